// setup.hip -- device-side construction of the block sparsity and scatter maps.
// Replaces DoFTools::make_sparsity_pattern + distribute_sparsity_pattern + PETSc matrix preallocation
// (mpi_fluid_solver.cpp:311-322).  One-off per mesh; uses rocPRIM (sort/scan) as plumbing.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

__global__ void k_gen_keys(int64_t n_cells, int R, int C, const int32_t *__restrict__ rows,
                           const int32_t *__restrict__ cols, int64_t n_rows_owned, uint64_t *__restrict__ keys) {
  const int64_t total = n_cells * R * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = t / (R * C);
    const int rc = int(t - cell * R * C);
    const int r = rc / C, c = rc - r * C;
    const int32_t row = rows[cell * R + r], col = cols[cell * C + c];
    keys[t] = (row < n_rows_owned) ? ((uint64_t(uint32_t(row)) << 32) | uint32_t(col)) : ~uint64_t(0);
  }
}

__global__ void k_flag_unique(int64_t n, const uint64_t *__restrict__ keys, int64_t *__restrict__ flags) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = keys[t];
    flags[t] = (k != ~uint64_t(0) && (t == 0 || keys[t - 1] != k)) ? 1 : 0;
  }
}

__global__ void k_emit(int64_t n, const uint64_t *__restrict__ keys, const int64_t *__restrict__ pos,
                       int32_t *__restrict__ col, int64_t *__restrict__ rowcnt) {
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
    const uint64_t k = keys[t];
    if (k != ~uint64_t(0) && (t == 0 || keys[t - 1] != k)) {
      col[pos[t]] = int32_t(uint32_t(k));
      atomicAdd((unsigned long long *)&rowcnt[k >> 32], 1ull);
    }
  }
}

__global__ void k_pos_map(int64_t n_cells, int R, int C, const int32_t *__restrict__ rows,
                          const int32_t *__restrict__ cols, int64_t n_rows_owned, const int64_t *__restrict__ rowptr,
                          const int32_t *__restrict__ col, uint16_t *__restrict__ pos) {
  const int64_t total = n_cells * R * C;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = t / (R * C);
    const int rc = int(t - cell * R * C);
    const int r = rc / C, c = rc - r * C;
    const int32_t row = rows[cell * R + r], cc = cols[cell * C + c];
    uint16_t out = 0xFFFF;
    if (row < n_rows_owned) {
      int64_t lo = rowptr[row], hi = rowptr[row + 1] - 1;
      const int64_t base = lo;
      while (lo <= hi) {
        const int64_t mid = (lo + hi) >> 1;
        const int32_t v = col[mid];
        if (v == cc) { out = uint16_t(mid - base); break; }
        if (v < cc) lo = mid + 1; else hi = mid - 1;
      }
    }
    pos[t] = out;
  }
}

static int grid_for(int64_t n) {
  int64_t g = (n + 255) / 256;
  return int(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

// Sorts / uniques N (row << 32 | col) keys (invalid = ~0) into the CSR pattern of M; k0 is consumed.
static void pattern_from_keys(ifem_ctx *ctx, PlanarCsr &M, int bs, int64_t n_rows_owned, DBuf<uint64_t> &k0, int64_t N) {
  hipStream_t s = ctx->stream;
  DBuf<uint64_t> k1;
  k1.alloc(N, "the (row, column) keys of the sparsity pattern");
  size_t tmp_bytes = 0;
  IFEM_HIP_CHECK(rocprim::radix_sort_keys(nullptr, tmp_bytes, k0.p, k1.p, (size_t)N, 0, 64, s));
  DBuf<char> tmp;
  tmp.alloc(tmp_bytes + 16);
  IFEM_HIP_CHECK(rocprim::radix_sort_keys(tmp.p, tmp_bytes, k0.p, k1.p, (size_t)N, 0, 64, s));
  int64_t *flags = reinterpret_cast<int64_t *>(k0.p); // k0 is dead after the sort
  hipLaunchKernelGGL(k_flag_unique, dim3(grid_for(N)), dim3(256), 0, s, N, k1.p, flags);
  DBuf<int64_t> posbuf;
  posbuf.alloc(N + 1);
  size_t tmp2 = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp2, flags, posbuf.p, int64_t(0), (size_t)N,
                                         rocprim::plus<int64_t>(), s));
  DBuf<char> tmpb;
  tmpb.alloc(tmp2 + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmpb.p, tmp2, flags, posbuf.p, int64_t(0), (size_t)N,
                                         rocprim::plus<int64_t>(), s));
  int64_t last_pos = 0, last_flag = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&last_pos, posbuf.p + (N - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipMemcpyAsync(&last_flag, flags + (N - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  const int64_t nnzb = last_pos + last_flag;
  M.n_rows = n_rows_owned;
  M.nnzb = nnzb;
  M.bs = bs;
  M.col.alloc(nnzb, "the column indices of a block of the sparsity pattern");
  DBuf<int64_t> rowcnt;
  rowcnt.alloc(n_rows_owned + 1);
  IFEM_HIP_CHECK(hipMemsetAsync(rowcnt.p, 0, (n_rows_owned + 1) * 8, s));
  hipLaunchKernelGGL(k_emit, dim3(grid_for(N)), dim3(256), 0, s, N, k1.p, posbuf.p, M.col.p, rowcnt.p);
  M.rowptr.alloc(n_rows_owned + 1);
  size_t tmp3 = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tmp3, rowcnt.p, M.rowptr.p, int64_t(0), (size_t)(n_rows_owned + 1),
                                         rocprim::plus<int64_t>(), s));
  DBuf<char> tmpc;
  tmpc.alloc(tmp3 + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmpc.p, tmp3, rowcnt.p, M.rowptr.p, int64_t(0), (size_t)(n_rows_owned + 1),
                                         rocprim::plus<int64_t>(), s));
  // longest row must fit the 16-bit scatter map
  {
    std::vector<int64_t> rc = rowcnt.download(s);
    int64_t mx = 0;
    for (int64_t i = 0; i < n_rows_owned; ++i) mx = rc[i] > mx ? rc[i] : mx;
    if (mx >= 0xFFFF) throw Error(IFEM_E_BADPARAM, "row longer than 65534 blocks");
    M.max_row = (int)mx;
  }
  if (bs > 0 && &M != &ctx->Auu) { // bs == 0: pattern only (incidence lists); A_uu values: see ensure_auu_values
    M.val.alloc((size_t)nnzb * bs, "the values of a block of system_matrix (B, B^T or M_p)");
    IFEM_HIP_CHECK(hipMemsetAsync(M.val.p, 0, (size_t)nnzb * bs * sizeof(double), s));
  }
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
}

// ---- scatter order of the A_uu rows (3D Q2/Q1 cell kernel).  The blocks of a row need not be stored in column order: nothing
// but the scatter map and the column array know where a block lives.  The cell kernel's atomics cost one memory-side request
// per 64-byte segment they touch, and a cell's 27 blocks of a row are contiguous only as far as no block of
// another cell lies between them.  Ordering the blocks of a row by (last cell, first cell, column) of the cells
// that touch them puts the blocks exclusive to a cell next to those it shares with its neighbours: a cell's part of a row becomes one to
// four runs instead of fourteen (Morton numbering) -- 862 instead of 957 segments per cell before packing losses
// (tools/scatter_sim.py).  Applied once, before the values exist; posUU is mapped through the permutation.  (Round 3 keyed on the
// MFMA column tile too: the two-waves-per-cell kernel scattered the two tiles of a row at different times; round 4's sends a row whole.)
__global__ void k_block_cells(int64_t n_cells, int NU, const int32_t *__restrict__ cu, int64_t n_rows_owned, const int64_t *__restrict__ rowptr,
                              const uint16_t *__restrict__ pos, int32_t *__restrict__ cmin, int32_t *__restrict__ cmax) {
  const int64_t total = n_cells * NU * NU;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = t / (NU * NU);
    const int a = int((t - cell * NU * NU) / NU);
    const int32_t row = cu[cell * NU + a];
    if (row >= n_rows_owned) continue;
    const int64_t e = rowptr[row] + pos[t];
    atomicMin(&cmin[e], int32_t(cell));
    atomicMax(&cmax[e], int32_t(cell));
  }
}
constexpr int kReorderMaxRow = 512;
// one wavefront per row: rank of every block by (cmax, cmin, col); newpos[old entry] = rank, col_out in the new order
__global__ __launch_bounds__(256) void k_row_reorder(int64_t n_rows, const int64_t *__restrict__ rowptr, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ cmin, const int32_t *__restrict__ cmax,
                                                     int32_t *__restrict__ col_out, uint16_t *__restrict__ newpos) {
  __shared__ uint64_t k1[4][kReorderMaxRow];
  __shared__ int32_t k2[4][kReorderMaxRow];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = int64_t(blockIdx.x) * 4 + wave;
  const bool active = row < n_rows;
  const int64_t rs = active ? rowptr[row] : 0;
  const int len = active ? int(rowptr[row + 1] - rs) : 0;
  const bool sortable = len <= kReorderMaxRow; // a longer row keeps its column order
  if (!sortable)
    for (int i = lane; i < len; i += 64) { col_out[rs + i] = col[rs + i]; newpos[rs + i] = uint16_t(i); }
  if (sortable)
    for (int i = lane; i < len; i += 64) {
      k1[wave][i] = (uint64_t(uint32_t(cmax[rs + i])) << 32) | uint32_t(cmin[rs + i]);
      k2[wave][i] = col[rs + i];
    }
  __syncthreads();
  if (sortable)
    for (int i = lane; i < len; i += 64) {
      const uint64_t a1 = k1[wave][i];
      const int32_t a2 = k2[wave][i];
      int rank = 0;
      for (int j = 0; j < len; ++j) {
        const uint64_t b1 = k1[wave][j];
        rank += (b1 < a1 || (b1 == a1 && k2[wave][j] < a2)) ? 1 : 0;
      }
      newpos[rs + i] = uint16_t(rank);
      col_out[rs + rank] = a2;
    }
}
__global__ void k_pos_remap(int64_t n_cells, int NU, const int32_t *__restrict__ cu, int64_t n_rows_owned, const int64_t *__restrict__ rowptr,
                            const uint16_t *__restrict__ newpos, uint16_t *__restrict__ pos) {
  const int64_t total = n_cells * NU * NU;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = t / (NU * NU);
    const int a = int((t - cell * NU * NU) / NU);
    const int32_t row = cu[cell * NU + a];
    if (row < n_rows_owned) pos[t] = newpos[rowptr[row] + pos[t]];
  }
}
static void reorder_uu_rows(ifem_ctx *ctx) {
  PlanarCsr &M = ctx->Auu;
  hipStream_t s = ctx->stream;
  const int64_t nnzb = M.nnzb, nc = ctx->n_cells;
  if (!nnzb || !nc || M.n_rows == 0) return;
  DBuf<int32_t> cmin, cmax, col2;
  DBuf<uint16_t> newpos;
  cmin.alloc((size_t)nnzb); cmax.alloc((size_t)nnzb); col2.alloc((size_t)nnzb); newpos.alloc((size_t)nnzb);
  IFEM_HIP_CHECK(hipMemsetAsync(cmin.p, 0x7f, (size_t)nnzb * sizeof(int32_t), s));
  IFEM_HIP_CHECK(hipMemsetAsync(cmax.p, 0, (size_t)nnzb * sizeof(int32_t), s));
  const int64_t N = nc * ctx->nu * ctx->nu;
  hipLaunchKernelGGL(k_block_cells, dim3(grid_for(N)), dim3(256), 0, s, nc, ctx->nu, ctx->cell_unodes.p, M.n_rows, M.rowptr.p, ctx->posUU.p, cmin.p, cmax.p);
  hipLaunchKernelGGL(k_row_reorder, dim3(unsigned((M.n_rows + 3) / 4)), dim3(256), 0, s, M.n_rows, M.rowptr.p, M.col.p, cmin.p, cmax.p, col2.p, newpos.p);
  hipLaunchKernelGGL(k_pos_remap, dim3(grid_for(N)), dim3(256), 0, s, nc, ctx->nu, ctx->cell_unodes.p, M.n_rows, M.rowptr.p, newpos.p, ctx->posUU.p);
  IFEM_HIP_CHECK(hipMemcpyAsync(M.col.p, col2.p, (size_t)nnzb * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_HIP_CHECK(hipGetLastError());
  M.n_interior = -1; // a row list built from the old order stays valid (it lists rows), but keep the invariant simple
  ctx->scat3.release(); // records of the old order
  ctx->hdr3_rows = false;
}

// ---- per-cell records of the 3D Q2/Q1 cell kernel (assemble3.hip; layouts: ctx.hpp kAsm3Rec / kAsm3Hdr).  Everything the
// kernel's scatter would otherwise work out per cell and per assembly from the node ids and the scatter map: the order of the
// cell's nodes by id, the rank of every block among the 27 the cell adds to a row (by position in the row) and the alignment of
// the row's first block -- functions of the mesh and of the pattern alone.
__global__ void k_scat3_hdr(int64_t n_cells, const int32_t *__restrict__ cu, const int32_t *__restrict__ cp, int64_t n_rows_owned,
                            const int64_t *__restrict__ rowptr, const uint16_t *__restrict__ pos, int with_rows, uint8_t *__restrict__ hdr) {
  const int64_t cell = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (cell >= n_cells) return;
  uint8_t *h = hdr + cell * kAsm3Hdr;
  const int32_t *nd = cu + cell * 27, *pd = cp + cell * 8;
  for (int k = 27; k < 32; ++k) { h[k] = uint8_t(k); h[32 + k] = uint8_t(k); }
  for (int a = 0; a < 27; ++a) {
    int rank = 0;
    for (int j = 0; j < 27; ++j) rank += nd[j] < nd[a] ? 1 : 0;
    h[rank] = uint8_t(a);      // perm
    h[32 + a] = uint8_t(rank); // iperm
  }
  for (int b = 0; b < 8; ++b) {
    int rank = 0;
    for (int j = 0; j < 8; ++j) rank += pd[j] < pd[b] ? 1 : 0;
    h[64 + rank] = uint8_t(b); // permp
  }
  for (int a = 0; a < 32; ++a) {
    uint8_t s = 0;
    if (with_rows && a < 27 && nd[a] < n_rows_owned) {
      int p0 = 0xFFFF;
      for (int b = 0; b < 27; ++b) { const int p = pos[(cell * 27 + a) * 27 + b]; p0 = p < p0 ? p : p0; }
      s = uint8_t((rowptr[nd[a]] + p0) & 7); // blocks are 9 doubles: (9 k) mod 8 = k mod 8
    }
    h[72 + a] = s;
  }
  for (int k = 104; k < kAsm3Hdr; ++k) h[k] = 0;
}
__global__ void k_scat3_rec(int64_t n_cells, const int32_t *__restrict__ cu, int64_t n_rows_owned, const uint16_t *__restrict__ pos,
                            const uint8_t *__restrict__ hdr, uint16_t *__restrict__ rec) {
  const int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (t >= n_cells * 128) return;
  const int64_t cell = t >> 7;
  const int ti = int(t >> 6) & 1, lane = int(t) & 63, l15 = lane & 15, g = lane >> 4;
  const uint8_t *perm = hdr + cell * kAsm3Hdr;
  uint16_t *out = rec + cell * kAsm3Rec + (ti * 64 + lane) * 8;
  for (int tj = 0; tj < 2; ++tj)
    for (int r = 0; r < 4; ++r) {
      const int a = 16 * ti + g + 4 * r, col = 16 * tj + l15;
      uint16_t v = 0xFFFF;
      if (a < 27 && col < 27 && cu[cell * 27 + a] < n_rows_owned) {
        const uint16_t *row = pos + (cell * 27 + a) * 27;
        const int p = row[perm[col]];
        int rank = 0;
        for (int j = 0; j < 27; ++j) rank += row[j] < p ? 1 : 0;
        v = uint16_t((p - rank) | (rank << 9)); // rank <= position: the blocks before it in the row include its predecessors of this cell
      }
      out[tj * 4 + r] = v;
    }
}
// headers always; records (2 KB per cell) and row alignments only for a context that assembles A_uu.  false: the rows are too
// long for the records (position < 512) -- the caller falls back to the general kernel of assemble2.hip
bool ensure_scat3(ifem_ctx *ctx, bool with_rows) {
  if (ctx->dim != 3 || ctx->kv != 2 || ctx->n_cells == 0) return false;
  if (with_rows && ctx->Auu.max_row >= 512) return false;
  hipStream_t s = ctx->stream;
  const int64_t nc = ctx->n_cells;
  const bool need_hdr = ctx->hdr3.n == 0 || (with_rows && !ctx->hdr3_rows);
  if (need_hdr) {
    if (ctx->hdr3.n == 0) ctx->hdr3.alloc(size_t(nc) * kAsm3Hdr);
    hipLaunchKernelGGL(k_scat3_hdr, dim3(unsigned((nc + 127) / 128)), dim3(128), 0, s, nc, ctx->cell_unodes.p, ctx->cell_pnodes.p, ctx->Auu.n_rows,
                       ctx->Auu.rowptr.p, ctx->posUU.p, with_rows ? 1 : 0, ctx->hdr3.p);
    ctx->hdr3_rows = with_rows;
  }
  if (with_rows && ctx->scat3.n == 0) {
    ctx->scat3.alloc(size_t(nc) * kAsm3Rec);
    hipLaunchKernelGGL(k_scat3_rec, dim3(unsigned((nc * 128 + 255) / 256)), dim3(256), 0, s, nc, ctx->cell_unodes.p, ctx->Auu.n_rows, ctx->posUU.p,
                       ctx->hdr3.p, ctx->scat3.p);
  }
  IFEM_HIP_CHECK(hipGetLastError());
  return true;
}
// position of the diagonal block in every owned row (the rows are not necessarily in column order)
__global__ void k_diag_pos(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, int32_t *__restrict__ dp) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  int32_t p = -1;
  for (int64_t k = rp[row]; k < rp[row + 1]; ++k)
    if (col[k] == row) { p = int32_t(k - rp[row]); break; }
  dp[row] = p;
}

// The values of A_uu (78 GB at 128^3 Q2) are allocated by the first assembly that writes them: coarse multigrid levels
// keep the pattern but never assemble the velocity block.
void ensure_auu_values(ifem_ctx *ctx) {
  PlanarCsr &M = ctx->Auu;
  const size_t n = (size_t)M.nnzb * M.bs;
  if (M.val.n == n) return;
  if (ctx->tune.uu_row_order && ctx->dim == 3 && ctx->kv == 2 && M.val.n == 0) reorder_uu_rows(ctx);
  ctx->uu_diag_pos.alloc((size_t)M.n_rows + 1);
  if (M.n_rows) hipLaunchKernelGGL(k_diag_pos, dim3(unsigned((M.n_rows + 255) / 256)), dim3(256), 0, ctx->stream, M.n_rows, M.rowptr.p, M.col.p, ctx->uu_diag_pos.p);
  M.val.alloc(n, "the values of A_uu (system_matrix.block(0, 0))");
  IFEM_HIP_CHECK(hipMemsetAsync(M.val.p, 0, n * sizeof(double), ctx->stream));
}

// Builds the sorted pattern of {(row, col)} pairs coupled through a common cell, plus the scatter map.
void build_pattern(ifem_ctx *ctx, PlanarCsr &M, int bs, int64_t n_rows_owned, int R, const int32_t *d_rows, int C,
                   const int32_t *d_cols, DBuf<uint16_t> &pos) {
  hipStream_t s = ctx->stream;
  const int64_t N = ctx->n_cells * R * C;
  DBuf<uint64_t> k0;
  k0.alloc(N);
  hipLaunchKernelGGL(k_gen_keys, dim3(grid_for(N)), dim3(256), 0, s, ctx->n_cells, R, C, d_rows, d_cols, n_rows_owned,
                     k0.p);
  pattern_from_keys(ctx, M, bs, n_rows_owned, k0, N);
  pos.alloc(N);
  hipLaunchKernelGGL(k_pos_map, dim3(grid_for(N)), dim3(256), 0, s, ctx->n_cells, R, C, d_rows, d_cols, n_rows_owned,
                     M.rowptr.p, M.col.p, pos.p);
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_HIP_CHECK(hipGetLastError());
}


// ---- interior / boundary rows of a matrix on several ranks: a row is "boundary" when it reads a ghost column (column id
// >= n_owned_cols) of its own pattern or of the optional second pattern M2 (A_uu rows also carry the B^T block); the
// list keeps the row order inside both groups.
__global__ void k_row_flag(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, int32_t n_owned_cols,
                           const int64_t *__restrict__ rp2, const int32_t *__restrict__ col2, int32_t n_owned_cols2,
                           int64_t *__restrict__ flag) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    bool b = false;
    for (int64_t k = rp[r]; k < rp[r + 1] && !b; ++k) b = col[k] >= n_owned_cols;
    if (rp2)
      for (int64_t k = rp2[r]; k < rp2[r + 1] && !b; ++k) b = col2[k] >= n_owned_cols2;
    flag[r] = b ? 1 : 0;
  }
}
__global__ void k_row_split(int64_t n_rows, const int64_t *__restrict__ flag, const int64_t *__restrict__ before, int64_t n_interior,
                            int32_t *__restrict__ rows) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x) {
    if (flag[r]) rows[n_interior + before[r]] = int32_t(r);
    else rows[r - before[r]] = int32_t(r);
  }
}
void build_row_split(ifem_ctx *ctx, PlanarCsr &M, int64_t n_owned_cols, const PlanarCsr *M2, int64_t n_owned_cols2) {
  if (M.n_interior >= 0) return;
  hipStream_t s = ctx->stream;
  const int64_t n = M.n_rows;
  M.n_interior = 0; M.n_boundary = 0;
  if (n == 0) return;
  if (M2 && M2->n_rows != n) throw Error(IFEM_E_BADPARAM, "row split: the two patterns differ in their row count");
  DBuf<int64_t> flag, before;
  flag.alloc(n); before.alloc(n);
  hipLaunchKernelGGL(k_row_flag, dim3(grid_for(n)), dim3(256), 0, s, n, M.rowptr.p, M.col.p, int32_t(n_owned_cols),
                     M2 ? M2->rowptr.p : nullptr, M2 ? M2->col.p : nullptr, int32_t(n_owned_cols2), flag.p);
  size_t tb = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, flag.p, before.p, int64_t(0), (size_t)n, rocprim::plus<int64_t>(), s));
  DBuf<char> tmp;
  tmp.alloc(tb + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmp.p, tb, flag.p, before.p, int64_t(0), (size_t)n, rocprim::plus<int64_t>(), s));
  int64_t last_before = 0, last_flag = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&last_before, before.p + (n - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipMemcpyAsync(&last_flag, flag.p + (n - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  M.n_boundary = last_before + last_flag;
  M.n_interior = n - M.n_boundary;
  M.split_rows.alloc(n);
  hipLaunchKernelGGL(k_row_split, dim3(grid_for(n)), dim3(256), 0, s, n, flag.p, before.p, M.n_interior, M.split_rows.p);
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_HIP_CHECK(hipGetLastError());
}

// ---- list of the rows i with flag[i] != 0 (ascending); returns their number
__global__ void k_row_pick(int64_t n_rows, const int64_t *__restrict__ flag, const int64_t *__restrict__ before, int32_t *__restrict__ rows) {
  for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
    if (flag[r]) rows[before[r]] = int32_t(r);
}
int64_t compact_flagged_rows(ifem_ctx *ctx, const int64_t *flag, int64_t n, DBuf<int32_t> &rows) {
  if (n == 0) return 0;
  hipStream_t s = ctx->stream;
  DBuf<int64_t> before;
  before.alloc(n);
  size_t tb = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, flag, before.p, int64_t(0), (size_t)n, rocprim::plus<int64_t>(), s));
  DBuf<char> tmp;
  tmp.alloc(tb + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmp.p, tb, flag, before.p, int64_t(0), (size_t)n, rocprim::plus<int64_t>(), s));
  int64_t lb = 0, lf = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&lb, before.p + (n - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipMemcpyAsync(&lf, flag + (n - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  const int64_t cnt = lb + lf;
  if ((int64_t)rows.n < cnt) rows.alloc((size_t)cnt);
  if (cnt) hipLaunchKernelGGL(k_row_pick, dim3(grid_for(n)), dim3(256), 0, s, n, flag, before.p, rows.p);
  IFEM_HIP_CHECK(hipStreamSynchronize(s)); // `before` leaves scope
  return cnt;
}

// ---- mass_schur(1,1) pattern (compute_mmult_pattern(B, B^T), mpi_fluid_solver.cpp:326-329).  For the Q1 pressure space
// pattern(B B^T) = pattern(M_p^2): p-nodes i, j couple iff cells c1 with i and c2 with j share a vertex.
// OWNED: only the owned x owned block through owned intermediate nodes (several ranks: rows of ghost nodes are not held here)
template <bool OWNED>
__global__ void k_sq_count(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                           int64_t *__restrict__ cnt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t c = 0;
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int32_t m = col[k];
      if (!OWNED) c += rp[m + 1] - rp[m];
      else if (m < n_rows)
        for (int64_t l = rp[m]; l < rp[m + 1]; ++l) c += col[l] < n_rows ? 1 : 0;
    }
    cnt[i] = c;
  }
}
template <bool OWNED>
__global__ void k_sq_fill(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                          const int64_t *__restrict__ off, uint64_t *__restrict__ keys) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t o = off[i];
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k) {
      const int32_t m = col[k];
      if (OWNED && m >= n_rows) continue;
      for (int64_t l = rp[m]; l < rp[m + 1]; ++l)
        if (!OWNED || col[l] < n_rows) keys[o++] = (uint64_t(uint32_t(i)) << 32) | uint32_t(col[l]);
    }
  }
}

// node -> (cell, local index) incidence lists for the row-owner assembly: CSR with entries (cell << 5 | a)
__global__ void k_gen_inc_keys(int64_t n_cells, int R, const int32_t *__restrict__ rows, int64_t n_rows_owned,
                               uint64_t *__restrict__ keys) {
  const int64_t total = n_cells * R;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t cell = t / R;
    const int a = int(t - cell * R);
    const int32_t row = rows[t];
    keys[t] = (row < n_rows_owned) ? ((uint64_t(uint32_t(row)) << 32) | uint32_t((cell << 5) | a)) : ~uint64_t(0);
  }
}

void build_incidence(ifem_ctx *ctx) {
  if (ctx->n_cells >= (int64_t(1) << 26)) throw Error(IFEM_E_BADPARAM, "incidence lists: too many local cells");
  hipStream_t s = ctx->stream;
  auto one = [&](PlanarCsr &M, int64_t n_rows, int R, const int32_t *rows) {
    const int64_t N = ctx->n_cells * R;
    DBuf<uint64_t> keys;
    keys.alloc(N);
    hipLaunchKernelGGL(k_gen_inc_keys, dim3(grid_for(N)), dim3(256), 0, s, ctx->n_cells, R, rows, n_rows, keys.p);
    pattern_from_keys(ctx, M, 0, n_rows, keys, N);
  };
  one(ctx->uinc, ctx->nUo, ctx->nu, ctx->mf_n_interior >= 0 ? ctx->mf_cell_unodes.p : ctx->cell_unodes.p);
}

// ---- cell tables of the matrix-free apply on several ranks: interior cells (all nodes owned) first, then the cells that
// touch a ghost node; order kept inside both groups (the Morton locality survives)
__global__ void k_cell_flag(int64_t n_cells, int nu, const int32_t *__restrict__ cell_unodes, int32_t n_owned, int64_t *__restrict__ flag) {
  for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_cells; c += (int64_t)gridDim.x * blockDim.x) {
    bool b = false;
    for (int a = 0; a < nu; ++a) b = b || cell_unodes[c * nu + a] >= n_owned;
    flag[c] = b ? 1 : 0;
  }
}
__global__ void k_cell_permute(int64_t n_cells, int nu, int nvd, const int64_t *__restrict__ flag, const int64_t *__restrict__ before,
                               int64_t n_interior, const int32_t *__restrict__ cu, const double *__restrict__ vc,
                               int32_t *__restrict__ cu_out, double *__restrict__ vc_out) {
  const int per = nu + nvd;
  for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n_cells * per; t += (int64_t)gridDim.x * blockDim.x) {
    const int64_t c = t / per;
    const int k = int(t - c * per);
    const int64_t dst = flag[c] ? n_interior + before[c] : c - before[c];
    if (k < nu) cu_out[dst * nu + k] = cu[c * nu + k];
    else vc_out[dst * nvd + (k - nu)] = vc[c * nvd + (k - nu)];
  }
}
void build_mf_cell_split(ifem_ctx *ctx) {
  if (ctx->mf_n_interior >= 0) return;
  hipStream_t s = ctx->stream;
  const int64_t n = ctx->n_cells;
  if (n == 0) { ctx->mf_n_interior = 0; return; }
  const int nu = ctx->nu, nvd = ctx->np * ctx->dim; // np = 2^dim vertices
  DBuf<int64_t> flag, before;
  flag.alloc(n); before.alloc(n);
  hipLaunchKernelGGL(k_cell_flag, dim3(grid_for(n)), dim3(256), 0, s, n, nu, ctx->cell_unodes.p, int32_t(ctx->nUo), flag.p);
  size_t tb = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, flag.p, before.p, int64_t(0), (size_t)n, rocprim::plus<int64_t>(), s));
  DBuf<char> tmp;
  tmp.alloc(tb + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmp.p, tb, flag.p, before.p, int64_t(0), (size_t)n, rocprim::plus<int64_t>(), s));
  int64_t lb = 0, lf = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&lb, before.p + (n - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipMemcpyAsync(&lf, flag.p + (n - 1), 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  const int64_t n_int = n - (lb + lf);
  ctx->mf_cell_unodes.alloc((size_t)n * nu);
  ctx->mf_vcoords.alloc((size_t)n * nvd);
  hipLaunchKernelGGL(k_cell_permute, dim3(grid_for(n * (nu + nvd))), dim3(256), 0, s, n, nu, nvd, flag.p, before.p, n_int,
                     ctx->cell_unodes.p, ctx->vcoords.p, ctx->mf_cell_unodes.p, ctx->mf_vcoords.p);
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_HIP_CHECK(hipGetLastError());
  ctx->mf_n_interior = n_int;
  ctx->uinc.n_rows = 0; // incidence lists of the natural numbering (if any) are stale
  ctx->uinc.n_interior = -1;
}

// ---- distributed explicit S_m on a structured pressure lattice (box meshes on several ranks).  Row i couples the
// lattice nodes within +-2 of it (clipped at the domain boundary) = pattern(M_p^2); columns are ids of the 2-deep column
// space (owned | far nodes by owner).  Values come from probing the distributed matrix-free operator B diag(M_u)^-1 B^T
// with the 5^dim vectors "nodes whose lattice coordinates are congruent to c mod 5": two columns of one colour never
// meet in a row, so (S_m e_c)_i is exactly the entry S_m[i, j_c(i)].
struct SBox {
  int64_t N[3], lo[3], n[3];
};
__device__ inline void sbox_decode(const SBox &B, int64_t gid, int64_t *g) {
  g[0] = gid % B.N[0];
  g[1] = (gid / B.N[0]) % B.N[1];
  g[2] = gid / (B.N[0] * B.N[1]);
}
__global__ void k_sbox_count(int64_t n_rows, SBox B, const int64_t *__restrict__ gid, int64_t *__restrict__ cnt) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t g[3], c = 1;
    sbox_decode(B, gid[i], g);
    for (int d = 0; d < 3; ++d) {
      const int64_t lo = g[d] - 2 > 0 ? g[d] - 2 : 0, hi = g[d] + 2 < B.N[d] - 1 ? g[d] + 2 : B.N[d] - 1;
      c *= hi - lo + 1;
    }
    cnt[i] = c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt[n_rows] = 0;
}
__global__ void k_sbox_fill(int64_t n_rows, SBox B, const int64_t *__restrict__ gid, const int32_t *__restrict__ box_id,
                            const int64_t *__restrict__ rp, int32_t *__restrict__ col) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t g[3], lo[3], hi[3];
    sbox_decode(B, gid[i], g);
    for (int d = 0; d < 3; ++d) {
      lo[d] = g[d] - 2 > 0 ? g[d] - 2 : 0;
      hi[d] = g[d] + 2 < B.N[d] - 1 ? g[d] + 2 : B.N[d] - 1;
    }
    int64_t o = rp[i];
    for (int64_t z = lo[2]; z <= hi[2]; ++z)
      for (int64_t y = lo[1]; y <= hi[1]; ++y)
        for (int64_t x = lo[0]; x <= hi[0]; ++x)
          col[o++] = box_id[((z - B.lo[2]) * B.n[1] + (y - B.lo[1])) * B.n[0] + (x - B.lo[0])];
  }
}
__global__ void k_probe_vector(int64_t n_rows, SBox B, const int64_t *__restrict__ gid, int c0, int c1, int c2,
                               double *__restrict__ x) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t g[3];
    sbox_decode(B, gid[i], g);
    x[i] = (g[0] % 5 == c0 && g[1] % 5 == c1 && g[2] % 5 == c2) ? 1.0 : 0.0;
  }
}
__global__ void k_probe_fill(int64_t n_rows, SBox B, const int64_t *__restrict__ gid, int c0, int c1, int c2,
                             const int64_t *__restrict__ rp, const double *__restrict__ y, double *__restrict__ val) {
  const int c[3] = {c0, c1, c2};
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_rows; i += (int64_t)gridDim.x * blockDim.x) {
    int64_t g[3], lo[3], w[3], t[3];
    sbox_decode(B, gid[i], g);
    bool ok = true;
    for (int d = 0; d < 3; ++d) {
      lo[d] = g[d] - 2 > 0 ? g[d] - 2 : 0;
      const int64_t hi = g[d] + 2 < B.N[d] - 1 ? g[d] + 2 : B.N[d] - 1;
      w[d] = hi - lo[d] + 1;
      const int64_t a = g[d] - 2; // the unique node of [a, a+4] congruent to c[d] mod 5
      t[d] = a + (((c[d] - a) % 5) + 5) % 5;
      if (t[d] < lo[d] || t[d] > hi) ok = false;
    }
    if (ok) val[rp[i] + ((t[2] - lo[2]) * w[1] + (t[1] - lo[1])) * w[0] + (t[0] - lo[0])] = y[i];
  }
}
static SBox sbox_of(const ifem_ctx *ctx) {
  SBox B;
  for (int d = 0; d < 3; ++d) { B.N[d] = ctx->halo.p_lattice_n[d]; B.lo[d] = ctx->halo.sm_box_lo[d]; B.n[d] = ctx->halo.sm_box_n[d]; }
  return B;
}
void build_schur_pattern_box(ifem_ctx *ctx) {
  if (!ctx->halo.has_s) throw Error(IFEM_E_BADPARAM, "explicit S_m on several ranks needs the 2-deep pressure halo plan");
  hipStream_t s = ctx->stream;
  const int64_t n = ctx->nPo;
  const SBox B = sbox_of(ctx);
  PlanarCsr &M = ctx->Sm;
  DBuf<int64_t> cnt;
  cnt.alloc(n + 1);
  hipLaunchKernelGGL(k_sbox_count, dim3(grid_for(n > 0 ? n : 1)), dim3(256), 0, s, n, B, ctx->halo.own_p_gid.p, cnt.p);
  M.rowptr.alloc(n + 1);
  size_t tb = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, cnt.p, M.rowptr.p, int64_t(0), (size_t)(n + 1), rocprim::plus<int64_t>(), s));
  DBuf<char> tmp;
  tmp.alloc(tb + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmp.p, tb, cnt.p, M.rowptr.p, int64_t(0), (size_t)(n + 1), rocprim::plus<int64_t>(), s));
  int64_t nnz = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&nnz, M.rowptr.p + n, 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  M.n_rows = n; M.nnzb = nnz; M.bs = 1; M.max_row = ctx->dim == 3 ? 125 : 25;
  M.col.alloc(nnz);
  M.val.alloc(nnz);
  IFEM_HIP_CHECK(hipMemsetAsync(M.val.p, 0, nnz * sizeof(double), s));
  if (n) hipLaunchKernelGGL(k_sbox_fill, dim3(grid_for(n)), dim3(256), 0, s, n, B, ctx->halo.own_p_gid.p, ctx->halo.sm_box_id.p, M.rowptr.p, M.col.p);
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_HIP_CHECK(hipGetLastError());
}
void schur_probe_vector(ifem_ctx *ctx, int color, double *x) {
  const int64_t n = ctx->nPo;
  if (n) hipLaunchKernelGGL(k_probe_vector, dim3(grid_for(n)), dim3(256), 0, ctx->stream, n, sbox_of(ctx), ctx->halo.own_p_gid.p,
                            color % 5, (color / 5) % 5, color / 25, x);
}
void schur_probe_fill(ifem_ctx *ctx, int color, const double *y) {
  const int64_t n = ctx->nPo;
  if (n) hipLaunchKernelGGL(k_probe_fill, dim3(grid_for(n)), dim3(256), 0, ctx->stream, n, sbox_of(ctx), ctx->halo.own_p_gid.p,
                            color % 5, (color / 5) % 5, color / 25, ctx->Sm.rowptr.p, y, ctx->Sm.val.p);
}

template <bool OWNED>
static void schur_pattern_into(ifem_ctx *ctx, PlanarCsr &M) {
  hipStream_t s = ctx->stream;
  const int64_t n = ctx->nPo;
  DBuf<int64_t> cnt, off;
  cnt.alloc(n + 1);
  off.alloc(n + 1);
  IFEM_HIP_CHECK(hipMemsetAsync(cnt.p, 0, (n + 1) * 8, s));
  hipLaunchKernelGGL((k_sq_count<OWNED>), dim3(grid_for(n)), dim3(256), 0, s, n, ctx->Mp.rowptr.p, ctx->Mp.col.p, cnt.p);
  size_t tb = 0;
  IFEM_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, cnt.p, off.p, int64_t(0), (size_t)(n + 1), rocprim::plus<int64_t>(), s));
  DBuf<char> tmp;
  tmp.alloc(tb + 16);
  IFEM_HIP_CHECK(rocprim::exclusive_scan(tmp.p, tb, cnt.p, off.p, int64_t(0), (size_t)(n + 1), rocprim::plus<int64_t>(), s));
  int64_t N = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&N, off.p + n, 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  DBuf<uint64_t> keys;
  keys.alloc(N);
  hipLaunchKernelGGL((k_sq_fill<OWNED>), dim3(grid_for(n)), dim3(256), 0, s, n, ctx->Mp.rowptr.p, ctx->Mp.col.p, off.p, keys.p);
  pattern_from_keys(ctx, M, 1, n, keys, N);
  IFEM_HIP_CHECK(hipGetLastError());
}
void build_schur_pattern(ifem_ctx *ctx) {
  if (ctx->halo.nranks != 1) throw Error(IFEM_E_BADPARAM, "explicit S_m needs a 2-deep pressure halo: single rank only");
  schur_pattern_into<false>(ctx, ctx->Sm);
}
// several ranks: the owned x owned block of pattern(M_p^2), local pressure numbering (the per-rank ILU(0) of T_pp, tpp.hip)
void build_schur_pattern_owned(ifem_ctx *ctx) { schur_pattern_into<true>(ctx, ctx->TppPat); }

} // namespace ifem
