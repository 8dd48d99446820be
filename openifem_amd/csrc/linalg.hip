// linalg.hip -- bandwidth-bound kernels of the Krylov path: row-planar block SpMV, fused vector ops,
// reductions.  These replace PETSc MatMult / Vec ops under deal.II's SolverFGMRES / SolverCG
// (mpi_insim.cpp:75-82,103-108,383-388).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdlib>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

// ---------------------------------------------------------------------------------------------------
// Row-planar block SpMV.  G lanes cooperate on one row; lane k walks blocks k, k+G, ... of the row and
// reads the BS = BR*BC planes of its block with stride len (coalesced across lanes), gathers BC values of x
// and accumulates BR partial sums that are reduced over the G lanes with DPP-free shuffles.
#ifndef IFEM_SPMV_UNROLL
#define IFEM_SPMV_UNROLL 4
#endif
template <int BR, int BC, int G, bool ACC, class VT = double>
__device__ inline void row_planar_dot(const int64_t rs, const int len, const int32_t *__restrict__ col,
                                      const VT *__restrict__ val, const double *__restrict__ x, const int lig,
                                      double *acc) {
  const VT *vbase = val + rs * (BR * BC);
  // U entries per lane and trip with all index / value loads issued before the dependent gathers of x: the loop is
  // latency-bound otherwise (one 12-byte load pair in flight per lane).  Out-of-range slots re-read entry 0 with weight 0.
  constexpr int U = IFEM_SPMV_UNROLL;
  for (int k0 = lig; k0 < len; k0 += G * U) {
    int32_t c[U];
    VT v[U][BR * BC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * G;
      const bool ok = k < len;
      const int ks = ok ? k : 0;
      c[u] = col[rs + ks];
#pragma unroll
      for (int e = 0; e < BR * BC; ++e) { const VT t = vbase[int64_t(e) * len + ks]; v[u][e] = ok ? t : VT(0); }
    }
    double xv[U][BC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < BC; ++j) xv[u][j] = x[int64_t(c[u]) * BC + j];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int j = 0; j < BC; ++j) acc[r] += v[u][r * BC + j] * xv[u][j];
  }
}

// A_uu block row in the block-interleaved layout: lane k reads the BR*BC entries of its block back to back (the lanes of
// a group together cover one contiguous span of the row)
#ifndef IFEM_SPMV_UU_UNROLL
#define IFEM_SPMV_UU_UNROLL 1
#endif
template <int BR, int BC, int G, class VT>
__device__ inline void row_interleaved_dot(const int64_t rs, const int len, const int32_t *__restrict__ col,
                                           const VT *__restrict__ val, const double *__restrict__ x, const int lig,
                                           double *acc) {
  // U blocks per lane and trip as in row_planar_dot; 72-byte blocks already keep enough bytes in flight: U = 2 measured
  // no faster (k_spmv_uu 21.0 vs 19.8 ms at 128^3), so U = 1
  constexpr int U = IFEM_SPMV_UU_UNROLL;
  for (int k0 = lig; k0 < len; k0 += G * U) {
    int32_t c[U];
    VT bv[U][BR * BC];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * G;
      const bool ok = k < len;
      const int ks = ok ? k : 0;
      c[u] = col[rs + ks];
      const VT *b = val + (rs + ks) * (BR * BC);
#pragma unroll
      for (int e = 0; e < BR * BC; ++e) { const VT t = b[e]; bv[u][e] = ok ? t : VT(0); }
    }
    double xv[U][BC];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int j = 0; j < BC; ++j) xv[u][j] = x[int64_t(c[u]) * BC + j];
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int r = 0; r < BR; ++r)
#pragma unroll
        for (int j = 0; j < BC; ++j) acc[r] += double(bv[u][r * BC + j]) * xv[u][j];
  }
}

template <int G>
__device__ inline double group_sum(double v) {
#pragma unroll
  for (int off = G / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// y_u = A_uu x_u + B^T x_p   (rows: owned velocity nodes)
template <int DIM, int G, class VT = double>
__global__ __launch_bounds__(256) void k_spmv_uu(int64_t n_rows, const int64_t *__restrict__ rp_a,
                                                 const int32_t *__restrict__ col_a, const VT *__restrict__ val_a,
                                                 const int64_t *__restrict__ rp_t, const int32_t *__restrict__ col_t,
                                                 const double *__restrict__ val_t, const double *__restrict__ xu,
                                                 const double *__restrict__ xp, double *__restrict__ yu,
                                                 const int32_t *__restrict__ rows) {
  const int64_t ridx = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int lig = threadIdx.x & (G - 1);
  if (ridx >= n_rows) return; // whole groups exit together
  const int64_t row = rows ? int64_t(rows[ridx]) : ridx; // a row list: n_rows counts its entries
  double acc[DIM];
#pragma unroll
  for (int r = 0; r < DIM; ++r) acc[r] = 0;
  {
    const int64_t rs = rp_a[row];
    const int len = int(rp_a[row + 1] - rs);
#if IFEM_UU_INTERLEAVED
    row_interleaved_dot<DIM, DIM, G, VT>(rs, len, col_a, val_a, xu, lig, acc);
#else
    row_planar_dot<DIM, DIM, G, true, VT>(rs, len, col_a, val_a, xu, lig, acc);
#endif
  }
  if (xp) {
    const int64_t rs = rp_t[row];
    const int len = int(rp_t[row + 1] - rs);
    row_planar_dot<DIM, 1, G, true>(rs, len, col_t, val_t, xp, lig, acc);
  }
#pragma unroll
  for (int r = 0; r < DIM; ++r) acc[r] = group_sum<G>(acc[r]);
  if (lig == 0) {
#pragma unroll
    for (int r = 0; r < DIM; ++r) yu[row * DIM + r] = acc[r];
  }
}

// Software-pipelined y_u = A_uu x_u for the block-interleaved 3 x 3 layout.  The row-per-group kernel above lives for three
// dependent round trips (row pointer -> indices / values -> x) and moves 4.6 TB/s at the pins where a plain read stream
// reaches 6.2 (tools/readbw.hip).  Here a block keeps a contiguous range of rows (row pointers in LDS), every half-wave walks
// its rows as a flat sequence of "items" (32 consecutive blocks of a row) and, in each iteration, requests the column
// indices of item i + 2 and the values of item i + 1, gathers x for item i + 1 (its indices arrived an iteration ago) and
// only then multiplies item i: three items are in flight per lane, no load is waited for next to its issue.
struct SpmvItem { int64_t off; int row, n; bool last; }; // first block of the item, row (block-local), blocks in it, closes its row
__global__ __launch_bounds__(256) void k_spmv_uu_pipe(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                      const double *__restrict__ val, const double *__restrict__ xu,
                                                      double *__restrict__ yu, const int32_t *__restrict__ rows, int rows_per_block) {
  extern __shared__ int64_t s_rp[]; // [rows_per_block][2]: begin / end of every row of the block
  const int hw = threadIdx.x >> 5, lig = threadIdx.x & 31;
  const int64_t r0 = int64_t(xcd_swizzle(blockIdx.x, gridDim.x)) * rows_per_block;
  const int nr = int((r0 + rows_per_block <= n_rows ? r0 + rows_per_block : n_rows) - r0);
  if (nr <= 0) return;
  for (int i = threadIdx.x; i < nr; i += blockDim.x) {
    const int64_t r = rows ? int64_t(rows[r0 + i]) : r0 + i;
    s_rp[2 * i] = rp[r]; s_rp[2 * i + 1] = rp[r + 1];
  }
  __syncthreads();
  // item iterator of this half-wave: rows hw, hw + 8, ... of the block, 32 blocks at a time
  auto first_item = [&](int row) -> SpmvItem {
    SpmvItem it{0, row, 0, true};
    if (row < nr) { const int64_t b = s_rp[2 * row], e = s_rp[2 * row + 1]; it.off = b; it.n = int(e - b < 32 ? e - b : 32); it.last = e - b <= 32; }
    return it;
  };
  auto next_item = [&](const SpmvItem &c) -> SpmvItem {
    if (c.row >= nr) return c;
    if (!c.last) {
      const int64_t e = s_rp[2 * c.row + 1], b = c.off + 32;
      return SpmvItem{b, c.row, int(e - b < 32 ? e - b : 32), e - b <= 32};
    }
    return first_item(c.row + 8);
  };
  SpmvItem i0 = first_item(hw), i1 = next_item(i0), i2 = next_item(i1);
  auto load_col = [&](const SpmvItem &it) -> int32_t { return (it.row < nr && lig < it.n) ? col[it.off + lig] : 0; };
  struct Vals { double v[9]; };
  auto load_val = [&](const SpmvItem &it) -> Vals {
    Vals o;
    const bool ok = it.row < nr && lig < it.n;
    const double *b = val + (ok ? (it.off + lig) * 9 : 0);
#pragma unroll
    for (int e = 0; e < 9; ++e) { const double t = b[e]; o.v[e] = ok ? t : 0.0; }
    return o;
  };
  struct X3 { double x[3]; };
  auto load_x = [&](int32_t c) -> X3 { X3 o; const double *p = xu + int64_t(c) * 3; o.x[0] = p[0]; o.x[1] = p[1]; o.x[2] = p[2]; return o; };
  // prologue: indices of items 0 and 1, values and x of item 0
  int32_t c1 = load_col(i1);
  Vals v0 = load_val(i0);
  X3 x0 = load_x(load_col(i0));
  double acc[3] = {0, 0, 0};
  while (i0.row < nr) {
    const int32_t c2 = load_col(i2);  // indices two items ahead
    const Vals v1 = load_val(i1);     // values one item ahead
    const X3 x1 = load_x(c1);         // x one item ahead (c1 was requested an iteration ago)
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[r] += v0.v[r * 3 + j] * x0.x[j];
    if (i0.last) { // the row is complete: sum over the half-wave, one lane stores
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) acc[r] += __shfl_xor(acc[r], off, 32);
      }
      if (lig == 0) {
        const int64_t row = rows ? int64_t(rows[r0 + i0.row]) : r0 + i0.row;
        yu[row * 3 + 0] = acc[0]; yu[row * 3 + 1] = acc[1]; yu[row * 3 + 2] = acc[2];
      }
      acc[0] = acc[1] = acc[2] = 0;
    }
    i0 = i1; i1 = i2; i2 = next_item(i2);
    c1 = c2; v0 = v1; x0 = x1;
  }
}

// y += M x (same layout and row lists as k_spmv_planar): the B^T x_p part of the outer operator behind k_spmv_uu_pipe (taking
// the B^T row into the pipeline as a second kind of item was measured: 19.6 ms against 14.9 + 2.1)
template <int BR, int BC, int G, class VT = double>
__global__ __launch_bounds__(256) void k_spmv_planar_add(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                         const VT *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
                                                         const int32_t *__restrict__ rows = nullptr) {
  const int64_t ridx = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int lig = threadIdx.x & (G - 1);
  if (ridx >= n_rows) return;
  const int64_t row = rows ? int64_t(rows[ridx]) : ridx;
  double acc[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r) acc[r] = 0;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
  row_planar_dot<BR, BC, G, true, VT>(rs, len, col, val, x, lig, acc);
#pragma unroll
  for (int r = 0; r < BR; ++r) acc[r] = group_sum<G>(acc[r]);
  if (lig == 0) {
#pragma unroll
    for (int r = 0; r < BR; ++r) y[row * BR + r] += acc[r];
  }
}

// y = M x with BR x BC blocks, generic (B: 1 x DIM, B^T: DIM x 1, M_p / S_m: 1 x 1)
template <int BR, int BC, int G, class VT = double>
__global__ __launch_bounds__(256) void k_spmv_planar(int64_t n_rows, const int64_t *__restrict__ rp,
                                                     const int32_t *__restrict__ col, const VT *__restrict__ val,
                                                     const double *__restrict__ x, double *__restrict__ y,
                                                     const int32_t *__restrict__ rows = nullptr) {
  const int64_t ridx = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int lig = threadIdx.x & (G - 1);
  if (ridx >= n_rows) return;
  const int64_t row = rows ? int64_t(rows[ridx]) : ridx; // a row list: n_rows counts its entries
  double acc[BR];
#pragma unroll
  for (int r = 0; r < BR; ++r) acc[r] = 0;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
  row_planar_dot<BR, BC, G, true, VT>(rs, len, col, val, x, lig, acc);
#pragma unroll
  for (int r = 0; r < BR; ++r) acc[r] = group_sum<G>(acc[r]);
  if (lig == 0) {
#pragma unroll
    for (int r = 0; r < BR; ++r) y[row * BR + r] = acc[r];
  }
}

static inline unsigned blocks_for_rows(int64_t n_rows, int G) { return unsigned((n_rows * G + 255) / 256); }
// rows of `part` (kernels.hpp): count and list (nullptr = all rows in natural order)
struct RowPart { int64_t n; const int32_t *rows; };
static inline RowPart row_part(const PlanarCsr &M, int part) {
  if (part == 0) return {M.n_rows, nullptr};
  if (M.n_interior < 0) throw Error(IFEM_E_BADPARAM, "row split not built");
  if (part == 1) return {M.n_interior, M.split_rows.p};
  return {M.n_boundary, M.split_rows.p + M.n_interior};
}
static inline unsigned vgrid(int64_t n);

__global__ void k_to_f32(int64_t n, const double *__restrict__ a, float *__restrict__ b) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) b[i] = float(a[i]);
}

// single-precision copy of the A_uu values for the inner (preconditioner-only) solver: same planar layout
void auu_f32_refresh(ifem_ctx *ctx) {
  if (ctx->auu_f32_valid) return;
  const int64_t n = (int64_t)ctx->Auu.val.n;
  if (ctx->Auu_f32.n != (size_t)n) ctx->Auu_f32.alloc(n);
  if (n) hipLaunchKernelGGL(k_to_f32, dim3(8192), dim3(256), 0, ctx->stream, n, ctx->Auu.val.p, ctx->Auu_f32.p);
  ctx->auu_f32_valid = true;
}

// algorithmic bytes of one planar-CSR product: values (vb bytes each) + 4-byte column index per block, row pointer + output
// per row, the gathered input once per column
static inline double planar_bytes(const PlanarCsr &M, int64_t rows, int vb, int64_t n_cols, int in_b, int out_b) {
  const double share = M.n_rows > 0 ? double(rows) / double(M.n_rows) : 0.0;
  return share * double(M.nnzb) * (double(M.bs) * vb + 4) + double(rows) * (8 + out_b) + double(n_cols) * in_b;
}

void spmv_uu(ifem_ctx *ctx, const double *xu, const double *xp, double *yu, bool use_f32, int part) {
  const RowPart rp_ = row_part(ctx->Auu, part);
  const int64_t n = rp_.n;
  const int32_t *rows = rp_.rows;
  if (n == 0) return;
  if (ctx->Auu.val.n == 0) throw Error(IFEM_E_BADPARAM, "A_uu has no stored values (ifem_tuning::stored_uu = 0, or no assembly yet): this operation needs the block CSR");
  hipStream_t s = ctx->stream;
  const bool time_it = ctx->profile && xp == nullptr; // the A_uu-only launches of the inner solver: the dominant kernel
  if (use_f32) auu_f32_refresh(ctx);
  if (time_it) IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
  if (ctx->dim == 3 && !use_f32 && IFEM_UU_INTERLEAVED && ctx->tune.spmv_pipe) {
    const int rpb = 16; // rows per block = two per half-wave (measured at 128^3: 8 / 16 / 32 / 64 rows -> 14.4 / 14.4 / 15.1 / 16.0 ms)
    const unsigned nb = unsigned((n + rpb - 1) / rpb);
    {
      KScope ks(ctx, IFEM_KC_SPMV_UU, planar_bytes(ctx->Auu, n, 8, ctx->nUl, 24, 24));
      hipLaunchKernelGGL(k_spmv_uu_pipe, dim3(nb), dim3(256), size_t(2 * rpb) * sizeof(int64_t), s, n, ctx->Auu.rowptr.p, ctx->Auu.col.p,
                         ctx->Auu.val.p, xu, yu, rows, rpb);
    }
    if (xp && ctx->Bt.n_rows) { // + B^T x_p on the same rows
      KScope ks(ctx, IFEM_KC_SPMV_BBT, planar_bytes(ctx->Bt, n, 8, ctx->nPl, 8, 48));
      hipLaunchKernelGGL((k_spmv_planar_add<3, 1, 8>), dim3(blocks_for_rows(n, 8)), dim3(256), 0, s, n, ctx->Bt.rowptr.p, ctx->Bt.col.p,
                         ctx->Bt.val.p, xp, yu, rows);
    }
  } else if (ctx->dim == 3) {
    KScope ks(ctx, IFEM_KC_SPMV_UU, planar_bytes(ctx->Auu, n, use_f32 ? 4 : 8, ctx->nUl, 24, 24) + (xp ? planar_bytes(ctx->Bt, n, 8, ctx->nPl, 8, 0) : 0.0));
    const int Gsel = ctx->tune.spmv_lanes;
#define IFEM_SPMV3(G)                                                                                                  \
  if (use_f32)                                                                                                         \
    hipLaunchKernelGGL((k_spmv_uu<3, G, float>), dim3(blocks_for_rows(n, G)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,    \
                       ctx->Auu.col.p, ctx->Auu_f32.p, ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt.val.p, xu, xp, yu, rows); \
  else                                                                                                                 \
    hipLaunchKernelGGL((k_spmv_uu<3, G, double>), dim3(blocks_for_rows(n, G)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,   \
                       ctx->Auu.col.p, ctx->Auu.val.p, ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt.val.p, xu, xp, yu, rows);
    if (Gsel == 16) { IFEM_SPMV3(16) } else if (Gsel == 64) { IFEM_SPMV3(64) } else if (Gsel == 8) { IFEM_SPMV3(8) } else { IFEM_SPMV3(32) }
#undef IFEM_SPMV3
  } else {
    KScope ks(ctx, IFEM_KC_SPMV_UU, planar_bytes(ctx->Auu, n, use_f32 ? 4 : 8, ctx->nUl, 16, 16) + (xp ? planar_bytes(ctx->Bt, n, 8, ctx->nPl, 8, 0) : 0.0));
    constexpr int G = 16;
    if (use_f32)
      hipLaunchKernelGGL((k_spmv_uu<2, G, float>), dim3(blocks_for_rows(n, G)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,
                         ctx->Auu.col.p, ctx->Auu_f32.p, ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt.val.p, xu, xp, yu, rows);
    else
      hipLaunchKernelGGL((k_spmv_uu<2, G, double>), dim3(blocks_for_rows(n, G)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,
                         ctx->Auu.col.p, ctx->Auu.val.p, ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt.val.p, xu, xp, yu, rows);
  }
  ctx->last_spmv_f32 = use_f32;
  if (time_it) {
    IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->spmv_uu_ms_total += ms;
    ctx->timing.spmv_uu_calls++;
    // algorithmic bytes of this launch: values (+4-byte block column index) + row pointers + x + y
    const int d = ctx->dim;
    ctx->timing.spmv_uu_bytes = double(ctx->Auu.nnzb) * (d * d * (use_f32 ? 4 : 8) + 4) + double(n) * (8 + 2 * d * 8);
  }
}

// ---------------------------------------------------------------------------------------------------
// Scalar velocity operator (IFEM_AINV_SCALAR_GMRES): A_uu ~ blockdiag_c( P_c S^ P_c + (I - P_c) diag ), one scalar
// value per node pair applied to all dim components at once: 8 (4) bytes of matrix per block instead of 72 (36).
template <int DIM, int G, class VT>
__global__ __launch_bounds__(256) void k_spmv_scalar(int64_t n_rows, const int64_t *__restrict__ rp,
                                                     const int32_t *__restrict__ col, const VT *__restrict__ val,
                                                     const uint8_t *__restrict__ is_c, const double *__restrict__ sdiag,
                                                     const double *__restrict__ x, double *__restrict__ y) {
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int lig = threadIdx.x & (G - 1);
  if (row >= n_rows) return;
  double acc[DIM];
#pragma unroll
  for (int j = 0; j < DIM; ++j) acc[j] = 0;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
  for (int k = lig; k < len; k += G) {
    const int32_t c = col[rs + k];
    const double v = double(val[rs + k]);
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      double xv = x[int64_t(c) * DIM + j];
      if (is_c && is_c[int64_t(c) * DIM + j]) xv = 0.0;
      acc[j] += v * xv;
    }
  }
#pragma unroll
  for (int j = 0; j < DIM; ++j) acc[j] = group_sum<G>(acc[j]);
  if (lig == 0) {
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      const bool cst = is_c && is_c[row * DIM + j];
      y[row * DIM + j] = cst ? sdiag[row] * x[row * DIM + j] : acc[j];
    }
  }
}

__global__ void k_csr_diag(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                           const double *__restrict__ val, double *__restrict__ d) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t rs = rp[row];
  int lo = 0, hi = int(rp[row + 1] - rs) - 1;
  double out = 1.0;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int32_t v = col[rs + mid];
    if (v == row) { out = val[rs + mid]; break; }
    if (v < row) lo = mid + 1; else hi = mid - 1;
  }
  if (lo > hi) // not found by bisection: the distributed S_m keeps its columns in lattice-window order, not sorted by id
    for (int64_t k = rs; k < rp[row + 1]; ++k)
      if (col[k] == row) { out = val[k]; break; }
  d[row] = out;
}

template <int DIM>
__global__ void k_node_scale(int64_t n_nodes, const double *__restrict__ d, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_nodes * DIM; i += int64_t(gridDim.x) * blockDim.x)
    y[i] = x[i] / d[i / DIM];
}

void shat_refresh(ifem_ctx *ctx, bool f32) {
  if (!ctx->shat_valid) throw Error(IFEM_E_BADPARAM, "scalar operator not assembled: call ifem_set_ainv_kind before ifem_ins_assemble");
  if (ctx->shat_aux_valid) return;
  const int64_t n = ctx->nUo, nnz = (int64_t)ctx->Shat.n;
  if (ctx->shat_dinv.n != (size_t)n) ctx->shat_dinv.alloc(n);
  if (n) hipLaunchKernelGGL(k_csr_diag, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->Auu.rowptr.p,
                            ctx->Auu.col.p, ctx->Shat.p, ctx->shat_dinv.p);
  if (f32) {
    if (ctx->Shat_f32.n != (size_t)nnz) ctx->Shat_f32.alloc(nnz);
    if (nnz) hipLaunchKernelGGL(k_to_f32, dim3(8192), dim3(256), 0, ctx->stream, nnz, ctx->Shat.p, ctx->Shat_f32.p);
  }
  ctx->shat_aux_valid = true;
}

// y_u = S^ x_u per component with the constrained dofs of set `cset` kept as scaled identity rows
void spmv_shat(ifem_ctx *ctx, const double *xu, double *yu, bool f32) {
  const int64_t n = ctx->nUo;
  if (n == 0) return;
  hipStream_t s = ctx->stream;
  const int cset = ctx->asm_constraint_set;
  const uint8_t *isc = ctx->has_c[cset] ? ctx->is_c[cset].p : nullptr;
  const bool time_it = ctx->profile;
  if (time_it) IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
  if (ctx->dim == 3) {
    if (f32) hipLaunchKernelGGL((k_spmv_scalar<3, 32, float>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,
                                ctx->Auu.col.p, ctx->Shat_f32.p, isc, ctx->shat_dinv.p, xu, yu);
    else hipLaunchKernelGGL((k_spmv_scalar<3, 32, double>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,
                            ctx->Auu.col.p, ctx->Shat.p, isc, ctx->shat_dinv.p, xu, yu);
  } else {
    if (f32) hipLaunchKernelGGL((k_spmv_scalar<2, 16, float>), dim3(blocks_for_rows(n, 16)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,
                                ctx->Auu.col.p, ctx->Shat_f32.p, isc, ctx->shat_dinv.p, xu, yu);
    else hipLaunchKernelGGL((k_spmv_scalar<2, 16, double>), dim3(blocks_for_rows(n, 16)), dim3(256), 0, s, n, ctx->Auu.rowptr.p,
                            ctx->Auu.col.p, ctx->Shat.p, isc, ctx->shat_dinv.p, xu, yu);
  }
  if (time_it) {
    IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
    IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->spmv_uu_ms_total += ms;
    ctx->timing.spmv_uu_calls++;
    const int d = ctx->dim;
    ctx->timing.spmv_uu_bytes = double(ctx->Auu.nnzb) * ((f32 ? 4 : 8) + 4) + double(n) * (8 + 2 * d * 8 + 2 * d + 8);
  }
}

void shat_jacobi(ifem_ctx *ctx, const double *x, double *y) {
  const int64_t n = ctx->nUo;
  if (!n) return;
  if (ctx->dim == 3) hipLaunchKernelGGL((k_node_scale<3>), dim3(vgrid(n * 3)), dim3(256), 0, ctx->stream, n, ctx->shat_dinv.p, x, y);
  else hipLaunchKernelGGL((k_node_scale<2>), dim3(vgrid(n * 2)), dim3(256), 0, ctx->stream, n, ctx->shat_dinv.p, x, y);
}

void spmv_b(ifem_ctx *ctx, const double *xu, double *yp, int part) {
  const RowPart rp_ = row_part(ctx->B, part);
  const int64_t n = rp_.n;
  if (n == 0) return;
  KScope ks(ctx, IFEM_KC_SPMV_BBT, planar_bytes(ctx->B, n, 8, ctx->nUl, 8 * ctx->dim, 8));
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_spmv_planar<1, 3, 32>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, ctx->stream, n,
                       ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, xu, yp, rp_.rows);
  else
    hipLaunchKernelGGL((k_spmv_planar<1, 2, 16>), dim3(blocks_for_rows(n, 16)), dim3(256), 0, ctx->stream, n,
                       ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, xu, yp, rp_.rows);
}

void spmv_bt(ifem_ctx *ctx, const double *xp, double *yu) {
  const int64_t n = ctx->Bt.n_rows;
  if (n == 0) return;
  KScope ks(ctx, IFEM_KC_SPMV_BBT, planar_bytes(ctx->Bt, n, 8, ctx->nPl, 8, 8 * ctx->dim));
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_spmv_planar<3, 1, 8>), dim3(blocks_for_rows(n, 8)), dim3(256), 0, ctx->stream, n,
                       ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt.val.p, xp, yu);
  else
    hipLaunchKernelGGL((k_spmv_planar<2, 1, 4>), dim3(blocks_for_rows(n, 4)), dim3(256), 0, ctx->stream, n,
                       ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt.val.p, xp, yu);
}

// single-precision copies of B and B^T for the matrix-free S_m = B diag(M_u)^-1 B^T inside the approximate-preconditioner
// kinds on several ranks (an explicit S_m would need a 2-deep pressure halo): rebuilt lazily after every assemble
static void bbt_f32_refresh(ifem_ctx *ctx) {
  if (ctx->bbt_f32_valid) return;
  const int64_t nb = (int64_t)ctx->B.val.n, nt = (int64_t)ctx->Bt.val.n;
  if (ctx->B_f32.n != (size_t)nb) ctx->B_f32.alloc(nb);
  if (ctx->Bt_f32.n != (size_t)nt) ctx->Bt_f32.alloc(nt);
  if (nb) hipLaunchKernelGGL(k_to_f32, dim3(8192), dim3(256), 0, ctx->stream, nb, ctx->B.val.p, ctx->B_f32.p);
  if (nt) hipLaunchKernelGGL(k_to_f32, dim3(8192), dim3(256), 0, ctx->stream, nt, ctx->Bt.val.p, ctx->Bt_f32.p);
  ctx->bbt_f32_valid = true;
}
void spmv_b_f32(ifem_ctx *ctx, const double *xu, double *yp) {
  const int64_t n = ctx->B.n_rows;
  if (n == 0) return;
  bbt_f32_refresh(ctx);
  KScope ks(ctx, IFEM_KC_SPMV_BBT, planar_bytes(ctx->B, n, 4, ctx->nUl, 8 * ctx->dim, 8));
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_spmv_planar<1, 3, 32, float>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, ctx->stream, n,
                       ctx->B.rowptr.p, ctx->B.col.p, ctx->B_f32.p, xu, yp);
  else
    hipLaunchKernelGGL((k_spmv_planar<1, 2, 16, float>), dim3(blocks_for_rows(n, 16)), dim3(256), 0, ctx->stream, n,
                       ctx->B.rowptr.p, ctx->B.col.p, ctx->B_f32.p, xu, yp);
}
void spmv_bt_f32(ifem_ctx *ctx, const double *xp, double *yu) {
  const int64_t n = ctx->Bt.n_rows;
  if (n == 0) return;
  bbt_f32_refresh(ctx);
  KScope ks(ctx, IFEM_KC_SPMV_BBT, planar_bytes(ctx->Bt, n, 4, ctx->nPl, 8, 8 * ctx->dim));
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_spmv_planar<3, 1, 8, float>), dim3(blocks_for_rows(n, 8)), dim3(256), 0, ctx->stream, n,
                       ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt_f32.p, xp, yu);
  else
    hipLaunchKernelGGL((k_spmv_planar<2, 1, 4, float>), dim3(blocks_for_rows(n, 4)), dim3(256), 0, ctx->stream, n,
                       ctx->Bt.rowptr.p, ctx->Bt.col.p, ctx->Bt_f32.p, xp, yu);
}

void spmv_app(ifem_ctx *ctx, const double *xp, double *yp) {
  const int64_t n = ctx->Mp.n_rows;
  if (n == 0) return;
  KScope ks(ctx, IFEM_KC_SPMV_MP, planar_bytes(ctx->Mp, n, 8, ctx->nPl, 8, 8));
  hipLaunchKernelGGL((k_spmv_planar<1, 1, 8>), dim3(blocks_for_rows(n, 8)), dim3(256), 0, ctx->stream, n,
                     ctx->Mp.rowptr.p, ctx->Mp.col.p, ctx->App.p, xp, yp);
}

// d = diag(M) of a scalar planar CSR matrix with owned rows (M_p, S_m): Jacobi preconditioner of their CG solves
void scalar_diag(ifem_ctx *ctx, const PlanarCsr &M, const double *val, double *d) {
  const int64_t n = M.n_rows;
  if (n) hipLaunchKernelGGL(k_csr_diag, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, M.rowptr.p, M.col.p, val, d);
}

// diag(B diag(M_u)^-1 B^T) from the rows of B (1 x DIM blocks, planar) and the ghost-extended inverse lumped velocity mass: the
// Jacobi scaling of the S_m smoother where S_m itself is only applied as two SpMVs (several ranks without a 2-deep pressure halo)
template <int DIM>
__global__ void k_sm_diag_b(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col, const double *__restrict__ val,
                            const double *__restrict__ dinv_ext, double *__restrict__ d) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
  const double *vb = val + rs * DIM;
  double s = 0;
  for (int k = 0; k < len; ++k) {
    const int64_t c = col[rs + k];
#pragma unroll
    for (int j = 0; j < DIM; ++j) { const double v = vb[int64_t(j) * len + k]; s += v * v * dinv_ext[c * DIM + j]; }
  }
  d[row] = s;
}
void sm_diag_from_blocks(ifem_ctx *ctx, const double *dinv_mu_ext, double *d) {
  const int64_t n = ctx->B.n_rows;
  if (!n) return;
  if (ctx->dim == 3) hipLaunchKernelGGL(k_sm_diag_b<3>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, dinv_mu_ext, d);
  else hipLaunchKernelGGL(k_sm_diag_b<2>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, dinv_mu_ext, d);
}

void app_diag_setup(ifem_ctx *ctx) {
  const int64_t n = ctx->Mp.n_rows;
  if (ctx->app_diag.n != (size_t)n) ctx->app_diag.alloc(n);
  if (n) hipLaunchKernelGGL(k_csr_diag, dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->Mp.rowptr.p,
                            ctx->Mp.col.p, ctx->App.p, ctx->app_diag.p);
}

// use_f32: single-precision copy of the values (the preconditioner's CG(M_p) only; refreshed when M_p is re-integrated)
void spmv_mp(ifem_ctx *ctx, const double *xp, double *yp, int part, bool use_f32) {
  const RowPart rp_ = row_part(ctx->Mp, part);
  const int64_t n = rp_.n;
  if (n == 0) return;
  if (use_f32 && !ctx->mp_f32_valid) {
    const int64_t nv = (int64_t)ctx->Mp.val.n;
    if (ctx->Mp_f32.n != (size_t)nv) ctx->Mp_f32.alloc(nv);
    hipLaunchKernelGGL(k_to_f32, dim3(8192), dim3(256), 0, ctx->stream, nv, ctx->Mp.val.p, ctx->Mp_f32.p);
    ctx->mp_f32_valid = true;
  }
  const unsigned nb = blocks_for_rows(n, 8);
  KScope ks(ctx, IFEM_KC_SPMV_MP, planar_bytes(ctx->Mp, n, use_f32 ? 4 : 8, ctx->nPl, 8, 8));
  if (use_f32) hipLaunchKernelGGL((k_spmv_planar<1, 1, 8, float>), dim3(nb), dim3(256), 0, ctx->stream, n, ctx->Mp.rowptr.p, ctx->Mp.col.p, ctx->Mp_f32.p, xp, yp, rp_.rows);
  else hipLaunchKernelGGL((k_spmv_planar<1, 1, 8>), dim3(nb), dim3(256), 0, ctx->stream, n, ctx->Mp.rowptr.p, ctx->Mp.col.p, ctx->Mp.val.p, xp, yp, rp_.rows);
}

// ---------------------------------------------------------------------------------------------------
// mass_schur(1,1) = B diag(1/diag M_u) B^T (mpi_insim.cpp:44-49, PETSc MatMatMult in the reference): one wave per
// pressure row i.  The row's column ids go into a small open-addressing hash table in LDS (column -> position), the
// blocks k of B's row i (scaled by 1/diag M_u) into LDS arrays; then each half of the wave walks one (short) row k of B^T
// with consecutive lanes on consecutive entries (coalesced), looks the position up with ~1 probe and adds into the row's
// LDS accumulators.  (First version: one lane per block k, a serial walk of row k with a 7-step bisection per entry --
// 102 ms at 128^3, bound by the uncoalesced loads and the dependent LDS reads.)
__device__ inline void wsync() { // LDS operations of one wave execute in order: a wave-local barrier is enough
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int DIM>
__global__ __launch_bounds__(256) void k_schur_numeric(int64_t n_rows, int maxlen, int maxb, int hsize, const int64_t *__restrict__ rpS,
                                                       const int32_t *__restrict__ colS, double *__restrict__ valS,
                                                       const int64_t *__restrict__ rpB, const int32_t *__restrict__ colB,
                                                       const double *__restrict__ valB, const int64_t *__restrict__ rpT,
                                                       const int32_t *__restrict__ colT, const double *__restrict__ valT,
                                                       const double *__restrict__ dinv, const int32_t *__restrict__ rows) {
  extern __shared__ __align__(16) unsigned char smem_s[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // per wave: acc[maxlen] | bd[maxb][DIM] | ts[maxb] (int64) | hkey[hsize] | tlen[maxb] | hpos[hsize] (uint16)
  const size_t per_wave = size_t(maxlen) * 8 + size_t(maxb) * DIM * 8 + size_t(maxb) * 8 + size_t(hsize) * 4 + size_t(maxb) * 4 + size_t(hsize) * 2;
  unsigned char *base = smem_s + size_t(wave) * ((per_wave + 15) & ~size_t(15));
  double *acc = reinterpret_cast<double *>(base);
  double *bdl = acc + maxlen;
  int64_t *tsl = reinterpret_cast<int64_t *>(bdl + size_t(maxb) * DIM);
  int32_t *hkey = reinterpret_cast<int32_t *>(tsl + maxb);
  int32_t *tll = hkey + hsize;
  uint16_t *hpos = reinterpret_cast<uint16_t *>(tll + maxb);
  // XCD-aware row order: neighbouring pressure rows (Morton order) read the same rows of B^T -- 27 pressure rows share
  // each -- so every XCD gets one contiguous range of rows and finds them in its own L2
  const int64_t ridx = int64_t(xcd_swizzle(blockIdx.x, gridDim.x)) * 4 + wave;
  const bool active = ridx < n_rows;
  const int64_t row = active ? (rows ? int64_t(rows[ridx]) : ridx) : 0; // a row list: n_rows counts its entries
  const int64_t rs = active ? rpS[row] : 0;
  const int len = active ? int(rpS[row + 1] - rs) : 0;
  const unsigned hmask = unsigned(hsize - 1);
  for (int i = lane; i < hsize; i += 64) hkey[i] = -1;
  for (int i = lane; i < len; i += 64) acc[i] = 0.0;
  wsync();
  for (int i = lane; i < len; i += 64) { // insert (column -> position)
    const int32_t j = colS[rs + i];
    unsigned h = (unsigned(j) * 2654435761u >> 8) & hmask;
    while (true) {
      const int32_t old = atomicCAS(&hkey[h], -1, j);
      if (old == -1) { hpos[h] = uint16_t(i); break; }
      h = (h + 1) & hmask;
    }
  }
  const int64_t bs = active ? rpB[row] : 0;
  const int blen = active ? int(rpB[row + 1] - bs) : 0;
  for (int kb = lane; kb < blen; kb += 64) {
    const int32_t k = colB[bs + kb];
#pragma unroll
    for (int c = 0; c < DIM; ++c) bdl[kb * DIM + c] = valB[bs * DIM + int64_t(c) * blen + kb] * dinv[int64_t(k) * DIM + c];
    const int64_t ts = rpT[k];
    tsl[kb] = ts;
    tll[kb] = int32_t(rpT[k + 1] - ts);
  }
  wsync();
  const int half = lane >> 5, tl = lane & 31;
  // U rows of B^T per half-wave and trip: all their index / value loads are issued before the first lookup (one L2 round
  // trip per trip instead of one per row: the loop is latency-bound otherwise, 30 -> 12 ms at 128^3)
  constexpr int U = 4;
  for (int kb0 = 0; kb0 < blen; kb0 += 2 * U) {
    int32_t j[U];
    double v[U];
    bool ok[U];
    int tlen_[U];
    int64_t ts_[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int kb = kb0 + 2 * u + half;
      const bool live = kb < blen;
      const int kbs = live ? kb : 0;
      ts_[u] = tsl[kbs];
      tlen_[u] = tll[kbs];
      ok[u] = live && tl < tlen_[u];
      const int t = ok[u] ? tl : 0;
      j[u] = colT[ts_[u] + t];
      double s_ = 0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) s_ += bdl[kbs * DIM + c] * valT[ts_[u] * DIM + int64_t(c) * tlen_[u] + t];
      v[u] = s_;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (ok[u]) {
        unsigned h = (unsigned(j[u]) * 2654435761u >> 8) & hmask;
        while (true) { // every column of B^T's row k is a column of S_m's row i (pattern(S_m) = pattern(B B^T))
          const int32_t key = hkey[h];
          if (key == j[u]) { unsafeAtomicAdd(&acc[hpos[h]], v[u]); break; }
          if (key == -1) break;
          h = (h + 1) & hmask;
        }
      }
      // rows of B^T longer than half a wave (more than 8 cells around a node): the remaining entries, one trip each
      const int kb = kb0 + 2 * u + half;
      if (kb < blen)
        for (int t = tl + 32; t < tlen_[u]; t += 32) {
          const int32_t jj = colT[ts_[u] + t];
          double vv = 0;
#pragma unroll
          for (int c = 0; c < DIM; ++c) vv += bdl[kb * DIM + c] * valT[ts_[u] * DIM + int64_t(c) * tlen_[u] + t];
          unsigned h = (unsigned(jj) * 2654435761u >> 8) & hmask;
          while (true) {
            const int32_t key = hkey[h];
            if (key == jj) { unsafeAtomicAdd(&acc[hpos[h]], vv); break; }
            if (key == -1) break;
            h = (h + 1) & hmask;
          }
        }
    }
  }
  wsync();
  for (int i = lane; i < len; i += 64) valS[rs + i] = acc[i];
}

// rows of B with a constrained velocity dof among their columns (the rows of S_m a constrained-dof set changes)
template <int DIM>
__global__ void k_sm_row_flag(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                              const uint8_t *__restrict__ is_c, int64_t *__restrict__ flag) {
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  const int lig = threadIdx.x & 31;
  if (row >= n_rows) return;
  int v = 0; // 32 lanes per row: every lane looks at its share of the columns, then an OR over the half-wave
  for (int64_t k = rp[row] + lig; k < rp[row + 1] && !v; k += 32) {
    const int64_t nd = col[k];
#pragma unroll
    for (int c = 0; c < DIM; ++c) v |= is_c[nd * DIM + c];
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v |= __shfl_xor(v, off, 32);
  if (lig == 0) flag[row] = v ? 1 : 0;
}

static void launch_schur(ifem_ctx *ctx, int64_t n_list, const int32_t *rows, const double *vB, const double *vT, double *out) {
  const int maxlen = (ctx->Sm.max_row + 1) & ~1, maxb = (ctx->B.max_row + 1) & ~1;
  int hsize = 64;
  while (hsize < 2 * maxlen) hsize *= 2;
  const size_t per_wave = size_t(maxlen) * 8 + size_t(maxb) * ctx->dim * 8 + size_t(maxb) * 8 + size_t(hsize) * 4 + size_t(maxb) * 4 + size_t(hsize) * 2;
  const size_t smem = 4 * ((per_wave + 15) & ~size_t(15));
  if (smem > 160 * 1024) throw Error(IFEM_E_BADPARAM, "explicit S_m: a pressure row is too long for the LDS row buffers");
  const unsigned blocks = unsigned((n_list + 3) / 4);
  static bool attr_set[2] = {false, false};
  if (smem > 48 * 1024 && !attr_set[ctx->dim == 3]) {
    if (ctx->dim == 3) IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_schur_numeric<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    else IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_schur_numeric<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set[ctx->dim == 3] = true;
  }
  if (!n_list) return;
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_schur_numeric<3>), dim3(blocks), dim3(256), smem, ctx->stream, n_list, maxlen, maxb, hsize, ctx->Sm.rowptr.p,
                       ctx->Sm.col.p, out, ctx->B.rowptr.p, ctx->B.col.p, vB, ctx->Bt.rowptr.p, ctx->Bt.col.p, vT, ctx->dinvMu.p, rows);
  else
    hipLaunchKernelGGL((k_schur_numeric<2>), dim3(blocks), dim3(256), smem, ctx->stream, n_list, maxlen, maxb, hsize, ctx->Sm.rowptr.p,
                       ctx->Sm.col.p, out, ctx->B.rowptr.p, ctx->B.col.p, vB, ctx->Bt.rowptr.p, ctx->Bt.col.p, vT, ctx->dinvMu.p, rows);
  IFEM_HIP_CHECK(hipGetLastError());
}

void schur_numeric(ifem_ctx *ctx) {
  if (ctx->sm_valid) return;
  const int64_t n = ctx->Sm.n_rows;
  if (n == 0) return;
  hipStream_t s = ctx->stream;
  KScope ks(ctx, IFEM_KC_SCHUR_SETUP, 8.0 * double(ctx->Sm.val.n + ctx->B.val.n + ctx->Bt.val.n));
  // With the unconstrained blocks at hand (assemble.hip: B / B^T are masked copies of them) only the rows whose B row touches
  // a constrained dof differ from the S_m of the unconstrained blocks: that one is formed once per mesh, a new set copies it
  // and recomputes the touched rows (a few per cent of them on a box with Dirichlet walls).
  const bool partial = ctx->tune.geo_cache >= 1 && ctx->geo0_valid && ctx->geo_valid && ctx->B0.n == ctx->B.val.n;
  if (partial) {
    if (!ctx->sm0_valid) {
      if (ctx->Sm0.n != ctx->Sm.val.n) ctx->Sm0.alloc(ctx->Sm.val.n);
      launch_schur(ctx, n, nullptr, ctx->B0.p, ctx->Bt0.p, ctx->Sm0.p);
      ctx->sm0_valid = true;
    }
    IFEM_HIP_CHECK(hipMemcpyAsync(ctx->Sm.val.p, ctx->Sm0.p, ctx->Sm.val.n * sizeof(double), hipMemcpyDeviceToDevice, s));
    const int w = ctx->asm_constraint_set;
    if (ctx->has_c[w]) {
      DBuf<int64_t> flag;
      flag.alloc(n);
      const unsigned g = unsigned((n * 32 + 255) / 256);
      if (ctx->dim == 3) hipLaunchKernelGGL((k_sm_row_flag<3>), dim3(g), dim3(256), 0, s, n, ctx->B.rowptr.p, ctx->B.col.p, ctx->is_c[w].p, flag.p);
      else hipLaunchKernelGGL((k_sm_row_flag<2>), dim3(g), dim3(256), 0, s, n, ctx->B.rowptr.p, ctx->B.col.p, ctx->is_c[w].p, flag.p);
      const int64_t cnt = compact_flagged_rows(ctx, flag.p, n, ctx->sm_rows);
      launch_schur(ctx, cnt, ctx->sm_rows.p, ctx->B.val.p, ctx->Bt.val.p, ctx->Sm.val.p);
    }
  } else
    launch_schur(ctx, n, nullptr, ctx->B.val.p, ctx->Bt.val.p, ctx->Sm.val.p);
  ctx->sm_valid = true;
  ctx->sm_f32_valid = false;
  ctx->sm_version++;
}

// y = M x for a scalar matrix on the pattern of `M` with the values `val` (explicit T_pp on the pattern of S_m)
void spmv_planar_scalar(ifem_ctx *ctx, const PlanarCsr &M, const double *val, const double *xp, double *yp) {
  const int64_t n = M.n_rows;
  if (n == 0) return;
  KScope ks(ctx, IFEM_KC_SPMV_SM, planar_bytes(M, n, 8, n, 8, 8));
  hipLaunchKernelGGL((k_spmv_planar<1, 1, 32>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, ctx->stream, n, M.rowptr.p, M.col.p,
                     val, xp, yp);
}

void spmv_sm(ifem_ctx *ctx, const double *xp, double *yp, bool use_f32, int part) {
  if (ctx->Sm.n_rows == 0) return;
  if (use_f32 && !ctx->sm_f32_valid) {
    const int64_t nv = (int64_t)ctx->Sm.val.n;
    if (ctx->Sm_f32.n != (size_t)nv) ctx->Sm_f32.alloc(nv);
    hipLaunchKernelGGL(k_to_f32, dim3(8192), dim3(256), 0, ctx->stream, nv, ctx->Sm.val.p, ctx->Sm_f32.p);
    ctx->sm_f32_valid = true;
  }
  const RowPart rp_ = row_part(ctx->Sm, part);
  const int64_t n = rp_.n;
  if (n == 0) return;
  const int Gs = ctx->tune.sm_lanes;
  KScope ks(ctx, IFEM_KC_SPMV_SM, planar_bytes(ctx->Sm, n, use_f32 ? 4 : 8, ctx->halo.nranks > 1 ? ctx->halo.n_s_cols : ctx->nPl, 8, 8));
  if (use_f32 && Gs == 64)
    hipLaunchKernelGGL((k_spmv_planar<1, 1, 64, float>), dim3(blocks_for_rows(n, 64)), dim3(256), 0, ctx->stream, n,
                       ctx->Sm.rowptr.p, ctx->Sm.col.p, ctx->Sm_f32.p, xp, yp, rp_.rows);
  else if (use_f32 && Gs == 16)
    hipLaunchKernelGGL((k_spmv_planar<1, 1, 16, float>), dim3(blocks_for_rows(n, 16)), dim3(256), 0, ctx->stream, n,
                       ctx->Sm.rowptr.p, ctx->Sm.col.p, ctx->Sm_f32.p, xp, yp, rp_.rows);
  else if (use_f32)
    hipLaunchKernelGGL((k_spmv_planar<1, 1, 32, float>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, ctx->stream, n,
                       ctx->Sm.rowptr.p, ctx->Sm.col.p, ctx->Sm_f32.p, xp, yp, rp_.rows);
  else
    hipLaunchKernelGGL((k_spmv_planar<1, 1, 32>), dim3(blocks_for_rows(n, 32)), dim3(256), 0, ctx->stream, n,
                       ctx->Sm.rowptr.p, ctx->Sm.col.p, ctx->Sm.val.p, xp, yp, rp_.rows);
}

// ---------------------------------------------------------------------------------------------------
// vector kernels: grid-stride, 16-byte accesses where alignment allows
static inline unsigned vgrid(int64_t n) {
  int64_t g = (n + 511) / 512;
  return unsigned(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}

__global__ void k_axpy(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] += a * x[i];
}
__global__ void k_axpby(int64_t n, double a, const double *__restrict__ x, double b, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = a * x[i] + b * y[i];
}
__global__ void k_scale(int64_t n, double a, double *__restrict__ x) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) x[i] *= a;
}
__global__ void k_mul(int64_t n, const double *__restrict__ d, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = d[i] * x[i];
}
__global__ void k_recip(int64_t n, const double *__restrict__ d, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = 1.0 / d[i];
}

void v_axpy(ifem_ctx *ctx, int64_t n, double a, const double *x, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 24.0 * double(n));
  if (n) hipLaunchKernelGGL(k_axpy, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, a, x, y);
}
void v_axpby(ifem_ctx *ctx, int64_t n, double a, const double *x, double b, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 24.0 * double(n));
  if (n) hipLaunchKernelGGL(k_axpby, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, a, x, b, y);
}
void v_scale(ifem_ctx *ctx, int64_t n, double a, double *x) {
  KScope ks(ctx, IFEM_KC_VECTOR, 16.0 * double(n));
  if (n) hipLaunchKernelGGL(k_scale, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, a, x);
}
__global__ void k_scale_to(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = a * x[i];
}
// y = a x (the normalised Krylov vector in one pass instead of a copy and a scaling)
void v_scale_to(ifem_ctx *ctx, int64_t n, double a, const double *x, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 16.0 * double(n));
  if (n) hipLaunchKernelGGL(k_scale_to, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, a, x, y);
}
__global__ void k_scale_to2(int64_t n, double a, const double *__restrict__ x, double *__restrict__ y, double *__restrict__ z) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) { const double v = a * x[i]; y[i] = v; z[i] = v; }
}
// y = z = a x (the normalised Krylov vector into its basis column and into the fixed input vector of a captured operator)
void v_scale_to2(ifem_ctx *ctx, int64_t n, double a, const double *x, double *y, double *z) {
  KScope ks(ctx, IFEM_KC_VECTOR, 24.0 * double(n));
  if (n) hipLaunchKernelGGL(k_scale_to2, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, a, x, y, z);
}
void v_copy(ifem_ctx *ctx, int64_t n, const double *x, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 16.0 * double(n));
  if (n) IFEM_HIP_CHECK(hipMemcpyAsync(y, x, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
}
void v_zero(ifem_ctx *ctx, int64_t n, double *x) {
  KScope ks(ctx, IFEM_KC_VECTOR, 8.0 * double(n));
  if (n) IFEM_HIP_CHECK(hipMemsetAsync(x, 0, n * sizeof(double), ctx->stream));
}
void vec_mul(ifem_ctx *ctx, int64_t n, const double *d, const double *x, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 24.0 * double(n));
  if (n) hipLaunchKernelGGL(k_mul, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, d, x, y);
}
__global__ void k_div(int64_t n, const double *__restrict__ d, const double *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = x[i] / (d[i] != 0.0 ? d[i] : 1.0);
}
void vec_div(ifem_ctx *ctx, int64_t n, const double *d, const double *x, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 24.0 * double(n));
  if (n) hipLaunchKernelGGL(k_div, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, d, x, y);
}
void dinv_setup(ifem_ctx *ctx) {
  const int64_t n = ctx->diagMu.n;
  if (n) hipLaunchKernelGGL(k_recip, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, ctx->diagMu.p, ctx->dinvMu.p);
}

// Two-stage reduction: every block writes one partial per dot product (same-address atomics serialise in L2:
// 16k of them cost more than streaming the vectors), k_reduce_final sums the partials.  Deterministic.
constexpr int MDOT_MAXB = 4096;
template <int K>
__device__ inline void block_reduce_store(double *v, double *part /* [k][MDOT_MAXB] */) {
  __shared__ double sh[4][K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double t = v[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    v[k] = t;
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) sh[wave][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double t = 0;
    for (int w = 0; w < int(blockDim.x >> 6); ++w) t += sh[w][threadIdx.x];
    part[int64_t(threadIdx.x) * MDOT_MAXB + blockIdx.x] = t;
  }
}

__global__ __launch_bounds__(256) void k_reduce_final(int nblk, const double *__restrict__ part, double *__restrict__ out) {
  const int k = blockIdx.x;
  double t = 0;
  for (int i = threadIdx.x; i < nblk; i += blockDim.x) t += part[int64_t(k) * MDOT_MAXB + i];
  __shared__ double sh[4];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) out[k] = sh[0] + sh[1] + sh[2] + sh[3];
}

template <int K>
__global__ __launch_bounds__(256) void k_mdot(int64_t n, int k0, const double *__restrict__ V, int64_t ld,
                                              const double *__restrict__ w, double *__restrict__ out) {
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double wi = w[i];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] += V[int64_t(k0 + k) * ld + i] * wi;
  }
  block_reduce_store<K>(acc, out + int64_t(k0) * MDOT_MAXB);
}

// the K coefficients of one pass travel as kernel arguments (no staging buffer, no host wait); NORM: the pass also leaves the
// block sums of ||w||^2 of the updated vector in `part` (the last pass of a Gram-Schmidt sweep: the norm costs no extra pass)
struct MaxpyCoef { double v[8]; };
template <int K, bool NORM>
__global__ __launch_bounds__(256) void k_maxpy(int64_t n, int k0, const double *__restrict__ V, int64_t ld, MaxpyCoef hk,
                                               double *__restrict__ w, double *__restrict__ part) {
  double acc[1] = {0};
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    double t = w[i];
#pragma unroll
    for (int k = 0; k < K; ++k) t -= hk.v[k] * V[int64_t(k0 + k) * ld + i];
    w[i] = t;
    if (NORM) acc[0] += t * t;
  }
  if (NORM) block_reduce_store<1>(acc, part);
}

// Short vectors (the pressure space of the reference's SCnsIM tests: a few thousand entries, GMRES(200) on T_pp): the passes above
// cost a launch per 8 columns plus the final reduction, and the launches -- not the bytes -- are the time.  One launch for any number
// of columns: block b owns column b and writes its dot product directly; the multi-axpy takes up to 64 coefficients as arguments.
constexpr int64_t kShortVector = 32768;
__global__ __launch_bounds__(256) void k_mdot_cols(int64_t n, const double *__restrict__ V, int64_t ld, const double *__restrict__ w,
                                                   double *__restrict__ out) {
  const double *v = V + int64_t(blockIdx.x) * ld;
  double t = 0;
  for (int64_t i = threadIdx.x; i < n; i += 256) t += v[i] * w[i];
  __shared__ double sh[4];
  for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
struct MaxpyCoef64 { double v[64]; };
template <bool NORM>
__global__ __launch_bounds__(256) void k_maxpy_cols(int64_t n, int k, const double *__restrict__ V, int64_t ld, MaxpyCoef64 hk,
                                                    double *__restrict__ w, double *__restrict__ part) {
  double acc[1] = {0};
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    double t = w[i];
    for (int c = 0; c < k; ++c) t -= hk.v[c] * V[int64_t(c) * ld + i];
    w[i] = t;
    if (NORM) acc[0] += t * t;
  }
  if (NORM) block_reduce_store<1>(acc, part);
}

// out_host[i] = <V_i, w>, i < k.  Device scalars live in ctx->scal[0..63].  all_ranks = false: the local sums; true: summed
// over the ranks ON THE STREAM (comm.hip::allreduce_sum_dev on the device scalars) before the one copy to the host -- the
// Gram-Schmidt coefficients of FGMRES / the inner GMRES cross PCIe once per pass and wait for the device once.
void v_mdot(ifem_ctx *ctx, int64_t n, int k, const double *V, int64_t ld, const double *w, double *out_host, bool all_ranks) {
  if (k > 64) { // the staging buffers hold 64 dot products: long bases (GMRES(200)) go in chunks
    for (int k0 = 0; k0 < k; k0 += 64) v_mdot(ctx, n, std::min(64, k - k0), V + int64_t(k0) * ld, ld, w, out_host + k0, all_ranks);
    return;
  }
  hipStream_t s = ctx->stream;
  all_ranks = all_ranks && ctx->halo.nranks > 1;
  if (ctx->partials.n == 0) ctx->partials.alloc(size_t(64) * MDOT_MAXB);
  if (n == 0 && !all_ranks) { for (int i = 0; i < k; ++i) out_host[i] = 0; return; }
  if (n == 0) { // a rank without entries still takes part in the sum
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->scal.p, 0, k * sizeof(double), s));
    allreduce_sum_dev(ctx, ctx->scal.p, k);
    IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->scal.p, k * sizeof(double), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
    for (int i = 0; i < k; ++i) out_host[i] = ctx->h_scal[i];
    return;
  }
  const unsigned nblk = vgrid(n);
  if (n <= kShortVector) {
    KScope ks(ctx, IFEM_KC_MDOT, 16.0 * double(n) * k, 2.0 * double(n) * k);
    hipLaunchKernelGGL(k_mdot_cols, dim3(k), dim3(256), 0, s, n, V, ld, w, ctx->scal.p);
  } else {
  // every pass of K columns re-reads w; a column that IS w (norms) is counted once
  KScope ks(ctx, IFEM_KC_MDOT, 8.0 * double(n) * (double(k) + double((k + 7) / 8) - (V == w ? 1.0 : 0.0)), 2.0 * double(n) * k);
  int k0 = 0;
  while (k0 < k && n > 0) {
    const int r = k - k0;
    if (r >= 8) { hipLaunchKernelGGL((k_mdot<8>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 8; }
    else if (r >= 4) { hipLaunchKernelGGL((k_mdot<4>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 4; }
    else if (r >= 2) { hipLaunchKernelGGL((k_mdot<2>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 2; }
    else { hipLaunchKernelGGL((k_mdot<1>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 1; }
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(k), dim3(256), 0, s, (int)nblk, ctx->partials.p, ctx->scal.p);
  }
  if (all_ranks) allreduce_sum_dev(ctx, ctx->scal.p, k);
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->scal.p, k * sizeof(double), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  for (int i = 0; i < k; ++i) out_host[i] = ctx->h_scal[i];
}

// w -= sum_i h_i V_i.  norm2_out != nullptr: ||w||^2 of the result comes back with it (fused into the last pass; summed over the ranks
// on the stream when all_ranks) -- the one host wait of the call; without it the call does not wait for the device at all.
void v_maxpy(ifem_ctx *ctx, int64_t n, int k, const double *V, int64_t ld, const double *h_host, double *w, double *norm2_out, bool all_ranks) {
  hipStream_t s = ctx->stream;
  all_ranks = all_ranks && ctx->halo.nranks > 1;
  if (norm2_out && ctx->partials.n == 0) ctx->partials.alloc(size_t(64) * MDOT_MAXB);
  const unsigned nblk = vgrid(n);
  if (n > 0 && k > 0 && n <= kShortVector) {
    KScope ks(ctx, IFEM_KC_MAXPY, 8.0 * double(n) * (double(k) + 2.0 * double((k + 63) / 64)), 2.0 * double(n) * k + (norm2_out ? 2.0 * double(n) : 0.0));
    for (int k0 = 0; k0 < k; k0 += 64) {
      const int kk = std::min(64, k - k0);
      MaxpyCoef64 c{};
      for (int i = 0; i < kk; ++i) c.v[i] = h_host[k0 + i];
      if (norm2_out && k0 + kk >= k) hipLaunchKernelGGL((k_maxpy_cols<true>), dim3(nblk), dim3(256), 0, s, n, kk, V + int64_t(k0) * ld, ld, c, w, ctx->partials.p);
      else hipLaunchKernelGGL((k_maxpy_cols<false>), dim3(nblk), dim3(256), 0, s, n, kk, V + int64_t(k0) * ld, ld, c, w, ctx->partials.p);
    }
  } else if (n > 0 && k > 0) {
    KScope ks(ctx, IFEM_KC_MAXPY, 8.0 * double(n) * (double(k) + 2.0 * double((k + 7) / 8)), 2.0 * double(n) * k + (norm2_out ? 2.0 * double(n) : 0.0));
    int k0 = 0;
    while (k0 < k) {
      const int r = k - k0;
      const int K = r >= 8 ? 8 : (r >= 4 ? 4 : (r >= 2 ? 2 : 1));
      const bool last = norm2_out && k0 + K >= k;
      MaxpyCoef c{};
      for (int i = 0; i < K; ++i) c.v[i] = h_host[k0 + i];
#define IFEM_MAXPY_D(KK)                                                                                                \
      { if (last) hipLaunchKernelGGL((k_maxpy<KK, true>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, c, w, ctx->partials.p);   \
        else hipLaunchKernelGGL((k_maxpy<KK, false>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, c, w, ctx->partials.p); }
      if (K == 8) IFEM_MAXPY_D(8) else if (K == 4) IFEM_MAXPY_D(4) else if (K == 2) IFEM_MAXPY_D(2) else IFEM_MAXPY_D(1)
#undef IFEM_MAXPY_D
      k0 += K;
    }
  }
  if (!norm2_out) return;
  if (n > 0 && k > 0) hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, (int)nblk, ctx->partials.p, ctx->scal.p);
  else if (n > 0) { // nothing to subtract: the plain norm
    double out = 0;
    v_mdot(ctx, n, 1, w, n, w, &out, all_ranks);
    *norm2_out = out;
    return;
  } else IFEM_HIP_CHECK(hipMemsetAsync(ctx->scal.p, 0, sizeof(double), s)); // a rank without entries still takes part in the sum
  if (all_ranks) allreduce_sum_dev(ctx, ctx->scal.p, 1);
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  *norm2_out = ctx->h_scal[0];
}

// ---- single-precision Krylov basis of the inner (preconditioner-only) GMRES: V float, every other vector and all
// accumulation double.  k is padded to a multiple of 4 by the callers' allocation (columns beyond k exist and are finite).
template <int K>
__global__ __launch_bounds__(256) void k_mdot_f32(int64_t n, int k0, const float *__restrict__ V, int64_t ld,
                                                  const double *__restrict__ w, double *__restrict__ out) {
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; ++k) acc[k] = 0;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double wi = w[i];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] += double(V[int64_t(k0 + k) * ld + i]) * wi;
  }
  block_reduce_store<K>(acc, out + int64_t(k0) * MDOT_MAXB);
}
// w -= sum_k h_k V_k; NORM: also the block partials of ||w||^2 of the result (slot `slot` of the partials)
template <int K, bool NORM>
__global__ __launch_bounds__(256) void k_maxpy_f32(int64_t n, int k0, const float *__restrict__ V, int64_t ld,
                                                   const double *__restrict__ h, double *__restrict__ w,
                                                   double *__restrict__ part, int slot) {
  double hk[K];
#pragma unroll
  for (int k = 0; k < K; ++k) hk[k] = h[k0 + k];
  double nn[1] = {0};
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    double t = w[i];
#pragma unroll
    for (int k = 0; k < K; ++k) t -= hk[k] * double(V[int64_t(k0 + k) * ld + i]);
    w[i] = t;
    if (NORM) nn[0] += t * t;
  }
  if (NORM) block_reduce_store<1>(nn, part + int64_t(slot) * MDOT_MAXB);
}
__global__ void k_scale_store_f32(int64_t n, double s, const double *__restrict__ w, float *__restrict__ v) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) v[i] = float(w[i] * s);
}

static inline int pad4(int k) { return (k + 3) & ~3; }

void v_mdot_f32(ifem_ctx *ctx, int64_t n, int k, const float *V, int64_t ld, const double *w, double *out_host) {
  hipStream_t s = ctx->stream;
  if (ctx->partials.n == 0) ctx->partials.alloc(size_t(64) * MDOT_MAXB);
  if (n == 0) { for (int i = 0; i < k; ++i) out_host[i] = 0; return; }
  if (pad4(k) > 64) throw Error(IFEM_E_BADPARAM, "single-precision basis: at most 64 vectors");
  const unsigned nblk = vgrid(n);
  const int kp = pad4(k);
  {
  KScope ks(ctx, IFEM_KC_MDOT, double(n) * (4.0 * kp + 8.0 * ((kp + 15) / 16)), 2.0 * double(n) * k);
  for (int k0 = 0; k0 < kp;) {
    const int r = kp - k0;
    if (r >= 16) { hipLaunchKernelGGL((k_mdot_f32<16>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 16; }
    else if (r >= 12) { hipLaunchKernelGGL((k_mdot_f32<12>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 12; }
    else if (r >= 8) { hipLaunchKernelGGL((k_mdot_f32<8>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 8; }
    else { hipLaunchKernelGGL((k_mdot_f32<4>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, w, ctx->partials.p); k0 += 4; }
  }
  hipLaunchKernelGGL(k_reduce_final, dim3(k), dim3(256), 0, s, (int)nblk, ctx->partials.p, ctx->scal.p);
  }
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->scal.p, k * sizeof(double), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  for (int i = 0; i < k; ++i) out_host[i] = ctx->h_scal[i];
}

// w -= sum_{i<k} h[i] V_i; when norm2_out != nullptr also returns ||w||^2 of the result (local, not all-reduced)
void v_maxpy_f32(ifem_ctx *ctx, int64_t n, int k, const float *V, int64_t ld, const double *h_host, double *w, double *norm2_out) {
  if (n == 0 || k == 0) { if (norm2_out) *norm2_out = 0; return; }
  if (pad4(k) > 64) throw Error(IFEM_E_BADPARAM, "single-precision basis: at most 64 vectors");
  hipStream_t s = ctx->stream;
  const int kp = pad4(k);
  for (int i = 0; i < kp; ++i) ctx->h_scal[64 + i] = i < k ? h_host[i] : 0.0;
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->scal.p + 64, ctx->h_scal + 64, kp * sizeof(double), hipMemcpyHostToDevice, s));
  const unsigned nblk = vgrid(n);
  double *part = ctx->partials.p;
  const double *hd = ctx->scal.p + 64;
  {
  KScope ks(ctx, IFEM_KC_MAXPY, double(n) * (4.0 * kp + 16.0 * ((kp + 15) / 16)), 2.0 * double(n) * k);
#define IFEM_MAXPY(K)                                                                                                  \
  { if (last && norm2_out) hipLaunchKernelGGL((k_maxpy_f32<K, true>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, hd, w, part, 0); \
    else hipLaunchKernelGGL((k_maxpy_f32<K, false>), dim3(nblk), dim3(256), 0, s, n, k0, V, ld, hd, w, part, 0);        \
    k0 += K; }
  for (int k0 = 0; k0 < kp;) {
    const int r = kp - k0;
    const int step = r >= 16 ? 16 : (r >= 12 ? 12 : (r >= 8 ? 8 : 4));
    const bool last = k0 + step >= kp;
    if (step == 16) IFEM_MAXPY(16) else if (step == 12) IFEM_MAXPY(12) else if (step == 8) IFEM_MAXPY(8) else IFEM_MAXPY(4)
  }
#undef IFEM_MAXPY
  }
  if (norm2_out) {
    hipLaunchKernelGGL(k_reduce_final, dim3(1), dim3(256), 0, s, (int)nblk, part, ctx->scal.p);
    IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->scal.p, sizeof(double), hipMemcpyDeviceToHost, s));
  }
  IFEM_HIP_CHECK(hipStreamSynchronize(s)); // h_scal[64..] must stay untouched until the copy has been consumed
  if (norm2_out) *norm2_out = ctx->h_scal[0];
}

void v_scale_store_f32(ifem_ctx *ctx, int64_t n, double a, const double *w, float *v) {
  KScope ks(ctx, IFEM_KC_VECTOR, 12.0 * double(n));
  if (n) hipLaunchKernelGGL(k_scale_store_f32, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, a, w, v);
}

// ---- CG with the recurrence scalars on the device: no host round trip per iteration (the host only looks at the
// residual every few iterations).  Scalars live in ctx->scal[160..]: rr, rz, pq, alpha, beta.
constexpr int CGD = 160;
__global__ __launch_bounds__(256) void k_cgd_init(int64_t n, const double *__restrict__ b, const double *__restrict__ diag,
                                                  double *__restrict__ x, double *__restrict__ r, double *__restrict__ z,
                                                  double *__restrict__ p, double *__restrict__ part) {
  double acc[2] = {0, 0};
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double ri = b[i], zi = diag ? ri / (diag[i] != 0.0 ? diag[i] : 1.0) : ri;
    x[i] = 0.0; r[i] = ri; p[i] = zi;
    if (diag) z[i] = zi;
    acc[0] += ri * ri; acc[1] += ri * zi;
  }
  block_reduce_store<2>(acc, part);
}
// one block: sums the partials of `nrow` dot products and advances the recurrence
//   stage 0: rr, rz            stage 1: pq -> alpha = rz / pq            stage 2: rr', rz' -> beta = rz' / rz
// part 0: both in one launch (single rank); part 1: sums only, left in sc[CGD + 8..9] for the all-reduce over the ranks;
// part 2: the scalar update from those sums
__global__ __launch_bounds__(256) void k_cgd_scalars(int nblk, const double *__restrict__ part, double *__restrict__ sc, int stage, int piece) {
  __shared__ double sh[2][4];
  double a = 0, b = 0;
  if (piece != 2) {
    double t[2] = {0, 0};
    const int nrow = stage == 1 ? 1 : 2;
    for (int k = 0; k < nrow; ++k)
      for (int i = threadIdx.x; i < nblk; i += blockDim.x) t[k] += part[int64_t(k) * MDOT_MAXB + i];
    for (int k = 0; k < 2; ++k) {
      for (int off = 32; off > 0; off >>= 1) t[k] += __shfl_xor(t[k], off, 64);
      if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = t[k];
    }
    __syncthreads();
    a = sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3]; b = sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
  }
  if (threadIdx.x == 0) {
    if (piece == 1) { sc[CGD + 8] = a; sc[CGD + 9] = b; return; }
    if (piece == 2) { a = sc[CGD + 8]; b = sc[CGD + 9]; }
    if (stage == 0) { sc[CGD + 0] = a; sc[CGD + 1] = b; }
    else if (stage == 1) { sc[CGD + 2] = a; sc[CGD + 3] = a != 0.0 ? sc[CGD + 1] / a : 0.0; }
    else { sc[CGD + 4] = sc[CGD + 1] != 0.0 ? b / sc[CGD + 1] : 0.0; sc[CGD + 0] = a; sc[CGD + 1] = b; }
  }
}
// the scalar step of one stage: on several ranks the local sums are all-reduced on the stream between the two pieces
static void cgd_scalars(ifem_ctx *ctx, unsigned nblk, int stage) {
  hipStream_t s = ctx->stream;
  if (ctx->halo.nranks == 1) {
    hipLaunchKernelGGL(k_cgd_scalars, dim3(1), dim3(256), 0, s, (int)nblk, ctx->partials.p, ctx->scal.p, stage, 0);
    return;
  }
  hipLaunchKernelGGL(k_cgd_scalars, dim3(1), dim3(256), 0, s, (int)nblk, ctx->partials.p, ctx->scal.p, stage, 1);
  allreduce_sum_dev(ctx, ctx->scal.p + CGD + 8, stage == 1 ? 1 : 2);
  hipLaunchKernelGGL(k_cgd_scalars, dim3(1), dim3(64), 0, s, (int)nblk, ctx->partials.p, ctx->scal.p, stage, 2);
}
__global__ __launch_bounds__(256) void k_cgd_update(int64_t n, const double *__restrict__ sc, const double *__restrict__ diag,
                                                    const double *__restrict__ p, const double *__restrict__ q,
                                                    double *__restrict__ x, double *__restrict__ r, double *__restrict__ z,
                                                    double *__restrict__ part) {
  const double al = sc[CGD + 3];
  double acc[2] = {0, 0};
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    x[i] += al * p[i];
    const double ri = r[i] - al * q[i];
    r[i] = ri;
    const double zi = diag ? ri / (diag[i] != 0.0 ? diag[i] : 1.0) : ri;
    if (diag) z[i] = zi;
    acc[0] += ri * ri; acc[1] += ri * zi;
  }
  block_reduce_store<2>(acc, part);
}
__global__ void k_cgd_p(int64_t n, const double *__restrict__ sc, const double *__restrict__ z, double *__restrict__ p) {
  const double be = sc[CGD + 4];
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) p[i] = z[i] + be * p[i];
}
void cgd_init(ifem_ctx *ctx, int64_t n, const double *b, const double *diag, double *x, double *r, double *z, double *p) {
  KScope ks(ctx, IFEM_KC_CG_RECURRENCE, double(n) * (diag ? 48.0 : 32.0));
  if (ctx->partials.n == 0) ctx->partials.alloc(size_t(64) * MDOT_MAXB);
  const unsigned nblk = vgrid(n);
  hipLaunchKernelGGL(k_cgd_init, dim3(nblk), dim3(256), 0, ctx->stream, n, b, diag, x, r, z, p, ctx->partials.p);
  cgd_scalars(ctx, nblk, 0);
}
void cgd_alpha(ifem_ctx *ctx, int64_t n, const double *p, const double *q) {
  KScope ks(ctx, IFEM_KC_CG_RECURRENCE, 16.0 * double(n));
  const unsigned nblk = vgrid(n);
  hipLaunchKernelGGL((k_mdot<1>), dim3(nblk), dim3(256), 0, ctx->stream, n, 0, p, n, q, ctx->partials.p);
  cgd_scalars(ctx, nblk, 1);
}
void cgd_update(ifem_ctx *ctx, int64_t n, const double *diag, double *p, const double *q, double *x, double *r, double *z) {
  KScope ks(ctx, IFEM_KC_CG_RECURRENCE, double(n) * (diag ? 88.0 : 64.0));
  const unsigned nblk = vgrid(n);
  hipLaunchKernelGGL(k_cgd_update, dim3(nblk), dim3(256), 0, ctx->stream, n, ctx->scal.p, diag, p, q, x, r, z, ctx->partials.p);
  cgd_scalars(ctx, nblk, 2);
  hipLaunchKernelGGL(k_cgd_p, dim3(nblk), dim3(256), 0, ctx->stream, n, ctx->scal.p, diag ? z : r, p);
}
double cgd_rr(ifem_ctx *ctx) { // the only host synchronisation of the loop
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal + 200, ctx->scal.p + CGD, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return ctx->h_scal[200];
}

// ---- the same with ONE reduction per iteration (Chronopoulos / Gear): with u = D^-1 r and w = A u at hand,
//   gamma = <r,u>, delta = <w,u>, rho = <r,r>   (one fused pass, one all-reduce of three numbers on several ranks)
//   beta = gamma / gamma_old, alpha = gamma / (delta - beta gamma / alpha_old)
//   p = u + beta p,  s = w + beta s (= A p),  x += alpha p,  r -= alpha s,  u = D^-1 r          (one fused pass)
// Same iterates as the two-reduction recurrence in exact arithmetic; per iteration one all-reduce instead of two (CG(M_p):
// 36 iterations x every preconditioner application), 4 launches instead of 6 and 120 instead of 128 bytes per entry.
// Scalars in ctx->scal[CGD ..]: 0 rho, 1 gamma, 2 delta, 3 alpha, 4 beta; 8..10 staging of the all-reduce.
__global__ __launch_bounds__(256) void k_cg1_init(int64_t n, const double *__restrict__ b, const double *__restrict__ diag,
                                                  double *__restrict__ x, double *__restrict__ r, double *__restrict__ u,
                                                  double *__restrict__ p, double *__restrict__ s) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double ri = b[i];
    x[i] = 0.0; r[i] = ri; p[i] = 0.0; s[i] = 0.0;
    if (diag) u[i] = ri / (diag[i] != 0.0 ? diag[i] : 1.0);
  }
}
__global__ __launch_bounds__(256) void k_cg1_dots(int64_t n, const double *__restrict__ r, const double *__restrict__ u,
                                                  const double *__restrict__ w, double *__restrict__ part) {
  double acc[3] = {0, 0, 0};
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double ri = r[i], ui = u[i];
    acc[0] += ri * ui; acc[1] += w[i] * ui; acc[2] += ri * ri;
  }
  block_reduce_store<3>(acc, part);
}
__global__ __launch_bounds__(256) void k_cg1_scalars(int nblk, const double *__restrict__ part, double *__restrict__ sc, int first, int piece) {
  __shared__ double sh[3][4];
  double v[3] = {0, 0, 0};
  if (piece != 2) {
    for (int k = 0; k < 3; ++k) {
      double t = 0;
      for (int i = threadIdx.x; i < nblk; i += blockDim.x) t += part[int64_t(k) * MDOT_MAXB + i];
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
      if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = t;
    }
    __syncthreads();
    for (int k = 0; k < 3; ++k) v[k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
  }
  if (threadIdx.x == 0) {
    if (piece == 1) { sc[CGD + 8] = v[0]; sc[CGD + 9] = v[1]; sc[CGD + 10] = v[2]; return; }
    if (piece == 2) { v[0] = sc[CGD + 8]; v[1] = sc[CGD + 9]; v[2] = sc[CGD + 10]; }
    const double gam = v[0], del = v[1];
    double be = 0.0, al;
    if (first) al = del != 0.0 ? gam / del : 0.0;
    else {
      const double go = sc[CGD + 1], ao = sc[CGD + 3];
      be = go != 0.0 ? gam / go : 0.0;
      const double den = del - (ao != 0.0 ? be * gam / ao : 0.0);
      al = den != 0.0 ? gam / den : 0.0;
    }
    sc[CGD + 0] = v[2]; sc[CGD + 1] = gam; sc[CGD + 2] = del; sc[CGD + 3] = al; sc[CGD + 4] = be;
  }
}
__global__ __launch_bounds__(256) void k_cg1_update(int64_t n, const double *__restrict__ sc, const double *__restrict__ diag,
                                                    double *u, const double *__restrict__ w, double *__restrict__ p,
                                                    double *__restrict__ s, double *__restrict__ x, double *r) {
  const double al = sc[CGD + 3], be = sc[CGD + 4];
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double pi = u[i] + be * p[i], si = w[i] + be * s[i];
    p[i] = pi; s[i] = si;
    x[i] += al * pi;
    const double ri = r[i] - al * si;
    r[i] = ri;
    if (diag) u[i] = ri / (diag[i] != 0.0 ? diag[i] : 1.0); // (no diag: u IS r -- plain CG, the caller passes the same array -- or the output of an external preconditioner, pcg_mg_sm)
  }
}
void cg1_init(ifem_ctx *ctx, int64_t n, const double *b, const double *diag, double *x, double *r, double *u, double *p, double *s) {
  KScope ks(ctx, IFEM_KC_CG_RECURRENCE, double(n) * (diag ? 56.0 : 40.0));
  if (ctx->partials.n == 0) ctx->partials.alloc(size_t(64) * MDOT_MAXB);
  if (n) hipLaunchKernelGGL(k_cg1_init, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, b, diag, x, r, u, p, s);
}
void cg1_dots(ifem_ctx *ctx, int64_t n, const double *r, const double *u, const double *w, bool first) {
  KScope ks(ctx, IFEM_KC_CG_RECURRENCE, double(n) * (u == r ? 16.0 : 24.0), 6.0 * double(n));
  hipStream_t st = ctx->stream;
  const unsigned nblk = vgrid(n);
  hipLaunchKernelGGL(k_cg1_dots, dim3(nblk), dim3(256), 0, st, n, r, u, w, ctx->partials.p);
  if (ctx->halo.nranks == 1) {
    hipLaunchKernelGGL(k_cg1_scalars, dim3(1), dim3(256), 0, st, (int)nblk, ctx->partials.p, ctx->scal.p, first ? 1 : 0, 0);
    return;
  }
  hipLaunchKernelGGL(k_cg1_scalars, dim3(1), dim3(256), 0, st, (int)nblk, ctx->partials.p, ctx->scal.p, first ? 1 : 0, 1);
  allreduce_sum_dev(ctx, ctx->scal.p + CGD + 8, 3);
  hipLaunchKernelGGL(k_cg1_scalars, dim3(1), dim3(64), 0, st, (int)nblk, ctx->partials.p, ctx->scal.p, first ? 1 : 0, 2);
}
void cg1_update(ifem_ctx *ctx, int64_t n, const double *diag, double *u, const double *w, double *p, double *s, double *x, double *r) {
  KScope ks(ctx, IFEM_KC_CG_RECURRENCE, double(n) * (diag ? 104.0 : 88.0), 8.0 * double(n));
  if (n) hipLaunchKernelGGL(k_cg1_update, dim3(vgrid(n)), dim3(256), 0, ctx->stream, n, ctx->scal.p, diag, u, w, p, s, x, r);
}

double v_dot(ifem_ctx *ctx, int64_t n, const double *x, const double *y) {
  double out = 0;
  v_mdot(ctx, n, 1, x, n, y, &out);
  return out;
}

double bv_dot(ifem_ctx *ctx, const double *x, const double *y) {
  double d = v_dot(ctx, ctx->dim * ctx->nUo + ctx->nPo, x, y);
  allreduce_sum(ctx, &d, 1);
  return d;
}

__global__ __launch_bounds__(256) void k_minmax(int64_t n, const double *__restrict__ x, double *__restrict__ out) {
  double mn = 1e300, mx = -1e300;
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double v = x[i];
    mn = fmin(mn, v); mx = fmax(mx, v);
  }
  for (int off = 32; off > 0; off >>= 1) { mn = fmin(mn, __shfl_xor(mn, off, 64)); mx = fmax(mx, __shfl_xor(mx, off, 64)); }
  if ((threadIdx.x & 63) == 0) {
    // f64 min/max through order-preserving CAS loops
    unsigned long long *pmn = (unsigned long long *)&out[0], *pmx = (unsigned long long *)&out[1];
    unsigned long long old = *pmn;
    while (__longlong_as_double((long long)old) > mn) {
      const unsigned long long prev = atomicCAS(pmn, old, (unsigned long long)__double_as_longlong(mn));
      if (prev == old) break;
      old = prev;
    }
    old = *pmx;
    while (__longlong_as_double((long long)old) < mx) {
      const unsigned long long prev = atomicCAS(pmx, old, (unsigned long long)__double_as_longlong(mx));
      if (prev == old) break;
      old = prev;
    }
  }
}

void v_minmax(ifem_ctx *ctx, int64_t n, const double *x, double *mn, double *mx) {
  hipStream_t s = ctx->stream;
  ctx->h_scal[0] = 1e300; ctx->h_scal[1] = -1e300;
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->scal.p, ctx->h_scal, 2 * sizeof(double), hipMemcpyHostToDevice, s));
  if (n) hipLaunchKernelGGL(k_minmax, dim3(vgrid(n)), dim3(256), 0, s, n, x, ctx->scal.p);
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal, ctx->scal.p, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  *mn = ctx->h_scal[0]; *mx = ctx->h_scal[1];
}

// ---------------------------------------------------------------------------------------------------
// node-block Jacobi: inverse of the dim x dim diagonal blocks of A_uu
template <int DIM>
__global__ void k_bjac_setup(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ diag_pos,
                             const double *__restrict__ val, double *__restrict__ out) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
  const int pos = diag_pos[row]; // (the blocks of a row are not necessarily in column order: setup.hip)
  double D[DIM * DIM], Di[DIM * DIM];
  for (int e = 0; e < DIM * DIM; ++e) D[e] = (pos >= 0) ? val[uu_base(rs, len, pos, DIM * DIM) + int64_t(e) * uu_estride(len)] : ((e / DIM == e % DIM) ? 1.0 : 0.0);
  if constexpr (DIM == 2) {
    const double r = 1.0 / (D[0] * D[3] - D[1] * D[2]);
    Di[0] = D[3] * r; Di[1] = -D[1] * r; Di[2] = -D[2] * r; Di[3] = D[0] * r;
  } else {
    const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
    const double r = 1.0 / (D[0] * c00 + D[1] * c01 + D[2] * c02);
    Di[0] = c00 * r; Di[3] = c01 * r; Di[6] = c02 * r;
    Di[1] = (D[2] * D[7] - D[1] * D[8]) * r; Di[4] = (D[0] * D[8] - D[2] * D[6]) * r; Di[7] = (D[1] * D[6] - D[0] * D[7]) * r;
    Di[2] = (D[1] * D[5] - D[2] * D[4]) * r; Di[5] = (D[2] * D[3] - D[0] * D[5]) * r; Di[8] = (D[0] * D[4] - D[1] * D[3]) * r;
  }
  for (int e = 0; e < DIM * DIM; ++e) out[row * DIM * DIM + e] = Di[e];
}

// the same through LDS for the block-interleaved layout (a diagonal block is DIM^2 contiguous doubles): 256 rows per workgroup, the
// blocks fetched and the inverses stored with consecutive lanes on consecutive doubles (one thread per row reads its 72 bytes with nine
// separate instructions and a whole wave touches 64 different lines per instruction: 1.5 ms for the 17 M rows of the 128^3 mesh)
template <int DIM>
__global__ __launch_bounds__(256) void k_bjac_setup_lds(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ diag_pos,
                                                        const double *__restrict__ val, double *__restrict__ out) {
  constexpr int BS = DIM * DIM;
  __shared__ double blk[256 * BS];
  __shared__ int64_t base[256];
  const int64_t row0 = int64_t(blockIdx.x) * 256;
  const int nr = int(n_rows - row0 < 256 ? n_rows - row0 : 256);
  if (int(threadIdx.x) < nr) {
    const int64_t row = row0 + threadIdx.x;
    const int pos = diag_pos[row];
    base[threadIdx.x] = pos >= 0 ? (rp[row] + pos) * BS : int64_t(-1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr * BS; i += 256) {
    const int r = i / BS, e = i - r * BS;
    blk[i] = base[r] >= 0 ? val[base[r] + e] : ((e / DIM == e % DIM) ? 1.0 : 0.0);
  }
  __syncthreads();
  if (int(threadIdx.x) < nr) {
    double D[BS], Di[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) D[e] = blk[threadIdx.x * BS + e];
    if constexpr (DIM == 2) {
      const double r = 1.0 / (D[0] * D[3] - D[1] * D[2]);
      Di[0] = D[3] * r; Di[1] = -D[1] * r; Di[2] = -D[2] * r; Di[3] = D[0] * r;
    } else {
      const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
      const double r = 1.0 / (D[0] * c00 + D[1] * c01 + D[2] * c02);
      Di[0] = c00 * r; Di[3] = c01 * r; Di[6] = c02 * r;
      Di[1] = (D[2] * D[7] - D[1] * D[8]) * r; Di[4] = (D[0] * D[8] - D[2] * D[6]) * r; Di[7] = (D[1] * D[6] - D[0] * D[7]) * r;
      Di[2] = (D[1] * D[5] - D[2] * D[4]) * r; Di[5] = (D[2] * D[3] - D[0] * D[5]) * r; Di[8] = (D[0] * D[4] - D[1] * D[3]) * r;
    }
#pragma unroll
    for (int e = 0; e < BS; ++e) blk[threadIdx.x * BS + e] = Di[e]; // (a thread rewrites the entries it read)
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nr * BS; i += 256) out[row0 * BS + i] = blk[i];
}

template <int DIM>
__global__ void k_bjac_apply(int64_t n_rows, const double *__restrict__ bj, const double *__restrict__ x,
                             double *__restrict__ y) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  double xv[DIM];
  for (int j = 0; j < DIM; ++j) xv[j] = x[row * DIM + j];
  for (int r = 0; r < DIM; ++r) {
    double t = 0;
    for (int j = 0; j < DIM; ++j) t += bj[row * DIM * DIM + r * DIM + j] * xv[j];
    y[row * DIM + r] = t;
  }
}

// node-block Jacobi on a single-precision vector with single-precision blocks (inner solver only): y (double) = bj * x
template <int DIM>
__global__ void k_bjac_apply_f32(int64_t n_rows, const float *__restrict__ bj, const float *__restrict__ x,
                                 double *__restrict__ y) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n_rows) return;
  float xv[DIM];
  for (int j = 0; j < DIM; ++j) xv[j] = x[row * DIM + j];
  for (int r = 0; r < DIM; ++r) {
    double t = 0;
    for (int j = 0; j < DIM; ++j) t += double(bj[row * DIM * DIM + r * DIM + j]) * double(xv[j]);
    y[row * DIM + r] = t;
  }
}
// single-precision copy of the inverse node blocks (preconditioner-only consumers), refreshed after every bjac set-up
const float *bjac_f32_ptr(ifem_ctx *ctx) {
  if (!ctx->bjac_f32_valid && ctx->bjac.n) {
    if (ctx->bjac_f32.n != ctx->bjac.n) ctx->bjac_f32.alloc(ctx->bjac.n);
    hipLaunchKernelGGL(k_to_f32, dim3(vgrid((int64_t)ctx->bjac.n)), dim3(256), 0, ctx->stream, (int64_t)ctx->bjac.n, ctx->bjac.p, ctx->bjac_f32.p);
    ctx->bjac_f32_valid = true;
  }
  return ctx->bjac_f32.p;
}
void bjac_apply_f32(ifem_ctx *ctx, const float *x, double *y) {
  const int64_t n = ctx->nUo;
  if (!n) return;
  (void)bjac_f32_ptr(ctx);
  KScope ks(ctx, IFEM_KC_VECTOR, double(n) * ctx->dim * (4.0 * ctx->dim + 4 + 8));
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_bjac_apply_f32<3>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->bjac_f32.p, x, y);
  else
    hipLaunchKernelGGL((k_bjac_apply_f32<2>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->bjac_f32.p, x, y);
}

void bjac_setup(ifem_ctx *ctx) {
  ctx->bjac_f32_valid = false;
  const int64_t n = ctx->nUo;
  if (!n) return;
  KScope ks(ctx, IFEM_KC_SMOOTHER_SETUP, double(n) * ctx->dim * ctx->dim * 16.0);
  if (IFEM_UU_INTERLEAVED) {
    if (ctx->dim == 3)
      hipLaunchKernelGGL((k_bjac_setup_lds<3>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                         ctx->Auu.rowptr.p, ctx->uu_diag_pos.p, ctx->Auu.val.p, ctx->bjac.p);
    else
      hipLaunchKernelGGL((k_bjac_setup_lds<2>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                         ctx->Auu.rowptr.p, ctx->uu_diag_pos.p, ctx->Auu.val.p, ctx->bjac.p);
  } else if (ctx->dim == 3)
    hipLaunchKernelGGL((k_bjac_setup<3>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       ctx->Auu.rowptr.p, ctx->uu_diag_pos.p, ctx->Auu.val.p, ctx->bjac.p);
  else
    hipLaunchKernelGGL((k_bjac_setup<2>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n,
                       ctx->Auu.rowptr.p, ctx->uu_diag_pos.p, ctx->Auu.val.p, ctx->bjac.p);
}

void bjac_apply(ifem_ctx *ctx, const double *x, double *y) {
  const int64_t n = ctx->nUo;
  if (!n) return;
  KScope ks(ctx, IFEM_KC_VECTOR, double(n) * ctx->dim * (8.0 * ctx->dim + 16));
  if (ctx->dim == 3)
    hipLaunchKernelGGL((k_bjac_apply<3>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->bjac.p, x, y);
  else
    hipLaunchKernelGGL((k_bjac_apply<2>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, ctx->stream, n, ctx->bjac.p, x, y);
}

// AffineConstraints::distribute for Dirichlet lines on a compact owned vector [u_o | p_o]
__global__ void k_apply_constraints(int64_t n_u_owned, int64_t n_owned, int64_t p_off_ext,
                                    const uint8_t *__restrict__ is_c, const double *__restrict__ cval,
                                    double *__restrict__ x) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n_owned; i += int64_t(gridDim.x) * blockDim.x) {
    const int64_t e = (i < n_u_owned) ? i : (p_off_ext + (i - n_u_owned));
    if (is_c[e]) x[i] = cval[e];
  }
}

void apply_constraints(ifem_ctx *ctx, int which, double *x) {
  if (!ctx->has_c[which]) return;
  const int64_t nuo = ctx->dim * ctx->nUo, n = nuo + ctx->nPo;
  KScope ks(ctx, IFEM_KC_OTHER, double(n));
  hipLaunchKernelGGL(k_apply_constraints, dim3(vgrid(n)), dim3(256), 0, ctx->stream, nuo, n, ctx->dim * ctx->nUl,
                     ctx->is_c[which].p, ctx->cval[which].p, x);
}

} // namespace ifem
