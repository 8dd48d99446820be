// tpp.hip -- the pressure Schur complement of SUPGFluidSolver::BlockIncompSchurPreconditioner as an explicit matrix.
//   reference (mpi_supg_solver.cpp:19-32, 56-133, 163-179): T_pp = A_pp - A_pv P_vv^-1 A_vp is an OPERATOR (P_vv = ILU(0)
//   of A_vv), solved by GMRES(200) that is preconditioned with the ILU(0) of the assembled B2pp = A_pp - A_pv D^-1 A_vp.
// Here P_vv^-1 is the node-block Jacobi of A_vv, so T_pp itself has the sparsity of A_pv A_vp (= the pattern of
// mass_schur) and is assembled once per Newton iteration:  T~[i,j] = A_pp[i,j] - sum_k A_pv[i,k] Binv_k A_vp[k,j].
// The inner GMRES(200) on it is preconditioned by an ILU(0) of T~ -- the reference's choice for B2pp (Euclid ILU(0),
// preconditioner_pilut.cpp:124-138) -- factorised and applied on the device by level scheduling: rows are grouped into
// levels of the elimination DAG (natural order, as Euclid's serial sweep) or into the colours of a greedy colouring of the
// matrix graph (multicolour ILU(0): a few dozen levels whatever the mesh size); one launch per level, one wave per row in
// the factorisation, 16 lanes per row in the triangular solves.  Only the preconditioner is affected.
// Several ranks (round 4): the operator T_pp stays distributed (solver.hip), its preconditioner is the ILU(0) of the OWNED x OWNED
// block of T~ on every rank -- block-Jacobi ILU, what Euclid does by default across ranks (mpi_supg_solver.cpp:49-53,120-133):
// pattern = owned block of pattern(M_p^2) (setup.hip::build_schur_pattern_owned), products through owned velocity nodes only.
// A zero, tiny or non-finite pivot (T~ is not an M-matrix, the relaxed MILU moves dropped fill onto the diagonal) is detected
// after the factorisation; the caller falls back to Jacobi.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

__device__ inline void wsync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the pattern T~ lives on: mass_schur's on one rank, the owned x owned block on several
static PlanarCsr &tpp_pat(ifem_ctx *ctx) { return ctx->halo.nranks > 1 ? ctx->TppPat : ctx->Sm; }
static const PlanarCsr &tpp_pat(const ifem_ctx *ctx) { return ctx->halo.nranks > 1 ? ctx->TppPat : ctx->Sm; }

// one wave per pressure row i, as k_schur_numeric; the row of A_pp is merged in at the end
template <int DIM>
__global__ __launch_bounds__(256) void k_tpp_numeric(int64_t n_rows, int maxlen, const int64_t *__restrict__ rpS,
                                                     const int32_t *__restrict__ colS, double *__restrict__ valS,
                                                     const int64_t *__restrict__ rpB, const int32_t *__restrict__ colB,
                                                     const double *__restrict__ valB, const int64_t *__restrict__ rpT,
                                                     const int32_t *__restrict__ colT, const double *__restrict__ valT,
                                                     const double *__restrict__ binv, const int64_t *__restrict__ rpM,
                                                     const int32_t *__restrict__ colM, const double *__restrict__ app, int64_t n_u_owned) {
  extern __shared__ __align__(16) unsigned char smem_t[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *acc = reinterpret_cast<double *>(smem_t) + size_t(wave) * maxlen;
  int32_t *cols = reinterpret_cast<int32_t *>(reinterpret_cast<double *>(smem_t) + size_t(4) * maxlen) + size_t(wave) * maxlen;
  const int64_t row = int64_t(blockIdx.x) * 4 + wave;
  const bool active = row < n_rows;
  const int64_t rs = active ? rpS[row] : 0;
  const int len = active ? int(rpS[row + 1] - rs) : 0;
  for (int i = lane; i < len; i += 64) { cols[i] = colS[rs + i]; acc[i] = 0.0; }
  __syncthreads();
  auto find = [&](int32_t j) {
    int lo = 0, hi = len - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const int32_t cv = cols[mid];
      if (cv == j) return mid;
      if (cv < j) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
  };
  if (active) {
    const int64_t bs = rpB[row];
    const int blen = int(rpB[row + 1] - bs);
    for (int kb = lane; kb < blen; kb += 64) {
      const int32_t k = colB[bs + kb];
      if (k >= n_u_owned) continue; // a ghost velocity node: its row of A_vp and its diagonal block live on another rank
      double a[DIM], bd[DIM];
#pragma unroll
      for (int e = 0; e < DIM; ++e) a[e] = valB[bs * DIM + int64_t(e) * blen + kb];
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double t = 0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) t += a[e] * binv[int64_t(k) * DIM * DIM + e * DIM + c];
        bd[c] = t;
      }
      const int64_t ts = rpT[k];
      const int tlen = int(rpT[k + 1] - ts);
      for (int t = 0; t < tlen; ++t) {
        double v = 0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) v += bd[c] * valT[ts * DIM + int64_t(c) * tlen + t];
        const int p = find(colT[ts + t]);
        if (p >= 0) unsafeAtomicAdd(&acc[p], -v);
      }
    }
  }
  __syncthreads();
  if (active) {
    const int64_t ms = rpM[row];
    const int mlen = int(rpM[row + 1] - ms);
    for (int t = lane; t < mlen; t += 64) {
      const int p = find(colM[ms + t]);
      if (p >= 0) acc[p] += app[ms + t]; // distinct columns: no two lanes meet
    }
  }
  __syncthreads();
  for (int i = lane; i < len; i += 64) valS[rs + i] = acc[i];
}

const PlanarCsr &tpp_pattern(ifem_ctx *ctx) {
  PlanarCsr &Pt = tpp_pat(ctx);
  if (Pt.n_rows == 0) { if (ctx->halo.nranks > 1) build_schur_pattern_owned(ctx); else build_schur_pattern(ctx); }
  return Pt;
}

// out = A_pp - A_pv blockdiag(binv) A_vp on the pattern of T_pp (owned velocity nodes only on several ranks)
void schur_pp_numeric(ifem_ctx *ctx, const double *binv, double *out) {
  const PlanarCsr &Pt = tpp_pattern(ctx);
  const int64_t n = Pt.n_rows;
  if (n == 0) return;
  const int maxlen = (Pt.max_row + 1) & ~1;
  const size_t smem = size_t(4) * maxlen * (sizeof(double) + sizeof(int32_t));
  const unsigned blocks = unsigned((n + 3) / 4);
  KScope ks(ctx, IFEM_KC_TPP);
#define IFEM_TPP(D)                                                                                                     \
  hipLaunchKernelGGL((k_tpp_numeric<D>), dim3(blocks), dim3(256), smem, ctx->stream, n, maxlen, Pt.rowptr.p,            \
                     Pt.col.p, out, ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, ctx->Bt.rowptr.p,                      \
                     ctx->Bt.col.p, ctx->Bt.val.p, binv, ctx->Mp.rowptr.p, ctx->Mp.col.p, ctx->App.p, ctx->nUo)
  if (ctx->dim == 3) IFEM_TPP(3); else IFEM_TPP(2);
#undef IFEM_TPP
  IFEM_HIP_CHECK(hipGetLastError());
}

void tpp_numeric(ifem_ctx *ctx) {
  if (ctx->tpp_valid) return;
  PlanarCsr &Pt = tpp_pat(ctx);
  if (Pt.n_rows == 0) { if (ctx->halo.nranks > 1) build_schur_pattern_owned(ctx); else build_schur_pattern(ctx); }
  const int64_t n = Pt.n_rows;
  if (n == 0) return;
  if (ctx->Tpp.n != (size_t)Pt.nnzb) ctx->Tpp.alloc((size_t)Pt.nnzb);
  const int maxlen = (Pt.max_row + 1) & ~1;
  const size_t smem = size_t(4) * maxlen * (sizeof(double) + sizeof(int32_t));
  const unsigned blocks = unsigned((n + 3) / 4);
  KScope ks(ctx, IFEM_KC_TPP);
#define IFEM_TPP(D)                                                                                                     \
  hipLaunchKernelGGL((k_tpp_numeric<D>), dim3(blocks), dim3(256), smem, ctx->stream, n, maxlen, Pt.rowptr.p,            \
                     Pt.col.p, ctx->Tpp.p, ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, ctx->Bt.rowptr.p,               \
                     ctx->Bt.col.p, ctx->Bt.val.p, ctx->bjac.p, ctx->Mp.rowptr.p, ctx->Mp.col.p, ctx->App.p, ctx->nUo)
  if (ctx->dim == 3) IFEM_TPP(3); else IFEM_TPP(2);
#undef IFEM_TPP
  IFEM_HIP_CHECK(hipGetLastError());
  if (ctx->tpp_diag.n != (size_t)n) ctx->tpp_diag.alloc((size_t)n);
  scalar_diag(ctx, Pt, ctx->Tpp.p, ctx->tpp_diag.p);
  ctx->tpp_valid = true;
  ctx->tpp_ilu.factored = false;
}

void spmv_tpp(ifem_ctx *ctx, const double *xp, double *yp) {
  const PlanarCsr &Pt = tpp_pat(ctx);
  KScope ks(ctx, IFEM_KC_TPP, double(Pt.nnzb) * 12.0 + double(Pt.n_rows) * 24.0);
  if (Pt.n_rows == 0) return;
  spmv_planar_scalar(ctx, Pt, ctx->Tpp.p, xp, yp);
}


// ---- ILU(0) of T~ on its own pattern ---------------------------------------------------------------------------------
// Elimination order `ord` (a permutation position per row); "lower" entries of row i are the columns k with ord[k] <
// ord[i].  Per row the entry positions are listed sorted by the order of their column (ent), n_low of them lower, then
// the diagonal, then the upper ones: the factorisation walks the lower ones in elimination order, the triangular solves
// take either side.  Levels: rows of one level have no lower (forward) / upper (backward) entry in the same level.
struct IluHost {
  std::vector<int32_t> ent, n_low, diag, rows_f, rows_b;
  std::vector<int64_t> lvl_f, lvl_b; // level pointers into rows_f / rows_b
};

static void ilu_analyse(const std::vector<int64_t> &rp, const std::vector<int32_t> &col, int order_kind, IluHost &H) {
  const int64_t n = (int64_t)rp.size() - 1;
  std::vector<int32_t> ord((size_t)n);
  if (order_kind == 1) { // greedy colouring of the (structurally symmetric) matrix graph, rows ordered by (colour, index)
    std::vector<int32_t> colour((size_t)n, -1), mark;
    int32_t ncol = 0;
    for (int64_t i = 0; i < n; ++i) {
      mark.assign((size_t)ncol + 1, 0);
      for (int64_t k = rp[i]; k < rp[i + 1]; ++k) { const int32_t c = colour[col[k]]; if (c >= 0 && col[k] != i) mark[c] = 1; }
      int32_t c = 0;
      while (c < ncol && mark[c]) ++c;
      colour[i] = c;
      if (c == ncol) ++ncol;
    }
    std::vector<int32_t> idx((size_t)n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int32_t a, int32_t b) { return colour[a] < colour[b]; });
    for (int64_t p = 0; p < n; ++p) ord[idx[p]] = (int32_t)p;
  } else
    std::iota(ord.begin(), ord.end(), 0);
  H.ent.resize(col.size()); H.n_low.assign((size_t)n, 0); H.diag.assign((size_t)n, -1);
  std::vector<int32_t> lf((size_t)n, 0), lb((size_t)n, 0), by_ord((size_t)n);
  for (int64_t i = 0; i < n; ++i) by_ord[ord[i]] = (int32_t)i;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t rs = rp[i];
    const int len = int(rp[i + 1] - rs);
    int32_t *e = &H.ent[(size_t)rs];
    std::iota(e, e + len, 0);
    std::sort(e, e + len, [&](int32_t a, int32_t b) { return ord[col[rs + a]] < ord[col[rs + b]]; });
    int nl = 0;
    for (int t = 0; t < len; ++t) {
      const int32_t c = col[rs + e[t]];
      if (c == i) H.diag[i] = e[t];
      else if (ord[c] < ord[i]) ++nl;
    }
    if (H.diag[i] < 0) throw Error(IFEM_E_BADPARAM, "ILU(0) of T_pp: a row has no diagonal entry");
    H.n_low[i] = nl;
  }
  // forward levels in elimination order, backward levels in reverse
  int32_t nlf = 0, nlb = 0;
  for (int64_t p = 0; p < n; ++p) {
    const int32_t i = by_ord[p];
    int32_t l = 0;
    for (int t = 0; t < H.n_low[i]; ++t) l = std::max(l, lf[col[rp[i] + H.ent[(size_t)rp[i] + t]]] + 1);
    lf[i] = l; nlf = std::max(nlf, l + 1);
  }
  for (int64_t p = n - 1; p >= 0; --p) {
    const int32_t i = by_ord[p];
    const int len = int(rp[i + 1] - rp[i]);
    int32_t l = 0;
    for (int t = H.n_low[i] + 1; t < len; ++t) l = std::max(l, lb[col[rp[i] + H.ent[(size_t)rp[i] + t]]] + 1);
    lb[i] = l; nlb = std::max(nlb, l + 1);
  }
  auto bucket = [&](const std::vector<int32_t> &lv, int32_t nl, std::vector<int64_t> &ptr, std::vector<int32_t> &rows) {
    ptr.assign((size_t)nl + 1, 0);
    for (int64_t i = 0; i < n; ++i) ++ptr[(size_t)lv[i] + 1];
    for (int32_t l = 0; l < nl; ++l) ptr[(size_t)l + 1] += ptr[l];
    rows.resize((size_t)n);
    std::vector<int64_t> fill(ptr.begin(), ptr.end() - 1);
    for (int64_t i = 0; i < n; ++i) rows[(size_t)fill[lv[i]]++] = (int32_t)i;
  };
  bucket(lf, nlf, H.lvl_f, H.rows_f);
  bucket(lb, nlb, H.lvl_b, H.rows_b);
}

// factorisation of the rows of one level: one wave per row, the row's values in LDS.  For every lower entry (i, k) in
// elimination order: l_ik = a_ik / u_kk, then a_ij -= l_ik u_kj over the upper entries of row k that exist in row i.
__global__ __launch_bounds__(256) void k_ilu_factor(int64_t n_rows, const int32_t *__restrict__ rows, int maxlen,
                                                    const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                    const int32_t *__restrict__ ent, const int32_t *__restrict__ n_low,
                                                    const int32_t *__restrict__ diag, double *__restrict__ LU, double omega) {
  extern __shared__ __align__(16) unsigned char smem_i[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *w = reinterpret_cast<double *>(smem_i) + size_t(wave) * maxlen;
  const int64_t r = int64_t(blockIdx.x) * 4 + wave;
  if (r >= n_rows) return; // whole waves leave: no workgroup barrier below
  const int32_t i = rows[r];
  const int64_t rs = rp[i];
  const int len = int(rp[i + 1] - rs);
  for (int t = lane; t < len; t += 64) w[t] = LU[rs + t];
  wsync_lds();
  const int nl = n_low[i];
  const int di = diag[i];
  for (int t = 0; t < nl; ++t) {
    const int32_t e = ent[rs + t];
    const int32_t k = col[rs + e];
    const int64_t ks = rp[k];
    const int klen = int(rp[k + 1] - ks);
    const double lik = w[e] / LU[ks + diag[k]];
    double dropped = 0.0; // fill-in this lane drops in this step (relaxed MILU: it goes to the diagonal, below)
    for (int u = n_low[k] + 1 + lane; u < klen; u += 64) { // upper entries of row k (its final values: an earlier level)
      const int32_t e2 = ent[ks + u];
      const int32_t j = col[ks + e2];
      int lo = 0, hi = len - 1, p = -1; // columns of a row are sorted
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t cv = col[rs + mid];
        if (cv == j) { p = mid; break; }
        if (cv < j) lo = mid + 1; else hi = mid - 1;
      }
      if (p >= 0) w[p] -= lik * LU[ks + e2]; // distinct j, distinct p
      else dropped += lik * LU[ks + e2];
    }
    // one lane updates the diagonal after the wave has finished the step: no lane races the w[di] update of the j == i entry
    // above, and the sum has a fixed order (deterministic factors)
    if (omega != 0.0) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dropped += __shfl_xor(dropped, off);
    }
    wsync_lds();
    if (lane == 0) { w[e] = lik; if (omega != 0.0) w[di] -= omega * dropped; }
    wsync_lds();
  }
  for (int t = lane; t < len; t += 64) LU[rs + t] = w[t];
}

// one level of a triangular solve, 16 lanes per row.  forward: y_i = x_i - sum_lower l_ik y_k (unit diagonal);
// backward: y_i = (y_i - sum_upper u_ij y_j) / u_ii, in place
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_ilu_solve(int64_t n_rows, const int32_t *__restrict__ rows, const int64_t *__restrict__ rp,
                                                   const int32_t *__restrict__ col, const int32_t *__restrict__ ent,
                                                   const int32_t *__restrict__ n_low, const int32_t *__restrict__ diag,
                                                   const double *__restrict__ LU, const double *__restrict__ x, double *__restrict__ y) {
  const int64_t r = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
  const int lig = threadIdx.x & 15;
  const bool active = r < n_rows;
  const int32_t i = active ? rows[r] : 0;
  const int64_t rs = rp[i];
  const int len = int(rp[i + 1] - rs), nl = n_low[i];
  const int t0 = FORWARD ? 0 : nl + 1, t1 = FORWARD ? nl : len;
  double s = 0;
  if (active)
    for (int t = t0 + lig; t < t1; t += 16) {
      const int32_t e = ent[rs + t];
      s += LU[rs + e] * y[col[rs + e]];
    }
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
  if (active && lig == 0) y[i] = FORWARD ? x[i] - s : (y[i] - s) / LU[rs + diag[i]];
}

// Jacobi sweeps on a triangular system instead of its exact solution (iterative triangular solves, Anzt / Chow / Dongarra):
// forward  y <- x - (L - I) y_old,  backward  y <- (z - (U - D) y_old) / D, every row at once.  k sweeps reproduce the first k
// terms of the Neumann series of the triangular inverse: an approximate application of the same ILU(0) factors in 2 k
// launches whatever the number of levels.  A fixed k is a fixed linear operator: plain (non-flexible) GMRES may use it.
template <bool FORWARD>
__global__ __launch_bounds__(256) void k_ilu_jacobi_sweep(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                          const int32_t *__restrict__ ent, const int32_t *__restrict__ n_low,
                                                          const int32_t *__restrict__ diag, const double *__restrict__ LU,
                                                          const double *__restrict__ rhs, const double *__restrict__ y_old,
                                                          double *__restrict__ y_new) {
  const int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 4;
  const int lig = threadIdx.x & 15;
  const bool active = i < n;
  const int64_t rs = active ? rp[i] : 0;
  const int len = active ? int(rp[i + 1] - rs) : 0, nl = active ? n_low[i] : 0;
  const int t0 = FORWARD ? 0 : nl + 1, t1 = FORWARD ? nl : len;
  double s = 0;
  for (int t = t0 + lig; t < t1; t += 16) {
    const int32_t e = ent[rs + t];
    s += LU[rs + e] * y_old[col[rs + e]];
  }
  for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
  if (active && lig == 0) y_new[i] = FORWARD ? rhs[i] - s : (rhs[i] - s) / LU[rs + diag[i]];
}
__global__ void k_ilu_diag_scale(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ diag,
                                 const double *__restrict__ LU, const double *__restrict__ x, double *__restrict__ y) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i < n) y[i] = x[i] / LU[rp[i] + diag[i]];
}

// a run of consecutive SMALL levels in one launch: one workgroup of 64 row groups walks the levels with a workgroup barrier in
// between (natural-order ILU(0) of a 2D stencil has hundreds of levels of a few dozen rows: one launch per level is
// launch-bound).  y is read and written with agent-scope accesses: the rows of the previous level were written by other
// waves of this workgroup.
template <bool FORWARD>
__global__ __launch_bounds__(1024) void k_ilu_solve_batch(int l0, int l1, const int64_t *__restrict__ lvl, const int32_t *__restrict__ rows,
                                                          const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                          const int32_t *__restrict__ ent, const int32_t *__restrict__ n_low,
                                                          const int32_t *__restrict__ diag, const double *__restrict__ LU,
                                                          const double *__restrict__ x, double *y) {
  const int lig = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (int l = l0; l < l1; ++l) {
    const int64_t first = lvl[l], cnt = lvl[l + 1] - first;
    for (int64_t r = grp; r < cnt; r += 64) {
      const int32_t i = rows[first + r];
      const int64_t rs = rp[i];
      const int len = int(rp[i + 1] - rs), nl = n_low[i];
      const int t0 = FORWARD ? 0 : nl + 1, t1 = FORWARD ? nl : len;
      double s = 0;
      for (int t = t0 + lig; t < t1; t += 16) {
        const int32_t e = ent[rs + t];
        s += LU[rs + e] * __hip_atomic_load(&y[col[rs + e]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      for (int off = 8; off > 0; off >>= 1) s += __shfl_xor(s, off, 16);
      if (lig == 0) {
        const double v = FORWARD ? x[i] - s : (__hip_atomic_load(&y[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - s) / LU[rs + diag[i]];
        __hip_atomic_store(&y[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    // the stores above are agent-scope (write-through to L2), the loads of the next level agent-scope too (past the L1):
    // the workgroup barrier with its release / acquire orders them
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
  }
}

// after the factorisation: smallest and largest |pivot| and the number of non-finite entries of the factors
__global__ void k_ilu_check(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ diag, const double *__restrict__ LU,
                            unsigned long long *__restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long bad = 0;
  for (int64_t k = rp[i]; k < rp[i + 1]; ++k) bad += isfinite(LU[k]) ? 0 : 1;
  const double d = fabs(LU[rp[i] + diag[i]]);
  if (bad || !isfinite(d)) { atomicAdd(&out[2], bad ? bad : 1ull); return; }
  const unsigned long long bits = (unsigned long long)__double_as_longlong(d); // order-preserving for non-negative doubles
  atomicMin(&out[0], bits);
  atomicMax(&out[1], bits);
}

// launch plan of one triangular sweep: runs of consecutive levels with at most kBatchRows rows each go to the batch kernel
static constexpr int64_t kBatchRows = 256;
static void plan_sweep(const std::vector<int64_t> &lvl, std::vector<std::array<int32_t, 2>> &plan) {
  plan.clear();
  const int nl = (int)lvl.size() - 1;
  int l = 0;
  while (l < nl) {
    if (lvl[l + 1] - lvl[l] > kBatchRows) { plan.push_back({l, -1}); ++l; continue; } // a wide level: its own multi-workgroup launch
    int e = l;
    while (e < nl && lvl[e + 1] - lvl[e] <= kBatchRows) ++e;
    plan.push_back({l, e});
    l = e;
  }
}

bool tpp_ilu_factor(ifem_ctx *ctx) {
  KScope ks(ctx, IFEM_KC_TPP);
  TppIlu &I = ctx->tpp_ilu;
  const PlanarCsr &Pt = tpp_pat(ctx);
  const int64_t n = Pt.n_rows;
  if (n == 0) return true;
  if (I.factored) return !I.broken;
  hipStream_t s = ctx->stream;
  // 2 (default): natural order where its levels are wide enough to fill the device, multicolour otherwise (below)
  const int order_kind = ctx->tune.tpp_ilu_order >= 2 ? 2 : (ctx->tune.tpp_ilu_order == 1 ? 1 : 0);
  if (!I.analysed || I.order_kind != order_kind) { // once per pattern: the elimination DAG on the host
    std::vector<int64_t> rp((size_t)n + 1);
    std::vector<int32_t> col((size_t)Pt.nnzb);
    IFEM_HIP_CHECK(hipMemcpyAsync(rp.data(), Pt.rowptr.p, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipMemcpyAsync(col.data(), Pt.col.p, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
    IluHost H;
    ilu_analyse(rp, col, order_kind == 1 ? 1 : 0, H);
    if (order_kind == 2) {
      // The natural order preconditions better (cylinder refined once more: 49 against 154 inner iterations per application) but
      // its elimination DAG has O(n^(1/dim)) levels: below ~1000 rows per level the triangular sweeps are launch- and
      // barrier-bound and the multicolour order wins on the clock by 4-7 x (profiles/r03_tpp_ilu_sweep.txt: 0.17 / 0.64 / 4.5 s
      // against 0.69 / 4.1 / 17 s per solve at 36 k / 6 k / 24 k rows; profiles/r04_ilu_order.txt: 22 against 159 ms per time step
      // of tests/fluid_body_force_mpi)
      const int64_t levels = (int64_t)H.lvl_f.size() - 1;
      if (levels > 0 && n / levels < 1024) { H = IluHost(); ilu_analyse(rp, col, 1, H); I.order_used = 1; }
      else I.order_used = 0;
    } else I.order_used = order_kind;
    I.ent.upload(H.ent.data(), H.ent.size(), s);
    I.n_low.upload(H.n_low.data(), H.n_low.size(), s);
    I.diag.upload(H.diag.data(), H.diag.size(), s);
    I.rows_f.upload(H.rows_f.data(), H.rows_f.size(), s);
    I.rows_b.upload(H.rows_b.data(), H.rows_b.size(), s);
    I.d_lvl_f.upload(H.lvl_f.data(), H.lvl_f.size(), s);
    I.d_lvl_b.upload(H.lvl_b.data(), H.lvl_b.size(), s);
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
    I.lvl_f = H.lvl_f; I.lvl_b = H.lvl_b;
    plan_sweep(I.lvl_f, I.plan_f);
    plan_sweep(I.lvl_b, I.plan_b);
    I.analysed = true; I.order_kind = order_kind;
  }
  if (I.LU.n != ctx->Tpp.n) I.LU.alloc(ctx->Tpp.n);
  IFEM_HIP_CHECK(hipMemcpyAsync(I.LU.p, ctx->Tpp.p, ctx->Tpp.n * sizeof(double), hipMemcpyDeviceToDevice, s));
  const int maxlen = (Pt.max_row + 1) & ~1;
  const size_t smem = size_t(4) * maxlen * sizeof(double);
  const double omega = 1e-3 * std::min(std::max(ctx->tune.tpp_milu_permille, 0), 1000);
  for (size_t l = 0; l + 1 < I.lvl_f.size(); ++l) {
    const int64_t first = I.lvl_f[l], cnt = I.lvl_f[l + 1] - first;
    if (cnt <= 0) continue;
    hipLaunchKernelGGL(k_ilu_factor, dim3(unsigned((cnt + 3) / 4)), dim3(256), smem, s, cnt, I.rows_f.p + first, maxlen, Pt.rowptr.p,
                       Pt.col.p, I.ent.p, I.n_low.p, I.diag.p, I.LU.p, omega);
  }
  // breakdown check: the pivots divide in every later elimination step and in every triangular solve
  if (I.chk.n != 3) I.chk.alloc(3);
  const unsigned long long init[3] = {~0ull, 0ull, 0ull};
  IFEM_HIP_CHECK(hipMemcpyAsync(I.chk.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_ilu_check, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, Pt.rowptr.p, I.diag.p, I.LU.p, I.chk.p);
  unsigned long long res[3];
  IFEM_HIP_CHECK(hipMemcpyAsync(res, I.chk.p, sizeof(res), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_HIP_CHECK(hipGetLastError());
  double pmin, pmax;
  static_assert(sizeof(double) == sizeof(unsigned long long), "bit copy");
  std::memcpy(&pmin, &res[0], 8); std::memcpy(&pmax, &res[1], 8);
  I.pivot_min = res[2] ? 0.0 : pmin; I.pivot_max = res[2] ? 0.0 : pmax;
  I.broken = res[2] != 0 || !(pmin > 1e-14 * pmax);
  I.factored = true;
  return !I.broken;
}

// y = (LU)^-1 x
void tpp_ilu_apply(ifem_ctx *ctx, const double *x, double *y) {
  KScope ks(ctx, IFEM_KC_TPP, double(tpp_pat(ctx).nnzb) * 12.0 + double(tpp_pat(ctx).n_rows) * 32.0);
  TppIlu &I = ctx->tpp_ilu;
  const PlanarCsr &Pt = tpp_pat(ctx);
  hipStream_t s = ctx->stream;
  const int sweeps = ctx->tune.tpp_tri_sweeps;
  if (sweeps > 0) { // approximate triangular solves: 2 * sweeps row-parallel launches
    const int64_t n = Pt.n_rows;
    if ((int64_t)I.t0.n < n) { I.t0.alloc((size_t)n); I.t1.alloc((size_t)n); }
    const dim3 g(unsigned((n * 16 + 255) / 256)), b(256);
    // forward: start from y0 = x (the zeroth Neumann term), ping-pong between t0 and t1
    const double *cur = x;
    double *bufs[2] = {I.t0.p, I.t1.p};
    for (int k = 0; k < sweeps; ++k) {
      double *nxt = bufs[k & 1];
      hipLaunchKernelGGL((k_ilu_jacobi_sweep<true>), g, b, 0, s, n, Pt.rowptr.p, Pt.col.p, I.ent.p, I.n_low.p, I.diag.p, I.LU.p, x, cur, nxt);
      cur = nxt;
    }
    const double *z = cur; // L^-1 x (approximately): the right-hand side of the backward system, lives in one of the buffers
    // backward: y0 = D^-1 z into y, then sweeps ping-pong between y and the buffer that does not hold z
    hipLaunchKernelGGL(k_ilu_diag_scale, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, Pt.rowptr.p, I.diag.p, I.LU.p, z, y);
    double *other = (z == I.t0.p) ? I.t1.p : I.t0.p;
    const double *curb = y;
    for (int k = 0; k < sweeps; ++k) {
      double *nxt = (curb == y) ? other : y;
      hipLaunchKernelGGL((k_ilu_jacobi_sweep<false>), g, b, 0, s, n, Pt.rowptr.p, Pt.col.p, I.ent.p, I.n_low.p, I.diag.p, I.LU.p, z, curb, nxt);
      curb = nxt;
    }
    if (curb != y) IFEM_HIP_CHECK(hipMemcpyAsync(y, curb, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, s));
    return;
  }
  auto sweep = [&](bool forward) {
    const auto &plan = forward ? I.plan_f : I.plan_b;
    const auto &lvl = forward ? I.lvl_f : I.lvl_b;
    const int64_t *d_lvl = forward ? I.d_lvl_f.p : I.d_lvl_b.p;
    const int32_t *rows = forward ? I.rows_f.p : I.rows_b.p;
    for (const auto &st : plan) {
      if (st[1] < 0) {
        const int64_t first = lvl[st[0]], cnt = lvl[st[0] + 1] - first;
        if (forward)
          hipLaunchKernelGGL((k_ilu_solve<true>), dim3(unsigned((cnt * 16 + 255) / 256)), dim3(256), 0, s, cnt, rows + first, Pt.rowptr.p,
                             Pt.col.p, I.ent.p, I.n_low.p, I.diag.p, I.LU.p, x, y);
        else
          hipLaunchKernelGGL((k_ilu_solve<false>), dim3(unsigned((cnt * 16 + 255) / 256)), dim3(256), 0, s, cnt, rows + first, Pt.rowptr.p,
                             Pt.col.p, I.ent.p, I.n_low.p, I.diag.p, I.LU.p, x, y);
      } else if (forward)
        hipLaunchKernelGGL((k_ilu_solve_batch<true>), dim3(1), dim3(1024), 0, s, st[0], st[1], d_lvl, rows, Pt.rowptr.p, Pt.col.p,
                           I.ent.p, I.n_low.p, I.diag.p, I.LU.p, x, y);
      else
        hipLaunchKernelGGL((k_ilu_solve_batch<false>), dim3(1), dim3(1024), 0, s, st[0], st[1], d_lvl, rows, Pt.rowptr.p, Pt.col.p,
                           I.ent.p, I.n_low.p, I.diag.p, I.LU.p, x, y);
    }
  };
  sweep(true);
  sweep(false);
}
int tpp_ilu_levels(const ifem_ctx *ctx) { return (int)ctx->tpp_ilu.lvl_f.size() - 1; }

} // namespace ifem
