// bilu.hip -- ILU(0) with DIM x DIM blocks on the device, for the two Euclid factorisations of
// SUPGFluidSolver::BlockIncompSchurPreconditioner (mpi_supg_solver.cpp:49-53: Pvv_inverse = ILU(0) of A_vv; :133: B2pp_inverse =
// ILU(0) of B2pp = A_pp - A_pv rowsum(|A_vv|)^-1 A_vp; Euclid's default level 0, preconditioner_pilut.cpp:100,124-138).
//   * natural row order, as Euclid's serial sweep.  For A_vv the rows are velocity NODES with dim x dim blocks: the reference's
//     scalar ILU(0) runs on a pattern made of full dim x dim blocks (every velocity component couples to every other in the
//     DoFHandler's sparsity), and on such a pattern the scalar and the block factorisation are the same operator L U.
//   * several ranks: the factorisation of the owned x owned block on every rank (block-Jacobi ILU: the ordering Euclid's parallel
//     ILU uses across ranks is not reproducible, SURVEY 8c "parity unpinned").
//   * factorisation: level-scheduled along the elimination DAG -- one wavefront per block row, the row in LDS; runs of small levels
//     (a 2D stencil in natural order has hundreds of levels of a few dozen rows) are walked by ONE workgroup inside one launch.
//   * application: k Jacobi sweeps on each triangular system (y <- x - (L - I) y, y <- D^-1 (z - (U - D) y)) instead of
//     substitution: every sweep is one row-parallel launch whatever the number of levels, and the triangular factors of an ILU(0)
//     are so diagonally dominant that 4 / 6 sweeps reproduce the iteration counts of exact substitution to ~15 % (measured on the
//     cylinder benchmark at 3 and 4 refinements, profiles/r06_scns_pc_sweep.txt).  A fixed k is a fixed linear operator.
//     sweeps < 0: exact substitution, level by level in one workgroup (verification of the factors; slow).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <numeric>
#include <vector>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

namespace {

__device__ inline void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int DIM>
__device__ inline void block_inverse(const double *D, double *Di) {
  if constexpr (DIM == 1) Di[0] = 1.0 / D[0];
  else if constexpr (DIM == 2) {
    const double r = 1.0 / (D[0] * D[3] - D[1] * D[2]);
    Di[0] = D[3] * r; Di[1] = -D[1] * r; Di[2] = -D[2] * r; Di[3] = D[0] * r;
  } else {
    const double c00 = D[4] * D[8] - D[5] * D[7], c01 = D[5] * D[6] - D[3] * D[8], c02 = D[3] * D[7] - D[4] * D[6];
    const double r = 1.0 / (D[0] * c00 + D[1] * c01 + D[2] * c02);
    Di[0] = c00 * r; Di[3] = c01 * r; Di[6] = c02 * r;
    Di[1] = (D[2] * D[7] - D[1] * D[8]) * r; Di[4] = (D[0] * D[8] - D[2] * D[6]) * r; Di[7] = (D[1] * D[6] - D[0] * D[7]) * r;
    Di[2] = (D[1] * D[5] - D[2] * D[4]) * r; Di[5] = (D[2] * D[3] - D[0] * D[5]) * r; Di[8] = (D[0] * D[4] - D[1] * D[3]) * r;
  }
}

// LU[t][e] = src[srcpos[t]][e]: the factor's own (column-sorted, owned x owned) copy of the matrix
template <int BS>
__global__ void k_bilu_gather(int64_t nnz, const int64_t *__restrict__ srcpos, const double *__restrict__ src, double *__restrict__ LU) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= nnz * BS) return;
  const int64_t t = i / BS;
  const int e = int(i - t * BS);
  LU[i] = src[srcpos[t] * BS + e];
}

// one block row (wavefront-wide): w = the row in LDS.  For every lower entry (i, k) in column order: L_ik = W_ik U_kk^-1, then
// W_ij -= L_ik U_kj over the upper entries of row k that exist in row i; finally U_ii^-1.  COHERENT: the rows of earlier levels were
// written by other waves of the SAME launch (batch kernel): read them past the L1.
template <int DIM, bool COHERENT>
__device__ inline void bilu_factor_row(const int32_t i, double *w, const int lane, const int64_t *__restrict__ rp,
                                       const int32_t *__restrict__ col, const int32_t *__restrict__ diag, double *LU, double *dinv) {
  constexpr int BS = DIM * DIM;
  const int64_t rs = rp[i];
  const int len = int(rp[i + 1] - rs);
  auto ld = [&](const double *p) -> double {
    if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
  };
  for (int t = lane; t < len * BS; t += 64) w[t] = LU[rs * BS + t]; // this row: written by the gather kernel before the launch
  wave_lds_sync();
  const int di = diag[i];
  for (int t = 0; t < di; ++t) { // lower entries: columns sorted, so t < diag position
    const int32_t k = col[rs + t];
    const int64_t ks = rp[k];
    const int klen = int(rp[k + 1] - ks), kd = diag[k];
    // L_ik = W_ik * Uinv_kk  (every lane computes it: BS <= 9 products on LDS / L2 values it needs anyway)
    double Lik[BS];
    {
      double Ui[BS], Wt[BS];
#pragma unroll
      for (int e = 0; e < BS; ++e) { Ui[e] = ld(&dinv[int64_t(k) * BS + e]); Wt[e] = w[t * BS + e]; }
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double s = 0;
#pragma unroll
          for (int m = 0; m < DIM; ++m) s += Wt[r * DIM + m] * Ui[m * DIM + c];
          Lik[r * DIM + c] = s;
        }
    }
    for (int u = kd + 1 + lane; u < klen; u += 64) { // upper entries of row k (final values: an earlier level)
      const int32_t j = col[ks + u];
      int lo = 0, hi = len - 1, p = -1; // columns of a row are sorted
      while (lo <= hi) {
        const int mid = (lo + hi) >> 1;
        const int32_t cv = col[rs + mid];
        if (cv == j) { p = mid; break; }
        if (cv < j) lo = mid + 1; else hi = mid - 1;
      }
      if (p < 0) continue; // ILU(0): fill outside the pattern is dropped
      double Uk[BS];
#pragma unroll
      for (int e = 0; e < BS; ++e) Uk[e] = ld(&LU[(ks + u) * BS + e]);
#pragma unroll
      for (int r = 0; r < DIM; ++r)
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          double s = 0;
#pragma unroll
          for (int m = 0; m < DIM; ++m) s += Lik[r * DIM + m] * Uk[m * DIM + c];
          w[p * BS + r * DIM + c] -= s; // distinct j, distinct p (p > t: upper entries of row k have j > k)
        }
    }
    wave_lds_sync();
    if (lane < BS) w[t * BS + lane] = Lik[lane];
    wave_lds_sync();
  }
  if (lane == 0) {
    double D[BS], Di[BS];
#pragma unroll
    for (int e = 0; e < BS; ++e) D[e] = w[di * BS + e];
    block_inverse<DIM>(D, Di);
#pragma unroll
    for (int e = 0; e < BS; ++e) {
      if constexpr (COHERENT) __hip_atomic_store(&dinv[int64_t(i) * BS + e], Di[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else dinv[int64_t(i) * BS + e] = Di[e];
    }
  }
  for (int t = lane; t < len * BS; t += 64) {
    if constexpr (COHERENT) __hip_atomic_store(&LU[rs * BS + t], w[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else LU[rs * BS + t] = w[t];
  }
}

// one wide level: one wavefront per row
template <int DIM>
__global__ __launch_bounds__(256) void k_bilu_factor(int64_t n_rows, const int32_t *__restrict__ rows, int maxlen,
                                                     const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                     const int32_t *__restrict__ diag, double *LU, double *dinv) {
  extern __shared__ __align__(16) unsigned char smem_b[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *w = reinterpret_cast<double *>(smem_b) + size_t(wave) * maxlen * DIM * DIM;
  const int64_t r = int64_t(blockIdx.x) * 4 + wave;
  if (r >= n_rows) return; // whole waves leave: no workgroup barrier below
  bilu_factor_row<DIM, false>(rows[r], w, lane, rp, col, diag, LU, dinv);
}

// a run of consecutive small levels in one launch: the 16 waves of one workgroup take the rows of a level, a workgroup barrier
// (with agent-scope release / acquire: the factors travel through the L2) separates the levels
template <int DIM>
__global__ __launch_bounds__(1024) void k_bilu_factor_batch(int l0, int l1, const int64_t *__restrict__ lvl, const int32_t *__restrict__ rows,
                                                            int maxlen, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                            const int32_t *__restrict__ diag, double *LU, double *dinv) {
  extern __shared__ __align__(16) unsigned char smem_b[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double *w = reinterpret_cast<double *>(smem_b) + size_t(wave) * maxlen * DIM * DIM;
  for (int l = l0; l < l1; ++l) {
    const int64_t first = lvl[l], cnt = lvl[l + 1] - first;
    for (int64_t r = wave; r < cnt; r += 16) {
      bilu_factor_row<DIM, true>(rows[first + r], w, lane, rp, col, diag, LU, dinv);
      wave_lds_sync(); // w is reused by this wave's next row
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
    __syncthreads();
  }
}

// one Jacobi sweep on a triangular system, G lanes per block row.
//   FORWARD : y_new_i = rhs_i - sum_{j < i} L_ij y_old_j                          (unit block diagonal)
//             with y_scaled != NULL also y_scaled_i = U_ii^-1 y_new_i  (the start vector of the backward sweeps, fused)
//   BACKWARD: y_new_i = U_ii^-1 (rhs_i - sum_{j > i} U_ij y_old_j)
template <int DIM, bool FORWARD, int G>
__global__ __launch_bounds__(256) void k_bilu_sweep(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                    const int32_t *__restrict__ diag, const double *__restrict__ LU,
                                                    const double *__restrict__ dinv, const double *__restrict__ rhs,
                                                    const double *__restrict__ y_old, double *__restrict__ y_new,
                                                    double *__restrict__ y_scaled) {
  constexpr int BS = DIM * DIM;
  const int64_t i = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const int lig = threadIdx.x & (G - 1);
  const bool active = i < n;
  const int64_t rs = active ? rp[i] : 0;
  const int len = active ? int(rp[i + 1] - rs) : 0, di = active ? diag[i] : 0;
  const int t0 = FORWARD ? 0 : di + 1, t1 = FORWARD ? di : len;
  double s[DIM];
#pragma unroll
  for (int r = 0; r < DIM; ++r) s[r] = 0;
  for (int t = t0 + lig; t < t1; t += G) {
    const int32_t j = col[rs + t];
    double yv[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) yv[c] = y_old[int64_t(j) * DIM + c];
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int c = 0; c < DIM; ++c) s[r] += LU[(rs + t) * BS + r * DIM + c] * yv[c];
  }
#pragma unroll
  for (int r = 0; r < DIM; ++r)
    for (int off = G / 2; off > 0; off >>= 1) s[r] += __shfl_xor(s[r], off, G);
  if (active && lig == 0) {
    double v[DIM];
#pragma unroll
    for (int r = 0; r < DIM; ++r) v[r] = rhs[i * DIM + r] - s[r];
    if (FORWARD) {
#pragma unroll
      for (int r = 0; r < DIM; ++r) y_new[i * DIM + r] = v[r];
    }
    if (!FORWARD || y_scaled) {
      double *out = FORWARD ? y_scaled : y_new;
#pragma unroll
      for (int r = 0; r < DIM; ++r) {
        double q = 0;
#pragma unroll
        for (int c = 0; c < DIM; ++c) q += dinv[i * BS + r * DIM + c] * v[c];
        out[i * DIM + r] = q;
      }
    }
  }
}

// exact substitution, every level in ONE workgroup (verification: the Jacobi sweeps converge to this)
template <int DIM, bool FORWARD>
__global__ __launch_bounds__(1024) void k_bilu_solve_exact(int n_levels, const int64_t *__restrict__ lvl, const int32_t *__restrict__ rows,
                                                           const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                                                           const int32_t *__restrict__ diag, const double *__restrict__ LU,
                                                           const double *__restrict__ dinv, const double *__restrict__ x, double *y) {
  constexpr int BS = DIM * DIM;
  const int lig = threadIdx.x & 15, grp = threadIdx.x >> 4;
  for (int l = 0; l < n_levels; ++l) {
    const int64_t first = lvl[l], cnt = lvl[l + 1] - first;
    for (int64_t r = grp; r < cnt; r += 64) {
      const int32_t i = rows[first + r];
      const int64_t rs = rp[i];
      const int len = int(rp[i + 1] - rs), di = diag[i];
      const int t0 = FORWARD ? 0 : di + 1, t1 = FORWARD ? di : len;
      double s[DIM];
#pragma unroll
      for (int q = 0; q < DIM; ++q) s[q] = 0;
      for (int t = t0 + lig; t < t1; t += 16) {
        const int32_t j = col[rs + t];
#pragma unroll
        for (int q = 0; q < DIM; ++q)
#pragma unroll
          for (int c = 0; c < DIM; ++c)
            s[q] += LU[(rs + t) * BS + q * DIM + c] * __hip_atomic_load(&y[int64_t(j) * DIM + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int q = 0; q < DIM; ++q)
        for (int off = 8; off > 0; off >>= 1) s[q] += __shfl_xor(s[q], off, 16);
      if (lig == 0) {
        double v[DIM];
#pragma unroll
        for (int q = 0; q < DIM; ++q)
          v[q] = (FORWARD ? x[int64_t(i) * DIM + q] : __hip_atomic_load(&y[int64_t(i) * DIM + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - s[q];
#pragma unroll
        for (int q = 0; q < DIM; ++q) {
          double o = v[q];
          if (!FORWARD) {
            o = 0;
#pragma unroll
            for (int c = 0; c < DIM; ++c) o += dinv[int64_t(i) * BS + q * DIM + c] * v[c];
          }
          __hip_atomic_store(&y[int64_t(i) * DIM + q], o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
    __syncthreads();
  }
}

// after the factorisation: non-finite entries of the factors / inverse pivots, largest |entry| of the inverse pivot blocks
template <int BS>
__global__ void k_bilu_check(int64_t n, const int64_t *__restrict__ rp, const double *__restrict__ LU, const double *__restrict__ dinv,
                             unsigned long long *__restrict__ out) {
  const int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long bad = 0;
  for (int64_t k = rp[i] * BS; k < rp[i + 1] * BS; ++k) bad += isfinite(LU[k]) ? 0 : 1;
  double m = 0;
  for (int e = 0; e < BS; ++e) { const double d = dinv[i * BS + e]; bad += isfinite(d) ? 0 : 1; m = fmax(m, fabs(d)); }
  if (bad) { atomicAdd(&out[2], bad); return; }
  const unsigned long long bits = (unsigned long long)__double_as_longlong(m); // order-preserving for non-negative doubles
  atomicMin(&out[0], bits);
  atomicMax(&out[1], bits);
}

constexpr int64_t kBatchRows = 64; // levels up to this many rows are walked by the single-workgroup kernel

template <int DIM>
void factor_t(ifem_ctx *ctx, BIlu &I, const double *src) {
  constexpr int BS = DIM * DIM;
  hipStream_t s = ctx->stream;
  const int64_t n = I.n;
  hipLaunchKernelGGL((k_bilu_gather<BS>), dim3(unsigned((I.nnz * BS + 255) / 256)), dim3(256), 0, s, I.nnz, I.srcpos.p, src, I.LU.p);
  const int maxlen = (I.max_row + 1) & ~1;
  const size_t smem1 = size_t(4) * maxlen * BS * sizeof(double), smem16 = size_t(16) * maxlen * BS * sizeof(double);
  const bool batch_ok = smem16 <= 64 * 1024;
  const int nl = (int)I.lvl_f.size() - 1;
  int l = 0;
  while (l < nl) {
    const int64_t first = I.lvl_f[l], cnt = I.lvl_f[l + 1] - first;
    if (batch_ok && cnt <= kBatchRows) {
      int e = l;
      while (e < nl && I.lvl_f[e + 1] - I.lvl_f[e] <= kBatchRows) ++e;
      hipLaunchKernelGGL((k_bilu_factor_batch<DIM>), dim3(1), dim3(1024), smem16, s, l, e, I.d_lvl_f.p, I.rows_f.p, maxlen, I.rp.p, I.col.p,
                         I.diag.p, I.LU.p, I.dinv.p);
      l = e;
    } else {
      if (cnt > 0)
        hipLaunchKernelGGL((k_bilu_factor<DIM>), dim3(unsigned((cnt + 3) / 4)), dim3(256), smem1, s, cnt, I.rows_f.p + first, maxlen, I.rp.p,
                           I.col.p, I.diag.p, I.LU.p, I.dinv.p);
      ++l;
    }
  }
  if (I.chk.n != 3) I.chk.alloc(3);
  const unsigned long long init[3] = {~0ull, 0ull, 0ull};
  IFEM_HIP_CHECK(hipMemcpyAsync(I.chk.p, init, sizeof(init), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL((k_bilu_check<BS>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, I.rp.p, I.LU.p, I.dinv.p, I.chk.p);
}

template <int DIM>
void apply_t(ifem_ctx *ctx, BIlu &I, int sweeps, const double *x, double *y) {
  hipStream_t s = ctx->stream;
  const int64_t n = I.n;
  if (sweeps < 0) { // exact substitution
    const int nlf = (int)I.lvl_f.size() - 1, nlb = (int)I.lvl_b.size() - 1;
    hipLaunchKernelGGL((k_bilu_solve_exact<DIM, true>), dim3(1), dim3(1024), 0, s, nlf, I.d_lvl_f.p, I.rows_f.p, I.rp.p, I.col.p, I.diag.p,
                       I.LU.p, I.dinv.p, x, y);
    hipLaunchKernelGGL((k_bilu_solve_exact<DIM, false>), dim3(1), dim3(1024), 0, s, nlb, I.d_lvl_b.p, I.rows_b.p, I.rp.p, I.col.p, I.diag.p,
                       I.LU.p, I.dinv.p, x, y);
    return;
  }
  constexpr int G = DIM == 1 ? 16 : 8;
  const size_t nv = (size_t)n * DIM;
  if (I.t0.n < nv) { I.t0.alloc(nv); I.t1.alloc(nv); I.t2.alloc(nv); }
  const dim3 g(unsigned((n * G + 255) / 256)), b(256);
  const int k = std::max(1, sweeps);
  // forward from y0 = x; the last sweep also writes the start vector of the backward sweeps, D^-1 z
  const double *cur = x;
  double *fb[2] = {I.t0.p, I.t1.p};
  double *z = nullptr, *y0 = I.t2.p;
  for (int q = 0; q < k; ++q) {
    double *nxt = fb[q & 1];
    hipLaunchKernelGGL((k_bilu_sweep<DIM, true, G>), g, b, 0, s, n, I.rp.p, I.col.p, I.diag.p, I.LU.p, I.dinv.p, x, cur, nxt,
                       q == k - 1 ? y0 : nullptr);
    cur = nxt; z = nxt;
  }
  // backward: ping-pong between y0's buffer and the forward buffer that does not hold z; the last sweep writes y
  double *other = (z == I.t0.p) ? I.t1.p : I.t0.p;
  const double *curb = y0;
  for (int q = 0; q < k; ++q) {
    double *nxt = q == k - 1 ? y : (curb == y0 ? other : y0);
    hipLaunchKernelGGL((k_bilu_sweep<DIM, false, G>), g, b, 0, s, n, I.rp.p, I.col.p, I.diag.p, I.LU.p, I.dinv.p, z, curb, nxt, nullptr);
    curb = nxt;
  }
}

} // namespace

// once per pattern: the factor's own CSR = rows [0, n) of (rp_dev, col_dev) restricted to the columns < n, sorted by column;
// srcpos = where each kept entry sits in the source value array; elimination levels of the natural order
void bilu_analyse(ifem_ctx *ctx, BIlu &I, int dim, int64_t n, const int64_t *rp_dev, const int32_t *col_dev) {
  hipStream_t s = ctx->stream;
  std::vector<int64_t> rp((size_t)n + 1);
  IFEM_HIP_CHECK(hipMemcpyAsync(rp.data(), rp_dev, rp.size() * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<int32_t> col((size_t)rp[n]);
  IFEM_HIP_CHECK(hipMemcpyAsync(col.data(), col_dev, col.size() * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  std::vector<int64_t> rp2((size_t)n + 1, 0), src;
  std::vector<int32_t> col2, diag((size_t)n, -1);
  src.reserve(col.size()); col2.reserve(col.size());
  std::vector<std::pair<int32_t, int64_t>> row;
  int max_row = 0;
  for (int64_t i = 0; i < n; ++i) {
    row.clear();
    for (int64_t k = rp[i]; k < rp[i + 1]; ++k)
      if (col[k] < n) row.emplace_back(col[k], k);
    std::sort(row.begin(), row.end());
    for (size_t t = 0; t < row.size(); ++t) {
      if (row[t].first == i) diag[i] = (int32_t)t;
      col2.push_back(row[t].first); src.push_back(row[t].second);
    }
    if (diag[i] < 0) throw Error(IFEM_E_BADPARAM, "ILU(0): a row has no diagonal entry");
    rp2[i + 1] = (int64_t)col2.size();
    max_row = std::max(max_row, (int)row.size());
  }
  std::vector<int32_t> lf((size_t)n, 0), lb((size_t)n, 0);
  int32_t nlf = 0, nlb = 0;
  for (int64_t i = 0; i < n; ++i) {
    int32_t l = 0;
    for (int64_t k = rp2[i]; k < rp2[i] + diag[i]; ++k) l = std::max(l, lf[col2[k]] + 1);
    lf[i] = l; nlf = std::max(nlf, l + 1);
  }
  for (int64_t i = n - 1; i >= 0; --i) {
    int32_t l = 0;
    for (int64_t k = rp2[i] + diag[i] + 1; k < rp2[i + 1]; ++k) l = std::max(l, lb[col2[k]] + 1);
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
  std::vector<int32_t> rows_f, rows_b;
  bucket(lf, nlf, I.lvl_f, rows_f);
  bucket(lb, nlb, I.lvl_b, rows_b);
  I.dim = dim; I.n = n; I.nnz = (int64_t)col2.size(); I.max_row = max_row;
  I.rp.upload(rp2.data(), rp2.size(), s);
  I.col.upload(col2.data(), col2.size(), s);
  I.diag.upload(diag.data(), diag.size(), s);
  I.srcpos.upload(src.data(), src.size(), s);
  I.rows_f.upload(rows_f.data(), rows_f.size(), s);
  I.rows_b.upload(rows_b.data(), rows_b.size(), s);
  I.d_lvl_f.upload(I.lvl_f.data(), I.lvl_f.size(), s);
  I.d_lvl_b.upload(I.lvl_b.data(), I.lvl_b.size(), s);
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  const size_t bs = (size_t)dim * dim;
  I.LU.alloc((size_t)I.nnz * bs);
  I.dinv.alloc((size_t)n * bs);
  I.analysed = true; I.factored = false;
}

// numeric factorisation from the source value array (blocks of dim*dim contiguous doubles at srcpos); false: broke down
bool bilu_factor(ifem_ctx *ctx, BIlu &I, const double *src) {
  KScope ks(ctx, IFEM_KC_TPP, double(I.nnz) * I.dim * I.dim * 24.0);
  if (I.n == 0) { I.factored = true; I.broken = false; return true; }
  if (I.dim == 1) factor_t<1>(ctx, I, src);
  else if (I.dim == 2) factor_t<2>(ctx, I, src);
  else factor_t<3>(ctx, I, src);
  unsigned long long res[3];
  IFEM_HIP_CHECK(hipMemcpyAsync(res, I.chk.p, sizeof(res), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_HIP_CHECK(hipGetLastError());
  double imin, imax;
  std::memcpy(&imin, &res[0], 8); std::memcpy(&imax, &res[1], 8);
  // pivots as 1 / (largest entry of the inverse pivot block)
  I.pivot_min = res[2] || !(imax > 0) ? 0.0 : 1.0 / imax;
  I.pivot_max = res[2] || !(imin > 0) ? 0.0 : 1.0 / imin;
  I.broken = res[2] != 0 || !(I.pivot_min > 1e-14 * I.pivot_max);
  I.factored = true;
  return !I.broken;
}

// y = (L U)^-1 x on compact vectors [n][dim]; sweeps > 0: Jacobi sweeps per triangular system, < 0: exact substitution
void bilu_apply(ifem_ctx *ctx, BIlu &I, int sweeps, const double *x, double *y) {
  if (I.n == 0) return;
  const int k = sweeps < 0 ? 2 : 2 * std::max(1, sweeps);
  KScope ks(ctx, IFEM_KC_TPP, double(k) * 0.5 * (double(I.nnz) * (I.dim * I.dim * 8.0 + 4.0) + double(I.n) * I.dim * 24.0));
  if (I.dim == 1) apply_t<1>(ctx, I, sweeps, x, y);
  else if (I.dim == 2) apply_t<2>(ctx, I, sweeps, x, y);
  else apply_t<3>(ctx, I, sweeps, x, y);
}

// ReverseRowSum as diagonal blocks (mpi_supg_solver.cpp:62-124): out[k] = diag_c( 1 / sum_j sum_d |A_vv[(k,c),(j,d)]| ), the shape
// k_tpp_numeric reads its node-block inverses in -- so the same kernel forms A_pp - A_pv rowsum(|A_vv|)^-1 A_vp
template <int DIM>
__global__ void k_rowsum_abs_inv(int64_t n_rows, const int64_t *__restrict__ rp, const double *__restrict__ val, double *__restrict__ out) {
  constexpr int BS = DIM * DIM;
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 3;
  const int lig = threadIdx.x & 7;
  const bool active = row < n_rows;
  const int64_t rs = active ? rp[row] : 0;
  const int len = active ? int(rp[row + 1] - rs) : 0;
  double s[DIM];
#pragma unroll
  for (int r = 0; r < DIM; ++r) s[r] = 0;
  for (int k = lig; k < len; k += 8)
#pragma unroll
    for (int r = 0; r < DIM; ++r)
#pragma unroll
      for (int c = 0; c < DIM; ++c) s[r] += fabs(val[uu_base(rs, len, k, BS) + int64_t(r * DIM + c) * uu_estride(len)]);
#pragma unroll
  for (int r = 0; r < DIM; ++r)
    for (int off = 4; off > 0; off >>= 1) s[r] += __shfl_xor(s[r], off, 8);
  if (active && lig == 0)
#pragma unroll
    for (int e = 0; e < BS; ++e) out[row * BS + e] = (e / DIM == e % DIM) ? 1.0 / s[e / DIM] : 0.0;
}

void rowsum_abs_inv(ifem_ctx *ctx, double *out) {
  const int64_t n = ctx->nUo;
  if (!n) return;
  KScope ks(ctx, IFEM_KC_TPP, double(ctx->Auu.val.n) * 8.0);
  const dim3 g(unsigned((n * 8 + 255) / 256)), b(256);
  if (ctx->dim == 3) hipLaunchKernelGGL((k_rowsum_abs_inv<3>), g, b, 0, ctx->stream, n, ctx->Auu.rowptr.p, ctx->Auu.val.p, out);
  else hipLaunchKernelGGL((k_rowsum_abs_inv<2>), g, b, 0, ctx->stream, n, ctx->Auu.rowptr.p, ctx->Auu.val.p, out);
}

} // namespace ifem
