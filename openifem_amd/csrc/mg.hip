// mg.hip -- device kernels of the geometric multigrid used inside the block preconditioner (solver.hip holds the
// V-cycle itself): CSR transfers between neighbouring levels (ifem_mg_attach), the fused vector updates of the
// Chebyshev-Jacobi smoother, small helpers of its set-up.  The reference has no counterpart: its CG(S_m) is
// unpreconditioned (mpi_insim.cpp:86-112) and its A~^-1 is MUMPS (:124-127).
#include <hip/hip_runtime.h>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

static inline unsigned mgrid(int64_t n) {
  const int64_t g = (n + 255) / 256;
  return unsigned(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

// y_row (+)= sum_k w_k x[col_k]: one thread per row (<= 8 entries for a prolongation row, <= 27 for a restriction row)
template <bool ADD>
__global__ __launch_bounds__(256) void k_mg_csr(int64_t n_rows, const int64_t *__restrict__ ptr, const int32_t *__restrict__ col,
                                                const double *__restrict__ w, const double *__restrict__ x,
                                                double *__restrict__ y) {
  for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n_rows; r += int64_t(gridDim.x) * blockDim.x) {
    double s = 0;
    for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) s += w[k] * x[col[k]];
    if (ADD) y[r] += s; else y[r] = s;
  }
}

void mg_csr_apply(ifem_ctx *ctx, const MgCsr &M, const double *x, double *y, bool add) {
  if (!M.n_rows) return;
  KScope ks(ctx, IFEM_KC_MG_TRANSFER, double(M.col.n) * 20.0 + double(M.n_rows) * 16.0);
  if (add) hipLaunchKernelGGL((k_mg_csr<true>), dim3(mgrid(M.n_rows)), dim3(256), 0, ctx->stream, M.n_rows, M.ptr.p, M.col.p, M.w.p, x, y);
  else hipLaunchKernelGGL((k_mg_csr<false>), dim3(mgrid(M.n_rows)), dim3(256), 0, ctx->stream, M.n_rows, M.ptr.p, M.col.p, M.w.p, x, y);
}

// Chebyshev iteration on D^-1 A (Adams, Brezina, Hu, Tuminaro 2003): d_0 = (1/theta) D^-1 r
__global__ void k_cheb_init(int64_t n, double c0, const double *__restrict__ dinv, const double *__restrict__ r,
                            double *__restrict__ d) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) d[i] = c0 * dinv[i] * r[i];
}
// x += d; r -= t (t = A d); d = a d + b D^-1 r
__global__ void k_cheb_step(int64_t n, double a, double b, const double *__restrict__ dinv, const double *__restrict__ t,
                            double *__restrict__ x, double *__restrict__ r, double *__restrict__ d) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double di = d[i], ri = r[i] - t[i];
    x[i] += di;
    r[i] = ri;
    d[i] = a * di + b * dinv[i] * ri;
  }
}
void cheb_init(ifem_ctx *ctx, int64_t n, double c0, const double *dinv, const double *r, double *d) {
  KScope ks(ctx, IFEM_KC_VECTOR, 24.0 * double(n));
  if (n) hipLaunchKernelGGL(k_cheb_init, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, c0, dinv, r, d);
}
void cheb_step(ifem_ctx *ctx, int64_t n, double a, double b, const double *dinv, const double *t, double *x, double *r, double *d) {
  KScope ks(ctx, IFEM_KC_VECTOR, 64.0 * double(n));
  if (n) hipLaunchKernelGGL(k_cheb_step, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, a, b, dinv, t, x, r, d);
}

__global__ void k_vec_recip(int64_t n, double *__restrict__ d) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    const double v = d[i];
    d[i] = v != 0.0 ? 1.0 / v : 1.0;
  }
}
void vec_recip(ifem_ctx *ctx, int64_t n, double *d) {
  if (n) hipLaunchKernelGGL(k_vec_recip, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, d);
}

// deterministic rough start vector of the power iteration (the same on every run; `offset` decorrelates the ranks)
__global__ void k_vec_rough(int64_t n, int64_t offset, double *__restrict__ x) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    uint64_t h = uint64_t(i + offset) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    x[i] = double(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
  }
}
void vec_rough(ifem_ctx *ctx, int64_t n, int64_t offset, double *x) {
  if (n) hipLaunchKernelGGL(k_vec_rough, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, offset, x);
}

} // namespace ifem

// ---------------------------------------------------------------------------------------------------------------------
// Node-block diagonal of A_uu without the assembled matrix: the dim x dim blocks Ke[(a,.),(a,.)] of every cell summed per
// node, with the constraint rule of the assembly (a constrained dof keeps |Ke_rr| on its diagonal, its row and
// column vanish), then inverted: the block-Jacobi data of a COARSE multigrid level, where nothing is assembled.  Same
// integrand as apply_mf.hip / assemble*.hip at the evaluation point ctx->mf_eval.  One wavefront per cell.
#include "assemble_common.hpp"

namespace ifem {

struct DiagArgs {
  int64_t n_cells, nUo;
  const double *vcoords;
  const int32_t *cell_unodes;
  const uint8_t *is_c;
  const double *eval;
  double *out; // [nUo][DIM*DIM], zeroed
  double mu, rho, gamma, inv_dt;
  int conv;
  Tab1D t;
};

// Half a wavefront per cell (8 cells per block), 1D tables in LDS, every loop over nodes / points / components unrolled (the
// node or point index of the inner loops is a compile-time constant, the lane's own index selects the LDS table row).
// R: arithmetic type of the integrals (float on the levels of the V-cycle: the smoother applies the inverse blocks in single precision
// anyway, and the kernel is bound by its ~65 kFLOP per cell); the per-node sums are accumulated in double
template <int DIM, int KV, typename R>
__global__ __launch_bounds__(256) void k_uu_diag(DiagArgs A) {
  constexpr int N1 = KV + 1, NN = (DIM == 2) ? N1 * N1 : N1 * N1 * N1, NV = 1 << DIM, CPB = 8;
  __shared__ R sX[CPB][NV * DIM], sJi[CPB][NN * DIM * DIM], sW[CPB][NN], sU[CPB][NN * DIM], sGu[CPB][NN * DIM * DIM], sE[CPB][NN * DIM];
  __shared__ R tN[9], tD[9];
  __shared__ int32_t sNode[CPB][NN];
  const int slot = threadIdx.x >> 5, hl = threadIdx.x & 31;
  const int64_t cell = int64_t(blockIdx.x) * CPB + slot;
  const bool active = cell < A.n_cells;
  const int64_t cc = active ? cell : 0;
  if (threadIdx.x < 9) { tN[threadIdx.x] = R(A.t.N[threadIdx.x]); tD[threadIdx.x] = R(A.t.dN[threadIdx.x]); }
  if (hl < NN) {
    const int32_t nd = A.cell_unodes[cc * NN + hl];
    sNode[slot][hl] = nd;
#pragma unroll
    for (int c = 0; c < DIM; ++c) sE[slot][hl * DIM + c] = A.conv ? R(A.eval[int64_t(DIM) * nd + c]) : R(0);
  }
  if (hl < NV * DIM) sX[slot][hl] = R(A.vcoords[cc * NV * DIM + hl]);
  __syncthreads();
  const int li = hl < NN ? hl : 0; // my point (first stage) / my node (second stage)
  const int l0 = li % N1, l1 = (li / N1) % N1, l2 = DIM == 3 ? li / (N1 * N1) : 0;
  if (hl < NN) { // lane = quadrature point: Jacobian of the d-linear map, fields of the evaluation point
    const int q = hl;
    const int qi[3] = {l0, l1, l2};
    R L[3][2], J[DIM * DIM], Ji[DIM * DIM], wq = 1;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      R x_ = R(A.t.xi[0]), w_ = R(A.t.w[0]);
#pragma unroll
      for (int k = 1; k < N1; ++k) { x_ = qi[d] == k ? R(A.t.xi[k]) : x_; w_ = qi[d] == k ? R(A.t.w[k]) : w_; }
      L[d][1] = x_; L[d][0] = R(1) - x_; wq *= w_;
    }
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) J[i] = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const int b[3] = {v & 1, (v >> 1) & 1, (v >> 2) & 1};
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        R g = b[d] ? R(1) : R(-1);
#pragma unroll
        for (int o = 0; o < DIM; ++o) if (o != d) g *= L[o][b[o]];
#pragma unroll
        for (int e = 0; e < DIM; ++e) J[e * DIM + d] += sX[slot][v * DIM + e] * g;
      }
    }
    const R det = inv_small<DIM, R>(J, Ji);
    sW[slot][q] = R(fabs(det)) * wq;
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) sJi[slot][q * DIM * DIM + i] = Ji[i];
    R u[DIM], gr[DIM * DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) u[c] = 0;
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) gr[i] = 0;
    // node loop: x index unrolled, the two slower indices walk the LDS tables
#pragma unroll 1
    for (int a21 = 0; a21 < NN / N1; ++a21) {
      const int a1 = a21 % N1, a2 = a21 / N1;
      const R n1v = tN[l1 * N1 + a1], d1v = tD[l1 * N1 + a1];
      const R n2v = DIM == 3 ? tN[l2 * N1 + a2] : R(1), d2v = DIM == 3 ? tD[l2 * N1 + a2] : R(0);
#pragma unroll
      for (int a0 = 0; a0 < N1; ++a0) {
        const int a = a21 * N1 + a0;
        const R n0v = tN[l0 * N1 + a0], d0v = tD[l0 * N1 + a0];
        const R N = n0v * n1v * n2v;
        const R dr[3] = {d0v * n1v * n2v, n0v * d1v * n2v, n0v * n1v * d2v};
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const R uv = sE[slot][a * DIM + c];
          u[c] += N * uv;
#pragma unroll
          for (int e = 0; e < DIM; ++e) gr[c * DIM + e] += uv * dr[e];
        }
      }
    }
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      sU[slot][q * DIM + c] = u[c];
#pragma unroll
      for (int d = 0; d < DIM; ++d) { // physical gradient d_d u_c = sum_e (d^_e u_c) Ji[e][d]
        R t = 0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) t += gr[c * DIM + e] * Ji[e * DIM + d];
        sGu[slot][q * DIM * DIM + c * DIM + d] = t;
      }
    }
  }
  __syncthreads();
  if (hl < NN && active) { // lane = node a: its diagonal block
    const int a = hl;
    const R mu = R(A.mu), rho = R(A.rho), gamma = R(A.gamma), inv_dt = R(A.inv_dt);
    R s = 0, D[DIM * DIM];
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) D[i] = 0;
#pragma unroll 1
    for (int q21 = 0; q21 < NN / N1; ++q21) {
      const int q1 = q21 % N1, q2 = q21 / N1;
      const R n1v = tN[q1 * N1 + l1], d1v = tD[q1 * N1 + l1];
      const R n2v = DIM == 3 ? tN[q2 * N1 + l2] : R(1), d2v = DIM == 3 ? tD[q2 * N1 + l2] : R(0);
#pragma unroll
      for (int q0 = 0; q0 < N1; ++q0) {
        const int q = q21 * N1 + q0;
        const R n0v = tN[q0 * N1 + l0], d0v = tD[q0 * N1 + l0];
        const R N = n0v * n1v * n2v;
        const R dr[3] = {d0v * n1v * n2v, n0v * d1v * n2v, n0v * n1v * d2v};
        R ga[DIM];
        const R *Ji = &sJi[slot][q * DIM * DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          R t = 0;
#pragma unroll
          for (int e = 0; e < DIM; ++e) t += dr[e] * Ji[e * DIM + d];
          ga[d] = t;
        }
        const R w = sW[slot][q];
        R gg = 0, ug = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) { gg += ga[d] * ga[d]; ug += sU[slot][q * DIM + d] * ga[d]; }
        s += w * (mu * gg + rho * N * ug + rho * inv_dt * N * N);
#pragma unroll
        for (int c = 0; c < DIM; ++c)
#pragma unroll
          for (int d = 0; d < DIM; ++d)
            D[c * DIM + d] += w * (rho * N * N * sGu[slot][q * DIM * DIM + c * DIM + d] + gamma * rho * ga[c] * ga[d]);
      }
    }
#pragma unroll
    for (int c = 0; c < DIM; ++c) D[c * DIM + c] += s;
    const int32_t nd = sNode[slot][a];
    const bool own = nd < A.nUo;
#pragma unroll
    for (int c = 0; c < DIM; ++c)
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        const bool rc = own && A.is_c && A.is_c[int64_t(DIM) * nd + c], cd = own && A.is_c && A.is_c[int64_t(DIM) * nd + d];
        R v = D[c * DIM + d];
        if (rc || cd) v = (c == d) ? R(fabs(v)) : R(0);
        sGu[slot][a * DIM * DIM + c * DIM + d] = own ? v : R(0); // staged: the gradient table is consumed (same index range)
      }
  }
  __syncthreads();
  // scatter with lane = (node, entry): consecutive lanes add to consecutive doubles of a node block (the f64 atomic unit
  // works per 64-byte segment: one lane per node block costs 9 segments per block, this costs ~1.5)
  if (active)
    for (int t = hl; t < NN * DIM * DIM; t += 32) {
      const int a = t / (DIM * DIM), e = t - a * (DIM * DIM);
      const R v = sGu[slot][t];
      if (v != R(0)) unsafeAtomicAdd(&A.out[int64_t(sNode[slot][a]) * DIM * DIM + e], double(v));
    }
}

template <int DIM>
__global__ void k_block_invert(int64_t n, double *__restrict__ b) {
  const int64_t row = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (row >= n) return;
  double D[DIM * DIM], Di[DIM * DIM];
  for (int e = 0; e < DIM * DIM; ++e) D[e] = b[row * DIM * DIM + e];
  bool zero = true;
  for (int e = 0; e < DIM * DIM; ++e) zero = zero && D[e] == 0.0;
  if (zero) for (int c = 0; c < DIM; ++c) D[c * DIM + c] = 1.0;
  inv_small<DIM>(D, Di);
  for (int e = 0; e < DIM * DIM; ++e) b[row * DIM * DIM + e] = Di[e];
}

// ctx->bjac := inverse node blocks of the matrix-free A_uu of this level (state: mf_eval, mf_params, active constraint set)
void uu_block_diag_mf(ifem_ctx *ctx) {
  if (!ctx->mf_valid) throw Error(IFEM_E_BADPARAM, "matrix-free block diagonal: no operator state on this level");
  const int64_t n = ctx->nUo;
  const int dim = ctx->dim;
  if ((int64_t)ctx->bjac.n != n * dim * dim) ctx->bjac.alloc((size_t)n * dim * dim);
  ctx->bjac_f32_valid = false;
  if (!n) return;
  hipStream_t s = ctx->stream;
  KScope ks(ctx, IFEM_KC_SMOOTHER_SETUP, double(ctx->n_cells) * ctx->nu * dim * dim * 8.0 + double(n) * dim * dim * 24.0);
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->bjac.p, 0, ctx->bjac.n * sizeof(double), s));
  DiagArgs a{};
  a.n_cells = ctx->n_cells; a.nUo = ctx->nUo;
  a.vcoords = ctx->vcoords.p; a.cell_unodes = ctx->cell_unodes.p;
  a.is_c = ctx->has_c[ctx->asm_constraint_set] ? ctx->is_c[ctx->asm_constraint_set].p : nullptr;
  a.eval = ctx->mf_eval.p; a.out = ctx->bjac.p;
  a.mu = ctx->mf_params.viscosity; a.rho = ctx->mf_params.rho; a.gamma = ctx->mf_params.grad_div; a.inv_dt = 1.0 / ctx->mf_params.dt;
  a.conv = ctx->mf_noconv ? 0 : 1;
  tab1d(a.t, ctx->kv);
  const dim3 grid(unsigned((ctx->n_cells + 7) / 8)), block(256); // 8 cells per block
  // single-precision integrals where the blocks serve the single-precision V-cycle (ifem_tuning::mf_f32, the default)
  if (ctx->tune.mf_f32) {
    if (dim == 3 && ctx->kv == 2) hipLaunchKernelGGL((k_uu_diag<3, 2, float>), grid, block, 0, s, a);
    else if (dim == 3) hipLaunchKernelGGL((k_uu_diag<3, 1, float>), grid, block, 0, s, a);
    else if (ctx->kv == 2) hipLaunchKernelGGL((k_uu_diag<2, 2, float>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_uu_diag<2, 1, float>), grid, block, 0, s, a);
  } else {
    if (dim == 3 && ctx->kv == 2) hipLaunchKernelGGL((k_uu_diag<3, 2, double>), grid, block, 0, s, a);
    else if (dim == 3) hipLaunchKernelGGL((k_uu_diag<3, 1, double>), grid, block, 0, s, a);
    else if (ctx->kv == 2) hipLaunchKernelGGL((k_uu_diag<2, 2, double>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_uu_diag<2, 1, double>), grid, block, 0, s, a);
  }
  if (dim == 3) hipLaunchKernelGGL((k_block_invert<3>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, ctx->bjac.p);
  else hipLaunchKernelGGL((k_block_invert<2>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, n, ctx->bjac.p);
  IFEM_HIP_CHECK(hipGetLastError());
}

} // namespace ifem

// ---------------------------------------------------------------------------------------------------------------------
// Velocity-space pieces of the A_uu V-cycle: node-based CSR transfers acting on the dim components of a node, masked by
// the Dirichlet flags of the two levels (a constrained dof is a decoupled 1 x 1 equation on its level: it neither sends
// nor receives corrections), injection of the evaluation point, and the Chebyshev updates with the inverse node blocks.
namespace ifem {

// y[row][c] = flag_out ? 0 : sum_k w_k (flag_in[col_k][c] ? 0 : x[col_k][c]).  G lanes share a row (prolongation rows hold
// 1..27 weights, restriction rows up to 125): index / weight reads are contiguous per group, the partial sums meet by
// shuffles.  G is chosen from the mean row length at launch.  The two flag tests are folded into one byte per weight
// (bit c: component c of this weight is dropped), rebuilt when the constrained-dof set of either level changes.
// Round 5: what the transfer kernel reads per weight is ONE 8-byte entry {column | drop bits << 29, weight as float} instead of column (4) +
// weight (8) + mask (1) from three arrays.  The weights of the nested Q_k interpolations are dyadic rationals (products of -1/8, 3/8, 3/4, 1):
// exact in single precision, the products are formed in double as before -- same results, 8 instead of 13 bytes per weight.
template <int DIM>
__global__ void k_mg_mask(int64_t n_rows, const int64_t *__restrict__ ptr, const int32_t *__restrict__ col, const double *__restrict__ w,
                          const uint8_t *__restrict__ flag_in, const uint8_t *__restrict__ flag_out, uint2 *__restrict__ pk) {
  for (int64_t r = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; r < n_rows; r += int64_t(gridDim.x) * blockDim.x) {
    uint32_t mo = 0;
#pragma unroll
    for (int c = 0; c < DIM; ++c) mo |= (flag_out && flag_out[r * DIM + c]) ? (1u << c) : 0u;
    for (int64_t k = ptr[r]; k < ptr[r + 1]; ++k) {
      uint32_t m = mo;
      const int64_t j = int64_t(col[k]) * DIM;
#pragma unroll
      for (int c = 0; c < DIM; ++c) m |= (flag_in && flag_in[j + c]) ? (1u << c) : 0u;
      pk[k] = make_uint2(uint32_t(col[k]) | (m << 29), __float_as_uint(float(w[k])));
    }
  }
}
void mg_csr_mask(ifem_ctx *ctx, const MgCsr &M, const uint8_t *flag_in, const uint8_t *flag_out, DBuf<uint8_t> &mask) {
  if (!M.n_rows) return;
  if (int64_t(ctx->nUl) >= (int64_t(1) << 29)) throw Error(IFEM_E_BADPARAM, "multigrid transfer: packed entries hold 29 column bits");
  if (mask.n != M.col.n * 8) mask.alloc(M.col.n * 8);
  KScope ks(ctx, IFEM_KC_MG_TRANSFER, double(M.col.n) * 23.0);
  uint2 *pk = reinterpret_cast<uint2 *>(mask.p);
  if (ctx->dim == 3) hipLaunchKernelGGL((k_mg_mask<3>), dim3(mgrid(M.n_rows)), dim3(256), 0, ctx->stream, M.n_rows, M.ptr.p, M.col.p, M.w.p, flag_in, flag_out, pk);
  else hipLaunchKernelGGL((k_mg_mask<2>), dim3(mgrid(M.n_rows)), dim3(256), 0, ctx->stream, M.n_rows, M.ptr.p, M.col.p, M.w.p, flag_in, flag_out, pk);
}

template <int DIM, int G, typename V>
__global__ __launch_bounds__(256) void k_mg_csr_nodes(int64_t n_rows, const int64_t *__restrict__ ptr, const uint2 *__restrict__ pk,
                                                      const V *__restrict__ x, V *__restrict__ y) {
  const int lig = threadIdx.x % G;
  const int64_t r = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) / G;
  const bool live = r < n_rows;
  double s[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) s[c] = 0;
  if (live) {
    const int64_t k1 = ptr[r + 1];
    for (int64_t k = ptr[r] + lig; k < k1; k += G) {
      const uint2 e = pk[k];
      const int64_t j = int64_t(e.x & 0x1FFFFFFFu) * DIM;
      const double wk = double(__uint_as_float(e.y));
      const unsigned m = e.x >> 29;
      double xv[DIM];
#pragma unroll
      for (int c = 0; c < DIM; ++c) xv[c] = double(x[j + c]);
#pragma unroll
      for (int c = 0; c < DIM; ++c) s[c] += ((m >> c) & 1u) ? 0.0 : wk * xv[c];
    }
  }
#pragma unroll
  for (int c = 0; c < DIM; ++c)
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) s[c] += __shfl_xor(s[c], o, G);
  if (live && lig < DIM) {
    double v = s[0];
#pragma unroll
    for (int c = 1; c < DIM; ++c) v = lig == c ? s[c] : v;
    y[r * DIM + lig] = V(v);
  }
}

template <int DIM, typename V>
static void launch_csr_nodes(ifem_ctx *ctx, const MgCsr &M, const V *x, const uint8_t *mask, V *y) {
  const double mean = double(M.col.n) / double(M.n_rows);
  const int g = mean <= 6 ? 4 : (mean <= 12 ? 8 : (mean <= 24 ? 16 : 32));
  const unsigned blocks = unsigned((M.n_rows * g + 255) / 256);
  // one packed 8-byte entry per weight; per row its pointer and DIM outputs; the gathered input is re-used from cache
  KScope ks(ctx, IFEM_KC_MG_TRANSFER, double(M.col.n) * 8.0 + double(M.n_rows) * (8.0 + DIM * sizeof(V)));
  const uint2 *pk = reinterpret_cast<const uint2 *>(mask);
#define IFEM_CSRN(G) hipLaunchKernelGGL((k_mg_csr_nodes<DIM, G, V>), dim3(blocks), dim3(256), 0, ctx->stream, M.n_rows, M.ptr.p, pk, x, y)
  if (g == 4) IFEM_CSRN(4); else if (g == 8) IFEM_CSRN(8); else if (g == 16) IFEM_CSRN(16); else IFEM_CSRN(32);
#undef IFEM_CSRN
}

void mg_csr_apply_nodes(ifem_ctx *ctx, const MgCsr &M, const double *x, const DBuf<uint8_t> &mask, double *y) {
  if (!M.n_rows) return;
  if (mask.n != M.col.n * 8) throw Error(IFEM_E_BADPARAM, "multigrid transfer without its packed entries (mg_csr_mask)");
  if (ctx->dim == 3) launch_csr_nodes<3, double>(ctx, M, x, mask.p, y);
  else launch_csr_nodes<2, double>(ctx, M, x, mask.p, y);
}
void mg_csr_apply_nodes_f32(ifem_ctx *ctx, const MgCsr &M, const float *x, const DBuf<uint8_t> &mask, float *y) {
  if (!M.n_rows) return;
  if (mask.n != M.col.n * 8) throw Error(IFEM_E_BADPARAM, "multigrid transfer without its packed entries (mg_csr_mask)");
  if (ctx->dim == 3) launch_csr_nodes<3, float>(ctx, M, x, mask.p, y);
  else launch_csr_nodes<2, float>(ctx, M, x, mask.p, y);
}

__global__ void k_mg_inject(int64_t n_nodes, int dim, const int32_t *__restrict__ inj, const double *__restrict__ fine,
                            double *__restrict__ coarse) {
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n_nodes * dim; t += int64_t(gridDim.x) * blockDim.x) {
    const int64_t i = t / dim;
    const int c = int(t - i * dim);
    coarse[t] = inj[i] >= 0 ? fine[int64_t(inj[i]) * dim + c] : 0.0; // -1: another rank's node (replicated coarse level: summed over the ranks)
  }
}
void mg_inject_nodes(ifem_ctx *ctx, int64_t n_nodes, const int32_t *inj, const double *fine, double *coarse) {
  KScope ks(ctx, IFEM_KC_MG_TRANSFER, double(n_nodes) * (4.0 + 16.0 * ctx->dim));
  if (n_nodes) hipLaunchKernelGGL(k_mg_inject, dim3(mgrid(n_nodes * ctx->dim)), dim3(256), 0, ctx->stream, n_nodes, ctx->dim, inj, fine, coarse);
}

// d = c0 B r per node (B = inverse diagonal block)
template <int DIM, typename V>
__global__ void k_cheb_init_block(int64_t n_nodes, double c0, const float *__restrict__ bj, const V *__restrict__ r,
                                  V *__restrict__ d) {
  const int64_t nd = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (nd >= n_nodes) return;
  double rv[DIM];
#pragma unroll
  for (int j = 0; j < DIM; ++j) rv[j] = double(r[nd * DIM + j]);
#pragma unroll
  for (int i = 0; i < DIM; ++i) {
    double t = 0;
#pragma unroll
    for (int j = 0; j < DIM; ++j) t += double(bj[nd * DIM * DIM + i * DIM + j]) * rv[j];
    d[nd * DIM + i] = V(c0 * t);
  }
}
template <typename V>
static void cheb_init_block_t(ifem_ctx *ctx, double c0, const V *r, V *d) {
  const int64_t n = ctx->nUo;
  if (!n) return;
  const dim3 g(unsigned((n + 255) / 256)), b(256);
  const float *bjf = bjac_f32_ptr(ctx);
  KScope ks(ctx, IFEM_KC_VECTOR, double(n) * ctx->dim * (4.0 * ctx->dim + 2.0 * sizeof(V)));
  if (ctx->dim == 3) hipLaunchKernelGGL((k_cheb_init_block<3, V>), g, b, 0, ctx->stream, n, c0, bjf, r, d);
  else hipLaunchKernelGGL((k_cheb_init_block<2, V>), g, b, 0, ctx->stream, n, c0, bjf, r, d);
}
void cheb_init_block(ifem_ctx *ctx, double c0, const double *r, double *d) { cheb_init_block_t<double>(ctx, c0, r, d); }
void cheb_init_block_f32(ifem_ctx *ctx, double c0, const float *r, float *d) { cheb_init_block_t<float>(ctx, c0, r, d); }

// conversions between the double vectors of the Krylov solvers and the single-precision level vectors, y (+)= a x
__global__ void k_cvt_d2f(int64_t n, const double *__restrict__ x, float *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = float(x[i]);
}
__global__ void k_cvt_f2d(int64_t n, const float *__restrict__ x, double *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] = double(x[i]);
}
__global__ void k_axpy_f32v(int64_t n, float a, const float *__restrict__ x, float *__restrict__ y) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) y[i] += a * x[i];
}
void v_cvt_d2f(ifem_ctx *ctx, int64_t n, const double *x, float *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 12.0 * double(n));
  if (n) hipLaunchKernelGGL(k_cvt_d2f, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, x, y);
}
void v_cvt_f2d(ifem_ctx *ctx, int64_t n, const float *x, double *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 12.0 * double(n));
  if (n) hipLaunchKernelGGL(k_cvt_f2d, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, x, y);
}
void v_axpy_f32v(ifem_ctx *ctx, int64_t n, float a, const float *x, float *y) {
  KScope ks(ctx, IFEM_KC_VECTOR, 12.0 * double(n));
  if (n) hipLaunchKernelGGL(k_axpy_f32v, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, a, x, y);
}
} // namespace ifem
