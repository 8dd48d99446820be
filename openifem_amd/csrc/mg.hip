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
  if (n) hipLaunchKernelGGL(k_cheb_init, dim3(mgrid(n)), dim3(256), 0, ctx->stream, n, c0, dinv, r, d);
}
void cheb_step(ifem_ctx *ctx, int64_t n, double a, double b, const double *dinv, const double *t, double *x, double *r, double *d) {
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
