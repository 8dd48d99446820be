// hanging.hip -- hanging-node lines of the AffineConstraints the reference assembles through
// (DoFTools::make_hanging_node_constraints, source/mpi_fluid_solver.cpp:182-184, consumed by
// distribute_local_to_global, mpi_insim.cpp:343-355, and constraints.distribute, :390).
//
// distribute_local_to_global with lines x_h = sum_k w_hk x_k yields the condensed system C^T A^ C (A^ = the matrix
// assembled as if the hanging dofs were ordinary ones, C = identity on regular dofs, the weights on hanging rows),
// a diagonal entry on every hanging row and the right-hand side C^T (b^ - A^ c0), c0 = the inhomogeneity a hanging
// dof inherits from Dirichlet masters.  This build keeps the cell kernels and the sparsity of A^ untouched and applies
// C and C^T to the VECTORS around the operator of the outer Krylov solver instead: a handful of rows per refinement
// interface, two tiny kernels per application.  The block preconditioner is built from the blocks of A^ (it is only a
// preconditioner).  Single-rank contexts.
#include <hip/hip_runtime.h>
#include <vector>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

// x_h = sum over masters that are not Dirichlet-constrained in the active set (those columns are eliminated)
__global__ void k_hang_interp(int32_t n, const int32_t *__restrict__ dof, const int32_t *__restrict__ ptr,
                              const int32_t *__restrict__ master, const double *__restrict__ w,
                              const uint8_t *__restrict__ is_c, double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k)
    if (!(is_c && is_c[master[k]])) s += w[k] * x[master[k]];
  x[dof[i]] = s;
}

// y_k += w_hk y_h for the free masters (C^T), in place: hanging rows are read here and overwritten by k_hang_rows
__global__ void k_hang_scatter(int32_t n, const int32_t *__restrict__ dof, const int32_t *__restrict__ ptr,
                               const int32_t *__restrict__ master, const double *__restrict__ w,
                               const uint8_t *__restrict__ is_c, double *__restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = y[dof[i]];
  for (int k = ptr[i]; k < ptr[i + 1]; ++k)
    if (!(is_c && is_c[master[k]])) unsafeAtomicAdd(&y[master[k]], w[k] * v);
}

// y_h = d_h x_h (scale = 1) or b_h = d_h c0_h
__global__ void k_hang_rows(int32_t n, const int32_t *__restrict__ dof, const double *__restrict__ d,
                            const double *__restrict__ x, double *__restrict__ y) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[dof[i]] = d[i] * x[dof[i]];
}

// c0_h = sum over Dirichlet masters of w_hk g_k (the inhomogeneity of the closed line), zero elsewhere
__global__ void k_hang_offset(int32_t n, const int32_t *__restrict__ dof, const int32_t *__restrict__ ptr,
                              const int32_t *__restrict__ master, const double *__restrict__ w,
                              const uint8_t *__restrict__ is_c, const double *__restrict__ cval, double *__restrict__ c0,
                              int *__restrict__ any) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k)
    if (is_c && is_c[master[k]]) s += w[k] * cval[master[k]];
  c0[dof[i]] = s;
  if (s != 0.0) *any = 1;
}

// AffineConstraints::distribute: x_h = sum over ALL masters (Dirichlet masters carry their values in x already)
__global__ void k_hang_distribute(int32_t n, const int32_t *__restrict__ dof, const int32_t *__restrict__ ptr,
                                  const int32_t *__restrict__ master, const double *__restrict__ w, double *__restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k) s += w[k] * x[master[k]];
  x[dof[i]] = s;
}

// diagonal of A^_uu at the hanging velocity dofs from the inverse node blocks of the block-Jacobi set-up; hanging
// pressure dofs (no diagonal in A^: the p-p block is zero or tiny) take the mean of the velocity values, or 1
template <int DIM>
__global__ void k_hang_diag(int32_t n, const int32_t *__restrict__ dof, int64_t n_u, const double *__restrict__ bjac,
                            double *__restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t g = dof[i];
  if (g >= n_u) { d[i] = -1.0; return; }
  const int64_t nd = g / DIM;
  const int c = int(g - nd * DIM);
  const double *b = bjac + nd * DIM * DIM;
  double a; // (B^-1)_cc of the DIM x DIM block B = inverse diagonal block
  if (DIM == 2) {
    const double det = b[0] * b[3] - b[1] * b[2];
    a = (c == 0 ? b[3] : b[0]) / det;
  } else {
    const double c00 = b[4] * b[8] - b[5] * b[7], c11 = b[0] * b[8] - b[2] * b[6], c22 = b[0] * b[4] - b[1] * b[3];
    const double det = b[0] * c00 - b[1] * (b[3] * b[8] - b[5] * b[6]) + b[2] * (b[3] * b[7] - b[4] * b[6]);
    a = (c == 0 ? c00 : (c == 1 ? c11 : c22)) / det;
  }
  d[i] = fabs(a);
}

static inline dim3 hgrid(int32_t n) { return dim3(unsigned((n + 127) / 128)); }

void hanging_set(ifem_ctx *ctx, int32_t n, const int32_t *dof, const int32_t *ptr, const int32_t *master, const double *weight) {
  Hanging &h = ctx->hang;
  h.n = 0;
  if (n <= 0) return;
  if (ctx->halo.nranks > 1) throw Error(IFEM_E_BADPARAM, "hanging-node constraints: single-rank contexts only in this build");
  if (!dof || !ptr || !master || !weight) throw Error(IFEM_E_BADPARAM, "null argument");
  std::vector<uint8_t> is_h((size_t)ctx->n_local, 0);
  for (int32_t i = 0; i < n; ++i) {
    if (dof[i] < 0 || dof[i] >= ctx->n_local) throw Error(IFEM_E_BADPARAM, "hanging dof out of range");
    if (is_h[dof[i]]) throw Error(IFEM_E_BADPARAM, "hanging dof listed twice");
    is_h[dof[i]] = 1;
    if (ptr[i + 1] < ptr[i]) throw Error(IFEM_E_BADPARAM, "hanging ptr must be non-decreasing");
  }
  for (int32_t k = ptr[0]; k < ptr[n]; ++k) {
    if (master[k] < 0 || master[k] >= ctx->n_local) throw Error(IFEM_E_BADPARAM, "hanging master out of range");
    if (is_h[master[k]]) throw Error(IFEM_E_BADPARAM, "hanging lines must be closed (a master is itself a hanging dof)");
  }
  if (ptr[0] != 0) throw Error(IFEM_E_BADPARAM, "hanging ptr[0] must be 0");
  h.dof.upload(dof, (size_t)n, ctx->stream);
  h.ptr.upload(ptr, (size_t)n + 1, ctx->stream);
  h.master.upload(master, (size_t)ptr[n], ctx->stream);
  h.w.upload(weight, (size_t)ptr[n], ctx->stream);
  h.d.alloc((size_t)n);
  h.x.alloc((size_t)ctx->n_local);
  h.c0.alloc((size_t)ctx->n_local);
  h.flag.alloc(1);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  h.host_dof.assign(dof, dof + n);
  h.n = n;
}

static const uint8_t *active_flags(const ifem_ctx *ctx) {
  return ctx->has_c[ctx->asm_constraint_set] ? ctx->is_c[ctx->asm_constraint_set].p : nullptr;
}

const double *hanging_input(ifem_ctx *ctx, const double *x) {
  Hanging &h = ctx->hang;
  const int64_t n = int64_t(ctx->dim) * ctx->nUo + ctx->nPo;
  v_copy(ctx, n, x, h.x.p);
  hipLaunchKernelGGL(k_hang_interp, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                     active_flags(ctx), h.x.p);
  return h.x.p;
}

void hanging_output(ifem_ctx *ctx, const double *x, double *y) {
  Hanging &h = ctx->hang;
  hipLaunchKernelGGL(k_hang_scatter, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                     active_flags(ctx), y);
  hipLaunchKernelGGL(k_hang_rows, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.d.p, x, y);
}

void hanging_distribute(ifem_ctx *ctx, double *x) {
  Hanging &h = ctx->hang;
  if (!h.n) return;
  hipLaunchKernelGGL(k_hang_distribute, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p, x);
}

// diagonal entries of the hanging rows (after bjac_setup of the current assembly)
void hanging_refresh_diag(ifem_ctx *ctx) {
  Hanging &h = ctx->hang;
  const int64_t nu = int64_t(ctx->dim) * ctx->nUo;
  if (ctx->dim == 3) hipLaunchKernelGGL((k_hang_diag<3>), hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, nu, ctx->bjac.p, h.d.p);
  else hipLaunchKernelGGL((k_hang_diag<2>), hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, nu, ctx->bjac.p, h.d.p);
  std::vector<double> d((size_t)h.n);
  IFEM_HIP_CHECK(hipMemcpyAsync(d.data(), h.d.p, d.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  double s = 0; int m = 0;
  for (double v : d) if (v > 0) { s += v; ++m; }
  const double mean = m ? s / m : 1.0;
  bool touched = false;
  for (double &v : d) if (!(v > 0)) { v = mean; touched = true; }
  if (touched) h.d.upload(d.data(), d.size(), ctx->stream);
}

// c0 of the active constraint set into h.c0; returns whether any entry is non-zero
bool hanging_offset(ifem_ctx *ctx, int use_nonzero) {
  Hanging &h = ctx->hang;
  const int set = use_nonzero ? 1 : 0;
  if (!ctx->has_c[set]) return false;
  const int64_t n = int64_t(ctx->dim) * ctx->nUo + ctx->nPo;
  v_zero(ctx, n, h.c0.p);
  IFEM_HIP_CHECK(hipMemsetAsync(h.flag.p, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_hang_offset, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                     ctx->is_c[set].p, ctx->cval[set].p, h.c0.p, h.flag.p);
  int any = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&any, h.flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  return any != 0;
}

} // namespace ifem
