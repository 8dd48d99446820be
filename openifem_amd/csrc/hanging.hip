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
// preconditioner).
//
// Partitioned contexts: the lines list every LOCAL hanging dof (owned and ghost) with local (ghost-extended) master ids --
// the caller's ghost layer holds the masters of its ghost hanging nodes.  C acts on a ghost-extended copy of x after the
// halo refresh (every rank interpolates all its local hanging entries itself: one exchange); C^T adds w y_h to masters
// that may be ghosts here, so the extended result travels back to the owners (halo_reverse_add, comm.hip) before the
// hanging rows are overwritten.  Only the owner of a hanging dof holds its row.
#include <hip/hip_runtime.h>
#include <vector>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

// ghost-extended block vector [u (dim*nUl) | p (nPl)] vs compact owned vector [u (dim*nUo) | p (nPo)]
struct HLayout {
  int64_t nu_ext, nu_own, np_own;
  __host__ __device__ int64_t compact(int64_t e) const { // -1: not owned here
    if (e < nu_ext) return e < nu_own ? e : -1;
    const int64_t p = e - nu_ext;
    return p < np_own ? nu_own + p : -1;
  }
};
static HLayout layout(const ifem_ctx *c) { return {int64_t(c->dim) * c->nUl, int64_t(c->dim) * c->nUo, c->nPo}; }

// x_h = sum over masters that are not Dirichlet-constrained in the active set (those columns are eliminated); extended x
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

// y_k += w_hk y_h for the free masters (C^T) on the extended y, lines of OWNED hanging dofs only (their rows live here)
__global__ void k_hang_scatter(int32_t n, const int32_t *__restrict__ dof, const int32_t *__restrict__ ptr,
                               const int32_t *__restrict__ master, const double *__restrict__ w,
                               const uint8_t *__restrict__ is_c, double *__restrict__ y, HLayout L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || L.compact(dof[i]) < 0) return;
  const double v = y[dof[i]];
  for (int k = ptr[i]; k < ptr[i + 1]; ++k)
    if (!(is_c && is_c[master[k]])) unsafeAtomicAdd(&y[master[k]], w[k] * v);
}

// y_h = d_h x_h on the compact y (owned hanging rows); x compact or extended
__global__ void k_hang_rows(int32_t n, const int32_t *__restrict__ dof, const double *__restrict__ d,
                            const double *__restrict__ x, int x_extended, double *__restrict__ y, HLayout L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = L.compact(dof[i]);
  if (c < 0) return;
  y[c] = d[i] * x[x_extended ? int64_t(dof[i]) : c];
}

// c0_h = sum over Dirichlet masters of w_hk g_k (the inhomogeneity of the closed line) on the extended c0, zero elsewhere
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

// AffineConstraints::distribute: x_h = sum over ALL masters (Dirichlet masters carry their values already); reads the
// extended copy, writes the owned hanging entries of the compact vector
__global__ void k_hang_distribute(int32_t n, const int32_t *__restrict__ dof, const int32_t *__restrict__ ptr,
                                  const int32_t *__restrict__ master, const double *__restrict__ w,
                                  const double *__restrict__ xe, double *__restrict__ x, HLayout L) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t c = L.compact(dof[i]);
  if (c < 0) return;
  double s = 0;
  for (int k = ptr[i]; k < ptr[i + 1]; ++k) s += w[k] * xe[master[k]];
  x[c] = s;
}

// diagonal of A^_uu at the owned hanging velocity dofs from the inverse node blocks of the block-Jacobi set-up; hanging
// pressure dofs (no diagonal in A^: the p-p block is zero or tiny) are marked -1 and take the mean of the velocity values
// (or 1) on the host; ghost lines are marked -2 (their rows live on the owner)
template <int DIM>
__global__ void k_hang_diag(int32_t n, const int32_t *__restrict__ dof, HLayout L, const double *__restrict__ bjac,
                            double *__restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t g = dof[i];
  if (L.compact(g) < 0) { d[i] = -2.0; return; }
  if (g >= L.nu_ext) { d[i] = -1.0; return; }
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
  // collective over ranks: a rank without hanging lines still takes part in the exchanges of the others
  double any = n > 0 ? 1.0 : 0.0;
  allreduce_max(ctx, &any, 1);
  h.active = any != 0.0;
  if (n <= 0) return;
  if (!dof || !ptr || !master || !weight) throw Error(IFEM_E_BADPARAM, "null argument");
  std::vector<uint8_t> is_h((size_t)ctx->n_local, 0);
  for (int32_t i = 0; i < n; ++i) {
    if (dof[i] < 0 || dof[i] >= ctx->n_local) throw Error(IFEM_E_BADPARAM, "hanging dof out of range");
    if (is_h[dof[i]]) throw Error(IFEM_E_BADPARAM, "hanging dof listed twice");
    is_h[dof[i]] = 1;
    if (ptr[i + 1] < ptr[i]) throw Error(IFEM_E_BADPARAM, "hanging ptr must be non-decreasing");
  }
  for (int32_t k = ptr[0]; k < ptr[n]; ++k) {
    if (master[k] < 0 || master[k] >= ctx->n_local) throw Error(IFEM_E_BADPARAM, "hanging master out of range (masters of every listed line must be local nodes)");
    if (is_h[master[k]]) throw Error(IFEM_E_BADPARAM, "hanging lines must be closed (a master is itself a hanging dof)");
  }
  if (ptr[0] != 0) throw Error(IFEM_E_BADPARAM, "hanging ptr[0] must be 0");
  h.dof.upload(dof, (size_t)n, ctx->stream);
  h.ptr.upload(ptr, (size_t)n + 1, ctx->stream);
  h.master.upload(master, (size_t)ptr[n], ctx->stream);
  h.w.upload(weight, (size_t)ptr[n], ctx->stream);
  h.d.alloc((size_t)n);
  h.flag.alloc(1);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  h.host_dof.assign(dof, dof + n);
  h.n = n;
}

static void ensure_buffers(ifem_ctx *ctx) {
  Hanging &h = ctx->hang;
  if (h.x.n != (size_t)ctx->n_local) {
    h.x.alloc((size_t)ctx->n_local);
    h.c0.alloc((size_t)ctx->n_local);
    h.y.alloc((size_t)ctx->n_local);
    if (!h.flag.p) h.flag.alloc(1);
  }
}

static const uint8_t *active_flags(const ifem_ctx *ctx) {
  return ctx->has_c[ctx->asm_constraint_set] ? ctx->is_c[ctx->asm_constraint_set].p : nullptr;
}

// ghost-extended copy of a compact owned vector, ghosts refreshed
static void extend_full(ifem_ctx *ctx, const double *x, double *xe) {
  const HLayout L = layout(ctx);
  v_copy(ctx, L.nu_own, x, xe);
  v_copy(ctx, L.np_own, x + L.nu_own, xe + L.nu_ext);
  halo_exchange(ctx, xe);
  halo_exchange_p(ctx, xe + L.nu_ext);
}

const double *hanging_input(ifem_ctx *ctx, const double *x) {
  Hanging &h = ctx->hang;
  ensure_buffers(ctx);
  extend_full(ctx, x, h.x.p);
  if (h.n)
    hipLaunchKernelGGL(k_hang_interp, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                       active_flags(ctx), h.x.p);
  return h.x.p;
}

void hanging_output(ifem_ctx *ctx, const double *x, bool x_extended, double *y) {
  Hanging &h = ctx->hang;
  const HLayout L = layout(ctx);
  if (ctx->halo.nranks == 1) { // compact == extended: scatter in place
    if (!h.n) return;
    hipLaunchKernelGGL(k_hang_scatter, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                       active_flags(ctx), y, L);
    hipLaunchKernelGGL(k_hang_rows, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.d.p, x, int(x_extended), y, L);
    return;
  }
  ensure_buffers(ctx);
  double *ye = h.y.p;
  v_zero(ctx, ctx->n_local, ye);
  v_copy(ctx, L.nu_own, y, ye);
  v_copy(ctx, L.np_own, y + L.nu_own, ye + L.nu_ext);
  if (h.n)
    hipLaunchKernelGGL(k_hang_scatter, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                       active_flags(ctx), ye, L);
  halo_reverse_add(ctx, ye);
  halo_reverse_add_p(ctx, ye + L.nu_ext);
  v_copy(ctx, L.nu_own, ye, y);
  v_copy(ctx, L.np_own, ye + L.nu_ext, y + L.nu_own);
  if (h.n)
    hipLaunchKernelGGL(k_hang_rows, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.d.p, x, int(x_extended), y, L);
}

void hanging_distribute(ifem_ctx *ctx, double *x) {
  Hanging &h = ctx->hang;
  if (!h.active) return;
  ensure_buffers(ctx);
  extend_full(ctx, x, h.x.p);
  if (h.n)
    hipLaunchKernelGGL(k_hang_distribute, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                       h.x.p, x, layout(ctx));
}

// diagonal entries of the hanging rows (after bjac_setup of the current assembly)
void hanging_refresh_diag(ifem_ctx *ctx) {
  Hanging &h = ctx->hang;
  const HLayout L = layout(ctx);
  std::vector<double> d((size_t)h.n);
  if (h.n) {
    if (ctx->dim == 3) hipLaunchKernelGGL((k_hang_diag<3>), hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, L, ctx->bjac.p, h.d.p);
    else hipLaunchKernelGGL((k_hang_diag<2>), hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, L, ctx->bjac.p, h.d.p);
    IFEM_HIP_CHECK(hipMemcpyAsync(d.data(), h.d.p, d.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  double sm[2] = {0, 0}; // sum and count of the velocity diagonals, over all ranks
  for (double v : d) if (v > 0) { sm[0] += v; sm[1] += 1; }
  allreduce_sum(ctx, sm, 2);
  const double mean = sm[1] > 0 ? sm[0] / sm[1] : 1.0;
  bool touched = false;
  for (double &v : d) if (v == -1.0) { v = mean; touched = true; }
  if (touched) {
    IFEM_HIP_CHECK(hipMemcpyAsync(h.d.p, d.data(), d.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // d is a local
  }
}

// c0 of the active constraint set into h.c0 (ghost-extended: every rank evaluates all its local lines); returns whether
// any entry is non-zero on any rank
bool hanging_offset(ifem_ctx *ctx, int use_nonzero) {
  Hanging &h = ctx->hang;
  const int set = use_nonzero ? 1 : 0;
  ensure_buffers(ctx);
  v_zero(ctx, ctx->n_local, h.c0.p);
  double any = 0;
  if (ctx->has_c[set] && h.n) {
    IFEM_HIP_CHECK(hipMemsetAsync(h.flag.p, 0, sizeof(int), ctx->stream));
    hipLaunchKernelGGL(k_hang_offset, hgrid(h.n), dim3(128), 0, ctx->stream, h.n, h.dof.p, h.ptr.p, h.master.p, h.w.p,
                       ctx->is_c[set].p, ctx->cval[set].p, h.c0.p, h.flag.p);
    int a = 0;
    IFEM_HIP_CHECK(hipMemcpyAsync(&a, h.flag.p, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    any = a ? 1.0 : 0.0;
  }
  allreduce_max(ctx, &any, 1);
  return any != 0.0;
}

} // namespace ifem
