// tpp.hip -- the pressure Schur complement of SUPGFluidSolver::BlockIncompSchurPreconditioner as an explicit matrix.
//   reference (mpi_supg_solver.cpp:19-32, 56-133, 163-179): T_pp = A_pp - A_pv P_vv^-1 A_vp is an OPERATOR (P_vv = ILU(0)
//   of A_vv), solved by GMRES(200) that is preconditioned with the ILU(0) of the assembled B2pp = A_pp - A_pv D^-1 A_vp.
// Here P_vv^-1 is the node-block Jacobi of A_vv, so T_pp itself has the sparsity of A_pv A_vp (= the pattern of
// mass_schur) and is assembled once per Newton iteration:  T~[i,j] = A_pp[i,j] - sum_k A_pv[i,k] Binv_k A_vp[k,j].
// Its inverse is then applied exactly where the pressure space is small (dense LU through rocSOLVER, up to
// ifem_tuning::tpp_dense_max rows: the 2D benchmark meshes of the reference), and by Jacobi-preconditioned GMRES on one SpMV per
// iteration beyond.  Only the preconditioner is affected.  Single-rank contexts (the pattern needs a 2-deep halo).
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <rocsolver/rocsolver.h>
#include <cstdlib>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

// rocSOLVER is linked at build time: bound at load, its code objects stay deferred until the first factorisation (a
// dlopen after the HIP runtime is up loads them eagerly, which takes minutes).
#define IFEM_ROCBLAS_CHECK(expr)                                                                              \
  do {                                                                                                        \
    rocblas_status st_ = (expr);                                                                              \
    if (st_ != rocblas_status_success)                                                                        \
      throw ::ifem::Error(IFEM_E_HIP, std::string(#expr) + ": rocblas status " + std::to_string(int(st_)));   \
  } while (0)

// one wave per pressure row i, as k_schur_numeric; the row of A_pp is merged in at the end
template <int DIM>
__global__ __launch_bounds__(256) void k_tpp_numeric(int64_t n_rows, int maxlen, const int64_t *__restrict__ rpS,
                                                     const int32_t *__restrict__ colS, double *__restrict__ valS,
                                                     const int64_t *__restrict__ rpB, const int32_t *__restrict__ colB,
                                                     const double *__restrict__ valB, const int64_t *__restrict__ rpT,
                                                     const int32_t *__restrict__ colT, const double *__restrict__ valT,
                                                     const double *__restrict__ binv, const int64_t *__restrict__ rpM,
                                                     const int32_t *__restrict__ colM, const double *__restrict__ app) {
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

__global__ void k_csr_to_dense(int64_t n, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                               const double *__restrict__ val, double *__restrict__ D) {
  const int64_t row = int64_t(blockIdx.x) * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  for (int64_t k = rp[row] + (threadIdx.x & 63); k < rp[row + 1]; k += 64) D[row + int64_t(col[k]) * n] = val[k]; // column-major
}

void tpp_numeric(ifem_ctx *ctx) {
  if (ctx->tpp_valid) return;
  if (ctx->halo.nranks > 1) throw Error(IFEM_E_BADPARAM, "explicit T_pp: single-rank contexts only");
  if (ctx->Sm.n_rows == 0) build_schur_pattern(ctx);
  const int64_t n = ctx->Sm.n_rows;
  if (n == 0) return;
  if (ctx->Tpp.n != ctx->Sm.val.n) ctx->Tpp.alloc(ctx->Sm.val.n);
  const int maxlen = (ctx->Sm.max_row + 1) & ~1;
  const size_t smem = size_t(4) * maxlen * (sizeof(double) + sizeof(int32_t));
  const unsigned blocks = unsigned((n + 3) / 4);
#define IFEM_TPP(D)                                                                                                     \
  hipLaunchKernelGGL((k_tpp_numeric<D>), dim3(blocks), dim3(256), smem, ctx->stream, n, maxlen, ctx->Sm.rowptr.p,       \
                     ctx->Sm.col.p, ctx->Tpp.p, ctx->B.rowptr.p, ctx->B.col.p, ctx->B.val.p, ctx->Bt.rowptr.p,          \
                     ctx->Bt.col.p, ctx->Bt.val.p, ctx->bjac.p, ctx->Mp.rowptr.p, ctx->Mp.col.p, ctx->App.p)
  if (ctx->dim == 3) IFEM_TPP(3); else IFEM_TPP(2);
#undef IFEM_TPP
  IFEM_HIP_CHECK(hipGetLastError());
  if (ctx->tpp_diag.n != (size_t)n) ctx->tpp_diag.alloc((size_t)n);
  scalar_diag(ctx, ctx->Sm, ctx->Tpp.p, ctx->tpp_diag.p);
  ctx->tpp_valid = true;
  ctx->tpp_dense_valid = false;
}

void spmv_tpp(ifem_ctx *ctx, const double *xp, double *yp) {
  const int64_t n = ctx->Sm.n_rows;
  if (n == 0) return;
  spmv_planar_scalar(ctx, ctx->Sm, ctx->Tpp.p, xp, yp);
}


static rocblas_handle handle_of(ifem_ctx *ctx) {
  if (!ctx->rocblas) {
    rocblas_handle h;
    IFEM_ROCBLAS_CHECK(rocblas_create_handle(&h));
    IFEM_ROCBLAS_CHECK(rocblas_set_stream(h, ctx->stream));
    ctx->rocblas = h;
  }
  return static_cast<rocblas_handle>(ctx->rocblas);
}
void tpp_release(ifem_ctx *ctx) {
  if (ctx->rocblas) rocblas_destroy_handle(static_cast<rocblas_handle>(ctx->rocblas));
  ctx->rocblas = nullptr;
}

// LU factors of the dense copy of T~ (partial pivoting); false when the pressure space is too large for this path
bool tpp_dense_setup(ifem_ctx *ctx) {
  const int64_t n = ctx->Sm.n_rows;
  if (n == 0 || n > ctx->tune.tpp_dense_max) return false;
  if (ctx->tpp_dense_valid) return true;
  rocblas_handle h = handle_of(ctx);
  if (ctx->tpp_dense.n != size_t(n) * size_t(n)) ctx->tpp_dense.alloc(size_t(n) * size_t(n));
  if (ctx->tpp_ipiv.n != size_t(n) + 1) ctx->tpp_ipiv.alloc(size_t(n) + 1);
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->tpp_dense.p, 0, ctx->tpp_dense.n * sizeof(double), ctx->stream));
  hipLaunchKernelGGL(k_csr_to_dense, dim3(unsigned((n + 3) / 4)), dim3(256), 0, ctx->stream, n, ctx->Sm.rowptr.p, ctx->Sm.col.p,
                     ctx->Tpp.p, ctx->tpp_dense.p);
  int *info = ctx->tpp_ipiv.p + n;
  IFEM_ROCBLAS_CHECK(rocsolver_dgetrf(h, (rocblas_int)n, (rocblas_int)n, ctx->tpp_dense.p, (rocblas_int)n, ctx->tpp_ipiv.p, info));
  int hinfo = 0;
  IFEM_HIP_CHECK(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (hinfo != 0) throw Error(IFEM_E_KRYLOV_NOCONV, "T_pp is singular (dense LU pivot " + std::to_string(hinfo) + ")");
  ctx->tpp_dense_valid = true;
  return true;
}

// y = T~^-1 x
void tpp_dense_solve(ifem_ctx *ctx, const double *x, double *y) {
  const int64_t n = ctx->Sm.n_rows;
  v_copy(ctx, n, x, y);
  IFEM_ROCBLAS_CHECK(rocsolver_dgetrs(handle_of(ctx), rocblas_operation_none, (rocblas_int)n, 1, ctx->tpp_dense.p, (rocblas_int)n, ctx->tpp_ipiv.p, y, (rocblas_int)n));
}

} // namespace ifem
