// comm.hip -- the two collectives of the fluid step over RCCL / xGMI:
//   halo exchange of a DoF vector   (PETSc VecScatter in MatMult and ghosted-vector assignment)
//   all-reduce of a few scalars     (l2_norm(), Krylov dot products)
// Point-to-point ncclSend/ncclRecv grouped per exchange: one message per neighbour = one per xGMI link for
// an octant partition.  Single-rank runs never touch RCCL.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

#define IFEM_NCCL_CHECK(expr)                                                                              \
  do {                                                                                                     \
    ncclResult_t r_ = (expr);                                                                              \
    if (r_ != ncclSuccess)                                                                                 \
      throw ::ifem::Error(IFEM_E_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));                \
  } while (0)

void comm_init(ifem_ctx *ctx, const ifem_partition *part) {
  Halo &h = ctx->halo;
  if (!part || part->nranks <= 1) { h.rank = 0; h.nranks = 1; return; }
  h.rank = part->rank; h.nranks = part->nranks;
  h.nbr.assign(part->neighbor_rank, part->neighbor_rank + part->n_neighbors);
  const int nn = part->n_neighbors;
  h.send_u_ptr.assign(part->send_u_ptr, part->send_u_ptr + nn + 1);
  h.recv_u_ptr.assign(part->recv_u_ptr, part->recv_u_ptr + nn + 1);
  h.send_p_ptr.assign(part->send_p_ptr, part->send_p_ptr + nn + 1);
  h.recv_p_ptr.assign(part->recv_p_ptr, part->recv_p_ptr + nn + 1);
  h.send_u_idx.upload(part->send_u_idx, h.send_u_ptr[nn], ctx->stream);
  h.send_p_idx.upload(part->send_p_idx, h.send_p_ptr[nn], ctx->stream);
  h.sendbuf.alloc((size_t)ctx->dim * h.send_u_ptr[nn] + h.send_p_ptr[nn] + 8);
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
  std::memcpy(&id, part->nccl_unique_id, 128);
  ncclComm_t comm;
  IFEM_NCCL_CHECK(ncclCommInitRank(&comm, h.nranks, id, h.rank));
  h.comm = comm;
}

void comm_destroy(ifem_ctx *ctx) {
  if (ctx->halo.comm) ncclCommDestroy((ncclComm_t)ctx->halo.comm);
  ctx->halo.comm = nullptr;
}

__global__ void k_pack(int64_t n, int bs, const int32_t *__restrict__ idx, const double *__restrict__ x,
                       double *__restrict__ buf) {
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n * bs; t += int64_t(gridDim.x) * blockDim.x) {
    const int64_t i = t / bs;
    const int c = int(t - i * bs);
    buf[t] = x[int64_t(idx[i]) * bs + c];
  }
}

static void exchange(ifem_ctx *ctx, double *x, int bs, int64_t n_owned, const std::vector<int32_t> &sptr,
                     const DBuf<int32_t> &sidx, const std::vector<int32_t> &rptr, double *sendbuf) {
  Halo &h = ctx->halo;
  const int nn = (int)h.nbr.size();
  const int64_t ns = sptr[nn];
  if (ns) {
    int64_t g = (ns * bs + 255) / 256;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, ctx->stream, ns, bs, sidx.p, x, sendbuf);
  }
  IFEM_NCCL_CHECK(ncclGroupStart());
  for (int k = 0; k < nn; ++k) {
    const int64_t sc = int64_t(sptr[k + 1] - sptr[k]) * bs, rc = int64_t(rptr[k + 1] - rptr[k]) * bs;
    if (sc) IFEM_NCCL_CHECK(ncclSend(sendbuf + int64_t(sptr[k]) * bs, sc, ncclDouble, h.nbr[k], (ncclComm_t)h.comm, ctx->stream));
    if (rc) IFEM_NCCL_CHECK(ncclRecv(x + (n_owned + rptr[k]) * bs, rc, ncclDouble, h.nbr[k], (ncclComm_t)h.comm, ctx->stream));
  }
  IFEM_NCCL_CHECK(ncclGroupEnd());
}

void halo_exchange(ifem_ctx *ctx, double *xu_ext) {
  if (ctx->halo.nranks == 1) return;
  exchange(ctx, xu_ext, ctx->dim, ctx->nUo, ctx->halo.send_u_ptr, ctx->halo.send_u_idx, ctx->halo.recv_u_ptr,
           ctx->halo.sendbuf.p);
}

void halo_exchange_p(ifem_ctx *ctx, double *xp_ext) {
  if (ctx->halo.nranks == 1) return;
  exchange(ctx, xp_ext, 1, ctx->nPo, ctx->halo.send_p_ptr, ctx->halo.send_p_idx, ctx->halo.recv_p_ptr,
           ctx->halo.sendbuf.p + (size_t)ctx->dim * ctx->halo.send_u_ptr.back());
}

int comm_unique_id(uint8_t out[128]) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return IFEM_E_COMM;
  std::memcpy(out, &id, 128);
  return IFEM_OK;
}

static void allreduce(ifem_ctx *ctx, double *host_vals, int n, ncclRedOp_t op);
void allreduce_sum(ifem_ctx *ctx, double *host_vals, int n) { allreduce(ctx, host_vals, n, ncclSum); }
void allreduce_max(ifem_ctx *ctx, double *host_vals, int n) { allreduce(ctx, host_vals, n, ncclMax); }

static void allreduce(ifem_ctx *ctx, double *host_vals, int n, ncclRedOp_t op) {
  if (ctx->halo.nranks == 1) return;
  hipStream_t s = ctx->stream;
  double *d = ctx->scal.p + 128;
  std::memcpy(ctx->h_scal + 128, host_vals, n * sizeof(double));
  IFEM_HIP_CHECK(hipMemcpyAsync(d, ctx->h_scal + 128, n * sizeof(double), hipMemcpyHostToDevice, s));
  IFEM_NCCL_CHECK(ncclAllReduce(d, d, n, ncclDouble, op, (ncclComm_t)ctx->halo.comm, s));
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal + 128, d, n * sizeof(double), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  std::memcpy(host_vals, ctx->h_scal + 128, n * sizeof(double));
}

} // namespace ifem
