// comm.hip -- the two collectives of the fluid step:
//   halo exchange of a DoF vector   (PETSc VecScatter in MatMult and ghosted-vector assignment)
//   all-reduce of a few scalars     (l2_norm(), Krylov dot products)
// Transport 1 (product): RCCL over xGMI, one process per GPU.  Point-to-point ncclSend/ncclRecv grouped per
// exchange: one message per neighbour = one per xGMI link for an octant partition.
// Transport 2 (validation): "local world" -- several contexts (virtual ranks, one host thread each) inside ONE
// process sharing one GPU; exchanges are device-to-device copies between the contexts, rendezvous by a host
// barrier.  It exists so that the partitioned algorithm (ownership, ghost layers, halo plans, distributed Krylov)
// can be validated on a single-GPU box; it shares everything with the RCCL path except the transport calls.
// Single-rank runs never touch either.
#include <hip/hip_runtime.h>
#include <chrono>
#include <rccl/rccl.h>
#include <algorithm>
#include <array>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <atomic>
#include <thread>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

#define IFEM_NCCL_CHECK(expr)                                                                              \
  do {                                                                                                     \
    ncclResult_t r_ = (expr);                                                                              \
    if (r_ != ncclSuccess)                                                                                 \
      throw ::ifem::Error(IFEM_E_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));                \
  } while (0)

struct LocalWorld {
  int nranks;
  std::mutex mu;
  std::condition_variable cv;
  int waiting = 0;
  long generation = 0;
  std::vector<ifem_ctx *> ctx;           // published contexts
  std::vector<std::vector<double>> red;  // all-reduce staging
  // stream-ordered exchanges: rank r records ev_packed[r] behind its packing kernel and ev_copied[r] behind its copies out of
  // the peers' send buffers; the peers' streams wait on those events, no host thread waits for the device
  std::vector<hipEvent_t> ev_packed, ev_copied;
  std::vector<const double *> red_ptr; // device scalars of every rank during a stream-ordered all-reduce
  std::vector<const void *> red_vec;   // ... and device vectors (hand-over to a replicated coarse level)
  explicit LocalWorld(int n) : nranks(n), ctx(n, nullptr), red(n), ev_packed(n, nullptr), ev_copied(n, nullptr), red_ptr(n, nullptr), red_vec(n, nullptr) {}
  ~LocalWorld() {
    for (auto e : ev_packed) if (e) (void)hipEventDestroy(e);
    for (auto e : ev_copied) if (e) (void)hipEventDestroy(e);
  }
  // A world whose `aborted` flag is up (a rank left an exchange by an exception) is DEAD: every later barrier / rendezvous of it
  // fails with IFEM_E_COMM on every rank instead of waiting for the rank that will never come; the flag is never cleared --
  // destroy the contexts and the world and build new ones.
  void barrier() {
    std::unique_lock<std::mutex> lk(mu);
    if (aborted.load(std::memory_order_acquire)) throw Error(IFEM_E_COMM, "local world: a peer rank aborted an exchange (the world is dead)");
    const long gen = generation;
    if (++waiting == nranks) { waiting = 0; ++generation; cv.notify_all(); }
    else
      while (!cv.wait_for(lk, std::chrono::milliseconds(20), [&] { return generation != gen; }))
        if (aborted.load(std::memory_order_acquire)) { --waiting; throw Error(IFEM_E_COMM, "local world: a peer rank aborted an exchange (the world is dead)"); }
  }
  // host-only rendezvous of the exchanges (no device synchronisation around it): spin briefly, then yield
  std::atomic<long> spin_count{0};
  std::atomic<long> spin_gen{0};
  std::atomic<bool> aborted{false}; // a rank left an exchange by an exception: its peers must not wait for it for ever
  void rendezvous() {
    const long gen = spin_gen.load(std::memory_order_acquire);
    if (spin_count.fetch_add(1, std::memory_order_acq_rel) + 1 == nranks) {
      spin_count.store(0, std::memory_order_relaxed);
      spin_gen.store(gen + 1, std::memory_order_release);
    } else {
      int spins = 0;
      while (spin_gen.load(std::memory_order_acquire) == gen) {
        if (aborted.load(std::memory_order_acquire)) throw Error(IFEM_E_COMM, "local world: a peer rank aborted an exchange");
        if (++spins > 2000) std::this_thread::yield();
      }
    }
  }
  // the halo plans of all ranks against each other, once (comm_init, every context registered): what the exchanges would
  // otherwise find out between two rendezvous, leaving the peers of the throwing rank spinning
  void validate(int rank) const {
    const ifem_ctx *c = ctx[rank];
    const Halo &h = c->halo;
    for (size_t k = 0; k < h.nbr.size(); ++k) {
      const int pr = h.nbr[k];
      if (pr < 0 || pr >= nranks || !ctx[pr]) throw Error(IFEM_E_COMM, "local world: neighbour rank out of range");
      const ifem_ctx *peer = ctx[pr];
      if (peer->device != c->device) throw Error(IFEM_E_COMM, "local world: all contexts must live on one device (the all-reduce reads the peers' scalars directly)");
      const Halo &ph = peer->halo;
      int me = -1;
      for (size_t j = 0; j < ph.nbr.size(); ++j) if (ph.nbr[j] == rank) me = (int)j;
      if (me < 0) throw Error(IFEM_E_COMM, "local world: neighbour lists are not symmetric");
      if (ph.send_u_ptr[me + 1] - ph.send_u_ptr[me] != h.recv_u_ptr[k + 1] - h.recv_u_ptr[k] ||
          ph.send_p_ptr[me + 1] - ph.send_p_ptr[me] != h.recv_p_ptr[k + 1] - h.recv_p_ptr[k])
        throw Error(IFEM_E_COMM, "local world: send/recv count mismatch between rank " + std::to_string(rank) + " and " + std::to_string(pr));
      if (h.has_s != ph.has_s || (h.has_s && ph.send_s_ptr[me + 1] - ph.send_s_ptr[me] != h.recv_s_ptr[k + 1] - h.recv_s_ptr[k]))
        throw Error(IFEM_E_COMM, "local world: 2-deep pressure halo plans do not match");
    }
  }
};

// Communicators are shared by the contexts of one process that were given the same unique id (the levels of a multigrid
// chain: same ranks, same neighbours, one stream): ncclCommInitRank is collective and allocates its channel buffers per
// communicator, one per level would multiply both.  Reference-counted; the last context destroys it.
struct SharedComm { ncclComm_t comm, comm2; int refs; };
static std::mutex g_comm_mu;
static std::map<std::array<uint8_t, 128>, SharedComm> g_comms;

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
  if (h.recv_u_ptr[nn] != ctx->nUl - ctx->nUo || h.recv_p_ptr[nn] != ctx->nPl - ctx->nPo)
    throw Error(IFEM_E_BADPARAM, "ifem_partition: receive counts do not match the number of ghost nodes");
  h.send_u_idx.upload(part->send_u_idx, h.send_u_ptr[nn], ctx->stream);
  h.send_p_idx.upload(part->send_p_idx, h.send_p_ptr[nn], ctx->stream);
  size_t n_send_s = 0;
  if (part->sm_box_id && part->l2g_p && part->send_s_ptr && part->recv_s_ptr) {
    h.send_s_ptr.assign(part->send_s_ptr, part->send_s_ptr + nn + 1);
    h.recv_s_ptr.assign(part->recv_s_ptr, part->recv_s_ptr + nn + 1);
    n_send_s = (size_t)h.send_s_ptr[nn];
    h.send_s_idx.upload(part->send_s_idx, n_send_s, ctx->stream);
    for (int d = 0; d < 3; ++d) { h.p_lattice_n[d] = part->p_lattice_n[d]; h.sm_box_lo[d] = part->sm_box_lo[d]; h.sm_box_n[d] = part->sm_box_n[d]; }
    h.sm_box_id.upload(part->sm_box_id, (size_t)(h.sm_box_n[0] * h.sm_box_n[1] * h.sm_box_n[2]), ctx->stream);
    h.own_p_gid.upload(part->l2g_p, (size_t)ctx->nPo, ctx->stream);
    h.n_s_cols = ctx->nPo + h.recv_s_ptr[nn];
    h.has_s = true;
  }
  h.sendbuf.alloc((size_t)ctx->dim * h.send_u_ptr[nn] + h.send_p_ptr[nn] + n_send_s + 8);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  if (part->local_world) {
    auto *w = static_cast<LocalWorld *>(part->local_world);
    if (w->nranks != h.nranks) throw Error(IFEM_E_BADPARAM, "local world size mismatch");
    h.local = w;
    w->ctx[h.rank] = ctx;
    if (!w->ev_packed[h.rank]) {
      IFEM_HIP_CHECK(hipEventCreateWithFlags(&w->ev_packed[h.rank], hipEventDisableTiming));
      IFEM_HIP_CHECK(hipEventCreateWithFlags(&w->ev_copied[h.rank], hipEventDisableTiming));
    }
    w->barrier();
    // every rank validates (and every rank passes the second barrier even if it throws afterwards: no peer is left waiting)
    std::string bad;
    try { w->validate(h.rank); } catch (const Error &e) { bad = e.what(); }
    w->barrier();
    if (!bad.empty()) throw Error(IFEM_E_COMM, bad);
  } else {
    if (!part->nccl_unique_id) throw Error(IFEM_E_BADPARAM, "ifem_partition needs nccl_unique_id or local_world");
    ncclUniqueId id;
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId size");
    std::memcpy(&id, part->nccl_unique_id, 128);
    std::array<uint8_t, 128> key;
    std::memcpy(key.data(), part->nccl_unique_id, 128);
    std::lock_guard<std::mutex> lk(g_comm_mu);
    auto it = g_comms.find(key);
    if (it != g_comms.end()) {
      int cr = -1, cn = -1;
      IFEM_NCCL_CHECK(ncclCommUserRank(it->second.comm, &cr));
      IFEM_NCCL_CHECK(ncclCommCount(it->second.comm, &cn));
      if (cr != h.rank || cn != h.nranks) throw Error(IFEM_E_BADPARAM, "ifem_partition: this unique id already names a communicator of another rank / size");
      it->second.refs++;
      h.comm = it->second.comm;
      h.comm2 = it->second.comm2;
    } else {
      ncclComm_t comm, comm2 = nullptr;
      IFEM_NCCL_CHECK(ncclCommInitRank(&comm, h.nranks, id, h.rank));
      // the halo traffic gets a communicator of its own: its groups run on another stream than the all-reduces of `comm`.
      // Without it (an RCCL that cannot split) the exchanges simply stay on the context stream.
      if (ncclCommSplit(comm, 0, h.rank, &comm2, nullptr) != ncclSuccess) comm2 = nullptr;
      g_comms[key] = SharedComm{comm, comm2, 1};
      h.comm = comm;
      h.comm2 = comm2;
    }
    if (h.comm2) {
      int lo = 0, hi = 0;
      IFEM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi)); // hi = numerically lowest = highest priority
      IFEM_HIP_CHECK(hipStreamCreateWithPriority(&h.hstream, hipStreamNonBlocking, hi));
      h.owns_hstream = true;
      IFEM_HIP_CHECK(hipEventCreateWithFlags(&h.ev_pack, hipEventDisableTiming));
      IFEM_HIP_CHECK(hipEventCreateWithFlags(&h.ev_done, hipEventDisableTiming));
    }
  }
}

void comm_destroy(ifem_ctx *ctx) {
  Halo &hh = ctx->halo;
  if (hh.ev_pack) (void)hipEventDestroy(hh.ev_pack);
  if (hh.ev_done) (void)hipEventDestroy(hh.ev_done);
  if (hh.hstream && hh.owns_hstream) (void)hipStreamDestroy(hh.hstream);
  hh.ev_pack = hh.ev_done = nullptr; hh.hstream = nullptr; hh.comm2 = nullptr;
  if (ctx->halo.comm) {
    std::lock_guard<std::mutex> lk(g_comm_mu);
    for (auto it = g_comms.begin(); it != g_comms.end(); ++it)
      if (it->second.comm == (ncclComm_t)ctx->halo.comm) {
        if (--it->second.refs == 0) {
          if (it->second.comm2) ncclCommDestroy(it->second.comm2);
          ncclCommDestroy(it->second.comm);
          g_comms.erase(it);
        }
        break;
      }
  }
  ctx->halo.comm = nullptr;
}

template <typename T>
__global__ void k_pack(int64_t n, int bs, const int32_t *__restrict__ idx, const T *__restrict__ x, T *__restrict__ buf) {
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n * bs; t += int64_t(gridDim.x) * blockDim.x) {
    const int64_t i = t / bs;
    const int c = int(t - i * bs);
    buf[t] = x[int64_t(idx[i]) * bs + c];
  }
}

// which: 0 = velocity nodes (bs = dim), 1 = pressure nodes (bs = 1)
// which: 0 velocity halo, 1 pressure halo, 2 the 2-deep pressure halo of the distributed S_m
// async: the send/recv group goes to the halo stream (second communicator) behind an event recorded after the packing
// kernel, and leaves an event for halo_wait; otherwise everything is ordered on the context stream
template <typename T> struct NcclType;
template <> struct NcclType<double> { static constexpr ncclDataType_t v = ncclDouble; };
template <> struct NcclType<float> { static constexpr ncclDataType_t v = ncclFloat; };

// T = double for the vectors of the Krylov solvers, float for the level vectors of the A_uu V-cycle (the send buffer is
// sized in doubles and reinterpreted)
template <typename T>
static void exchange(ifem_ctx *ctx, T *x, int which, bool async = false) {
  Halo &h = ctx->halo;
  const int bs = which == 0 ? ctx->dim : 1;
  const int64_t n_owned = which == 0 ? ctx->nUo : ctx->nPo;
  const std::vector<int32_t> &sptr = which == 0 ? h.send_u_ptr : (which == 1 ? h.send_p_ptr : h.send_s_ptr);
  const std::vector<int32_t> &rptr = which == 0 ? h.recv_u_ptr : (which == 1 ? h.recv_p_ptr : h.recv_s_ptr);
  const DBuf<int32_t> &sidx = which == 0 ? h.send_u_idx : (which == 1 ? h.send_p_idx : h.send_s_idx);
  auto sendbuf_of = [&](ifem_ctx *c) {
    size_t off = 0;
    if (which >= 1) off += (size_t)c->dim * c->halo.send_u_ptr.back();
    if (which >= 2) off += (size_t)c->halo.send_p_ptr.back();
    return reinterpret_cast<T *>(c->halo.sendbuf.p) + off;
  };
  T *sendbuf = sendbuf_of(ctx);
  const int nn = (int)h.nbr.size();
  const int64_t ns = sptr[nn];
  ++h.n_exchanges;
  if (ns) {
    int64_t g = (ns * bs + 255) / 256;
    hipLaunchKernelGGL((k_pack<T>), dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, ctx->stream, ns, bs, sidx.p, (const T *)x, sendbuf);
  }
  if (h.local) {
    // stream-ordered like the RCCL path: my copies wait (on the device) for the peers' packing kernels, the peers' next
    // packing kernels wait for my copies; the host threads only meet to know that the events have been recorded
    auto *w = static_cast<LocalWorld *>(h.local);
    try {
    IFEM_HIP_CHECK(hipEventRecord(w->ev_packed[h.rank], ctx->stream)); // my send buffer is complete at this point of my stream
    w->rendezvous();
    for (int k = 0; k < nn; ++k) {
      const int64_t rc = int64_t(rptr[k + 1] - rptr[k]) * bs;
      if (!rc) continue;
      ifem_ctx *peer = w->ctx[h.nbr[k]];
      const Halo &ph = peer->halo;
      int me = -1;
      for (size_t j = 0; j < ph.nbr.size(); ++j) if (ph.nbr[j] == h.rank) me = (int)j;
      if (me < 0) throw Error(IFEM_E_COMM, "local world: neighbour lists are not symmetric");
      const std::vector<int32_t> &psptr = which == 0 ? ph.send_u_ptr : (which == 1 ? ph.send_p_ptr : ph.send_s_ptr);
      if (int64_t(psptr[me + 1] - psptr[me]) * bs != rc) throw Error(IFEM_E_COMM, "local world: send/recv count mismatch");
      IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, w->ev_packed[h.nbr[k]], 0));
      IFEM_HIP_CHECK(hipMemcpyAsync(x + (n_owned + rptr[k]) * bs, sendbuf_of(peer) + int64_t(psptr[me]) * bs,
                                    rc * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    }
    IFEM_HIP_CHECK(hipEventRecord(w->ev_copied[h.rank], ctx->stream));
    w->rendezvous();
    // whatever I launch next (the next packing kernel first of all) must not overwrite what a peer is still reading
    for (int k = 0; k < nn; ++k) IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, w->ev_copied[h.nbr[k]], 0));
    } catch (...) { w->aborted.store(true, std::memory_order_release); throw; } // the peers leave their rendezvous with an error
    return;
  }
  async = async && h.comm2 && h.hstream;
  hipStream_t st = async ? h.hstream : ctx->stream;
  ncclComm_t cm = (ncclComm_t)(async ? h.comm2 : h.comm);
  if (async) { // the halo stream picks up behind the packing kernel (and everything before it on the context stream)
    IFEM_HIP_CHECK(hipEventRecord(h.ev_pack, ctx->stream));
    IFEM_HIP_CHECK(hipStreamWaitEvent(st, h.ev_pack, 0));
  }
  IFEM_NCCL_CHECK(ncclGroupStart());
  for (int k = 0; k < nn; ++k) {
    const int64_t sc = int64_t(sptr[k + 1] - sptr[k]) * bs, rc = int64_t(rptr[k + 1] - rptr[k]) * bs;
    if (sc) IFEM_NCCL_CHECK(ncclSend(sendbuf + int64_t(sptr[k]) * bs, sc, NcclType<T>::v, h.nbr[k], cm, st));
    if (rc) IFEM_NCCL_CHECK(ncclRecv(x + (n_owned + rptr[k]) * bs, rc, NcclType<T>::v, h.nbr[k], cm, st));
  }
  IFEM_NCCL_CHECK(ncclGroupEnd());
  if (async) IFEM_HIP_CHECK(hipEventRecord(h.ev_done, st));
}

// Overlapped halo exchange: halo_start packs on the context stream and lets the transfers run on the halo stream;
// kernels launched on the context stream up to halo_wait must not read ghost entries of x (nor write its owned ones).
// which: 0 velocity, 1 pressure, 2 the 2-deep pressure halo of S_m.  The validation transport exchanges synchronously
// in halo_start (the split of the consumers is exercised all the same).
bool halo_overlap_ok(const ifem_ctx *ctx) {
  const Halo &h = ctx->halo;
  return h.nranks > 1 && ctx->tune.halo_overlap && (h.local || (h.comm2 && h.hstream));
}
void halo_start(ifem_ctx *ctx, double *x_ext, int which) {
  if (ctx->halo.nranks == 1) return;
  if (which == 2 && !ctx->halo.has_s) throw Error(IFEM_E_BADPARAM, "no 2-deep pressure halo plan in this context");
  exchange(ctx, x_ext, which, true);
}
void halo_start_f32(ifem_ctx *ctx, float *xu_ext) {
  if (ctx->halo.nranks == 1) return;
  exchange(ctx, xu_ext, 0, true);
}
void halo_wait(ifem_ctx *ctx) {
  Halo &h = ctx->halo;
  if (h.nranks == 1 || h.local || !h.comm2 || !h.hstream) return;
  IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, h.ev_done, 0));
}

template <typename T>
__global__ void k_unpack_add(int64_t n, int bs, const int32_t *__restrict__ idx, const T *__restrict__ buf, T *__restrict__ x) {
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n * bs; t += int64_t(gridDim.x) * blockDim.x) {
    const int64_t i = t / bs;
    const int c = int(t - i * bs);
    unsafeAtomicAdd(&x[int64_t(idx[i]) * bs + c], buf[t]); // a node may be a ghost of several neighbours
  }
}

// Transpose of the halo exchange: the ghost entries of x_ext travel back to their owners and are ADDED to the owned
// entries (PETSc VecScatter in reverse / ADD_VALUES mode; deal.II compress(VectorOperation::add)).  Used by C^T of the
// hanging-node lines whose masters live on another rank.  which: 0 velocity nodes (bs = dim), 1 pressure nodes.
template <typename T>
static void reverse_add(ifem_ctx *ctx, T *x, int which) {
  Halo &h = ctx->halo;
  const int bs = which == 0 ? ctx->dim : 1;
  const int64_t n_owned = which == 0 ? ctx->nUo : ctx->nPo;
  const std::vector<int32_t> &sptr = which == 0 ? h.send_u_ptr : h.send_p_ptr;
  const std::vector<int32_t> &rptr = which == 0 ? h.recv_u_ptr : h.recv_p_ptr;
  const DBuf<int32_t> &sidx = which == 0 ? h.send_u_idx : h.send_p_idx;
  T *buf = reinterpret_cast<T *>(h.sendbuf.p) + (which == 0 ? 0 : (size_t)ctx->dim * h.send_u_ptr.back()); // the forward send region, now receiving
  const int nn = (int)h.nbr.size();
  const int64_t ns = sptr[nn];
  ++h.n_exchanges;
  if (h.local) {
    auto *w = static_cast<LocalWorld *>(h.local);
    h.rev_src = x;
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // my ghost entries are complete
    w->barrier();
    for (int k = 0; k < nn; ++k) {
      const int64_t sc = int64_t(sptr[k + 1] - sptr[k]) * bs;
      if (!sc) continue;
      ifem_ctx *peer = w->ctx[h.nbr[k]];
      const Halo &ph = peer->halo;
      int me = -1;
      for (size_t j = 0; j < ph.nbr.size(); ++j) if (ph.nbr[j] == h.rank) me = (int)j;
      if (me < 0) throw Error(IFEM_E_COMM, "local world: neighbour lists are not symmetric");
      const std::vector<int32_t> &prptr = which == 0 ? ph.recv_u_ptr : ph.recv_p_ptr;
      const int64_t pn_owned = which == 0 ? peer->nUo : peer->nPo;
      if (int64_t(prptr[me + 1] - prptr[me]) * bs != sc) throw Error(IFEM_E_COMM, "local world: send/recv count mismatch");
      IFEM_HIP_CHECK(hipMemcpyAsync(buf + int64_t(sptr[k]) * bs, static_cast<const T *>(ph.rev_src) + (pn_owned + prptr[me]) * bs, sc * sizeof(T),
                                    hipMemcpyDeviceToDevice, ctx->stream));
    }
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    w->barrier(); // peers may now reuse their ghost entries
  } else {
    IFEM_NCCL_CHECK(ncclGroupStart());
    for (int k = 0; k < nn; ++k) {
      const int64_t sc = int64_t(sptr[k + 1] - sptr[k]) * bs, rc = int64_t(rptr[k + 1] - rptr[k]) * bs;
      if (rc) IFEM_NCCL_CHECK(ncclSend(x + (n_owned + rptr[k]) * bs, rc, NcclType<T>::v, h.nbr[k], (ncclComm_t)h.comm, ctx->stream));
      if (sc) IFEM_NCCL_CHECK(ncclRecv(buf + int64_t(sptr[k]) * bs, sc, NcclType<T>::v, h.nbr[k], (ncclComm_t)h.comm, ctx->stream));
    }
    IFEM_NCCL_CHECK(ncclGroupEnd());
  }
  if (ns) {
    int64_t g = (ns * bs + 255) / 256;
    hipLaunchKernelGGL((k_unpack_add<T>), dim3((unsigned)(g > 4096 ? 4096 : g)), dim3(256), 0, ctx->stream, ns, bs, sidx.p, (const T *)buf, x);
  }
}

void halo_reverse_add(ifem_ctx *ctx, double *xu_ext) {
  if (ctx->halo.nranks == 1) return;
  reverse_add(ctx, xu_ext, 0);
}
void halo_reverse_add_f32(ifem_ctx *ctx, float *xu_ext) {
  if (ctx->halo.nranks == 1) return;
  reverse_add(ctx, xu_ext, 0);
}
void halo_exchange_f32(ifem_ctx *ctx, float *xu_ext) {
  if (ctx->halo.nranks == 1) return;
  exchange(ctx, xu_ext, 0);
}
void halo_reverse_add_p(ifem_ctx *ctx, double *xp_ext) {
  if (ctx->halo.nranks == 1) return;
  reverse_add(ctx, xp_ext, 1);
}

void halo_exchange(ifem_ctx *ctx, double *xu_ext) {
  if (ctx->halo.nranks == 1) return;
  exchange(ctx, xu_ext, 0);
}

void halo_exchange_s(ifem_ctx *ctx, double *xs_ext) {
  if (ctx->halo.nranks == 1) return;
  if (!ctx->halo.has_s) throw Error(IFEM_E_BADPARAM, "no 2-deep pressure halo plan in this context");
  exchange(ctx, xs_ext, 2);
}

void halo_exchange_p(ifem_ctx *ctx, double *xp_ext) {
  if (ctx->halo.nranks == 1) return;
  exchange(ctx, xp_ext, 1);
}

int comm_unique_id(uint8_t out[128]) {
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return IFEM_E_COMM;
  std::memcpy(out, &id, 128);
  return IFEM_OK;
}

static void allreduce(ifem_ctx *ctx, double *host_vals, int n, bool is_max) {
  Halo &h = ctx->halo;
  if (h.nranks == 1) return;
  ++h.n_allreduce_host;
  if (h.local) {
    auto *w = static_cast<LocalWorld *>(h.local);
    w->red[h.rank].assign(host_vals, host_vals + n);
    w->barrier();
    std::vector<double> out(w->red[0]);
    for (int r = 1; r < h.nranks; ++r)
      for (int i = 0; i < n; ++i) out[i] = is_max ? std::max(out[i], w->red[r][i]) : out[i] + w->red[r][i];
    w->barrier(); // everybody has read the staging slots
    std::memcpy(host_vals, out.data(), n * sizeof(double));
    return;
  }
  // the staging slots hold kScalStage doubles (scal / h_scal[kScalStageOff...]): longer lists (the 200-column inner GMRES of
  // scns_solve, a user fgmres_restart >= 128) go through in chunks
  hipStream_t s = ctx->stream;
  double *d = ctx->scal.p + kScalStageOff;
  for (int first = 0; first < n; first += kScalStage) {
    const int m = std::min(kScalStage, n - first);
    std::memcpy(ctx->h_scal + kScalStageOff, host_vals + first, m * sizeof(double));
    IFEM_HIP_CHECK(hipMemcpyAsync(d, ctx->h_scal + kScalStageOff, m * sizeof(double), hipMemcpyHostToDevice, s));
    IFEM_NCCL_CHECK(ncclAllReduce(d, d, m, ncclDouble, is_max ? ncclMax : ncclSum, (ncclComm_t)h.comm, s));
    IFEM_HIP_CHECK(hipMemcpyAsync(ctx->h_scal + kScalStageOff, d, m * sizeof(double), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
    std::memcpy(host_vals + first, ctx->h_scal + kScalStageOff, m * sizeof(double));
  }
}

void allreduce_sum(ifem_ctx *ctx, double *host_vals, int n) { allreduce(ctx, host_vals, n, false); }

constexpr int kMaxLocalPeers = 8;
struct PeerPtrs { const double *p[kMaxLocalPeers]; };
__global__ void k_sum_peers(int n, int nranks, PeerPtrs pp, double *__restrict__ out) {
  const int i = threadIdx.x;
  if (i >= n) return;
  double s = 0;
  for (int r = 0; r < nranks; ++r) s += pp.p[r][i];
  out[i] = s;
}

// Sum of n DEVICE-resident scalars over the ranks, in place, ordered on the context stream and without a host round trip
// (RCCL): what the device-resident CG recurrences use between their partial reductions and the scalar update.  The
// validation transport has no device-side collective: it synchronises and goes through the host path.
void allreduce_sum_dev(ifem_ctx *ctx, double *dev_vals, int n) {
  Halo &h = ctx->halo;
  if (h.nranks == 1 || n <= 0) return;
  ++h.n_allreduce_dev;
  if (h.local) {
    auto *w = static_cast<LocalWorld *>(h.local);
    if (h.nranks > kMaxLocalPeers || n > 64) { // beyond the fixed-size argument block: through the host
      std::vector<double> tmp(n);
      IFEM_HIP_CHECK(hipMemcpyAsync(tmp.data(), dev_vals, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
      IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
      allreduce(ctx, tmp.data(), n, false);
      --h.n_allreduce_host; // counted as a device one above
      IFEM_HIP_CHECK(hipMemcpyAsync(dev_vals, tmp.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
      IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // tmp leaves scope
      return;
    }
    // stream-ordered, like ncclAllReduce on the device scalars: every rank sums the peers' values (in rank order: the same
    // result everywhere) into a staging slot once their producers have run, and takes it over once everybody has read
    w->red_ptr[h.rank] = dev_vals;
    IFEM_HIP_CHECK(hipEventRecord(w->ev_packed[h.rank], ctx->stream));
    w->rendezvous();
    PeerPtrs pp{};
    for (int r = 0; r < h.nranks; ++r) {
      pp.p[r] = w->red_ptr[r];
      if (r != h.rank) IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, w->ev_packed[r], 0));
    }
    double *stage = ctx->scal.p + kScalStageOff + kScalStage - 64; // the tail of the staging slots (n <= 64)
    hipLaunchKernelGGL(k_sum_peers, dim3(1), dim3(64), 0, ctx->stream, n, h.nranks, pp, stage);
    IFEM_HIP_CHECK(hipEventRecord(w->ev_copied[h.rank], ctx->stream));
    w->rendezvous();
    for (int r = 0; r < h.nranks; ++r)
      if (r != h.rank) IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, w->ev_copied[r], 0));
    IFEM_HIP_CHECK(hipMemcpyAsync(dev_vals, stage, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    return;
  }
  IFEM_NCCL_CHECK(ncclAllReduce(dev_vals, dev_vals, n, ncclDouble, ncclSum, (ncclComm_t)h.comm, ctx->stream));
}
void allreduce_max(ifem_ctx *ctx, double *host_vals, int n) { allreduce(ctx, host_vals, n, true); }

// ---- vector all-reduce: the hand-over to a replicated coarse level (ifem_mg_attach with a single-rank coarse context).  Every rank
// holds its partial restriction over ALL coarse nodes; the sum is the restricted residual, identical on every rank (RCCL rings and
// the rank-ordered sum below both give every rank the same bits).  The replicas themselves agree to rounding only (their level
// operators use atomics whose order differs between devices); solver.hip::replica_world agrees the smoother bounds across ranks.
struct PeerVecs { const void *p[kMaxLocalPeers]; };
template <typename T>
__global__ void k_sum_peer_vecs(int64_t n, int nranks, PeerVecs pp, T *__restrict__ out) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n; i += int64_t(gridDim.x) * blockDim.x) {
    T s = 0;
    for (int r = 0; r < nranks; ++r) s += static_cast<const T *>(pp.p[r])[i];
    out[i] = s;
  }
}
template <typename T>
static void allreduce_vec_t(ifem_ctx *ctx, T *dev, int64_t n, T *scratch) {
  Halo &h = ctx->halo;
  if (h.nranks == 1 || n <= 0) return;
  ++h.n_allreduce_vec;
  if (h.local) {
    auto *w = static_cast<LocalWorld *>(h.local);
    if (h.nranks > kMaxLocalPeers) throw Error(IFEM_E_COMM, "local world: a replicated coarse level needs at most 8 virtual ranks");
    // same protocol as the stream-ordered scalar all-reduce: publish, sum the peers' vectors into scratch once their producers
    // have run, take the sum over once everybody has read
    try {
      w->red_vec[h.rank] = dev;
      IFEM_HIP_CHECK(hipEventRecord(w->ev_packed[h.rank], ctx->stream));
      w->rendezvous();
      PeerVecs pp{};
      for (int r = 0; r < h.nranks; ++r) {
        pp.p[r] = w->red_vec[r];
        if (r != h.rank) IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, w->ev_packed[r], 0));
      }
      const unsigned g = unsigned(std::min<int64_t>((n + 255) / 256, 4096));
      hipLaunchKernelGGL((k_sum_peer_vecs<T>), dim3(g), dim3(256), 0, ctx->stream, n, h.nranks, pp, scratch);
      IFEM_HIP_CHECK(hipEventRecord(w->ev_copied[h.rank], ctx->stream));
      w->rendezvous();
      for (int r = 0; r < h.nranks; ++r)
        if (r != h.rank) IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->stream, w->ev_copied[r], 0));
      IFEM_HIP_CHECK(hipMemcpyAsync(dev, scratch, size_t(n) * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream));
    } catch (...) { w->aborted.store(true, std::memory_order_release); throw; } // the peers leave their rendezvous with an error
    return;
  }
  IFEM_NCCL_CHECK(ncclAllReduce(dev, dev, size_t(n), sizeof(T) == 4 ? ncclFloat : ncclDouble, ncclSum, (ncclComm_t)h.comm, ctx->stream));
}
void allreduce_sum_vec(ifem_ctx *ctx, double *dev, int64_t n, double *scratch) { allreduce_vec_t<double>(ctx, dev, n, scratch); }
void allreduce_sum_vec_f32(ifem_ctx *ctx, float *dev, int64_t n, float *scratch) { allreduce_vec_t<float>(ctx, dev, n, scratch); }

// One-rank RCCL round trip (communicator, all-reduce, grouped send/recv to self) on `device`: checks on a
// single-GPU box that the library, the RCCL it resolves at run time and the stream semantics fit together.
int comm_selftest(int device) {
  IFEM_HIP_CHECK(hipSetDevice(device));
  hipStream_t s;
  IFEM_HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  ncclUniqueId id;
  IFEM_NCCL_CHECK(ncclGetUniqueId(&id));
  ncclComm_t comm;
  IFEM_NCCL_CHECK(ncclCommInitRank(&comm, 1, id, 0));
  DBuf<double> a, b;
  a.alloc(64); b.alloc(64);
  std::vector<double> h(64);
  for (int i = 0; i < 64; ++i) h[i] = i + 0.5;
  IFEM_HIP_CHECK(hipMemcpyAsync(a.p, h.data(), 64 * 8, hipMemcpyHostToDevice, s));
  IFEM_NCCL_CHECK(ncclAllReduce(a.p, a.p, 64, ncclDouble, ncclSum, comm, s));
  IFEM_NCCL_CHECK(ncclGroupStart());
  IFEM_NCCL_CHECK(ncclSend(a.p, 64, ncclDouble, 0, comm, s));
  IFEM_NCCL_CHECK(ncclRecv(b.p, 64, ncclDouble, 0, comm, s));
  IFEM_NCCL_CHECK(ncclGroupEnd());
  std::vector<double> out(64);
  IFEM_HIP_CHECK(hipMemcpyAsync(out.data(), b.p, 64 * 8, hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  // the overlapped exchange's plumbing (comm_init / exchange(async)): split communicator, high-priority stream, the
  // event pair, a send/recv group on the second stream and a stream-ordered all-reduce of device scalars
  ncclComm_t comm2 = nullptr;
  IFEM_NCCL_CHECK(ncclCommSplit(comm, 0, 0, &comm2, nullptr));
  int lo = 0, hi = 0;
  IFEM_HIP_CHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t hs;
  IFEM_HIP_CHECK(hipStreamCreateWithPriority(&hs, hipStreamNonBlocking, hi));
  hipEvent_t e0, e1;
  IFEM_HIP_CHECK(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
  IFEM_HIP_CHECK(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
  DBuf<double> c2;
  c2.alloc(64);
  IFEM_HIP_CHECK(hipMemsetAsync(c2.p, 0, 64 * 8, s));
  IFEM_HIP_CHECK(hipEventRecord(e0, s));
  IFEM_HIP_CHECK(hipStreamWaitEvent(hs, e0, 0));
  IFEM_NCCL_CHECK(ncclGroupStart());
  IFEM_NCCL_CHECK(ncclSend(b.p, 64, ncclDouble, 0, comm2, hs));
  IFEM_NCCL_CHECK(ncclRecv(c2.p, 64, ncclDouble, 0, comm2, hs));
  IFEM_NCCL_CHECK(ncclGroupEnd());
  IFEM_HIP_CHECK(hipEventRecord(e1, hs));
  IFEM_NCCL_CHECK(ncclAllReduce(a.p, a.p, 2, ncclDouble, ncclSum, comm, s)); // meanwhile on the first communicator
  IFEM_HIP_CHECK(hipStreamWaitEvent(s, e1, 0));
  std::vector<double> out2(64);
  IFEM_HIP_CHECK(hipMemcpyAsync(out2.data(), c2.p, 64 * 8, hipMemcpyDeviceToHost, s));
  // the hand-over to a replicated coarse level (allreduce_sum_vec_f32): an in-place single-precision all-reduce of a whole level vector
  const size_t nv = size_t(1) << 20;
  DBuf<float> fv;
  fv.alloc(nv);
  std::vector<float> hf(nv);
  for (size_t i = 0; i < nv; ++i) hf[i] = float(i % 1021) * 0.25f;
  IFEM_HIP_CHECK(hipMemcpyAsync(fv.p, hf.data(), nv * sizeof(float), hipMemcpyHostToDevice, s));
  IFEM_NCCL_CHECK(ncclAllReduce(fv.p, fv.p, nv, ncclFloat, ncclSum, comm, s));
  std::vector<float> of(nv);
  IFEM_HIP_CHECK(hipMemcpyAsync(of.data(), fv.p, nv * sizeof(float), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  for (size_t i = 0; i < nv; ++i) if (of[i] != hf[i]) throw Error(IFEM_E_COMM, "RCCL self-test: wrong data after the vector all-reduce");
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  ncclCommDestroy(comm2);
  ncclCommDestroy(comm);
  (void)hipStreamDestroy(hs);
  (void)hipStreamDestroy(s);
  for (int i = 0; i < 64; ++i) if (out2[i] != h[i]) throw Error(IFEM_E_COMM, "RCCL self-test: wrong data on the halo stream");
  for (int i = 0; i < 64; ++i) if (out[i] != h[i]) throw Error(IFEM_E_COMM, "RCCL self-test: wrong data");
  return IFEM_OK;
}

// what the context's communicator looks like and how much it was used (the whole multigrid chain below ctx included)
void comm_stats(ifem_ctx *ctx, ifem_comm_stats *out, bool reset, bool single_level) {
  std::memset(out, 0, sizeof(*out));
  const Halo &h0 = ctx->halo;
  out->nranks = h0.nranks;
  out->rank = h0.rank;
  out->n_neighbors = (int32_t)h0.nbr.size();
  out->transport = h0.nranks == 1 ? 0 : (h0.local ? 2 : 1);
  if (h0.comm) {
    int cn = 0;
    IFEM_NCCL_CHECK(ncclCommCount((ncclComm_t)h0.comm, &cn));
    out->rccl_nranks = cn;
    out->halo_stream = h0.comm2 && h0.hstream ? 1 : 0;
  }
  for (ifem_ctx *c = ctx; c; c = c->mg_coarse) {
    Halo &h = c->halo;
    out->halo_exchanges += h.n_exchanges;
    out->allreduce_dev += h.n_allreduce_dev;
    out->allreduce_host += h.n_allreduce_host;
    out->allreduce_vec += h.n_allreduce_vec;
    ++out->levels;
    if (reset) h.n_exchanges = h.n_allreduce_dev = h.n_allreduce_host = h.n_allreduce_vec = 0;
    if (single_level) break; // (ifem_comm_stats_level: this context alone -- the chain itself is never edited)
  }
}

void *local_world_create(int nranks) { return new LocalWorld(nranks); }
void local_world_destroy(void *w) { delete static_cast<LocalWorld *>(w); }

} // namespace ifem
