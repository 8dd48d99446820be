// api.hip -- the extern "C" surface declared in include/ifem_hip.h.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/ifem_hip_testing.h"
#include "ctx.hpp"
#include "kernels.hpp"

using namespace ifem;

static thread_local std::string g_err;

#define IFEM_API_BEGIN try {
#define IFEM_API_END                                           \
  }                                                            \
  catch (const ifem::Error &e) { g_err = e.what(); return e.code; } \
  catch (const std::exception &e) { g_err = e.what(); return IFEM_E_HIP; } \
  return IFEM_OK;

namespace ifem {
__global__ void k_flags_diff(int64_t n, const uint8_t *a, const uint8_t *b, unsigned long long *out) {
  unsigned long long d = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d += a[i] != b[i];
  if (d) atomicAdd(out, d);
}
__global__ void k_constraint_scatter(int32_t n, const int32_t *dof, const double *val, uint8_t *flag, double *cval) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[dof[i]] = 1;
  cval[dof[i]] = val ? val[i] : 0.0;
}

// out[k] = 1 when the flag arrays of pair k differ somewhere (a never-set array differs from everything), compared on the
// device: the flags of a 128^3 mesh are 53 MB, the answer is a word
void flags_differ(ifem_ctx *ctx, int npairs, const DBuf<uint8_t> *const *a, const DBuf<uint8_t> *const *b, double *out) {
  const size_t N = (size_t)ctx->n_local;
  unsigned long long *cnt = reinterpret_cast<unsigned long long *>(ctx->scal.p + kScalStageOff); // staging slots, idle here
  IFEM_HIP_CHECK(hipMemsetAsync(cnt, 0, npairs * sizeof(unsigned long long), ctx->stream));
  for (int k = 0; k < npairs; ++k)
    if (a[k]->n == N && b[k]->n == N && N)
      hipLaunchKernelGGL(k_flags_diff, dim3(1024), dim3(256), 0, ctx->stream, (int64_t)N, a[k]->p, b[k]->p, cnt + k);
  unsigned long long h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  IFEM_HIP_CHECK(hipMemcpyAsync(h, cnt, npairs * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  for (int k = 0; k < npairs; ++k) out[k] = (a[k]->n != N || b[k]->n != N || h[k] != 0) ? 1.0 : 0.0;
}

// identity of the constrained-dof set (ctx.hpp) after the flags of set `which` changed: differs_self = they are not what
// they were, differs_other = they are not the other set's either.  Decided over ALL ranks, since a stale cache triggers
// collective work.
void constraint_set_identity(ifem_ctx *ctx, int which, bool differs_self, bool differs_other) {
  const int other = 1 - which;
  double differs[2] = {differs_self ? 1.0 : 0.0, differs_other ? 1.0 : 0.0};
  allreduce_max(ctx, differs, 2);
  if (differs[0] != 0.0) ctx->flag_id[which] = differs[1] == 0.0 ? ctx->flag_id[other] : ++ctx->flag_counter;
}
} // namespace ifem

namespace {
template <typename T>
std::vector<T> download_range(const ifem::DBuf<T> &b, int64_t off, int64_t count, hipStream_t s) {
  if (off < 0 || count < 0 || size_t(off + count) > b.n) throw ifem::Error(IFEM_E_BADPARAM, "ifem_export_rows: slice outside its array");
  std::vector<T> h((size_t)count);
  if (count) {
    IFEM_HIP_CHECK(hipMemcpyAsync(h.data(), b.p + off, size_t(count) * sizeof(T), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
  }
  return h;
}
} // namespace

extern "C" {

const char *ifem_last_error(void) { return g_err.c_str(); }

int64_t ifem_abi_sizeof(int which) {
  switch (which) {
  case 0: return sizeof(ifem_mesh_desc);
  case 1: return sizeof(ifem_partition);
  case 2: return sizeof(ifem_ins_params);
  case 3: return sizeof(ifem_solver_opts);
  case 4: return sizeof(ifem_solve_stats);
  case 5: return sizeof(ifem_scns_params);
  case 6: return sizeof(ifem_timing);
  case 7: return sizeof(ifem_tuning);
  case 8: return sizeof(ifem_mg_transfer);
  case 9: return sizeof(ifem_fsi_solid);
  case 10: return sizeof(ifem_fsi_stats);
  case 11: return sizeof(ifem_comm_stats);
  case 12: return sizeof(ifem_kprof_entry);
  default: return -1;
  }
}

int ifem_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void ifem_default_solver_opts(ifem_solver_opts *o) {
  o->fgmres_restart = 30; o->fgmres_maxit = 0; o->fgmres_rel = 1e-4; o->fgmres_abs = 1e-12;
  o->mp_rel = 1e-6; o->mp_abs = 1e-10; o->sm_rel = 1e-3; o->sm_abs = 1e-10;
  o->ainv_kind = IFEM_AINV_GMRES_BJACOBI; o->inner_restart = 30; o->inner_maxit = 400; o->inner_rel = 1e-2;
  o->explicit_schur = 1;
  o->outer_matrix_free = 0;
  o->device_cg = 1;
  o->verbose = 0;
  o->sm_mg = 1; o->mg_smooth = 2; o->mg_cheb_ratio = 4.0;
  o->mg_smooth_u = 2; o->mg_smooth_u_post = 0; o->mg_cheb_ratio_u = 4.0;
  o->inner_rel_first = 0.0;
  o->inner_first_pshare = 0.0;
}

void ifem_default_tuning(ifem_tuning *t) {
  t->geo_cache = 1; t->xcd_swizzle = 1; t->asm_skip = 0; t->spmv_lanes = 32; t->sm_lanes = 32; t->mf_f32 = 1;
  t->tpp_operator = 0; t->spmv_pipe = 1; t->halo_overlap = 1; t->asm3_variant = 0; t->cg_single_reduction = 1; t->asm3_cpb = 2; t->tpp_milu_permille = 950; t->tpp_ilu_order = 2; t->basis_pad = 32 * 33; t->tpp_tri_sweeps = 0; t->uu_row_order = 1; t->eig_steps = 0; t->vcycle_graph_cells = 262144;
  t->scns_pc = 2; t->pvv_sweeps = 4; t->b2pp_sweeps = 6; t->scns_inner_reorth = 0; t->scns_inner_left = 1; t->scns_graph = 0; t->stored_uu = 1;
}

int ifem_set_tuning(ifem_ctx *ctx, const ifem_tuning *t) {
  IFEM_API_BEGIN
  if (!t) throw Error(IFEM_E_BADPARAM, "null tuning");
  const int g[2] = {t->spmv_lanes, t->sm_lanes};
  for (int v : g)
    if (v != 8 && v != 16 && v != 32 && v != 64) throw Error(IFEM_E_BADPARAM, "lanes per row must be 8, 16, 32 or 64");
  if (t->basis_pad < 0) throw Error(IFEM_E_BADPARAM, "negative size in ifem_tuning");
  if (t->asm3_cpb > 0 && t->asm3_cpb != 2 && t->asm3_cpb != 4 && t->asm3_cpb != 8) throw Error(IFEM_E_BADPARAM, "asm3_cpb must be 2, 4 or 8 (cells per workgroup of the 3D Q2/Q1 cell kernel)");
  if (t->tpp_ilu_order < -1 || t->tpp_ilu_order > 2) throw Error(IFEM_E_BADPARAM, "tpp_ilu_order must be -1, 0, 1 or 2");
  if (t->scns_pc != 0 && t->scns_pc != 1 && t->scns_pc != 2) throw Error(IFEM_E_BADPARAM, "scns_pc must be 1 (explicit T_pp) or 2 (the reference's structure); 0 = default");
  ctx->tune = *t;
  ++ctx->graph_epoch;
  if (ctx->tune.asm3_cpb <= 0) ctx->tune.asm3_cpb = 2; // a zero-initialised struct: the defaults of the fields added after round 4
  if (ctx->tune.scns_pc == 0) { ctx->tune.scns_pc = 2; if (!ctx->tune.pvv_sweeps) ctx->tune.pvv_sweeps = 4; if (!ctx->tune.b2pp_sweeps) ctx->tune.b2pp_sweeps = 6; ctx->tune.scns_inner_left = 1; ctx->tune.stored_uu = 1; }
  IFEM_API_END
}

int ifem_ctx_create(const ifem_mesh_desc *m, const ifem_partition *part, int device, ifem_ctx **out) {
  ifem_ctx *ctx = nullptr;
  IFEM_API_BEGIN
  if (!m || !out) throw Error(IFEM_E_BADPARAM, "null argument");
  if ((m->dim != 2 && m->dim != 3) || (m->kv != 1 && m->kv != 2)) throw Error(IFEM_E_BADPARAM, "dim must be 2|3, kv 1|2");
  if (ifem_device_count() <= device) throw Error(IFEM_E_NODEVICE, "no HIP device: libifem_hip has no CPU fallback");
  IFEM_HIP_CHECK(hipSetDevice(device));
  ctx = new ifem_ctx();
  ctx->device = device;
  ifem_default_tuning(&ctx->tune);
  IFEM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
  IFEM_HIP_CHECK(hipEventCreate(&ctx->ev0));
  IFEM_HIP_CHECK(hipEventCreate(&ctx->ev1));
  hipStream_t s = ctx->stream;
  ctx->dim = m->dim; ctx->kv = m->kv;
  build_fe_tables(ctx->fe, m->dim, m->kv);
  ctx->nu = ctx->fe.nu; ctx->np = ctx->fe.np; ctx->nq = ctx->fe.nq;
  ctx->d_fe.upload(&ctx->fe, 1, s);
  ctx->n_cells = m->n_cells;
  ctx->nUo = m->n_unodes_owned; ctx->nUl = m->n_unodes_local;
  ctx->nPo = m->n_pnodes_owned; ctx->nPl = m->n_pnodes_local;
  ctx->n_local = ctx->dim * ctx->nUl + ctx->nPl;
  const int dim = ctx->dim, nu = ctx->nu, np = ctx->np;
  ctx->vcoords.upload(m->vcoords, (size_t)m->n_cells * np * dim, s);
  ctx->cell_unodes.upload(m->cell_unodes, (size_t)m->n_cells * nu, s);
  ctx->cell_pnodes.upload(m->cell_pnodes, (size_t)m->n_cells * np, s);
  if (m->cell_face_bid) ctx->cell_face_bid.upload(m->cell_face_bid, (size_t)m->n_cells * 2 * dim, s);
  else {
    std::vector<int32_t> none((size_t)m->n_cells * 2 * dim, -1);
    ctx->cell_face_bid.upload(none.data(), none.size(), s);
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
  }
  IFEM_HIP_CHECK(hipHostMalloc((void **)&ctx->h_scal, kScalSlots * sizeof(double)));
  ctx->scal.alloc(kScalSlots);
  comm_init(ctx, part);
  {
    double g[2] = {double(ctx->dim * ctx->nUo), double(ctx->nPo)};
    allreduce_sum(ctx, g, 2);
    ctx->n_global_u = (int64_t)g[0]; ctx->n_global_p = (int64_t)g[1];
  }
  // block sparsity + scatter maps (make_sparsity_pattern / matrix.reinit, mpi_fluid_solver.cpp:311-322)
  build_pattern(ctx, ctx->Auu, dim * dim, ctx->nUo, nu, ctx->cell_unodes.p, nu, ctx->cell_unodes.p, ctx->posUU);
  build_pattern(ctx, ctx->Bt, dim, ctx->nUo, nu, ctx->cell_unodes.p, np, ctx->cell_pnodes.p, ctx->posUP);
  build_pattern(ctx, ctx->B, dim, ctx->nPo, np, ctx->cell_pnodes.p, nu, ctx->cell_unodes.p, ctx->posPU);
  build_pattern(ctx, ctx->Mp, 1, ctx->nPo, np, ctx->cell_pnodes.p, np, ctx->cell_pnodes.p, ctx->posPP);
  ctx->diagMu.alloc((size_t)dim * ctx->nUo);
  ctx->dinvMu.alloc((size_t)dim * ctx->nUo);
  ctx->bjac.alloc((size_t)dim * dim * ctx->nUo);
  for (int v = 0; v < IFEM_N_VECS; ++v) {
    ctx->vec[v].alloc((size_t)ctx->n_local);
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->vec[v].p, 0, ctx->n_local * sizeof(double), s));
  }
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  *out = ctx;
  }
  catch (const ifem::Error &e) { g_err = e.what(); if (ctx) ifem_ctx_destroy(ctx); return e.code; }
  catch (const std::exception &e) { g_err = e.what(); if (ctx) ifem_ctx_destroy(ctx); return IFEM_E_HIP; }
  return IFEM_OK;
}

void ifem_ctx_destroy(ifem_ctx *ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  // unhook from a multigrid chain.  The level above forgets this one; the levels below ran on this context's stream(s) (or
  // on those of a level further up) and stay usable by themselves: the chain below gets streams of its own.
  if (ctx->mg_fine && ctx->mg_fine->mg_coarse == ctx) {
    ctx->mg_fine->mg_coarse = nullptr;
    ctx->mg_fine->mg_replica = false;
    ctx->mg_fine->sm_mg_version = -1;
    ctx->mg_fine->uu_mg_version = -1;
  }
  if (ifem_ctx *below = ctx->mg_coarse) {
    below->mg_fine = nullptr;
    hipStream_t ns = nullptr, nh = nullptr;
    if (!below->owns_stream && hipStreamCreateWithFlags(&ns, hipStreamNonBlocking) == hipSuccess) {
      for (ifem_ctx *c = below; c; c = c->mg_coarse) { c->stream = ns; c->owns_stream = c == below; }
    }
    if (below->halo.hstream && !below->halo.owns_hstream) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&nh, hipStreamNonBlocking, hi) == hipSuccess)
        for (ifem_ctx *c = below; c; c = c->mg_coarse) { c->halo.hstream = nh; c->halo.owns_hstream = c == below; }
    }
  }
  comm_destroy(ctx);
  if (ctx->h_scal) (void)hipHostFree(ctx->h_scal);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (hipEvent_t e : ctx->pc_ev) (void)hipEventDestroy(e);
  ctx->vc_graph.destroy();
  ctx->sm_graph.destroy();
  ctx->pa_graph.destroy();
  hipStream_t s = ctx->owns_stream ? ctx->stream : nullptr;
  delete ctx;
  if (s) (void)hipStreamDestroy(s);
}

int64_t ifem_n_local_dofs(const ifem_ctx *ctx) { return ctx->n_local; }
int64_t ifem_nnz(const ifem_ctx *ctx, int block) {
  switch (block) {
  case 0: return ctx->Auu.nnzb;
  case 1: return ctx->B.nnzb;
  case 2: return ctx->Mp.nnzb;
  default: return 0;
  }
}

int ifem_set_constraints(ifem_ctx *ctx, int which, int32_t n, const int32_t *dof, const double *inhom) {
  IFEM_API_BEGIN
  if (which < 0 || which > 1) throw Error(IFEM_E_BADPARAM, "which must be 0 or 1");
  if (n < 0 || (n > 0 && !dof)) throw Error(IFEM_E_BADPARAM, "bad constraint list");
  // Only the n lines cross the bus; flags and inhomogeneities over the local dofs are built and compared on the device
  // (FSI re-makes the constraints every time step, mpi_fsi.cpp:1191: at 128^3 the dense host arrays were 480 MB per call).
  // A dof listed twice keeps its LAST line, as the sequential host loop did.
  const size_t N = (size_t)ctx->n_local;
  std::vector<int32_t> kd;
  std::vector<double> kv;
  kd.reserve((size_t)n);
  kv.reserve((size_t)n);
  {
    std::vector<uint8_t> &seen = ctx->seen_scratch; // all zero between calls
    if (seen.size() != N) seen.assign(N, 0);
    int32_t bad = 0;
    for (int32_t i = n - 1; i >= 0; --i) {
      const int32_t d = dof[i];
      if (d < 0 || (size_t)d >= N) { bad = 1; break; }
      if (d >= ctx->dim * ctx->nUl) { bad = 2; break; }
      if (seen[d]) continue;
      seen[d] = 1;
      kd.push_back(d);
      kv.push_back(inhom ? inhom[i] : 0.0);
    }
    for (int32_t d : kd) seen[d] = 0;
    // the call is collective on partitioned contexts (constraint_set_identity all-reduces): every rank must fail together,
    // or the ranks with good lists would wait in that all-reduce for the one that threw
    double worst = bad;
    allreduce_max(ctx, &worst, 1);
    if (worst == 1.0) throw Error(IFEM_E_BADPARAM, bad == 1 ? "constraint dof out of range" : "constraint dof out of range on another rank");
    if (worst == 2.0) throw Error(IFEM_E_BADPARAM, "pressure Dirichlet constraints are not supported");
  }
  hipStream_t s = ctx->stream;
  DBuf<int32_t> d_dof;
  DBuf<double> d_val;
  DBuf<uint8_t> nf;
  DBuf<double> nv;
  nf.alloc(N);
  nv.alloc(N);
  if (N) {
    IFEM_HIP_CHECK(hipMemsetAsync(nf.p, 0, N, s));
    IFEM_HIP_CHECK(hipMemsetAsync(nv.p, 0, N * sizeof(double), s));
  }
  if (!kd.empty()) {
    d_dof.upload(kd.data(), kd.size(), s);
    d_val.upload(kv.data(), kv.size(), s);
    hipLaunchKernelGGL(k_constraint_scatter, dim3((unsigned)((kd.size() + 255) / 256)), dim3(256), 0, s, (int32_t)kd.size(), d_dof.p, d_val.p, nf.p, nv.p);
  }
  const DBuf<uint8_t> *pa[2] = {&nf, &nf}, *pb[2] = {&ctx->is_c[which], &ctx->is_c[1 - which]};
  double differs[2];
  flags_differ(ctx, 2, pa, pb, differs); // synchronises: the host lists may go away
  ctx->is_c[which].swap(nf);
  ctx->cval[which].swap(nv);
  ctx->has_c[which] = !kd.empty();
  {
    double any = 0; // (collective on partitioned contexts: a rank without inhomogeneous lines must agree with one that has them)
    for (double v : kv) if (v != 0.0) { any = 1; break; }
    allreduce_max(ctx, &any, 1);
    ctx->inhom_any[which] = any != 0.0;
  }
  constraint_set_identity(ctx, which, differs[0] != 0.0, differs[1] != 0.0);
  IFEM_API_END
}

int ifem_fsi_set_solid(ifem_ctx *ctx, const ifem_fsi_solid *solid) {
  IFEM_API_BEGIN
  ifem::fsi_set_solid(ctx, solid);
  IFEM_API_END
}
int ifem_fsi_update_indicator(ifem_ctx *ctx, int32_t *host_out, int64_t *n_artificial) {
  IFEM_API_BEGIN
  ifem::fsi_update_indicator(ctx, host_out, n_artificial);
  IFEM_API_END
}
int ifem_fsi_find_fluid_bc(ifem_ctx *ctx, double dt, int use_dirichlet_bc, const int32_t *cell_order, ifem_fsi_stats *stats) {
  IFEM_API_BEGIN
  if (!(dt > 0)) throw Error(IFEM_E_BADPARAM, "ifem_fsi_find_fluid_bc: dt must be positive");
  ifem::fsi_find_fluid_bc(ctx, dt, use_dirichlet_bc, cell_order, stats);
  IFEM_API_END
}
int ifem_fsi_fluid_at_points(ifem_ctx *ctx, int32_t n, const double *points, double *values, double *stress, int32_t *cell) {
  IFEM_API_BEGIN
  ifem::fsi_fluid_at_points(ctx, n, points, values, stress, cell);
  IFEM_API_END
}
int ifem_fsi_get_stress(ifem_ctx *ctx, double *host_out) {
  IFEM_API_BEGIN
  const size_t n = (size_t)(ctx->dim * (ctx->dim + 1) / 2) * ctx->nUl;
  if (!host_out) throw Error(IFEM_E_BADPARAM, "null output");
  if (ctx->fsi_stress.n != n) throw Error(IFEM_E_BADPARAM, "the context holds no nodal fsi_stress");
  IFEM_HIP_CHECK(hipMemcpyAsync(host_out, ctx->fsi_stress.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}
int ifem_get_constraints(ifem_ctx *ctx, int which, uint8_t *flags, double *inhom) {
  IFEM_API_BEGIN
  if (which < 0 || which > 1) throw Error(IFEM_E_BADPARAM, "which must be 0 or 1");
  const size_t n = (size_t)ctx->n_local;
  if (ctx->is_c[which].n != n) { // never set: no line
    if (flags) std::fill(flags, flags + n, uint8_t(0));
    if (inhom) std::fill(inhom, inhom + n, 0.0);
  } else {
    if (flags) IFEM_HIP_CHECK(hipMemcpyAsync(flags, ctx->is_c[which].p, n, hipMemcpyDeviceToHost, ctx->stream));
    if (inhom) IFEM_HIP_CHECK(hipMemcpyAsync(inhom, ctx->cval[which].p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  IFEM_API_END
}

int ifem_set_hanging_constraints(ifem_ctx *ctx, int32_t n, const int32_t *dof, const int32_t *ptr, const int32_t *master,
                                 const double *weight) {
  IFEM_API_BEGIN
  ifem::hanging_set(ctx, n, dof, ptr, master, weight);
  IFEM_API_END
}

int ifem_mg_attach(ifem_ctx *fine, ifem_ctx *coarse, const ifem_mg_transfer *t) {
  IFEM_API_BEGIN
  if (!fine || !coarse || !t || fine == coarse) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: two contexts and a transfer table");
  // a single-rank coarse context below a partitioned fine one: the REPLICATED coarse level (see ifem_hip.h)
  const bool replica = fine->halo.nranks > 1 && coarse->halo.nranks == 1;
  if (fine->dim != coarse->dim || fine->kv != coarse->kv || fine->device != coarse->device ||
      (!replica && (fine->halo.nranks != coarse->halo.nranks || fine->halo.rank != coarse->halo.rank)))
    throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: the levels must share dimension, velocity degree, device, rank and rank count "
                                 "(or the coarse context is a single-rank replica of the whole coarse mesh)");
  if (coarse->mg_fine && coarse->mg_fine != fine)
    throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: the coarse context already hangs below another level");
  for (const ifem_ctx *c = coarse; c; c = c->mg_coarse)
    if (c == fine) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: the fine context is a level below the coarse one");
  // row pointers non-decreasing and closed: a malformed table would send the transfer kernels out of bounds
  auto monotone = [](const int64_t *ptr, int64_t n_rows, int64_t nnz_, const char *what) {
    if (ptr[0] != 0 || ptr[n_rows] != nnz_) throw Error(IFEM_E_BADPARAM, std::string("ifem_mg_attach: ") + what + " row pointers do not span the table");
    for (int64_t r = 0; r < n_rows; ++r)
      if (ptr[r + 1] < ptr[r]) throw Error(IFEM_E_BADPARAM, std::string("ifem_mg_attach: ") + what + " row pointers decrease");
  };
  if (t->n_fine_p_owned != fine->nPo || t->n_coarse_p_local != coarse->nPl)
    throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: P_p needs one row per owned fine pressure node, R_p one per local coarse pressure node");
  if (!t->pp_ptr || !t->pp_col || !t->pp_w || !t->rp_ptr || !t->rp_col || !t->rp_w) throw Error(IFEM_E_BADPARAM, "null transfer table");
  const int64_t nnz = t->pp_ptr[fine->nPo];
  if (nnz < 0 || t->rp_ptr[coarse->nPl] != nnz) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: R_p is not the transpose of P_p");
  monotone(t->pp_ptr, fine->nPo, nnz, "P_p");
  monotone(t->rp_ptr, coarse->nPl, nnz, "R_p");
  for (int64_t k = 0; k < nnz; ++k) {
    if (t->pp_col[k] < 0 || t->pp_col[k] >= coarse->nPl) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: P_p column out of range");
    if (t->rp_col[k] < 0 || t->rp_col[k] >= fine->nPo) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: R_p column out of range");
  }
  hipStream_t s = fine->stream;
  fine->mg_Pp.n_rows = fine->nPo;
  fine->mg_Pp.ptr.upload(t->pp_ptr, (size_t)fine->nPo + 1, s);
  fine->mg_Pp.col.upload(t->pp_col, (size_t)nnz, s);
  fine->mg_Pp.w.upload(t->pp_w, (size_t)nnz, s);
  fine->mg_Rp.n_rows = coarse->nPl;
  fine->mg_Rp.ptr.upload(t->rp_ptr, (size_t)coarse->nPl + 1, s);
  fine->mg_Rp.col.upload(t->rp_col, (size_t)nnz, s);
  fine->mg_Rp.w.upload(t->rp_w, (size_t)nnz, s);
  fine->mg_Pu.n_rows = fine->mg_Ru.n_rows = 0;
  // the packed transfer entries (columns, weights and drop bits baked in) and the captured cycles belong to the tables that are replaced
  fine->mg_mask_key[0] = fine->mg_mask_key[1] = -1;
  ++fine->graph_epoch; ++coarse->graph_epoch;
  fine->vc_graph.destroy(); fine->sm_graph.destroy();
  fine->vc_graph.armed = fine->sm_graph.armed = false; fine->vc_graph.key.clear(); fine->sm_graph.key.clear();
  if (t->pu_ptr) { // velocity-node transfers: optional
    if (!t->pu_col || !t->pu_w || !t->ru_ptr || !t->ru_col || !t->ru_w || !t->inj_u) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: incomplete velocity transfer tables");
    if (t->n_fine_u_owned != fine->nUo || t->n_coarse_u_local != coarse->nUl)
      throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: P_u needs one row per owned fine velocity node, R_u one per local coarse velocity node");
    const int64_t nu = t->pu_ptr[fine->nUo];
    if (nu < 0 || t->ru_ptr[coarse->nUl] != nu) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: R_u is not the transpose of P_u");
    monotone(t->pu_ptr, fine->nUo, nu, "P_u");
    monotone(t->ru_ptr, coarse->nUl, nu, "R_u");
    for (int64_t k = 0; k < nu; ++k) {
      if (t->pu_col[k] < 0 || t->pu_col[k] >= coarse->nUl) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: P_u column out of range");
      if (t->ru_col[k] < 0 || t->ru_col[k] >= fine->nUo) throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: R_u column out of range");
    }
    for (int64_t i = 0; i < coarse->nUo; ++i)
      if ((t->inj_u[i] < 0 && !(replica && t->inj_u[i] == -1)) || t->inj_u[i] >= fine->nUo)
        throw Error(IFEM_E_BADPARAM, "ifem_mg_attach: inj_u must name owned fine nodes (-1 below a replicated level: another rank owns the fine node)");
    fine->mg_Pu.n_rows = fine->nUo;
    fine->mg_Pu.ptr.upload(t->pu_ptr, (size_t)fine->nUo + 1, s);
    fine->mg_Pu.col.upload(t->pu_col, (size_t)nu, s);
    fine->mg_Pu.w.upload(t->pu_w, (size_t)nu, s);
    fine->mg_Ru.n_rows = coarse->nUl;
    fine->mg_Ru.ptr.upload(t->ru_ptr, (size_t)coarse->nUl + 1, s);
    fine->mg_Ru.col.upload(t->ru_col, (size_t)nu, s);
    fine->mg_Ru.w.upload(t->ru_w, (size_t)nu, s);
    fine->mg_inj_u.upload(t->inj_u, (size_t)coarse->nUo, s);
  }
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  if (fine->mg_coarse && fine->mg_coarse != coarse) fine->mg_coarse->mg_fine = nullptr; // re-attach: the old level is on its own
  fine->mg_coarse = coarse;
  fine->mg_replica = replica;
  coarse->mg_fine = fine;
  fine->sm_mg_version = -1;
  fine->uu_mg_version = -1;
  // one stream for the whole chain: the V-cycle walks up and down the levels and every launch must stay in order
  for (ifem_ctx *c = coarse; c; c = c->mg_coarse) {
    if (c->stream == s) continue;
    IFEM_HIP_CHECK(hipStreamSynchronize(c->stream));
    if (c->owns_stream) (void)hipStreamDestroy(c->stream);
    c->stream = s;
    c->owns_stream = false;
    // likewise one halo stream (the levels share the communicators, comm.hip)
    if (fine->halo.hstream && c->halo.hstream && c->halo.hstream != fine->halo.hstream) {
      IFEM_HIP_CHECK(hipStreamSynchronize(c->halo.hstream));
      if (c->halo.owns_hstream) (void)hipStreamDestroy(c->halo.hstream);
      c->halo.hstream = fine->halo.hstream;
      c->halo.owns_hstream = false;
    }
  }
  IFEM_API_END
}

int ifem_mg_depth(const ifem_ctx *ctx) {
  int d = 0;
  for (const ifem_ctx *c = ctx ? ctx->mg_coarse : nullptr; c; c = c->mg_coarse) ++d;
  return d;
}

int ifem_set_cell_fields(ifem_ctx *ctx, const int32_t *indicator) {
  IFEM_API_BEGIN
  if (indicator) {
    ctx->indicator.upload(indicator, (size_t)ctx->n_cells, ctx->stream);
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  } else
    ctx->indicator.release();
  IFEM_API_END
}

static inline bool vec_ok(int v) { return v >= 0 && v < IFEM_N_VECS; }
// PRESENT / EVAL / FSI_ACC / INCREMENT are ghosted (extended) vectors, the others hold owned entries only
static inline bool is_ext(int v) { return v == IFEM_VEC_PRESENT || v == IFEM_VEC_EVAL || v == IFEM_VEC_FSI_ACC || v == IFEM_VEC_INCREMENT; }
static inline int64_t vec_len(const ifem_ctx *c, int v) { return is_ext(v) ? c->n_local : c->dim * c->nUo + c->nPo; }
static inline int64_t p_off(const ifem_ctx *c, int v) { return c->dim * (is_ext(v) ? c->nUl : c->nUo); }

int ifem_vec_set(ifem_ctx *ctx, int vec, const double *host) {
  IFEM_API_BEGIN
  if (!vec_ok(vec)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->vec[vec].p, host, vec_len(ctx, vec) * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}
int ifem_vec_get(ifem_ctx *ctx, int vec, double *host) {
  IFEM_API_BEGIN
  if (!vec_ok(vec)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  IFEM_HIP_CHECK(hipMemcpyAsync(host, ctx->vec[vec].p, vec_len(ctx, vec) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}
int ifem_vec_zero(ifem_ctx *ctx, int vec) {
  IFEM_API_BEGIN
  if (!vec_ok(vec)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  v_zero(ctx, ctx->n_local, ctx->vec[vec].p);
  IFEM_API_END
}
// owned entries of src -> dst (layouts may differ); ghosts of an extended dst are refreshed
static void copy_owned(ifem_ctx *ctx, int dst, int src, double a_src, double b_dst) {
  const int64_t nuo = ctx->dim * ctx->nUo;
  v_axpby(ctx, nuo, a_src, ctx->vec[src].p, b_dst, ctx->vec[dst].p);
  v_axpby(ctx, ctx->nPo, a_src, ctx->vec[src].p + p_off(ctx, src), b_dst, ctx->vec[dst].p + p_off(ctx, dst));
  if (is_ext(dst)) {
    halo_exchange(ctx, ctx->vec[dst].p);
    halo_exchange_p(ctx, ctx->vec[dst].p + p_off(ctx, dst));
  }
}
int ifem_vec_copy(ifem_ctx *ctx, int dst, int src) {
  IFEM_API_BEGIN
  if (!vec_ok(dst) || !vec_ok(src)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  copy_owned(ctx, dst, src, 1.0, 0.0);
  IFEM_API_END
}
int ifem_vec_axpy(ifem_ctx *ctx, double a, int x, int y) {
  IFEM_API_BEGIN
  if (!vec_ok(x) || !vec_ok(y)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  copy_owned(ctx, y, x, a, 1.0);
  IFEM_API_END
}
static double norm2_owned(ifem_ctx *ctx, int vec) {
  const int64_t nuo = ctx->dim * ctx->nUo;
  double d[2];
  d[0] = v_dot(ctx, nuo, ctx->vec[vec].p, ctx->vec[vec].p);
  d[1] = v_dot(ctx, ctx->nPo, ctx->vec[vec].p + p_off(ctx, vec), ctx->vec[vec].p + p_off(ctx, vec));
  double s = d[0] + d[1];
  allreduce_sum(ctx, &s, 1);
  return std::sqrt(s);
}
int ifem_vec_norm2(ifem_ctx *ctx, int vec, double *out) {
  IFEM_API_BEGIN
  if (!vec_ok(vec)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  *out = norm2_owned(ctx, vec);
  IFEM_API_END
}
int ifem_vec_minmax(ifem_ctx *ctx, int vec, int block, double *vmin, double *vmax) {
  IFEM_API_BEGIN
  if (!vec_ok(vec)) throw Error(IFEM_E_BADPARAM, "bad vector id");
  double mn, mx;
  if (block == 0) v_minmax(ctx, ctx->dim * ctx->nUo, ctx->vec[vec].p, &mn, &mx);
  else v_minmax(ctx, ctx->nPo, ctx->vec[vec].p + p_off(ctx, vec), &mn, &mx);
  if (ctx->halo.nranks > 1) { // max = -min(-x): reuse the sum all-reduce on {mn, -mx} is wrong; do two max-reductions
    double t[2] = {-mn, mx};
    allreduce_max(ctx, t, 2);
    mn = -t[0]; mx = t[1];
  }
  if (vmin) *vmin = mn;
  if (vmax) *vmax = mx;
  IFEM_API_END
}
int ifem_halo_exchange(ifem_ctx *ctx, int vec) {
  IFEM_API_BEGIN
  if (!vec_ok(vec) || !is_ext(vec)) throw Error(IFEM_E_BADPARAM, "not a ghosted vector");
  halo_exchange(ctx, ctx->vec[vec].p);
  halo_exchange_p(ctx, ctx->vec[vec].p + p_off(ctx, vec));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_ins_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) {
  IFEM_API_BEGIN
  if (!p || p->dt <= 0) throw Error(IFEM_E_BADPARAM, "bad ifem_ins_params");
  auto t0 = std::chrono::steady_clock::now();
  launch_ins_assemble(ctx, p, use_nonzero);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->timing.assemble_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  IFEM_API_END
}

int ifem_imex_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero, int assemble_system) {
  IFEM_API_BEGIN
  if (!p || p->dt <= 0) throw Error(IFEM_E_BADPARAM, "bad ifem_ins_params");
  auto t0 = std::chrono::steady_clock::now();
  launch_ins_assemble_ex(ctx, p, use_nonzero, 1, assemble_system);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->timing.assemble_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  IFEM_API_END
}

static int imex_solve_impl(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats) {
  ifem_solver_opts oo;
  if (o) oo = *o; else { ifem_default_solver_opts(&oo); oo.inner_rel = 1e-4; }
  const double bn = norm2_owned(ctx, IFEM_VEC_RHS);
  oo.fgmres_rel = 0.0;
  oo.fgmres_abs = std::min(1e-9, 1e-8 * bn); // mpi_insimex.cpp:369-370
  return ins_solve(ctx, p, &oo, use_nonzero, stats);
}

int ifem_imex_solve(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats) {
  IFEM_API_BEGIN
  if (!p) throw Error(IFEM_E_BADPARAM, "null params");
  const int rc = imex_solve_impl(ctx, p, o, use_nonzero, stats);
  if (rc != 0) { g_err = "FGMRES did not converge (SolverControl::NoConvergence)"; return rc; }
  IFEM_API_END
}

int ifem_imex_step(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int apply_nonzero, int assemble_system,
                   ifem_solve_stats *stats) {
  IFEM_API_BEGIN
  if (!p || p->dt <= 0) throw Error(IFEM_E_BADPARAM, "bad ifem_ins_params");
  v_zero(ctx, ctx->n_local, ctx->vec[IFEM_VEC_UPDATE].p); // solution_time_increment = 0
  launch_ins_assemble_ex(ctx, p, apply_nonzero, 1, assemble_system);
  const int rc = imex_solve_impl(ctx, p, o, apply_nonzero, stats);
  if (rc != 0) throw Error(rc, "FGMRES did not converge (SolverControl::NoConvergence)");
  copy_owned(ctx, IFEM_VEC_PRESENT, IFEM_VEC_UPDATE, 1.0, 1.0); // present_solution += solution_time_increment
  launch_update_stress(ctx, p->viscosity);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_solve(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats) {
  IFEM_API_BEGIN
  ifem_solver_opts def;
  if (!o) { ifem_default_solver_opts(&def); o = &def; }
  const int rc = ins_solve(ctx, p, o, use_nonzero, stats);
  if (rc != 0) throw Error(rc, "FGMRES did not converge (SolverControl::NoConvergence)");
  IFEM_API_END
}

int ifem_rhs_norm(ifem_ctx *ctx, double *l2) { return ifem_vec_norm2(ctx, IFEM_VEC_RHS, l2); }

int ifem_ins_newton_step(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int apply_nonzero,
                         double tolerance, int max_iterations, double *log) {
  int outer = 0;
  try {
    ifem_solver_opts def;
    if (!o) { ifem_default_solver_opts(&def); o = &def; }
    double cur = 1.0, init = 1.0, rel = 1.0;
    copy_owned(ctx, IFEM_VEC_EVAL, IFEM_VEC_PRESENT, 1.0, 0.0); // evaluation_point = present_solution (:420)
    while (rel > tolerance && cur > 1e-11) {
      if (outer >= max_iterations) throw Error(IFEM_E_NEWTON_MAXIT, "Too many Newton iterations!");
      const int nz = apply_nonzero && outer == 0;
      v_zero(ctx, ctx->n_local, ctx->vec[IFEM_VEC_UPDATE].p); // newton_update = 0 (:427)
      launch_ins_assemble(ctx, p, nz);
      ifem_solve_stats st{};
      const int rc = ins_solve(ctx, p, o, nz, &st);
      if (rc != 0) throw Error(rc, "FGMRES did not converge (SolverControl::NoConvergence)");
      cur = norm2_owned(ctx, IFEM_VEC_RHS);                      // system_rhs.l2_norm() (:438)
      copy_owned(ctx, IFEM_VEC_EVAL, IFEM_VEC_UPDATE, 1.0, 1.0); // evaluation_point += newton_update (:444-448)
      if (outer == 0) init = cur;
      rel = cur / init;
      if (log) { log[outer * 4 + 0] = cur; log[outer * 4 + 1] = rel; log[outer * 4 + 2] = st.fgmres_iters; log[outer * 4 + 3] = st.fgmres_res; }
      ++outer;
    }
    // solution_increment = present - evaluation; present = evaluation (:465-473)
    copy_owned(ctx, IFEM_VEC_INCREMENT, IFEM_VEC_PRESENT, 1.0, 0.0);
    copy_owned(ctx, IFEM_VEC_INCREMENT, IFEM_VEC_EVAL, -1.0, 1.0);
    copy_owned(ctx, IFEM_VEC_PRESENT, IFEM_VEC_EVAL, 1.0, 0.0);
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  catch (const ifem::Error &e) { g_err = e.what(); return e.code; }
  catch (const std::exception &e) { g_err = e.what(); return IFEM_E_HIP; }
  return outer;
}

int ifem_set_scns_fields(ifem_ctx *ctx, const double *sigma_pml, const double *body_force, const double *fsi_stress) {
  IFEM_API_BEGIN
  hipStream_t s = ctx->stream;
  const size_t ncq = (size_t)ctx->n_cells * ctx->nq;
  if (sigma_pml) ctx->sigma_pml.upload(sigma_pml, ncq, s); else ctx->sigma_pml.release();
  if (body_force) ctx->body_force.upload(body_force, ncq * ctx->dim, s); else ctx->body_force.release();
  if (fsi_stress) ctx->fsi_stress.upload(fsi_stress, (size_t)(ctx->dim * (ctx->dim + 1) / 2) * ctx->nUl, s); else ctx->fsi_stress.release();
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_API_END
}

int ifem_set_eddy_viscosity(ifem_ctx *ctx, const double *nodal) {
  IFEM_API_BEGIN
  if (nodal) ctx->eddy_viscosity.upload(nodal, (size_t)ctx->nUl, ctx->stream); else ctx->eddy_viscosity.release();
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_update_stress(ifem_ctx *ctx, double viscosity, double *host_out) {
  IFEM_API_BEGIN
  launch_update_stress(ctx, viscosity);
  if (host_out) {
    IFEM_HIP_CHECK(hipMemcpyAsync(host_out, ctx->stress.p, ctx->stress.n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  IFEM_API_END
}

int ifem_scns_assemble(ifem_ctx *ctx, const ifem_scns_params *p, int use_nonzero) {
  IFEM_API_BEGIN
  if (!p || p->dt <= 0 || p->viscosity <= 0) throw Error(IFEM_E_BADPARAM, "bad ifem_scns_params");
  launch_scns_assemble(ctx, p, use_nonzero);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_scns_solve(ifem_ctx *ctx, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats) {
  IFEM_API_BEGIN
  ifem_solver_opts def;
  if (!o) { ifem_default_solver_opts(&def); o = &def; }
  const int rc = scns_solve(ctx, o, use_nonzero, stats);
  if (rc != 0) throw Error(rc, "FGMRES did not converge (SolverControl::NoConvergence)");
  IFEM_API_END
}

int ifem_scns_newton_step(ifem_ctx *ctx, const ifem_scns_params *p, const ifem_solver_opts *o, int apply_nonzero,
                          double tolerance, int max_iterations, double *log) {
  int outer = 0;
  try {
    ifem_solver_opts def;
    if (!o) { ifem_default_solver_opts(&def); o = &def; }
    double cur = 1.0, init = 1.0, rel = 1.0;
    copy_owned(ctx, IFEM_VEC_EVAL, IFEM_VEC_PRESENT, 1.0, 0.0);
    while (rel > tolerance && cur > 1e-14) { // mpi_supg_solver.cpp:354-355
      if (outer >= max_iterations) throw Error(IFEM_E_NEWTON_MAXIT, "Too many Newton iterations!");
      const int nz = apply_nonzero && outer == 0;
      v_zero(ctx, ctx->n_local, ctx->vec[IFEM_VEC_UPDATE].p);
      launch_scns_assemble(ctx, p, nz);
      ifem_solve_stats st{};
      const int rc = scns_solve(ctx, o, nz, &st);
      if (rc != 0) throw Error(rc, "FGMRES did not converge (SolverControl::NoConvergence)");
      cur = norm2_owned(ctx, IFEM_VEC_RHS);
      copy_owned(ctx, IFEM_VEC_EVAL, IFEM_VEC_UPDATE, 1.0, 1.0);
      if (outer == 0) init = cur;
      rel = cur / init;
      if (log) { log[outer * 4 + 0] = cur; log[outer * 4 + 1] = rel; log[outer * 4 + 2] = st.fgmres_iters; log[outer * 4 + 3] = st.inner_iters; }
      ++outer;
    }
    copy_owned(ctx, IFEM_VEC_INCREMENT, IFEM_VEC_PRESENT, 1.0, 0.0);
    copy_owned(ctx, IFEM_VEC_INCREMENT, IFEM_VEC_EVAL, -1.0, 1.0);
    copy_owned(ctx, IFEM_VEC_PRESENT, IFEM_VEC_EVAL, 1.0, 0.0);
    launch_update_stress(ctx, p->viscosity); // update_stress() (:396), feeds the next assemble
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  catch (const ifem::Error &e) { g_err = e.what(); return e.code; }
  catch (const std::exception &e) { g_err = e.what(); return IFEM_E_HIP; }
  return outer;
}

int ifem_system_vmult(ifem_ctx *ctx, int dst, int src) {
  IFEM_API_BEGIN
  if (!vec_ok(dst) || !vec_ok(src) || is_ext(dst) || is_ext(src)) throw Error(IFEM_E_BADPARAM, "use non-ghosted vectors");
  ins_system_vmult(ctx, ctx->vec[src].p, ctx->vec[dst].p);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_true_residual(ifem_ctx *ctx, double *residual_l2, double *rhs_l2) {
  IFEM_API_BEGIN
  if (!ctx->assembled || !residual_l2) throw Error(IFEM_E_BADPARAM, "ifem_true_residual after an assembly and a solve");
  const int64_t n = int64_t(ctx->dim) * ctx->nUo + ctx->nPo;
  DBuf<double> r;
  r.alloc((size_t)n + 8);
  ins_system_vmult(ctx, ctx->vec[IFEM_VEC_UPDATE].p, r.p);
  v_axpby(ctx, n, 1.0, ctx->vec[IFEM_VEC_RHS].p, -1.0, r.p); // r = b - A x
  apply_constraints(ctx, 0, r.p); // zero_constraints: the constrained rows are set to 0 and drop out of the norm
  *residual_l2 = std::sqrt(bv_dot(ctx, r.p, r.p));
  if (rhs_l2) *rhs_l2 = std::sqrt(bv_dot(ctx, ctx->vec[IFEM_VEC_RHS].p, ctx->vec[IFEM_VEC_RHS].p));
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_mass_vmult(ifem_ctx *ctx, int dst, int src) {
  IFEM_API_BEGIN
  if (!vec_ok(dst) || !vec_ok(src) || is_ext(dst) || is_ext(src) || dst == src) throw Error(IFEM_E_BADPARAM, "use two non-ghosted vectors");
  if (!ctx->assembled) throw Error(IFEM_E_BADPARAM, "ifem_mass_vmult called before an assembly");
  const int64_t nuo = int64_t(ctx->dim) * ctx->nUo;
  vec_mul(ctx, nuo, ctx->diagMu.p, ctx->vec[src].p, ctx->vec[dst].p);
  if ((int64_t)ctx->work.n < ctx->nPl) ctx->work.alloc(ctx->nPl);
  double *xe = ctx->work.p; // ghost-extended copy of the pressure part
  v_copy(ctx, ctx->nPo, ctx->vec[src].p + nuo, xe);
  if (ctx->halo.nranks > 1) halo_exchange_p(ctx, xe);
  spmv_mp(ctx, xe, ctx->vec[dst].p + nuo);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_uu_block_diag(ifem_ctx *ctx, int which, double *host_out) {
  IFEM_API_BEGIN
  if (!ctx->assembled || !host_out) throw Error(IFEM_E_BADPARAM, "ifem_uu_block_diag after an assembly, with an output buffer");
  const size_t n = (size_t)ctx->nUo * ctx->dim * ctx->dim;
  if (which == 1) {
    std::vector<double> keep(n);
    IFEM_HIP_CHECK(hipMemcpyAsync(keep.data(), ctx->bjac.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    uu_block_diag_mf(ctx);
    IFEM_HIP_CHECK(hipMemcpyAsync(host_out, ctx->bjac.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipMemcpyAsync(ctx->bjac.p, keep.data(), n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    ctx->bjac_f32_valid = false;
  } else {
    IFEM_HIP_CHECK(hipMemcpyAsync(host_out, ctx->bjac.p, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  }
  IFEM_API_END
}

int ifem_uu_vmult(ifem_ctx *ctx, int dst, int src, int variant) {
  IFEM_API_BEGIN
  if (!vec_ok(dst) || !vec_ok(src) || is_ext(dst) || is_ext(src) || dst == src) throw Error(IFEM_E_BADPARAM, "use two non-ghosted vectors");
  if (!ctx->assembled) throw Error(IFEM_E_BADPARAM, "ifem_uu_vmult called before ifem_ins_assemble");
  const int64_t nul = int64_t(ctx->dim) * ctx->nUl, nuo = int64_t(ctx->dim) * ctx->nUo;
  if ((int64_t)ctx->work.n < nul) ctx->work.alloc(nul);
  double *xe = ctx->work.p; // ghost-extended copy of the velocity part
  v_copy(ctx, nuo, ctx->vec[src].p, xe);
  if (ctx->halo.nranks > 1) halo_exchange(ctx, xe);
  if (variant == IFEM_AINV_GMRES_BJACOBI_MF) apply_uu_mf(ctx, xe, ctx->vec[dst].p);
  else if (variant == IFEM_AINV_MG) apply_uu_mf(ctx, xe, ctx->vec[dst].p, /*single=*/true); // the inner solve's single-precision operator
  else if (variant == IFEM_AINV_GMRES_BJACOBI || variant == IFEM_AINV_GMRES_BJACOBI_F32)
    spmv_uu(ctx, xe, nullptr, ctx->vec[dst].p, variant == IFEM_AINV_GMRES_BJACOBI_F32);
  else throw Error(IFEM_E_BADPARAM, "variant: IFEM_AINV_GMRES_BJACOBI, _F32, _MF or IFEM_AINV_MG (single-precision matrix-free)");
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_precond_vmult(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int dst, int src) {
  IFEM_API_BEGIN
  if (!vec_ok(dst) || !vec_ok(src) || is_ext(dst) || is_ext(src)) throw Error(IFEM_E_BADPARAM, "use non-ghosted vectors");
  ifem_solver_opts def;
  if (!o) { ifem_default_solver_opts(&def); o = &def; }
  ins_precond_vmult(ctx, p, o, ctx->vec[src].p, ctx->vec[dst].p);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  IFEM_API_END
}

int ifem_tpp_ilu_probe(ifem_ctx *ctx, int64_t *rowptr, int32_t *col, double *val, const double *x, double *y, int32_t *levels) {
  IFEM_API_BEGIN
  if (!ctx->assembled || !ctx->has_app) throw Error(IFEM_E_BADPARAM, "ifem_tpp_ilu_probe after ifem_scns_assemble");
  if (ctx->halo.nranks > 1) throw Error(IFEM_E_BADPARAM, "ifem_tpp_ilu_probe: single-rank contexts only");
  if (!rowptr) throw Error(IFEM_E_BADPARAM, "ifem_tpp_ilu_probe: rowptr is required");
  hipStream_t s = ctx->stream;
  bjac_setup(ctx);
  ifem::tpp_numeric(ctx);
  const int64_t n = ctx->Sm.n_rows;
  IFEM_HIP_CHECK(hipMemcpyAsync(rowptr, ctx->Sm.rowptr.p, (size_t)(n + 1) * sizeof(int64_t), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  if (col && val) {
    IFEM_HIP_CHECK(hipMemcpyAsync(col, ctx->Sm.col.p, (size_t)rowptr[n] * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipMemcpyAsync(val, ctx->Tpp.p, (size_t)rowptr[n] * sizeof(double), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
  }
  if (x && y) {
    if (ctx->tune.tpp_ilu_order < 0) throw Error(IFEM_E_BADPARAM, "ifem_tpp_ilu_probe: tpp_ilu_order = -1 keeps no ILU");
    if (!ifem::tpp_ilu_factor(ctx))
      throw Error(IFEM_E_KRYLOV_NOCONV, "ILU(0) of T_pp broke down: zero, tiny or non-finite pivot (|pivot| in [" + std::to_string(ctx->tpp_ilu.pivot_min) +
                                            ", " + std::to_string(ctx->tpp_ilu.pivot_max) + "])");
    DBuf<double> dx, dy;
    dx.upload(x, (size_t)n, s);
    dy.alloc((size_t)n);
    ifem::tpp_ilu_apply(ctx, dx.p, dy.p);
    IFEM_HIP_CHECK(hipMemcpyAsync(y, dy.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, s));
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
    if (levels) *levels = ifem::tpp_ilu_levels(ctx);
  }
  IFEM_API_END
}

int ifem_scns_pc_probe(ifem_ctx *ctx, int which, const double *x, double *y) {
  IFEM_API_BEGIN
  if (!ctx->assembled || !ctx->has_app) throw Error(IFEM_E_BADPARAM, "ifem_scns_pc_probe after ifem_scns_assemble");
  if (ctx->halo.nranks > 1) throw Error(IFEM_E_BADPARAM, "ifem_scns_pc_probe: single-rank contexts only");
  if (!x || !y) throw Error(IFEM_E_BADPARAM, "ifem_scns_pc_probe: null vector");
  hipStream_t s = ctx->stream;
  const size_t n = which == 0 ? (size_t)ctx->dim * ctx->nUo : (size_t)ctx->nPo;
  DBuf<double> dx, dy;
  dx.upload(x, n, s);
  dy.alloc(n);
  ifem::scns_pc_probe(ctx, which, dx.p, dy.p);
  IFEM_HIP_CHECK(hipMemcpyAsync(y, dy.p, n * sizeof(double), hipMemcpyDeviceToHost, s));
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  IFEM_API_END
}

int ifem_tpp_override(ifem_ctx *ctx, const double *val) {
  IFEM_API_BEGIN
  if (!ctx) throw Error(IFEM_E_BADPARAM, "ifem_tpp_override: null context");
  if (ctx->halo.nranks > 1) throw Error(IFEM_E_BADPARAM, "ifem_tpp_override is a single-rank test aid");
  if (!ctx->tpp_valid || !val) throw Error(IFEM_E_BADPARAM, "ifem_tpp_override after ifem_tpp_ilu_probe, with values");
  IFEM_HIP_CHECK(hipMemcpyAsync(ctx->Tpp.p, val, ctx->Tpp.n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  scalar_diag(ctx, ctx->Sm, ctx->Tpp.p, ctx->tpp_diag.p); // (single rank: T_pp lives on the pattern of S_m) the Jacobi fallback of a broken-down ILU divides by THIS matrix's diagonal
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  ctx->tpp_ilu.factored = false;
  IFEM_API_END
}

int ifem_export_csr(ifem_ctx *ctx, int which, int64_t *rowptr, int32_t *col, double *val) {
  IFEM_API_BEGIN
  hipStream_t s = ctx->stream;
  const int dim = ctx->dim;
  const int64_t nuo = dim * ctx->nUo, npo = ctx->nPo, n = nuo + npo, poff = dim * ctx->nUl;
  auto rpA = ctx->Auu.rowptr.download(s), rpT = ctx->Bt.rowptr.download(s), rpB = ctx->B.rowptr.download(s),
       rpM = ctx->Mp.rowptr.download(s);
  std::vector<double> vApp;
  if (ctx->has_app) vApp = ctx->App.download(s);
  // pattern: [A_uu | B^T ; B | M_p-pattern]  (explicit zeros kept so that both `which` share one pattern)
  rowptr[0] = 0;
  for (int64_t A = 0; A < ctx->nUo; ++A)
    for (int c = 0; c < dim; ++c) rowptr[A * dim + c + 1] = (rpA[A + 1] - rpA[A]) * dim + (rpT[A + 1] - rpT[A]);
  for (int64_t i = 0; i < npo; ++i) rowptr[nuo + i + 1] = (rpB[i + 1] - rpB[i]) * dim + (rpM[i + 1] - rpM[i]);
  for (int64_t i = 0; i < n; ++i) rowptr[i + 1] += rowptr[i];
  if (!col || !val) return IFEM_OK;
  auto cA = ctx->Auu.col.download(s), cT = ctx->Bt.col.download(s), cB = ctx->B.col.download(s), cM = ctx->Mp.col.download(s);
  auto vA = ctx->Auu.val.download(s), vT = ctx->Bt.val.download(s), vB = ctx->B.val.download(s), vM = ctx->Mp.val.download(s);
  auto dM = ctx->diagMu.download(s);
  for (int64_t A = 0; A < ctx->nUo; ++A) {
    const int64_t rs = rpA[A], len = rpA[A + 1] - rs, ts = rpT[A], tlen = rpT[A + 1] - ts;
    for (int c = 0; c < dim; ++c) {
      int64_t o = rowptr[A * dim + c];
      for (int64_t k = 0; k < len; ++k)
        for (int d = 0; d < dim; ++d) {
          col[o] = cA[rs + k] * dim + d;
          if (which == 0) val[o] = vA[uu_base(rs, len, k, dim * dim) + (c * dim + d) * uu_estride(len)];
          else val[o] = (cA[rs + k] == A && c == d) ? dM[A * dim + c] : 0.0;
          ++o;
        }
      for (int64_t k = 0; k < tlen; ++k) {
        col[o] = int32_t(poff + cT[ts + k]);
        val[o] = (which == 0) ? vT[ts * dim + c * tlen + k] : 0.0;
        ++o;
      }
    }
  }
  for (int64_t i = 0; i < npo; ++i) {
    const int64_t rs = rpB[i], len = rpB[i + 1] - rs, ms = rpM[i], mlen = rpM[i + 1] - ms;
    int64_t o = rowptr[nuo + i];
    for (int64_t k = 0; k < len; ++k)
      for (int d = 0; d < dim; ++d) {
        col[o] = cB[rs + k] * dim + d;
        val[o] = (which == 0) ? vB[rs * dim + d * len + k] : 0.0;
        ++o;
      }
    for (int64_t k = 0; k < mlen; ++k) {
      col[o] = int32_t(poff + cM[ms + k]);
      val[o] = (which == 0) ? (ctx->has_app ? vApp[ms + k] : 0.0) : vM[ms + k];
      ++o;
    }
  }
  IFEM_API_END
}

// rows [row0, row0 + nrows) of the CSR ifem_export_csr describes, downloading only the slices of the device arrays those rows own
// (the whole A_uu of the 128^3 channel is 78 GB: ifem_export_csr cannot be called there)
int ifem_export_rows(ifem_ctx *ctx, int which, int64_t row0, int64_t nrows, int64_t *rowptr, int32_t *col, double *val) {
  IFEM_API_BEGIN
  hipStream_t s = ctx->stream;
  const int dim = ctx->dim, bs = dim * dim;
  const int64_t nuo = dim * ctx->nUo, npo = ctx->nPo, n = nuo + npo, poff = dim * ctx->nUl;
  if (!rowptr || row0 < 0 || nrows < 0 || row0 + nrows > n) throw Error(IFEM_E_BADPARAM, "ifem_export_rows: row range outside the local system");
  rowptr[0] = 0;
  // ---- velocity rows: nodes A0 .. A1-1 cover them
  const int64_t u0 = std::min(row0, nuo), u1 = std::min(row0 + nrows, nuo);
  if (u1 > u0) {
    const int64_t A0 = u0 / dim, A1 = (u1 + dim - 1) / dim;
    auto rpA = download_range(ctx->Auu.rowptr, A0, A1 - A0 + 1, s), rpT = download_range(ctx->Bt.rowptr, A0, A1 - A0 + 1, s);
    for (int64_t r = u0; r < u1; ++r) {
      const int64_t a = r / dim - A0;
      rowptr[r - row0 + 1] = rowptr[r - row0] + (rpA[a + 1] - rpA[a]) * dim + (rpT[a + 1] - rpT[a]);
    }
    if (col && val) {
      const int64_t b0 = rpA[0], b1 = rpA[A1 - A0], t0 = rpT[0], t1 = rpT[A1 - A0];
      auto cA = download_range(ctx->Auu.col, b0, b1 - b0, s);
      auto vA = download_range(ctx->Auu.val, b0 * bs, (b1 - b0) * bs, s);
      auto cT = download_range(ctx->Bt.col, t0, t1 - t0, s);
      auto vT = download_range(ctx->Bt.val, t0 * dim, (t1 - t0) * dim, s);
      auto dM = download_range(ctx->diagMu, A0 * dim, (A1 - A0) * dim, s);
      for (int64_t r = u0; r < u1; ++r) {
        const int64_t A = r / dim, a = A - A0;
        const int c = int(r % dim);
        const int64_t rs = rpA[a], len = rpA[a + 1] - rs, ts = rpT[a], tlen = rpT[a + 1] - ts;
        int64_t o = rowptr[r - row0];
        for (int64_t k = 0; k < len; ++k)
          for (int d = 0; d < dim; ++d) {
            col[o] = cA[rs - b0 + k] * dim + d;
            if (which == 0) val[o] = vA[uu_base(rs, len, k, bs) + (c * dim + d) * uu_estride(len) - b0 * bs];
            else val[o] = (cA[rs - b0 + k] == A && c == d) ? dM[a * dim + c] : 0.0;
            ++o;
          }
        for (int64_t k = 0; k < tlen; ++k) {
          col[o] = int32_t(poff + cT[ts - t0 + k]);
          val[o] = (which == 0) ? vT[ts * dim + c * tlen + k - t0 * dim] : 0.0;
          ++o;
        }
      }
    }
  }
  // ---- pressure rows
  const int64_t p0 = std::max(row0, nuo) - nuo, p1 = row0 + nrows - nuo;
  if (p1 > p0) {
    auto rpB = download_range(ctx->B.rowptr, p0, p1 - p0 + 1, s), rpM = download_range(ctx->Mp.rowptr, p0, p1 - p0 + 1, s);
    for (int64_t i = p0; i < p1; ++i) {
      const int64_t j = i - p0, r = nuo + i - row0;
      rowptr[r + 1] = rowptr[r] + (rpB[j + 1] - rpB[j]) * dim + (rpM[j + 1] - rpM[j]);
    }
    if (col && val) {
      const int64_t b0 = rpB[0], b1 = rpB[p1 - p0], m0 = rpM[0], m1 = rpM[p1 - p0];
      auto cB = download_range(ctx->B.col, b0, b1 - b0, s);
      auto vB = download_range(ctx->B.val, b0 * dim, (b1 - b0) * dim, s);
      auto cM = download_range(ctx->Mp.col, m0, m1 - m0, s);
      auto vM = download_range(ctx->Mp.val, m0, m1 - m0, s);
      std::vector<double> vApp;
      if (ctx->has_app) vApp = download_range(ctx->App, m0, m1 - m0, s);
      for (int64_t i = p0; i < p1; ++i) {
        const int64_t j = i - p0;
        const int64_t rs = rpB[j], len = rpB[j + 1] - rs, ms = rpM[j], mlen = rpM[j + 1] - ms;
        int64_t o = rowptr[nuo + i - row0];
        for (int64_t k = 0; k < len; ++k)
          for (int d = 0; d < dim; ++d) {
            col[o] = cB[rs - b0 + k] * dim + d;
            val[o] = (which == 0) ? vB[rs * dim + d * len + k - b0 * dim] : 0.0;
            ++o;
          }
        for (int64_t k = 0; k < mlen; ++k) {
          col[o] = int32_t(poff + cM[ms - m0 + k]);
          val[o] = (which == 0) ? (ctx->has_app ? vApp[ms - m0 + k] : 0.0) : vM[ms - m0 + k];
          ++o;
        }
      }
    }
  }
  IFEM_API_END
}

// the stored block pattern of A_uu for the nodes [node0, node0 + n_nodes): ABSOLUTE block offsets (block k of the array holds the
// values [k * dim^2, (k + 1) * dim^2)) and block columns in storage order -- what the scatter of the cell kernel addresses
int ifem_export_uu_pattern(ifem_ctx *ctx, int64_t node0, int64_t n_nodes, int64_t *rowptr, int32_t *col) {
  IFEM_API_BEGIN
  if (!rowptr || node0 < 0 || n_nodes < 0 || node0 + n_nodes > ctx->Auu.n_rows) throw Error(IFEM_E_BADPARAM, "ifem_export_uu_pattern: node range outside the owned rows");
  hipStream_t s = ctx->stream;
  auto rp = download_range(ctx->Auu.rowptr, node0, n_nodes + 1, s);
  std::copy(rp.begin(), rp.end(), rowptr);
  if (col) {
    auto c = download_range(ctx->Auu.col, rp.front(), rp.back() - rp.front(), s);
    std::copy(c.begin(), c.end(), col);
  }
  IFEM_API_END
}

int ifem_get_timing(ifem_ctx *ctx, ifem_timing *t) {
  IFEM_API_BEGIN
  *t = ctx->timing; // spmv_uu_bytes is set by the profiled launches themselves (linalg.hip::spmv_uu)
  IFEM_API_END
}

int ifem_comm_stats_get(ifem_ctx *ctx, ifem_comm_stats *out, int reset) {
  IFEM_API_BEGIN
  if (!out) throw Error(IFEM_E_BADPARAM, "ifem_comm_stats_get: null output");
  ifem::comm_stats(ctx, out, reset != 0);
  IFEM_API_END
}

int ifem_comm_stats_level(ifem_ctx *ctx, int level, ifem_comm_stats *out) {
  IFEM_API_BEGIN
  if (!out || level < 0) throw Error(IFEM_E_BADPARAM, "ifem_comm_stats_level: an output and a level >= 0");
  ifem_ctx *c = ctx;
  for (int k = 0; k < level && c; ++k) c = c->mg_coarse;
  if (!c) throw Error(IFEM_E_BADPARAM, "ifem_comm_stats_level: the chain has fewer levels");
  ifem::comm_stats(c, out, false, /*single_level=*/true);
  IFEM_API_END
}

int ifem_test_restart_fits(ifem_ctx *ctx, int columns) {
  if (!ctx || columns < 0) return IFEM_E_BADPARAM;
  ctx->test_restart_fits = columns;
  return IFEM_OK;
}

int ifem_inner_restart_length(ifem_ctx *ctx) { return ctx ? ctx->inner_restart_eff : IFEM_E_BADPARAM; }

int ifem_vcycle_graph_stats(ifem_ctx *ctx, uint64_t *captures, uint64_t *launches) {
  IFEM_API_BEGIN
  if (captures) *captures = ctx->vc_graph.captures + ctx->sm_graph.captures + ctx->pa_graph.captures;
  if (launches) *launches = ctx->vc_graph.launches + ctx->sm_graph.launches + ctx->pa_graph.launches;
  IFEM_API_END
}

int ifem_kprof_begin(ifem_ctx *ctx) {
  IFEM_API_BEGIN
  ifem::KProf &k = ifem::kprof_root(ctx);
  k.recs.clear();
  k.used = 0;
  k.depth = 0;
  k.on = true;
  IFEM_API_END
}

int ifem_kprof_end(ifem_ctx *ctx, ifem_kprof_entry *out, int32_t max_entries) {
  int written = 0;
  IFEM_API_BEGIN
  ifem::KProf &k = ifem::kprof_root(ctx);
  k.on = false;
  IFEM_HIP_CHECK(hipSetDevice(ctx->device));
  IFEM_HIP_CHECK(hipDeviceSynchronize());
  ifem_kprof_entry acc[IFEM_KC_COUNT];
  for (int f = 0; f < IFEM_KC_COUNT; ++f) acc[f] = {f, 0u, 0.0, 0.0, 0.0};
  for (const auto &r : k.recs) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, k.ev[r.e0], k.ev[r.e1]) != hipSuccess) continue; // a scope cut short by an exception
    ifem_kprof_entry &a = acc[r.cat >= 0 && r.cat < IFEM_KC_COUNT ? r.cat : IFEM_KC_OTHER];
    a.scopes++; a.ms += ms; a.bytes += r.bytes; a.flops += r.flops;
  }
  k.recs.clear();
  k.used = 0;
  for (int f = 0; f < IFEM_KC_COUNT && out && written < max_entries; ++f)
    if (acc[f].scopes) out[written++] = acc[f];
  return written;
  IFEM_API_END
}

const char *ifem_kprof_family_name(int32_t family) {
  static const char *names[IFEM_KC_COUNT] = {"assemble_cells", "zero_fill", "spmv_uu", "spmv_b_bt", "mf_cell", "mf_gather", "spmv_sm", "spmv_mp",
                                             "mdot", "maxpy", "vector_ops", "mg_transfer", "smoother_setup", "cg_recurrence", "schur_setup", "other", "tpp_ilu"};
  return family >= 0 && family < IFEM_KC_COUNT ? names[family] : "?";
}

int ifem_set_ainv_kind(ifem_ctx *ctx, int kind) { ctx->want_shat = kind == IFEM_AINV_SCALAR_GMRES; return IFEM_OK; }
int ifem_set_profiling(ifem_ctx *ctx, int on) { ctx->profile = on != 0; ++ctx->graph_epoch; return IFEM_OK; }
int ifem_synchronize(ifem_ctx *ctx) {
  IFEM_API_BEGIN
  IFEM_HIP_CHECK(hipSetDevice(ctx->device));
  IFEM_HIP_CHECK(hipDeviceSynchronize());
  IFEM_API_END
}

int ifem_comm_unique_id(uint8_t out[128]) { return ifem::comm_unique_id(out); }
int ifem_comm_selftest(int device) {
  IFEM_API_BEGIN
  ifem::comm_selftest(device);
  IFEM_API_END
}
void *ifem_local_world_create(int nranks) { return ifem::local_world_create(nranks); }
void ifem_local_world_destroy(void *w) { ifem::local_world_destroy(w); }

} // extern "C"
