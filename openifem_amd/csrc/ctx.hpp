// ctx.hpp -- device-side state of one ifem_ctx (one per GPU / process).
#pragma once
#include <array>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <utility>
#include <string>
#include <vector>
#include "../../include/ifem_hip.h"

namespace ifem {

struct Error : std::runtime_error {
  int code;
  Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

#define IFEM_HIP_CHECK(expr)                                                                        \
  do {                                                                                              \
    hipError_t e_ = (expr);                                                                         \
    if (e_ != hipSuccess)                                                                           \
      throw ::ifem::Error(IFEM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_) + " at " + \
                                          __FILE__ + ":" + std::to_string(__LINE__));               \
  } while (0)

// ctx->scal / ctx->h_scal slots: [0,64) fused dot products, [64,128) multi-axpy coefficients, [128,256) recurrence scalars
// of the device-resident CG (linalg.hip), [kScalStageOff, +kScalStage) staging of the RCCL all-reduce (comm.hip)
constexpr int kScalSlots = 512, kScalStageOff = 256, kScalStage = 256;

template <class T>
struct DBuf { // owning device buffer
  T *p = nullptr;
  size_t n = 0;
  DBuf() = default;
  DBuf(const DBuf &) = delete;
  DBuf &operator=(const DBuf &) = delete;
  ~DBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    n = 0;
  }
  void swap(DBuf &o) {
    std::swap(p, o.p);
    std::swap(n, o.n);
  }
  // `what` names the array in the error message of a failed allocation (an over-sized mesh must say which table did not fit)
  void alloc(size_t count, const char *what = nullptr) {
    release();
    if (!count) return;
    const hipError_t e = hipMalloc((void **)&p, count * sizeof(T));
    if (e != hipSuccess) {
      p = nullptr;
      (void)hipGetLastError(); // the failed allocation must not poison the next call's error check
      size_t fr = 0, tot = 0;
      (void)hipMemGetInfo(&fr, &tot);
      throw ::ifem::Error(IFEM_E_HIP, std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) + " bytes for " + (what ? what : "a device array") +
                                          " (" + std::to_string(count) + " x " + std::to_string(sizeof(T)) + " B): " + hipGetErrorString(e) + "; device memory free " +
                                          std::to_string(fr >> 20) + " of " + std::to_string(tot >> 20) + " MiB");
    }
    n = count;
  }
  void upload(const T *h, size_t count, hipStream_t s) {
    alloc(count);
    if (count) IFEM_HIP_CHECK(hipMemcpyAsync(p, h, count * sizeof(T), hipMemcpyHostToDevice, s));
  }
  std::vector<T> download(hipStream_t s) const {
    std::vector<T> h(n);
    if (n) {
      IFEM_HIP_CHECK(hipMemcpyAsync(h.data(), p, n * sizeof(T), hipMemcpyDeviceToHost, s));
      IFEM_HIP_CHECK(hipStreamSynchronize(s));
    }
    return h;
  }
};

// Row-planar block-CSR: row r owns blocks [rowptr[r], rowptr[r+1]); entry e (of BS per block) of the k-th
// block of the row lives at val[BS*rowptr[r] + e*len_r + k].  Consecutive lanes (k) read consecutive
// doubles for every e: fully coalesced without LDS staging.  BS = dim*dim (A_uu), dim (B, B^T) or 1.
// XCD-aware block index: MI355X dispatches block b to XCD b % 8 (observed, not a contract -- only speed depends on it);
// the remap hands every XCD one contiguous range of the Morton-ordered cells, so the nodes shared by neighbouring cells
// are fetched into ONE private L2 instead of eight.  Bijective for any grid size.
__device__ inline unsigned xcd_swizzle(unsigned bid, unsigned nwg) {
  constexpr unsigned NX = 8;
  const unsigned q = nwg / NX, r = nwg % NX, xcd = bid % NX, k = bid / NX;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

// Value layout of A_uu (bs = dim*dim entries per block).  Block-interleaved: the bs entries of a block are contiguous,
// so the assembly scatters one node pair into one or two 64-byte segments (the f64 atomic unit retires ~24 G segments/s
// whatever the number of lanes that hit a segment) instead of into bs planes.  B, B^T, M_p, S_m stay row-planar
// (entry e of the k-th block of a row at val[bs*rowptr + e*len + k]).  IFEM_UU_INTERLEAVED=0 restores the planar A_uu.
#ifndef IFEM_UU_INTERLEAVED
#define IFEM_UU_INTERLEAVED 1
#endif
__host__ __device__ inline int64_t uu_base(int64_t rs, int64_t len, int64_t k, int bs) { // offset of entry 0 of block k
#if IFEM_UU_INTERLEAVED
  (void)len;
  return (rs + k) * bs;
#else
  (void)len;
  return rs * bs + k;
#endif
}
__host__ __device__ inline int64_t uu_estride(int64_t len) { // distance between consecutive entries of one block
#if IFEM_UU_INTERLEAVED
  (void)len;
  return 1;
#else
  return len;
#endif
}

struct PlanarCsr {
  int64_t n_rows = 0, nnzb = 0;
  int bs = 1, max_row = 0;
  DBuf<int64_t> rowptr;
  DBuf<int32_t> col;
  DBuf<double> val;
  // several ranks: rows whose columns are all owned (they can be multiplied while the halo is in flight) listed first,
  // then the rows that read a ghost column; built on first use (setup.hip::build_row_split)
  DBuf<int32_t> split_rows;
  int64_t n_interior = -1, n_boundary = 0;
};

// FE tables for one (dim, kv): reference-cell shape values/gradients at the volume quadrature points.
struct FeTables {
  int dim, kv, nu, np, nq;
  double phi[27 * 27];       // [q][a]   Q_kv
  double dphi[27 * 27 * 3];  // [q][a][e]
  double psi[27 * 8];        // [q][b]   Q1 (pressure + geometry mapping)
  double dpsi[27 * 8 * 3];   // [q][b][e]
  double w[27];
  // face quadrature: per face f, nqf points: Q_kv values and Q1 gradients
  int nqf;
  double fphi[6 * 9 * 27];     // [f][qf][a]
  double fdpsi[6 * 9 * 8 * 3]; // [f][qf][v][e]
  double fw[9];
};
void build_fe_tables(FeTables &t, int dim, int kv);

struct Halo {
  int rank = 0, nranks = 1;
  std::vector<int> nbr;
  std::vector<int32_t> send_u_ptr, recv_u_ptr, send_p_ptr, recv_p_ptr;
  DBuf<int32_t> send_u_idx, send_p_idx;
  // 2-deep pressure halo of the distributed explicit S_m (empty: not available)
  std::vector<int32_t> send_s_ptr, recv_s_ptr;
  DBuf<int32_t> send_s_idx, sm_box_id;
  DBuf<int64_t> own_p_gid;
  int64_t p_lattice_n[3] = {1, 1, 1}, sm_box_lo[3] = {0, 0, 0}, sm_box_n[3] = {0, 0, 0};
  int64_t n_s_cols = 0; // nPo + far nodes
  bool has_s = false;
  DBuf<double> sendbuf;
  void *comm = nullptr;  // ncclComm_t
  // overlapped exchanges (halo_start / halo_wait): a second communicator (ncclCommSplit of `comm`) whose send/recv groups
  // run on `hstream` (high priority) while the context stream works on the rows / cells that need no ghost value
  void *comm2 = nullptr;
  hipStream_t hstream = nullptr;
  bool owns_hstream = false;
  hipEvent_t ev_pack = nullptr, ev_done = nullptr;
  void *local = nullptr; // LocalWorld* (in-process virtual ranks, validation transport)
  const void *rev_src = nullptr; // local world only: the extended vector whose ghosts the peers fetch in a reverse exchange
  // counters since the last ifem_comm_stats(reset): halo exchanges (forward + reverse), all-reduces ordered on the stream,
  // all-reduces that made the host wait for the device
  uint64_t n_exchanges = 0, n_allreduce_dev = 0, n_allreduce_host = 0, n_allreduce_vec = 0;
};

} // namespace ifem

namespace ifem {
// ILU(0) of the explicit T_pp on its own pattern (tpp.hip): analysis once per pattern, numeric factors per Newton iteration
struct TppIlu {
  DBuf<int32_t> ent, n_low, diag, rows_f, rows_b;
  std::vector<int64_t> lvl_f, lvl_b; // level pointers into rows_f / rows_b (host: the launch plan)
  DBuf<int64_t> d_lvl_f, d_lvl_b;    // the same on the device (batched runs of small levels)
  std::vector<std::array<int32_t, 2>> plan_f, plan_b; // {level, -1}: one wide level; {l0, l1}: a run of small levels in one launch
  DBuf<double> LU, t0, t1; // factors; scratch of the Jacobi-sweep triangular solves
  bool analysed = false, factored = false;
  double pivot_min = 0, pivot_max = 0;
  bool broken = false; // the last factorisation met a zero / non-finite pivot: the caller falls back to Jacobi
  DBuf<unsigned long long> chk; // [3] device: min |pivot|, max |pivot| (bit patterns), non-finite entries
  int order_kind = 0, order_used = 0; // ifem_tuning::tpp_ilu_order the analysis was made for; the order it chose (0 natural, 1 multicolour)
};
// ILU(0) with dim x dim blocks in the natural row order (bilu.hip): the two Euclid factorisations of the SUPG block preconditioner
struct BIlu {
  int dim = 1, max_row = 0;
  int64_t n = 0, nnz = 0;
  DBuf<int64_t> rp, srcpos;          // the factor's own column-sorted CSR (owned x owned) and where each entry sits in the source values
  DBuf<int32_t> col, diag, rows_f, rows_b;
  std::vector<int64_t> lvl_f, lvl_b; // elimination levels (host: launch plan of the factorisation)
  DBuf<int64_t> d_lvl_f, d_lvl_b;
  DBuf<double> LU, dinv, t0, t1, t2; // factors (unit-lower L and U without its diagonal blocks), inverse pivot blocks, sweep scratch
  DBuf<unsigned long long> chk;
  bool analysed = false, factored = false, broken = false;
  double pivot_min = 0, pivot_max = 0;
};
// hanging-node constraint lines x[dof_i] = sum_k w_k x[master_k] (closed), see hanging.hip
struct Hanging {
  int32_t n = 0;
  DBuf<int32_t> dof, ptr, master; // local (ghost-extended) dof ids; lines of owned AND ghost hanging dofs
  DBuf<double> w, d, x, c0, y; // weights, diagonal of the hanging rows, extended scratch input / inhomogeneity / output
  DBuf<int> flag;
  std::vector<int32_t> host_dof;
  bool active = false; // some rank holds hanging lines: every rank takes part in their exchanges
};
} // namespace ifem

namespace ifem {
// the solid as MPI::FSI sees it on every rank and the scratch of the device-side FSI inputs (fsi.hip)
struct FsiState {
  int32_t nv = 0, nc = 0, nbf = 0;
  DBuf<double> vert;   // [nv][dim]
  DBuf<int32_t> cells; // [nc][2^dim]
  DBuf<double> rec;    // per solid cell: bounding box lo[dim], hi[dim], then the vertex coordinates [2^dim][dim]
  DBuf<double> bface;  // dim 2: per boundary face p1x p1y p2x p2y
  DBuf<double> vel, acc, stress;
  double box[6] = {0, 0, 0, 0, 0, 0}; // solid_box: lo, hi per direction
  DBuf<int32_t> bin_ptr, bin_cells;   // uniform grid over solid_box -> solid cells whose bounding box meets the bin
  int G[3] = {1, 1, 1};
  double inv_h[3] = {0, 0, 0};
  bool valid = false, has_fields = false, has_stress = false;
  DBuf<uint32_t> order_min, first; // first-touch cell of every local velocity node: (cell << 5 | local node) or 0xFFFFFFFF
  DBuf<int32_t> cand;              // velocity nodes whose support point lies in solid_box
  DBuf<uint8_t> taken;             // dofs that already carry a boundary or hanging line
  DBuf<uint8_t> prev[2];           // the flags of the two constraint sets before the merge (identity of the sets afterwards)
  DBuf<int64_t> counters;          // [8] device counters
  // uniform grid over the local fluid cells (point evaluation of the fluid solution, built on first use)
  DBuf<int32_t> fbin_ptr, fbin_cells;
  int fG[3] = {1, 1, 1};
  double fbox_lo[3] = {0, 0, 0}, finv_h[3] = {0, 0, 0};
  bool fbins_valid = false;
};
} // namespace ifem

namespace ifem {
// Per-kernel-family timing of one profiled step (ifem_kprof_begin / ifem_kprof_end): every launch wrapper opens a KScope, which
// records an event pair on the context stream around its launches -- nothing waits until ifem_kprof_end reads them all, so the
// queue keeps running ahead of the host as in a timed step.  One recorder per level chain: coarse multigrid levels log into
// the finest level's (they run on its stream).  The reference's counterpart: the TimerOutput sections of InsIM
// (mpi_insim.cpp:33,70,87,125,155,368).
struct KProf {
  bool on = false;
  int depth = 0; // a scope opened inside another one is part of the outer scope
  std::vector<hipEvent_t> ev;
  size_t used = 0;
  struct Rec { int cat; uint32_t e0, e1; double bytes, flops; };
  std::vector<Rec> recs;
  ~KProf() { for (hipEvent_t e : ev) (void)hipEventDestroy(e); }
};
struct KScope {
  KProf *k = nullptr;
  hipStream_t s = nullptr;
  uint32_t e1 = 0;
  inline KScope(ifem_ctx *ctx, int cat, double bytes = 0, double flops = 0);
  inline ~KScope();
  KScope(const KScope &) = delete;
  KScope &operator=(const KScope &) = delete;
};
} // namespace ifem

namespace ifem {
// one CSR transfer table of the multigrid hierarchy (row-parallel gather on the device)
struct MgCsr {
  int64_t n_rows = 0;
  DBuf<int64_t> ptr;
  DBuf<int32_t> col;
  DBuf<double> w;
};
} // namespace ifem

namespace ifem {
// reference tables of the 3D Q2/Q1 cell kernel (assemble3.hip), built on the host, copied into LDS by every workgroup
struct Tabs3 {
  double Nz[16], dNz[16];           // [q2 (4)][a2 (4)] 1D shape values / derivatives, zero for q2 = 3 or a2 = 3 (MFMA padding)
  double N2[81], DX2[81], DY2[81];  // [(q0 q1)][(a0 a1)]: N_x N_y, N'_x N_y, N_x N'_y
  double psi[27 * 8];               // [q][b] trilinear (pressure, geometry)
};
// per-cell records of that kernel (setup.hip::ensure_scat3): kAsm3Rec uint16 = [row tile 2][lane 64][column tile 2][r 4] holding
// (position of the block in its A_uu row - rank | rank << 9), rank = rank of that position among the row's 27 blocks of this cell,
// 0xFFFF = no block; kAsm3Hdr bytes = perm[32] (tile column -> local node, the cell's nodes in the order of their ids) | iperm[32] |
// permp[8] (the same for the pressure nodes) | srow[32] (position of the row's first block inside a 64-byte segment, in doubles) | padding
constexpr int kAsm3Rec = 1024, kAsm3Hdr = 128;

} // namespace ifem

struct ifem_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = true; // false: a coarse multigrid level running on the stream of the finest level (ifem_mg_attach)
  int dim = 0, kv = 0, nu = 0, np = 0, nq = 0;
  int64_t n_cells = 0;
  int64_t nUo = 0, nUl = 0, nPo = 0, nPl = 0; // velocity / pressure nodes owned / local
  int64_t n_local = 0;                        // dim*nUl + nPl
  int64_t n_global_u = 0, n_global_p = 0;     // owned counts summed over ranks (iteration limits must agree on all ranks)
  ifem::FeTables fe;
  ifem::DBuf<ifem::FeTables> d_fe;
  // mesh
  ifem::DBuf<double> vcoords;
  ifem::DBuf<int32_t> cell_unodes, cell_pnodes, cell_face_bid, indicator;
  // matrices
  ifem::PlanarCsr Auu; // rows: owned velocity nodes, cols: local velocity nodes, bs = dim*dim
  ifem::PlanarCsr Bt;  // rows: owned velocity nodes, cols: local pressure nodes, bs = dim   (block (0,1))
  ifem::PlanarCsr B;   // rows: owned pressure nodes, cols: local velocity nodes, bs = dim   (block (1,0))
  ifem::PlanarCsr Mp;  // rows: owned pressure nodes, cols: local pressure nodes, bs = 1     (mass (1,1))
  // SCnsIM (slightly compressible, SUPG): pressure-pressure block on the M_p pattern, nodal stress fields, cell fields
  ifem::DBuf<double> App, app_diag, stress, fsi_stress, sigma_pml, body_force, xinv, eddy_viscosity;
  bool has_app = false, stress_valid = false;
  ifem::PlanarCsr uinc;  // velocity node -> (cell << 5 | local index) incidence lists (gather stage of the matrix-free apply)
  // several ranks: the matrix-free apply keeps its own copy of the cell tables with the cells whose nodes are all owned
  // first (they are processed while the halo is in flight); the incidence lists then refer to this numbering
  ifem::DBuf<int32_t> mf_cell_unodes;
  ifem::DBuf<double> mf_vcoords;
  int64_t mf_n_interior = -1;
  ifem_tuning tune{};    // ifem_set_tuning
  ifem::PlanarCsr Sm;  // mass_schur(1,1) = B diag(M_u)^-1 B^T, explicit (single rank only; empty otherwise)
  bool sm_valid = false;
  ifem::DBuf<float> B_f32, Bt_f32; // single-precision copies for the matrix-free S_m of the approximate-preconditioner kinds
  bool bbt_f32_valid = false;
  ifem::DBuf<double> xs_ext; // [halo.n_s_cols] input of the distributed S_m SpMV: owned entries + 2-deep far nodes
  ifem::DBuf<float> Sm_f32; // single-precision copy of the S_m values for its SpMV (approximate-preconditioner kinds)
  bool sm_f32_valid = false;
  ifem::DBuf<float> Mp_f32; // the same for M_p (CG(M_p) of the preconditioner)
  bool mp_f32_valid = false;
  int64_t sm_key = -1;
  // identity of the constrained-dof SET of each AffineConstraints object (which dofs, not their values): B, B^T, M_p,
  // diag(M_u) and S_m depend on nothing else, so zero_ / nonzero_constraints with the same lines share one cache entry and
  // re-making identical constraints (time-dependent boundary values) keeps it
  // (decided by comparing the flag arrays on the device, api.hip::flags_differ)
  int64_t flag_id[2] = {0, 0}, flag_counter = 0;
  std::vector<uint8_t> seen_scratch; // ifem_set_constraints: duplicate detection over the local dofs, all zero between calls
  // scalar velocity operator S^ = mu K + rho C(u) + rho/dt M on the A_uu block pattern (IFEM_AINV_SCALAR_*)
  ifem::DBuf<double> Shat, shat_dinv;
  ifem::DBuf<float> Shat_f32;
  bool want_shat = false, shat_valid = false, shat_aux_valid = false;
  int asm_constraint_set = 0;
  bool geo_valid = false; // B, B^T, M_p, diag(M_u) hold the blocks of constraint set geo_key (assemble.hip)
  int64_t geo_key = -1;
  // the same blocks integrated WITHOUT any constraint (functions of the mesh alone), kept once built: the blocks of a new
  // constrained-dof set are masked copies of them instead of a re-integration (M_p and diag(M_u) do not depend on the set)
  ifem::DBuf<double> B0, Bt0;
  bool geo0_valid = false;
  int geo_unchanged = 0; // consecutive assemblies OF THE FINEST LEVEL that kept the cached blocks (assemble.hip: the copies are released at kGeoKeep = 2 ...)
  int geo_set_changes = 0; // ... unless the constrained-dof set has ever changed after the first assembly (an FSI run): then they stay
  uint64_t geo_seen_asm = 0; // a multigrid level: the finest level's asm_version its geo_unchanged last counted
  // S_m of the unconstrained blocks (same mesh-only idea): a constrained-dof set only changes the rows whose B row touches a
  // constrained dof, so S_m of a new set = this copy with those rows recomputed (linalg.hip::schur_numeric)
  ifem::DBuf<double> Sm0;
  bool sm0_valid = false;
  ifem::DBuf<int32_t> sm_rows;
  int64_t geo_refresh_stamp = -1; // a coarse multigrid level: the finest level's assembly its blocks were last refreshed for
  ifem::Hanging hang; // hanging-node lines (hanging.hip)
  ifem::FsiState fsi; // solid + scratch of the device-side FSI inputs (fsi.hip)
  // multigrid (ifem_mg_attach): the next coarser level (not owned) and the pressure transfers to it; per-level state of
  // the S_m V-cycle: 1/diag(S_m), largest eigenvalue of D^-1 S_m, scratch vectors [nPl]
  ifem_ctx *mg_coarse = nullptr;
  ifem_ctx *mg_fine = nullptr; // back link (not owned): the level this context hangs below, whose stream(s) it borrows
  bool mg_replica = false;     // mg_coarse is a REPLICATED single-rank context of the whole coarse mesh (several ranks here): restrictions are
                               // summed over the ranks by a vector all-reduce, nothing below this level exchanges anything
  ifem::MgCsr mg_Pp, mg_Rp, mg_Pu, mg_Ru;
  ifem::DBuf<int32_t> mg_inj_u; // [nUo of the coarse level] coincident owned velocity node of this level
  ifem::DBuf<uint8_t> mg_Pu_mask, mg_Ru_mask; // per weight 8 bytes: {column | components dropped by the Dirichlet flags of the two levels << 29, weight as float} (mg.hip::mg_csr_mask)
  bool uu_is_stored = true;                   // the last full assembly scattered A_uu (false: ifem_tuning::stored_uu = 0 took the matrix-free path)
  bool inhom_any[2] = {false, false};         // constraint object `which` carries a non-zero inhomogeneity somewhere (any rank)
  uint64_t graph_epoch = 0;                   // bumped by ifem_set_tuning / ifem_set_profiling / ifem_mg_attach: part of every hipGraph replay key
  int64_t mg_mask_key[2] = {-1, -1};          // constrained-dof sets (flag ids of this level and the coarser one) of the masks
  ifem::DBuf<double> sm_dinv, mg_vec[6], mgu_vec[5];
  ifem::DBuf<float> mguf_vec[5]; // single-precision level vectors of the A_uu V-cycle (solver.hip)
  double sm_lmax = 0, uu_lmax = 0;
  double uu_evn = 0, uu_lmax_evn = -1; // ||evaluation point|| of the current assembly (finest level) / of the cached A_uu bound
  int64_t uu_evn_asm = -1;
  // last iterate of the two power iterations: the next estimate (new constrained-dof set, same mesh) starts from it
  ifem::DBuf<double> sm_eig, uu_eig;
  int64_t uu_lmax_asm = -1; // ifem_tuning::geo_cache = 2: the finest level's assembly the A_uu bound was last refreshed for
  int64_t asm_version = 0, uu_mg_version = -1; // full assemblies done / the assembly the A_uu V-cycle data belong to
  double uu_lmax_key[6] = {0, 0, 0, 0, 0, -1};  // (mu, rho, gamma, dt, noconv, constrained-dof set) of the cached eigenvalue bound
  int64_t sm_version = 0, sm_mg_version = -1; // S_m values rebuilt / the version the V-cycle data belong to
  // explicit T_pp = A_pp - A_pv Binv A_vp on the pattern of Sm and its dense LU (tpp.hip)
  ifem::DBuf<double> Tpp, tpp_diag;
  bool tpp_valid = false;
  ifem::TppIlu tpp_ilu; // level-scheduled ILU(0) of T_pp (tpp.hip)
  // the reference's structure (ifem_tuning::scns_pc = 2): ILU(0)(A_vv), B2pp = A_pp - A_pv rowsum(|A_vv|)^-1 A_vp and its ILU(0)
  ifem::BIlu pvv_ilu, b2_ilu;
  ifem::DBuf<double> B2pp, rsinv;
  bool b2_valid = false;
  ifem::PlanarCsr TppPat; // several ranks: pattern of the owned x owned block of T_pp (the per-rank ILU(0), tpp.hip)
  // matrix-free A_uu (IFEM_AINV_GMRES_BJACOBI_MF): state of the last ifem_ins_assemble
  ifem::DBuf<double> mf_ycell; // per-cell results of the matrix-free apply [n_cells][nu][dim] (two-stage scatter)
  ifem::DBuf<double> mf_eval;  // velocity part of the evaluation point, ghost-extended
  ifem_ins_params mf_params{};
  bool mf_valid = false;
  bool mf_noconv = false; // the assembled matrix has no convective terms (InsIMEX): the operator skips the second field group
  double mf_ms_total = 0;
  ifem::DBuf<float> Auu_f32;   // single-precision copy of Auu.val for the inner (preconditioner-only) solver
  bool auu_f32_valid = false, last_spmv_f32 = false;
  ifem::DBuf<double> diagMu;   // diag of mass (0,0), per velocity dof (owned)
  ifem::DBuf<double> dinvMu;   // 1/diagMu
  ifem::DBuf<double> bjac;     // inverse diagonal node blocks of A_uu [nUo][dim*dim]
  ifem::DBuf<float> bjac_f32;  // single-precision copy for the inner solver (built on first use after bjac_setup)
  bool bjac_f32_valid = false;
  // scatter maps: position of the column inside the row, 0xFFFF = row not owned here
  ifem::DBuf<uint16_t> posUU, posUP, posPU, posPP;
  ifem::DBuf<int32_t> uu_diag_pos; // position of the diagonal block in every owned A_uu row (setup.hip::ensure_auu_values)
  // 3D Q2/Q1 cell kernel (assemble3.hip): per-cell scatter records and headers (setup.hip::ensure_scat3), reference tables
  ifem::DBuf<uint16_t> scat3;
  ifem::DBuf<uint8_t> hdr3;
  ifem::DBuf<ifem::Tabs3> tabs3;
  bool hdr3_rows = false; // the headers carry the row alignments of the present A_uu row order
  // constraints (local dof numbering), sets 0 = zero, 1 = nonzero
  ifem::DBuf<uint8_t> is_c[2];
  ifem::DBuf<double> cval[2];
  bool has_c[2] = {false, false};
  // vectors
  ifem::DBuf<double> vec[IFEM_N_VECS];
  // Krylov workspace
  ifem::DBuf<double> krylovV, krylovZ, innerV, innerZ, work;
  ifem::DBuf<double> scal; // device scalars for reductions
  ifem::DBuf<double> partials; // per-block partial sums of the fused dot products [64][4096]
  double *h_scal = nullptr; // pinned host mirror
  ifem::Halo halo;
  bool assembled = false;
  bool profile = false; // HIP-event timing of every A_uu SpMV launch (bench only: adds a sync per launch)
  // timing
  ifem_timing timing{};
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int tight_first_misses = 0, tight_first_backoff = 0; // ifem_solver_opts::inner_rel_first: consecutive misses, qualifying solves left to skip
  double spmv_uu_ms_total = 0;
  ifem::KProf kprof; // per-kernel-family event log of a profiled step (used on the finest level of a chain only)
  // The A_uu V-cycle as a hipGraph (finest level of a single-rank chain with at most ifem_tuning::vcycle_graph_cells cells): a fixed
  // sequence of ~100-200 short launches whose arguments (level vectors, tables, Chebyshev coefficients) do not change between
  // applications -- captured once per state (`key`: pointers, bounds, parameters, constraint set, sweep counts) and replayed.  `armed`:
  // the state has been seen once and ran eagerly (lazy buffers exist); the next application with the same key is captured.
  struct VcGraph {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    std::vector<uint64_t> key;
    bool armed = false;
    uint64_t launches = 0, captures = 0;
    void destroy() {
      if (exec) (void)hipGraphExecDestroy(exec);
      if (graph) (void)hipGraphDestroy(graph);
      exec = nullptr; graph = nullptr;
    }
  } vc_graph, sm_graph, pa_graph; // (pa_graph: the inner iteration's B2pp_inverse T_pp of the SCnsIM preconditioner)  // (sm_graph: the same for the V-cycle on S_m inside CG(S_m))
  // section marks of the preconditioner applications of one solve (start, after CG(M_p), after CG(S_m) + B^T, end): recorded on the
  // stream, read once when the solve has finished (ifem_solve_stats::t_cg_mp_ms / t_cg_sm_ms / t_ainv_ms) -- no host wait per section
  int test_restart_fits = 0; // test aid: stands in for the free-memory bound of the restart lengthening on this rank
  int inner_restart_eff = 0; // restart length of the inner GMRES of IFEM_AINV_MG once an application stagnated across restarts (solver.hip)
  std::vector<hipEvent_t> pc_ev;
  size_t pc_used = 0;
};

namespace ifem {
inline KProf &kprof_root(ifem_ctx *c) {
  while (c->mg_fine) c = c->mg_fine;
  return c->kprof;
}
inline KScope::KScope(ifem_ctx *ctx, int cat, double bytes, double flops) {
  KProf &r = kprof_root(ctx);
  if (!r.on) return;
  k = &r;
  if (r.depth++ > 0) return;
  s = ctx->stream;
  while (r.ev.size() < r.used + 2) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { r.depth--; k = nullptr; return; }
    r.ev.push_back(e);
  }
  const uint32_t e0 = uint32_t(r.used);
  e1 = e0 + 1;
  r.used += 2;
  r.recs.push_back({cat, e0, e1, bytes, flops});
  (void)hipEventRecord(r.ev[e0], s);
}
inline KScope::~KScope() {
  if (!k) return;
  if (--k->depth > 0) return;
  (void)hipEventRecord(k->ev[e1], s);
}
} // namespace ifem
