// kernels.hpp -- host-callable launchers implemented in the .hip translation units.
#pragma once
#include "ctx.hpp"

namespace ifem {

// setup.hip
void build_pattern(ifem_ctx *ctx, PlanarCsr &M, int bs, int64_t n_rows_owned, int R, const int32_t *d_rows, int C,
                   const int32_t *d_cols, DBuf<uint16_t> &pos);

void ensure_auu_values(ifem_ctx *ctx);
bool ensure_scat3(ifem_ctx *ctx, bool with_rows); // records of the 3D Q2/Q1 cell kernel (assemble3.hip); false: not applicable
void build_schur_pattern(ifem_ctx *ctx);
void build_schur_pattern_owned(ifem_ctx *ctx); // several ranks: owned x owned block, into ctx->TppPat
void build_incidence(ifem_ctx *ctx);
int64_t compact_flagged_rows(ifem_ctx *ctx, const int64_t *flag, int64_t n, DBuf<int32_t> &rows); // ascending list of the flagged rows
void build_mf_cell_split(ifem_ctx *ctx); // several ranks: interior-first copy of the cell tables for the matrix-free apply

// assemble.hip
void launch_ins_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero);
void launch_ins_assemble_ex(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero, int imex, int assemble_system);
// B, B^T, M_p, diag(M_u) only (multigrid levels of the pressure Schur complement): no A_uu, no right-hand side state
void launch_ins_assemble_geometry(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero);

// fsi.hip -- fluid-side inputs of MPI::FSI (source/mpi_fsi.cpp:96-127,142-223,291-663)
void fsi_set_solid(ifem_ctx *ctx, const ifem_fsi_solid *s);
void fsi_update_indicator(ifem_ctx *ctx, int32_t *host_out, int64_t *n_artificial);
void fsi_find_fluid_bc(ifem_ctx *ctx, double dt, int use_dirichlet_bc, const int32_t *cell_order, ifem_fsi_stats *stats);
void fsi_fluid_at_points(ifem_ctx *ctx, int32_t n, const double *points, double *values, double *stress, int32_t *cell);
// api.hip: do the flag arrays of pair k differ (compared on the device); identity of the constrained-dof set `which` after
// its flags changed
void flags_differ(ifem_ctx *ctx, int npairs, const DBuf<uint8_t> *const *a, const DBuf<uint8_t> *const *b, double *out);
void constraint_set_identity(ifem_ctx *ctx, int which, bool differs_self, bool differs_other);

// assemble_scns.hip
void launch_scns_assemble(ifem_ctx *ctx, const ifem_scns_params *p, int use_nonzero);
void launch_update_stress(ifem_ctx *ctx, double mu);

// linalg.hip -- all on ctx->stream.  Block vectors are [u (dim*nUl) | p (nPl)]; "owned" ranges only.
struct VecLayout {
  int64_t n_u_owned; // dim*nUo
  int64_t p_off;     // dim*nUl
  int64_t n_p_owned; // nPo
};
inline VecLayout layout_of(const ifem_ctx *c) { return {c->dim * c->nUo, c->dim * c->nUl, c->nPo}; }

// y_u = A_uu x_u (+ B^T x_p when xp != nullptr)
void spmv_uu(ifem_ctx *ctx, const double *xu, const double *xp, double *yu, bool use_f32, int part = 0);
// apply_mf.hip: y_u = A_uu x_u without the stored matrix (sum-factorised cell kernel on the state of the last assemble)
// optional epilogue of the matrix-free product t = A_uu x (owned rows), see apply_mf.hip::k_mf_gather: t is consumed instead of
// stored.  mode 1: xs += x, r -= t; mode 2 additionally d = a x + b B r (B = inverse node block, single-precision copy) -- the
// Chebyshev step of the multigrid smoother, d being the owned part of x itself; mode 3: mode 1, then d = b B r into another
// vector d -- the first direction of the smoothing sweep that follows the coarse correction
template <typename V>
struct MfFuseT {
  int mode = 0;
  double a = 0, b = 0;
  V *xs = nullptr, *r = nullptr, *d = nullptr;
};
using MfFuse = MfFuseT<double>;
// part (several ranks, build_mf_cell_split): 0 every cell, then the node gather; 1 only the cells whose nodes are all owned
// (no ghost entry of xu is read: the halo may still be in flight); 2 the remaining cells, then the gather
void apply_uu_mf(ifem_ctx *ctx, const double *xu, double *yu, bool single = false, const MfFuse *fuse = nullptr, int part = 0);
// the same on SINGLE-PRECISION vectors, fused form only (the level vectors of the A_uu V-cycle, solver.hip): single-precision
// cell arithmetic, x and the vectors of `fuse` are float
void apply_uu_mf_f32v(ifem_ctx *ctx, const float *xu, const MfFuseT<float> *fuse, int part = 0);
// scalar velocity operator S^ (IFEM_AINV_SCALAR_GMRES): auxiliary data, SpMV on all components, Jacobi
void shat_refresh(ifem_ctx *ctx, bool f32);
void spmv_shat(ifem_ctx *ctx, const double *xu, double *yu, bool f32);
void shat_jacobi(ifem_ctx *ctx, const double *x, double *y);
// y_p = B x_u
// `part` of the row-parallel products below: 0 all rows, 1 the rows that read owned columns only, 2 the others (several
// ranks: 1 runs while the halo of x is in flight, 2 after halo_wait; see PlanarCsr::split_rows)
void build_row_split(ifem_ctx *ctx, PlanarCsr &M, int64_t n_owned_cols, const PlanarCsr *M2 = nullptr, int64_t n_owned_cols2 = 0);
void spmv_b(ifem_ctx *ctx, const double *xu, double *yp, int part = 0);
// y_u = B^T x_p
void spmv_bt(ifem_ctx *ctx, const double *xp, double *yu);
void spmv_b_f32(ifem_ctx *ctx, const double *xu, double *yp);  // same with single-precision copies of the values
void spmv_bt_f32(ifem_ctx *ctx, const double *xp, double *yu); // (matrix-free S_m inside the preconditioner only)
// y_p = A_pp x_p (SCnsIM pressure block on the M_p pattern); 1/diag(A_pp) for its Jacobi preconditioner
void spmv_app(ifem_ctx *ctx, const double *xp, double *yp);
void app_diag_setup(ifem_ctx *ctx);
void sm_diag_from_blocks(ifem_ctx *ctx, const double *dinv_mu_ext, double *d); // diag(B diag(M_u)^-1 B^T), owned pressure rows
void scalar_diag(ifem_ctx *ctx, const PlanarCsr &M, const double *val, double *d);
// y_p = M_p x_p
void spmv_mp(ifem_ctx *ctx, const double *xp, double *yp, int part = 0, bool use_f32 = false);
// explicit S_m: numeric product B diag(1/diag M_u) B^T into ctx->Sm (pattern must exist), and y_p = S_m x_p
void schur_numeric(ifem_ctx *ctx);
void spmv_sm(ifem_ctx *ctx, const double *xp, double *yp, bool use_f32, int part = 0);
// y_u = d .* x_u (diagonal scaling with 1/diag(M_u))
void vec_mul(ifem_ctx *ctx, int64_t n, const double *d, const double *x, double *y);
void vec_div(ifem_ctx *ctx, int64_t n, const double *d, const double *x, double *y); // y = x ./ d (d == 0 -> 1)
// y = bjac * x  (node-block Jacobi)
void bjac_apply(ifem_ctx *ctx, const double *x, double *y);
const float *bjac_f32_ptr(ifem_ctx *ctx); // single-precision copy of the inverse node blocks (refreshed on demand)
void bjac_setup(ifem_ctx *ctx);
void dinv_setup(ifem_ctx *ctx);

// generic vector kernels over one contiguous range
void v_axpy(ifem_ctx *ctx, int64_t n, double a, const double *x, double *y);            // y += a x
void v_axpby(ifem_ctx *ctx, int64_t n, double a, const double *x, double b, double *y); // y = a x + b y
void v_scale(ifem_ctx *ctx, int64_t n, double a, double *x);
void v_copy(ifem_ctx *ctx, int64_t n, const double *x, double *y);
void v_scale_to(ifem_ctx *ctx, int64_t n, double a, const double *x, double *y); // y = a x
void v_scale_to2(ifem_ctx *ctx, int64_t n, double a, const double *x, double *y, double *z); // y = z = a x
void v_zero(ifem_ctx *ctx, int64_t n, double *x);
double v_dot(ifem_ctx *ctx, int64_t n, const double *x, const double *y); // local (no all-reduce), syncs
// multi-dot: out[i] = <V_i, w> for i < k (V column-major with leading dimension ld), one pass, syncs
void v_mdot(ifem_ctx *ctx, int64_t n, int k, const double *V, int64_t ld, const double *w, double *out_host, bool all_ranks = false);
// w -= sum_i h[i] V_i
void v_maxpy(ifem_ctx *ctx, int64_t n, int k, const double *V, int64_t ld, const double *h_host, double *w, double *norm2_out = nullptr,
             bool all_ranks = false);
// single-precision Krylov basis (inner solver): V float, w / coefficients / accumulation double
void v_mdot_f32(ifem_ctx *ctx, int64_t n, int k, const float *V, int64_t ld, const double *w, double *out_host);
void v_maxpy_f32(ifem_ctx *ctx, int64_t n, int k, const float *V, int64_t ld, const double *h_host, double *w, double *norm2_out);
void v_scale_store_f32(ifem_ctx *ctx, int64_t n, double a, const double *w, float *v);
void bjac_apply_f32(ifem_ctx *ctx, const float *x, double *y);
// CG with device-resident recurrence scalars (single rank): init, alpha = rz / <p,q>, update of x, r, z, p; cgd_rr syncs
void cgd_init(ifem_ctx *ctx, int64_t n, const double *b, const double *diag, double *x, double *r, double *z, double *p);
void cgd_alpha(ifem_ctx *ctx, int64_t n, const double *p, const double *q);
void cgd_update(ifem_ctx *ctx, int64_t n, const double *diag, double *p, const double *q, double *x, double *r, double *z);
double cgd_rr(ifem_ctx *ctx);
// the same recurrence with one fused reduction per iteration (Chronopoulos / Gear): u = D^-1 r (u == r without diag), w = A u, s = A p
void cg1_init(ifem_ctx *ctx, int64_t n, const double *b, const double *diag, double *x, double *r, double *u, double *p, double *s);
void cg1_dots(ifem_ctx *ctx, int64_t n, const double *r, const double *u, const double *w, bool first);
void cg1_update(ifem_ctx *ctx, int64_t n, const double *diag, double *u, const double *w, double *p, double *s, double *x, double *r);
void v_minmax(ifem_ctx *ctx, int64_t n, const double *x, double *mn, double *mx);
// x[dof] = value for constrained dofs (AffineConstraints::distribute, Dirichlet lines)
void apply_constraints(ifem_ctx *ctx, int which, double *x);
void spmv_planar_scalar(ifem_ctx *ctx, const PlanarCsr &M, const double *val, const double *xp, double *yp);
// explicit pressure Schur complement of the SUPG block preconditioner (tpp.hip)
void tpp_numeric(ifem_ctx *ctx);
void spmv_tpp(ifem_ctx *ctx, const double *xp, double *yp);
bool tpp_ilu_factor(ifem_ctx *ctx); // false: zero / tiny / non-finite pivot (ctx->tpp_ilu.broken): do not apply the factors
void tpp_ilu_apply(ifem_ctx *ctx, const double *x, double *y);
int tpp_ilu_levels(const ifem_ctx *ctx);
void schur_pp_numeric(ifem_ctx *ctx, const double *binv, double *out); // out = A_pp - A_pv blockdiag(binv) A_vp on the pattern of T_pp
const PlanarCsr &tpp_pattern(ifem_ctx *ctx);                          // that pattern (built on first use)
// ILU(0) with dim x dim blocks, natural order, Jacobi-sweep application (bilu.hip)
void bilu_analyse(ifem_ctx *ctx, BIlu &I, int dim, int64_t n, const int64_t *rp_dev, const int32_t *col_dev);
bool bilu_factor(ifem_ctx *ctx, BIlu &I, const double *src);
void bilu_apply(ifem_ctx *ctx, BIlu &I, int sweeps, const double *x, double *y);
void rowsum_abs_inv(ifem_ctx *ctx, double *out);
void scns_refpc_setup(ifem_ctx *ctx, int verbose, bool *pvv_ok, bool *b2_ok);
void scns_pc_probe(ifem_ctx *ctx, int which, const double *x, double *y);
// hanging-node lines (hanging.hip): C x on a copy of x, C^T and the hanging rows on y, distribute, set-up
void hanging_set(ifem_ctx *ctx, int32_t n, const int32_t *dof, const int32_t *ptr, const int32_t *master, const double *weight);
const double *hanging_input(ifem_ctx *ctx, const double *x); // ghost-extended [u_l | p_l] copy of x with C applied
void hanging_output(ifem_ctx *ctx, const double *x, bool x_extended, double *y);
void hanging_distribute(ifem_ctx *ctx, double *x);
void hanging_refresh_diag(ifem_ctx *ctx);
bool hanging_offset(ifem_ctx *ctx, int use_nonzero);
void hanging_condense_rhs(ifem_ctx *ctx, int use_nonzero); // solver.hip: b = C^T (b^ - A^ c0), hanging rows d c0

// mg.hip: transfers and smoother updates of the multigrid inside the preconditioner
void mg_csr_apply(ifem_ctx *ctx, const MgCsr &M, const double *x, double *y, bool add);
void cheb_init(ifem_ctx *ctx, int64_t n, double c0, const double *dinv, const double *r, double *d);
void cheb_step(ifem_ctx *ctx, int64_t n, double a, double b, const double *dinv, const double *t, double *x, double *r, double *d);
void vec_recip(ifem_ctx *ctx, int64_t n, double *d);
void vec_rough(ifem_ctx *ctx, int64_t n, int64_t offset, double *x);
void uu_block_diag_mf(ifem_ctx *ctx); // ctx->bjac := inverse node blocks of the matrix-free A_uu (coarse multigrid levels)
void mg_csr_mask(ifem_ctx *ctx, const MgCsr &M, const uint8_t *flag_in, const uint8_t *flag_out, DBuf<uint8_t> &mask);
void mg_csr_apply_nodes(ifem_ctx *ctx, const MgCsr &M, const double *x, const DBuf<uint8_t> &mask, double *y);
void mg_inject_nodes(ifem_ctx *ctx, int64_t n_nodes, const int32_t *inj, const double *fine, double *coarse);
void cheb_init_block(ifem_ctx *ctx, double c0, const double *r, double *d);
void cheb_init_block_f32(ifem_ctx *ctx, double c0, const float *r, float *d);
void mg_csr_apply_nodes_f32(ifem_ctx *ctx, const MgCsr &M, const float *x, const DBuf<uint8_t> &mask, float *y);
void v_cvt_d2f(ifem_ctx *ctx, int64_t n, const double *x, float *y);
void v_cvt_f2d(ifem_ctx *ctx, int64_t n, const float *x, double *y);
void v_axpy_f32v(ifem_ctx *ctx, int64_t n, float a, const float *x, float *y);

// all-reduce helpers (identity for a single rank)
void allreduce_sum(ifem_ctx *ctx, double *host_vals, int n);
void allreduce_max(ifem_ctx *ctx, double *host_vals, int n);
void allreduce_sum_dev(ifem_ctx *ctx, double *dev_vals, int n); // device scalars, in place, stream-ordered
// sum of a device VECTOR over the ranks, in place, stream-ordered (the hand-over to a replicated coarse level); scratch: n entries
void allreduce_sum_vec(ifem_ctx *ctx, double *dev, int64_t n, double *scratch);
void allreduce_sum_vec_f32(ifem_ctx *ctx, float *dev, int64_t n, float *scratch);
int comm_unique_id(uint8_t out[128]);
int comm_selftest(int device);
void comm_stats(ifem_ctx *ctx, ifem_comm_stats *out, bool reset, bool single_level = false);
void *local_world_create(int nranks);
void local_world_destroy(void *w);
// ghost refresh of a ghost-extended velocity buffer [dim*nUl] / pressure buffer [nPl] (RCCL send/recv over xGMI)
void halo_exchange(ifem_ctx *ctx, double *xu_ext);
void halo_exchange_p(ifem_ctx *ctx, double *xp_ext);
// transpose: ghost entries are sent back to their owners and added there (C^T of hanging lines across ranks)
void halo_reverse_add(ifem_ctx *ctx, double *xu_ext);
// single-precision velocity vectors (level vectors of the A_uu V-cycle): same plans, half the bytes
void halo_exchange_f32(ifem_ctx *ctx, float *xu_ext);
void halo_reverse_add_f32(ifem_ctx *ctx, float *xu_ext);
void halo_start_f32(ifem_ctx *ctx, float *xu_ext);
void halo_reverse_add_p(ifem_ctx *ctx, double *xp_ext);
// overlapped form: halo_start (pack + transfers on the halo stream), work that reads no ghost entry, halo_wait
bool halo_overlap_ok(const ifem_ctx *ctx);
void halo_start(ifem_ctx *ctx, double *x_ext, int which); // which: 0 velocity, 1 pressure, 2 the 2-deep halo of S_m
void halo_wait(ifem_ctx *ctx);
void halo_exchange_s(ifem_ctx *ctx, double *xs_ext); // [n_s_cols]: owned pressure nodes, then the 2-deep far nodes
void build_schur_pattern_box(ifem_ctx *ctx);          // distributed explicit S_m on a structured pressure lattice
void schur_probe_fill(ifem_ctx *ctx, int color, const double *y); // S_m[i, j(color)] = y_i
void schur_probe_vector(ifem_ctx *ctx, int color, double *x);     // x_i = [color(i) == color]
void comm_init(ifem_ctx *ctx, const ifem_partition *part);
void comm_destroy(ifem_ctx *ctx);

// solver.hip
int ins_solve(ifem_ctx *ctx, const ifem_ins_params *P, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats);
void ins_precond_vmult(ifem_ctx *ctx, const ifem_ins_params *P, const ifem_solver_opts *o, const double *src, double *dst);
void ins_system_vmult(ifem_ctx *ctx, const double *src, double *dst);
int scns_solve(ifem_ctx *ctx, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats);

// owned-range dot over a block vector (u range + p range), all-reduced
double bv_dot(ifem_ctx *ctx, const double *x, const double *y);
inline int64_t bv_len(const ifem_ctx *c) { return c->n_local; }

} // namespace ifem
