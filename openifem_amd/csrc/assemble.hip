// assemble.hip -- host side of InsIM::assemble / InsIMEX::assemble (reference: source/mpi_insim.cpp:153-362,
// source/mpi_insimex.cpp:150-355): zeroing, argument block, kernel selection, epilogue.
//
// The cell kernels integrate the Newton-linearised INS weak form in component-block form:
//   Ke[(a,c),(b,d)] = sum_q JxW { d_cd [ mu gN_a.gN_b + rho N_a (u.gN_b) + rho/dt N_a N_b ]
//                                 + rho N_a N_b d_d u_c + gamma rho d_c N_a d_d N_b }
//   Ke[(a,c),p_b]   = -sum_q JxW d_c N_a  psi_b          (and its transpose)
// and scatter with AffineConstraints::distribute_local_to_global(..., true) semantics straight into the
// device block matrices:
//   assemble3.hip  3D Q2/Q1: contraction on the FP64 matrix cores, two wavefronts per cell
//   assemble2.hip  every other (dim, kv): quadrature-point-outer vector kernel, one wavefront per cell
#include <hip/hip_runtime.h>
#include "ctx.hpp"
#include "kernels.hpp"
#include "assemble_common.hpp"

namespace ifem {

static void assemble_epilogue(ifem_ctx *ctx, int use_nonzero);
static void ifem_ctx_unconstrained_geometry(ifem_ctx *ctx, const ifem_ins_params *p);
void launch_ins_assemble2_kernel(ifem_ctx *ctx, const AsmArgs &A);
bool launch_ins_assemble3_kernel(ifem_ctx *ctx, const AsmArgs &A);

// B / B^T of a constrained-dof set from the unconstrained blocks: distribute_local_to_global(..., true) drops the rows and
// columns of constrained dofs, i.e. plane c of the B^T row of node a, and entry c of every B block in the
// column of node a, when velocity dof (a, c) is constrained -- the kept entries are the unconstrained sums unchanged.
template <int DIM>
__global__ void k_mask_bt(int64_t n_rows, const int64_t *__restrict__ rp, const uint8_t *__restrict__ is_c,
                          const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 5; // 32 lanes per row
  const int lig = threadIdx.x & 31;
  if (row >= n_rows) return;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
#pragma unroll
  for (int c = 0; c < DIM; ++c) {
    const bool drop = is_c && is_c[row * DIM + c];
    for (int k = lig; k < len; k += 32) dst[rs * DIM + int64_t(c) * len + k] = drop ? 0.0 : src[rs * DIM + int64_t(c) * len + k];
  }
}
template <int DIM>
__global__ void k_mask_b(int64_t n_rows, const int64_t *__restrict__ rp, const int32_t *__restrict__ col,
                         const uint8_t *__restrict__ is_c, const double *__restrict__ src, double *__restrict__ dst) {
  const int64_t row = (int64_t(blockIdx.x) * blockDim.x + threadIdx.x) >> 5;
  const int lig = threadIdx.x & 31;
  if (row >= n_rows) return;
  const int64_t rs = rp[row];
  const int len = int(rp[row + 1] - rs);
  for (int k = lig; k < len; k += 32) {
    const int64_t nd = col[rs + k];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const bool drop = is_c && is_c[nd * DIM + c];
      dst[rs * DIM + int64_t(c) * len + k] = drop ? 0.0 : src[rs * DIM + int64_t(c) * len + k];
    }
  }
}
// assemblies with an unchanged (and never yet changed) constrained-dof set after which the unconstrained copies of B / B^T / S_m are given
// back: the second cached assembly.  A run whose set does change later (FSI: every time step) re-integrates them once and keeps them from then on.
constexpr int kGeoKeep = 2;
static void masked_geometry_blocks(ifem_ctx *ctx, int use_nonzero) {
  KScope ks(ctx, IFEM_KC_SCHUR_SETUP, 16.0 * double(ctx->B.val.n + ctx->Bt.val.n));
  const int w = use_nonzero ? 1 : 0;
  const uint8_t *flags = ctx->has_c[w] ? ctx->is_c[w].p : nullptr;
  hipStream_t s = ctx->stream;
  const int64_t nu = ctx->Bt.n_rows, np = ctx->B.n_rows;
  if (ctx->dim == 3) {
    if (nu) hipLaunchKernelGGL((k_mask_bt<3>), dim3(unsigned((nu * 32 + 255) / 256)), dim3(256), 0, s, nu, ctx->Bt.rowptr.p, flags, ctx->Bt0.p, ctx->Bt.val.p);
    if (np) hipLaunchKernelGGL((k_mask_b<3>), dim3(unsigned((np * 32 + 255) / 256)), dim3(256), 0, s, np, ctx->B.rowptr.p, ctx->B.col.p, flags, ctx->B0.p, ctx->B.val.p);
  } else {
    if (nu) hipLaunchKernelGGL((k_mask_bt<2>), dim3(unsigned((nu * 32 + 255) / 256)), dim3(256), 0, s, nu, ctx->Bt.rowptr.p, flags, ctx->Bt0.p, ctx->Bt.val.p);
    if (np) hipLaunchKernelGGL((k_mask_b<2>), dim3(unsigned((np * 32 + 255) / 256)), dim3(256), 0, s, np, ctx->B.rowptr.p, ctx->B.col.p, flags, ctx->B0.p, ctx->B.val.p);
  }
}

void launch_ins_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) { launch_ins_assemble_ex(ctx, p, use_nonzero, 0, 1); }
void launch_ins_assemble_geometry(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) { launch_ins_assemble_ex(ctx, p, use_nonzero, 0, 2); }

// imex = 1: InsIMEX::assemble (mpi_insimex.cpp:150-355): every field comes from the present solution, the matrix has no
// convective terms; assemble_system = 0 integrates the right-hand side only and leaves the matrices untouched;
// assemble_system = 2 (internal): B, B^T, M_p, diag(M_u) only -- a multigrid level of the pressure Schur complement
void launch_ins_assemble_ex(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero, int imex, int assemble_system) {
  hipStream_t s = ctx->stream;
  const int dim = ctx->dim;
  const bool geo_only = assemble_system >= 2; // 3: the same without constraints (pristine blocks)
  if (assemble_system == 2) {
    const int64_t key = ctx->flag_id[use_nonzero ? 1 : 0];
    // a multigrid level is asked once per preconditioner application: it counts ASSEMBLIES of the finest level (its version stamp)
    const ifem_ctx *f0 = ctx;
    while (f0->mg_fine) f0 = f0->mg_fine;
    const bool new_assembly = uint64_t(f0->asm_version) != ctx->geo_seen_asm;
    ctx->geo_seen_asm = uint64_t(f0->asm_version);
    if (ctx->geo_valid && ctx->geo_key == key && ctx->tune.geo_cache != 2) { // still the blocks of this constrained-dof set
      if (new_assembly && ++ctx->geo_unchanged == kGeoKeep && ctx->geo0_valid && ctx->geo_set_changes == 0) { // its unconstrained copies go the way of the finest level's (below)
        ctx->B0.release(); ctx->Bt0.release(); ctx->Sm0.release();
        ctx->geo0_valid = false; ctx->sm0_valid = false;
      }
      return;
    }
    if (ctx->geo_valid) ctx->geo_set_changes++;
    ctx->geo_unchanged = 0;
  }
  // ifem_tuning::stored_uu = 0: the velocity-velocity block is never stored.  The cell kernel integrates the right-hand side (and,
  // through the geometry path, B / B^T / M_p / diag(M_u)); A_uu is applied matrix-free in fp64 by the outer operator (the same
  // operator to 1e-13, test_matrix_free_uu_apply_equals_assembled_block) and its node-block diagonal comes from the cell integrals.
  // An assembly with inhomogeneous constraint values needs the element matrix columns (distribute_local_to_global moves K g into the
  // right-hand side): that one -- the first Newton iteration of a step with non-zero boundary values -- takes the stored path.
  const bool mf_only = assemble_system == 1 && ctx->tune.stored_uu == 0 && !(use_nonzero && ctx->inhom_any[1]);
  if (assemble_system == 1 && ctx->tune.stored_uu == 0 && ctx->hang.active)
    throw Error(IFEM_E_BADPARAM, "stored_uu = 0 with hanging-node constraints is not supported");
  if (assemble_system == 1 && !mf_only) ensure_auu_values(ctx);
  if (assemble_system == 1) ctx->uu_is_stored = !mf_only;
  if (!assemble_system && !ctx->assembled) throw Error(IFEM_E_BADPARAM, "rhs-only assembly before any matrix assembly");
  if (assemble_system == 1) { // state the matrix-free A_uu needs to reproduce this matrix (apply_mf.hip)
    const size_t nu = size_t(dim) * size_t(ctx->nUl);
    if (ctx->mf_eval.n != nu) ctx->mf_eval.alloc(nu);
    if (imex) IFEM_HIP_CHECK(hipMemsetAsync(ctx->mf_eval.p, 0, nu * sizeof(double), s)); // no convection in the IMEX matrix
    else IFEM_HIP_CHECK(hipMemcpyAsync(ctx->mf_eval.p, ctx->vec[IFEM_VEC_EVAL].p, nu * sizeof(double), hipMemcpyDeviceToDevice, s));
    ctx->mf_params = *p;
    ctx->mf_valid = true;
    ctx->mf_noconv = imex != 0;
    ctx->asm_version++;
  }
  // B, B^T, M_p and diag(M_u) depend on the mesh and on WHICH dofs are constrained, not on the solution, the parameters
  // or the inhomogeneities: an assembly whose constrained-dof set equals that of the previous one (zero_ and
  // nonzero_constraints of make_constraints list the same dofs) keeps them (bit-identical to re-integrating
  // them) and integrates A_uu and the right-hand side only.  ifem_tuning::geo_cache = 0 switches it off.
  const int64_t geo_key = ctx->flag_id[use_nonzero ? 1 : 0];
  bool skip_geo = ctx->tune.geo_cache == 1 && assemble_system && assemble_system != 3 && ctx->geo_valid && ctx->geo_key == geo_key;
  // the unconstrained copies of B / B^T / S_m (19 + 4 GB at 128^3) only serve a CHANGE of the constrained-dof set (FSI steps): a run
  // whose set has NEVER changed and has stood still for a few assemblies (pure-fluid runs) gives them back; a context that has seen
  // a change (an FSI run: a new set every time step, several Newton assemblies in between) keeps them -- releasing them there would
  // repeat a hipFree / hipMalloc / geometry launch every time step
  if (assemble_system == 1) {
    if (!skip_geo && ctx->geo_valid && ctx->geo_key != geo_key) ctx->geo_set_changes++;
    ctx->geo_unchanged = skip_geo ? ctx->geo_unchanged + 1 : 0;
    if (ctx->geo_unchanged == kGeoKeep && ctx->geo0_valid && ctx->geo_set_changes == 0) {
      ctx->B0.release(); ctx->Bt0.release(); ctx->Sm0.release();
      ctx->geo0_valid = false; ctx->sm0_valid = false;
    }
  }
  // A NEW constrained-dof set (every FSI step): the blocks are masked copies of the unconstrained ones, which are integrated
  // once per mesh (one geometry-only launch of the cell kernel without constraints); M_p and diag(M_u) do not depend on
  // the set at all.  Same values as re-integrating them under the new set (the kept entries are the same sums).
  if (assemble_system && assemble_system != 3 && !skip_geo && ctx->tune.geo_cache) {
    if (!ctx->geo0_valid) {
      ifem_ctx_unconstrained_geometry(ctx, p);
      ctx->geo0_valid = true;
    }
    masked_geometry_blocks(ctx, use_nonzero);
    ctx->geo_valid = true; ctx->geo_key = geo_key;
    skip_geo = true;
    if (geo_only) { // a multigrid level of S_m: nothing else to integrate
      dinv_setup(ctx);
      ctx->bbt_f32_valid = false;
      ctx->sm_valid = false; ctx->sm_key = geo_key;
      ctx->asm_constraint_set = use_nonzero ? 1 : 0;
      return;
    }
  }
  // system_matrix = 0; mass_matrix = 0; system_rhs = 0  (:163-165)
  {
  KScope ks_fill(ctx, IFEM_KC_ZERO_FILL, 8.0 * ((assemble_system && !geo_only && !mf_only ? double(ctx->Auu.val.n) : 0.0) + double(ctx->vec[IFEM_VEC_RHS].n) +
                                                (assemble_system && !skip_geo ? double(ctx->Bt.val.n + ctx->B.val.n + ctx->Mp.val.n + ctx->diagMu.n) : 0.0)));
  if (assemble_system) {
    // (a hand-written fill kernel with 16-byte non-temporal stores measures the same 15 ms for the 78 GB at 128^3)
    if (!geo_only && !mf_only) IFEM_HIP_CHECK(hipMemsetAsync(ctx->Auu.val.p, 0, ctx->Auu.val.n * sizeof(double), s));
    if (!skip_geo) {
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->Bt.val.p, 0, ctx->Bt.val.n * sizeof(double), s));
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->B.val.p, 0, ctx->B.val.n * sizeof(double), s));
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->Mp.val.p, 0, ctx->Mp.val.n * sizeof(double), s));
      ctx->mp_f32_valid = false;
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->diagMu.p, 0, ctx->diagMu.n * sizeof(double), s));
    }
  }
  if (ctx->want_shat && assemble_system) {
    if (ctx->Shat.n != (size_t)ctx->Auu.nnzb) ctx->Shat.alloc((size_t)ctx->Auu.nnzb);
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->Shat.p, 0, ctx->Shat.n * sizeof(double), s));
  }
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->vec[IFEM_VEC_RHS].p, 0, ctx->vec[IFEM_VEC_RHS].n * sizeof(double), s));
  }
  AsmArgs A{};
  A.n_cells = ctx->n_cells; A.nUo = ctx->nUo; A.nUl = ctx->nUl; A.nPo = ctx->nPo;
  A.fe = ctx->d_fe.p;
  A.vcoords = ctx->vcoords.p; A.cell_unodes = ctx->cell_unodes.p; A.cell_pnodes = ctx->cell_pnodes.p;
  A.cell_face_bid = ctx->cell_face_bid.p; A.indicator = ctx->indicator.p;
  A.posUU = ctx->posUU.p; A.posUP = ctx->posUP.p; A.posPU = ctx->posPU.p; A.posPP = ctx->posPP.p;
  A.rp_uu = ctx->Auu.rowptr.p; A.rp_bt = ctx->Bt.rowptr.p; A.rp_b = ctx->B.rowptr.p; A.rp_mp = ctx->Mp.rowptr.p;
  A.v_uu = ctx->Auu.val.p; A.v_bt = ctx->Bt.val.p; A.v_b = ctx->B.val.p; A.v_mp = ctx->Mp.val.p;
  A.diagMu = ctx->diagMu.p; A.rhs = ctx->vec[IFEM_VEC_RHS].p;
  A.v_s = ctx->want_shat ? ctx->Shat.p : nullptr;
  const int w = use_nonzero ? 1 : 0;
  const bool unconstrained = assemble_system == 3; // internal: the mesh-only blocks (ifem_ctx_unconstrained_geometry)
  A.is_c = (ctx->has_c[w] && !unconstrained) ? ctx->is_c[w].p : nullptr;
  A.cval = (ctx->has_c[w] && !unconstrained) ? ctx->cval[w].p : nullptr;
  A.use_inhom = (use_nonzero && ctx->has_c[1] && !unconstrained) ? 1 : 0;
  A.skip_geo = skip_geo ? 1 : 0;
  A.skip_uu = geo_only ? 1 : 0;
  A.debug_skip = ctx->tune.asm_skip;
  A.xcd_swizzle = ctx->tune.xcd_swizzle;
  A.eval = ctx->vec[imex ? IFEM_VEC_PRESENT : IFEM_VEC_EVAL].p; A.present = ctx->vec[IFEM_VEC_PRESENT].p;
  A.imex = imex; A.rhs_only = assemble_system && !mf_only ? 0 : 1;
  if (mf_only && !skip_geo) throw Error(IFEM_E_BADPARAM, "stored_uu = 0 needs ifem_tuning::geo_cache >= 1 (the geometry blocks come from their own launch)");
  A.fsi_acc = ctx->indicator.p ? ctx->vec[IFEM_VEC_FSI_ACC].p : nullptr;
  A.mu = p->viscosity; A.rho = p->rho; A.gamma = p->grad_div; A.inv_dt = 1.0 / p->dt;
  for (int i = 0; i < 3; ++i) A.g[i] = p->gravity[i];
  A.n_neumann = p->n_neumann;
  for (int i = 0; i < 8; ++i) { A.neumann_id[i] = p->neumann_id[i]; A.neumann_p[i] = p->neumann_p[i]; }
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
  {
    // algorithmic traffic / work of the cell kernel: every stored value of the blocks it integrates
    // written once, the right-hand side, per cell the mesh tables and the three nodal vectors it gathers; flops of the
    // component-block form: per (node pair, point) dim^2 (2 FMA) + dim (2 FMA) + 5 products (53 flop in 3D), per (velocity
    // node, pressure node, point) 2 (1 + dim) when B / B^T / M_p are integrated.  MFMA padding is not counted.
    const double nuu = A.skip_uu || A.rhs_only ? 0.0 : double(ctx->Auu.val.n);
    const double ngeo = A.skip_geo || A.rhs_only ? 0.0 : double(ctx->Bt.val.n + ctx->B.val.n + ctx->Mp.val.n + ctx->diagMu.n);
    const int nd = ctx->nu * dim + ctx->np, npc = 1 << dim;
    const double pair = 2.0 * (2 * dim * dim + 2 * dim) + 5.0; // 24 FMA + 5 products in 3D
    KScope ks_asm(ctx, IFEM_KC_ASSEMBLE, 8.0 * (nuu + ngeo + double(ctx->nUo) * dim + double(ctx->nPo)) + double(ctx->n_cells) * (npc * dim * 8.0 + (ctx->nu + ctx->np) * 4.0 + 3.0 * nd * 8.0),
                  double(ctx->n_cells) * ctx->nq * ((nuu > 0 ? double(ctx->nu) * ctx->nu * pair : 0.0) + (ngeo > 0 ? double(ctx->nu) * ctx->np * 2.0 * (1 + dim) : 0.0)));
  if (!launch_ins_assemble3_kernel(ctx, A)) // assemble3.hip: 3D Q2/Q1 on the FP64 matrix cores
    launch_ins_assemble2_kernel(ctx, A);    // assemble2.hip (quadrature-point-outer, register accumulators)
  }
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
  if (unconstrained) return; // the caller copies the blocks away
  if (assemble_system) { ctx->geo_valid = true; ctx->geo_key = geo_key; }
  if (geo_only) { // what the Schur complement of this level needs: 1/diag(M_u); S_m is stale if the blocks were re-integrated
    dinv_setup(ctx);
    ctx->bbt_f32_valid = false;
    ctx->sm_valid = false; ctx->sm_key = geo_key;
    ctx->asm_constraint_set = use_nonzero ? 1 : 0;
    return;
  }
  if (assemble_system) assemble_epilogue(ctx, use_nonzero);
  else {
    IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->timing.assemble_kernel_ms = ms;
  }
  hanging_condense_rhs(ctx, use_nonzero);
}

// B, B^T, M_p, diag(M_u) of the mesh alone: one geometry-only launch with no constraint set, B / B^T copied away
static void ifem_ctx_unconstrained_geometry(ifem_ctx *ctx, const ifem_ins_params *p) {
  launch_ins_assemble_ex(ctx, p, 0, 0, 3);
  hipStream_t s = ctx->stream;
  if (ctx->B0.n != ctx->B.val.n) ctx->B0.alloc(ctx->B.val.n);
  if (ctx->Bt0.n != ctx->Bt.val.n) ctx->Bt0.alloc(ctx->Bt.val.n);
  if (ctx->B.val.n) IFEM_HIP_CHECK(hipMemcpyAsync(ctx->B0.p, ctx->B.val.p, ctx->B.val.n * sizeof(double), hipMemcpyDeviceToDevice, s));
  if (ctx->Bt.val.n) IFEM_HIP_CHECK(hipMemcpyAsync(ctx->Bt0.p, ctx->Bt.val.p, ctx->Bt.val.n * sizeof(double), hipMemcpyDeviceToDevice, s));
  ctx->mp_f32_valid = false;
}

static void assemble_epilogue(ifem_ctx *ctx, int use_nonzero) {
  dinv_setup(ctx);
  if (!ctx->uu_is_stored) { ctx->asm_constraint_set = use_nonzero ? 1 : 0; uu_block_diag_mf(ctx); }
  else bjac_setup(ctx);
  IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
  float ms = 0;
  IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->timing.assemble_kernel_ms = ms;
  ctx->assembled = true;
  ctx->has_app = false;
  ctx->auu_f32_valid = false;
  ctx->bbt_f32_valid = false;
  // S_m = B diag(M_u)^-1 B^T depends only on the mesh and on WHICH dofs are constrained (not on the solution):
  // keep it across assemblies until the constraint set changes (the reference rebuilds it every solve(); same values)
  {
    const int64_t key = ctx->flag_id[use_nonzero ? 1 : 0];
    if (key != ctx->sm_key || ctx->tune.geo_cache != 1) { ctx->sm_valid = false; ctx->sm_key = key; }
  }
  ctx->shat_valid = ctx->want_shat;
  ctx->shat_aux_valid = false;
  ctx->asm_constraint_set = use_nonzero ? 1 : 0;
}

} // namespace ifem
