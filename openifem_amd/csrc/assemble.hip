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
void launch_ins_assemble2_kernel(ifem_ctx *ctx, const AsmArgs &A);
bool launch_ins_assemble3_kernel(ifem_ctx *ctx, const AsmArgs &A);

void launch_ins_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) { launch_ins_assemble_ex(ctx, p, use_nonzero, 0, 1); }
void launch_ins_assemble_geometry(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) { launch_ins_assemble_ex(ctx, p, use_nonzero, 0, 2); }

// imex = 1: InsIMEX::assemble (mpi_insimex.cpp:150-355): every field comes from the present solution, the matrix has no
// convective terms; assemble_system = 0 integrates the right-hand side only and leaves the matrices untouched;
// assemble_system = 2 (internal): B, B^T, M_p, diag(M_u) only -- a multigrid level of the pressure Schur complement
void launch_ins_assemble_ex(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero, int imex, int assemble_system) {
  hipStream_t s = ctx->stream;
  const int dim = ctx->dim;
  const bool geo_only = assemble_system == 2;
  if (geo_only) {
    const int64_t key = ctx->flag_id[use_nonzero ? 1 : 0];
    if (ctx->geo_valid && ctx->geo_key == key) return; // still the blocks of this constrained-dof set
  } else if (assemble_system)
    ensure_auu_values(ctx);
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
  const bool skip_geo = ctx->tune.geo_cache && assemble_system && ctx->geo_valid && ctx->geo_key == geo_key;
  // system_matrix = 0; mass_matrix = 0; system_rhs = 0  (:163-165)
  if (assemble_system) {
    // (a hand-written fill kernel with 16-byte non-temporal stores measures the same 15 ms for the 78 GB at 128^3)
    if (!geo_only) IFEM_HIP_CHECK(hipMemsetAsync(ctx->Auu.val.p, 0, ctx->Auu.val.n * sizeof(double), s));
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
  A.is_c = ctx->has_c[w] ? ctx->is_c[w].p : nullptr;
  A.cval = ctx->has_c[w] ? ctx->cval[w].p : nullptr;
  A.use_inhom = (use_nonzero && ctx->has_c[1]) ? 1 : 0;
  A.skip_geo = skip_geo ? 1 : 0;
  A.skip_uu = geo_only ? 1 : 0;
  A.debug_skip = ctx->tune.asm_skip;
  A.xcd_swizzle = ctx->tune.xcd_swizzle;
  A.eval = ctx->vec[imex ? IFEM_VEC_PRESENT : IFEM_VEC_EVAL].p; A.present = ctx->vec[IFEM_VEC_PRESENT].p;
  A.imex = imex; A.rhs_only = assemble_system ? 0 : 1;
  A.fsi_acc = ctx->indicator.p ? ctx->vec[IFEM_VEC_FSI_ACC].p : nullptr;
  A.mu = p->viscosity; A.rho = p->rho; A.gamma = p->grad_div; A.inv_dt = 1.0 / p->dt;
  for (int i = 0; i < 3; ++i) A.g[i] = p->gravity[i];
  A.n_neumann = p->n_neumann;
  for (int i = 0; i < 8; ++i) { A.neumann_id[i] = p->neumann_id[i]; A.neumann_p[i] = p->neumann_p[i]; }
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
  if (!launch_ins_assemble3_kernel(ctx, A)) // assemble3.hip: 3D Q2/Q1 on the FP64 matrix cores
    launch_ins_assemble2_kernel(ctx, A);    // assemble2.hip (quadrature-point-outer, register accumulators)
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
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

static void assemble_epilogue(ifem_ctx *ctx, int use_nonzero) {
  dinv_setup(ctx);
  bjac_setup(ctx);
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
    if (key != ctx->sm_key || !ctx->tune.geo_cache) { ctx->sm_valid = false; ctx->sm_key = key; }
  }
  ctx->shat_valid = ctx->want_shat;
  ctx->shat_aux_valid = false;
  ctx->asm_constraint_set = use_nonzero ? 1 : 0;
}

} // namespace ifem
