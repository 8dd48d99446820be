/*
 * ifem_hip_testing.h -- TEST AIDS of libifem_hip.so.  Not part of the drop-in surface (include/ifem_hip.h): nothing here replaces
 * a member of the reference's FluidSolver hierarchy; the parity tests use these entry points to look at intermediate objects
 * of the SCnsIM block preconditioner (mpi_supg_solver.cpp:35-192) that the reference keeps private.
 */
#ifndef IFEM_HIP_TESTING_H
#define IFEM_HIP_TESTING_H
#include "ifem_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Test hook of the SCnsIM preconditioner (single-rank contexts, after ifem_scns_assemble): forms the explicit
 * T_pp = A_pp - A_pv P_vv^-1 A_vp and its ILU(0) (ifem_tuning::tpp_ilu_order) as ifem_scns_solve would.  Call with rowptr only
 * to size the arrays (n_p + 1 entries, nnz = rowptr[n_p]); with col / val it also returns the CSR of T_pp, and with x / y
 * (host, n_p entries) y = (LU)^-1 x.  *levels (may be NULL) receives the number of forward levels of the schedule. */
int ifem_tpp_ilu_probe(ifem_ctx *ctx, int64_t *rowptr, int32_t *col, double *val, const double *x, double *y, int32_t *levels);
/* Test aid: replace the values of the explicit T_pp (pattern and order as returned by ifem_tpp_ilu_probe) so that the next
 * factorisation sees them -- the breakdown path of the ILU(0) (zero / non-finite pivot -> IFEM_E_KRYLOV_NOCONV from the probe, Jacobi
 * in ifem_scns_solve) cannot be reached from an assembled fluid matrix at will. */
int ifem_tpp_override(ifem_ctx *ctx, const double *val); /* TEST AID ONLY: single rank, not part of the reference's interface */

/* The pieces of the reference-structure preconditioner (ifem_tuning::scns_pc = 2; single-rank contexts, after ifem_scns_assemble),
 * host vectors in the compact owned layout:
 *   which = 0: y = P_vv^-1 x, the ILU(0) of A_vv          (n = dim * n_unodes; mpi_supg_solver.cpp:49-51)
 *   which = 1: y = B2pp_inverse x, the ILU(0) of B2pp     (n = n_pnodes; :133)
 *   which = 2: y = B2pp x, B2pp = A_pp - A_pv rowsum(|A_vv|)^-1 A_vp   (:56-132)
 *   which = 3: y = T_pp x = A_pp x - A_pv P_vv^-1 A_vp x  (:19-32)
 * ifem_tuning::pvv_sweeps / b2pp_sweeps < 0 make the two ILU applications exact substitutions. */
int ifem_scns_pc_probe(ifem_ctx *ctx, int which, const double *x, double *y);

/* Stand-in for the free-memory bound of the self-lengthening inner restart (solver.hip::precond_vmult) on THIS rank: `columns` > 0 is
 * the number of basis column pairs this rank says it can afford, 0 restores the hipMemGetInfo estimate.  The decision itself is
 * collective (the smallest wish of all ranks wins): a test gives the ranks different bounds and checks that they agree. */
int ifem_test_restart_fits(ifem_ctx *ctx, int columns);

#ifdef __cplusplus
}
#endif
#endif
