/*
 * oracle/oracle.h -- CPU restatement of the OpenIFEM incompressible Navier-Stokes fluid step.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load liboracle.so.  The product path (openifem_amd/) never
 * includes, links or calls anything in this directory.
 *
 * What it restates (reference = /root/reference, OpenIFEM @ 2025-07-25):
 *   orc_ins_assemble      source/mpi_insim.cpp:153-362   (InsIM::assemble: cell loop, Neumann faces,
 *                                                          distribute_local_to_global)
 *   orc_schur_setup       source/mpi_insim.cpp:13-50     (BlockSchurPreconditioner ctor: S_m = B diag(Mu)^-1 B^T)
 *   orc_precond_vmult     source/mpi_insim.cpp:57-128    (BlockSchurPreconditioner::vmult)
 *   orc_ins_solve         source/mpi_insim.cpp:365-395   (InsIM::solve: FGMRES(30) + constraints.distribute)
 *   orc_ins_run_one_step  source/mpi_insim.cpp:398-490   (Newton loop)
 *   orc_update_stress     source/mpi_fluid_solver.cpp:716-811
 *
 * Parity pinning: the reference cannot be built in this image (deal.II/PETSc/MUMPS absent), and its
 * tests dump no element matrices, so the oracle is pinned against the reference tests' known answers
 * (tests/fluid_pressure_driven: vmax = 2.5e-2; tests/fluid_gravity: pmax-pmin = 20;
 * tests/fluid_pipe_mpi: vmax = 1.5) -- see tests/test_oracle_kat.py.  Element matrices and Krylov
 * iterates are "parity unpinned" (SURVEY 8c): no reference artefact exists for them.
 *
 * Third-party arithmetic restated here because it is not under /root/reference (deal.II >= 9.3,
 * unpinned): FE_Q Lagrange shapes on equidistant nodes, QGauss, MappingQ1, AffineConstraints::
 * distribute_local_to_global, SolverFGMRES/SolverCG.  PETSc KSPCG -> plain CG; MUMPS -> caller-supplied
 * exact solve callback (tests pass scipy splu) or the built-in iterative inner solver.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Mesh + DoF tables.  Global DoF numbering is the reference's block layout [velocity | pressure]
 * (mpi_fluid_solver.cpp:125-128) with velocity components interleaved per node:
 *   u-dof = dim*unode + c,   p-dof = dim*n_unodes + pnode.
 * Local node order is tensor-lexicographic (x fastest) for both Q_kv and Q1. */
typedef struct {
  int32_t dim;           /* 2 or 3 */
  int32_t kv;            /* velocity degree: 1 or 2 (pressure degree is 1) */
  int32_t n_cells, n_unodes, n_pnodes;
  const double  *vcoords;       /* [n_cells][2^dim][dim] vertex coords (lexicographic vertex order) */
  const int32_t *cell_unodes;   /* [n_cells][(kv+1)^dim] */
  const int32_t *cell_pnodes;   /* [n_cells][2^dim] */
  const int32_t *cell_face_bid; /* [n_cells][2*dim] boundary id or -1; faces x-,x+,y-,y+,z-,z+ */
  const int32_t *indicator;     /* [n_cells] or NULL (0 real fluid, 1 artificial) */
} orc_mesh;

typedef struct {
  double mu, rho, gamma, dt;
  double g[3];
  int32_t n_neumann;
  int32_t neumann_id[8];
  double  neumann_p[8];
} orc_params;

/* A^-1 u-block solve callback (stands in for MUMPS, mpi_insim.cpp:124-127).
 * Called with the CSR of A_uu after every assemble when `refresh` != 0, then y = A_uu^-1 x. */
typedef void (*orc_ainv_fn)(void *user, int32_t refresh, int32_t n, const int64_t *rowptr,
                            const int32_t *col, const double *val, const double *x, double *y);

typedef struct {
  int32_t fgmres_restart;     /* 30 (deal.II default) */
  int32_t fgmres_maxit;       /* n_dofs in the reference */
  double  fgmres_rel, fgmres_abs; /* 1e-4, 1e-12 (mpi_insim.cpp:379-380) */
  /* built-in inner solver for A^-1 when no callback is given: GMRES(m) + node-block Jacobi */
  int32_t inner_restart, inner_maxit;
  double  inner_rel;
  int32_t n_threads;          /* OpenMP threads (0 = default) */
} orc_opts;

typedef struct orc_system orc_system;

orc_system *orc_create(const orc_mesh *m);
void orc_destroy(orc_system *s);
int32_t orc_n_dofs(const orc_system *s);
int32_t orc_n_u(const orc_system *s);
/* which: 0 = zero_constraints, 1 = nonzero_constraints (Dirichlet lines: dof, inhomogeneity) */
void orc_set_constraints(orc_system *s, int32_t which, int32_t n, const int32_t *dof, const double *val);
void orc_default_opts(orc_opts *o);

/* CSR of the full block system (all couplings, mpi_fluid_solver.cpp:311-322) */
const int64_t *orc_rowptr(const orc_system *s);
const int32_t *orc_col(const orc_system *s);
double *orc_A(orc_system *s);      /* system_matrix values */
double *orc_M(orc_system *s);      /* mass_matrix values (same pattern) */
double *orc_rhs(orc_system *s);

void orc_ins_assemble(orc_system *s, const orc_params *p, int32_t use_nonzero, const double *eval,
                      const double *present, const double *fsi_acc);
/* the same matrix and right-hand side assembled by subdomains without atomics (owner computes row; cell_part[cell] in
 * [0, n_parts) names the subdomain of a cell): the CPU baseline leg */
void orc_ins_assemble_subdomains(orc_system *s, const orc_params *p, int32_t use_nonzero, const double *eval,
                                 const double *present, const double *fsi_acc, const int32_t *cell_part, int32_t n_parts,
                                 int32_t n_threads /* 0: OpenMP default */);
/* the same assembly through constraints that also hold hanging-node lines x[dof[l]] = sum_k weight[k] x[master[k]],
 * k in [ptr[l], ptr[l+1]); dense n x n output (row-major) for small meshes; Dirichlet lines from orc_set_constraints */
void orc_ins_assemble_affine_dense(orc_system *s, const orc_params *p, int32_t use_nonzero, const double *eval,
                                   const double *present, const double *fsi_acc, int32_t n_lines, const int32_t *dof,
                                   const int32_t *ptr, const int32_t *master, const double *weight, double *A, double *rhs);
/* single-cell dense Ke/Me/fe (ndof x ndof row-major, local dof = [a*dim+c | p]) before constraints */
void orc_ins_cell(const orc_mesh *m, const orc_params *p, int32_t cell, const double *eval,
                  const double *present, const double *fsi_acc, double *Ke, double *Me, double *fe);

int32_t orc_ins_solve(orc_system *s, const orc_params *p, int32_t use_nonzero, const orc_opts *o,
                      orc_ainv_fn ainv, void *user, double *newton_update, int32_t *iters, double *res);
/* returns number of Newton iterations, <0 on failure (-1 too many Newton its, -2 Krylov no convergence) */
int32_t orc_ins_run_one_step(orc_system *s, const orc_params *p, int32_t apply_nonzero, double newton_tol,
                             int32_t newton_maxit, const orc_opts *o, orc_ainv_fn ainv, void *user,
                             double *present, const double *fsi_acc, double *log /* [maxit][4] or NULL */);

/* y = A x with the assembled system matrix */
void orc_spmv(const orc_system *s, const double *x, double *y);
/* S_m explicit (mpi_insim.cpp:44-49): returns CSR of mass_schur(1,1) built by the last solve (may be NULL) */
void orc_schur_csr(const orc_system *s, const int64_t **rowptr, const int32_t **col, const double **val);
/* z = P^-1 v (block Schur preconditioner, mpi_insim.cpp:57-128) on the last assembled system */
void orc_precond_vmult(orc_system *s, const orc_params *p, const orc_opts *o, orc_ainv_fn ainv, void *user,
                       const double *v, double *z);

/* ---- Fluid::MPI::InsIMEX (source/mpi_insimex.cpp:150-446): symmetric solution-independent matrix, explicit convection.
 * assemble_system = 0 re-assembles the rhs only (distribute_local_to_global of the vector: constrained rows dropped).
 * Pinned by tests/fluid_cylinder_mpi_insimex (vmax 0.374062, pmax 46.5308 after one step). */
void orc_imex_assemble(orc_system *s, const orc_params *p, int32_t use_nonzero, int32_t assemble_system,
                       const double *present, const double *fsi_acc);
/* InsIMEX::run_one_step (:396-446): FGMRES to min(1e-9, 1e-8 ||rhs||), present += solution_time_increment */
int32_t orc_imex_run_one_step(orc_system *s, const orc_params *p, int32_t apply_nonzero, int32_t assemble_system,
                              const orc_opts *o, orc_ainv_fn ainv, void *user, double *present, const double *fsi_acc,
                              int32_t *iters, double *res);

/* ---- slightly compressible NS with SUPG/PSPG/LSIC: Fluid::MPI::SCnsIM (source/mpi_scnsim.cpp:15-568) ------------
 * The preconditioner of SUPGFluidSolver (Hypre-Euclid ILU(0), mpi_supg_solver.cpp:35-192) is third-party and only
 * changes iteration counts; the oracle solves each Newton system through a caller-supplied full-system solve
 * (tests pass scipy splu).  Parity pinning: tests/fluid_cylinder_mpi_scnsim (pmax = 1.03544). */
typedef struct {
  double mu, rho, dt, solid_rho;
  double g[3];
  int32_t n_neumann;
  int32_t neumann_id[8];
  double  neumann_p[8];
  const double *stress;     /* [dim][dim][n_unodes] projected nodal viscous stress (update_stress) or NULL (= 0) */
  const double *fsi_stress; /* [dim(dim+1)/2][n_unodes] nodal FSI stress or NULL */
  const double *sigma_pml;  /* [n_cells][n_q] or NULL */
  const double *body_force; /* [n_cells][n_q][dim] or NULL */
  const double *eddy_viscosity; /* [n_unodes] nodal eddy viscosity of an attached turbulence model (mpi_scnsim.cpp:198-216:
                                   viscosity_q += max(nu_t(q), 0)) or NULL */
  int32_t formulation;      /* 0: SCnsIM (source/mpi_scnsim.cpp:137-563); 1: SUPGInsIM, the incompressible SUPG/PSPG/LSIC
                               integrand of source/mpi_insim_supg.cpp:100-262 (constant density, div-free continuity, no
                               PML / stress / FSI terms).  Pinned by tests/fluid_pressure_driven_mpi_insim_supg (vmax
                               2.5e-2) and tests/fluid_plane_wall_driven_mpi_insim_supg (|v|_2 = 4.7112) */
} orc_scns_params;

/* full-system solve callback: CSR of the assembled system (n x n), rhs -> x */
typedef void (*orc_full_solve_fn)(void *user, int32_t n, const int64_t *rowptr, const int32_t *col, const double *val,
                                  const double *rhs, double *x);

void orc_scns_cell(const orc_mesh *m, const orc_scns_params *p, int32_t cell, const double *eval, const double *present,
                   const double *fsi_acc, double *Ke, double *fe);
void orc_scns_assemble(orc_system *s, const orc_scns_params *p, int32_t use_nonzero, const double *eval,
                       const double *present, const double *fsi_acc);
/* Newton loop of SUPGFluidSolver::run_one_step (mpi_supg_solver.cpp:331-425): floor 1e-14; returns iterations or <0 */
/* SUPGFluidSolver::solve with BlockIncompSchurPreconditioner as the reference builds it (mpi_supg_solver.cpp:19-192, 297-328):
 * ILU(0)(A_vv), T_pp as an operator, ILU(0)(B2pp) inside left-preconditioned GMRES(200); on the system of the last
 * orc_scns_assemble.  counts[4] = FGMRES iterations, Tpp_itr, preconditioner applications, Pvv^-1 applications.
 * perm_v / perm_p: elimination order of the two ILU(0) (NULL: natural = Euclid on one rank) -- measurement hooks. */
int32_t orc_scns_solve(orc_system *s, int32_t use_nonzero, int32_t fgmres_restart, const int32_t *perm_v, const int32_t *perm_p,
                       double *newton_update, int64_t *counts, double *res);
/* pieces of that preconditioner (exact substitutions): which 0: y = Pvv^-1 x; 1: y = B2pp_inverse x; 2: y = B2pp x; 3: y = T_pp x */
int32_t orc_scns_pc_probe(orc_system *s, int32_t which, const double *x, double *y);
void orc_set_tri_sweeps(int32_t kv, int32_t kp); /* measurement hook: Jacobi sweeps instead of substitution in orc_scns_solve */
int32_t orc_scns_run_one_step(orc_system *s, const orc_scns_params *p, int32_t apply_nonzero, double newton_tol,
                              int32_t newton_maxit, orc_full_solve_fn solve, void *user, double *present,
                              const double *fsi_acc, double *log);
/* FluidSolver::update_stress (mpi_fluid_solver.cpp:716-811): stress[i][j][node] = nodal average of the per-cell
 * projection of 2 mu sym(grad u) from the quadrature points to the Q_kv nodes */
void orc_update_stress(const orc_mesh *m, double mu, const double *present, double *stress /*[dim][dim][n_unodes]*/);

/* ---- fluid-side inputs of MPI::FSI (source/mpi_fsi.cpp:96-127,142-223,291-663), oracle_fsi.c ----------------------------
 * The solid is a Q1 mesh (parameters "Degree = 1" in every FSI test) at its CURRENT position (move_solid_mesh(true)), with
 * the localized nodal fields the reference gathers on every rank (:350-362). */
typedef struct {
  int32_t dim, n_vertices, n_cells, n_bfaces;
  const double  *vertices;       /* [n_vertices][dim] */
  const int32_t *cell_vertices;  /* [n_cells][2^dim] lexicographic vertex order */
  const int32_t *bface_vertices; /* dim 2: [n_bfaces][2] solid_boundaries (collect_solid_boundaries :78-94); dim 3: unused */
  const double *velocity, *acceleration; /* [n_vertices][dim] */
  const double *stress;          /* [dim(dim+1)/2][n_vertices], component order (0,0),(1,0),(1,1),(2,0).. (:459-474); or NULL */
} orc_fsi_solid;
void orc_fsi_solid_box(const orc_fsi_solid *s, double *box /* [2*dim] lo,hi per direction */);
int32_t orc_fsi_point_in_solid(const orc_fsi_solid *s, const double *box, const double *point);
int32_t orc_fsi_real_to_unit(int32_t dim, const double *X, const double *p, double *xi);
/* the solid cell around p and the (projected) unit-cell point, or -1 */
int32_t orc_fsi_locate(const orc_fsi_solid *s, const double *p, double *xi);
void orc_fsi_update_indicator(const orc_mesh *m, const orc_fsi_solid *s, int32_t *indicator);
/* FSI::find_fluid_bc on one rank.  m->indicator = result of update_indicator.  fluid_stress [dim][dim][n_unodes] (may be
 * NULL = 0); fsi_stress [ncomp][n_unodes] is updated in place where the reference assigns it; fsi_acc [n_dofs] is
 * overwritten; use_dirichlet_bc: line_flag / line_val [dim*n_unodes] receive the proposed lines BEFORE the merge with
 * left_object_wins (:641-651).  Returns the number of points the cell search failed on (the reference throws). */
int32_t orc_fsi_find_fluid_bc(const orc_mesh *m, const orc_fsi_solid *s, double dt, int32_t use_dirichlet_bc,
                              const double *present, const double *fluid_stress, double *fsi_stress, double *fsi_acc,
                              int32_t *line_flag, double *line_val);

/* the fluid solution (u, p) and the nodal viscous stress at points of the solid (find_solid_bc :727-760,
 * update_solid_displacement :268-271): values [n][dim+1], stress [n][dim][dim] (may be NULL), cell [n] (-1: not found) */
void orc_fsi_fluid_at_points(const orc_mesh *m, const double *present, const double *fluid_stress, int32_t n,
                             const double *points, double *values, double *stress, int32_t *cell);

/* FE tables for cross-checks: phi[q][a], dphi[q][a][dim] on the reference cell, weights */
int32_t orc_fe_tables(int32_t dim, int32_t k, int32_t nq1d, double *phi, double *dphi, double *w, double *qp);

#ifdef __cplusplus
}
#endif
#endif
