/*
 * ifem_hip.h -- C ABI of libifem_hip.so: the MI355X (gfx950) implementation of OpenIFEM's implicit
 * incompressible Navier-Stokes fluid step (assemble + block-preconditioned FGMRES solve).
 *
 * The reference exposes no FFI for this path; the boundary is the C++ class hierarchy
 *   Fluid::MPI::FluidSolver<dim>  include/mpi_fluid_solver.h:89-151,185-206
 *   Fluid::MPI::InsIM<dim>        include/mpi_insim.h:35-99
 * Each entry point below names the reference member it replaces.  A maintainer binds them from the
 * reference's own C++; the host-side mirror of FluidSolver/InsIM that ships in this
 * repository (openifem_amd/csrc/host/) is one such caller.
 *
 * Conventions
 *  - return 0 on success, a negative IFEM_E_* code on failure; ifem_last_error() gives the message
 *    (maps the reference's AssertThrow exceptions, e.g. "Too many Newton iterations!" mpi_insim.cpp:424).
 *  - plain pointers and sizes only.  The caller owns host buffers; the context owns device buffers.
 *  - one context per GPU / per process; in a multi-GPU run every call is collective over the ranks
 *    (like the reference's MPI_COMM_WORLD collectives, mpi_fluid_solver.cpp:37).  Not re-entrant per context.
 *  - there is NO CPU fallback: every compute entry point fails with IFEM_E_NODEVICE without a HIP device.
 *
 * DoF layout (reference: DoFRenumbering::component_wise with blocks [velocity | pressure],
 * mpi_fluid_solver.cpp:125-128): a block vector is [ u | p ] with u = dim values per velocity node
 * (components interleaved), local nodes ordered owned-first then ghosts:
 *     index(u, node, c) = dim*node + c                      node in [0, n_unodes_local)
 *     index(p, node)    = dim*n_unodes_local + node         node in [0, n_pnodes_local)
 */
#ifndef IFEM_HIP_H
#define IFEM_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IFEM_OK 0
#define IFEM_E_BADPARAM (-1)
#define IFEM_E_NODEVICE (-2)
#define IFEM_E_HIP (-3)
#define IFEM_E_KRYLOV_NOCONV (-4) /* deal.II SolverControl::NoConvergence */
#define IFEM_E_NEWTON_MAXIT (-5)  /* "Too many Newton iterations!" mpi_insim.cpp:424-425 */
#define IFEM_E_COMM (-6)

typedef struct ifem_ctx ifem_ctx;

/* Replaces what FluidSolver::setup_dofs (mpi_fluid_solver.cpp:116-162) leaves behind: the locally relevant
 * part of the mesh and its cell -> DoF tables.  Local cell-node order is tensor-lexicographic (x fastest).
 * Cells listed are the ones this rank assembles (owned cells plus the ghost layer touching owned rows). */
typedef struct {
  int32_t dim;               /* 2 or 3 */
  int32_t kv;                /* velocity degree 1 or 2; pressure degree is 1 */
  int32_t n_cells;
  int32_t n_unodes_owned, n_unodes_local; /* owned first, then ghosts grouped by neighbour rank */
  int32_t n_pnodes_owned, n_pnodes_local;
  const double  *vcoords;       /* [n_cells][2^dim][dim] d-linear (MappingQ1) vertex coordinates */
  const int32_t *cell_unodes;   /* [n_cells][(kv+1)^dim] local velocity-node ids */
  const int32_t *cell_pnodes;   /* [n_cells][2^dim] local pressure-node ids */
  const int32_t *cell_face_bid; /* [n_cells][2*dim] boundary id or -1 (faces x-,x+,y-,y+,z-,z+) */
} ifem_mesh_desc;

/* Replaces the ghost-exchange plans PETSc builds for the ghosted vectors / MatMult
 * (mpi_fluid_solver.cpp:333-338).  NULL for a single-GPU run. */
typedef struct {
  int32_t rank, nranks;
  int32_t n_neighbors;
  const int32_t *neighbor_rank; /* [n_neighbors] */
  const int32_t *send_u_ptr;    /* [n_neighbors+1] into send_u_idx */
  const int32_t *send_u_idx;    /* owned velocity nodes to send, per neighbour */
  const int32_t *recv_u_ptr;    /* [n_neighbors+1]: ghost velocity nodes of neighbour k are local ids
                                   n_unodes_owned + [recv_u_ptr[k], recv_u_ptr[k+1]) */
  const int32_t *send_p_ptr, *send_p_idx, *recv_p_ptr; /* same for pressure nodes */
  const uint8_t *nccl_unique_id; /* 128 bytes from ifem_comm_unique_id(), identical on all ranks */
  void *local_world;             /* NULL for RCCL.  Validation transport: a handle from ifem_local_world_create()
                                    shared by several contexts (virtual ranks, one host thread each) of ONE process */
  /* Optional (structured pressure lattices, i.e. box meshes): 2-deep pressure halo that lets several ranks keep
   * mass_schur(1,1) = B diag(M_u)^-1 B^T explicit as the reference does (mpi_insim.cpp:44-49) -- its rows couple
   * pressure nodes two cells apart.  Column space of the distributed S_m: [owned pressure nodes | the other nodes of
   * the lattice box "owned range +-2", grouped by owner rank in global order].  sm_box_id == NULL: S_m is applied
   * matrix-free (two SpMVs) on several ranks. */
  int64_t p_lattice_n[3];        /* global pressure lattice (nodes per direction, 1 beyond dim) */
  int64_t sm_box_lo[3], sm_box_n[3]; /* the box in lattice coordinates */
  const int32_t *sm_box_id;      /* [sm_box_n[0]*sm_box_n[1]*sm_box_n[2]] (x fastest) S_m column id of every box node */
  const int64_t *l2g_p;          /* [n_pnodes_owned] global lattice id ((z*Ny + y)*Nx + x) of the owned pressure nodes */
  const int32_t *send_s_ptr, *send_s_idx, *recv_s_ptr; /* halo plan of that column space, same layout as send_p_* */
} ifem_partition;

/* Parameters::AllParameters subset used by InsIM::assemble (mpi_insim.cpp:157-161,240,272; parameters.cpp) */
typedef struct {
  double viscosity, rho, grad_div, dt;
  double gravity[3];
  int32_t n_neumann;          /* parameters.fluid_neumann_bcs (pressure BCs), mpi_insim.cpp:313-341 */
  int32_t neumann_id[8];
  double  neumann_p[8];
} ifem_ins_params;

/* How A~^-1 (MUMPS in the reference, mpi_insim.cpp:124-127) is replaced */
#define IFEM_AINV_GMRES_BJACOBI 0     /* inner GMRES(m) on A_uu, node-block Jacobi preconditioned */
#define IFEM_AINV_GMRES_BJACOBI_F32 1 /* same, with the inner SpMV reading a single-precision copy of A_uu (the outer
                                         FGMRES operator stays fp64; only the preconditioner is approximated) */

#define IFEM_AINV_SCALAR_GMRES 2      /* inner GMRES(m) on the scalar operator S^ = mu K + rho C(u) + rho/dt M applied to
                                         every velocity component (drops the grad-div and u-gradient couplings of A_uu:
                                         8x less matrix traffic; adequate while gamma*rho*dim <~ mu).  Needs
                                         ifem_set_ainv_kind(ctx, 2) before ifem_ins_assemble. */
#define IFEM_AINV_GMRES_BJACOBI_MF 3  /* inner GMRES(m) whose operator applies A_uu matrix-free (sum-factorised cell
                                         kernel on the evaluation point of the last ifem_ins_assemble): same operator
                                         as kind 0 up to fp64 rounding, ~2 kB instead of ~40 kB of HBM traffic per cell */

#define IFEM_AINV_MG 4                /* inner GMRES(m) on the matrix-free A_uu (as kind 3) preconditioned by one geometric multigrid
                                         V-cycle over the levels of ifem_mg_attach: matrix-free rediscretised operators at
                                         the injected evaluation point, Chebyshev / node-block-Jacobi smoothing */

typedef struct {
  int32_t fgmres_restart;     /* 30: deal.II SolverFGMRES default */
  int32_t fgmres_maxit;       /* 0 -> n_dofs (mpi_insim.cpp:379-380) */
  double  fgmres_rel;         /* 1e-4 */
  double  fgmres_abs;         /* 1e-12 */
  double  mp_rel, mp_abs;     /* CG(M_p) 1e-6, 1e-10 (mpi_insim.cpp:73-74) */
  double  sm_rel, sm_abs;     /* CG(S_m) 1e-3, 1e-10 (mpi_insim.cpp:88-89) */
  int32_t ainv_kind;          /* IFEM_AINV_* */
  int32_t inner_restart, inner_maxit; /* IFEM_AINV_MG: inner_maxit = 0 makes A~^-1 exactly one V-cycle, -k makes it k
                                         stationary V-cycle sweeps x += V(b - A x) (no inner Krylov loop in either case); the Krylov kinds
                                         read inner_maxit <= 0 as the default cap of 400 */
  double  inner_rel;          /* relative residual target of the inner A_uu solve */
  int32_t explicit_schur;     /* 1: form S_m = B diag(M_u)^-1 B^T explicitly like the reference (on several GPUs this
                                 needs the 2-deep pressure halo plan of ifem_partition); 0: apply it as two SpMVs */
  int32_t verbose;
  int32_t device_cg;          /* 1 (default): single-rank CG(M_p) / CG(S_m) keep their recurrence scalars on the device and
                                 the host checks the residual every 4 iterations; 0: one host round trip per dot product */
  int32_t outer_matrix_free;  /* 0 (default): the outer FGMRES operator is the assembled block matrix, as in the reference;
                                 1: its velocity-velocity block is applied matrix-free (same operator to 1e-12, a
                                 fifth of the time) -- an experiment switch, off by default */
  int32_t sm_mg;              /* 1 (default): when coarser levels are attached (ifem_mg_attach) and S_m is explicit, the
                                 S_m solve of the preconditioner (mpi_insim.cpp:86-112) is CG preconditioned by one multigrid
                                 V-cycle, same stopping rule on the true residual; 0: plain CG as in the reference */
  int32_t mg_smooth;          /* 2: Chebyshev-Jacobi smoothing steps before and after the coarse correction */
  double  mg_cheb_ratio;      /* 4: the smoother targets the eigenvalues of D^-1 S_m in [lambda_max / ratio, lambda_max] */
  int32_t mg_smooth_u;        /* 2: smoothing steps of the A_uu V-cycle (IFEM_AINV_MG) */
  int32_t mg_smooth_u_post;   /* 0: as many after the coarse correction as before it; > 0: that many (the cycle is then no
                                 longer symmetric, which the flexible inner GMRES does not need) */
  double  mg_cheb_ratio_u;    /* 4: its Chebyshev interval [lambda_max / ratio, lambda_max] of (block D)^-1 A_uu */
  double  inner_rel_first;    /* 0: off.  > 0: the FIRST preconditioner application of a solve runs its inner A_uu solve to
                                 this relative residual instead of inner_rel.  The first Krylov direction decides whether
                                 the outer iteration ends at its first check: at 128^3, 5e-5 there (four inner iterations)
                                 gives 7.8e-5 ||rhs|| after one outer iteration where 1e-2 gives 9.1e-4 and needs a second
                                 one.  Only used when that first residual is velocity-dominated (pressure share below
                                 inner_first_pshare): a residual that is mostly continuity equation (later Newton iterations)
                                 needs several outer iterations whatever the velocity solve does.  Self-correcting: when a solve
                                 that used it still needed a second outer iteration (the extra inner iterations bought nothing:
                                 64^3 channel), the context skips it for its next 8 (16, 32, 64 after repeated misses) qualifying
                                 solves before trying again */
  double  inner_first_pshare; /* 0 (default): 10 fgmres_rel.  The largest pressure share ||r_p|| / ||r|| of the first Krylov vector
                                 for which inner_rel_first is used -- a heuristic: the block-triangular preconditioner leaves
                                 O(0.1) of the pressure part of a residual behind per outer iteration */
} ifem_solver_opts;

/* Tuning / measurement knobs of one context (defaults = the measured best; nothing here changes results beyond fp64
 * rounding).  Replaces the environment switches of the first round: the library reads no environment variable. */
typedef struct {
  int32_t geo_cache;     /* 1: B, B^T, M_p, diag(M_u) and S_m are kept across assemblies with an unchanged constrained-dof set; a
                            new set takes B / B^T as masked copies of the unconstrained blocks (integrated once per mesh) and
                            re-forms S_m.  2 (measurement): every assembly is treated as a new set.  0: everything is
                            re-integrated by every assembly as the reference does (mpi_insim.cpp:163-165) */
  int32_t xcd_swizzle;   /* 1: cell kernels hand every XCD one contiguous range of the (Morton-ordered) cells */
  int32_t asm_skip;      /* 0; read by measurement builds only (-DIFEM_ASM_PROBES): drop parts of the 3D Q2/Q1 assembly kernel (results
                            invalid); the default build ignores it */
  int32_t spmv_lanes;    /* 32: lanes per block row of the A_uu SpMV (8/16/32/64) */
  int32_t sm_lanes;      /* 32: lanes per row of the S_m SpMV */
  int32_t mf_f32;        /* 1: single-precision cell arithmetic in the matrix-free A_uu of the INNER solve */
  int32_t tpp_operator;  /* 0; 1: SCnsIM preconditioner applies T_pp as an operator instead of the explicit matrix */
  int32_t spmv_pipe;     /* 1: the fp64 A_uu SpMV (3 x 3 blocks) as a software-pipelined walk of row ranges (k_spmv_uu_pipe); 0: one
                            group of lanes per row */
  int32_t halo_overlap;  /* 1: several ranks: the halo of an operator input travels on a second stream / communicator
                            while the rows (SpMV) or cells (matrix-free A_uu) that read no ghost value are processed; 0: the
                            exchange completes on the context stream before the operator starts */
  int32_t asm3_variant;  /* 3D Q2/Q1 cell kernel: 0 (default) = the matrix-core kernel of assemble3.hip (one wavefront per cell); 1 = the
                            general vector kernel of assemble2.hip (also taken when an A_uu row holds 512 blocks or more) */
  int32_t cg_single_reduction; /* 1 (default): the device-resident CG of the pressure solves (CG(M_p), plain CG(S_m)) forms its three dot
                            products in one pass and one all-reduce per iteration (Chronopoulos / Gear recurrence); 0: the textbook
                            recurrence with two reductions per iteration */
  int32_t asm3_cpb;      /* 2: cells (= wavefronts) per workgroup of that kernel (2, 4 or 8; anything else is IFEM_E_BADPARAM) */
  int32_t tpp_milu_permille; /* SCnsIM, ILU(0) of T_pp: 950 (default); 0 plain ILU(0); w in (0, 1000]: relaxed modified ILU, w/1000 of every dropped
                                fill-in entry is added to the diagonal of its row */
  int64_t tpp_ilu_order; /* SCnsIM, explicit T_pp: preconditioner of its inner GMRES -- 2 (default): ILU(0), natural row order when its
                            elimination levels hold >= 1024 rows on average, multicolour otherwise; 0: natural order always (fewest
                            iterations, O(n^(1/dim)) level-scheduled launches); 1: multicolour always (a few dozen levels whatever
                            the mesh, ~3 x the iterations); -1 Jacobi(T_pp) */
  int64_t basis_pad;     /* 32*33 doubles of padding between Krylov basis columns (HBM channel spread) */
  int32_t tpp_tri_sweeps; /* SCnsIM, ILU(0) of T_pp: 0 (default) exact level-scheduled triangular solves; k > 0: k Jacobi sweeps per
                             triangular system instead (2 k row-parallel launches whatever the number of levels) */
  int32_t uu_row_order;  /* 1 (default): 3D Q2/Q1 contexts store the blocks of an A_uu row in the order (last cell, first cell, column) of the
                            cells that touch them instead of column order, so that a cell's part of a row is a few contiguous runs for the
                            cell kernel's atomics (takes effect before the first assembly of the context); 0: column order */
  int32_t eig_steps;     /* 0 (default: 12 / 14): power-iteration steps of a COLD estimate of the smoothers' eigenvalue bounds (lambda_max of
                            (block D)^-1 A_uu and of D^-1 S_m on every level); > 0: that many */
  int32_t vcycle_graph_cells; /* 262144 (default): on a single-rank level chain whose finest level has at most this many cells the A_uu V-cycle
                                 of IFEM_AINV_MG and the S_m V-cycle inside CG(S_m) are captured into hipGraphs once per state and replayed
                                 (their ~100-200 short launches are launch latency there: the reference's test meshes); 0: always launched
                                 eagerly */
  int32_t scns_pc;       /* SCnsIM / SUPGInsIM block preconditioner.  2 (default): the reference's structure (mpi_supg_solver.cpp:35-192):
                            P_vv^-1 = ILU(0) of A_vv (node blocks, natural order; the owned x owned block per rank), T_pp applied as the
                            OPERATOR A_pp - A_pv P_vv^-1 A_vp, inner GMRES(200) preconditioned by the ILU(0) of the assembled
                            B2pp = A_pp - A_pv rowsum(|A_vv|)^-1 A_vp.  1: rounds 2-5: node-block Jacobi P_vv^-1, T_pp explicit (tpp_*) */
  int32_t pvv_sweeps;    /* 4: Jacobi sweeps per triangular system when ILU(0)(A_vv) is applied (bilu.hip); < 0: exact substitution */
  int32_t b2pp_sweeps;   /* 6: the same for ILU(0)(B2pp) (measured on the cylinder at 3 / 4 refinements: 3/5 -> 85 / 455 ms per solve,
                            4/6 -> 79 / 437, 5/8 -> 83 / 417; exact substitution: 1.4 / 10.7 s) */
  int32_t scns_inner_reorth; /* 0 (default): the inner GMRES(200) on T_pp orthogonalises once per iteration (classical Gram-Schmidt, one
                            fused pass: it is a preconditioner solved to 1e-3); 1: twice, as the outer FGMRES */
  int32_t scns_inner_left; /* 1 (default): that GMRES is LEFT-preconditioned and stops on the preconditioned residual, as deal.II's SolverGMRES
                            does with its defaults (mpi_supg_solver.cpp:174-182); 0: right-preconditioned, true residual */
  int32_t scns_graph;    /* 0 (default): eager launches.  1: single rank: the ~20 short launches of one inner iteration's B2pp_inverse (T_pp v)
                            are replayed as a captured hipGraph -- measured: no gain (84.6 against 85.3 ms per solve: the device-side
                            dependency chain of the short kernels, not the host's launch rate, sets the pace) */
  int32_t stored_uu;     /* 1 (default): ifem_ins_assemble scatters the velocity-velocity block into the block CSR, as the reference does
                            (mpi_insim.cpp:343-361).  0: A_uu is never stored: the assembly integrates the right-hand side (B, B^T, M_p, diag(M_u)
                            through the cached geometry path), the outer operator applies A_uu matrix-free in fp64 (equal to the stored block to
                            1e-13) and the smoothers take their node blocks from the cell integrals.  Needs the matrix-free inner solvers
                            (IFEM_AINV_MG / _BJACOBI_MF), geo_cache >= 1 and no hanging nodes.  An assembly whose constraint set carries
                            non-zero values (the first Newton iteration of a step with inhomogeneous boundary values: distribute_local_to_global
                            needs the element matrix columns) takes the stored path by itself -- and allocates the values then.  Without such
                            assemblies 78 GB less memory at 128^3; ~2 x the step rate.  The block CSR stays the default (north star). */
} ifem_tuning;
/* Initialise an ifem_tuning with ifem_default_tuning before changing fields: a zero-initialised struct gets the documented defaults
 * only for the fields where 0 is not a meaningful value (asm3_cpb, scns_pc, pvv_sweeps, b2pp_sweeps). */
void ifem_default_tuning(ifem_tuning *t);
int ifem_set_tuning(ifem_ctx *ctx, const ifem_tuning *t);

/* counters of the last ifem_solve (the timer2 sections of mpi_insim.cpp:70,87,125) */
typedef struct {
  uint32_t fgmres_iters; double fgmres_res;
  uint32_t precond_applies, cg_mp_iters, cg_sm_iters, inner_iters;
  double t_schur_setup_ms, t_cg_mp_ms, t_cg_sm_ms, t_ainv_ms, t_spmv_ms, t_total_ms;
  uint32_t sm_mg_levels; /* levels used by the multigrid-preconditioned CG(S_m) of the last solve (0: plain CG) */
  uint32_t inner_first_tight; /* 1: the first preconditioner application of this solve ran with inner_rel_first */
} ifem_solve_stats;

/* context-resident block vectors (reference members of FluidSolver / InsIM) */
enum {
  IFEM_VEC_PRESENT = 0,   /* present_solution       mpi_fluid_solver.h:203 */
  IFEM_VEC_EVAL = 1,      /* evaluation_point       mpi_insim.h */
  IFEM_VEC_FSI_ACC = 2,   /* fsi_acceleration       mpi_fluid_solver.h:209 */
  IFEM_VEC_UPDATE = 3,    /* newton_update */
  IFEM_VEC_RHS = 4,       /* system_rhs */
  IFEM_VEC_INCREMENT = 5, /* solution_increment */
  IFEM_VEC_TMP = 6,
  IFEM_N_VECS = 7
};

const char *ifem_last_error(void);
/* sizeof of the structs above as this library was compiled, for a binding to check its mirror against:
 * 0 ifem_mesh_desc, 1 ifem_partition, 2 ifem_ins_params, 3 ifem_solver_opts, 4 ifem_solve_stats, 5 ifem_scns_params,
 * 6 ifem_timing, 7 ifem_tuning, 8 ifem_mg_transfer, 9 ifem_fsi_solid, 10 ifem_fsi_stats, 11 ifem_comm_stats, 12 ifem_kprof_entry;
 * -1 for anything else */
int64_t ifem_abi_sizeof(int which);
int ifem_device_count(void);
void ifem_default_solver_opts(ifem_solver_opts *o);
/* RCCL bootstrap: rank 0 calls this and ships the 128 bytes to the other ranks by any channel */
int ifem_comm_unique_id(uint8_t out[128]);
/* one-rank RCCL round trip (communicator + all-reduce + grouped send/recv to self) on `device` */
int ifem_comm_selftest(int device);
/* What the communicator of a context looks like and how often it was used since the last reset, summed over the context
 * and the multigrid levels attached below it: the reference's counterpart is PETSc's -log_view line of VecScatter /
 * MPI_Allreduce counts.  rccl_nranks is ncclCommCount of the context's communicator (0: no RCCL communicator). */
typedef struct {
  int32_t nranks, rank, n_neighbors;
  int32_t transport;     /* 0 single rank, 1 RCCL, 2 validation transport (local world) */
  int32_t rccl_nranks;   /* ncclCommCount */
  int32_t halo_stream;   /* 1: overlapped exchanges on the second communicator + priority stream */
  int32_t levels;        /* contexts summed over (1 + attached multigrid levels) */
  int32_t reserved_;     /* padding, always 0 */
  uint64_t halo_exchanges; /* packed send/recv groups (forward and reverse) */
  uint64_t allreduce_dev;  /* all-reduces of device scalars ordered on the stream (no host wait) */
  uint64_t allreduce_host; /* all-reduces the host waited for (Gram-Schmidt coefficients, norms) */
  uint64_t allreduce_vec;  /* stream-ordered all-reduces of whole level vectors: the hand-over to a replicated coarse level */
} ifem_comm_stats;
int ifem_comm_stats_get(ifem_ctx *ctx, ifem_comm_stats *out, int reset);
/* the same counters of ONE level of the chain (0: ctx itself, k: the k-th context attached below it); `levels` = 1 */
int ifem_comm_stats_level(ifem_ctx *ctx, int level, ifem_comm_stats *out);
/* validation transport (see ifem_partition::local_world): nranks contexts of one process on one GPU */
void *ifem_local_world_create(int nranks);
void ifem_local_world_destroy(void *world);

/* FluidSolver::initialize_system (mpi_fluid_solver.cpp:305-365, mpi_insim.cpp:143-150): builds the block
 * sparsity (A_uu as dim x dim BSR, B, B^T, M_p, S_m) on the device, scatter maps, halo plans and vectors. */
int ifem_ctx_create(const ifem_mesh_desc *mesh, const ifem_partition *part, int device, ifem_ctx **out);
void ifem_ctx_destroy(ifem_ctx *ctx);
int64_t ifem_n_local_dofs(const ifem_ctx *ctx);
int64_t ifem_nnz(const ifem_ctx *ctx, int block); /* 0: A_uu blocks, 1: B blocks, 2: M_p, 3: S_m */

/* FluidSolver::make_constraints result (mpi_fluid_solver.cpp:165-280): which = 0 zero_constraints,
 * 1 nonzero_constraints; Dirichlet lines (local dof, inhomogeneity; inhom == NULL: all zero).  A dof listed twice keeps
 * its last line.  Only the n lines are uploaded: the flags / inhomogeneities over the local dofs are scattered on the
 * device and compared there with the previous sets (which decides whether B, B^T, S_m of the last set can be kept), so
 * re-making the constraints every time step, as MPI::FSI does (mpi_fsi.cpp:1191), costs ~2 ms per call at 128^3. */
int ifem_set_constraints(ifem_ctx *ctx, int which, int32_t n, const int32_t *dof, const double *inhom);
/* Hanging-node lines of both AffineConstraints objects (DoFTools::make_hanging_node_constraints,
 * mpi_fluid_solver.cpp:182-184; consumed by distribute_local_to_global, mpi_insim.cpp:343-355, and by
 * constraints.distribute, :390):  x[dof[i]] = sum_{k in [ptr[i], ptr[i+1])} weight[k] * x[master[k]],  local dof ids as in
 * ifem_set_constraints (velocity dofs, then n_u + pressure node).  The lines must be closed with respect to each other
 * (no master is itself a hanging dof); masters may be Dirichlet-constrained (their inhomogeneity is inherited, as
 * AffineConstraints::close() does).  A hanging dof must not also be listed in ifem_set_constraints
 * (interpolate_boundary_values skips constrained dofs).  Assemble, solve and the *_step calls then work on the condensed
 * system; the returned update has its hanging entries interpolated.  n = 0 removes the lines.
 * Partitioned contexts (collective call; a rank without hanging nodes passes n = 0): list the lines of every LOCAL hanging
 * dof, owned or ghost, with local master ids -- the ghost layer of the rank must therefore hold the masters of its ghost
 * hanging nodes (p4est's ghost layer plus DoFTools::extract_locally_relevant_dofs does the same for the reference,
 * mpi_fluid_solver.cpp:140-152,182-184). */
int ifem_set_hanging_constraints(ifem_ctx *ctx, int32_t n, const int32_t *dof, const int32_t *ptr, const int32_t *master,
                                 const double *weight);
/* cell_property[*].indicator written by MPI::FSI::update_indicator (mpi_fsi.cpp:291-321); NULL = all 0 */
int ifem_set_cell_fields(ifem_ctx *ctx, const int32_t *indicator);

/* block-vector plumbing (PETScWrappers::MPI::BlockVector assignments in run_one_step) */
int ifem_vec_set(ifem_ctx *ctx, int vec, const double *host);
int ifem_vec_get(ifem_ctx *ctx, int vec, double *host);
int ifem_vec_copy(ifem_ctx *ctx, int dst, int src);
int ifem_vec_zero(ifem_ctx *ctx, int vec);
int ifem_vec_axpy(ifem_ctx *ctx, double a, int x, int y); /* y += a x */
int ifem_vec_norm2(ifem_ctx *ctx, int vec, double *out);  /* l2_norm(), all-reduced over ranks */
/* Utils::PETScVectorMax/Min over one block (source/utilities.cpp:635-651): block 0 velocity, 1 pressure */
int ifem_vec_minmax(ifem_ctx *ctx, int vec, int block, double *vmin, double *vmax);
int ifem_halo_exchange(ifem_ctx *ctx, int vec);           /* ghosted-vector assignment */

/* ---- Geometric multigrid levels for the preconditioner's inner solves.  The reference gets mesh-independent inner solves
 * from MUMPS (mpi_insim.cpp:124-127); its CG(S_m) (:86-112) is unpreconditioned.  Here the caller may attach a chain of
 * coarser contexts -- the same problem (domain, boundary ids, constraint sets, partition over the same ranks) on coarser
 * meshes, e.g. the levels a Triangulation passes through under refine_global, or semi-coarsened box meshes for stretched
 * cells -- with the nodal prolongation between neighbouring levels.  Coarse operators are REDISCRETISED on those contexts
 * (S_m: geometry blocks + the same B diag(M_u)^-1 B^T), nothing is assembled by Galerkin products.  Only the
 * preconditioner changes: the outer FGMRES operator and stopping rules stay the reference's.
 * P_p: pressure-node prolongation, rows = the OWNED pressure nodes of `fine`, columns = LOCAL (owned + ghost) pressure
 * nodes of `coarse` on the same rank; R_p = P_p^T, rows = LOCAL pressure nodes of `coarse`, columns = owned pressure
 * nodes of `fine` (ghost rows are sent to their owners and added, like a PETSc reverse scatter).  The caller keeps
 * `coarse` alive and keeps its BOUNDARY constraint sets in step with those of `fine`; the tables are copied.  Lines that only
 * the fine context carries (the artificial-fluid Dirichlet lines ifem_fsi_find_fluid_bc merges into its sets every coupled step)
 * need not be mirrored: the coarse levels then treat those dofs as free, the transfers still drop them on the fine side
 * (per-weight masks rebuilt whenever a constrained-dof set changes) -- a weaker coarse correction inside the preconditioner,
 * the outer operator and its stopping rule are untouched (tests/test_gpu_fsi_caller.py runs InsIM with attached levels under
 * such lines).  The V-cycle of the A_uu block always runs on single-precision level vectors, whatever ifem_tuning::mf_f32 says
 * (that switch selects the cell arithmetic of the inner GMRES's operator only).
 * REPLICATED coarse level (several ranks): `coarse` may be a SINGLE-RANK context of the WHOLE coarse mesh, created with identical
 * content on every rank, below a partitioned `fine`.  All its nodes are "local": P columns index them directly, R has one row per
 * coarse node and holds this rank's owned fine columns only -- the ranks' partial restrictions are summed by one stream-ordered
 * vector all-reduce (ncclAllReduce) per V-cycle instead of a reverse halo exchange, the prolongation needs no exchange, and every
 * level attached below the replica (ordinary single-rank levels) runs redundantly on every rank without communication.  inj_u[k] = -1
 * where another rank owns the fine node under coarse node k (the injected evaluation point is summed over the ranks as well).
 * Meant for the levels whose meshes are too small to be worth a message per smoothing step (the reference has
 * no counterpart -- MUMPS gathers its coarse fronts the same way). */
typedef struct {
  int64_t n_fine_p_owned, n_coarse_p_local;
  const int64_t *pp_ptr; const int32_t *pp_col; const double *pp_w; /* CSR of P_p */
  const int64_t *rp_ptr; const int32_t *rp_col; const double *rp_w; /* CSR of R_p = P_p^T */
  /* optional (all NULL: pressure levels only): the same for the velocity NODES (applied to the dim components of a node),
   * rows of P_u = owned velocity nodes of `fine`, columns = local velocity nodes of `coarse`, R_u = P_u^T; inj_u
   * [n_unodes_owned of coarse]: the owned velocity node of `fine` at the same point (nested meshes: the coarse evaluation
   * point is the fine one restricted by injection) -- enables IFEM_AINV_MG */
  int64_t n_fine_u_owned, n_coarse_u_local;
  const int64_t *pu_ptr; const int32_t *pu_col; const double *pu_w;
  const int64_t *ru_ptr; const int32_t *ru_col; const double *ru_w;
  const int32_t *inj_u;
} ifem_mg_transfer;
int ifem_mg_attach(ifem_ctx *fine, ifem_ctx *coarse, const ifem_mg_transfer *t);
/* number of levels below ctx (0: none attached) */
int ifem_mg_depth(const ifem_ctx *ctx);

/* Tells the next ifem_ins_assemble which extra operators the chosen A_uu^-1 replacement needs (IFEM_AINV_*). */
int ifem_set_ainv_kind(ifem_ctx *ctx, int kind);
/* InsIM::assemble(use_nonzero_constraints), mpi_insim.cpp:153-362.  Reads IFEM_VEC_EVAL, _PRESENT,
 * _FSI_ACC; writes the device matrices (A_uu, B, B^T, diag(M_u), M_p) and IFEM_VEC_RHS. */
int ifem_ins_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero);
/* InsIM::solve(use_nonzero_constraints), mpi_insim.cpp:365-395 incl. BlockSchurPreconditioner :13-128.
 * Solves into IFEM_VEC_UPDATE and applies constraints.distribute(). */
int ifem_solve(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int use_nonzero,
               ifem_solve_stats *stats);
/* The TRUE residual of the last solve, recomputed with the assembled operator: ||system_rhs - system_matrix * newton_update||_2
 * over the rows that carry an equation (constrained rows hold d x = 0 after constraints.distribute), all-reduced, next to
 * ||system_rhs||_2.  What SolverControl's last_value() (mpi_insim.cpp:379-388) estimates through the Krylov recurrence. */
int ifem_true_residual(ifem_ctx *ctx, double *residual_l2, double *rhs_l2);
/* system_rhs.l2_norm(), mpi_insim.cpp:438 */
int ifem_rhs_norm(ifem_ctx *ctx, double *l2);
/* the Newton loop of InsIM::run_one_step, mpi_insim.cpp:416-473.  log (may be NULL) receives
 * [ABS_RES, REL_RES, GMRES_ITR, GMRES_RES] per iteration.  Returns the number of Newton iterations. */
int ifem_ins_newton_step(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int apply_nonzero,
                         double tolerance, int max_iterations, double *log);

/* ---- Fluid::MPI::InsIMEX (source/mpi_insimex.cpp): implicit-explicit incompressible NS.  The matrix (viscous +
 * grad-div + mass/dt + B, B^T: symmetric, independent of the solution) is assembled when assemble_system != 0; the
 * right-hand side (explicit convection, everything evaluated at IFEM_VEC_PRESENT) always.  With assemble_system = 0 only
 * the rhs is integrated and constrained rows are dropped (AffineConstraints::distribute_local_to_global of a vector,
 * mpi_insimex.cpp:343-346). */
int ifem_imex_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero, int assemble_system);
/* InsIMEX::solve (:358-393): FGMRES to min(1e-9, 1e-8 ||rhs||) with the block Schur preconditioner of :8-131 (CG(M_p),
 * CG(S_m), A_uu^-1 through the inner solver selected by o->ainv_kind; the reference's CG(A_uu) tolerance is
 * o->inner_rel = 1e-4), result in IFEM_VEC_UPDATE (solution_time_increment) */
int ifem_imex_solve(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats);
/* InsIMEX::run_one_step (:396-446) without output/checkpoint: increment = 0, assemble, solve, present += increment,
 * update_stress */
int ifem_imex_step(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int apply_nonzero, int assemble_system,
                   ifem_solve_stats *stats);

/* ---- Fluid::MPI::SCnsIM / SUPGFluidSolver (source/mpi_scnsim.cpp, source/mpi_supg_solver.cpp): slightly compressible
 * Navier-Stokes with SUPG / PSPG / LSIC.  Partitioned contexts work as for InsIM (owner-computes rows, halo refresh of
 * the Krylov vectors and of the projected stress). */
typedef struct {
  double viscosity, rho, dt, solid_rho; /* parameters.viscosity, fluid_rho, time step, solid_rho (artificial fluid) */
  double gravity[3];
  int32_t n_neumann;
  int32_t neumann_id[8];
  double  neumann_p[8];
  int32_t formulation; /* IFEM_FORM_SCNSIM (SCnsIM::assemble, mpi_scnsim.cpp:137-563) or IFEM_FORM_SUPG_INSIM
                          (SUPGInsIM::assemble, mpi_insim_supg.cpp:15-327: incompressible, constant density, no PML /
                          projected-stress / FSI terms); both are solved by ifem_scns_solve (SUPGFluidSolver::solve) */
} ifem_scns_params;
#define IFEM_FORM_SCNSIM 0
#define IFEM_FORM_SUPG_INSIM 1
/* optional cell / nodal fields of SCnsIM::assemble (each may be NULL = absent): sigma_pml [n_cells][n_q]
 * (set_sigma_pml_field evaluated at the quadrature points, mpi_scnsim.cpp:188-192), body_force [n_cells][n_q][dim]
 * (set_body_force, :193-197), fsi_stress [dim(dim+1)/2][n_unodes_local] (MPI::FSI, mpi_fsi.cpp:469-471) */
int ifem_set_scns_fields(ifem_ctx *ctx, const double *sigma_pml, const double *body_force, const double *fsi_stress);
/* nodal eddy viscosity of an attached turbulence model ([n_unodes_local] on the scalar Q_kv space, or NULL to detach):
 * SCnsIM::assemble adds max(nu_t(q), 0) to the viscosity at every quadrature point (mpi_scnsim.cpp:198-216).  The
 * turbulence model itself (Spalart-Allmaras, source/mpi_spalart_allmaras.cpp) is outside the path. */
int ifem_set_eddy_viscosity(ifem_ctx *ctx, const double *nodal);
/* FluidSolver::update_stress (mpi_fluid_solver.cpp:716-811) from IFEM_VEC_PRESENT into the context's nodal stress
 * (read by the next ifem_scns_assemble); host_out (may be NULL) receives [dim][dim][n_unodes_local] */
int ifem_update_stress(ifem_ctx *ctx, double viscosity, double *host_out);
/* SCnsIM::assemble (mpi_scnsim.cpp:15-568): reads IFEM_VEC_EVAL/_PRESENT/_FSI_ACC and the fields above */
int ifem_scns_assemble(ifem_ctx *ctx, const ifem_scns_params *p, int use_nonzero);
/* SUPGFluidSolver::solve (mpi_supg_solver.cpp:297-328): FGMRES to 1e-6 ||rhs|| with the block Schur preconditioner of
 * :35-192 in which the two Euclid ILU(0) factorisations are replaced by node-block Jacobi / Jacobi */
int ifem_scns_solve(ifem_ctx *ctx, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats);
/* Newton loop of SUPGFluidSolver::run_one_step (:331-425, floor 1e-14) followed by update_stress */
int ifem_scns_newton_step(ifem_ctx *ctx, const ifem_scns_params *p, const ifem_solver_opts *o, int apply_nonzero,
                          double tolerance, int max_iterations, double *log);

/* ---- fluid-side inputs of MPI::FSI produced on the device (source/mpi_fsi.cpp) --------------------
 * What the FSI driver computes on the fluid mesh before every fluid step (mpi_fsi.cpp:1189-1208), from the solid as every
 * rank sees it: the cell indicator, fsi_acceleration, the nodal fsi_stress and -- with use_dirichlet_bc -- the Dirichlet
 * lines of the artificial fluid merged into both constraint objects.  Nothing of it leaves the device; the host hands over
 * the (small) solid only.  The solid solver itself is outside the path.
 * The solid is a Q1 mesh ("Degree = 1" in every FSI test of the reference) at its CURRENT position
 * (FSI::move_solid_mesh(true), :40-76) with the localized nodal fields of :350-362. */
typedef struct {
  int32_t n_vertices, n_cells, n_boundary_faces;
  const double  *vertices;      /* [n_vertices][dim] */
  const int32_t *cell_vertices; /* [n_cells][2^dim], lexicographic (deal.II) vertex order */
  const int32_t *boundary_face_vertices; /* dim 2: [n_boundary_faces][2] = solid_boundaries (collect_solid_boundaries,
                                   :78-94), vertex order of face->vertex(0), ->vertex(1); dim 3: unused, may be NULL */
  const double *velocity, *acceleration; /* [n_vertices][dim] localized_solid_velocity / _acceleration; may be NULL until
                                   ifem_fsi_find_fluid_bc is called */
  const double *stress;         /* [dim(dim+1)/2][n_vertices] localized_stress[i][j], j <= i in the loop order of :459-474,
                                   or NULL: the nodal fsi_stress is then left alone */
} ifem_fsi_solid;
/* uploads the solid and computes solid_box (FSI::update_solid_box, :96-127) */
int ifem_fsi_set_solid(ifem_ctx *ctx, const ifem_fsi_solid *solid);
/* FSI::update_indicator (:291-319): indicator = 1 on the local cells (owned and ghost layer) whose vertices all lie in the
 * solid (FSI::point_in_solid, :142-223), written to the context's cell indicator (what ifem_set_cell_fields sets);
 * host_out [n_cells] and n_artificial may be NULL */
int ifem_fsi_update_indicator(ifem_ctx *ctx, int32_t *host_out, int64_t *n_artificial);
typedef struct {
  int64_t n_candidates;  /* velocity nodes of the first-touch sets inside solid_box */
  int64_t n_inside;      /* of them inside the solid */
  int64_t n_lines;       /* Dirichlet lines added to each constraint object (use_dirichlet_bc) */
  int64_t n_not_found;   /* points inside the solid no solid cell was found for: != 0 makes the call fail like the
                            reference's AssertThrow "Cannot find point in solid" (:526-533) */
} ifem_fsi_stats;
/* FSI::find_fluid_bc (:323-663).  Reads IFEM_VEC_PRESENT, the projected stress of the last ifem_update_stress (zero before
 * the first), the cell indicator; writes the nodal fsi_stress (:415-480, entries elsewhere keep their values),
 * IFEM_VEC_FSI_ACC (:489-556; zero with use_dirichlet_bc) and, with use_dirichlet_bc, merges the lines
 * v_solid - present (nonzero set) / 0 (zero set) of the velocity dofs inside the solid into the two constraint objects of
 * ifem_set_constraints with left_object_wins: dofs that already carry a boundary or hanging-node line keep it (:569-651).
 * The caller re-makes the boundary lines first, as the reference does (fluid_solver.make_constraints(), :1191).
 * The nodal fsi_stress lives in the context from the first call on (zero-initialised, as fluid_solver.fsi_stress is);
 * ifem_set_scns_fields(..., fsi_stress = NULL) releases it, a non-NULL pointer replaces it.
 * First-touch rule (:437-441, :506-508): the reference evaluates a node in the first cell of its loop that touches it;
 * here that is the touching cell of smallest cell_order[c] (NULL: the local cell index, i.e. the loop order of a single
 * rank).  On several ranks pass the global active-cell index: every rank then picks the same cell for the nodes it
 * owns (the reference's VectorOperation::insert leaves that choice to message order) and ghosts take the owner's value. */
int ifem_fsi_find_fluid_bc(ifem_ctx *ctx, double dt, int use_dirichlet_bc, const int32_t *cell_order, ifem_fsi_stats *stats);
/* The other direction of the coupling: the fluid solution at points of the solid -- Utils::GridInterpolator(fluid
 * dof_handler, point).point_value(present_solution) and the scalar interpolator of the nodal viscous stress in the same
 * cell, as FSI::find_solid_bc (:727-760: sigma = -p I + viscous stress at the solid's boundary vertices) and
 * FSI::update_solid_displacement (:268-271) use them.  Reads IFEM_VEC_PRESENT and the projected stress of the last
 * ifem_update_stress on the device; only the n points and their results cross the bus (instead of the whole solution).
 * values [n][dim+1] = (u, p); stress [n][dim][dim] or NULL; cell [n] = the local cell around the point (smallest
 * distance_to_unit_cell below 1e-10, lowest index on ties) or -1 with zero values, which is what point_value returns for a
 * point outside the locally owned cells.  On several ranks a point is found by the rank(s) whose local cells hold it. */
int ifem_fsi_fluid_at_points(ifem_ctx *ctx, int32_t n, const double *points, double *values, double *stress, int32_t *cell);
/* read-back hooks: nodal fsi_stress [dim(dim+1)/2][n_unodes_local]; constraint object `which`: flags and inhomogeneities
 * over the local dofs [n_local] (either pointer may be NULL) */
int ifem_fsi_get_stress(ifem_ctx *ctx, double *host_out);
int ifem_get_constraints(ifem_ctx *ctx, int which, uint8_t *flags, double *inhom);

/* y = [A Bt; B 0] x on context vectors (system_matrix.vmult) -- test / bench hook */
int ifem_system_vmult(ifem_ctx *ctx, int dst, int src);
/* y = [diag(M_u) x_u ; M_p x_p] on context vectors: the two blocks of mass_matrix the reference's preconditioner reads
 * (mpi_insim.cpp:27-49: diag of block (0,0), block (1,1)) -- test hook */
int ifem_mass_vmult(ifem_ctx *ctx, int dst, int src);
/* inverse dim x dim diagonal node blocks of A_uu [n_unodes_owned][dim*dim] -- test hook.  which = 0: from the assembled
 * matrix (the block-Jacobi data of the last assembly); 1: recomputed matrix-free from the operator state of the last assembly
 * (what a coarse multigrid level uses, mg.hip::k_uu_diag) */
int ifem_uu_block_diag(ifem_ctx *ctx, int which, double *host_out);
/* y_u = A_uu x_u on the velocity part of two context vectors -- test / bench hook.  variant: IFEM_AINV_GMRES_BJACOBI
 * (stored fp64 matrix), _F32 (its single-precision copy), _MF (matrix-free, fp64 cell arithmetic) or IFEM_AINV_MG (matrix-free
 * with the single-precision cell arithmetic of the inner solve: ifem_tuning::mf_f32) */
int ifem_uu_vmult(ifem_ctx *ctx, int dst, int src, int variant);
/* z = P^-1 v, BlockSchurPreconditioner::vmult (mpi_insim.cpp:57-128) on context vectors -- test hook */
int ifem_precond_vmult(ifem_ctx *ctx, const ifem_ins_params *p, const ifem_solver_opts *o, int dst, int src);

/* (the test aids of the SCnsIM preconditioner -- ifem_tpp_ilu_probe, ifem_tpp_override, ifem_scns_pc_probe -- live in
 * ifem_hip_testing.h: they are not part of the drop-in surface) */

/* Export the assembled block system as one CSR over the local dofs [u|p] (host arrays; call twice: first
 * with col = val = NULL to get nnz through rowptr[n]).  which: 0 system_matrix, 1 mass (diag(M_u), M_p).
 * The columns of a row are NOT sorted in general: 3D Q2/Q1 contexts store the velocity-velocity blocks of a row in scatter
 * order (ifem_tuning::uu_row_order); every (row, column) appears once. */
int ifem_export_csr(ifem_ctx *ctx, int which, int64_t *rowptr, int32_t *col, double *val);
/* The same for the rows [row0, row0 + nrows) only: rowptr has nrows + 1 entries and starts at 0; only the slices of the device
 * arrays these rows own are downloaded (at 128^3 the whole A_uu is 78 GB).  Call with col = val = NULL first to size the arrays
 * (nnz = rowptr[nrows]).  A parity test's window onto a bench-size system: a slab of rows against the oracle's assembly of the
 * cells that touch it (the reference's counterpart: one rank's rows of system_matrix, mpi_insim.cpp:343-361). */
int ifem_export_rows(ifem_ctx *ctx, int which, int64_t row0, int64_t nrows, int64_t *rowptr, int32_t *col, double *val);
/* The stored block pattern of A_uu for the velocity nodes [node0, node0 + n_nodes): rowptr (n_nodes + 1 entries) holds ABSOLUTE
 * block offsets into the value array (block k owns the doubles [k dim^2, (k + 1) dim^2)), col (rowptr[n_nodes] - rowptr[0] entries,
 * may be NULL) the block columns in storage order (ifem_tuning::uu_row_order).  Measurement aid: the addresses the fused scatter
 * of the cell kernel hits (tools/scatter_sim.py replays them into 64-byte segments per cell on any mesh). */
int ifem_export_uu_pattern(ifem_ctx *ctx, int64_t node0, int64_t n_nodes, int64_t *rowptr, int32_t *col);

/* per-kernel timing of the last assemble/solve, HIP events on the context stream */
typedef struct {
  double assemble_ms;        /* whole ifem_ins_assemble */
  double assemble_kernel_ms; /* the cell-integration + scatter kernel alone */
  double spmv_uu_ms_avg; uint64_t spmv_uu_calls; /* A_uu BSR SpMV (dominant kernel of the solve) */
  double spmv_uu_bytes;      /* algorithmic bytes per call */
  double mf_ms_avg; uint64_t mf_calls; /* matrix-free A_uu application (IFEM_AINV_GMRES_BJACOBI_MF) */
} ifem_timing;
int ifem_get_timing(ifem_ctx *ctx, ifem_timing *t);
/* Per-kernel-family log of one profiled step -- the reference's TimerOutput sections ("Assemble system", "CG for Mp",
 * "CG for Sm", "MUMPS for A_inv", "Solve linear system": mpi_insim.cpp:33,70,87,125,155,368) at kernel granularity.
 * ifem_kprof_begin starts recording: every launch wrapper of the library then brackets its launches with an event pair on the
 * context stream (all multigrid levels log into the context they hang below); nothing waits for the device until
 * ifem_kprof_end, which stops recording, waits once, and sums per family: scopes logged, milliseconds, and the ALGORITHMIC
 * bytes / flops the wrappers state for their launches (DESIGN.md section 4 gives the formulas).  Returns the number of
 * families written (<= max_entries, families without a launch are skipped). */
#define IFEM_KC_ASSEMBLE 0      /* the cell kernel (k_ins_assemble3 / k_ins_assemble2) */
#define IFEM_KC_ZERO_FILL 1     /* system_matrix = 0, system_rhs = 0 (mpi_insim.cpp:163-165) */
#define IFEM_KC_SPMV_UU 2       /* k_spmv_uu_pipe / k_spmv_uu: the stored fp64 A_uu of the outer operator */
#define IFEM_KC_SPMV_BBT 3      /* k_spmv_planar on B and B^T (fp64) */
#define IFEM_KC_MF_CELL 4       /* k_apply_uu_mf2: matrix-free A_uu, cell kernel */
#define IFEM_KC_MF_GATHER 5     /* k_mf_gather: node gather of the matrix-free product (+ fused smoother update) */
#define IFEM_KC_SPMV_SM 6       /* k_spmv_planar on S_m */
#define IFEM_KC_SPMV_MP 7       /* k_spmv_planar on M_p */
#define IFEM_KC_MDOT 8          /* k_mdot<K>: fused dot products */
#define IFEM_KC_MAXPY 9         /* k_maxpy<K>: fused multi-axpy */
#define IFEM_KC_VECTOR 10       /* axpy / axpby / scale / copy / convert / Chebyshev updates */
#define IFEM_KC_MG_TRANSFER 11  /* k_mg_csr*: prolongation / restriction / injection */
#define IFEM_KC_SMOOTHER_SETUP 12 /* k_uu_diag, k_bjac_setup, k_block_invert, eigenvalue bounds */
#define IFEM_KC_CG_RECURRENCE 13 /* k_cgd_*: device-resident CG recurrences */
#define IFEM_KC_SCHUR_SETUP 14  /* geometry blocks, k_schur_numeric */
#define IFEM_KC_OTHER 15        /* constraints, hanging nodes, halo packing */
#define IFEM_KC_TPP 16          /* SCnsIM: k_tpp_numeric, the ILU(0) of T_pp (factorisation, level-scheduled triangular solves), its SpMV */
#define IFEM_KC_COUNT 17
typedef struct {
  int32_t family;   /* IFEM_KC_* */
  uint32_t scopes;  /* launch-wrapper calls logged (a scope may hold two launches, e.g. the two stages of a reduction) */
  double ms;        /* device time between the scopes' event pairs, summed */
  double bytes;     /* algorithmic bytes, summed (0: not stated for this family) */
  double flops;     /* algorithmic flops, summed */
} ifem_kprof_entry;
/* how often the V-cycles of this context (A_uu and S_m) were captured into a hipGraph and how often a captured graph was launched
 * (ifem_tuning::vcycle_graph_cells) */
int ifem_vcycle_graph_stats(ifem_ctx *ctx, uint64_t *captures, uint64_t *launches);
/* the restart length the inner GMRES of IFEM_AINV_MG has lengthened itself to on this context (0: never -- ifem_solver_opts::inner_restart
 * is in use): an application that needs more than two restart cycles doubles it, up to 128 columns and a quarter of the free memory */
int ifem_inner_restart_length(ifem_ctx *ctx);
int ifem_kprof_begin(ifem_ctx *ctx);
int ifem_kprof_end(ifem_ctx *ctx, ifem_kprof_entry *out, int32_t max_entries);
const char *ifem_kprof_family_name(int32_t family);
/* on != 0: time every A_uu SpMV launch with HIP events on the context stream (one sync per launch) */
int ifem_set_profiling(ifem_ctx *ctx, int on);
/* block until everything queued on the context's stream (and on its halo-exchange stream) has finished: the bracket of
   a timed region.  The reference needs no counterpart (its PETSc calls are synchronous). */
int ifem_synchronize(ifem_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
