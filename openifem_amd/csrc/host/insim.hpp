// insim.hpp -- host-side mirror of the reference interface for the fluid step:
//   Utils::Time                      include/utilities.h, source/utilities.cpp:6-36
//   Fluid::MPI::FluidSolver<dim>     include/mpi_fluid_solver.h:89-151   (run, run_one_step, setup_dofs,
//                                    make_constraints, initialize_system, get_current_solution, hooks)
//   Fluid::MPI::InsIM<dim>           include/mpi_insim.h:35-99           (assemble, solve, run_one_step, run)
// Same member names, argument meaning and error behaviour (exceptions with the reference's messages); the
// PETSc matrices/vectors are replaced by one ifem_ctx (device) reached only through the C ABI of ifem_hip.h.
#pragma once
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../../include/ifem_hip.h"
#include "grid.hpp"
#include "multigrid.hpp"
#include "parameters.hpp"

namespace ifem_host {
namespace Utils {
class Time {
public:
  Time(double time_end, double delta_t, double output_interval, double refinement_interval, double save_interval)
      : timestep(0), time_current(0.0), time_end(time_end), delta_t(delta_t), output_interval(output_interval),
        refinement_interval(refinement_interval), save_interval(save_interval) {}
  double current() const { return time_current; }
  double end() const { return time_end; }
  double get_delta_t() const { return delta_t; }
  unsigned int get_timestep() const { return timestep; }
  bool time_to_output() const;
  bool time_to_refine() const;
  bool time_to_save() const;
  void increment();
  void decrement();
  void set_delta_t(double delta) { delta_t = delta; }

private:
  unsigned int timestep;
  double time_current, time_end, delta_t;
  const double output_interval, refinement_interval, save_interval;
};
// Utils::PVDWriter (include/utilities.h, source/utilities.cpp:38-81): the .pvd collection of the written time steps
class PVDWriter {
public:
  PVDWriter(const Time &t, const std::string &filename = "solution.pvd");
  void write_current_timestep(const std::string &pvtu_prefix, unsigned n_digits_for_counter);

private:
  const Time *time;
  std::string name, entries;
};
} // namespace Utils

namespace Fluid {
namespace MPI {

// one .vtu piece (linear patch per cell, the reference's field names); free function so that it is testable without a device
template <int dim>
void write_vtu_piece(const std::string &filename, const DoFTables<dim> &dofs, const std::vector<double> &solution,
                     const std::vector<double> &fsi_acc, const std::vector<double> &stress, const std::vector<int32_t> &indicator,
                     int subdomain, std::vector<std::string> *names_out);

struct SolverFailure : std::runtime_error {
  int code;
  SolverFailure(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};

template <int dim>
class FluidSolver {
public:
  using Point = std::array<double, dim>;
  FluidSolver(Triangulation<dim> &, const Parameters::AllParameters &, int device = 0);
  virtual ~FluidSolver();
  virtual void run() = 0;
  void add_hard_coded_boundary_condition(const int id,
                                         const std::function<double(const Point &, const unsigned int, const double)> &);
  void set_initial_condition(const std::function<double(const Point &, const unsigned int)> &);
  // SCnsIM inputs evaluated at the quadrature points (mpi_fluid_solver.cpp:82-103, mpi_scnsim.cpp:188-197)
  void set_body_force(const std::function<double(const Point &, const unsigned int)> &);
  void set_sigma_pml_field(const std::function<double(const Point &, const unsigned int)> &);
  // projected nodal viscous stress [dim][dim][n_unodes] of the present solution (mpi_fluid_solver.cpp:716-811)
  std::vector<double> update_stress();
  // FluidSolver::output_results (mpi_fluid_solver.cpp:491-579): fluid_<index>.<rank>.vtu (+ .pvtu and fluid.pvd on rank 0)
  void output_results(const unsigned int output_index);
  // FluidSolver::save_checkpoint / load_checkpoint (mpi_fluid_solver.cpp:582-713): <step>.fluid_checkpoint (+ .info,
  // _fixed.data) in output_dir; run() tries load_checkpoint() first and run_one_step saves whenever time.time_to_save()
  void save_checkpoint(const int output_index);
  bool load_checkpoint();
  std::string output_dir = "./";
  bool output_enabled = false; // run_one_step writes results at step 0 and whenever time.time_to_output() (off by default)
  // block vector [velocity | pressure] (PETScWrappers::MPI::BlockVector get_current_solution())
  std::vector<double> get_current_solution() const;
  std::pair<size_t, size_t> dofs_per_block_sizes() const { return {(size_t)dofs.n_u(), (size_t)dofs.n_pnodes}; }
  // Multi-GPU: this process is rank `rank` of a Px x Py x Pz block partition of the (box) triangulation
  // (p4est Morton partition in the reference, include/mpi_fluid_solver.h:187).  Exactly one transport is set:
  // nccl_unique_id (128 bytes, RCCL) or local_world (validation transport, see ifem_hip.h).
  void set_partition(const std::array<int, 3> &P, int rank, const uint8_t *nccl_unique_id, void *local_world);
  const PartitionTables &partition() const { return part; }
  // node/cell ordering used by setup_dofs: Morton curve (default, cache-friendly) or lexicographic
  void set_node_order(bool morton) { dofs.morton = morton; }
  ifem_ctx *context() const { return ctx; }
  unsigned int current_timestep() const { return time.get_timestep(); }
  double current_time() const { return time.current(); }
  const DoFTables<dim> &dof_tables() const { return dofs; }
  // the hanging-node lines of a locally refined triangulation (empty otherwise), block numbering [u | p]
  const HangingLines &hanging_lines() const { return hanging; }
  const Triangulation<dim> &get_triangulation() const { return triangulation; }
  void constraint_lines(std::vector<int32_t> &d, std::vector<double> &v) const { d = constraint_dofs; v = nonzero_values; }
  std::ostream *pcout = &std::cout; // ConditionalOStream on rank 0; nullptr silences
  // Geometric multigrid levels for the preconditioner's inner solves.  When the triangulation is a box
  // and the formulation uses them (InsIM, InsIMEX), initialize_system() builds the chain of coarser box meshes
  // (multigrid.hpp::next_coarser_level), sets each up as a solver of the same problem on the same partition and hangs it
  // below this context with ifem_mg_attach.  The reference needs none: MUMPS (mpi_insim.cpp:124-127).
  bool multigrid = true;
  int mg_min_cells = 4; // a direction is halved only while it keeps this many cells per rank
  // several ranks: a coarser level whose WHOLE mesh has at most this many cells is not partitioned but REPLICATED -- a single-rank
  // solver of the whole coarse mesh on every rank (ifem_mg_attach's replicated coarse level), which then builds its own chain below
  // it without any communication; 0 keeps every level partitioned
  int64_t mg_replica_cells = 32768;
  // validation transport only (set_partition with local_world): the worlds of the coarser levels, finest coarse level
  // first -- every level's virtual ranks meet in a world of their own; the chain ends where the list does
  std::vector<void *> mg_local_worlds;
  // the coarser levels hanging below this solver, finest first (borrowed pointers; each level owns the next)
  std::vector<const FluidSolver<dim> *> multigrid_levels() const;

  virtual void run_one_step(bool apply_nonzero_constraints, bool assemble_system = true) = 0;
  void setup_dofs();
  void make_constraints();
  virtual void initialize_system();
  // counters of the most recent run_one_step (the reference prints them per iteration: " ITR = .. GMRES_ITR = ..")
  unsigned int last_newton_iterations = 0, last_fgmres_iterations = 0;

protected:
  void check(int rc, const char *what) const;
  // the same formulation on another triangulation (a coarser multigrid level); nullptr: this family keeps no levels
  virtual std::unique_ptr<FluidSolver<dim>> make_level_solver(Triangulation<dim> &) const { return nullptr; }
  // builds the next coarser level (recursively the whole chain) and attaches it; false when there is none
  bool attach_multigrid_levels();
  bool attach_nested_levels(); // unstructured meshes with a refinement history (cylinder under refine_global), single rank
  std::unique_ptr<Triangulation<dim>> mg_tria;        // declared before mg_coarse: destroyed after it
  std::unique_ptr<FluidSolver<dim>> mg_coarse;        // the next coarser level (its context borrows this one's stream)
  // refine_mesh (mpi_fluid_solver.cpp:418-488, called from run_one_step when time_to_refine() fires in a pure-fluid run,
  // mpi_insim.cpp:485-489) is not part of the host mirror: stop with a message instead of computing on another mesh
  void refine_mesh_not_supported() const;
  Triangulation<dim> &triangulation;
  Parameters::AllParameters parameters;
  DoFTables<dim> dofs;
  HangingLines hanging;
  std::vector<size_t> dofs_per_block;
  std::map<int, std::function<double(const Point &, const unsigned int, const double)>> hard_coded_boundary_values;
  std::shared_ptr<std::function<double(const Point &, const unsigned int)>> initial_condition_field;
  std::shared_ptr<std::function<double(const Point &, const unsigned int)>> body_force, sigma_pml_field;
  // the time the hard coded boundary Fields carry (Field::advance_time): never advanced by InsIM::run
  // (mpi_insim.cpp:493-519), advanced by dt before every step of SUPGFluidSolver::run (mpi_supg_solver.cpp:438-480)
  double field_time = 0.0;
  Utils::Time time;
  std::unique_ptr<Utils::PVDWriter> pvd_writer;
  ifem_ctx *ctx = nullptr;
  int device;
  std::vector<int32_t> constraint_dofs;
  std::vector<double> nonzero_values;
  PartitionTables part;
  std::array<int, 3> proc_grid{1, 1, 1};
  int part_rank = 0;
  std::vector<uint8_t> nccl_id;
  void *local_world = nullptr;
};

template <int dim>
class InsIM : public FluidSolver<dim> {
public:
  InsIM(Triangulation<dim> &, const Parameters::AllParameters &, int device = 0);
  void run() override;
  void run_one_step(bool apply_nonzero_constraints, bool assemble_system = true) override;
  void initialize_system() override;
  // exposed for tests/bench (private in the reference, mpi_insim.h:66,75)
  void assemble(const bool use_nonzero_constraints);
  std::pair<unsigned int, double> solve(const bool use_nonzero_constraints);
  ifem_solver_opts solver_opts;
  ifem_solve_stats last_stats{};
  ifem_ins_params ins_params() const;

protected:
  std::unique_ptr<FluidSolver<dim>> make_level_solver(Triangulation<dim> &t) const override {
    return std::unique_ptr<FluidSolver<dim>>(new InsIM<dim>(t, parameters, this->device));
  }

private:
  using FluidSolver<dim>::parameters;
  using FluidSolver<dim>::time;
  using FluidSolver<dim>::ctx;
  using FluidSolver<dim>::check;
};

// Fluid::MPI::InsIMEX<dim> (include/mpi_insimex.h, source/mpi_insimex.cpp): implicit-explicit scheme, no Newton loop;
// the matrix is assembled in the first two steps only (run(), :455-470) and the right-hand side every step
template <int dim>
class InsIMEX : public FluidSolver<dim> {
public:
  InsIMEX(Triangulation<dim> &, const Parameters::AllParameters &, int device = 0);
  void run() override;
  void run_one_step(bool apply_nonzero_constraints, bool assemble_system = true) override;
  void initialize_system() override;
  void assemble(bool use_nonzero_constraints, bool assemble_system);
  std::pair<unsigned int, double> solve(bool use_nonzero_constraints, bool assemble_system);
  void assemble(bool use_nonzero_constraints) { assemble(use_nonzero_constraints, true); }
  std::pair<unsigned int, double> solve(bool use_nonzero_constraints) { return solve(use_nonzero_constraints, true); }
  ifem_solver_opts solver_opts;
  ifem_solve_stats last_stats{};
  ifem_ins_params ins_params() const;

protected:
  std::unique_ptr<FluidSolver<dim>> make_level_solver(Triangulation<dim> &t) const override {
    return std::unique_ptr<FluidSolver<dim>>(new InsIMEX<dim>(t, parameters, this->device));
  }

private:
  using FluidSolver<dim>::parameters;
  using FluidSolver<dim>::time;
  using FluidSolver<dim>::ctx;
  using FluidSolver<dim>::check;
};

// Fluid::MPI::SUPGFluidSolver<dim> (include/mpi_supg_solver.h:39-104): Newton loop + FGMRES of the stabilised
// solvers; assemble() is the hook of the derived formulation.
template <int dim>
class SUPGFluidSolver : public FluidSolver<dim> {
public:
  SUPGFluidSolver(Triangulation<dim> &, const Parameters::AllParameters &, int device = 0);
  void run() override;
  void run_one_step(bool apply_nonzero_constraints, bool assemble_system = true) override;
  void initialize_system() override;
  virtual void assemble(const bool use_nonzero_constraints) = 0;
  std::pair<unsigned int, double> solve(const bool use_nonzero_constraints);
  ifem_solver_opts solver_opts;
  ifem_solve_stats last_stats{};

protected:
  // evaluates body_force / sigma_pml_field at the quadrature points and uploads them (once per initialize_system)
  void upload_fields();
  using FluidSolver<dim>::parameters;
  using FluidSolver<dim>::time;
  using FluidSolver<dim>::ctx;
  using FluidSolver<dim>::check;
};

// Fluid::MPI::SUPGInsIM<dim> (include/mpi_insim_supg.h:26-60, source/mpi_insim_supg.cpp:15-327): incompressible NS with
// SUPG / PSPG / LSIC on equal-order elements
template <int dim>
class SUPGInsIM : public SUPGFluidSolver<dim> {
public:
  SUPGInsIM(Triangulation<dim> &, const Parameters::AllParameters &, int device = 0);
  void assemble(const bool use_nonzero_constraints) override;

private:
  using FluidSolver<dim>::parameters;
  using FluidSolver<dim>::time;
  using FluidSolver<dim>::ctx;
  using FluidSolver<dim>::check;
};

// Fluid::MPI::SCnsIM<dim> (include/mpi_scnsim.h, source/mpi_scnsim.cpp:15-568)
template <int dim>
class SCnsIM : public SUPGFluidSolver<dim> {
public:
  SCnsIM(Triangulation<dim> &, const Parameters::AllParameters &, int device = 0);
  void assemble(const bool use_nonzero_constraints) override;
  ifem_scns_params scns_params() const;

private:
  using FluidSolver<dim>::parameters;
  using FluidSolver<dim>::time;
  using FluidSolver<dim>::ctx;
  using FluidSolver<dim>::check;
};

} // namespace MPI
} // namespace Fluid

namespace Utils {
// The timed state of the benchmark channel [0,L]x[0,H](x[0,W]): present = analytic plane Poiseuille of the
// pressure-driven channel, evaluation point = present + a seeded perturbation keyed by the GLOBAL dof id (uniform in
// +-rel*Umax on unconstrained velocity dofs, +-rel*dP on pressure), both uploaded; Dirichlet dofs keep the present values.
template <int dim>
void channel_bench_state(Fluid::MPI::FluidSolver<dim> &solver, double L = 2.0, double H = 0.2, double dP = 10.0, double mu = 1.0,
                         uint64_t seed = 1234, double rel = 1e-3);
} // namespace Utils
} // namespace ifem_host
