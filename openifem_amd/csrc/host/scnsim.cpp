// scnsim.cpp -- host mirror of Fluid::MPI::SUPGFluidSolver<dim> (source/mpi_supg_solver.cpp:297-484) and
// Fluid::MPI::SCnsIM<dim> (source/mpi_scnsim.cpp:15-568) over the C ABI of ifem_hip.h.
#include <cmath>
#include <iomanip>
#include "../ctx.hpp"
#include "insim.hpp"

namespace ifem_host {
namespace Fluid {
namespace MPI {

template <int dim>
SUPGFluidSolver<dim>::SUPGFluidSolver(Triangulation<dim> &tria, const Parameters::AllParameters &parameters, int device)
    : FluidSolver<dim>(tria, parameters, device) {
  // mpi_supg_solver.cpp:213-215: equal-order elements only
  if (parameters.fluid_velocity_degree != parameters.fluid_pressure_degree)
    throw std::invalid_argument("Velocity finite element should be the same as pressure!");
  ifem_default_solver_opts(&solver_opts);
}

template <int dim>
void SUPGFluidSolver<dim>::upload_fields() {
  if (!this->body_force && !this->sigma_pml_field) return;
  const auto &d = this->dofs;
  ifem::FeTables fe;
  ifem::build_fe_tables(fe, dim, d.kv);
  constexpr int nv = 1 << dim;
  const size_t n_cells = d.cell_unodes.size() / d.nu;
  std::vector<double> pml, bf;
  if (this->sigma_pml_field) pml.resize(n_cells * fe.nq);
  if (this->body_force) bf.resize(n_cells * fe.nq * dim);
  for (size_t c = 0; c < n_cells; ++c)
    for (int q = 0; q < fe.nq; ++q) {
      std::array<double, dim> x{};
      for (int v = 0; v < nv; ++v)
        for (int e = 0; e < dim; ++e) x[e] += fe.psi[q * nv + v] * d.vcoords[(c * nv + v) * dim + e];
      if (this->sigma_pml_field) pml[c * fe.nq + q] = (*this->sigma_pml_field)(x, 0);
      if (this->body_force)
        for (int e = 0; e < dim; ++e) bf[(c * fe.nq + q) * dim + e] = (*this->body_force)(x, e);
    }
  check(ifem_set_scns_fields(ctx, pml.empty() ? nullptr : pml.data(), bf.empty() ? nullptr : bf.data(), nullptr), "initialize_system");
}

template <int dim>
void SUPGFluidSolver<dim>::initialize_system() {
  FluidSolver<dim>::initialize_system();
  upload_fields();
}

template <int dim>
std::pair<unsigned int, double> SUPGFluidSolver<dim>::solve(const bool use_nonzero_constraints) {
  check(ifem_scns_solve(ctx, &solver_opts, use_nonzero_constraints, &last_stats), "solve");
  return {last_stats.fgmres_iters, last_stats.fgmres_res};
}

template <int dim>
void SUPGFluidSolver<dim>::run_one_step(bool apply_nonzero_constraints, bool assemble_system) {
  static_cast<void>(assemble_system);
  time.increment();
  if (this->pcout)
    *this->pcout << std::string(96, '*') << std::endl
                 << "Time step = " << time.get_timestep() << ", at t = " << std::scientific << time.current() << std::endl;
  double current_residual = 1.0, initial_residual = 1.0, relative_residual = 1.0;
  unsigned int outer_iteration = 0;
  this->last_newton_iterations = this->last_fgmres_iterations = 0;
  check(ifem_vec_copy(ctx, IFEM_VEC_EVAL, IFEM_VEC_PRESENT), "run_one_step");
  while (relative_residual > parameters.fluid_tolerance && current_residual > 1e-14) {
    if (!(outer_iteration < parameters.fluid_max_iterations))
      throw SolverFailure(IFEM_E_NEWTON_MAXIT, "Too many Newton iterations!");
    check(ifem_vec_zero(ctx, IFEM_VEC_UPDATE), "run_one_step");
    assemble(apply_nonzero_constraints && outer_iteration == 0);
    auto state = solve(apply_nonzero_constraints && outer_iteration == 0);
    check(ifem_rhs_norm(ctx, &current_residual), "run_one_step");
    check(ifem_vec_axpy(ctx, 1.0, IFEM_VEC_UPDATE, IFEM_VEC_EVAL), "run_one_step");
    if (outer_iteration == 0) initial_residual = current_residual;
    relative_residual = current_residual / initial_residual;
    if (this->pcout)
      *this->pcout << std::scientific << std::left << " ITR = " << std::setw(2) << outer_iteration
                   << " ABS_RES = " << current_residual << " REL_RES = " << relative_residual
                   << " GMRES_ITR = " << std::setw(3) << state.first << " GMRES_RES = " << state.second
                   << " INNER_GMRES_ITR = " << std::setw(3) << last_stats.inner_iters << std::endl;
    outer_iteration++;
    this->last_newton_iterations = outer_iteration;
    this->last_fgmres_iterations += state.first;
  }
  check(ifem_vec_copy(ctx, IFEM_VEC_INCREMENT, IFEM_VEC_PRESENT), "run_one_step");
  check(ifem_vec_axpy(ctx, -1.0, IFEM_VEC_EVAL, IFEM_VEC_INCREMENT), "run_one_step");
  check(ifem_vec_copy(ctx, IFEM_VEC_PRESENT, IFEM_VEC_EVAL), "run_one_step");
  // update_stress feeds the next assemble (mpi_scnsim.cpp:178-186); 
  check(ifem_update_stress(ctx, parameters.viscosity, nullptr), "update_stress");
  if (parameters.simulation_type == "Fluid" && time.time_to_save()) this->save_checkpoint((int)time.get_timestep()); // (:416-419)
  if (this->output_enabled && time.time_to_output()) this->output_results(time.get_timestep());
  this->refine_mesh_not_supported(); // (mpi_supg_solver.cpp:420-424)
}

template <int dim>
void SUPGFluidSolver<dim>::run() {
  if (this->pcout) *this->pcout << "Running with HIP on 1 MI355X rank(s)..." << std::endl;
  // hard coded boundary Fields are advanced by dt before the first and every later step (:438-444, :470-480)
  const bool success_load = this->load_checkpoint(); // (:434-450); a restarted run goes straight into the time loop
  if (!success_load) {
    if (!this->hard_coded_boundary_values.empty()) this->field_time += time.get_delta_t();
    this->triangulation.refine_global(parameters.global_refinements[0]);
    this->setup_dofs();
    this->make_constraints();
    this->initialize_system();
    run_one_step(true);
  }
  while (time.end() - time.current() > 1e-12) {
    if (!this->hard_coded_boundary_values.empty()) {
      this->field_time += time.get_delta_t();
      this->make_constraints();
      run_one_step(true);
    } else
      run_one_step(false);
  }
}

template <int dim>
SCnsIM<dim>::SCnsIM(Triangulation<dim> &tria, const Parameters::AllParameters &parameters, int device)
    : SUPGFluidSolver<dim>(tria, parameters, device) {}

template <int dim>
ifem_scns_params SCnsIM<dim>::scns_params() const {
  ifem_scns_params p{};
  p.viscosity = parameters.viscosity; p.rho = parameters.fluid_rho; p.dt = time.get_delta_t();
  p.solid_rho = parameters.solid_rho;
  p.formulation = IFEM_FORM_SCNSIM;
  for (int i = 0; i < dim; ++i) p.gravity[i] = parameters.gravity[i];
  p.n_neumann = 0;
  if (parameters.n_fluid_neumann_bcs != 0)
    for (auto &kv : parameters.fluid_neumann_bcs) {
      if (p.n_neumann >= 8) throw std::invalid_argument("at most 8 Neumann boundaries are supported");
      p.neumann_id[p.n_neumann] = (int32_t)kv.first;
      p.neumann_p[p.n_neumann++] = kv.second;
    }
  return p;
}

template <int dim>
void SCnsIM<dim>::assemble(const bool use_nonzero_constraints) {
  const ifem_scns_params p = scns_params();
  check(ifem_scns_assemble(ctx, &p, use_nonzero_constraints), "assemble");
}

template <int dim>
SUPGInsIM<dim>::SUPGInsIM(Triangulation<dim> &tria, const Parameters::AllParameters &parameters, int device)
    : SUPGFluidSolver<dim>(tria, parameters, device) {}

template <int dim>
void SUPGInsIM<dim>::assemble(const bool use_nonzero_constraints) {
  ifem_scns_params p{};
  p.viscosity = parameters.viscosity; p.rho = parameters.fluid_rho; p.dt = time.get_delta_t();
  p.solid_rho = parameters.solid_rho;
  for (int i = 0; i < dim; ++i) p.gravity[i] = parameters.gravity[i];
  if (parameters.n_fluid_neumann_bcs != 0)
    for (auto &kv : parameters.fluid_neumann_bcs) {
      if (p.n_neumann >= 8) throw std::invalid_argument("at most 8 Neumann boundaries are supported");
      p.neumann_id[p.n_neumann] = (int32_t)kv.first;
      p.neumann_p[p.n_neumann++] = kv.second;
    }
  p.formulation = IFEM_FORM_SUPG_INSIM;
  check(ifem_scns_assemble(ctx, &p, use_nonzero_constraints), "assemble");
}

template class SUPGInsIM<2>;
template class SUPGInsIM<3>;
template class SUPGFluidSolver<2>;
template class SUPGFluidSolver<3>;
template class SCnsIM<2>;
template class SCnsIM<3>;

} // namespace MPI
} // namespace Fluid
} // namespace ifem_host
