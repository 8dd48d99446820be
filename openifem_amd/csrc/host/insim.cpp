#include "insim.hpp"
#include <cmath>
#include <iomanip>

namespace ifem_host {
namespace Utils {
bool Time::time_to_output() const {
  auto delta = static_cast<unsigned int>(output_interval / delta_t);
  return (timestep >= delta && timestep % delta == 0);
}
bool Time::time_to_refine() const {
  auto delta = static_cast<unsigned int>(refinement_interval / delta_t);
  return (timestep >= delta && timestep % delta == 0);
}
bool Time::time_to_save() const {
  auto delta = static_cast<unsigned int>(save_interval / delta_t);
  return (timestep >= delta && timestep % delta == 0);
}
void Time::increment() { time_current += delta_t; ++timestep; }
void Time::decrement() { time_current -= delta_t; --timestep; }
} // namespace Utils

namespace Fluid {
namespace MPI {

template <int dim>
FluidSolver<dim>::FluidSolver(Triangulation<dim> &tria, const Parameters::AllParameters &parameters, int device)
    : triangulation(tria), parameters(parameters),
      time(parameters.end_time, parameters.time_step, parameters.output_interval, parameters.refinement_interval,
           parameters.save_interval),
      device(device) {}

template <int dim>
FluidSolver<dim>::~FluidSolver() {
  mg_coarse.reset(); // coarser levels first: their contexts run on this context's stream
  if (ctx) ifem_ctx_destroy(ctx);
}

template <int dim>
std::vector<const FluidSolver<dim> *> FluidSolver<dim>::multigrid_levels() const {
  std::vector<const FluidSolver<dim> *> out;
  for (const FluidSolver<dim> *s = mg_coarse.get(); s; s = s->mg_coarse.get()) out.push_back(s);
  return out;
}

// The chain of coarser levels below this solver: the next coarser box mesh of multigrid.hpp, the same
// formulation / parameters / boundary conditions / partition on it (a full solver: it makes its own constraints and
// builds ITS coarser level in its initialize_system), the nodal transfers between the two lattices, ifem_mg_attach.
template <int dim>
bool FluidSolver<dim>::attach_multigrid_levels() {
  mg_coarse.reset();
  mg_tria.reset();
  if (multigrid && ctx && !triangulation.is_box) return attach_nested_levels();
  if (!multigrid || !triangulation.is_box || triangulation.locally_refined || !ctx) return false;
  std::array<int, 3> n{1, 1, 1}, next;
  std::array<double, 3> extent{1, 1, 1};
  for (int d = 0; d < dim; ++d) {
    n[d] = triangulation.reps[d] / proc_grid[d];
    extent[d] = triangulation.p1[d] - triangulation.p0[d];
  }
  const int nranks = proc_grid[0] * proc_grid[1] * proc_grid[2];
  bool have_next = next_coarser_level(dim, n, proc_grid, extent, mg_min_cells, next);
  // several ranks: a small enough coarse mesh is replicated (one single-rank solver of the whole mesh per rank) instead of partitioned
  bool replica = false;
  std::array<int, 3> next_global{1, 1, 1};
  if (nranks > 1 && mg_replica_cells > 0) {
    bool cand = have_next;
    if (have_next) {
      for (int d = 0; d < dim; ++d) next_global[d] = next[d] * proc_grid[d];
    } else { // the blocks cannot be halved any more: the whole mesh may still be
      std::array<int, 3> whole{1, 1, 1};
      for (int d = 0; d < dim; ++d) whole[d] = triangulation.reps[d];
      cand = next_coarser_level(dim, whole, {1, 1, 1}, extent, mg_min_cells, next_global);
    }
    int64_t cells = 1;
    for (int d = 0; d < dim; ++d) cells *= next_global[d];
    replica = cand && cells <= mg_replica_cells;
  }
  if (!replica && !have_next) return false;
  // validation transport: every partitioned level rendezvous in a world of its own, handed in by the caller; none left = chain ends
  void *level_world = nullptr;
  if (local_world && !replica) {
    if (mg_local_worlds.empty()) return false;
    level_world = mg_local_worlds.front();
  }
  mg_tria.reset(new Triangulation<dim>());
  std::vector<unsigned> reps(dim);
  std::array<double, dim> a, b;
  for (int d = 0; d < dim; ++d) {
    reps[d] = (unsigned)(replica ? next_global[d] : next[d] * proc_grid[d]);
    a[d] = triangulation.p0[d]; b[d] = triangulation.p1[d];
  }
  GridGenerator::subdivided_hyper_rectangle<dim>(*mg_tria, reps, a, b, triangulation.colorized, /*lazy=*/true);
  std::unique_ptr<FluidSolver<dim>> c = make_level_solver(*mg_tria);
  if (!c) { mg_tria.reset(); return false; }
  c->pcout = nullptr;
  c->multigrid = true;
  c->mg_min_cells = mg_min_cells;
  c->dofs.morton = dofs.morton;
  c->hard_coded_boundary_values = hard_coded_boundary_values;
  c->field_time = field_time;
  c->mg_replica_cells = mg_replica_cells;
  if (replica) {
    c->set_partition({1, 1, 1}, 0, nullptr, nullptr);
  } else {
    c->set_partition(proc_grid, part_rank, nccl_id.empty() ? nullptr : nccl_id.data(), level_world);
    c->mg_local_worlds.assign(mg_local_worlds.begin() + (mg_local_worlds.empty() ? 0 : 1), mg_local_worlds.end());
  }
  c->setup_dofs();
  c->make_constraints();
  c->initialize_system(); // recursion: attaches the levels below c
  // transfers between the two node lattices (owned fine rows, local coarse columns)
  const int kv = dofs.kv;
  std::array<int, 3> rf{1, 1, 1}, rc{1, 1, 1};
  for (int d = 0; d < dim; ++d) { rf[d] = triangulation.reps[d]; rc[d] = mg_tria->reps[d]; }
  CsrTransfer Pp, Rp, Pu, Ru;
  box_prolongation(dim, rf, rc, 1, part.l2g_p.data(), dofs.n_pnodes_owned, c->part.l2g_p.data(), c->dofs.n_pnodes, Pp);
  transpose_transfer(Pp, Rp);
  box_prolongation(dim, rf, rc, kv, part.l2g_u.data(), dofs.n_unodes_owned, c->part.l2g_u.data(), c->dofs.n_unodes, Pu);
  transpose_transfer(Pu, Ru);
  const std::vector<int32_t> inj = box_injection(dim, rf, rc, kv, c->part.l2g_u.data(), c->dofs.n_unodes_owned,
                                                 part.l2g_u.data(), dofs.n_unodes_owned, /*allow_missing=*/replica);
  ifem_mg_transfer t{};
  t.n_fine_p_owned = Pp.n_rows; t.n_coarse_p_local = Pp.n_cols;
  t.pp_ptr = Pp.ptr.data(); t.pp_col = Pp.col.data(); t.pp_w = Pp.w.data();
  t.rp_ptr = Rp.ptr.data(); t.rp_col = Rp.col.data(); t.rp_w = Rp.w.data();
  t.n_fine_u_owned = Pu.n_rows; t.n_coarse_u_local = Pu.n_cols;
  t.pu_ptr = Pu.ptr.data(); t.pu_col = Pu.col.data(); t.pu_w = Pu.w.data();
  t.ru_ptr = Ru.ptr.data(); t.ru_col = Ru.col.data(); t.ru_w = Ru.w.data();
  t.inj_u = inj.data();
  check(ifem_mg_attach(ctx, c->ctx, &t), "attach_multigrid_levels");
  mg_coarse = std::move(c);
  return true;
}

// Unstructured meshes that know their refinement history (Utils::GridCreator::flow_around_cylinder under refine_global,
// source/utilities.cpp:345-570, mpi_insim.cpp:493-519): the level chain IS the history -- the same generator one level down, the same
// formulation / boundary conditions on it, transfers from the parent-child tables (host/multigrid.cpp::nested_prolongation).
// Several ranks: the strips of partition_unstructured are cut per level and their ghost layers do not cover each other's transfers, so
// the coarser meshes are not partitioned at all: the next level is a REPLICATED single-rank solver of the whole coarser mesh on every rank
// (ifem_mg_attach's replicated coarse level; these meshes are small), with the rows of the global transfer tables this rank owns.
template <int dim>
bool FluidSolver<dim>::attach_nested_levels() {
  if (!triangulation.generator || !triangulation.parent_of || triangulation.level < 1) return false;
  if (triangulation.locally_refined) return false;
  const bool replica = proc_grid[0] * proc_grid[1] * proc_grid[2] > 1;
  if (replica && mg_replica_cells == 0) return false;
  mg_tria.reset(new Triangulation<dim>());
  mg_tria->generator = triangulation.generator;
  mg_tria->parent_of = triangulation.parent_of;
  mg_tria->generator(*mg_tria, triangulation.level - 1);
  std::unique_ptr<FluidSolver<dim>> c = make_level_solver(*mg_tria);
  if (!c) { mg_tria.reset(); return false; }
  c->pcout = nullptr;
  c->multigrid = true;
  c->mg_min_cells = mg_min_cells;
  c->dofs.morton = dofs.morton;
  c->hard_coded_boundary_values = hard_coded_boundary_values;
  c->field_time = field_time;
  c->mg_replica_cells = mg_replica_cells;
  c->setup_dofs(); // a level solver starts as a single rank: the replica of the whole coarser mesh when this level is partitioned
  c->make_constraints();
  c->initialize_system(); // recursion: attaches the levels below c
  // the tables of the WHOLE fine mesh in the generator's numbering (= this solver's own on one rank; rebuilt on several: these meshes
  // are small and every rank built them once already in setup_dofs)
  DoFTables<dim> whole_tables;
  PartitionTables whole_part;
  if (replica) distribute_dofs_unstructured<dim>(triangulation, (int)parameters.fluid_velocity_degree, whole_tables, whole_part);
  const DoFTables<dim> &gt = replica ? whole_tables : dofs;
  const size_t ncf = gt.cell_unodes.size() / gt.nu, ncc = c->dofs.cell_unodes.size() / c->dofs.nu;
  std::vector<size_t> parent(ncf);
  std::vector<int> offset(ncf);
  for (size_t k = 0; k < ncf; ++k) {
    triangulation.parent_of(triangulation.level, k, parent[k], offset[k]);
    if (parent[k] >= ncc) throw std::logic_error("attach_nested_levels: parent cell out of range");
  }
  CsrTransfer Pp, Rp, Pu, Ru;
  nested_prolongation(dim, 1, gt.cell_pnodes.data(), ncf, gt.n_pnodes, c->dofs.cell_pnodes.data(), c->dofs.n_pnodes, parent, offset, Pp);
  nested_prolongation(dim, gt.kv, gt.cell_unodes.data(), ncf, gt.n_unodes, c->dofs.cell_unodes.data(), c->dofs.n_unodes, parent, offset, Pu);
  std::vector<int32_t> inj = nested_injection(dim, gt.kv, gt.cell_unodes.data(), ncf, c->dofs.cell_unodes.data(), ncc,
                                              c->dofs.n_unodes, parent, offset);
  if (replica) { // keep the rows of the nodes this rank owns (local numbering: owned first); the columns are the replica's nodes already
    auto owned_rows = [](const CsrTransfer &P, const std::vector<int64_t> &l2g, int64_t n_owned) {
      CsrTransfer Q;
      Q.n_rows = n_owned; Q.n_cols = P.n_cols;
      Q.ptr.assign((size_t)n_owned + 1, 0);
      for (int64_t i = 0; i < n_owned; ++i) {
        const int64_t g = l2g[(size_t)i];
        if (g < 0 || g >= P.n_rows) throw std::logic_error("attach_nested_levels: global node id out of range");
        Q.ptr[(size_t)i + 1] = Q.ptr[(size_t)i] + (P.ptr[(size_t)g + 1] - P.ptr[(size_t)g]);
      }
      Q.col.reserve((size_t)Q.ptr[(size_t)n_owned]); Q.w.reserve((size_t)Q.ptr[(size_t)n_owned]);
      for (int64_t i = 0; i < n_owned; ++i) {
        const int64_t g = l2g[(size_t)i];
        Q.col.insert(Q.col.end(), P.col.begin() + P.ptr[(size_t)g], P.col.begin() + P.ptr[(size_t)g + 1]);
        Q.w.insert(Q.w.end(), P.w.begin() + P.ptr[(size_t)g], P.w.begin() + P.ptr[(size_t)g + 1]);
      }
      return Q;
    };
    Pp = owned_rows(Pp, part.l2g_p, dofs.n_pnodes_owned);
    Pu = owned_rows(Pu, part.l2g_u, dofs.n_unodes_owned);
    std::vector<int32_t> g2l((size_t)gt.n_unodes, -1); // owned velocity nodes only: exactly one rank injects a coarse node
    for (int64_t i = 0; i < dofs.n_unodes_owned; ++i) g2l[(size_t)part.l2g_u[(size_t)i]] = (int32_t)i;
    for (auto &v : inj) v = g2l[(size_t)v];
  }
  transpose_transfer(Pp, Rp);
  transpose_transfer(Pu, Ru);
  ifem_mg_transfer t{};
  t.n_fine_p_owned = Pp.n_rows; t.n_coarse_p_local = Pp.n_cols;
  t.pp_ptr = Pp.ptr.data(); t.pp_col = Pp.col.data(); t.pp_w = Pp.w.data();
  t.rp_ptr = Rp.ptr.data(); t.rp_col = Rp.col.data(); t.rp_w = Rp.w.data();
  t.n_fine_u_owned = Pu.n_rows; t.n_coarse_u_local = Pu.n_cols;
  t.pu_ptr = Pu.ptr.data(); t.pu_col = Pu.col.data(); t.pu_w = Pu.w.data();
  t.ru_ptr = Ru.ptr.data(); t.ru_col = Ru.col.data(); t.ru_w = Ru.w.data();
  t.inj_u = inj.data();
  check(ifem_mg_attach(ctx, c->ctx, &t), "attach_nested_levels");
  mg_coarse = std::move(c);
  return true;
}

template <int dim>
void FluidSolver<dim>::check(int rc, const char *what) const {
  if (rc < 0) throw SolverFailure(rc, std::string(what) + ": " + ifem_last_error());
}

template <int dim>
void FluidSolver<dim>::refine_mesh_not_supported() const {
  if (parameters.simulation_type == "Fluid" && time.time_to_refine())
    throw std::runtime_error("refine_mesh: the reference refines the fluid mesh at this step (Refinement interval reached); "
                             "adaptive refinement is not supported by the HIP host mirror -- raise 'Refinement interval' "
                             "above 'End time' (hanging-node lines of an externally refined mesh go through "
                             "ifem_set_hanging_constraints)");
}

template <int dim>
void FluidSolver<dim>::add_hard_coded_boundary_condition(
    const int id, const std::function<double(const Point &, const unsigned int, const double)> &value_function) {
  if (parameters.fluid_dirichlet_bcs.find(id) == parameters.fluid_dirichlet_bcs.end())
    throw std::invalid_argument("Hard coded BC ID not included in parameters file!");
  if (!hard_coded_boundary_values.insert({id, value_function}).second)
    throw std::invalid_argument("Duplicated hard coded boundary conditions!");
}

template <int dim>
void FluidSolver<dim>::set_initial_condition(const std::function<double(const Point &, const unsigned int)> &condition) {
  initial_condition_field.reset(new std::function<double(const Point &, const unsigned int)>(condition));
}

template <int dim>
void FluidSolver<dim>::set_body_force(const std::function<double(const Point &, const unsigned int)> &bf) {
  body_force.reset(new std::function<double(const Point &, const unsigned int)>(bf));
}

template <int dim>
void FluidSolver<dim>::set_sigma_pml_field(const std::function<double(const Point &, const unsigned int)> &pml) {
  sigma_pml_field.reset(new std::function<double(const Point &, const unsigned int)>(pml));
}

template <int dim>
std::vector<double> FluidSolver<dim>::update_stress() {
  std::vector<double> s((size_t)(dim * dim) * (size_t)dofs.n_unodes);
  check(ifem_update_stress(ctx, parameters.viscosity, s.data()), "update_stress");
  return s;
}

template <int dim>
void FluidSolver<dim>::set_partition(const std::array<int, 3> &P, int rank, const uint8_t *nccl_unique_id, void *world) {
  proc_grid = P;
  part_rank = rank;
  nccl_id.clear();
  if (nccl_unique_id) nccl_id.assign(nccl_unique_id, nccl_unique_id + 128);
  local_world = world;
}

template <int dim>
void FluidSolver<dim>::setup_dofs() {
  if (!triangulation.is_box) {
    const int nranks = proc_grid[0] * proc_grid[1] * proc_grid[2];
    distribute_dofs_unstructured<dim>(triangulation, (int)parameters.fluid_velocity_degree, dofs, part);
    if (nranks > 1) { // every rank builds the (small) global tables and keeps its strip
      const DoFTables<dim> global = dofs;
      partition_unstructured<dim>(global, nranks, part_rank, dofs, part);
    }
  } else if (triangulation.locally_refined) { // one level of local refinement (execute_coarsening_and_refinement)
    const int nranks = proc_grid[0] * proc_grid[1] * proc_grid[2];
    const bool morton = dofs.morton;
    distribute_dofs_refined_box<dim>(triangulation, (int)parameters.fluid_velocity_degree, dofs, part, hanging);
    dofs.morton = morton;
    if (nranks > 1) { // round 4: every rank builds the global tables and lines (these meshes are small) and keeps its strip, the
                      // masters of its hanging nodes included (the reference runs these meshes on >= 2 ranks, fsi_leaflet_mpi.cpp:65-76)
      const DoFTables<dim> global = dofs;
      const HangingLines global_lines = hanging;
      partition_unstructured<dim>(global, nranks, part_rank, dofs, part, &global_lines, &hanging);
      dofs.morton = morton;
    }
  } else {
    hanging.clear();
    distribute_dofs_box<dim>(triangulation.reps, triangulation.p0, triangulation.p1, triangulation.colorized,
                             (int)parameters.fluid_velocity_degree, proc_grid, part_rank, dofs, part);
  }
  dofs_per_block = {(size_t)(dim * part.n_unodes_global), (size_t)part.n_pnodes_global};
  if (this->pcout)
    *this->pcout << "   Number of active fluid cells: " << triangulation.n_active_cells() << std::endl
                 << "   Number of degrees of freedom: " << dofs_per_block[0] + dofs_per_block[1] << " (" << dofs_per_block[0]
                 << '+' << dofs_per_block[1] << ')' << std::endl;
}

template <int dim>
void FluidSolver<dim>::make_constraints() {
  std::map<int, std::function<double(const Point &, unsigned)>> hc;
  const double t = field_time;
  for (auto &kv : hard_coded_boundary_values) {
    auto f = kv.second;
    hc[kv.first] = [f, t](const Point &p, unsigned c) { return f(p, c, t); };
  }
  // hanging-node lines first (mpi_fluid_solver.cpp:182-184), then the boundary values, which skip dofs that carry a line
  make_dirichlet<dim>(dofs, parameters.fluid_dirichlet_bcs, hc, constraint_dofs, nonzero_values, hanging.dof.empty() ? nullptr : &hanging.dof);
  if (ctx) {
    check(ifem_set_constraints(ctx, 1, (int32_t)constraint_dofs.size(), constraint_dofs.data(), nonzero_values.data()), "make_constraints");
    check(ifem_set_constraints(ctx, 0, (int32_t)constraint_dofs.size(), constraint_dofs.data(), nullptr), "make_constraints");
  }
  if (mg_coarse) { // the levels keep their constraint sets in step with this one (ifem_hip.h, ifem_mg_attach)
    mg_coarse->field_time = field_time;
    mg_coarse->make_constraints();
  }
}

template <int dim>
void FluidSolver<dim>::initialize_system() {
  if (ctx) { ifem_ctx_destroy(ctx); ctx = nullptr; }
  ifem_mesh_desc m{};
  m.dim = dim; m.kv = dofs.kv; m.n_cells = (int32_t)(dofs.cell_unodes.size() / dofs.nu);
  m.n_unodes_owned = (int32_t)dofs.n_unodes_owned; m.n_unodes_local = (int32_t)dofs.n_unodes;
  m.n_pnodes_owned = (int32_t)dofs.n_pnodes_owned; m.n_pnodes_local = (int32_t)dofs.n_pnodes;
  m.vcoords = dofs.vcoords.data(); m.cell_unodes = dofs.cell_unodes.data(); m.cell_pnodes = dofs.cell_pnodes.data();
  m.cell_face_bid = dofs.cell_face_bid.data();
  ifem_partition ip{};
  if (part.nranks > 1) {
    ip.rank = part.rank; ip.nranks = part.nranks; ip.n_neighbors = (int32_t)part.neighbors.size();
    ip.neighbor_rank = part.neighbors.data();
    ip.send_u_ptr = part.send_u_ptr.data(); ip.send_u_idx = part.send_u_idx.data(); ip.recv_u_ptr = part.recv_u_ptr.data();
    ip.send_p_ptr = part.send_p_ptr.data(); ip.send_p_idx = part.send_p_idx.data(); ip.recv_p_ptr = part.recv_p_ptr.data();
    ip.nccl_unique_id = nccl_id.empty() ? nullptr : nccl_id.data();
    ip.local_world = local_world;
    if (!part.sm_box_id.empty()) { // box meshes: 2-deep pressure halo for the explicit S_m
      for (int d = 0; d < 3; ++d) { ip.p_lattice_n[d] = part.p_lattice_n[d]; ip.sm_box_lo[d] = part.sm_box_lo[d]; ip.sm_box_n[d] = part.sm_box_n[d]; }
      ip.sm_box_id = part.sm_box_id.data();
      ip.l2g_p = part.l2g_p.data();
      ip.send_s_ptr = part.send_s_ptr.data(); ip.send_s_idx = part.send_s_idx.data(); ip.recv_s_ptr = part.recv_s_ptr.data();
    }
  }
  check(ifem_ctx_create(&m, part.nranks > 1 ? &ip : nullptr, device, &ctx), "initialize_system");
  // collective: a rank whose strip holds no hanging node still takes part in the exchanges of the lines of the others
  if (!hanging.dof.empty() || (triangulation.locally_refined && part.nranks > 1))
    check(ifem_set_hanging_constraints(ctx, (int32_t)hanging.dof.size(), hanging.dof.data(), hanging.ptr.data(), hanging.master.data(),
                                       hanging.weight.data()), "initialize_system");
  check(ifem_set_constraints(ctx, 1, (int32_t)constraint_dofs.size(), constraint_dofs.data(), nonzero_values.data()), "initialize_system");
  check(ifem_set_constraints(ctx, 0, (int32_t)constraint_dofs.size(), constraint_dofs.data(), nullptr), "initialize_system");
  if (initial_condition_field) { // apply_initial_condition (mpi_fluid_solver.cpp:367-415): nodal interpolation
    std::vector<double> x((size_t)dofs.n_dofs(), 0.0);
    for (int64_t nd = 0; nd < dofs.n_unodes; ++nd)
      for (int c = 0; c < dim; ++c) x[nd * dim + c] = (*initial_condition_field)(dofs.unode_coords[nd], c);
    for (int64_t nd = 0; nd < dofs.n_pnodes; ++nd) x[dofs.n_u() + nd] = (*initial_condition_field)(dofs.pnode_coords[nd], dim);
    check(ifem_vec_set(ctx, IFEM_VEC_PRESENT, x.data()), "apply_initial_condition");
  }
}

template <int dim>
std::vector<double> FluidSolver<dim>::get_current_solution() const {
  std::vector<double> x((size_t)dofs.n_dofs());
  check(ifem_vec_get(ctx, IFEM_VEC_PRESENT, x.data()), "get_current_solution");
  return x;
}

template <int dim>
InsIM<dim>::InsIM(Triangulation<dim> &tria, const Parameters::AllParameters &parameters, int device)
    : FluidSolver<dim>(tria, parameters, device) {
  if (parameters.fluid_velocity_degree - parameters.fluid_pressure_degree != 1)
    throw std::invalid_argument("Velocity finite element should be one order higher than pressure!");
  ifem_default_solver_opts(&solver_opts);
}

template <int dim>
void InsIM<dim>::initialize_system() {
  FluidSolver<dim>::initialize_system();
  // box meshes: multigrid levels for A~^-1 and CG(S_m); the measured best inner solver on them becomes the default
  // (matrix-free operator + V-cycle, restart 16 = one multi-dot pass of the single-precision basis)
  if (this->attach_multigrid_levels() && solver_opts.ainv_kind == IFEM_AINV_GMRES_BJACOBI) {
    solver_opts.ainv_kind = IFEM_AINV_MG;
    solver_opts.inner_restart = 16;
    // first preconditioner application of a velocity-dominated solve to 5e-5: ends the outer iteration at its first check on
    // fine meshes (128^3: one FGMRES iteration instead of two for the same four inner iterations); self-correcting where it
    // does not pay (ifem_solver_opts::inner_rel_first: the context backs off after a miss).  These are the values bench.py times.
    solver_opts.inner_rel_first = 5e-5;
  }
}

template <int dim>
ifem_ins_params InsIM<dim>::ins_params() const {
  ifem_ins_params p{};
  p.viscosity = parameters.viscosity; p.rho = parameters.fluid_rho; p.grad_div = parameters.grad_div;
  p.dt = time.get_delta_t();
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
void InsIM<dim>::assemble(const bool use_nonzero_constraints) {
  const ifem_ins_params p = ins_params();
  check(ifem_set_ainv_kind(ctx, solver_opts.ainv_kind), "assemble");
  check(ifem_ins_assemble(ctx, &p, use_nonzero_constraints), "assemble");
}

template <int dim>
std::pair<unsigned int, double> InsIM<dim>::solve(const bool use_nonzero_constraints) {
  const ifem_ins_params p = ins_params();
  check(ifem_solve(ctx, &p, &solver_opts, use_nonzero_constraints, &last_stats), "solve");
  return {last_stats.fgmres_iters, last_stats.fgmres_res};
}

template <int dim>
void InsIM<dim>::run_one_step(bool apply_nonzero_constraints, bool assemble_system) {
  static_cast<void>(assemble_system);
  if (this->output_enabled && time.get_timestep() == 0) this->output_results(0);
  time.increment();
  if (this->pcout)
    *this->pcout << std::string(96, '*') << std::endl
           << "Time step = " << time.get_timestep() << ", at t = " << std::scientific << time.current() << std::endl;
  double current_residual = 1.0, initial_residual = 1.0, relative_residual = 1.0;
  unsigned int outer_iteration = 0;
  this->last_newton_iterations = this->last_fgmres_iterations = 0;
  check(ifem_vec_copy(ctx, IFEM_VEC_EVAL, IFEM_VEC_PRESENT), "run_one_step"); // evaluation_point = present_solution
  while (relative_residual > parameters.fluid_tolerance && current_residual > 1e-11) {
    if (!(outer_iteration < parameters.fluid_max_iterations))
      throw SolverFailure(IFEM_E_NEWTON_MAXIT, "Too many Newton iterations!");
    check(ifem_vec_zero(ctx, IFEM_VEC_UPDATE), "run_one_step"); // newton_update = 0
    assemble(apply_nonzero_constraints && outer_iteration == 0);
    auto state = solve(apply_nonzero_constraints && outer_iteration == 0);
    check(ifem_rhs_norm(ctx, &current_residual), "run_one_step");
    check(ifem_vec_axpy(ctx, 1.0, IFEM_VEC_UPDATE, IFEM_VEC_EVAL), "run_one_step"); // evaluation_point += newton_update
    if (outer_iteration == 0) initial_residual = current_residual;
    relative_residual = current_residual / initial_residual;
    if (this->pcout)
      *this->pcout << std::scientific << std::left << " ITR = " << std::setw(2) << outer_iteration
             << " ABS_RES = " << current_residual << " REL_RES = " << relative_residual
             << " GMRES_ITR = " << std::setw(3) << state.first << " GMRES_RES = " << state.second << std::endl;
    outer_iteration++;
    this->last_newton_iterations = outer_iteration;
    this->last_fgmres_iterations += state.first;
  }
  // solution_increment = present - evaluation; present_solution = evaluation_point
  check(ifem_vec_copy(ctx, IFEM_VEC_INCREMENT, IFEM_VEC_PRESENT), "run_one_step");
  check(ifem_vec_axpy(ctx, -1.0, IFEM_VEC_EVAL, IFEM_VEC_INCREMENT), "run_one_step");
  check(ifem_vec_copy(ctx, IFEM_VEC_PRESENT, IFEM_VEC_EVAL), "run_one_step");
  // Update stress for output -- and for FSI::find_fluid_bc, which reads the projected stress (mpi_insim.cpp:474-475)
  check(ifem_update_stress(ctx, parameters.viscosity, nullptr), "update_stress");
  if (parameters.simulation_type == "Fluid" && time.time_to_save()) this->save_checkpoint((int)time.get_timestep()); // (:477-480)
  if (this->output_enabled && time.time_to_output()) this->output_results(time.get_timestep()); // (:481-484)
  this->refine_mesh_not_supported(); // (:485-489)
}

template <int dim>
void InsIM<dim>::run() {
  if (this->pcout) *this->pcout << "Running with HIP on " << this->proc_grid[0] * this->proc_grid[1] * this->proc_grid[2] << " MI355X rank(s)..." << std::endl;
  const bool success_load = this->load_checkpoint(); // try load from previous computation (:499-507)
  if (!success_load) {
    this->triangulation.refine_global(parameters.global_refinements[0]);
    this->setup_dofs();
    this->make_constraints();
    this->initialize_system();
  }
  run_one_step(true);
  while (time.end() - time.current() > 1e-12) run_one_step(false);
}

template <int dim>
InsIMEX<dim>::InsIMEX(Triangulation<dim> &tria, const Parameters::AllParameters &parameters, int device)
    : FluidSolver<dim>(tria, parameters, device) {
  if (parameters.fluid_velocity_degree - parameters.fluid_pressure_degree != 1)
    throw std::invalid_argument("Velocity finite element should be one order higher than pressure!");
  ifem_default_solver_opts(&solver_opts);
  solver_opts.inner_rel = 1e-4; // CG for A: max(1e-12, 1e-4 ||.||)  (mpi_insimex.cpp:117-118)
}

template <int dim>
void InsIMEX<dim>::initialize_system() {
  FluidSolver<dim>::initialize_system();
  // box meshes: CG(S_m) is multigrid-preconditioned, and the measured best A~^-1 of this symmetric operator becomes the default
  // (InsIMEX: exactly one V-cycle on the matrix-free A_uu -- 16 outer iterations x 58 ms against 13 x 80 ms with
  // two inner Krylov steps at 128^3 -- and the velocity block of the OUTER operator applied matrix-free in fp64: the same operator to
  // 1e-12, a fifth of the stored block's time, 16 times per time step)
  if (this->attach_multigrid_levels() && solver_opts.ainv_kind == IFEM_AINV_GMRES_BJACOBI) {
    solver_opts.ainv_kind = IFEM_AINV_MG;
    solver_opts.inner_restart = 16;
    solver_opts.inner_rel = 1e-2;
    solver_opts.inner_maxit = 0;
    solver_opts.outer_matrix_free = 1;
  }
}

template <int dim>
ifem_ins_params InsIMEX<dim>::ins_params() const {
  ifem_ins_params p{};
  p.viscosity = parameters.viscosity; p.rho = parameters.fluid_rho; p.grad_div = parameters.grad_div;
  p.dt = time.get_delta_t();
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
void InsIMEX<dim>::assemble(bool use_nonzero_constraints, bool assemble_system) {
  const ifem_ins_params p = ins_params();
  check(ifem_imex_assemble(ctx, &p, use_nonzero_constraints, assemble_system), "assemble");
}

template <int dim>
std::pair<unsigned int, double> InsIMEX<dim>::solve(bool use_nonzero_constraints, bool assemble_system) {
  static_cast<void>(assemble_system); // the preconditioner data (S_m) is cached by the context until the matrix changes
  const ifem_ins_params p = ins_params();
  check(ifem_imex_solve(ctx, &p, &solver_opts, use_nonzero_constraints, &last_stats), "solve");
  return {last_stats.fgmres_iters, last_stats.fgmres_res};
}

template <int dim>
void InsIMEX<dim>::run_one_step(bool apply_nonzero_constraints, bool assemble_system) {
  if (this->output_enabled && time.get_timestep() == 0) this->output_results(0); // (mpi_insimex.cpp:402-405)
  time.increment();
  if (this->pcout)
    *this->pcout << std::string(96, '*') << std::endl
                 << "Time step = " << time.get_timestep() << ", at t = " << std::scientific << time.current() << std::endl;
  check(ifem_vec_zero(ctx, IFEM_VEC_UPDATE), "run_one_step"); // solution_time_increment = 0
  assemble(apply_nonzero_constraints, assemble_system);
  auto state = solve(apply_nonzero_constraints, assemble_system);
  check(ifem_vec_axpy(ctx, 1.0, IFEM_VEC_UPDATE, IFEM_VEC_PRESENT), "run_one_step"); // present_solution += increment
  if (this->pcout)
    *this->pcout << std::scientific << std::left << " GMRES_ITR = " << std::setw(3) << state.first
                 << " GMRES_RES = " << state.second << std::endl;
  check(ifem_update_stress(ctx, parameters.viscosity, nullptr), "update_stress");
  if (parameters.simulation_type == "Fluid" && time.time_to_save()) this->save_checkpoint((int)time.get_timestep()); // (mpi_insimex.cpp:433-436)
  if (this->output_enabled && time.time_to_output()) this->output_results(time.get_timestep());
  this->refine_mesh_not_supported(); // (mpi_insimex.cpp:438-442; the reference also re-assembles on that trigger, :416-418)
}

template <int dim>
void InsIMEX<dim>::run() {
  if (this->pcout) *this->pcout << "Running with HIP on " << this->proc_grid[0] * this->proc_grid[1] * this->proc_grid[2] << " MI355X rank(s)..." << std::endl;
  const bool success_load = this->load_checkpoint(); // (mpi_insimex.cpp:455-463)
  if (!success_load) {
    this->triangulation.refine_global(parameters.global_refinements[0]);
    this->setup_dofs();
    this->make_constraints();
    this->initialize_system();
  }
  // the left-hand side is assembled in the first two steps, and again after a restart (:466-472)
  while (time.end() - time.current() > 1e-12) run_one_step(time.get_timestep() == 0, time.get_timestep() < 2 || success_load);
}

template class InsIMEX<2>;
template class InsIMEX<3>;
template class FluidSolver<2>;
template class FluidSolver<3>;
template class InsIM<2>;
template class InsIM<3>;

} // namespace MPI
} // namespace Fluid
} // namespace ifem_host
