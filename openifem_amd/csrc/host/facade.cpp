// facade.cpp -- flat C entry points over the C++ host mirror, so that tests/bench (ctypes) can drive
// Fluid::MPI::InsIM<dim> exactly the way the reference's test drivers do (tests/*/*.cpp).
#include <cstring>
#include <random>
#include <sstream>
#include "insim.hpp"
#include "multigrid.hpp"

using namespace ifem_host;

namespace {
thread_local std::string g_err;
struct Handle {
  int dim;
  std::unique_ptr<Triangulation<2>> t2;
  std::unique_ptr<Triangulation<3>> t3;
  std::unique_ptr<Fluid::MPI::FluidSolver<2>> s2; // InsIM or SCnsIM
  std::unique_ptr<Fluid::MPI::FluidSolver<3>> s3;
  std::ostringstream log;
};
// the formulation named by the driver: "InsIM" (mpi_insim.h) or "SCnsIM" (mpi_scnsim.h)
template <int dim>
std::unique_ptr<Fluid::MPI::FluidSolver<dim>> make_solver(const char *kind, Triangulation<dim> &t,
                                                          const Parameters::AllParameters &params, int device) {
  const std::string k = kind ? kind : "InsIM";
  if (k == "InsIM") return std::unique_ptr<Fluid::MPI::FluidSolver<dim>>(new Fluid::MPI::InsIM<dim>(t, params, device));
  if (k == "SCnsIM") return std::unique_ptr<Fluid::MPI::FluidSolver<dim>>(new Fluid::MPI::SCnsIM<dim>(t, params, device));
  if (k == "InsIMEX") return std::unique_ptr<Fluid::MPI::FluidSolver<dim>>(new Fluid::MPI::InsIMEX<dim>(t, params, device));
  if (k == "SUPGInsIM") return std::unique_ptr<Fluid::MPI::FluidSolver<dim>>(new Fluid::MPI::SUPGInsIM<dim>(t, params, device));
  throw std::invalid_argument("unknown fluid solver '" + k + "' (InsIM, InsIMEX, SCnsIM, SUPGInsIM)");
}
// assemble / solve / solver_opts live in the two solver families, not in FluidSolver (as in the reference)
template <int dim, class FI, class FS>
void with_family(Fluid::MPI::FluidSolver<dim> *s, FI fi, FS fs) {
  if (auto *a = dynamic_cast<Fluid::MPI::InsIM<dim> *>(s)) fi(*a);
  else if (auto *x = dynamic_cast<Fluid::MPI::InsIMEX<dim> *>(s)) fi(*x);
  else if (auto *b = dynamic_cast<Fluid::MPI::SUPGFluidSolver<dim> *>(s)) fs(*b);
  else throw std::logic_error("unknown solver family");
}
template <class F>
int guard(F f) {
  try { f(); return 0; }
  catch (const Fluid::MPI::SolverFailure &e) { g_err = e.what(); return e.code; }
  catch (const std::exception &e) { g_err = e.what(); return IFEM_E_BADPARAM; }
}
} // namespace

// the nested transfers of the cylinder meshes (multigrid.hpp::nested_prolongation / nested_injection) between refinement levels
// `level` and level - 1, checked on the host -- out[0]: max |row sum - 1| of P_u and P_p (partition of unity), out[1]: number of
// coarse velocity nodes c whose fine twin's row of P_u is not the unit vector e_c, out[2]: max error of a linear function at the
// fine velocity nodes that sit where the parent's d-linear map puts them (every patch but the curved ring), out[3]: their share
template <int dim>
static void nested_check(int level, int kv, double *out) {
  Triangulation<dim> tf, tc;
  Utils::GridCreator<dim>::flow_around_cylinder(tf);
  tf.refine_global(level);
  Utils::GridCreator<dim>::flow_around_cylinder(tc);
  tc.refine_global(level - 1);
  DoFTables<dim> df, dc;
  PartitionTables pf, pc;
  distribute_dofs_unstructured<dim>(tf, kv, df, pf);
  distribute_dofs_unstructured<dim>(tc, kv, dc, pc);
  const size_t ncf = df.cell_unodes.size() / df.nu, ncc = dc.cell_unodes.size() / dc.nu;
  std::vector<size_t> parent(ncf);
  std::vector<int> offset(ncf);
  for (size_t k = 0; k < ncf; ++k) tf.parent_of(level, k, parent[k], offset[k]);
  CsrTransfer Pu, Pp;
  nested_prolongation(dim, kv, df.cell_unodes.data(), ncf, df.n_unodes, dc.cell_unodes.data(), dc.n_unodes, parent, offset, Pu);
  nested_prolongation(dim, 1, df.cell_pnodes.data(), ncf, df.n_pnodes, dc.cell_pnodes.data(), dc.n_pnodes, parent, offset, Pp);
  const std::vector<int32_t> inj = nested_injection(dim, kv, df.cell_unodes.data(), ncf, dc.cell_unodes.data(), ncc, dc.n_unodes, parent, offset);
  double rs = 0;
  for (const CsrTransfer *P : {&Pu, &Pp})
    for (int64_t i = 0; i < P->n_rows; ++i) {
      double s = 0;
      for (int64_t k = P->ptr[(size_t)i]; k < P->ptr[(size_t)i + 1]; ++k) s += P->w[(size_t)k];
      rs = std::max(rs, std::fabs(s - 1.0));
    }
  int64_t bad = 0;
  for (int64_t c = 0; c < dc.n_unodes; ++c) {
    const int64_t i = inj[(size_t)c];
    bool ok = Pu.ptr[(size_t)i + 1] - Pu.ptr[(size_t)i] == 1 && Pu.col[(size_t)Pu.ptr[(size_t)i]] == c && std::fabs(Pu.w[(size_t)Pu.ptr[(size_t)i]] - 1.0) < 1e-13;
    bad += ok ? 0 : 1;
  }
  auto f = [](const std::array<double, dim> &x) { double v = 1.0; const double g[3] = {2.0, -3.0, 0.5}; for (int d = 0; d < dim; ++d) v += g[d] * x[d]; return v; };
  double err = 0;
  int64_t n_straight = 0;
  for (int64_t i = 0; i < df.n_unodes; ++i) {
    double v = 0;
    std::array<double, dim> xi{};
    for (int64_t k = Pu.ptr[(size_t)i]; k < Pu.ptr[(size_t)i + 1]; ++k) {
      v += Pu.w[(size_t)k] * f(dc.unode_coords[(size_t)Pu.col[(size_t)k]]);
      for (int d = 0; d < dim; ++d) xi[d] += Pu.w[(size_t)k] * dc.unode_coords[(size_t)Pu.col[(size_t)k]][d];
    }
    double dist = 0;
    for (int d = 0; d < dim; ++d) dist = std::max(dist, std::fabs(xi[d] - df.unode_coords[(size_t)i][d]));
    if (dist > 1e-12) continue; // a node the curved ring moved
    ++n_straight;
    err = std::max(err, std::fabs(v - f(df.unode_coords[(size_t)i])));
  }
  out[0] = rs; out[1] = double(bad); out[2] = err; out[3] = double(n_straight) / double(df.n_unodes);
}

extern "C" {

const char *ifemx_last_error(void) { return g_err.c_str(); }

// Equivalent of a reference test driver: AllParameters(prm) + subdivided_hyper_rectangle(reps, p0, p1, true)
// + InsIM<dim>(tria, params).  prm_text is the parameter file CONTENT.
int ifemx_solver_create_box(const char *kind, const char *prm_text, int dim, const unsigned *reps, const double *p0,
                            const double *p1, int device, int verbose, void **out) {
  return guard([&] {
    auto params = Parameters::AllParameters::from_string(prm_text);
    if (params.dimension != dim) throw std::invalid_argument("Dimension in the parameter file differs from the mesh");
    auto *h = new Handle();
    h->dim = dim;
    std::vector<unsigned> r(reps, reps + dim);
    if (dim == 2) {
      h->t2.reset(new Triangulation<2>());
      GridGenerator::subdivided_hyper_rectangle<2>(*h->t2, r, {p0[0], p0[1]}, {p1[0], p1[1]}, true, /*lazy=*/true);
      h->s2 = make_solver<2>(kind, *h->t2, params, device);
      h->s2->pcout = verbose ? &std::cout : nullptr;
    } else {
      h->t3.reset(new Triangulation<3>());
      GridGenerator::subdivided_hyper_rectangle<3>(*h->t3, r, {p0[0], p0[1], p0[2]}, {p1[0], p1[1], p1[2]}, true, /*lazy=*/true);
      h->s3 = make_solver<3>(kind, *h->t3, params, device);
      h->s3->pcout = verbose ? &std::cout : nullptr;
    }
    *out = h;
  });
}

// tests/fluid_cylinder_mpi driver: AllParameters(prm) + GridCreator<2>::flow_around_cylinder(tria) + InsIM<2>(tria, params)
int ifemx_solver_create_cylinder(const char *kind, const char *prm_text, int device, int verbose, void **out) {
  return guard([&] {
    auto params = Parameters::AllParameters::from_string(prm_text);
    auto *h = new Handle();
    h->dim = params.dimension == 3 ? 3 : 2;
    if (h->dim == 2) {
      h->t2.reset(new Triangulation<2>());
      Utils::GridCreator<2>::flow_around_cylinder(*h->t2);
      h->s2 = make_solver<2>(kind, *h->t2, params, device);
      h->s2->pcout = verbose ? &std::cout : nullptr;
    } else { // tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:98-104: the extruded mesh
      h->t3.reset(new Triangulation<3>());
      Utils::GridCreator<3>::flow_around_cylinder(*h->t3);
      h->s3 = make_solver<3>(kind, *h->t3, params, device);
      h->s3->pcout = verbose ? &std::cout : nullptr;
    }
    *out = h;
  });
}

// FluidSolver::add_hard_coded_boundary_condition(id, f): f(point[dim], component, time) -> value
typedef double (*ifemx_bc_fn)(const double *point, unsigned component, double time);
int ifemx_add_hard_coded_boundary_condition(void *hv, int id, ifemx_bc_fn fn) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) h->s2->add_hard_coded_boundary_condition(id, [fn](const std::array<double, 2> &p, unsigned c, double t) { return fn(p.data(), c, t); });
    else h->s3->add_hard_coded_boundary_condition(id, [fn](const std::array<double, 3> &p, unsigned c, double t) { return fn(p.data(), c, t); });
  });
}

int ifemx_insim_create_box(const char *prm_text, int dim, const unsigned *reps, const double *p0, const double *p1,
                           int device, int verbose, void **out) {
  return ifemx_solver_create_box("InsIM", prm_text, dim, reps, p0, p1, device, verbose, out);
}
int ifemx_insim_create_cylinder(const char *prm_text, int device, int verbose, void **out) {
  return ifemx_solver_create_cylinder("InsIM", prm_text, device, verbose, out);
}
// FluidSolver::set_body_force / set_sigma_pml_field: f(point[dim], component) -> value
typedef double (*ifemx_field_fn)(const double *point, unsigned component);
int ifemx_set_body_force(void *hv, ifemx_field_fn fn) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) h->s2->set_body_force([fn](const std::array<double, 2> &p, unsigned c) { return fn(p.data(), c); });
    else h->s3->set_body_force([fn](const std::array<double, 3> &p, unsigned c) { return fn(p.data(), c); });
  });
}
int ifemx_set_sigma_pml_field(void *hv, ifemx_field_fn fn) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) h->s2->set_sigma_pml_field([fn](const std::array<double, 2> &p, unsigned c) { return fn(p.data(), c); });
    else h->s3->set_sigma_pml_field([fn](const std::array<double, 3> &p, unsigned c) { return fn(p.data(), c); });
  });
}
int ifemx_set_initial_condition(void *hv, ifemx_field_fn fn) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) h->s2->set_initial_condition([fn](const std::array<double, 2> &p, unsigned c) { return fn(p.data(), c); });
    else h->s3->set_initial_condition([fn](const std::array<double, 3> &p, unsigned c) { return fn(p.data(), c); });
  });
}
// FluidSolver::update_stress: out[dim][dim][n_unodes]
int ifemx_update_stress(void *hv, double *out) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto s = h->dim == 2 ? h->s2->update_stress() : h->s3->update_stress();
    std::memcpy(out, s.data(), s.size() * sizeof(double));
  });
}

// FluidSolver::output_results(index) into directory `dir` (must end with '/')
int ifemx_output_results(void *hv, const char *dir, unsigned index) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { h->s2->output_dir = dir; h->s2->output_results(index); }
    else { h->s3->output_dir = dir; h->s3->output_results(index); }
  });
}
// FluidSolver::save_checkpoint(index) / load_checkpoint() in directory `dir` (must end with '/'); *found = 0 when the
// directory holds no checkpoint (the solver is then untouched: set it up as for a fresh start)
int ifemx_save_checkpoint(void *hv, const char *dir, int index) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { h->s2->output_dir = dir; h->s2->save_checkpoint(index); }
    else { h->s3->output_dir = dir; h->s3->save_checkpoint(index); }
  });
}
int ifemx_load_checkpoint(void *hv, const char *dir, int *found) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { h->s2->output_dir = dir; *found = h->s2->load_checkpoint(); }
    else { h->s3->output_dir = dir; *found = h->s3->load_checkpoint(); }
  });
}
// directory of output_results / checkpoints for run() (the reference uses the working directory)
int ifemx_set_output_dir(void *hv, const char *dir, int enable_output) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { h->s2->output_dir = dir; h->s2->output_enabled = enable_output != 0; }
    else { h->s3->output_dir = dir; h->s3->output_enabled = enable_output != 0; }
  });
}
// Utils::Time of the solver: current step and time
int ifemx_time(void *hv, unsigned *timestep, double *current) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { *timestep = h->s2->current_timestep(); *current = h->s2->current_time(); }
    else { *timestep = h->s3->current_timestep(); *current = h->s3->current_time(); }
  });
}
// the .vtu writer on host data only (no device): solution [n_dofs], optional fsi_acc [n_dofs], stress [dim][dim][n_unodes]
int ifemx_write_vtu(void *hv, const char *filename, const double *solution, const double *fsi_acc, const double *stress, int subdomain) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto run = [&](auto &s, auto dimtag) {
      constexpr int D = decltype(dimtag)::value;
      auto &d = s.dof_tables();
      std::vector<double> sol(solution, solution + d.n_dofs()), acc, st;
      if (fsi_acc) acc.assign(fsi_acc, fsi_acc + d.n_dofs());
      if (stress) st.assign(stress, stress + (size_t)D * D * d.n_unodes);
      Fluid::MPI::write_vtu_piece<D>(filename, d, sol, acc, st, {}, subdomain, nullptr);
    };
    if (h->dim == 2) run(*h->s2, std::integral_constant<int, 2>()); else run(*h->s3, std::integral_constant<int, 3>());
  });
}

void ifemx_destroy(void *hv) { delete static_cast<Handle *>(hv); }

// rank `rank` of a P[0] x P[1] x P[2] block partition; call before ifemx_setup.  Transport: nccl_unique_id (128 B)
// or local_world (ifem_local_world_create), the other NULL.
int ifemx_set_partition(void *hv, const int *P, int rank, const uint8_t *nccl_unique_id, void *local_world) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    std::array<int, 3> p{P[0], P[1], P[2]};
    if (h->dim == 2) h->s2->set_partition(p, rank, nccl_unique_id, local_world);
    else h->s3->set_partition(p, rank, nccl_unique_id, local_world);
  });
}
int ifemx_set_node_order(void *hv, int morton) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] { if (h->dim == 2) h->s2->set_node_order(morton != 0); else h->s3->set_node_order(morton != 0); });
}
// sizes of the partition tables: [n_unodes_owned, n_unodes_local, n_pnodes_owned, n_pnodes_local, n_neighbors,
//  n_send_u, n_send_p, n_unodes_global, n_pnodes_global, n_cells_local]
int ifemx_partition_sizes(void *hv, int64_t *out) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto fill = [&](auto &s) {
      auto &d = s.dof_tables(); auto &p = s.partition();
      out[0] = d.n_unodes_owned; out[1] = d.n_unodes; out[2] = d.n_pnodes_owned; out[3] = d.n_pnodes;
      out[4] = (int64_t)p.neighbors.size(); out[5] = (int64_t)p.send_u_idx.size(); out[6] = (int64_t)p.send_p_idx.size();
      out[7] = p.n_unodes_global; out[8] = p.n_pnodes_global; out[9] = (int64_t)(d.cell_unodes.size() / d.nu);
    };
    if (h->dim == 2) fill(*h->s2); else fill(*h->s3);
  });
}
int ifemx_partition_tables(void *hv, int64_t *l2g_u, int64_t *l2g_p, int32_t *neighbors, int32_t *send_u_ptr,
                           int32_t *send_u_idx, int32_t *recv_u_ptr, int32_t *send_p_ptr, int32_t *send_p_idx,
                           int32_t *recv_p_ptr) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto fill = [&](auto &s) {
      auto &p = s.partition();
      auto cp = [](auto &v, auto *dst) { if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
      cp(p.l2g_u, l2g_u); cp(p.l2g_p, l2g_p); cp(p.neighbors, neighbors);
      cp(p.send_u_ptr, send_u_ptr); cp(p.send_u_idx, send_u_idx); cp(p.recv_u_ptr, recv_u_ptr);
      cp(p.send_p_ptr, send_p_ptr); cp(p.send_p_idx, send_p_idx); cp(p.recv_p_ptr, recv_p_ptr);
    };
    if (h->dim == 2) fill(*h->s2); else fill(*h->s3);
  });
}

// 2-deep pressure halo plan of the distributed explicit S_m: out = [box_lo[3], box_n[3], lattice_n[3], n_send_s, n_far]
// (all zero when the plan is not available); tables sized accordingly
int ifemx_sm_plan_sizes(void *hv, int64_t *out) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto fill = [&](auto &s) {
      auto &p = s.partition();
      for (int d = 0; d < 3; ++d) { out[d] = p.sm_box_lo[d]; out[3 + d] = p.sm_box_n[d]; out[6 + d] = p.p_lattice_n[d]; }
      out[9] = (int64_t)p.send_s_idx.size();
      out[10] = p.recv_s_ptr.empty() ? 0 : p.recv_s_ptr.back();
    };
    if (h->dim == 2) fill(*h->s2); else fill(*h->s3);
  });
}
int ifemx_sm_plan_tables(void *hv, int32_t *box_id, int32_t *send_s_ptr, int32_t *send_s_idx, int32_t *recv_s_ptr) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto fill = [&](auto &s) {
      auto &p = s.partition();
      auto cp = [](auto &v, auto *dst) { if (!v.empty()) std::memcpy(dst, v.data(), v.size() * sizeof(v[0])); };
      cp(p.sm_box_id, box_id); cp(p.send_s_ptr, send_s_ptr); cp(p.send_s_idx, send_s_idx); cp(p.recv_s_ptr, recv_s_ptr);
    };
    if (h->dim == 2) fill(*h->s2); else fill(*h->s3);
  });
}

// The refinement loop of the reference's FSI drivers (tests/fsi_leaflet_mpi/fsi_leaflet_mpi.cpp:65-75): set_refine_flag on every
// coarse cell whose centre lies in [lo, hi] along direction `dir`, then execute_coarsening_and_refinement.  Before ifemx_setup.
int ifemx_refine_band(void *hv, int dir, double lo, double hi, int64_t *n_flagged) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto run = [&](auto &tria) {
      int64_t cnt = 0;
      for (size_t c = 0; c < tria.n_active_cells(); ++c) {
        const auto center = tria.cell_center(c);
        if (center[dir] >= lo && center[dir] <= hi) { tria.set_refine_flag(c); ++cnt; }
      }
      tria.execute_coarsening_and_refinement();
      if (n_flagged) *n_flagged = cnt;
    };
    if (dir < 0 || dir >= h->dim) throw std::invalid_argument("ifemx_refine_band: bad direction");
    if (h->dim == 2) run(*h->t2); else run(*h->t3);
  });
}
// hanging-node lines of the solver's DoF tables: sizes = [n_lines, n_entries]; then the arrays (dof [n], ptr [n + 1],
// master / weight [n_entries]); any output may be NULL
int ifemx_hanging_lines(void *hv, int64_t *sizes, int32_t *dof, int32_t *ptr, int32_t *master, double *weight) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto fill = [&](auto &s) {
      const HangingLines &L = s.hanging_lines();
      if (sizes) { sizes[0] = (int64_t)L.dof.size(); sizes[1] = (int64_t)L.master.size(); }
      if (dof && !L.dof.empty()) std::memcpy(dof, L.dof.data(), L.dof.size() * 4);
      if (ptr) std::memcpy(ptr, L.ptr.data(), L.ptr.size() * 4);
      if (master && !L.master.empty()) std::memcpy(master, L.master.data(), L.master.size() * 4);
      if (weight && !L.weight.empty()) std::memcpy(weight, L.weight.data(), L.weight.size() * 8);
    };
    if (h->dim == 2) fill(*h->s2); else fill(*h->s3);
  });
}
// FluidSolver::make_constraints(): the boundary lines re-made (MPI::FSI::run does this every time step, mpi_fsi.cpp:1191-1198);
// zero_inhomogeneities != 0: nonzero_constraints := copy of zero_constraints, as the FSI driver does after the first step
int ifemx_make_constraints(void *hv, int zero_inhomogeneities) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto run = [&](auto &s) {
      s.make_constraints();
      if (zero_inhomogeneities) {
        std::vector<int32_t> d; std::vector<double> v;
        s.constraint_lines(d, v);
        if (ifem_set_constraints(s.context(), 1, (int32_t)d.size(), d.data(), nullptr) < 0) throw std::runtime_error(ifem_last_error());
      }
    };
    if (h->dim == 2) run(*h->s2); else run(*h->s3);
  });
}

// Multigrid levels of the host mirror (insim.hpp, FluidSolver::multigrid): on / off, the smallest number of cells per rank a
// halved direction keeps, and -- validation transport only -- the local worlds of the coarser levels.  Before ifemx_setup.
int ifemx_set_multigrid(void *hv, int on, int min_cells, void *const *level_worlds, int n_worlds) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto set = [&](auto &s) {
      s.multigrid = on != 0;
      if (min_cells > 0) s.mg_min_cells = min_cells;
      s.mg_local_worlds.assign(level_worlds, level_worlds + (level_worlds ? n_worlds : 0));
    };
    if (h->dim == 2) set(*h->s2); else set(*h->s3);
  });
}
// FluidSolver::mg_replica_cells: coarse meshes of at most this many cells are replicated on every rank instead of partitioned
// (0: never).  Before ifemx_setup.
int ifemx_set_mg_replica_cells(void *hv, int64_t cells) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (cells < 0) throw std::invalid_argument("ifemx_set_mg_replica_cells: a count >= 0");
    if (h->dim == 2) h->s2->mg_replica_cells = cells; else h->s3->mg_replica_cells = cells;
  });
}
// the levels attached below the solver's context: *n_levels, their global repetitions reps[level][3] and contexts (each
// array may be NULL; room for 16 levels)
int ifemx_mg_levels(void *hv, int *n_levels, int32_t *reps, void **ctxs) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto fill = [&](auto &s) {
      auto lv = s.multigrid_levels();
      if (lv.size() > 16) throw std::runtime_error("ifemx_mg_levels: more than 16 levels");
      if (n_levels) *n_levels = (int)lv.size();
      for (size_t k = 0; k < lv.size(); ++k) {
        if (reps) for (int d = 0; d < 3; ++d) reps[3 * k + d] = lv[k]->get_triangulation().reps[d];
        if (ctxs) ctxs[k] = lv[k]->context();
      }
    };
    if (h->dim == 2) fill(*h->s2); else fill(*h->s3);
  });
}
// multigrid.hpp::coarse_level_chain: out[level][3] cells per rank (room for 16 levels), *n_levels
int ifemx_coarse_level_chain(int dim, const int *cells_per_rank, const int *P, const double *extent, int min_cells,
                             int32_t *out, int *n_levels) {
  return guard([&] {
    std::array<int, 3> n{1, 1, 1}, p{1, 1, 1};
    std::array<double, 3> e{1, 1, 1};
    for (int d = 0; d < dim; ++d) { n[d] = cells_per_rank[d]; p[d] = P[d]; e[d] = extent[d]; }
    auto chain = coarse_level_chain(dim, n, p, e, min_cells, 16);
    *n_levels = (int)chain.size();
    for (size_t k = 0; k < chain.size(); ++k) for (int d = 0; d < 3; ++d) out[3 * k + d] = chain[k][d];
  });
}
// multigrid.hpp::box_prolongation for the tests' checker.  Two calls: with col = NULL it fills ptr [n_fine + 1] only
// (the number of entries is ptr[n_fine]), with col / w it fills them too.
int ifemx_box_prolongation(int dim, const int *reps_fine, const int *reps_coarse, int degree, const int64_t *l2g_fine,
                           int64_t n_fine, const int64_t *l2g_coarse, int64_t n_coarse, int64_t *ptr, int32_t *col, double *w) {
  return guard([&] {
    std::array<int, 3> rf{1, 1, 1}, rc{1, 1, 1};
    for (int d = 0; d < dim; ++d) { rf[d] = reps_fine[d]; rc[d] = reps_coarse[d]; }
    CsrTransfer P;
    box_prolongation(dim, rf, rc, degree, l2g_fine, n_fine, l2g_coarse, n_coarse, P);
    std::memcpy(ptr, P.ptr.data(), P.ptr.size() * sizeof(int64_t));
    if (col) { std::memcpy(col, P.col.data(), P.col.size() * sizeof(int32_t)); std::memcpy(w, P.w.data(), P.w.size() * sizeof(double)); }
  });
}
int ifemx_box_injection(int dim, const int *reps_fine, const int *reps_coarse, int degree, const int64_t *l2g_coarse,
                        int64_t n_coarse, const int64_t *l2g_fine, int64_t n_fine, int32_t *out) {
  return guard([&] {
    std::array<int, 3> rf{1, 1, 1}, rc{1, 1, 1};
    for (int d = 0; d < dim; ++d) { rf[d] = reps_fine[d]; rc[d] = reps_coarse[d]; }
    auto inj = box_injection(dim, rf, rc, degree, l2g_coarse, n_coarse, l2g_fine, n_fine);
    std::memcpy(out, inj.data(), inj.size() * sizeof(int32_t));
  });
}

// ... below a replicated coarse level: -1 where the fine node under a coarse node is not in the (owned) fine list
int ifemx_box_injection_partial(int dim, const int *reps_fine, const int *reps_coarse, int degree, const int64_t *l2g_coarse,
                                int64_t n_coarse, const int64_t *l2g_fine, int64_t n_fine, int32_t *out) {
  return guard([&] {
    std::array<int, 3> rf{1, 1, 1}, rc{1, 1, 1};
    for (int d = 0; d < dim; ++d) { rf[d] = reps_fine[d]; rc[d] = reps_coarse[d]; }
    auto inj = box_injection(dim, rf, rc, degree, l2g_coarse, n_coarse, l2g_fine, n_fine, /*allow_missing=*/true);
    std::memcpy(out, inj.data(), inj.size() * sizeof(int32_t));
  });
}

int ifemx_nested_transfer_check(int dim, int level, int kv, double *out) {
  return guard([&] { if (dim == 2) nested_check<2>(level, kv, out); else nested_check<3>(level, kv, out); });
}

#define DISPATCH(h, expr2, expr3) (static_cast<Handle *>(h)->dim == 2 ? (expr2) : (expr3))

int ifemx_run(void *hv) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] { if (h->dim == 2) h->s2->run(); else h->s3->run(); });
}
// setup_dofs + make_constraints + initialize_system without running (refinements from the .prm applied)
int ifemx_setup(void *hv, int global_refinements) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { h->t2->refine_global(global_refinements); h->s2->setup_dofs(); h->s2->make_constraints(); h->s2->initialize_system(); }
    else { h->t3->refine_global(global_refinements); h->s3->setup_dofs(); h->s3->make_constraints(); h->s3->initialize_system(); }
  });
}
// host-only part of the set-up (no device needed): refine + setup_dofs + make_constraints
int ifemx_setup_host_only(void *hv, int global_refinements) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { h->t2->refine_global(global_refinements); h->s2->setup_dofs(); h->s2->make_constraints(); }
    else { h->t3->refine_global(global_refinements); h->s3->setup_dofs(); h->s3->make_constraints(); }
  });
}
// nonzero_constraints lines; call with dof = NULL to get the count
int ifemx_constraints(void *hv, int32_t *dof, double *val, int64_t *n) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    std::vector<int32_t> d; std::vector<double> v;
    if (h->dim == 2) h->s2->constraint_lines(d, v); else h->s3->constraint_lines(d, v);
    *n = (int64_t)d.size();
    if (dof) { std::memcpy(dof, d.data(), d.size() * 4); std::memcpy(val, v.data(), v.size() * 8); }
  });
}
int ifemx_run_one_step(void *hv, int apply_nonzero) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] { if (h->dim == 2) h->s2->run_one_step(apply_nonzero); else h->s3->run_one_step(apply_nonzero); });
}
// run_one_step(apply_nonzero_constraints, assemble_system): the second flag matters to InsIMEX only
int ifemx_run_one_step2(void *hv, int apply_nonzero, int assemble_system) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) h->s2->run_one_step(apply_nonzero, assemble_system); else h->s3->run_one_step(apply_nonzero, assemble_system);
  });
}
// counters of the solver's most recent solve (also the one inside run_one_step)
int ifemx_last_stats(void *hv, ifem_solve_stats *st) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (!st) throw std::runtime_error("ifemx_last_stats: null output");
    auto get = [&](auto &s) { *st = s.last_stats; };
    if (h->dim == 2) with_family<2>(h->s2.get(), get, get); else with_family<3>(h->s3.get(), get, get);
  });
}
// Newton iterations and summed FGMRES iterations of the solver's most recent run_one_step
int ifemx_last_newton(void *hv, int32_t *newton_iterations, int32_t *fgmres_iterations) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto get = [&](auto &s) {
      if (newton_iterations) *newton_iterations = int32_t(s.last_newton_iterations);
      if (fgmres_iterations) *fgmres_iterations = int32_t(s.last_fgmres_iterations);
    };
    if (h->dim == 2) with_family<2>(h->s2.get(), get, get); else with_family<3>(h->s3.get(), get, get);
  });
}
int ifemx_assemble(void *hv, int use_nonzero) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) with_family<2>(h->s2.get(), [&](auto &s) { s.assemble(use_nonzero); }, [&](auto &s) { s.assemble(use_nonzero); });
    else with_family<3>(h->s3.get(), [&](auto &s) { s.assemble(use_nonzero); }, [&](auto &s) { s.assemble(use_nonzero); });
  });
}
int ifemx_solve(void *hv, int use_nonzero, ifem_solve_stats *st) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto run = [&](auto &s) { s.solve(use_nonzero); if (st) *st = s.last_stats; };
    if (h->dim == 2) with_family<2>(h->s2.get(), run, run); else with_family<3>(h->s3.get(), run, run);
  });
}
ifem_solver_opts *ifemx_solver_opts(void *hv) {
  auto *h = static_cast<Handle *>(hv);
  ifem_solver_opts *o = nullptr;
  auto get = [&](auto &s) { o = &s.solver_opts; };
  if (h->dim == 2) with_family<2>(h->s2.get(), get, get); else with_family<3>(h->s3.get(), get, get);
  return o;
}
ifem_ctx *ifemx_ctx(void *hv) {
  auto *h = static_cast<Handle *>(hv);
  return h->dim == 2 ? h->s2->context() : h->s3->context();
}
int ifemx_sizes(void *hv, int64_t *n_cells, int64_t *n_u, int64_t *n_p) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) { auto &d = h->s2->dof_tables(); *n_cells = (int64_t)(d.cell_unodes.size() / d.nu); *n_u = d.n_u(); *n_p = d.n_pnodes; }
    else { auto &d = h->s3->dof_tables(); *n_cells = (int64_t)(d.cell_unodes.size() / d.nu); *n_u = d.n_u(); *n_p = d.n_pnodes; }
  });
}
int ifemx_get_solution(void *hv, double *out) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto x = h->dim == 2 ? h->s2->get_current_solution() : h->s3->get_current_solution();
    std::memcpy(out, x.data(), x.size() * sizeof(double));
  });
}
// support-point coordinates of velocity nodes [n_unodes][dim] and pressure nodes [n_pnodes][dim]
int ifemx_node_coords(void *hv, double *ucoords, double *pcoords) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) {
      auto &d = h->s2->dof_tables();
      std::memcpy(ucoords, d.unode_coords.data(), d.unode_coords.size() * 2 * sizeof(double));
      std::memcpy(pcoords, d.pnode_coords.data(), d.pnode_coords.size() * 2 * sizeof(double));
    } else {
      auto &d = h->s3->dof_tables();
      std::memcpy(ucoords, d.unode_coords.data(), d.unode_coords.size() * 3 * sizeof(double));
      std::memcpy(pcoords, d.pnode_coords.data(), d.pnode_coords.size() * 3 * sizeof(double));
    }
  });
}
// cell -> node tables (for cross-checks against the tests' independent builder)
int ifemx_cell_tables(void *hv, int32_t *cell_unodes, int32_t *cell_pnodes, int32_t *face_bid, double *vcoords) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    auto copy = [&](auto &d) {
      std::memcpy(cell_unodes, d.cell_unodes.data(), d.cell_unodes.size() * 4);
      std::memcpy(cell_pnodes, d.cell_pnodes.data(), d.cell_pnodes.size() * 4);
      std::memcpy(face_bid, d.cell_face_bid.data(), d.cell_face_bid.size() * 4);
      std::memcpy(vcoords, d.vcoords.data(), d.vcoords.size() * 8);
    };
    if (h->dim == 2) copy(h->s2->dof_tables()); else copy(h->s3->dof_tables());
  });
}

// The timed state of the bench: present = analytic plane Poiseuille for the pressure-driven channel
// [0,L]x[0,H](x[0,W]), evaluation point = present + seeded perturbation (mt19937_64(seed), uniform +-rel*Umax on
// unconstrained velocity dofs, +-rel*dP on pressure).  Uploads both; Dirichlet dofs keep the present values.
int ifemx_channel_state(void *hv, double L, double H, double dP, double mu, uint64_t seed, double rel) {
  auto *h = static_cast<Handle *>(hv);
  return guard([&] {
    if (h->dim == 2) Utils::channel_bench_state<2>(*h->s2, L, H, dP, mu, seed, rel);
    else Utils::channel_bench_state<3>(*h->s3, L, H, dP, mu, seed, rel);
  });
}

} // extern "C"
