// channel3d.cpp -- a C++ caller of the host mirror for BASELINE config 3 (3D channel flow, Q2/Q1, mpi_insim), shaped like
// the reference's test drivers (tests/fluid_cylinder_mpi/fluid_cylinder_mpi.cpp:19-105, tests/fluid_pressure_driven):
// AllParameters(prm) + subdivided_hyper_rectangle + Fluid::MPI::InsIM<3>(tria, params), linked against libifem_hip.so.
// Nothing here builds a multigrid hierarchy: InsIM<3>::initialize_system() does (csrc/host/insim.cpp).
//
//   channel3d <prm> run                      flow.run() to the .prm's end time, then the Poiseuille check of
//                                            tests/fluid_pressure_driven (vmax = dP H^2 / (8 mu L) = 2.5e-2)
//   channel3d <prm> bench <cells> <K> <W>    the timed state of bench.py: K Newton steps (assemble + solve) after W
//                                            warm-up steps; prints one JSON line (ms per step, iteration counts, true
//                                            residual of the last solve)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include "insim.hpp"

using namespace ifem_host;

int main(int argc, char *argv[]) {
  try {
    std::string infile("parameters.prm");
    if (argc > 1) infile = argv[1];
    const std::string mode = argc > 2 ? argv[2] : "run";
    Parameters::AllParameters params(infile);
    if (params.dimension != 3) throw std::runtime_error("This test should be run in 3D!");
    const unsigned cells = argc > 3 ? (unsigned)std::atoi(argv[3]) : 16;
    const int steps = argc > 4 ? std::atoi(argv[4]) : 3, warmup = argc > 5 ? std::atoi(argv[5]) : 1;
    const double L = 2.0, H = 0.2;

    Triangulation<3> tria;
    GridGenerator::subdivided_hyper_rectangle<3>(tria, {cells, cells, cells}, {0, 0, 0}, {L, H, H}, true, /*lazy=*/true);
    Fluid::MPI::InsIM<3> flow(tria, params);

    if (mode == "run") {
      flow.run();
      // Check the max value of the velocity (tests/fluid_pressure_driven/fluid_pressure_driven.cpp:40-47)
      double vmin = 0, vmax = 0;
      if (ifem_vec_minmax(flow.context(), IFEM_VEC_PRESENT, 0, &vmin, &vmax) < 0) throw std::runtime_error(ifem_last_error());
      const double expected = 10.0 * H * H / (8 * params.viscosity * L);
      const double verror = std::abs(vmax - expected) / expected;
      std::printf("{\"mode\": \"run\", \"vmax\": %.10g, \"expected\": %.10g, \"rel_error\": %.3e, \"multigrid_levels\": %d, "
                  "\"ainv_kind\": %d}\n", vmax, expected, verror, ifem_mg_depth(flow.context()), flow.solver_opts.ainv_kind);
      if (!(verror < 1e-3)) throw std::runtime_error("Maximum velocity is incorrect!");
      return 0;
    }

    flow.pcout = nullptr;
    flow.setup_dofs();
    flow.make_constraints();
    flow.initialize_system();
    // no knobs: bench.py times exactly what initialize_system leaves in solver_opts (argv[6]: experiment override of inner_rel_first)
    if (argc > 6) flow.solver_opts.inner_rel_first = std::atof(argv[6]);
    Utils::channel_bench_state<3>(flow);
    for (int i = 0; i < warmup; ++i) { flow.assemble(false); flow.solve(false); }
    if (ifem_synchronize(flow.context()) < 0) throw std::runtime_error(ifem_last_error());
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < steps; ++i) { flow.assemble(false); flow.solve(false); }
    if (ifem_synchronize(flow.context()) < 0) throw std::runtime_error(ifem_last_error());
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() / steps;
    double res = 0, rhs = 0;
    if (ifem_true_residual(flow.context(), &res, &rhs) < 0) throw std::runtime_error(ifem_last_error());
    const auto sizes = flow.dofs_per_block_sizes();
    const double n_dofs = double(sizes.first + sizes.second);
    const ifem_solve_stats &st = flow.last_stats;
    std::printf("{\"mode\": \"bench\", \"cells\": %u, \"n_dofs\": %.0f, \"ms_per_step\": %.3f, \"dofs_per_s\": %.6g, \"fgmres_iters\": %u, "
                "\"inner_iters\": %u, \"cg_mp_iters\": %u, \"cg_sm_iters\": %u, \"true_rel_residual\": %.6e, \"multigrid_levels\": %d, "
                "\"ainv_kind\": %d, \"inner_restart\": %d, \"inner_rel\": %g, \"inner_rel_first\": %g}\n",
                cells, n_dofs, ms, n_dofs / (ms * 1e-3), st.fgmres_iters, st.inner_iters, st.cg_mp_iters, st.cg_sm_iters, res / rhs,
                ifem_mg_depth(flow.context()), flow.solver_opts.ainv_kind, flow.solver_opts.inner_restart, flow.solver_opts.inner_rel, flow.solver_opts.inner_rel_first);
  } catch (std::exception &exc) {
    std::cerr << std::endl << "Exception on processing: " << std::endl << exc.what() << std::endl << "Aborting!" << std::endl;
    return 1;
  }
  return 0;
}
