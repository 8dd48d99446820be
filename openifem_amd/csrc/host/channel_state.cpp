// channel_state.cpp -- the timed state of the benchmark channel for any driver of the
// host mirror: tests, bench.py (through the facade) and tests/cpp/channel3d.cpp.
#include <random>
#include "insim.hpp"

namespace ifem_host {
namespace Utils {

template <int dim>
void channel_bench_state(Fluid::MPI::FluidSolver<dim> &solver, double L, double H, double dP, double mu, uint64_t seed, double rel) {
  auto &d = solver.dof_tables();
  const int64_t n_u = d.n_u(), n = d.n_dofs();
  std::vector<double> present((size_t)n, 0.0), ev;
  const double umax = dP * H * H / (8 * mu * L);
  for (int64_t nd = 0; nd < d.n_unodes; ++nd) {
    const double y = d.unode_coords[nd][1];
    present[nd * dim] = dP / (2 * mu * L) * y * (H - y);
  }
  for (int64_t nd = 0; nd < d.n_pnodes; ++nd) present[n_u + nd] = dP * (1.0 - d.pnode_coords[nd][0] / L);
  ev = present;
  // perturbation keyed by the GLOBAL dof so that every partition of the mesh sees the same field:
  // one mt19937_64(seed) draw sequence would depend on the local numbering
  auto &pt = solver.partition();
  // (a stateless hash of (seed, key) -- splitmix64 -- not a generator per dof: seeding 53 M Mersenne twisters took 20 s at 128^3)
  auto unit = [&](uint64_t key) {
    uint64_t z = seed + (key + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return double(z >> 11) * (2.0 / 9007199254740992.0) - 1.0; // 53 bits -> [-1, 1)
  };
  for (int64_t nd = 0; nd < d.n_unodes; ++nd)
    for (int c = 0; c < dim; ++c) ev[nd * dim + c] += rel * umax * unit((uint64_t)pt.l2g_u[nd] * dim + c);
  for (int64_t nd = 0; nd < d.n_pnodes; ++nd) ev[n_u + nd] += rel * dP * unit((uint64_t)(dim * pt.n_unodes_global + pt.l2g_p[nd]));
  // constrained dofs keep the boundary values
  std::vector<int32_t> cd;
  std::vector<double> cv;
  solver.constraint_lines(cd, cv);
  for (size_t k = 0; k < cd.size(); ++k) ev[cd[k]] = present[cd[k]];
  if (ifem_vec_set(solver.context(), IFEM_VEC_PRESENT, present.data()) < 0 ||
      ifem_vec_set(solver.context(), IFEM_VEC_EVAL, ev.data()) < 0)
    throw std::runtime_error(ifem_last_error());
}
template void channel_bench_state<2>(Fluid::MPI::FluidSolver<2> &, double, double, double, double, uint64_t, double);
template void channel_bench_state<3>(Fluid::MPI::FluidSolver<3> &, double, double, double, double, uint64_t, double);

} // namespace Utils
} // namespace ifem_host
