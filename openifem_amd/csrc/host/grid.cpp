// grid.cpp -- see grid.hpp.
#include "grid.hpp"
#include <cmath>
#include <numeric>

namespace ifem_host {

namespace GridGenerator {
template <int dim>
void subdivided_hyper_rectangle(Triangulation<dim> &tria, const std::vector<unsigned> &repetitions,
                                const std::array<double, dim> &p0, const std::array<double, dim> &p1, bool colorize) {
  if ((int)repetitions.size() != dim) throw std::invalid_argument("subdivided_hyper_rectangle: repetitions size");
  std::array<int, 3> r{1, 1, 1};
  for (int d = 0; d < dim; ++d) r[d] = (int)repetitions[d];
  tria.is_box = true;
  tria.reps = r;
  for (int d = 0; d < dim; ++d) { tria.p0[d] = p0[d]; tria.p1[d] = p1[d]; }
  const int nvx = r[0] + 1, nvy = r[1] + 1, nvz = (dim == 3) ? r[2] + 1 : 1;
  tria.vertices.resize((size_t)nvx * nvy * nvz);
  for (int k = 0; k < nvz; ++k)
    for (int j = 0; j < nvy; ++j)
      for (int i = 0; i < nvx; ++i) {
        auto &v = tria.vertices[((size_t)k * nvy + j) * nvx + i];
        const int idx[3] = {i, j, k};
        // p0 + idx * h with h computed once per direction, as the tests' BoxMesh does
        for (int d = 0; d < dim; ++d) v[d] = p0[d] + idx[d] * ((p1[d] - p0[d]) / r[d]);
      }
  const size_t nc = (size_t)r[0] * r[1] * ((dim == 3) ? r[2] : 1);
  tria.cells.resize(nc);
  tria.face_bid.resize(nc);
  const int rz = (dim == 3) ? r[2] : 1;
  for (int k = 0; k < rz; ++k)
    for (int j = 0; j < r[1]; ++j)
      for (int i = 0; i < r[0]; ++i) {
        const size_t c = ((size_t)k * r[1] + j) * r[0] + i;
        for (int v = 0; v < (1 << dim); ++v) {
          const int di = v & 1, dj = (v >> 1) & 1, dk = (v >> 2) & 1;
          tria.cells[c][v] = int32_t(((size_t)(k + dk) * nvy + (j + dj)) * nvx + (i + di));
        }
        const int ci[3] = {i, j, k};
        for (int d = 0; d < dim; ++d) {
          tria.face_bid[c][2 * d] = (ci[d] == 0) ? (colorize ? 2 * d : 0) : -1;
          tria.face_bid[c][2 * d + 1] = (ci[d] == r[d] - 1) ? (colorize ? 2 * d + 1 : 0) : -1;
        }
      }
}
template void subdivided_hyper_rectangle<2>(Triangulation<2> &, const std::vector<unsigned> &, const std::array<double, 2> &,
                                            const std::array<double, 2> &, bool);
template void subdivided_hyper_rectangle<3>(Triangulation<3> &, const std::vector<unsigned> &, const std::array<double, 3> &,
                                            const std::array<double, 3> &, bool);
} // namespace GridGenerator

template <int dim>
void Triangulation<dim>::refine_global(int times) {
  if (times <= 0) return;
  if (!is_box) throw std::runtime_error("refine_global: only box triangulations are supported in this build");
  std::vector<unsigned> r(dim);
  for (int d = 0; d < dim; ++d) r[d] = (unsigned)reps[d] << times;
  std::array<double, dim> a, b;
  for (int d = 0; d < dim; ++d) { a[d] = p0[d]; b[d] = p1[d]; }
  // colorised ids survive refinement (children inherit the face's boundary id)
  bool colorize = false;
  for (auto &f : face_bid) for (int k = 0; k < 2 * dim; ++k) if (f[k] > 0) colorize = true;
  GridGenerator::subdivided_hyper_rectangle<dim>(*this, r, a, b, colorize);
}
template struct Triangulation<2>;
template struct Triangulation<3>;

template <int dim>
static void map_point(const double *X /*[NV][dim]*/, const double *xi, double *out) {
  for (int d = 0; d < dim; ++d) out[d] = 0;
  for (int v = 0; v < (1 << dim); ++v) {
    double w = 1;
    for (int d = 0; d < dim; ++d) w *= ((v >> d) & 1) ? xi[d] : (1.0 - xi[d]);
    for (int d = 0; d < dim; ++d) out[d] += w * X[v * dim + d];
  }
}

template <int dim>
void distribute_dofs(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out) {
  constexpr int NV = 1 << dim;
  const int n1 = kv + 1;
  int nu = 1;
  for (int d = 0; d < dim; ++d) nu *= n1;
  const size_t nc = tria.cells.size();
  out.kv = kv; out.nu = nu; out.np = NV;
  out.vcoords.resize(nc * NV * dim);
  out.cell_face_bid.resize(nc * 2 * dim);
  for (size_t c = 0; c < nc; ++c) {
    for (int v = 0; v < NV; ++v)
      for (int d = 0; d < dim; ++d) out.vcoords[(c * NV + v) * dim + d] = tria.vertices[tria.cells[c][v]][d];
    for (int f = 0; f < 2 * dim; ++f) out.cell_face_bid[c * 2 * dim + f] = tria.face_bid[c][f];
  }
  out.cell_unodes.resize(nc * nu);
  out.cell_pnodes.resize(nc * NV);
  if (tria.is_box) {
    // lattice numbering, x fastest
    const auto &r = tria.reps;
    const int64_t nux = kv * r[0] + 1, nuy = kv * r[1] + 1, nuz = (dim == 3) ? kv * r[2] + 1 : 1;
    const int64_t npx = r[0] + 1, npy = r[1] + 1, npz = (dim == 3) ? r[2] + 1 : 1;
    out.n_unodes = nux * nuy * nuz;
    out.n_pnodes = npx * npy * npz;
    const int rz = (dim == 3) ? r[2] : 1;
    for (int k = 0; k < rz; ++k)
      for (int j = 0; j < r[1]; ++j)
        for (int i = 0; i < r[0]; ++i) {
          const size_t c = ((size_t)k * r[1] + j) * r[0] + i;
          for (int a = 0; a < nu; ++a) {
            const int ai = a % n1, aj = (a / n1) % n1, ak = (dim == 3) ? a / (n1 * n1) : 0;
            out.cell_unodes[c * nu + a] = int32_t(((int64_t)(kv * k + ak) * nuy + (kv * j + aj)) * nux + (kv * i + ai));
          }
          for (int v = 0; v < NV; ++v) {
            const int di = v & 1, dj = (v >> 1) & 1, dk = (v >> 2) & 1;
            out.cell_pnodes[c * NV + v] = int32_t(((int64_t)(k + dk) * npy + (j + dj)) * npx + (i + di));
          }
        }
  } else {
    throw std::runtime_error("distribute_dofs: unstructured triangulations are not supported in this build");
  }
  // support points
  out.unode_coords.assign((size_t)out.n_unodes, {});
  out.pnode_coords.assign((size_t)out.n_pnodes, {});
  std::vector<uint8_t> done_u((size_t)out.n_unodes, 0);
  for (size_t c = 0; c < nc; ++c) {
    const double *X = &out.vcoords[c * NV * dim];
    for (int a = 0; a < nu; ++a) {
      const int32_t nd = out.cell_unodes[c * nu + a];
      if (done_u[nd]) continue;
      done_u[nd] = 1;
      double xi[3] = {0, 0, 0};
      int t = a;
      for (int d = 0; d < dim; ++d) { xi[d] = double(t % n1) / kv; t /= n1; }
      map_point<dim>(X, xi, out.unode_coords[nd].data());
    }
    for (int v = 0; v < NV; ++v)
      for (int d = 0; d < dim; ++d) out.pnode_coords[out.cell_pnodes[c * NV + v]][d] = X[v * dim + d];
  }
}
template void distribute_dofs<2>(const Triangulation<2> &, int, DoFTables<2> &);
template void distribute_dofs<3>(const Triangulation<3> &, int, DoFTables<3> &);

template <int dim>
void make_dirichlet(const Triangulation<dim> &tria, const DoFTables<dim> &dofs,
                    const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &bcs,
                    const std::map<int, std::function<double(const std::array<double, dim> &, unsigned)>> &hard_coded,
                    std::vector<int32_t> &dof, std::vector<double> &value) {
  dof.clear();
  value.clear();
  const int kv = dofs.kv, n1 = kv + 1, nu = dofs.nu;
  std::vector<uint8_t> seen((size_t)dofs.n_u(), 0);
  // per boundary id, in ascending id order; an already constrained dof keeps its first line
  // (interpolate_boundary_values + AffineConstraints::add_line semantics, mpi_fluid_solver.cpp:185-272)
  for (const auto &bc : bcs) {
    const int id = (int)bc.first;
    const unsigned flag = bc.second.first;
    const std::vector<double> &vals = bc.second.second;
    if (flag < 1 || flag > 7 || (dim == 2 && flag > 3)) throw std::invalid_argument("Unrecogonized component flag!");
    std::vector<int> comps;
    for (int c = 0; c < dim; ++c) if (flag & (1u << c)) comps.push_back(c);
    if (vals.size() < comps.size()) throw std::invalid_argument("Dirichlet boundary values: too few entries");
    auto hc = hard_coded.find(id);
    for (size_t cell = 0; cell < tria.cells.size(); ++cell)
      for (int f = 0; f < 2 * dim; ++f) {
        if (tria.face_bid[cell][f] != id) continue;
        const int nd = f / 2, side = (f % 2) ? kv : 0;
        for (int a = 0; a < nu; ++a) {
          int idx[3] = {a % n1, (a / n1) % n1, (dim == 3) ? a / (n1 * n1) : 0};
          if (idx[nd] != side) continue;
          const int32_t node = dofs.cell_unodes[cell * nu + a];
          for (size_t k = 0; k < comps.size(); ++k) {
            const int64_t g = (int64_t)dim * node + comps[k];
            if (seen[g]) continue;
            seen[g] = 1;
            dof.push_back((int32_t)g);
            value.push_back(hc != hard_coded.end() ? hc->second(dofs.unode_coords[node], (unsigned)comps[k]) : vals[k]);
          }
        }
      }
  }
}
template void make_dirichlet<2>(const Triangulation<2> &, const DoFTables<2> &,
                                const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &,
                                const std::map<int, std::function<double(const std::array<double, 2> &, unsigned)>> &,
                                std::vector<int32_t> &, std::vector<double> &);
template void make_dirichlet<3>(const Triangulation<3> &, const DoFTables<3> &,
                                const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &,
                                const std::map<int, std::function<double(const std::array<double, 3> &, unsigned)>> &,
                                std::vector<int32_t> &, std::vector<double> &);

} // namespace ifem_host
