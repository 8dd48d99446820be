// grid.cpp -- see grid.hpp.
#include "grid.hpp"
#include <cmath>
#include <map>
#include <numeric>
#include <unordered_map>
#include <cmath>
#include <functional>
#include <map>
#include <numeric>

namespace ifem_host {

namespace GridGenerator {
template <int dim>
void subdivided_hyper_rectangle(Triangulation<dim> &tria, const std::vector<unsigned> &repetitions,
                                const std::array<double, dim> &p0, const std::array<double, dim> &p1, bool colorize,
                                bool lazy) {
  if ((int)repetitions.size() != dim) throw std::invalid_argument("subdivided_hyper_rectangle: repetitions size");
  std::array<int, 3> r{1, 1, 1};
  for (int d = 0; d < dim; ++d) r[d] = (int)repetitions[d];
  tria.is_box = true;
  tria.colorized = colorize;
  tria.reps = r;
  for (int d = 0; d < dim; ++d) { tria.p0[d] = p0[d]; tria.p1[d] = p1[d]; }
  tria.vertices.clear(); tria.cells.clear(); tria.face_bid.clear();
  if (lazy) return;
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
                                            const std::array<double, 2> &, bool, bool);
template void subdivided_hyper_rectangle<3>(Triangulation<3> &, const std::vector<unsigned> &, const std::array<double, 3> &,
                                            const std::array<double, 3> &, bool, bool);
} // namespace GridGenerator

template <int dim>
void Triangulation<dim>::refine_global(int times) {
  if (times <= 0) return;
  if (!is_box) {
    if (!generator) throw std::runtime_error("refine_global: triangulation has neither box metadata nor a generator");
    level += times;
    auto gen = generator; // the generator rewrites *this
    gen(*this, level);
    return;
  }
  std::vector<unsigned> r(dim);
  for (int d = 0; d < dim; ++d) r[d] = (unsigned)reps[d] << times;
  std::array<double, dim> a, b;
  for (int d = 0; d < dim; ++d) { a[d] = p0[d]; b[d] = p1[d]; }
  // colorised ids survive refinement (children inherit the face's boundary id)
  GridGenerator::subdivided_hyper_rectangle<dim>(*this, r, a, b, colorized, cells.empty());
}
template <int dim>
void Triangulation<dim>::set_refine_flag(size_t coarse_cell) {
  if (!is_box || locally_refined) throw std::runtime_error("set_refine_flag: one level of local refinement on a box triangulation");
  const size_t nc = n_active_cells();
  if (coarse_cell >= nc) throw std::out_of_range("set_refine_flag: no such cell");
  if (refine_flags.size() != nc) refine_flags.assign(nc, 0);
  refine_flags[coarse_cell] = 1;
}
template <int dim>
std::array<double, dim> Triangulation<dim>::cell_center(size_t c) const {
  std::array<double, dim> x{};
  size_t rem = c;
  for (int d = 0; d < dim; ++d) {
    const size_t i = rem % (size_t)reps[d];
    rem /= (size_t)reps[d];
    x[d] = p0[d] + (double(i) + 0.5) * ((p1[d] - p0[d]) / reps[d]);
  }
  return x;
}
template <int dim>
void Triangulation<dim>::execute_coarsening_and_refinement() {
  if (!is_box) throw std::runtime_error("execute_coarsening_and_refinement: box triangulations only");
  bool any = false;
  for (uint8_t f : refine_flags) any = any || f;
  locally_refined = any;
  if (!any) refine_flags.clear();
}
template struct Triangulation<2>;
template struct Triangulation<3>;

namespace Utils {
// The refined mesh is generated directly at `level`: bulk cells refine uniformly; in the 8 ring cells deal.II's
// TransfiniteInterpolationManifold (one curved edge: the PolarManifold arc) reduces to
//   x(xi, eta) = (1 - xi) Arc(eta) + xi Out(eta)
// evaluated at the dyadic points (new vertices are chart-space averages pushed forward).  merge_triangulations keeps
// the bulk's coordinates for the seam vertices; the circle is re-centred to (0.2, 0.2) (utilities.cpp:451-479).
// for_3d: the variant flow_around_cylinder_2d(tria, false) extrudes (utilities.cpp:348-355): the channel starts at x = -0.3
// (25 instead of 22 bulk columns), everything else -- ring, removed cells, boundary ids -- is the same.
static void cylinder_2d(Triangulation<2> &tria, int level, bool for_3d = false) {
  const int s = 1 << level;
  const double left = for_3d ? -0.3 : 0.0;
  const int ncol = for_3d ? 25 : 22, hole0 = for_3d ? 4 : 1;
  const double hx = 2.2 / 22, hy = 0.41 / 4, r_in = 0.05, PI = 3.14159265358979323846;
  const double cx = 0.2, cy = 0.2, sx = 0.2, sy = 0.205;
  const double O[8][2] = {{sx + 0.1, sy}, {sx + 0.1, sy + 0.1025}, {sx, sy + 0.1025}, {sx - 0.1, sy + 0.1025},
                          {sx - 0.1, sy}, {sx - 0.1, sy - 0.1025}, {sx, sy - 0.1025}, {sx + 0.1, sy - 0.1025}};
  tria.is_box = false;
  tria.vertices.clear(); tria.cells.clear(); tria.face_bid.clear();
  std::map<std::pair<long long, long long>, int32_t> vid;
  auto vertex = [&](double x, double y) {
    const std::pair<long long, long long> key{std::llround(x * 1e9), std::llround(y * 1e9)};
    auto it = vid.find(key);
    if (it != vid.end()) return it->second;
    const int32_t id = (int32_t)tria.vertices.size();
    vid[key] = id;
    tria.vertices.push_back({x, y});
    return id;
  };
  std::vector<int32_t> ids((size_t)(s + 1) * (s + 1));
  auto emit_patch = [&](const std::function<void(double, double, double &, double &)> &map) {
    for (int b = 0; b <= s; ++b)
      for (int a = 0; a <= s; ++a) {
        double x, y;
        map(double(a) / s, double(b) / s, x, y);
        ids[(size_t)a * (s + 1) + b] = vertex(x, y);
      }
    for (int b = 0; b < s; ++b)
      for (int a = 0; a < s; ++a)
        tria.cells.push_back({ids[(size_t)a * (s + 1) + b], ids[(size_t)(a + 1) * (s + 1) + b],
                              ids[(size_t)a * (s + 1) + b + 1], ids[(size_t)(a + 1) * (s + 1) + b + 1]});
  };
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < ncol; ++i) {
      if ((i == hole0 || i == hole0 + 1) && (j == 1 || j == 2)) continue; // cells within 0.15 of (0.2, 0.2) are removed
      emit_patch([&](double xi, double eta, double &x, double &y) { x = left + (i + xi) * hx; y = (j + eta) * hy; });
    }
  for (int k = 0; k < 8; ++k)
    emit_patch([&](double xi, double eta, double &x, double &y) {
      const double th = 2 * PI * k / 8 + eta * (PI / 4);
      const double ax = cx + r_in * std::cos(th), ay = cy + r_in * std::sin(th);
      const double ox = (1 - eta) * O[k][0] + eta * O[(k + 1) % 8][0], oy = (1 - eta) * O[k][1] + eta * O[(k + 1) % 8][1];
      x = (1 - xi) * ax + xi * ox; y = (1 - xi) * ay + xi * oy;
    });
  // boundary faces = edges used by one cell; ids from the face centre (utilities.cpp:493-523)
  static const int fv[4][2] = {{0, 2}, {1, 3}, {0, 1}, {2, 3}};
  std::map<std::pair<int32_t, int32_t>, int> count;
  for (auto &c : tria.cells)
    for (auto &f : fv) count[{std::min(c[f[0]], c[f[1]]), std::max(c[f[0]], c[f[1]])}]++;
  tria.face_bid.assign(tria.cells.size(), {-1, -1, -1, -1});
  for (size_t ci = 0; ci < tria.cells.size(); ++ci)
    for (int f = 0; f < 4; ++f) {
      const auto &c = tria.cells[ci];
      if (count[{std::min(c[fv[f][0]], c[fv[f][1]]), std::max(c[fv[f][0]], c[fv[f][1]])}] != 1) continue;
      const double mx = 0.5 * (tria.vertices[c[fv[f][0]]][0] + tria.vertices[c[fv[f][1]]][0]);
      const double my = 0.5 * (tria.vertices[c[fv[f][0]]][1] + tria.vertices[c[fv[f][1]]][1]);
      int id = 4;
      if (std::abs(mx - 2.2) < 1e-12) id = 1;
      else if (std::abs(mx - left) < 1e-12) id = 0;
      else if (std::abs(my - 0.41) < 1e-12) id = 3;
      else if (std::abs(my) < 1e-12) id = 2;
      tria.face_bid[ci][f] = id;
    }
}

template <>
void GridCreator<2>::flow_around_cylinder(Triangulation<2> &tria) {
  tria.level = 0;
  tria.generator = [](Triangulation<2> &t, int level) {
    auto gen = t.generator;
    auto par = t.parent_of;
    const int lv = level;
    cylinder_2d(t, lv);
    t.generator = gen;
    t.parent_of = par;
    t.level = lv;
  };
  // every coarse patch emits its (2^level)^2 cells row by row (emit_patch): the parent of cell (a, b) of a patch is (a/2, b/2)
  tria.parent_of = [](int level, size_t c, size_t &parent, int &offset) {
    const size_t sf = size_t(1) << level, sc = sf / 2;
    const size_t patch = c / (sf * sf), r = c % (sf * sf), b = r / sf, a = r % sf;
    parent = patch * sc * sc + (b / 2) * sc + a / 2;
    offset = int(a & 1) | int(b & 1) << 1;
  };
  cylinder_2d(tria, 0);
}
// GridCreator<3>::flow_around_cylinder (utilities.cpp:526-570): the 2D mesh of the x in [-0.3, 2.2] channel extruded to
// z in [0, 0.41] with 9 slices (8 layers); boundary ids x: 0 / 1, y: 2 / 3, z: 4 / 5, cylinder surface 6.  A refined level
// regenerates the 2D mesh at that level (curved ring as in 2D) and doubles the layers.
static void cylinder_3d(Triangulation<3> &tria, int level) {
  Triangulation<2> t2;
  cylinder_2d(t2, level, true);
  const int layers = 8 << level;
  const double hz = 0.41 / layers;
  const size_t nv2 = t2.vertices.size(), nc2 = t2.cells.size();
  tria.is_box = false;
  tria.vertices.clear(); tria.cells.clear(); tria.face_bid.clear();
  tria.vertices.reserve(nv2 * (layers + 1));
  for (int k = 0; k <= layers; ++k)
    for (size_t v = 0; v < nv2; ++v) tria.vertices.push_back({t2.vertices[v][0], t2.vertices[v][1], k == layers ? 0.41 : k * hz});
  for (int k = 0; k < layers; ++k)
    for (size_t c = 0; c < nc2; ++c) {
      std::array<int32_t, 8> cv;
      for (int v = 0; v < 4; ++v) {
        cv[v] = int32_t(t2.cells[c][v] + size_t(k) * nv2);
        cv[4 + v] = int32_t(t2.cells[c][v] + size_t(k + 1) * nv2);
      }
      tria.cells.push_back(cv);
      std::array<int32_t, 6> fb;
      for (int f = 0; f < 4; ++f) fb[f] = t2.face_bid[c][f] == 4 ? 6 : t2.face_bid[c][f];
      fb[4] = k == 0 ? 4 : -1;
      fb[5] = k == layers - 1 ? 5 : -1;
      tria.face_bid.push_back(fb);
    }
}
template <>
void GridCreator<3>::flow_around_cylinder(Triangulation<3> &tria) {
  tria.level = 0;
  tria.generator = [](Triangulation<3> &t, int level) {
    auto gen = t.generator;
    auto par = t.parent_of;
    const int lv = level;
    cylinder_3d(t, lv);
    t.generator = gen;
    t.parent_of = par;
    t.level = lv;
  };
  // layer k of the extrusion holds the 2D mesh of the level (116 patches of (2^level)^2 cells): 2D rule + the layer pair
  tria.parent_of = [](int level, size_t c, size_t &parent, int &offset) {
    const size_t n_patches = 25 * 4 - 4 + 8;
    const size_t sf = size_t(1) << level, sc = sf / 2, nc2f = n_patches * sf * sf, nc2c = n_patches * sc * sc;
    const size_t k = c / nc2f, c2 = c % nc2f;
    const size_t patch = c2 / (sf * sf), r = c2 % (sf * sf), b = r / sf, a = r % sf;
    parent = (k / 2) * nc2c + patch * sc * sc + (b / 2) * sc + a / 2;
    offset = int(a & 1) | int(b & 1) << 1 | int(k & 1) << 2;
  };
  cylinder_3d(tria, 0);
}
} // namespace Utils

// 3D hexahedral meshes: a Q2 node (i, j, k) in {0, 1, 2}^3 of a cell sits at the centre of the vertices it "spans" (index 0 / 2
// fixes the vertex bit of that direction, 1 frees it): 1 vertex, the 2 of an edge, the 4 of a face or all 8.  The sorted
// vertex tuple identifies the entity in every cell that shares it; its coordinates are their mean (the d-linear map at
// the midpoint), which is where MappingQ1 puts FE_Q(2)'s support points.
static void distribute_dofs_hex(const Triangulation<3> &tria, int kv, DoFTables<3> &out, PartitionTables &part) {
  const size_t nc = tria.cells.size(), nV = tria.vertices.size();
  const int n1 = kv + 1, nu = n1 * n1 * n1;
  out.kv = kv; out.nu = nu; out.np = 8;
  out.vcoords.resize(nc * 24);
  out.cell_face_bid.resize(nc * 6);
  out.cell_pnodes.resize(nc * 8);
  out.cell_unodes.resize(nc * nu);
  for (size_t c = 0; c < nc; ++c) {
    for (int v = 0; v < 8; ++v) {
      for (int d = 0; d < 3; ++d) out.vcoords[(c * 8 + v) * 3 + d] = tria.vertices[tria.cells[c][v]][d];
      out.cell_pnodes[c * 8 + v] = tria.cells[c][v];
    }
    for (int f = 0; f < 6; ++f) out.cell_face_bid[c * 6 + f] = tria.face_bid[c][f];
  }
  out.pnode_coords.assign(tria.vertices.begin(), tria.vertices.end());
  out.n_pnodes = out.n_pnodes_owned = (int64_t)nV;
  if (kv == 1) {
    out.cell_unodes = out.cell_pnodes;
    out.unode_coords = out.pnode_coords;
  } else {
    out.unode_coords.assign(tria.vertices.begin(), tria.vertices.end());
    std::map<std::array<int32_t, 8>, int32_t> entity; // sorted spanned vertices, padded with -1
    for (size_t c = 0; c < nc; ++c) {
      const auto &cv = tria.cells[c];
      for (int a = 0; a < 27; ++a) {
        const int idx[3] = {a % 3, (a / 3) % 3, a / 9};
        std::array<int32_t, 8> key;
        key.fill(-1);
        int n = 0;
        for (int v = 0; v < 8; ++v) {
          bool in = true;
          for (int d = 0; d < 3; ++d) {
            const int bit = (v >> d) & 1;
            if ((idx[d] == 0 && bit != 0) || (idx[d] == 2 && bit != 1)) in = false;
          }
          if (in) key[n++] = cv[v];
        }
        if (n == 1) { out.cell_unodes[c * 27 + a] = key[0]; continue; }
        std::sort(key.begin(), key.begin() + n);
        auto it = entity.find(key);
        if (it == entity.end()) {
          it = entity.emplace(key, (int32_t)out.unode_coords.size()).first;
          std::array<double, 3> x{0, 0, 0};
          for (int i = 0; i < n; ++i)
            for (int d = 0; d < 3; ++d) x[d] += tria.vertices[key[i]][d] / n;
          out.unode_coords.push_back(x);
        }
        out.cell_unodes[c * 27 + a] = it->second;
      }
    }
  }
  out.n_unodes = out.n_unodes_owned = (int64_t)out.unode_coords.size();
  part = PartitionTables();
  part.l2g_u.resize((size_t)out.n_unodes);
  part.l2g_p.resize((size_t)out.n_pnodes);
  std::iota(part.l2g_u.begin(), part.l2g_u.end(), 0);
  std::iota(part.l2g_p.begin(), part.l2g_p.end(), 0);
  part.send_u_ptr = part.recv_u_ptr = part.send_p_ptr = part.recv_p_ptr = {0};
  part.n_unodes_global = out.n_unodes; part.n_pnodes_global = out.n_pnodes; part.n_cells_global = (int64_t)nc;
}

template <int dim>
void distribute_dofs_unstructured(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out, PartitionTables &part) {
  if constexpr (dim != 2) {
    distribute_dofs_hex(tria, kv, out, part);
  } else {
    const size_t nc = tria.cells.size(), nV = tria.vertices.size();
    const int n1 = kv + 1, nu = n1 * n1;
    out.kv = kv; out.nu = nu; out.np = 4;
    out.vcoords.resize(nc * 8);
    out.cell_face_bid.resize(nc * 4);
    out.cell_pnodes.resize(nc * 4);
    out.cell_unodes.resize(nc * nu);
    for (size_t c = 0; c < nc; ++c)
      for (int v = 0; v < 4; ++v) {
        out.vcoords[(c * 4 + v) * 2] = tria.vertices[tria.cells[c][v]][0];
        out.vcoords[(c * 4 + v) * 2 + 1] = tria.vertices[tria.cells[c][v]][1];
        out.cell_pnodes[c * 4 + v] = tria.cells[c][v];
        out.cell_face_bid[c * 4 + v] = tria.face_bid[c][v];
      }
    out.pnode_coords.assign(tria.vertices.begin(), tria.vertices.end());
    out.n_pnodes = out.n_pnodes_owned = (int64_t)nV;
    if (kv == 1) {
      out.cell_unodes = out.cell_pnodes;
      out.unode_coords = out.pnode_coords;
      out.n_unodes = out.n_unodes_owned = (int64_t)nV;
    } else {
      // Q2: vertices, then edge midpoints (one per vertex pair), then cell centres; local order x fastest
      static const int ev[4][3] = {{1, 0, 1}, {3, 0, 2}, {5, 1, 3}, {7, 2, 3}}; // local node, vertex a, vertex b
      static const int vloc[4] = {0, 2, 6, 8};
      std::map<std::pair<int32_t, int32_t>, int32_t> eid;
      out.unode_coords.assign(tria.vertices.begin(), tria.vertices.end());
      for (size_t c = 0; c < nc; ++c) {
        const auto &cv = tria.cells[c];
        for (int v = 0; v < 4; ++v) out.cell_unodes[c * 9 + vloc[v]] = cv[v];
        for (auto &e : ev) {
          const std::pair<int32_t, int32_t> key{std::min(cv[e[1]], cv[e[2]]), std::max(cv[e[1]], cv[e[2]])};
          auto it = eid.find(key);
          if (it == eid.end()) {
            it = eid.emplace(key, (int32_t)out.unode_coords.size()).first;
            out.unode_coords.push_back({0.5 * (tria.vertices[cv[e[1]]][0] + tria.vertices[cv[e[2]]][0]),
                                        0.5 * (tria.vertices[cv[e[1]]][1] + tria.vertices[cv[e[2]]][1])});
          }
          out.cell_unodes[c * 9 + e[0]] = it->second;
        }
      }
      for (size_t c = 0; c < nc; ++c) {
        const auto &cv = tria.cells[c];
        out.cell_unodes[c * 9 + 4] = (int32_t)out.unode_coords.size();
        double x = 0, y = 0;
        for (int v = 0; v < 4; ++v) { x += 0.25 * tria.vertices[cv[v]][0]; y += 0.25 * tria.vertices[cv[v]][1]; }
        out.unode_coords.push_back({x, y});
      }
      out.n_unodes = out.n_unodes_owned = (int64_t)out.unode_coords.size();
    }
    part = PartitionTables();
    part.l2g_u.resize((size_t)out.n_unodes);
    part.l2g_p.resize((size_t)out.n_pnodes);
    std::iota(part.l2g_u.begin(), part.l2g_u.end(), 0);
    std::iota(part.l2g_p.begin(), part.l2g_p.end(), 0);
    part.send_u_ptr = part.recv_u_ptr = part.send_p_ptr = part.recv_p_ptr = {0};
    part.n_unodes_global = out.n_unodes; part.n_pnodes_global = out.n_pnodes; part.n_cells_global = (int64_t)nc;
  }
}
template void distribute_dofs_unstructured<2>(const Triangulation<2> &, int, DoFTables<2> &, PartitionTables &);
template void distribute_dofs_unstructured<3>(const Triangulation<3> &, int, DoFTables<3> &, PartitionTables &);

template <int dim>
void partition_unstructured(const DoFTables<dim> &g, int nranks, int rank, DoFTables<dim> &out, PartitionTables &part,
                            const HangingLines *lines, HangingLines *local_lines) {
  constexpr int NV = 1 << dim;
  const int nu = g.nu;
  const size_t nc = g.cell_unodes.size() / nu;
  if (nranks < 1 || rank < 0 || rank >= nranks) throw std::invalid_argument("partition_unstructured: bad rank");
  // 1. strips of equal cell count along x (centroid), ties by y
  std::vector<size_t> order(nc);
  std::iota(order.begin(), order.end(), size_t(0));
  auto centroid = [&](size_t c, int d) { double s = 0; for (int v = 0; v < NV; ++v) s += g.vcoords[(c * NV + v) * dim + d]; return s / NV; };
  std::sort(order.begin(), order.end(), [&](size_t a, size_t b) {
    const double xa = centroid(a, 0), xb = centroid(b, 0);
    if (xa != xb) return xa < xb;
    const double ya = centroid(a, 1), yb = centroid(b, 1);
    return ya != yb ? ya < yb : a < b;
  });
  std::vector<int> cell_rank(nc);
  for (size_t k = 0; k < nc; ++k) cell_rank[order[k]] = (int)std::min<size_t>(k * nranks / nc, nranks - 1);
  // 2. node owners: lowest rank among the cells around the node
  std::vector<int> uown((size_t)g.n_unodes, nranks), pown((size_t)g.n_pnodes, nranks);
  for (size_t c = 0; c < nc; ++c) {
    for (int a = 0; a < nu; ++a) { int &o = uown[g.cell_unodes[c * nu + a]]; o = std::min(o, cell_rank[c]); }
    for (int v = 0; v < NV; ++v) { int &o = pown[g.cell_pnodes[c * NV + v]]; o = std::min(o, cell_rank[c]); }
  }
  // 3. local cells and local node numbering of ANY rank t (needed for t = rank and for the neighbours' ghost lists)
  struct Local { std::vector<size_t> cells; std::vector<int64_t> lu, lp; int64_t nuo = 0, npo = 0; };
  auto build = [&](int t) {
    Local L;
    std::vector<char> useu((size_t)g.n_unodes, 0), usep((size_t)g.n_pnodes, 0);
    for (size_t c = 0; c < nc; ++c) {
      bool touch = false;
      for (int a = 0; a < nu && !touch; ++a) touch = uown[g.cell_unodes[c * nu + a]] == t;
      if (!touch) continue;
      L.cells.push_back(c);
      for (int a = 0; a < nu; ++a) useu[g.cell_unodes[c * nu + a]] = 1;
      for (int v = 0; v < NV; ++v) usep[g.cell_pnodes[c * NV + v]] = 1;
    }
    // hanging-node lines: the masters of every local hanging node join the ghost layer (the interpolation x_h = sum w x_m and
    // its transpose run on local data; masters are regular nodes of an unrefined cell, so one pass closes the set)
    if (lines) {
      const int64_t n_u_glob = int64_t(dim) * g.n_unodes;
      for (size_t i = 0; i < lines->dof.size(); ++i) {
        const int64_t d = lines->dof[i];
        const bool vel = d < n_u_glob;
        if (!(vel ? useu[d / dim] : usep[d - n_u_glob])) continue;
        for (int32_t k = lines->ptr[i]; k < lines->ptr[i + 1]; ++k) {
          const int64_t md = lines->master[k];
          if (vel) useu[md / dim] = 1; else usep[md - n_u_glob] = 1;
        }
      }
    }
    auto number = [&](const std::vector<char> &use, const std::vector<int> &own, std::vector<int64_t> &l2g, int64_t &n_owned) {
      for (int64_t i = 0; i < (int64_t)use.size(); ++i) if (use[i] && own[i] == t) l2g.push_back(i);
      n_owned = (int64_t)l2g.size();
      std::vector<std::pair<int, int64_t>> gh;
      for (int64_t i = 0; i < (int64_t)use.size(); ++i) if (use[i] && own[i] != t) gh.push_back({own[i], i});
      std::sort(gh.begin(), gh.end());
      for (auto &x : gh) l2g.push_back(x.second);
    };
    number(useu, uown, L.lu, L.nuo);
    number(usep, pown, L.lp, L.npo);
    return L;
  };
  const Local me = build(rank);
  // 4. local tables
  out = DoFTables<dim>();
  out.kv = g.kv; out.nu = nu; out.np = NV; out.morton = g.morton;
  out.n_unodes = (int64_t)me.lu.size(); out.n_pnodes = (int64_t)me.lp.size();
  out.n_unodes_owned = me.nuo; out.n_pnodes_owned = me.npo;
  std::vector<int32_t> gu2l((size_t)g.n_unodes, -1), gp2l((size_t)g.n_pnodes, -1);
  for (size_t i = 0; i < me.lu.size(); ++i) gu2l[me.lu[i]] = (int32_t)i;
  for (size_t i = 0; i < me.lp.size(); ++i) gp2l[me.lp[i]] = (int32_t)i;
  out.unode_coords.resize(me.lu.size()); out.pnode_coords.resize(me.lp.size());
  for (size_t i = 0; i < me.lu.size(); ++i) out.unode_coords[i] = g.unode_coords[me.lu[i]];
  for (size_t i = 0; i < me.lp.size(); ++i) out.pnode_coords[i] = g.pnode_coords[me.lp[i]];
  const size_t nl = me.cells.size();
  out.vcoords.resize(nl * NV * dim); out.cell_face_bid.resize(nl * 2 * dim);
  out.cell_unodes.resize(nl * nu); out.cell_pnodes.resize(nl * NV);
  for (size_t k = 0; k < nl; ++k) {
    const size_t c = me.cells[k];
    for (int i = 0; i < NV * dim; ++i) out.vcoords[k * NV * dim + i] = g.vcoords[c * NV * dim + i];
    for (int f = 0; f < 2 * dim; ++f) out.cell_face_bid[k * 2 * dim + f] = g.cell_face_bid[c * 2 * dim + f];
    for (int a = 0; a < nu; ++a) out.cell_unodes[k * nu + a] = gu2l[g.cell_unodes[c * nu + a]];
    for (int v = 0; v < NV; ++v) out.cell_pnodes[k * NV + v] = gp2l[g.cell_pnodes[c * NV + v]];
  }
  // 5. halo plans: my ghosts owned by s arrive in global order; I send to s what s holds of my owned nodes, same order
  part = PartitionTables();
  part.rank = rank; part.nranks = nranks; part.P = {nranks, 1, 1};
  part.l2g_u = me.lu; part.l2g_p = me.lp;
  part.n_unodes_global = g.n_unodes; part.n_pnodes_global = g.n_pnodes; part.n_cells_global = (int64_t)nc;
  part.send_u_ptr = {0}; part.recv_u_ptr = {0}; part.send_p_ptr = {0}; part.recv_p_ptr = {0};
  for (int s = 0; s < nranks; ++s) {
    if (s == rank) continue;
    const Local other = build(s);
    std::vector<int32_t> su, sp;
    for (size_t i = (size_t)other.nuo; i < other.lu.size(); ++i) if (uown[other.lu[i]] == rank) su.push_back(gu2l[other.lu[i]]);
    for (size_t i = (size_t)other.npo; i < other.lp.size(); ++i) if (pown[other.lp[i]] == rank) sp.push_back(gp2l[other.lp[i]]);
    int32_t ru = 0, rp = 0;
    for (size_t i = (size_t)me.nuo; i < me.lu.size(); ++i) if (uown[me.lu[i]] == s) ++ru;
    for (size_t i = (size_t)me.npo; i < me.lp.size(); ++i) if (pown[me.lp[i]] == s) ++rp;
    if (su.empty() && sp.empty() && !ru && !rp) continue;
    part.neighbors.push_back(s);
    part.send_u_idx.insert(part.send_u_idx.end(), su.begin(), su.end());
    part.send_p_idx.insert(part.send_p_idx.end(), sp.begin(), sp.end());
    part.send_u_ptr.push_back((int32_t)part.send_u_idx.size());
    part.send_p_ptr.push_back((int32_t)part.send_p_idx.size());
    part.recv_u_ptr.push_back(part.recv_u_ptr.back() + ru);
    part.recv_p_ptr.push_back(part.recv_p_ptr.back() + rp);
  }
  if (part.recv_u_ptr.back() != out.n_unodes - out.n_unodes_owned || part.recv_p_ptr.back() != out.n_pnodes - out.n_pnodes_owned)
    throw std::logic_error("partition_unstructured: ghost bookkeeping is inconsistent");
  // 6. the lines of the local (owned AND ghost) hanging dofs in local block numbering [dim * unode + c | dim * n_unodes + pnode]
  if (lines && local_lines) {
    local_lines->clear();
    const int64_t n_u_glob = int64_t(dim) * g.n_unodes, n_u_loc = int64_t(dim) * out.n_unodes;
    auto loc = [&](int64_t d) -> int64_t {
      if (d < n_u_glob) { const int32_t l = gu2l[d / dim]; return l < 0 ? -1 : int64_t(dim) * l + d % dim; }
      const int32_t l = gp2l[d - n_u_glob];
      return l < 0 ? -1 : n_u_loc + l;
    };
    std::map<int64_t, size_t> mine; // ascending local dof
    for (size_t i = 0; i < lines->dof.size(); ++i) { const int64_t l = loc(lines->dof[i]); if (l >= 0) mine[l] = i; }
    for (auto &kv_ : mine) {
      const size_t i = kv_.second;
      local_lines->dof.push_back((int32_t)kv_.first);
      for (int32_t k = lines->ptr[i]; k < lines->ptr[i + 1]; ++k) {
        const int64_t lm = loc(lines->master[k]);
        if (lm < 0) throw std::logic_error("partition_unstructured: a master of a local hanging node is not local");
        local_lines->master.push_back((int32_t)lm);
        local_lines->weight.push_back(lines->weight[k]);
      }
      local_lines->ptr.push_back((int32_t)local_lines->master.size());
    }
  }
}
template void partition_unstructured<2>(const DoFTables<2> &, int, int, DoFTables<2> &, PartitionTables &, const HangingLines *, HangingLines *);
template void partition_unstructured<3>(const DoFTables<3> &, int, int, DoFTables<3> &, PartitionTables &, const HangingLines *, HangingLines *);

template <int dim>
static void map_point(const double *X /*[NV][dim]*/, const double *xi, double *out) {
  for (int d = 0; d < dim; ++d) out[d] = 0;
  for (int v = 0; v < (1 << dim); ++v) {
    double w = 1;
    for (int d = 0; d < dim; ++d) w *= ((v >> d) & 1) ? xi[d] : (1.0 - xi[d]);
    for (int d = 0; d < dim; ++d) out[d] += w * X[v * dim + d];
  }
}

// one lattice (velocity nodes with k = kv, pressure nodes with k = 1) of the partitioned box
struct Lattice {
  int k;
  int64_t N[3];        // global nodes per direction
  int64_t lo[3], hi[3]; // local box (inclusive)
  int64_t ln[3];        // local box extents
  std::vector<int32_t> local_id; // local box position -> local node id
  int64_t gid(const int64_t *g) const { return (g[2] * N[1] + g[1]) * N[0] + g[0]; }
  int64_t lpos(const int64_t *g) const { return ((g[2] - lo[2]) * ln[1] + (g[1] - lo[1])) * ln[0] + (g[0] - lo[0]); }
};

// Morton (Z-order) key of a lattice point: nodes and cells that are close in space get close indices, so the x
// gathers of the SpMV and the scatter targets of the assembly stay within the XCD L2s (lexicographic numbering
// spreads a 5x5x5 node neighbourhood over five 1.6 MB planes at 128^3)
static uint64_t morton3(uint64_t x, uint64_t y, uint64_t z) {
  auto spread = [](uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
  };
  return spread(x) | (spread(y) << 1) | (spread(z) << 2);
}

static int owner_block(int64_t g, int k, int n, int P) {
  if (g == 0) return 0;
  const int64_t b = (g - 1) / (int64_t(k) * n);
  return int(b < P ? b : P - 1);
}

template <int dim>
void distribute_dofs_box(const std::array<int, 3> &reps, const std::array<double, 3> &p0, const std::array<double, 3> &p1,
                         bool colorize, int kv, const std::array<int, 3> &P, int rank, DoFTables<dim> &out,
                         PartitionTables &part) {
  constexpr int NV = 1 << dim;
  const int n1 = kv + 1;
  int nu = 1;
  for (int d = 0; d < dim; ++d) nu *= n1;
  int G[3] = {1, 1, 1}, Pd[3] = {1, 1, 1}, n[3] = {1, 1, 1}, b[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) {
    G[d] = reps[d]; Pd[d] = P[d];
    if (Pd[d] < 1 || G[d] % Pd[d] != 0) throw std::invalid_argument("box partition: repetitions must be divisible by the process grid");
    n[d] = G[d] / Pd[d];
  }
  for (int d = dim; d < 3; ++d) if (P[d] != 1) throw std::invalid_argument("box partition: process grid exceeds the dimension");
  const int nranks = Pd[0] * Pd[1] * Pd[2];
  if (rank < 0 || rank >= nranks) throw std::invalid_argument("box partition: bad rank");
  b[0] = rank % Pd[0]; b[1] = (rank / Pd[0]) % Pd[1]; b[2] = rank / (Pd[0] * Pd[1]);
  auto rank_of = [&](const int *bb) { return (bb[2] * Pd[1] + bb[1]) * Pd[0] + bb[0]; };
  auto cell_range = [&](const int *bb, int d, int &c0, int &c1) {
    if (d >= dim) { c0 = 0; c1 = 1; return; }
    c0 = bb[d] * n[d];
    c1 = std::min((bb[d] + 1) * n[d] + 1, G[d]);
  };
  auto make_lattice = [&](int k, const int *bb, Lattice &L) {
    L.k = k;
    for (int d = 0; d < 3; ++d) {
      int c0, c1;
      cell_range(bb, d, c0, c1);
      L.N[d] = (d < dim) ? int64_t(k) * G[d] + 1 : 1;
      L.lo[d] = (d < dim) ? int64_t(k) * c0 : 0;
      L.hi[d] = (d < dim) ? int64_t(k) * c1 : 0;
      L.ln[d] = L.hi[d] - L.lo[d] + 1;
    }
  };
  auto owner_of = [&](int k, const int64_t *g) {
    int ob[3] = {0, 0, 0};
    for (int d = 0; d < dim; ++d) ob[d] = owner_block(g[d], k, n[d], Pd[d]);
    return rank_of(ob);
  };

  part.rank = rank; part.nranks = nranks; part.P = {Pd[0], Pd[1], Pd[2]};
  part.n_cells_global = int64_t(G[0]) * G[1] * G[2];

  // ---- node numbering of one lattice: owned first (lexicographic), then ghosts grouped by owner rank
  struct Ghost { int32_t owner; int64_t gid; int64_t lpos; };
  auto number_lattice = [&](int k, Lattice &L, std::vector<int64_t> &l2g, int64_t &n_owned, std::vector<Ghost> &ghosts) {
    make_lattice(k, b, L);
    const int64_t nloc = L.ln[0] * L.ln[1] * L.ln[2];
    L.local_id.assign((size_t)nloc, -1);
    l2g.clear(); ghosts.clear();
    struct Owned { uint64_t key; int64_t gid, lpos; };
    std::vector<Owned> owned;
    int64_t g[3];
    for (g[2] = L.lo[2]; g[2] <= L.hi[2]; ++g[2])
      for (g[1] = L.lo[1]; g[1] <= L.hi[1]; ++g[1])
        for (g[0] = L.lo[0]; g[0] <= L.hi[0]; ++g[0]) {
          const int ow = owner_of(k, g);
          if (ow == rank) owned.push_back({out.morton ? morton3(g[0], g[1], g[2]) : (uint64_t)L.gid(g), L.gid(g), L.lpos(g)});
          else ghosts.push_back({(int32_t)ow, L.gid(g), L.lpos(g)});
        }
    if (out.morton) std::sort(owned.begin(), owned.end(), [](const Owned &a, const Owned &c) { return a.key < c.key; });
    for (auto &o : owned) { L.local_id[(size_t)o.lpos] = (int32_t)l2g.size(); l2g.push_back(o.gid); }
    n_owned = (int64_t)l2g.size();
    std::stable_sort(ghosts.begin(), ghosts.end(), [](const Ghost &a, const Ghost &c) { return a.owner != c.owner ? a.owner < c.owner : a.gid < c.gid; });
    for (auto &gh : ghosts) { L.local_id[(size_t)gh.lpos] = (int32_t)l2g.size(); l2g.push_back(gh.gid); }
  };
  Lattice LU, LP;
  std::vector<Ghost> gu, gp;
  number_lattice(kv, LU, part.l2g_u, out.n_unodes_owned, gu);
  number_lattice(1, LP, part.l2g_p, out.n_pnodes_owned, gp);
  out.n_unodes = (int64_t)part.l2g_u.size();
  out.n_pnodes = (int64_t)part.l2g_p.size();
  part.n_unodes_global = LU.N[0] * LU.N[1] * LU.N[2];
  part.n_pnodes_global = LP.N[0] * LP.N[1] * LP.N[2];

  // ---- halo plans: neighbours are the adjacent blocks; both sides enumerate the shared nodes in global order
  part.neighbors.clear();
  std::vector<std::vector<int32_t>> send_u, send_p;
  std::vector<int32_t> recv_u_cnt, recv_p_cnt;
  for (int dz = -1; dz <= 1; ++dz)
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        if (!dx && !dy && !dz) continue;
        int bb[3] = {b[0] + dx, b[1] + dy, b[2] + dz};
        bool ok = true;
        for (int d = 0; d < 3; ++d) if (bb[d] < 0 || bb[d] >= Pd[d]) ok = false;
        if (!ok) continue;
        const int s = rank_of(bb);
        auto sends_to = [&](int k, const Lattice &mine) {
          Lattice Ls;
          make_lattice(k, bb, Ls);
          std::vector<int32_t> idx;
          int64_t g[3];
          for (g[2] = Ls.lo[2]; g[2] <= Ls.hi[2]; ++g[2])
            for (g[1] = Ls.lo[1]; g[1] <= Ls.hi[1]; ++g[1])
              for (g[0] = Ls.lo[0]; g[0] <= Ls.hi[0]; ++g[0])
                if (owner_of(k, g) == rank) idx.push_back(mine.local_id[(size_t)mine.lpos(g)]);
          return idx;
        };
        std::vector<int32_t> su = sends_to(kv, LU), sp = sends_to(1, LP);
        int32_t ru = 0, rp = 0;
        for (auto &gh : gu) if (gh.owner == s) ++ru;
        for (auto &gh : gp) if (gh.owner == s) ++rp;
        if (su.empty() && sp.empty() && !ru && !rp) continue;
        part.neighbors.push_back(s);
        send_u.push_back(su); send_p.push_back(sp);
        recv_u_cnt.push_back(ru); recv_p_cnt.push_back(rp);
      }
  // sort neighbours by rank (ghosts are grouped by ascending owner rank)
  std::vector<int> order(part.neighbors.size());
  std::iota(order.begin(), order.end(), 0);
  std::sort(order.begin(), order.end(), [&](int a, int c) { return part.neighbors[a] < part.neighbors[c]; });
  std::vector<int32_t> nb;
  part.send_u_ptr = {0}; part.recv_u_ptr = {0}; part.send_p_ptr = {0}; part.recv_p_ptr = {0};
  part.send_u_idx.clear(); part.send_p_idx.clear();
  for (int o : order) {
    nb.push_back(part.neighbors[o]);
    part.send_u_idx.insert(part.send_u_idx.end(), send_u[o].begin(), send_u[o].end());
    part.send_p_idx.insert(part.send_p_idx.end(), send_p[o].begin(), send_p[o].end());
    part.send_u_ptr.push_back((int32_t)part.send_u_idx.size());
    part.send_p_ptr.push_back((int32_t)part.send_p_idx.size());
    part.recv_u_ptr.push_back(part.recv_u_ptr.back() + recv_u_cnt[o]);
    part.recv_p_ptr.push_back(part.recv_p_ptr.back() + recv_p_cnt[o]);
  }
  part.neighbors = nb;
  if (part.recv_u_ptr.back() != out.n_unodes - out.n_unodes_owned || part.recv_p_ptr.back() != out.n_pnodes - out.n_pnodes_owned)
    throw std::logic_error("box partition: a ghost node has a non-adjacent owner");

  // ---- 2-deep pressure halo plan (explicit S_m on several ranks)
  part.p_lattice_n = {LP.N[0], LP.N[1], LP.N[2]};
  part.sm_box_id.clear(); part.send_s_ptr.clear(); part.send_s_idx.clear(); part.recv_s_ptr.clear();
  part.sm_box_n = {0, 0, 0};
  {
    bool wide = nranks > 1;
    for (int d = 0; d < dim; ++d) if (n[d] < 2) wide = false;
    if (wide) {
      auto own_range = [&](const int *bb, int d, int64_t &lo, int64_t &hi) { // owned pressure nodes of block bb (inclusive)
        if (d >= dim) { lo = hi = 0; return; }
        lo = bb[d] == 0 ? 0 : int64_t(bb[d]) * n[d] + 1;
        hi = int64_t(bb[d] + 1) * n[d];
      };
      auto s_box = [&](const int *bb, int64_t *lo, int64_t *hi) {
        for (int d = 0; d < 3; ++d) {
          int64_t a, c;
          own_range(bb, d, a, c);
          lo[d] = d < dim ? std::max<int64_t>(0, a - 2) : 0;
          hi[d] = d < dim ? std::min<int64_t>(LP.N[d] - 1, c + 2) : 0;
        }
      };
      int64_t slo[3], shi[3];
      s_box(b, slo, shi);
      for (int d = 0; d < 3; ++d) { part.sm_box_lo[d] = slo[d]; part.sm_box_n[d] = shi[d] - slo[d] + 1; }
      const int64_t vol = part.sm_box_n[0] * part.sm_box_n[1] * part.sm_box_n[2];
      part.sm_box_id.assign((size_t)vol, -1);
      auto spos = [&](const int64_t *g) { return ((g[2] - slo[2]) * part.sm_box_n[1] + (g[1] - slo[1])) * part.sm_box_n[0] + (g[0] - slo[0]); };
      struct G2 { int32_t owner; int64_t gid; int64_t pos; };
      std::vector<G2> far;
      int64_t g[3];
      for (g[2] = slo[2]; g[2] <= shi[2]; ++g[2])
        for (g[1] = slo[1]; g[1] <= shi[1]; ++g[1])
          for (g[0] = slo[0]; g[0] <= shi[0]; ++g[0]) {
            const int ow = owner_of(1, g);
            if (ow == rank) part.sm_box_id[(size_t)spos(g)] = LP.local_id[(size_t)LP.lpos(g)];
            else far.push_back({(int32_t)ow, LP.gid(g), spos(g)});
          }
      std::stable_sort(far.begin(), far.end(), [](const G2 &a, const G2 &c) { return a.owner != c.owner ? a.owner < c.owner : a.gid < c.gid; });
      for (size_t i = 0; i < far.size(); ++i) part.sm_box_id[(size_t)far[i].pos] = (int32_t)(out.n_pnodes_owned + (int64_t)i);
      part.send_s_ptr = {0}; part.recv_s_ptr = {0};
      for (int32_t s : part.neighbors) { // same neighbour order as the other plans
        int bb[3] = {s % Pd[0], (s / Pd[0]) % Pd[1], s / (Pd[0] * Pd[1])};
        int64_t nlo[3], nhi[3];
        s_box(bb, nlo, nhi);
        for (g[2] = nlo[2]; g[2] <= nhi[2]; ++g[2])   // what the neighbour's box holds of my owned nodes, in global order
          for (g[1] = nlo[1]; g[1] <= nhi[1]; ++g[1])
            for (g[0] = nlo[0]; g[0] <= nhi[0]; ++g[0])
              if (owner_of(1, g) == rank) part.send_s_idx.push_back(LP.local_id[(size_t)LP.lpos(g)]);
        part.send_s_ptr.push_back((int32_t)part.send_s_idx.size());
        int32_t cnt = 0;
        for (auto &f : far) if (f.owner == s) ++cnt;
        part.recv_s_ptr.push_back(part.recv_s_ptr.back() + cnt);
      }
      if (part.recv_s_ptr.back() != (int32_t)far.size()) throw std::logic_error("box partition: a 2-deep pressure neighbour is not an adjacent block");
    }
  }

  // ---- local cells (lexicographic over the local cell box)
  int c0[3], c1[3];
  for (int d = 0; d < 3; ++d) cell_range(b, d, c0[d], c1[d]);
  const size_t nc = size_t(c1[0] - c0[0]) * (c1[1] - c0[1]) * (c1[2] - c0[2]);
  out.kv = kv; out.nu = nu; out.np = NV;
  out.vcoords.resize(nc * NV * dim);
  out.cell_face_bid.resize(nc * 2 * dim);
  out.cell_unodes.resize(nc * nu);
  out.cell_pnodes.resize(nc * NV);
  double h[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) h[d] = (p1[d] - p0[d]) / G[d];
  struct CellKey { uint64_t key; int ci, cj, ck; };
  std::vector<CellKey> corder;
  corder.reserve(nc);
  for (int ck = c0[2]; ck < c1[2]; ++ck)
    for (int cj = c0[1]; cj < c1[1]; ++cj)
      for (int ci = c0[0]; ci < c1[0]; ++ci) corder.push_back({out.morton ? morton3(ci, cj, ck) : (uint64_t)corder.size(), ci, cj, ck});
  if (out.morton) std::sort(corder.begin(), corder.end(), [](const CellKey &a, const CellKey &c) { return a.key < c.key; });
  for (size_t c = 0; c < nc; ++c) {
      {
        const int ci = corder[c].ci, cj = corder[c].cj, ck = corder[c].ck;
        const int cc[3] = {ci, cj, ck};
        for (int v = 0; v < NV; ++v) {
          int64_t g[3] = {ci + (v & 1), cj + ((v >> 1) & 1), (dim == 3) ? ck + ((v >> 2) & 1) : 0};
          for (int d = 0; d < dim; ++d) out.vcoords[(c * NV + v) * dim + d] = p0[d] + g[d] * h[d];
          out.cell_pnodes[c * NV + v] = LP.local_id[(size_t)LP.lpos(g)];
        }
        for (int a = 0; a < nu; ++a) {
          const int ai = a % n1, aj = (a / n1) % n1, ak = (dim == 3) ? a / (n1 * n1) : 0;
          int64_t g[3] = {int64_t(kv) * ci + ai, int64_t(kv) * cj + aj, (dim == 3) ? int64_t(kv) * ck + ak : 0};
          out.cell_unodes[c * nu + a] = LU.local_id[(size_t)LU.lpos(g)];
        }
        for (int d = 0; d < dim; ++d) {
          out.cell_face_bid[c * 2 * dim + 2 * d] = (cc[d] == 0) ? (colorize ? 2 * d : 0) : -1;
          out.cell_face_bid[c * 2 * dim + 2 * d + 1] = (cc[d] == G[d] - 1) ? (colorize ? 2 * d + 1 : 0) : -1;
        }
      }
  }
  // ---- support points (lattice points of the box; identical on every rank sharing the node)
  out.unode_coords.assign((size_t)out.n_unodes, {});
  out.pnode_coords.assign((size_t)out.n_pnodes, {});
  for (int64_t i = 0; i < out.n_unodes; ++i) {
    int64_t t = part.l2g_u[i];
    for (int d = 0; d < dim; ++d) { out.unode_coords[i][d] = p0[d] + double(t % LU.N[d]) * (h[d] / kv); t /= LU.N[d]; }
  }
  for (int64_t i = 0; i < out.n_pnodes; ++i) {
    int64_t t = part.l2g_p[i];
    for (int d = 0; d < dim; ++d) { out.pnode_coords[i][d] = p0[d] + double(t % LP.N[d]) * h[d]; t /= LP.N[d]; }
  }
}
template void distribute_dofs_box<2>(const std::array<int, 3> &, const std::array<double, 3> &, const std::array<double, 3> &, bool,
                                     int, const std::array<int, 3> &, int, DoFTables<2> &, PartitionTables &);
template void distribute_dofs_box<3>(const std::array<int, 3> &, const std::array<double, 3> &, const std::array<double, 3> &, bool,
                                     int, const std::array<int, 3> &, int, DoFTables<3> &, PartitionTables &);

template <int dim>
void distribute_dofs(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out) {
  if (!tria.is_box) throw std::runtime_error("distribute_dofs: unstructured triangulations are not supported in this build");
  PartitionTables part;
  distribute_dofs_box<dim>(tria.reps, tria.p0, tria.p1, tria.colorized, kv, {1, 1, 1}, 0, out, part);
}
template void distribute_dofs<2>(const Triangulation<2> &, int, DoFTables<2> &);
template void distribute_dofs<3>(const Triangulation<3> &, int, DoFTables<3> &);

template <int dim>
void distribute_dofs_refined_box(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out, PartitionTables &part,
                                 HangingLines &lines) {
  if (!tria.is_box) throw std::runtime_error("distribute_dofs_refined_box: box triangulations only");
  constexpr int NV = 1 << dim;
  const int n1 = kv + 1;
  int nu = 1;
  for (int d = 0; d < dim; ++d) nu *= n1;
  int reps[3] = {1, 1, 1};
  double h[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) { reps[d] = tria.reps[d]; h[d] = (tria.p1[d] - tria.p0[d]) / reps[d]; }
  const size_t n_coarse = (size_t)reps[0] * reps[1] * reps[2];
  const bool flagged = tria.locally_refined && tria.refine_flags.size() == n_coarse;
  // cells on the half-cell lattice: origin and size (1: child of a refined cell, 2: unrefined coarse cell)
  struct Cell { int org[3]; int size; };
  std::vector<Cell> cells;
  for (size_t c = 0; c < n_coarse; ++c) {
    int ci[3] = {int(c % reps[0]), int((c / reps[0]) % reps[1]), int(c / ((size_t)reps[0] * reps[1]))};
    if (flagged && tria.refine_flags[c]) {
      for (int ch = 0; ch < NV; ++ch) {
        Cell k{{0, 0, 0}, 1};
        for (int d = 0; d < dim; ++d) k.org[d] = 2 * ci[d] + ((ch >> d) & 1);
        cells.push_back(k);
      }
    } else {
      Cell k{{0, 0, 0}, 2};
      for (int d = 0; d < dim; ++d) k.org[d] = 2 * ci[d];
      cells.push_back(k);
    }
  }
  const size_t nc = cells.size();
  auto pack = [](const int64_t *k) { return (uint64_t(k[2]) << 42) | (uint64_t(k[1]) << 21) | uint64_t(k[0]); };
  std::unordered_map<uint64_t, int32_t> uid, pid;
  std::vector<std::array<int64_t, 3>> ukeys, pkeys;
  out = DoFTables<dim>();
  out.kv = kv; out.nu = nu; out.np = NV;
  out.cell_unodes.resize(nc * nu); out.cell_pnodes.resize(nc * NV);
  out.vcoords.resize(nc * NV * dim); out.cell_face_bid.assign(nc * 2 * dim, -1);
  for (size_t c = 0; c < nc; ++c) {
    const Cell &k = cells[c];
    for (int a = 0; a < nu; ++a) { // velocity lattice: kv units per half cell
      const int l[3] = {a % n1, (a / n1) % n1, a / (n1 * n1)};
      int64_t key[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) key[d] = int64_t(k.org[d]) * kv + int64_t(l[d]) * k.size;
      auto it = uid.emplace(pack(key), (int32_t)uid.size());
      if (it.second) ukeys.push_back({key[0], key[1], key[2]});
      out.cell_unodes[c * nu + a] = it.first->second;
    }
    for (int a = 0; a < NV; ++a) {
      int64_t key[3] = {0, 0, 0};
      for (int d = 0; d < dim; ++d) key[d] = k.org[d] + ((a >> d) & 1) * k.size;
      auto it = pid.emplace(pack(key), (int32_t)pid.size());
      if (it.second) pkeys.push_back({key[0], key[1], key[2]});
      out.cell_pnodes[c * NV + a] = it.first->second;
      for (int d = 0; d < dim; ++d) out.vcoords[(c * NV + a) * dim + d] = tria.p0[d] + double(key[d]) * h[d] / 2;
    }
    for (int d = 0; d < dim; ++d) { // colorised boundary ids 2d / 2d + 1 (children inherit them)
      if (k.org[d] == 0) out.cell_face_bid[c * 2 * dim + 2 * d] = tria.colorized ? 2 * d : 0;
      if (k.org[d] + k.size == 2 * reps[d]) out.cell_face_bid[c * 2 * dim + 2 * d + 1] = tria.colorized ? 2 * d + 1 : 0;
    }
  }
  out.n_unodes = out.n_unodes_owned = (int64_t)uid.size();
  out.n_pnodes = out.n_pnodes_owned = (int64_t)pid.size();
  out.unode_coords.resize(ukeys.size());
  out.pnode_coords.resize(pkeys.size());
  for (size_t i = 0; i < ukeys.size(); ++i)
    for (int d = 0; d < dim; ++d) out.unode_coords[i][d] = tria.p0[d] + double(ukeys[i][d]) * h[d] / (2 * kv);
  for (size_t i = 0; i < pkeys.size(); ++i)
    for (int d = 0; d < dim; ++d) out.pnode_coords[i][d] = tria.p0[d] + double(pkeys[i][d]) * h[d] / 2;
  part = PartitionTables();
  part.l2g_u.resize(ukeys.size()); part.l2g_p.resize(pkeys.size());
  std::iota(part.l2g_u.begin(), part.l2g_u.end(), 0);
  std::iota(part.l2g_p.begin(), part.l2g_p.end(), 0);
  part.n_unodes_global = out.n_unodes; part.n_pnodes_global = out.n_pnodes; part.n_cells_global = (int64_t)nc;
  part.send_u_ptr = {0}; part.recv_u_ptr = {0}; part.send_p_ptr = {0}; part.recv_p_ptr = {0};
  // hanging lines: the first unrefined coarse cell (in cell order) whose closure holds a foreign node interpolates it
  lines.clear();
  const int64_t n_u = dim * out.n_unodes;
  std::map<int32_t, std::pair<std::vector<int32_t>, std::vector<double>>> found;
  auto lagrange = [](int k, double t, double *w) {
    for (int i = 0; i <= k; ++i) {
      w[i] = 1.0;
      for (int j = 0; j <= k; ++j)
        if (j != i) w[i] *= (t - double(j) / k) / (double(i) / k - double(j) / k);
    }
  };
  for (size_t c = 0; c < nc; ++c) {
    const Cell &k = cells[c];
    if (k.size != 2) continue;
    for (int space = 0; space < 2; ++space) { // velocity nodes (degree kv, dim components), pressure nodes (degree 1)
      const int deg = space == 0 ? kv : 1, unit = deg, m1 = deg + 1;
      const auto &table = space == 0 ? uid : pid;
      const int32_t *cell_nodes = space == 0 ? &out.cell_unodes[c * nu] : &out.cell_pnodes[c * NV];
      const int n_cell_nodes = space == 0 ? nu : NV;
      const int span = 2 * unit; // lattice units across the coarse cell
      int64_t cnt = 1;
      for (int d = 0; d < dim; ++d) cnt *= span + 1;
      for (int64_t q = 0; q < cnt; ++q) {
        int64_t key[3] = {0, 0, 0}, rem = q;
        int off[3] = {0, 0, 0};
        for (int d = 0; d < dim; ++d) { off[d] = int(rem % (span + 1)); rem /= span + 1; key[d] = int64_t(k.org[d]) * unit + off[d]; }
        auto it = table.find(pack(key));
        if (it == table.end()) continue;
        const int32_t nd = it->second;
        bool mine = false;
        for (int a = 0; a < n_cell_nodes; ++a) mine = mine || cell_nodes[a] == nd;
        if (mine) continue;
        double w1[3][4];
        for (int d = 0; d < dim; ++d) lagrange(deg, double(off[d]) / span, w1[d]);
        std::vector<int32_t> ms;
        std::vector<double> ws;
        for (int a = 0; a < n_cell_nodes; ++a) {
          const int l[3] = {a % m1, (a / m1) % m1, a / (m1 * m1)};
          double w = 1.0;
          for (int d = 0; d < dim; ++d) w *= w1[d][l[d]];
          if (std::fabs(w) > 1e-13) { ms.push_back(cell_nodes[a]); ws.push_back(w); }
        }
        const int ncomp = space == 0 ? dim : 1;
        for (int cpt = 0; cpt < ncomp; ++cpt) {
          const int32_t dof = space == 0 ? int32_t(dim * nd + cpt) : int32_t(n_u + nd);
          if (found.count(dof)) continue;
          std::vector<int32_t> md(ms.size());
          for (size_t i = 0; i < ms.size(); ++i) md[i] = space == 0 ? int32_t(dim * ms[i] + cpt) : int32_t(n_u + ms[i]);
          found.emplace(dof, std::make_pair(md, ws));
        }
      }
    }
  }
  for (auto &kv_ : found) { // ascending dof
    lines.dof.push_back(kv_.first);
    lines.master.insert(lines.master.end(), kv_.second.first.begin(), kv_.second.first.end());
    lines.weight.insert(lines.weight.end(), kv_.second.second.begin(), kv_.second.second.end());
    lines.ptr.push_back((int32_t)lines.master.size());
  }
}
template void distribute_dofs_refined_box<2>(const Triangulation<2> &, int, DoFTables<2> &, PartitionTables &, HangingLines &);
template void distribute_dofs_refined_box<3>(const Triangulation<3> &, int, DoFTables<3> &, PartitionTables &, HangingLines &);

template <int dim>
void make_dirichlet(const DoFTables<dim> &dofs,
                    const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &bcs,
                    const std::map<int, std::function<double(const std::array<double, dim> &, unsigned)>> &hard_coded,
                    std::vector<int32_t> &dof, std::vector<double> &value, const std::vector<int32_t> *skip) {
  dof.clear();
  value.clear();
  const int kv = dofs.kv, n1 = kv + 1, nu = dofs.nu;
  std::vector<uint8_t> seen((size_t)dofs.n_u(), 0);
  if (skip)
    for (int32_t d : *skip) if (d >= 0 && d < dofs.n_u()) seen[(size_t)d] = 1;
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
    const size_t n_cells = dofs.cell_unodes.size() / nu;
    for (size_t cell = 0; cell < n_cells; ++cell)
      for (int f = 0; f < 2 * dim; ++f) {
        if (dofs.cell_face_bid[cell * 2 * dim + f] != id) continue;
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
template void make_dirichlet<2>(const DoFTables<2> &,
                                const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &,
                                const std::map<int, std::function<double(const std::array<double, 2> &, unsigned)>> &,
                                std::vector<int32_t> &, std::vector<double> &, const std::vector<int32_t> *);
template void make_dirichlet<3>(const DoFTables<3> &,
                                const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &,
                                const std::map<int, std::function<double(const std::array<double, 3> &, unsigned)>> &,
                                std::vector<int32_t> &, std::vector<double> &, const std::vector<int32_t> *);

} // namespace ifem_host
