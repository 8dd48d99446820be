// grid.hpp -- minimal mesh + DoF layer standing in for the deal.II pieces the fluid step needs:
//   Triangulation / GridGenerator::subdivided_hyper_rectangle(.., colorize=true)   (tests/*/.cpp drivers)
//   DoFHandler::distribute_dofs + block renumbering [velocity | pressure]           (mpi_fluid_solver.cpp:116-162)
//   VectorTools::interpolate_boundary_values on boundary ids with component masks   (mpi_fluid_solver.cpp:165-280)
// Nodes are numbered lexicographically by support-point coordinates (z, then y, then x): a bandwidth-friendly
// order like the reference's Cuthill-McKee pass, and identical to the lattice order on box meshes.
#pragma once
#include <algorithm>
#include <array>
#include <cstdint>
#include <functional>
#include <map>
#include <stdexcept>
#include <vector>

namespace ifem_host {

template <int dim>
struct Triangulation {
  static constexpr int NV = 1 << dim;
  std::vector<std::array<double, dim>> vertices;
  std::vector<std::array<int32_t, NV>> cells;          // lexicographic vertex order (x fastest)
  std::vector<std::array<int32_t, 2 * dim>> face_bid;  // boundary id per face x-,x+,y-,y+,z-,z+ ; -1 interior
  // structured provenance (set by subdivided_hyper_rectangle, cleared by anything unstructured)
  bool is_box = false;
  bool colorized = false;
  std::array<int, 3> reps{1, 1, 1};
  std::array<double, 3> p0{0, 0, 0}, p1{1, 1, 1};
  // box triangulations may be "lazy": only the metadata is kept and cells are generated per rank on demand
  size_t n_active_cells() const { return is_box ? size_t(reps[0]) * reps[1] * reps[2] : cells.size(); }
  // unstructured triangulations built by a generator (Utils::GridCreator) are re-generated at the new level by
  // refine_global: the generator places the new vertices on the manifolds the reference attaches
  std::function<void(Triangulation<dim> &, int level)> generator;
  int level = 0;
  // ... and know their refinement history: cell `fine_cell` of level `level` (>= 1) is child `offset` (bit d: upper half in
  // direction d of the parent's reference cell) of cell `parent` of level - 1 -- what deal.II's cell->parent() / child(i) give
  // the multigrid transfers (host/multigrid.cpp::nested_prolongation)
  std::function<void(int level, size_t fine_cell, size_t &parent, int &offset)> parent_of;
  void refine_global(int times);
  // One level of LOCAL refinement of a box triangulation: cell->set_refine_flag() on coarse cells (numbered x fastest) and
  // execute_coarsening_and_refinement(), as the reference's FSI drivers refine a band of the fluid mesh
  // (tests/fsi_leaflet_mpi/fsi_leaflet_mpi.cpp:65-75).  Flagged cells are replaced by their 2^dim children; the mesh is
  // one-irregular by construction (one level) and the hanging-node lines come from distribute_dofs_refined_box.
  std::vector<uint8_t> refine_flags; // per coarse cell, empty: none
  bool locally_refined = false;
  void set_refine_flag(size_t coarse_cell);
  std::array<double, dim> cell_center(size_t coarse_cell) const;
  void execute_coarsening_and_refinement();
};

// hanging-node lines x[dof] = sum_k weight[k] x[master[k]], k in [ptr[i], ptr[i+1]), in the block numbering
// [dim * unode + c | n_u + pnode] (DoFTools::make_hanging_node_constraints, mpi_fluid_solver.cpp:182-184)
struct HangingLines {
  std::vector<int32_t> dof, ptr{0}, master;
  std::vector<double> weight;
  void clear() { dof.clear(); ptr.assign(1, 0); master.clear(); weight.clear(); }
};

namespace Utils {
// GridCreator<2>::flow_around_cylinder (reference source/utilities.cpp:345-524): the DFG cylinder benchmark mesh,
// 22 x 4 bulk cells with the 8-cell polar/transfinite ring around the cylinder, boundary ids 0 inflow, 1 outflow,
// 2 y=0, 3 y=0.41, 4 cylinder.  GridCreator<3> (utilities.cpp:526-570): the x in [-0.3, 2.2] variant extruded to
// z in [0, 0.41] in 8 layers, boundary ids 0 / 1 (x), 2 / 3 (y), 4 / 5 (z), 6 cylinder surface.
template <int dim>
struct GridCreator {
  static void flow_around_cylinder(Triangulation<dim> &tria);
};
} // namespace Utils

namespace GridGenerator {
template <int dim>
void subdivided_hyper_rectangle(Triangulation<dim> &tria, const std::vector<unsigned> &repetitions,
                                const std::array<double, dim> &p0, const std::array<double, dim> &p1, bool colorize,
                                bool lazy = false);
}

// cell -> node tables for FESystem(FE_Q(kv)^dim, FE_Q(1)); local order tensor-lexicographic
template <int dim>
struct DoFTables {
  int kv = 2, nu = 0, np = 0;
  int64_t n_unodes = 0, n_pnodes = 0;
  std::vector<int32_t> cell_unodes, cell_pnodes;
  std::vector<double> vcoords;        // [n_cells][2^dim][dim]
  std::vector<int32_t> cell_face_bid; // [n_cells][2*dim]
  std::vector<std::array<double, dim>> unode_coords, pnode_coords; // support points (d-linear map of the unit lattice)
  // multi-GPU: nodes [0, n_*_owned) are owned by this rank, the rest are ghosts (grouped by owner rank)
  int64_t n_unodes_owned = 0, n_pnodes_owned = 0;
  // input of distribute_dofs*: number owned nodes and order cells along a Morton curve (default) or lexicographically
  bool morton = true;
  int64_t n_u() const { return dim * n_unodes; }
  int64_t n_dofs() const { return dim * n_unodes + n_pnodes; }
};

// Block partition of a box mesh over a P[0] x P[1] x P[2] process grid: rank r owns the cells of its
// lattice block; a node belongs to the lowest rank touching it; every rank also assembles the ghost cell layer on
// its upper faces so that all cells touching an owned row are local ("owner computes row").
struct PartitionTables {
  int rank = 0, nranks = 1;
  std::array<int, 3> P{1, 1, 1};
  std::vector<int32_t> neighbors;                        // sorted ranks exchanging halo data with this rank
  std::vector<int32_t> send_u_ptr, send_u_idx, recv_u_ptr; // ifem_partition layout
  std::vector<int32_t> send_p_ptr, send_p_idx, recv_p_ptr;
  std::vector<int64_t> l2g_u, l2g_p;                     // local node -> global lattice id
  int64_t n_unodes_global = 0, n_pnodes_global = 0, n_cells_global = 0;
  // 2-deep pressure halo for an explicit S_m = B diag(M_u)^-1 B^T on several ranks (its rows couple pressure nodes two
  // cells apart).  Column space: [owned pressure nodes | nodes of the box "owned range +-2" owned elsewhere, grouped by
  // owner rank in global order].  Empty when a block is narrower than 2 cells.
  std::array<int64_t, 3> p_lattice_n{1, 1, 1};           // global pressure lattice
  std::array<int64_t, 3> sm_box_lo{0, 0, 0}, sm_box_n{0, 0, 0};
  std::vector<int32_t> sm_box_id;                        // [box] lattice position -> S_m column id
  std::vector<int32_t> send_s_ptr, send_s_idx, recv_s_ptr; // same layout as the pressure halo plan
};

// local tables of rank `rank` for the box mesh reps x [p0,p1] without ever building the global mesh
template <int dim>
void distribute_dofs_box(const std::array<int, 3> &reps, const std::array<double, 3> &p0, const std::array<double, 3> &p1,
                         bool colorize, int kv, const std::array<int, 3> &P, int rank, DoFTables<dim> &out,
                         PartitionTables &part);

template <int dim>
void distribute_dofs(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out);
// a box triangulation after execute_coarsening_and_refinement (single rank): coarse cells x fastest with the children of
// a refined cell in its place, nodes numbered in order of first appearance, and the hanging-node lines of the
// refinement interfaces (a node in the closure of an unrefined coarse cell that is not one of its nodes is interpolated
// from that cell's shape functions)
template <int dim>
void distribute_dofs_refined_box(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out, PartitionTables &part,
                                 HangingLines &lines);
// general (unstructured, single rank) variant: vertices, edge midpoints, (face centres,) cell centres
template <int dim>
void distribute_dofs_unstructured(const Triangulation<dim> &tria, int kv, DoFTables<dim> &out, PartitionTables &part);
// Partition of an unstructured mesh given by its GLOBAL tables (every rank builds them: these meshes are small) into the
// tables of rank `rank` of `nranks`: cells are cut into strips of equal count along x (p4est / METIS stand-in: results
// do not depend on the partition), a node belongs to the lowest rank among the cells around it, the local cells are all
// cells touching an owned node ("owner computes row"), ghosts are grouped by owner in global order.
// `lines` (global hanging-node lines, e.g. of distribute_dofs_refined_box): the masters of every local hanging node join the
// ghost layer, `local_lines` receives the lines of the local (owned and ghost) hanging dofs in local numbering -- what
// ifem_set_hanging_constraints takes on a partitioned context (the reference runs its locally refined FSI meshes on >= 2 ranks:
// tests/fsi_leaflet_mpi/fsi_leaflet_mpi.cpp:65-76, mpi_fluid_solver.cpp:182-184)
template <int dim>
void partition_unstructured(const DoFTables<dim> &global, int nranks, int rank, DoFTables<dim> &out, PartitionTables &part,
                            const HangingLines *lines = nullptr, HangingLines *local_lines = nullptr);

// Dirichlet lines (dof, value) in block numbering [u|p]; `bcs`: id -> (component flag 1..7, values);
// `hard_coded`: id -> f(point, component) overriding the constant values (add_hard_coded_boundary_condition).
// `skip` (may be NULL): dofs that already carry a line (hanging nodes): interpolate_boundary_values leaves them alone
template <int dim>
void make_dirichlet(const DoFTables<dim> &dofs,
                    const std::map<unsigned, std::pair<unsigned, std::vector<double>>> &bcs,
                    const std::map<int, std::function<double(const std::array<double, dim> &, unsigned)>> &hard_coded,
                    std::vector<int32_t> &dof, std::vector<double> &value, const std::vector<int32_t> *skip = nullptr);

} // namespace ifem_host
