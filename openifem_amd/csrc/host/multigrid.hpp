// multigrid.hpp -- the geometric multigrid hierarchy of the host mirror for box triangulations: which coarser meshes
// hang below a level and the nodal transfers between two neighbouring levels, in the layout ifem_mg_attach takes.
// The reference has no counterpart: its A~^-1 is a MUMPS factorisation (source/mpi_insim.cpp:124-127) and its CG(S_m)
// is unpreconditioned (:86-112); the levels replace both inside the preconditioner only.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

namespace ifem_host {

// The next coarser level of a box mesh with `cells_per_rank` cells per rank on a P process grid over `extent`: the
// directions with the smallest cells are halved until the cells are within 1.5x of isotropic (semi-coarsening for
// stretched cells), then all directions, as long as every halved direction keeps min_cells cells per rank.
// Returns false when the chain ends here.
bool next_coarser_level(int dim, const std::array<int, 3> &cells_per_rank, const std::array<int, 3> &P,
                        const std::array<double, 3> &extent, int min_cells, std::array<int, 3> &out);
// the whole chain below a mesh (finest coarse level first), at most max_levels long
std::vector<std::array<int, 3>> coarse_level_chain(int dim, std::array<int, 3> cells_per_rank, const std::array<int, 3> &P,
                                                   const std::array<double, 3> &extent, int min_cells, int max_levels = 8);

struct CsrTransfer {
  int64_t n_rows = 0, n_cols = 0;
  std::vector<int64_t> ptr;
  std::vector<int32_t> col;
  std::vector<double> w;
};

// Nodal prolongation between two nested box meshes of the same domain (ratio 1 or 2 per direction) on the Q_degree node
// lattice: row i = the fine node with global lattice id l2g_fine[i] (x fastest), columns = positions in l2g_coarse of the
// coarse nodes whose Lagrange polynomial does not vanish there.  Throws when a stencil node is missing from l2g_coarse
// (the coarse ghost layer must cover the owned fine nodes: same block partition on both levels).
void box_prolongation(int dim, const std::array<int, 3> &reps_fine, const std::array<int, 3> &reps_coarse, int degree,
                      const int64_t *l2g_fine, int64_t n_fine, const int64_t *l2g_coarse, int64_t n_coarse, CsrTransfer &P);
// R = P^T with sorted rows
void transpose_transfer(const CsrTransfer &P, CsrTransfer &R);
// for every given coarse lattice node the position in l2g_fine of the fine node at the same point
std::vector<int32_t> box_injection(int dim, const std::array<int, 3> &reps_fine, const std::array<int, 3> &reps_coarse,
                                   int degree, const int64_t *l2g_coarse, int64_t n_coarse, const int64_t *l2g_fine,
                                   int64_t n_fine, bool allow_missing = false); // allow_missing: -1 where the fine node is not in the list (replicated coarse level)

// Nodal transfers between two levels of a globally refined unstructured mesh (the cylinder benchmark under refine_global,
// source/utilities.cpp:345-570): row i = fine node i, interpolated from the Q_degree shape functions of the PARENT of a fine cell
// that holds it, at the node's position in the parent's reference cell ((child offset + reference position in the child) / 2 --
// the transfer deal.II's MGTransfer builds from cell->child(i); curved patches move the nodes, not the reference positions).
// cell_nodes_* : [cells][n1^dim] tensor-lexicographic local order; parent[c] / offset[c] of every fine cell.
void nested_prolongation(int dim, int degree, const int32_t *cell_nodes_fine, size_t n_cells_fine, int64_t n_nodes_fine,
                         const int32_t *cell_nodes_coarse, int64_t n_nodes_coarse, const std::vector<size_t> &parent,
                         const std::vector<int> &offset, CsrTransfer &P);
// for every coarse node the fine node at the same reference point of the refinement tree
std::vector<int32_t> nested_injection(int dim, int degree, const int32_t *cell_nodes_fine, size_t n_cells_fine,
                                      const int32_t *cell_nodes_coarse, size_t n_cells_coarse, int64_t n_nodes_coarse,
                                      const std::vector<size_t> &parent, const std::vector<int> &offset);

} // namespace ifem_host
