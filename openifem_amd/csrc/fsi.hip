// fsi.hip -- the fluid-side inputs MPI::FSI computes before every fluid step, on the device.
//
// Reference (source/mpi_fsi.cpp): update_solid_box :96-127, point_in_solid :142-223, update_indicator :291-319,
// find_fluid_bc :323-663 (nodal fsi_stress :415-480, fsi_acceleration :489-556, Dirichlet lines of the artificial fluid
// :569-651); Utils::GridInterpolator / CellLocator (source/utilities.cpp:193-244, :295-341) for the solid cell around a
// point and the Q1 point value.
//
// Shape of the work: a few 10^6..10^7 fluid points (cell vertices, velocity support points) against a solid of 10^2..10^4
// cells that every rank holds whole.  Almost every point fails the solid_box test: a workgroup whose points all fail
// leaves at once.  For the others
//  - the 2D crossing number needs every boundary face: the workgroup streams them through LDS in tiles and every lane walks
//    the tile for its own point;
//  - CellAccessor<3>::point_inside and the cell search only concern the cells whose bounding box holds the point: a
//    uniform grid over solid_box (about one bin per solid cell, built on the host when the solid is handed over) lists
//    them per bin in ascending order, so "some cell contains the point" / "the cell of smallest distance, lowest index on
//    ties" are the answers a serial loop over the whole solid gives (brute force over 3456 cells measured 9.4 + 12.5 ms
//    at 128^3 for update_indicator + find_fluid_bc).
//
// The reference evaluates a node in the first cell of its cell loop that touches it (dof_touched, :437-441, :506-508);
// the gradient of the fluid velocity at the support point depends on that cell.  Here: two passes of atomicMin over the
// (cell, local node) pairs give every node the touching cell of smallest order (the local cell index, or the caller's global
// active-cell index on several ranks), then one lane per node does what the reference does in that cell.  No value is
// accumulated, so the result does not depend on scheduling.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <vector>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int kBlock = 256;

template <int DIM> struct SolidView {
  int32_t nc, nbf;
  const double *rec;   // [nc][REC]
  const double *bface; // [nbf][4]
  const int32_t *cells;
  double box[6];
  // uniform grid over solid_box: the cells whose (slightly inflated) bounding box meets a bin, ascending per bin
  const int32_t *bin_ptr, *bin_cells;
  int G[3];
  double inv_h[3];
};
template <int DIM> __device__ inline int bin_of(const SolidView<DIM> &S, const double *p) {
  int idx = 0, stride = 1;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    int i = int((p[d] - S.box[2 * d]) * S.inv_h[d]);
    i = i < 0 ? 0 : (i > S.G[d] - 1 ? S.G[d] - 1 : i);
    idx += i * stride;
    stride *= S.G[d];
  }
  return idx;
}
template <int DIM> constexpr int rec_len() { return 2 * DIM + DIM * (1 << DIM); }

// d-linear shape functions of the unit cell, lexicographic vertex order
template <int DIM> __device__ inline void q1_shape(const double *xi, double *N, double *dN) {
  constexpr int NV = 1 << DIM;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    double val = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) val *= ((v >> d) & 1) ? xi[d] : 1.0 - xi[d];
    N[v] = val;
    if (dN) {
#pragma unroll
      for (int e = 0; e < DIM; ++e) {
        double g = 1.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          const int hi = (v >> d) & 1;
          g *= d == e ? (hi ? 1.0 : -1.0) : (hi ? xi[d] : 1.0 - xi[d]);
        }
        dN[v * DIM + e] = g;
      }
    }
  }
}

template <int DIM> __device__ inline double det_inv(const double *J, double *Ji) {
  if constexpr (DIM == 2) {
    const double det = J[0] * J[3] - J[1] * J[2];
    Ji[0] = J[3] / det; Ji[1] = -J[1] / det; Ji[2] = -J[2] / det; Ji[3] = J[0] / det;
    return det;
  } else {
    const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    const double det = J[0] * c00 + J[1] * c01 + J[2] * c02;
    Ji[0] = c00 / det; Ji[1] = (J[2] * J[7] - J[1] * J[8]) / det; Ji[2] = (J[1] * J[5] - J[2] * J[4]) / det;
    Ji[3] = c01 / det; Ji[4] = (J[0] * J[8] - J[2] * J[6]) / det; Ji[5] = (J[2] * J[3] - J[0] * J[5]) / det;
    Ji[6] = c02 / det; Ji[7] = (J[1] * J[6] - J[0] * J[7]) / det; Ji[8] = (J[0] * J[4] - J[1] * J[3]) / det;
    return det;
  }
}

// MappingQ1::transform_real_to_unit_cell: Newton on x(xi) = p from the cell centre; false = ExcTransformationFailed
template <int DIM> __device__ inline bool real_to_unit(const double *X, const double *p, double *xi) {
  constexpr int NV = 1 << DIM;
  double N[NV], dN[NV * DIM], J[DIM * DIM], Ji[DIM * DIM], F[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) xi[d] = 0.5;
  for (int it = 0; it < 30; ++it) {
    q1_shape<DIM>(xi, N, dN);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      double x = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) x += N[v] * X[v * DIM + c];
      F[c] = x - p[c];
#pragma unroll
      for (int e = 0; e < DIM; ++e) {
        double g = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) g += dN[v * DIM + e] * X[v * DIM + c];
        J[c * DIM + e] = g;
      }
    }
    const double det = det_inv<DIM>(J, Ji);
    if (!(fabs(det) > 0)) return false;
    double step = 0;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < DIM; ++e) {
      double dx = 0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) dx += Ji[e * DIM + c] * F[c];
      xi[e] -= dx;
      step = fmax(step, fabs(dx));
      if (!(fabs(xi[e]) < 1e3)) bad = true;
    }
    if (bad) return false;
    if (step < 1e-15) return true;
  }
  return true;
}

// FSI::point_in_solid for the point of every lane (`active` lanes only; all lanes of the workgroup must call).
// lds: kBlock * 4 doubles (2D: one tile of boundary faces); unused in 3D
template <int DIM> __device__ inline bool block_point_in_solid(bool active, const double *p, const SolidView<DIM> &S, double *lds) {
#pragma unroll
  for (int i = 0; i < DIM; ++i)
    if (p[i] < S.box[2 * i] || p[i] > S.box[2 * i + 1]) active = false; // :147-151
  if (!__syncthreads_or(active)) return false;
  if constexpr (DIM == 2) { // crossing number over the boundary faces, :154-213
    unsigned cross = 0, half = 0;
    bool decided = false;
    for (int32_t f0 = 0; f0 < S.nbf; f0 += kBlock) {
      const int nt = min(kBlock, S.nbf - f0);
      __syncthreads();
      for (int k = threadIdx.x; k < nt * 4; k += kBlock) lds[k] = S.bface[(size_t)f0 * 4 + k];
      __syncthreads();
      if (active && !decided)
        for (int f = 0; f < nt; ++f) {
          const double p1x = lds[4 * f], p1y = lds[4 * f + 1], p2x = lds[4 * f + 2], p2y = lds[4 * f + 3];
          const double y_diff1 = p1y - p[1], y_diff2 = p2y - p[1], x_diff1 = p1x - p[0], x_diff2 = p2x - p[0];
          const double r1x = p1x - p2x, r1y = p1y - p2y;
          double r2x = 0.0;
          if (r1y != 0.0) r2x = r1x * (p[1] - p2y) / r1y;
          if (y_diff1 * y_diff2 < 0) {
            if (r2x + p2x > p[0]) ++cross;
            else if (r2x + p2x == p[0]) { decided = true; break; }
          } else if (y_diff1 * y_diff2 == 0) {
            if (y_diff1 == 0 && y_diff2 == 0) {
              if (x_diff1 * x_diff2 < 0) { decided = true; break; }
              else continue;
            } else if (r2x + p2x > p[0]) {
              if (p[1] != S.box[2] && p[1] != S.box[3]) ++half;
            } else if ((p[0] == p1x && p[1] == p1y) || (p[0] == p2x && p[1] == p2y)) { decided = true; break; }
          }
        }
    }
    if (!active) return false;
    if (decided) return true;
    cross += half / 2;
    return cross % 2 != 0;
  } else { // CellAccessor<3>::point_inside of every cell, :215-222: only the cells of the point's bin can pass its
           // bounding-box test, and "some cell contains the point" does not depend on the order they are asked in
    constexpr int REC = rec_len<3>();
    if (!active) return false;
    const int b = bin_of<3>(S, p);
    for (int32_t k = S.bin_ptr[b]; k < S.bin_ptr[b + 1]; ++k) {
      const double *R = S.rec + (size_t)S.bin_cells[k] * REC;
      if (p[0] < R[0] || p[0] > R[3] || p[1] < R[1] || p[1] > R[4] || p[2] < R[2] || p[2] > R[5]) continue;
      double X[24], xi[3];
#pragma unroll
      for (int i = 0; i < 24; ++i) X[i] = R[6 + i];
      if (!real_to_unit<3>(X, p, xi)) continue;
      if (xi[0] >= 0.0 && xi[0] <= 1.0 && xi[1] >= 0.0 && xi[1] <= 1.0 && xi[2] >= 0.0 && xi[2] <= 1.0) return true;
    }
    return false;
  }
}

// the solid cell around the point (GridTools::find_active_cell_around_point as Utils::GridInterpolator uses it): smallest
// distance of the unit-cell image to the unit cell, lowest cell index on ties, accepted below 1e-10; xi projected to the
// unit cell.  -1: none.
template <int DIM> __device__ inline int32_t locate(bool active, const double *p, const SolidView<DIM> &S, double *xi_out) {
  constexpr int REC = rec_len<DIM>();
  if (!active) return -1;
  int32_t best = -1;
  double best_d = 1e300;
  const int b = bin_of<DIM>(S, p);
  for (int32_t k = S.bin_ptr[b]; k < S.bin_ptr[b + 1]; ++k) { // ascending cell index: ties go to the lowest, as in a serial loop
    const int32_t c = S.bin_cells[k];
    const double *R = S.rec + (size_t)c * REC;
    double ext = 0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) ext = fmax(ext, R[DIM + d] - R[d]);
    bool out = false;
#pragma unroll
    for (int d = 0; d < DIM; ++d)
      if (p[d] < R[d] - 1e-9 * ext || p[d] > R[DIM + d] + 1e-9 * ext) out = true;
    if (out) continue;
    double X[DIM * (1 << DIM)], xi[DIM];
#pragma unroll
    for (int i = 0; i < DIM * (1 << DIM); ++i) X[i] = R[2 * DIM + i];
    if (!real_to_unit<DIM>(X, p, xi)) continue;
    double dd = 0.0; // GeometryInfo::distance_to_unit_cell
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      if (-xi[d] > dd) dd = -xi[d];
      else if (xi[d] - 1.0 > dd) dd = xi[d] - 1.0;
    }
    if (dd < best_d) {
      best_d = dd;
      best = c;
#pragma unroll
      for (int d = 0; d < DIM; ++d) xi_out[d] = xi[d];
    }
  }
  if (best < 0 || !(best_d < 1e-10)) return -1;
#pragma unroll
  for (int d = 0; d < DIM; ++d) xi_out[d] = xi_out[d] < 0.0 ? 0.0 : (xi_out[d] > 1.0 ? 1.0 : xi_out[d]);
  return best;
}

template <int DIM> constexpr int lds_doubles() { return DIM == 2 ? kBlock * 4 : 1; } // 2D: one tile of boundary faces

// bounding box + vertex coordinates of every solid cell, one contiguous record per cell
template <int DIM> __global__ void k_fsi_cell_records(int32_t nc, const double *vert, const int32_t *cells, double *rec) {
  constexpr int NV = 1 << DIM, REC = rec_len<DIM>();
  const int32_t c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  double lo[DIM], hi[DIM];
  for (int v = 0; v < NV; ++v) {
    const int32_t g = cells[(size_t)c * NV + v];
    for (int d = 0; d < DIM; ++d) {
      const double x = vert[(size_t)g * DIM + d];
      rec[(size_t)c * REC + 2 * DIM + v * DIM + d] = x;
      if (v == 0) lo[d] = hi[d] = x;
      else {
        if (x < lo[d]) lo[d] = x;
        if (x > hi[d]) hi[d] = x;
      }
    }
  }
  for (int d = 0; d < DIM; ++d) {
    rec[(size_t)c * REC + d] = lo[d];
    rec[(size_t)c * REC + DIM + d] = hi[d];
  }
}

// FSI::update_indicator: one lane per (cell, vertex); the cell is artificial fluid when all its vertices are in the solid
template <int DIM> __global__ void __launch_bounds__(kBlock) k_fsi_indicator(int64_t n_cells, const double *vcoords, SolidView<DIM> S,
                                                                             int32_t *indicator, int64_t *counters) {
  constexpr int NV = 1 << DIM, CPB = kBlock / NV;
  __shared__ double lds[lds_doubles<DIM>()];
  __shared__ uint8_t in_s[kBlock];
  const int64_t cell = (int64_t)blockIdx.x * CPB + threadIdx.x / NV;
  const bool active = cell < n_cells;
  double p[DIM];
  if (active)
    for (int d = 0; d < DIM; ++d) p[d] = vcoords[((size_t)cell * NV + threadIdx.x % NV) * DIM + d];
  else
    for (int d = 0; d < DIM; ++d) p[d] = 0;
  const bool inside = block_point_in_solid<DIM>(active, p, S, lds);
  in_s[threadIdx.x] = inside;
  __syncthreads();
  if (active && threadIdx.x % NV == 0) {
    int all = 1;
    for (int v = 0; v < NV; ++v) all &= in_s[threadIdx.x + v];
    indicator[cell] = all;
    if (all) atomicAdd((unsigned long long *)&counters[0], 1ull);
  }
}

__device__ inline bool in_cell_point(int dim, int kv, int a) { // a support point in the interior of the cell (:588-600)
  int cnt = 0;
  for (int d = 0; d < dim; ++d) {
    const int i1 = a % (kv + 1);
    a /= kv + 1;
    if (i1 > 0 && i1 < kv) ++cnt;
  }
  return cnt == dim;
}

// first-touch, pass 0: smallest order among the eligible cells that touch the node; pass 1: that cell's (cell, local node)
__global__ void k_fsi_first_touch(int pass, int64_t n_cells, int dim, int kv, int nu, const int32_t *cell_unodes, const int32_t *indicator,
                                  const int32_t *cell_order, int all_cells, uint32_t *order_min, uint32_t *first) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells * nu) return;
  const int64_t c = i / nu;
  const int a = int(i % nu);
  if (all_cells) {
    if (in_cell_point(dim, kv, a)) return;
  } else if (indicator[c] == 0)
    return;
  const uint32_t ord = cell_order ? (uint32_t)cell_order[c] : (uint32_t)c;
  const int32_t node = cell_unodes[i];
  if (pass == 0) atomicMin(&order_min[node], ord);
  else if (order_min[node] == ord) first[node] = (uint32_t)c << 5 | (uint32_t)a;
}

template <int DIM> __device__ inline void support_point(int kv, const double *vc /* [NV][DIM] of the cell */, int a, double *xi, double *x) {
  constexpr int NV = 1 << DIM;
  for (int d = 0; d < DIM; ++d) {
    xi[d] = double(a % (kv + 1)) / kv;
    a /= kv + 1;
  }
  double N[NV];
  q1_shape<DIM>(xi, N, nullptr);
  for (int d = 0; d < DIM; ++d) {
    double v = 0;
#pragma unroll
    for (int k = 0; k < NV; ++k) v += N[k] * vc[k * DIM + d];
    x[d] = v;
  }
}

// nodes whose support point lies in solid_box
template <int DIM> __global__ void k_fsi_candidates(int64_t n_nodes, int kv, const uint32_t *first, const double *vcoords, SolidView<DIM> S,
                                                    int32_t *cand, int64_t *counters) {
  constexpr int NV = 1 << DIM;
  const int64_t node = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (node >= n_nodes) return;
  const uint32_t f = first[node];
  if (f == kNone) return;
  double xi[DIM], x[DIM];
  support_point<DIM>(kv, vcoords + (size_t)(f >> 5) * NV * DIM, int(f & 31), xi, x);
  for (int i = 0; i < DIM; ++i)
    if (x[i] < S.box[2 * i] || x[i] > S.box[2 * i + 1]) return;
  const unsigned long long k = atomicAdd((unsigned long long *)&counters[1], 1ull);
  cand[k] = (int32_t)node;
}

__device__ inline void lagrange(int kv, double x, double *L, double *dL) {
  if (kv == 1) {
    L[0] = 1 - x; L[1] = x;
    dL[0] = -1; dL[1] = 1;
  } else {
    L[0] = 2 * (x - 0.5) * (x - 1); L[1] = -4 * x * (x - 1); L[2] = 2 * x * (x - 0.5);
    dL[0] = 4 * x - 3; dL[1] = -8 * x + 4; dL[2] = 4 * x - 1;
  }
}

struct NodeBcArgs {
  int kv, nu, mode; // mode bit 0: nodal fsi_stress, bit 1: fsi_acceleration, bit 2: Dirichlet lines
  int64_t nUl;
  double dt;
  const uint32_t *first;
  const int32_t *cand;
  const double *vcoords;
  const int32_t *cell_unodes;
  const double *present;      // ghost-extended block vector
  const double *fluid_stress; // [dim][dim][nUl] or nullptr (= 0)
  const double *s_vel, *s_acc, *s_stress;
  int32_t s_nv;
  double *fsi_stress, *fsi_acc;
  const uint8_t *taken;
  uint8_t *is_c0, *is_c1;
  double *cval0, *cval1;
  int64_t *counters; // [1] candidates (input), [2] inside, [3] lines, [4] not found
};

// one lane per candidate node: what the reference does for the node in its first-touch cell
template <int DIM> __global__ void __launch_bounds__(kBlock) k_fsi_node_bc(NodeBcArgs A, SolidView<DIM> S) {
  constexpr int NV = 1 << DIM, NCOMP = DIM * (DIM + 1) / 2;
  __shared__ double lds[lds_doubles<DIM>()];
  const int64_t n_cand = A.counters[1];
  if ((int64_t)blockIdx.x * kBlock >= n_cand) return;
  const int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const bool active = k < n_cand;
  int32_t node = 0, cell = 0;
  int a = 0;
  double xi[DIM], x[DIM];
  for (int d = 0; d < DIM; ++d) x[d] = xi[d] = 0;
  if (active) {
    node = A.cand[k];
    const uint32_t f = A.first[node];
    cell = int32_t(f >> 5);
    a = int(f & 31);
    support_point<DIM>(A.kv, A.vcoords + (size_t)cell * NV * DIM, a, xi, x);
  }
  const bool inside = block_point_in_solid<DIM>(active, x, S, lds);
  if (!inside) return;
  double sxi[DIM];
  const int32_t sc = locate<DIM>(inside, x, S, sxi);
  atomicAdd((unsigned long long *)&A.counters[2], 1ull);
  double N[NV];
  int32_t sv[NV];
  if (sc >= 0) {
    q1_shape<DIM>(sxi, N, nullptr);
    for (int v = 0; v < NV; ++v) sv[v] = S.cells[(size_t)sc * NV + v];
  }
  if ((A.mode & 1) && A.s_stress) { // :459-474 (point_value gives 0 when the search found no cell)
    int idx = 0;
    for (int j = 0; j < DIM; ++j)
      for (int kk = 0; kk < j + 1; ++kk) {
        double s = 0;
        if (sc >= 0)
          for (int v = 0; v < NV; ++v) s += N[v] * A.s_stress[(size_t)idx * A.s_nv + sv[v]];
        const double fl = A.fluid_stress ? A.fluid_stress[((size_t)j * DIM + kk) * A.nUl + node] : 0.0;
        A.fsi_stress[(size_t)idx * A.nUl + node] = fl - s;
        ++idx;
      }
    (void)NCOMP;
  }
  if (!(A.mode & 6)) return;
  if (sc < 0) { // AssertThrow "Cannot find point in solid" (:526-533, :604-611)
    atomicAdd((unsigned long long *)&A.counters[4], 1ull);
    return;
  }
  double vs[DIM], as[DIM], v[DIM];
  for (int c = 0; c < DIM; ++c) {
    double s1 = 0, s2 = 0;
    for (int q = 0; q < NV; ++q) {
      s1 += N[q] * A.s_vel[(size_t)sv[q] * DIM + c];
      if (A.mode & 2) s2 += N[q] * A.s_acc[(size_t)sv[q] * DIM + c];
    }
    vs[c] = s1;
    as[c] = s2;
    v[c] = A.present[(size_t)DIM * node + c];
  }
  if (A.mode & 2) { // fluid_acc = (vs - v) / dt + grad_v v at the support point of the first-touch cell (:548-556)
    const double *vc = A.vcoords + (size_t)cell * NV * DIM;
    double Nf[NV], dN[NV * DIM], J[DIM * DIM], Ji[DIM * DIM], L[DIM][3], dL[DIM][3];
    q1_shape<DIM>(xi, Nf, dN);
    for (int c = 0; c < DIM; ++c)
      for (int e = 0; e < DIM; ++e) {
        double g = 0;
        for (int q = 0; q < NV; ++q) g += dN[q * DIM + e] * vc[q * DIM + c];
        J[c * DIM + e] = g;
      }
    det_inv<DIM>(J, Ji);
    for (int d = 0; d < DIM; ++d) lagrange(A.kv, xi[d], L[d], dL[d]);
    double gref[DIM * DIM];
    for (int i = 0; i < DIM * DIM; ++i) gref[i] = 0;
    const int n1 = A.kv + 1;
    for (int b = 0; b < A.nu; ++b) {
      int idx[DIM], r = b;
      for (int d = 0; d < DIM; ++d) { idx[d] = r % n1; r /= n1; }
      const int32_t nb = A.cell_unodes[(size_t)cell * A.nu + b];
      for (int e = 0; e < DIM; ++e) {
        double g = 1;
        for (int d = 0; d < DIM; ++d) g *= d == e ? dL[d][idx[d]] : L[d][idx[d]];
        for (int c = 0; c < DIM; ++c) gref[c * DIM + e] += g * A.present[(size_t)DIM * nb + c];
      }
    }
    for (int c = 0; c < DIM; ++c) {
      double conv = 0;
      for (int e = 0; e < DIM; ++e) {
        double g = 0;
        for (int q = 0; q < DIM; ++q) g += gref[c * DIM + q] * Ji[q * DIM + e];
        conv += g * v[e];
      }
      A.fsi_acc[(size_t)DIM * node + c] = (vs[c] - v[c]) / A.dt + conv - as[c];
    }
  }
  if (A.mode & 4) // the merge with left_object_wins: a dof that already carries a line keeps it (:641-651)
    for (int c = 0; c < DIM; ++c) {
      const size_t dof = (size_t)DIM * node + c;
      if (A.taken[dof]) continue;
      A.is_c0[dof] = 1;
      A.is_c1[dof] = 1;
      A.cval0[dof] = 0.0;
      A.cval1[dof] = vs[c] - v[c];
      atomicAdd((unsigned long long *)&A.counters[3], 1ull);
    }
}

__global__ void k_fsi_fill_u32(int64_t n, uint32_t v, uint32_t *x) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}
__global__ void k_fsi_taken(int64_t n, const uint8_t *c0, const uint8_t *c1, uint8_t *taken) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) taken[i] = (c0 ? c0[i] : 0) | (c1 ? c1[i] : 0);
}
__global__ void k_fsi_any(int64_t n, const uint8_t *c0, const uint8_t *c1, int64_t *out) { // [0] / [1]: lines of set 0 / 1
  unsigned long long a0 = 0, a1 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    a0 += c0[i] != 0;
    a1 += c1[i] != 0;
  }
  if (a0) atomicAdd((unsigned long long *)&out[0], a0);
  if (a1) atomicAdd((unsigned long long *)&out[1], a1);
}
__global__ void k_fsi_mark(int32_t n, const int32_t *dof, uint8_t *taken) {
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) taken[dof[i]] = 1;
}
// scalar nodal fields <-> the velocity-layout buffer the halo exchange moves ([node][dim]); unpack writes ghosts only
__global__ void k_fsi_pack(int64_t n, int dim, int nf, const double *f, double *buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int c = 0; c < dim; ++c) buf[i * dim + c] = c < nf ? f[(size_t)c * n + i] : 0.0;
}
__global__ void k_fsi_unpack(int64_t n, int64_t n_owned, int dim, int nf, const double *buf, double *f) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + n_owned;
  if (i >= n) return;
  for (int c = 0; c < nf; ++c) f[(size_t)c * n + i] = buf[i * dim + c];
}

// Dirichlet lines of ghost dofs are the owner's: flags / inhomogeneities travel as doubles in the velocity layout
__global__ void k_fsi_lines_pack(int64_t n, int what, const uint8_t *is_c1, const double *cval1, double *buf) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = what == 0 ? (is_c1[i] ? 1.0 : 0.0) : cval1[i];
}
__global__ void k_fsi_lines_unpack(int64_t n, int64_t n_owned, int what, const double *buf, uint8_t *is_c0, uint8_t *is_c1, double *cval1) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + n_owned;
  if (i >= n) return;
  if (what == 0) is_c0[i] = is_c1[i] = buf[i] > 0.5 ? 1 : 0;
  else cval1[i] = buf[i];
}

// ---- the other direction: the fluid solution at points of the solid (find_solid_bc :727-760, update_solid_displacement
// :268-271).  A uniform grid over the bounding box of the local fluid cells, built once per context on the device
// (count, host prefix sum, fill), lists the cells whose bounding box meets a bin; a lane per point walks its bin.
template <int DIM> struct FluidBins {
  const int32_t *ptr, *cells;
  double lo[3], inv_h[3];
  int G[3];
};
template <int DIM> __device__ inline void cell_bin_range(const double *X, const FluidBins<DIM> &B, int *lo, int *hi) {
  constexpr int NV = 1 << DIM;
  double ext = 0, blo[DIM], bhi[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    blo[d] = bhi[d] = X[d];
    for (int v = 1; v < NV; ++v) {
      blo[d] = fmin(blo[d], X[v * DIM + d]);
      bhi[d] = fmax(bhi[d], X[v * DIM + d]);
    }
    ext = fmax(ext, bhi[d] - blo[d]);
  }
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const int a = (int)floor((blo[d] - 1e-8 * ext - B.lo[d]) * B.inv_h[d]), b = (int)floor((bhi[d] + 1e-8 * ext - B.lo[d]) * B.inv_h[d]);
    lo[d] = max(0, min(B.G[d] - 1, a));
    hi[d] = max(0, min(B.G[d] - 1, b));
  }
}
__global__ void k_fluid_minmax(int64_t n, int dim, const double *v, double *partial) { // [blocks][2*dim]: min, max per direction
  __shared__ double s[256];
  for (int d = 0; d < dim; ++d)
    for (int mm = 0; mm < 2; ++mm) {
      double a = mm ? -1e300 : 1e300;
      for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        a = mm ? fmax(a, v[i * dim + d]) : fmin(a, v[i * dim + d]);
      s[threadIdx.x] = a;
      __syncthreads();
      for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s[threadIdx.x] = mm ? fmax(s[threadIdx.x], s[threadIdx.x + o]) : fmin(s[threadIdx.x], s[threadIdx.x + o]);
        __syncthreads();
      }
      if (threadIdx.x == 0) partial[(size_t)blockIdx.x * 2 * dim + 2 * d + mm] = s[0];
      __syncthreads();
    }
}
template <int DIM> __global__ void k_fluid_bin(int pass, int64_t n_cells, const double *vcoords, FluidBins<DIM> B, int32_t *count_or_cursor, int32_t *cells) {
  constexpr int NV = 1 << DIM;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_cells) return;
  int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  cell_bin_range<DIM>(vcoords + (size_t)c * NV * DIM, B, lo, hi);
  for (int k = lo[2]; k <= hi[2]; ++k)
    for (int j = lo[1]; j <= hi[1]; ++j)
      for (int i = lo[0]; i <= hi[0]; ++i) {
        const size_t b = (size_t)i + (size_t)B.G[0] * ((size_t)j + (size_t)B.G[1] * k);
        const int32_t pos = atomicAdd(&count_or_cursor[b], 1);
        if (pass == 1) cells[pos] = (int32_t)c;
      }
}
template <int DIM> __global__ void k_fluid_at_points(int32_t n, const double *pts, int64_t n_cells, int kv, int nu, int64_t nUl, const double *vcoords,
                                                     const int32_t *cell_unodes, const int32_t *cell_pnodes, const double *present,
                                                     const double *stress, FluidBins<DIM> B, double *values, double *st_out, int32_t *cell_out) {
  constexpr int NV = 1 << DIM;
  const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double p[DIM];
  int b = 0, stride = 1;
  bool in_box = true;
  for (int d = 0; d < DIM; ++d) {
    p[d] = pts[(size_t)i * DIM + d];
    const double t = (p[d] - B.lo[d]) * B.inv_h[d];
    if (!(t > -1.0 && t < B.G[d] + 1.0)) in_box = false; // far outside: no bin to ask (NaN lands here too)
    int k = (int)floor(t);
    k = k < 0 ? 0 : (k > B.G[d] - 1 ? B.G[d] - 1 : k);
    b += k * stride;
    stride *= B.G[d];
  }
  int32_t best = -1;
  double best_d = 1e300, bxi[DIM];
  if (in_box)
    for (int32_t q = B.ptr[b]; q < B.ptr[b + 1]; ++q) {
      const int32_t c = B.cells[q];
      double X[NV * DIM], ext = 0, lo[DIM], hi[DIM], xi[DIM];
      for (int k = 0; k < NV * DIM; ++k) X[k] = vcoords[(size_t)c * NV * DIM + k];
      bool out = false;
      for (int d = 0; d < DIM; ++d) {
        lo[d] = hi[d] = X[d];
        for (int v = 1; v < NV; ++v) {
          lo[d] = fmin(lo[d], X[v * DIM + d]);
          hi[d] = fmax(hi[d], X[v * DIM + d]);
        }
        ext = fmax(ext, hi[d] - lo[d]);
      }
      for (int d = 0; d < DIM; ++d)
        if (p[d] < lo[d] - 1e-9 * ext || p[d] > hi[d] + 1e-9 * ext) out = true;
      if (out || !real_to_unit<DIM>(X, p, xi)) continue;
      double dd = 0.0;
      for (int d = 0; d < DIM; ++d) {
        if (-xi[d] > dd) dd = -xi[d];
        else if (xi[d] - 1.0 > dd) dd = xi[d] - 1.0;
      }
      if (dd < best_d || (dd == best_d && c < best)) { // the bin lists are unordered: the tie rule is explicit
        best_d = dd;
        best = c;
        for (int d = 0; d < DIM; ++d) bxi[d] = xi[d];
      }
    }
  for (int k = 0; k < DIM + 1; ++k) values[(size_t)i * (DIM + 1) + k] = 0.0;
  if (st_out)
    for (int k = 0; k < DIM * DIM; ++k) st_out[(size_t)i * DIM * DIM + k] = 0.0;
  if (best < 0 || !(best_d < 1e-10)) {
    cell_out[i] = -1;
    return;
  }
  cell_out[i] = best;
  double L[DIM][3], dL[DIM][3], acc[DIM + 1], sacc[DIM * DIM];
  for (int d = 0; d < DIM; ++d) {
    bxi[d] = bxi[d] < 0.0 ? 0.0 : (bxi[d] > 1.0 ? 1.0 : bxi[d]);
    lagrange(kv, bxi[d], L[d], dL[d]);
  }
  for (int k = 0; k < DIM + 1; ++k) acc[k] = 0;
  for (int k = 0; k < DIM * DIM; ++k) sacc[k] = 0;
  const int n1 = kv + 1;
  for (int a = 0; a < nu; ++a) {
    int r = a;
    double w = 1;
    for (int d = 0; d < DIM; ++d) {
      w *= L[d][r % n1];
      r /= n1;
    }
    const int32_t node = cell_unodes[(size_t)best * nu + a];
    for (int c = 0; c < DIM; ++c) acc[c] += w * present[(size_t)DIM * node + c];
    if (st_out && stress)
      for (int k = 0; k < DIM * DIM; ++k) sacc[k] += w * stress[(size_t)k * nUl + node];
  }
  double Nq[NV];
  q1_shape<DIM>(bxi, Nq, nullptr);
  for (int v = 0; v < NV; ++v) acc[DIM] += Nq[v] * present[(size_t)DIM * nUl + cell_pnodes[(size_t)best * NV + v]];
  for (int k = 0; k < DIM + 1; ++k) values[(size_t)i * (DIM + 1) + k] = acc[k];
  if (st_out)
    for (int k = 0; k < DIM * DIM; ++k) st_out[(size_t)i * DIM * DIM + k] = sacc[k];
}

inline dim3 grid_for(int64_t n, int block = kBlock) { return dim3((unsigned)std::max<int64_t>(1, (n + block - 1) / block)); }

template <int DIM> SolidView<DIM> view_of(const ifem_ctx *ctx) {
  const FsiState &F = ctx->fsi;
  SolidView<DIM> S;
  S.nc = F.nc;
  S.nbf = F.nbf;
  S.rec = F.rec.p;
  S.bface = F.bface.p;
  S.cells = F.cells.p;
  for (int i = 0; i < 6; ++i) S.box[i] = F.box[i];
  S.bin_ptr = F.bin_ptr.p;
  S.bin_cells = F.bin_cells.p;
  for (int d = 0; d < 3; ++d) {
    S.G[d] = F.G[d];
    S.inv_h[d] = F.inv_h[d];
  }
  return S;
}

void require_solid(const ifem_ctx *ctx) {
  if (!ctx->fsi.valid) throw Error(IFEM_E_BADPARAM, "no solid: call ifem_fsi_set_solid first");
}

} // namespace

void fsi_set_solid(ifem_ctx *ctx, const ifem_fsi_solid *s) {
  if (!s || !s->vertices || !s->cell_vertices || s->n_vertices <= 0 || s->n_cells <= 0) throw Error(IFEM_E_BADPARAM, "ifem_fsi_set_solid: empty solid");
  const int dim = ctx->dim, nv = 1 << dim, ncomp = dim * (dim + 1) / 2;
  if (dim == 2 && (s->n_boundary_faces <= 0 || !s->boundary_face_vertices)) throw Error(IFEM_E_BADPARAM, "ifem_fsi_set_solid: a 2D solid needs its boundary faces");
  for (int64_t i = 0; i < (int64_t)s->n_cells * nv; ++i)
    if (s->cell_vertices[i] < 0 || s->cell_vertices[i] >= s->n_vertices) throw Error(IFEM_E_BADPARAM, "ifem_fsi_set_solid: cell vertex out of range");
  FsiState &F = ctx->fsi;
  hipStream_t st = ctx->stream;
  std::vector<int32_t> F_bin_cells_host;
  F.nv = s->n_vertices;
  F.nc = s->n_cells;
  F.nbf = dim == 2 ? s->n_boundary_faces : 0;
  F.vert.upload(s->vertices, (size_t)F.nv * dim, st);
  F.cells.upload(s->cell_vertices, (size_t)F.nc * nv, st);
  if (dim == 2) {
    std::vector<double> bf((size_t)F.nbf * 4);
    for (int32_t f = 0; f < F.nbf; ++f)
      for (int k = 0; k < 2; ++k) {
        const int32_t v = s->boundary_face_vertices[2 * f + k];
        if (v < 0 || v >= F.nv) throw Error(IFEM_E_BADPARAM, "ifem_fsi_set_solid: boundary face vertex out of range");
        bf[(size_t)4 * f + 2 * k] = s->vertices[(size_t)2 * v];
        bf[(size_t)4 * f + 2 * k + 1] = s->vertices[(size_t)2 * v + 1];
      }
    F.bface.upload(bf.data(), bf.size(), st);
  } else
    F.bface.release();
  F.has_fields = s->velocity && s->acceleration;
  if (F.has_fields) {
    F.vel.upload(s->velocity, (size_t)F.nv * dim, st);
    F.acc.upload(s->acceleration, (size_t)F.nv * dim, st);
  }
  F.has_stress = s->stress != nullptr;
  if (F.has_stress) F.stress.upload(s->stress, (size_t)ncomp * F.nv, st);
  // FSI::update_solid_box (:96-127): the solid is a few thousand vertices, the host loop is the reference's own
  for (int i = 0; i < dim; ++i) F.box[2 * i] = F.box[2 * i + 1] = s->vertices[i];
  for (int32_t v = 0; v < F.nv; ++v)
    for (int i = 0; i < dim; ++i) {
      const double x = s->vertices[(size_t)v * dim + i];
      if (x < F.box[2 * i]) F.box[2 * i] = x;
      else if (x > F.box[2 * i + 1]) F.box[2 * i + 1] = x;
    }
  { // bins: about one per solid cell; a cell is listed in every bin its bounding box (inflated by more than the 1e-9 of the
    // cell search) meets.  The solid is small: two host passes over its cells.
    int g = std::max(1, std::min(64, (int)std::lround(std::pow((double)F.nc, 1.0 / dim))));
    int64_t nb = 1;
    for (int d = 0; d < 3; ++d) {
      F.G[d] = d < dim ? g : 1;
      const double len = d < dim ? F.box[2 * d + 1] - F.box[2 * d] : 0.0;
      F.inv_h[d] = len > 0 ? F.G[d] / len : 0.0;
      nb *= F.G[d];
    }
    std::vector<int32_t> ptr((size_t)nb + 1, 0), lo_hi((size_t)F.nc * 6);
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<int32_t> fill;
      if (pass == 1) fill.assign(ptr.begin(), ptr.end() - 1);
      for (int32_t c = 0; c < F.nc; ++c) {
        int lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        if (pass == 0) {
          double ext = 0, blo[3], bhi[3];
          for (int d = 0; d < dim; ++d) {
            blo[d] = bhi[d] = s->vertices[(size_t)s->cell_vertices[(size_t)c * nv] * dim + d];
            for (int v = 1; v < nv; ++v) {
              const double x = s->vertices[(size_t)s->cell_vertices[(size_t)c * nv + v] * dim + d];
              blo[d] = std::min(blo[d], x);
              bhi[d] = std::max(bhi[d], x);
            }
            ext = std::max(ext, bhi[d] - blo[d]);
          }
          for (int d = 0; d < dim; ++d) {
            const int a = (int)std::floor((blo[d] - 1e-8 * ext - F.box[2 * d]) * F.inv_h[d]); // the device applies the same monotone map to p
            const int b = (int)std::floor((bhi[d] + 1e-8 * ext - F.box[2 * d]) * F.inv_h[d]);
            lo_hi[(size_t)c * 6 + d] = std::max(0, std::min(F.G[d] - 1, a));
            lo_hi[(size_t)c * 6 + 3 + d] = std::max(0, std::min(F.G[d] - 1, b));
          }
        }
        for (int d = 0; d < dim; ++d) {
          lo[d] = lo_hi[(size_t)c * 6 + d];
          hi[d] = lo_hi[(size_t)c * 6 + 3 + d];
        }
        for (int k = lo[2]; k <= hi[2]; ++k)
          for (int j = lo[1]; j <= hi[1]; ++j)
            for (int i = lo[0]; i <= hi[0]; ++i) {
              const size_t b = (size_t)i + (size_t)F.G[0] * ((size_t)j + (size_t)F.G[1] * k);
              if (pass == 0) ++ptr[b + 1];
              else F_bin_cells_host[fill[b]++] = c;
            }
      }
      if (pass == 0) {
        for (size_t b = 0; b < (size_t)nb; ++b) ptr[b + 1] += ptr[b];
        F_bin_cells_host.assign((size_t)ptr[nb], 0);
      }
    }
    F.bin_ptr.upload(ptr.data(), ptr.size(), st);
    F.bin_cells.upload(F_bin_cells_host.data(), std::max<size_t>(F_bin_cells_host.size(), 1), st);
    IFEM_HIP_CHECK(hipStreamSynchronize(st));
  }
  F.rec.alloc((size_t)F.nc * (dim == 2 ? rec_len<2>() : rec_len<3>()));
  if (dim == 2) hipLaunchKernelGGL(k_fsi_cell_records<2>, grid_for(F.nc), dim3(kBlock), 0, st, F.nc, F.vert.p, F.cells.p, F.rec.p);
  else hipLaunchKernelGGL(k_fsi_cell_records<3>, grid_for(F.nc), dim3(kBlock), 0, st, F.nc, F.vert.p, F.cells.p, F.rec.p);
  if (F.counters.n != 8) F.counters.alloc(8);
  IFEM_HIP_CHECK(hipStreamSynchronize(st)); // the host arrays may go away
  F.valid = true;
}

void fsi_update_indicator(ifem_ctx *ctx, int32_t *host_out, int64_t *n_artificial) {
  require_solid(ctx);
  FsiState &F = ctx->fsi;
  hipStream_t st = ctx->stream;
  if (ctx->indicator.n != (size_t)ctx->n_cells) ctx->indicator.alloc((size_t)ctx->n_cells);
  IFEM_HIP_CHECK(hipMemsetAsync(F.counters.p, 0, 8 * sizeof(int64_t), st));
  const int nv = 1 << ctx->dim, cpb = kBlock / nv;
  const dim3 grid((unsigned)((ctx->n_cells + cpb - 1) / cpb));
  if (ctx->dim == 2) hipLaunchKernelGGL(k_fsi_indicator<2>, grid, dim3(kBlock), 0, st, ctx->n_cells, ctx->vcoords.p, view_of<2>(ctx), ctx->indicator.p, F.counters.p);
  else hipLaunchKernelGGL(k_fsi_indicator<3>, grid, dim3(kBlock), 0, st, ctx->n_cells, ctx->vcoords.p, view_of<3>(ctx), ctx->indicator.p, F.counters.p);
  if (host_out) IFEM_HIP_CHECK(hipMemcpyAsync(host_out, ctx->indicator.p, (size_t)ctx->n_cells * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  int64_t cnt = 0;
  if (n_artificial) IFEM_HIP_CHECK(hipMemcpyAsync(&cnt, F.counters.p, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  IFEM_HIP_CHECK(hipStreamSynchronize(st));
  if (n_artificial) *n_artificial = cnt;
}

namespace {
// first-touch table of the local velocity nodes over the indicator cells (all_cells = 0) or over every local cell without the
// in-cell support points (all_cells = 1), then the candidate list; returns nothing, counters[1] holds the count
void build_first_touch(ifem_ctx *ctx, const int32_t *d_order, int all_cells) {
  FsiState &F = ctx->fsi;
  hipStream_t st = ctx->stream;
  const int64_t n = ctx->nUl, pairs = ctx->n_cells * ctx->nu;
  if (F.first.n != (size_t)n) {
    F.first.alloc((size_t)n);
    F.order_min.alloc((size_t)n);
    F.cand.alloc((size_t)n);
  }
  hipLaunchKernelGGL(k_fsi_fill_u32, grid_for(n), dim3(kBlock), 0, st, n, kNone, F.order_min.p);
  hipLaunchKernelGGL(k_fsi_fill_u32, grid_for(n), dim3(kBlock), 0, st, n, kNone, F.first.p);
  for (int pass = 0; pass < 2; ++pass)
    hipLaunchKernelGGL(k_fsi_first_touch, grid_for(pairs), dim3(kBlock), 0, st, pass, ctx->n_cells, ctx->dim, ctx->kv, ctx->nu, ctx->cell_unodes.p,
                       ctx->indicator.p, d_order, all_cells, F.order_min.p, F.first.p);
  IFEM_HIP_CHECK(hipMemsetAsync(F.counters.p + 1, 0, sizeof(int64_t), st));
  if (ctx->dim == 2) hipLaunchKernelGGL(k_fsi_candidates<2>, grid_for(n), dim3(kBlock), 0, st, n, ctx->kv, F.first.p, ctx->vcoords.p, view_of<2>(ctx), F.cand.p, F.counters.p);
  else hipLaunchKernelGGL(k_fsi_candidates<3>, grid_for(n), dim3(kBlock), 0, st, n, ctx->kv, F.first.p, ctx->vcoords.p, view_of<3>(ctx), F.cand.p, F.counters.p);
}

void launch_node_bc(ifem_ctx *ctx, NodeBcArgs &A) {
  // the grid covers every node; workgroups beyond the candidate count (known on the device only) leave at once
  const dim3 grid = grid_for(ctx->nUl);
  if (ctx->dim == 2) hipLaunchKernelGGL(k_fsi_node_bc<2>, grid, dim3(kBlock), 0, ctx->stream, A, view_of<2>(ctx));
  else hipLaunchKernelGGL(k_fsi_node_bc<3>, grid, dim3(kBlock), 0, ctx->stream, A, view_of<3>(ctx));
}
} // namespace

void fsi_find_fluid_bc(ifem_ctx *ctx, double dt, int use_dirichlet_bc, const int32_t *cell_order, ifem_fsi_stats *stats) {
  require_solid(ctx);
  FsiState &F = ctx->fsi;
  if (!F.has_fields) throw Error(IFEM_E_BADPARAM, "ifem_fsi_find_fluid_bc: the solid carries no velocity / acceleration");
  if (ctx->indicator.n != (size_t)ctx->n_cells) throw Error(IFEM_E_BADPARAM, "ifem_fsi_find_fluid_bc: no cell indicator (ifem_fsi_update_indicator)");
  if (ctx->n_cells >= (int64_t(1) << 27)) throw Error(IFEM_E_BADPARAM, "ifem_fsi_find_fluid_bc: more than 2^27 local cells");
  if (cell_order)
    for (int64_t c = 0; c < ctx->n_cells; ++c)
      if (cell_order[c] < 0) throw Error(IFEM_E_BADPARAM, "ifem_fsi_find_fluid_bc: negative cell_order");
  hipStream_t st = ctx->stream;
  const int dim = ctx->dim, ncomp = dim * (dim + 1) / 2;
  const int64_t nUl = ctx->nUl, nloc = ctx->n_local;
  DBuf<int32_t> d_order;
  if (cell_order) d_order.upload(cell_order, (size_t)ctx->n_cells, st);
  if (ctx->fsi_stress.n != (size_t)ncomp * nUl) { // fluid_solver.fsi_stress starts at zero
    ctx->fsi_stress.alloc((size_t)ncomp * nUl);
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->fsi_stress.p, 0, ctx->fsi_stress.n * sizeof(double), st));
  }
  IFEM_HIP_CHECK(hipMemsetAsync(F.counters.p, 0, 8 * sizeof(int64_t), st));
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->vec[IFEM_VEC_FSI_ACC].p, 0, (size_t)nloc * sizeof(double), st)); // tmp_fsi_acceleration (:346-348)

  NodeBcArgs A{};
  A.kv = ctx->kv;
  A.nu = ctx->nu;
  A.nUl = nUl;
  A.dt = dt;
  A.vcoords = ctx->vcoords.p;
  A.cell_unodes = ctx->cell_unodes.p;
  A.present = ctx->vec[IFEM_VEC_PRESENT].p;
  A.fluid_stress = (ctx->stress_valid && ctx->stress.n == (size_t)dim * dim * nUl) ? ctx->stress.p : nullptr;
  A.s_vel = F.vel.p;
  A.s_acc = F.acc.p;
  A.s_stress = F.has_stress ? F.stress.p : nullptr;
  A.s_nv = F.nv;
  A.fsi_stress = ctx->fsi_stress.p;
  A.fsi_acc = ctx->vec[IFEM_VEC_FSI_ACC].p;
  A.counters = F.counters.p;

  // nodal fsi_stress (:415-480) and fsi_acceleration (:489-556): the nodes of the indicator cells
  const int mode_ind = (F.has_stress ? 1 : 0) | (use_dirichlet_bc ? 0 : 2);
  int64_t h_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0}, n_cand = 0, n_inside = 0;
  if (mode_ind) {
    build_first_touch(ctx, d_order.p, 0);
    A.first = F.first.p;
    A.cand = F.cand.p;
    A.mode = mode_ind;
    launch_node_bc(ctx, A);
    IFEM_HIP_CHECK(hipMemcpyAsync(h_cnt, F.counters.p, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    IFEM_HIP_CHECK(hipStreamSynchronize(st));
    n_cand = h_cnt[1];
    n_inside = h_cnt[2];
  }
  if (use_dirichlet_bc) { // Dirichlet lines of the artificial fluid (:569-651): every local cell, no in-cell support point
    for (int w = 0; w < 2; ++w)
      if (ctx->is_c[w].n != (size_t)nloc) {
        ctx->is_c[w].alloc((size_t)nloc);
        ctx->cval[w].alloc((size_t)nloc);
        IFEM_HIP_CHECK(hipMemsetAsync(ctx->is_c[w].p, 0, (size_t)nloc, st));
        IFEM_HIP_CHECK(hipMemsetAsync(ctx->cval[w].p, 0, (size_t)nloc * sizeof(double), st));
      }
    for (int w = 0; w < 2; ++w) { // what the sets were, for their identity afterwards
      if (F.prev[w].n != (size_t)nloc) F.prev[w].alloc((size_t)nloc);
      IFEM_HIP_CHECK(hipMemcpyAsync(F.prev[w].p, ctx->is_c[w].p, (size_t)nloc, hipMemcpyDeviceToDevice, st));
    }
    if (F.taken.n != (size_t)nloc) F.taken.alloc((size_t)nloc);
    hipLaunchKernelGGL(k_fsi_taken, grid_for(nloc), dim3(kBlock), 0, st, nloc, ctx->is_c[0].p, ctx->is_c[1].p, F.taken.p);
    if (ctx->hang.n) hipLaunchKernelGGL(k_fsi_mark, grid_for(ctx->hang.n), dim3(kBlock), 0, st, ctx->hang.n, ctx->hang.dof.p, F.taken.p);
    IFEM_HIP_CHECK(hipMemsetAsync(F.counters.p + 2, 0, sizeof(int64_t), st));
    build_first_touch(ctx, d_order.p, 1);
    A.first = F.first.p;
    A.cand = F.cand.p;
    A.mode = 4;
    A.taken = F.taken.p;
    A.is_c0 = ctx->is_c[0].p;
    A.is_c1 = ctx->is_c[1].p;
    A.cval0 = ctx->cval[0].p;
    A.cval1 = ctx->cval[1].p;
    ctx->inhom_any[1] = true; // the merged lines carry v_solid - present: inhomogeneous in general
    launch_node_bc(ctx, A);
    if (ctx->halo.nranks > 1) {
      // a ghost dof carries its owner's line: the owner saw every cell that touches the node (a ghost may be the master of a
      // local hanging node and lie in no local cell at all).  Every rank takes part, with or without ghosts of its own.
      DBuf<double> buf;
      buf.alloc((size_t)dim * std::max<int64_t>(nUl, 1));
      for (int what = 0; what < 2; ++what) {
        hipLaunchKernelGGL(k_fsi_lines_pack, grid_for(dim * nUl), dim3(kBlock), 0, st, dim * nUl, what, ctx->is_c[1].p, ctx->cval[1].p, buf.p);
        halo_exchange(ctx, buf.p);
        if (nUl > ctx->nUo)
          hipLaunchKernelGGL(k_fsi_lines_unpack, grid_for(dim * (nUl - ctx->nUo)), dim3(kBlock), 0, st, dim * nUl, dim * ctx->nUo, what, buf.p,
                             ctx->is_c[0].p, ctx->is_c[1].p, ctx->cval[1].p);
      }
      IFEM_HIP_CHECK(hipStreamSynchronize(st));
    }
    IFEM_HIP_CHECK(hipMemcpyAsync(h_cnt, F.counters.p, sizeof(h_cnt), hipMemcpyDeviceToHost, st));
    IFEM_HIP_CHECK(hipStreamSynchronize(st));
    n_cand = h_cnt[1];
    n_inside = h_cnt[2];
  }
  double nf = (double)h_cnt[4];
  allreduce_max(ctx, &nf, 1);
  if (stats) {
    stats->n_candidates = n_cand;
    stats->n_inside = n_inside;
    stats->n_lines = h_cnt[3];
    stats->n_not_found = (int64_t)nf;
  }
  // (the reference's "Cannot find point in solid", mpi_fsi.cpp:526-533, is raised at the END of this function: by then the
  // Dirichlet-merge kernels have written the flags of both sets, and a caller that catches the error must find the set
  // identities -- which decide whether B, B^T and S_m are reused -- in step with those flags)
  if (use_dirichlet_bc) { // identity of the two constrained-dof sets (ctx.hpp), in the order two ifem_set_constraints calls
                          // would decide it: set 0 against (old 0, old 1), then set 1 against (old 1, new 0)
    const DBuf<uint8_t> *pa[4] = {&ctx->is_c[0], &ctx->is_c[0], &ctx->is_c[1], &ctx->is_c[1]};
    const DBuf<uint8_t> *pb[4] = {&F.prev[0], &F.prev[1], &F.prev[1], &ctx->is_c[0]};
    double d[4];
    int64_t any[2] = {0, 0};
    IFEM_HIP_CHECK(hipMemsetAsync(F.counters.p + 5, 0, 2 * sizeof(int64_t), st));
    hipLaunchKernelGGL(k_fsi_any, dim3(1024), dim3(kBlock), 0, st, nloc, ctx->is_c[0].p, ctx->is_c[1].p, F.counters.p + 5);
    IFEM_HIP_CHECK(hipMemcpyAsync(any, F.counters.p + 5, sizeof(any), hipMemcpyDeviceToHost, st));
    flags_differ(ctx, 4, pa, pb, d); // synchronises
    ctx->has_c[0] = any[0] != 0;
    ctx->has_c[1] = any[1] != 0;
    constraint_set_identity(ctx, 0, d[0] != 0.0, d[1] != 0.0);
    constraint_set_identity(ctx, 1, d[2] != 0.0, d[3] != 0.0);
  }
  if (ctx->halo.nranks > 1) { // ghosts take the owner's values (fsi_acceleration and fsi_stress are ghosted vectors)
    if (!use_dirichlet_bc) halo_exchange(ctx, ctx->vec[IFEM_VEC_FSI_ACC].p);
    if (F.has_stress) {
      DBuf<double> buf;
      buf.alloc((size_t)dim * nUl);
      for (int c0 = 0; c0 < ncomp; c0 += dim) {
        const int nf_ = std::min(dim, ncomp - c0);
        hipLaunchKernelGGL(k_fsi_pack, grid_for(nUl), dim3(kBlock), 0, st, nUl, dim, nf_, ctx->fsi_stress.p + (size_t)c0 * nUl, buf.p);
        halo_exchange(ctx, buf.p);
        if (nUl > ctx->nUo)
          hipLaunchKernelGGL(k_fsi_unpack, grid_for(nUl - ctx->nUo), dim3(kBlock), 0, st, nUl, ctx->nUo, dim, nf_, buf.p, ctx->fsi_stress.p + (size_t)c0 * nUl);
      }
      IFEM_HIP_CHECK(hipStreamSynchronize(st));
    }
  }
  IFEM_HIP_CHECK(hipStreamSynchronize(st));
  if (nf != 0.0) throw Error(IFEM_E_BADPARAM, "Cannot find point in solid (mpi_fsi.cpp:526-533): " + std::to_string((int64_t)nf) + " support point(s)");
}


namespace {
template <int DIM> FluidBins<DIM> fluid_bins_of(const ifem_ctx *ctx) {
  const FsiState &F = ctx->fsi;
  FluidBins<DIM> B;
  B.ptr = F.fbin_ptr.p;
  B.cells = F.fbin_cells.p;
  for (int d = 0; d < 3; ++d) {
    B.lo[d] = F.fbox_lo[d];
    B.inv_h[d] = F.finv_h[d];
    B.G[d] = F.fG[d];
  }
  return B;
}
void build_fluid_bins(ifem_ctx *ctx) {
  FsiState &F = ctx->fsi;
  if (F.fbins_valid) return;
  hipStream_t st = ctx->stream;
  const int dim = ctx->dim, nv = 1 << dim;
  const int blocks = 256;
  DBuf<double> partial;
  partial.alloc((size_t)blocks * 2 * dim);
  hipLaunchKernelGGL(k_fluid_minmax, dim3(blocks), dim3(256), 0, st, ctx->n_cells * nv, dim, ctx->vcoords.p, partial.p);
  std::vector<double> hp = partial.download(st);
  double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  for (int d = 0; d < dim; ++d) {
    lo[d] = 1e300;
    hi[d] = -1e300;
    for (int b = 0; b < blocks; ++b) {
      lo[d] = std::min(lo[d], hp[(size_t)b * 2 * dim + 2 * d]);
      hi[d] = std::max(hi[d], hp[(size_t)b * 2 * dim + 2 * d + 1]);
    }
  }
  const int g = std::max(1, std::min(256, (int)std::lround(std::pow((double)ctx->n_cells, 1.0 / dim))));
  int64_t nb = 1;
  for (int d = 0; d < 3; ++d) {
    F.fG[d] = d < dim ? g : 1;
    F.fbox_lo[d] = lo[d];
    F.finv_h[d] = (d < dim && hi[d] > lo[d]) ? F.fG[d] / (hi[d] - lo[d]) : 0.0;
    nb *= F.fG[d];
  }
  DBuf<int32_t> count;
  count.alloc((size_t)nb);
  IFEM_HIP_CHECK(hipMemsetAsync(count.p, 0, (size_t)nb * sizeof(int32_t), st));
  F.fbin_ptr.alloc((size_t)nb + 1); // so that fluid_bins_of has its pointers; filled below
  const dim3 grid = grid_for(ctx->n_cells);
  if (dim == 2) hipLaunchKernelGGL(k_fluid_bin<2>, grid, dim3(kBlock), 0, st, 0, ctx->n_cells, ctx->vcoords.p, fluid_bins_of<2>(ctx), count.p, (int32_t *)nullptr);
  else hipLaunchKernelGGL(k_fluid_bin<3>, grid, dim3(kBlock), 0, st, 0, ctx->n_cells, ctx->vcoords.p, fluid_bins_of<3>(ctx), count.p, (int32_t *)nullptr);
  std::vector<int32_t> hc = count.download(st), ptr((size_t)nb + 1, 0);
  int64_t total = 0;
  for (int64_t b = 0; b < nb; ++b) {
    ptr[b] = (int32_t)total;
    total += hc[b];
    if (total > INT32_MAX) throw Error(IFEM_E_BADPARAM, "fluid bins: more than 2^31 (cell, bin) pairs");
  }
  ptr[nb] = (int32_t)total;
  F.fbin_ptr.upload(ptr.data(), ptr.size(), st);
  F.fbin_cells.alloc((size_t)std::max<int64_t>(total, 1));
  IFEM_HIP_CHECK(hipMemcpyAsync(count.p, F.fbin_ptr.p, (size_t)nb * sizeof(int32_t), hipMemcpyDeviceToDevice, st)); // cursors
  if (dim == 2) hipLaunchKernelGGL(k_fluid_bin<2>, grid, dim3(kBlock), 0, st, 1, ctx->n_cells, ctx->vcoords.p, fluid_bins_of<2>(ctx), count.p, F.fbin_cells.p);
  else hipLaunchKernelGGL(k_fluid_bin<3>, grid, dim3(kBlock), 0, st, 1, ctx->n_cells, ctx->vcoords.p, fluid_bins_of<3>(ctx), count.p, F.fbin_cells.p);
  IFEM_HIP_CHECK(hipStreamSynchronize(st));
  F.fbins_valid = true;
}
} // namespace

void fsi_fluid_at_points(ifem_ctx *ctx, int32_t n, const double *points, double *values, double *stress, int32_t *cell) {
  if (n < 0 || (n > 0 && (!points || !values || !cell))) throw Error(IFEM_E_BADPARAM, "ifem_fsi_fluid_at_points: null argument");
  if (n == 0) return;
  build_fluid_bins(ctx);
  hipStream_t st = ctx->stream;
  const int dim = ctx->dim;
  DBuf<double> d_pts, d_val, d_st;
  DBuf<int32_t> d_cell;
  d_pts.upload(points, (size_t)n * dim, st);
  d_val.alloc((size_t)n * (dim + 1));
  d_cell.alloc((size_t)n);
  if (stress) d_st.alloc((size_t)n * dim * dim);
  const double *fl = (ctx->stress_valid && ctx->stress.n == (size_t)dim * dim * ctx->nUl) ? ctx->stress.p : nullptr;
  if (dim == 2)
    hipLaunchKernelGGL(k_fluid_at_points<2>, grid_for(n, 128), dim3(128), 0, st, n, d_pts.p, ctx->n_cells, ctx->kv, ctx->nu, ctx->nUl, ctx->vcoords.p,
                       ctx->cell_unodes.p, ctx->cell_pnodes.p, ctx->vec[IFEM_VEC_PRESENT].p, fl, fluid_bins_of<2>(ctx), d_val.p, d_st.p, d_cell.p);
  else
    hipLaunchKernelGGL(k_fluid_at_points<3>, grid_for(n, 128), dim3(128), 0, st, n, d_pts.p, ctx->n_cells, ctx->kv, ctx->nu, ctx->nUl, ctx->vcoords.p,
                       ctx->cell_unodes.p, ctx->cell_pnodes.p, ctx->vec[IFEM_VEC_PRESENT].p, fl, fluid_bins_of<3>(ctx), d_val.p, d_st.p, d_cell.p);
  IFEM_HIP_CHECK(hipMemcpyAsync(values, d_val.p, (size_t)n * (dim + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
  IFEM_HIP_CHECK(hipMemcpyAsync(cell, d_cell.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
  if (stress) IFEM_HIP_CHECK(hipMemcpyAsync(stress, d_st.p, (size_t)n * dim * dim * sizeof(double), hipMemcpyDeviceToHost, st));
  IFEM_HIP_CHECK(hipStreamSynchronize(st));
}

} // namespace ifem
