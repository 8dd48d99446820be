// apply_mf.hip -- matrix-free application of the velocity block A_uu of InsIM's Newton matrix:
//   y_(a,c) = sum_q JxW { mu gN_a.g x_c + rho N_a (u.g x_c) + rho/dt N_a x_c + rho N_a x.g u_c + gamma rho d_c N_a div x }
// which is row (a,c) of Ke x summed over the cells (reference: the u-u part of local_matrix, mpi_insim.cpp:263-289, after
// distribute_local_to_global).  Used as the operator of the inner (preconditioner-only) Krylov solve that stands in for
// MUMPS (mpi_insim.cpp:124-127): the assembled block matrix stays the operator of the outer FGMRES.
//
// One wavefront per cell, sum factorisation on the tensor-product element: nodal values -> Gauss points by one 1D
// interpolation per direction, reference gradients by the collocation derivative on the Gauss points (exact: the
// restriction of Q_k to a grid line has degree k and there are k+1 points), the weak form at the quadrature point
// (one lane per point, MappingQ1 Jacobian from the cell vertices), then the transposed passes.  ~10 kFLOP and ~2.3 kB of
// HBM traffic per 3D Q2 cell instead of 40 B per stored block entry (64 entries per row) of the assembled SpMV.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstring>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

template <typename R>
struct MfTablesT {
  R N[9];  // [q][i]  1D Lagrange shape i (equidistant nodes) at Gauss point q
  R D[9];  // [q][q'] derivative at Gauss point q of the Lagrange polynomial of Gauss point q'
  R xi[3]; // Gauss points on [0,1]
  R w[3];
};
using MfTables = MfTablesT<double>;

// R = arithmetic type of the cell kernel: double for the operator of an outer solve, float for the preconditioner-only
// inner Krylov solve (tolerance 1e-2; its basis is single precision already).  Vectors in HBM stay double either way.
template <typename R, typename XT = double>
struct MfArgsT {
  int64_t n_cells, nUo; // n_cells: one past the last cell of this launch
  int64_t first_cell;   // first cell of this launch
  const double *vcoords;
  const int32_t *cell_unodes;
  const uint8_t *is_c; // constraint flags of the set the matrix was assembled with (local dofs) or nullptr
  const double *eval;  // evaluation point of the assembled matrix, velocity part, ghost-extended
  const XT *x;         // ghost-extended input (float: a level vector of the V-cycle)
  double *y;           // owned rows
  R mu, rho, gamma, inv_dt;
  int xcd;
  R *ycell; // per-cell results of the two-stage scatter
  MfTablesT<R> t;
};

template <int DIM, int N1>
struct MfGeo {
  static constexpr int NN = (DIM == 2) ? N1 * N1 : N1 * N1 * N1;
  static constexpr int NV = 1 << DIM;
};

// ---------------------------------------------------------------------------------------------------------------
// Half a wavefront (32 lanes) per cell, two cells per wave.  Every 1D pass is an in-register "pencil" operation
// (one lane owns the N1 values of a grid line: N1 LDS reads, N1^2 FMAs against wave-uniform table entries that live in
// SGPRs, N1 LDS writes, in place), the quadrature-point stage runs on 27 of 32 lanes, the trilinear geometry comes
// from 8 monomial coefficients, and the only synchronisation is wave-local (LDS operations of one wave execute in
// order).  ~330 wave instructions per 3D Q2 cell (the first, item-per-lane version of round 1 needed ~1200).
// Round 4 measured a variant with TWO cells per lane in packed arithmetic (v_pk_fma_f32 by construction, four cells per wave trip;
// profiles/r04_valu_rate.txt, r04_mf_packed.txt): 1.97 against 1.75 ms per application at 128^3 -- v_pk_fma_f32 issues in 4.7
// cycles against 4.1 for v_fma_f32 (1.7 x per FMA), but packed multiplies / adds gain nothing (the two-operand forms issue in 2.5
// cycles), the pair needs 195 registers (two waves per SIMD, where every instruction issues ~15 % slower) and 8-byte LDS items.
__device__ inline void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// LDS layout of the nodal / quadrature-point fields of a cell.  The pencil passes touch, per instruction, one entry of every
// (pencil, component) pair with one index fixed; the quadrature-point stage one entry per point with the component fixed.  On 3D
// Q2 the compact layout (component * 27 + i + 3 j + 9 k) puts two lanes of a 32-lane group on one bank in two of the three pass
// directions (29 % of the LDS cycles of rounds 1-3 were conflict cycles, and the LDS array is this kernel's busiest unit).  Entries
// that share a bank must differ in all four indices (i, j, k, component): the cosets of (1,1,1,1) in Z_3^4 are such sets, so entry
// (c; i,j,k) goes to row c (32 entries) at column (i-c) + 3 (j-c) + 9 (k-c), differences mod 3 -- every access pattern of the kernel
// is conflict-free.  Other element types keep the compact layout.
template <int DIM, int N1>
struct MfLay {
  static constexpr int NN = MfGeo<DIM, N1>::NN;
  static constexpr bool SWZ = DIM == 3 && N1 == 3;
  static constexpr int CS = SWZ ? 32 : NN; // entries per component
  static constexpr int SZ = DIM * CS;
  __device__ static int rot(int x) { return x < 0 ? x + N1 : x; }
  __device__ static int off(int c, int i0, int i1, int i2) {
    if constexpr (SWZ) return CS * c + rot(i0 - c) + 3 * rot(i1 - c) + 9 * rot(i2 - c);
    else return CS * c + i0 + N1 * i1 + N1 * N1 * i2;
  }
};

// NF fields per entry: the increment x and (CONV) the evaluation point u side by side, so that one ds_read_b64 (same LDS cycles as
// a ds_read_b32) fetches both
template <typename R, int NF>
struct alignas(sizeof(R) * NF) MfVal { R f[NF]; };

template <int DIM, int N1, typename R, int NF>
struct alignas(16) MfCell { // per-cell LDS scratch
  static constexpr int SZ = MfLay<DIM, N1>::SZ;
  MfVal<R, NF> V[SZ]; // nodal values -> values at the Gauss points; then, as R[SZ] over its head, integrand -> nodal result
  MfVal<R, NF> G[SZ]; // one reference-gradient direction at a time; R[SZ] over its head in the transposed passes
  R C[8 * DIM];       // [e][k]: monomial coefficients of the d-linear map
  R X[(1 << DIM) * DIM];
};

// CONV = false: the evaluation point is zero (InsIMEX matrix: no convective / Newton terms) -- the second field is neither gathered
// nor interpolated
template <int DIM, int KV, int WPB, bool CONV, typename R, typename XT>
__global__ __launch_bounds__(64 * WPB) void k_apply_uu_mf2(MfArgsT<R, XT> A) {
  constexpr int N1 = KV + 1, NN = MfGeo<DIM, N1>::NN, NV = 1 << DIM, NF = CONV ? 2 : 1;
  constexpr int NP = NN / N1;        // pencils per field and direction
  constexpr int NPL = DIM * NP;      // pencil lanes (one per pencil and component)
  using Lay = MfLay<DIM, N1>;
  using Val = MfVal<R, NF>;
  __shared__ MfCell<DIM, N1, R, NF> SS[2 * WPB];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, half = lane >> 5, hl = lane & 31;
  MfCell<DIM, N1, R, NF> &S = SS[2 * wave + half];
  R *const Vs = reinterpret_cast<R *>(S.V), *const Gs = reinterpret_cast<R *>(S.G); // single-field views (transposed passes)
  // ---- per-lane roles (fixed for the life of the wave)
  const bool pen_lane = hl < NPL;
  const int comp = pen_lane ? hl / NP : 0, pen = pen_lane ? hl % NP : 0;
  int padr[DIM][N1]; // entry of point i of this lane's pencil in direction d
  {
    const int a = pen % N1, b = pen / N1;
#pragma unroll
    for (int i = 0; i < N1; ++i) {
      if constexpr (DIM == 3) {
        padr[0][i] = Lay::off(comp, i, a, b); padr[1][i] = Lay::off(comp, a, i, b); padr[2][i] = Lay::off(comp, a, b, i);
      } else {
        padr[0][i] = Lay::off(comp, i, pen, 0); padr[1][i] = Lay::off(comp, pen, i, 0);
      }
    }
  }
  const bool q_lane = hl < NN;
  const int q = q_lane ? hl : 0;
  const int qi[3] = {q % N1, (q / N1) % N1, q / (N1 * N1)};
  int qoff[DIM]; // entry of node / quadrature point q, component c
#pragma unroll
  for (int c = 0; c < DIM; ++c) qoff[c] = Lay::off(c, qi[0], qi[1], DIM == 3 ? qi[2] : 0);
  R xi[3] = {0, 0, 0}, wq = 1;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    // select from the wave-uniform tables without dynamic indexing of kernel arguments
    R x_ = A.t.xi[0], w_ = A.t.w[0];
#pragma unroll
    for (int k = 1; k < N1; ++k) { x_ = qi[d] == k ? A.t.xi[k] : x_; w_ = qi[d] == k ? A.t.w[k] : w_; }
    xi[d] = x_; wq *= w_;
  }

  // every block owns one contiguous range of cell pairs (XCD-aware: neighbouring ranges run on the same XCD)
  const int64_t n_pairs = (A.n_cells - A.first_cell + 1) / 2;
  const int64_t per_block = (n_pairs + gridDim.x - 1) / gridDim.x;
  const int64_t vb = A.xcd ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x;
  const int64_t p_end = (vb + 1) * per_block < n_pairs ? (vb + 1) * per_block : n_pairs;
  // The gather of a cell is a chain of two dependent HBM accesses (node id, then the nodal values): it is software-
  // pipelined over the cells of the wave -- ids two cells ahead (issued at the top of a cell), values, constraint flags and
  // vertex coordinates one cell ahead (issued after the quadrature-point stage, where few registers are live) -- and
  // everything is kept RAW (double / byte) until it is consumed, so that no conversion forces a wait next to a load.
  const int64_t p_first = vb * per_block + wave;
  // 32-bit index arithmetic in the prefetch (cells * nodes-per-cell and dim * nodes are below 2^31 by the int32 node ids)
  auto cell_of = [&](int64_t pr) { const int64_t c = A.first_cell + 2 * pr + half; return (pr < p_end && c < A.n_cells) ? uint32_t(c) : 0u; };
  struct Pre { XT x[DIM]; double u[DIM], vc; uint8_t f[DIM]; } pre;
  auto load_id = [&](int64_t pr) -> int32_t { return q_lane ? A.cell_unodes[cell_of(pr) * uint32_t(NN) + uint32_t(hl)] : 0; };
  auto load_vals = [&](int64_t pr, int32_t nd, Pre &o) {
    if (q_lane) {
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const uint32_t dof = uint32_t(DIM) * uint32_t(nd) + uint32_t(c);
        o.x[c] = A.x[dof];
        if constexpr (CONV) o.u[c] = A.eval[dof];
        o.f[c] = A.is_c ? A.is_c[dof] : uint8_t(0); // one byte per dof: stays in L2
      }
    }
    if (hl < NV * DIM) o.vc = A.vcoords[int64_t(cell_of(pr)) * (NV * DIM) + hl];
  };
  int32_t nd_ahead = load_id(p_first + WPB);
  load_vals(p_first, load_id(p_first), pre);
  for (int64_t pair = p_first; pair < p_end; pair += WPB) {
    const int64_t cell = A.first_cell + 2 * pair + half;
    const bool active = cell < A.n_cells;
    // ---- gather: this cell's values are in registers; start the id load of the cell after the next
    const Pre cur = pre;
    const int32_t nd_next = nd_ahead;
    nd_ahead = load_id(pair + 2 * WPB);
    if (q_lane) {
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        Val v;
        v.f[0] = cur.f[c] != 0 ? R(0) : R(cur.x[c]);
        if constexpr (CONV) v.f[1] = R(cur.u[c]);
        S.V[qoff[c]] = v;
      }
    }
    if (hl < NV * DIM) S.X[hl] = R(cur.vc);
    wsync();
    // monomial coefficients of x(xi) = sum_k C_k prod_{d in k} xi_d:  C_k = sum_{v subset of k} (-1)^{|k|-|v|} X_v
    if (hl < NV * DIM) {
      const int k = hl / DIM, e = hl % DIM;
      R acc = 0;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const bool sub = (v & ~k) == 0;
        const int par = __builtin_popcount(k ^ v) & 1;
        const R xv = S.X[v * DIM + e];
        acc += sub ? (par ? -xv : xv) : R(0);
      }
      S.C[e * 8 + k] = acc;
    }
    // ---- nodal values -> Gauss points (both fields of an entry at once), in place
    if (pen_lane) {
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        Val in[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) in[i] = S.V[padr[d][i]];
#pragma unroll
        for (int o = 0; o < N1; ++o) {
          Val acc;
#pragma unroll
          for (int r = 0; r < NF; ++r) {
            R s = 0;
#pragma unroll
            for (int i = 0; i < N1; ++i) s += A.t.N[o * N1 + i] * in[i].f[r];
            acc.f[r] = s;
          }
          S.V[padr[d][o]] = acc;
        }
        wsync();
      }
    } else {
#pragma unroll
      for (int d = 0; d < DIM; ++d) wsync();
    }
    // ---- geometry at the quadrature point
    R Ji[DIM * DIM], JxW = 0;
    {
      R J[DIM * DIM];
      if constexpr (DIM == 3) {
        const R e_ = xi[1], z_ = xi[2], x_ = xi[0];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          typedef R R4 __attribute__((ext_vector_type(4)));
          const R4 lo = *reinterpret_cast<const R4 *>(&S.C[e * 8]), hi = *reinterpret_cast<const R4 *>(&S.C[e * 8 + 4]); // wave-uniform addresses: wide broadcast reads
          J[e * 3 + 0] = lo[1] + lo[3] * e_ + hi[1] * z_ + hi[3] * (e_ * z_);
          J[e * 3 + 1] = lo[2] + lo[3] * x_ + hi[2] * z_ + hi[3] * (x_ * z_);
          J[e * 3 + 2] = hi[0] + hi[1] * x_ + hi[2] * e_ + hi[3] * (x_ * e_);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const R c1 = S.C[e * 8 + 1], c2 = S.C[e * 8 + 2], c3 = S.C[e * 8 + 3];
          J[e * 2 + 0] = c1 + c3 * xi[1];
          J[e * 2 + 1] = c2 + c3 * xi[0];
        }
      }
      R det;
      if constexpr (DIM == 2) {
        det = J[0] * J[3] - J[1] * J[2];
        const R r = R(1) / det;
        Ji[0] = J[3] * r; Ji[1] = -J[1] * r; Ji[2] = -J[2] * r; Ji[3] = J[0] * r;
      } else {
        const R c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
        det = J[0] * c00 + J[1] * c01 + J[2] * c02;
        const R r = R(1) / det;
        Ji[0] = c00 * r; Ji[3] = c01 * r; Ji[6] = c02 * r;
        Ji[1] = (J[2] * J[7] - J[1] * J[8]) * r; Ji[4] = (J[0] * J[8] - J[2] * J[6]) * r; Ji[7] = (J[1] * J[6] - J[0] * J[7]) * r;
        Ji[2] = (J[1] * J[5] - J[2] * J[4]) * r; Ji[5] = (J[2] * J[3] - J[0] * J[5]) * r; Ji[8] = (J[0] * J[4] - J[1] * J[3]) * r;
      }
      JxW = fabs(det) * wq;
    }
    // ---- physical gradients: one reference direction at a time through G
    R gx[DIM][DIM], gu[DIM][DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c)
#pragma unroll
      for (int e = 0; e < DIM; ++e) { gx[c][e] = 0; gu[c][e] = 0; }
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      if (pen_lane) {
        Val in[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) in[i] = S.V[padr[d][i]];
#pragma unroll
        for (int o = 0; o < N1; ++o) {
          Val acc;
#pragma unroll
          for (int r = 0; r < NF; ++r) {
            R s = 0;
#pragma unroll
            for (int i = 0; i < N1; ++i) s += A.t.D[o * N1 + i] * in[i].f[r];
            acc.f[r] = s;
          }
          S.G[padr[d][o]] = acc;
        }
      }
      wsync();
      if (q_lane) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const Val g = S.G[qoff[c]];
#pragma unroll
          for (int e = 0; e < DIM; ++e) gx[c][e] += Ji[d * DIM + e] * g.f[0];
          if constexpr (CONV) {
#pragma unroll
            for (int e = 0; e < DIM; ++e) gu[c][e] += Ji[d * DIM + e] * g.f[1];
          }
        }
      }
      wsync();
    }
    // ---- weak form at the quadrature point
    R That[DIM][DIM]; // [comp][ref dir]
    if (q_lane) {
      R xq[DIM], uq[DIM];
#pragma unroll
      for (int c = 0; c < DIM; ++c) { const Val v = S.V[qoff[c]]; xq[c] = v.f[0]; uq[c] = CONV ? v.f[NF - 1] : R(0); }
      R divx = 0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) divx += gx[c][c];
      R mass[DIM];
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        R conv = 0, newt = 0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) { conv += uq[e] * gx[c][e]; newt += xq[e] * gu[c][e]; }
        mass[c] = JxW * A.rho * (conv + A.inv_dt * xq[c] + newt);
        R tp[DIM];
#pragma unroll
        for (int e = 0; e < DIM; ++e) tp[e] = JxW * (A.mu * gx[c][e] + (e == c ? A.gamma * A.rho * divx : R(0)));
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          R t = 0;
#pragma unroll
          for (int e = 0; e < DIM; ++e) t += Ji[d * DIM + e] * tp[e];
          That[c][d] = t;
        }
      }
      // the integrand goes to the single-field view over the head of V: every paired value of the wave has been read above
      // (the LDS executes one wave's operations in order)
#pragma unroll
      for (int c = 0; c < DIM; ++c) Vs[qoff[c]] = mass[c];
    }
    // ---- the next cell's values: their latency hides behind the transposed passes below
    load_vals(pair + WPB, nd_next, pre);
    // ---- V_c += D_d^T T^_{c,d}, one direction at a time
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      if (q_lane) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) Gs[qoff[c]] = That[c][d];
      }
      wsync();
      if (pen_lane) {
        R t[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) t[i] = Gs[padr[d][i]];
#pragma unroll
        for (int o = 0; o < N1; ++o) {
          R acc = Vs[padr[d][o]];
#pragma unroll
          for (int i = 0; i < N1; ++i) acc += A.t.D[i * N1 + o] * t[i];
          Vs[padr[d][o]] = acc;
        }
      }
      wsync();
    }
    // ---- transposed interpolation back to the nodes, in place
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      if (pen_lane) {
        R in[N1];
#pragma unroll
        for (int i = 0; i < N1; ++i) in[i] = Vs[padr[d][i]];
#pragma unroll
        for (int o = 0; o < N1; ++o) {
          R acc = 0;
#pragma unroll
          for (int i = 0; i < N1; ++i) acc += A.t.N[i * N1 + o] * in[i];
          Vs[padr[d][o]] = acc;
        }
      }
      wsync();
    }
    // ---- two-stage scatter: plain stores of the cell's result ([node][component]), summed per node by k_mf_gather
    // (atomics-free, deterministic); the DIM stores of a wave fill the same lines
    if (active && q_lane) {
#pragma unroll
      for (int c = 0; c < DIM; ++c) A.ycell[cell * (DIM * NN) + hl * DIM + c] = Vs[qoff[c]];
    }
    wsync();
  }
}

// Constrained rows of the assembled matrix carry only their diagonal: y_r = d_r x_r, d_r recovered from the
// inverse node block (row and column r of the block are zero apart from d_r, so its inverse has 1/d_r there).
// second stage of the atomics-free scatter: y_i = sum over the cells touching node(i) of the cell's local result (fixed
// order: deterministic), constrained rows y_r = d_r x_r as above
// FUSE (the multigrid smoother, solver.hip::mg_uu_smooth): the product t = (A x)_node is not stored but consumed on the
// spot -- fuse.mode 1: xs += x, r -= t (residual update after the coarse correction); mode 2: the Chebyshev step
// xs += x, r -= t, x <- a x + b B r with the inverse node block B (x is the smoother's direction vector and is updated in
// place: the cell kernel that read it has completed, and a thread only touches the entries of its own node).
template <int DIM, typename R, bool FUSE, typename V>
__global__ void k_mf_gather(int64_t n, int nn, const int64_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc,
                            const R *__restrict__ ycell, const uint8_t *__restrict__ is_c,
                            const double *__restrict__ bjac, const float *__restrict__ bjf, const V *x, double *y, MfFuseT<V> fuse) {
  // one thread per node: the incidence list is walked once for the DIM components (24 contiguous bytes per entry)
  const int64_t nd = int64_t(blockIdx.x) * blockDim.x + threadIdx.x;
  if (nd * DIM >= n) return;
  double s[DIM];
#pragma unroll
  for (int c = 0; c < DIM; ++c) s[c] = 0;
  const int64_t k0 = inc_ptr[nd], k1 = inc_ptr[nd + 1];
  uint8_t fl[DIM]; // constraint flags: requested now, looked at after the sums
#pragma unroll
  for (int c = 0; c < DIM; ++c) fl[c] = is_c ? is_c[nd * DIM + c] : uint8_t(0);
  // up to 8 incident cells per trip (a vertex node of a hexahedral mesh has 8): all incidence entries are loaded before
  // the dependent loads of the cell results; slots past the end re-read the first entry with weight 0
  for (int64_t k = k0; k < k1; k += 8) {
    int32_t e[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) e[u] = inc[k + u < k1 ? k + u : k0];
    R v[8][DIM];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const R *src = ycell + int64_t(e[u] >> 5) * (DIM * nn) + (e[u] & 31) * DIM;
#pragma unroll
      for (int c = 0; c < DIM; ++c) v[u][c] = src[c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (k + u < k1) {
#pragma unroll
        for (int c = 0; c < DIM; ++c) s[c] += v[u][c];
      }
  }
  if constexpr (!FUSE) {
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int64_t i = nd * DIM + c;
      y[i] = fl[c] ? x[i] / bjac[nd * DIM * DIM + c * DIM + c] : s[c];
    }
  } else {
    double xv[DIM], rv[DIM], bj[DIM * DIM];
#pragma unroll
    for (int e = 0; e < DIM * DIM; ++e) bj[e] = double(bjf[nd * DIM * DIM + e]);
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int64_t i = nd * DIM + c;
      xv[c] = double(x[i]);
      const double t = fl[c] ? xv[c] / bj[c * DIM + c] : s[c];
      rv[c] = double(fuse.r[i]) - t;
      fuse.xs[i] = V(double(fuse.xs[i]) + xv[c]);
      fuse.r[i] = V(rv[c]);
    }
    if (fuse.mode >= 2) {
      const double a = fuse.mode == 2 ? fuse.a : 0.0;
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double z = 0;
#pragma unroll
        for (int j = 0; j < DIM; ++j) z += bj[c * DIM + j] * rv[j];
        fuse.d[nd * DIM + c] = V(a * xv[c] + fuse.b * z);
      }
    }
  }
}

template <typename R>
static void mf_tables_to(MfTablesT<R> &o, const MfTables &t) {
  for (int i = 0; i < 9; ++i) { o.N[i] = R(t.N[i]); o.D[i] = R(t.D[i]); }
  for (int i = 0; i < 3; ++i) { o.xi[i] = R(t.xi[i]); o.w[i] = R(t.w[i]); }
}

static void mf_tables(MfTables &t, int kv) {
  const int n1 = kv + 1;
  std::memset(&t, 0, sizeof(t));
  if (n1 == 2) {
    const double a = 0.5 / std::sqrt(3.0);
    t.xi[0] = 0.5 - a; t.xi[1] = 0.5 + a; t.w[0] = t.w[1] = 0.5;
  } else {
    const double a = 0.5 * std::sqrt(0.6);
    t.xi[0] = 0.5 - a; t.xi[1] = 0.5; t.xi[2] = 0.5 + a;
    t.w[0] = t.w[2] = 5.0 / 18.0; t.w[1] = 8.0 / 18.0;
  }
  for (int q = 0; q < n1; ++q)
    for (int i = 0; i < n1; ++i) {
      double v = 1; // Lagrange shape i on the equidistant nodes j/kv
      for (int j = 0; j < n1; ++j)
        if (j != i) v *= (t.xi[q] - double(j) / kv) / (double(i) / kv - double(j) / kv);
      t.N[q * n1 + i] = v;
      double d = 0; // derivative at xi_q of the Lagrange polynomial of Gauss point i
      for (int k = 0; k < n1; ++k) {
        if (k == i) continue;
        double p = 1.0 / (t.xi[i] - t.xi[k]);
        for (int j = 0; j < n1; ++j)
          if (j != i && j != k) p *= (t.xi[q] - t.xi[j]) / (t.xi[i] - t.xi[j]);
        d += p;
      }
      t.D[q * n1 + i] = d;
    }
}

template <typename R, typename XT>
static void apply_uu_mf_t(ifem_ctx *ctx, const XT *xu, double *yu, const MfFuseT<XT> *fuse, int part) {
  if (!ctx->mf_valid) throw Error(IFEM_E_BADPARAM, "matrix-free A_uu: no assembled state (call ifem_ins_assemble first)");
  const int64_t n = int64_t(ctx->dim) * ctx->nUo;
  hipStream_t s = ctx->stream;
  MfArgsT<R, XT> a{};
  // cell range of this launch; with the interior-first tables (several ranks) every part uses them, part 0 included
  if (part != 0 && ctx->mf_n_interior < 0) throw Error(IFEM_E_BADPARAM, "matrix-free A_uu: cell split not built");
  const bool perm = ctx->mf_n_interior >= 0;
  a.first_cell = part == 2 ? ctx->mf_n_interior : 0;
  a.n_cells = part == 1 ? ctx->mf_n_interior : ctx->n_cells;
  a.nUo = ctx->nUo;
  a.vcoords = perm ? ctx->mf_vcoords.p : ctx->vcoords.p; a.cell_unodes = perm ? ctx->mf_cell_unodes.p : ctx->cell_unodes.p;
  a.is_c = ctx->has_c[ctx->asm_constraint_set] ? ctx->is_c[ctx->asm_constraint_set].p : nullptr;
  a.eval = ctx->mf_eval.p; a.x = xu; a.y = yu;
  a.mu = R(ctx->mf_params.viscosity); a.rho = R(ctx->mf_params.rho); a.gamma = R(ctx->mf_params.grad_div);
  a.inv_dt = R(1.0 / ctx->mf_params.dt);
  { MfTables t; mf_tables(t, ctx->kv); mf_tables_to(a.t, t); }
  a.xcd = ctx->tune.xcd_swizzle;
  const bool time_it = ctx->profile;
  if (ctx->n_cells >= (int64_t(1) << 26) || ctx->nu > 32) throw Error(IFEM_E_BADPARAM, "matrix-free A_uu: incidence entries hold 26 cell bits and 5 node bits");
  if (ctx->uinc.n_rows == 0 && ctx->nUo) build_incidence(ctx);
  const size_t need = size_t(ctx->n_cells) * ctx->dim * ctx->nu; // in doubles: the float variant uses half of it
  if (ctx->mf_ycell.n < need) ctx->mf_ycell.alloc(need);
  a.ycell = reinterpret_cast<R *>(ctx->mf_ycell.p);
  if (time_it) IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s)); // the cell kernel alone (what rocprofv3 reports for it)
  constexpr int WPB = 4;
  const dim3 block(64 * WPB);
  const int64_t n_pairs = (a.n_cells - a.first_cell + 1) / 2;
  const bool conv = !ctx->mf_noconv;
  // every block walks one contiguous range of cell pairs: the grid is a whole number of resident rounds (4 per CU slot)
  // so that no round runs partly empty
  auto grid_for_kernel = [&](const void *fn) {
    int per_cu = 0, dev = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 64 * WPB, 0) != hipSuccess || per_cu < 1) per_cu = 6;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    return unsigned(cus) * unsigned(per_cu) * 4u;
  };
  const unsigned g_all = unsigned(std::min<int64_t>((n_pairs + WPB - 1) / WPB, int64_t(1) << 30));
#define IFEM_MF2(D, K)                                                                                                 \
  { if (conv) { static const unsigned cap = grid_for_kernel(reinterpret_cast<const void *>(&k_apply_uu_mf2<D, K, WPB, true, R, XT>));  \
                hipLaunchKernelGGL((k_apply_uu_mf2<D, K, WPB, true, R, XT>), dim3(std::min(cap, g_all)), block, 0, s, a); }       \
    else { static const unsigned cap = grid_for_kernel(reinterpret_cast<const void *>(&k_apply_uu_mf2<D, K, WPB, false, R, XT>));     \
           hipLaunchKernelGGL((k_apply_uu_mf2<D, K, WPB, false, R, XT>), dim3(std::min(cap, g_all)), block, 0, s, a); } }
  if (n_pairs > 0) {
    // algorithmic traffic of the cell kernel: x, evaluation point (fp64 in HBM), constraint flags once per entry,
    // the per-cell results once, vertex coordinates and node ids per cell; flops of the sum-factorised passes + the point stage
    const int64_t nc = a.n_cells - a.first_cell;
    const int d = ctx->dim, n1 = ctx->kv + 1, npc = 1 << d;
    const double share = ctx->n_cells > 0 ? double(nc) / double(ctx->n_cells) : 0.0;
    const double passes = double((2 * d) * d * ctx->nu * n1 * 2 * 2 + d * d * ctx->nu * n1 * 2 * 2);
    KScope ks(ctx, IFEM_KC_MF_CELL, share * double(n) * (sizeof(XT) + 8 + 1) + double(nc) * (npc * d * 8 + ctx->nu * 4 + ctx->nu * d * sizeof(R)),
              double(nc) * (passes + ctx->nu * 190.0));
    if (ctx->dim == 3 && ctx->kv == 2) IFEM_MF2(3, 2)
    else if (ctx->dim == 3) IFEM_MF2(3, 1)
    else if (ctx->kv == 2) IFEM_MF2(2, 2)
    else IFEM_MF2(2, 1)
  }
#undef IFEM_MF2
  if (part == 1) return; // the node gather follows the boundary cells
  if (time_it) IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
  const MfFuseT<XT> f0 = fuse ? *fuse : MfFuseT<XT>{};
  // node gather: the per-cell results and the incidence entries once, the node's row pointer; fused form: the inverse node block
  // (single precision) and the smoother's vectors (mode 1: xs, r read + written, x read; modes 2 / 3: d written too)
  const double per_node = ctx->dim * double(sizeof(XT)) * (fuse ? (fuse->mode >= 2 ? 6.0 : 5.0) : 2.0) + (fuse ? 4.0 * ctx->dim * ctx->dim : 0.0) + 8.0 + ctx->dim;
  KScope ksg(ctx, IFEM_KC_MF_GATHER, double(ctx->n_cells) * ctx->nu * (ctx->dim * sizeof(R) + 4.0) + double(ctx->nUo) * per_node);
#define IFEM_MFG(D, F)                                                                                                 \
  hipLaunchKernelGGL((k_mf_gather<D, R, F, XT>), dim3(unsigned((n / D + 255) / 256)), dim3(256), 0, s, n, ctx->nu, ctx->uinc.rowptr.p, \
                     ctx->uinc.col.p, a.ycell, a.is_c, ctx->bjac.p, fuse ? bjac_f32_ptr(ctx) : nullptr, xu, yu, f0)
  if (ctx->dim == 3) { if (fuse) IFEM_MFG(3, true); else IFEM_MFG(3, false); }
  else { if (fuse) IFEM_MFG(2, true); else IFEM_MFG(2, false); }
#undef IFEM_MFG
  if (time_it) {
    IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->mf_ms_total += ms;
    ctx->timing.mf_calls++;
  }
}

// single = true: single-precision cell arithmetic (the inner, preconditioner-only solve); ifem_tuning::mf_f32 = 0 forces double
void apply_uu_mf(ifem_ctx *ctx, const double *xu, double *yu, bool single, const MfFuse *fuse, int part) {
  if (single && ctx->tune.mf_f32) apply_uu_mf_t<float, double>(ctx, xu, yu, fuse, part);
  else apply_uu_mf_t<double, double>(ctx, xu, yu, fuse, part);
}
void apply_uu_mf_f32v(ifem_ctx *ctx, const float *xu, const MfFuseT<float> *fuse, int part) {
  if (!fuse) throw Error(IFEM_E_BADPARAM, "apply_uu_mf_f32v: the fused form only");
  apply_uu_mf_t<float, float>(ctx, xu, nullptr, fuse, part);
}

} // namespace ifem
