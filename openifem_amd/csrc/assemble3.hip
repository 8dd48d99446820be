// assemble3.hip -- InsIM::assemble for the 3D Q2/Q1 element on the FP64 matrix cores (reference:
// source/mpi_insim.cpp:153-362; same mathematics, scatter and constraint handling as assemble2.hip).
//
// The velocity-velocity block of the element matrix is a sum over the 27 quadrature points of outer products,
//   Ke[(a,c),(b,d)] = sum_q  (wg ga_c)(q,a) gb_d(q,b)  +  (N_a rho w d_d u_c)(q) N_b(q)           (grad-div, Newton term)
//                 + d_cd sum_q [ sum_e (w mu ga_e) gb_e + (w rho N_a)(u.gb) + (w rho/dt N_a) N_b ]  (scalar part),
// i.e. for every (c,d) a 27x27 GEMM with K = 54 (+ a shared 27x27 GEMM with K = 135).  Two wavefronts per cell run them
// as v_mfma_f64_16x16x4 on 2x2 tiles of 16x16 (27 padded to 32, K padded to 28): 644 MFMAs per cell.  FP64 MFMA has the
// same peak as the vector FMA on MI355X -- the point is the instruction stream: an MFMA retires 2048 flops for two
// 8-byte operands per lane, where the vector path of assemble2.hip issues ~47 instructions per 58 flops.
// The operands come from per-cell LDS tables tabN[q][a], tabG[d][q][a] (23 KB), built once per cell; rhs, B/B^T, M_p
// read the same tables (default build: no tables, shapes rebuilt where consumed from tensor factors, Cell3Otf).  Accumulator
// layout of the instruction (tools/microbench.hip): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15], register r
// of the result is D[(l>>4) + 4r][l&15].
// The scatter is what bounds the kernel: the memory-side atomic path retires ~24 G requests of up to 64 bytes per second
// whatever the type, scope or footprint (tools/atomics_types.hip), so everything after the contraction is organised around the
// number of 64-byte segments a cell's 729 blocks touch (tools/scatter_sim.py replays it on the CPU): tile columns in the order
// of the cell's node ids (perm), the 16 node pairs of a matrix row ranked by their position in the row (rank_in_row16; the rows
// themselves are stored in scatter order, setup.hip::reorder_uu_rows), every staged row shifted to a segment boundary.
#include <hip/hip_runtime.h>
#include <mutex>
#include <set>
#include <type_traits>
#include "kernels.hpp"
#include "assemble_common.hpp"

namespace ifem {

typedef double d4 __attribute__((ext_vector_type(4)));

// Per-cell LDS.  TABLES layout (OTF = false): the physical gradient / value tables of the cell, 23 KB, read by the MFMA
// loop, the rhs and the B / B^T integrals: two cells fill half a CU's LDS, two waves per SIMD.  OTF layout: no per-cell
// tables -- every consumer rebuilds N_a(q) and grad N_a(q) on the fly from the workgroup's 1D / 2D tensor factors
// (Shared3) and the cell's inverse Jacobians Ji[q] (1.9 KB): 18.7 KB per cell, four workgroups per CU, so that the
// atomic-unit time of one cell's scatter overlaps the integration of three others instead of one.
struct Cell3Tabs {
  double tabG[3][27][27]; // physical shape gradients
  double tabN[27][27];    // shape values
};
struct Cell3Otf {
  double Ji[27 * 9];      // [q][reference direction e][physical direction d]
  double part1[27 * 18];  // partial nodal sums of the cell's second wave (phase 1)
};
template <bool OTF>
struct Cell3 : std::conditional<OTF, Cell3Otf, Cell3Tabs>::type {
  static constexpr int DIM = 3, NU = 27, NP = 8, NQ = 27, ND = 89, BS = 9;
  static constexpr int NODAL = 3 * NU * DIM + NP, STAGE = 64 * BS + 64;
  double X[NP * DIM], C[8 * DIM];
  double JxW[NQ], uq[NQ * DIM];
  double gqs[NQ * 9]; // rho JxW grad u
  // Ji[243] (TABLES layout) | Vc[243] | Sc[81] | divw[27] until the rhs is integrated, then the scatter staging of the cell's second wave
  double dead[STAGE];
  // nodal values (phase 1) | B entries in row order (second wave, uncached assemblies) | scatter staging of the cell's first wave
  double scratch[(STAGE > NODAL ? STAGE : NODAL) > NU * NP * DIM ? (STAGE > NODAL ? STAGE : NODAL) : NU * NP * DIM];
  double fe[ND], cv[ND];
  int64_t rs_uu[NU], rs_bt[NU], rs_b[NP], rs_mp[NP];
  int32_t len_uu[NU], len_bt[NU], len_b[NP], len_mp[NP];
  int32_t un[NU], pn[NP];
  int32_t bid[6], ind; // boundary ids of the faces (only read with Neumann conditions), FSI indicator of the cell
  uint8_t cf[ND + 7];
  uint8_t iperm[32], permp[8]; // rank of a velocity node among the cell's 27 (inverse of perm); pressure nodes in the order of their ids
  uint8_t perm[32];    // tile column -> local node, the cell's nodes in the order of their (local) node ids: the order of the columns
                       // in every row of the sorted pattern, so that neighbouring lanes of the staged scatter hit neighbouring blocks
};

struct Shared3 {
  Tab1D t;
  double psi[27 * 8];
};
// OTF layout: + the 2D tensor factors over (q0 q1) x (a0 a1): N2 = Nx Ny, DX2 = Nx' Ny, DY2 = Nx Ny'.  (A separate type: the
// TABLES layout fills 80 of a CU's 160 KB with two workgroups, 2 KB more would leave room for one.)
struct Shared3Otf : Shared3 {
  double N2[81], DX2[81], DY2[81];
};

// N_a(q) and the physical gradient of N_a at q from the tensor factors and the inverse Jacobian of the point: the same
// products and sums, in the same order, as the table build of the TABLES layout
__device__ __forceinline__ void shape_ref(const Shared3Otf &T, int q, int a, double &N, double r[3]) {
  const int i2 = (q % 9) * 9 + (a % 9), i1 = (q / 9) * 3 + a / 9;
  const double n2 = T.N2[i2], dx2 = T.DX2[i2], dy2 = T.DY2[i2], nz = T.t.N[i1], dz = T.t.dN[i1];
  N = n2 * nz;
  r[0] = dx2 * nz; r[1] = dy2 * nz; r[2] = n2 * dz;
}
__device__ __forceinline__ void shape_otf(const Shared3Otf &T, const double *__restrict__ Jq, int q, int a, double &N, double g[3]) {
  double r[3];
  shape_ref(T, q, a, N, r);
#pragma unroll
  for (int d = 0; d < 3; ++d) g[d] = r[0] * Jq[d] + r[1] * Jq[3 + d] + r[2] * Jq[6 + d];
}

// rank of `key` among the 16 lanes of my row of lanes (keys distinct): 15 row rotations by DPP, no LDS.  The staged scatter
// orders the 16 node pairs of a matrix row by their position in that row, so that neighbouring lanes of its atomics hit
// neighbouring blocks whatever order the row stores its blocks in (setup.hip: scatter order of the A_uu rows).
__device__ __forceinline__ int rank_in_row16(unsigned key) {
  int rank = 0;
#define IFEM_ROR16(n) rank += unsigned(__builtin_amdgcn_update_dpp(0, int(key), 0x120 + n, 0xf, 0xf, false)) < key ? 1 : 0;
  IFEM_ROR16(1) IFEM_ROR16(2) IFEM_ROR16(3) IFEM_ROR16(4) IFEM_ROR16(5) IFEM_ROR16(6) IFEM_ROR16(7) IFEM_ROR16(8)
  IFEM_ROR16(9) IFEM_ROR16(10) IFEM_ROR16(11) IFEM_ROR16(12) IFEM_ROR16(13) IFEM_ROR16(14) IFEM_ROR16(15)
#undef IFEM_ROR16
  return rank;
}

// CPB cells per workgroup, TWO wavefronts per cell (h = 0, 1) sharing the cell's LDS tables: the tables cap the
// workgroup at ~150 KB of LDS, and one wave per SIMD leaves every LDS / global round trip exposed; with two waves per
// cell the SIMDs hold two waves each.  Phases are separated by workgroup barriers (uniform control flow).
// WAVES: waves per SIMD the register allocation is held to (2: 256 registers, 3: 168, 4: 128).  The TABLES layout runs at 2
// (its LDS allows no more); the OTF layout is built for 3 (ks loop not unrolled: the whole kernel spills 80 bytes per lane,
// none of it inside the contraction) and 4 (240 bytes, most of it around the staged scatter).
template <int CPB, bool OTF, int WAVES>
__global__ __launch_bounds__(128 * CPB) __attribute__((amdgpu_waves_per_eu(WAVES ? WAVES : 1, WAVES ? WAVES : 4))) void k_ins_assemble3(AsmArgs A, Tab1D t1) {
  constexpr int DIM = 3, N1 = 3, NU = 27, NP = 8, NQ = 27, ND = 89, BS = 9;
  constexpr int NBP = NU * NP, BROUNDS = (NBP + 63) / 64, FR = (ND + 63) / 64;
  constexpr int SPAN = 16 * BS + 8; // lanes per staged matrix row in the scatter: 16 blocks + up to 7 lanes of alignment shift
  extern __shared__ __align__(16) unsigned char smem[];
  using Shared = typename std::conditional<OTF, Shared3Otf, Shared3>::type;
  Shared &T = *reinterpret_cast<Shared *>(smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, slot = wave >> 1, h = wave & 1;
  using Cell = Cell3<OTF>;
  Cell &S = *reinterpret_cast<Cell *>(smem + ((sizeof(Shared) + 15) & ~size_t(15)) + size_t(slot) * ((sizeof(Cell) + 15) & ~size_t(15)));
  double *Ji_, *part1;
  if constexpr (OTF) { Ji_ = S.Ji; part1 = S.part1; }
  else { Ji_ = S.dead; part1 = &S.tabG[0][0][0]; } // the tables are not built yet when the partial sums are parked there
  double *const Vc_ = S.dead + 243, *const Sc_ = S.dead + 486, *const divw_ = S.dead + 567;
  if (threadIdx.x < 9) { T.t.N[threadIdx.x] = t1.N[threadIdx.x]; T.t.dN[threadIdx.x] = t1.dN[threadIdx.x]; }
  if (threadIdx.x < 3) { T.t.xi[threadIdx.x] = t1.xi[threadIdx.x]; T.t.w[threadIdx.x] = t1.w[threadIdx.x]; }
  for (int i = threadIdx.x; i < NQ * NP; i += blockDim.x) {
    const int q = i / NP, b = i - q * NP;
    double v = 1;
    for (int d = 0; d < DIM; ++d) {
      const int qd = d == 0 ? q % N1 : (d == 1 ? (q / N1) % N1 : q / (N1 * N1));
      const double x = t1.xi[0] * (qd == 0) + t1.xi[1] * (qd == 1) + t1.xi[2] * (qd == 2);
      v *= ((b >> d) & 1) ? x : 1.0 - x;
    }
    T.psi[i] = v;
  }
  if constexpr (OTF)
    for (int i = threadIdx.x; i < 81; i += blockDim.x) {
      const int q01 = i / 9, a01 = i - q01 * 9;
      const int ix = (q01 % 3) * N1 + (a01 % 3), iy = (q01 / 3) * N1 + (a01 / 3);
      T.N2[i] = t1.N[ix] * t1.N[iy]; T.DX2[i] = t1.dN[ix] * t1.N[iy]; T.DY2[i] = t1.N[ix] * t1.dN[iy];
    }
  __syncthreads();

  const int64_t idx = int64_t(A.xcd_swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x) * CPB + slot;
  const bool active = idx < A.count;
  const int64_t cc = active ? (A.order ? int64_t(A.order[A.first + idx]) : idx) : 0;
  const int64_t p_off = int64_t(DIM) * A.nUl;
  double *ue = S.scratch, *u0e = S.scratch + NU * DIM, *ae = S.scratch + 2 * NU * DIM, *pe = S.scratch + 3 * NU * DIM;

  // ---- phase 0: ids, coordinates, nodal values, row descriptors, constraint flags.  One workgroup fills a CU's LDS, so
  // nothing hides a global round trip: every wave issues ALL its loads (two dependent rounds: ids, then everything keyed
  // by the id) before the first LDS store that needs one.  Optional arrays fall back to a valid address + select.
  if (h == 0) {
    const int a = lane < NU ? lane : 0;
    const int32_t nd = A.cell_unodes[cc * NU + a];
    int32_t bid = -1;
    if (A.n_neumann != 0 && lane < 2 * DIM) bid = A.cell_face_bid[cc * 2 * DIM + lane];
    const bool own = nd < A.nUo;
    const int64_t ndr = own ? nd : 0;
    const int64_t r0 = A.rp_uu[ndr], r1 = A.rp_uu[ndr + 1], t0 = A.rp_bt[ndr], t1_ = A.rp_bt[ndr + 1];
    const int64_t dof = int64_t(DIM) * nd;
    const double *fa = A.fsi_acc ? A.fsi_acc : A.eval, *cvp = A.cval ? A.cval : A.eval;
    const uint8_t *icp = A.is_c ? A.is_c : reinterpret_cast<const uint8_t *>(A.eval);
    double ev[DIM], pv[DIM], av[DIM], cvv[DIM];
    uint8_t cfv[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      ev[c] = A.eval[dof + c]; pv[c] = A.present[dof + c]; av[c] = fa[dof + c]; cvv[c] = cvp[dof + c]; cfv[c] = icp[dof + c];
    }
    if (lane < NU) {
      S.un[a] = nd;
      S.rs_uu[a] = own ? r0 : 0; S.len_uu[a] = own ? int32_t(r1 - r0) : -1;
      S.rs_bt[a] = own ? t0 : 0; S.len_bt[a] = own ? int32_t(t1_ - t0) : -1;
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        ue[a * DIM + c] = ev[c];
        u0e[a * DIM + c] = pv[c];
        ae[a * DIM + c] = A.fsi_acc ? av[c] : 0.0;
        S.cf[a * DIM + c] = A.is_c ? cfv[c] : uint8_t(0);
        S.cv[a * DIM + c] = A.cval ? cvv[c] : 0.0;
      }
    }
    if (lane < 2 * DIM) S.bid[lane] = bid;
    for (int i = lane; i < ND; i += 64) S.fe[i] = 0.0;
  } else {
    const int b = lane < NP ? lane : 0;
    const int32_t nd = A.cell_pnodes[cc * NP + b];
    const double xv = A.vcoords[cc * NP * DIM + (lane < NP * DIM ? lane : 0)];
    const int32_t indv = (A.indicator && lane == 0) ? A.indicator[cc] : 0;
    const bool own = nd < A.nPo;
    const int64_t ndr = own ? nd : 0;
    const int64_t r0 = A.rp_b[ndr], r1 = A.rp_b[ndr + 1], m0 = A.rp_mp[ndr], m1 = A.rp_mp[ndr + 1];
    const double *cvp = A.cval ? A.cval : A.eval;
    const uint8_t *icp = A.is_c ? A.is_c : reinterpret_cast<const uint8_t *>(A.eval);
    const double pev = A.eval[p_off + nd], cvv = cvp[p_off + nd];
    const uint8_t cfv = icp[p_off + nd];
    if (lane < NP * DIM) S.X[lane] = xv;
    if (lane == 0) S.ind = indv;
    if (lane < NP) {
      S.pn[b] = nd;
      S.rs_b[b] = own ? r0 : 0; S.len_b[b] = own ? int32_t(r1 - r0) : -1;
      S.rs_mp[b] = own ? m0 : 0; S.len_mp[b] = own ? int32_t(m1 - m0) : -1;
      pe[b] = pev;
      S.cf[NU * DIM + b] = A.is_c ? cfv : uint8_t(0);
      S.cv[NU * DIM + b] = A.cval ? cvv : 0.0;
    }
  }
  __syncthreads();
  if (h == 1 && lane < 32) { // rank of every node id among the cell's 27 (ids are distinct); columns 27..31 of the tiles stay padding
    int rank = lane;
    if (lane < NU) {
      const int32_t mine = S.un[lane];
      rank = 0;
#pragma unroll 9
      for (int j = 0; j < NU; ++j) rank += S.un[j] < mine ? 1 : 0;
    }
    S.perm[rank] = uint8_t(lane);
    S.iperm[lane] = uint8_t(rank);
  }
  if (h == 1 && lane >= 32 && lane < 32 + NP) { // the same for the 8 pressure nodes (B^T and M_p rows are in column order)
    const int32_t mine = S.pn[lane - 32];
    int rank = 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) rank += S.pn[j] < mine ? 1 : 0;
    S.permp[rank] = uint8_t(lane - 32);
  }
  if (h == 0 && lane < NP * DIM) { // monomial coefficients of the trilinear map
    const int k = lane / DIM, e = lane % DIM;
    double acc = 0;
#pragma unroll
    for (int v = 0; v < NP; ++v) {
      const bool sub = (v & ~k) == 0;
      const int par = __builtin_popcount(k ^ v) & 1;
      const double xv = S.X[v * DIM + e];
      acc += sub ? (par ? -xv : xv) : 0.0;
    }
    S.C[k * DIM + e] = acc;
  }
  __syncthreads();
  const int ind = active ? S.ind : 0;

  // ---- phase 1: per quadrature point (lane = q): Jacobian, fields of the evaluation point, rhs coefficients
  // the nodal sums are split over the two waves of the cell: the second wave handles nodes 14..26 and parks its partial
  // sums in the (not yet built) gradient table
  // (part1: [27 lanes][18])
  if (h == 1 && lane < NQ) {
    const int q = lane;
    const int qi[3] = {q % N1, (q / N1) % N1, q / (N1 * N1)};
    double u[3] = {0, 0, 0}, u0[3] = {0, 0, 0}, ac[3] = {0, 0, 0}, gr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) gr[i] = 0;
#pragma unroll 1
    for (int a = 14; a < (A.debug_skip == 6 ? 15 : NU); ++a) {
      const int ai[3] = {a % N1, (a / N1) % N1, a / (N1 * N1)};
      const double nx = T.t.N[qi[0] * N1 + ai[0]], ny = T.t.N[qi[1] * N1 + ai[1]], nz = T.t.N[qi[2] * N1 + ai[2]];
      const double dx = T.t.dN[qi[0] * N1 + ai[0]], dy = T.t.dN[qi[1] * N1 + ai[1]], dz = T.t.dN[qi[2] * N1 + ai[2]];
      const double N = nx * ny * nz, dr[3] = {dx * ny * nz, nx * dy * nz, nx * ny * dz};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double uv = ue[a * 3 + c];
        u[c] += N * uv; u0[c] += N * u0e[a * 3 + c]; ac[c] += N * ae[a * 3 + c];
#pragma unroll
        for (int e = 0; e < 3; ++e) gr[c * 3 + e] += uv * dr[e];
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) { part1[q * 18 + c] = u[c]; part1[q * 18 + 3 + c] = u0[c]; part1[q * 18 + 6 + c] = ac[c]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) part1[q * 18 + 9 + i] = gr[i];
  }
  double u_[3] = {0, 0, 0}, u0_[3] = {0, 0, 0}, ac_[3] = {0, 0, 0}, gr_[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (h == 0 && lane < NQ) { // first half of the nodes, in registers across the barrier
    const int q = lane;
    const int qi[3] = {q % N1, (q / N1) % N1, q / (N1 * N1)};
#pragma unroll 1
    for (int a = 0; a < (A.debug_skip == 6 ? 1 : 14); ++a) {
      const int ai[3] = {a % N1, (a / N1) % N1, a / (N1 * N1)};
      const double nx = T.t.N[qi[0] * N1 + ai[0]], ny = T.t.N[qi[1] * N1 + ai[1]], nz = T.t.N[qi[2] * N1 + ai[2]];
      const double dx = T.t.dN[qi[0] * N1 + ai[0]], dy = T.t.dN[qi[1] * N1 + ai[1]], dz = T.t.dN[qi[2] * N1 + ai[2]];
      const double N = nx * ny * nz, dr[3] = {dx * ny * nz, nx * dy * nz, nx * ny * dz};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double uv = ue[a * 3 + c];
        u_[c] += N * uv; u0_[c] += N * u0e[a * 3 + c]; ac_[c] += N * ae[a * 3 + c];
#pragma unroll
        for (int e = 0; e < 3; ++e) gr_[c * 3 + e] += uv * dr[e];
      }
    }
  }
  __syncthreads();
  if (h == 0 && lane < NQ) {
    const int q = lane;
    const int qi[3] = {q % N1, (q / N1) % N1, q / (N1 * N1)};
    double xi[3], wq = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { xi[d] = T.t.xi[qi[d]]; wq *= T.t.w[qi[d]]; }
    double J[9], Ji[9];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const double c1 = S.C[1 * 3 + e], c2 = S.C[2 * 3 + e], c3 = S.C[3 * 3 + e], c4 = S.C[4 * 3 + e], c5 = S.C[5 * 3 + e],
                   c6 = S.C[6 * 3 + e], c7 = S.C[7 * 3 + e];
      J[e * 3 + 0] = c1 + c3 * xi[1] + c5 * xi[2] + c7 * (xi[1] * xi[2]);
      J[e * 3 + 1] = c2 + c3 * xi[0] + c6 * xi[2] + c7 * (xi[0] * xi[2]);
      J[e * 3 + 2] = c4 + c5 * xi[0] + c6 * xi[1] + c7 * (xi[0] * xi[1]);
    }
    const double det = inv_small<3>(J, Ji);
    const double w = fabs(det) * wq;
    double u[3], u0[3], ac[3], gr[9], p = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) { u[c] = u_[c] + part1[q * 18 + c]; u0[c] = u0_[c] + part1[q * 18 + 3 + c]; ac[c] = ac_[c] + part1[q * 18 + 6 + c]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) gr[i] = gr_[i] + part1[q * 18 + 9 + i];
#pragma unroll
    for (int b = 0; b < NP; ++b) p += T.psi[q * NP + b] * pe[b];
    double g[9], dv = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        double t = 0;
#pragma unroll
        for (int e = 0; e < 3; ++e) t += gr[c * 3 + e] * Ji[e * 3 + d];
        g[c * 3 + d] = t;
      }
    dv = g[0] + g[4] + g[8];
    S.JxW[q] = w;
    divw_[q] = w * dv;
#pragma unroll
    for (int i = 0; i < 9; ++i) { Ji_[q * 9 + i] = Ji[i]; S.gqs[q * 9 + i] = A.imex ? 0.0 : A.rho * w * g[i]; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      S.uq[q * 3 + c] = A.imex ? 0.0 : u[c]; // only the matrix reads uq (u . grad N_b): no convection in the IMEX matrix
      double adv = 0, vc[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        adv += g[c * 3 + d] * u[d];
        vc[d] = w * (-A.mu * g[c * 3 + d] + (c == d ? p - A.gamma * A.rho * dv : 0.0));
      }
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if constexpr (OTF) Vc_[(q * 3 + c) * 3 + d] = Ji[d * 3] * vc[0] + Ji[d * 3 + 1] * vc[1] + Ji[d * 3 + 2] * vc[2]; // reference-gradient basis
        else Vc_[(q * 3 + c) * 3 + d] = vc[d];
      }
      double sc = -A.rho * adv - A.rho * A.inv_dt * (u[c] - u0[c]) + A.rho * A.g[c];
      if (ind == 1) sc += A.rho * ac[c];
      Sc_[q * 3 + c] = w * sc;
    }
  }
  __syncthreads();
  // ---- node tables, once per cell: tabN[q][a], tabG[d][q][a] (both waves, interleaved rounds)
  if constexpr (!OTF)
  for (int t = lane + 64 * h; t < (A.debug_skip == 7 ? 128 : NQ * NU); t += 128) {
    const int q = t / NU, a = t - q * NU;
    const int qi[3] = {q % N1, (q / N1) % N1, q / (N1 * N1)}, ai[3] = {a % N1, (a / N1) % N1, a / (N1 * N1)};
    const double nx = T.t.N[qi[0] * N1 + ai[0]], ny = T.t.N[qi[1] * N1 + ai[1]], nz = T.t.N[qi[2] * N1 + ai[2]];
    const double dx = T.t.dN[qi[0] * N1 + ai[0]], dy = T.t.dN[qi[1] * N1 + ai[1]], dz = T.t.dN[qi[2] * N1 + ai[2]];
    const double dr[3] = {dx * ny * nz, nx * dy * nz, nx * ny * dz};
    S.tabN[q][a] = nx * ny * nz;
#pragma unroll
    for (int d = 0; d < 3; ++d) S.tabG[d][q][a] = dr[0] * Ji_[q * 9 + d] + dr[1] * Ji_[q * 9 + 3 + d] + dr[2] * Ji_[q * 9 + 6 + d];
  }
  __syncthreads();
  // ---- Neumann (pressure) boundary faces  (:313-341)
  if (A.n_neumann != 0 && active && h == 0) {
    for (int f = 0; f < 2 * DIM; ++f) {
      const int bid = S.bid[f];
      if (bid < 0) continue;
      double pbc = 0; bool hit = false;
      for (int k = 0; k < A.n_neumann; ++k) if (A.neumann_id[k] == bid) { pbc = A.neumann_p[k]; hit = true; }
      if (!hit) continue;
      const int nd = f >> 1; const double sgn = (f & 1) ? 1.0 : -1.0;
      for (int i = lane; i < NU * DIM; i += 64) {
        const int a = i / DIM, c = i - a * DIM;
        double acc = 0;
#pragma unroll 1
        for (int qf = 0; qf < A.fe->nqf; ++qf) {
          double J[9], Ji[9];
          for (int k = 0; k < 9; ++k) J[k] = 0;
          const double *dps = &A.fe->fdpsi[(f * A.fe->nqf + qf) * NP * DIM];
          for (int v = 0; v < NP; ++v)
            for (int d = 0; d < DIM; ++d)
              for (int e = 0; e < DIM; ++e) J[d * DIM + e] += S.X[v * DIM + d] * dps[v * DIM + e];
          const double det = inv_small<3>(J, Ji);
          double nv[3], nn = 0;
          for (int d = 0; d < DIM; ++d) { nv[d] = sgn * Ji[nd * DIM + d]; nn += nv[d] * nv[d]; }
          nn = sqrt(nn);
          acc += A.fe->fphi[(f * A.fe->nqf + qf) * NU + a] * (nv[c] / nn) * pbc * fabs(det) * nn * A.fe->fw[qf];
        }
        unsafeAtomicAdd(&S.fe[i], -acc); // the cell's second wave adds to S.fe concurrently
      }
    }
  }
  wsync2();
  // ---- local rhs (:281-304) from the tables: first wave; the second wave integrates B / B^T and M_p meanwhile
  if (h == 0)
#pragma unroll
  for (int k = 0; k < FR; ++k) {
    const int i = lane + 64 * k;
    double f = 0;
    if (i < NU * DIM) {
      const int a = i / DIM, c = i - a * DIM;
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) {
        if constexpr (OTF) {
          double N, r[3];
          shape_ref(T, q, a, N, r);
          f += Sc_[q * 3 + c] * N + Vc_[(q * 3 + c) * 3 + 0] * r[0] + Vc_[(q * 3 + c) * 3 + 1] * r[1] + Vc_[(q * 3 + c) * 3 + 2] * r[2];
        } else
        f += Sc_[q * 3 + c] * S.tabN[q][a] + Vc_[(q * 3 + c) * 3 + 0] * S.tabG[0][q][a] + Vc_[(q * 3 + c) * 3 + 1] * S.tabG[1][q][a] +
             Vc_[(q * 3 + c) * 3 + 2] * S.tabG[2][q][a];
      }
    } else if (i < ND) {
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) f += divw_[q] * T.psi[q * NP + (i - NU * DIM)];
    }
    if (i < ND) unsafeAtomicAdd(&S.fe[i], f);
  }

  // ---- velocity-pressure blocks: -JxW psi_b grad N_a
  bool need_b = !A.rhs_only && A.debug_skip < 3 && active && h == 1;
  if (need_b && A.skip_geo) { // cached blocks: only a cell with an inhomogeneous constrained dof still needs the entries
    const bool mine = (lane < ND && S.cf[lane] && S.cv[lane] != 0.0) || (lane + 64 < ND && S.cf[lane + 64] && S.cv[lane + 64] != 0.0);
    need_b = A.use_inhom && __any(mine);
  }
  // Lanes = (velocity node a, pressure node in id order): the eight entries of a B^T row land next to each other.  The B entries go
  // through LDS (bst, the idle scratch zone) into the order (pressure node, velocity nodes by id) = the order of B's rows, so that
  // neighbouring lanes of its atomics hit neighbouring entries too: 984 -> ~530 64-byte segments per cell for B, B^T and M_p
  double *const bst = S.scratch;
  if (need_b) {
#pragma unroll 1
    for (int k = 0; k < BROUNDS; ++k) {
      const int t = lane + 64 * k;
      if (t >= NBP) continue;
      const int a = t / NP, pb = S.permp[t - a * NP];
      double v[3] = {0, 0, 0};
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) {
        const double wpsi = S.JxW[q] * T.psi[q * NP + pb];
        if constexpr (OTF) {
          double N, g[3];
          shape_otf(T, Ji_ + q * 9, q, a, N, g);
          v[0] -= wpsi * g[0]; v[1] -= wpsi * g[1]; v[2] -= wpsi * g[2];
        } else {
        v[0] -= wpsi * S.tabG[0][q][a]; v[1] -= wpsi * S.tabG[1][q][a]; v[2] -= wpsi * S.tabG[2][q][a];
        }
      }
      const bool pc = S.cf[NU * DIM + pb];
      if (S.len_bt[a] >= 0) {
        const int len = S.len_bt[a];
        double *base = A.v_bt + S.rs_bt[a] * DIM + A.posUP[(cc * NU + a) * NP + pb];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          if (S.cf[a * DIM + c]) continue;
          if (!pc) { if (!A.skip_geo) unsafeAtomicAdd(base + int64_t(c) * len, v[c]); }
          else if (A.use_inhom && S.cv[NU * DIM + pb] != 0.0) unsafeAtomicAdd(&S.fe[a * DIM + c], -v[c] * S.cv[NU * DIM + pb]);
        }
      }
      const bool brow = S.len_b[pb] >= 0 && !pc;
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        double w = 0.0;
        if (brow) {
          if (!S.cf[a * DIM + c]) w = v[c];
          else if (A.use_inhom && S.cv[a * DIM + c] != 0.0) unsafeAtomicAdd(&S.fe[NU * DIM + pb], -v[c] * S.cv[a * DIM + c]);
        }
        if (!A.skip_geo) bst[(pb * NU + S.iperm[a]) * DIM + c] = w;
      }
    }
    if (!A.skip_geo) { // B in row order: lane = (pressure node, velocity node by id), one plane per instruction
      wsync2();
#pragma unroll 1
      for (int k = 0; k < BROUNDS; ++k) {
        const int t = lane + 64 * k;
        if (t >= NBP) continue;
        const int pb = t / NU, a = S.perm[t - pb * NU];
        if (S.len_b[pb] < 0) continue;
        const int len = S.len_b[pb];
        double *base = A.v_b + S.rs_b[pb] * DIM + A.posPU[(cc * NP + pb) * NU + a];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const double w = bst[t * DIM + c];
          if (w != 0.0) unsafeAtomicAdd(base + int64_t(c) * len, w);
        }
      }
    }
  }
  // ---- pressure mass matrix M_p and diag(M_u)
  if (!A.rhs_only && !A.skip_geo && A.debug_skip < 4) {
    if (h == 1 && lane < NP * NP) {
      const int pa = lane / NP, pb = S.permp[lane - pa * NP];
      double m = 0;
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) m += S.JxW[q] * T.psi[q * NP + pa] * T.psi[q * NP + pb];
      if (active && S.len_mp[pa] >= 0) {
        const bool ra = S.cf[NU * DIM + pa], cb = S.cf[NU * DIM + pb];
        double *dst = A.v_mp + S.rs_mp[pa] + A.posPP[(cc * NP + pa) * NP + pb];
        if (!ra && !cb) unsafeAtomicAdd(dst, m);
        else if (ra && pa == pb) unsafeAtomicAdd(dst, fabs(m));
      }
    }
    if (h == 0 && lane < NU) {
      double m = 0;
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) {
        double N;
        if constexpr (OTF) { double r[3]; shape_ref(T, q, lane, N, r); }
        else N = S.tabN[q][lane];
        m += S.JxW[q] * N * N;
      }
      if (active && S.len_uu[lane] >= 0)
        for (int c = 0; c < DIM; ++c) unsafeAtomicAdd(&A.diagMu[int64_t(DIM) * S.un[lane] + c], m);
    }
  }
  __syncthreads(); // rhs, B / B^T and M_p are integrated: the dead zone may be reused, S.fe is complete up to the scatter corrections
  // most cells carry no constrained dof: a wave-uniform flag lets their scatter skip the per-entry constraint logic
  bool any_c;
  {
    bool mine = false;
    for (int i = lane; i < ND; i += 64) mine = mine || S.cf[i];
    any_c = __any(mine);
  }
  const double wgam = A.gamma * A.rho, rdt = A.rho * A.inv_dt;
  double *stage = h == 0 ? S.scratch : S.dead; // the second wave stages in the dead zone (Ji, Vc, Sc, divw are consumed)
  int64_t *soff = reinterpret_cast<int64_t *>(stage + 64 * BS);
  // ---- velocity-velocity block on the matrix cores, one 16x16 tile pair (ti, tj) at a time
  if (!A.rhs_only && !A.skip_uu && A.debug_skip < 5) {
#pragma unroll 1
    for (int tp = 2 * h; tp < 2 * h + 2; ++tp) {
      const int ti = tp >> 1, tj = tp & 1;
      const int al = 16 * ti + (lane & 15), bl = S.perm[16 * tj + (lane & 15)]; // my A-row node, my B-column node (columns in node-id order)
      const bool av = al < NU, bv = bl < NU;
      const int ac_ = av ? al : 0, bc_ = bv ? bl : 0;
      d4 acc[BS], sac = {0, 0, 0, 0};
#pragma unroll
      for (int e = 0; e < BS; ++e) acc[e] = d4{0, 0, 0, 0};
      // scatter positions of my four pairs: loaded now, needed after the contraction
      uint16_t posr[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * ti + (lane >> 4) + 4 * r;
        posr[r] = A.posUU[(cc * NU + (a < NU ? a : 0)) * NU + bc_]; // unconditional (clamped): no branch, no wait here
      }
      if (A.debug_skip != 2) {
#pragma unroll OTF ? 1 : 7
        for (int ks = 0; ks < 7; ++ks) {
          const int q = 4 * ks + (lane >> 4);
          const bool qv = q < NQ;
          const int qq = qv ? q : 0;
          const double ma = (av && qv) ? 1.0 : 0.0, mb = (bv && qv) ? 1.0 : 0.0; // padding rows / columns / points contribute 0
          const double w = S.JxW[qq];
          double Na, Nb, ga[3], gb[3];
          if constexpr (OTF) {
            shape_otf(T, Ji_ + qq * 9, qq, ac_, Na, ga);
            shape_otf(T, Ji_ + qq * 9, qq, bc_, Nb, gb);
            Na *= ma; Nb *= mb;
#pragma unroll
            for (int d = 0; d < 3; ++d) { ga[d] *= ma; gb[d] *= mb; }
          } else {
            Na = ma * S.tabN[qq][ac_]; Nb = mb * S.tabN[qq][bc_];
#pragma unroll
            for (int d = 0; d < 3; ++d) { ga[d] = ma * S.tabG[d][qq][ac_]; gb[d] = mb * S.tabG[d][qq][bc_]; }
          }
          const double ugb = S.uq[qq * 3] * gb[0] + S.uq[qq * 3 + 1] * gb[1] + S.uq[qq * 3 + 2] * gb[2];
          const double wmu = w * A.mu, wNa = w * Na;
          // scalar part
          sac = __builtin_amdgcn_mfma_f64_16x16x4f64(wmu * ga[0], gb[0], sac, 0, 0, 0);
          // grad-div part of the nine blocks (independent accumulators between dependent MFMAs)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const double wga = w * wgam * ga[c];
#pragma unroll
            for (int d = 0; d < 3; ++d) acc[c * 3 + d] = __builtin_amdgcn_mfma_f64_16x16x4f64(wga, gb[d], acc[c * 3 + d], 0, 0, 0);
          }
          sac = __builtin_amdgcn_mfma_f64_16x16x4f64(wmu * ga[1], gb[1], sac, 0, 0, 0);
          if (!A.imex) { // Newton term rho N_a N_b d_d u_c
#pragma unroll
            for (int e = 0; e < BS; ++e) acc[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(Na * S.gqs[qq * 9 + e], Nb, acc[e], 0, 0, 0);
          }
          sac = __builtin_amdgcn_mfma_f64_16x16x4f64(wmu * ga[2], gb[2], sac, 0, 0, 0);
          if (!A.imex) sac = __builtin_amdgcn_mfma_f64_16x16x4f64(A.rho * wNa, ugb, sac, 0, 0, 0);
          sac = __builtin_amdgcn_mfma_f64_16x16x4f64(rdt * wNa, Nb, sac, 0, 0, 0);
        }
      }
      // ---- scatter: register r of a tile holds the pair (a = 16 ti + (lane>>4) + 4 r, b = 16 tj + (lane&15))
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * ti + (lane >> 4) + 4 * r, b = bl;
        const bool have = active && a < NU && b < NU && S.len_uu[a < NU ? a : 0] >= 0 && !A.debug_skip;
        // stage slot: my 16-lane group holds one matrix row; its pairs in the order of their positions in that row
        const int slot = (lane & 48) | rank_in_row16(bv ? (unsigned(posr[r]) << 4 | unsigned(lane & 15)) : (0x100000u | unsigned(lane & 15)));
        int64_t off = -1;
        if (have) {
          const uint16_t pos = posr[r];
          off = uu_base(S.rs_uu[a], S.len_uu[a], pos, BS);
          const double s = sac[r];
          const int64_t row_dof0 = int64_t(DIM) * S.un[a];
          if (A.v_s) unsafeAtomicAdd(A.v_s + S.rs_uu[a] + pos, s);
          if (!any_c) {
#pragma unroll
            for (int e = 0; e < BS; ++e) stage[slot * BS + e] = acc[e][r] + ((e == 0 || e == 4 || e == 8) ? s : 0.0);
          } else
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const bool rc = S.cf[a * 3 + c];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
              const bool ccn = S.cf[b * 3 + d];
              const double v = acc[c * 3 + d][r] + (c == d ? s : 0.0);
              double w = 0.0;
              if (!rc && !ccn) w = v;
              else if (rc) {
                if (a == b && c == d) { // |Ke(r,r)| on the diagonal, rhs so that the update equals the inhomogeneity
                  w = fabs(v);
                  if (A.use_inhom) unsafeAtomicAdd(&A.rhs[row_dof0 + c], S.cv[a * 3 + c] * fabs(v));
                }
              } else if (A.use_inhom) {
                const double g = S.cv[b * 3 + d];
                if (g != 0.0) unsafeAtomicAdd(&S.fe[a * 3 + c], -v * g);
              }
              stage[slot * BS + c * 3 + d] = w;
            }
          }
        }
        soff[slot] = off;
        wsync2();
        // lane = (pair, entry) of the four matrix rows staged above, each row's 144 values shifted by the position of its first
        // block inside a 64-byte segment: an instruction boundary (every 64 lanes) then falls on a segment boundary wherever the
        // row's blocks are contiguous, instead of making two instructions touch the same segment (tools/scatter_sim.py)
#pragma unroll
        for (int rr = 0; rr < BS + 1; ++rr) {
          const int t = lane + 64 * rr, g = t / SPAN;
          const int64_t o0 = soff[16 * (g < 4 ? g : 0)];
          const int u = t - g * SPAN - (o0 >= 0 ? int(o0 & 7) : 0);
          const bool in = g < 4 && u >= 0 && u < 16 * BS;
          const int uc = in ? u : 0, pl = uc / BS, e = uc - pl * BS;
          const int64_t o = in ? soff[16 * g + pl] : -1;
          const double w = stage[(16 * (g < 4 ? g : 0) + pl) * BS + e];
          if (o >= 0 && w != 0.0) unsafeAtomicAdd(A.v_uu + o + e, w);
        }
        wsync2();
      }
    }
  }
  __syncthreads();
  // ---- rhs scatter (unconstrained owned rows; constrained rows were handled with the diagonal)
  if (active && h == 0) {
    for (int i = lane; i < ND; i += 64) {
      if (S.cf[i]) continue;
      if (i < NU * DIM) {
        const int a = i / DIM, c = i - a * DIM;
        if (S.len_uu[a] >= 0) unsafeAtomicAdd(&A.rhs[int64_t(DIM) * S.un[a] + c], S.fe[i]);
      } else {
        const int b = i - NU * DIM;
        if (S.len_b[b] >= 0) unsafeAtomicAdd(&A.rhs[int64_t(DIM) * A.nUo + S.pn[b]], S.fe[i]);
      }
    }
  }
}

// 3D Q2/Q1 only; the block-interleaved A_uu layout is assumed by the staged scatter
template <bool OTF, int WAVES, int CPB> // CPB cells per workgroup (two waves each)
static void launch3(ifem_ctx *ctx, const AsmArgs &A) {
  const size_t smem = ((sizeof(typename std::conditional<OTF, Shared3Otf, Shared3>::type) + 15) & ~size_t(15)) + CPB * ((sizeof(Cell3<OTF>) + 15) & ~size_t(15));
  // the dynamic-LDS limit is an attribute of the function ON A DEVICE: remembered per device, not per process
  static std::mutex mu;
  static std::set<int> done;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count(ctx->device)) {
      IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ins_assemble3<CPB, OTF, WAVES>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      done.insert(ctx->device);
    }
  }
  Tab1D t;
  tab1d(t, 2);
  AsmArgs B = A;
  B.order = nullptr; B.first = 0; B.count = A.n_cells;
  const int64_t nblk = (B.count + CPB - 1) / CPB;
  hipLaunchKernelGGL((k_ins_assemble3<CPB, OTF, WAVES>), dim3((unsigned)nblk), dim3(128 * CPB), smem, ctx->stream, B, t);
  IFEM_HIP_CHECK(hipGetLastError());
}

bool launch_ins_assemble3_kernel(ifem_ctx *ctx, const AsmArgs &A) {
#if !IFEM_UU_INTERLEAVED
  return false;
#else
  if (ctx->dim != 3 || ctx->kv != 2) return false;
  const int v = ctx->tune.asm3_variant;
  const bool tables = v == 1 || (v == 2 && !A.skip_geo && !A.rhs_only);
  const int cpb = ctx->tune.asm3_cpb;
  if (tables) launch3<false, 0, 2>(ctx, A); // register allocation left to the compiler, as in rounds 1-2 (170 + 80 accumulation registers)
  else if (ctx->tune.asm3_waves == 4) launch3<true, 4, 2>(ctx, A);
  else if (ctx->tune.asm3_waves == 2) launch3<true, 2, 2>(ctx, A);
  else if (cpb == 1) launch3<true, 3, 1>(ctx, A);
  else if (cpb == 4) launch3<true, 3, 4>(ctx, A);
  else launch3<true, 3, 2>(ctx, A);
  return true;
#endif
}

} // namespace ifem
