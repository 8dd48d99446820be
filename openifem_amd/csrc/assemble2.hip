// assemble2.hip -- InsIM::assemble on gfx950, second version (reference: source/mpi_insim.cpp:153-362).
//
// Same mathematics and the same scatter as assemble.hip (component-block form of SURVEY A.2, distribute_local_to_global
// semantics of SURVEY A.4), reorganised around the register file instead of LDS:
//   * the quadrature-point loop is OUTSIDE: per point a small node table {N_a, grad N_a, u.grad N_a} is built in LDS
//     (double buffered) and every lane advances the accumulators of its (a,b) node pairs, which live in registers
//     (6 pairs x (dim*dim + 1) doubles per pass); the 17.5 KB physical-gradient table of version 1 is gone;
//   * shape values come from the 1D tensor factors (18 doubles) instead of 30 KB of reference tables;
//   * all scratch is per wave, so the only synchronisation is wave-local.
// Per wave ~13 KB of LDS instead of 57 KB: 3 waves per SIMD instead of 1, and ~9 LDS doubles per 29 FMAs in the hot
// loop instead of 21.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include "kernels.hpp"
#include "assemble_common.hpp"

#ifndef IFEM_ASM2_RP
#define IFEM_ASM2_RP 4
#endif

namespace ifem {

template <int DIM, int KV>
struct Cell2 {
  using G_ = Geo<DIM, KV>;
  static constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ, ND = G_::ND;
  static constexpr int TS = 6; // node-table stride in doubles: {N, g[3], u.g, pad} (16-byte aligned pairs)
  static constexpr int NODAL = 3 * NU * DIM + NP, TAB = 2 * NU * TS, STAGE = 64 * DIM * DIM + 64; // scatter staging
  double X[NP * DIM];
  double C[8 * DIM]; // monomial coefficients of the d-linear map
  // per quadrature point, uniform over the lanes
  double Ji[NQ * DIM * DIM], JxW[NQ], uq[NQ * DIM];
  double gqs[NQ * DIM * DIM]; // rho JxW grad u              (coefficient of N_a N_b in the Newton term)
  double Vc[NQ * DIM * DIM];  // JxW (-mu grad u_c + e_c (p - gamma rho div u))   . grad N_a  -> rhs
  double Sc[NQ * DIM];        // JxW rho (-(grad u u)_c - (u - u0)_c / dt + g_c [+ a_c]) N_a   -> rhs
  double divw[NQ];            // JxW div u
  static constexpr int SCR0 = NODAL > TAB ? NODAL : TAB;
  double scratch[SCR0 > STAGE ? SCR0 : STAGE]; // nodal values (phase 1) | node tables tab[2][NU][TS] (passes) | scatter staging
  double fe[ND], cv[ND];
  int64_t rs_uu[NU], rs_bt[NU], rs_b[NP], rs_mp[NP];
  int32_t len_uu[NU], len_bt[NU], len_b[NP], len_mp[NP];
  int32_t un[NU], pn[NP];
  uint8_t cf[ND + 7];
};

struct Shared2 {
  Tab1D t;
  double psi[27 * 8]; // [q][b] Q1 shapes at the Gauss points
};

template <int DIM, int KV, int WPB, bool ATOMIC, int RPMAX>
__global__ __launch_bounds__(64 * WPB) void k_ins_assemble2(AsmArgs A, Tab1D t1) {
  using G_ = Geo<DIM, KV>;
  using C2 = Cell2<DIM, KV>;
  constexpr int N1 = KV + 1, NU = G_::NU, NP = G_::NP, NQ = G_::NQ, ND = G_::ND, TS = C2::TS;
  constexpr int NPAIR = NU * NU, ROUNDS = (NPAIR + 63) / 64, RP = ROUNDS < RPMAX ? ROUNDS : RPMAX;
  constexpr int NPASS = (ROUNDS + RP - 1) / RP;
  constexpr int NBP = NU * NP, BROUNDS = (NBP + 63) / 64; // velocity-pressure pairs
  constexpr int BS = DIM * DIM;
  constexpr int FR = (ND + 63) / 64;                      // rhs items per lane
  extern __shared__ __align__(16) unsigned char smem[];
  Shared2 &T = *reinterpret_cast<Shared2 *>(smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  C2 &S = *reinterpret_cast<C2 *>(smem + ((sizeof(Shared2) + 15) & ~size_t(15)) + size_t(wave) * ((sizeof(C2) + 15) & ~size_t(15)));
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
  __syncthreads();

  const int64_t idx = int64_t(A.xcd_swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x) * WPB + wave;
  const bool active = idx < A.count;
  const int64_t cc = active ? (A.order ? int64_t(A.order[A.first + idx]) : idx) : 0;
  const int64_t p_off = int64_t(DIM) * A.nUl;
  double *ue = S.scratch, *u0e = S.scratch + NU * DIM, *ae = S.scratch + 2 * NU * DIM, *pe = S.scratch + 3 * NU * DIM;

  // ---- phase 0: ids, coordinates, nodal values, row descriptors, constraint flags
  for (int i = lane; i < NP * DIM; i += 64) S.X[i] = A.vcoords[cc * NP * DIM + i];
  for (int a = lane; a < NU; a += 64) {
    const int32_t nd = A.cell_unodes[cc * NU + a];
    S.un[a] = nd;
    const bool own = nd < A.nUo;
    const int64_t r0 = own ? A.rp_uu[nd] : 0, r1 = own ? A.rp_uu[nd + 1] : 0;
    S.rs_uu[a] = r0; S.len_uu[a] = own ? int32_t(r1 - r0) : -1;
    const int64_t t0 = own ? A.rp_bt[nd] : 0, t1_ = own ? A.rp_bt[nd + 1] : 0;
    S.rs_bt[a] = t0; S.len_bt[a] = own ? int32_t(t1_ - t0) : -1;
    for (int c = 0; c < DIM; ++c) {
      const int64_t dof = int64_t(DIM) * nd + c;
      ue[a * DIM + c] = A.eval[dof];
      u0e[a * DIM + c] = A.present[dof];
      ae[a * DIM + c] = A.fsi_acc ? A.fsi_acc[dof] : 0.0;
      S.cf[a * DIM + c] = A.is_c ? A.is_c[dof] : 0;
      S.cv[a * DIM + c] = A.cval ? A.cval[dof] : 0.0;
    }
  }
  for (int b = lane; b < NP; b += 64) {
    const int32_t nd = A.cell_pnodes[cc * NP + b];
    S.pn[b] = nd;
    const bool own = nd < A.nPo;
    const int64_t r0 = own ? A.rp_b[nd] : 0, r1 = own ? A.rp_b[nd + 1] : 0;
    S.rs_b[b] = r0; S.len_b[b] = own ? int32_t(r1 - r0) : -1;
    const int64_t m0 = own ? A.rp_mp[nd] : 0, m1 = own ? A.rp_mp[nd + 1] : 0;
    S.rs_mp[b] = m0; S.len_mp[b] = own ? int32_t(m1 - m0) : -1;
    pe[b] = A.eval[p_off + nd];
    S.cf[NU * DIM + b] = A.is_c ? A.is_c[p_off + nd] : 0;
    S.cv[NU * DIM + b] = A.cval ? A.cval[p_off + nd] : 0.0;
  }
  for (int i = lane; i < ND; i += 64) S.fe[i] = 0.0;
  wsync2();
  // monomial coefficients of x(xi) = sum_k C_k prod_{d in k} xi_d:  C_k = sum_{v subset of k} (-1)^{|k|-|v|} X_v
  if (lane < NP * DIM) {
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
  wsync2();
  const int ind = (active && A.indicator) ? A.indicator[cc] : 0;

  // ---- phase 1: per quadrature point (lane = q): MappingQ1 Jacobian, fields of the evaluation point, rhs coefficients
  if (lane < NQ) {
    const int q = lane;
    const int qi[3] = {q % N1, (q / N1) % N1, q / (N1 * N1)};
    double xi[3] = {0, 0, 0}, wq = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      xi[d] = T.t.xi[qi[d]];
      wq *= T.t.w[qi[d]];
    }
    double J[DIM * DIM], Ji[DIM * DIM]; // J[d][e] = d x_d / d xi_e
    if constexpr (DIM == 3) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const double c1 = S.C[1 * 3 + e], c2 = S.C[2 * 3 + e], c3 = S.C[3 * 3 + e], c4 = S.C[4 * 3 + e], c5 = S.C[5 * 3 + e],
                     c6 = S.C[6 * 3 + e], c7 = S.C[7 * 3 + e];
        J[e * 3 + 0] = c1 + c3 * xi[1] + c5 * xi[2] + c7 * (xi[1] * xi[2]);
        J[e * 3 + 1] = c2 + c3 * xi[0] + c6 * xi[2] + c7 * (xi[0] * xi[2]);
        J[e * 3 + 2] = c4 + c5 * xi[0] + c6 * xi[1] + c7 * (xi[0] * xi[1]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const double c1 = S.C[1 * 2 + e], c2 = S.C[2 * 2 + e], c3 = S.C[3 * 2 + e];
        J[e * 2 + 0] = c1 + c3 * xi[1];
        J[e * 2 + 1] = c2 + c3 * xi[0];
      }
    }
    const double det = inv_small<DIM>(J, Ji);
    const double w = fabs(det) * wq;
    double u[DIM], u0[DIM], ac[DIM], gr[DIM * DIM], p = 0; // gr[c][e]: reference gradient of u_c
#pragma unroll
    for (int c = 0; c < DIM; ++c) { u[c] = 0; u0[c] = 0; ac[c] = 0; }
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) gr[i] = 0;
#pragma unroll 1
    for (int a = 0; a < NU; ++a) { // rolled on purpose: unrolled, the scheduler hoists ~250 LDS loads into registers
      const int ai[3] = {a % N1, (a / N1) % N1, a / (N1 * N1)};
      const double nx = T.t.N[qi[0] * N1 + ai[0]], ny = T.t.N[qi[1] * N1 + ai[1]];
      const double dx = T.t.dN[qi[0] * N1 + ai[0]], dy = T.t.dN[qi[1] * N1 + ai[1]];
      double N, dr[DIM];
      if constexpr (DIM == 3) {
        const double nz = T.t.N[qi[2] * N1 + ai[2]], dz = T.t.dN[qi[2] * N1 + ai[2]];
        dr[0] = dx * ny * nz; dr[1] = nx * dy * nz; dr[2] = nx * ny * dz; N = nx * ny * nz;
      } else {
        dr[0] = dx * ny; dr[1] = nx * dy; N = nx * ny;
      }
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const double uv = ue[a * DIM + c];
        u[c] += N * uv; u0[c] += N * u0e[a * DIM + c]; ac[c] += N * ae[a * DIM + c];
#pragma unroll
        for (int e = 0; e < DIM; ++e) gr[c * DIM + e] += uv * dr[e];
      }
    }
#pragma unroll
    for (int b = 0; b < NP; ++b) p += T.psi[q * NP + b] * pe[b];
    double g[DIM * DIM], dv = 0; // physical gradient g[c][d] = sum_e gr[c][e] Ji[e][d]
#pragma unroll
    for (int c = 0; c < DIM; ++c)
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        double t = 0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) t += gr[c * DIM + e] * Ji[e * DIM + d];
        g[c * DIM + d] = t;
      }
#pragma unroll
    for (int c = 0; c < DIM; ++c) dv += g[c * DIM + c];
    S.JxW[q] = w;
    S.divw[q] = w * dv;
#pragma unroll
    for (int i = 0; i < DIM * DIM; ++i) { S.Ji[q * DIM * DIM + i] = Ji[i]; S.gqs[q * DIM * DIM + i] = A.rho * w * g[i]; }
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      S.uq[q * DIM + c] = u[c];
      double adv = 0;
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        adv += g[c * DIM + d] * u[d];
        S.Vc[(q * DIM + c) * DIM + d] = w * (-A.mu * g[c * DIM + d] + (c == d ? p - A.gamma * A.rho * dv : 0.0));
      }
      double sc = -A.rho * adv - A.rho * A.inv_dt * (u[c] - u0[c]) + A.rho * A.g[c];
      if (ind == 1) sc += A.rho * ac[c];
      S.Sc[q * DIM + c] = w * sc;
    }
  }
  wsync2();
  // ---- Neumann (pressure) boundary faces  (:313-341)
  if (A.n_neumann != 0 && active) {
    for (int f = 0; f < 2 * DIM; ++f) {
      const int bid = A.cell_face_bid[cc * 2 * DIM + f];
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
          double J[DIM * DIM], Ji[DIM * DIM];
          for (int k = 0; k < DIM * DIM; ++k) J[k] = 0;
          const double *dps = &A.fe->fdpsi[(f * A.fe->nqf + qf) * NP * DIM];
          for (int v = 0; v < NP; ++v)
            for (int d = 0; d < DIM; ++d)
              for (int e = 0; e < DIM; ++e) J[d * DIM + e] += S.X[v * DIM + d] * dps[v * DIM + e];
          const double det = inv_small<DIM>(J, Ji);
          double nv[DIM], nn = 0;
          for (int d = 0; d < DIM; ++d) { nv[d] = sgn * Ji[nd * DIM + d]; nn += nv[d] * nv[d]; }
          nn = sqrt(nn);
          const double JxWf = fabs(det) * nn * A.fe->fw[qf];
          acc += A.fe->fphi[(f * A.fe->nqf + qf) * NU + a] * (nv[c] / nn) * pbc * JxWf;
        }
        S.fe[i] -= acc;
      }
    }
  }
  wsync2();

  // ---- passes over the quadrature points with register accumulators
  // my node for the table build (lane = a) and its 1D indices
  const int ta = lane < NU ? lane : 0;
  const int tai[3] = {ta % N1, (ta / N1) % N1, ta / (N1 * N1)};
  const double wgam = A.gamma * A.rho, rdt = A.rho * A.inv_dt;
  double *tab = S.scratch;
  auto build_tab = [&](int q) -> double * { // node table of point q: {N_a, grad N_a (physical), u . grad N_a}
    double *tb = tab + (q & 1) * (NU * TS);
    if (lane < NU) {
      const int qi0 = q % N1, qi1 = (q / N1) % N1, qi2 = q / (N1 * N1);
      const double nx = T.t.N[qi0 * N1 + tai[0]], ny = T.t.N[qi1 * N1 + tai[1]];
      const double dx = T.t.dN[qi0 * N1 + tai[0]], dy = T.t.dN[qi1 * N1 + tai[1]];
      double N, dr[DIM];
      if constexpr (DIM == 3) {
        const double nz = T.t.N[qi2 * N1 + tai[2]], dz = T.t.dN[qi2 * N1 + tai[2]];
        dr[0] = dx * ny * nz; dr[1] = nx * dy * nz; dr[2] = nx * ny * dz; N = nx * ny * nz;
      } else {
        dr[0] = dx * ny; dr[1] = nx * dy; N = nx * ny;
      }
      double ug = 0;
      tb[lane * TS] = N;
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        double t = 0;
#pragma unroll
        for (int e = 0; e < DIM; ++e) t += dr[e] * S.Ji[(q * DIM + e) * DIM + d];
        tb[lane * TS + 1 + d] = t;
        ug += S.uq[q * DIM + d] * t;
      }
      tb[lane * TS + 4] = A.imex ? 0.0 : ug;
    }
    return tb;
  };
#pragma unroll 1
  for (int pass = 0; pass < NPASS; ++pass) {
    if (((A.rhs_only || A.skip_uu) && pass > 0) || A.debug_skip >= 5) break;
    const bool first = pass == 0;
    int oa[RP], ob[RP];
    bool pv[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      const int t = lane + 64 * (pass * RP + j);
      pv[j] = t < NPAIR;
      const int a = pv[j] ? t / NU : 0, b = pv[j] ? t - (t / NU) * NU : 0;
      oa[j] = a * TS; ob[j] = b * TS;
    }
    double acc[RP][DIM * DIM], sacc[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      sacc[j] = 0;
#pragma unroll
      for (int i = 0; i < DIM * DIM; ++i) acc[j][i] = 0;
    }
    // plain read-modify-write scatter: request the old matrix values now, consume them after the point loop
    double old[RP][DIM * DIM];
    double *pbase[RP];
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      pbase[j] = nullptr;
      const int a = oa[j] / TS, b = ob[j] / TS;
      if (pv[j] && active && S.len_uu[a] >= 0) {
        pbase[j] = A.v_uu + uu_base(S.rs_uu[a], S.len_uu[a], A.posUU[(cc * NU + a) * NU + b], DIM * DIM);
        if constexpr (!ATOMIC) {
          const int len = S.len_uu[a];
#pragma unroll
          for (int e = 0; e < DIM * DIM; ++e) old[j][e] = pbase[j][int64_t(e) * uu_estride(len)];
        }
      }
    }
    double fr[FR]; // local rhs items (first pass)
#pragma unroll
    for (int k = 0; k < FR; ++k) fr[k] = 0;

    // software pipeline: the table of point q+1 is written (other LDS buffer) before point q is consumed, so its
    // LDS round trip hides behind ~190 FMAs instead of stalling the wave at every point
    double *tb_next = build_tab(0);
    wsync2();
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
      double *tb = tb_next;
      if (q + 1 < NQ) tb_next = build_tab(q + 1);
      const double w = S.JxW[q];
      const double wmu = w * A.mu, wrho = w * A.rho, wrdt = w * rdt, wg = w * wgam;
      double gqs[DIM * DIM];
#pragma unroll
      for (int i = 0; i < DIM * DIM; ++i) gqs[i] = A.imex ? 0.0 : S.gqs[q * DIM * DIM + i];
#pragma unroll
      for (int j = 0; j < RP; ++j) {
        if (A.debug_skip == 2 || A.rhs_only || A.skip_uu) continue;
        const double *pa = tb + oa[j], *pb = tb + ob[j];
        const double Na = pa[0], Nb = pb[0], ugb = pb[4];
        double ga[DIM], gb[DIM], gg = 0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) { ga[d] = pa[1 + d]; gb[d] = pb[1 + d]; gg += ga[d] * gb[d]; }
        const double nn = Na * Nb;
        sacc[j] += wmu * gg + wrho * (Na * ugb) + wrdt * nn;
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          const double wga = wg * ga[c];
#pragma unroll
          for (int d = 0; d < DIM; ++d) acc[j][c * DIM + d] += nn * gqs[c * DIM + d] + wga * gb[d];
        }
      }
      if (first) { // local rhs (:281-304)
#pragma unroll
        for (int k = 0; k < FR; ++k) {
          const int i = lane + 64 * k;
          if (i < NU * DIM) {
            const int a = i / DIM, c = i - a * DIM;
            const double *pa = tb + a * TS;
            double t = S.Sc[q * DIM + c] * pa[0];
#pragma unroll
            for (int d = 0; d < DIM; ++d) t += S.Vc[(q * DIM + c) * DIM + d] * pa[1 + d];
            fr[k] += t;
          } else if (i < ND) {
            fr[k] += S.divw[q] * T.psi[q * NP + (i - NU * DIM)];
          }
        }
      }
      wsync2(); // table q+1 complete before the next iteration reads it; table q free for q+2
    }
    wsync2();
    // ---- scatter the velocity-velocity pairs of this pass
#if IFEM_UU_INTERLEAVED
    // Block-interleaved A_uu: the BS values of a pair are contiguous in memory.  Every lane settles the constraint
    // logic of its pair, parks the BS contributions in LDS, then the wave re-reads them flat, lane = (pair, entry): one
    // atomic instruction covers ~7 pairs x 9 consecutive doubles (~12 segments) instead of 64 pairs in 64 segments.
    {
      double *stage = S.scratch;                                          // [64][BS]
      int64_t *soff = reinterpret_cast<int64_t *>(S.scratch + 64 * BS);   // [64] element offset of the block, -1: none
#pragma unroll
      for (int j = 0; j < RP; ++j) {
        const bool have = pbase[j] && !A.debug_skip && !A.rhs_only && !A.skip_uu;
        soff[lane] = have ? int64_t(pbase[j] - A.v_uu) : int64_t(-1);
        if (have) {
          const int a = oa[j] / TS, b = ob[j] / TS;
          const int64_t row_dof0 = int64_t(DIM) * S.un[a];
          if (A.v_s) gadd<ATOMIC>(A.v_s + S.rs_uu[a] + A.posUU[(cc * NU + a) * NU + b], sacc[j]);
#pragma unroll
          for (int c = 0; c < DIM; ++c) {
            const bool rc = S.cf[a * DIM + c];
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
              const bool ccn = S.cf[b * DIM + d];
              const double v = acc[j][c * DIM + d] + (c == d ? sacc[j] : 0.0);
              double w = 0.0;
              if (!rc && !ccn) w = v;
              else if (rc) {
                if (a == b && c == d) { // |Ke(r,r)| on the diagonal, rhs so that the update equals the inhomogeneity
                  w = fabs(v);
                  if (A.use_inhom) gadd<ATOMIC>(&A.rhs[row_dof0 + c], S.cv[a * DIM + c] * fabs(v));
                }
              } else if (A.use_inhom) {
                const double g = S.cv[b * DIM + d];
                if (g != 0.0) unsafeAtomicAdd(&S.fe[a * DIM + c], -v * g);
              }
              stage[lane * BS + c * DIM + d] = w;
            }
          }
        }
        wsync2();
#pragma unroll
        for (int r = 0; r < BS; ++r) {
          const int t = lane + 64 * r, pl = t / BS, e = t - pl * BS;
          const int64_t off = soff[pl];
          const double w = stage[t];
          if (off >= 0 && w != 0.0) gadd<ATOMIC>(A.v_uu + off + e, w);
        }
        wsync2();
      }
    }
#else
#pragma unroll
    for (int j = 0; j < RP; ++j) {
      if (!pbase[j] || A.debug_skip || A.rhs_only || A.skip_uu) continue; // no pair in this slot, inactive cell or row owned by another rank
      const int a = oa[j] / TS, b = ob[j] / TS;
      const int len = S.len_uu[a];
      double *base = pbase[j];
      const int64_t row_dof0 = int64_t(DIM) * S.un[a];
      if (A.v_s) gadd<ATOMIC>(A.v_s + S.rs_uu[a] + A.posUU[(cc * NU + a) * NU + b], sacc[j]);
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        const bool rc = S.cf[a * DIM + c];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          const bool ccn = S.cf[b * DIM + d];
          const double v = acc[j][c * DIM + d] + (c == d ? sacc[j] : 0.0);
          double *dst = base + int64_t(c * DIM + d) * uu_estride(len);
          if (!rc && !ccn) { if constexpr (ATOMIC) unsafeAtomicAdd(dst, v); else *dst = old[j][c * DIM + d] + v; }
          else if (rc) {
            if (a == b && c == d) { // |Ke(r,r)| on the diagonal, rhs so that the update equals the inhomogeneity
              if constexpr (ATOMIC) unsafeAtomicAdd(dst, fabs(v)); else *dst = old[j][c * DIM + d] + fabs(v);
              if (A.use_inhom) gadd<ATOMIC>(&A.rhs[row_dof0 + c], S.cv[a * DIM + c] * fabs(v));
            }
          } else if (A.use_inhom) {
            const double g = S.cv[b * DIM + d];
            if (g != 0.0) unsafeAtomicAdd(&S.fe[a * DIM + c], -v * g);
          }
        }
      }
    }
#endif
    if (first) {
#pragma unroll
      for (int k = 0; k < FR; ++k) {
        const int i = lane + 64 * k;
        if (i < ND) unsafeAtomicAdd(&S.fe[i], fr[k]);
      }
    }
    wsync2();
  }
  bool need_b = !A.rhs_only && A.debug_skip < 3;
  if (need_b && A.skip_geo) { // cached blocks: only a cell with an inhomogeneous constrained dof still needs the entries
    bool mine = false;
    for (int i = lane; i < ND; i += 64) mine = mine || (S.cf[i] && S.cv[i] != 0.0);
    need_b = A.use_inhom && __any(mine);
  }
  if (need_b) { // ---- velocity-pressure blocks: -JxW psi_b grad N_a, own pass over the points
    double bacc[BROUNDS][DIM];
#pragma unroll
    for (int k = 0; k < BROUNDS; ++k)
#pragma unroll
      for (int c = 0; c < DIM; ++c) bacc[k][c] = 0;
    double *tb_next = build_tab(0);
    wsync2();
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
      double *tb = tb_next;
      if (q + 1 < NQ) tb_next = build_tab(q + 1);
      const double w = S.JxW[q];
      {
#pragma unroll
        for (int k = 0; k < BROUNDS; ++k) {
          const int t = lane + 64 * k;
          if (t < NBP) {
            const int a = t / NP, pb = t - a * NP;
            const double wpsi = w * T.psi[q * NP + pb];
            const double *pa = tb + a * TS;
#pragma unroll
            for (int c = 0; c < DIM; ++c) bacc[k][c] -= wpsi * pa[1 + c];
          }
        }
      }
      wsync2();
    }
    wsync2();
    if (active) { // block (0,1) = B^T and block (1,0) = B
#pragma unroll
      for (int k = 0; k < BROUNDS; ++k) {
        const int t = lane + 64 * k;
        if (t >= NBP) continue;
        const int a = t / NP, pb = t - a * NP;
        const bool pc = S.cf[NU * DIM + pb];
        if (S.len_bt[a] >= 0) {
          const int len = S.len_bt[a];
          double *base = A.v_bt + S.rs_bt[a] * DIM + A.posUP[(cc * NU + a) * NP + pb];
#pragma unroll
          for (int c = 0; c < DIM; ++c) {
            if (S.cf[a * DIM + c]) continue;
            if (!pc) { if (!A.skip_geo) gadd<ATOMIC>(base + int64_t(c) * len, bacc[k][c]); }
            else if (A.use_inhom && S.cv[NU * DIM + pb] != 0.0) unsafeAtomicAdd(&S.fe[a * DIM + c], -bacc[k][c] * S.cv[NU * DIM + pb]);
          }
        }
        if (S.len_b[pb] >= 0 && !pc) {
          const int len = S.len_b[pb];
          double *base = A.v_b + S.rs_b[pb] * DIM + A.posPU[(cc * NP + pb) * NU + a];
#pragma unroll
          for (int c = 0; c < DIM; ++c) {
            if (!S.cf[a * DIM + c]) { if (!A.skip_geo) gadd<ATOMIC>(base + int64_t(c) * len, bacc[k][c]); }
            else if (A.use_inhom && S.cv[a * DIM + c] != 0.0) unsafeAtomicAdd(&S.fe[NU * DIM + pb], -bacc[k][c] * S.cv[a * DIM + c]);
          }
        }
      }
    }
    wsync2();
  }
  // ---- pressure mass matrix M_p and diag(M_u)  (:274-276, only the (0,0) diagonal and (1,1) are used)
  for (int t = lane; t < ((A.rhs_only || A.skip_geo || A.debug_skip >= 4) ? 0 : NP * NP); t += 64) {
    const int pa = t / NP, pb = t - pa * NP;
    double m = 0;
#pragma unroll 3
    for (int q = 0; q < NQ; ++q) m += S.JxW[q] * T.psi[q * NP + pa] * T.psi[q * NP + pb];
    if (!active || S.len_mp[pa] < 0) continue;
    const bool ra = S.cf[NU * DIM + pa], cb = S.cf[NU * DIM + pb];
    double *dst = A.v_mp + S.rs_mp[pa] + A.posPP[(cc * NP + pa) * NP + pb];
    if (!ra && !cb) gadd<ATOMIC>(dst, m);
    else if (ra && pa == pb) gadd<ATOMIC>(dst, fabs(m));
  }
  if (lane < NU && !A.rhs_only && !A.skip_geo && A.debug_skip < 4) {
    double m = 0;
#pragma unroll 1
    for (int q = 0; q < NQ; ++q) {
      const int qi0 = q % N1, qi1 = (q / N1) % N1, qi2 = q / (N1 * N1);
      double N = T.t.N[qi0 * N1 + tai[0]] * T.t.N[qi1 * N1 + tai[1]];
      if constexpr (DIM == 3) N *= T.t.N[qi2 * N1 + tai[2]];
      m += S.JxW[q] * N * N;
    }
    if (active && S.len_uu[lane] >= 0)
      for (int c = 0; c < DIM; ++c) gadd<ATOMIC>(&A.diagMu[int64_t(DIM) * S.un[lane] + c], m);
  }
  wsync2();
  // ---- rhs scatter (unconstrained owned rows; constrained rows were handled with the diagonal)
  if (active) {
    for (int i = lane; i < ND; i += 64) {
      if (S.cf[i]) continue;
      if (i < NU * DIM) {
        const int a = i / DIM, c = i - a * DIM;
        if (S.len_uu[a] >= 0) gadd<ATOMIC>(&A.rhs[int64_t(DIM) * S.un[a] + c], S.fe[i]);
      } else {
        const int b = i - NU * DIM;
        if (S.len_b[b] >= 0) gadd<ATOMIC>(&A.rhs[int64_t(DIM) * A.nUo + S.pn[b]], S.fe[i]);
      }
    }
  }
}

template <int DIM, int KV>
static void launch2_t(ifem_ctx *ctx, const AsmArgs &A) {
  constexpr int WPB = 4, RPMAX = IFEM_ASM2_RP; // pairs per lane and pass: atomic scatter
  using C2 = Cell2<DIM, KV>;
  const size_t smem = ((sizeof(Shared2) + 15) & ~size_t(15)) + WPB * ((sizeof(C2) + 15) & ~size_t(15));
  static bool attr_set = false;
  if (!attr_set) {
    IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ins_assemble2<DIM, KV, WPB, true, RPMAX>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  Tab1D t;
  tab1d(t, KV);
  // (a coloured plain read-modify-write scatter was measured in round 1: 247-253 ms against 224 ms with atomics at 128^3)
  AsmArgs B = A;
  B.order = nullptr; B.first = 0; B.count = A.n_cells;
  const int64_t nblk = (B.count + WPB - 1) / WPB;
  hipLaunchKernelGGL((k_ins_assemble2<DIM, KV, WPB, true, RPMAX>), dim3((unsigned)nblk), dim3(64 * WPB), smem, ctx->stream, B, t);
  IFEM_HIP_CHECK(hipGetLastError());
}

void launch_ins_assemble2_kernel(ifem_ctx *ctx, const AsmArgs &A) {
  if (ctx->dim == 2 && ctx->kv == 1) launch2_t<2, 1>(ctx, A);
  else if (ctx->dim == 2 && ctx->kv == 2) launch2_t<2, 2>(ctx, A);
  else if (ctx->dim == 3 && ctx->kv == 1) launch2_t<3, 1>(ctx, A);
  else if (ctx->dim == 3 && ctx->kv == 2) launch2_t<3, 2>(ctx, A);
  else throw Error(IFEM_E_BADPARAM, "unsupported (dim, kv)");
}

} // namespace ifem
