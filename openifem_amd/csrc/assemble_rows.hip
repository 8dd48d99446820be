// assemble_rows.hip -- deterministic "owner computes row" assembly of InsIM::assemble (mpi_insim.cpp:153-362).
//
// The first assembly kernel (assemble.hip) integrates cell by cell and scatters with 1.7e10 f64 atomics at 128^3;
// measured atomic throughput (24-190 Gatom/s, profiles/r01_microbench.txt) makes that scatter 5x slower than the
// integration itself.  Here every matrix row is produced by exactly one wavefront and written once, coalesced:
//   pass 1  k_cell_qdata : one wave per cell, lane = quadrature point: MappingQ1 Jacobian inverse, JxW and the
//                          evaluation-point fields (u, grad u, p, u - u0, a_fsi, div u) -> 30 doubles per point
//   pass 2  k_rows_u     : one wave per owned velocity node I.  For every incidence (cell, a) of I the wave stages the
//                          cell's 6.5 kB of point data in LDS and forms the 27 blocks Ke[(a,:),(b,:)] (lane = column
//                          node b, the 27 quadrature points split over the two half-waves), accumulating them in an LDS
//                          copy of the row; then applies distribute_local_to_global's Dirichlet rules and
//                          streams the row, its B^T row, diag(M_u) and the rhs entries out.
//   pass 3  k_rows_p     : same for the pressure rows (B, M_p, rhs_p).
// No atomics, no memset of the matrices, bit-reproducible.  Reference-cell tables live in LDS.
#include <hip/hip_runtime.h>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

template <int DIM, int KV>
struct RG {
  static constexpr int N1 = KV + 1;
  static constexpr int NU = (DIM == 2) ? N1 * N1 : N1 * N1 * N1;
  static constexpr int NP = (DIM == 2) ? 4 : 8;
  static constexpr int NQ = NU;
  // per quadrature point: JxW, Jinv[D*D], u[D], G[D*D], p, du[D] (u - u0), acc[D], div
  static constexpr int F_W = 0, F_JI = 1, F_U = F_JI + DIM * DIM, F_G = F_U + DIM, F_P = F_G + DIM * DIM, F_DU = F_P + 1,
                       F_AC = F_DU + DIM, F_DIV = F_AC + DIM, QD = F_DIV + 1;
};

// LDS hand-off between the lanes of ONE wave: a wave's DS instructions execute in program order, so it is enough to
// keep the compiler from moving LDS accesses across this point.  Deliberately NOT a fence: a wavefront-scope release
// fence also drains vmcnt and would serialise the prefetched global loads of the next incidence.
__device__ inline void wave_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

template <int DIM>
__device__ inline double inv_dd(const double *J, double *Ji) {
  if constexpr (DIM == 2) {
    const double det = J[0] * J[3] - J[1] * J[2];
    const double r = 1.0 / det;
    Ji[0] = J[3] * r; Ji[1] = -J[1] * r; Ji[2] = -J[2] * r; Ji[3] = J[0] * r;
    return det;
  } else {
    const double c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    const double det = J[0] * c00 + J[1] * c01 + J[2] * c02;
    const double r = 1.0 / det;
    Ji[0] = c00 * r; Ji[3] = c01 * r; Ji[6] = c02 * r;
    Ji[1] = (J[2] * J[7] - J[1] * J[8]) * r; Ji[4] = (J[0] * J[8] - J[2] * J[6]) * r; Ji[7] = (J[1] * J[6] - J[0] * J[7]) * r;
    Ji[2] = (J[1] * J[5] - J[2] * J[4]) * r; Ji[5] = (J[2] * J[3] - J[0] * J[5]) * r; Ji[8] = (J[0] * J[4] - J[1] * J[3]) * r;
    return det;
  }
}

struct RowArgs {
  int64_t n_cells, nUo, nUl, nPo;
  const FeTables *fe;
  const double *vcoords;
  const int32_t *cell_unodes, *cell_pnodes, *cell_face_bid, *indicator;
  const uint16_t *posUU, *posUP, *posPU, *posPP;
  const int64_t *rp_uu, *rp_bt, *rp_b, *rp_mp;
  const int32_t *col_uu, *col_b;
  double *v_uu, *v_bt, *v_b, *v_mp, *diagMu, *rhs, *v_s;
  const int64_t *uinc_ptr, *pinc_ptr;
  const int32_t *uinc, *pinc;
  double *qdata;
  const uint8_t *is_c;
  const double *cval;
  const double *eval, *present, *fsi_acc;
  double mu, rho, gamma, inv_dt;
  double g[3];
  int n_neumann;
  int neumann_id[8];
  double neumann_p[8];
  int use_inhom, maxlen_uu, maxlen_bt, maxlen_b, maxlen_mp;
};

// ------------------------------------------------------------------------------------------------- pass 1
template <int DIM, int KV>
__global__ __launch_bounds__(256) void k_cell_qdata(RowArgs A) {
  using G_ = RG<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ, QD = G_::QD;
  __shared__ double sX[4][NP * DIM], sU[4][NU * DIM], sU0[4][NU * DIM], sA[4][NU * DIM], sP[4][NP];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t cell = int64_t(blockIdx.x) * 4 + wave;
  const bool active = cell < A.n_cells;
  const int64_t cc = active ? cell : 0;
  for (int i = lane; i < NP * DIM; i += 64) sX[wave][i] = A.vcoords[cc * NP * DIM + i];
  for (int a = lane; a < NU; a += 64) {
    const int64_t nd = A.cell_unodes[cc * NU + a];
    for (int c = 0; c < DIM; ++c) {
      sU[wave][a * DIM + c] = A.eval[nd * DIM + c];
      sU0[wave][a * DIM + c] = A.present[nd * DIM + c];
      sA[wave][a * DIM + c] = A.fsi_acc ? A.fsi_acc[nd * DIM + c] : 0.0;
    }
  }
  for (int b = lane; b < NP; b += 64) sP[wave][b] = A.eval[int64_t(DIM) * A.nUl + A.cell_pnodes[cc * NP + b]];
  __syncthreads();
  if (!active) return;
  const FeTables &T = *A.fe;
  for (int q = lane; q < NQ; q += 64) {
    double J[DIM * DIM], Ji[DIM * DIM];
    for (int i = 0; i < DIM * DIM; ++i) J[i] = 0;
    for (int v = 0; v < NP; ++v)
      for (int d = 0; d < DIM; ++d)
        for (int e = 0; e < DIM; ++e) J[d * DIM + e] += sX[wave][v * DIM + d] * T.dpsi[(q * NP + v) * DIM + e];
    const double det = inv_dd<DIM>(J, Ji);
    double u[DIM], u0[DIM], ac[DIM], gr[DIM * DIM], p = 0;
    for (int c = 0; c < DIM; ++c) { u[c] = 0; u0[c] = 0; ac[c] = 0; }
    for (int i = 0; i < DIM * DIM; ++i) gr[i] = 0;
    for (int a = 0; a < NU; ++a) {
      const double N = T.phi[q * NU + a];
      for (int c = 0; c < DIM; ++c) {
        const double ue = sU[wave][a * DIM + c];
        u[c] += N * ue; u0[c] += N * sU0[wave][a * DIM + c]; ac[c] += N * sA[wave][a * DIM + c];
        for (int e = 0; e < DIM; ++e) gr[c * DIM + e] += ue * T.dphi[(q * NU + a) * DIM + e]; // reference gradient
      }
    }
    for (int b = 0; b < NP; ++b) p += T.psi[q * NP + b] * sP[wave][b];
    double *o = A.qdata + cell * (NQ * QD);
    o[G_::F_W * NQ + q] = fabs(det) * T.w[q];
    double dv = 0;
    for (int c = 0; c < DIM; ++c)
      for (int d = 0; d < DIM; ++d) {
        double g = 0;
        for (int e = 0; e < DIM; ++e) g += gr[c * DIM + e] * Ji[e * DIM + d];
        o[(G_::F_G + c * DIM + d) * NQ + q] = g;
        if (c == d) dv += g;
      }
    for (int i = 0; i < DIM * DIM; ++i) o[(G_::F_JI + i) * NQ + q] = Ji[i];
    for (int c = 0; c < DIM; ++c) {
      o[(G_::F_U + c) * NQ + q] = u[c];
      o[(G_::F_DU + c) * NQ + q] = u[c] - u0[c];
      o[(G_::F_AC + c) * NQ + q] = ac[c];
    }
    o[G_::F_P * NQ + q] = p;
    o[G_::F_DIV * NQ + q] = dv;
  }
}

// ------------------------------------------------------------------------------------------------- pass 2
template <int DIM, int KV>
struct RowTables {
  using G_ = RG<DIM, KV>;
  double phi[G_::NQ * G_::NU];
  double dphi[G_::NQ * G_::NU * DIM];
  double psi[G_::NQ * G_::NP];
};

template <int DIM>
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Neumann (pressure) boundary faces of `cell` contribute -(phi_(a,c) . n) p_bc JxW_face to the rhs (:313-341)
template <int DIM, int KV>
__device__ inline void neumann_rhs(const RowArgs &A, int64_t cell, int a, double *out /*[DIM]*/) {
  using G_ = RG<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP;
  for (int c = 0; c < DIM; ++c) out[c] = 0;
  for (int f = 0; f < 2 * DIM; ++f) {
    const int bid = A.cell_face_bid[cell * 2 * DIM + f];
    if (bid < 0) continue;
    double pbc = 0; bool hit = false;
    for (int k = 0; k < A.n_neumann; ++k) if (A.neumann_id[k] == bid) { pbc = A.neumann_p[k]; hit = true; }
    if (!hit) continue;
    const int nd = f >> 1; const double sgn = (f & 1) ? 1.0 : -1.0;
    for (int qf = 0; qf < A.fe->nqf; ++qf) {
      double J[DIM * DIM], Ji[DIM * DIM];
      for (int k = 0; k < DIM * DIM; ++k) J[k] = 0;
      const double *dps = &A.fe->fdpsi[(f * A.fe->nqf + qf) * NP * DIM];
      for (int v = 0; v < NP; ++v)
        for (int d = 0; d < DIM; ++d)
          for (int e = 0; e < DIM; ++e) J[d * DIM + e] += A.vcoords[(cell * NP + v) * DIM + d] * dps[v * DIM + e];
      const double det = inv_dd<DIM>(J, Ji);
      double nv[DIM], nn = 0;
      for (int d = 0; d < DIM; ++d) { nv[d] = sgn * Ji[nd * DIM + d]; nn += nv[d] * nv[d]; }
      nn = sqrt(nn);
      const double t = A.fe->fphi[(f * A.fe->nqf + qf) * NU + a] * pbc * fabs(det) * A.fe->fw[qf]; // * nn / nn
      for (int c = 0; c < DIM; ++c) out[c] -= t * nv[c];
    }
  }
}

template <int DIM, int KV, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_rows_u(RowArgs A) {
  using G_ = RG<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ, QD = G_::QD, DD = DIM * DIM;
  constexpr int QH = (NQ + 1) / 2; // quadrature points of the first half-wave
  extern __shared__ __align__(16) unsigned char smem_r[];
  auto &T = *reinterpret_cast<RowTables<DIM, KV> *>(smem_r);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_wave = size_t(NQ) * QD + size_t(NQ) * (DIM + 1) + size_t(A.maxlen_uu) * (DD + 1) + size_t(A.maxlen_bt) * DIM + 8;
  double *W = reinterpret_cast<double *>(smem_r + sizeof(RowTables<DIM, KV>)) + size_t(wave) * per_wave;
  double *qd = W, *aside = qd + NQ * QD, *rowbuf = aside + NQ * (DIM + 1), *sbuf = rowbuf + size_t(A.maxlen_uu) * DD,
         *btbuf = sbuf + A.maxlen_uu, *misc = btbuf + size_t(A.maxlen_bt) * DIM;
  for (int i = threadIdx.x; i < NQ * NU; i += blockDim.x) T.phi[i] = A.fe->phi[i];
  for (int i = threadIdx.x; i < NQ * NU * DIM; i += blockDim.x) T.dphi[i] = A.fe->dphi[i];
  for (int i = threadIdx.x; i < NQ * NP; i += blockDim.x) T.psi[i] = A.fe->psi[i];
  __syncthreads();
  // persistent waves: the 25 kB of tables are staged once per workgroup, then each wave walks rows on its own
  // (only wave-level synchronisation below)
  for (int64_t I = int64_t(blockIdx.x) * WPB + wave; I < A.nUo; I += int64_t(gridDim.x) * WPB) {
  const int64_t rs = A.rp_uu[I], ts = A.rp_bt[I];
  const int len = int(A.rp_uu[I + 1] - rs), tlen = int(A.rp_bt[I + 1] - ts);
  for (int i = lane; i < len * DD; i += 64) rowbuf[i] = 0;
  for (int i = lane; i < len; i += 64) sbuf[i] = 0;
  for (int i = lane; i < tlen * DIM; i += 64) btbuf[i] = 0;
  double fe[DIM], mass = 0;
  for (int c = 0; c < DIM; ++c) fe[c] = 0;
  const double wgam = A.gamma * A.rho, rdt = A.rho * A.inv_dt;
  wave_sync();
  // Software pipeline over the incidences of the row: the dependent global loads of incidence j+1 (its cell's point
  // data, scatter positions, indicator) are issued before incidence j is integrated -- at 1.5 waves per SIMD nothing
  // else would hide their latency.
  const int64_t ip0 = A.uinc_ptr[I];
  const int ninc = int(A.uinc_ptr[I + 1] - ip0);
  const int32_t my_packed = (lane < ninc) ? A.uinc[ip0 + lane] : 0; // <= 64 incidences per node (checked at set-up)
  constexpr int NST = (NQ * QD + 63) / 64;
  double st_n[NST];
  uint16_t posuu_n = 0, posup_n = 0;
  int ind_n = 0;
  auto issue = [&](int j) {
    const int32_t pk = __shfl(my_packed, j, 64);
    const int64_t cl = pk >> 5;
    const int aa = pk & 31;
    const double *src = A.qdata + cl * (NQ * QD);
#pragma unroll
    for (int t = 0; t < NST; ++t) { const int i = lane + 64 * t; st_n[t] = (i < NQ * QD) ? src[i] : 0.0; }
    const int bb = lane & 31;
    if (bb < NU && lane < 32) posuu_n = A.posUU[(cl * NU + aa) * NU + bb];
    if (lane >= 32 && lane < 32 + NP) posup_n = A.posUP[(cl * NU + aa) * NP + (lane - 32)];
    ind_n = A.indicator ? A.indicator[cl] : 0;
  };
  if (ninc > 0) issue(0);
  for (int j = 0; j < ninc; ++j) {
    const int32_t packed = __shfl(my_packed, j, 64);
    const int64_t cell = packed >> 5;
    const int a = packed & 31;
    const uint16_t posuu = posuu_n, posup = posup_n;
    const int ind = ind_n;
#pragma unroll
    for (int t = 0; t < NST; ++t) { const int i = lane + 64 * t; if (i < NQ * QD) qd[i] = st_n[t]; }
    wave_sync();
    if (j + 1 < ninc) issue(j + 1);
    // ---- row-side data per quadrature point + rhs of row (a, :)  (:281-304)
    double t_fe[DIM], t_m = 0;
    for (int c = 0; c < DIM; ++c) t_fe[c] = 0;
    if (lane < NQ) {
      const int q = lane;
      const double w = qd[G_::F_W * NQ + q], Na = T.phi[q * NU + a];
      double ga[DIM];
      for (int d = 0; d < DIM; ++d) {
        double g = 0;
        for (int e = 0; e < DIM; ++e) g += T.dphi[(q * NU + a) * DIM + e] * qd[(G_::F_JI + e * DIM + d) * NQ + q];
        ga[d] = g;
        aside[q * (DIM + 1) + 1 + d] = g;
      }
      aside[q * (DIM + 1)] = Na;
      const double p = qd[G_::F_P * NQ + q], dv = qd[G_::F_DIV * NQ + q];
      for (int c = 0; c < DIM; ++c) {
        double visc = 0, adv = 0;
        for (int d = 0; d < DIM; ++d) {
          const double G = qd[(G_::F_G + c * DIM + d) * NQ + q];
          visc += G * ga[d];
          adv += G * qd[(G_::F_U + d) * NQ + q];
        }
        double t = -A.mu * visc - A.rho * adv * Na + p * ga[c] - wgam * dv * ga[c] - rdt * qd[(G_::F_DU + c) * NQ + q] * Na +
                   A.rho * A.g[c] * Na;
        if (ind == 1) t += A.rho * qd[(G_::F_AC + c) * NQ + q] * Na;
        t_fe[c] = t * w;
      }
      t_m = w * Na * Na;
    }
    for (int c = 0; c < DIM; ++c) fe[c] += wave_sum<DIM>(t_fe[c]);
    mass += wave_sum<DIM>(t_m);
    wave_sync();
    // ---- blocks Ke[(a,:),(b,:)]: lane = column node b, quadrature points split over the two half-waves  (:263-273)
    const int b = lane & 31, half = lane >> 5;
    const bool valid = b < NU;
    double s = 0, acc[DD];
    for (int i = 0; i < DD; ++i) acc[i] = 0;
    if (valid) {
      const int q0 = half ? QH : 0, q1 = half ? NQ : QH;
      for (int q = q0; q < q1; ++q) {
        const double w = qd[G_::F_W * NQ + q], Na = aside[q * (DIM + 1)], Nb = T.phi[q * NU + b];
        double gb[DIM], ga[DIM];
        for (int d = 0; d < DIM; ++d) {
          double g = 0;
          for (int e = 0; e < DIM; ++e) g += T.dphi[(q * NU + b) * DIM + e] * qd[(G_::F_JI + e * DIM + d) * NQ + q];
          gb[d] = g;
          ga[d] = aside[q * (DIM + 1) + 1 + d];
        }
        double gg = 0, ugb = 0;
        for (int d = 0; d < DIM; ++d) { gg += ga[d] * gb[d]; ugb += qd[(G_::F_U + d) * NQ + q] * gb[d]; }
        s += w * (A.mu * gg + A.rho * Na * ugb + rdt * Na * Nb);
        const double m = w * A.rho * Na * Nb, wg = w * wgam;
        for (int c = 0; c < DIM; ++c)
          for (int d = 0; d < DIM; ++d) acc[c * DIM + d] += m * qd[(G_::F_G + c * DIM + d) * NQ + q] + wg * ga[c] * gb[d];
      }
    }
    s += __shfl_xor(s, 32, 64);
    for (int i = 0; i < DD; ++i) acc[i] += __shfl_xor(acc[i], 32, 64);
    if (valid && half == 0) {
      const uint16_t pos = posuu;
      for (int c = 0; c < DIM; ++c) acc[c * DIM + c] += s;
      for (int i = 0; i < DD; ++i) rowbuf[size_t(pos) * DD + i] += acc[i];
      sbuf[pos] += s;
    }
    // ---- B^T row entries Ke[(a,c), p_b] = -sum_q JxW d_c N_a psi_b
    if (lane >= 32 && lane < 32 + NP) {
      const int pb = lane - 32;
      double v[DIM];
      for (int c = 0; c < DIM; ++c) v[c] = 0;
      for (int q = 0; q < NQ; ++q) {
        const double wpsi = qd[G_::F_W * NQ + q] * T.psi[q * NP + pb];
        for (int c = 0; c < DIM; ++c) v[c] -= wpsi * aside[q * (DIM + 1) + 1 + c];
      }
      const uint16_t pos = posup;
      for (int c = 0; c < DIM; ++c) btbuf[size_t(pos) * DIM + c] += v[c];
    }
    if (A.n_neumann != 0 && lane == 0) {
      double nr[DIM];
      neumann_rhs<DIM, KV>(A, cell, a, nr);
      for (int c = 0; c < DIM; ++c) misc[c] = nr[c];
    }
    wave_sync();
    if (A.n_neumann != 0) for (int c = 0; c < DIM; ++c) fe[c] += misc[c];
    wave_sync();
  }
  // ---- constraints (distribute_local_to_global(..., true)) and write-out
  const int64_t p_off = int64_t(DIM) * A.nUl;
  bool rc[DIM]; double rg[DIM];
  for (int c = 0; c < DIM; ++c) { rc[c] = A.is_c ? A.is_c[I * DIM + c] : false; rg[c] = A.cval ? A.cval[I * DIM + c] : 0.0; }
  double corr[DIM], dg[DIM];
  for (int c = 0; c < DIM; ++c) { corr[c] = 0; dg[c] = 0; }
  for (int k = lane; k < len; k += 64) {
    const int64_t J = A.col_uu[rs + k];
    bool cc_[DIM]; double cg[DIM];
    for (int d = 0; d < DIM; ++d) { cc_[d] = A.is_c ? A.is_c[J * DIM + d] : false; cg[d] = A.cval ? A.cval[J * DIM + d] : 0.0; }
    for (int c = 0; c < DIM; ++c)
      for (int d = 0; d < DIM; ++d) {
        const double v = rowbuf[size_t(k) * DD + c * DIM + d];
        double out;
        if (rc[c]) { out = (J == I && c == d) ? fabs(v) : 0.0; if (J == I && c == d) dg[c] = fabs(v); }
        else if (cc_[d]) { out = 0.0; if (A.use_inhom) corr[c] -= v * cg[d]; }
        else out = v;
        A.v_uu[uu_base(rs, len, k, DD) + int64_t(c * DIM + d) * uu_estride(len)] = out;
      }
    if (A.v_s) A.v_s[rs + k] = sbuf[k];
  }
  for (int k = lane; k < tlen; k += 64) {
    for (int c = 0; c < DIM; ++c) {
      const double v = btbuf[size_t(k) * DIM + c];
      A.v_bt[ts * DIM + int64_t(c) * tlen + k] = rc[c] ? 0.0 : v; // pressure dofs carry no Dirichlet lines (checked at set_constraints)
    }
  }
  for (int c = 0; c < DIM; ++c) { corr[c] = wave_sum<DIM>(corr[c]); dg[c] = wave_sum<DIM>(dg[c]); }
  if (lane == 0) {
    for (int c = 0; c < DIM; ++c) {
      A.rhs[I * DIM + c] = rc[c] ? (A.use_inhom ? rg[c] * dg[c] : 0.0) : fe[c] + corr[c];
      A.diagMu[I * DIM + c] = mass;
    }
  }
  (void)p_off;
  wave_sync();
  } // row loop
}

// ------------------------------------------------------------------------------------------------- pass 3
template <int DIM, int KV, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_rows_p(RowArgs A) {
  using G_ = RG<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ, QD = G_::QD;
  constexpr int QH = (NQ + 1) / 2;
  extern __shared__ __align__(16) unsigned char smem_p[];
  auto &T = *reinterpret_cast<RowTables<DIM, KV> *>(smem_p);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t per_wave = size_t(NQ) * QD + size_t(A.maxlen_b) * DIM + size_t(A.maxlen_mp) + 8;
  double *W = reinterpret_cast<double *>(smem_p + sizeof(RowTables<DIM, KV>)) + size_t(wave) * per_wave;
  double *qd = W, *brow = qd + NQ * QD, *mrow = brow + size_t(A.maxlen_b) * DIM;
  for (int i = threadIdx.x; i < NQ * NU; i += blockDim.x) T.phi[i] = A.fe->phi[i];
  for (int i = threadIdx.x; i < NQ * NU * DIM; i += blockDim.x) T.dphi[i] = A.fe->dphi[i];
  for (int i = threadIdx.x; i < NQ * NP; i += blockDim.x) T.psi[i] = A.fe->psi[i];
  __syncthreads();
  for (int64_t I = int64_t(blockIdx.x) * WPB + wave; I < A.nPo; I += int64_t(gridDim.x) * WPB) {
  const int64_t bs = A.rp_b[I], ms = A.rp_mp[I];
  const int blen = int(A.rp_b[I + 1] - bs), mlen = int(A.rp_mp[I + 1] - ms);
  for (int i = lane; i < blen * DIM; i += 64) brow[i] = 0;
  for (int i = lane; i < mlen; i += 64) mrow[i] = 0;
  double fe = 0;
  wave_sync();
  for (int64_t inc = A.pinc_ptr[I]; inc < A.pinc_ptr[I + 1]; ++inc) {
    const int32_t packed = A.pinc[inc];
    const int64_t cell = packed >> 5;
    const int pa = packed & 31;
    const double *src = A.qdata + cell * (NQ * QD);
    for (int i = lane; i < NQ * QD; i += 64) qd[i] = src[i];
    wave_sync();
    double t = 0;
    if (lane < NQ) t = qd[G_::F_DIV * NQ + lane] * T.psi[lane * NP + pa] * qd[G_::F_W * NQ + lane]; // (div u) psi_i (:290)
    fe += wave_sum<DIM>(t);
    const int b = lane & 31, half = lane >> 5;
    double v[DIM];
    for (int c = 0; c < DIM; ++c) v[c] = 0;
    if (b < NU) {
      const int q0 = half ? QH : 0, q1 = half ? NQ : QH;
      for (int q = q0; q < q1; ++q) {
        const double wpsi = qd[G_::F_W * NQ + q] * T.psi[q * NP + pa];
        for (int d = 0; d < DIM; ++d) {
          double g = 0;
          for (int e = 0; e < DIM; ++e) g += T.dphi[(q * NU + b) * DIM + e] * qd[(G_::F_JI + e * DIM + d) * NQ + q];
          v[d] -= wpsi * g;
        }
      }
    }
    for (int c = 0; c < DIM; ++c) v[c] += __shfl_xor(v[c], 32, 64);
    if (b < NU && half == 0) {
      const uint16_t pos = A.posPU[(cell * NP + pa) * NU + b];
      for (int c = 0; c < DIM; ++c) brow[size_t(pos) * DIM + c] += v[c];
    }
    if (lane >= 32 && lane < 32 + NP) {
      const int pb = lane - 32;
      double m = 0;
      for (int q = 0; q < NQ; ++q) m += qd[G_::F_W * NQ + q] * T.psi[q * NP + pa] * T.psi[q * NP + pb];
      mrow[A.posPP[(cell * NP + pa) * NP + pb]] += m;
    }
    wave_sync();
  }
  double corr = 0;
  for (int k = lane; k < blen; k += 64) {
    const int64_t J = A.col_b[bs + k];
    for (int d = 0; d < DIM; ++d) {
      const double v = brow[size_t(k) * DIM + d];
      const bool cst = A.is_c ? A.is_c[J * DIM + d] : false;
      if (cst && A.use_inhom) corr -= v * A.cval[J * DIM + d];
      A.v_b[bs * DIM + int64_t(d) * blen + k] = cst ? 0.0 : v;
    }
  }
  for (int k = lane; k < mlen; k += 64) A.v_mp[ms + k] = mrow[k];
  corr = wave_sum<DIM>(corr);
  if (lane == 0) A.rhs[int64_t(DIM) * A.nUo + I] = fe + corr;
  wave_sync();
  } // row loop
}

// ------------------------------------------------------------------------------------------------- host side
template <int DIM, int KV>
static void launch_rows_t(ifem_ctx *ctx, RowArgs &A) {
  using G_ = RG<DIM, KV>;
  constexpr int WPB = 6; // 6 x 18 kB of per-wave scratch + 25 kB of tables < 160 kB LDS
  hipStream_t s = ctx->stream;
  const size_t need = size_t(ctx->n_cells) * G_::NQ * G_::QD;
  if (ctx->qdata.n < need) ctx->qdata.alloc(need);
  A.qdata = ctx->qdata.p;
  hipLaunchKernelGGL((k_cell_qdata<DIM, KV>), dim3(unsigned((ctx->n_cells + 3) / 4)), dim3(256), 0, s, A);
  const size_t smem_u = sizeof(RowTables<DIM, KV>) +
                        WPB * sizeof(double) * (size_t(G_::NQ) * G_::QD + size_t(G_::NQ) * (DIM + 1) +
                                                size_t(A.maxlen_uu) * (DIM * DIM + 1) + size_t(A.maxlen_bt) * DIM + 8);
  const size_t smem_p = sizeof(RowTables<DIM, KV>) +
                        WPB * sizeof(double) * (size_t(G_::NQ) * G_::QD + size_t(A.maxlen_b) * DIM + size_t(A.maxlen_mp) + 8);
  if (smem_u > 160 * 1024 || smem_p > 160 * 1024) throw Error(IFEM_E_BADPARAM, "row assembly: rows too long for LDS");
  IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows_u<DIM, KV, WPB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_u));
  IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rows_p<DIM, KV, WPB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_p));
  auto grid = [](int64_t rows) { const int64_t g = (rows + WPB - 1) / WPB; return unsigned(g < 2048 ? g : 2048); };
  if (ctx->nUo) hipLaunchKernelGGL((k_rows_u<DIM, KV, WPB>), dim3(grid(ctx->nUo)), dim3(64 * WPB), smem_u, s, A);
  if (ctx->nPo) hipLaunchKernelGGL((k_rows_p<DIM, KV, WPB>), dim3(grid(ctx->nPo)), dim3(64 * WPB), smem_p, s, A);
  IFEM_HIP_CHECK(hipGetLastError());
}

void launch_ins_assemble_rows(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) {
  hipStream_t s = ctx->stream;
  if (ctx->uinc.n_rows == 0 && ctx->nUo) build_incidence(ctx);
  RowArgs A{};
  A.n_cells = ctx->n_cells; A.nUo = ctx->nUo; A.nUl = ctx->nUl; A.nPo = ctx->nPo;
  A.fe = ctx->d_fe.p;
  A.vcoords = ctx->vcoords.p; A.cell_unodes = ctx->cell_unodes.p; A.cell_pnodes = ctx->cell_pnodes.p;
  A.cell_face_bid = ctx->cell_face_bid.p; A.indicator = ctx->indicator.p;
  A.posUU = ctx->posUU.p; A.posUP = ctx->posUP.p; A.posPU = ctx->posPU.p; A.posPP = ctx->posPP.p;
  A.rp_uu = ctx->Auu.rowptr.p; A.rp_bt = ctx->Bt.rowptr.p; A.rp_b = ctx->B.rowptr.p; A.rp_mp = ctx->Mp.rowptr.p;
  A.col_uu = ctx->Auu.col.p; A.col_b = ctx->B.col.p;
  A.v_uu = ctx->Auu.val.p; A.v_bt = ctx->Bt.val.p; A.v_b = ctx->B.val.p; A.v_mp = ctx->Mp.val.p;
  A.diagMu = ctx->diagMu.p; A.rhs = ctx->vec[IFEM_VEC_RHS].p;
  if (ctx->want_shat && ctx->Shat.n != (size_t)ctx->Auu.nnzb) ctx->Shat.alloc((size_t)ctx->Auu.nnzb);
  A.v_s = ctx->want_shat ? ctx->Shat.p : nullptr;
  A.uinc_ptr = ctx->uinc.rowptr.p; A.uinc = ctx->uinc.col.p;
  A.pinc_ptr = ctx->pinc.rowptr.p; A.pinc = ctx->pinc.col.p;
  const int w = use_nonzero ? 1 : 0;
  A.is_c = ctx->has_c[w] ? ctx->is_c[w].p : nullptr;
  A.cval = ctx->has_c[w] ? ctx->cval[w].p : nullptr;
  A.use_inhom = (use_nonzero && ctx->has_c[1]) ? 1 : 0;
  A.eval = ctx->vec[IFEM_VEC_EVAL].p; A.present = ctx->vec[IFEM_VEC_PRESENT].p;
  A.fsi_acc = ctx->indicator.p ? ctx->vec[IFEM_VEC_FSI_ACC].p : nullptr;
  A.mu = p->viscosity; A.rho = p->rho; A.gamma = p->grad_div; A.inv_dt = 1.0 / p->dt;
  for (int i = 0; i < 3; ++i) A.g[i] = p->gravity[i];
  A.n_neumann = p->n_neumann;
  for (int i = 0; i < 8; ++i) { A.neumann_id[i] = p->neumann_id[i]; A.neumann_p[i] = p->neumann_p[i]; }
  A.maxlen_uu = ctx->Auu.max_row; A.maxlen_bt = ctx->Bt.max_row; A.maxlen_b = ctx->B.max_row; A.maxlen_mp = ctx->Mp.max_row;
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
  const int dim = ctx->dim;
  if (dim == 2 && ctx->kv == 1) launch_rows_t<2, 1>(ctx, A);
  else if (dim == 2 && ctx->kv == 2) launch_rows_t<2, 2>(ctx, A);
  else if (dim == 3 && ctx->kv == 1) launch_rows_t<3, 1>(ctx, A);
  else if (dim == 3 && ctx->kv == 2) launch_rows_t<3, 2>(ctx, A);
  else throw Error(IFEM_E_BADPARAM, "unsupported (dim, kv)");
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
}

} // namespace ifem
