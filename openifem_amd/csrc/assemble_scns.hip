// assemble_scns.hip -- Fluid::MPI::SCnsIM::assemble (source/mpi_scnsim.cpp:15-568) and FluidSolver::update_stress
// (source/mpi_fluid_solver.cpp:716-811) on gfx950.
//
// Slightly compressible Navier-Stokes with SUPG / PSPG / LSIC stabilisation (Tezduyar UGN parameters from the PRESENT
// velocity, including the reference's length-scale quirk), PML damping, body force, FSI terms and the
// divergence of the projected nodal viscous stress.  One wavefront per cell; per quadrature point the fields and the
// stabilisation parameters are staged in LDS, then every lane integrates (i, j) entries of the local matrix with the
// reference's expression (deal.II Tensor operator* semantics) and scatters them with the Dirichlet rules of
// distribute_local_to_global into A_uu / A_up / A_pu / A_pp.  All shipped SCnsIM cases are Q1/Q1
// (12 or 32 local dofs), where this kernel is latency-, not throughput-bound.
#include <hip/hip_runtime.h>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

template <int DIM, int KV>
struct SG {
  static constexpr int N1 = KV + 1;
  static constexpr int NU = (DIM == 2) ? N1 * N1 : N1 * N1 * N1;
  static constexpr int NP = (DIM == 2) ? 4 : 8;
  static constexpr int NQ = NU;
  static constexpr int ND = NU * DIM + NP;
  static constexpr int NS = DIM * (DIM + 1) / 2;
};

struct ScnsArgs {
  int64_t n_cells, nUo, nUl, nPo;
  const FeTables *fe;
  const double *vcoords;
  const int32_t *cell_unodes, *cell_pnodes, *cell_face_bid, *indicator;
  const uint16_t *posUU, *posUP, *posPU, *posPP;
  const int64_t *rp_uu, *rp_bt, *rp_b, *rp_mp;
  double *v_uu, *v_bt, *v_b, *v_pp, *rhs;
  const uint8_t *is_c;
  const double *cval;
  const double *eval, *present, *fsi_acc;
  const double *stress;     // [DIM*DIM][nUl] or nullptr
  const double *fsi_stress; // [NS][nUl] or nullptr
  const double *sigma_pml;  // [n_cells][NQ] or nullptr
  const double *body_force; // [n_cells][NQ][DIM] or nullptr
  const double *eddy;       // [nUl] nodal eddy viscosity or nullptr
  double mu, rho_f, rho_s, dt;
  double g[3];
  int n_neumann;
  int neumann_id[8];
  double neumann_p[8];
  int use_inhom;
  int inc; // IFEM_FORM_SUPG_INSIM: the incompressible integrand of mpi_insim_supg.cpp
};

template <int DIM>
__device__ inline double inv_s(const double *J, double *Ji) {
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

// per quadrature point data kept in LDS
template <int DIM>
struct QPoint {
  double JxW, rho, visc, tS, tP, tL, pr, p0, sigma, cdiv;
  double u[DIM], u0[DIM], du[DIM], acc[DIM], gp[DIM], sdiv[DIM], gbf[DIM], uG[DIM], Gu[DIM], R[DIM];
  double G[DIM * DIM], fT[DIM * DIM];
};

template <int DIM, int KV>
struct ScnsScratch {
  using G_ = SG<DIM, KV>;
  double X[G_::NP * DIM];
  double gN[G_::NQ * G_::NU * DIM];
  double gP[G_::NQ * G_::NP * DIM];
  QPoint<DIM> qp[G_::NQ];
  double ue[G_::NU * DIM], u0e[G_::NU * DIM], ae[G_::NU * DIM], pe[G_::NP], p0e[G_::NP];
  double se[DIM * DIM * G_::NU], fse[G_::NS * G_::NU], eve[G_::NU];
  double fe[G_::ND], cv[G_::ND];
  int64_t rs_uu[G_::NU], rs_bt[G_::NU], rs_b[G_::NP], rs_pp[G_::NP];
  int32_t len_uu[G_::NU], len_bt[G_::NU], len_b[G_::NP], len_pp[G_::NP];
  int32_t un[G_::NU], pn[G_::NP];
  uint8_t cf[G_::ND + 7];
};

template <int DIM>
__device__ inline double dotd(const double *a, const double *b) {
  double t = 0;
#pragma unroll
  for (int i = 0; i < DIM; ++i) t += a[i] * b[i];
  return t;
}

// shape data of system shape function k at quadrature point q: component (DIM = pressure), value, gradient
template <int DIM, int KV>
struct Shape {
  int c;
  double N;
  double g[DIM];
  __device__ Shape(const ScnsScratch<DIM, KV> &S, const FeTables &T, int k, int q) {
    using G_ = SG<DIM, KV>;
    if (k < G_::NU * DIM) {
      const int a = k / DIM;
      c = k - a * DIM;
      N = T.phi[q * G_::NU + a];
      for (int d = 0; d < DIM; ++d) g[d] = S.gN[(q * G_::NU + a) * DIM + d];
    } else {
      const int b = k - G_::NU * DIM;
      c = DIM;
      N = T.psi[q * G_::NP + b];
      for (int d = 0; d < DIM; ++d) g[d] = S.gP[(q * G_::NP + b) * DIM + d];
    }
  }
  __device__ void phi_u(double *v) const { for (int d = 0; d < DIM; ++d) v[d] = (d == c) ? N : 0.0; }
  __device__ double div_phi_u() const { return c < DIM ? g[c] : 0.0; }
  __device__ double phi_p() const { return c == DIM ? N : 0.0; }
  __device__ void grad_phi_p(double *v) const { for (int d = 0; d < DIM; ++d) v[d] = (c == DIM) ? g[d] : 0.0; }
  // v * grad_phi_u: r_b = sum_a v_a (grad phi)_ab = v_c g_b
  __device__ void v_times_grad(const double *v, double *r) const { const double s = c < DIM ? v[c] : 0.0; for (int d = 0; d < DIM; ++d) r[d] = s * g[d]; }
  // grad_phi_u * v: r_a = (a == c) g . v
  __device__ void grad_times_v(const double *v, double *r) const { const double s = c < DIM ? dotd<DIM>(g, v) : 0.0; for (int d = 0; d < DIM; ++d) r[d] = (d == c) ? s : 0.0; }
};

template <int DIM, int KV, int WPB>
__global__ __launch_bounds__(64 * WPB) void k_scns_assemble(ScnsArgs A) {
  using G_ = SG<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ, ND = G_::ND, NS = G_::NS;
  extern __shared__ __align__(16) unsigned char smem_c[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto &S = *reinterpret_cast<ScnsScratch<DIM, KV> *>(smem_c + size_t(wave) * sizeof(ScnsScratch<DIM, KV>));
  const FeTables &T = *A.fe;
  const int64_t cell = int64_t(blockIdx.x) * WPB + wave;
  const bool active = cell < A.n_cells;
  const int64_t cc = active ? cell : 0;
  const int64_t p_off = int64_t(DIM) * A.nUl;
  const bool inc = A.inc != 0;
  const int ind = (active && A.indicator && !inc) ? A.indicator[cc] : 0; // SUPGInsIM has no artificial-fluid terms
  const double cp_to_cv = 1.4, atm = 1013250, kappa_s = 1e4; // mpi_scnsim.cpp:124-126
  // ---- phase 0: gather
  for (int i = lane; i < NP * DIM; i += 64) S.X[i] = A.vcoords[cc * NP * DIM + i];
  for (int a = lane; a < NU; a += 64) {
    const int32_t nd = A.cell_unodes[cc * NU + a];
    S.un[a] = nd;
    const bool own = nd < A.nUo;
    const int64_t r0 = own ? A.rp_uu[nd] : 0, r1 = own ? A.rp_uu[nd + 1] : 0;
    S.rs_uu[a] = r0; S.len_uu[a] = own ? int32_t(r1 - r0) : -1;
    const int64_t t0 = own ? A.rp_bt[nd] : 0, t1 = own ? A.rp_bt[nd + 1] : 0;
    S.rs_bt[a] = t0; S.len_bt[a] = own ? int32_t(t1 - t0) : -1;
    for (int c = 0; c < DIM; ++c) {
      const int64_t dof = int64_t(DIM) * nd + c;
      S.ue[a * DIM + c] = A.eval[dof];
      S.u0e[a * DIM + c] = A.present[dof];
      S.ae[a * DIM + c] = A.fsi_acc ? A.fsi_acc[dof] : 0.0;
      S.cf[a * DIM + c] = A.is_c ? A.is_c[dof] : 0;
      S.cv[a * DIM + c] = A.cval ? A.cval[dof] : 0.0;
    }
    for (int k = 0; k < DIM * DIM; ++k) S.se[k * NU + a] = A.stress ? A.stress[int64_t(k) * A.nUl + nd] : 0.0;
    for (int k = 0; k < NS; ++k) S.fse[k * NU + a] = A.fsi_stress ? A.fsi_stress[int64_t(k) * A.nUl + nd] : 0.0;
    S.eve[a] = (A.eddy && !inc) ? A.eddy[nd] : 0.0;
  }
  for (int b = lane; b < NP; b += 64) {
    const int32_t nd = A.cell_pnodes[cc * NP + b];
    S.pn[b] = nd;
    const bool own = nd < A.nPo;
    const int64_t r0 = own ? A.rp_b[nd] : 0, r1 = own ? A.rp_b[nd + 1] : 0;
    S.rs_b[b] = r0; S.len_b[b] = own ? int32_t(r1 - r0) : -1;
    const int64_t m0 = own ? A.rp_mp[nd] : 0, m1 = own ? A.rp_mp[nd + 1] : 0;
    S.rs_pp[b] = m0; S.len_pp[b] = own ? int32_t(m1 - m0) : -1;
    S.pe[b] = A.eval[p_off + nd];
    S.p0e[b] = A.present[p_off + nd];
    S.cf[NU * DIM + b] = A.is_c ? A.is_c[p_off + nd] : 0;
    S.cv[NU * DIM + b] = A.cval ? A.cval[p_off + nd] : 0.0;
  }
  for (int i = lane; i < ND; i += 64) S.fe[i] = 0;
  __syncthreads();
  // ---- phase 1: geometry, fields and stabilisation parameters per quadrature point (:146-289)
  for (int q = lane; q < NQ; q += 64) {
    double J[DIM * DIM], Ji[DIM * DIM];
    for (int i = 0; i < DIM * DIM; ++i) J[i] = 0;
    for (int v = 0; v < NP; ++v)
      for (int d = 0; d < DIM; ++d)
        for (int e = 0; e < DIM; ++e) J[d * DIM + e] += S.X[v * DIM + d] * T.dpsi[(q * NP + v) * DIM + e];
    const double det = inv_s<DIM>(J, Ji);
    QPoint<DIM> &Q = S.qp[q];
    Q.JxW = fabs(det) * T.w[q];
    for (int a = 0; a < NU; ++a)
      for (int d = 0; d < DIM; ++d) {
        double g = 0;
        for (int e = 0; e < DIM; ++e) g += T.dphi[(q * NU + a) * DIM + e] * Ji[e * DIM + d];
        S.gN[(q * NU + a) * DIM + d] = g;
      }
    for (int b = 0; b < NP; ++b)
      for (int d = 0; d < DIM; ++d) {
        double g = 0;
        for (int e = 0; e < DIM; ++e) g += T.dpsi[(q * NP + b) * DIM + e] * Ji[e * DIM + d];
        S.gP[(q * NP + b) * DIM + d] = g;
      }
    double u[DIM], u0[DIM], acc[DIM], G[DIM * DIM], gp[DIM], sg[DIM * DIM * DIM], fs[NS];
    double pr = 0, p0 = 0, evq = 0;
    for (int c = 0; c < DIM; ++c) { u[c] = 0; u0[c] = 0; acc[c] = 0; gp[c] = 0; }
    for (int i = 0; i < DIM * DIM; ++i) G[i] = 0;
    for (int i = 0; i < DIM * DIM * DIM; ++i) sg[i] = 0;
    for (int i = 0; i < NS; ++i) fs[i] = 0;
    for (int a = 0; a < NU; ++a) {
      const double N = T.phi[q * NU + a];
      const double *ga = &S.gN[(q * NU + a) * DIM];
      for (int c = 0; c < DIM; ++c) {
        const double ue = S.ue[a * DIM + c];
        u[c] += N * ue; u0[c] += N * S.u0e[a * DIM + c]; acc[c] += N * S.ae[a * DIM + c];
        for (int d = 0; d < DIM; ++d) G[c * DIM + d] += ue * ga[d];
      }
      for (int k = 0; k < DIM * DIM; ++k)
        for (int d = 0; d < DIM; ++d) sg[k * DIM + d] += S.se[k * NU + a] * ga[d];
      for (int k = 0; k < NS; ++k) fs[k] += N * S.fse[k * NU + a];
      evq += N * S.eve[a];
    }
    for (int b = 0; b < NP; ++b) {
      pr += T.psi[q * NP + b] * S.pe[b];
      p0 += T.psi[q * NP + b] * S.p0e[b];
      for (int d = 0; d < DIM; ++d) gp[d] += S.pe[b] * S.gP[(q * NP + b) * DIM + d];
    }
    const double sigma = (A.sigma_pml && !inc) ? A.sigma_pml[cc * NQ + q] : 0.0;
    const double rho = inc ? A.rho_f : A.rho_f * (1 + p0 / atm) * (1 - ind) + ind * A.rho_s; // :210-213 | insim_supg :109
    const double visc = (ind == 1 ? 1.0 : A.mu) + (evq > 0.0 ? evq : 0.0);    // :214-216
    // UGN length scale: first ND/(DIM+1) system shape functions in deal.II's vertex-major order (:252-258)
    double h = 0;
    for (int a = 0; a < ND / (DIM + 1); ++a) {
      const int v = a / (DIM + 1), comp = a % (DIM + 1);
      const double *gs;
      if (comp < DIM) {
        int la = 0, stride = 1;
        for (int d = 0; d < DIM; ++d) { la += ((v >> d) & 1) * KV * stride; stride *= (KV + 1); }
        gs = &S.gN[(q * NU + la) * DIM];
      } else gs = &S.gP[(q * NP + v) * DIM];
      h += fabs(dotd<DIM>(u0, gs));
    }
    const double vn = sqrt(dotd<DIM>(u0, u0));
    h = (h != 0.0) ? 2 * vn / h : 0.0;
    const double nu = visc / rho;
    double tS;
    if (h != 0.0) { const double a1 = 2 / A.dt, a2 = 2 * vn / h, a3 = 4 * nu / (h * h); tS = 1 / sqrt(a1 * a1 + a2 * a2 + a3 * a3); }
    else tS = A.dt / 2;
    const double Re = vn * h / (2 * nu);
    const double z = Re <= 3 ? (Re / 3) : 1.0;
    Q.rho = rho; Q.visc = visc; Q.tS = tS; Q.tP = tS / rho; Q.tL = h / 2 * vn * z;
    Q.pr = pr; Q.p0 = p0; Q.sigma = sigma;
    double cdiv = 0;
    for (int c = 0; c < DIM; ++c) cdiv += G[c * DIM + c];
    Q.cdiv = cdiv;
    for (int c = 0; c < DIM; ++c) {
      Q.u[c] = u[c]; Q.u0[c] = u0[c]; Q.du[c] = u[c] - u0[c]; Q.acc[c] = acc[c]; Q.gp[c] = gp[c];
      double sd = 0;
      for (int j = 0; j < DIM; ++j) sd += sg[(c * DIM + j) * DIM + j];
      Q.sdiv[c] = inc ? 0.0 : sd * visc / A.mu;
      Q.gbf[c] = A.g[c] + (A.body_force ? A.body_force[(cc * NQ + q) * DIM + c] : 0.0);
      double t1 = 0, t2 = 0;
      for (int a = 0; a < DIM; ++a) { t1 += u[a] * G[a * DIM + c]; t2 += G[c * DIM + a] * u[a]; }
      Q.uG[c] = t1; Q.Gu[c] = t2;
    }
    for (int i = 0; i < DIM * DIM; ++i) { Q.G[i] = G[i]; Q.fT[i] = 0; }
    if (ind != 0) { int si = 0; for (int k = 0; k < DIM; ++k) for (int m = 0; m <= k; ++m) { Q.fT[k * DIM + m] = Q.fT[m * DIM + k] = fs[si++]; } }
    for (int c = 0; c < DIM; ++c)
      Q.R[c] = rho * (Q.du[c] / A.dt + Q.uG[c]) + gp[c] - Q.sdiv[c] - rho * Q.gbf[c] + rho * sigma * u[c];
  }
  __syncthreads();
  // ---- phase 2: rhs (:425-512)
  for (int i = lane; i < ND; i += 64) {
    double f = 0;
    for (int q = 0; q < NQ; ++q) {
      const QPoint<DIM> &Q = S.qp[q];
      const Shape<DIM, KV> si(S, T, i, q);
      double pu[DIM], gpi[DIM], uGi[DIM];
      si.phi_u(pu); si.grad_phi_p(gpi); si.v_times_grad(Q.u, uGi);
      const double dvi = si.div_phi_u(), ppi = si.phi_p();
      double sp = 0; // scalar_product(grad u, grad_phi_u[i]) = sum_b G[c][b] g_b
      if (si.c < DIM) for (int b = 0; b < DIM; ++b) sp += Q.G[si.c * DIM + b] * si.g[b];
      double r = ((-Q.visc * sp - Q.rho * dotd<DIM>(Q.Gu, pu) + Q.pr * dvi) - Q.rho * dotd<DIM>(Q.du, pu) / A.dt +
                  dotd<DIM>(Q.gbf, pu) * Q.rho);
      if (inc) { // mpi_insim_supg.cpp:236-262 (sigma = sdiv = 0 make Q.R the incompressible residual)
        r += -(Q.cdiv * ppi);
        r += -(Q.tS * dotd<DIM>(uGi, Q.R) + Q.tP * dotd<DIM>(gpi, Q.R));
        r += -(Q.tL * Q.rho * dvi) * Q.cdiv;
      } else {
      r += -(Q.rho * Q.sigma * dotd<DIM>(Q.u, pu) + Q.sigma * Q.pr * ppi / atm);
      r += -(cp_to_cv * (atm + Q.pr * (1 - ind)) * Q.cdiv * ppi + dotd<DIM>(Q.u, Q.gp) * ppi * (1 - ind) +
             (Q.pr - Q.p0) * ppi / A.dt * (1 - ind)) / atm -
           1 / kappa_s * (Q.pr - Q.p0) * ppi * ind / A.dt;
      r += -(Q.tS * dotd<DIM>(uGi, Q.R) + Q.tP * dotd<DIM>(gpi, Q.R));
      r += -((Q.tL * Q.rho * dvi) * ((Q.pr - Q.p0) / A.dt * (1 - ind) + cp_to_cv * atm * Q.cdiv +
                                      cp_to_cv * Q.pr * Q.cdiv * (1 - ind) + dotd<DIM>(Q.u, Q.gp) * (1 - ind)) / atm +
             (Q.tL * Q.rho * dvi) * (1 / kappa_s * (Q.pr - Q.p0) / A.dt) * ind);
      }
      if (ind == 1) {
        double spf = 0;
        if (si.c < DIM) for (int b = 0; b < DIM; ++b) spf += si.g[b] * Q.fT[si.c * DIM + b];
        double w[DIM], ar[DIM];
        for (int c = 0; c < DIM; ++c) { w[c] = pu[c] + Q.tP * gpi[c] + Q.tS * uGi[c]; ar[c] = Q.acc[c] * Q.rho; }
        r += spf + dotd<DIM>(ar, w);
      }
      f += r * Q.JxW;
    }
    S.fe[i] = f;
  }
  __syncthreads();
  // Neumann faces (:521-549)
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
        for (int qf = 0; qf < T.nqf; ++qf) {
          double J[DIM * DIM], Ji[DIM * DIM];
          for (int k = 0; k < DIM * DIM; ++k) J[k] = 0;
          const double *dps = &T.fdpsi[(f * T.nqf + qf) * NP * DIM];
          for (int v = 0; v < NP; ++v)
            for (int d = 0; d < DIM; ++d)
              for (int e = 0; e < DIM; ++e) J[d * DIM + e] += S.X[v * DIM + d] * dps[v * DIM + e];
          const double det = inv_s<DIM>(J, Ji);
          acc += T.fphi[(f * T.nqf + qf) * NU + a] * (sgn * Ji[nd * DIM + c]) * pbc * fabs(det) * T.fw[qf];
        }
        S.fe[i] -= acc;
      }
    }
  }
  __syncthreads();
  // ---- phase 3: local matrix entries (:291-421) + scatter
  for (int t = lane; t < ND * ND; t += 64) {
    const int i = t / ND, j = t - i * ND;
    double v = 0;
    for (int q = 0; q < NQ; ++q) {
      const QPoint<DIM> &Q = S.qp[q];
      const Shape<DIM, KV> si(S, T, i, q), sj(S, T, j, q);
      double pui[DIM], puj[DIM], gpi[DIM], gpj[DIM], uGi[DIM], uGj[DIM], pjGi[DIM], pjG[DIM], Gpj[DIM], Gju[DIM];
      si.phi_u(pui); sj.phi_u(puj); si.grad_phi_p(gpi); sj.grad_phi_p(gpj);
      si.v_times_grad(Q.u, uGi); sj.v_times_grad(Q.u, uGj); si.v_times_grad(puj, pjGi);
      for (int b = 0; b < DIM; ++b) { // phi_u[j] * grad u (row c_j of G); grad u * phi_u[j] (column c_j of G)
        pjG[b] = (sj.c < DIM) ? sj.N * Q.G[sj.c * DIM + b] : 0.0;
        Gpj[b] = (sj.c < DIM) ? Q.G[b * DIM + sj.c] * sj.N : 0.0;
      }
      sj.grad_times_v(Q.u, Gju);
      const double dvi = si.div_phi_u(), dvj = sj.div_phi_u(), ppi = si.phi_p(), ppj = sj.phi_p();
      const double sp = (si.c < DIM && si.c == sj.c) ? dotd<DIM>(si.g, sj.g) : 0.0;
      const double pipj = dotd<DIM>(pui, puj);
      const double rho = Q.rho, tS = Q.tS, tP = Q.tP, tL = Q.tL, sig = Q.sigma;
      double e = ((Q.visc * sp + rho * dotd<DIM>(Gpj, pui) + rho * dotd<DIM>(Gju, pui) - dvi * ppj) + rho * pipj / A.dt);
      if (inc) { // mpi_insim_supg.cpp:181-232
        e += (tS * rho * dotd<DIM>(uGi, pjG) + tS * rho * dotd<DIM>(uGi, uGj) + tS * rho * dotd<DIM>(pjGi, Q.uG) +
              tS * rho * dotd<DIM>(uGi, puj) / A.dt + tS * rho * dotd<DIM>(pjGi, Q.du) / A.dt + tS * dotd<DIM>(uGi, gpj) +
              tS * dotd<DIM>(pjGi, Q.gp) - tS * rho * dotd<DIM>(pjGi, Q.gbf) + tP * rho * dotd<DIM>(gpi, pjG) +
              tP * rho * dotd<DIM>(gpi, uGj) + tP * rho * dotd<DIM>(gpi, puj) / A.dt + tP * dotd<DIM>(gpi, gpj) +
              tL * rho * dvi * dvj);
        e += dvj * ppi;
        v += e * Q.JxW;
        continue;
      }
      e += (rho * sig * pipj + sig * ppj * ppi / atm);
      e += (tS * rho * dotd<DIM>(uGi, pjG) + tS * rho * dotd<DIM>(uGi, uGj) + tS * rho * dotd<DIM>(pjGi, Q.uG) +
            tS * rho * dotd<DIM>(uGi, puj) / A.dt + tS * rho * dotd<DIM>(pjGi, Q.du) / A.dt + tS * dotd<DIM>(uGi, gpj) +
            tS * dotd<DIM>(pjGi, Q.gp) - tS * dotd<DIM>(pjGi, Q.sdiv) - tS * rho * dotd<DIM>(pjGi, Q.gbf) +
            tS * rho * sig * dotd<DIM>(uGi, puj) + tS * rho * sig * dotd<DIM>(pjGi, Q.u) + tP * rho * dotd<DIM>(gpi, pjG) +
            tP * rho * dotd<DIM>(gpi, uGj) + tP * rho * dotd<DIM>(gpi, puj) / A.dt + tP * dotd<DIM>(gpi, gpj) +
            tP * rho * sig * dotd<DIM>(gpi, puj) + tL * rho * dvi * ppj / A.dt * (1 - ind) / atm +
            tL * rho * 1 / kappa_s * dvi * ppj / A.dt * ind + tL * rho * cp_to_cv * dvi * dvj +
            tL * rho * cp_to_cv * dvi * Q.pr * (1 - ind) * dvj / atm + tL * rho * cp_to_cv * dvi * ppj * (1 - ind) * Q.cdiv / atm +
            tL * rho * dvi * dotd<DIM>(Q.u, gpj) / atm * (1 - ind) + tL * rho * dvi * dotd<DIM>(puj, Q.gp) / atm * (1 - ind));
      e += (cp_to_cv * (atm + Q.pr * (1 - ind)) * dvj * ppi + ppj * Q.cdiv * ppi * (1 - ind) +
            dotd<DIM>(Q.u, gpj) * ppi * (1 - ind) + dotd<DIM>(puj, Q.gp) * ppi * (1 - ind) + ppi * ppj / A.dt * (1 - ind)) / atm +
           1 / kappa_s * ppi * ppj * ind / A.dt;
      if (ind == 1) { double ar[DIM]; for (int c = 0; c < DIM; ++c) ar[c] = Q.acc[c] * rho; e += -(tS * dotd<DIM>(pjGi, ar)); }
      v += e * Q.JxW;
    }
    if (!active) continue;
    // scatter with Dirichlet elimination
    const bool ri = S.cf[i], cj = S.cf[j];
    const bool iu = i < NU * DIM, ju = j < NU * DIM;
    const int ai = iu ? i / DIM : i - NU * DIM, ci = iu ? i - ai * DIM : 0;
    const int aj = ju ? j / DIM : j - NU * DIM, cjj = ju ? j - aj * DIM : 0;
    const int rlen = iu ? (ju ? S.len_uu[ai] : S.len_bt[ai]) : (ju ? S.len_b[ai] : S.len_pp[ai]);
    if (rlen < 0) continue; // row owned elsewhere
    double *dst;
    if (iu && ju) dst = A.v_uu + uu_base(S.rs_uu[ai], rlen, A.posUU[(cc * NU + ai) * NU + aj], DIM * DIM) + int64_t(ci * DIM + cjj) * uu_estride(rlen);
    else if (iu) dst = A.v_bt + S.rs_bt[ai] * DIM + int64_t(ci) * rlen + A.posUP[(cc * NU + ai) * NP + aj];
    else if (ju) dst = A.v_b + S.rs_b[ai] * DIM + int64_t(cjj) * rlen + A.posPU[(cc * NP + ai) * NU + aj];
    else dst = A.v_pp + S.rs_pp[ai] + A.posPP[(cc * NP + ai) * NP + aj];
    const int64_t row_dof = iu ? int64_t(DIM) * S.un[ai] + ci : int64_t(DIM) * A.nUo + S.pn[ai];
    if (!ri && !cj) unsafeAtomicAdd(dst, v);
    else if (ri) {
      if (i == j) {
        unsafeAtomicAdd(dst, fabs(v));
        if (A.use_inhom) unsafeAtomicAdd(&A.rhs[row_dof], S.cv[i] * fabs(v));
      }
    } else if (A.use_inhom && S.cv[j] != 0.0) unsafeAtomicAdd(&S.fe[i], -v * S.cv[j]);
  }
  __syncthreads();
  if (active)
    for (int i = lane; i < ND; i += 64) {
      if (S.cf[i]) continue;
      if (i < NU * DIM) { const int a = i / DIM, c = i - a * DIM; if (S.len_uu[a] >= 0) unsafeAtomicAdd(&A.rhs[int64_t(DIM) * S.un[a] + c], S.fe[i]); }
      else { const int b = i - NU * DIM; if (S.len_b[b] >= 0) unsafeAtomicAdd(&A.rhs[int64_t(DIM) * A.nUo + S.pn[b]], S.fe[i]); }
    }
}

template <int DIM, int KV>
static void launch_scns_t(ifem_ctx *ctx, const ScnsArgs &A) {
  constexpr int WPB = (DIM == 3 && KV == 2) ? 2 : 4;
  const size_t smem = WPB * sizeof(ScnsScratch<DIM, KV>);
  if (smem > 160 * 1024) throw Error(IFEM_E_BADPARAM, "SCnsIM kernel: element too large for LDS");
  IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scns_assemble<DIM, KV, WPB>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  const int64_t nblk = (A.n_cells + WPB - 1) / WPB;
  hipLaunchKernelGGL((k_scns_assemble<DIM, KV, WPB>), dim3((unsigned)nblk), dim3(64 * WPB), smem, ctx->stream, A);
  IFEM_HIP_CHECK(hipGetLastError());
}

void launch_scns_assemble(ifem_ctx *ctx, const ifem_scns_params *p, int use_nonzero) {
  hipStream_t s = ctx->stream;
  KScope ks(ctx, IFEM_KC_ASSEMBLE, 16.0 * double(ctx->Auu.val.n + ctx->Bt.val.n + ctx->B.val.n + ctx->Mp.val.n)); // (zero fill + cell kernel + block-Jacobi set-up)
  if (ctx->App.n != ctx->Mp.val.n) ctx->App.alloc(ctx->Mp.val.n);
  ensure_auu_values(ctx);
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->Auu.val.p, 0, ctx->Auu.val.n * sizeof(double), s));
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->Bt.val.p, 0, ctx->Bt.val.n * sizeof(double), s));
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->B.val.p, 0, ctx->B.val.n * sizeof(double), s));
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->App.p, 0, ctx->App.n * sizeof(double), s));
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->vec[IFEM_VEC_RHS].p, 0, ctx->vec[IFEM_VEC_RHS].n * sizeof(double), s));
  ScnsArgs A{};
  A.n_cells = ctx->n_cells; A.nUo = ctx->nUo; A.nUl = ctx->nUl; A.nPo = ctx->nPo;
  A.fe = ctx->d_fe.p;
  A.vcoords = ctx->vcoords.p; A.cell_unodes = ctx->cell_unodes.p; A.cell_pnodes = ctx->cell_pnodes.p;
  A.cell_face_bid = ctx->cell_face_bid.p; A.indicator = ctx->indicator.p;
  A.posUU = ctx->posUU.p; A.posUP = ctx->posUP.p; A.posPU = ctx->posPU.p; A.posPP = ctx->posPP.p;
  A.rp_uu = ctx->Auu.rowptr.p; A.rp_bt = ctx->Bt.rowptr.p; A.rp_b = ctx->B.rowptr.p; A.rp_mp = ctx->Mp.rowptr.p;
  A.v_uu = ctx->Auu.val.p; A.v_bt = ctx->Bt.val.p; A.v_b = ctx->B.val.p; A.v_pp = ctx->App.p;
  A.rhs = ctx->vec[IFEM_VEC_RHS].p;
  const int w = use_nonzero ? 1 : 0;
  A.is_c = ctx->has_c[w] ? ctx->is_c[w].p : nullptr;
  A.cval = ctx->has_c[w] ? ctx->cval[w].p : nullptr;
  A.use_inhom = (use_nonzero && ctx->has_c[1]) ? 1 : 0;
  A.eval = ctx->vec[IFEM_VEC_EVAL].p; A.present = ctx->vec[IFEM_VEC_PRESENT].p;
  A.fsi_acc = ctx->indicator.p ? ctx->vec[IFEM_VEC_FSI_ACC].p : nullptr;
  A.stress = ctx->stress_valid ? ctx->stress.p : nullptr;
  A.fsi_stress = ctx->fsi_stress.n ? ctx->fsi_stress.p : nullptr;
  A.sigma_pml = ctx->sigma_pml.n ? ctx->sigma_pml.p : nullptr;
  A.body_force = ctx->body_force.n ? ctx->body_force.p : nullptr;
  A.eddy = ctx->eddy_viscosity.n ? ctx->eddy_viscosity.p : nullptr;
  A.mu = p->viscosity; A.rho_f = p->rho; A.rho_s = p->solid_rho; A.dt = p->dt;
  if (p->formulation != IFEM_FORM_SCNSIM && p->formulation != IFEM_FORM_SUPG_INSIM) throw Error(IFEM_E_BADPARAM, "ifem_scns_params.formulation");
  A.inc = p->formulation == IFEM_FORM_SUPG_INSIM;
  for (int i = 0; i < 3; ++i) A.g[i] = p->gravity[i];
  A.n_neumann = p->n_neumann;
  for (int i = 0; i < 8; ++i) { A.neumann_id[i] = p->neumann_id[i]; A.neumann_p[i] = p->neumann_p[i]; }
  const int dim = ctx->dim;
  if (dim == 2 && ctx->kv == 1) launch_scns_t<2, 1>(ctx, A);
  else if (dim == 2 && ctx->kv == 2) launch_scns_t<2, 2>(ctx, A);
  else if (dim == 3 && ctx->kv == 1) launch_scns_t<3, 1>(ctx, A);
  else if (dim == 3 && ctx->kv == 2) launch_scns_t<3, 2>(ctx, A);
  else throw Error(IFEM_E_BADPARAM, "unsupported (dim, kv)");
  bjac_setup(ctx);
  ctx->assembled = true;
  ctx->has_app = true;
  ctx->mf_valid = false;
  ctx->auu_f32_valid = false;
  ctx->bbt_f32_valid = false;
  ctx->sm_valid = false; ctx->sm_key = -1;
  ctx->shat_valid = false;
  ctx->tpp_valid = false; ctx->tpp_ilu.factored = false;
  ctx->b2_valid = false; ctx->pvv_ilu.factored = false; ctx->b2_ilu.factored = false;
  ctx->geo_valid = false; // B / B^T now hold the SUPG-stabilised blocks
  ctx->asm_constraint_set = use_nonzero ? 1 : 0;
  hanging_condense_rhs(ctx, use_nonzero);
}

// ---------------------------------------------------------------------------------------------------------------
// FluidSolver::update_stress (mpi_fluid_solver.cpp:716-811): tau = 2 mu sym(grad u) at the quadrature points of the
// PRESENT solution, per-cell projection to the Q_kv nodes (n_q == n_nodes: interpolation through Phi^-1), nodal average
template <int DIM, int KV>
__global__ __launch_bounds__(256) void k_update_stress(int64_t n_cells, int64_t nUl, const FeTables *fe,
                                                       const double *__restrict__ Xinv, const double *__restrict__ vcoords,
                                                       const int32_t *__restrict__ cell_unodes, const double *__restrict__ present,
                                                       double mu, double *__restrict__ stress, double *__restrict__ cnt) {
  using G_ = SG<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ;
  __shared__ double qs[4][DIM * DIM][NQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t cell = int64_t(blockIdx.x) * 4 + wave;
  const bool active = cell < n_cells;
  const int64_t cc = active ? cell : 0;
  const FeTables &T = *fe;
  for (int q = lane; q < NQ; q += 64) {
    double J[DIM * DIM], Ji[DIM * DIM], G[DIM * DIM];
    for (int i = 0; i < DIM * DIM; ++i) { J[i] = 0; G[i] = 0; }
    for (int v = 0; v < NP; ++v)
      for (int d = 0; d < DIM; ++d)
        for (int e = 0; e < DIM; ++e) J[d * DIM + e] += vcoords[(cc * NP + v) * DIM + d] * T.dpsi[(q * NP + v) * DIM + e];
    inv_s<DIM>(J, Ji);
    for (int a = 0; a < NU; ++a) {
      const int64_t nd = cell_unodes[cc * NU + a];
      double g[DIM];
      for (int d = 0; d < DIM; ++d) { double t = 0; for (int e = 0; e < DIM; ++e) t += T.dphi[(q * NU + a) * DIM + e] * Ji[e * DIM + d]; g[d] = t; }
      for (int c = 0; c < DIM; ++c) { const double ue = present[nd * DIM + c]; for (int d = 0; d < DIM; ++d) G[c * DIM + d] += ue * g[d]; }
    }
    for (int i = 0; i < DIM; ++i) for (int j = 0; j < DIM; ++j) qs[wave][i * DIM + j][q] = mu * (G[i * DIM + j] + G[j * DIM + i]);
  }
  __syncthreads();
  if (!active) return;
  for (int a = lane; a < NU; a += 64) {
    const int64_t nd = cell_unodes[cc * NU + a];
    for (int k = 0; k < DIM * DIM; ++k) {
      double t = 0;
      for (int q = 0; q < NQ; ++q) t += Xinv[a * NQ + q] * qs[wave][k][q];
      unsafeAtomicAdd(&stress[int64_t(k) * nUl + nd], t);
    }
    unsafeAtomicAdd(&cnt[nd], 1.0);
  }
}

// [plane i*dim + j][node] <-> interleaved [node][j] for one i: lets the velocity halo plan refresh the ghost stresses
__global__ void k_stress_pack(int64_t n, int dim, int i, const double *__restrict__ stress, double *__restrict__ buf, int dir) {
  for (int64_t t = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; t < n * dim; t += int64_t(gridDim.x) * blockDim.x) {
    const int64_t nd = t / dim;
    const int j = int(t - nd * dim);
    if (dir == 0) buf[t] = stress[int64_t(i * dim + j) * n + nd];
    else const_cast<double *>(stress)[int64_t(i * dim + j) * n + nd] = buf[t];
  }
}

__global__ void k_stress_avg(int64_t n, int nk, const double *__restrict__ cnt, double *__restrict__ stress) {
  for (int64_t i = int64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < n * nk; i += int64_t(gridDim.x) * blockDim.x)
    stress[i] /= cnt[i % n];
}

void launch_update_stress(ifem_ctx *ctx, double mu) {
  hipStream_t s = ctx->stream;
  const int dim = ctx->dim, nu = ctx->nu, nq = ctx->nq;
  const int64_t n = ctx->nUl;
  if (ctx->stress.n != (size_t)dim * dim * n) ctx->stress.alloc((size_t)dim * dim * n);
  if (ctx->xinv.n == 0) { // Phi^-1 on the host (Gauss-Jordan, partial pivoting)
    std::vector<double> P((size_t)nq * nu), X((size_t)nu * nq, 0.0);
    for (int q = 0; q < nq; ++q) for (int i = 0; i < nu; ++i) P[(size_t)q * nu + i] = ctx->fe.phi[q * nu + i];
    for (int i = 0; i < nu; ++i) X[(size_t)i * nq + i] = 1.0;
    for (int c = 0; c < nu; ++c) {
      int piv = c;
      for (int r = c + 1; r < nu; ++r) if (std::fabs(P[(size_t)r * nu + c]) > std::fabs(P[(size_t)piv * nu + c])) piv = r;
      if (piv != c) for (int k = 0; k < nu; ++k) { std::swap(P[(size_t)c * nu + k], P[(size_t)piv * nu + k]); std::swap(X[(size_t)c * nq + k], X[(size_t)piv * nq + k]); }
      const double d = P[(size_t)c * nu + c];
      for (int k = 0; k < nu; ++k) { P[(size_t)c * nu + k] /= d; X[(size_t)c * nq + k] /= d; }
      for (int r = 0; r < nu; ++r) if (r != c) { const double f = P[(size_t)r * nu + c]; for (int k = 0; k < nu; ++k) { P[(size_t)r * nu + k] -= f * P[(size_t)c * nu + k]; X[(size_t)r * nq + k] -= f * X[(size_t)c * nq + k]; } }
    }
    ctx->xinv.upload(X.data(), X.size(), s);
    IFEM_HIP_CHECK(hipStreamSynchronize(s));
  }
  DBuf<double> cnt;
  cnt.alloc(n);
  IFEM_HIP_CHECK(hipMemsetAsync(cnt.p, 0, n * sizeof(double), s));
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->stress.p, 0, ctx->stress.n * sizeof(double), s));
  const unsigned blocks = unsigned((ctx->n_cells + 3) / 4);
#define IFEM_US(D, K) hipLaunchKernelGGL((k_update_stress<D, K>), dim3(blocks), dim3(256), 0, s, ctx->n_cells, n, ctx->d_fe.p, \
                                         ctx->xinv.p, ctx->vcoords.p, ctx->cell_unodes.p, ctx->vec[IFEM_VEC_PRESENT].p, mu,   \
                                         ctx->stress.p, cnt.p)
  if (dim == 2 && ctx->kv == 1) IFEM_US(2, 1);
  else if (dim == 2 && ctx->kv == 2) IFEM_US(2, 2);
  else if (dim == 3 && ctx->kv == 1) IFEM_US(3, 1);
  else IFEM_US(3, 2);
#undef IFEM_US
  hipLaunchKernelGGL(k_stress_avg, dim3(1024), dim3(256), 0, s, n, dim * dim, cnt.p, ctx->stress.p);
  if (ctx->halo.nranks > 1) { // owned nodes saw every cell that touches them; ghosts take the owner's average
    DBuf<double> buf;
    buf.alloc((size_t)dim * n);
    for (int i = 0; i < dim; ++i) {
      hipLaunchKernelGGL(k_stress_pack, dim3(1024), dim3(256), 0, s, n, dim, i, ctx->stress.p, buf.p, 0);
      halo_exchange(ctx, buf.p);
      hipLaunchKernelGGL(k_stress_pack, dim3(1024), dim3(256), 0, s, n, dim, i, ctx->stress.p, buf.p, 1);
    }
  }
  IFEM_HIP_CHECK(hipStreamSynchronize(s));
  ctx->stress_valid = true;
}

} // namespace ifem
