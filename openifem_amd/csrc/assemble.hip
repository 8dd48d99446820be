// assemble.hip -- InsIM::assemble on gfx950 (reference: source/mpi_insim.cpp:153-362).
//
// One wavefront (64 lanes) integrates one cell: MappingQ1 Jacobians per quadrature point, physical shape
// gradients, evaluation-point fields, then the Newton-linearised INS weak form in component-block form
// (SURVEY A.2):
//   Ke[(a,c),(b,d)] = sum_q JxW { d_cd [ mu gN_a.gN_b + rho N_a (u.gN_b) + rho/dt N_a N_b ]
//                                 + rho N_a N_b d_d u_c + gamma rho d_c N_a d_d N_b }
//   Ke[(a,c),p_b]   = -sum_q JxW d_c N_a  psi_b          (and its transpose)
// and scatters with AffineConstraints::distribute_local_to_global(..., true) semantics straight
// into the row-planar block matrices with hardware f64 atomics (global_atomic_add_f64).
// Reference-cell tables (shape values / gradients at the Gauss points) are staged in LDS once per workgroup.
#include <hip/hip_runtime.h>
#include "ctx.hpp"
#include "kernels.hpp"
#include "assemble_common.hpp"

namespace ifem {

template <int DIM, int KV>
struct CellScratch {
  using G_ = Geo<DIM, KV>;
  double X[G_::NP * DIM];
  double G[G_::NQ * G_::NU * DIM]; // physical gradients [q][a][d]
  double JxW[G_::NQ];
  double uq[G_::NQ * DIM], gq[G_::NQ * DIM * DIM], pq[G_::NQ], u0q[G_::NQ * DIM], aq[G_::NQ * DIM], divq[G_::NQ];
  double ue[G_::NU * DIM], u0e[G_::NU * DIM], ae[G_::NU * DIM], pe[G_::NP];
  double fe[G_::ND];
  double cv[G_::ND];
  int64_t rs_uu[G_::NU], rs_bt[G_::NU], rs_b[G_::NP], rs_mp[G_::NP];
  int32_t len_uu[G_::NU], len_bt[G_::NU], len_b[G_::NP], len_mp[G_::NP];
  int32_t un[G_::NU], pn[G_::NP];
  uint8_t cf[G_::ND + 7];
};

template <int DIM, int KV>
struct SharedTables {
  using G_ = Geo<DIM, KV>;
  double phi[G_::NQ * G_::NU];
  double dphi[G_::NQ * G_::NU * DIM];
  double psi[G_::NQ * G_::NP];
  double dpsi[G_::NQ * G_::NP * DIM];
  double w[G_::NQ];
};

template <int DIM, int KV, int WPB, bool ATOMIC>
__global__ __launch_bounds__(64 * WPB) void k_ins_assemble(AsmArgs A) {
  using G_ = Geo<DIM, KV>;
  constexpr int NU = G_::NU, NP = G_::NP, NQ = G_::NQ, ND = G_::ND;
  extern __shared__ __align__(16) unsigned char smem[];
  auto &T = *reinterpret_cast<SharedTables<DIM, KV> *>(smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  auto &S = *reinterpret_cast<CellScratch<DIM, KV> *>(smem + sizeof(SharedTables<DIM, KV>) +
                                                     size_t(wave) * sizeof(CellScratch<DIM, KV>));
  // ---- stage reference tables (FeTables arrays are dimensioned for the largest element: re-stride)
  for (int i = threadIdx.x; i < NQ * NU; i += blockDim.x) T.phi[i] = A.fe->phi[i];
  for (int i = threadIdx.x; i < NQ * NU * DIM; i += blockDim.x) T.dphi[i] = A.fe->dphi[i];
  for (int i = threadIdx.x; i < NQ * NP; i += blockDim.x) T.psi[i] = A.fe->psi[i];
  for (int i = threadIdx.x; i < NQ * NP * DIM; i += blockDim.x) T.dpsi[i] = A.fe->dpsi[i];
  for (int i = threadIdx.x; i < NQ; i += blockDim.x) T.w[i] = A.fe->w[i];

  const int64_t idx = int64_t(blockIdx.x) * WPB + wave;
  const bool active = idx < A.count;
  const int64_t cc = active ? (A.order ? int64_t(A.order[A.first + idx]) : idx) : 0;
  const int64_t p_off = int64_t(DIM) * A.nUl;

  // ---- phase 0: ids, coordinates, nodal values, row descriptors, constraint flags
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
  }
  for (int b = lane; b < NP; b += 64) {
    const int32_t nd = A.cell_pnodes[cc * NP + b];
    S.pn[b] = nd;
    const bool own = nd < A.nPo;
    const int64_t r0 = own ? A.rp_b[nd] : 0, r1 = own ? A.rp_b[nd + 1] : 0;
    S.rs_b[b] = r0; S.len_b[b] = own ? int32_t(r1 - r0) : -1;
    const int64_t m0 = own ? A.rp_mp[nd] : 0, m1 = own ? A.rp_mp[nd + 1] : 0;
    S.rs_mp[b] = m0; S.len_mp[b] = own ? int32_t(m1 - m0) : -1;
    S.pe[b] = A.eval[p_off + nd];
    S.cf[NU * DIM + b] = A.is_c ? A.is_c[p_off + nd] : 0;
    S.cv[NU * DIM + b] = A.cval ? A.cval[p_off + nd] : 0.0;
  }
  __syncthreads();

  // ---- phase 1: per quadrature point Jacobian (MappingQ1), JxW, physical gradients  (fe_values.reinit, :213)
  for (int q = lane; q < NQ; q += 64) {
    double J[DIM * DIM], Ji[DIM * DIM];
    for (int i = 0; i < DIM * DIM; ++i) J[i] = 0;
    for (int v = 0; v < NP; ++v)
      for (int d = 0; d < DIM; ++d)
        for (int e = 0; e < DIM; ++e) J[d * DIM + e] += S.X[v * DIM + d] * T.dpsi[(q * NP + v) * DIM + e];
    const double det = inv_small<DIM>(J, Ji);
    S.JxW[q] = fabs(det) * T.w[q];
    for (int a = 0; a < NU; ++a)
      for (int d = 0; d < DIM; ++d) {
        double g = 0;
        for (int e = 0; e < DIM; ++e) g += T.dphi[(q * NU + a) * DIM + e] * Ji[e * DIM + d];
        S.G[(q * NU + a) * DIM + d] = g;
      }
  }
  __syncthreads();
  // ---- phase 2: fields at quadrature points (get_function_values / gradients, :219-232)
  for (int q = lane; q < NQ; q += 64) {
    double u[DIM], u0[DIM], ac[DIM], g[DIM * DIM], p = 0;
    for (int c = 0; c < DIM; ++c) { u[c] = 0; u0[c] = 0; ac[c] = 0; }
    for (int i = 0; i < DIM * DIM; ++i) g[i] = 0;
    for (int a = 0; a < NU; ++a) {
      const double N = T.phi[q * NU + a];
      for (int c = 0; c < DIM; ++c) {
        const double ue = S.ue[a * DIM + c];
        u[c] += N * ue; u0[c] += N * S.u0e[a * DIM + c]; ac[c] += N * S.ae[a * DIM + c];
        for (int d = 0; d < DIM; ++d) g[c * DIM + d] += ue * S.G[(q * NU + a) * DIM + d];
      }
    }
    for (int b = 0; b < NP; ++b) p += T.psi[q * NP + b] * S.pe[b];
    double dv = 0;
    for (int c = 0; c < DIM; ++c) { S.uq[q * DIM + c] = u[c]; S.u0q[q * DIM + c] = u0[c]; S.aq[q * DIM + c] = ac[c]; dv += g[c * DIM + c]; }
    for (int i = 0; i < DIM * DIM; ++i) S.gq[q * DIM * DIM + i] = g[i];
    S.pq[q] = p; S.divq[q] = dv;
  }
  __syncthreads();
  const int ind = (active && A.indicator) ? A.indicator[cc] : 0;
  // ---- phase 3: local rhs  (:281-304)
  for (int i = lane; i < ND; i += 64) {
    double f = 0;
    if (i < NU * DIM) {
      const int a = i / DIM, c = i - a * DIM;
      for (int q = 0; q < NQ; ++q) {
        const double N = T.phi[q * NU + a];
        const double *g = &S.G[(q * NU + a) * DIM];
        double visc = 0, adv = 0;
        for (int d = 0; d < DIM; ++d) { visc += S.gq[q * DIM * DIM + c * DIM + d] * g[d]; adv += S.gq[q * DIM * DIM + c * DIM + d] * S.uq[q * DIM + d]; }
        double t = -A.mu * visc - A.rho * adv * N + S.pq[q] * g[c] - A.gamma * A.rho * S.divq[q] * g[c] -
                   A.rho * A.inv_dt * (S.uq[q * DIM + c] - S.u0q[q * DIM + c]) * N + A.rho * A.g[c] * N;
        if (ind == 1) t += A.rho * S.aq[q * DIM + c] * N;
        f += t * S.JxW[q];
      }
    } else {
      const int b = i - NU * DIM;
      for (int q = 0; q < NQ; ++q) f += S.divq[q] * T.psi[q * NP + b] * S.JxW[q];
    }
    S.fe[i] = f;
  }
  __syncthreads();
  // ---- phase 3b: Neumann (pressure) boundary faces  (:313-341)
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
  __syncthreads();

  const double wgam = A.gamma * A.rho, rdt = A.rho * A.inv_dt;
  // ---- phase 4: velocity-velocity blocks + scatter
  // (plain read-modify-write variant: the old values are requested before the contraction and consumed after it, so
  //  the load latency hides behind ~1000 FMAs instead of stalling the wave)
  for (int t = lane; t < NU * NU; t += 64) {
    const int a = t / NU, b = t - a * NU;
    const int len = S.len_uu[a];
    const bool row_here = active && len >= 0; // row owned by this rank
    double *base = nullptr;
    double old[DIM * DIM];
    if (row_here) {
      const uint16_t pos = A.posUU[(cc * NU + a) * NU + b];
      base = A.v_uu + uu_base(S.rs_uu[a], len, pos, DIM * DIM);
      if constexpr (!ATOMIC) {
#pragma unroll
        for (int e = 0; e < DIM * DIM; ++e) old[e] = base[int64_t(e) * uu_estride(len)];
      }
    }
    double s = 0, acc[DIM * DIM];
    for (int i = 0; i < DIM * DIM; ++i) acc[i] = 0;
    for (int q = 0; q < NQ; ++q) {
      const double w = S.JxW[q], Na = T.phi[q * NU + a], Nb = T.phi[q * NU + b];
      const double *ga = &S.G[(q * NU + a) * DIM], *gb = &S.G[(q * NU + b) * DIM];
      double gg = 0, ugb = 0;
      for (int d = 0; d < DIM; ++d) { gg += ga[d] * gb[d]; ugb += S.uq[q * DIM + d] * gb[d]; }
      s += w * (A.mu * gg + A.rho * Na * ugb + rdt * Na * Nb);
      const double m = w * A.rho * Na * Nb, wg = w * wgam;
      for (int c = 0; c < DIM; ++c)
        for (int d = 0; d < DIM; ++d) acc[c * DIM + d] += m * S.gq[q * DIM * DIM + c * DIM + d] + wg * ga[c] * gb[d];
    }
    for (int c = 0; c < DIM; ++c) acc[c * DIM + c] += s;
    if (!row_here) continue;
    const int64_t row_dof0 = int64_t(DIM) * S.un[a];
    if (A.v_s) gadd<ATOMIC>(A.v_s + S.rs_uu[a] + A.posUU[(cc * NU + a) * NU + b], s);
    for (int c = 0; c < DIM; ++c) {
      const bool rc = S.cf[a * DIM + c];
      for (int d = 0; d < DIM; ++d) {
        const bool ccn = S.cf[b * DIM + d];
        const double v = acc[c * DIM + d];
        double *dst = base + int64_t(c * DIM + d) * uu_estride(len);
        if (!rc && !ccn) {
          if constexpr (ATOMIC) unsafeAtomicAdd(dst, v); else *dst = old[c * DIM + d] + v;
        } else if (rc) {
          if (a == b && c == d) { // |Ke(r,r)| on the diagonal, rhs so that the update equals the inhomogeneity
            if constexpr (ATOMIC) unsafeAtomicAdd(dst, fabs(v)); else *dst = old[c * DIM + d] + fabs(v);
            if (A.use_inhom) gadd<ATOMIC>(&A.rhs[row_dof0 + c], S.cv[a * DIM + c] * fabs(v));
          }
        } else if (A.use_inhom) {
          const double g = S.cv[b * DIM + d];
          if (g != 0.0) unsafeAtomicAdd(&S.fe[a * DIM + c], -v * g);
        }
      }
    }
  }
  // ---- phase 5: velocity-pressure blocks (block (0,1) = B^T and block (1,0) = B)
  for (int t = lane; t < NU * NP; t += 64) {
    const int a = t / NP, pb = t - a * NP;
    const bool bt_here = active && S.len_bt[a] >= 0, b_here = active && S.len_b[pb] >= 0;
    double *base_bt = nullptr, *base_b = nullptr;
    double old_bt[DIM], old_b[DIM];
    if (bt_here) {
      base_bt = A.v_bt + S.rs_bt[a] * DIM + A.posUP[(cc * NU + a) * NP + pb];
      if constexpr (!ATOMIC)
        for (int c = 0; c < DIM; ++c) old_bt[c] = base_bt[int64_t(c) * S.len_bt[a]];
    }
    if (b_here) {
      base_b = A.v_b + S.rs_b[pb] * DIM + A.posPU[(cc * NP + pb) * NU + a];
      if constexpr (!ATOMIC)
        for (int c = 0; c < DIM; ++c) old_b[c] = base_b[int64_t(c) * S.len_b[pb]];
    }
    double v[DIM];
    for (int c = 0; c < DIM; ++c) v[c] = 0;
    for (int q = 0; q < NQ; ++q) {
      const double wpsi = S.JxW[q] * T.psi[q * NP + pb];
      for (int c = 0; c < DIM; ++c) v[c] -= wpsi * S.G[(q * NU + a) * DIM + c];
    }
    if (!active) continue;
    const bool pc = S.cf[NU * DIM + pb];
    if (bt_here) {
      const int len = S.len_bt[a];
      for (int c = 0; c < DIM; ++c) {
        if (S.cf[a * DIM + c]) continue;
        if (!pc) { if constexpr (ATOMIC) unsafeAtomicAdd(base_bt + int64_t(c) * len, v[c]); else base_bt[int64_t(c) * len] = old_bt[c] + v[c]; }
        else if (A.use_inhom && S.cv[NU * DIM + pb] != 0.0) unsafeAtomicAdd(&S.fe[a * DIM + c], -v[c] * S.cv[NU * DIM + pb]);
      }
    }
    if (b_here && !pc) {
      const int len = S.len_b[pb];
      for (int c = 0; c < DIM; ++c) {
        if (!S.cf[a * DIM + c]) { if constexpr (ATOMIC) unsafeAtomicAdd(base_b + int64_t(c) * len, v[c]); else base_b[int64_t(c) * len] = old_b[c] + v[c]; }
        else if (A.use_inhom && S.cv[a * DIM + c] != 0.0) unsafeAtomicAdd(&S.fe[NU * DIM + pb], -v[c] * S.cv[a * DIM + c]);
      }
    }
  }
  // ---- phase 6: pressure mass matrix M_p and diag(M_u)  (:274-276, only (0,0) diagonal and (1,1) are used)
  for (int t = lane; t < NP * NP; t += 64) {
    const int pa = t / NP, pb = t - pa * NP;
    double m = 0;
    for (int q = 0; q < NQ; ++q) m += S.JxW[q] * T.psi[q * NP + pa] * T.psi[q * NP + pb];
    if (!active || S.len_mp[pa] < 0) continue;
    const bool ra = S.cf[NU * DIM + pa], cb = S.cf[NU * DIM + pb];
    double *dst = A.v_mp + S.rs_mp[pa] + A.posPP[(cc * NP + pa) * NP + pb];
    if (!ra && !cb) gadd<ATOMIC>(dst, m);
    else if (ra && pa == pb) gadd<ATOMIC>(dst, fabs(m));
  }
  for (int a = lane; a < NU; a += 64) {
    double m = 0;
    for (int q = 0; q < NQ; ++q) { const double N = T.phi[q * NU + a]; m += S.JxW[q] * N * N; }
    if (active && S.len_uu[a] >= 0)
      for (int c = 0; c < DIM; ++c) gadd<ATOMIC>(&A.diagMu[int64_t(DIM) * S.un[a] + c], m);
  }
  __syncthreads();
  // ---- phase 7: rhs scatter (unconstrained owned rows; constrained rows were handled with the diagonal)
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
static void launch_t(ifem_ctx *ctx, const AsmArgs &A) {
  constexpr int WPB = (DIM == 3 && KV == 2) ? 4 : 4;
  const size_t smem = sizeof(SharedTables<DIM, KV>) + WPB * sizeof(CellScratch<DIM, KV>);
  static bool attr_set = false;
  if (!attr_set) {
    IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ins_assemble<DIM, KV, WPB, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ins_assemble<DIM, KV, WPB, false>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    attr_set = true;
  }
  static const bool use_rmw = [] { const char *e = getenv("IFEM_ASM_SCATTER"); return e && std::string(e) == "rmw"; }();
  if (ctx->color_ptr.empty() || !use_rmw) { // one launch, hardware atomics (see assemble2.hip for the measurement)
    AsmArgs B = A;
    B.order = nullptr; B.first = 0; B.count = A.n_cells;
    const int64_t nblk = (B.count + WPB - 1) / WPB;
    hipLaunchKernelGGL((k_ins_assemble<DIM, KV, WPB, true>), dim3((unsigned)nblk), dim3(64 * WPB), smem, ctx->stream, B);
  } else { // one launch per colour (stream order separates them): conflict-free plain read-modify-write
    for (size_t k = 0; k + 1 < ctx->color_ptr.size(); ++k) {
      AsmArgs B = A;
      B.order = ctx->color_order.p; B.first = ctx->color_ptr[k]; B.count = ctx->color_ptr[k + 1] - ctx->color_ptr[k];
      if (B.count == 0) continue;
      const int64_t nblk = (B.count + WPB - 1) / WPB;
      hipLaunchKernelGGL((k_ins_assemble<DIM, KV, WPB, false>), dim3((unsigned)nblk), dim3(64 * WPB), smem, ctx->stream, B);
    }
  }
  IFEM_HIP_CHECK(hipGetLastError());
}

static void assemble_epilogue(ifem_ctx *ctx, int use_nonzero);
void launch_ins_assemble2_kernel(ifem_ctx *ctx, const AsmArgs &A);
bool launch_ins_assemble3_kernel(ifem_ctx *ctx, const AsmArgs &A);

void launch_ins_assemble(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero) { launch_ins_assemble_ex(ctx, p, use_nonzero, 0, 1); }

// imex = 1: InsIMEX::assemble (mpi_insimex.cpp:150-355): every field comes from the present solution, the matrix has no
// Zeroing the 78 GB of A_uu values (128^3 Q2) in front of the scatter costs 12 ms.  With a second value buffer the fill
// runs on a side stream and the next assembly swaps the buffers.  Measured at 128^3: underneath the cell kernel the fill
// competes for the write path of the atomics (112.6 -> 124 ms); started behind the cell kernel it slows the block-Jacobi
// set-up and the first pressure solves by about what it saves (step 513 -> 509..515 ms) -- 12 ms of HBM writes cost 12 ms
// wherever they run in this pipeline.  Not worth 78 GB: opt-in only, IFEM_AUU_SPARE=1 (when the matrix is large enough to
// matter and free memory >= 2 x the buffer), =2 always (tests).
static bool spare_buffer_ready(ifem_ctx *ctx) {
  if (ctx->spare_state) return ctx->spare_state > 0;
  const char *e = getenv("IFEM_AUU_SPARE");
  const int mode = e ? atoi(e) : 0;
  const size_t bytes = ctx->Auu.val.n * sizeof(double);
  size_t free_b = 0, total_b = 0;
  bool ok = mode != 0 && bytes > 0 && hipMemGetInfo(&free_b, &total_b) == hipSuccess;
  if (ok && mode != 2) ok = bytes >= (size_t(256) << 20) && free_b >= 2 * bytes;
  if (ok) ok = hipMalloc((void **)&ctx->Auu_spare.p, bytes) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); ctx->spare_state = -1; return false; }
  ctx->Auu_spare.n = ctx->Auu.val.n;
  IFEM_HIP_CHECK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
  IFEM_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_main, hipEventDisableTiming));
  IFEM_HIP_CHECK(hipEventCreateWithFlags(&ctx->ev_spare, hipEventDisableTiming));
  ctx->spare_state = 1;
  return true;
}

// convective terms; assemble_system = 0 integrates the right-hand side only and leaves the matrices untouched
void launch_ins_assemble_ex(ifem_ctx *ctx, const ifem_ins_params *p, int use_nonzero, int imex, int assemble_system) {
  hipStream_t s = ctx->stream;
  const int dim = ctx->dim;
  static const bool v1_env = [] { const char *e = getenv("IFEM_ASM"); return e && std::string(e) == "v1"; }();
  if (imex && (v1_env || ctx->asm_rows)) throw Error(IFEM_E_BADPARAM, "InsIMEX assembly needs the default assembly kernel (unset IFEM_ASM)");
  if (!assemble_system && !ctx->assembled) throw Error(IFEM_E_BADPARAM, "rhs-only assembly before any matrix assembly");
  if (assemble_system) { // state the matrix-free A_uu needs to reproduce this matrix (apply_mf.hip)
    const size_t nu = size_t(dim) * size_t(ctx->nUl);
    if (ctx->mf_eval.n != nu) ctx->mf_eval.alloc(nu);
    if (imex) IFEM_HIP_CHECK(hipMemsetAsync(ctx->mf_eval.p, 0, nu * sizeof(double), s)); // no convection in the IMEX matrix
    else IFEM_HIP_CHECK(hipMemcpyAsync(ctx->mf_eval.p, ctx->vec[IFEM_VEC_EVAL].p, nu * sizeof(double), hipMemcpyDeviceToDevice, s));
    ctx->mf_params = *p;
    ctx->mf_valid = true;
    ctx->mf_noconv = imex != 0;
  }
  if (ctx->asm_rows) {
    launch_ins_assemble_rows(ctx, p, use_nonzero);
    assemble_epilogue(ctx, use_nonzero);
    hanging_condense_rhs(ctx, use_nonzero);
    return;
  }
  // B, B^T, M_p and diag(M_u) depend on the mesh and on WHICH dofs are constrained, not on the solution or the
  // parameters: an assembly with the constraint set of the previous one keeps them (bit-identical to re-integrating
  // them) and integrates A_uu and the right-hand side only (not the first-generation kernel IFEM_ASM=v1).  IFEM_GEO_CACHE=0
  // switches it off.
  static const bool geo_cache_on = [] { const char *e = getenv("IFEM_GEO_CACHE"); return !e || atoi(e) != 0; }();
  static const bool other_kernel = [] { const char *e = getenv("IFEM_ASM"); return e && std::string(e) == "v1"; }();
  const int64_t geo_key = ctx->constraints_epoch * 2 + (use_nonzero ? 1 : 0);
  const bool skip_geo = geo_cache_on && assemble_system && !other_kernel && ctx->geo_valid && ctx->geo_key == geo_key;
  // system_matrix = 0; mass_matrix = 0; system_rhs = 0  (:163-165)
  bool refill_spare = false;
  if (assemble_system) {
    if (spare_buffer_ready(ctx)) { // the other buffer was zeroed while the previous matrix was in use
      if (ctx->spare_zeroing) {
        IFEM_HIP_CHECK(hipStreamWaitEvent(s, ctx->ev_spare, 0));
        std::swap(ctx->Auu.val.p, ctx->Auu_spare.p);
      } else
        IFEM_HIP_CHECK(hipMemsetAsync(ctx->Auu.val.p, 0, ctx->Auu.val.n * sizeof(double), s));
      refill_spare = true; // enqueued behind the cell kernel below
    } else
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->Auu.val.p, 0, ctx->Auu.val.n * sizeof(double), s));
    if (!skip_geo) {
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->Bt.val.p, 0, ctx->Bt.val.n * sizeof(double), s));
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->B.val.p, 0, ctx->B.val.n * sizeof(double), s));
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->Mp.val.p, 0, ctx->Mp.val.n * sizeof(double), s));
      IFEM_HIP_CHECK(hipMemsetAsync(ctx->diagMu.p, 0, ctx->diagMu.n * sizeof(double), s));
    }
  }
  if (ctx->want_shat && assemble_system) {
    if (ctx->Shat.n != (size_t)ctx->Auu.nnzb) ctx->Shat.alloc((size_t)ctx->Auu.nnzb);
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->Shat.p, 0, ctx->Shat.n * sizeof(double), s));
  }
  IFEM_HIP_CHECK(hipMemsetAsync(ctx->vec[IFEM_VEC_RHS].p, 0, ctx->vec[IFEM_VEC_RHS].n * sizeof(double), s));
  AsmArgs A{};
  A.n_cells = ctx->n_cells; A.nUo = ctx->nUo; A.nUl = ctx->nUl; A.nPo = ctx->nPo;
  A.fe = ctx->d_fe.p;
  A.vcoords = ctx->vcoords.p; A.cell_unodes = ctx->cell_unodes.p; A.cell_pnodes = ctx->cell_pnodes.p;
  A.cell_face_bid = ctx->cell_face_bid.p; A.indicator = ctx->indicator.p;
  A.posUU = ctx->posUU.p; A.posUP = ctx->posUP.p; A.posPU = ctx->posPU.p; A.posPP = ctx->posPP.p;
  A.rp_uu = ctx->Auu.rowptr.p; A.rp_bt = ctx->Bt.rowptr.p; A.rp_b = ctx->B.rowptr.p; A.rp_mp = ctx->Mp.rowptr.p;
  A.v_uu = ctx->Auu.val.p; A.v_bt = ctx->Bt.val.p; A.v_b = ctx->B.val.p; A.v_mp = ctx->Mp.val.p;
  A.diagMu = ctx->diagMu.p; A.rhs = ctx->vec[IFEM_VEC_RHS].p;
  A.v_s = ctx->want_shat ? ctx->Shat.p : nullptr;
  const int w = use_nonzero ? 1 : 0;
  A.is_c = ctx->has_c[w] ? ctx->is_c[w].p : nullptr;
  A.cval = ctx->has_c[w] ? ctx->cval[w].p : nullptr;
  A.use_inhom = (use_nonzero && ctx->has_c[1]) ? 1 : 0;
  A.skip_geo = skip_geo ? 1 : 0;
  { const char *e = getenv("IFEM_ASM_SKIP"); A.debug_skip = e ? atoi(e) : 0; }
  { const char *e = getenv("IFEM_XCD"); A.xcd_swizzle = e ? atoi(e) : 1; }
  A.eval = ctx->vec[imex ? IFEM_VEC_PRESENT : IFEM_VEC_EVAL].p; A.present = ctx->vec[IFEM_VEC_PRESENT].p;
  A.imex = imex; A.rhs_only = assemble_system ? 0 : 1;
  A.fsi_acc = ctx->indicator.p ? ctx->vec[IFEM_VEC_FSI_ACC].p : nullptr;
  A.mu = p->viscosity; A.rho = p->rho; A.gamma = p->grad_div; A.inv_dt = 1.0 / p->dt;
  for (int i = 0; i < 3; ++i) A.g[i] = p->gravity[i];
  A.n_neumann = p->n_neumann;
  for (int i = 0; i < 8; ++i) { A.neumann_id[i] = p->neumann_id[i]; A.neumann_p[i] = p->neumann_p[i]; }
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev0, s));
  static const bool v1 = [] { const char *e = getenv("IFEM_ASM"); return e && std::string(e) == "v1"; }();
  static const bool v2 = [] { const char *e = getenv("IFEM_ASM"); return e && std::string(e) == "v2"; }();
  if (!v1 && !v2 && launch_ins_assemble3_kernel(ctx, A)) {} // assemble3.hip: 3D Q2/Q1 on the FP64 matrix cores
  else if (!v1) launch_ins_assemble2_kernel(ctx, A);           // assemble2.hip (quadrature-point-outer, register accumulators)
  else if (dim == 2 && ctx->kv == 1) launch_t<2, 1>(ctx, A);
  else if (dim == 2 && ctx->kv == 2) launch_t<2, 2>(ctx, A);
  else if (dim == 3 && ctx->kv == 1) launch_t<3, 1>(ctx, A);
  else if (dim == 3 && ctx->kv == 2) launch_t<3, 2>(ctx, A);
  else throw Error(IFEM_E_BADPARAM, "unsupported (dim, kv)");
  IFEM_HIP_CHECK(hipEventRecord(ctx->ev1, s));
  if (refill_spare) {
    // the previous matrix (now the spare buffer) lost its last reader before this assembly; its fill starts when the cell
    // kernel is done
    IFEM_HIP_CHECK(hipEventRecord(ctx->ev_main, s));
    IFEM_HIP_CHECK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_main, 0));
    IFEM_HIP_CHECK(hipMemsetAsync(ctx->Auu_spare.p, 0, ctx->Auu_spare.n * sizeof(double), ctx->side_stream));
    IFEM_HIP_CHECK(hipEventRecord(ctx->ev_spare, ctx->side_stream));
    ctx->spare_zeroing = true;
  }
  if (assemble_system) { ctx->geo_valid = true; ctx->geo_key = geo_key; }
  if (assemble_system) assemble_epilogue(ctx, use_nonzero);
  else {
    IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
    float ms = 0;
    IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    ctx->timing.assemble_kernel_ms = ms;
  }
  hanging_condense_rhs(ctx, use_nonzero);
}

static void assemble_epilogue(ifem_ctx *ctx, int use_nonzero) {
  dinv_setup(ctx);
  bjac_setup(ctx);
  IFEM_HIP_CHECK(hipEventSynchronize(ctx->ev1));
  float ms = 0;
  IFEM_HIP_CHECK(hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
  ctx->timing.assemble_kernel_ms = ms;
  ctx->assembled = true;
  ctx->has_app = false;
  ctx->auu_f32_valid = false;
  ctx->bbt_f32_valid = false;
  // S_m = B diag(M_u)^-1 B^T depends only on the mesh and on WHICH dofs are constrained (not on the solution):
  // keep it across assemblies until the constraint set changes (the reference rebuilds it every solve(); same values)
  {
    const int64_t key = ctx->constraints_epoch * 2 + (use_nonzero ? 1 : 0);
    if (key != ctx->sm_key) { ctx->sm_valid = false; ctx->sm_key = key; }
  }
  ctx->shat_valid = ctx->want_shat;
  ctx->shat_aux_valid = false;
  ctx->asm_constraint_set = use_nonzero ? 1 : 0;
}

} // namespace ifem
