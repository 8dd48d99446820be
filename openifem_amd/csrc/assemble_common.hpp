// assemble_common.hpp -- argument block and small helpers shared by the InsIM assembly kernels (assemble.hip: first
// version with the physical-gradient table in LDS; assemble2.hip: quadrature-point-outer version with register
// accumulators).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include "ctx.hpp"

namespace ifem {

template <int DIM, int KV>
struct Geo {
  static constexpr int N1 = KV + 1;
  static constexpr int NU = (DIM == 2) ? N1 * N1 : N1 * N1 * N1;
  static constexpr int NP = (DIM == 2) ? 4 : 8;
  static constexpr int NQ = NU;
  static constexpr int ND = NU * DIM + NP;
};

struct AsmArgs {
  int64_t n_cells;
  const int32_t *order; // cells of the colour being assembled (nullptr: all cells, atomics)
  int64_t first, count; // range of `order` handled by this launch
  int64_t nUo, nUl, nPo;
  const FeTables *fe;
  const double *vcoords;
  const int32_t *cell_unodes, *cell_pnodes, *cell_face_bid, *indicator;
  const uint16_t *posUU, *posUP, *posPU, *posPP;
  const uint16_t *scat3; const uint8_t *hdr3; const Tabs3 *tabs3; // 3D Q2/Q1 cell kernel only (assemble3.hip)
  const int64_t *rp_uu, *rp_bt, *rp_b, *rp_mp;
  double *v_uu, *v_bt, *v_b, *v_mp, *diagMu, *rhs;
  double *v_s; // scalar velocity operator (one value per A_uu block) or nullptr
  const uint8_t *is_c;
  const double *cval;
  const double *eval, *present, *fsi_acc;
  double mu, rho, gamma, inv_dt;
  double g[3];
  int n_neumann;
  int neumann_id[8];
  double neumann_p[8];
  int use_inhom; // constraint set carries non-zero inhomogeneities
  int imex;     // InsIMEX (mpi_insimex.cpp:248-262): the matrix drops the two convective terms (explicit convection)
  int rhs_only; // InsIMEX with assemble_system = false: only the right-hand side is integrated and scattered (:343-346)
  int xcd_swizzle; // contiguous cell ranges per XCD (IFEM_XCD=0 switches it off)
  int skip_geo;   // B, B^T, M_p and diag(M_u) of the previous assembly are still valid (same mesh, same constraint set):
                  // integrate them only where a constrained dof needs their entries for the right-hand side
  int skip_uu;    // geometry-only assembly (multigrid levels): the velocity-velocity block is neither integrated nor scattered
  int debug_skip; // measurement builds only (-DIFEM_ASM_PROBES, ifem_tuning::asm_skip): 1 = no A_uu scatter, 2 = no contraction either
};

template <int DIM, typename R = double>
__device__ inline R inv_small(const R *J, R *Ji) {
  if constexpr (DIM == 2) {
    const R det = J[0] * J[3] - J[1] * J[2];
    const R r = R(1) / det;
    Ji[0] = J[3] * r; Ji[1] = -J[1] * r; Ji[2] = -J[2] * r; Ji[3] = J[0] * r;
    return det;
  } else {
    const R c00 = J[4] * J[8] - J[5] * J[7], c01 = J[5] * J[6] - J[3] * J[8], c02 = J[3] * J[7] - J[4] * J[6];
    const R det = J[0] * c00 + J[1] * c01 + J[2] * c02;
    const R r = R(1) / det;
    Ji[0] = c00 * r; Ji[3] = c01 * r; Ji[6] = c02 * r;
    Ji[1] = (J[2] * J[7] - J[1] * J[8]) * r; Ji[4] = (J[0] * J[8] - J[2] * J[6]) * r; Ji[7] = (J[1] * J[6] - J[0] * J[7]) * r;
    Ji[2] = (J[1] * J[5] - J[2] * J[4]) * r; Ji[5] = (J[2] * J[3] - J[0] * J[5]) * r; Ji[8] = (J[0] * J[4] - J[1] * J[3]) * r;
    return det;
  }
}

// scatter add into global memory: hardware atomic, or plain read-modify-write when the launch covers one colour
template <bool ATOMIC>
__device__ inline void gadd(double *p, double v) {
  if constexpr (ATOMIC) unsafeAtomicAdd(p, v);
  else *p += v;
}

// 1D tensor factors of the Q_kv shape functions at the Gauss points (assemble2.hip, assemble3.hip)
struct Tab1D {
  double N[9];  // [q][i] 1D Lagrange shape i (equidistant nodes) at Gauss point q
  double dN[9]; // its derivative
  double xi[3], w[3];
};

__device__ inline void wsync2() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

inline void tab1d(Tab1D &t, int kv) {
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
      const double xi_i = double(i) / kv;
      double v = 1, d = 0;
      for (int j = 0; j < n1; ++j)
        if (j != i) v *= (t.xi[q] - double(j) / kv) / (xi_i - double(j) / kv);
      for (int k = 0; k < n1; ++k) {
        if (k == i) continue;
        double p = 1.0 / (xi_i - double(k) / kv);
        for (int j = 0; j < n1; ++j)
          if (j != i && j != k) p *= (t.xi[q] - double(j) / kv) / (xi_i - double(j) / kv);
        d += p;
      }
      t.N[q * n1 + i] = v;
      t.dN[q * n1 + i] = d;
    }
}

} // namespace ifem
