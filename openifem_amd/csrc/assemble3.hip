// assemble3.hip -- InsIM::assemble for the 3D Q2/Q1 element on the FP64 matrix cores (reference:
// source/mpi_insim.cpp:153-362; same mathematics, scatter and constraint handling as assemble2.hip).
//
// The velocity-velocity block of the element matrix is a sum over the 27 quadrature points of outer products,
//   Ke[(a,c),(b,d)] = sum_q  (w gam ga_c)(q,a) gb_d(q,b)  +  (N_a (rho w d_d u_c + d_cd rho/dt w))(q) N_b(q)     (grad-div, Newton + mass)
//                 + d_cd sum_q sum_e (w mu ga_e + rho w u_e N_a)(q,a) gb_e(q,b)                                  (viscous + convective),
// i.e. for every (c,d) a 27x27 GEMM with K = 54 plus a shared 27x27 GEMM with K = 81, run as v_mfma_f64_16x16x4 on 2x2 tiles of
// 16x16 (27 padded to 32, K to 28): 588 MFMAs per cell.
//
// Round 4: ONE wavefront per cell.  On gfx950 the FP64 MFMA runs on the vector pipe: a wave's MFMAs and its other VALU
// instructions add up (tools/overlap_mfma.hip: 4 MFMAs 274 cycles, + 64 v_fma / v_add 600; an MFMA wave and a VALU wave on one
// SIMD take the sum of their times), so the kernel is bound by the INSTRUCTIONS it issues -- 68 cycles per MFMA, 4.5 per VALU
// instruction -- and by the memory-side rate of its atomics, not by latency or occupancy.  Everything here is
// arranged to issue few instructions:
//  * fields at the quadrature points and the local right-hand side by sum factorisation (pencil passes over 27-value arrays
//    in LDS, 1D tables in SGPRs) instead of 27-lane loops over the nodes;
//  * both column tiles of a row tile are integrated together (80 accumulator registers per lane, two waves per SIMD): the
//    row-side operands are built once per point, and a matrix row's 27 blocks leave in one piece;
//  * the scatter stages a matrix row as the IMAGE of its memory: block k of the row (in the order of the blocks' positions,
//    setup.hip::build_scat3) at double s + 9 k, s = alignment of the row's first block inside a 64-byte segment, each double
//    with the byte address "block address - 8 * image index" beside it.  The reading lanes are linear in the image: lane l of
//    round rr adds image[64 rr + l] to base + 8 (64 rr + l) -- two VALU instructions per atomic instruction, instruction
//    boundaries on segment boundaries wherever the row is contiguous (tools/scatter_sim.py: 871 segments per cell);
//  * ranks, positions and alignments come from per-cell records built once per pattern; reference tables from the host.
// Accumulator layout of the instruction (tools/microbench.hip): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15],
// register r of the result is D[(l>>4) + 4r][l&15].
#include <hip/hip_runtime.h>
#include <mutex>
#include <set>
#include "kernels.hpp"
#include "assemble_common.hpp"

namespace ifem {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) double gdouble; // global address space: global_atomic_add_f64, not the flat form

#ifdef IFEM_ASM_PROBES // measurement build (tools/asmbench.py): ifem_tuning::asm_skip drops parts of the kernel, results invalid
#define IFEM_PROBE(x) (x)
#else
#define IFEM_PROBE(x) false
#endif

// per-cell LDS of one wavefront; the scatter stages two matrix rows per sub-step (image of 2 x 256 doubles + their bases)
struct Cell4 {
  static constexpr int NU = 27, NP = 8, ND = 89;
  static constexpr int IMG = 2 * 256;
  static constexpr int ZONE = 2 * IMG;
  double Ji[28 * 9];  // [q][reference direction e][physical direction d]; point 27 = padding of the K dimension, all zero
  double gqs[28 * 9]; // rho JxW grad u + d_cd rho/dt JxW
  double rwu[28 * 3]; // rho JxW u
  double JxW[28];
  double X[24], C[24];
  double fe[96]; // velocity part by component: [c][a], pressure part at 81 + b
  double cv[96]; // inhomogeneities in dof order (a * 3 + c, 81 + b)
  // phase 1: nodal values [9][27] | pencil intermediates; then rhs coefficient fields; B staging (uncached assemblies); scatter image
  double zone[ZONE];
  int64_t rs_uu[27], rs_bt[27], rs_b[8], rs_mp[8];
  int32_t len_uu[27], len_bt[27], len_b[8], len_mp[8];
  int32_t un[27], pn[8];
  int32_t bid[6], ind;
  uint8_t cf[96];
  uint8_t hdr[kAsm3Hdr]; // perm[32] | iperm[32] | permp[8] | srow[32] (setup.hip::build_scat3)
};

// N_a(q) and the reference gradient of N_a at q from the tensor factors; q9 = (q % 9) * 9, q4 = (q / 9) * 4, a9 = a % 9, a3 = a / 9.
// The z factor is zero for q = 27 and a >= 27 (padding of the MFMA tiles): no masks in the contraction.
__device__ __forceinline__ void shape_ref4(const Tabs3 &T, int q9, int q4, int a9, int a3, double &N, double r[3]) {
  const int i2 = q9 + a9, i1 = q4 + a3;
  const double n2 = T.N2[i2], dx2 = T.DX2[i2], dy2 = T.DY2[i2], nz = T.Nz[i1], dz = T.dNz[i1];
  N = n2 * nz;
  r[0] = dx2 * nz; r[1] = dy2 * nz; r[2] = n2 * dz;
}
__device__ __forceinline__ void shape_phys4(const Tabs3 &T, const double *Jq, int q9, int q4, int a9, int a3, double &N, double g[3]) {
  double r[3];
  shape_ref4(T, q9, q4, a9, a3, N, r);
#pragma unroll
  for (int d = 0; d < 3; ++d) g[d] = r[0] * Jq[d] + r[1] * Jq[3 + d] + r[2] * Jq[6 + d];
}

// One tensor direction of narr arrays of 27 values: out[arr][.. o ..] (+)= sum_j M[o][j] in[arr][.. j ..], M = N or N' of the 1D
// element (TRANS: its transpose -- the test-function side).  Item = (array, pencil of three values); the table sits in SGPRs.
template <int DIR, bool TRANS, bool DERIV, bool ACC>
__device__ __forceinline__ void pencil_pass(const Tab1D &t, const double *__restrict__ in, double *__restrict__ out, int narr, int lane) {
  constexpr int st = DIR == 0 ? 1 : (DIR == 1 ? 3 : 9);
  for (int it = lane; it < narr * 9; it += 64) {
    const int arr = it / 9, p = it - arr * 9;
    const int b = DIR == 0 ? 3 * p : (DIR == 1 ? p + 6 * (p / 3) : p);
    const double *src = in + arr * 27 + b;
    double *dst = out + arr * 27 + b;
    const double v0 = src[0], v1 = src[st], v2 = src[2 * st];
#pragma unroll
    for (int o = 0; o < 3; ++o) {
      const double c0 = DERIV ? (TRANS ? t.dN[o] : t.dN[o * 3]) : (TRANS ? t.N[o] : t.N[o * 3]);
      const double c1 = DERIV ? (TRANS ? t.dN[3 + o] : t.dN[o * 3 + 1]) : (TRANS ? t.N[3 + o] : t.N[o * 3 + 1]);
      const double c2 = DERIV ? (TRANS ? t.dN[6 + o] : t.dN[o * 3 + 2]) : (TRANS ? t.N[6 + o] : t.N[o * 3 + 2]);
      double r = c0 * v0 + c1 * v1 + c2 * v2;
      if (ACC) r += dst[o * st];
      dst[o * st] = r;
    }
  }
}

// One row tile (matrix rows 16 TI .. 16 TI + 15) of the velocity-velocity block: the contraction over the 27 points for both column
// tiles, then the scatter of its rows, two per sub-step (see the head of the file).
template <int TI>
__device__ __forceinline__ void row_tile(const AsmArgs &A, const Tabs3 &T, Cell4 &S, const uint4 rc, const int lane, const bool any_c, const bool mass_s) {
  constexpr int NU = 27, BS = 9, DIM = 3;
  const uint8_t *const perm = S.hdr, *const srow = S.hdr + 72;
  const double wgam = A.gamma * A.rho, rdt = A.rho * A.inv_dt;
  double *const imv = S.zone;                                               // image of two matrix rows: values ...
  uint64_t *const imb = reinterpret_cast<uint64_t *>(S.zone + Cell4::IMG);  // ... and "address - 8 * image index" of every double
  const int l15 = lane & 15, g = lane >> 4;
  const int bn0 = perm[l15], bn1 = perm[16 + l15]; // my column nodes (columns in node-id order; 27..31: padding)
  const int b9_0 = bn0 % 9, b3_0 = bn0 / 9, b9_1 = bn1 % 9, b3_1 = bn1 / 9;
  const int al = 16 * TI + l15; // my A-row node (27..31: padding, zero through the z factor)
  const int a9 = al % 9, a3 = al / 9;
  d4 acc0[BS], acc1[BS], sac0 = {0, 0, 0, 0}, sac1 = {0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < BS; ++e) { acc0[e] = d4{0, 0, 0, 0}; acc1[e] = d4{0, 0, 0, 0}; }
  if (!IFEM_PROBE(A.debug_skip == 2)) {
#pragma unroll 1
    for (int ks = 0; ks < 7; ++ks) {
      const int q = 4 * ks + g; // 27: the zero point
      const int q3 = q / 9, q9 = (q - q3 * 9) * 9, q4 = q3 * 4;
      const double *Jq = S.Ji + q * 9;
      const double w = S.JxW[q];
      double Na, Nb0, Nb1, ga[3], gb0[3], gb1[3];
      shape_phys4(T, Jq, q9, q4, a9, a3, Na, ga);
      shape_phys4(T, Jq, q9, q4, b9_0, b3_0, Nb0, gb0);
      shape_phys4(T, Jq, q9, q4, b9_1, b3_1, Nb1, gb1);
      const double wmu = w * A.mu, wg = w * wgam;
      // viscous + convective: sum_e (w mu ga_e + rho w u_e N_a) gb_e
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const double av = S.rwu[q * 3 + e] * Na + wmu * ga[e];
        sac0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, gb0[e], sac0, 0, 0, 0);
        sac1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, gb1[e], sac1, 0, 0, 0);
      }
      // grad-div part of the nine blocks
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const double wga = wg * ga[c];
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          acc0[c * 3 + d] = __builtin_amdgcn_mfma_f64_16x16x4f64(wga, gb0[d], acc0[c * 3 + d], 0, 0, 0);
          acc1[c * 3 + d] = __builtin_amdgcn_mfma_f64_16x16x4f64(wga, gb1[d], acc1[c * 3 + d], 0, 0, 0);
        }
      }
      // Newton term rho N_a N_b d_d u_c with the mass term on its diagonal blocks
      if (!A.imex) {
#pragma unroll
        for (int e = 0; e < BS; ++e) {
          const double an = Na * S.gqs[q * 9 + e];
          acc0[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(an, Nb0, acc0[e], 0, 0, 0);
          acc1[e] = __builtin_amdgcn_mfma_f64_16x16x4f64(an, Nb1, acc1[e], 0, 0, 0);
        }
      }
      if (mass_s) {
        const double am = rdt * w * Na;
        sac0 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, Nb0, sac0, 0, 0, 0);
        sac1 = __builtin_amdgcn_mfma_f64_16x16x4f64(am, Nb1, sac1, 0, 0, 0);
      }
    }
  }
  // ---- scatter: register r of a tile holds the pair (a = 16 TI + (lane>>4) + 4 r, b = column node of the lane)
  // the doubles of a row image no block covers (alignment head, tail) carry null bases: written before the blocks of every sub-step
  const int nrow = lane >= 24 ? 1 : 0, nj = lane - 24 * nrow;
  const int nulli = 256 * nrow + (nj < 8 ? nj : 232 + nj); // lanes 0..47: doubles 0..7 and 240..255 of the two rows
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row0 = 16 * TI + 4 * r;
    if (row0 >= NU) continue; // rows 28..31 do not exist
    const int arow = row0 + g;
    const int ar = arow < NU ? arow : NU - 1;
    const bool rowok = arow < NU && S.len_uu[ar] >= 0 && !IFEM_PROBE(A.debug_skip == 1);
    const uint64_t okm = __ballot(rowok); // bit 16 g: row g of this step exists and is owned
    const int sp = srow[ar];              // position of the row's first block inside its 64-byte segment, in doubles
    const int64_t rs = S.rs_uu[ar];
    const uint64_t rowbase = uint64_t(A.v_uu) + 72ull * uint64_t(rs) - 8ull * unsigned(sp); // address of double 0 of the row's image
    const int64_t row_dof0 = int64_t(DIM) * S.un[ar];
    int i0[2];
    uint64_t base[2];
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const unsigned w32 = tj == 0 ? ((r & 2) ? rc.y : rc.x) : ((r & 2) ? rc.w : rc.z);
      const unsigned sc = (r & 1) ? (w32 >> 16) : (w32 & 0xffffu);
      const unsigned pr = sc & 511u, rank = (sc >> 9) & 31u; // position of my block in the row minus its rank among the cell's 27, rank
      i0[tj] = sp + BS * int(rank);
      base[tj] = rowbase + 72ull * pr;
      if (A.v_s && rowok && 16 * tj + l15 < NU) unsafeAtomicAdd(A.v_s + rs + pr + rank, tj == 0 ? sac0[r] : sac1[r]);
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      if (row0 + 2 * ss >= NU) continue;
      if (lane < 48) imb[nulli] = 0ull;
      const bool writer = rowok && (g >> 1) == ss;
      const int rowoff = (g & 1) * 256;
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        if (!writer || 16 * tj + l15 >= NU) continue;
        const double s = tj == 0 ? sac0[r] : sac1[r];
        const int at = rowoff + i0[tj];
#pragma unroll
        for (int e = 0; e < BS; ++e) {
          imv[at + e] = (tj == 0 ? acc0[e][r] : acc1[e][r]) + ((e == 0 || e == 4 || e == 8) ? s : 0.0);
          imb[at + e] = base[tj];
        }
        if (any_c) { // rows / columns of constrained dofs: patch my block in the image
          const int b = tj == 0 ? bn0 : bn1;
          const unsigned rm = unsigned(S.cf[ar * 3]) | unsigned(S.cf[ar * 3 + 1]) << 1 | unsigned(S.cf[ar * 3 + 2]) << 2;
          const unsigned cm = unsigned(S.cf[b * 3]) | unsigned(S.cf[b * 3 + 1]) << 1 | unsigned(S.cf[b * 3 + 2]) << 2;
          if (rm | cm) {
#pragma unroll 1
            for (int e = 0; e < BS; ++e) {
              const int c = e / 3, d = e - 3 * c;
              const bool rcn = (rm >> c) & 1u, ccn = (cm >> d) & 1u;
              if (!rcn && !ccn) continue;
              const double v = imv[at + e];
              double w = 0.0;
              if (rcn) {
                if (ar == b && c == d) { // |Ke(r,r)| on the diagonal, rhs so that the update equals the inhomogeneity
                  w = fabs(v);
                  if (A.use_inhom) unsafeAtomicAdd(&A.rhs[row_dof0 + c], S.cv[ar * 3 + c] * fabs(v));
                }
              } else if (A.use_inhom) {
                const double gi = S.cv[b * 3 + d];
                if (gi != 0.0) unsafeAtomicAdd(&S.fe[c * NU + ar], -v * gi);
              }
              imv[at + e] = w;
              imb[at + e] = w != 0.0 ? base[tj] : 0ull;
            }
          }
        }
      }
      wsync2();
      // lane l of round rr owns double 64 rr + l of the image: of row rr / 4, at 8 (64 (rr % 4) + l) bytes behind its base
      {
        uint64_t bs[8];
        double wv[8];
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const bool on = row0 + 2 * ss + rr / 4 < NU && ((okm >> (16 * (2 * ss + rr / 4))) & 1ull); // compile-time && wave-uniform: a row of this rank
          bs[rr] = on ? imb[64 * rr + lane] : 0ull;
          wv[rr] = on ? imv[64 * rr + lane] : 0.0;
        }
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          if (row0 + 2 * ss + rr / 4 >= NU) continue;
          if (bs[rr] != 0ull) {
            gdouble *dst = reinterpret_cast<gdouble *>(bs[rr] + 8ull * unsigned(lane)) + 64 * (rr & 3);
            if (IFEM_PROBE(A.debug_skip == 6)) *dst = wv[rr]; // rate of plain stores in place of the atomics (results invalid)
            else __builtin_amdgcn_global_atomic_fadd_f64(dst, wv[rr]);
          }
        }
      }
      wsync2();
    }
  }
}

// CPB cells per workgroup, one wavefront each; the workgroup shares the reference tables only.
template <int CPB>
__global__ __launch_bounds__(64 * CPB) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_ins_assemble3(AsmArgs A, Tab1D t1) {
  constexpr int DIM = 3, NU = 27, NP = 8, NQ = 27, ND = 89;
  constexpr int NBP = NU * NP, BROUNDS = (NBP + 63) / 64;
  using Cell = Cell4;
  extern __shared__ __align__(16) unsigned char smem[];
  Tabs3 &T = *reinterpret_cast<Tabs3 *>(smem);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  Cell &S = *reinterpret_cast<Cell *>(smem + ((sizeof(Tabs3) + 15) & ~size_t(15)) + size_t(wave) * ((sizeof(Cell) + 15) & ~size_t(15)));
  {
    const double *src = reinterpret_cast<const double *>(A.tabs3);
    double *dst = reinterpret_cast<double *>(&T);
    for (int i = threadIdx.x; i < int(sizeof(Tabs3) / 8); i += 64 * CPB) dst[i] = src[i];
  }
  __syncthreads(); // the only workgroup barrier: from here on every wave works on its own cell
  const int64_t idx = int64_t(A.xcd_swizzle ? xcd_swizzle(blockIdx.x, gridDim.x) : blockIdx.x) * CPB + wave;
  if (idx >= A.count) return;
  const int64_t cc = A.order ? int64_t(A.order[A.first + idx]) : idx;
  const int64_t p_off = int64_t(DIM) * A.nUl;
  double *const zone = S.zone;
  const uint8_t *const perm = S.hdr, *const iperm = S.hdr + 32, *const permp = S.hdr + 64;

  // ---- phase 0: ids, coordinates, nodal values, row descriptors, constraint flags, scatter records.  Lanes 0..26: velocity
  // nodes, 32..39: pressure nodes, 40..63: vertex coordinates.  Two dependent rounds of loads (ids, then everything keyed by an
  // id), all issued before the first LDS store; optional arrays fall back to a valid address + select.
  const bool isU = lane < NU, isP = lane >= 32 && lane < 32 + NP, isX = lane >= 40;
  const int nf = (A.fsi_acc && A.indicator) ? 9 : 6; // nodal fields carried through phase 1: u, u0 (+ a_fsi)
  const uint4 *recp = reinterpret_cast<const uint4 *>(A.scat3 + cc * kAsm3Rec);
  {
    const int32_t nd = isP ? A.cell_pnodes[cc * NP + (lane - 32)] : A.cell_unodes[cc * NU + (isU ? lane : 0)];
    const double xv = A.vcoords[cc * NP * DIM + (isX ? lane - 40 : 0)];
    int32_t bidv = -1;
    if (A.n_neumann != 0) bidv = A.cell_face_bid[cc * 2 * DIM + (lane < 2 * DIM ? lane : 0)];
    const int32_t indv = A.indicator ? A.indicator[cc] : 0;
    const uint32_t hdrw = reinterpret_cast<const uint32_t *>(A.hdr3 + cc * kAsm3Hdr)[lane & 31];
    const bool own = nd < (isP ? A.nPo : A.nUo);
    const int64_t ndr = own ? nd : 0;
    const int64_t *rpa = isP ? A.rp_b : A.rp_uu, *rpb = isP ? A.rp_mp : A.rp_bt;
    const int64_t r0 = rpa[ndr], r1 = rpa[ndr + 1], t0 = rpb[ndr], t1_ = rpb[ndr + 1];
    const int64_t dof = isP ? p_off + nd : int64_t(DIM) * nd;
    const double *fa = A.fsi_acc ? A.fsi_acc : A.eval, *cvp = A.cval ? A.cval : A.eval;
    const uint8_t *icp = A.is_c ? A.is_c : reinterpret_cast<const uint8_t *>(A.eval);
    double ev[DIM], pv[DIM], av[DIM], cvv[DIM];
    uint8_t cfv[DIM];
#pragma unroll
    for (int c = 0; c < DIM; ++c) {
      const int64_t k = dof + (isP ? 0 : c);
      ev[c] = A.eval[k]; pv[c] = A.present[k]; av[c] = fa[k]; cvv[c] = cvp[k]; cfv[c] = icp[k];
    }
    if (isU) {
      const int a = lane;
      S.un[a] = nd;
      S.rs_uu[a] = own ? r0 : 0; S.len_uu[a] = own ? int32_t(r1 - r0) : -1;
      S.rs_bt[a] = own ? t0 : 0; S.len_bt[a] = own ? int32_t(t1_ - t0) : -1;
#pragma unroll
      for (int c = 0; c < DIM; ++c) {
        zone[c * 27 + a] = ev[c];
        zone[(3 + c) * 27 + a] = pv[c];
        if (nf == 9) zone[(6 + c) * 27 + a] = av[c];
        S.cf[a * DIM + c] = A.is_c ? cfv[c] : uint8_t(0);
        S.cv[a * DIM + c] = A.cval ? cvv[c] : 0.0;
      }
    }
    if (isP) {
      const int b = lane - 32;
      S.pn[b] = nd;
      S.rs_b[b] = own ? r0 : 0; S.len_b[b] = own ? int32_t(r1 - r0) : -1;
      S.rs_mp[b] = own ? t0 : 0; S.len_mp[b] = own ? int32_t(t1_ - t0) : -1;
      zone[972 + b] = ev[0];
      S.cf[NU * DIM + b] = A.is_c ? cfv[0] : uint8_t(0);
      S.cv[NU * DIM + b] = A.cval ? cvv[0] : 0.0;
    }
    if (isX) S.X[lane - 40] = xv;
    if (lane < 2 * DIM) S.bid[lane] = bidv;
    if (lane == 0) S.ind = indv;
    if (lane < 32) reinterpret_cast<uint32_t *>(S.hdr)[lane] = hdrw;
    for (int i = lane; i < 96; i += 64) S.fe[i] = 0.0;
  }
  uint4 rec0 = {0, 0, 0, 0}, rec1 = {0, 0, 0, 0}; // [row tile][lane]: (position | rank << 9) of my pairs [column tile][r]
  if (!A.rhs_only && !A.skip_uu) { rec0 = recp[lane]; rec1 = recp[64 + lane]; }
  wsync2();
  const int ind = S.ind;
  if (lane < NP * DIM) { // monomial coefficients of the trilinear map
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
  // ---- phase 1: the nodal fields at the 27 points by sum factorisation: values of u, u0 (a_fsi) and the reference gradient of u.
  // zone: nodal [nf][27] | X pass [nf + 3][27] | Y pass [nf + 6][27] at 567; the Z pass overwrites the first two
  {
    double *const nodal = zone, *const PX = zone + 243, *const PY = zone + 567, *const PZ = zone;
    pencil_pass<0, false, false, false>(t1, nodal, PX, nf, lane);         // N_x of every field
    pencil_pass<0, false, true, false>(t1, nodal, PX + nf * 27, 3, lane); // N'_x of u
    wsync2();
    pencil_pass<1, false, false, false>(t1, PX, PY, nf + 3, lane);          // N_y of all of them
    pencil_pass<1, false, true, false>(t1, PX, PY + (nf + 3) * 27, 3, lane); // N'_y of (N_x u)
    wsync2();
    pencil_pass<2, false, false, false>(t1, PY, PZ, nf + 6, lane);          // values [0, nf), d/dxi_0 u at nf, d/dxi_1 u at nf + 3
    pencil_pass<2, false, true, false>(t1, PY, PZ + (nf + 6) * 27, 3, lane); // d/dxi_2 u at nf + 6
    wsync2();
  }
  // ---- per quadrature point (lane = q): Jacobian, physical gradient, coefficients of the matrix and of the right-hand side
  // rhs coefficient fields for the transposed passes at zone + 567: [S | V0 | V1 | V2][c][27], div term at zone + 891
  double *const fld = zone + 567, *const divw_ = zone + 891;
  // the mass term rho/dt N_a N_b rides on the Newton term's diagonal blocks; without a Newton term (IMEX), or when the scalar
  // part is wanted by itself (v_s), it is one more product of the scalar part
  const bool mass_s = A.imex || A.v_s != nullptr;
  if (lane < NQ) {
    const int q = lane;
    const int qi[3] = {q % 3, (q / 3) % 3, q / 9};
    double xi[3], wq = 1.0;
#pragma unroll
    for (int d = 0; d < DIM; ++d) { xi[d] = t1.xi[0] * (qi[d] == 0) + t1.xi[1] * (qi[d] == 1) + t1.xi[2] * (qi[d] == 2); wq *= t1.w[0] * (qi[d] == 0) + t1.w[1] * (qi[d] == 1) + t1.w[2] * (qi[d] == 2); }
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
    for (int c = 0; c < 3; ++c) {
      u[c] = zone[c * 27 + q]; u0[c] = zone[(3 + c) * 27 + q]; ac[c] = nf == 9 ? zone[(6 + c) * 27 + q] : 0.0;
#pragma unroll
      for (int e = 0; e < 3; ++e) gr[c * 3 + e] = zone[(nf + 3 * e + c) * 27 + q];
    }
#pragma unroll
    for (int b = 0; b < NP; ++b) p += T.psi[q * NP + b] * zone[972 + b];
    double g[9];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int d = 0; d < 3; ++d) g[c * 3 + d] = gr[c * 3] * Ji[d] + gr[c * 3 + 1] * Ji[3 + d] + gr[c * 3 + 2] * Ji[6 + d];
    const double dv = g[0] + g[4] + g[8];
    const double rdtw = A.rho * A.inv_dt * w;
    S.JxW[q] = w;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      S.Ji[q * 9 + i] = Ji[i];
      S.gqs[q * 9 + i] = (A.imex ? 0.0 : A.rho * w * g[i]) + (((i == 0 || i == 4 || i == 8) && !mass_s) ? rdtw : 0.0);
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      S.rwu[q * 3 + c] = A.imex ? 0.0 : A.rho * w * u[c]; // only the matrix reads it (u . grad N_b): no convection in the IMEX matrix
      double adv = 0, vc[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        adv += g[c * 3 + d] * u[d];
        vc[d] = w * (-A.mu * g[c * 3 + d] + (c == d ? p - A.gamma * A.rho * dv : 0.0));
      }
#pragma unroll
      for (int e = 0; e < 3; ++e) fld[((1 + e) * 3 + c) * 27 + q] = Ji[e * 3] * vc[0] + Ji[e * 3 + 1] * vc[1] + Ji[e * 3 + 2] * vc[2]; // reference-gradient basis
      double sc = -A.rho * adv - A.rho * A.inv_dt * (u[c] - u0[c]) + A.rho * A.g[c];
      if (ind == 1) sc += A.rho * ac[c];
      fld[c * 27 + q] = w * sc;
    }
    divw_[q] = w * dv;
  } else if (lane == NQ) { // the padding point of the K dimension
    S.JxW[NQ] = 0.0;
#pragma unroll
    for (int i = 0; i < 9; ++i) { S.Ji[NQ * 9 + i] = 0.0; S.gqs[NQ * 9 + i] = 0.0; }
#pragma unroll
    for (int c = 0; c < 3; ++c) S.rwu[NQ * 3 + c] = 0.0;
  }
  wsync2();
  // ---- Neumann (pressure) boundary faces  (:313-341)
  if (A.n_neumann != 0) {
    for (int f = 0; f < 2 * DIM; ++f) {
      const int bid = S.bid[f];
      if (bid < 0) continue;
      double pbc = 0; bool hit = false;
      for (int k = 0; k < A.n_neumann; ++k) if (A.neumann_id[k] == bid) { pbc = A.neumann_p[k]; hit = true; }
      if (!hit) continue;
      const int nd = f >> 1; const double sgn = (f & 1) ? 1.0 : -1.0;
      for (int i = lane; i < NU * DIM; i += 64) {
        const int c = i / NU, a = i - c * NU;
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
        S.fe[i] -= acc;
      }
    }
    wsync2();
  }
  // ---- local rhs (:281-304): the transposed passes, fe[c][a] += sum_q N_a S_c + d_e N_a V_ce
  {
    double *const ZA = zone, *const YB = zone + 243;
    pencil_pass<2, true, false, false>(t1, fld, ZA, 9, lane);         // N_z^T of S, V0, V1
    wsync2();
    pencil_pass<2, true, true, true>(t1, fld + 9 * 27, ZA, 3, lane);  // + N'_z^T V2 onto the S part
    wsync2();
    pencil_pass<1, true, false, false>(t1, ZA, YB, 6, lane);          // N_y^T of (S + V2) and V0
    wsync2();
    pencil_pass<1, true, true, true>(t1, ZA + 6 * 27, YB, 3, lane);   // + N'_y^T V1
    wsync2();
    pencil_pass<0, true, false, true>(t1, YB, S.fe, 3, lane);         // N_x^T
    wsync2();
    pencil_pass<0, true, true, true>(t1, YB + 3 * 27, S.fe, 3, lane); // + N'_x^T V0
    { // pressure rows: sum_q (w div u) psi_b
      const int b = lane & 7, k = lane >> 3;
      double f = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = k + 8 * j;
        if (q < NQ) f += divw_[q] * T.psi[q * NP + b];
      }
      f += __shfl_xor(f, 8); f += __shfl_xor(f, 16); f += __shfl_xor(f, 32);
      if (lane < NP) S.fe[NU * DIM + lane] += f;
    }
    wsync2();
  }

  // ---- velocity-pressure blocks: -JxW psi_b grad N_a
  bool need_b = !A.rhs_only && !IFEM_PROBE(A.debug_skip >= 3);
  if (need_b && A.skip_geo) { // cached blocks: only a cell with an inhomogeneous constrained dof still needs the entries
    const bool mine = (lane < ND && S.cf[lane] && S.cv[lane] != 0.0) || (lane + 64 < ND && S.cf[lane + 64] && S.cv[lane + 64] != 0.0);
    need_b = A.use_inhom && __any(mine);
  }
  // Lanes = (velocity node a, pressure node in id order): the eight entries of a B^T row land next to each other.  The B entries go
  // through LDS (the idle zone) into the order (pressure node, velocity nodes by id) = the order of B's rows, so that
  // neighbouring lanes of its atomics hit neighbouring entries too
  if (need_b) {
    double *const bst = zone;
#pragma unroll 1
    for (int k = 0; k < BROUNDS; ++k) {
      const int t = lane + 64 * k;
      if (t >= NBP) continue;
      const int a = t / NP, pb = permp[t - a * NP];
      const int a9 = a % 9, a3 = a / 9;
      double v[3] = {0, 0, 0};
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) {
        const double wpsi = S.JxW[q] * T.psi[q * NP + pb];
        double N, g[3];
        shape_phys4(T, S.Ji + q * 9, (q % 9) * 9, (q / 9) * 4, a9, a3, N, g);
        v[0] -= wpsi * g[0]; v[1] -= wpsi * g[1]; v[2] -= wpsi * g[2];
      }
      const bool pc = S.cf[NU * DIM + pb];
      if (S.len_bt[a] >= 0) {
        const int len = S.len_bt[a];
        double *base = A.v_bt + S.rs_bt[a] * DIM + A.posUP[(cc * NU + a) * NP + pb];
#pragma unroll
        for (int c = 0; c < DIM; ++c) {
          if (S.cf[a * DIM + c]) continue;
          if (!pc) { if (!A.skip_geo) unsafeAtomicAdd(base + int64_t(c) * len, v[c]); }
          else if (A.use_inhom && S.cv[NU * DIM + pb] != 0.0) unsafeAtomicAdd(&S.fe[c * NU + a], -v[c] * S.cv[NU * DIM + pb]);
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
        if (!A.skip_geo) bst[(pb * NU + iperm[a]) * DIM + c] = w;
      }
    }
    if (!A.skip_geo) { // B in row order: lane = (pressure node, velocity node by id), one plane per instruction
      wsync2();
#pragma unroll 1
      for (int k = 0; k < BROUNDS; ++k) {
        const int t = lane + 64 * k;
        if (t >= NBP) continue;
        const int pb = t / NU, a = perm[t - pb * NU];
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
  if (!A.rhs_only && !A.skip_geo && !IFEM_PROBE(A.debug_skip >= 4)) {
    {
      const int pa = lane / NP, pb = permp[lane - pa * NP];
      double m = 0;
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) m += S.JxW[q] * T.psi[q * NP + pa] * T.psi[q * NP + pb];
      if (S.len_mp[pa] >= 0) {
        const bool ra = S.cf[NU * DIM + pa], cb = S.cf[NU * DIM + pb];
        double *dst = A.v_mp + S.rs_mp[pa] + A.posPP[(cc * NP + pa) * NP + pb];
        if (!ra && !cb) unsafeAtomicAdd(dst, m);
        else if (ra && pa == pb) unsafeAtomicAdd(dst, fabs(m));
      }
    }
    if (lane < NU) {
      const int a9 = lane % 9, a3 = lane / 9;
      double m = 0;
#pragma unroll 3
      for (int q = 0; q < NQ; ++q) {
        double N, r[3];
        shape_ref4(T, (q % 9) * 9, (q / 9) * 4, a9, a3, N, r);
        m += S.JxW[q] * N * N;
      }
      if (S.len_uu[lane] >= 0)
        for (int c = 0; c < DIM; ++c) unsafeAtomicAdd(&A.diagMu[int64_t(DIM) * S.un[lane] + c], m);
    }
  }
  wsync2(); // rhs, B / B^T and M_p are integrated: the zone may be reused, S.fe is complete up to the scatter corrections
  // most cells carry no constrained dof: a wave-uniform flag lets their scatter skip the per-entry constraint logic
  bool any_c;
  {
    bool mine = false;
    for (int i = lane; i < ND; i += 64) mine = mine || S.cf[i];
    any_c = __any(mine);
  }
  // ---- velocity-velocity block on the matrix cores, one row tile (16 matrix rows) after the other
  if (!A.rhs_only && !A.skip_uu && !IFEM_PROBE(A.debug_skip == 5)) {
    row_tile<0>(A, T, S, rec0, lane, any_c, mass_s);
    row_tile<1>(A, T, S, rec1, lane, any_c, mass_s);
  }
  wsync2();
  // ---- rhs scatter (unconstrained owned rows; constrained rows were handled with the diagonal)
  for (int i = lane; i < ND; i += 64) {
    if (i < NU * DIM) {
      const int c = i / NU, a = i - c * NU;
      if (!S.cf[a * DIM + c] && S.len_uu[a] >= 0) unsafeAtomicAdd(&A.rhs[int64_t(DIM) * S.un[a] + c], S.fe[i]);
    } else {
      const int b = i - NU * DIM;
      if (!S.cf[i] && S.len_b[b] >= 0) unsafeAtomicAdd(&A.rhs[int64_t(DIM) * A.nUo + S.pn[b]], S.fe[i]);
    }
  }
}

// reference tables of the Q2/Q1 hexahedron at the 27 Gauss points (built once per process, uploaded once per device)
static void build_tabs3(Tabs3 &T, const Tab1D &t) {
  std::memset(&T, 0, sizeof(T));
  for (int q = 0; q < 3; ++q)
    for (int a = 0; a < 3; ++a) { T.Nz[q * 4 + a] = t.N[q * 3 + a]; T.dNz[q * 4 + a] = t.dN[q * 3 + a]; }
  for (int i = 0; i < 81; ++i) {
    const int q01 = i / 9, a01 = i - q01 * 9;
    const int ix = (q01 % 3) * 3 + (a01 % 3), iy = (q01 / 3) * 3 + (a01 / 3);
    T.N2[i] = t.N[ix] * t.N[iy]; T.DX2[i] = t.dN[ix] * t.N[iy]; T.DY2[i] = t.N[ix] * t.dN[iy];
  }
  for (int i = 0; i < 27 * 8; ++i) {
    const int q = i / 8, b = i - q * 8;
    double v = 1;
    for (int d = 0; d < 3; ++d) {
      const int qd = d == 0 ? q % 3 : (d == 1 ? (q / 3) % 3 : q / 9);
      const double x = t.xi[qd];
      v *= ((b >> d) & 1) ? x : 1.0 - x;
    }
    T.psi[i] = v;
  }
}

template <int CPB>
static void launch3(ifem_ctx *ctx, const AsmArgs &A) {
  const size_t smem = ((sizeof(Tabs3) + 15) & ~size_t(15)) + CPB * ((sizeof(Cell4) + 15) & ~size_t(15));
  // the dynamic-LDS limit is an attribute of the function ON A DEVICE: remembered per device, not per process
  static std::mutex mu;
  static std::set<int> done;
  {
    std::lock_guard<std::mutex> lk(mu);
    if (!done.count(ctx->device)) {
      IFEM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_ins_assemble3<CPB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
      done.insert(ctx->device);
    }
  }
  Tab1D t;
  tab1d(t, 2);
  if (ctx->tabs3.n == 0) {
    Tabs3 T;
    build_tabs3(T, t);
    ctx->tabs3.upload(&T, 1, ctx->stream);
    IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream)); // T is a stack object
  }
  AsmArgs B = A;
  B.order = nullptr; B.first = 0; B.count = A.n_cells;
  B.tabs3 = ctx->tabs3.p; B.scat3 = ctx->scat3.p; B.hdr3 = ctx->hdr3.p;
  const int64_t nblk = (B.count + CPB - 1) / CPB;
  hipLaunchKernelGGL((k_ins_assemble3<CPB>), dim3((unsigned)nblk), dim3(64 * CPB), smem, ctx->stream, B, t);
  IFEM_HIP_CHECK(hipGetLastError());
}

// 3D Q2/Q1 only; the block-interleaved A_uu layout is assumed by the staged scatter, rows of at most 511 blocks by its records
bool launch_ins_assemble3_kernel(ifem_ctx *ctx, const AsmArgs &A) {
#if !IFEM_UU_INTERLEAVED
  return false;
#else
  if (ctx->dim != 3 || ctx->kv != 2 || ctx->tune.asm3_variant == 1) return false;
  if (!ensure_scat3(ctx, !A.skip_uu && !A.rhs_only)) return false;
  switch (ctx->tune.asm3_cpb) { // (one cell per workgroup spilled 2 VGPRs and was never the fastest: not built since round 5)
  case 4: launch3<4>(ctx, A); break;
  case 8: launch3<8>(ctx, A); break;
  default: launch3<2>(ctx, A);
  }
  return true;
#endif
}

} // namespace ifem
