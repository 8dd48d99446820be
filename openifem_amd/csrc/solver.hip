// solver.hip -- host drivers of the Krylov path; every vector/matrix operation is a HIP kernel on ctx->stream.
//   fgmres()            deal.II SolverFGMRES as used by InsIM::solve            (mpi_insim.cpp:379-388)
//   cg()                PETSc KSPCG + PreconditionNone                          (mpi_insim.cpp:73-82,88-108)
//   precond_vmult()     InsIM::BlockSchurPreconditioner::vmult                  (mpi_insim.cpp:57-128)
//   ins_solve()         InsIM::solve                                            (mpi_insim.cpp:365-395)
//   ins_newton_step()   Newton loop of InsIM::run_one_step                      (mpi_insim.cpp:416-473)
// A~^-1 (MUMPS in the reference) is an inner right-preconditioned GMRES(m) on the BSR A_uu with node-block
// Jacobi; the outer solver is *flexible* GMRES precisely so that such an inexact inner solve is admissible.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <vector>
#include "ctx.hpp"
#include "kernels.hpp"

namespace ifem {

using OpFn = std::function<void(const double *, double *)>;
using DotFn = std::function<void(int, const double *, const double *, double *)>; // out[i] = <V_i, w>, all-reduced

struct Clock {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double ms() const { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

// (flexible) right-preconditioned restarted GMRES, x0 = 0.  V: (m+1) x n, Z: m x n (flexible) or 1 x n.
// Orthogonalisation: classical Gram-Schmidt with one re-orthogonalisation pass (two fused multi-dot /
// multi-axpy sweeps instead of deal.II's j+1 sequential dots; same Krylov space, fewer host round trips).
static int gmres(ifem_ctx *ctx, int64_t n, int64_t ld, bool reorth, const OpFn &A, const OpFn &Pinv, bool flexible,
                 const double *b, double *x, int m, int maxit, double tol, double *V, double *Z, double *w,
                 double *res_out,
                 const std::function<void(int, const double *, int64_t, const double *, double *)> &mdot,
                 std::vector<double> *history = nullptr, // residual norm after every iteration (verbose runs)
                 const std::function<void(int, double *&, double *&)> *ensure = nullptr, // bases that grow with the iteration count: called
                                                                                        // with the V columns the next iteration needs
                 bool left = false, // LEFT preconditioning (not flexible): the Krylov space of P^-1 A, the stopping test reads the
                                      // PRECONDITIONED residual -- deal.II's SolverGMRES with its defaults (mpi_supg_solver.cpp:176-182)
                 double *stage = nullptr,    // left only: every new basis vector is also written here and the operators read IT, so that
                 const OpFn *PA = nullptr) { // ... the fused w = P^-1 A stage (a sequence with constant kernel arguments: a hipGraph) can replace the pair
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m), h(m + 1), h2(m + 1);
  v_zero(ctx, n, x);
  int it = 0;
  double res = 0;
  bool first = true;
  while (true) {
    const double *r0 = w; // the residual the cycle starts from: b itself in the first cycle (x = 0), b - A x after a restart
    if (first) { r0 = b; first = false; }
    else { A(x, w); v_axpby(ctx, n, 1.0, b, -1.0, w); }
    if (left) { Pinv(r0, Z); r0 = Z; }
    double bb;
    mdot(1, r0, n, r0, &bb);
    const double beta = std::sqrt(bb);
    res = beta;
    if (res <= tol || it >= maxit || !std::isfinite(res)) break; // (deal.II's SolverControl::check fails on a NaN as well)
    if (left && stage) v_scale_to2(ctx, n, 1.0 / beta, r0, V, stage);
    else v_scale_to(ctx, n, 1.0 / beta, r0, V);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    int j = 0;
    bool done = false;
    for (; j < m && it < maxit; ++j) {
      if (ensure) (*ensure)(j + 2, V, Z); // columns 0 .. j + 1 of V, 0 .. j of Z
      double *vj = V + (int64_t)j * ld;
      double *zj = flexible ? Z + (int64_t)j * ld : Z;
      if (left && stage && PA) (*PA)(stage, w);
      else if (left) { A(stage ? stage : vj, zj); Pinv(zj, w); }
      else { Pinv(vj, zj); A(zj, w); }
      // classical Gram-Schmidt (twice with reorth); ||w||^2 comes out of the last multi-axpy pass (summed over the ranks like the
      // dot products of `mdot`): per iteration the host waits for the device once per pass, not three / five times
      double ww = 0;
      mdot(j + 1, V, ld, w, h.data());
      if (reorth) {
        v_maxpy(ctx, n, j + 1, V, ld, h.data(), w);
        mdot(j + 1, V, ld, w, h2.data());
        v_maxpy(ctx, n, j + 1, V, ld, h2.data(), w, &ww, true);
      } else {
        v_maxpy(ctx, n, j + 1, V, ld, h.data(), w, &ww, true);
        std::fill(h2.begin(), h2.end(), 0.0);
      }
      for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = h[i] + h2[i];
      const double hn = std::sqrt(ww);
      H[(size_t)(j + 1) * m + j] = hn;
      if (hn > 0) {
        if (left && stage) v_scale_to2(ctx, n, 1.0 / hn, w, V + (int64_t)(j + 1) * ld, stage);
        else v_scale_to(ctx, n, 1.0 / hn, w, V + (int64_t)(j + 1) * ld);
      }
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = t;
      }
      const double a = H[(size_t)j * m + j], c = H[(size_t)(j + 1) * m + j], r = std::hypot(a, c);
      cs[j] = a / r; sn[j] = c / r;
      H[(size_t)j * m + j] = r; H[(size_t)(j + 1) * m + j] = 0;
      g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      ++it;
      if (history) history->push_back(res);
      if (res <= tol || hn == 0 || !std::isfinite(res)) { ++j; done = true; break; }
    }
    for (int i = j - 1; i >= 0; --i) {
      double t = g[i];
      for (int k = i + 1; k < j; ++k) t -= H[(size_t)i * m + k] * y[k];
      y[i] = t / H[(size_t)i * m + i];
    }
    if (flexible || left) {
      for (int i = 0; i < j; ++i) h[i] = -y[i];
      v_maxpy(ctx, n, j, left ? V : Z, ld, h.data(), x); // x += sum y_i z_i   (left preconditioning: x += V y)
    } else {
      // x += P^-1 (V y): one preconditioner application instead of storing every z_j
      v_zero(ctx, n, w);
      for (int i = 0; i < j; ++i) h[i] = -y[i];
      v_maxpy(ctx, n, j, V, ld, h.data(), w);
      Pinv(w, Z);
      v_axpy(ctx, n, 1.0, Z, x);
    }
    if (done || it >= maxit) break;
  }
  if (res_out) *res_out = res;
  return it;
}

// Krylov bases of the flexible solvers grow with the iteration count instead of being sized for the restart length: FGMRES(30) at
// 128^3 would hold 61 vectors of 425 MB where the bench's solves use 2 to 14.  Contents are kept; new columns are zero.
// max_cols: the most columns the solver can ever ask for (restart length + 1): growth never goes beyond it (rounded up to the
// multiple of 4 the fused kernels read), so FGMRES(30) ends at 32 columns, not at 36
static void grow_basis(ifem_ctx *c, DBuf<double> &B, int64_t ld, int64_t cols, int64_t max_cols) {
  const int64_t have = ld > 0 ? int64_t(B.n) / ld : 0;
  if (have >= cols || ld <= 0) return;
  const int64_t cap = std::max<int64_t>((std::max(max_cols, cols) + 3) / 4 * 4, cols);
  const int64_t want = std::min(cap, std::max<int64_t>((cols + 7) / 8 * 8, have + have / 2));
  DBuf<double> nb;
  nb.alloc(size_t(want) * size_t(ld));
  if (have > 0) IFEM_HIP_CHECK(hipMemcpyAsync(nb.p, B.p, size_t(have) * size_t(ld) * sizeof(double), hipMemcpyDeviceToDevice, c->stream));
  IFEM_HIP_CHECK(hipMemsetAsync(nb.p + size_t(have) * size_t(ld), 0, size_t(want - have) * size_t(ld) * sizeof(double), c->stream));
  IFEM_HIP_CHECK(hipStreamSynchronize(c->stream));
  B.swap(nb);
}
// the callback gmres() gets for the pair (V: up to max_v columns, Z: up to max_v - 1)
static std::function<void(int, double *&, double *&)> basis_grower(ifem_ctx *c, DBuf<double> &Vb, DBuf<double> &Zb, int64_t ld, int max_v) {
  return [c, &Vb, &Zb, ld, max_v](int cols, double *&V, double *&Z) {
    grow_basis(c, Vb, ld, std::min(cols, max_v), max_v);
    grow_basis(c, Zb, ld, std::min(cols - 1, max_v - 1), max_v - 1);
    V = Vb.p; Z = Zb.p;
  };
}
constexpr int kBasisStart = 8; // columns a basis starts with

// Inner solver of the preconditioner: right-preconditioned restarted GMRES, x0 = 0, Krylov basis stored in single
// precision (half the Gram-Schmidt traffic; everything else fp64), single-pass classical Gram-Schmidt with the norm of
// the orthogonalised vector fused into the multi-axpy sweep.  Only used where an approximate A^-1 is admissible.
using OpF32 = std::function<void(const float *, double *)>;
static int gmres_f32basis(ifem_ctx *ctx, int64_t n, int64_t ld, const OpFn &A, const OpF32 &Pinv, const double *b, double *x,
                          int m, int maxit, double tol, float *V, double *z, double *w, double *res_out,
                          const std::function<void(double *, int)> &allreduce) {
  std::vector<double> H((size_t)(m + 1) * m, 0.0), cs(m), sn(m), g(m + 1), y(m), h(m + 4);
  v_zero(ctx, n, x);
  int it = 0;
  double res = 0;
  bool first = true;
  while (true) {
    if (first) { v_copy(ctx, n, b, w); first = false; }
    else { A(x, w); v_axpby(ctx, n, 1.0, b, -1.0, w); }
    double bb = v_dot(ctx, n, w, w);
    allreduce(&bb, 1);
    const double beta = std::sqrt(bb);
    res = beta;
    if (res <= tol || it >= maxit || !std::isfinite(res)) break;
    v_scale_store_f32(ctx, n, 1.0 / beta, w, V);
    std::fill(g.begin(), g.end(), 0.0);
    g[0] = beta;
    int j = 0;
    bool done = false;
    for (; j < m && it < maxit; ++j) {
      Pinv(V + (int64_t)j * ld, z);
      A(z, w);
      v_mdot_f32(ctx, n, j + 1, V, ld, w, h.data());
      allreduce(h.data(), j + 1);
      double ww;
      v_maxpy_f32(ctx, n, j + 1, V, ld, h.data(), w, &ww);
      allreduce(&ww, 1);
      for (int i = 0; i <= j; ++i) H[(size_t)i * m + j] = h[i];
      const double hn = std::sqrt(ww);
      H[(size_t)(j + 1) * m + j] = hn;
      if (hn > 0) v_scale_store_f32(ctx, n, 1.0 / hn, w, V + (int64_t)(j + 1) * ld);
      for (int i = 0; i < j; ++i) {
        const double t = cs[i] * H[(size_t)i * m + j] + sn[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)(i + 1) * m + j] = -sn[i] * H[(size_t)i * m + j] + cs[i] * H[(size_t)(i + 1) * m + j];
        H[(size_t)i * m + j] = t;
      }
      const double a = H[(size_t)j * m + j], c = H[(size_t)(j + 1) * m + j], r = std::hypot(a, c);
      cs[j] = a / r; sn[j] = c / r;
      H[(size_t)j * m + j] = r; H[(size_t)(j + 1) * m + j] = 0;
      g[j + 1] = -sn[j] * g[j]; g[j] = cs[j] * g[j];
      res = std::fabs(g[j + 1]);
      ++it;
      if (res <= tol || hn == 0 || !std::isfinite(res)) { ++j; done = true; break; }
    }
    for (int i = j - 1; i >= 0; --i) {
      double t = g[i];
      for (int k = i + 1; k < j; ++k) t -= H[(size_t)i * m + k] * y[k];
      y[i] = t / H[(size_t)i * m + i];
    }
    // x += P^-1 (V y): round V y to the basis precision (column m+1 of the basis is free at this point)
    v_zero(ctx, n, w);
    for (int i = 0; i < j; ++i) h[i] = -y[i];
    v_maxpy_f32(ctx, n, j, V, ld, h.data(), w, nullptr);
    float *vy = V + (int64_t)(m + 1) * ld;
    v_scale_store_f32(ctx, n, 1.0, w, vy);
    Pinv(vy, z);
    v_axpy(ctx, n, 1.0, z, x);
    if (done || it >= maxit) break;
  }
  if (res_out) *res_out = res;
  return it;
}

// plain CG, zero initial guess, absolute tolerance on ||r||_2
static int cg(ifem_ctx *ctx, int64_t n, const OpFn &A, const double *b, double *x, double tol, int maxit, double *r,
              double *p, double *q, const std::function<double(const double *, const double *)> &dot) {
  v_zero(ctx, n, x);
  v_copy(ctx, n, b, r);
  v_copy(ctx, n, b, p);
  double rr = dot(r, r);
  int it = 0;
  while (std::sqrt(rr) > tol && it < maxit) {
    A(p, q);
    const double al = rr / dot(p, q);
    v_axpy(ctx, n, al, p, x);
    v_axpy(ctx, n, -al, q, r);
    const double rn = dot(r, r);
    v_axpby(ctx, n, 1.0, r, rn / rr, p);
    rr = rn;
    ++it;
  }
  return it;
}

// CG / Jacobi-PCG whose recurrence scalars stay on the device: five launches and no host round trip per iteration; the
// host reads ||r||^2 every `chk` iterations, so up to chk - 1 iterations run beyond the tolerance (harmless: they only
// tighten the solve).  Same iterates as cg() / pcg_jacobi() up to that point.  On several ranks the partial sums are
// all-reduced on the stream (RCCL) between the reduction and the scalar update: still no host round trip per iteration.
static int cg_device(ifem_ctx *ctx, int64_t n, const OpFn &A, const double *diag, const double *b, double *x, double tol,
                     int maxit, double *r, double *z, double *p, double *q, int chk, double *s = nullptr) {
  if (s && ctx->tune.cg_single_reduction) {
    // one reduction per iteration (linalg.hip::cg1_*): u = D^-1 r lives in z (or IS r), w = A u in q, s = A p
    double *u = diag ? z : r, *w = q;
    cg1_init(ctx, n, b, diag, x, r, u, p, s);
    A(u, w);
    cg1_dots(ctx, n, r, u, w, true);
    int it = 0;
    double rr = cgd_rr(ctx);
    while (std::sqrt(rr) > tol && it < maxit) {
      const int burst = std::min(chk, maxit - it);
      for (int k = 0; k < burst; ++k) {
        cg1_update(ctx, n, diag, u, w, p, s, x, r);
        A(u, w);
        cg1_dots(ctx, n, r, u, w, false);
      }
      it += burst;
      rr = cgd_rr(ctx);
      if (!(rr == rr)) break; // NaN guard
    }
    return it;
  }
  cgd_init(ctx, n, b, diag, x, r, z, p);
  int it = 0;
  double rr = cgd_rr(ctx);
  while (std::sqrt(rr) > tol && it < maxit) {
    const int burst = std::min(chk, maxit - it);
    for (int k = 0; k < burst; ++k) {
      A(p, q);
      cgd_alpha(ctx, n, p, q);
      cgd_update(ctx, n, diag, p, q, x, r, z);
    }
    it += burst;
    rr = cgd_rr(ctx);
    if (!(rr == rr)) break; // NaN guard
  }
  return it;
}

// Jacobi-preconditioned CG, zero initial guess, same stopping rule (absolute tolerance on the TRUE residual ||r||_2).
// r and z must be adjacent (z = r + ld) so that <r,r> and <z,r> come out of one fused reduction.
static int pcg_jacobi(ifem_ctx *ctx, int64_t n, const OpFn &A, const double *diag, const double *b, double *x, double tol,
                      int maxit, double *r, int64_t ld, double *p, double *q,
                      const std::function<void(int, const double *, int64_t, const double *, double *)> &mdot) {
  double *z = r + ld;
  v_zero(ctx, n, x);
  v_copy(ctx, n, b, r);
  vec_div(ctx, n, diag, r, z);
  v_copy(ctx, n, z, p);
  double d2[2];
  mdot(2, r, ld, r, d2);
  double rr = d2[0], rz = d2[1];
  int it = 0;
  while (std::sqrt(rr) > tol && it < maxit) {
    A(p, q);
    double pq;
    mdot(1, p, n, q, &pq);
    const double al = rz / pq;
    v_axpy(ctx, n, al, p, x);
    v_axpy(ctx, n, -al, q, r);
    vec_div(ctx, n, diag, r, z);
    mdot(2, r, ld, r, d2);
    v_axpby(ctx, n, 1.0, z, d2[1] / rz, p);
    rr = d2[0]; rz = d2[1];
    ++it;
  }
  return it;
}

// leading dimension of a Krylov basis: a multiple of 64 doubles plus an odd number of 256-byte lines, so that the
// K+1 streams of a fused multi-dot do not start on the same HBM channel
static int64_t basis_ld(const ifem_ctx *ctx, int64_t n) { return ((n + 63) / 64) * 64 + ctx->tune.basis_pad; }

struct SolveState {
  ifem_ctx *ctx;
  const ifem_ins_params *P;
  const ifem_solver_opts *o;
  ifem_solve_stats st{};
  int64_t nuo, npo, n;
  double p_src_norm = 0, u_src_norm = 0; // norms of the pressure / velocity parts of the vector the preconditioner is applied to
  bool tight_candidate = false, tight_used = false; // inner_rel_first: the solve qualified for it / its first application ran with it
  // workspace carved out of ctx->work
  double *xu_ext, *xp_ext, *tu, *tp[7], *utmp, *inner_w, *inner_z, *outer_w;
};

// ghost-extended views of a compact owned vector [u_o | p_o]
static void extend_u(SolveState &S, const double *xu, const double **out) {
  ifem_ctx *c = S.ctx;
  if (c->halo.nranks == 1) { *out = xu; return; }
  v_copy(c, S.nuo, xu, S.xu_ext);
  halo_exchange(c, S.xu_ext); // exchanges the u part only when given a u-extended buffer (see comm.hip)
  *out = S.xu_ext;
}
static void extend_p(SolveState &S, const double *xp, const double **out) {
  ifem_ctx *c = S.ctx;
  if (c->halo.nranks == 1) { *out = xp; return; }
  v_copy(c, S.npo, xp, S.xp_ext);
  halo_exchange_p(c, S.xp_ext);
  *out = S.xp_ext;
}

// the assembled operator A^ = [A_uu B^T; B A_pp]: ghost-extended input (xu [dim*nUl], xp [nPl]), compact owned output
static void system_apply_ext(SolveState &S, const double *xu, const double *xp, double *y, bool time_it);

// the same on a compact owned vector (ghosts refreshed here).  Several ranks: both halos travel on the halo stream while the
// rows without a ghost column are multiplied (comm.hip::halo_start / halo_wait, PlanarCsr::split_rows)
static void system_apply_raw(SolveState &S, const double *x, double *y, bool time_it) {
  ifem_ctx *c = S.ctx;
  if (halo_overlap_ok(c) && !c->has_app && !((S.o->outer_matrix_free || !c->uu_is_stored) && c->mf_valid)) {
    build_row_split(c, c->Auu, c->nUo, &c->Bt, c->nPo);
    build_row_split(c, c->B, c->nUo);
    v_copy(c, S.nuo, x, S.xu_ext);
    v_copy(c, S.npo, x + S.nuo, S.xp_ext);
    halo_start(c, S.xu_ext, 0);
    halo_start(c, S.xp_ext, 1);
    spmv_uu(c, S.xu_ext, S.xp_ext, y, false, 1);
    spmv_b(c, S.xu_ext, y + S.nuo, 1);
    halo_wait(c);
    spmv_uu(c, S.xu_ext, S.xp_ext, y, false, 2);
    spmv_b(c, S.xu_ext, y + S.nuo, 2);
    return;
  }
  const double *xu, *xp;
  extend_u(S, x, &xu);
  extend_p(S, x + S.nuo, &xp);
  system_apply_ext(S, xu, xp, y, time_it);
}

// operator of the outer Krylov solver: A^, or C^T A^ C with the hanging rows replaced by their diagonal (hanging.hip)
static void system_apply(SolveState &S, const double *x, double *y, bool time_it) {
  ifem_ctx *c = S.ctx;
  if (!c->hang.active) { system_apply_raw(S, x, y, time_it); return; }
  const double *xe = hanging_input(c, x); // ghost-extended, hanging entries interpolated
  system_apply_ext(S, xe, xe + int64_t(c->dim) * c->nUl, y, time_it);
  hanging_output(c, x, false, y);
}

static void system_apply_ext(SolveState &S, const double *xu, const double *xp, double *y, bool time_it) {
  ifem_ctx *c = S.ctx;
  if ((S.o->outer_matrix_free || !c->uu_is_stored) && c->mf_valid && !c->has_app) { // A_uu x_u without the stored matrix (fp64 cell arithmetic)
    apply_uu_mf(c, xu, y);
    spmv_bt(c, xp, S.tu);
    v_axpy(c, S.nuo, 1.0, S.tu, y);
  } else
    spmv_uu(c, xu, xp, y, false);
  spmv_b(c, xu, y + S.nuo);
  if (c->has_app) { // SCnsIM: y_p += A_pp x_p
    spmv_app(c, xp, S.tp[5]);
    v_axpy(c, S.npo, 1.0, S.tp[5], y + S.nuo);
  }
}

static double dot_all(SolveState &S, int64_t n, const double *a, const double *b) {
  double d = v_dot(S.ctx, n, a, b);
  allreduce_sum(S.ctx, &d, 1);
  return d;
}

// ---- the pressure Schur complement S_m = B diag(M_u)^-1 B^T of one level (mass_schur(1,1), mpi_insim.cpp:44-49)
// S_m applied with two SpMVs (ghost refreshes in between): several ranks without the 2-deep halo plan, and the probing
static void sm_matrix_free(SolveState &S, const double *x, double *y, bool lowp) {
  ifem_ctx *c = S.ctx;
  const double *xe; extend_p(S, x, &xe);
  if (lowp) spmv_bt_f32(c, xe, S.tu); else spmv_bt(c, xe, S.tu);
  vec_mul(c, S.nuo, c->dinvMu.p, S.tu, S.tu);
  const double *te; extend_u(S, S.tu, &te);
  if (lowp) spmv_b_f32(c, te, y); else spmv_b(c, te, y);
}

static bool sm_is_explicit(const SolveState &S) {
  const ifem_ctx *c = S.ctx;
  return S.o->explicit_schur && (c->halo.nranks == 1 || c->halo.has_s);
}

// BlockSchurPreconditioner ctor (:44-49): S_m formed explicitly, once per constrained-dof set
static void sm_ensure(SolveState &S) {
  ifem_ctx *c = S.ctx;
  if (!sm_is_explicit(S)) return;
  if (c->halo.nranks == 1) {
    if (c->Sm.n_rows == 0) build_schur_pattern(c);
    schur_numeric(c);
    return;
  }
  // same matrix, distributed: pattern from the pressure lattice, values by probing
  if (c->Sm.n_rows == 0 && c->nPo) build_schur_pattern_box(c);
  if ((int64_t)c->xs_ext.n < c->halo.n_s_cols) c->xs_ext.alloc(c->halo.n_s_cols);
  if (!c->sm_valid) {
    const int ncol = c->dim == 3 ? 125 : 25;
    for (int col = 0; col < ncol; ++col) {
      schur_probe_vector(c, col, S.tp[1]);
      sm_matrix_free(S, S.tp[1], S.tp[3], false);
      schur_probe_fill(c, col, S.tp[3]);
    }
    c->sm_valid = true;
    c->sm_f32_valid = false;
    c->sm_version++;
  }
}

// y = S_m x on compact owned pressure vectors
static void sm_apply(SolveState &S, const double *x, double *y, bool lowp) {
  ifem_ctx *c = S.ctx;
  if (sm_is_explicit(S) && c->halo.nranks > 1) {
    v_copy(c, S.npo, x, c->xs_ext.p);
    if (halo_overlap_ok(c)) { // rows of S_m without a far column while the 2-deep halo travels
      build_row_split(c, c->Sm, c->nPo);
      halo_start(c, c->xs_ext.p, 2);
      spmv_sm(c, c->xs_ext.p, y, lowp, 1);
      halo_wait(c);
      spmv_sm(c, c->xs_ext.p, y, lowp, 2);
      return;
    }
    halo_exchange_s(c, c->xs_ext.p);
    spmv_sm(c, c->xs_ext.p, y, lowp);
    return;
  }
  // the approximate-preconditioner kinds (1, 3) also stream S_m in single precision: it only ever acts inside CG to
  // 1e-3 ||v|| and the rounding (6e-8 relative) is far below the eigenvalue ratio of this Laplacian-like operator
  if (sm_is_explicit(S)) { spmv_sm(c, x, y, lowp); return; }
  sm_matrix_free(S, x, y, lowp);
}

// ---- multigrid V-cycle for S_m over the levels attached with ifem_mg_attach (geometric, rediscretised coarse operators).
// Smoother: Chebyshev iteration on D^-1 S_m (D = diag S_m) over [lambda_max / ratio, 1.1 lambda_max]: no inner products, a
// fixed polynomial, so the V-cycle (same polynomial before and after the coarse correction) is a fixed SPD operator and
// CG may use it as its preconditioner.  Level vectors (ctx->mg_vec, nPl + 8 each): 0 right-hand side / residual,
// 1 solution, 2 Chebyshev direction, 3 operator product, 4 prolongated correction.  Compact owned entries come first, so
// the same buffers serve as ghost-extended pressure vectors for the transfers.
// A fixed launch sequence on fixed buffers as a hipGraph (ctx.hpp::VcGraph): replayed while `key` -- every pointer, bound, parameter and count the
// sequence's kernel arguments are made of -- stays what it was when the graph was captured.  A new key runs `body` eagerly once (whatever is
// allocated or converted lazily inside exists afterwards) and is captured at its next occurrence.  Returns false when the runtime refused the
// capture: the caller stops asking (body has run eagerly by then).
static bool graph_run(ifem_ctx *c, ifem_ctx::VcGraph &G, std::vector<uint64_t> &key, const std::function<void()> &body) {
  if (G.exec && key == G.key) {
    if (hipGraphLaunch(G.exec, c->stream) == hipSuccess) { ++G.launches; return true; }
    (void)hipGetLastError(); // a replay the runtime refuses: give the graph up and run the sequence eagerly
    G.destroy();
    G.armed = false;
    body();
    return false;
  }
  if (!(G.armed && key == G.key)) {
    G.destroy();
    G.key = key; G.armed = true;
    body();
    return true;
  }
  bool captured = hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
  if (captured) {
    bool threw = false;
    try { body(); }
    catch (...) { threw = true; } // e.g. a call that is illegal during capture on a path the eager warm-up did not reach
    if (threw) { // end and discard the capture, clear the error, and run the sequence eagerly: an eager run may well succeed
      hipGraph_t dead = nullptr;
      (void)hipStreamEndCapture(c->stream, &dead);
      if (dead) (void)hipGraphDestroy(dead);
      (void)hipGetLastError();
      G.destroy();
      G.armed = false;
      body();
      return false;
    }
    captured = hipStreamEndCapture(c->stream, &G.graph) == hipSuccess && G.graph != nullptr;
    captured = captured && hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0) == hipSuccess;
    captured = captured && hipGraphLaunch(G.exec, c->stream) == hipSuccess;
  }
  if (captured) { ++G.captures; ++G.launches; return true; }
  (void)hipGetLastError();
  G.destroy();
  G.armed = false;
  body();
  return false;
}
// rocprofv3 (rocprofiler-sdk 7.2) dies with a segmentation fault inside hipGraphLaunch when kernel tracing is on: under a profiler the cycles are
// launched eagerly.  Detected once per process by the tool library being loaded (or announced: rocprofv3 exports ROCP_TOOL_LIBRARIES for its
// child) -- the one place where this library looks at its environment.
static bool profiler_attached() {
  static const bool on = [] {
    if (std::getenv("ROCP_TOOL_LIBRARIES") || std::getenv("ROCPROFILER_REGISTER_FORCE_LOAD")) return true;
    for (const char *lib : {"librocprofiler-sdk-tool.so", "librocprofiler-sdk-tool.so.1", "librocprofiler-sdk-tool.so.0"}) {
      if (void *h = dlopen(lib, RTLD_NOLOAD | RTLD_LAZY)) { dlclose(h); return true; }
    }
    return false;
  }();
  return on;
}
static inline void key_ptr(std::vector<uint64_t> &key, const void *p) { key.push_back(uint64_t(reinterpret_cast<uintptr_t>(p))); }
static inline void key_f64(std::vector<uint64_t> &key, double v) { uint64_t b; std::memcpy(&b, &v, 8); key.push_back(b); }

// A replicated coarse level (a single-rank context of the whole coarse mesh below a partitioned level) is computed by every rank
// on its own device: its operators carry float / double atomics whose order differs between devices, so the replicas agree to
// rounding only, and replica-local decisions (the 1 % early exit of a power iteration) may differ.  Quantities that steer the
// smoother -- the Chebyshev bounds -- are therefore agreed on over the partitioned level's communicator (maximum over the ranks):
// every rank then applies the SAME polynomial, and the level stays one (symmetric) preconditioner for all of them.
static ifem_ctx *replica_world(ifem_ctx *c) {
  if (c->halo.nranks > 1) return nullptr;
  for (ifem_ctx *p = c->mg_fine; p; p = p->mg_fine)
    if (p->halo.nranks > 1) return p;
  return nullptr;
}

struct MgSm {
  std::vector<SolveState> L; // level 0 = the context being solved
  bool lowp = false;
  int nu = 2;
  double ratio = 4.0;
};

// geometry blocks, S_m, its diagonal and the eigenvalue bound of every level; cheap when nothing changed
static void mg_sm_setup(MgSm &M, int use_nonzero) {
  for (size_t l = 0; l < M.L.size(); ++l) {
    SolveState &S = M.L[l];
    ifem_ctx *c = S.ctx;
    if (l > 0) { // coarse levels are rediscretised: B, B^T, M_p, diag(M_u) of the level's own mesh and constraint set
      // (ifem_tuning::geo_cache = 2, "every assembly is a new constrained-dof set": once per assembly of the finest level,
      // not once per preconditioner application)
      ifem_ctx *f0 = M.L[0].ctx;
      if (!(c->tune.geo_cache == 2 && c->geo_refresh_stamp == f0->asm_version)) {
        launch_ins_assemble_geometry(c, S.P, use_nonzero);
        c->geo_refresh_stamp = f0->asm_version;
      }
      sm_ensure(S);
    }
    for (auto &v : c->mg_vec)
      if ((int64_t)v.n < c->nPl + 8) { v.alloc((size_t)c->nPl + 8); IFEM_HIP_CHECK(hipMemsetAsync(v.p, 0, v.n * sizeof(double), c->stream)); }
    // S_m in operator form (the finest level of a partition without the 2-deep pressure halo: unstructured strips): the smoother
    // needs its diagonal only, which the rows of B give; "valid" = the diagonal belongs to the present blocks
    const bool opform = !sm_is_explicit(S);
    if (opform && !c->sm_valid) { c->sm_valid = true; c->sm_version++; }
    if (c->sm_mg_version == c->sm_version && c->sm_lmax > 0) continue;
    if ((int64_t)c->sm_dinv.n != c->nPo) c->sm_dinv.alloc((size_t)c->nPo);
    if (opform) {
      const double *de; extend_u(S, c->dinvMu.p, &de);
      sm_diag_from_blocks(c, de, c->sm_dinv.p);
    } else
    scalar_diag(c, c->Sm, c->Sm.val.p, c->sm_dinv.p); // owned rows; the diagonal entry has a local column id on every layout
    vec_recip(c, S.npo, c->sm_dinv.p);
    // largest eigenvalue of D^-1 S_m: power iteration (set-up only: host-synchronised norms) from a fixed rough vector, or --
    // when S_m was re-formed for another constrained-dof set of the same mesh -- from the previous run's last iterate, then
    // stopped once the estimate moves by less than 1 %
    double *x = c->mg_vec[1].p, *y = c->mg_vec[3].p;
    const bool warm = c->sm_lmax > 0 && (int64_t)c->sm_eig.n == S.npo && S.npo > 0;
    if (warm) v_copy(c, S.npo, c->sm_eig.p, x);
    else vec_rough(c, S.npo, int64_t(c->halo.rank) * 1000003, x);
    double lam = 0;
    const int n_pow = c->tune.eig_steps > 0 ? c->tune.eig_steps : 14;
    for (int it = 0; it < n_pow; ++it) {
      double nx = v_dot(c, S.npo, x, x);
      allreduce_sum(c, &nx, 1);
      if (!(nx > 0)) break;
      v_scale(c, S.npo, 1.0 / std::sqrt(nx), x);
      sm_apply(S, x, y, false);
      vec_mul(c, S.npo, c->sm_dinv.p, y, y);
      double ny = v_dot(c, S.npo, y, y);
      allreduce_sum(c, &ny, 1);
      const double prev = lam;
      lam = std::sqrt(ny);
      v_copy(c, S.npo, y, x);
      if (warm && it >= 1 && std::fabs(lam - prev) <= 0.01 * lam) break;
    }
    if (S.npo > 0) {
      if ((int64_t)c->sm_eig.n != S.npo) c->sm_eig.alloc((size_t)S.npo);
      v_copy(c, S.npo, x, c->sm_eig.p);
    }
    if (ifem_ctx *w = replica_world(c)) allreduce_max(w, &lam, 1);
    c->sm_lmax = lam > 0 ? lam : 1.0;
    c->sm_mg_version = c->sm_version;
  }
}

// nsteps Chebyshev steps on S x = b from the pair (x, r = b - S x); with keep_r the residual is kept up to date on exit
// (one operator product per step), without it the last product is skipped.  lo / hi: target interval of D^-1 S.
static void mg_sm_smooth(MgSm &M, size_t l, int nsteps, double lo, double hi, double *x, double *r, bool keep_r) {
  SolveState &S = M.L[l];
  ifem_ctx *c = S.ctx;
  double *d = c->mg_vec[2].p, *t = c->mg_vec[3].p;
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
  double rho_old = 1.0 / sigma;
  cheb_init(c, S.npo, 1.0 / theta, c->sm_dinv.p, r, d);
  for (int k = 0; k < nsteps; ++k) {
    const bool last = k == nsteps - 1;
    if (last && !keep_r) { v_axpy(c, S.npo, 1.0, d, x); break; }
    sm_apply(S, d, t, M.lowp);
    const double rho_new = 1.0 / (2.0 * sigma - rho_old);
    // x += d; r -= S d; d = rho_new rho_old d + (2 rho_new / delta) D^-1 r   (the new d is unused after the last step)
    cheb_step(c, S.npo, rho_new * rho_old, 2.0 * rho_new / delta, c->sm_dinv.p, t, x, r, d);
    rho_old = rho_new;
  }
}

// level l: mg_vec[1] = V(mg_vec[0]); mg_vec[0] is overwritten by the residual
static void mg_sm_vcycle(MgSm &M, size_t l) {
  SolveState &S = M.L[l];
  ifem_ctx *c = S.ctx;
  double *r = c->mg_vec[0].p, *x = c->mg_vec[1].p;
  const double hi = 1.1 * c->sm_lmax;
  v_zero(c, S.npo, x);
  if (l + 1 == M.L.size()) { // coarsest level: a longer polynomial over a wide interval stands in for a direct solve
    mg_sm_smooth(M, l, 30, hi / 900.0, hi, x, r, false);
    return;
  }
  const double lo = hi / M.ratio;
  mg_sm_smooth(M, l, M.nu, lo, hi, x, r, true);
  // restriction r_c = P^T r: gather per LOCAL coarse node from the owned fine nodes, ghost rows travel to their owners
  SolveState &Sc = M.L[l + 1];
  ifem_ctx *cc = Sc.ctx;
  mg_csr_apply(c, c->mg_Rp, r, cc->mg_vec[0].p, false);
  if (c->mg_replica) allreduce_sum_vec(c, cc->mg_vec[0].p, cc->nPo, cc->mg_vec[4].p); // replicated coarse level: sum of the ranks' partial rows
  else halo_reverse_add_p(cc, cc->mg_vec[0].p);
  mg_sm_vcycle(M, l + 1);
  // prolongation e = P x_c from the ghost-extended coarse correction, then x += e, r -= S e
  halo_exchange_p(cc, cc->mg_vec[1].p);
  double *e = c->mg_vec[4].p, *t = c->mg_vec[3].p;
  mg_csr_apply(c, c->mg_Pp, cc->mg_vec[1].p, e, false);
  sm_apply(S, e, t, M.lowp);
  v_axpy(c, S.npo, 1.0, e, x);
  v_axpy(c, S.npo, -1.0, t, r);
  mg_sm_smooth(M, l, M.nu, lo, hi, x, r, false);
}

// CG preconditioned by one V-cycle, zero initial guess, absolute tolerance on the true residual ||b - S x||_2 (the
// reference's stopping rule for CG(S_m), mpi_insim.cpp:88-89)
static int pcg_mg_sm(MgSm &M, const double *b, double *x, double tol, int maxit, double *r, double *p, double *q, double *sv = nullptr) {
  SolveState &S = M.L[0];
  ifem_ctx *c = S.ctx;
  auto dot2 = [&](const double *a1, const double *b1, const double *a2, const double *b2, double *out) {
    out[0] = v_dot(c, S.npo, a1, b1);
    out[1] = (a1 == a2 && b1 == b2) ? out[0] : v_dot(c, S.npo, a2, b2);
    allreduce_sum(c, out, 2);
  };
  double *zin = c->mg_vec[0].p, *z = c->mg_vec[1].p;
  // the cycle: eagerly, or replayed as a hipGraph on small single-rank chains (as the A_uu V-cycle, precond_vmult)
  bool graph_ok = c->tune.vcycle_graph_cells > 0 && c->n_cells <= c->tune.vcycle_graph_cells && !c->profile && !kprof_root(c).on && !profiler_attached();
  for (const SolveState &L : M.L) graph_ok = graph_ok && L.ctx->halo.nranks == 1 && !L.ctx->mg_replica && sm_is_explicit(L);
  auto sm_cycle = [&]() {
    if (!graph_ok) { mg_sm_vcycle(M, 0); return; }
    std::vector<uint64_t> key;
    key.push_back(M.L.size()); key.push_back(uint64_t(M.nu)); key_f64(key, M.ratio); key.push_back(uint64_t(M.lowp));
    for (const SolveState &L : M.L) {
      ifem_ctx *lc = L.ctx;
      for (auto &v : lc->mg_vec) key_ptr(key, v.p);
      key_ptr(key, lc->sm_dinv.p); key_ptr(key, lc->Sm.val.p); key_ptr(key, lc->Sm_f32.p); key_ptr(key, lc->Sm.rowptr.p);
      key_ptr(key, lc->mg_Rp.col.p); key_ptr(key, lc->mg_Pp.col.p);
      key.push_back(uint64_t(lc->nPo)); key.push_back(uint64_t(lc->sm_version)); key.push_back(uint64_t(lc->tune.sm_lanes)); key.push_back(uint64_t(lc->sm_f32_valid));
      key_f64(key, lc->sm_lmax);
    }
    if (!graph_run(c, c->sm_graph, key, [&]() { mg_sm_vcycle(M, 0); })) { c->tune.vcycle_graph_cells = 0; graph_ok = false; }
  };
  if (sv && c->tune.cg_single_reduction) {
    // the recurrence of linalg.hip::cg1_* with the V-cycle as its preconditioner: u = V r, w = S u, the two inner products of the step in one
    // pass with alpha / beta formed on the device (several ranks: one all-reduce on the stream), the vectors updated in one pass; the host
    // waits once per iteration, for ||r||^2 of the stopping rule (it waited six times: three pairs of separate dot products)
    auto norm2 = [&](const double *v) { double t = v_dot(c, S.npo, v, v); allreduce_sum(c, &t, 1); return t; };
    cg1_init(c, S.npo, b, nullptr, x, r, r, p, sv);
    double rr = norm2(r);
    int it = 0;
    while (std::sqrt(rr) > tol && it < maxit) {
      v_copy(c, S.npo, r, zin);
      sm_cycle();                      // z = V r
      sm_apply(S, z, q, M.lowp);       // w = S z
      cg1_dots(c, S.npo, r, z, q, it == 0);
      cg1_update(c, S.npo, nullptr, z, q, p, sv, x, r);
      rr = norm2(r);
      ++it;
      if (!(rr == rr)) break; // NaN guard
    }
    return it;
  }
  v_zero(c, S.npo, x);
  v_copy(c, S.npo, b, r);
  double d2[2];
  dot2(r, r, r, r, d2);
  double rr = d2[0], rz = 0;
  int it = 0;
  while (std::sqrt(rr) > tol && it < maxit) {
    v_copy(c, S.npo, r, zin);
    sm_cycle();
    double rz_new[2];
    dot2(r, z, r, z, rz_new);
    if (it == 0) v_copy(c, S.npo, z, p);
    else v_axpby(c, S.npo, 1.0, z, rz_new[0] / rz, p);
    rz = rz_new[0];
    sm_apply(S, p, q, M.lowp);
    double pq[2];
    dot2(p, q, p, q, pq);
    const double al = rz / pq[0];
    v_axpy(c, S.npo, al, p, x);
    v_axpy(c, S.npo, -al, q, r);
    dot2(r, r, r, r, d2);
    rr = d2[0];
    ++it;
    if (!(rr == rr)) break; // NaN guard
  }
  return it;
}

static void carve_workspace(SolveState &S, bool krylov = true);

// ---- multigrid V-cycle for A_uu (IFEM_AINV_MG): stands where the reference has MUMPS (mpi_insim.cpp:124-127).  Every level
// applies its own matrix-free A_uu (apply_mf.hip, single-precision cell arithmetic) at the evaluation point injected from
// the level above; smoother = Chebyshev iteration on (node-block diagonal)^-1 A_uu.  The coarse block diagonals are
// integrated matrix-free (mg.hip::k_uu_diag), the finest level uses the blocks of the assembled matrix.  Level vectors
// (ctx->mguf_vec, SINGLE precision since round 3 -- the cell arithmetic of the level operators, the node blocks of the smoother
// and the basis of the Krylov solver around the cycle are single precision already -- dim * nUl + 8 each): 0 right-hand side /
// residual, 1 solution, 2 direction, 3 (unused: the operator product is consumed inside the fused gather), 4 prolongated
// correction; compact owned entries first, so the buffers double as ghost-extended velocity vectors.  The double vectors
// ctx->mgu_vec[1..3] remain as scratch of the eigenvalue estimates.
struct MgUu {
  std::vector<SolveState> L;
  int nu = 3, nu_post = 3;
  double ratio = 8.0;
};

// matrix-free A_uu of one level on a compact owned vector.  Several ranks: the cells whose nodes are all owned are
// processed while the velocity halo travels, the cells of the ghost layer and the node gather after it has arrived
static void uu_apply_level(SolveState &S, const double *x, double *y, const MfFuse *fuse = nullptr) {
  ifem_ctx *c = S.ctx;
  if (halo_overlap_ok(c)) {
    build_mf_cell_split(c);
    v_copy(c, S.nuo, x, S.xu_ext);
    halo_start(c, S.xu_ext, 0);
    apply_uu_mf(c, S.xu_ext, y, true, fuse, 1);
    halo_wait(c);
    apply_uu_mf(c, S.xu_ext, y, true, fuse, 2);
    return;
  }
  const double *xe; extend_u(S, x, &xe);
  apply_uu_mf(c, xe, y, true, fuse);
}

// the same on a level vector of the V-cycle (single precision, ghost-extended in place: owned entries first), fused form
static void uu_apply_level_f32(SolveState &S, float *x_ext, const MfFuseT<float> *fuse) {
  ifem_ctx *c = S.ctx;
  if (halo_overlap_ok(c)) {
    build_mf_cell_split(c);
    halo_start_f32(c, x_ext);
    apply_uu_mf_f32v(c, x_ext, fuse, 1);
    halo_wait(c);
    apply_uu_mf_f32v(c, x_ext, fuse, 2);
    return;
  }
  halo_exchange_f32(c, x_ext);
  apply_uu_mf_f32v(c, x_ext, fuse);
}

static void mg_uu_setup(MgUu &M, bool force_bounds = false) {
  ifem_ctx *f0 = M.L[0].ctx;
  // size of the evaluation point the bounds below were estimated at: A_uu carries rho C(u), so a bound taken at a small
  // velocity must not outlive a grown convective part.  One norm per assembly on the finest level (all levels see the same
  // field by injection); a change of more than 10 % refreshes the estimates (warm: 2-3 power steps per level)
  if (f0->uu_evn_asm != f0->asm_version) {
    double e2 = v_dot(f0, int64_t(f0->dim) * f0->nUo, f0->mf_eval.p, f0->mf_eval.p);
    allreduce_sum(f0, &e2, 1);
    f0->uu_evn = std::sqrt(e2);
    f0->uu_evn_asm = f0->asm_version;
  }
  for (size_t l = 0; l < M.L.size(); ++l) {
    SolveState &S = M.L[l];
    ifem_ctx *c = S.ctx;
    const int64_t nv = int64_t(c->dim) * c->nUl + 8;
    for (int k = 1; k <= 3; ++k) {
      auto &v = c->mgu_vec[k];
      if ((int64_t)v.n < nv) { v.alloc((size_t)nv); IFEM_HIP_CHECK(hipMemsetAsync(v.p, 0, v.n * sizeof(double), c->stream)); }
    }
    for (auto &v : c->mguf_vec)
      if ((int64_t)v.n < nv) { v.alloc((size_t)nv); IFEM_HIP_CHECK(hipMemsetAsync(v.p, 0, v.n * sizeof(float), c->stream)); }
    if (l > 0 && c->uu_mg_version != f0->asm_version) { // operator state of a coarse level: rediscretisation at the injected point
      ifem_ctx *p = M.L[l - 1].ctx;
      c->mf_params = f0->mf_params;
      c->mf_noconv = f0->mf_noconv;
      c->asm_constraint_set = f0->asm_constraint_set;
      const size_t ne = size_t(c->dim) * size_t(c->nUl);
      if (c->mf_eval.n != ne) c->mf_eval.alloc(ne);
      mg_inject_nodes(c, c->nUo, p->mg_inj_u.p, p->mf_eval.p, c->mf_eval.p);
      if (p->mg_replica) allreduce_sum_vec(p, c->mf_eval.p, int64_t(c->dim) * c->nUo, c->mgu_vec[1].p); // every rank injects the nodes it owns
      else halo_exchange(c, c->mf_eval.p);
      c->mf_valid = true;
      uu_block_diag_mf(c);
      c->uu_mg_version = f0->asm_version;
    }
    if (l + 1 < M.L.size()) { // transfer masks towards the next level: functions of the two constrained-dof sets
      ifem_ctx *cc = M.L[l + 1].ctx;
      const int cs = f0->asm_constraint_set;
      const int64_t key[2] = {c->flag_id[cs], cc->flag_id[cs]};
      if (c->mg_Pu_mask.n != c->mg_Pu.col.n * 8 || c->mg_mask_key[0] != key[0] || c->mg_mask_key[1] != key[1]) {
        const uint8_t *ff = c->has_c[cs] ? c->is_c[cs].p : nullptr, *fc = cc->has_c[cs] ? cc->is_c[cs].p : nullptr;
        mg_csr_mask(c, c->mg_Ru, ff, fc, c->mg_Ru_mask); // rows: local coarse nodes, columns: owned fine nodes
        mg_csr_mask(c, c->mg_Pu, fc, ff, c->mg_Pu_mask); // rows: owned fine nodes, columns: local coarse nodes
        c->mg_mask_key[0] = key[0]; c->mg_mask_key[1] = key[1];
      }
    }
    // eigenvalue bound of (block D)^-1 A_uu: depends on the parameters and the constrained-dof set, hardly on the evaluation
    // point (the viscous and mass terms carry the top of the spectrum): estimated once per such state
    const double key[6] = {f0->mf_params.viscosity, f0->mf_params.rho, f0->mf_params.grad_div, f0->mf_params.dt,
                           double(f0->mf_noconv), double(c->flag_id[c->asm_constraint_set])};
    bool same = c->uu_lmax > 0 && !force_bounds;
    for (int i = 0; i < 6; ++i) same = same && key[i] == c->uu_lmax_key[i];
    same = same && std::fabs(f0->uu_evn - c->uu_lmax_evn) <= 0.1 * std::max(c->uu_lmax_evn, 1e-300);
    if (c->tune.geo_cache == 2 && c->uu_lmax_asm != f0->asm_version) same = false; // measurement mode: a new set per assembly
    c->uu_lmax_asm = f0->asm_version;
    if (same) continue;
    if (force_bounds) c->uu_lmax = 0; // cold estimate: all 12 steps from a rough vector
    // power iteration on B A_uu.  A level that has an estimate from another constrained-dof set starts from that run's last
    // iterate and stops once the estimate moves by less than 1 % (the top of this spectrum belongs to the mesh, not to the
    // set); the first estimate starts from a fixed rough vector and runs all 12 steps.
    double *x = c->mgu_vec[1].p, *y = c->mgu_vec[3].p, *z = c->mgu_vec[2].p;
    const bool warm = c->uu_lmax > 0 && (int64_t)c->uu_eig.n == S.nuo && S.nuo > 0;
    if (warm) v_copy(c, S.nuo, c->uu_eig.p, x);
    else vec_rough(c, S.nuo, int64_t(c->halo.rank) * 7000003, x);
    double lam = 0;
    const int n_pow = c->tune.eig_steps > 0 ? c->tune.eig_steps : 12;
    for (int it = 0; it < n_pow; ++it) {
      double nx = v_dot(c, S.nuo, x, x);
      allreduce_sum(c, &nx, 1);
      if (!(nx > 0)) break;
      v_scale(c, S.nuo, 1.0 / std::sqrt(nx), x);
      uu_apply_level(S, x, y);
      bjac_apply(c, y, z);
      double nz = v_dot(c, S.nuo, z, z);
      allreduce_sum(c, &nz, 1);
      const double prev = lam;
      lam = std::sqrt(nz);
      v_copy(c, S.nuo, z, x);
      if (warm && it >= 1 && std::fabs(lam - prev) <= 0.01 * lam) break;
    }
    if (S.nuo > 0) {
      if ((int64_t)c->uu_eig.n != S.nuo) c->uu_eig.alloc((size_t)S.nuo);
      v_copy(c, S.nuo, x, c->uu_eig.p);
    }
    if (ifem_ctx *w = replica_world(c)) allreduce_max(w, &lam, 1);
    c->uu_lmax = lam > 0 && std::isfinite(lam) ? lam : 1.0;
    if (S.o->verbose) fprintf(stderr, "[ifem] A_uu V-cycle level %zu (%lld cells): lambda_max estimate %.4f (%s)\n", l, (long long)c->n_cells, c->uu_lmax, warm ? "warm" : "cold");
    c->uu_lmax_evn = f0->uu_evn;
    for (int i = 0; i < 6; ++i) c->uu_lmax_key[i] = key[i];
  }
}

// d_ready: the first direction d = (1/theta) B r is already in place (written by the fused residual update, MfFuse mode 3)
static void mg_uu_smooth(MgUu &M, size_t l, int nsteps, double lo, double hi, float *x, float *r, bool keep_r, bool d_ready = false) {
  SolveState &S = M.L[l];
  ifem_ctx *c = S.ctx;
  float *d = c->mguf_vec[2].p;
  const double theta = 0.5 * (hi + lo), delta = 0.5 * (hi - lo), sigma = theta / delta;
  double rho_old = 1.0 / sigma;
  if (!d_ready) cheb_init_block_f32(c, 1.0 / theta, r, d);
  for (int k = 0; k < nsteps; ++k) {
    const bool last = k == nsteps - 1;
    if (last && !keep_r) { v_axpy_f32v(c, S.nuo, 1.0f, d, x); break; }
    const double rho_new = 1.0 / (2.0 * sigma - rho_old);
    // x += d; r -= A d; d = rho_new rho_old d + (2 rho_new / delta) B r, fused into the node gather of the product
    MfFuseT<float> f;
    f.mode = 2; f.a = rho_new * rho_old; f.b = 2.0 * rho_new / delta; f.xs = x; f.r = r; f.d = d;
    uu_apply_level_f32(S, d, &f);
    rho_old = rho_new;
  }
}

// level l: mguf_vec[1] = V(mguf_vec[0]); mguf_vec[0] is overwritten by the residual
static void mg_uu_vcycle(MgUu &M, size_t l) {
  SolveState &S = M.L[l];
  ifem_ctx *c = S.ctx;
  float *r = c->mguf_vec[0].p, *x = c->mguf_vec[1].p;
  const double hi = 1.1 * c->uu_lmax;
  IFEM_HIP_CHECK(hipMemsetAsync(x, 0, size_t(S.nuo) * sizeof(float), c->stream));
  if (l + 1 == M.L.size()) {
    mg_uu_smooth(M, l, 24, hi / 400.0, hi, x, r, false);
    return;
  }
  const double lo = hi / M.ratio;
  mg_uu_smooth(M, l, M.nu, lo, hi, x, r, true);
  SolveState &Sc = M.L[l + 1];
  ifem_ctx *cc = Sc.ctx;
  mg_csr_apply_nodes_f32(c, c->mg_Ru, r, c->mg_Ru_mask, cc->mguf_vec[0].p);
  // partial restrictions -> the coarse residual: ghost rows to their owners, or (replicated coarse level: every rank holds all rows) the
  // sum over the ranks; from there down nothing is exchanged and the prolongation reads the replica directly
  if (c->mg_replica) allreduce_sum_vec_f32(c, cc->mguf_vec[0].p, int64_t(cc->dim) * cc->nUo, cc->mguf_vec[4].p);
  else halo_reverse_add_f32(cc, cc->mguf_vec[0].p);
  mg_uu_vcycle(M, l + 1);
  halo_exchange_f32(cc, cc->mguf_vec[1].p);
  float *e = c->mguf_vec[4].p;
  mg_csr_apply_nodes_f32(c, c->mg_Pu, cc->mguf_vec[1].p, c->mg_Pu_mask, e);
  MfFuseT<float> f; // x += e; r -= A e; d = (1/theta) B r: the first direction of the post-smoothing sweep
  f.mode = 3; f.xs = x; f.r = r; f.d = c->mguf_vec[2].p; f.b = 1.0 / (0.5 * (hi + lo));
  uu_apply_level_f32(S, e, &f);
  mg_uu_smooth(M, l, M.nu_post, lo, hi, x, r, false, true);
}

static void precond_vmult(SolveState &S, const double *src, double *dst) {
  ifem_ctx *c = S.ctx;
  const ifem_ins_params *P = S.P;
  const ifem_solver_opts *o = S.o;
  const double *src0 = src, *src1 = src + S.nuo;
  double *dst0 = dst, *dst1 = dst + S.nuo;
  double *tmp = S.tp[0], *r = S.tp[1], *p = S.tp[2], *q = S.tp[3];
  auto pdot = [&](const double *a, const double *b) { return dot_all(S, S.npo, a, b); };
  const double n1 = std::sqrt(pdot(src1, src1));
  S.p_src_norm = n1;
  if (S.st.precond_applies == 0 && o->inner_rel_first > 0) { // pressure share of the first Krylov vector (see the inner solve below)
    double uu = 0;
    v_mdot(S.ctx, S.nuo, 1, src0, S.nuo, src0, &uu);
    allreduce_sum(S.ctx, &uu, 1);
    S.u_src_norm = std::sqrt(uu);
  }
  // section marks on the stream (read by precond_section_times once the solve is over)
  auto mark = [&]() {
    if (c->pc_ev.size() <= c->pc_used) {
      hipEvent_t e = nullptr;
      IFEM_HIP_CHECK(hipEventCreate(&e));
      c->pc_ev.push_back(e);
    }
    IFEM_HIP_CHECK(hipEventRecord(c->pc_ev[c->pc_used++], c->stream));
  };
  mark();
  // CG for Mp (:69-84)
  // the approximate-preconditioner kinds stream M_p in single precision like S_m (the solve is to 1e-6, the rounding of the
  // values 6e-8).  (p.q fused into this SpMV was measured: a block reduction in each of its 67 k small blocks costs more
  // than the separate pass over the two vectors.)
  const bool mp_f32 = o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_F32 || o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_MF || o->ainv_kind == IFEM_AINV_MG;
  OpFn mp = [&](const double *x, double *y) {
    if (halo_overlap_ok(c)) {
      build_row_split(c, c->Mp, c->nPo);
      v_copy(c, S.npo, x, S.xp_ext);
      halo_start(c, S.xp_ext, 1);
      spmv_mp(c, S.xp_ext, y, 1, mp_f32);
      halo_wait(c);
      spmv_mp(c, S.xp_ext, y, 2, mp_f32);
      return;
    }
    const double *xe; extend_p(S, x, &xe);
    spmv_mp(c, xe, y, 0, mp_f32);
  };
  // kinds 1 and 3 (approximate preconditioner) also put a Jacobi preconditioner on the two pressure CG solves: same
  // stopping rule on the true residual, fewer iterations (the reference uses PreconditionNone; counts are no parity target)
  const bool pjac = o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_F32 || o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_MF || o->ainv_kind == IFEM_AINV_MG;
  auto pmdot = [&](int k, const double *V, int64_t ld, const double *w, double *out) {
    v_mdot(c, S.npo, k, V, ld, w, out, /*all_ranks=*/true);
  };
  const int pmax = (int)std::min<int64_t>(std::max<int64_t>(c->n_global_p, 1), 1 << 30);
  // device-resident recurrences on any number of ranks: the dot products are all-reduced on the stream (comm.hip::allreduce_sum_dev)
  const bool dev_cg = o->device_cg != 0;
  if (dev_cg) {
    if (pjac) scalar_diag(c, c->Mp, c->Mp.val.p, S.tp[5]);
    S.st.cg_mp_iters += cg_device(c, S.npo, mp, pjac ? S.tp[5] : nullptr, src1, tmp, std::max(o->mp_abs, o->mp_rel * n1), pmax,
                                  S.tp[1], S.tp[2], S.tp[3], S.tp[4], 4, S.tp[6]);
  } else if (pjac) {
    scalar_diag(c, c->Mp, c->Mp.val.p, S.tp[5]);
    S.st.cg_mp_iters += pcg_jacobi(c, S.npo, mp, S.tp[5], src1, tmp, std::max(o->mp_abs, o->mp_rel * n1), pmax, S.tp[1], c->nPl, S.tp[3], S.tp[4], pmdot); // r = tp[1], z = tp[2]
  } else
  S.st.cg_mp_iters += cg(c, S.npo, mp, src1, tmp, std::max(o->mp_abs, o->mp_rel * n1), pmax, r, p, q, pdot);
  v_scale(c, S.npo, -(P->viscosity + P->grad_div * P->rho), tmp);
  mark();
  // CG for Sm (:86-112)
  const bool lowp_all = o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_F32 || o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_MF || o->ainv_kind == IFEM_AINV_MG;
  sm_ensure(S);
  OpFn sm = [&](const double *x, double *y) { sm_apply(S, x, y, lowp_all); };
  // multigrid-preconditioned CG when coarser levels are attached (every level needs its S_m explicitly: Jacobi smoothing)
  // (the finest level may apply S_m as two SpMVs -- several ranks without the 2-deep pressure halo, e.g. the strips of an unstructured
  // mesh above replicated coarse levels: the V-cycle needs the operator and its diagonal there, not the matrix)
  bool use_mg = o->sm_mg && c->mg_coarse && (sm_is_explicit(S) || (o->explicit_schur && c->halo.nranks > 1));
  MgSm M;
  if (use_mg) {
    M.lowp = lowp_all; M.nu = std::max(1, o->mg_smooth); M.ratio = std::max(1.5, o->mg_cheb_ratio);
    M.L.push_back(S);
    for (ifem_ctx *cc = c->mg_coarse; cc; cc = cc->mg_coarse) {
      SolveState Sc{cc, P, o};
      carve_workspace(Sc, false);
      if (!sm_is_explicit(Sc)) { use_mg = false; break; }
      M.L.push_back(Sc);
    }
  }
  if (use_mg) {
    mg_sm_setup(M, c->asm_constraint_set);
    S.st.cg_sm_iters += pcg_mg_sm(M, src1, dst1, std::max(o->sm_abs, o->sm_rel * n1), pmax, S.tp[1], S.tp[2], S.tp[3], S.tp[4]);
    S.st.sm_mg_levels = (uint32_t)M.L.size();
  } else if (dev_cg) // (Jacobi on S_m was measured too: 176 instead of 172 iterations -- its diagonal is nearly constant)
    S.st.cg_sm_iters += cg_device(c, S.npo, sm, nullptr, src1, dst1, std::max(o->sm_abs, o->sm_rel * n1), pmax, S.tp[1], S.tp[2],
                                  S.tp[3], S.tp[4], 4, S.tp[6]);
  else
    S.st.cg_sm_iters += cg(c, S.npo, sm, src1, dst1, std::max(o->sm_abs, o->sm_rel * n1), pmax, r, p, q, pdot);
  v_axpby(c, S.npo, 1.0, tmp, -P->rho / P->dt, dst1);
  // utmp = src0 - B^T dst1 (:116-120)
  {
    const double *xe; extend_p(S, dst1, &xe);
    spmv_bt(c, xe, S.utmp);
    v_axpby(c, S.nuo, 1.0, src0, -1.0, S.utmp);
  }
  mark();
  // A~^-1 utmp (:124-127)
  const bool f32 = o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_F32;
  if (!c->uu_is_stored && o->ainv_kind != IFEM_AINV_GMRES_BJACOBI_MF && o->ainv_kind != IFEM_AINV_MG)
    throw Error(IFEM_E_BADPARAM, "ifem_tuning::stored_uu = 0 keeps no A_uu values: use IFEM_AINV_MG or IFEM_AINV_GMRES_BJACOBI_MF (the matrix-free inner operators)");
  OpFn Auu = [&](const double *x, double *y) { const double *xe; extend_u(S, x, &xe); spmv_uu(c, xe, nullptr, y, f32); };
  if (o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_MF)
    Auu = [&](const double *x, double *y) { uu_apply_level(S, x, y); };
  const bool scalar_op = o->ainv_kind == IFEM_AINV_SCALAR_GMRES;
  if (scalar_op) shat_refresh(c, true);
  if (scalar_op) Auu = [&](const double *x, double *y) { const double *xe; extend_u(S, x, &xe); spmv_shat(c, xe, y, true); };
  OpFn Pj = [&](const double *x, double *y) { if (scalar_op) shat_jacobi(c, x, y); else bjac_apply(c, x, y); };
  auto mdot = [&](int k, const double *V, int64_t ld, const double *w, double *out) {
    v_mdot(c, S.nuo, k, V, ld, w, out, /*all_ranks=*/true);
  };
  double un;
  mdot(1, S.utmp, S.nuo, S.utmp, &un);
  un = std::sqrt(un);
  // the first application of a solve may ask for a tighter inner solve (ifem_solver_opts::inner_rel_first)
  // ... when the residual it is applied to is velocity-dominated.  The block-triangular preconditioner leaves O(0.1) of the
  // pressure part of a residual behind per outer iteration (the pressure Schur complement is only approximated, mpi_insim.cpp:
  // 57-112), so a first Krylov vector whose pressure share exceeds ~10 fgmres_rel cannot be finished in one iteration by
  // a better velocity solve -- the later Newton iterations, whose residual is almost all continuity equation (shares
  // 0.996 / 0.66 against 7e-5 in the first one at 128^3): there the cheap setting is the better one (time_step leg of
  // bench.py: 1.16 s against 1.27 s with the tight first application everywhere)
  const double pshare_max = o->inner_first_pshare > 0 ? o->inner_first_pshare : 10.0 * o->fgmres_rel;
  const bool pressure_dominated = S.p_src_norm > pshare_max * std::hypot(S.u_src_norm, S.p_src_norm);
  // ... and while it pays: a solve whose tight first application did NOT end the outer iteration at its first check has spent
  // the extra inner iterations for nothing (64^3 channel: 2 outer iterations either way, 29.8 instead of 24.5 ms).  After such a
  // miss the context leaves the option off for its next 8 / 16 / 32 / 64 qualifying solves (ins_solve keeps the count), then tries again.
  bool tight = false;
  if (S.st.precond_applies == 0 && o->inner_rel_first > 0 && !pressure_dominated) {
    S.tight_candidate = true;
    tight = S.tight_used = c->tight_first_backoff == 0;
  }
  const double inner_rel_now = tight ? o->inner_rel_first : o->inner_rel;
  if (o->verbose && S.st.precond_applies == 0)
    fprintf(stderr, "[ifem] first preconditioner application: pressure share of the residual %.3e, inner tolerance %.1e\n",
            S.p_src_norm / std::max(std::hypot(S.u_src_norm, S.p_src_norm), 1e-300), inner_rel_now);
  double res = 0;
  if (o->ainv_kind == IFEM_AINV_MG) { // inner GMRES on the matrix-free operator, one V-cycle as its preconditioner
    MgUu Mu;
    Mu.nu = std::max(1, o->mg_smooth_u); Mu.nu_post = o->mg_smooth_u_post > 0 ? o->mg_smooth_u_post : Mu.nu; Mu.ratio = std::max(1.5, o->mg_cheb_ratio_u);
    Mu.L.push_back(S);
    for (ifem_ctx *p = c; p->mg_coarse && p->mg_Pu.n_rows == p->nUo && p->nUo > 0; p = p->mg_coarse) {
      SolveState Sc{p->mg_coarse, P, o};
      carve_workspace(Sc, false);
      Mu.L.push_back(Sc);
    }
    if (!c->mf_valid) throw Error(IFEM_E_BADPARAM, "IFEM_AINV_MG needs the operator state of ifem_ins_assemble / ifem_imex_assemble");
    mg_uu_setup(Mu);
    OpFn Amf = [&](const double *x, double *y) { uu_apply_level(S, x, y); };
    // the cycle itself: eagerly, or as a captured hipGraph (ctx.hpp::VcGraph) on small single-rank chains
    bool graph_ok = c->tune.vcycle_graph_cells > 0 && c->n_cells <= c->tune.vcycle_graph_cells && !c->profile && !kprof_root(c).on && !profiler_attached();
    for (const SolveState &L : Mu.L) graph_ok = graph_ok && L.ctx->halo.nranks == 1 && !L.ctx->mg_replica && !L.ctx->profile;
    auto run_vcycle = [&]() {
      if (!graph_ok) { mg_uu_vcycle(Mu, 0); return; }
      std::vector<uint64_t> key;
      auto put = [&](const void *ptr) { key_ptr(key, ptr); };
      auto putd = [&](double v) { key_f64(key, v); };
      key.push_back(Mu.L.size()); key.push_back(uint64_t(Mu.nu)); key.push_back(uint64_t(Mu.nu_post)); putd(Mu.ratio);
      for (const SolveState &L : Mu.L) {
        ifem_ctx *lc = L.ctx;
        (void)bjac_f32_ptr(lc); // lazy state (the single-precision copy of the inverse node blocks) stays outside the graph
        for (auto &v : lc->mguf_vec) put(v.p);
        put(lc->bjac_f32.p); put(lc->mf_eval.p); put(lc->mf_ycell.p); put(lc->mg_Ru_mask.p); put(lc->mg_Pu_mask.p);
        put(lc->has_c[lc->asm_constraint_set] ? lc->is_c[lc->asm_constraint_set].p : nullptr);
        key.push_back(uint64_t(lc->nUo)); key.push_back(uint64_t(lc->n_cells)); key.push_back(uint64_t(lc->mf_noconv)); key.push_back(uint64_t(lc->tune.xcd_swizzle));
        // everything else the captured launches are made of: the epoch moves with ifem_set_tuning / ifem_set_profiling / ifem_mg_attach
        key.push_back(lc->graph_epoch); key.push_back(uint64_t(lc->tune.mf_f32));
        put(lc->bjac.p); put(lc->vcoords.p); put(lc->cell_unodes.p); put(lc->uinc.col.p);
        putd(lc->uu_lmax); putd(lc->mf_params.viscosity); putd(lc->mf_params.rho); putd(lc->mf_params.grad_div); putd(lc->mf_params.dt);
      }
      if (!graph_run(c, c->vc_graph, key, [&]() { mg_uu_vcycle(Mu, 0); })) {
        c->tune.vcycle_graph_cells = 0;
        graph_ok = false;
        if (o->verbose) fprintf(stderr, "[ifem] hipGraph capture of the A_uu V-cycle failed: eager launches from now on\n");
      }
    };
    OpFn Vc = [&](const double *x, double *y) {
      v_cvt_d2f(c, S.nuo, x, c->mguf_vec[0].p);
      run_vcycle();
      v_cvt_f2d(c, S.nuo, c->mguf_vec[1].p, y);
    };
    // one attempt of A~^-1 with the V-cycle; returns false when the result is not finite (a Chebyshev bound below the
    // spectral radius turns the smoothers into amplifiers)
    auto attempt = [&]() -> bool {
      if (o->inner_maxit == 0) { // A~^-1 := one V-cycle, no inner Krylov iteration (the outer solver is flexible)
        Vc(S.utmp, dst0);
        S.st.inner_iters += 1;
      } else if (o->inner_maxit < 0) { // -k: k stationary V-cycle sweeps x += V(b - A x): no Arnoldi process, k - 1 operator products
        const int k = -o->inner_maxit;
        Vc(S.utmp, dst0);
        for (int it = 1; it < k; ++it) {
          Amf(dst0, S.inner_w);
          v_axpby(c, S.nuo, 1.0, S.utmp, -1.0, S.inner_w); // r = b - A x
          Vc(S.inner_w, S.inner_z);
          v_axpy(c, S.nuo, 1.0, S.inner_z, dst0);
        }
        S.st.inner_iters += k;
      } else { // flexible GMRES: the preconditioned directions are kept, so the update needs no extra V-cycle
        const int64_t ld = basis_ld(S.ctx, S.nuo);
        // restart length: the caller's, lengthened by the context when an application needed more than two restart cycles -- a V-cycle
        // that has lost its mesh independence (the refined cylinder, DESIGN section 6: 147 inner iterations with GMRES(16), 75 with
        // GMRES(40), 59 with GMRES(100)) stagnates across restarts; the bases grow on demand, so the longer cycle costs memory
        // only where it is used.  Capped at 128 columns and at a quarter of the free device memory.
        const int mi = std::max(std::max(1, o->inner_restart), c->inner_restart_eff);
        grow_basis(c, c->innerV, ld, std::min(mi + 1, kBasisStart), mi + 1);
        grow_basis(c, c->innerZ, ld, std::min(mi, kBasisStart), mi);
        const auto grow = basis_grower(c, c->innerV, c->innerZ, ld, mi + 1);
        const int its = gmres(c, S.nuo, ld, /*reorth=*/false, Amf, Vc, true, S.utmp, dst0, mi, o->inner_maxit, inner_rel_now * un,
                              c->innerV.p, c->innerZ.p, S.inner_w, &res, mdot, nullptr, &grow);
        S.st.inner_iters += its;
        if (its > 2 * mi && mi < 128) {
          size_t fr = 0, tot = 0;
          (void)hipMemGetInfo(&fr, &tot);
          int want = std::min(128, 2 * mi);
          const double per_col = 2.0 * double(ld) * sizeof(double);
          int fits = int(std::min<double>(128.0, 0.25 * double(fr) / std::max(per_col, 1.0)));
          if (c->test_restart_fits > 0) fits = c->test_restart_fits; // test aid (ifem_test_restart_fits): a rank that is short of memory
          want = std::min(want, std::max(fits, mi));
          // `its` and `mi` are the same on every rank, free memory and `ld` are not: the restart length must be (the ranks restart
          // together -- their all-reduces and halo exchanges pair up), so the smallest wish of all ranks wins
          if (c->halo.nranks > 1) { double w = -double(want); allreduce_max(c, &w, 1); want = int(-w); }
          if (want > mi) {
            if (o->verbose) fprintf(stderr, "[ifem] inner GMRES(%d) needed %d iterations: restart length %d from now on\n", mi, its, want);
            c->inner_restart_eff = want;
          }
        }
      }
      if (o->inner_maxit > 0) return std::isfinite(res); // the Arnoldi recurrence carries any NaN / Inf of the V-cycle
      double dn;
      mdot(1, dst0, S.nuo, dst0, &dn);
      return std::isfinite(dn);
    };
    if (!attempt()) {
      // re-estimate every level's bound from scratch and try once more; if that fails too, this application falls back to
      // the node-block Jacobi preconditioned inner solve (the preconditioner degrades, the outer solver stays correct)
      if (o->verbose) fprintf(stderr, "[ifem] A_uu V-cycle returned non-finite values: re-estimating the Chebyshev bounds\n");
      mg_uu_setup(Mu, /*force_bounds=*/true);
      res = 0;
      if (!attempt()) {
        if (o->verbose) fprintf(stderr, "[ifem] A_uu V-cycle still non-finite: node-block Jacobi for this application\n");
        OpFn Pbj = [&](const double *x, double *y) { bjac_apply(c, x, y); };
        res = 0;
        const int64_t ldj = basis_ld(S.ctx, S.nuo);
        const int mj = std::max(1, o->inner_restart);
        const std::function<void(int, double *&, double *&)> growv = [&](int cols, double *&V, double *&) {
          grow_basis(c, c->innerV, ldj, std::min(cols, mj + 1), mj + 1); // (not flexible: one z vector)
          V = c->innerV.p;
        };
        S.st.inner_iters += gmres(c, S.nuo, ldj, /*reorth=*/false, Amf, Pbj, false, S.utmp, dst0, mj,
                                  std::max(o->inner_maxit, 50), inner_rel_now * un, c->innerV.p, S.inner_z, S.inner_w, &res, mdot, nullptr, &growv);
      }
    }
    mark();
    S.st.precond_applies++;
    return;
  }
  // inner_maxit <= 0 means "one V-cycle / stationary sweeps" to IFEM_AINV_MG only; for the Krylov kinds it is the library default cap
  const int inner_cap = o->inner_maxit > 0 ? o->inner_maxit : 400;
  const bool f32_basis = (f32 || o->ainv_kind == IFEM_AINV_GMRES_BJACOBI_MF) && o->inner_restart + 6 <= 64;
  if (f32_basis) { // columns 0..m: basis, m+1: scratch for V y, up to the next multiple of 4: padding read by the fused kernels
    OpF32 Pf = [&](const float *x, double *y) { bjac_apply_f32(c, x, y); };
    S.st.inner_iters += gmres_f32basis(c, S.nuo, basis_ld(S.ctx, S.nuo), Auu, Pf, S.utmp, dst0, o->inner_restart, inner_cap,
                                       inner_rel_now * un, reinterpret_cast<float *>(c->innerV.p), S.inner_z, S.inner_w, &res,
                                       [&](double *v, int k) { allreduce_sum(c, v, k); });
  } else
  S.st.inner_iters += gmres(c, S.nuo, basis_ld(S.ctx, S.nuo), /*reorth=*/false, Auu, Pj, false, S.utmp, dst0, o->inner_restart,
                            inner_cap, inner_rel_now * un, c->innerV.p, S.inner_z, S.inner_w, &res, mdot);
  mark();
  S.st.precond_applies++;
}

// device time of the three sections of every preconditioner application since pc_used was reset; the stream must be idle
static void precond_section_times(ifem_ctx *c, ifem_solve_stats &st) {
  for (size_t k = 0; k + 3 < c->pc_used; k += 4) {
    float a = 0, b = 0, d = 0;
    if (hipEventElapsedTime(&a, c->pc_ev[k], c->pc_ev[k + 1]) == hipSuccess && hipEventElapsedTime(&b, c->pc_ev[k + 1], c->pc_ev[k + 2]) == hipSuccess &&
        hipEventElapsedTime(&d, c->pc_ev[k + 2], c->pc_ev[k + 3]) == hipSuccess) {
      st.t_cg_mp_ms += a; st.t_cg_sm_ms += b; st.t_ainv_ms += d;
    }
  }
  c->pc_used = 0;
}

// The inner Krylov basis is shared by every solver of the context.  The single-precision-basis kernels read (with zero
// coefficients) up to 3 columns past the ones in use, so the buffer is zero-filled whenever it is (re)allocated: 0 * NaN
// from uninitialised memory would poison the preconditioner.
static void grow_inner_basis(ifem_ctx *c, int64_t need) {
  if ((int64_t)c->innerV.n >= need) return;
  c->innerV.alloc((size_t)need);
  IFEM_HIP_CHECK(hipMemsetAsync(c->innerV.p, 0, c->innerV.n * sizeof(double), c->stream));
}

static void carve_workspace(SolveState &S, bool krylov) {
  ifem_ctx *c = S.ctx;
  const int64_t nul = c->dim * c->nUl, npl = c->nPl;
  S.nuo = c->dim * c->nUo; S.npo = c->nPo; S.n = S.nuo + S.npo;
  const int64_t need = nul + npl + 4 * nul + 7 * npl + S.n + 64;
  if ((int64_t)c->work.n < need) c->work.alloc(need);
  double *p = c->work.p;
  S.xu_ext = p; p += nul;
  S.xp_ext = p; p += npl;
  S.tu = p; p += nul;
  S.utmp = p; p += nul;
  S.inner_w = p; p += nul;
  S.inner_z = p; p += nul;
  for (int i = 0; i < 7; ++i) { S.tp[i] = p; p += npl; }
  S.outer_w = p;
  if (!krylov) return; // a coarse multigrid level: vector scratch only
  const int m = S.o->fgmres_restart, mi = S.o->inner_restart;
  grow_basis(c, c->krylovV, basis_ld(S.ctx, S.n), std::min(m + 1, kBasisStart), m + 1); // the rest on demand (basis_grower)
  grow_basis(c, c->krylovZ, basis_ld(S.ctx, S.n), std::min(m, kBasisStart), m);
  // the inner solve of IFEM_AINV_MG is flexible too and grows its bases the same way; the other kinds' kernels want theirs whole
  grow_inner_basis(c, (int64_t)(S.o->ainv_kind == IFEM_AINV_MG ? std::min(mi + 1, kBasisStart) : mi + 1) * basis_ld(S.ctx, S.nuo));
}

void ins_precond_vmult(ifem_ctx *ctx, const ifem_ins_params *P, const ifem_solver_opts *o, const double *src, double *dst) {
  SolveState S{ctx, P, o};
  carve_workspace(S);
  ctx->pc_used = 0;
  precond_vmult(S, src, dst);
  ctx->pc_used = 0;
}

// right-hand side of the condensed system after an assembly that treated the hanging dofs as ordinary ones:
// b = C^T (b^ - A^ c0) on the regular rows, d_h c0_h on the hanging rows (distribute_local_to_global semantics)
void hanging_condense_rhs(ifem_ctx *ctx, int use_nonzero) {
  if (!ctx->hang.active) return;
  ifem_solver_opts o;
  ifem_default_solver_opts(&o);
  SolveState S{ctx, nullptr, &o};
  carve_workspace(S);
  hanging_refresh_diag(ctx);
  double *rhs = ctx->vec[IFEM_VEC_RHS].p;
  const bool inhom = hanging_offset(ctx, use_nonzero); // c0: ghost-extended, zero away from the hanging entries
  if (inhom) {
    system_apply_ext(S, ctx->hang.c0.p, ctx->hang.c0.p + int64_t(ctx->dim) * ctx->nUl, S.outer_w, false);
    v_axpy(ctx, S.n, -1.0, S.outer_w, rhs);
  }
  hanging_output(ctx, ctx->hang.c0.p, true, rhs);
}

void ins_system_vmult(ifem_ctx *ctx, const double *src, double *dst) {
  ifem_solver_opts o;
  ifem_default_solver_opts(&o);
  SolveState S{ctx, nullptr, &o};
  carve_workspace(S);
  system_apply(S, src, dst, false);
}

// SUPGFluidSolver::solve + BlockIncompSchurPreconditioner::vmult (mpi_supg_solver.cpp:35-192, 297-328).
//   P_vv^-1  : node-block Jacobi of A_vv            (reference: Hypre-Euclid ILU(0))
//   T_pp     : A_pp - A_pv P_vv^-1 A_vp, solved by GMRES(200) to 1e-3 ||.|| with the ILU(0) of the explicit T_pp
//                                                   (reference: ILU(0) of B2pp = A_pp - A_pv rowsum|A_vv|^-1 A_vp)
// what deal.II's SolverControl::NoConvergence carries: the last step and the last residual
static void throw_noconv(const char *solver, int it, double res, double tol) {
  char msg[256];
  snprintf(msg, sizeof msg, "%s did not converge (SolverControl::NoConvergence): residual %.6e after %d iteration%s, tolerance %.6e%s", solver, res, it,
           it == 1 ? "" : "s", tol, std::isfinite(res) ? "" : " -- the system holds non-finite values (check the state vectors)");
  throw Error(IFEM_E_KRYLOV_NOCONV, msg);
}

// constructor of BlockIncompSchurPreconditioner (mpi_supg_solver.cpp:35-134) in the reference's structure: ILU(0)(A_vv), B2pp and its
// ILU(0); cached until the next assembly
void scns_refpc_setup(ifem_ctx *ctx, int verbose, bool *pvv_ok, bool *b2_ok) {
  BIlu &Iv = ctx->pvv_ilu, &Ip = ctx->b2_ilu;
#if !IFEM_UU_INTERLEAVED
  throw Error(IFEM_E_BADPARAM, "scns_pc = 2 needs the block-interleaved A_uu layout");
#endif
  if (!Iv.analysed) bilu_analyse(ctx, Iv, ctx->dim, ctx->nUo, ctx->Auu.rowptr.p, ctx->Auu.col.p);
  *pvv_ok = Iv.factored ? !Iv.broken : bilu_factor(ctx, Iv, ctx->Auu.val.p);
  if (!*pvv_ok && verbose) fprintf(stderr, "[ifem] scns solve: ILU(0) of A_vv broke down: node-block Jacobi instead\n");
  const PlanarCsr &Pt = tpp_pattern(ctx);
  if (!ctx->b2_valid) {
    const size_t nb = (size_t)ctx->nUo * ctx->dim * ctx->dim;
    if (ctx->rsinv.n != nb) ctx->rsinv.alloc(nb);
    if (ctx->B2pp.n != (size_t)Pt.nnzb) ctx->B2pp.alloc((size_t)Pt.nnzb);
    rowsum_abs_inv(ctx, ctx->rsinv.p);
    schur_pp_numeric(ctx, ctx->rsinv.p, ctx->B2pp.p);
    ctx->b2_valid = true; Ip.factored = false;
  }
  if (!Ip.analysed) bilu_analyse(ctx, Ip, 1, Pt.n_rows, Pt.rowptr.p, Pt.col.p);
  *b2_ok = Ip.factored ? !Ip.broken : bilu_factor(ctx, Ip, ctx->B2pp.p);
  if (!*b2_ok && verbose) fprintf(stderr, "[ifem] scns solve: ILU(0) of B2pp broke down: Jacobi instead\n");
}

// test hook (ifem_scns_pc_probe, single rank): the pieces of the preconditioner on device vectors
void scns_pc_probe(ifem_ctx *ctx, int which, const double *x, double *y) {
  ifem_solver_opts o;
  ifem_default_solver_opts(&o);
  SolveState S{ctx, nullptr, &o};
  carve_workspace(S);
  bool pvv_ok = false, b2_ok = false;
  scns_refpc_setup(ctx, 0, &pvv_ok, &b2_ok);
  if (!pvv_ok || !b2_ok) throw Error(IFEM_E_KRYLOV_NOCONV, "ILU(0) broke down");
  switch (which) {
  case 0: bilu_apply(ctx, ctx->pvv_ilu, ctx->tune.pvv_sweeps, x, y); break;
  case 1: bilu_apply(ctx, ctx->b2_ilu, ctx->tune.b2pp_sweeps, x, y); break;
  case 2: spmv_planar_scalar(ctx, tpp_pattern(ctx), ctx->B2pp.p, x, y); break;
  case 3: {
    spmv_bt(ctx, x, S.tu);
    bilu_apply(ctx, ctx->pvv_ilu, ctx->tune.pvv_sweeps, S.tu, S.utmp);
    spmv_b(ctx, S.utmp, S.tp[4]);
    spmv_app(ctx, x, y);
    v_axpy(ctx, S.npo, -1.0, S.tp[4], y);
    break;
  }
  default: throw Error(IFEM_E_BADPARAM, "ifem_scns_pc_probe: which must be 0..3");
  }
}

int scns_solve(ifem_ctx *ctx, const ifem_solver_opts *o, int use_nonzero, ifem_solve_stats *stats) {
  if (!ctx->assembled || !ctx->has_app) throw Error(IFEM_E_BADPARAM, "ifem_scns_solve called before ifem_scns_assemble");
  SolveState S{ctx, nullptr, o};
  carve_workspace(S);
  Clock total;
  app_diag_setup(ctx);
  const int mt = 200;
  grow_inner_basis(ctx, (int64_t)(mt + 1) * basis_ld(S.ctx, S.npo));
  double *rhs = ctx->vec[IFEM_VEC_RHS].p, *upd = ctx->vec[IFEM_VEC_UPDATE].p;
  auto mdot = [&](int k, const double *V, int64_t ld, const double *w, double *out) {
    v_mdot(ctx, S.n, k, V, ld, w, out, /*all_ranks=*/true);
  };
  auto mdot_p = [&](int k, const double *V, int64_t ld, const double *w, double *out) {
    v_mdot(ctx, S.npo, k, V, ld, w, out, /*all_ranks=*/true);
  };
  double bn;
  mdot(1, rhs, S.n, rhs, &bn);
  bn = std::sqrt(bn);
  const double tol = 1e-6 * bn; // mpi_supg_solver.cpp:311-312
  const int64_t n_glob = ctx->n_global_u + ctx->n_global_p; // identical on all ranks
  const int maxit = o->fgmres_maxit > 0 ? o->fgmres_maxit : (int)std::min<int64_t>(n_glob, 1 << 30);
  OpFn Aop = [&](const double *x, double *y) { system_apply(S, x, y, false); };
  // y_u = A_vp x_p and y_p = A_pv x_u on compact owned vectors (ghosts refreshed first on several ranks)
  auto bt_apply = [&](const double *xp, double *yu) { const double *xe; extend_p(S, xp, &xe); spmv_bt(ctx, xe, yu); };
  auto b_apply = [&](const double *xu, double *yp) { const double *xe; extend_u(S, xu, &xe); spmv_b(ctx, xe, yp); };
  OpFn Tpp = [&](const double *x, double *y) {
    const double *xe; extend_p(S, x, &xe);
    spmv_bt(ctx, xe, S.tu);
    bjac_apply(ctx, S.tu, S.utmp);
    b_apply(S.utmp, S.tp[4]);
    spmv_app(ctx, xe, y);
    v_axpy(ctx, S.npo, -1.0, S.tp[4], y);
  };
  OpFn Jpp = [&](const double *x, double *y) { vec_div(ctx, S.npo, ctx->app_diag.p, x, y); };
  // one rank: T_pp as an explicit matrix (tpp.hip): one SpMV per inner iteration, and the inner GMRES preconditioned by the
  // ILU(0) of that matrix (the reference: Euclid ILU(0) of B2pp, mpi_supg_solver.cpp:141-192, preconditioner_pilut.cpp:124-138),
  // factorised once per Newton iteration and applied by level-scheduled triangular solves.  Several ranks: the operator stays
  // distributed, the preconditioner is the ILU(0) of the owned x owned block of T_pp on every rank (block-Jacobi ILU: what
  // Euclid does across ranks, mpi_supg_solver.cpp:49-53,120-133).  ifem_tuning::tpp_operator keeps the operator form with
  // Jacobi, tpp_ilu_order = -1 the explicit matrix with Jacobi.  A factorisation that breaks down (zero / tiny / non-finite
  // pivot) is not applied: Jacobi instead.
  // ifem_tuning::scns_pc = 2 (default): the reference's structure.  P_vv^-1 = ILU(0) of A_vv (mpi_supg_solver.cpp:49-51), T_pp the
  // operator A_pp - A_pv P_vv^-1 A_vp (:19-32), the inner GMRES(200) preconditioned by the ILU(0) of the assembled
  // B2pp = A_pp - A_pv rowsum(|A_vv|)^-1 A_vp (:56-133); both factorisations once per Newton iteration, applied by Jacobi sweeps on
  // the triangular systems (bilu.hip).  A factorisation that breaks down falls back to (node-block) Jacobi.
  const bool refpc = ctx->tune.scns_pc == 2;
  bool pvv_ok = false;
  OpFn Pvv = [&](const double *x, double *y) {
    if (pvv_ok) bilu_apply(ctx, ctx->pvv_ilu, ctx->tune.pvv_sweeps, x, y);
    else bjac_apply(ctx, x, y);
  };
  bool b2_ok = false;
  if (refpc) {
    scns_refpc_setup(ctx, o->verbose, &pvv_ok, &b2_ok);
    Tpp = [&](const double *x, double *y) { // SchurComplementTpp::vmult
      const double *xe; extend_p(S, x, &xe);
      spmv_bt(ctx, xe, S.tu);
      Pvv(S.tu, S.utmp);
      b_apply(S.utmp, S.tp[4]);
      spmv_app(ctx, xe, y);
      v_axpy(ctx, S.npo, -1.0, S.tp[4], y);
    };
    if (b2_ok) Jpp = [&](const double *x, double *y) { bilu_apply(ctx, ctx->b2_ilu, ctx->tune.b2pp_sweeps, x, y); };
  }
  // w = B2pp_inverse (T_pp stage) of the inner iteration: ~20 short launches with constant arguments (the Jacobi sweeps of the two ILU(0)
  // applications, three SpMVs) whose cost is the host's launch rate -- one captured hipGraph on a single rank (ifem_tuning::scns_graph)
  bool pa_graph_ok = refpc && ctx->tune.scns_graph != 0 && ctx->halo.nranks == 1 && !ctx->profile && !ctx->kprof.on && !profiler_attached();
  OpFn PAg = [&](const double *x, double *y) {
    auto body = [&]() { Tpp(x, S.tp[1]); Jpp(S.tp[1], y); };
    if (!pa_graph_ok) { body(); return; }
    std::vector<uint64_t> key;
    for (const void *p : {(const void *)x, (const void *)y, (const void *)S.tp[1], (const void *)S.tp[4], (const void *)S.tu, (const void *)S.utmp,
                          (const void *)ctx->pvv_ilu.LU.p, (const void *)ctx->pvv_ilu.t0.p, (const void *)ctx->b2_ilu.LU.p, (const void *)ctx->b2_ilu.t0.p,
                          (const void *)ctx->App.p, (const void *)ctx->B.val.p, (const void *)ctx->Bt.val.p})
      key_ptr(key, p);
    key.push_back(uint64_t(ctx->tune.pvv_sweeps)); key.push_back(uint64_t(ctx->tune.b2pp_sweeps)); key.push_back(uint64_t(pvv_ok)); key.push_back(uint64_t(b2_ok));
    key.push_back(uint64_t(ctx->pvv_ilu.nnz)); key.push_back(uint64_t(ctx->b2_ilu.nnz));
    if (!graph_run(ctx, ctx->pa_graph, key, body)) pa_graph_ok = false;
  };
  const bool tpp_explicit = !refpc && ctx->halo.nranks == 1 && !ctx->tune.tpp_operator;
  auto ilu_or_warn = [&]() {
    const bool ok = tpp_ilu_factor(ctx);
    if (!ok && o->verbose)
      fprintf(stderr, "[ifem] scns solve: ILU(0) of T_pp broke down (|pivot| in [%.3e, %.3e]): Jacobi instead\n", ctx->tpp_ilu.pivot_min,
              ctx->tpp_ilu.pivot_max);
    return ok;
  };
  if (tpp_explicit) {
    tpp_numeric(ctx);
    Tpp = [&](const double *x, double *y) { spmv_tpp(ctx, x, y); };
    if (ctx->tune.tpp_ilu_order >= 0 && ilu_or_warn())
      Jpp = [&](const double *x, double *y) { tpp_ilu_apply(ctx, x, y); };
    else
      Jpp = [&](const double *x, double *y) { vec_div(ctx, S.npo, ctx->tpp_diag.p, x, y); };
  } else if (!refpc && ctx->halo.nranks > 1 && !ctx->tune.tpp_operator && ctx->tune.tpp_ilu_order >= 0) {
    tpp_numeric(ctx); // the owned x owned block
    if (ilu_or_warn()) Jpp = [&](const double *x, double *y) { tpp_ilu_apply(ctx, x, y); };
  }
  OpFn Pop = [&](const double *src, double *dst) {
    const double *src0 = src, *src1 = src + S.nuo;
    double *dst0 = dst, *dst1 = dst + S.nuo;
    Pvv(src0, S.inner_w);                          // ptmp1 = P_vv^-1 src0
    b_apply(S.inner_w, S.tp[0]);                   // A_pv ptmp1
    v_axpby(ctx, S.npo, 1.0, src1, -1.0, S.tp[0]); // ptmp = src1 - A_pv ptmp1
    double pn;
    mdot_p(1, S.tp[0], S.npo, S.tp[0], &pn);
    double res = 0;
    const double inner_tol = 1e-3 * std::sqrt(pn);
    if (refpc && pn > 0) {
      // initial guess alpha ptmp with alpha = (ptmp . ptmp) / (T_pp ptmp . ptmp) (:165-171); the Krylov solve runs on the correction
      // with the same ABSOLUTE tolerance 1e-3 ||ptmp|| (:174-175)
      Tpp(S.tp[0], S.tp[6]);
      double sc;
      mdot_p(1, S.tp[6], S.npo, S.tp[0], &sc);
      const double alpha = sc != 0 && std::isfinite(sc) ? pn / sc : 0.0;
      v_axpby(ctx, S.npo, 1.0, S.tp[0], -alpha, S.tp[6]); // r0 = ptmp - alpha T_pp ptmp
      const bool left = ctx->tune.scns_inner_left != 0;
      S.st.inner_iters += gmres(ctx, S.npo, basis_ld(S.ctx, S.npo), ctx->tune.scns_inner_reorth != 0, Tpp, Jpp, false, S.tp[6], dst1, mt, 100000,
                                inner_tol, ctx->innerV.p, S.tp[1], S.tp[2], &res, mdot_p, nullptr, nullptr, left, left ? S.tp[5] : nullptr,
                                left && pa_graph_ok ? &PAg : nullptr);
      v_axpy(ctx, S.npo, alpha, S.tp[0], dst1);
    } else
      S.st.inner_iters += gmres(ctx, S.npo, basis_ld(S.ctx, S.npo), /*reorth=*/true, Tpp, Jpp, false, S.tp[0], dst1, mt, 100000, inner_tol,
                                ctx->innerV.p, S.tp[1], S.tp[2], &res, mdot_p);
    bt_apply(dst1, S.tu);                          // A_vp dst1
    Pvv(S.tu, S.utmp);
    v_copy(ctx, S.nuo, S.inner_w, dst0);
    v_axpy(ctx, S.nuo, -1.0, S.utmp, dst0);        // dst0 = P_vv^-1 src0 - P_vv^-1 A_vp dst1
    S.st.precond_applies++;
  };
  double res = 0;
  const auto grow = basis_grower(ctx, ctx->krylovV, ctx->krylovZ, basis_ld(S.ctx, S.n), o->fgmres_restart + 1);
  const int it = gmres(ctx, S.n, basis_ld(S.ctx, S.n), true, Aop, Pop, true, rhs, upd, o->fgmres_restart, maxit, tol, ctx->krylovV.p,
                       ctx->krylovZ.p, S.outer_w, &res, mdot, nullptr, &grow);
  apply_constraints(ctx, use_nonzero ? 1 : 0, upd);
  hanging_distribute(ctx, upd);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  S.st.fgmres_iters = it; S.st.fgmres_res = res; S.st.t_total_ms = total.ms();
  if (stats) *stats = S.st;
  if (o->verbose)
    fprintf(stderr, "[ifem] scns solve: fgmres %d its res %.3e (tol %.3e) | inner Tpp its %u | %.1f ms\n", it, res, tol,
            S.st.inner_iters, S.st.t_total_ms);
  if (!(res <= tol)) throw_noconv("FGMRES", it, res, tol);
  return 0;
}

int ins_solve(ifem_ctx *ctx, const ifem_ins_params *P, const ifem_solver_opts *o, int use_nonzero,
              ifem_solve_stats *stats) {
  if (!ctx->assembled) throw Error(IFEM_E_BADPARAM, "ifem_solve called before ifem_ins_assemble");
  SolveState S{ctx, P, o};
  carve_workspace(S);
  Clock total;
  ctx->pc_used = 0;
  ctx->spmv_uu_ms_total = 0;
  ctx->timing.spmv_uu_calls = 0;
  ctx->mf_ms_total = 0;
  ctx->timing.mf_calls = 0;
  double *rhs = ctx->vec[IFEM_VEC_RHS].p, *upd = ctx->vec[IFEM_VEC_UPDATE].p;
  auto mdot = [&](int k, const double *V, int64_t ld, const double *w, double *out) {
    v_mdot(ctx, S.n, k, V, ld, w, out, /*all_ranks=*/true);
  };
  double bn;
  mdot(1, rhs, S.n, rhs, &bn);
  bn = std::sqrt(bn);
  const double tol = std::max(o->fgmres_abs, o->fgmres_rel * bn);
  const int64_t n_glob = ctx->n_global_u + ctx->n_global_p; // SolverControl(system_matrix.m(), ...): identical on all ranks
  const int maxit = o->fgmres_maxit > 0 ? o->fgmres_maxit : (int)std::min<int64_t>(n_glob, 1 << 30);
  OpFn Aop = [&](const double *x, double *y) { system_apply(S, x, y, true); };
  OpFn Pop = [&](const double *x, double *y) { precond_vmult(S, x, y); };
  double res = 0;
  std::vector<double> hist;
  const auto grow = basis_grower(ctx, ctx->krylovV, ctx->krylovZ, basis_ld(S.ctx, S.n), o->fgmres_restart + 1);
  const int it = gmres(ctx, S.n, basis_ld(S.ctx, S.n), /*reorth=*/true, Aop, Pop, true, rhs, upd, o->fgmres_restart, maxit, tol, ctx->krylovV.p, ctx->krylovZ.p,
                       S.outer_w, &res, mdot, o->verbose ? &hist : nullptr, &grow);
  apply_constraints(ctx, use_nonzero ? 1 : 0, upd); // constraints_used.distribute(newton_update)
  hanging_distribute(ctx, upd);
  IFEM_HIP_CHECK(hipStreamSynchronize(ctx->stream));
  // inner_rel_first pays only when it ends the outer iteration at its first check (precond_vmult); `it` is the same on all ranks
  if (S.tight_used) {
    if (it <= 1) ctx->tight_first_misses = 0;
    else { ctx->tight_first_misses = std::min(ctx->tight_first_misses + 1, 4); ctx->tight_first_backoff = 4 << ctx->tight_first_misses; }
  } else if (S.tight_candidate && ctx->tight_first_backoff > 0)
    ctx->tight_first_backoff--;
  S.st.inner_first_tight = S.tight_used ? 1u : 0u;
  precond_section_times(ctx, S.st); // (the stream was synchronised above)
  S.st.fgmres_iters = it;
  S.st.fgmres_res = res;
  S.st.t_total_ms = total.ms();
  S.st.t_spmv_ms = ctx->spmv_uu_ms_total;
  ctx->timing.mf_ms_avg = ctx->timing.mf_calls ? ctx->mf_ms_total / ctx->timing.mf_calls : 0;
  ctx->timing.spmv_uu_ms_avg = ctx->timing.spmv_uu_calls ? ctx->spmv_uu_ms_total / ctx->timing.spmv_uu_calls : 0;
  if (stats) *stats = S.st;
  if (o->verbose)
    fprintf(stderr, "[ifem] solve: fgmres %d its res %.3e (tol %.3e) | P applies %u CG(Mp) %u (%.1f ms) CG(Sm) %u (%.1f ms) inner %u (%.1f ms) | %.1f ms\n",
            it, res, tol, S.st.precond_applies, S.st.cg_mp_iters, S.st.t_cg_mp_ms, S.st.cg_sm_iters, S.st.t_cg_sm_ms, S.st.inner_iters, S.st.t_ainv_ms, S.st.t_total_ms);
  if (o->verbose) {
    fprintf(stderr, "[ifem] solve: ||rhs|| %.3e, relative residual per FGMRES iteration:", bn);
    for (double r : hist) fprintf(stderr, " %.2e", r / bn);
    fprintf(stderr, "\n");
  }
  if (!(res <= tol)) throw_noconv("FGMRES", it, res, tol);
  return 0;
}

} // namespace ifem
