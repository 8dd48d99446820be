#include "multigrid.hpp"
#include <algorithm>
#include <cmath>
#include <stdexcept>
#include <string>

namespace ifem_host {

bool next_coarser_level(int dim, const std::array<int, 3> &n, const std::array<int, 3> &P,
                        const std::array<double, 3> &extent, int min_cells, std::array<int, 3> &out) {
  double h[3], hmin = 0;
  for (int d = 0; d < dim; ++d) {
    h[d] = extent[d] / (double(n[d]) * P[d]);
    hmin = d ? std::min(hmin, h[d]) : h[d];
  }
  bool any = false;
  out = n;
  for (int d = 0; d < dim; ++d) {
    if (h[d] > 1.5 * hmin) continue; // a long direction: waits until the others have caught up
    if (n[d] % 2 != 0 || n[d] / 2 < min_cells) return false; // a direction that should be halved cannot be
    out[d] = n[d] / 2;
    any = true;
  }
  return any;
}

std::vector<std::array<int, 3>> coarse_level_chain(int dim, std::array<int, 3> n, const std::array<int, 3> &P,
                                                   const std::array<double, 3> &extent, int min_cells, int max_levels) {
  std::vector<std::array<int, 3>> chain;
  std::array<int, 3> next;
  while ((int)chain.size() < max_levels && next_coarser_level(dim, n, P, extent, min_cells, next)) {
    chain.push_back(next);
    n = next;
  }
  return chain;
}

namespace {
// 1D nodal interpolation between nested uniform lattices of Q_k nodes (n_fine = n_coarse: identity, or 2 n_coarse):
// fine lattice point i = sum_a w[i][a] * coarse point idx[i][a]; vanishing weights are dropped (cnt[i] entries kept)
struct Interp1D {
  int k = 1;
  std::vector<int64_t> idx; // [N_f][k + 1]
  std::vector<double> w;
  std::vector<int> cnt;
};
Interp1D interp_1d(int n_fine, int n_coarse, int k) {
  Interp1D I;
  I.k = k;
  const int64_t Nf = int64_t(k) * n_fine + 1;
  I.idx.assign((size_t)Nf * (k + 1), 0);
  I.w.assign((size_t)Nf * (k + 1), 0.0);
  I.cnt.assign((size_t)Nf, 0);
  if (n_fine == n_coarse) {
    for (int64_t i = 0; i < Nf; ++i) { I.idx[i * (k + 1)] = i; I.w[i * (k + 1)] = 1.0; I.cnt[i] = 1; }
    return I;
  }
  if (n_fine != 2 * n_coarse) throw std::invalid_argument("multigrid levels must be nested with ratio 1 or 2 per direction");
  for (int64_t i = 0; i < Nf; ++i) {
    const int64_t cell = std::min<int64_t>(i / (2 * k), n_coarse - 1); // coarse cell holding the point (2k fine steps per cell)
    const double t = double(i - 2 * k * cell) / (2.0 * k);             // position in that cell, in [0, 1]
    int c = 0;
    for (int a = 0; a <= k; ++a) { // Lagrange polynomial a on the coarse cell's nodes
      double la = 1.0;
      for (int b = 0; b <= k; ++b)
        if (b != a) la *= (t - double(b) / k) / (double(a) / k - double(b) / k);
      if (std::fabs(la) < 1e-14) continue;
      I.idx[i * (k + 1) + c] = int64_t(k) * cell + a;
      I.w[i * (k + 1) + c] = la;
      ++c;
    }
    I.cnt[i] = c;
  }
  return I;
}

// position lookup of lattice ids inside the bounding box of a node list
struct BoxLookup {
  int64_t N[3] = {1, 1, 1}, lo[3] = {0, 0, 0}, n[3] = {1, 1, 1};
  std::vector<int32_t> pos;
  BoxLookup(int dim, const int64_t *Nlat, const int64_t *l2g, int64_t count) {
    int64_t hi[3] = {0, 0, 0};
    for (int d = 0; d < 3; ++d) { N[d] = d < dim ? Nlat[d] : 1; lo[d] = N[d]; hi[d] = -1; }
    for (int64_t i = 0; i < count; ++i) {
      int64_t r = l2g[i];
      for (int d = 0; d < 3; ++d) { const int64_t c = r % N[d]; r /= N[d]; lo[d] = std::min(lo[d], c); hi[d] = std::max(hi[d], c); }
    }
    if (count == 0) for (int d = 0; d < 3; ++d) { lo[d] = 0; hi[d] = -1; }
    for (int d = 0; d < 3; ++d) n[d] = hi[d] - lo[d] + 1;
    pos.assign((size_t)std::max<int64_t>(n[0] * n[1] * n[2], 0), -1);
    for (int64_t i = 0; i < count; ++i) {
      int64_t r = l2g[i], c[3];
      for (int d = 0; d < 3; ++d) { c[d] = r % N[d] - lo[d]; r /= N[d]; }
      pos[(size_t)((c[2] * n[1] + c[1]) * n[0] + c[0])] = (int32_t)i;
    }
  }
  int32_t find(const int64_t *c) const {
    for (int d = 0; d < 3; ++d) if (c[d] < lo[d] || c[d] >= lo[d] + n[d]) return -1;
    return pos[(size_t)(((c[2] - lo[2]) * n[1] + (c[1] - lo[1])) * n[0] + (c[0] - lo[0]))];
  }
};
} // namespace

void box_prolongation(int dim, const std::array<int, 3> &reps_f, const std::array<int, 3> &reps_c, int degree,
                      const int64_t *l2g_f, int64_t n_f, const int64_t *l2g_c, int64_t n_c, CsrTransfer &P) {
  int64_t Nf[3] = {1, 1, 1}, Nc[3] = {1, 1, 1};
  Interp1D one[3];
  for (int d = 0; d < dim; ++d) {
    Nf[d] = int64_t(degree) * reps_f[d] + 1;
    Nc[d] = int64_t(degree) * reps_c[d] + 1;
    one[d] = interp_1d(reps_f[d], reps_c[d], degree);
  }
  for (int d = dim; d < 3; ++d) one[d] = interp_1d(0, 0, degree); // the single point 0 -> 0
  const BoxLookup look(dim, Nc, l2g_c, n_c);
  const int s = degree + 1;
  P.n_rows = n_f; P.n_cols = n_c;
  P.ptr.assign((size_t)n_f + 1, 0);
  // two passes: count, then fill with the columns of a row in ascending order
  for (int64_t i = 0; i < n_f; ++i) {
    int64_t r = l2g_f[i], c[3];
    for (int d = 0; d < 3; ++d) { c[d] = r % Nf[d]; r /= Nf[d]; }
    P.ptr[(size_t)i + 1] = P.ptr[(size_t)i] + int64_t(one[0].cnt[c[0]]) * one[1].cnt[c[1]] * one[2].cnt[c[2]];
  }
  P.col.resize((size_t)P.ptr[(size_t)n_f]);
  P.w.resize(P.col.size());
  std::vector<std::pair<int32_t, double>> row;
  for (int64_t i = 0; i < n_f; ++i) {
    int64_t r = l2g_f[i], c[3];
    for (int d = 0; d < 3; ++d) { c[d] = r % Nf[d]; r /= Nf[d]; }
    row.clear();
    for (int az = 0; az < one[2].cnt[c[2]]; ++az)
      for (int ay = 0; ay < one[1].cnt[c[1]]; ++ay)
        for (int ax = 0; ax < one[0].cnt[c[0]]; ++ax) {
          const int64_t cc[3] = {one[0].idx[c[0] * s + ax], one[1].idx[c[1] * s + ay], one[2].idx[c[2] * s + az]};
          const int32_t lc = look.find(cc);
          if (lc < 0) throw std::runtime_error("multigrid transfer: a coarse node of the interpolation stencil is not local on this rank");
          row.emplace_back(lc, one[0].w[c[0] * s + ax] * one[1].w[c[1] * s + ay] * one[2].w[c[2] * s + az]);
        }
    std::sort(row.begin(), row.end());
    int64_t k = P.ptr[(size_t)i];
    for (auto &e : row) { P.col[(size_t)k] = e.first; P.w[(size_t)k] = e.second; ++k; }
  }
}

void transpose_transfer(const CsrTransfer &P, CsrTransfer &R) {
  R.n_rows = P.n_cols; R.n_cols = P.n_rows;
  R.ptr.assign((size_t)R.n_rows + 1, 0);
  for (int32_t c : P.col) ++R.ptr[(size_t)c + 1];
  for (int64_t r = 0; r < R.n_rows; ++r) R.ptr[(size_t)r + 1] += R.ptr[(size_t)r];
  R.col.resize(P.col.size());
  R.w.resize(P.w.size());
  std::vector<int64_t> fill(R.ptr.begin(), R.ptr.end() - 1);
  for (int64_t i = 0; i < P.n_rows; ++i) // ascending i: the rows of R come out sorted
    for (int64_t k = P.ptr[(size_t)i]; k < P.ptr[(size_t)i + 1]; ++k) {
      const int64_t q = fill[(size_t)P.col[(size_t)k]]++;
      R.col[(size_t)q] = (int32_t)i;
      R.w[(size_t)q] = P.w[(size_t)k];
    }
}

std::vector<int32_t> box_injection(int dim, const std::array<int, 3> &reps_f, const std::array<int, 3> &reps_c, int degree,
                                   const int64_t *l2g_c, int64_t n_c, const int64_t *l2g_f, int64_t n_f, bool allow_missing) {
  int64_t Nf[3] = {1, 1, 1}, Nc[3] = {1, 1, 1}, ratio[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d) {
    Nf[d] = int64_t(degree) * reps_f[d] + 1;
    Nc[d] = int64_t(degree) * reps_c[d] + 1;
    if (reps_f[d] % reps_c[d] != 0) throw std::invalid_argument("multigrid levels must be nested");
    ratio[d] = reps_f[d] / reps_c[d];
  }
  const BoxLookup look(dim, Nf, l2g_f, n_f);
  std::vector<int32_t> out((size_t)n_c);
  for (int64_t i = 0; i < n_c; ++i) {
    int64_t r = l2g_c[i], c[3];
    for (int d = 0; d < 3; ++d) { c[d] = (r % Nc[d]) * ratio[d]; r /= Nc[d]; }
    const int32_t p = look.find(c);
    if (p < 0 && !allow_missing) throw std::runtime_error("multigrid transfer: the fine node under an owned coarse node is not owned by the same rank");
    out[(size_t)i] = p;
  }
  return out;
}

void nested_prolongation(int dim, int degree, const int32_t *cnf, size_t ncf, int64_t n_fine, const int32_t *cnc, int64_t n_coarse,
                         const std::vector<size_t> &parent, const std::vector<int> &offset, CsrTransfer &P) {
  const int n1 = degree + 1;
  int nn = 1;
  for (int d = 0; d < dim; ++d) nn *= n1;
  // one (cell, local index) per fine node: the first cell that holds it
  std::vector<int64_t> where((size_t)n_fine, -1);
  for (size_t c = 0; c < ncf; ++c)
    for (int a = 0; a < nn; ++a) {
      int64_t &w = where[(size_t)cnf[c * nn + a]];
      if (w < 0) w = int64_t(c) * nn + a;
    }
  P.n_rows = n_fine; P.n_cols = n_coarse;
  P.ptr.assign((size_t)n_fine + 1, 0);
  P.col.clear(); P.w.clear();
  std::vector<std::pair<int32_t, double>> row;
  for (int64_t i = 0; i < n_fine; ++i) {
    if (where[(size_t)i] < 0) throw std::runtime_error("nested_prolongation: a fine node belongs to no cell");
    const size_t c = size_t(where[(size_t)i] / nn);
    const int a = int(where[(size_t)i] % nn);
    double w1[3][4];
    for (int d = 0; d < dim; ++d) {
      const int l = d == 0 ? a % n1 : (d == 1 ? (a / n1) % n1 : a / (n1 * n1));
      const double t = (double((offset[c] >> d) & 1) + double(l) / degree) / 2.0; // position in the parent's reference cell
      for (int q = 0; q <= degree; ++q) {
        double la = 1.0;
        for (int b = 0; b <= degree; ++b)
          if (b != q) la *= (t - double(b) / degree) / (double(q) / degree - double(b) / degree);
        w1[d][q] = la;
      }
    }
    row.clear();
    for (int q = 0; q < nn; ++q) {
      double w = 1.0;
      for (int d = 0; d < dim; ++d) w *= w1[d][d == 0 ? q % n1 : (d == 1 ? (q / n1) % n1 : q / (n1 * n1))];
      if (std::fabs(w) < 1e-14) continue;
      row.emplace_back(cnc[parent[c] * nn + q], w);
    }
    std::sort(row.begin(), row.end());
    for (auto &e : row) { P.col.push_back(e.first); P.w.push_back(e.second); }
    P.ptr[(size_t)i + 1] = (int64_t)P.col.size();
  }
}

std::vector<int32_t> nested_injection(int dim, int degree, const int32_t *cnf, size_t ncf, const int32_t *cnc, size_t ncc,
                                      int64_t n_coarse, const std::vector<size_t> &parent, const std::vector<int> &offset) {
  if (degree != 1 && degree != 2) throw std::invalid_argument("nested_injection: degree 1 or 2");
  const int n1 = degree + 1;
  int nn = 1;
  for (int d = 0; d < dim; ++d) nn *= n1;
  const int nch = 1 << dim;
  std::vector<size_t> child(ncc * nch, size_t(-1));
  for (size_t c = 0; c < ncf; ++c) child[parent[c] * nch + offset[c]] = c;
  std::vector<int32_t> out((size_t)n_coarse, -1);
  for (size_t K = 0; K < ncc; ++K)
    for (int q = 0; q < nn; ++q) {
      const int32_t nd = cnc[K * nn + q];
      if (out[(size_t)nd] >= 0) continue;
      int off = 0, af = 0, stride = 1;
      for (int d = 0; d < dim; ++d) {
        const int l = d == 0 ? q % n1 : (d == 1 ? (q / n1) % n1 : q / (n1 * n1));
        // coarse reference position l / degree in {0, 1/2, 1} (degree 2) or {0, 1} (degree 1): child and position in the child
        const int twice = 2 * l;                       // position in units of 1 / (2 degree)
        const int o = twice >= 2 * degree ? 1 : (twice > degree ? 1 : 0);
        const int lf = twice - o * degree;             // in units of 1 / degree of the child
        off |= o << d;
        af += lf * stride;
        stride *= n1;
      }
      const size_t cf = child[K * nch + off];
      if (cf == size_t(-1)) throw std::runtime_error("nested_injection: a coarse cell misses a child");
      out[(size_t)nd] = cnf[cf * nn + af];
    }
  for (int64_t i = 0; i < n_coarse; ++i)
    if (out[(size_t)i] < 0) throw std::runtime_error("nested_injection: a coarse node belongs to no cell");
  return out;
}

} // namespace ifem_host
