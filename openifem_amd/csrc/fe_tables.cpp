// fe_tables.cpp -- reference-cell tables the kernels stage in LDS.
// Restates deal.II's FE_Q (Lagrange on equidistant nodes), QGauss(kv+1) and the Q1 geometry mapping that
// FEValues(fe, quad, flags) uses with the default mapping (mpi_insim.cpp:167-175, mpi_fluid_solver.cpp:27-35).
#include <cmath>
#include <cstring>
#include "ctx.hpp"

namespace ifem {

static void lagrange(int k, double x, double *N, double *dN) {
  const int n = k + 1;
  for (int j = 0; j < n; ++j) {
    const double xj = double(j) / k;
    double v = 1, dv = 0;
    for (int m = 0; m < n; ++m)
      if (m != j) v *= (x - double(m) / k) / (xj - double(m) / k);
    for (int l = 0; l < n; ++l) {
      if (l == j) continue;
      double t = 1.0 / (xj - double(l) / k);
      for (int m = 0; m < n; ++m)
        if (m != j && m != l) t *= (x - double(m) / k) / (xj - double(m) / k);
      dv += t;
    }
    N[j] = v;
    dN[j] = dv;
  }
}

static void gauss01(int n, double *x, double *w) {
  if (n == 2) {
    const double a = 0.5 / std::sqrt(3.0);
    x[0] = 0.5 - a; x[1] = 0.5 + a; w[0] = w[1] = 0.5;
  } else if (n == 3) {
    const double a = 0.5 * std::sqrt(0.6);
    x[0] = 0.5 - a; x[1] = 0.5; x[2] = 0.5 + a;
    w[0] = w[2] = 5.0 / 18.0; w[1] = 8.0 / 18.0;
  } else
    throw Error(IFEM_E_BADPARAM, "QGauss order unsupported");
}

// tensor-product shapes of degree k at reference point xi; local index x-fastest
static void tensor_shapes(int dim, int k, const double *xi, double *N, double *dN /*[a][dim]*/) {
  double n1[3][3], d1[3][3];
  for (int d = 0; d < dim; ++d) lagrange(k, xi[d], n1[d], d1[d]);
  const int n = k + 1;
  int nn = 1;
  for (int d = 0; d < dim; ++d) nn *= n;
  for (int a = 0; a < nn; ++a) {
    int ia[3] = {0, 0, 0}, t = a;
    for (int d = 0; d < dim; ++d) { ia[d] = t % n; t /= n; }
    double v = 1;
    for (int d = 0; d < dim; ++d) v *= n1[d][ia[d]];
    N[a] = v;
    for (int e = 0; e < dim; ++e) {
      double g = 1;
      for (int d = 0; d < dim; ++d) g *= (d == e) ? d1[d][ia[d]] : n1[d][ia[d]];
      dN[a * dim + e] = g;
    }
  }
}

void build_fe_tables(FeTables &t, int dim, int kv) {
  std::memset(&t, 0, sizeof(t));
  t.dim = dim; t.kv = kv;
  const int n1 = kv + 1;
  t.nu = 1; t.np = 1; t.nq = 1;
  for (int d = 0; d < dim; ++d) { t.nu *= n1; t.np *= 2; t.nq *= n1; }
  double gx[3], gw[3];
  gauss01(n1, gx, gw);
  for (int q = 0; q < t.nq; ++q) {
    double xi[3] = {0, 0, 0}, w = 1;
    int r = q;
    for (int d = 0; d < dim; ++d) { xi[d] = gx[r % n1]; w *= gw[r % n1]; r /= n1; }
    t.w[q] = w;
    tensor_shapes(dim, kv, xi, t.phi + q * t.nu, t.dphi + q * t.nu * dim);
    tensor_shapes(dim, 1, xi, t.psi + q * t.np, t.dpsi + q * t.np * dim);
  }
  // faces: QGauss<dim-1>(kv+1) on the reference faces x-,x+,y-,y+,z-,z+
  t.nqf = (dim == 2) ? n1 : n1 * n1;
  for (int qf = 0; qf < t.nqf; ++qf) {
    double w = 1; int r = qf;
    for (int d = 0; d < dim - 1; ++d) { w *= gw[r % n1]; r /= n1; }
    t.fw[qf] = w;
  }
  for (int f = 0; f < 2 * dim; ++f) {
    const int nd = f / 2;
    for (int qf = 0; qf < t.nqf; ++qf) {
      double xi[3] = {0, 0, 0}; int r = qf;
      for (int d = 0; d < dim; ++d) {
        if (d == nd) xi[d] = double(f % 2);
        else { xi[d] = gx[r % n1]; r /= n1; }
      }
      double dn_u[27 * 3], n_p[8];
      tensor_shapes(dim, kv, xi, t.fphi + (f * t.nqf + qf) * t.nu, dn_u);
      tensor_shapes(dim, 1, xi, n_p, t.fdpsi + (f * t.nqf + qf) * t.np * dim);
    }
  }
}

} // namespace ifem
