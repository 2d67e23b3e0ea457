// Tiny dense fp64 helpers used on the host at context creation (D x D, D <= a few thousand).
#include <cmath>
#include <vector>

#include "vgpa_internal.h"

namespace vgpa {

// Lower Cholesky factor from the LOWER triangle of a (like LAPACK potrf('L'), which
// numpy.linalg.cholesky calls; the upper triangle is never read).
bool host_cholesky_lower(int n, const double* a, double* l) {
  for (int i = 0; i < n * n; i++) l[i] = 0.0;
  for (int j = 0; j < n; j++) {
    double d = a[j * n + j];
    for (int k = 0; k < j; k++) d -= l[j * n + k] * l[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    l[j * n + j] = d;
    for (int i = j + 1; i < n; i++) {
      double s = a[i * n + j];
      for (int k = 0; k < j; k++) s -= l[i * n + k] * l[j * n + k];
      l[i * n + j] = s / d;
    }
  }
  return true;
}

void host_lower_inverse(int n, const double* l, double* linv) {
  for (int i = 0; i < n * n; i++) linv[i] = 0.0;
  for (int c = 0; c < n; c++) {
    for (int i = c; i < n; i++) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; k++) s -= l[i * n + k] * linv[k * n + c];
      linv[i * n + c] = s / l[i * n + i];
    }
  }
}

// a^-1 = C^T C with C = L^-1 (utilities.py:203-237, chol_inv); logdet = 2 sum log diag(L)
bool host_spd_inverse(int n, const double* a, double* ainv, double* logdet) {
  std::vector<double> l(n * n), c(n * n);
  if (!host_cholesky_lower(n, a, l.data())) return false;
  host_lower_inverse(n, l.data(), c.data());
  host_matmul(n, c.data(), c.data(), ainv, true, false);
  if (logdet) {
    double s = 0.0;
    for (int i = 0; i < n; i++) s += std::log(l[i * n + i]);
    *logdet = 2.0 * s;
  }
  return true;
}

void host_matmul(int n, const double* a, const double* b, double* c, bool ta, bool tb) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      double s = 0.0;
      for (int k = 0; k < n; k++) s += (ta ? a[k * n + i] : a[i * n + k]) * (tb ? b[j * n + k] : b[k * n + j]);
      c[i * n + j] = s;
    }
}

}  // namespace vgpa
