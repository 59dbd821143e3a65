#include "dense.hpp"
#include <limits>

namespace airhip {

static inline double sign_of(double a, double b) { return b >= 0.0 ? std::fabs(a) : -std::fabs(a); }

std::vector<double> ls_qr(const Dense& A0, const std::vector<double>& b0) {
  Dense A = A0;
  std::vector<double> b = b0;
  const int m = A.m, n = A.n;
  for (int j = 0; j < n; j++) {
    double nrm = 0.0;
    for (int i = j; i < m; i++) nrm += A(i, j) * A(i, j);
    nrm = std::sqrt(nrm);
    if (nrm == 0.0) continue;
    double alpha = A(j, j);
    double beta = -sign_of(nrm, alpha);
    // v = x - beta e1, stored in place, v0 kept separately
    double v0 = alpha - beta;
    double vnorm2 = v0 * v0;
    for (int i = j + 1; i < m; i++) vnorm2 += A(i, j) * A(i, j);
    if (vnorm2 == 0.0) continue;
    auto apply = [&](double* col) {
      double s = v0 * col[j];
      for (int i = j + 1; i < m; i++) s += A(i, j) * col[i];
      s = 2.0 * s / vnorm2;
      col[j] -= s * v0;
      for (int i = j + 1; i < m; i++) col[i] -= s * A(i, j);
    };
    for (int c = j + 1; c < n; c++) apply(&A.a[(size_t)c * m]);
    apply(b.data());
    A(j, j) = beta;
  }
  std::vector<double> x(n, 0.0);
  for (int j = n - 1; j >= 0; j--) {
    double s = b[j];
    for (int c = j + 1; c < n; c++) s -= A(j, c) * x[c];
    x[j] = s / A(j, j);
  }
  return x;
}

std::vector<double> lu_solve(const Dense& A0, const std::vector<double>& b0) {
  Dense A = A0;
  std::vector<double> x = b0;
  const int n = A.n;
  for (int k = 0; k < n; k++) {
    int p = k; double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; i++) if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); p = i; }
    if (best == 0.0) continue;
    if (p != k) {
      for (int c = 0; c < n; c++) std::swap(A(k, c), A(p, c));
      std::swap(x[k], x[p]);
    }
    const double piv = 1.0 / A(k, k);
    for (int i = k + 1; i < n; i++) {
      const double l = A(i, k) * piv;
      if (l == 0.0) continue;
      A(i, k) = l;
      for (int c = k + 1; c < n; c++) A(i, c) -= l * A(k, c);
      x[i] -= l * x[k];
    }
  }
  for (int k = n - 1; k >= 0; k--) {
    double s = x[k];
    for (int c = k + 1; c < n; c++) s -= A(k, c) * x[c];
    x[k] = (A(k, k) == 0.0) ? 0.0 : s / A(k, k);
  }
  return x;
}

Dense lu_inverse(const Dense& A0) {
  Dense A = A0;
  const int n = A.n;
  std::vector<int> piv((size_t)n);
  for (int k = 0; k < n; k++) {
    int p = k; double best = std::fabs(A(k, k));
    for (int i = k + 1; i < n; i++) if (std::fabs(A(i, k)) > best) { best = std::fabs(A(i, k)); p = i; }
    piv[k] = p;
    if (best == 0.0) continue;
    if (p != k) for (int c = 0; c < n; c++) std::swap(A(k, c), A(p, c));
    const double d = 1.0 / A(k, k);
    for (int i = k + 1; i < n; i++) A(i, k) *= d;
#pragma omp parallel for schedule(static) if (n - k > 256)
    for (int c = k + 1; c < n; c++) {
      const double u = A(k, c);
      if (u == 0.0) continue;
      for (int i = k + 1; i < n; i++) A(i, c) -= A(i, k) * u;
    }
  }
  Dense X(n, n);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n; j++) {
    double* x = &X.a[(size_t)j * n];
    for (int i = 0; i < n; i++) x[i] = 0.0;
    x[j] = 1.0;
    for (int k = 0; k < n; k++) if (piv[k] != k) std::swap(x[k], x[piv[k]]);           // P e_j
    for (int k = 0; k < n; k++) { const double xk = x[k]; if (xk != 0.0) for (int i = k + 1; i < n; i++) x[i] -= A(i, k) * xk; }
    for (int k = n - 1; k >= 0; k--) {
      x[k] = (A(k, k) == 0.0) ? 0.0 : x[k] / A(k, k);
      const double xk = x[k];
      if (xk != 0.0) for (int i = 0; i < k; i++) x[i] -= A(i, k) * xk;
    }
  }
  return X;
}

// One-sided Jacobi SVD of an m x n matrix with m >= n: on return the columns of A are
// U*diag(s), V holds the right singular vectors.
static void jacobi_svd(Dense& A, Dense& V, std::vector<double>& s) {
  const int m = A.m, n = A.n;
  V = Dense(n, n);
  for (int i = 0; i < n; i++) V(i, i) = 1.0;
  const double eps = std::numeric_limits<double>::epsilon();
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double app = 0, aqq = 0, apq = 0;
        for (int i = 0; i < m; i++) { app += A(i, p) * A(i, p); aqq += A(i, q) * A(i, q); apq += A(i, p) * A(i, q); }
        if (std::fabs(apq) <= eps * std::sqrt(app * aqq) || apq == 0.0) continue;
        rotated = true;
        double zeta = (aqq - app) / (2.0 * apq);
        double t = sign_of(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; i++) {
          double x = A(i, p), y = A(i, q);
          A(i, p) = c * x - sn * y; A(i, q) = sn * x + c * y;
        }
        for (int i = 0; i < n; i++) {
          double x = V(i, p), y = V(i, q);
          V(i, p) = c * x - sn * y; V(i, q) = sn * x + c * y;
        }
      }
    if (!rotated) break;
  }
  s.assign(n, 0.0);
  for (int j = 0; j < n; j++) {
    double nrm = 0; for (int i = 0; i < m; i++) nrm += A(i, j) * A(i, j);
    s[j] = std::sqrt(nrm);
  }
}

std::vector<double> ls_minnorm_svd(const Dense& A0, const std::vector<double>& b, double rcond) {
  const int m = A0.m, n = A0.n;
  if (rcond < 0) rcond = std::numeric_limits<double>::epsilon();
  std::vector<double> x(n, 0.0);
  if (m >= n) {
    Dense A = A0, V; std::vector<double> s;
    jacobi_svd(A, V, s);
    double smax = 0; for (double v : s) smax = std::max(smax, v);
    for (int j = 0; j < n; j++) {
      if (s[j] <= rcond * smax || s[j] == 0.0) continue;
      double ub = 0; for (int i = 0; i < m; i++) ub += A(i, j) * b[i];  // (s_j u_j).b
      double coef = ub / (s[j] * s[j]);
      for (int i = 0; i < n; i++) x[i] += coef * V(i, j);
    }
    return x;
  }
  // m < n: SVD of A^T = U S V^T  ->  A = V S U^T, x = U S^+ V^T b
  Dense At(n, m);
  for (int i = 0; i < m; i++) for (int j = 0; j < n; j++) At(j, i) = A0(i, j);
  Dense V; std::vector<double> s;
  jacobi_svd(At, V, s);
  double smax = 0; for (double v : s) smax = std::max(smax, v);
  for (int j = 0; j < m; j++) {
    if (s[j] <= rcond * smax || s[j] == 0.0) continue;
    double vb = 0; for (int i = 0; i < m; i++) vb += V(i, j) * b[i];
    double coef = vb / (s[j] * s[j]);
    for (int i = 0; i < n; i++) x[i] += coef * At(i, j);
  }
  return x;
}

// Francis double-shift QR on an upper Hessenberg matrix (the classical EISPACK `hqr`
// iteration), eigenvalues only.
bool hessenberg_eig(Dense a, std::vector<double>& wr, std::vector<double>& wi) {
  const int n = a.n;
  wr.assign(n, 0.0); wi.assign(n, 0.0);
  if (n == 0) return true;
  double anorm = 0.0;
  for (int i = 0; i < n; i++)
    for (int j = std::max(i - 1, 0); j < n; j++) anorm += std::fabs(a(i, j));
  int nn = n - 1;
  double t = 0.0, p = 0, q = 0, r = 0, s = 0, w, x, y, z;
  while (nn >= 0) {
    int its = 0, l;
    do {
      for (l = nn; l >= 1; l--) {
        s = std::fabs(a(l - 1, l - 1)) + std::fabs(a(l, l));
        if (s == 0.0) s = anorm;
        if (std::fabs(a(l, l - 1)) + s == s) { a(l, l - 1) = 0.0; break; }
      }
      x = a(nn, nn);
      if (l == nn) {
        wr[nn] = x + t; wi[nn--] = 0.0;
      } else {
        y = a(nn - 1, nn - 1);
        w = a(nn, nn - 1) * a(nn - 1, nn);
        if (l == nn - 1) {
          p = 0.5 * (y - x);
          q = p * p + w;
          z = std::sqrt(std::fabs(q));
          x += t;
          if (q >= 0.0) {
            z = p + sign_of(z, p);
            wr[nn - 1] = wr[nn] = x + z;
            if (z != 0.0) wr[nn] = x - w / z;
            wi[nn - 1] = wi[nn] = 0.0;
          } else {
            wr[nn - 1] = wr[nn] = x + p;
            wi[nn - 1] = z; wi[nn] = -z;
          }
          nn -= 2;
        } else {
          if (its == 200) return false;
          if (its > 0 && its % 10 == 0) {
            t += x;
            for (int i = 0; i <= nn; i++) a(i, i) -= x;
            s = std::fabs(a(nn, nn - 1)) + std::fabs(a(nn - 1, nn - 2));
            y = x = 0.75 * s;
            w = -0.4375 * s * s;
          }
          ++its;
          int mm;
          for (mm = nn - 2; mm >= l; mm--) {
            z = a(mm, mm);
            r = x - z; s = y - z;
            p = (r * s - w) / a(mm + 1, mm) + a(mm, mm + 1);
            q = a(mm + 1, mm + 1) - z - r - s;
            r = a(mm + 2, mm + 1);
            s = std::fabs(p) + std::fabs(q) + std::fabs(r);
            p /= s; q /= s; r /= s;
            if (mm == l) break;
            double u = std::fabs(a(mm, mm - 1)) * (std::fabs(q) + std::fabs(r));
            double v = std::fabs(p) * (std::fabs(a(mm - 1, mm - 1)) + std::fabs(z) + std::fabs(a(mm + 1, mm + 1)));
            if (u + v == v) break;
          }
          for (int i = mm + 2; i <= nn; i++) {
            a(i, i - 2) = 0.0;
            if (i != mm + 2) a(i, i - 3) = 0.0;
          }
          for (int k = mm; k <= nn - 1; k++) {
            if (k != mm) {
              p = a(k, k - 1);
              q = a(k + 1, k - 1);
              r = 0.0;
              if (k != nn - 1) r = a(k + 2, k - 1);
              if ((x = std::fabs(p) + std::fabs(q) + std::fabs(r)) != 0.0) { p /= x; q /= x; r /= x; }
            }
            if ((s = sign_of(std::sqrt(p * p + q * q + r * r), p)) != 0.0) {
              if (k == mm) {
                if (l != mm) a(k, k - 1) = -a(k, k - 1);
              } else a(k, k - 1) = -s * x;
              p += s; x = p / s; y = q / s; z = r / s; q /= p; r /= p;
              for (int j = k; j <= nn; j++) {
                p = a(k, j) + q * a(k + 1, j);
                if (k != nn - 1) { p += r * a(k + 2, j); a(k + 2, j) -= p * z; }
                a(k + 1, j) -= p * y;
                a(k, j) -= p * x;
              }
              int mmin = nn < k + 3 ? nn : k + 3;
              for (int i = l; i <= mmin; i++) {
                p = x * a(i, k) + y * a(i, k + 1);
                if (k != nn - 1) { p += z * a(i, k + 2); a(i, k + 2) -= p * r; }
                a(i, k + 1) -= p * q;
                a(i, k) -= p;
              }
            }
          }
        }
      }
    } while (l < nn - 1);
  }
  return true;
}

Dense tall_qr_R(std::vector<Arr<double>>& cols, int64_t n) {
  const int k = (int)cols.size();
  Dense R(k, k);
  for (int j = 0; j < k && j < n; j++) {
    double* aj = cols[j].data();
    const double nrm2 = det_sum(j, n, [&](int64_t i) { return aj[i] * aj[i]; });
    double nrm = std::sqrt(nrm2);
    for (int i = 0; i < j; i++) R(i, j) = aj[i];
    if (nrm == 0.0) { R(j, j) = 0.0; continue; }
    double alpha = aj[j];
    double beta = -sign_of(nrm, alpha);
    double v0 = alpha - beta;
    double vnorm2 = nrm2 - alpha * alpha + v0 * v0;
    R(j, j) = beta;
    if (vnorm2 == 0.0) continue;
    for (int c = j + 1; c < k; c++) {
      double* ac = cols[c].data();
      double s = det_sum(j + 1, n, [&](int64_t i) { return aj[i] * ac[i]; });
      s += v0 * ac[j];
      s = 2.0 * s / vnorm2;
      ac[j] -= s * v0;
#pragma omp parallel for schedule(static)
      for (int64_t i = j + 1; i < n; i++) ac[i] -= s * aj[i];
    }
  }
  return R;
}

}  // namespace airhip
