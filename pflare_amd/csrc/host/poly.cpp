// Polynomial approximate inverses: GMRES-polynomial coefficients (Arnoldi and power
// basis), Newton-basis roots, Neumann, (weighted) Jacobi, and the fixed-sparsity
// assembled polynomial.  Follows src/Gmres_Poly.F90:41-1337,1522-1928,
// src/Gmres_Poly_Newton.F90:21-712, src/Neumann_Poly.F90:59-215,
// src/Weighted_Jacobi.F90:15-86, src/Approx_Inverse_Setup.F90:207-500.
#include "air.hpp"
#include "dense.hpp"
#include <limits>

namespace airhip {

// ---------------------------------------------------------------------------------------
// vector helpers
// ---------------------------------------------------------------------------------------
static double vdot(const double* a, const double* b, int64_t n) {
  return det_sum(0, n, [&](int64_t i) { return a[i] * b[i]; });
}
static double vnorm(const double* a, int64_t n) { return std::sqrt(vdot(a, a, n)); }

struct MatVec {
  OpView A;
  bool dscale = false;          // y = D^-1 (A x)
  const double* d = nullptr;
  void operator()(const double* x, double* y) const {
    const int64_t n = A.n;
    if (A.diag) {
      const double* ad = A.diag;
      if (dscale) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) y[i] = (ad[i] * x[i]) / d[i];
      } else {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) y[i] = ad[i] * x[i];
      }
      return;
    }
    if (dscale) spmv_dscale(*A.A, d, x, y); else spmv(*A.A, x, y);
  }
};

static void box_muller(uint64_t seed, int64_t n, double* out) {
  const uint64_t key = rng_key(seed);
  const double two_pi = 2.0 * 3.14159265358979323846;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    double u1 = uniform01(key, (uint64_t)i), u2 = uniform01(key, (uint64_t)(n + i));
    if (u1 == 0.0) u1 = 1.0 / 9007199254740992.0;
    out[i] = std::sqrt(-2.0 * std::log(u1)) * std::cos(two_pi * u2);
  }
}

// ---------------------------------------------------------------------------------------
// Arnoldi (src/Gmres_Poly.F90:308-456)
// ---------------------------------------------------------------------------------------
struct ArnoldiOut { double beta = 0; Dense H, C; int m = 0; std::vector<double> y; };

static std::vector<double> ls_solve_arnoldi(double beta, int m, const Dense& H) {
  Dense Hc(m + 1, m);
  for (int j = 0; j < m; j++) for (int i = 0; i <= m; i++) Hc(i, j) = H(i, j);
  std::vector<double> g0(m + 1, 0.0); g0[0] = beta;
  std::vector<double> x = ls_qr(Hc, g0);
  std::vector<double> y(H.n, 0.0);
  for (int i = 0; i < m; i++) y[i] = x[i];
  return y;
}

static ArnoldiOut arnoldi(const MatVec& mv, int64_t n, int poly_order, uint64_t seed,
                          double rel_tol, bool compute_cn) {
  const int sub = poly_order + 1;
  ArnoldiOut o;
  o.H = Dense(poly_order + 2, poly_order + 1);
  if (compute_cn) o.C = Dense(poly_order + 2, poly_order + 2);
  o.y.assign(poly_order + 1, 0.0);
  std::vector<Arr<double>> V(sub + 1);
  V[0].alloc(n);
  box_muller(seed, n, V[0].data());
  Arr<double> w(n);
  o.beta = vnorm(V[0].data(), n);
  { double ib = o.beta; double* v0 = V[0].data();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) v0[i] = v0[i] / ib; }
  if (compute_cn) o.C(0, 0) = 1.0 / o.beta;
  o.m = sub;
  std::vector<double> c_j(poly_order + 2), h(poly_order + 1);
  for (int m = 1; m <= sub; m++) {
    mv(V[m - 1].data(), w.data());
    if (compute_cn) {
      std::fill(c_j.begin(), c_j.end(), 0.0);
      for (int i = 0; i < m; i++) c_j[i + 1] = o.C(i, m - 1);
    }
    for (int i = 0; i < m; i++) { h[i] = vdot(w.data(), V[i].data(), n); o.H(i, m - 1) = h[i]; }
    for (int i = 0; i < m; i++) {
      const double hi = h[i]; const double* vi = V[i].data(); double* wp = w.data();
#pragma omp parallel for schedule(static)
      for (int64_t q = 0; q < n; q++) wp[q] -= hi * vi[q];
    }
    if (compute_cn)
      for (int i = 0; i < m; i++)
        for (int q = 0; q <= i; q++) c_j[q] -= o.C(q, i) * o.H(i, m - 1);
    double hn = vnorm(w.data(), n);
    o.H(m, m - 1) = hn;
    if (hn < TOL_LUCKY) {
      if (rel_tol > 0) o.y = ls_solve_arnoldi(o.beta, m, o.H);
      o.m = m;
      break;
    }
    V[m].alloc(n);
    { double* vm = V[m].data(); const double* wp = w.data();
#pragma omp parallel for schedule(static)
      for (int64_t q = 0; q < n; q++) vm[q] = wp[q] / hn; }
    if (compute_cn) for (int q = 0; q <= m; q++) o.C(q, m) = c_j[q] / hn;
    if (rel_tol > 0) {
      o.y = ls_solve_arnoldi(o.beta, m, o.H);
      double nr = 0.0;
      for (int i = 0; i < m; i++) {   // norm2(g0(1:m)) of H y - beta e1 (Gmres_Poly.F90:438-446)
        double g = 0.0;
        for (int j = 0; j < m; j++) g += o.H(i, j) * o.y[j];
        if (i == 0) g -= o.beta;
        nr += g * g;
      }
      if (std::sqrt(nr) / o.beta < rel_tol) { o.m = m; break; }
    }
  }
  return o;
}

static std::vector<double> coeffs_arnoldi(const MatVec& mv, int64_t n, int poly_order, uint64_t seed) {
  ArnoldiOut a = arnoldi(mv, n, poly_order, seed, TOL_ARNOLDI, true);
  std::vector<double> c(poly_order + 1, 0.0);
  for (int i = 0; i < a.m; i++) {
    double s = 0.0;
    for (int j = 0; j < a.m; j++) s += a.C(i, j) * a.y[j];
    c[i] = s;
  }
  return c;
}

// src/Gmres_Poly.F90:552-773
static std::vector<double> coeffs_power(const MatVec& mv, int64_t n, int poly_order, uint64_t seed) {
  const int sub = poly_order + 1;
  std::vector<Arr<double>> K(sub + 1);
  for (auto& k : K) k.alloc(n);
  box_muller(seed, n, K[0].data());
  for (int i = 0; i < sub; i++) mv(K[i].data(), K[i + 1].data());
  Dense R = tall_qr_R(K, n);
  Dense a(sub + 1, sub);
  for (int j = 0; j < sub; j++) for (int i = 0; i <= sub; i++) a(i, j) = R(i, j + 1);
  std::vector<double> g0(sub + 1, 0.0); g0[0] = R(0, 0);
  std::vector<double> x = ls_minnorm_svd(a, g0, -1.0);
  x.resize(sub);
  return x;
}

// ---------------------------------------------------------------------------------------
// Newton basis roots (src/Gmres_Poly_Newton.F90:21-712)
// ---------------------------------------------------------------------------------------
static std::vector<int> modified_leja(const std::vector<double>& re, const std::vector<double>& im) {
  const int n = (int)re.size();
  std::vector<int> order; order.reserve(n);
  std::vector<char> done(n, 0);
  auto push = [&](int i) {
    order.push_back(i); done[i] = 1;
    if (im[i] != 0.0) { int j = im[i] > 0 ? i + 1 : i - 1; order.push_back(j); done[j] = 1; }
  };
  int best = 0; double bm = -1.0;
  for (int i = 0; i < n; i++) { double m = std::sqrt(re[i] * re[i] + im[i] * im[i]); if (m > bm) { bm = m; best = i; } }
  push(best);
  while ((int)order.size() < n) {
    double max_mag = -std::numeric_limits<double>::max(); int max_i = -1;
    for (int i = 0; i < n; i++) {
      if (done[i]) continue;
      double val = 1.0;
      for (int k : order) {
        double a = re[i] - re[k], b = im[i] - im[k];
        val += std::log10(std::sqrt(a * a + b * b));
      }
      if (val > max_mag) { max_mag = val; max_i = i; }
    }
    if (max_mag < 0 || max_i < 0)
      for (int i = 0; i < n; i++) if (!done[i]) { max_i = i; break; }
    push(max_i);
  }
  return order;
}

static void cluster_eigenvalues(std::vector<double>& re, std::vector<double>& im, double rel_tol, double abs_tol) {
  const int n = (int)re.size();
  std::vector<char> used(n, 0);
  std::vector<double> rt, it;
  for (int i = 0; i < n; i++) {
    if (used[i]) continue;
    if (re[i] == 0.0 && im[i] == 0.0) { used[i] = 1; continue; }
    double sr = re[i], si = im[i]; int cnt = 1; used[i] = 1;
    double mag_i = std::sqrt(re[i] * re[i] + im[i] * im[i]);
    for (int j = i + 1; j < n; j++) {
      if (used[j]) continue;
      if (re[j] == 0.0 && im[j] == 0.0) { used[j] = 1; continue; }
      double mag_j = std::sqrt(re[j] * re[j] + im[j] * im[j]);
      double dist = std::sqrt((re[j] - re[i]) * (re[j] - re[i]) + (im[j] - im[i]) * (im[j] - im[i]));
      double scale = std::max(std::max(mag_i, mag_j), 1.0);
      if (dist <= abs_tol + rel_tol * scale) { sr += re[j]; si += im[j]; cnt++; used[j] = 1; }
    }
    rt.push_back(sr / cnt); it.push_back(si / cnt);
  }
  std::fill(re.begin(), re.end(), 0.0); std::fill(im.begin(), im.end(), 0.0);
  for (size_t k = 0; k < rt.size(); k++) { re[k] = rt[k]; im[k] = it[k]; }
}

static void compute_extra_roots(const std::vector<double>& re, const std::vector<double>& im,
                                std::vector<double>& re_o, std::vector<double>& im_o) {
  const int n = (int)re.size();
  std::vector<double> pof(n, 1.0);
  std::vector<int> extra(n, 0), overflow(n, 0);
  int total_extra = 0;
  for (int k = 0; k < n; k++) {
    double a = re[k], b = im[k];
    if (b < 0) continue;
    if (std::fabs(a) < TOL_ZERO) continue;
    if (a * a + b * b < TOL_ZERO) continue;
    for (int i = 0; i < n; i++) {
      if (i == k) continue;
      double c = re[i], d = im[i];
      if (std::fabs(c) < TOL_ZERO) continue;
      if (c * c + d * d < TOL_ZERO) continue;
      double dr = (a * c + b * d) / (c * c + d * d), di = (b * c - a * d) / (c * c + d * d);
      double dm = std::sqrt((1 - dr) * (1 - dr) + di * di);
      if (std::log10(pof[k]) + std::log10(dm) > 307) { overflow[k] += (int)std::log10(pof[k]); pof[k] = 1.0; }
      pof[k] *= dm;
    }
    double lp = std::log10(pof[k]);
    if (lp > 4 || overflow[k] != 0) {
      extra[k] = (int)std::ceil((lp + overflow[k] - 4.0) / 14.0);
      total_extra += extra[k];
      if (b > 0) { pof[k + 1] = -1; total_extra += extra[k]; }
    }
  }
  re_o.assign(n + total_extra, 0.0); im_o.assign(n + total_extra, 0.0);
  for (int i = 0; i < n; i++) { re_o[i] = re[i]; im_o[i] = im[i]; }
  int cnt = n;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < extra[i]; j++) {
      re_o[cnt] = re[i]; im_o[cnt] = im[i];
      if (im[i] > 0) { re_o[cnt + 1] = re[i]; im_o[cnt + 1] = -im[i]; }
      cnt++;
      if (im[i] > 0) cnt++;
    }
}

static void roots_from_arnoldi(const ArnoldiOut& a, int poly_order, bool add_roots,
                               std::vector<double>& re_out, std::vector<double>& im_out);

static void roots_newton(const MatVec& mv, int64_t n, int poly_order, bool add_roots, uint64_t seed,
                         std::vector<double>& re_out, std::vector<double>& im_out) {
  ArnoldiOut a = arnoldi(mv, n, poly_order, seed, -1.0, false);
  roots_from_arnoldi(a, poly_order, add_roots, re_out, im_out);
}

// entry points for an Arnoldi process run elsewhere (the device-side auto-truncation test, device/setup_accel.cpp)
void krylov_start_vector(uint64_t seed, int64_t n, double* out) { box_muller(seed, n, out); }
void newton_roots_from_hessenberg(const double* H, double beta, int m, int poly_order, bool add_roots,
                                  std::vector<double>& re, std::vector<double>& im) {
  ArnoldiOut a;
  a.beta = beta; a.m = m;
  a.H = Dense(poly_order + 2, poly_order + 1);
  for (int j = 0; j < poly_order + 1; j++) for (int i = 0; i < poly_order + 2; i++) a.H(i, j) = H[(size_t)j * (poly_order + 2) + i];
  roots_from_arnoldi(a, poly_order, add_roots, re, im);
}

static void roots_from_arnoldi(const ArnoldiOut& a, int poly_order, bool add_roots,
                               std::vector<double>& re_out, std::vector<double>& im_out) {
  const int p1 = poly_order + 1;
  // harmonic Ritz: H(:,d) += h_{d+1,d}^2 * (H^-T e_d), min-norm solve with rcond
  Dense HT(p1, p1);
  for (int i = 0; i < p1; i++) for (int j = 0; j < p1; j++) HT(i, j) = a.H(j, i);
  std::vector<double> e_d(p1, 0.0); e_d[poly_order] = 1.0;
  std::vector<double> sol = ls_minnorm_svd(HT, e_d, TOL_RCOND);
  double hd = a.H(poly_order + 1, poly_order);
  Dense Hs(p1, p1);
  for (int i = 0; i < p1; i++) for (int j = 0; j < p1; j++) Hs(i, j) = a.H(i, j);
  for (int i = 0; i < p1; i++) Hs(i, poly_order) += sol[i] * hd * hd;
  double hn2 = 0.0;
  for (int i = 0; i < a.m; i++) for (int j = 0; j < a.m; j++) hn2 += Hs(i, j) * Hs(i, j);
  const double H_norm = std::sqrt(hn2);
  std::vector<double> re, im;
  if (!hessenberg_eig(Hs, re, im)) throw std::runtime_error("Eig decomposition failed");
  const double eps = std::numeric_limits<double>::epsilon();
  const double rel_tol = std::sqrt(eps), abs_tol = eps * std::max(H_norm, a.beta);
  const double thr = (abs_tol + rel_tol * H_norm) * (abs_tol + rel_tol * H_norm);
  for (int i = 0; i < p1; i++)
    if (re[i] * re[i] + im[i] * im[i] < thr) { re[i] = 0.0; im[i] = 0.0; }
  cluster_eigenvalues(re, im, rel_tol, abs_tol);
  std::vector<double> rt, it;
  for (int i = 0; i < p1; i++) if (re[i] != 0.0 || im[i] != 0.0) { rt.push_back(re[i]); it.push_back(im[i]); }
  const int numerical_order = (int)rt.size();
  if (numerical_order == 0) { re_out.assign(p1, 0.0); im_out.assign(p1, 0.0); return; }
  if (add_roots) {
    std::vector<double> ra, ia;
    compute_extra_roots(rt, it, ra, ia);
    std::vector<double> pr = ra, pi = ia;
    for (int i = 0; i < numerical_order; i++) {
      int k = 0;
      for (size_t j = numerical_order; j < ra.size(); j++)
        if (ra[j] == rt[i] && std::fabs(ia[j]) == std::fabs(it[i])) { k++; pr[j] = ra[j] + k * TOL_LEJA_PERTURB; }
    }
    std::vector<int> idx = modified_leja(pr, pi);
    size_t total = ra.size() + (size_t)(p1 - numerical_order);
    re_out.assign(total, 0.0); im_out.assign(total, 0.0);
    for (size_t q = 0; q < ra.size(); q++) { re_out[q] = ra[idx[q]]; im_out[q] = ia[idx[q]]; }
    return;
  }
  std::vector<int> idx = modified_leja(rt, it);
  re_out.assign(p1, 0.0); im_out.assign(p1, 0.0);
  for (int q = 0; q < numerical_order; q++) { re_out[q] = rt[idx[q]]; im_out[q] = it[idx[q]]; }
}

static int newton_spmv_count(const std::vector<double>& re, const std::vector<double>& im) {
  const int n = (int)re.size();
  int i = 0, cnt = 0;
  while (i <= n - 2) {
    if (im[i] == 0.0) {
      if (std::fabs(re[i]) < TOL_ZERO) { i++; continue; }
      cnt++; i++;
    } else {
      if (re[i] * re[i] + im[i] * im[i] < TOL_ZERO) { i += 2; continue; }
      cnt++;
      if (i <= n - 3) cnt++;
      i += 2;
    }
  }
  return cnt;
}

// ---------------------------------------------------------------------------------------
// Fixed-sparsity assembled polynomial (src/Gmres_Poly.F90:920-1337)
// ---------------------------------------------------------------------------------------
thread_local SetupAccel* g_setup_accel = nullptr;

static void poly_diag_sparsity(const double* d, int64_t n, const std::vector<double>& c, InvOp& op) {
  // build_gmres_polynomial_inverse_0th_order_sparsity (Gmres_Poly.F90:1868-1928)
  op.kind = InvOp::DIAG; op.d.alloc(n);
  const int order = (int)c.size() - 1;
  double* out = op.d.data();
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    double inv = c[0], pw = d[i];
    for (int k = 1; k <= order; k++) { inv += c[k] * pw; if (k != order) pw *= d[i]; }
    out[i] = inv;
  }
}

static Csr mat_mult_powers_share_sparsity(const Csr& A, const std::vector<double>& c, int sparsity_order) {
  // matrix powers up to the sparsity order (unconstrained)
  std::vector<Csr> powers;  // powers[k] = A^(k+2)
  const Csr* pat = &A;
  for (int o = 2; o <= sparsity_order; o++) {
    powers.push_back(spgemm(A, o == 2 ? A : powers.back()));
    pat = &powers.back();
  }
  Csr cmat = copy_plus_diag(*pat);
  const int64_t m = A.m;
  const int nterms = (int)c.size();
  // values up to the sparsity order
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    for (int64_t k = cmat.rp[i]; k < cmat.rp[i + 1]; k++) cmat.v[k] *= c[sparsity_order];
    for (int o = sparsity_order - 1; o >= 1; o--) {
      const Csr& Po = (o == 1) ? A : powers[o - 2];
      int64_t a = Po.rp[i], ae = Po.rp[i + 1];
      for (int64_t k = cmat.rp[i]; k < cmat.rp[i + 1] && a < ae; k++) {
        while (a < ae && Po.ci[a] < cmat.ci[k]) a++;
        if (a < ae && Po.ci[a] == cmat.ci[k]) cmat.v[k] += c[o] * Po.v[a];
      }
    }
    for (int64_t k = cmat.rp[i]; k < cmat.rp[i + 1]; k++) if (cmat.ci[k] == i) { cmat.v[k] += c[0]; break; }
  }
  if (nterms <= sparsity_order + 1) return cmat;
  // constrained higher powers, one row at a time
#pragma omp parallel
  {
    std::vector<int32_t> match_idx;     // flattened: for each j in P(i), list of positions in P(i)
    std::vector<double> match_val;
    std::vector<int64_t> match_ptr;
    std::vector<double> prev, cur, acc;
    std::vector<int64_t> cpos;
#pragma omp for schedule(dynamic, 1024)
    for (int64_t i = 0; i < m; i++) {
      const int64_t p0 = pat->rp[i], ncols = pat->rp[i + 1] - p0;
      if (ncols == 0) continue;
      const int32_t* cols = pat->ci.data() + p0;
      match_idx.clear(); match_val.clear(); match_ptr.assign(ncols + 1, 0);
      for (int64_t j = 0; j < ncols; j++) {
        int32_t row = cols[j];
        int64_t a = A.rp[row], ae = A.rp[row + 1], b = 0;
        while (a < ae && b < ncols) {
          if (A.ci[a] == cols[b]) { match_idx.push_back((int32_t)b); match_val.push_back(A.v[a]); a++; b++; }
          else if (A.ci[a] < cols[b]) a++; else b++;
        }
        match_ptr[j + 1] = (int64_t)match_idx.size();
      }
      prev.assign(pat->v.data() + p0, pat->v.data() + p0 + ncols);
      cur.resize(ncols); acc.assign(ncols, 0.0);
      for (int term = sparsity_order + 1; term < nterms; term++) {
        std::fill(cur.begin(), cur.end(), 0.0);
        for (int64_t j = 0; j < ncols; j++) {
          const double pj = prev[j];
          for (int64_t q = match_ptr[j]; q < match_ptr[j + 1]; q++) cur[match_idx[q]] += pj * match_val[q];
        }
        if (c[term] != 0.0) for (int64_t q = 0; q < ncols; q++) acc[q] += c[term] * cur[q];
        prev.swap(cur);
      }
      // add into cmat (its row pattern is P(i) plus possibly the diagonal)
      int64_t k = cmat.rp[i];
      for (int64_t q = 0; q < ncols; q++) {
        while (cmat.ci[k] != cols[q]) k++;
        cmat.v[k] += acc[q];
      }
    }
  }
  return cmat;
}

// Assembled Newton-basis GMRES polynomial with fixed sparsity (counterpart of
// mat_mult_powers_share_sparsity_newton, src/Gmres_Poly_Newton.F90:1203-1872): the matrix recurrence of
// petsc_newton (Loe et al. Alg. 3; real roots and conjugate pairs in real arithmetic, zero roots skipped)
// carried out row by row, every product "row * A" constrained to the pattern S = pattern(A^sparsity_order) + I
// - the same "share sparsity" rule as the monomial version above (terms up to the sparsity order are exact
// because their pattern lies inside S).
static Csr mat_newton_share_sparsity(const Csr& A, const std::vector<double>& re, const std::vector<double>& im, int sparsity_order) {
  std::vector<Csr> powers;
  const Csr* pat = &A;
  for (int o = 2; o <= sparsity_order; o++) {
    powers.push_back(spgemm(A, o == 2 ? A : powers.back()));
    pat = &powers.back();
  }
  Csr cmat = copy_plus_diag(*pat);
  const int64_t m = A.m;
  const int nr = (int)re.size();
#pragma omp parallel
  {
    std::vector<int32_t> match_idx; std::vector<double> match_val; std::vector<int64_t> match_ptr;
    std::vector<double> t, t2, t3, y;
#pragma omp for schedule(dynamic, 1024)
    for (int64_t i = 0; i < m; i++) {
      const int64_t p0 = cmat.rp[i], ncols = cmat.rp[i + 1] - p0;
      const int32_t* cols = cmat.ci.data() + p0;
      match_idx.clear(); match_val.clear(); match_ptr.assign(ncols + 1, 0);
      int64_t dpos = -1;
      for (int64_t j = 0; j < ncols; j++) {
        if (cols[j] == i) dpos = j;
        const int32_t row = cols[j];
        int64_t a = A.rp[row], ae = A.rp[row + 1], b = 0;
        while (a < ae && b < ncols) {
          if (A.ci[a] == cols[b]) { match_idx.push_back((int32_t)b); match_val.push_back(A.v[a]); a++; b++; }
          else if (A.ci[a] < cols[b]) a++; else b++;
        }
        match_ptr[j + 1] = (int64_t)match_idx.size();
      }
      // out = in * A restricted to the row's pattern
      auto mult = [&](const std::vector<double>& in, std::vector<double>& out) {
        std::fill(out.begin(), out.end(), 0.0);
        for (int64_t j = 0; j < ncols; j++) {
          const double pj = in[j];
          if (pj == 0.0) continue;
          for (int64_t q = match_ptr[j]; q < match_ptr[j + 1]; q++) out[match_idx[q]] += pj * match_val[q];
        }
      };
      t.assign(ncols, 0.0); t2.assign(ncols, 0.0); t3.assign(ncols, 0.0); y.assign(ncols, 0.0);
      t[dpos] = 1.0;                                   // row i of the identity
      int k = 0;
      while (k <= nr - 2) {
        if (im[k] == 0.0) {
          if (std::fabs(re[k]) < TOL_ZERO) { k++; continue; }
          const double a = 1.0 / re[k];
          for (int64_t q = 0; q < ncols; q++) y[q] += a * t[q];
          mult(t, t2);
          for (int64_t q = 0; q < ncols; q++) t[q] += -a * t2[q];
          k++;
        } else {
          const double sq = re[k] * re[k] + im[k] * im[k];
          if (sq < TOL_ZERO) { k += 2; continue; }
          const double tr = 2 * re[k], isq = 1.0 / sq;
          mult(t, t2);
          for (int64_t q = 0; q < ncols; q++) { t2[q] = tr * t[q] - t2[q]; y[q] += isq * t2[q]; }
          if (k <= nr - 3) {
            mult(t2, t3);
            for (int64_t q = 0; q < ncols; q++) t[q] += -isq * t3[q];
          }
          k += 2;
        }
      }
      if (im[nr - 1] == 0.0 && std::fabs(re[nr - 1]) > TOL_ZERO) {
        const double a = 1.0 / re[nr - 1];
        for (int64_t q = 0; q < ncols; q++) y[q] += a * t[q];
      }
      for (int64_t q = 0; q < ncols; q++) cmat.v[p0 + q] = y[q];
    }
  }
  return cmat;
}

// scalar version for a MATDIAGONAL operator: p(d_i)
static void newton_diag(const double* d, int64_t n, const std::vector<double>& re, const std::vector<double>& im, InvOp& op) {
  op.kind = InvOp::DIAG; op.d.alloc(n);
  const int nr = (int)re.size();
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    double t = 1.0, y = 0.0; const double a_ = d[i];
    int k = 0;
    while (k <= nr - 2) {
      if (im[k] == 0.0) {
        if (std::fabs(re[k]) < TOL_ZERO) { k++; continue; }
        const double a = 1.0 / re[k];
        y += a * t; t += -a * (t * a_); k++;
      } else {
        const double sq = re[k] * re[k] + im[k] * im[k];
        if (sq < TOL_ZERO) { k += 2; continue; }
        const double t2 = 2 * re[k] * t - t * a_;
        y += t2 / sq;
        if (k <= nr - 3) t += -(t2 * a_) / sq;
        k += 2;
      }
    }
    if (im[nr - 1] == 0.0 && std::fabs(re[nr - 1]) > TOL_ZERO) y += t / re[nr - 1];
    op.d[i] = y;
  }
}

// src/Gmres_Poly.F90:1522-1813
static void build_gmres_polynomial_inverse(const OpView& A, int poly_order, const std::vector<double>& c,
                                           int sparsity_order, bool matrix_free, bool diag_scale, InvOp& op) {
  const int64_t n = A.n;
  if (matrix_free) {
    op.kind = InvOp::HORNER; op.coeffs = c; op.diag_scale = diag_scale; op.neumann = false;
    int nz = 0; for (size_t k = 0; k + 1 < c.size(); k++) if (c[k] != 0.0) nz++;
    op.spmvs = nz;
    return;
  }
  Arr<double> dinv;
  Csr scaled;
  const Csr* As = A.A;
  Arr<double> dscaled;
  if (A.diag) {
    // MATDIAGONAL operator: everything is pointwise
    dscaled.alloc(n);
    for (int64_t i = 0; i < n; i++) dscaled[i] = diag_scale ? 1.0 : A.diag[i];
    if (diag_scale) { dinv.alloc(n); for (int64_t i = 0; i < n; i++) dinv[i] = 1.0 / A.diag[i]; }
    if (poly_order == 0) { op.kind = InvOp::DIAG; op.d.alloc(n); op.d.fill(c[0]); }
    else poly_diag_sparsity(dscaled.data(), n, c, op);
    if (diag_scale) for (int64_t i = 0; i < n; i++) op.d[i] *= dinv[i];
    return;
  }
  if (diag_scale) {
    dinv.alloc(n); get_diagonal(*A.A, dinv.data());
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) dinv[i] = 1.0 / dinv[i];
    scaled = A.A->clone();
    scale_rows(scaled, dinv.data());
    As = &scaled;
  }
  if (poly_order == 0) {
    op.kind = InvOp::DIAG; op.d.alloc(n); op.d.fill(c[0]);
  } else if (sparsity_order == 0) {
    Arr<double> d(n); get_diagonal(*As, d.data());
    poly_diag_sparsity(d.data(), n, c, op);
  } else if (poly_order == 1 && sparsity_order == 1) {
    op.kind = InvOp::ASSEMBLED;
    op.mat = copy_plus_diag(*As);
    Csr& Mx = op.mat;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
      for (int64_t k = Mx.rp[i]; k < Mx.rp[i + 1]; k++) {
        Mx.v[k] *= c[1];
        if (Mx.ci[k] == i) Mx.v[k] += c[0];
      }
  } else if (sparsity_order < poly_order) {
    op.kind = InvOp::ASSEMBLED;
    // the default case (sparsity order 1) may run on the device, bit-identically (device/polyasm.hip)
    if (!(sparsity_order == 1 && g_setup_accel && g_setup_accel->poly_assemble(*As, c, op.mat, !diag_scale)))
      op.mat = mat_mult_powers_share_sparsity(*As, c, sparsity_order);
  } else {
    // unconstrained: sum of true matrix powers
    op.kind = InvOp::ASSEMBLED;
    op.mat = mat_mult_powers_share_sparsity(*As, c, poly_order);
  }
  if (diag_scale) {
    if (op.kind == InvOp::DIAG) { for (int64_t i = 0; i < n; i++) op.d[i] *= dinv[i]; }
    else scale_cols(op.mat, dinv.data());
  }
}

static void clamp_poly(int64_t global_rows, int& poly_order, int& sparsity_order) {
  if (poly_order + 1 > global_rows) poly_order = (int)(global_rows - 1);
  if (sparsity_order > poly_order) sparsity_order = poly_order;
}

void build_inverse(const OpView& A, int inverse_type, int poly_order, int sparsity_order,
                   bool matrix_free, bool diag_scale, uint64_t seed, InvOp& op, InvOp* asm_too) {
  const int64_t n = A.n;
  clamp_poly(n, poly_order, sparsity_order);
  Arr<double> d(n);
  if (A.diag) { for (int64_t i = 0; i < n; i++) d[i] = A.diag[i]; }
  else get_diagonal(*A.A, d.data());
  const bool is_gmres = inverse_type == INV_POWER || inverse_type == INV_ARNOLDI ||
                        inverse_type == INV_NEWTON || inverse_type == INV_NEWTON_NO_EXTRA;
  MatVec mv; mv.A = A; mv.dscale = is_gmres && diag_scale; mv.d = d.data();

  if (inverse_type == INV_POWER || inverse_type == INV_ARNOLDI) {
    std::vector<double> c = inverse_type == INV_POWER ? coeffs_power(mv, n, poly_order, seed)
                                                      : coeffs_arnoldi(mv, n, poly_order, seed);
    build_gmres_polynomial_inverse(A, poly_order, c, sparsity_order, matrix_free, diag_scale, op);
    if (matrix_free && asm_too)
      build_gmres_polynomial_inverse(A, poly_order, c, sparsity_order, false, diag_scale, *asm_too);
    return;
  }
  if (inverse_type == INV_NEWTON || inverse_type == INV_NEWTON_NO_EXTRA) {
    std::vector<double> re, im;
    roots_newton(mv, n, poly_order, inverse_type == INV_NEWTON, seed, re, im);
    // assembled form (build_gmres_polynomial_newton_inverse, src/Gmres_Poly_Newton.F90:1876-1990)
    auto assemble = [&](InvOp& out) {
      out = InvOp();
      Arr<double> dinv, dd(n);
      if (diag_scale) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) dd[i] = 1.0;            // diagonal of D^-1 A
        dinv.alloc(n);
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) dinv[i] = 1.0 / d[i];
      } else {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; i++) dd[i] = d[i];
      }
      if (A.diag || sparsity_order == 0) {
        newton_diag(dd.data(), n, re, im, out);
      } else {
        out.kind = InvOp::ASSEMBLED;
        if (diag_scale) {
          Csr scaled = A.A->clone();
          scale_rows(scaled, dinv.data());
          out.mat = mat_newton_share_sparsity(scaled, re, im, sparsity_order);
        } else out.mat = mat_newton_share_sparsity(*A.A, re, im, sparsity_order);
      }
      if (diag_scale) {
        if (out.kind == InvOp::DIAG) { for (int64_t i = 0; i < n; i++) out.d[i] *= dinv[i]; }
        else scale_cols(out.mat, dinv.data());
      }
    };
    if (!matrix_free) { assemble(op); return; }
    op.kind = InvOp::NEWTON; op.diag_scale = diag_scale; op.re = re; op.im = im;
    op.spmvs = newton_spmv_count(op.re, op.im);
    if (asm_too) assemble(*asm_too);
    return;
  }
  if (inverse_type == INV_NEUMANN) {
    std::vector<double> c(poly_order + 1, 1.0);
    if (matrix_free) {
      op.kind = InvOp::HORNER; op.coeffs = c; op.diag_scale = true; op.neumann = true; op.spmvs = poly_order;
      if (asm_too) build_inverse(A, INV_NEUMANN, poly_order, sparsity_order, false, false, seed, *asm_too, nullptr);
      return;
    }
    // assembled: q(I - D^-1 A) D^-1 (src/Neumann_Poly.F90:178-213)
    if (A.diag) {
      Arr<double> t(n); t.fill(0.0);   // I - D^-1 D = 0
      InvOp tmp; OpView tv; tv.diag = t.data(); tv.n = n;
      build_gmres_polynomial_inverse(tv, poly_order, c, sparsity_order, false, false, op);
      for (int64_t i = 0; i < n; i++) op.d[i] *= 1.0 / d[i];
      return;
    }
    Csr T = copy_plus_diag(*A.A);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
      for (int64_t k = T.rp[i]; k < T.rp[i + 1]; k++) {
        double v = -(T.v[k] * (1.0 / d[i]));
        if (T.ci[k] == i) v += 1.0;
        T.v[k] = v;
      }
    OpView tv; tv.A = &T; tv.n = n;
    build_gmres_polynomial_inverse(tv, poly_order, c, sparsity_order, false, false, op);
    Arr<double> dinv(n);
    for (int64_t i = 0; i < n; i++) dinv[i] = 1.0 / d[i];
    if (op.kind == InvOp::DIAG) { for (int64_t i = 0; i < n; i++) op.d[i] *= dinv[i]; }
    else scale_cols(op.mat, dinv.data());
    return;
  }
  if (inverse_type == INV_SAI || inverse_type == INV_ISAI) {
    // always assembled (src/Approx_Inverse_Setup.F90:465-482); a MATDIAGONAL operand is converted first
    op.kind = InvOp::ASSEMBLED;
    if (A.diag) { Csr D = csr_from_diag(A.diag, n); op.mat = sai_inverse(D, sparsity_order, inverse_type == INV_ISAI); }
    else op.mat = sai_inverse(*A.A, sparsity_order, inverse_type == INV_ISAI);
    return;
  }
  if (inverse_type == INV_JACOBI || inverse_type == INV_WJACOBI) {
    double weight = 1.0;
    if (inverse_type == INV_WJACOBI) {
      double norm_inf = 0.0;
      if (A.diag) norm_inf = 1.0;
      else {
        const Csr& M = *A.A;
#pragma omp parallel for schedule(static) reduction(max : norm_inf)
        for (int64_t i = 0; i < n; i++) {
          double s = 0.0, si = 1.0 / std::sqrt(std::fabs(d[i]));
          for (int64_t k = M.rp[i]; k < M.rp[i + 1]; k++)
            s += std::fabs(si * M.v[k] * (1.0 / std::sqrt(std::fabs(d[M.ci[k]]))));
          norm_inf = std::max(norm_inf, s);
        }
      }
      weight = 3.0 / (4.0 * norm_inf);
    }
    op.kind = InvOp::DIAG; op.d.alloc(n);
    for (int64_t i = 0; i < n; i++) op.d[i] = weight / d[i];
    return;
  }
  throw std::runtime_error("inverse type not supported by this build");
}

// ---------------------------------------------------------------------------------------
// Host application of an approximate inverse: used ONLY by the auto-truncation test of
// the setup phase (src/AIR_MG_Setup.F90:158-256), never by the solve phase.
// ---------------------------------------------------------------------------------------
void host_apply_inv(const InvOp& op, const OpView& A, const double* x, double* y) {
  const int64_t n = A.n;
  if (op.kind == InvOp::ASSEMBLED) { spmv(op.mat, x, y); return; }
  auto par = [&](auto&& f) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) f(i);
  };
  if (op.kind == InvOp::DIAG) { par([&](int64_t i) { y[i] = op.d[i] * x[i]; }); return; }
  Arr<double> d(n), rhs(n), t(n), t2(n), t3(n);
  if (A.diag) par([&](int64_t i) { d[i] = A.diag[i]; }); else get_diagonal(*A.A, d.data());
  MatVec plain; plain.A = A;
  auto mvop = [&](const double* in, double* out) {
    plain(in, out);
    if (op.diag_scale) {
      if (op.neumann) par([&](int64_t i) { out[i] = in[i] - out[i] / d[i]; });
      else par([&](int64_t i) { out[i] /= d[i]; });
    }
  };
  const double* r = x;
  if (op.diag_scale) { par([&](int64_t i) { rhs[i] = x[i] / d[i]; }); r = rhs.data(); }
  if (op.kind == InvOp::HORNER) {
    const auto& c = op.coeffs;
    par([&](int64_t i) { y[i] = c.back() * r[i]; });
    for (int k = (int)c.size() - 2; k >= 0; k--) {
      if (c[k] == 0.0) continue;
      par([&](int64_t i) { t[i] = y[i]; });
      mvop(t.data(), y);
      const double ck = c[k];
      par([&](int64_t i) { y[i] = ck * r[i] + y[i]; });
    }
    return;
  }
  // Newton
  const auto& re = op.re; const auto& im = op.im;
  const int nr = (int)re.size();
  par([&](int64_t i) { t[i] = r[i]; y[i] = 0.0; });
  int i = 0;
  while (i <= nr - 2) {
    if (im[i] == 0.0) {
      if (std::fabs(re[i]) < TOL_ZERO) { i++; continue; }
      const double a = 1.0 / re[i];
      par([&](int64_t q) { y[q] += a * t[q]; });
      mvop(t.data(), t2.data());
      par([&](int64_t q) { t[q] += -a * t2[q]; });
      i++;
    } else {
      const double sq = re[i] * re[i] + im[i] * im[i];
      if (sq < TOL_ZERO) { i += 2; continue; }
      const double tr = 2 * re[i], isq = 1.0 / sq;
      mvop(t.data(), t2.data());
      par([&](int64_t q) { t2[q] = tr * t[q] - t2[q]; y[q] += isq * t2[q]; });
      if (i <= nr - 3) {
        mvop(t2.data(), t3.data());
        par([&](int64_t q) { t[q] += -isq * t3[q]; });
      }
      i += 2;
    }
  }
  if (im[nr - 1] == 0.0 && std::fabs(re[nr - 1]) > TOL_ZERO) {
    const double a = 1.0 / re[nr - 1];
    par([&](int64_t q) { y[q] += a * t[q]; });
  }
}

}  // namespace airhip
