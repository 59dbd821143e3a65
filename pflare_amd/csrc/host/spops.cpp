#include "spops.hpp"
#include <limits>

namespace airhip {

// ---------------------------------------------------------------------------------------
// Generators
// ---------------------------------------------------------------------------------------
Csr gen_adv_1d(int64_t n) {
  Csr A; A.m = A.n = n;
  A.rp.alloc(n + 1);
  int64_t nnz = 2 * n - 1;
  A.ci.alloc(nnz); A.v.alloc(nnz);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    if (i == 0) { A.rp[0] = 0; A.ci[0] = 0; A.v[0] = 1.0; }
    else {
      int64_t o = 2 * i - 1;
      A.rp[i] = o;
      A.ci[o] = (int32_t)(i - 1); A.v[o] = -1.0;
      A.ci[o + 1] = (int32_t)i;   A.v[o + 1] = 1.0;
    }
  }
  A.rp[n] = nnz;
  return A;
}

namespace {
// Adds (col,val) to a tiny sorted row buffer with ADD_VALUES semantics; zero values are
// never inserted (MAT_IGNORE_ZERO_ENTRIES, tests/adv_diff_fd.c:173).
struct RowBuf {
  int64_t c[8]; double v[8]; int n = 0;
  inline void add(int64_t col, double val) {
    if (val == 0.0) return;
    for (int k = 0; k < n; k++) if (c[k] == col) { v[k] += val; return; }
    int k = n++;
    while (k > 0 && c[k - 1] > col) { c[k] = c[k - 1]; v[k] = v[k - 1]; k--; }
    c[k] = col; v[k] = val;
  }
};
}  // namespace

Csr gen_adv_diff_fd(const FdOptions& o, const int64_t*) {
  const int dim = o.dim;
  const int64_t M = o.M, N = o.N, P = (dim == 3 ? o.P : 1);
  double vel[3] = {o.u, o.v, o.w};
  if (o.unit_velocity) {
    double mag = 0; for (int d = 0; d < dim; d++) mag += vel[d] * vel[d];
    mag = std::sqrt(mag);
    if (mag > 1e-12) for (int d = 0; d < dim; d++) vel[d] /= mag;
  }
  const double u = vel[0], v = vel[1], w = vel[2];
  const double alpha = o.alpha;
  const bool nondim = o.adv_nondim < 0 ? (alpha == 0.0) : (o.adv_nondim != 0);
  const double Hx = o.L_x / (double)(M - 1), Hy = o.L_y / (double)(N - 1);
  const double Hz = dim == 3 ? o.L_z / (double)(P - 1) : 1.0;
  const double HxdHy = Hx / Hy, HydHx = Hy / Hx;
  double adv_x_scale = Hx, adv_y_scale = Hy;
  if (nondim) { adv_x_scale = 1.0; adv_y_scale = HydHx; }
  const double HyHz_Hx = Hy * Hz / Hx, HxHz_Hy = Hx * Hz / Hy, HxHy_Hz = Hx * Hy / Hz;
  double adv_yz = Hy * Hz, adv_xz = Hx * Hz, adv_xy = Hx * Hy;
  if (nondim) { adv_yz = HyHz_Hx / Hx; adv_xz = Hz / Hx; adv_xy = Hy / Hx; }

  const int64_t n = M * N * P;
  if (n > (int64_t)std::numeric_limits<int32_t>::max()) throw std::runtime_error("grid too large for int32 indices");
  auto emit = [&](int64_t row, RowBuf& rb) {
    rb.n = 0;
    int64_t i = row % M, j = (row / M) % N, k = row / (M * N);
    if (dim == 2) {
      bool inflow = (i == 0 || j == 0);
      bool bnd = inflow || i == M - 1 || j == N - 1;
      if (bnd) {
        if (inflow) { rb.add(row, 1.0); return; }
        if (alpha == 0.0) {
          rb.add(row - M, -v * adv_x_scale);
          rb.add(row - 1, -u * adv_y_scale);
          rb.add(row, u * adv_y_scale + v * adv_x_scale);
        } else rb.add(row, 1.0);
        return;
      }
      if (alpha != 0.0) {
        rb.add(row - M, -alpha * HxdHy);
        rb.add(row - 1, -alpha * HydHx);
        rb.add(row, alpha * 2.0 * (HxdHy + HydHx));
        rb.add(row + 1, -alpha * HydHx);
        rb.add(row + M, -alpha * HxdHy);
      }
      if (u != 0.0 || v != 0.0) {
        rb.add(row - M, -v * adv_x_scale);
        rb.add(row - 1, -u * adv_y_scale);
        rb.add(row, u * adv_y_scale + v * adv_x_scale);
      }
      return;
    }
    const int64_t MN = M * N;
    bool inflow = (i == 0 || j == 0 || k == 0);
    bool bnd = inflow || i == M - 1 || j == N - 1 || k == P - 1;
    if (bnd) {
      if (inflow) { rb.add(row, 1.0); return; }
      if (alpha == 0.0) {
        rb.add(row - MN, -w * adv_xy);
        rb.add(row - M, -v * adv_xz);
        rb.add(row - 1, -u * adv_yz);
        rb.add(row, u * adv_yz + v * adv_xz + w * adv_xy);
      } else rb.add(row, 1.0);
      return;
    }
    if (alpha != 0.0) {
      rb.add(row - MN, -alpha * HxHy_Hz);
      rb.add(row - M, -alpha * HxHz_Hy);
      rb.add(row - 1, -alpha * HyHz_Hx);
      rb.add(row, alpha * 2.0 * (HyHz_Hx + HxHz_Hy + HxHy_Hz));
      rb.add(row + 1, -alpha * HyHz_Hx);
      rb.add(row + M, -alpha * HxHz_Hy);
      rb.add(row + MN, -alpha * HxHy_Hz);
    }
    if (u != 0.0 || v != 0.0 || w != 0.0) {
      rb.add(row - MN, -w * adv_xy);
      rb.add(row - M, -v * adv_xz);
      rb.add(row - 1, -u * adv_yz);
      rb.add(row, u * adv_yz + v * adv_xz + w * adv_xy);
    }
  };

  Csr A; A.m = A.n = n;
  A.rp.alloc(n + 1);
  Arr<int64_t> cnt(n);
#pragma omp parallel
  {
    RowBuf rb;
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n; r++) { emit(r, rb); cnt[r] = rb.n; }
  }
  exclusive_scan(cnt.data(), n, A.rp.data());
  A.ci.alloc(A.rp[n]); A.v.alloc(A.rp[n]);
#pragma omp parallel
  {
    RowBuf rb;
#pragma omp for schedule(static)
    for (int64_t r = 0; r < n; r++) {
      emit(r, rb);
      int64_t o = A.rp[r];
      for (int k = 0; k < rb.n; k++) { A.ci[o + k] = (int32_t)rb.c[k]; A.v[o + k] = rb.v[k]; }
    }
  }
  return A;
}

// ---------------------------------------------------------------------------------------
// remove_small_from_sparse
// ---------------------------------------------------------------------------------------
Csr remove_small(const Csr& A, double tol, int rel, bool lump, int drop_diag) {
  const int64_t m = A.m;
  const double BIG = std::numeric_limits<double>::max();
  Arr<double> rowtol(m);
  Arr<int64_t> cnt(m);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    double t = tol;
    if (rel == 1) {
      double mx = 0.0;
      for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) mx = std::max(mx, std::fabs(A.v[k]));
      t = tol * mx;
    } else if (rel == -1) {
      double mx = -BIG;
      for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++)
        if (A.ci[k] != i) mx = std::max(mx, std::fabs(A.v[k]));
      t = tol * mx;
    }
    rowtol[i] = t;
    int64_t c = 0; bool have_diag = false, need_diag = false;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
      bool isd = (A.ci[k] == i);
      if (std::fabs(A.v[k]) >= t) {
        if (drop_diag == -1 && isd) continue;
        c++; if (isd) have_diag = true;
      } else if (lump || (drop_diag == 0 && isd)) need_diag = true;
    }
    if (need_diag && !have_diag) c++;
    cnt[i] = c;
  }
  Csr B; B.m = A.m; B.n = A.n; B.rp.alloc(m + 1);
  exclusive_scan(cnt.data(), m, B.rp.data());
  B.ci.alloc(B.rp[m]); B.v.alloc(B.rp[m]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    const double t = rowtol[i];
    // first the kept diagonal value (if any), then the lumped entries in column order
    double dsum = 0.0; bool have_diag = false, need_diag = false;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++)
      if (A.ci[k] == i && std::fabs(A.v[k]) >= t && drop_diag != -1) { dsum = A.v[k]; have_diag = true; }
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
      bool isd = (A.ci[k] == i);
      if (std::fabs(A.v[k]) >= t) continue;
      if (lump || (drop_diag == 0 && isd)) { dsum += A.v[k]; need_diag = true; }
    }
    bool write_diag = have_diag || need_diag;
    int64_t o = B.rp[i];
    bool diag_written = false;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
      int32_t c = A.ci[k];
      bool isd = (c == i);
      if (write_diag && !diag_written && c >= i) {
        B.ci[o] = (int32_t)i; B.v[o] = dsum; o++; diag_written = true;
      }
      if (isd) continue;  // handled through dsum (or dropped)
      if (std::fabs(A.v[k]) >= t) { B.ci[o] = c; B.v[o] = A.v[k]; o++; }
    }
    if (write_diag && !diag_written) { B.ci[o] = (int32_t)i; B.v[o] = dsum; o++; }
  }
  return B;
}

// ---------------------------------------------------------------------------------------
// Strength graph
// ---------------------------------------------------------------------------------------
void strength_graph(const Csr& A, double theta, bool symmetrize, Arr<int64_t>& rp, Arr<int32_t>& ci) {
  const int64_t m = A.m;
  const double BIG = std::numeric_limits<double>::max();
  // directed strength pattern S
  Arr<int64_t> cnt(m);
  Arr<double> thr(m);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    double mx = -BIG;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++)
      if (A.ci[k] != i) mx = std::max(mx, std::fabs(A.v[k]));
    double t = theta * mx;
    thr[i] = t;
    int64_t c = 0;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++)
      if (A.ci[k] != i && std::fabs(A.v[k]) >= t) c++;
    cnt[i] = c;
  }
  Arr<int64_t> srp(m + 1);
  exclusive_scan(cnt.data(), m, srp.data());
  Arr<int32_t> sci(srp[m]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    int64_t o = srp[i];
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++)
      if (A.ci[k] != i && std::fabs(A.v[k]) >= thr[i]) sci[o++] = A.ci[k];
  }
  if (!symmetrize) { rp = std::move(srp); ci = std::move(sci); return; }
  // transpose pattern
  Arr<int64_t> tc(m); tc.fill(0);
  const int64_t snnz = srp[m];
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < snnz; k++) {
#pragma omp atomic
    tc[sci[k]]++;
  }
  Arr<int64_t> trp(m + 1);
  exclusive_scan(tc.data(), m, trp.data());
  Arr<int64_t> cur(m);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) cur[i] = trp[i];
  Arr<int32_t> tci(snnz);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++)
    for (int64_t k = srp[i]; k < srp[i + 1]; k++) {
      int64_t pos;
      int32_t c = sci[k];
#pragma omp atomic capture
      pos = cur[c]++;
      tci[pos] = (int32_t)i;
    }
#pragma omp parallel for schedule(dynamic, 4096)
  for (int64_t i = 0; i < m; i++) std::sort(tci.data() + trp[i], tci.data() + trp[i + 1]);
  // union of S and S^T rows
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    int64_t a = srp[i], ae = srp[i + 1], b = trp[i], be = trp[i + 1], c = 0;
    while (a < ae && b < be) {
      if (sci[a] == tci[b]) { a++; b++; } else if (sci[a] < tci[b]) a++; else b++;
      c++;
    }
    cnt[i] = c + (ae - a) + (be - b);
  }
  rp.alloc(m + 1);
  exclusive_scan(cnt.data(), m, rp.data());
  ci.alloc(rp[m]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    int64_t a = srp[i], ae = srp[i + 1], b = trp[i], be = trp[i + 1], o = rp[i];
    while (a < ae && b < be) {
      if (sci[a] == tci[b]) { ci[o++] = sci[a]; a++; b++; }
      else if (sci[a] < tci[b]) ci[o++] = sci[a++];
      else ci[o++] = tci[b++];
    }
    while (a < ae) ci[o++] = sci[a++];
    while (b < be) ci[o++] = tci[b++];
  }
}

// ---------------------------------------------------------------------------------------
// F/C split extraction
// ---------------------------------------------------------------------------------------
void split_fc(const Csr& A, const int32_t* markers, const int32_t* newidx,
              const Arr<int32_t>& F, const Arr<int32_t>& C,
              Csr& A_ff, Csr& A_fc, Csr& A_cf, Csr* A_cc) {
  const int64_t nf = F.size(), nc = C.size();
  auto extract = [&](const Arr<int32_t>& rows, Csr& X, int xmark, int64_t xn, Csr* Y, int64_t yn) {
    const int64_t mr = rows.size();
    Arr<int64_t> cx(mr), cy(Y ? mr : 0);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < mr; r++) {
      int64_t i = rows[r], a = 0, b = 0;
      for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
        if (markers[A.ci[k]] == xmark) a++; else b++;
      }
      cx[r] = a; if (Y) cy[r] = b;
    }
    X.m = mr; X.n = xn; X.rp.alloc(mr + 1);
    exclusive_scan(cx.data(), mr, X.rp.data());
    X.ci.alloc(X.rp[mr]); X.v.alloc(X.rp[mr]);
    if (Y) {
      Y->m = mr; Y->n = yn; Y->rp.alloc(mr + 1);
      exclusive_scan(cy.data(), mr, Y->rp.data());
      Y->ci.alloc(Y->rp[mr]); Y->v.alloc(Y->rp[mr]);
    }
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < mr; r++) {
      int64_t i = rows[r], ox = X.rp[r], oy = Y ? Y->rp[r] : 0;
      for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
        int32_t c = A.ci[k];
        if (markers[c] == xmark) { X.ci[ox] = newidx[c]; X.v[ox] = A.v[k]; ox++; }
        else if (Y) { Y->ci[oy] = newidx[c]; Y->v[oy] = A.v[k]; oy++; }
      }
    }
  };
  // rows in F: F columns -> A_ff, C columns -> A_fc.  newidx is monotone inside each set,
  // so the sub-rows stay sorted.
  extract(F, A_ff, F_POINT, nf, &A_fc, nc);
  extract(C, A_cf, F_POINT, nf, A_cc, nc);
}

// ---------------------------------------------------------------------------------------
// SpGEMM
// ---------------------------------------------------------------------------------------
namespace {
// Per-thread sparse accumulator: open-addressing hash on the column index.  Values are
// accumulated in the order the candidates arrive (k ascending), i.e. exactly the order of a
// sequential Gustavson sparse accumulator, so results do not depend on the thread count.
struct RowAcc {
  std::vector<int32_t> key; std::vector<double> val; std::vector<int32_t> used;
  uint32_t mask = 0;
  void ensure(int64_t want) {
    uint32_t cap = 64;
    while ((int64_t)cap < 2 * want) cap <<= 1;
    if (cap > key.size()) { key.assign(cap, -1); val.assign(cap, 0.0); mask = cap - 1; }
  }
  inline void add(int32_t c, double x) {
    uint32_t h = ((uint32_t)c * 2654435761u) & mask;
    for (;;) {
      int32_t k = key[h];
      if (k == c) { val[h] += x; return; }
      if (k < 0) { key[h] = c; val[h] = x; used.push_back((int32_t)h); return; }
      h = (h + 1) & mask;
    }
  }
  // sorted (col,val) pairs of the current row into out, then reset
  void flush(std::vector<std::pair<int32_t, double>>& out) {
    out.clear();
    for (int32_t h : used) { out.emplace_back(key[h], val[h]); key[h] = -1; }
    used.clear();
    std::sort(out.begin(), out.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
  }
};

// remove_small semantics applied to one sorted row (see remove_small above)
inline void filter_row(int64_t i, std::vector<std::pair<int32_t, double>>& row, const SpgemmDrop& d,
                       std::vector<std::pair<int32_t, double>>& tmp) {
  if (d.negate) for (auto& e : row) e.second = -e.second;
  if (!d.enabled) return;
  double mx = 0.0;
  for (auto& e : row) mx = std::max(mx, std::fabs(e.second));
  const double t = d.tol * mx;       // relative to the row max including the diagonal
  double dsum = 0.0; bool have_diag = false, need_diag = false;
  for (auto& e : row) if (e.first == i && std::fabs(e.second) >= t) { dsum = e.second; have_diag = true; }
  for (auto& e : row) {
    if (std::fabs(e.second) >= t) continue;
    if (d.lump || e.first == i) { dsum += e.second; need_diag = true; }
  }
  const bool write_diag = have_diag || need_diag;
  tmp.clear();
  bool diag_written = false;
  for (auto& e : row) {
    if (write_diag && !diag_written && e.first >= i) { tmp.emplace_back((int32_t)i, dsum); diag_written = true; }
    if (e.first == i) continue;
    if (std::fabs(e.second) >= t) tmp.push_back(e);
  }
  if (write_diag && !diag_written) tmp.emplace_back((int32_t)i, dsum);
  row.swap(tmp);
}
}  // namespace

Csr spgemm(const Csr& A, const Csr& B, const SpgemmDrop& drop) {
  if (A.n != B.m) throw std::runtime_error("spgemm: shape mismatch");
  const int64_t m = A.m;
  const int nt = omp_get_max_threads();
  Csr C; C.m = m; C.n = B.n; C.rp.alloc(m + 1);
  Arr<int64_t> cnt(m);
  std::vector<std::vector<int32_t>> tci(nt);
  std::vector<std::vector<double>> tv(nt);
  std::vector<int64_t> tlo(nt + 1);
  // weight rows by candidate count so the static chunks are balanced
  Arr<int64_t> work(m + 1);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    int64_t w = 1;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) { int32_t j = A.ci[k]; w += B.rp[j + 1] - B.rp[j]; }
    cnt[i] = w;
  }
  exclusive_scan(cnt.data(), m, work.data());
  const int64_t total = work[m];
  for (int t = 0; t <= nt; t++) {
    if (t == nt) { tlo[t] = m; continue; }
    int64_t target = total / nt * t;
    tlo[t] = std::lower_bound(work.data(), work.data() + m, target) - work.data();
  }
  tlo[0] = 0;
#pragma omp parallel num_threads(nt)
  {
    int t = omp_get_thread_num();
    RowAcc acc;
    std::vector<std::pair<int32_t, double>> row, tmp;
    std::vector<int32_t>& oc = tci[t];
    std::vector<double>& ov = tv[t];
    for (int64_t i = tlo[t]; i < tlo[t + 1]; i++) {
      const int64_t ub = std::min<int64_t>(work[i + 1] - work[i], B.n);
      acc.ensure(ub);
      for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
        const int32_t j = A.ci[k]; const double a = A.v[k];
        for (int64_t q = B.rp[j]; q < B.rp[j + 1]; q++) acc.add(B.ci[q], a * B.v[q]);
      }
      acc.flush(row);
      filter_row(i, row, drop, tmp);
      for (auto& e : row) { oc.push_back(e.first); ov.push_back(e.second); }
      cnt[i] = (int64_t)row.size();
    }
  }
  exclusive_scan(cnt.data(), m, C.rp.data());
  C.ci.alloc(C.rp[m]); C.v.alloc(C.rp[m]);
#pragma omp parallel num_threads(nt)
  {
    int t = omp_get_thread_num();
    int64_t base = C.rp[tlo[t]];
    if (!tci[t].empty()) {
      std::memcpy(C.ci.data() + base, tci[t].data(), tci[t].size() * sizeof(int32_t));
      std::memcpy(C.v.data() + base, tv[t].data(), tv[t].size() * sizeof(double));
    }
    std::vector<int32_t>().swap(tci[t]);
    std::vector<double>().swap(tv[t]);
  }
  return C;
}

Csr fused_first_operator(const Csr& A_ff, const Csr& A_fc, const int32_t* wcol) {
  const int64_t nf = A_fc.m, nc = A_fc.n;
  if (A_ff.m != nf || A_ff.n != nf) throw std::runtime_error("fused_first_operator: A_ff / A_fc shape mismatch");
  // [A_ff | A_fc]: the entries of A_ff, then those of A_fc with their columns moved behind the F columns
  Csr cat; cat.m = nf; cat.n = nf + nc;
  cat.rp.alloc(nf + 1);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i <= nf; i++) cat.rp[i] = A_ff.rp[i] + A_fc.rp[i];
  cat.ci.alloc(cat.rp[nf]); cat.v.alloc(cat.rp[nf]);
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < nf; i++) {
    int64_t o = cat.rp[i];
    for (int64_t k = A_ff.rp[i]; k < A_ff.rp[i + 1]; k++, o++) { cat.ci[o] = A_ff.ci[k]; cat.v[o] = A_ff.v[k]; }
    for (int64_t k = A_fc.rp[i]; k < A_fc.rp[i + 1]; k++, o++) { cat.ci[o] = (int32_t)(A_fc.ci[k] + nf); cat.v[o] = A_fc.v[k]; }
  }
  // [W; I]
  Csr P; P.m = nf + nc; P.n = nc;
  P.rp.alloc(P.m + 1);
  {
    Arr<int64_t> cnt(P.m);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < P.m; r++) cnt[r] = r < nf ? (wcol[r] >= 0 ? 1 : 0) : 1;
    exclusive_scan(cnt.data(), P.m, P.rp.data());
  }
  P.ci.alloc(P.rp[P.m]); P.v.alloc(P.rp[P.m]);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < P.m; r++) {
    const int32_t c = r < nf ? wcol[r] : (int32_t)(r - nf);
    if (c >= 0) { P.ci[P.rp[r]] = c; P.v[P.rp[r]] = 1.0; }
  }
  return spgemm(cat, P);
}

void spmv(const Csr& A, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.m; i++) {
    double s = 0.0;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) s += A.v[k] * x[A.ci[k]];
    y[i] = s;
  }
}

void spmv_dscale(const Csr& A, const double* d, const double* x, double* y) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.m; i++) {
    double s = 0.0;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) s += A.v[k] * x[A.ci[k]];
    y[i] = s / d[i];
  }
}

Csr copy_plus_diag(const Csr& A) {
  const int64_t m = A.m;
  Arr<int64_t> cnt(m);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    bool has = false;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) if (A.ci[k] == i) { has = true; break; }
    cnt[i] = (A.rp[i + 1] - A.rp[i]) + (has ? 0 : 1);
  }
  Csr B; B.m = A.m; B.n = A.n; B.rp.alloc(m + 1);
  exclusive_scan(cnt.data(), m, B.rp.data());
  B.ci.alloc(B.rp[m]); B.v.alloc(B.rp[m]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < m; i++) {
    int64_t o = B.rp[i]; bool placed = false;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
      int32_t c = A.ci[k];
      if (!placed && c >= i) {
        if (c != i) { B.ci[o] = (int32_t)i; B.v[o] = 0.0; o++; }
        placed = true;
      }
      B.ci[o] = c; B.v[o] = A.v[k]; o++;
    }
    if (!placed) { B.ci[o] = (int32_t)i; B.v[o] = 0.0; o++; }
  }
  return B;
}

void get_diagonal(const Csr& A, double* d) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.m; i++) {
    double x = 0.0;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) if (A.ci[k] == i) { x = A.v[k]; break; }
    d[i] = x;
  }
}

bool is_diag_only(const Csr& A) {
  if (A.nnz() != A.m) return false;
  int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
  for (int64_t i = 0; i < A.m; i++)
    if (A.rp[i + 1] - A.rp[i] != 1 || A.ci[A.rp[i]] != i) bad++;
  return bad == 0;
}

Csr csr_from_diag(const double* d, int64_t n) {
  Csr A; A.m = A.n = n; A.rp.alloc(n + 1); A.ci.alloc(n); A.v.alloc(n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) { A.rp[i] = i; A.ci[i] = (int32_t)i; A.v[i] = d[i]; }
  A.rp[n] = n;
  return A;
}

void scale_cols(Csr& A, const double* s) {
  const int64_t nnz = A.nnz();
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < nnz; k++) A.v[k] *= s[A.ci[k]];
}

void scale_rows(Csr& A, const double* s) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < A.m; i++)
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) A.v[k] *= s[i];
}

}  // namespace airhip
