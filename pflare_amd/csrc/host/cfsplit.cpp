// CF splitting: PMISR (Luby rounds, smallest measure -> F) followed by the DDC cleanup.
// Follows src/PMISR_Module.F90:121-671, src/DDC_Module.F90:221-481,
// src/MatDiagDom.F90:98-270 and the driver src/CF_Splitting.F90:235-392 (CF_PMISR_DDC).
#include "air.hpp"
#include <limits>

namespace airhip {

static void pmisr(const Arr<int64_t>& rp, const Arr<int32_t>& ci, int64_t n, uint64_t seed,
                  int max_luby_steps, Arr<int32_t>& markers) {
  const uint64_t key = rng_key(seed);
  Arr<double> measure(n);
  Arr<uint8_t> assigned(n), in_set(n);
  markers.alloc(n);
  int64_t undecided = 0;
#pragma omp parallel for schedule(static) reduction(+ : undecided)
  for (int64_t i = 0; i < n; i++) {
    measure[i] = uniform01(key, (uint64_t)i) + (double)(rp[i + 1] - rp[i]);
    markers[i] = 0; assigned[i] = 0;
    // rows without strong neighbours go straight to F (PMISR_Module.F90:409-436)
    if (std::fabs(measure[i]) < 1.0) { assigned[i] = 1; markers[i] = F_POINT; }
    else undecided++;
  }
  int loops = 0;
  while (undecided != 0) {
    if (max_luby_steps > 0 && loops == max_luby_steps) break;
#pragma omp parallel for schedule(dynamic, 8192)
    for (int64_t i = 0; i < n; i++) {
      uint8_t in = 0;
      if (!assigned[i]) {
        in = 1;
        const double mi = measure[i];
        for (int64_t k = rp[i]; k < rp[i + 1]; k++) {
          int32_t j = ci[k];
          if (assigned[j]) continue;
          double mj = measure[j];
          if (mi > mj || (mi == mj && i > j)) { in = 0; break; }
        }
      }
      in_set[i] = in;
    }
    // record the new F points, then mark their neighbours as assigned (-> C at the end)
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
      if (in_set[i]) { assigned[i] = 1; markers[i] = F_POINT; }
#pragma omp parallel for schedule(dynamic, 8192)
    for (int64_t i = 0; i < n; i++)
      if (in_set[i])
        for (int64_t k = rp[i]; k < rp[i + 1]; k++) assigned[ci[k]] = 1;  // benign same-value race
    loops++;
    undecided = 0;
#pragma omp parallel for schedule(static) reduction(+ : undecided)
    for (int64_t i = 0; i < n; i++) undecided += assigned[i] ? 0 : 1;
  }
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) if (markers[i] == 0) markers[i] = C_POINT;
}

static void ddc(const Csr& A, double fraction_swap, Arr<int32_t>& markers) {
  if (fraction_swap == 0.0) return;
  const int64_t n = A.m;
  // ratio per F row (0 for non-F rows, never read)
  Arr<double> ratio(n);
  int64_t fine_size = 0;
#pragma omp parallel for schedule(static) reduction(+ : fine_size)
  for (int64_t i = 0; i < n; i++) {
    ratio[i] = 0.0;
    if (markers[i] != F_POINT) continue;
    fine_size++;
    double diag = 0.0, off = 0.0;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
      int32_t c = A.ci[k];
      if (markers[c] != F_POINT) continue;
      if (c == i) diag = std::fabs(A.v[k]); else off += std::fabs(A.v[k]);
    }
    if (diag != 0.0) ratio[i] = off / diag;
  }
  int64_t search_size = fraction_swap < 0 ? fine_size : (int64_t)((double)fine_size * fraction_swap);
  if (search_size <= 0) return;
  const int NB = 1000;
  auto bin_of = [&](double r) {
    double b = std::floor(r * NB) + 1.0;
    if (!(b == b) || b > (double)NB) return NB;   // NaN / overflow -> top bin
    int bi = (int)b;
    if (bi < 0) bi = NB;
    return std::min(bi, NB);
  };
  std::vector<int64_t> bins(NB + 1, 0);
#pragma omp parallel
  {
    std::vector<int64_t> lb(NB + 1, 0);
#pragma omp for schedule(static) nowait
    for (int64_t i = 0; i < n; i++) if (markers[i] == F_POINT) lb[bin_of(ratio[i])]++;
#pragma omp critical
    for (int b = 0; b <= NB; b++) bins[b] += lb[b];
  }
  double swap_val;
  if (fraction_swap < 0) swap_val = -fraction_swap;
  else {
    int64_t bin_sum = 0; int bb = NB;
    for (bb = NB; bb >= 1; bb--) { bin_sum += bins[bb]; if (bin_sum >= search_size) break; }
    if (bb < 1) bb = 1;
    swap_val = (double)(bb - 1) / (double)NB;
  }
  Arr<uint8_t> flip(n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++)
    flip[i] = (markers[i] == F_POINT && !(ratio[i] == 0.0 || ratio[i] < swap_val)) ? 1 : 0;
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) if (flip[i]) markers[i] = -markers[i];
}

// pmisr_existing_measure_implicit_transpose on A_ff + A_ff^T (src/PMISR_Module.F90): among the candidate F rows
// repeatedly take the local minima of `measure` (ties: the larger index loses) and block their neighbours; the rows
// taken become C points.  Returns the number of rows swapped.
static int64_t swap_independent_set(const Csr& A, const Arr<uint8_t>& cand, const Arr<double>& measure, Arr<int32_t>& markers) {
  const int64_t n = A.m;
  // adjacency of A_ff + A_ff^T (duplicates are harmless)
  Arr<int64_t> cnt(n + 1), rp(n + 1);
  for (int64_t i = 0; i <= n; i++) cnt[i] = 0;
  auto ff_edge = [&](int64_t i, int64_t k) { const int32_t c = A.ci[k]; return c != i && markers[c] == F_POINT; };
  for (int64_t i = 0; i < n; i++) {
    if (markers[i] != F_POINT) continue;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) if (ff_edge(i, k)) { cnt[i]++; cnt[A.ci[k]]++; }
  }
  rp[0] = 0;
  for (int64_t i = 0; i < n; i++) rp[i + 1] = rp[i] + cnt[i];
  Arr<int32_t> adj(std::max<int64_t>(rp[n], 1));
  for (int64_t i = 0; i < n; i++) cnt[i] = rp[i];
  for (int64_t i = 0; i < n; i++) {
    if (markers[i] != F_POINT) continue;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++)
      if (ff_edge(i, k)) { adj[cnt[i]++] = A.ci[k]; adj[cnt[A.ci[k]]++] = (int32_t)i; }
  }
  Arr<uint8_t> assigned(n), in_set(n), swap(n);
  int64_t undecided = 0;
#pragma omp parallel for schedule(static) reduction(+ : undecided)
  for (int64_t i = 0; i < n; i++) { swap[i] = 0; assigned[i] = cand[i] ? 0 : 1; undecided += cand[i] ? 1 : 0; }
  while (undecided != 0) {
#pragma omp parallel for schedule(dynamic, 8192)
    for (int64_t i = 0; i < n; i++) {
      uint8_t in = 0;
      if (!assigned[i]) {
        in = 1;
        const double mi = measure[i];
        for (int64_t k = rp[i]; k < rp[i + 1]; k++) {
          const int32_t j = adj[k];
          if (assigned[j]) continue;
          const double mj = measure[j];
          if (mi > mj || (mi == mj && i > j)) { in = 0; break; }
        }
      }
      in_set[i] = in;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) if (in_set[i]) { assigned[i] = 1; swap[i] = 1; }
#pragma omp parallel for schedule(dynamic, 8192)
    for (int64_t i = 0; i < n; i++)
      if (in_set[i]) for (int64_t k = rp[i]; k < rp[i + 1]; k++) assigned[adj[k]] = 1;   // benign same-value race
    undecided = 0;
#pragma omp parallel for schedule(static) reduction(+ : undecided)
    for (int64_t i = 0; i < n; i++) undecided += assigned[i] ? 0 : 1;
  }
  int64_t nsw = 0;
#pragma omp parallel for schedule(static) reduction(+ : nsw)
  for (int64_t i = 0; i < n; i++) if (swap[i]) { markers[i] = -markers[i]; nsw++; }
  return nsw;
}

// One pass of compatible-relaxation coarsening (src/CR_Splitting.F90:36-336): relax A_ff e = 0 from a random e with the
// (always assembled) A_ff inverse AIRG would smooth with; if the rate (||e_nu|| / ||e_0||)^(1/nu) is above the target,
// promote an independent set of the rows where the error stayed largest.  Returns the rate; n_swapped by reference.
static double cr_pass(const Csr& A, const AirOptions& o, Arr<int32_t>& markers, int64_t& n_swapped) {
  const int CR_NU = 10, CR_NU_POLY = 1;                 // src/Pflare_Parameters.F90:44-50
  const double CR_CANDIDATE = 0.1;                      // :62
  const int64_t n = A.m;
  n_swapped = 0;
  Arr<int32_t> pos(n);
  std::vector<int32_t> F;
  for (int64_t i = 0; i < n; i++) { pos[i] = -1; if (markers[i] == F_POINT) { pos[i] = (int32_t)F.size(); F.push_back((int32_t)i); } }
  const int64_t nf = (int64_t)F.size();
  if (nf == 0) return 0.0;
  Csr Aff_store;
  const Csr* Aff = &A;
  if (nf != n) {
    Aff_store.m = Aff_store.n = nf; Aff_store.rp.alloc(nf + 1);
    Aff_store.rp[0] = 0;
    for (int64_t r = 0; r < nf; r++) {
      int64_t c = 0;
      for (int64_t k = A.rp[F[r]]; k < A.rp[F[r] + 1]; k++) c += pos[A.ci[k]] >= 0 ? 1 : 0;
      Aff_store.rp[r + 1] = Aff_store.rp[r] + c;
    }
    Aff_store.ci.alloc(Aff_store.rp[nf]); Aff_store.v.alloc(Aff_store.rp[nf]);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nf; r++) {
      int64_t q = Aff_store.rp[r];
      for (int64_t k = A.rp[F[r]]; k < A.rp[F[r] + 1]; k++)
        if (pos[A.ci[k]] >= 0) { Aff_store.ci[q] = pos[A.ci[k]]; Aff_store.v[q] = A.v[k]; q++; }
    }
    Aff = &Aff_store;
  }
  const bool use_poly = o.inverse_type != INV_WJACOBI && o.inverse_type != INV_JACOBI && nf > (int64_t)o.poly_order + 1;
  const int nu = use_poly ? CR_NU_POLY : CR_NU;
  InvOp M;
  std::vector<uint8_t> forced((size_t)nf, 0);
  if (use_poly) {
    OpView v; v.A = Aff; v.n = nf;
    build_inverse(v, o.inverse_type, o.poly_order, o.inverse_sparsity_order, false, o.diag_scale_polys, o.seed, M, nullptr);
  } else {
    // weighted Jacobi with exactly zero diagonals sanitised (those rows are promoted first)
    M.kind = InvOp::DIAG; M.d.alloc(nf);
    Arr<double> d(nf);
    get_diagonal(*Aff, d.data());
    for (int64_t i = 0; i < nf; i++) if (d[i] == 0.0) { d[i] = 1.0; forced[i] = 1; }
    double weight = 1.0;
    if (o.inverse_type != INV_JACOBI) {
      double norm_inf = 0.0;
#pragma omp parallel for schedule(static) reduction(max : norm_inf)
      for (int64_t i = 0; i < nf; i++) {
        double sacc = 0.0; const double si = 1.0 / std::sqrt(std::fabs(d[i]));
        for (int64_t k = Aff->rp[i]; k < Aff->rp[i + 1]; k++)
          sacc += std::fabs(si * Aff->v[k] * (1.0 / std::sqrt(std::fabs(d[Aff->ci[k]]))));
        norm_inf = std::max(norm_inf, sacc);
      }
      if (norm_inf != 0.0) weight = 3.0 / (4.0 * norm_inf);
    }
    for (int64_t i = 0; i < nf; i++) M.d[i] = forced[i] ? 0.0 : weight / d[i];
  }
  const uint64_t key = rng_key(o.seed + 15485863);
  Arr<double> e(nf), w1(nf), w2(nf);
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nf; r++) e[r] = uniform01(key, (uint64_t)F[r]);
  const double norm_e0 = std::sqrt(det_sum(0, nf, [&](int64_t r) { return e[r] * e[r]; }));
  OpView av; av.A = Aff; av.n = nf;
  for (int sweep = 0; sweep < nu; sweep++) {
    spmv(*Aff, e.data(), w1.data());
    host_apply_inv(M, av, w1.data(), w2.data());
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < nf; r++) e[r] -= w2[r];
  }
  const double norm_enu = std::sqrt(det_sum(0, nf, [&](int64_t r) { return e[r] * e[r]; }));
  const double rate = norm_e0 != 0.0 ? std::pow(norm_enu / norm_e0, 1.0 / (double)nu) : 0.0;
  if (!(rate > o.strong_threshold)) return rate;
  double e_inf = 0.0;
  for (int64_t r = 0; r < nf; r++) e_inf = std::max(e_inf, std::fabs(e[r]));
  Arr<uint8_t> cand(n); Arr<double> measure(n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) { cand[i] = 0; measure[i] = std::numeric_limits<double>::max(); }
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nf; r++) {
    const double cm = forced[r] ? 2.0 : (e_inf != 0.0 ? std::fabs(e[r]) / e_inf : 0.0);
    if (cm >= CR_CANDIDATE) { cand[F[r]] = 1; measure[F[r]] = 10.0 - (cm - uniform01(key, (uint64_t)F[r]) / 1e10); }
  }
  n_swapped = swap_independent_set(A, cand, measure, markers);
  return rate;
}

// The max_dd_ratio path of ddc / ddc_cpu (src/DDC_Module.F90:56-82, 301-392): while some F row is less diagonally
// dominant in A_ff than `target`, swap to C an independent set (in A_ff + A_ff^T) of the worst of those rows - Luby
// rounds on the measure max(10, 2 max_ratio) - (ratio - rand * 1e-10), smallest first, rows already below the target
// excluded.  Returns the largest ratio found before the swap.
static double ddc_diag_dom(const Csr& A, double target, uint64_t seed, Arr<int32_t>& markers) {
  const int64_t n = A.m;
  Arr<double> ratio(n);
  double max_ratio = 0.0;
#pragma omp parallel for schedule(static) reduction(max : max_ratio)
  for (int64_t i = 0; i < n; i++) {
    ratio[i] = 0.0;
    if (markers[i] != F_POINT) continue;
    double diag = 0.0, off = 0.0;
    for (int64_t k = A.rp[i]; k < A.rp[i + 1]; k++) {
      const int32_t c = A.ci[k];
      if (markers[c] != F_POINT) continue;
      if (c == i) diag = std::fabs(A.v[k]); else off += std::fabs(A.v[k]);
    }
    if (diag != 0.0) ratio[i] = off / diag;
    max_ratio = std::max(max_ratio, ratio[i]);
  }
  if (max_ratio < target) return max_ratio;
  const uint64_t key = rng_key(seed + 104729);
  const double base = std::max(10.0, 2.0 * max_ratio);
  Arr<double> measure(n);
  Arr<uint8_t> cand(n);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    cand[i] = (markers[i] == F_POINT && ratio[i] >= target) ? 1 : 0;
    measure[i] = base - (ratio[i] - uniform01(key, (uint64_t)i) / 1e10);
  }
  swap_independent_set(A, cand, measure, markers);
  return max_ratio;
}

void compute_cf_splitting(const Csr& A, const AirOptions& o, Arr<int32_t>& markers) {
  if (o.cf_splitting_type == 2) {
    // CF_CR (src/CF_Splitting.F90:303-340): everything starts as F, strong_threshold carries the target CR rate
    markers.alloc(A.m);
    for (int64_t i = 0; i < A.m; i++) markers[i] = F_POINT;
    for (int its = 0; its < 100; its++) {             // PFLARE_CR_MAX_ITS
      int64_t nsw = 0;
      const double rate = cr_pass(A, o, markers, nsw);
      if (rate <= o.strong_threshold || nsw == 0) break;
    }
    return;
  }
  Arr<int64_t> rp; Arr<int32_t> ci;
  strength_graph(A, o.strong_threshold, !o.symmetric, rp, ci);
  pmisr(rp, ci, A.m, o.seed, o.max_luby_steps, markers);
  rp.release(); ci.release();
  if (o.strong_threshold == 0.0) return;
  if (o.cf_splitting_type == 1) {
    // CF_DIAG_DOM (src/CF_Splitting.F90:361-392): repeat until every F row reaches the ratio given by strong_threshold
    while (ddc_diag_dom(A, o.strong_threshold, o.seed, markers) >= o.strong_threshold) {}
    return;
  }
  for (int it = 0; it < o.ddc_its; it++) ddc(A, o.ddc_fraction, markers);
}

}  // namespace airhip
