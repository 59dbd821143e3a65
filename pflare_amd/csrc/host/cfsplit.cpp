// CF splitting: PMISR (Luby rounds, smallest measure -> F) followed by the DDC cleanup.
// Follows src/PMISR_Module.F90:121-671, src/DDC_Module.F90:221-481,
// src/MatDiagDom.F90:98-270 and the driver src/CF_Splitting.F90:235-392 (CF_PMISR_DDC).
#include "air.hpp"

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
  const uint64_t key = rng_key(seed + 104729);
  const double base = std::max(10.0, 2.0 * max_ratio);
  Arr<double> measure(n);
  Arr<uint8_t> assigned(n), in_set(n), swap(n);
  int64_t undecided = 0;
#pragma omp parallel for schedule(static) reduction(+ : undecided)
  for (int64_t i = 0; i < n; i++) {
    swap[i] = 0;
    const bool cand = markers[i] == F_POINT && ratio[i] >= target;
    assigned[i] = cand ? 0 : 1;
    measure[i] = base - (ratio[i] - uniform01(key, (uint64_t)i) / 1e10);
    undecided += cand ? 1 : 0;
  }
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
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) if (swap[i]) markers[i] = -markers[i];
  return max_ratio;
}

void compute_cf_splitting(const Csr& A, const AirOptions& o, Arr<int32_t>& markers) {
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
