// CF splitting on the device: strength graph, PMISR (Luby rounds) and the DDC cleanup of the default
// CF_PMISR_DDC splitting, for a level operator that is already resident (it is the R*A*P product of the previous
// level).  Same rules and the same arithmetic as host/cfsplit.cpp + host/spops.cpp::strength_graph - which restate
// src/SAbs.F90:62-88, src/PETSc_Helper.F90:336-399, src/PMISR_Module.F90:121-671, src/DDC_Module.F90:221-481,
// src/MatDiagDom.F90:98-270 - so the markers are IDENTICAL to the host's (tests/test_gpu_setup.py); the reference's
// device twins are src/PMISR_Modulek.kokkos.cxx, src/DDC_Modulek.kokkos.cxx, src/MatDiagDomk.kokkos.cxx.
//
// What makes identity cheap: the random part of the measure is a counter-based integer hash (splitmix64), the Luby
// decision of a row is a conjunction over its neighbours (their order does not matter), "assigned" only ever goes
// 0 -> 1, the diagonal-dominance ratio is summed by one thread per row in column order (the host's order), and the
// DDC histogram is an integer count.  The symmetrised graph S + S^T is never formed: a row walks its S row and its
// S^T row, and its degree |S_i u S^T_i| comes from a binary search of the reverse edge.
#include "dev.hpp"
#include "kcommon.hpp"
#include "rb.hpp"
#include "spgemm.hpp"
#include "cfsplit_dev.hpp"
#include "devmem.hpp"
#include <vector>
#include <cmath>
#include <algorithm>

namespace airhip {

namespace {

constexpr int F_POINT_D = -1, C_POINT_D = 1;      // host/air.hpp

__device__ __forceinline__ uint64_t splitmix64_d(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform01_d(uint64_t key, uint64_t i) {
  return (double)(splitmix64_d(key + i) >> 11) * (1.0 / 9007199254740992.0);
}

int grid1(int64_t n) {
  int64_t b = (n + BLOCK - 1) / BLOCK;
  if (b > 65535 * 16) b = 65535 * 16;
  return (int)(b < 1 ? 1 : b);
}

// strength threshold of every row and its number of strong off-diagonal entries (src/PETSc_Helper.F90:350-399)
__global__ __launch_bounds__(BLOCK) void k_strength_count(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                         const double* __restrict__ v, double theta, double* __restrict__ thr,
                                                         int64_t* __restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    double mx = -1.7976931348623157e308;
    for (int64_t k = rp[i]; k < rp[i + 1]; k++) if (ci[k] != i) mx = fmax(mx, fabs(v[k]));
    const double t = theta * mx;
    thr[i] = t;
    int64_t c = 0;
    for (int64_t k = rp[i]; k < rp[i + 1]; k++) if (ci[k] != i && fabs(v[k]) >= t) c++;
    cnt[i] = c;
  }
}

__global__ __launch_bounds__(BLOCK) void k_strength_fill(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                        const double* __restrict__ v, const double* __restrict__ thr,
                                                        const int64_t* __restrict__ srp, int32_t* __restrict__ sci,
                                                        unsigned long long* __restrict__ tcnt) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    int64_t o = srp[i];
    const double t = thr[i];
    for (int64_t k = rp[i]; k < rp[i + 1]; k++)
      if (ci[k] != i && fabs(v[k]) >= t) { sci[o++] = ci[k]; if (tcnt) atomicAdd(&tcnt[ci[k]], 1ull); }
  }
}

// The same two passes with one WAVE per row, for operators with long rows (the deep levels have hundreds of entries
// per row: a thread per row reads them uncoalesced).  Maximum and count are order-free; the strong entries keep their
// column order through ballot prefixes.
__global__ __launch_bounds__(BLOCK) void k_strength_count_w(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                           const double* __restrict__ v, double theta, double* __restrict__ thr,
                                                           int64_t* __restrict__ cnt) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < m; i += nw) {
    const int64_t a0 = rp[i], a1 = rp[i + 1];
    double mx = -1.7976931348623157e308;
    for (int64_t k = a0 + lane; k < a1; k += 64) if (ci[k] != i) mx = fmax(mx, fabs(v[k]));
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
    const double t = theta * mx;
    int c = 0;
    for (int64_t k = a0 + lane; k < a1; k += 64) c += (ci[k] != i && fabs(v[k]) >= t) ? 1 : 0;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if (lane == 0) { thr[i] = t; cnt[i] = c; }
  }
}

__global__ __launch_bounds__(BLOCK) void k_strength_fill_w(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                          const double* __restrict__ v, const double* __restrict__ thr,
                                                          const int64_t* __restrict__ srp, int32_t* __restrict__ sci,
                                                          unsigned long long* __restrict__ tcnt) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < m; i += nw) {
    const int64_t a0 = rp[i], a1 = rp[i + 1];
    int64_t o = srp[i];
    const double t = thr[i];
    for (int64_t kb = a0; kb < a1; kb += 64) {
      const int64_t k = kb + lane;
      const bool strong = k < a1 && ci[k] != i && fabs(v[k]) >= t;
      const unsigned long long b = __ballot(strong);
      if (strong) { const int32_t c = ci[k]; sci[o + __popcll(b & lt)] = c; if (tcnt) atomicAdd(&tcnt[c], 1ull); }
      o += __popcll(b);
    }
  }
}

// S^T adjacency (any order inside a row) and, per row, how many of its S entries have the reverse edge in S
__global__ __launch_bounds__(BLOCK) void k_transpose_fill(int64_t m, const int64_t* __restrict__ srp, const int32_t* __restrict__ sci,
                                                         unsigned long long* __restrict__ cursor, int32_t* __restrict__ tci,
                                                         int32_t* __restrict__ both) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    int32_t nb = 0;
    for (int64_t k = srp[i]; k < srp[i + 1]; k++) {
      const int32_t j = sci[k];
      tci[atomicAdd(&cursor[j], 1ull)] = (int32_t)i;
      // is (j, i) in S?  rows of S are in ascending column order
      int64_t lo = srp[j], hi = srp[j + 1];
      while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (sci[mid] < (int32_t)i) lo = mid + 1; else hi = mid; }
      if (lo < srp[j + 1] && sci[lo] == (int32_t)i) nb++;
    }
    both[i] = nb;
  }
}

// measure = rand + number of distinct strong neighbours; rows without any go straight to F
__global__ __launch_bounds__(BLOCK) void k_pmisr_init(int64_t m, uint64_t key, const int64_t* __restrict__ srp, const int64_t* __restrict__ trp,
                                                     const int32_t* __restrict__ both, double* __restrict__ measure,
                                                     int32_t* __restrict__ markers, uint8_t* __restrict__ assigned) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    int64_t deg = srp[i + 1] - srp[i];
    if (trp) deg += (trp[i + 1] - trp[i]) - both[i];
    const double ms = uniform01_d(key, (uint64_t)i) + (double)deg;
    measure[i] = ms;
    const bool lone = fabs(ms) < 1.0;
    markers[i] = lone ? F_POINT_D : 0;
    assigned[i] = lone ? 1 : 0;
  }
}

__device__ __forceinline__ bool loses_to(const double mi, int64_t i, int32_t j, const double* __restrict__ measure,
                                         const uint8_t* __restrict__ assigned) {
  if (assigned[j]) return false;
  const double mj = measure[j];
  return mi > mj || (mi == mj && i > (int64_t)j);
}

// one Luby round, part 1: a row joins the set if no unassigned neighbour has a smaller measure
__global__ __launch_bounds__(BLOCK) void k_luby_select(int64_t m, const int64_t* __restrict__ srp, const int32_t* __restrict__ sci,
                                                      const int64_t* __restrict__ trp, const int32_t* __restrict__ tci,
                                                      const double* __restrict__ measure, const uint8_t* __restrict__ assigned,
                                                      uint8_t* __restrict__ in_set) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    uint8_t in = 0;
    if (!assigned[i]) {
      in = 1;
      const double mi = measure[i];
      for (int64_t k = srp[i]; in && k < srp[i + 1]; k++) if (loses_to(mi, i, sci[k], measure, assigned)) in = 0;
      if (trp) for (int64_t k = trp[i]; in && k < trp[i + 1]; k++) if (loses_to(mi, i, tci[k], measure, assigned)) in = 0;
    }
    in_set[i] = in;
  }
}

// part 2: the new set members become F, they and their neighbours are assigned; counts the rows still undecided
__global__ __launch_bounds__(BLOCK) void k_luby_mark(int64_t m, const int64_t* __restrict__ srp, const int32_t* __restrict__ sci,
                                                    const int64_t* __restrict__ trp, const int32_t* __restrict__ tci,
                                                    const uint8_t* __restrict__ in_set, int32_t* __restrict__ markers,
                                                    uint8_t* __restrict__ assigned) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    if (!in_set[i]) continue;
    markers[i] = F_POINT_D; assigned[i] = 1;
    for (int64_t k = srp[i]; k < srp[i + 1]; k++) assigned[sci[k]] = 1;      // same-value stores
    if (trp) for (int64_t k = trp[i]; k < trp[i + 1]; k++) assigned[tci[k]] = 1;
  }
}

__global__ __launch_bounds__(BLOCK) void k_count_unassigned(int64_t m, const uint8_t* __restrict__ assigned, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[BLOCK];
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) c += assigned[i] ? 0 : 1;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int off = BLOCK / 2; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0 && red[0]) atomicAdd(out, red[0]);
}

__global__ __launch_bounds__(BLOCK) void k_finish_markers(int64_t m, int32_t* __restrict__ markers, unsigned long long* __restrict__ nf) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    if (markers[i] == 0) markers[i] = C_POINT_D;
    c += markers[i] == F_POINT_D ? 1 : 0;
  }
  if (c) atomicAdd(nf, c);
}

__device__ __forceinline__ int ddc_bin(double r) {
  constexpr int NB = 1000;
  const double b = floor(r * NB) + 1.0;
  if (!(b == b) || b > (double)NB) return NB;     // NaN / overflow -> top bin
  int bi = (int)b;
  if (bi < 0) bi = NB;
  return bi < NB ? bi : NB;
}

// diagonal-dominance ratio of the F rows inside A_ff, summed in column order, + the 1000-bin histogram
__global__ __launch_bounds__(BLOCK) void k_ddc_ratio(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                    const double* __restrict__ v, const int32_t* __restrict__ markers,
                                                    double* __restrict__ ratio, unsigned long long* __restrict__ bins) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  // every lane of a wave makes the same number of trips (the ballots below need the whole wave)
  const int64_t m_up = (m + WAVE - 1) / WAVE * WAVE;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m_up; i += stride) {
    double r = 0.0;
    int bin = -1;
    if (i < m && markers[i] == F_POINT_D) {
      double diag = 0.0, off = 0.0;
      for (int64_t k = rp[i]; k < rp[i + 1]; k++) {
        const int32_t c = ci[k];
        if (markers[c] != F_POINT_D) continue;
        if (c == i) diag = fabs(v[k]); else off += fabs(v[k]);
      }
      if (diag != 0.0) r = off / diag;
      bin = ddc_bin(r);
    }
    if (i < m) ratio[i] = r;
    // one atomic per distinct bin of the wave: on a constant-coefficient operator every F row lands in the same bin, and
    // 3e7 same-address atomics took 0.23 s per level (12 ns each) - the histogram is an integer count, order-free
    unsigned long long todo = __ballot(bin >= 0);
    while (todo) {
      const int leader = __ffsll((long long)todo) - 1;
      const int b = __shfl(bin, leader, WAVE);
      const unsigned long long same = __ballot(bin == b);
      if ((int)(threadIdx.x & (WAVE - 1)) == leader) atomicAdd(&bins[b], (unsigned long long)__popcll(same));
      todo &= ~same;
    }
  }
}

// one wave per row; the off-diagonal sum is taken in column order like the host's (the F-F entries of a 64-entry chunk
// are added one after the other by all lanes on broadcast values), so the ratios and with them the histogram are the same
__global__ __launch_bounds__(BLOCK) void k_ddc_ratio_w(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                      const double* __restrict__ v, const int32_t* __restrict__ markers,
                                                      double* __restrict__ ratio, unsigned long long* __restrict__ bins) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < m; i += nw) {
    double r = 0.0;
    if (markers[i] == F_POINT_D) {
      const int64_t a0 = rp[i], a1 = rp[i + 1];
      double diag = 0.0, off = 0.0;
      for (int64_t kb = a0; kb < a1; kb += 64) {
        const int64_t k = kb + lane;
        int32_t c = -1; double av = 0.0;
        if (k < a1) { c = ci[k]; av = fabs(v[k]); }
        const bool ff = c >= 0 && markers[c] == F_POINT_D;
        const unsigned long long bd = __ballot(ff && c == (int32_t)i);
        if (bd) diag = __shfl(av, __ffsll((long long)bd) - 1, 64);
        unsigned long long b = __ballot(ff && c != (int32_t)i);
        while (b) { const int t = __ffsll((long long)b) - 1; off += __shfl(av, t, 64); b &= b - 1; }
      }
      if (diag != 0.0) r = off / diag;
      if (lane == 0) atomicAdd(&bins[ddc_bin(r)], 1ull);
    }
    if (lane == 0) ratio[i] = r;
  }
}

__global__ __launch_bounds__(BLOCK) void k_ddc_flip(int64_t m, const double* __restrict__ ratio, double swap_val, int32_t* __restrict__ markers) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride)
    if (markers[i] == F_POINT_D && !(ratio[i] == 0.0 || ratio[i] < swap_val)) markers[i] = C_POINT_D;
}

struct Scratch {
  std::vector<void*> p;
  ~Scratch() { for (void* q : p) dev_free(q); }
  template <class T> T* get(int64_t n) { void* q = dev_alloc(sizeof(T) * (size_t)std::max<int64_t>(n, 1)); p.push_back(q); return (T*)q; }
};

}  // namespace

// markers_host[i] = F_POINT (-1) / C_POINT (1).  seed_key = rng_key(seed) of the host.
void dev_cf_splitting_pmisr_ddc(const DCsr& A, double strong_threshold, bool symmetrize, int max_luby_steps, uint64_t seed_key,
                                int ddc_its, double ddc_fraction, int32_t* markers_host, hipStream_t s) {
  const int64_t m = A.m;
  Scratch S;
  double* thr = S.get<double>(m);
  int64_t* cnt = S.get<int64_t>(m);
  int64_t* srp = S.get<int64_t>(m + 1);
  const bool long_rows = A.nnz >= 12 * m;            // one wave per row pays from a dozen entries per row on
  const int gw = (int)std::min<int64_t>((m + 3) / 4 > 0 ? (m + 3) / 4 : 1, 65535 * 16);
  if (long_rows) hipLaunchKernelGGL(k_strength_count_w, dim3(gw), dim3(BLOCK), 0, s, m, A.rp, A.ci, A.v, strong_threshold, thr, cnt);
  else hipLaunchKernelGGL(k_strength_count, dim3(grid1(m)), dim3(BLOCK), 0, s, m, A.rp, A.ci, A.v, strong_threshold, thr, cnt);
  const int64_t snnz = device_exclusive_scan(m, cnt, srp, s);
  HIP_CHECK(hipMemcpyAsync(srp + m, &snnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  int32_t* sci = S.get<int32_t>(snnz);
  int64_t* trp = nullptr; int32_t* tci = nullptr; int32_t* both = nullptr;
  unsigned long long* tcnt = nullptr;
  if (symmetrize) {
    tcnt = S.get<unsigned long long>(m);
    HIP_CHECK(hipMemsetAsync(tcnt, 0, sizeof(unsigned long long) * (size_t)m, s));
  }
  if (long_rows) hipLaunchKernelGGL(k_strength_fill_w, dim3(gw), dim3(BLOCK), 0, s, m, A.rp, A.ci, A.v, thr, srp, sci, tcnt);
  else hipLaunchKernelGGL(k_strength_fill, dim3(grid1(m)), dim3(BLOCK), 0, s, m, A.rp, A.ci, A.v, thr, srp, sci, tcnt);
  if (symmetrize) {
    trp = S.get<int64_t>(m + 1);
    static_assert(sizeof(unsigned long long) == sizeof(int64_t), "");
    const int64_t tn = device_exclusive_scan(m, (const int64_t*)tcnt, trp, s);
    HIP_CHECK(hipMemcpyAsync(trp + m, &tn, sizeof(int64_t), hipMemcpyHostToDevice, s));
    unsigned long long* cursor = S.get<unsigned long long>(m);
    HIP_CHECK(hipMemcpyAsync(cursor, trp, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToDevice, s));
    tci = S.get<int32_t>(snnz);
    both = S.get<int32_t>(m);
    hipLaunchKernelGGL(k_transpose_fill, dim3(grid1(m)), dim3(BLOCK), 0, s, m, srp, sci, cursor, tci, both);
  }
  double* measure = S.get<double>(m);
  int32_t* markers = S.get<int32_t>(m);
  uint8_t* assigned = S.get<uint8_t>(m);
  uint8_t* in_set = S.get<uint8_t>(m);
  unsigned long long* counter = S.get<unsigned long long>(1);
  hipLaunchKernelGGL(k_pmisr_init, dim3(grid1(m)), dim3(BLOCK), 0, s, m, seed_key, srp, trp, both, measure, markers, assigned);
  auto undecided = [&]() {
    unsigned long long h = 0;
    HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(h), s));
    hipLaunchKernelGGL(k_count_unassigned, dim3(std::min(grid1(m), 2048)), dim3(BLOCK), 0, s, m, assigned, counter);
    HIP_CHECK(hipMemcpyAsync(&h, counter, sizeof(h), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return (int64_t)h;
  };
  int loops = 0;
  while (undecided() != 0) {
    if (max_luby_steps > 0 && loops == max_luby_steps) break;
    hipLaunchKernelGGL(k_luby_select, dim3(grid1(m)), dim3(BLOCK), 0, s, m, srp, sci, trp, tci, measure, assigned, in_set);
    hipLaunchKernelGGL(k_luby_mark, dim3(grid1(m)), dim3(BLOCK), 0, s, m, srp, sci, trp, tci, in_set, markers, assigned);
    loops++;
  }
  HIP_CHECK(hipMemsetAsync(counter, 0, sizeof(unsigned long long), s));
  hipLaunchKernelGGL(k_finish_markers, dim3(grid1(m)), dim3(BLOCK), 0, s, m, markers, counter);
  // DDC (skipped entirely for strong_threshold = 0: src/CF_Splitting.F90:360-361)
  if (strong_threshold != 0.0 && ddc_fraction != 0.0) {
    double* ratio = measure;     // reuse
    unsigned long long* bins = S.get<unsigned long long>(1001);
    for (int it = 0; it < ddc_its; it++) {
      HIP_CHECK(hipMemsetAsync(bins, 0, sizeof(unsigned long long) * 1001, s));
      if (long_rows) hipLaunchKernelGGL(k_ddc_ratio_w, dim3(gw), dim3(BLOCK), 0, s, m, A.rp, A.ci, A.v, markers, ratio, bins);
      else hipLaunchKernelGGL(k_ddc_ratio, dim3(grid1(m)), dim3(BLOCK), 0, s, m, A.rp, A.ci, A.v, markers, ratio, bins);
      unsigned long long hb[1001];
      HIP_CHECK(hipMemcpyAsync(hb, bins, sizeof(hb), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      int64_t fine_size = 0;
      for (int b = 0; b <= 1000; b++) fine_size += (int64_t)hb[b];
      const int64_t search_size = ddc_fraction < 0 ? fine_size : (int64_t)((double)fine_size * ddc_fraction);
      if (search_size <= 0) continue;
      double swap_val;
      if (ddc_fraction < 0) swap_val = -ddc_fraction;
      else {
        int64_t bin_sum = 0; int bb = 1000;
        for (bb = 1000; bb >= 1; bb--) { bin_sum += (int64_t)hb[bb]; if (bin_sum >= search_size) break; }
        if (bb < 1) bb = 1;
        swap_val = (double)(bb - 1) / 1000.0;
      }
      hipLaunchKernelGGL(k_ddc_flip, dim3(grid1(m)), dim3(BLOCK), 0, s, m, ratio, swap_val, markers);
    }
  }
  HIP_CHECK(hipMemcpyAsync(markers_host, markers, sizeof(int32_t) * (size_t)m, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e) + " in the device CF splitting");
}

}  // namespace airhip
