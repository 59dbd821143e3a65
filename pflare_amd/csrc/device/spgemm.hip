// Device SpGEMM with fused relative drop / lump, used by the PCAIR setup for the two
// products that dominate the hierarchy construction:  Z = -A_cf * M  (r_drop) and
// A_c = R * (A P)  (a_drop, optional lumping)  [src/AIR_Operators_Setup.F90:784-795,
// 1008-1069; device twins exist in the reference as Kokkos SpGEMMs].
//
// One WAVE computes one output row.  The sparse accumulator is an open-addressing hash
// table keyed by the column (LDS for the common case, a global-memory table for rows that
// do not fit).  The rows of B are consumed one after the other (k ascending) with the 64
// lanes spread over the entries of B(k,:), whose columns are distinct - so no two lanes ever
// accumulate into the same key in the same step and every output entry is summed in exactly
// the order of a sequential Gustavson accumulator.  The finished row is sorted by column
// (bitonic, in the table), filtered with the semantics of remove_small_from_sparse
// (src/PETSc_Helper.F90:207-412; lumped sum taken in column order) and written in CSR order.
// Results are therefore bit-identical to the host implementation (host/spops.cpp).
#include "dev.hpp"
#include "kcommon.hpp"
#include "spgemm.hpp"
#include <stdexcept>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace airhip {

namespace {

constexpr int32_t EMPTY = 0x7fffffff;   // sorts last

struct CsrView { int64_t m, n; const int64_t* rp; const int32_t* ci; const double* v; };
static CsrView cview(const DCsr& A) { return CsrView{A.m, A.n, A.rp, A.ci, A.v}; }

struct DropParams { int enabled; double tol; int lump; int negate; int dbg; };

__device__ __forceinline__ uint32_t hash_col(int32_t c) { return (uint32_t)c * 2654435761u; }

// per-row upper bound of candidates: ub_i = sum_k nnz(B(k,:))
__global__ __launch_bounds__(BLOCK) void k_row_ub(CsrView A, CsrView B, int64_t* __restrict__ ub) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < A.m; i += nw) {
    int64_t s = 0;
    for (int64_t k = A.rp[i] + lane; k < A.rp[i + 1]; k += WAVE) { const int32_t j = A.ci[k]; s += B.rp[j + 1] - B.rp[j]; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    if (lane == 0) ub[i] = s;
  }
}

// Table accessors.  LDS tables are touched with plain ds ops (one wave owns the table and the
// wave executes its LDS instructions in order).  Global tables are touched only with relaxed
// agent-scope atomics so that every access is performed at L2 and no stale L1 line can be read.
template <bool GLOBAL> struct Tab {
  int32_t* keys; double* vals;
  __device__ __forceinline__ int32_t ldk(int h) const {
    if (GLOBAL) return __hip_atomic_load(&keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return keys[h];
  }
  __device__ __forceinline__ double ldv(int h) const {
    if (GLOBAL) return __hip_atomic_load(&vals[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return vals[h];
  }
  __device__ __forceinline__ void stk(int h, int32_t k) const {
    if (GLOBAL) __hip_atomic_store(&keys[h], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else keys[h] = k;
  }
  __device__ __forceinline__ void stv(int h, double v) const {
    if (GLOBAL) __hip_atomic_store(&vals[h], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else vals[h] = v;
  }
  __device__ __forceinline__ int32_t cas(int h, int32_t c) const {
    int32_t expected = EMPTY;
    __hip_atomic_compare_exchange_strong(&keys[h], &expected, c, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         GLOBAL ? __HIP_MEMORY_SCOPE_AGENT : __HIP_MEMORY_SCOPE_WAVEFRONT);
    return expected;   // EMPTY if we claimed the slot, else the resident key
  }
  __device__ __forceinline__ void sync() const {
    if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// One output row by one wave.  tab: table of HT entries owned by this wave (HT power of two).
// Count mode stores the number of kept entries; fill mode writes them at out_rp[i].
// Returns false if the table overflowed (row must be redone with a larger table).
template <bool FILL, bool GLOBAL>
__device__ __forceinline__ bool spgemm_row(const CsrView& A, const CsrView& B, int64_t i, const Tab<GLOBAL> tab, int HT,
                                           const DropParams& dp, int64_t* count_out, const int64_t* out_rp,
                                           int32_t* __restrict__ out_ci, double* __restrict__ out_v) {
#pragma clang fp contract(off)   // a*b + c must round twice, as the host code does
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t mask = (uint32_t)HT - 1;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int h = lane; h < HT; h += WAVE) { tab.stk(h, EMPTY); tab.stv(h, 0.0); }
  tab.sync();
  int unique = 0;
  bool overflow = false;
  const int64_t a0 = A.rp[i], a1 = A.rp[i + 1];
  // The candidate stream of the row is consumed in "steps": one step = up to 64 consecutive
  // entries of one row B(j,:) (distinct columns -> conflict-free lanes), steps ordered by k.
  // The loads of PIPE steps are issued back to back before their insertions so that a wave keeps
  // PIPE x 12 B x 64 in flight (the kernel is latency-bound otherwise: LDS limits the occupancy).
  constexpr int PIPE = 8;
  for (int64_t kb = a0; kb < a1 && !overflow && !(dp.dbg & 1); kb += WAVE) {
    // a chunk of A(i,:): lane t holds entry k = kb + t and the extent of B(j_t,:)
    const int64_t kk = kb + lane;
    const bool valid = kk < a1;
    const int32_t jl = valid ? A.ci[kk] : 0;
    const double al = valid ? A.v[kk] : 0.0;
    const int64_t b0 = valid ? B.rp[jl] : 0, b1 = valid ? B.rp[jl + 1] : 0;
    const int nch = (int)((b1 - b0 + WAVE - 1) >> 6);
    int incl = nch;                                   // inclusive scan of the step counts
    for (int off = 1; off < WAVE; off <<= 1) { const int t = __shfl_up(incl, off, WAVE); if (lane >= off) incl += t; }
    const int total = __shfl(incl, WAVE - 1, WAVE);
    for (int s0 = 0; s0 < total && !overflow; s0 += PIPE) {
      int32_t cc[PIPE]; double bv[PIPE], av[PIPE]; bool act[PIPE];
#pragma unroll
      for (int d = 0; d < PIPE; d++) {
        const int st = s0 + d;
        act[d] = false; cc[d] = 0; bv[d] = 0.0; av[d] = 0.0;
        if (st < total) {
          const int t = __ffsll((long long)__ballot(incl > st)) - 1;     // owner of step st
          const int64_t tb0 = __shfl(b0, t, WAVE), tb1 = __shfl(b1, t, WAVE);
          const int first = __shfl(incl, t, WAVE) - __shfl(nch, t, WAVE);
          const int64_t q = tb0 + (int64_t)(st - first) * WAVE + lane;
          av[d] = __shfl(al, t, WAVE);
          act[d] = q < tb1;
          if (act[d]) { cc[d] = B.ci[q]; bv[d] = B.v[q]; }
        }
      }
#pragma unroll
      for (int d = 0; d < PIPE; d++) {
        if (s0 + d >= total || overflow) continue;
        bool claimed = false, failed = false;
        if (act[d]) {
          const int32_t c = cc[d];
          const double x = av[d] * bv[d];
          uint32_t h = hash_col(c) & mask;
          int probes = 0;
          for (;;) {
            const int32_t old = tab.cas((int)h, c);
            if (old == EMPTY) claimed = true;
            if (old == EMPTY || old == c) {
              // the columns of B(j,:) are distinct: no other lane touches this slot in this step
              tab.stv((int)h, tab.ldv((int)h) + x);
              break;
            }
            h = (h + 1) & mask;
            if (++probes >= HT) { failed = true; break; }
          }
        }
        unique += __popcll(__ballot(claimed));
        if (__any(failed) || unique > (HT - HT / 8)) overflow = true;
        tab.sync();
      }
    }
  }
  if (overflow) return false;
  if (!FILL && !dp.dbg) {
    // count mode: the number of entries the filter keeps needs neither the compaction nor the sort - the row
    // maximum, "is the diagonal kept / created" and the number of large off-diagonal entries are order-free
    if (!dp.enabled) { if (lane == 0) *count_out = unique; return true; }
    double mx = 0.0;
    for (int h = lane; h < HT; h += WAVE) if (tab.ldk(h) != EMPTY) mx = fmax(mx, fabs(tab.ldv(h)));
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, WAVE));
    const double tolc = dp.tol * mx;
    bool have_diag = false, need_diag = false;
    int cnt = 0;
    for (int base = 0; base < HT; base += WAVE) {
      const int32_t k = tab.ldk(base + lane);
      const bool in = k != EMPTY;
      const double v = in ? tab.ldv(base + lane) : 0.0;
      const bool big = in && fabs(v) >= tolc;
      const bool isd = in && k == (int32_t)i;
      if (__any(isd && big)) have_diag = true;
      if (__any(in && !big && (dp.lump || isd))) need_diag = true;
      cnt += __popcll(__ballot(big && !isd));
    }
    if (lane == 0) *count_out = (int64_t)cnt + ((have_diag || need_diag) ? 1 : 0);
    return true;
  }
  // ---- compact the used slots to the front of the table (in place, chunk by chunk)
  int nu = 0;
  if (dp.dbg & 2) { if (lane == 0) *count_out = unique; return true; }
  for (int base = 0; base < HT; base += WAVE) {
    const int32_t k = tab.ldk(base + lane);
    const double v = tab.ldv(base + lane);
    const bool used = k != EMPTY;
    const unsigned long long bal = __ballot(used);
    const int pos = nu + __popcll(bal & lt);
    tab.sync();
    if (used) { tab.stk(pos, k); tab.stv(pos, dp.negate ? -v : v); }
    nu += __popcll(bal);
    tab.sync();
  }
  if (dp.dbg & 4) { if (lane == 0) *count_out = nu; return true; }
  int P2 = 64; while (P2 < nu) P2 <<= 1;
  for (int h = nu + lane; h < P2; h += WAVE) { tab.stk(h, EMPTY); tab.stv(h, 0.0); }
  tab.sync();
  // ---- bitonic sort of (key,val) by key
  for (int size = 2; size <= P2; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = lane; t < P2 / 2; t += WAVE) {
        const int a = 2 * t - (t & (stride - 1));
        const int b = a + stride;
        const bool asc = (a & size) == 0;
        const int32_t ka = tab.ldk(a), kb2 = tab.ldk(b);
        if ((ka > kb2) == asc) {
          tab.stk(a, kb2); tab.stk(b, ka);
          const double va = tab.ldv(a), vb = tab.ldv(b);
          tab.stv(a, vb); tab.stv(b, va);
        }
      }
      tab.sync();
    }
  if (dp.dbg & 8) { if (lane == 0) *count_out = nu; return true; }
  // ---- filter (remove_small_from_sparse, relative to the row max incl. the diagonal)
  int64_t kept = nu;
  double dsum = 0.0; bool write_diag = false; double tol = 0.0;
  if (dp.enabled) {
    double mx = 0.0;
    for (int h = lane; h < nu; h += WAVE) mx = fmax(mx, fabs(tab.ldv(h)));
    for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, WAVE));
    tol = dp.tol * mx;
    bool have_diag = false, need_diag = false;
    int cnt = 0;
    for (int base = 0; base < nu; base += WAVE) {
      const int h = base + lane;
      const bool in = h < nu;
      const int32_t k = in ? tab.ldk(h) : EMPTY;
      const double v = in ? tab.ldv(h) : 0.0;
      const bool big = in && fabs(v) >= tol;
      const bool isd = in && k == (int32_t)i;
      if (__any(isd && big)) have_diag = true;
      if (__any(in && !big && (dp.lump || isd))) need_diag = true;
      cnt += __popcll(__ballot(big && !isd));
    }
    write_diag = have_diag || need_diag;
    kept = cnt + (write_diag ? 1 : 0);
    if (write_diag && FILL) {
      // kept diagonal value first, then the removed entries in column order, added one after the other so that
      // the result is bit-identical to the sequential host code.  The table is read 64 slots at a time; only the
      // additions are sequential (every lane performs them on broadcast values, so the sum stays wave-uniform).
      double sacc = 0.0;
      for (int base = 0; base < nu; base += WAVE) {
        const int h = base + lane;
        const bool in = h < nu;
        const double v = in ? tab.ldv(h) : 0.0;
        const bool hit = in && tab.ldk(h) == (int32_t)i && fabs(v) >= tol;
        const unsigned long long b = __ballot(hit);
        if (b) sacc = __shfl(v, __ffsll((long long)b) - 1, WAVE);       // the kept diagonal (unique column)
      }
      for (int base = 0; base < nu; base += WAVE) {
        const int h = base + lane;
        const bool in = h < nu;
        const double v = in ? tab.ldv(h) : 0.0;
        const bool add = in && !(fabs(v) >= tol) && (dp.lump || tab.ldk(h) == (int32_t)i);
        unsigned long long b = __ballot(add);
        while (b) {
          const int t = __ffsll((long long)b) - 1;
          sacc += __shfl(v, t, WAVE);
          b &= b - 1;
        }
      }
      dsum = sacc;
    }
  }
  if (!FILL) { if (lane == 0) *count_out = kept; return true; }
  // ---- write the row in column order
  int64_t o = out_rp[i];
  if (!dp.enabled) {
    for (int h = lane; h < nu; h += WAVE) { out_ci[o + h] = tab.ldk(h); out_v[o + h] = tab.ldv(h); }
    return true;
  }
  bool diag_written = !write_diag;
  for (int base = 0; base < nu; base += WAVE) {
    const int h = base + lane;
    const bool in = h < nu;
    const int32_t k = in ? tab.ldk(h) : EMPTY;
    const double v = in ? tab.ldv(h) : 0.0;
    const bool isd = in && k == (int32_t)i;
    const bool keep = in && !isd && fabs(v) >= tol;
    const unsigned long long kbal = __ballot(keep);
    if (!diag_written) {
      // the diagonal goes in front of the first entry of this chunk with column >= i
      const unsigned long long ge = __ballot(in && k >= (int32_t)i);
      if (ge) {
        const int first = __ffsll((long long)ge) - 1;
        const unsigned long long lo = (1ull << first) - 1ull;
        const unsigned long long before = kbal & lo, after = kbal & ~lo;
        if (keep && lane < first) { const int p = __popcll(before & lt); out_ci[o + p] = k; out_v[o + p] = v; }
        o += __popcll(before);
        if (lane == 0) { out_ci[o] = (int32_t)i; out_v[o] = dsum; }
        o += 1;
        if (keep && lane >= first) { const int p = __popcll(after & lt); out_ci[o + p] = k; out_v[o + p] = v; }
        o += __popcll(after);
        diag_written = true;
        continue;
      }
    }
    if (keep) { const int p = __popcll(kbal & lt); out_ci[o + p] = k; out_v[o + p] = v; }
    o += __popcll(kbal);
  }
  if (!diag_written && lane == 0) { out_ci[o] = (int32_t)i; out_v[o] = dsum; }
  return true;
}

template <int HT, bool FILL>
__global__ __launch_bounds__(HT <= 1024 ? BLOCK : WAVE) void k_spgemm_lds(CsrView A, CsrView B, const int32_t* __restrict__ rows,
                                                                            int64_t nrows, DropParams dp, int64_t* __restrict__ cnt,
                                                                            const int64_t* __restrict__ out_rp, int32_t* out_ci,
                                                                            double* out_v, int32_t* __restrict__ failed) {
  constexpr int WPB = HT <= 1024 ? WAVES_PER_BLOCK : 1;
  __shared__ int32_t keys_all[WPB][HT];
  __shared__ double vals_all[WPB][HT];
  const int wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * WPB + wid;
  const int64_t nw = (int64_t)gridDim.x * WPB;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const Tab<false> tab{keys_all[wid], vals_all[wid]};
    const bool ok = spgemm_row<FILL, false>(A, B, i, tab, HT, dp, cnt + i, out_rp, out_ci, out_v);
    if (!ok && (threadIdx.x & 63) == 0) failed[w] = 1;
    __builtin_amdgcn_wave_barrier();
  }
}

// rows that overflowed the LDS table: one wave per row with a private global-memory table
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void k_spgemm_glb(CsrView A, CsrView B, const int32_t* __restrict__ rows, int64_t nrows,
                                                     const int64_t* __restrict__ tab_off, const int32_t* __restrict__ tab_size,
                                                     int32_t* keys_pool, double* vals_pool, DropParams dp, int64_t* __restrict__ cnt,
                                                     const int64_t* __restrict__ out_rp, int32_t* out_ci, double* out_v,
                                                     int32_t* __restrict__ failed) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const Tab<true> tab{keys_pool + tab_off[w], vals_pool + tab_off[w]};
    const bool ok = spgemm_row<FILL, true>(A, B, i, tab, tab_size[w], dp, cnt + i, out_rp, out_ci, out_v);
    if (!ok && (threadIdx.x & 63) == 0) failed[w] = 1;
  }
}

template <class T> T* dmalloc(int64_t n) { T* p = nullptr; HIP_CHECK(hipMalloc(&p, sizeof(T) * (size_t)(n > 0 ? n : 1))); return p; }

}  // namespace

void dcsr_free(DCsr& A) {
  if (A.rp) (void)hipFree(A.rp);
  if (A.ci) (void)hipFree(A.ci);
  if (A.v) (void)hipFree(A.v);
  A = DCsr();
}

DCsr dcsr_upload(int64_t m, int64_t n, const int64_t* rp, const int32_t* ci, const double* v) {
  DCsr d; d.m = m; d.n = n; d.nnz = rp[m];
  d.rp = dmalloc<int64_t>(m + 1); d.ci = dmalloc<int32_t>(d.nnz); d.v = dmalloc<double>(d.nnz);
  HIP_CHECK(hipMemcpy(d.rp, rp, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyHostToDevice));
  if (d.nnz) {
    HIP_CHECK(hipMemcpy(d.ci, ci, sizeof(int32_t) * (size_t)d.nnz, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d.v, v, sizeof(double) * (size_t)d.nnz, hipMemcpyHostToDevice));
  }
  return d;
}

void dcsr_download(const DCsr& d, int64_t* rp, int32_t* ci, double* v) {
  HIP_CHECK(hipMemcpy(rp, d.rp, sizeof(int64_t) * (size_t)(d.m + 1), hipMemcpyDeviceToHost));
  if (d.nnz) {
    HIP_CHECK(hipMemcpy(ci, d.ci, sizeof(int32_t) * (size_t)d.nnz, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(v, d.v, sizeof(double) * (size_t)d.nnz, hipMemcpyDeviceToHost));
  }
}

// host-side exclusive scan helper on a downloaded count array (counts are m int64: cheap next
// to the product itself)
static int64_t scan_counts(std::vector<int64_t>& c) {
  int64_t run = 0;
  for (auto& x : c) { int64_t t = x; x = run; run += t; }
  c.push_back(run);
  return run;
}

DCsr dev_spgemm(const DCsr& A, const DCsr& B, bool drop_enabled, double tol, bool lump, bool negate, hipStream_t s,
                SpgemmStats* stats) {
  if (A.n != B.m) throw std::runtime_error("dev_spgemm: shape mismatch");
  static const bool trace = std::getenv("AIRHIP_SPGEMM_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  double t_ub = 0, t_bin = 0, t_p1 = 0, t_scan = 0, t_p2 = 0;
  const int64_t m = A.m;
  DCsr C; C.m = m; C.n = B.n;
  C.rp = dmalloc<int64_t>(m + 1);
  if (m == 0) { int64_t z = 0; HIP_CHECK(hipMemcpy(C.rp, &z, sizeof(int64_t), hipMemcpyHostToDevice)); return C; }
  const CsrView av = cview(A), bv = cview(B);
  DropParams dp{drop_enabled ? 1 : 0, tol, lump ? 1 : 0, negate ? 1 : 0, 0};
  // ---- candidate upper bounds and binning (host side: m ints)
  int64_t* d_ub = dmalloc<int64_t>(m);
  hipLaunchKernelGGL(k_row_ub, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, av, bv, d_ub);
  std::vector<int64_t> ub(m);
  HIP_CHECK(hipMemcpyAsync(ub.data(), d_ub, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  t_ub = now() - t_begin;
  std::vector<int32_t> small, big;
  small.reserve(m);
  for (int64_t i = 0; i < m; i++) (ub[i] <= 768 ? small : big).push_back((int32_t)i);
  int32_t* d_small = dmalloc<int32_t>((int64_t)small.size());
  int32_t* d_big = dmalloc<int32_t>((int64_t)big.size());
  if (!small.empty()) HIP_CHECK(hipMemcpyAsync(d_small, small.data(), sizeof(int32_t) * small.size(), hipMemcpyHostToDevice, s));
  if (!big.empty()) HIP_CHECK(hipMemcpyAsync(d_big, big.data(), sizeof(int32_t) * big.size(), hipMemcpyHostToDevice, s));
  int32_t* d_fail_s = dmalloc<int32_t>((int64_t)small.size());
  int32_t* d_fail_b = dmalloc<int32_t>((int64_t)big.size());
  HIP_CHECK(hipMemsetAsync(d_fail_s, 0, sizeof(int32_t) * std::max<size_t>(small.size(), 1), s));
  HIP_CHECK(hipMemsetAsync(d_fail_b, 0, sizeof(int32_t) * std::max<size_t>(big.size(), 1), s));
  int64_t* d_cnt = dmalloc<int64_t>(m + 1);

  auto grid_small = [&](size_t n) { int64_t b = ((int64_t)n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(b, 256 * 12))); };
  auto grid_big = [&](size_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)n, 256 * 3))); };

  if (trace && big.size() > 100000) {
    // phase breakdown of the big-bin kernel (timing only; results are overwritten below)
    const char* nm[5] = {"init only", "init+insert", "+compact", "+sort", "all"};
    const int flag[5] = {1 | 2, 2, 4, 8, 0};
    for (int v = 0; v < 5; v++) {
      HIP_CHECK(hipStreamSynchronize(s));
      const double t0 = now();
      DropParams d2 = dp; d2.dbg = flag[v];
      hipLaunchKernelGGL((k_spgemm_lds<4096, false>), grid_big(big.size()), dim3(WAVE), 0, s, av, bv, d_big, (int64_t)big.size(), d2, d_cnt,
                         (const int64_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, d_fail_b);
      HIP_CHECK(hipStreamSynchronize(s));
      std::printf("      big-bin %-12s %.3f s\n", nm[v], now() - t0);
    }
    HIP_CHECK(hipMemsetAsync(d_fail_b, 0, sizeof(int32_t) * big.size(), s));
  }
  t_bin = now() - t_begin - t_ub;
  // ---- pass 1: counts
  if (!small.empty())
    hipLaunchKernelGGL((k_spgemm_lds<1024, false>), grid_small(small.size()), dim3(BLOCK), 0, s, av, bv, d_small, (int64_t)small.size(),
                       dp, d_cnt, (const int64_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, d_fail_s);
  if (!big.empty())
    hipLaunchKernelGGL((k_spgemm_lds<4096, false>), grid_big(big.size()), dim3(WAVE), 0, s, av, bv, d_big, (int64_t)big.size(), dp, d_cnt,
                       (const int64_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, d_fail_b);
  // rows that overflowed their LDS table -> global-memory tables
  std::vector<int32_t> fs(small.size()), fb(big.size());
  if (!small.empty()) HIP_CHECK(hipMemcpyAsync(fs.data(), d_fail_s, sizeof(int32_t) * fs.size(), hipMemcpyDeviceToHost, s));
  if (!big.empty()) HIP_CHECK(hipMemcpyAsync(fb.data(), d_fail_b, sizeof(int32_t) * fb.size(), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  std::vector<int32_t> huge;
  for (size_t w = 0; w < fs.size(); w++) if (fs[w]) huge.push_back(small[w]);
  for (size_t w = 0; w < fb.size(); w++) if (fb[w]) huge.push_back(big[w]);
  int32_t* d_huge = nullptr; int64_t* d_toff = nullptr; int32_t* d_tsize = nullptr; int32_t* d_kpool = nullptr; double* d_vpool = nullptr;
  int32_t* d_fail_h = nullptr;
  if (!huge.empty()) {
    std::vector<int64_t> toff(huge.size()); std::vector<int32_t> tsize(huge.size());
    int64_t run = 0;
    for (size_t w = 0; w < huge.size(); w++) {
      int64_t want = 2 * std::min<int64_t>(ub[huge[w]], B.n) + 64;
      int64_t cap = 8192; while (cap < want) cap <<= 1;
      if (cap > 0x40000000ll) throw std::runtime_error("dev_spgemm: row too long");
      toff[w] = run; tsize[w] = (int32_t)cap; run += cap;
    }
    d_huge = dmalloc<int32_t>((int64_t)huge.size()); d_toff = dmalloc<int64_t>((int64_t)huge.size());
    d_tsize = dmalloc<int32_t>((int64_t)huge.size()); d_kpool = dmalloc<int32_t>(run); d_vpool = dmalloc<double>(run);
    d_fail_h = dmalloc<int32_t>((int64_t)huge.size());
    HIP_CHECK(hipMemcpy(d_huge, huge.data(), sizeof(int32_t) * huge.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_toff, toff.data(), sizeof(int64_t) * huge.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_tsize, tsize.data(), sizeof(int32_t) * huge.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemsetAsync(d_fail_h, 0, sizeof(int32_t) * huge.size(), s));
    const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)huge.size() + 3) / 4, 2048);
    hipLaunchKernelGGL((k_spgemm_glb<false>), dim3(gb), dim3(BLOCK), 0, s, av, bv, d_huge, (int64_t)huge.size(), d_toff, d_tsize,
                       d_kpool, d_vpool, dp, d_cnt, (const int64_t*)nullptr, (int32_t*)nullptr, (double*)nullptr, d_fail_h);
    std::vector<int32_t> fh(huge.size());
    HIP_CHECK(hipMemcpyAsync(fh.data(), d_fail_h, sizeof(int32_t) * fh.size(), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int32_t f : fh) if (f) throw std::runtime_error("dev_spgemm: global hash table overflow");
  }
  t_p1 = now() - t_begin - t_ub - t_bin;
  // ---- row pointers
  std::vector<int64_t> cnt(m);
  HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  C.nnz = scan_counts(cnt);
  HIP_CHECK(hipMemcpyAsync(C.rp, cnt.data(), sizeof(int64_t) * (size_t)(m + 1), hipMemcpyHostToDevice, s));
  C.ci = dmalloc<int32_t>(C.nnz); C.v = dmalloc<double>(C.nnz);
  t_scan = now() - t_begin - t_ub - t_bin - t_p1;
  // ---- pass 2: fill (rows that overflowed in pass 1 are skipped here and done by the global kernel)
  if (!huge.empty()) {
    // remove the overflow rows from the LDS lists
    std::vector<int32_t> s2, b2;
    for (size_t w = 0; w < fs.size(); w++) if (!fs[w]) s2.push_back(small[w]);
    for (size_t w = 0; w < fb.size(); w++) if (!fb[w]) b2.push_back(big[w]);
    small.swap(s2); big.swap(b2);
    if (!small.empty()) HIP_CHECK(hipMemcpyAsync(d_small, small.data(), sizeof(int32_t) * small.size(), hipMemcpyHostToDevice, s));
    if (!big.empty()) HIP_CHECK(hipMemcpyAsync(d_big, big.data(), sizeof(int32_t) * big.size(), hipMemcpyHostToDevice, s));
  }
  if (!small.empty())
    hipLaunchKernelGGL((k_spgemm_lds<1024, true>), grid_small(small.size()), dim3(BLOCK), 0, s, av, bv, d_small, (int64_t)small.size(),
                       dp, d_cnt, C.rp, C.ci, C.v, d_fail_s);
  if (!big.empty())
    hipLaunchKernelGGL((k_spgemm_lds<4096, true>), grid_big(big.size()), dim3(WAVE), 0, s, av, bv, d_big, (int64_t)big.size(), dp, d_cnt,
                       C.rp, C.ci, C.v, d_fail_b);
  if (!huge.empty()) {
    const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)huge.size() + 3) / 4, 2048);
    hipLaunchKernelGGL((k_spgemm_glb<true>), dim3(gb), dim3(BLOCK), 0, s, av, bv, d_huge, (int64_t)huge.size(), d_toff, d_tsize, d_kpool,
                       d_vpool, dp, d_cnt, C.rp, C.ci, C.v, d_fail_h);
  }
  HIP_CHECK(hipStreamSynchronize(s));
  t_p2 = now() - t_begin - t_ub - t_bin - t_p1 - t_scan;
  if (trace) {
    int64_t ubt = 0; for (int64_t i = 0; i < m; i++) ubt += ub[i];
    std::printf("   [spgemm] m=%lld nnzA=%lld nnzB=%lld cand=%lld nnzC=%lld bins=%zu/%zu/%zu  ub %.3f bin %.3f pass1 %.3f scan %.3f pass2 %.3f s\n",
                (long long)m, (long long)A.nnz, (long long)B.nnz, (long long)ubt, (long long)C.nnz, small.size(), big.size(), huge.size(),
                t_ub, t_bin, t_p1, t_scan, t_p2);
    std::fflush(stdout);
  }
  if (stats) { stats->rows_small = (int64_t)small.size(); stats->rows_big = (int64_t)big.size(); stats->rows_global = (int64_t)huge.size(); }
  (void)hipFree(d_ub); (void)hipFree(d_small); (void)hipFree(d_big); (void)hipFree(d_fail_s); (void)hipFree(d_fail_b); (void)hipFree(d_cnt);
  if (d_huge) { (void)hipFree(d_huge); (void)hipFree(d_toff); (void)hipFree(d_tsize); (void)hipFree(d_kpool); (void)hipFree(d_vpool); (void)hipFree(d_fail_h); }
  return C;
}

}  // namespace airhip
