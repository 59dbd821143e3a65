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
                                           const DropParams& dp, int64_t* count_out, int64_t out_base,
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
  if (lane == 0) *count_out = kept;
  if (!FILL) return true;
  // ---- write the row in column order
  int64_t o = out_base;
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
                                                                            const int64_t* __restrict__ out_off, int32_t* out_ci,
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
    const bool ok = spgemm_row<FILL, false>(A, B, i, tab, HT, dp, cnt + i, FILL ? out_off[w] : 0, out_ci, out_v);
    if (!ok && (threadIdx.x & 63) == 0) failed[w] = 1;
    __builtin_amdgcn_wave_barrier();
  }
}

// rows that overflowed the LDS table: one wave per row with a private global-memory table
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void k_spgemm_glb(CsrView A, CsrView B, const int32_t* __restrict__ rows, int64_t nrows,
                                                     const int64_t* __restrict__ tab_off, const int32_t* __restrict__ tab_size,
                                                     int32_t* keys_pool, double* vals_pool, DropParams dp, int64_t* __restrict__ cnt,
                                                     const int64_t* __restrict__ out_off, int32_t* out_ci, double* out_v,
                                                     int32_t* __restrict__ failed) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const Tab<true> tab{keys_pool + tab_off[w], vals_pool + tab_off[w]};
    const bool ok = spgemm_row<FILL, true>(A, B, i, tab, tab_size[w], dp, cnt + i, FILL ? out_off[w] : 0, out_ci, out_v);
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

// rows of a chunk from the scratch pool (at src_off[w], cnt entries) to their final place (dst_off[w]); one wave per row
__global__ __launch_bounds__(BLOCK) void k_compact_rows(int64_t nrows, const int64_t* __restrict__ src_off, const int64_t* __restrict__ dst_off,
                                                       const int64_t* __restrict__ n_of, const int32_t* __restrict__ tci,
                                                       const double* __restrict__ tv, int32_t* __restrict__ oci, double* __restrict__ ov) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t a = src_off[w], o = dst_off[w], n = n_of[w];
    for (int64_t k = lane; k < n; k += WAVE) { oci[o + k] = tci[a + k]; ov[o + k] = tv[a + k]; }
  }
}

// One numeric pass (round 2; round 1 ran every product twice, once to count and once to fill): the rows are processed
// in chunks of consecutive rows; a chunk's finished (sorted, filtered) rows go to a scratch pool at offsets given by
// their upper bounds, the counts come back, and the chunk is compacted into its final, contiguous piece of C.  The
// pieces are consecutive row ranges, so C is their concatenation.
DCsr dev_spgemm(const DCsr& A, const DCsr& B, bool drop_enabled, double tol, bool lump, bool negate, hipStream_t s,
                SpgemmStats* stats) {
  if (A.n != B.m) throw std::runtime_error("dev_spgemm: shape mismatch");
  static const bool trace = std::getenv("AIRHIP_SPGEMM_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  const int64_t m = A.m;
  DCsr C; C.m = m; C.n = B.n;
  C.rp = dmalloc<int64_t>(m + 1);
  if (m == 0) { int64_t z = 0; HIP_CHECK(hipMemcpy(C.rp, &z, sizeof(int64_t), hipMemcpyHostToDevice)); return C; }
  const CsrView av = cview(A), bv = cview(B);
  const DropParams dp{drop_enabled ? 1 : 0, tol, lump ? 1 : 0, negate ? 1 : 0, 0};
  // ---- candidate upper bounds (host side: m ints)
  int64_t* d_ub = dmalloc<int64_t>(m);
  hipLaunchKernelGGL(k_row_ub, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, av, bv, d_ub);
  std::vector<int64_t> ub(m);
  HIP_CHECK(hipMemcpyAsync(ub.data(), d_ub, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d_ub);
  const double t_ub = now() - t_begin;
  // scratch capacity of a row: it cannot keep more entries than it has candidates, than its hash table holds, or than
  // B has columns - plus one for a diagonal created by lumping
  constexpr int64_t SMALL_UB = 768, SMALL_CAP = 1024 - 1024 / 8, BIG_CAP = 4096 - 4096 / 8;
  auto cap_lds = [&](int64_t u) { return std::min<int64_t>(u, u <= SMALL_UB ? SMALL_CAP : BIG_CAP) + 1; };
  const int64_t POOL = std::max<int64_t>(4ll << 20, std::min<int64_t>(512ll << 20, (A.nnz + B.nnz) * 2));   // entries (12 B each)
  int32_t* pool_ci = dmalloc<int32_t>(POOL); double* pool_v = dmalloc<double>(POOL);
  int64_t* d_cnt = dmalloc<int64_t>(m + 1);
  struct Piece { int64_t r0, r1, nnz; int32_t* ci; double* v; };
  std::vector<Piece> pieces;
  std::vector<int64_t> rp_host(m + 1);
  rp_host[0] = 0;
  int64_t n_small = 0, n_big = 0, n_huge = 0;
  double t_kern = 0, t_compact = 0;
  auto grid_small = [&](size_t n) { int64_t b = ((int64_t)n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(b, 256 * 12))); };
  auto grid_big = [&](size_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)n, 256 * 3))); };

  std::vector<int32_t> small, big, fs, fb;
  std::vector<int64_t> off_s, off_b, src_all, dst_all, n_all, cnt_h;
  for (int64_t r0 = 0; r0 < m;) {
    // ---- a chunk of consecutive rows whose scratch needs fit the pool
    small.clear(); big.clear(); off_s.clear(); off_b.clear();
    int64_t used = 0, r1 = r0;
    while (r1 < m) {
      const int64_t c = cap_lds(ub[r1]);
      if (used + c > POOL && r1 > r0) break;
      if (ub[r1] <= SMALL_UB) { small.push_back((int32_t)r1); off_s.push_back(used); } else { big.push_back((int32_t)r1); off_b.push_back(used); }
      used += c; r1++;
    }
    if (used > POOL) throw std::runtime_error("dev_spgemm: scratch pool too small for one row");
    const int64_t nrows = r1 - r0;
    const double tk0 = now();
    int32_t* d_small = dmalloc<int32_t>((int64_t)small.size()); int32_t* d_big = dmalloc<int32_t>((int64_t)big.size());
    int64_t* d_off_s = dmalloc<int64_t>((int64_t)small.size()); int64_t* d_off_b = dmalloc<int64_t>((int64_t)big.size());
    int32_t* d_fail_s = dmalloc<int32_t>((int64_t)small.size()); int32_t* d_fail_b = dmalloc<int32_t>((int64_t)big.size());
    if (!small.empty()) {
      HIP_CHECK(hipMemcpyAsync(d_small, small.data(), sizeof(int32_t) * small.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(d_off_s, off_s.data(), sizeof(int64_t) * small.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemsetAsync(d_fail_s, 0, sizeof(int32_t) * small.size(), s));
      hipLaunchKernelGGL((k_spgemm_lds<1024, true>), grid_small(small.size()), dim3(BLOCK), 0, s, av, bv, d_small, (int64_t)small.size(), dp,
                         d_cnt, d_off_s, pool_ci, pool_v, d_fail_s);
    }
    if (!big.empty()) {
      HIP_CHECK(hipMemcpyAsync(d_big, big.data(), sizeof(int32_t) * big.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(d_off_b, off_b.data(), sizeof(int64_t) * big.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemsetAsync(d_fail_b, 0, sizeof(int32_t) * big.size(), s));
      hipLaunchKernelGGL((k_spgemm_lds<4096, true>), grid_big(big.size()), dim3(WAVE), 0, s, av, bv, d_big, (int64_t)big.size(), dp, d_cnt,
                         d_off_b, pool_ci, pool_v, d_fail_b);
    }
    fs.assign(small.size(), 0); fb.assign(big.size(), 0);
    if (!small.empty()) HIP_CHECK(hipMemcpyAsync(fs.data(), d_fail_s, sizeof(int32_t) * fs.size(), hipMemcpyDeviceToHost, s));
    if (!big.empty()) HIP_CHECK(hipMemcpyAsync(fb.data(), d_fail_b, sizeof(int32_t) * fb.size(), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    // ---- rows that overflowed their LDS table: global-memory tables, their own scratch
    std::vector<int32_t> huge;
    for (size_t w = 0; w < fs.size(); w++) if (fs[w]) huge.push_back(small[w]);
    for (size_t w = 0; w < fb.size(); w++) if (fb[w]) huge.push_back(big[w]);
    std::sort(huge.begin(), huge.end());
    int32_t* hpool_ci = nullptr; double* hpool_v = nullptr;
    std::vector<int64_t> hoff(huge.size());
    if (!huge.empty()) {
      std::vector<int64_t> toff(huge.size()); std::vector<int32_t> tsize(huge.size());
      int64_t run = 0, hrun = 0;
      for (size_t w = 0; w < huge.size(); w++) {
        const int64_t want = 2 * std::min<int64_t>(ub[huge[w]], B.n) + 64;
        int64_t cap = 8192; while (cap < want) cap <<= 1;
        if (cap > 0x40000000ll) throw std::runtime_error("dev_spgemm: row too long");
        toff[w] = run; tsize[w] = (int32_t)cap; run += cap;
        hoff[w] = hrun; hrun += std::min<int64_t>(ub[huge[w]], B.n) + 1;
      }
      int32_t* d_huge = dmalloc<int32_t>((int64_t)huge.size()); int64_t* d_toff = dmalloc<int64_t>((int64_t)huge.size());
      int32_t* d_tsize = dmalloc<int32_t>((int64_t)huge.size()); int32_t* d_kpool = dmalloc<int32_t>(run); double* d_vpool = dmalloc<double>(run);
      int32_t* d_fail_h = dmalloc<int32_t>((int64_t)huge.size()); int64_t* d_hoff = dmalloc<int64_t>((int64_t)huge.size());
      hpool_ci = dmalloc<int32_t>(hrun); hpool_v = dmalloc<double>(hrun);
      HIP_CHECK(hipMemcpy(d_huge, huge.data(), sizeof(int32_t) * huge.size(), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_toff, toff.data(), sizeof(int64_t) * huge.size(), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_tsize, tsize.data(), sizeof(int32_t) * huge.size(), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_hoff, hoff.data(), sizeof(int64_t) * huge.size(), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemsetAsync(d_fail_h, 0, sizeof(int32_t) * huge.size(), s));
      const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)huge.size() + 3) / 4, 2048);
      hipLaunchKernelGGL((k_spgemm_glb<true>), dim3(gb), dim3(BLOCK), 0, s, av, bv, d_huge, (int64_t)huge.size(), d_toff, d_tsize, d_kpool,
                         d_vpool, dp, d_cnt, d_hoff, hpool_ci, hpool_v, d_fail_h);
      std::vector<int32_t> fh(huge.size());
      HIP_CHECK(hipMemcpyAsync(fh.data(), d_fail_h, sizeof(int32_t) * fh.size(), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      (void)hipFree(d_huge); (void)hipFree(d_toff); (void)hipFree(d_tsize); (void)hipFree(d_kpool); (void)hipFree(d_vpool);
      (void)hipFree(d_fail_h); (void)hipFree(d_hoff);
      for (int32_t f : fh) if (f) throw std::runtime_error("dev_spgemm: global hash table overflow");
      n_huge += (int64_t)huge.size();
    }
    n_small += (int64_t)small.size(); n_big += (int64_t)big.size();
    t_kern += now() - tk0;
    // ---- counts of the chunk, its row pointers, compaction into its piece of C
    const double tc0 = now();
    cnt_h.resize(nrows);
    HIP_CHECK(hipMemcpyAsync(cnt_h.data(), d_cnt + r0, sizeof(int64_t) * (size_t)nrows, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    src_all.assign(nrows, 0); dst_all.assign(nrows, 0); n_all.assign(nrows, 0);
    int64_t run = 0;
    for (int64_t r = r0; r < r1; r++) { dst_all[r - r0] = run; n_all[r - r0] = cnt_h[r - r0]; run += cnt_h[r - r0]; rp_host[r + 1] = rp_host[r] + cnt_h[r - r0]; }
    for (size_t w = 0; w < small.size(); w++) src_all[small[w] - r0] = off_s[w];
    for (size_t w = 0; w < big.size(); w++) src_all[big[w] - r0] = off_b[w];
    Piece pc{r0, r1, run, dmalloc<int32_t>(run), dmalloc<double>(run)};
    int64_t* d_src = dmalloc<int64_t>(nrows); int64_t* d_dst = dmalloc<int64_t>(nrows); int64_t* d_n = dmalloc<int64_t>(nrows);
    if (!huge.empty()) {
      // the overflow rows come from their own pool: compact them first (their src offsets refer to hpool), then mask them
      std::vector<int64_t> hs(huge.size()), hd(huge.size()), hn(huge.size());
      for (size_t w = 0; w < huge.size(); w++) { hs[w] = hoff[w]; hd[w] = dst_all[huge[w] - r0]; hn[w] = n_all[huge[w] - r0]; n_all[huge[w] - r0] = 0; }
      int64_t* d_hs = dmalloc<int64_t>((int64_t)huge.size()); int64_t* d_hd = dmalloc<int64_t>((int64_t)huge.size()); int64_t* d_hn = dmalloc<int64_t>((int64_t)huge.size());
      HIP_CHECK(hipMemcpy(d_hs, hs.data(), sizeof(int64_t) * hs.size(), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_hd, hd.data(), sizeof(int64_t) * hd.size(), hipMemcpyHostToDevice));
      HIP_CHECK(hipMemcpy(d_hn, hn.data(), sizeof(int64_t) * hn.size(), hipMemcpyHostToDevice));
      const unsigned g = (unsigned)std::min<int64_t>(((int64_t)huge.size() + 3) / 4, 2048);
      hipLaunchKernelGGL(k_compact_rows, dim3(g), dim3(BLOCK), 0, s, (int64_t)huge.size(), d_hs, d_hd, d_hn, hpool_ci, hpool_v, pc.ci, pc.v);
      HIP_CHECK(hipStreamSynchronize(s));
      (void)hipFree(d_hs); (void)hipFree(d_hd); (void)hipFree(d_hn); (void)hipFree(hpool_ci); (void)hipFree(hpool_v);
    }
    HIP_CHECK(hipMemcpyAsync(d_src, src_all.data(), sizeof(int64_t) * (size_t)nrows, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_dst, dst_all.data(), sizeof(int64_t) * (size_t)nrows, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_n, n_all.data(), sizeof(int64_t) * (size_t)nrows, hipMemcpyHostToDevice, s));
    const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nrows + 3) / 4, 4096));
    hipLaunchKernelGGL(k_compact_rows, dim3(g), dim3(BLOCK), 0, s, nrows, d_src, d_dst, d_n, pool_ci, pool_v, pc.ci, pc.v);
    HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(d_src); (void)hipFree(d_dst); (void)hipFree(d_n);
    (void)hipFree(d_small); (void)hipFree(d_big); (void)hipFree(d_off_s); (void)hipFree(d_off_b); (void)hipFree(d_fail_s); (void)hipFree(d_fail_b);
    pieces.push_back(pc);
    t_compact += now() - tc0;
    r0 = r1;
  }
  (void)hipFree(pool_ci); (void)hipFree(pool_v); (void)hipFree(d_cnt);
  // ---- C = concatenation of the pieces
  C.nnz = rp_host[m];
  HIP_CHECK(hipMemcpyAsync(C.rp, rp_host.data(), sizeof(int64_t) * (size_t)(m + 1), hipMemcpyHostToDevice, s));
  if (pieces.size() == 1) { C.ci = pieces[0].ci; C.v = pieces[0].v; }
  else {
    C.ci = dmalloc<int32_t>(C.nnz); C.v = dmalloc<double>(C.nnz);
    for (const Piece& pc : pieces) {
      if (pc.nnz) {
        HIP_CHECK(hipMemcpyAsync(C.ci + rp_host[pc.r0], pc.ci, sizeof(int32_t) * (size_t)pc.nnz, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipMemcpyAsync(C.v + rp_host[pc.r0], pc.v, sizeof(double) * (size_t)pc.nnz, hipMemcpyDeviceToDevice, s));
      }
    }
    HIP_CHECK(hipStreamSynchronize(s));
    for (const Piece& pc : pieces) { (void)hipFree(pc.ci); (void)hipFree(pc.v); }
  }
  HIP_CHECK(hipStreamSynchronize(s));
  if (trace) {
    int64_t ubt = 0; for (int64_t i = 0; i < m; i++) ubt += ub[i];
    std::printf("   [spgemm] m=%lld nnzA=%lld nnzB=%lld cand=%lld nnzC=%lld bins=%lld/%lld/%lld chunks=%zu  ub %.3f kernels %.3f compact %.3f total %.3f s\n",
                (long long)m, (long long)A.nnz, (long long)B.nnz, (long long)ubt, (long long)C.nnz, (long long)n_small, (long long)n_big,
                (long long)n_huge, pieces.size(), t_ub, t_kern, t_compact, now() - t_begin);
    std::fflush(stdout);
  }
  if (stats) { stats->rows_small = n_small; stats->rows_big = n_big; stats->rows_global = n_huge; }
  return C;
}

}  // namespace airhip
