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
#include "staged_copy.hpp"
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

struct DropParams { int enabled; double tol; int lump; int negate; };

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

// where the finished rows go: a pool with a bump allocator (the rows end up in completion order; a gather pass puts
// them in row order once all counts are known)
struct RowPool { unsigned long long* cursor; int64_t capacity; int32_t* ci; double* v; int64_t* row_pos; };
constexpr int ROW_OK = 0, ROW_TABLE_OVERFLOW = 1, ROW_POOL_FULL = 2;

// One output row by one wave.  tab: table of HT entries owned by this wave (HT power of two).
// Stores the number of kept entries in *count_out, takes that many entries from the pool and writes the row there.
// ROW_TABLE_OVERFLOW: the row must be redone with a larger table; ROW_POOL_FULL: with a fresh pool.
template <bool GLOBAL>
__device__ __forceinline__ int spgemm_row(const CsrView& A, const CsrView& B, int64_t i, const Tab<GLOBAL> tab, int HT,
                                          const DropParams& dp, int64_t* count_out, const RowPool& pool) {
#pragma clang fp contract(off)   // a*b + c must round twice, as the host code does
  constexpr bool FILL = true;
  int32_t* __restrict__ out_ci = pool.ci; double* __restrict__ out_v = pool.v;
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
  for (int64_t kb = a0; kb < a1 && !overflow; kb += WAVE) {
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
  if (overflow) return ROW_TABLE_OVERFLOW;
  // ---- compact the used slots to the front of the table (in place, chunk by chunk)
  int nu = 0;
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
  // ---- a place in the pool
  unsigned long long base = 0;
  if (lane == 0) base = atomicAdd(pool.cursor, (unsigned long long)kept);
  base = (unsigned long long)uniform64((int64_t)base);
  if ((int64_t)base + kept > pool.capacity) return ROW_POOL_FULL;
  if (lane == 0) { *count_out = kept; pool.row_pos[i] = (int64_t)base; }
  // ---- write the row in column order
  int64_t o = (int64_t)base;
  if (!dp.enabled) {
    for (int h = lane; h < nu; h += WAVE) { out_ci[o + h] = tab.ldk(h); out_v[o + h] = tab.ldv(h); }
    return ROW_OK;
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
  return ROW_OK;
}

template <int HT>
__global__ __launch_bounds__(HT <= 1024 ? BLOCK : WAVE) void k_spgemm_lds(CsrView A, CsrView B, const int32_t* __restrict__ rows,
                                                                            int64_t nrows, DropParams dp, int64_t* __restrict__ cnt,
                                                                            RowPool pool, int32_t* __restrict__ status) {
  constexpr int WPB = HT <= 1024 ? WAVES_PER_BLOCK : 1;
  __shared__ int32_t keys_all[WPB][HT];
  __shared__ double vals_all[WPB][HT];
  const int wid = threadIdx.x >> 6;
  const int64_t wave = (int64_t)blockIdx.x * WPB + wid;
  const int64_t nw = (int64_t)gridDim.x * WPB;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const Tab<false> tab{keys_all[wid], vals_all[wid]};
    const int st = spgemm_row<false>(A, B, i, tab, HT, dp, cnt + i, pool);
    if (st != ROW_OK && (threadIdx.x & 63) == 0) status[w] = st;
    __builtin_amdgcn_wave_barrier();
  }
}

// rows that overflowed the LDS table: one wave per row with a private global-memory table
__global__ __launch_bounds__(BLOCK) void k_spgemm_glb(CsrView A, CsrView B, const int32_t* __restrict__ rows, int64_t nrows,
                                                     const int64_t* __restrict__ tab_off, const int32_t* __restrict__ tab_size,
                                                     int32_t* keys_pool, double* vals_pool, DropParams dp, int64_t* __restrict__ cnt,
                                                     RowPool pool, int32_t* __restrict__ status) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const Tab<true> tab{keys_pool + tab_off[w], vals_pool + tab_off[w]};
    const int st = spgemm_row<true>(A, B, i, tab, tab_size[w], dp, cnt + i, pool);
    if (st != ROW_OK && (threadIdx.x & 63) == 0) status[w] = st;
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
  staged_h2d(d.rp, rp, sizeof(int64_t) * (size_t)(m + 1));
  if (d.nnz) {
    staged_h2d(d.ci, ci, sizeof(int32_t) * (size_t)d.nnz);
    staged_h2d(d.v, v, sizeof(double) * (size_t)d.nnz);
  }
  return d;
}

void dcsr_download(const DCsr& d, int64_t* rp, int32_t* ci, double* v) {
  staged_d2h(rp, d.rp, sizeof(int64_t) * (size_t)(d.m + 1));
  if (d.nnz) {
    staged_d2h(ci, d.ci, sizeof(int32_t) * (size_t)d.nnz);
    staged_d2h(v, d.v, sizeof(double) * (size_t)d.nnz);
  }
}

// host-side exclusive scan of a downloaded count array (m int64: cheap next to the product itself)
static int64_t scan_counts(std::vector<int64_t>& c) {
  int64_t run = 0;
  for (auto& x : c) { int64_t t = x; x = run; run += t; }
  c.push_back(run);
  return run;
}

// the finished rows from their places in a pool to row order; one wave per row
__global__ __launch_bounds__(BLOCK) void k_gather_rows(int64_t nrows, const int32_t* __restrict__ rows, const int64_t* __restrict__ row_pos,
                                                      const int64_t* __restrict__ rp, const int32_t* __restrict__ pci,
                                                      const double* __restrict__ pv, int32_t* __restrict__ oci, double* __restrict__ ov) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows ? (int64_t)rows[w] : w;
    const int64_t a = row_pos[i], o = rp[i], n = rp[i + 1] - o;
    for (int64_t k = lane; k < n; k += WAVE) { oci[o + k] = pci[a + k]; ov[o + k] = pv[a + k]; }
  }
}

// One numeric pass (round 2; round 1 ran every product twice, once to count and once to fill): every wave computes
// its row, takes as many entries as the row keeps from a pool (atomic bump allocator) and writes the sorted, filtered
// row there; once all counts are known a gather pass puts the rows in row order.  The pool is sized generously from
// the free device memory; if it still fills up, the rows that found no place are redone with a fresh pool.
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
  const DropParams dp{drop_enabled ? 1 : 0, tol, lump ? 1 : 0, negate ? 1 : 0};
  // ---- candidate upper bounds and binning (host side: m ints)
  int64_t* d_ub = dmalloc<int64_t>(m);
  hipLaunchKernelGGL(k_row_ub, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, av, bv, d_ub);
  std::vector<int64_t> ub(m);
  HIP_CHECK(hipMemcpyAsync(ub.data(), d_ub, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d_ub);
  const double t_ub = now() - t_begin;
  std::vector<int32_t> small, big, huge;
  small.reserve(m);
  int64_t ub_total = 0;
  for (int64_t i = 0; i < m; i++) { (ub[i] <= 768 ? small : big).push_back((int32_t)i); ub_total += std::min<int64_t>(ub[i], B.n) + 1; }
  int64_t* d_cnt = dmalloc<int64_t>(m + 1);
  int64_t* d_pos = dmalloc<int64_t>(m);
  unsigned long long* d_cursor = dmalloc<unsigned long long>(1);
  auto grid_small = [&](size_t n) { int64_t b = ((int64_t)n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK; return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(b, 256 * 12))); };
  auto grid_big = [&](size_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((int64_t)n, 256 * 3))); };
  // run one list through a kernel and split it by outcome
  auto run_list = [&](std::vector<int32_t>& rows, int kind, const RowPool& pool, std::vector<int32_t>& done, std::vector<int32_t>& overflow,
                      std::vector<int32_t>& retry) {
    if (rows.empty()) return;
    int32_t* d_rows = dmalloc<int32_t>((int64_t)rows.size());
    int32_t* d_st = dmalloc<int32_t>((int64_t)rows.size());
    HIP_CHECK(hipMemcpyAsync(d_rows, rows.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemsetAsync(d_st, 0, sizeof(int32_t) * rows.size(), s));
    int32_t* d_kpool = nullptr; double* d_vpool = nullptr; int64_t* d_toff = nullptr; int32_t* d_tsize = nullptr;
    if (kind == 0) hipLaunchKernelGGL((k_spgemm_lds<1024>), grid_small(rows.size()), dim3(BLOCK), 0, s, av, bv, d_rows, (int64_t)rows.size(), dp, d_cnt, pool, d_st);
    else if (kind == 1) hipLaunchKernelGGL((k_spgemm_lds<4096>), grid_big(rows.size()), dim3(WAVE), 0, s, av, bv, d_rows, (int64_t)rows.size(), dp, d_cnt, pool, d_st);
    else {
      std::vector<int64_t> toff(rows.size()); std::vector<int32_t> tsize(rows.size());
      int64_t run = 0;
      for (size_t w = 0; w < rows.size(); w++) {
        const int64_t want = 2 * std::min<int64_t>(ub[rows[w]], B.n) + 64;
        int64_t cap = 8192; while (cap < want) cap <<= 1;
        if (cap > 0x40000000ll) throw std::runtime_error("dev_spgemm: row too long");
        toff[w] = run; tsize[w] = (int32_t)cap; run += cap;
      }
      d_toff = dmalloc<int64_t>((int64_t)rows.size()); d_tsize = dmalloc<int32_t>((int64_t)rows.size());
      d_kpool = dmalloc<int32_t>(run); d_vpool = dmalloc<double>(run);
      HIP_CHECK(hipMemcpyAsync(d_toff, toff.data(), sizeof(int64_t) * rows.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipMemcpyAsync(d_tsize, tsize.data(), sizeof(int32_t) * rows.size(), hipMemcpyHostToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
      const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)rows.size() + 3) / 4, 2048);
      hipLaunchKernelGGL(k_spgemm_glb, dim3(gb), dim3(BLOCK), 0, s, av, bv, d_rows, (int64_t)rows.size(), d_toff, d_tsize, d_kpool, d_vpool, dp,
                         d_cnt, pool, d_st);
    }
    std::vector<int32_t> st(rows.size());
    HIP_CHECK(hipMemcpyAsync(st.data(), d_st, sizeof(int32_t) * st.size(), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (size_t w = 0; w < rows.size(); w++) {
      if (st[w] == ROW_OK) done.push_back(rows[w]);
      else if (st[w] == ROW_TABLE_OVERFLOW) { if (kind == 2) throw std::runtime_error("dev_spgemm: global hash table overflow"); overflow.push_back(rows[w]); }
      else retry.push_back(rows[w]);
    }
    (void)hipFree(d_rows); (void)hipFree(d_st);
    if (d_kpool) { (void)hipFree(d_kpool); (void)hipFree(d_vpool); (void)hipFree(d_toff); (void)hipFree(d_tsize); }
  };
  struct Round { int32_t* ci; double* v; std::vector<int32_t> done; };
  std::vector<Round> rounds;
  int64_t n_small = (int64_t)small.size(), n_big = (int64_t)big.size(), n_huge = 0;
  double t_kern = 0;
  int64_t longest = 0;
  for (int64_t i = 0; i < m; i++) longest = std::max<int64_t>(longest, std::min<int64_t>(ub[i], B.n) + 1);
  while (!small.empty() || !big.empty() || !huge.empty()) {
    // pool: a few times the inputs, never more than the sum of the bounds, never more than a third of the free memory
    size_t mem_free = 0, mem_total = 0;
    HIP_CHECK(hipMemGetInfo(&mem_free, &mem_total));
    int64_t capacity = std::max<int64_t>(4 * (A.nnz + B.nnz) + (1 << 20), 2 * longest);
    capacity = std::min<int64_t>(capacity, ub_total);
    capacity = std::min<int64_t>(capacity, (int64_t)(mem_free / 3 / 12));
    if (capacity < longest) throw std::runtime_error("dev_spgemm: not enough device memory for the output pool");
    Round rd; rd.ci = dmalloc<int32_t>(capacity); rd.v = dmalloc<double>(capacity);
    HIP_CHECK(hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s));
    const RowPool pool{d_cursor, capacity, rd.ci, rd.v, d_pos};
    const double tk0 = now();
    std::vector<int32_t> retry_s, retry_b, retry_h, over;
    run_list(small, 0, pool, rd.done, over, retry_s);
    run_list(big, 1, pool, rd.done, over, retry_b);
    huge.insert(huge.end(), over.begin(), over.end());
    n_huge += (int64_t)over.size();
    std::vector<int32_t> over2;
    run_list(huge, 2, pool, rd.done, over2, retry_h);
    t_kern += now() - tk0;
    const size_t left = retry_s.size() + retry_b.size() + retry_h.size();
    if (left && rd.done.empty()) throw std::runtime_error("dev_spgemm: no progress (output pool too small)");
    small.swap(retry_s); big.swap(retry_b); huge.swap(retry_h);
    rounds.push_back(std::move(rd));
  }
  // ---- row pointers, then the rows of every round into row order
  const double tg0 = now();
  std::vector<int64_t> cnt(m);
  HIP_CHECK(hipMemcpyAsync(cnt.data(), d_cnt, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  C.nnz = scan_counts(cnt);
  HIP_CHECK(hipMemcpyAsync(C.rp, cnt.data(), sizeof(int64_t) * (size_t)(m + 1), hipMemcpyHostToDevice, s));
  C.ci = dmalloc<int32_t>(C.nnz); C.v = dmalloc<double>(C.nnz);
  for (Round& rd : rounds) {
    const int64_t nr = (int64_t)rd.done.size();
    if (nr) {
      int32_t* d_rows = nullptr;
      if (rounds.size() > 1) {
        d_rows = dmalloc<int32_t>(nr);
        HIP_CHECK(hipMemcpyAsync(d_rows, rd.done.data(), sizeof(int32_t) * (size_t)nr, hipMemcpyHostToDevice, s));
      }
      const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nr + 3) / 4, 8192));
      hipLaunchKernelGGL(k_gather_rows, dim3(g), dim3(BLOCK), 0, s, nr, d_rows, d_pos, C.rp, rd.ci, rd.v, C.ci, C.v);
      HIP_CHECK(hipStreamSynchronize(s));
      if (d_rows) (void)hipFree(d_rows);
    }
    (void)hipFree(rd.ci); (void)hipFree(rd.v);
  }
  const double t_gather = now() - tg0;
  (void)hipFree(d_cnt); (void)hipFree(d_pos); (void)hipFree(d_cursor);
  if (trace) {
    std::printf("   [spgemm] m=%lld nnzA=%lld nnzB=%lld bound=%lld nnzC=%lld bins=%lld/%lld/%lld rounds=%zu  ub %.3f kernels %.3f gather %.3f total %.3f s\n",
                (long long)m, (long long)A.nnz, (long long)B.nnz, (long long)ub_total, (long long)C.nnz, (long long)n_small, (long long)n_big,
                (long long)n_huge, rounds.size(), t_ub, t_kern, t_gather, now() - t_begin);
    std::fflush(stdout);
  }
  if (stats) { stats->rows_small = n_small; stats->rows_big = n_big; stats->rows_global = n_huge; }
  return C;
}

}  // namespace airhip
