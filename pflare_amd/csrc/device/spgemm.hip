// Device SpGEMM with fused relative drop / lump, used by the PCAIR setup for the two
// products that dominate the hierarchy construction:  Z = -A_cf * M  (r_drop) and
// A_c = R * (A P)  (a_drop, optional lumping)  [src/AIR_Operators_Setup.F90:784-795,
// 1008-1069; device twins exist in the reference as Kokkos SpGEMMs].
//
// One WAVE computes one output row.  The sparse accumulator is an open-addressing hash
// table keyed by the column (LDS for the common case, a global-memory table for rows that
// do not fit; rows with at most 64 candidates skip the table and are sorted in registers).
// The rows of B are consumed one after the other (k ascending) with the 64
// lanes spread over the entries of B(k,:), whose columns are distinct - so no two lanes ever
// accumulate into the same key in the same step and every output entry is summed in exactly
// the order of a sequential Gustavson accumulator.  The finished row is sorted by column
// (bitonic, in the table), filtered with the semantics of remove_small_from_sparse
// (src/PETSc_Helper.F90:207-412; lumped sum taken in column order) and written in CSR order.
// Results are therefore bit-identical to the host implementation (host/spops.cpp).
#include "dev.hpp"
#include "kcommon.hpp"
#include "rb.hpp"
#include "spgemm.hpp"
#include "staged_copy.hpp"
#include "devmem.hpp"
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

// Accumulator kinds.  ub_i = sum_k nnz(B(k,:)) bounds the candidates of row i:
//   REG    ub <= 64 (and <= 64 entries in A(i,:)): the candidates sit one per lane and are sorted in registers
//   L1K    LDS hash table of 1024 slots, four waves (rows) per workgroup, 12 waves per CU
//   L2K    2048 slots, two waves per workgroup, 6 waves per CU
//   L4K    4096 slots, one wave per workgroup, 3 waves per CU
//   GLOBAL a hash table in device memory (rows with more than 3584 distinct columns)
// A table holds the DISTINCT columns of the row, and on the AIR operators those are 3-15 times fewer than the
// candidates (the rows of B that one row of A meets overlap heavily).  Sizing by ub put nearly every row of the deep
// levels into the 4k kernel, which runs at a quarter of the occupancy and pays for clearing, compacting and sorting
// 4096 slots; so the ratio distinct / candidates is measured on a sample of rows first (k_spgemm_sample) and a row
// goes to the smallest table that ub * ratio fits.  A row that overflows its table all the same moves up one kind
// (1k -> 2k -> 4k through device lists, 4k -> GLOBAL through the host).
constexpr int KIND_REG = 0, KIND_L1K = 1, KIND_L2K = 2, KIND_L4K = 3, KIND_GLOBAL = 4, NKIND = 5;
constexpr int TINY_MAX = 64, L1K_MAX = 768, L2K_MAX = 1536;     // bounds a table of that kind cannot overflow on
constexpr int L1K_FILL = 800, L2K_FILL = 1650;                  // distinct columns the tables are planned for (overflow: 896 / 1792)
constexpr int MAX_HT = 4096;
struct Plan { int64_t max1k, max2k; float ratio; };              // largest bounds the 1k / 2k tables take; distinct per candidate
__host__ __device__ inline int kind_of(int64_t ub, int64_t na, const Plan& p) {
  return (ub <= TINY_MAX && na <= TINY_MAX) ? KIND_REG : ub <= p.max1k ? KIND_L1K : ub <= p.max2k ? KIND_L2K : KIND_L4K;
}

// what the bounds add up to (decides the pool size; no per-row list leaves the device)
struct UbStats { unsigned long long total, longest, above_1k; };

// per-row upper bound of candidates
__global__ __launch_bounds__(BLOCK) void k_row_ub(CsrView A, CsrView B, int64_t* __restrict__ ub, UbStats* __restrict__ st) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  unsigned long long tot = 0, longest = 0, above = 0;
  for (int64_t i = wave; i < A.m; i += nw) {
    int64_t s = 0;
    const int64_t a0 = A.rp[i], a1 = A.rp[i + 1];
    for (int64_t k = a0 + lane; k < a1; k += WAVE) { const int32_t j = A.ci[k]; s += B.rp[j + 1] - B.rp[j]; }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    if (lane == 0) ub[i] = s;
    const unsigned long long c = (unsigned long long)(s < B.n ? s : B.n) + 1ull;
    tot += c; longest = c > longest ? c : longest; above += s > L1K_MAX ? 1 : 0;
  }
  if (lane == 0) { atomicAdd(&st->total, tot); atomicMax(&st->longest, longest); if (above) atomicAdd(&st->above_1k, above); }
}

// longest row of a CSR matrix (is B a matrix of single-entry rows?)
__global__ __launch_bounds__(BLOCK) void k_longest_row(int64_t m, const int64_t* __restrict__ rp, unsigned long long* __restrict__ out) {
  unsigned long long mx = 0;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) { const unsigned long long l = (unsigned long long)(rp[i + 1] - rp[i]); mx = l > mx ? l : mx; }
  for (int off = 32; off > 0; off >>= 1) { const unsigned long long t = __shfl_xor(mx, off, WAVE); mx = t > mx ? t : mx; }
  if ((threadIdx.x & (WAVE - 1)) == 0 && mx) atomicMax(out, mx);
}

// the rows a kernel works on: an explicit list, or (rows == nullptr) every row of A whose bound selects `kind`
struct RowSel { const int32_t* rows; int64_t nrows; const int64_t* ub; int kind; Plan plan; int b_single; };   // b_single: no row of B has more than one entry

// rows that move up one accumulator kind, and the rows each kind has seen
struct Escalation { int32_t* rows; unsigned int* count; };

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
  // vals[h] += x, one rounding.  LDS: ds_add_f64 without a return value - the wave does not wait for it, and since a
  // wave's LDS instructions execute in order, additions to one slot land in the order they were issued.
  __device__ __forceinline__ void add(int h, double x) const {
    if (GLOBAL) stv(h, ldv(h) + x);
    else (void)__hip_atomic_fetch_add(&vals[h], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
  __device__ __forceinline__ void sync() const {
    if (GLOBAL) __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent"); else __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// where the finished rows go: a pool with a bump allocator (the rows end up in completion order; a gather pass puts
// them in row order once all counts are known)
// One atomic on the cursor per row caps the product at 80 M rows/s (same-address atomics take ~12 ns each at L2), which
// is what the register kernel ran at on the fine levels whatever the rows held: a wave takes `chunk` entries at a time
// and hands them out to its rows itself (the rest of its last chunk stays unused; the host sizes the pool for that).
struct RowPool { unsigned long long* cursor; int64_t capacity; int32_t* ci; double* v; int64_t* row_pos; int chunk; };
struct WaveAlloc { int64_t pos = 0, end = 0; };
constexpr int ROW_OK = 0, ROW_TABLE_OVERFLOW = 1, ROW_POOL_FULL = 2;

// value of lane l (wave-uniform l) as a scalar
__device__ __forceinline__ int lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int64_t lane_i64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// ---- sort of an LDS table's nu compacted (column, value) pairs by column, sample sort instead of the bitonic network
// (which costs ~25 nu log^2 nu / 64 instructions per wave and was 25-40 % of the table kernels on the deep levels):
//   4 NB sampled columns are sorted in registers, every fourth becomes a splitter (NB = 64 or 128 buckets of ~nu / NB
//   columns, twice that in practice); every pair - all held in registers, CAP / 64 per lane - finds its bucket by
//   binary search and takes a place in it (LDS counter); the columns are regrouped by bucket in place; the rank of a
//   pair is its bucket's offset plus the number of smaller columns in its bucket, counted by reading the bucket
//   through; the pairs are stored at their ranks.  ~3 nu instructions per wave for nu ~ 1000.
// Returns false (table untouched) when the scratch words at the end of the key array are not free or a bucket holds
// more than BUCKET_MAX columns; the caller then runs the bitonic network.  CAP = slots of the wave's table region.
template <int CAP>
__device__ __forceinline__ bool lds_bucket_sort(const Tab<false> tab, int nu) {
  constexpr int E = CAP / WAVE;                      // pairs per lane
  constexpr int NB = CAP >= 2048 ? 128 : 64;         // buckets
  constexpr int Q = NB / WAVE;                       // buckets per lane in the scan
  constexpr int S = 4 * Q;                           // samples per lane (4 NB in all)
  constexpr int BUCKET_MAX = 192;
  if (nu > CAP - 3 * NB) return false;
  const int lane = threadIdx.x & (WAVE - 1);
  int32_t* spl = tab.keys + (CAP - 3 * NB);          // splitters | counters | offsets
  int32_t* cnt = spl + NB;
  int32_t* off = cnt + NB;
  // ---- samples, blocked over the lanes (element S * lane + r), bitonic on 32-bit keys: min / max only
  int32_t sm[S];
#pragma unroll
  for (int r = 0; r < S; r++) sm[r] = tab.ldk((int)(((long long)(lane * S + r) * nu) / (WAVE * S)));
  const int e0 = lane * S;
#pragma unroll
  for (int size = 2; size <= WAVE * S; size <<= 1)
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      if (stride < S) {
#pragma unroll
        for (int r = 0; r < S; r++) {
          if (r & stride) continue;
          const bool asc = (((e0 + r) & size) == 0);
          const int32_t x = sm[r], y = sm[r | stride];
          const int32_t lo = x < y ? x : y, hi = x < y ? y : x;
          sm[r] = asc ? lo : hi; sm[r | stride] = asc ? hi : lo;
        }
      } else {
        const int addr = (lane ^ (stride / S)) << 2;
        const bool want_min = ((e0 & stride) == 0) == ((e0 & size) == 0);
#pragma unroll
        for (int r = 0; r < S; r++) {
          const int32_t o = __builtin_amdgcn_ds_bpermute(addr, sm[r]);
          const int32_t lo = sm[r] < o ? sm[r] : o, hi = sm[r] < o ? o : sm[r];
          sm[r] = want_min ? lo : hi;
        }
      }
    }
  // splitter j = sample 4 j + 3 = lane j / Q, register 4 (j % Q) + 3; the last bucket is open-ended
#pragma unroll
  for (int q = 0; q < Q; q++) {
    const int j = lane * Q + q;
    spl[j] = j < NB - 1 ? sm[4 * q + 3] : 0x7fffffff;
    cnt[j] = 0;
  }
  tab.sync();
  // ---- every pair into registers, its bucket and its place in the bucket
  int32_t k[E]; double v[E]; int place[E];           // place = bucket << 16 | index in the bucket, then the rank
#pragma unroll
  for (int r = 0; r < E; r++) {
    k[r] = 0; v[r] = 0.0; place[r] = 0;
    if (r * WAVE >= nu) continue;                    // wave-uniform
    const int e = r * WAVE + lane;
    if (e < nu) {
      k[r] = tab.ldk(e); v[r] = tab.ldv(e);
      int b = 0;                                     // number of splitters < k: branch-free binary search
#pragma unroll
      for (int step = NB / 2; step > 0; step >>= 1) b += (spl[b + step - 1] < k[r]) ? step : 0;
      const int li = (int)__hip_atomic_fetch_add((unsigned int*)&cnt[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
      place[r] = (b << 16) | li;
    }
  }
  tab.sync();
  // ---- bucket offsets (exclusive scan); a bucket that outgrew BUCKET_MAX columns sends the row to the network
  int c[Q], mine = 0, longest = 0;
#pragma unroll
  for (int q = 0; q < Q; q++) { c[q] = cnt[lane * Q + q]; mine += c[q]; longest = c[q] > longest ? c[q] : longest; }
  for (int d = 32; d > 0; d >>= 1) { const int t = __shfl_xor(longest, d, WAVE); longest = t > longest ? t : longest; }
  longest = __builtin_amdgcn_readfirstlane(longest);
  if (longest > BUCKET_MAX) return false;            // nothing has been moved yet
  int incl = mine;
  for (int d = 1; d < WAVE; d <<= 1) { const int t = __shfl_up(incl, d, WAVE); if (lane >= d) incl += t; }
  int run = incl - mine;
#pragma unroll
  for (int q = 0; q < Q; q++) { off[lane * Q + q] = run; run += c[q]; }
  tab.sync();
  // ---- columns regrouped by bucket, in place (every column is in a register by now)
#pragma unroll
  for (int r = 0; r < E; r++) {
    if (r * WAVE >= nu) continue;
    if (r * WAVE + lane < nu) tab.stk(off[place[r] >> 16] + (place[r] & 0xffff), k[r]);
  }
  tab.sync();
  // ---- rank = bucket offset + smaller columns in the bucket
#pragma unroll
  for (int r = 0; r < E; r++) {
    if (r * WAVE >= nu) continue;
    const bool in = r * WAVE + lane < nu;
    const int b = place[r] >> 16;
    const int o = in ? off[b] : 0, n = in ? cnt[b] : 0;
    int rank = o;
    for (int j = 0; j < longest; j++) rank += (j < n && tab.ldk(o + j) < k[r]) ? 1 : 0;
    place[r] = rank;
  }
  tab.sync();
  // ---- the pairs at their ranks
#pragma unroll
  for (int r = 0; r < E; r++) {
    if (r * WAVE >= nu) continue;
    if (r * WAVE + lane < nu) { tab.stk(place[r], k[r]); tab.stv(place[r], v[r]); }
  }
  tab.sync();
  return true;
}

// ---- accumulate, hash table: leaves the nu distinct columns of row i of A*B with their sums in tab[0, nu), sorted
// by column; -1 = the table is too small for the row.  tab: HT slots owned by this wave (HT a power of two).
// CAP: slots of the wave's LDS region (0: no sample sort - global tables, counting only).
// dummy (LDS tables): first of 64 slots outside the table that stay EMPTY, one per lane (same-address LDS atomics
// serialise) - a lane without a candidate aims its compare-and-swaps and its additions of 0 at its own, so that the
// probe rounds need no per-step exec masks.
template <bool GLOBAL, int CAP>
__device__ __forceinline__ int hash_accumulate(const CsrView& A, const CsrView& B, int64_t i, const Tab<GLOBAL> tab, int HT, bool negate,
                                               bool count_only = false, int dummy = 0, bool b_single = false) {
#pragma clang fp contract(off)   // a*b + c must round twice, as the host code does
  const int lane = threadIdx.x & (WAVE - 1);
  const uint32_t mask = (uint32_t)HT - 1;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int h = lane; h < HT; h += WAVE) { tab.stk(h, EMPTY); tab.stv(h, 0.0); }
  if (!GLOBAL) { tab.stk(dummy + lane, EMPTY); tab.stv(dummy + lane, 0.0); }
  tab.sync();
  const int64_t a0 = uniform64(A.rp[i]), a1 = uniform64(A.rp[i + 1]);
  // The candidate stream of the row is consumed in "steps": one step = up to 64 consecutive entries of one row
  // B(j,:) (distinct columns -> conflict-free lanes), steps ordered by k.  The walk over (entry of A, offset in its
  // row of B) is scalar: lane e of a chunk holds entry kb + e of A(i,:) and the extent of B(j_e,:), the current
  // entry's base / length / value sit in SGPRs (readlane), so a step costs a handful of vector instructions.  PIPE
  // steps form a batch: their loads are issued back to back, then their slots are looked up TOGETHER (every probe
  // round issues the compare-and-swaps of all pending steps before it waits), then the products are added in step
  // order.  Which step claims a new column first does not matter (the row is sorted afterwards); the order of the
  // additions does, and that is the order of issue.
  constexpr int PIPE = 8;
  constexpr int MAX_PROBE_ROUNDS = 64;     // double hashing: a table filled to 3/4 needs more with probability 1e-8 per insert
  if (b_single) {
    // Every row of B has at most one entry (A * P with a one-point prolongator: the product is A with its columns
    // relabelled and merged).  The walk above would spend a whole step on each entry of A; here 64 entries of A form
    // a step: the slots are claimed together, and since several lanes of a step may carry the same column, the
    // additions are made one lane after the other, in the order of the entries of A - the Gustavson order.
    for (int64_t kb = a0; kb < a1; kb += WAVE) {
      const int64_t kk = kb + lane;
      int32_t c = EMPTY; double x = 0.0;
      if (kk < a1) {
        const int32_t j = A.ci[kk];
        const int64_t b0 = B.rp[j];
        if (B.rp[j + 1] > b0) { c = B.ci[b0]; x = A.v[kk] * B.v[b0]; }
      }
      const bool act = c != EMPTY;
      uint32_t h = 0;
      if (act) {
        const uint32_t g = (uint32_t)c * 2654435761u;
        h = (g ^ (g >> 15)) & mask;
        const uint32_t inc = ((g >> 19) | 1u) & mask;
        bool failed = false;
        for (int rounds = 0;; rounds++) {
          const int32_t old = tab.cas((int)h, c);
          if (old == EMPTY || old == c) break;
          h = (h + inc) & mask;
          if (rounds > MAX_PROBE_ROUNDS) { failed = true; break; }
        }
        if (failed) h = 0xffffffffu;
      }
      if (__any(act && h == 0xffffffffu)) return -1;      // (nearly) full table
      unsigned long long todo = __ballot(act);
      while (todo) {
        const int l = __ffsll((long long)todo) - 1;
        if (lane == l) tab.add((int)h, x);
        todo &= todo - 1;
      }
      tab.sync();
    }
  } else
  for (int64_t kb = a0; kb < a1; kb += WAVE) {
    const int nk = (int)(a1 - kb < WAVE ? a1 - kb : WAVE);
    double al = 0.0; int64_t b0 = 0; int len = 0;
    if (lane < nk) { const int32_t j = A.ci[kb + lane]; al = A.v[kb + lane]; b0 = B.rp[j]; len = (int)(B.rp[j + 1] - b0); }
    const long long al_bits = __double_as_longlong(al);
    int e = 0, off = 0;
    int cur_len = lane_i32(len, 0);
    int64_t cur_b0 = lane_i64(b0, 0);
    double cur_a = __longlong_as_double(lane_i64(al_bits, 0));
    while (e < nk) {
      int32_t cc[PIPE]; double x[PIPE], av[PIPE]; bool act[PIPE], pend[PIPE]; uint32_t hh[PIPE], inc[PIPE];
#pragma unroll
      for (int d = 0; d < PIPE; d++) {
        act[d] = false; cc[d] = 0; x[d] = 0.0; av[d] = 0.0;
        if (e < nk) {                                   // wave-uniform
          const uint32_t o = (uint32_t)(off + lane);
          act[d] = o < (uint32_t)cur_len;
          av[d] = cur_a;
          if (act[d]) { cc[d] = (B.ci + cur_b0)[o]; x[d] = (B.v + cur_b0)[o]; }   // no arithmetic: the batch's loads stay in flight together
          off += WAVE;
          if (off >= cur_len) {                         // next entry of A (an empty row of B costs one idle step)
            e++; off = 0;
            if (e < nk) { cur_len = lane_i32(len, e); cur_b0 = lane_i64(b0, e); cur_a = __longlong_as_double(lane_i64(al_bits, e)); }
          }
        }
      }
#pragma unroll
      for (int d = 0; d < PIPE; d++) {
        x[d] = av[d] * x[d];
        const uint32_t g = (uint32_t)cc[d] * 2654435761u;
        hh[d] = (g ^ (g >> 15)) & mask;
        inc[d] = ((g >> 19) | 1u) & mask;               // odd: the probe sequence visits every slot
        pend[d] = act[d];
      }
      int rounds = 0;
      if constexpr (GLOBAL) {
        for (;;) {
          int32_t old[PIPE];
#pragma unroll
          for (int d = 0; d < PIPE; d++) old[d] = pend[d] ? tab.cas((int)hh[d], cc[d]) : 0;
          bool more = false;
#pragma unroll
          for (int d = 0; d < PIPE; d++) {
            if (pend[d]) {
              if (old[d] == EMPTY || old[d] == cc[d]) pend[d] = false;
              else { hh[d] = (hh[d] + inc[d]) & mask; more = true; }
            }
          }
          if (!__any(more)) break;
          if (++rounds > MAX_PROBE_ROUNDS) return -1;     // (nearly) full table
        }
#pragma unroll
        for (int d = 0; d < PIPE; d++) if (act[d]) tab.add((int)hh[d], x[d]);
      } else {
        // branch-free rounds: a lane that has found its slot repeats the compare-and-swap there (it returns its own
        // column), a lane without a candidate works on the dummy slot
#pragma unroll
        for (int d = 0; d < PIPE; d++) if (!act[d]) { cc[d] = EMPTY; hh[d] = (uint32_t)(dummy + lane); inc[d] = 0; x[d] = 0.0; }
        for (;;) {
          int32_t old[PIPE];
#pragma unroll
          for (int d = 0; d < PIPE; d++) old[d] = tab.cas((int)hh[d], cc[d]);
          bool more = false;
#pragma unroll
          for (int d = 0; d < PIPE; d++) {
            const bool miss = old[d] != EMPTY && old[d] != cc[d];
            hh[d] = miss ? ((hh[d] + inc[d]) & mask) : hh[d];
            more |= miss;
          }
          if (!__any(more)) break;
          if (++rounds > MAX_PROBE_ROUNDS) return -1;     // (nearly) full table
        }
#pragma unroll
        for (int d = 0; d < PIPE; d++) tab.add((int)hh[d], x[d]);
      }
      tab.sync();
    }
  }
  // ---- compact the used slots to the front of the table (in place, chunk by chunk)
  int nu = 0;
  for (int base = 0; base < HT; base += WAVE) {
    const int32_t k = tab.ldk(base + lane);
    const double v = tab.ldv(base + lane);
    const bool used = k != EMPTY;
    const unsigned long long bal = __ballot(used);
    const int pos = nu + __popcll(bal & lt);
    tab.sync();
    if (used) { tab.stk(pos, k); tab.stv(pos, negate ? -v : v); }
    nu += __popcll(bal);
    tab.sync();
  }
  if (count_only) return nu;
  if constexpr (!GLOBAL && CAP > 0) {
    if (nu > 128 && lds_bucket_sort<CAP>(tab, nu)) return nu;
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
  return nu;
}

// ---- accumulate, registers: a row with at most 64 candidates (and at most 64 entries in A(i,:)).  Lane c holds
// candidate c of the sequential Gustavson sweep; a bitonic sort over the lanes by (column, c) brings equal columns
// together in sweep order, the first lane of every run adds the run up from left to right - the order of the host
// code - and the run heads go to tab[0, nu) (sorted).
__device__ __forceinline__ int tiny_accumulate(const CsrView& A, const CsrView& B, int64_t i, const Tab<false> tab, bool negate) {
#pragma clang fp contract(off)
  const int lane = threadIdx.x & (WAVE - 1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t a0 = A.rp[i];
  const int na = (int)(A.rp[i + 1] - a0);
  double al = 0.0; int64_t b0 = 0; int nb = 0;
  if (lane < na) { const int32_t j = A.ci[a0 + lane]; al = A.v[a0 + lane]; b0 = B.rp[j]; nb = (int)(B.rp[j + 1] - b0); }
  int incl = nb;
  for (int off = 1; off < WAVE; off <<= 1) { const int t = __shfl_up(incl, off, WAVE); if (lane >= off) incl += t; }
  const int excl = incl - nb;
  int64_t q = -1; double a = 0.0;
  for (int e = 0; e < na; e++) {          // e is wave-uniform: the owner's values come through readlane
    const int se = lane_i32(excl, e), ne = lane_i32(nb, e);
    const int64_t be = lane_i64(b0, e);
    const double ae = __longlong_as_double(lane_i64(__double_as_longlong(al), e));
    if (lane >= se && lane < se + ne) { q = be + (lane - se); a = ae; }
  }
  int32_t col = EMPTY; double x = 0.0;
  if (q >= 0) { col = B.ci[q]; x = a * B.v[q]; }
  uint64_t key = ((uint64_t)(uint32_t)col << 6) | (uint32_t)lane;          // distinct keys: no ties
  // bitonic network over the first P lanes (P = the candidates rounded up to a power of two; the other lanes hold
  // EMPTY, which sorts last anyway).  Partner values come through ds_bpermute with the lane address of the stride.
  const int total = lane_i32(incl, WAVE - 1);
#pragma unroll
  for (int size = 2; size <= WAVE; size <<= 1) {
    if (size / 2 >= total) break;                       // wave-uniform
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      const int addr = (lane ^ stride) << 2;
      const uint32_t klo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)key);
      const uint32_t khi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)(key >> 32));
      const long long xb = __double_as_longlong(x);
      const uint32_t xlo = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)xb);
      const uint32_t xhi = (uint32_t)__builtin_amdgcn_ds_bpermute(addr, (int)(uint32_t)((unsigned long long)xb >> 32));
      const uint64_t ok = ((uint64_t)khi << 32) | klo;
      const bool want_min = ((lane & stride) == 0) == ((lane & size) == 0);
      if ((ok < key) == want_min) { key = ok; x = __longlong_as_double((long long)(((uint64_t)xhi << 32) | xlo)); }
    }
  }
  col = (int32_t)(key >> 6);
  const int32_t prev = __shfl_up(col, 1, WAVE);
  const bool head = col != EMPTY && (lane == 0 || prev != col);
  double s = 0.0 + x;                                   // the accumulator of the host code starts from +0
  for (int r = 1; r < WAVE; r++) {
    const int32_t cn = __shfl_down(col, r, WAVE);
    const double xn = __shfl_down(x, r, WAVE);
    const bool same = head && lane + r < WAVE && cn == col;
    if (!__any(same)) break;                            // runs are contiguous: no run reaches further either
    if (same) s = s + xn;
  }
  const unsigned long long bal = __ballot(head);
  if (head) { const int pos = __popcll(bal & lt); tab.stk(pos, col); tab.stv(pos, negate ? -s : s); }
  tab.sync();
  return __popcll(bal);
}

// ---- filter and write: tab[0, nu) holds the sorted row.  Stores the number of kept entries in *count_out, takes that
// many entries from the pool and writes the row there.  ROW_POOL_FULL: the row must be redone with a fresh pool.
template <bool GLOBAL>
__device__ __forceinline__ int finish_row(const Tab<GLOBAL> tab, int nu, int64_t i, const DropParams& dp, int64_t* count_out, const RowPool& pool,
                                          WaveAlloc& wa) {
#pragma clang fp contract(off)
  int32_t* __restrict__ out_ci = pool.ci; double* __restrict__ out_v = pool.v;
  const int lane = threadIdx.x & (WAVE - 1);
  const unsigned long long lt = (1ull << lane) - 1ull;
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
    if (write_diag) {
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
  kept = uniform64(kept);
  if (wa.pos + kept > wa.end) {
    const int64_t want = kept > pool.chunk ? kept : (int64_t)pool.chunk;
    unsigned long long base = 0;
    if (lane == 0) base = atomicAdd(pool.cursor, (unsigned long long)want);
    const int64_t b = uniform64((int64_t)base);
    if (b + want > pool.capacity) return ROW_POOL_FULL;
    wa.pos = b; wa.end = b + want;
  }
  int64_t o = wa.pos;
  wa.pos += kept;
  if (lane == 0) { *count_out = kept; pool.row_pos[i] = o; }
  // ---- write the row in column order
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

// a row that found no table / no place: cnt[i] = -status, and the launch's failure counters go up
__device__ __forceinline__ void report(int st, int64_t i, int64_t* __restrict__ cnt, unsigned int* __restrict__ fail) {
  if (st != ROW_OK && (threadIdx.x & (WAVE - 1)) == 0) { cnt[i] = -(int64_t)st; atomicAdd(&fail[st - 1], 1u); }
}

// the smallest table (>= 128 slots, <= cap) that `expect` distinct columns fill to 3/4 at most
__device__ __forceinline__ int table_for(int64_t expect, int cap) {
  int ht = 128;
  while (ht < cap && (int64_t)ht * 3 < expect * 4) ht <<= 1;
  return ht;
}

// HT = the wave's share of LDS.  A row starts with the table its expected number of distinct columns asks for (short
// tables are cheaper to clear, compact and sort), and is redone with all HT slots if that overflows; a row that
// overflows HT slots moves up (esc) or is reported.
template <int HT>
__global__ __launch_bounds__(MAX_HT / HT * WAVE) void k_spgemm_lds(CsrView A, CsrView B, RowSel sel, DropParams dp, int64_t* __restrict__ cnt,
                                                                  RowPool pool, unsigned int* __restrict__ fail, Escalation esc,
                                                                  unsigned long long* __restrict__ seen) {
  constexpr int WPB = MAX_HT / HT;
  __shared__ int32_t keys_all[WPB][HT + WAVE];       // + the dummy slots (hash_accumulate)
  __shared__ double vals_all[WPB][HT + WAVE];
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * WPB;
  const int64_t count = sel.rows ? sel.nrows : A.m;
  const Tab<false> tab{keys_all[wid], vals_all[wid]};
  unsigned long long mine = 0;
  WaveAlloc wa;
  for (int64_t w = (int64_t)blockIdx.x * WPB + wid; w < count; w += nw) {
    const int64_t i = sel.rows ? (int64_t)sel.rows[w] : w;
    const int64_t ubi = sel.ub[i];
    if (!sel.rows && kind_of(ubi, A.rp[i + 1] - A.rp[i], sel.plan) != sel.kind) continue;
    mine++;
    const int64_t expect = ubi <= L1K_MAX ? ubi : (int64_t)((float)ubi * sel.plan.ratio) + 64;
    const int ht = table_for(expect < ubi ? expect : ubi, HT);
    int nu = hash_accumulate<false, HT>(A, B, i, tab, ht, dp.negate != 0, false, HT, sel.b_single != 0);
    if (nu < 0 && ht < HT) nu = hash_accumulate<false, HT>(A, B, i, tab, HT, dp.negate != 0, false, HT, sel.b_single != 0);
    if (nu < 0 && esc.rows) {
      if ((threadIdx.x & (WAVE - 1)) == 0) esc.rows[atomicAdd(esc.count, 1u)] = (int32_t)i;
    } else {
      report(nu < 0 ? ROW_TABLE_OVERFLOW : finish_row<false>(tab, nu, i, dp, cnt + i, pool, wa), i, cnt, fail);
    }
    __builtin_amdgcn_wave_barrier();
  }
  if ((threadIdx.x & (WAVE - 1)) == 0 && mine) atomicAdd(seen, mine);
}

__global__ __launch_bounds__(BLOCK) void k_spgemm_tiny(CsrView A, CsrView B, RowSel sel, DropParams dp, int64_t* __restrict__ cnt, RowPool pool,
                                                      unsigned int* __restrict__ fail, unsigned long long* __restrict__ seen) {
  __shared__ int32_t keys_all[WAVES_PER_BLOCK][WAVE];
  __shared__ double vals_all[WAVES_PER_BLOCK][WAVE];
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * WAVES_PER_BLOCK;
  const int64_t count = sel.rows ? sel.nrows : A.m;
  const Tab<false> tab{keys_all[wid], vals_all[wid]};
  unsigned long long mine = 0;
  WaveAlloc wa;
  for (int64_t w = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; w < count; w += nw) {
    const int64_t i = sel.rows ? (int64_t)sel.rows[w] : w;
    if (!sel.rows && kind_of(sel.ub[i], A.rp[i + 1] - A.rp[i], sel.plan) != KIND_REG) continue;
    mine++;
    const int nu = tiny_accumulate(A, B, i, tab, dp.negate != 0);
    report(finish_row<false>(tab, nu, i, dp, cnt + i, pool, wa), i, cnt, fail);
    tab.sync();
  }
  if ((threadIdx.x & (WAVE - 1)) == 0 && mine) atomicAdd(seen, mine);
}

// distinct columns of a sample of the rows with more than L1K_MAX candidates: sample w looks at row w * m / nsamples
// and stores its bound and its number of distinct columns (-1: more than the 4k table holds, -2: not such a row)
__global__ __launch_bounds__(WAVE) void k_spgemm_sample(CsrView A, CsrView B, const int64_t* __restrict__ ub, int nsamples,
                                                       int64_t* __restrict__ out_ub, int32_t* __restrict__ out_nu, int b_single) {
  __shared__ int32_t keys[MAX_HT + WAVE];
  __shared__ double vals[MAX_HT + WAVE];
  const Tab<false> tab{keys, vals};
  for (int w = blockIdx.x; w < nsamples; w += gridDim.x) {
    const int64_t i = ((int64_t)w * A.m) / nsamples;
    const int64_t u = ub[i];
    const int nu = u > L1K_MAX ? hash_accumulate<false, 0>(A, B, i, tab, MAX_HT, false, true, MAX_HT, b_single != 0) : -2;
    if (threadIdx.x == 0) { out_ub[w] = u; out_nu[w] = nu; }
    __builtin_amdgcn_wave_barrier();
  }
}

// rows that overflowed the LDS table: one wave per row with a private global-memory table
__global__ __launch_bounds__(BLOCK) void k_spgemm_glb(CsrView A, CsrView B, const int32_t* __restrict__ rows, int64_t nrows,
                                                     const int64_t* __restrict__ tab_off, const int32_t* __restrict__ tab_size,
                                                     int32_t* keys_pool, double* vals_pool, DropParams dp, int64_t* __restrict__ cnt,
                                                     RowPool pool, unsigned int* __restrict__ fail) {
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  WaveAlloc wa;
  for (int64_t w = wave; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const Tab<true> tab{keys_pool + tab_off[w], vals_pool + tab_off[w]};
    const int nu = hash_accumulate<true, 0>(A, B, i, tab, tab_size[w], dp.negate != 0);
    report(nu < 0 ? ROW_TABLE_OVERFLOW : finish_row<true>(tab, nu, i, dp, cnt + i, pool, wa), i, cnt, fail);
  }
}

template <class T> T* dmalloc(int64_t n) { return (T*)dev_alloc(sizeof(T) * (size_t)(n > 0 ? n : 1)); }

}  // namespace

static int64_t g_pool_limit = 0;
static int g_pool_chunk = -1;          // -1: chunks of 2048 entries on the large products, none on the small ones
void spgemm_set_pool_limit(int64_t entries) { g_pool_limit = entries > 0 ? entries : 0; }
void spgemm_set_pool_chunk(int entries) { g_pool_chunk = entries; }

void dcsr_free(DCsr& A) {
  if (A.rp) dev_free(A.rp);
  if (A.ci) dev_free(A.ci);
  if (A.v) dev_free(A.v);
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
// row there; once all counts are known a gather pass puts the rows in row order.  Nothing per row crosses the bus on
// the way: the bounds, the choice of accumulator and the counts stay on the device, the host reads a handful of
// totals.  Only when rows fail (LDS table too small -> global table; pool exhausted -> another round with a fresh
// pool) are the counts downloaded and the failed rows listed.
DCsr dev_spgemm(const DCsr& A, const DCsr& B, bool drop_enabled, double tol, bool lump, bool negate, hipStream_t s,
                SpgemmStats* stats) {
  if (A.n != B.m) throw std::runtime_error("dev_spgemm: shape mismatch");
  DevScope scope;      // every block below is a raw dev_alloc pointer: an exception on the way returns them all
  static const bool trace = std::getenv("AIRHIP_SPGEMM_TRACE") != nullptr;
  auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  const double t_begin = now();
  const int64_t m = A.m;
  DCsr C; C.m = m; C.n = B.n;
  C.rp = dmalloc<int64_t>(m + 1);
  if (m == 0) { int64_t z = 0; HIP_CHECK(hipMemcpy(C.rp, &z, sizeof(int64_t), hipMemcpyHostToDevice)); return C; }
  const CsrView av = cview(A), bv = cview(B);
  const DropParams dp{drop_enabled ? 1 : 0, tol, lump ? 1 : 0, negate ? 1 : 0};
  // ---- candidate upper bounds, and what they add up to
  int64_t* d_ub = dmalloc<int64_t>(m);
  UbStats* d_us = dmalloc<UbStats>(1);
  HIP_CHECK(hipMemsetAsync(d_us, 0, sizeof(UbStats), s));
  hipLaunchKernelGGL(k_row_ub, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, av, bv, d_ub, d_us);
  UbStats us;
  HIP_CHECK(hipMemcpyAsync(&us, d_us, sizeof(UbStats), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  dev_free(d_us);
  const int64_t ub_total = (int64_t)us.total, longest = (int64_t)us.longest;
  // B with at most one entry per row (A * P with a one-point prolongator): 64 entries of A per step (hash_accumulate)
  int b_single = 0;
  if (B.nnz <= B.m) {
    unsigned long long* d_lr = dmalloc<unsigned long long>(1);
    unsigned long long lr = 0;
    HIP_CHECK(hipMemsetAsync(d_lr, 0, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(k_longest_row, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, B.m, B.rp, d_lr);
    HIP_CHECK(hipMemcpyAsync(&lr, d_lr, sizeof(lr), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    dev_free(d_lr);
    b_single = lr <= 1 ? 1 : 0;
  }
  // ---- distinct columns per candidate on a sample of the long rows: which table each of them gets
  Plan plan{L1K_MAX, L2K_MAX, 1.0f};
  if (us.above_1k) {
    const int ns = (int)std::min<int64_t>(m, 1536);
    int64_t* d_sub = dmalloc<int64_t>(ns); int32_t* d_snu = dmalloc<int32_t>(ns);
    hipLaunchKernelGGL(k_spgemm_sample, dim3((unsigned)std::min(ns, 768)), dim3(WAVE), 0, s, av, bv, d_ub, ns, d_sub, d_snu, b_single);
    std::vector<int64_t> sub(ns); std::vector<int32_t> snu(ns);
    HIP_CHECK(hipMemcpyAsync(sub.data(), d_sub, sizeof(int64_t) * (size_t)ns, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(snu.data(), d_snu, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    dev_free(d_sub); dev_free(d_snu);
    std::vector<double> r;
    for (int w = 0; w < ns; w++) if (snu[w] != -2) r.push_back(snu[w] < 0 ? 1.0 : (double)snu[w] / (double)sub[w]);
    if (!r.empty()) {
      std::sort(r.begin(), r.end());
      const double r90 = std::max(r[std::min(r.size() - 1, r.size() * 9 / 10)], 1e-3);
      plan.ratio = (float)r90;
      plan.max1k = std::max<int64_t>(L1K_MAX, (int64_t)(L1K_FILL / r90));
      plan.max2k = std::max<int64_t>(L2K_MAX, (int64_t)(L2K_FILL / r90));
    }
  }
  const double t_ub = now() - t_begin;
  int64_t* d_cnt = dmalloc<int64_t>(m + 1);
  int64_t* d_pos = dmalloc<int64_t>(m);
  unsigned long long* d_cursor = dmalloc<unsigned long long>(1);
  unsigned int* d_fail = dmalloc<unsigned int>(4);       // [0] table overflow, [1] pool full, [2] / [3] rows moved up from 1k / 2k
  unsigned long long* d_seen = dmalloc<unsigned long long>(NKIND);
  int32_t* d_up1 = nullptr; int32_t* d_up2 = nullptr;    // rows that overflowed a 1k / 2k table
  if (us.above_1k) { d_up1 = dmalloc<int32_t>(m); d_up2 = dmalloc<int32_t>(m); }
  HIP_CHECK(hipMemsetAsync(d_seen, 0, NKIND * sizeof(unsigned long long), s));
  const int resident = 256 * 3;                          // workgroups of 48 KB LDS the chip holds
  auto grid_reg = [&](int64_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 256 * 32))); };
  auto grid_lds = [&](int64_t n, int wpb) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + wpb - 1) / wpb, 4 * resident))); };
  auto launch_lds = [&](int kind, const int32_t* rows, int64_t nrows, const RowPool& pool, Escalation esc) {
    const RowSel sel{rows, nrows, d_ub, kind, plan, b_single};
    const int64_t n = rows ? nrows : m;
    if (kind == KIND_L1K) hipLaunchKernelGGL((k_spgemm_lds<1024>), grid_lds(n, 4), dim3(4 * WAVE), 0, s, av, bv, sel, dp, d_cnt, pool, d_fail, esc, d_seen + kind);
    else if (kind == KIND_L2K) hipLaunchKernelGGL((k_spgemm_lds<2048>), grid_lds(n, 2), dim3(2 * WAVE), 0, s, av, bv, sel, dp, d_cnt, pool, d_fail, esc, d_seen + kind);
    else hipLaunchKernelGGL((k_spgemm_lds<4096>), grid_lds(n, 1), dim3(WAVE), 0, s, av, bv, sel, dp, d_cnt, pool, d_fail, esc, d_seen + kind);
  };
  auto read_u32 = [&](const unsigned int* d) {
    unsigned int v = 0;
    HIP_CHECK(hipMemcpyAsync(&v, d, sizeof(v), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    return v;
  };
  // all LDS kinds over their row selections (rows[k] == nullptr and all == true: the rows of A whose bound selects k),
  // longest rows first (they run at the lowest occupancy), then the rows that moved up
  auto run_tables = [&](const int32_t* const rows[NKIND], const int64_t nrows[NKIND], bool all, const RowPool& pool) {
    auto wanted = [&](int k) { return all ? (k == KIND_L1K || us.above_1k > 0) : rows[k] != nullptr; };
    const bool moving = d_up1 != nullptr;
    if (moving) HIP_CHECK(hipMemsetAsync(d_fail + 2, 0, 2 * sizeof(unsigned int), s));
    const Escalation none{nullptr, nullptr}, up1{d_up1, d_fail + 2}, up2{d_up2, d_fail + 3};
    if (wanted(KIND_L4K)) launch_lds(KIND_L4K, rows[KIND_L4K], nrows[KIND_L4K], pool, none);
    if (wanted(KIND_L2K)) launch_lds(KIND_L2K, rows[KIND_L2K], nrows[KIND_L2K], pool, moving ? up2 : none);
    if (wanted(KIND_L1K)) launch_lds(KIND_L1K, rows[KIND_L1K], nrows[KIND_L1K], pool, moving ? up1 : none);
    if (!moving) return;
    if (const unsigned int n1 = read_u32(d_fail + 2)) launch_lds(KIND_L2K, d_up1, n1, pool, up2);
    if (const unsigned int n2 = read_u32(d_fail + 3)) launch_lds(KIND_L4K, d_up2, n2, pool, none);
  };
  // rows with an explicit list (second and later attempts)
  struct List { std::vector<int32_t> rows; int32_t* d = nullptr; };
  auto upload_list = [&](List& l) {
    if (l.rows.empty()) return;
    l.d = dmalloc<int32_t>((int64_t)l.rows.size());
    HIP_CHECK(hipMemcpyAsync(l.d, l.rows.data(), sizeof(int32_t) * l.rows.size(), hipMemcpyHostToDevice, s));
  };
  std::vector<int64_t> ub_host;                         // only when rows fail
  auto need_ub_host = [&] {
    if (!ub_host.empty()) return;
    ub_host.resize(m);
    HIP_CHECK(hipMemcpyAsync(ub_host.data(), d_ub, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  };
  auto launch_global = [&](List& l, const RowPool& pool, std::vector<void*>& scratch) {
    need_ub_host();
    const size_t n = l.rows.size();
    std::vector<int64_t> toff(n); std::vector<int32_t> tsize(n);
    int64_t run = 0;
    for (size_t w = 0; w < n; w++) {
      const int64_t want = 2 * std::min<int64_t>(ub_host[l.rows[w]], B.n) + 64;
      int64_t cap = 8192; while (cap < want) cap <<= 1;
      if (cap > 0x40000000ll) throw std::runtime_error("dev_spgemm: row too long");
      toff[w] = run; tsize[w] = (int32_t)cap; run += cap;
    }
    int64_t* d_toff = dmalloc<int64_t>((int64_t)n); int32_t* d_tsize = dmalloc<int32_t>((int64_t)n);
    int32_t* d_kpool = dmalloc<int32_t>(run); double* d_vpool = dmalloc<double>(run);
    HIP_CHECK(hipMemcpyAsync(d_toff, toff.data(), sizeof(int64_t) * n, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_tsize, tsize.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));                  // toff / tsize are locals
    const unsigned gb = (unsigned)std::min<int64_t>(((int64_t)n + 3) / 4, 2048);
    hipLaunchKernelGGL(k_spgemm_glb, dim3(gb), dim3(BLOCK), 0, s, av, bv, l.d, (int64_t)n, d_toff, d_tsize, d_kpool, d_vpool, dp, d_cnt, pool, d_fail);
    scratch.insert(scratch.end(), {d_toff, d_tsize, d_kpool, d_vpool});
  };
  struct Round { int32_t* ci; double* v; bool all_rows; std::vector<int32_t> done; };
  std::vector<Round> rounds;
  List todo[NKIND];                                      // per accumulator kind; empty in the first round (= every row)
  int64_t n_global = 0;
  double t_kern = 0;
  std::vector<int64_t> cnt_host;
  for (bool first = true;; first = false) {
    // pool: twice the inputs (more when nothing will be dropped), never more than the sum of the bounds, never more than
    // a third of the free memory
    const size_t mem_free = dev_available_bytes();      // free memory + what the cache and the setup arena hold unused
    int64_t capacity = std::max<int64_t>(2 * (A.nnz + B.nnz) + (1 << 20), 2 * longest);
    // nothing is dropped: the rows keep all their distinct columns, which the sample has an estimate of
    if ((!drop_enabled || tol == 0.0) && plan.ratio < 1.0f) capacity = std::max<int64_t>(capacity, (int64_t)(1.1 * (double)plan.ratio * (double)ub_total));
    capacity = std::min<int64_t>(capacity, ub_total);
    capacity = std::min<int64_t>(capacity, (int64_t)(mem_free / 3 / 12));
    // chunked hand-out: every wave may leave most of one chunk unused (at most ~70k waves over the kernels of a round)
    // (only worth it on products with millions of rows)
    int chunk = g_pool_chunk >= 0 ? g_pool_chunk : 2048;
    const int64_t slack = (int64_t)chunk * 72 * 1024;
    if (!g_pool_limit && (g_pool_chunk >= 0 || m >= (1 << 20)) && (int64_t)(mem_free / 3 / 12) > capacity + slack) capacity += slack;
    else chunk = 0;
    if (g_pool_limit) capacity = std::max<int64_t>(longest, std::min<int64_t>(capacity, g_pool_limit));
    if (capacity < longest) throw std::runtime_error("dev_spgemm: not enough device memory for the output pool");
    // (tens of GB on the large levels: the block cache of device/devmem.hpp hands the previous product's pool back)
    Round rd; rd.all_rows = false;
    rd.ci = dmalloc<int32_t>(capacity); rd.v = dmalloc<double>(capacity);
    HIP_CHECK(hipMemsetAsync(d_cursor, 0, sizeof(unsigned long long), s));
    HIP_CHECK(hipMemsetAsync(d_fail, 0, 4 * sizeof(unsigned int), s));
    const RowPool pool{d_cursor, capacity, rd.ci, rd.v, d_pos, chunk};
    const double tk0 = now();
    std::vector<void*> scratch;
    const int32_t* lists[NKIND] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    int64_t counts[NKIND] = {0, 0, 0, 0, 0};
    if (!first) for (int k = 0; k < NKIND; k++) { upload_list(todo[k]); lists[k] = todo[k].d; counts[k] = (int64_t)todo[k].rows.size(); }
    if (first || lists[KIND_REG]) {
      const RowSel sel{lists[KIND_REG], counts[KIND_REG], d_ub, KIND_REG, plan, b_single};
      hipLaunchKernelGGL(k_spgemm_tiny, grid_reg(first ? m : counts[KIND_REG]), dim3(BLOCK), 0, s, av, bv, sel, dp, d_cnt, pool, d_fail, d_seen + KIND_REG);
    }
    run_tables(lists, counts, first, pool);
    if (lists[KIND_GLOBAL]) launch_global(todo[KIND_GLOBAL], pool, scratch);
    unsigned int fail[2];
    HIP_CHECK(hipMemcpyAsync(fail, d_fail, sizeof(fail), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (void* p : scratch) dev_free(p);
    t_kern += now() - tk0;
    const bool clean = fail[0] == 0 && fail[1] == 0;
    if (first && clean) { rd.all_rows = true; rounds.push_back(std::move(rd)); break; }
    // ---- some rows failed (or this is a later round): sort the rows of this round by outcome
    cnt_host.resize(m);
    HIP_CHECK(hipMemcpyAsync(cnt_host.data(), d_cnt, sizeof(int64_t) * (size_t)m, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    List next[NKIND];
    auto outcome = [&](int32_t i, int kind) {
      const int64_t c = cnt_host[i];
      if (c >= 0) rd.done.push_back(i);
      else if (c == -ROW_TABLE_OVERFLOW) {
        if (kind == KIND_GLOBAL) throw std::runtime_error("dev_spgemm: global hash table overflow");
        next[KIND_GLOBAL].rows.push_back(i); n_global++;
      } else next[kind].rows.push_back(i);               // no place in the pool: same kind again (it moves up again if it must)
    };
    if (first) {
      need_ub_host();
      std::vector<int64_t> arp(m + 1);
      HIP_CHECK(hipMemcpyAsync(arp.data(), A.rp, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyDeviceToHost, s));
      HIP_CHECK(hipStreamSynchronize(s));
      for (int64_t i = 0; i < m; i++) outcome((int32_t)i, kind_of(ub_host[i], arp[i + 1] - arp[i], plan));
    } else {
      for (int k = 0; k < NKIND; k++) for (int32_t i : todo[k].rows) outcome(i, k);
    }
    size_t left = 0;
    for (int k = 0; k < NKIND; k++) { if (todo[k].d) dev_free(todo[k].d); left += next[k].rows.size(); }
    const bool progress = !rd.done.empty() || !next[KIND_GLOBAL].rows.empty();
    rounds.push_back(std::move(rd));
    if (!left) break;
    if (!progress) throw std::runtime_error("dev_spgemm: no progress (output pool too small)");
    for (int k = 0; k < NKIND; k++) { todo[k].rows.swap(next[k].rows); todo[k].d = nullptr; }
  }
  // ---- row pointers (device scan of the counts), then the rows of every round into row order
  const double tg0 = now();
  C.nnz = device_exclusive_scan(m, d_cnt, C.rp, s);
  HIP_CHECK(hipMemcpyAsync(C.rp + m, &C.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  C.ci = dmalloc<int32_t>(C.nnz); C.v = dmalloc<double>(C.nnz);
  for (Round& rd : rounds) {
    const int64_t nr = rd.all_rows ? m : (int64_t)rd.done.size();
    if (nr) {
      int32_t* d_rows = nullptr;
      if (!rd.all_rows) {
        d_rows = dmalloc<int32_t>(nr);
        HIP_CHECK(hipMemcpyAsync(d_rows, rd.done.data(), sizeof(int32_t) * (size_t)nr, hipMemcpyHostToDevice, s));
      }
      const unsigned g = (unsigned)std::max<int64_t>(1, std::min<int64_t>((nr + 3) / 4, 8192));
      hipLaunchKernelGGL(k_gather_rows, dim3(g), dim3(BLOCK), 0, s, nr, d_rows, d_pos, C.rp, rd.ci, rd.v, C.ci, C.v);
      HIP_CHECK(hipStreamSynchronize(s));
      if (d_rows) dev_free(d_rows);
    }
    dev_free(rd.ci); dev_free(rd.v);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e) + " in the device SpGEMM");
  const double t_gather = now() - tg0;
  unsigned long long seen[NKIND];
  HIP_CHECK(hipMemcpy(seen, d_seen, sizeof(seen), hipMemcpyDeviceToHost));
  dev_free(d_ub); dev_free(d_cnt); dev_free(d_pos); dev_free(d_cursor); dev_free(d_fail); dev_free(d_seen);
  if (d_up1) { dev_free(d_up1); dev_free(d_up2); }
  if (trace) {
    std::printf("   [spgemm] m=%lld nnzA=%lld nnzB=%lld bound=%lld nnzC=%lld rows reg/1k/2k/4k/global=%llu/%llu/%llu/%llu/%lld ratio=%.3f rounds=%zu  plan %.3f kernels %.3f gather %.3f total %.3f s\n",
                (long long)m, (long long)A.nnz, (long long)B.nnz, (long long)ub_total, (long long)C.nnz, seen[0], seen[1], seen[2], seen[3],
                (long long)n_global, (double)plan.ratio, rounds.size(), t_ub, t_kern, t_gather, now() - t_begin);
    std::fflush(stdout);
  }
  if (stats) {
    stats->rows_reg = (int64_t)seen[KIND_REG]; stats->rows_1k = (int64_t)seen[KIND_L1K]; stats->rows_2k = (int64_t)seen[KIND_L2K];
    stats->rows_4k = (int64_t)seen[KIND_L4K]; stats->rows_global = n_global;
  }
  return C;
}

}  // namespace airhip
