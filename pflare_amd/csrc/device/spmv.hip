// Row-block CSR SpMV kernels (the dominant kernels of the AIR V-cycle) and the device-side
// CSR -> row-block CSR conversion.
//
// Format and mapping (gfx950, wave64):
//   * the matrix stays in CSR row order (no padding; SELL-64 padded the AIR operators by 1.75x, and
//     length-sorted SELL-C-sigma destroyed the locality of the x gathers);
//   * rows are partitioned into blocks of <= BLK_NNZ (512) non-zeros and <= BLK_ROWS (128) rows; a row
//     longer than that is a block of its own.  A block's index region has one of three forms (dev.hpp: BlkDesc).
//     MULTI-BASE (round 4, the default wherever it fits): entries in row-major order, one 16-BIT word each,
//     window << ob | offset, in front of them a table of up to 64 column windows (byte offset of a window's first
//     column) - 10 B per non-zero; also for rows longer than a block.  RAW: row-major order, 32-bit absolute
//     columns (blocks that fit no table).  PACKED (rounds 2-3, only with the multi-base form switched off): entries
//     SORTED BY COLUMN, 32-bit word ((column - cbase) * 8) << 10 | (row-major position + 1).  Measured at 464^3: the
//     column order of the packed form buys blocks of many short rows less than its 4-byte words cost them, and a
//     block of few long rows is in column order anyway (profiles/r04_mb16_rows.txt);
//   * the multi-base and raw blocks run in LEAN kernels of their own (k_rb_spmv, k_rb_fc_first: 57-76 VGPRs), the packed
//     ones in the *_g kernels: the block code waits on memory three times in a row, so what a CU streams is resident
//     waves x one block per wave lifetime, and the widest path of one kernel for all kinds had set the budget at 5
//     waves per SIMD.  The lean kernels are launched with enough dynamic LDS that 6 (products) / 5 (fused first F stage)
//     workgroups fit a CU - measured optima; 8 is slower again (profiles/r04_occupancy.txt);
//   * one WAVE owns exactly one block: the grid has as many waves as the matrix has blocks, and the hardware
//     dispatches workgroups in order, so the waves resident on an XCD at any moment cover one contiguous window
//     of rows - the matrix stream is read front to back and the x lines a window gathers stay in that XCD's L2.
//     (Measured at 256^3: a resident-size grid whose waves loop over their blocks, with or without a software
//     prefetch of the next block's stream, 3.9 TB/s; one block per wave 4.25 TB/s - the looping waves drift
//     apart and the window dissolves.)  Round 1 measured one block every ~7 us per wave = seven dependent memory
//     round trips with SQ_WAIT_ANY at 82 %; here the descriptor is a scalar load, every vector the block needs
//     (index words, values, row offsets, epilogue inputs) is requested before the first wait, and the gather is
//     the only second round trip;
//   * a block's 512 logical slots are 8 ROUNDS of 64 (slot 64 u + lane = register u of that lane).  32-bit index words are
//     stored in slot order (8 coalesced dword loads), 16-bit ones interleaved like the values.  VALUES are stored with every pair of full rounds (2i, 2i+1) interleaved -
//     memory position 128 i + 2 lane + t holds slot 64 (2i + t) + lane - so that one 16-byte load per lane fetches both
//     rounds: a dwordx4 load costs the address path what a dwordx2 load does (tools/ta_probe.hip), and the value stream is
//     the largest load of a block (tools/block_probe.hip: +5 % on a gather-bound block kernel).  Rounds that a block does
//     not have are not issued at all (a 425-entry row of the coarse operator issues 7 rounds, not 8);
//   * every per-block array is read through a buffer resource with the block's extent: the hardware range check
//     returns 0 past the end (and drops stores), which replaces clamps, exec-mask branches and 64-bit addresses;
//   * phase 1: products to LDS at their row-major positions (4 KB per wave); phase 2: lane l sums the LDS
//     segment of rows r0+l and r0+l+64 in column order (the summation order of a sequential CSR loop), or
//     T lanes share a row when the block holds <= 32 rows; a multi-base / raw block of ONE row is summed in
//     registers (same products, same order) and never touches LDS; the epilogue vectors are loaded before the
//     gather is waited for;
//   * XCD-aware assignment: blocks b and b+8 of the grid share an XCD (observed round-robin
//     dispatch; a speed assumption only), so each residue class of workgroups walks a contiguous
//     range of row blocks and an XCD's private L2 sees a contiguous window of x.
#include "dev.hpp"
#include "kcommon.hpp"
#include "rb.hpp"
#include "devmem.hpp"
#include <stdexcept>
#include <type_traits>
#include <vector>
#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace airhip {

// waves (= row blocks) per workgroup of the SpMV kernels
#ifndef AIRHIP_SPMV_WAVES
#define AIRHIP_SPMV_WAVES 4
#endif
constexpr int SPMV_WAVES = AIRHIP_SPMV_WAVES;
constexpr int SPMV_BLOCK = SPMV_WAVES * WAVE;

// XCD-aware assignment of row blocks to waves (gridDim.x is a multiple of 8): workgroup w of residue class
// g = blockIdx & 7 (one XCD under round-robin dispatch; a speed assumption only) takes SPMV_WAVES consecutive
// blocks of the g-th run of per_class = ceil(nblocks / 8) blocks.  Returns -1 when the wave has no block.
// 32-bit scalar arithmetic: round 2's 64-bit nblocks * g / 8 was 25 scalar instructions in front of the first load.
__device__ __forceinline__ int my_block(const RbView& A) {
  const int g = blockIdx.x & 7;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int k = (int)(blockIdx.x >> 3) * SPMV_WAVES + wid;          // position inside the class
  const int b = g * A.per_class + k;
  return (k < A.per_class && b < A.nblocks) ? b : -1;
}

// ---- tuning knobs (airhip_set_tuning): which blocks the conversion stores in the 16-bit multi-base form (dev.hpp).
//   mb16          0: none (the round-3 format: packed or raw), 1 (default): see mb16_max_rows
//   mb16_max_rows blocks of at most this many rows go multi-base (default 128 = every block that can), blocks whose
//                 columns span 2^19 or more (raw otherwise) always do.  Measured at 464^3 (profiles/r04_mb16_rows.txt):
//                 2 rows 20.7, 8 rows 21.3, 128 rows 21.5 iterations/s - the column order of the packed form buys the
//                 short-row blocks less than its 4-byte words cost them.
// They act when an operator is converted (upload of a hierarchy), not on operators that exist.
//   occ_spmv / occ_fc  workgroups (of 4 waves) of the lean k_rb_spmv / k_rb_fc_first kernels that a CU may hold = waves per
//                 SIMD (0: as many as the registers allow, 8).  Takes effect at the next launch: the launch asks for enough
//                 dynamic LDS that no more fit.  More resident waves stream more, but they also share the L2 and the
//                 address path: a kernel whose blocks gather many distinct lines (A_fc: columns in every section of the
//                 coarse vector) loses what a streaming-bound one gains.
static int g_mb16 = 1, g_mb16_max_rows = 128;
//   vdict         1 (default): multi-base blocks whose values take at most 256 distinct bit patterns store a dictionary and
//                 one-byte indices instead of the values (dev.hpp); 0: plain values everywhere.  Lossless either way.
static int g_vdict = 1;
static int g_occ_spmv = 6, g_occ_fc = 6;
int spmv_set_tuning(const char* key, int value) {
  const std::string k(key);
  if (k == "mb16") { g_mb16 = value != 0; return 0; }
  if (k == "mb16_max_rows") { g_mb16_max_rows = value < 0 ? 0 : value; return 0; }
  if (k == "vdict") { g_vdict = value != 0; return 0; }
  if (k == "occ_spmv") { g_occ_spmv = value < 0 ? 0 : value; return 0; }
  if (k == "occ_fc") { g_occ_fc = value < 0 ? 0 : value; return 0; }
  return -1;
}
// dynamic LDS that limits a kernel with LDS_STATIC bytes of its own to `wgs` workgroups per CU (160 KB of LDS, allocated
// in units of 1 KB here so that the rounding of the hardware's granule cannot cost a workgroup)
static unsigned occ_pad(int wgs) {
  // the LDS of one CU from the device (160 KB on gfx950, the only target; a part with 64 KB would otherwise be held to two
  // workgroups per CU by a padding computed for 160)
  static const int LDS_CU = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev) != hipSuccess || v <= 0) { (void)hipGetLastError(); v = 160 * 1024; }
    return v;
  }();
  constexpr int LDS_STATIC = SPMV_WAVES * LDS_SLOTS * 8;
  if (wgs <= 0 || wgs >= LDS_CU / LDS_STATIC) return 0;
  if (wgs < 3) wgs = 3;      // (a workgroup may not ask for more than 64 KB)
  const int alloc = LDS_CU / wgs / 1024 * 1024;
  return alloc > LDS_STATIC ? (unsigned)(alloc - LDS_STATIC) : 0;
}

struct BlkInfo { int64_t p0, ip; int32_t r0, cbase, nb, nr; };

// The block descriptor is read through the constant address space: the address is wave-uniform, so the compiler
// emits one scalar load (s_load_dwordx8, counted by lgkmcnt) instead of vector loads whose wait (vmcnt is in order)
// would drain the prefetched stream.  The arrays are never written by these kernels.
typedef const BlkDesc __attribute__((address_space(4))) * BlkDescConstPtr;

__device__ __forceinline__ BlkInfo load_blk(const RbView& A, int b) {
  BlkDescConstPtr bp = (BlkDescConstPtr)(uintptr_t)A.blk;
  BlkInfo o;
  o.p0 = bp[b].p0;
  o.r0 = bp[b].row;
  o.cbase = bp[b].cbase;
  o.nb = bp[b].nb;
  o.nr = bp[b].nr;
  o.ip = bp[b].ip;
  return o;
}

// ---- buffer addressing.  Every per-block array (index words, values, row offsets, epilogue vectors) is read
// through a buffer resource whose extent is the block's: lanes past the end get 0 from the hardware range check
// (and stores past the end are dropped), so the kernel needs no clamps, no exec-mask branches and one 32-bit
// offset register per array.  An out-of-range index word 0 decodes to the block's first column (packed: position 0 =
// the dummy LDS slot; multi-base: window 0, offset 0), so padding lanes gather a valid address of the block's own and
// their zero value makes a zero product.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, int64_t bytes) {
  const int nrec = bytes > 0x7fffffffll ? 0x7fffffff : (int)bytes;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, nrec, 0x00020000);
}
// the same for an extent known to fit 31 bits (everything per block does): no 64-bit clamp
__device__ __forceinline__ rsrc_t make_rsrc32(const void* base, int bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, bytes, 0x00020000);
}
__device__ __forceinline__ double buf_f64(rsrc_t r, int voff, int soff = 0) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
// Cache policy of the loads of the matrix stream (values, index words, dictionaries: bytes one wave reads once per product) -
// the `aux` operand of the buffer loads: 0 = default, 2 = nt (non-temporal: the line is the first to leave the caches, so the
// stream does not push the x lines the gathers share out of the XCD's L2).  The gathers and the vectors keep the default.
#ifndef AIRHIP_STREAM_AUX
#define AIRHIP_STREAM_AUX 0
#endif
constexpr int STREAM_AUX = AIRHIP_STREAM_AUX;
__device__ __forceinline__ double stream_f64(rsrc_t r, int voff, int soff = 0) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, STREAM_AUX));
}
template <typename T> __device__ __forceinline__ T stream_ld(const T* p) {
  if constexpr (STREAM_AUX == 2) return __builtin_nontemporal_load(p);
  else return *p;
}
__device__ __forceinline__ void buf_store_f64(rsrc_t r, int voff, double v) {
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(__attribute__((__vector_size__(2 * sizeof(unsigned int)))) unsigned int, v), r, voff, 0, 0);
}

// Rounds of a block (see the header): NF full rounds, then one more that is partial (or, when the block's size is a
// multiple of 64, empty: the hardware range check then returns zeros and the lanes park a zero product in the dummy
// slot).  NF is a compile-time constant - one straight-line variant of the block code per value (rb_block dispatches) -
// because wave-uniform branches around the single loads cost more than they save (256^3: restriction -10 %).
template <int NF> struct Rounds {
  static constexpr int nu = NF < EPL ? NF + 1 : EPL;      // rounds issued
  static constexpr int npair = NF >> 1;                   // interleaved pairs of full rounds
  __device__ __forceinline__ static constexpr bool has(int u) { return u < nu; }
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// the block's first 512 values into registers, pair by pair (16-byte loads on the interleaved placement)
template <int NF>
__device__ __forceinline__ void issue_values(const RbView& A, const BlkInfo& d, int n, int lane, double (&vv)[EPL]) {
  typedef Rounds<NF> R;
  const rsrc_t rv = make_rsrc32(A.val + d.p0, n * 8);
#pragma unroll
  for (int i = 0; i < EPL / 2; i++) {
    if (i < R::npair) {
      const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(rv, lane * 16, i * 2 * WAVE * 8, STREAM_AUX);
      vv[2 * i] = __builtin_bit_cast(double, (unsigned long long)q.x | ((unsigned long long)q.y << 32));
      vv[2 * i + 1] = __builtin_bit_cast(double, (unsigned long long)q.z | ((unsigned long long)q.w << 32));
    } else {
      vv[2 * i] = 0.0; vv[2 * i + 1] = 0.0;
      if (R::has(2 * i)) vv[2 * i] = stream_f64(rv, lane * 8, 2 * i * WAVE * 8);
      if (R::has(2 * i + 1)) vv[2 * i + 1] = stream_f64(rv, lane * 8, (2 * i + 1) * WAVE * 8);
    }
  }
}

// 32-bit index words (packed and raw blocks), round by round, then the values
template <int NF>
__device__ __forceinline__ void issue_stream(const RbView& A, const BlkInfo& d, int lane, uint32_t (&ix)[EPL], double (&vv)[EPL]) {
  typedef Rounds<NF> R;
  const int n = d.nb < BLK_NNZ ? d.nb : BLK_NNZ;
  const rsrc_t ri = make_rsrc32(A.idx + d.ip, n * 4);
#pragma unroll
  for (int u = 0; u < EPL; u++) { ix[u] = 0; if (R::has(u)) ix[u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ri, lane * 4, u * WAVE * 4, STREAM_AUX); }
  issue_values<NF>(A, d, n, lane, vv);
}

// 16-bit index words of a multi-base block: the words of every pair of full rounds are interleaved like the values (one
// dword per lane holds slot 64 (2i) + lane in its low half and slot 64 (2i + 1) + lane in its high half), the rest is in
// slot order; in front of them the table of windows, one per lane.  Then the values.  iw[i] keeps the two words of
// rounds 2i / 2i + 1 packed (mb_word extracts them).
template <int NF>
__device__ __forceinline__ void issue_index_mb(const RbView& A, const BlkInfo& d, int nwin, int lane, uint32_t& tab, uint32_t (&iw)[EPL / 2]) {
  typedef Rounds<NF> R;
  const uint8_t* __restrict__ ib = A.idx + d.ip;
  tab = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc32(ib, nwin * 4), lane * 4, 0, STREAM_AUX);
  const rsrc_t ri = make_rsrc32(ib + nwin * 4, d.nb * 2);
#pragma unroll
  for (int i = 0; i < EPL / 2; i++) {
    iw[i] = 0;
    if (i < R::npair) iw[i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(ri, lane * 4, i * 2 * WAVE * 2, STREAM_AUX);
    else {
      uint32_t lo = 0, hi = 0;
      if (R::has(2 * i)) lo = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(ri, lane * 2, 2 * i * WAVE * 2, STREAM_AUX);
      if (R::has(2 * i + 1)) hi = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(ri, lane * 2, (2 * i + 1) * WAVE * 2, STREAM_AUX);
      if (R::has(2 * i)) iw[i] = lo | (hi << 16);
    }
  }
}
template <int NF>
__device__ __forceinline__ void issue_stream_mb(const RbView& A, const BlkInfo& d, int nwin, int lane, uint32_t& tab, uint32_t (&iw)[EPL / 2], double (&vv)[EPL]) {
  issue_index_mb<NF>(A, d, nwin, lane, tab, iw);
  issue_values<NF>(A, d, d.nb, lane, vv);
}
__device__ __forceinline__ void wave_lds_sync();
// a block with a value dictionary (dev.hpp): its nd <= 256 distinct values (lane l takes entries l, l + 64, l + 128, l + 192;
// past the end the range check returns zero) and this lane's eight one-byte indices
__device__ __forceinline__ void issue_dictionary(const RbView& A, const BlkInfo& d, int nd, int lane, double (&tq)[4], unsigned long long& vi) {
  const rsrc_t rt = make_rsrc32(A.val + d.p0, nd * 8);
#pragma unroll
  for (int k = 0; k < 4; k++) tq[k] = stream_f64(rt, lane * 8, k * WAVE * 8);
  const rsrc_t rv = make_rsrc32(A.val + d.p0 + nd, WAVE * 8);
  vi = __builtin_bit_cast(unsigned long long, stream_f64(rv, lane * 8));
}
// ... into the wave's staging area, and from there the values of this lane's slots (entries past the block's end: zero, as
// the range check gives for plain values).  The staging area is free until the products are parked in it.
template <int NF>
__device__ __forceinline__ void values_from_dictionary(const BlkInfo& d, double* __restrict__ lds, int lane, const double (&tq)[4], unsigned long long vi, double (&vv)[EPL]) {
  typedef Rounds<NF> R;
#pragma unroll
  for (int k = 0; k < 4; k++) lds[lane + k * WAVE] = tq[k];
  wave_lds_sync();
#pragma unroll
  for (int u = 0; u < EPL; u++) {
    vv[u] = 0.0;
    if (R::has(u)) {
      const double t = lds[(int)((vi >> (8 * u)) & 255ull)];
      vv[u] = (u < NF || lane + u * WAVE < d.nb) ? t : 0.0;      // (u < NF: a full round, decided at compile time)
    }
  }
  wave_lds_sync();
}
template <int N>
__device__ __forceinline__ uint32_t mb_word(const uint32_t (&iw)[N], int u) { return (u & 1) ? (iw[u >> 1] >> 16) : (iw[u >> 1] & 0xffffu); }

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// sum of lds[j + 1] for j = a, a + step, ... < e, in that order; every lane of the wave takes part (a == e: no entry).
// Four entries per trip and a wave-uniform trip count: lanes whose row is shorter than the block's longest keep reading
// (clamped to the last slot of the staging area) and add zero - a per-lane loop cost six scalar instructions per entry
// of the longest row
__device__ __forceinline__ double row_sum(const double* __restrict__ lds, int a, int e, int step) {
  double s = 0.0;
  int j = a;
  while (__any(j < e)) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double v = lds[(j < BLK_NNZ ? j : BLK_NNZ) + 1];
      s += (j < e) ? v : 0.0;
      j += step;
    }
  }
  return s;
}

// row offsets of this lane's (at most two) rows: ra / re = first / one past the last entry, relative to the block
__device__ __forceinline__ void load_row_ranges(const RbView& A, const BlkInfo& d, int lane, bool two, int (&ra)[2], int (&re)[2]) {
  const rsrc_t rr = make_rsrc32(A.roff + d.r0, d.nr * 2);
#pragma unroll
  for (int q = 0; q < 2; q++) {
    ra[q] = 0; re[q] = 0;
    if (q == 0 || two) {
      const int r = lane + q * WAVE;
      ra[q] = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rr, r * 2, 0, 0);
      const int rnext = (int)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rr, r * 2 + 2, 0, 0);
      re[q] = (r + 1 < d.nr) ? rnext : (r < d.nr ? d.nb : 0);       // no row: ra = re = 0, nothing to sum
    }
  }
}

// phase 2 of a block whose products are staged in LDS at their row-major positions: the row sums, then the epilogue.
// pre[1] (the epilogue inputs of a lane's second row, blocks of more than 64 rows) may have been loaded by the caller
// (late2 = false) or is loaded here, after the first row has been stored (late2 = true: the lean multi-base kernel does not
// keep a second set of epilogue registers alive across its gather)
template <bool LATE2, class Epi>
__device__ __forceinline__ void sum_rows(const BlkInfo& d, const double* __restrict__ lds, const Epi& epi, int lane, bool two,
                                         const int (&ra)[2], const int (&re)[2], typename Epi::Pre (&pre)[2]) {
  const int nr = d.nr;
  if (nr > 32) {
    // one lane per row, column order
    epi.store(d.r0, nr, lane, row_sum(lds, ra[0], re[0], 1), pre[0]);
    if (two) {
      if constexpr (LATE2) pre[1] = epi.load(d.r0, nr, lane + WAVE, epi.early(d.r0, nr, lane + WAVE));
      epi.store(d.r0, nr, lane + WAVE, row_sum(lds, ra[1], re[1], 1), pre[1]);
    }
  } else {
    // few (longer) rows: T = 64 / 2^ceil(log2 nr) lanes share a row
    int sh = 6;
    while ((WAVE >> sh) < nr) { sh--; }       // 2^(6-sh) >= nr groups
    const int T = 1 << sh;
    const int g = lane >> sh, t = lane & (T - 1);
    const int a = __shfl(ra[0], g, WAVE), e = __shfl(re[0], g, WAVE);
    double s = row_sum(lds, a + t, e, T);
    for (int off = T >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    const double srow = __shfl(s, (lane << sh) & (WAVE - 1), WAVE);   // row `lane` lives in lanes lane*T ..
    epi.store(d.r0, nr, lane, srow, pre[0]);
  }
}

// packed and raw blocks (32-bit index words)
template <int NF, bool DIST, class Epi>
__device__ __forceinline__ void rb_body(const RbView& A, const BlkInfo& d, const XVec<DIST> x, double* __restrict__ lds, const Epi& epi, int lane) {
  const int nb = d.nb, nr = d.nr;
  const bool sorted = d.cbase >= 0;
  typedef Rounds<NF> R;
  uint32_t ix[EPL]; double vv[EPL];
  issue_stream<NF>(A, d, lane, ix, vv);
  // row offsets and epilogue inputs of this lane's (at most two) rows
  typename Epi::Early early[2];
  int ra[2], re[2];
  typename Epi::Pre pre[2];
  const bool two = nr > WAVE;          // blocks of <= 64 rows (most of them) issue nothing for a second row per lane
  early[0] = epi.early(d.r0, nr, lane);
  early[1] = early[0];
  if (two) early[1] = epi.early(d.r0, nr, lane + WAVE);
  load_row_ranges(A, d, lane, two, ra, re);
  // gather x for the rounds the block has
  double xx[EPL];
  if (sorted) {
    // packed block: the word's upper 22 bits are the byte offset of the column from the block's base column
    const XVec<DIST> xb = x.shifted(d.cbase);
#pragma unroll
    for (int u = 0; u < EPL; u++) { xx[u] = 0.0; if (R::has(u)) xx[u] = xb.at_byte_offset(ix[u] >> POS_BITS); }
  } else {
#pragma unroll
    for (int u = 0; u < EPL; u++) { xx[u] = 0.0; if (R::has(u)) xx[u] = x.at_byte_offset(ix[u] << 3); }           // raw block: absolute columns
  }
  pre[0] = epi.load(d.r0, nr, lane, early[0]);
  pre[1] = pre[0];
  if (two) pre[1] = epi.load(d.r0, nr, lane + WAVE, early[1]);
  if (nb <= BLK_NNZ) {
    // phase 1: products to their row-major positions (1-based; position 0 is the dummy slot of padding lanes)
    if (sorted) {
#pragma unroll
      for (int u = 0; u < EPL; u++) if (R::has(u)) lds[ix[u] & POS_MASK] = vv[u] * xx[u];
    } else {
#pragma unroll
      for (int u = 0; u < EPL; u++) if (R::has(u)) { const int k = lane + u * WAVE; lds[k < nb ? k + 1 : 0] = vv[u] * xx[u]; }
    }
    wave_lds_sync();
    sum_rows<false>(d, lds, epi, lane, two, ra, re, pre);
  } else {
    // one long row: the whole wave reduces it.  The first 512 entries are already in registers (stream and
    // gather above: all eight rounds exist); the rest is read with four independent chains per lane
    const uint32_t* __restrict__ ip = reinterpret_cast<const uint32_t*>(A.idx + d.ip);
    const double* __restrict__ vp = A.val + d.p0;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
    for (int u = 0; u < EPL; u += 4) {
      s0 = fma(vv[u], xx[u], s0); s1 = fma(vv[u + 1], xx[u + 1], s1);
      s2 = fma(vv[u + 2], xx[u + 2], s2); s3 = fma(vv[u + 3], xx[u + 3], s3);
    }
    int j = BLK_NNZ + lane;
    for (; j + 3 * WAVE < nb; j += 4 * WAVE) {
      const uint32_t c0 = stream_ld(ip + j), c1 = stream_ld(ip + j + WAVE), c2 = stream_ld(ip + j + 2 * WAVE), c3 = stream_ld(ip + j + 3 * WAVE);
      const double v0 = stream_ld(vp + j), v1 = stream_ld(vp + j + WAVE), v2 = stream_ld(vp + j + 2 * WAVE), v3 = stream_ld(vp + j + 3 * WAVE);
      const double x0 = x.at_byte_offset(c0 << 3), x1 = x.at_byte_offset(c1 << 3), x2 = x.at_byte_offset(c2 << 3), x3 = x.at_byte_offset(c3 << 3);
      s0 = fma(v0, x0, s0); s1 = fma(v1, x1, s1); s2 = fma(v2, x2, s2); s3 = fma(v3, x3, s3);
    }
    for (; j < nb; j += WAVE) s0 = fma(stream_ld(vp + j), x.at_byte_offset(stream_ld(ip + j) << 3), s0);
    double s = (s0 + s1) + (s2 + s3);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    epi.store(d.r0, nr, lane, s, pre[0]);      // nr = 1: only lane 0 is inside the block
  }
}

// multi-base blocks (16-bit index words, row-major order; at most 512 entries).  The lean path: it runs in kernels of its
// own (k_rb_spmv / k_rb_fc_first; the packed, raw and long-row blocks go through the *_g kernels) so that its register
// budget - 16 for the values, 16 for the gathered x, 4 for the index words, one set of epilogue inputs - and not that of
// the widest path decides how many waves a SIMD holds: the block code waits on memory three times in a row, and what it
// streams per microsecond is the number of resident waves times a block.
// RAW: the same path for a raw block (32-bit absolute columns; the few blocks that fit no table, e.g. a block of one entry or
// a long row whose columns lie in more than 64 windows) - so that a launch of the *_g kernels is only needed for packed
// blocks, i.e. when the multi-base form is switched off
// (The epilogue inputs of a lane's second row - blocks of more than 64 rows - are loaded only after the first row is stored:
// loading them with the first row's was measured on the levels 1-3, whose blocks all have 128 rows: no difference.)
// DICT: the block's values come out of its dictionary (dev.hpp; never a raw block, never a row longer than a block)
template <int NF, bool RAW, bool DIST, bool DICT, class Epi>
__device__ __forceinline__ void rb_body_mb(const RbView& A, const BlkInfo& d, const XVec<DIST> x, double* __restrict__ lds, const Epi& epi, int lane) {
  const int nr = d.nr;
  typedef Rounds<NF> R;
  const int meta = RAW ? 0 : -(d.cbase + 2);
  const int ob = meta & MB_OB_MASK, nwin = (meta >> MB_WIN_SHIFT) & MB_WIN_MASK;
  uint32_t tab = 0; uint32_t iw[RAW ? EPL : EPL / 2]; double vv[EPL];
  double tq[4]; unsigned long long vi = 0;
  if constexpr (RAW) issue_stream<NF>(A, d, lane, iw, vv);
  else if constexpr (DICT) { issue_index_mb<NF>(A, d, nwin, lane, tab, iw); issue_dictionary(A, d, meta >> MB_ND_SHIFT, lane, tq, vi); }
  else issue_stream_mb<NF>(A, d, nwin, lane, tab, iw, vv);
  int ra[2], re[2];
  typename Epi::Pre pre[2];
  const bool two = nr > WAVE;
  const bool one = nr == 1;            // the common case on the deep levels: no staging, no row offsets
  const typename Epi::Early early0 = epi.early(d.r0, nr, lane);
  ra[0] = ra[1] = re[0] = re[1] = 0;
  if (!one) load_row_ranges(A, d, lane, two, ra, re);
  // column = window's first column + offset: the table lives in one register across the lanes (ds_bpermute)
  const uint32_t omask = (1u << ob) - 1u;
  double xx[EPL];
#pragma unroll
  for (int u = 0; u < EPL; u++) {
    xx[u] = 0.0;
    if (R::has(u)) {
      if constexpr (RAW) xx[u] = x.at_byte_offset(iw[u] << 3);
      else {
        const uint32_t w = mb_word(iw, u);
        const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((w >> ob) << 2), (int)tab);
        xx[u] = x.at_byte_offset(base + ((w & omask) << 3));
      }
    }
  }
  pre[0] = epi.load(d.r0, nr, lane, early0);
  if constexpr (DICT) values_from_dictionary<NF>(d, lds, lane, tq, vi, vv);      // (while the gathers are in flight)
  if constexpr (NF == EPL && !DICT) {
    if (d.nb > BLK_NNZ) {
      // one long row: the whole wave reduces it, with the arithmetic of the general kernels' long-row path (rb_body).  The
      // first 512 entries are in registers; the words (16-bit, or 32-bit absolute columns of a raw row) and values of the
      // rest lie in plain order behind them
      typedef typename std::conditional<RAW, uint32_t, unsigned short>::type word_t;
      const word_t* __restrict__ wp = reinterpret_cast<const word_t*>(A.idx + d.ip + nwin * 4);
      const double* __restrict__ vp = A.val + d.p0;
      const int nb = d.nb;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
      for (int u = 0; u < EPL; u += 4) {
        s0 = fma(vv[u], xx[u], s0); s1 = fma(vv[u + 1], xx[u + 1], s1);
        s2 = fma(vv[u + 2], xx[u + 2], s2); s3 = fma(vv[u + 3], xx[u + 3], s3);
      }
      auto xat = [&](uint32_t w) {
        if constexpr (RAW) return x.at_byte_offset(w << 3);
        else {
          const uint32_t base = (uint32_t)__builtin_amdgcn_ds_bpermute((int)((w >> ob) << 2), (int)tab);
          return x.at_byte_offset(base + ((w & omask) << 3));
        }
      };
      // Wave-uniform trip counts with per-lane predicates: ds_bpermute only sees the table entries of ACTIVE lanes, so no
      // lane may leave these loops before the last one does.  Per lane the entries go to the same chains as in rb_body's
      // per-lane loops (four at a time while four remain for this lane, the rest to chain 0).
      int j = BLK_NNZ + lane;
      while (__any(j + 3 * WAVE < nb)) {
        const bool p = j + 3 * WAVE < nb;
        const int jj = p ? j : lane;         // (lanes that are done read positions inside the first 512: always the block's own)
        const uint32_t c0 = stream_ld(wp + jj), c1 = stream_ld(wp + jj + WAVE), c2 = stream_ld(wp + jj + 2 * WAVE), c3 = stream_ld(wp + jj + 3 * WAVE);
        const double v0 = stream_ld(vp + jj), v1 = stream_ld(vp + jj + WAVE), v2 = stream_ld(vp + jj + 2 * WAVE), v3 = stream_ld(vp + jj + 3 * WAVE);
        const double x0 = xat(c0), x1 = xat(c1), x2 = xat(c2), x3 = xat(c3);
        if (p) { s0 = fma(v0, x0, s0); s1 = fma(v1, x1, s1); s2 = fma(v2, x2, s2); s3 = fma(v3, x3, s3); j += 4 * WAVE; }
      }
      for (; __any(j < nb); j += WAVE) {
        const uint32_t w = j < nb ? (uint32_t)stream_ld(wp + j) : 0u;
        const double xv = xat(w);
        if (j < nb) s0 = fma(stream_ld(vp + j), xv, s0);
      }
      double s = (s0 + s1) + (s2 + s3);
      for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
      epi.store(d.r0, nr, lane, s, pre[0]);
      return;
    }
  }
  if (one) {
    // one row: lane l holds entries l, l + 64, ... - what it would read back from the staging area.  Same products, same
    // order of additions as the staged path (multiply, then add: no contraction)
    double s = 0.0;
    {
#pragma clang fp contract(off)
#pragma unroll
      for (int u = 0; u < EPL; u++) if (R::has(u)) { const double p = vv[u] * xx[u]; s = s + p; }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    epi.store(d.r0, nr, lane, s, pre[0]);
    return;
  }
  // phase 1: the slot is the row-major position
#pragma unroll
  for (int u = 0; u < EPL; u++) if (R::has(u)) lds[lane + u * WAVE + 1] = vv[u] * xx[u];
  wave_lds_sync();
  sum_rows<true>(d, lds, epi, lane, two, ra, re, pre);
}

// Epi interface (r0 = first row of the block, nr = its rows, lr = this lane's local row, may be >= nr):
//   Early early(r0, nr, lr)            loads whose result feeds another load (issued with the stream)
//   Pre   load(r0, nr, lr, Early)      the epilogue's inputs, issued before the gather is waited for
//   void  store(r0, nr, lr, acc, Pre)  after the row sum is known; rows past the block's end are dropped
// MB: the launch holds multi-base blocks only (lean path) / none of them (the other kinds)
template <bool MB, bool DIST, class Epi>
__device__ __forceinline__ void rb_block(const RbView& A, const XVec<DIST> x, double* __restrict__ lds, const Epi& epi) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int bi = my_block(A);
  if (bi < 0) return;
  const BlkInfo d = load_blk(A, bi);
  const int nf = (d.nb < BLK_NNZ ? d.nb : BLK_NNZ) >> 6;     // full rounds: one straight-line variant each (every long row: 8)
  if constexpr (MB) {
    if (d.cbase == -1) {
      switch (nf) {
        case 8: rb_body_mb<8, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 7: rb_body_mb<7, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 6: rb_body_mb<6, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 5: rb_body_mb<5, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 4: rb_body_mb<4, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 3: rb_body_mb<3, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 2: rb_body_mb<2, true, DIST, false>(A, d, x, lds, epi, lane); break;
        case 1: rb_body_mb<1, true, DIST, false>(A, d, x, lds, epi, lane); break;
        default: rb_body_mb<0, true, DIST, false>(A, d, x, lds, epi, lane); break;
      }
      return;
    }
    if ((-(d.cbase + 2)) >> MB_ND_SHIFT) {
      // a block with a value dictionary: straight-line variants of its own
      switch (nf) {
        case 8: rb_body_mb<8, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 7: rb_body_mb<7, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 6: rb_body_mb<6, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 5: rb_body_mb<5, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 4: rb_body_mb<4, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 3: rb_body_mb<3, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 2: rb_body_mb<2, false, DIST, true>(A, d, x, lds, epi, lane); break;
        case 1: rb_body_mb<1, false, DIST, true>(A, d, x, lds, epi, lane); break;
        default: rb_body_mb<0, false, DIST, true>(A, d, x, lds, epi, lane); break;
      }
      return;
    }
    switch (nf) {
      case 8: rb_body_mb<8, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 7: rb_body_mb<7, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 6: rb_body_mb<6, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 5: rb_body_mb<5, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 4: rb_body_mb<4, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 3: rb_body_mb<3, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 2: rb_body_mb<2, false, DIST, false>(A, d, x, lds, epi, lane); break;
      case 1: rb_body_mb<1, false, DIST, false>(A, d, x, lds, epi, lane); break;
      default: rb_body_mb<0, false, DIST, false>(A, d, x, lds, epi, lane); break;
    }
  } else {
    switch (nf) {
      case 8: rb_body<8, DIST>(A, d, x, lds, epi, lane); break;
      case 7: rb_body<7, DIST>(A, d, x, lds, epi, lane); break;
      case 6: rb_body<6, DIST>(A, d, x, lds, epi, lane); break;
      case 5: rb_body<5, DIST>(A, d, x, lds, epi, lane); break;
      case 4: rb_body<4, DIST>(A, d, x, lds, epi, lane); break;
      case 3: rb_body<3, DIST>(A, d, x, lds, epi, lane); break;
      case 2: rb_body<2, DIST>(A, d, x, lds, epi, lane); break;
      case 1: rb_body<1, DIST>(A, d, x, lds, epi, lane); break;
      default: rb_body<0, DIST>(A, d, x, lds, epi, lane); break;
    }
  }
}

// out_i = alpha * s_i [* dscale_i] + beta * z_i + gamma * w_i ;  y2_i += delta * (aux_out ? out_i : w2_i)
// Which of z / w / dscale / y2 are present is a compile-time mask (EPI_Z ...): the common smoothing and residual
// products carry one or no extra vector and must not pay registers and branches for the polynomial applies' fields.
constexpr int EPI_Z = 1, EPI_W = 2, EPI_D = 4, EPI_Y = 8;
template <int F>
struct EpiAxpby {
  SpmvEpi e;
  struct Early {};
  struct Pre { double z, w, d, y2, w2; };
  __device__ __forceinline__ Early early(int32_t, int, int) const { return Early{}; }
  __device__ __forceinline__ Pre load(int32_t r0, int nr, int lr, Early) const {
    Pre p; p.z = 0.0; p.w = 0.0; p.d = 1.0; p.y2 = 0.0; p.w2 = 0.0;
    const int ext = nr * 8;
    if constexpr ((F & EPI_Z) != 0) p.z = buf_f64(make_rsrc32(e.z + r0, ext), lr * 8);
    if constexpr ((F & EPI_W) != 0) p.w = buf_f64(make_rsrc32(e.w + r0, ext), lr * 8);
    if constexpr ((F & EPI_D) != 0) p.d = buf_f64(make_rsrc32(e.dscale + r0, ext), lr * 8);
    if constexpr ((F & EPI_Y) != 0) { p.y2 = buf_f64(make_rsrc32(e.y2 + r0, ext), lr * 8); if (!e.aux_out) p.w2 = buf_f64(make_rsrc32(e.w2 + r0, ext), lr * 8); }
    return p;
  }
  __device__ __forceinline__ void store(int32_t r0, int nr, int lr, double acc, const Pre& p) const {
    if constexpr ((F & EPI_D) != 0) acc *= p.d;
    double o = e.alpha * acc;
    if constexpr ((F & EPI_Z) != 0) o += e.beta * p.z;
    if constexpr ((F & EPI_W) != 0) o += e.gamma * p.w;
    const int ext = nr * 8;
    buf_store_f64(make_rsrc32(e.out + r0, ext), lr * 8, o);
    if constexpr ((F & EPI_Y) != 0) buf_store_f64(make_rsrc32(e.y2 + r0, ext), lr * 8, p.y2 + e.delta * (e.aux_out ? o : p.w2));
  }
};

// k_rb_spmv: multi-base blocks (the lean path); k_rb_spmv_g: the blocks of the other kinds
template <bool DIST, int F>
__global__ __launch_bounds__(SPMV_BLOCK) void k_rb_spmv(RbView A, XVec<DIST> x, EpiAxpby<F> epi) {
  __shared__ double lds_all[SPMV_WAVES][LDS_SLOTS];
  rb_block<true, DIST>(A, x, lds_all[threadIdx.x >> 6], epi);
}
template <bool DIST, int F>
__global__ __launch_bounds__(SPMV_BLOCK) void k_rb_spmv_g(RbView A, XVec<DIST> x, EpiAxpby<F> epi) {
  __shared__ double lds_all[SPMV_WAVES][LDS_SLOTS];
  rb_block<false, DIST>(A, x, lds_all[threadIdx.x >> 6], epi);
}
// the same kernels under other names for the products of the setup phase (SpmvEpi::setup_phase)
template <int F>
__global__ __launch_bounds__(SPMV_BLOCK) void k_rb_spmv_setup(RbView A, XVec<false> x, EpiAxpby<F> epi) {
  __shared__ double lds_all[SPMV_WAVES][LDS_SLOTS];
  rb_block<true, false>(A, x, lds_all[threadIdx.x >> 6], epi);
}
template <int F>
__global__ __launch_bounds__(SPMV_BLOCK) void k_rb_spmv_setup_g(RbView A, XVec<false> x, EpiAxpby<F> epi) {
  __shared__ double lds_all[SPMV_WAVES][LDS_SLOTS];
  rb_block<false, false>(A, x, lds_all[threadIdx.x >> 6], epi);
}

// Fused first stage of an F-point relaxation with a one-point prolongator (see dev.hpp)
// DIAG: A_ff and its inverse are diagonal and the whole F smooth rides along (a compile-time switch: the split mode must
// not carry the registers of the two diagonal vectors)
template <bool DIST, bool DIAG>
struct EpiFcFirst {
  const int32_t* __restrict__ wcol; const double* __restrict__ b_f; XVec<DIST> x_c;
  const double* __restrict__ aff_d; const double* __restrict__ minv;
  double* __restrict__ t4_out; double* __restrict__ x_f;
  struct Early { int32_t wc; };
  struct Pre { double b, x0, ad, mi; };
  __device__ __forceinline__ Early early(int32_t r0, int nr, int lr) const {
    // rows past the block's end read 0 = a valid column
    return Early{(int32_t)__builtin_amdgcn_raw_buffer_load_b32(make_rsrc32(wcol + r0, nr * 4), lr * 4, 0, 0)};
  }
  __device__ __forceinline__ Pre load(int32_t r0, int nr, int lr, Early ea) const {
    Pre p;
    const int ext = nr * 8;
    p.b = buf_f64(make_rsrc32(b_f + r0, ext), lr * 8);
    p.x0 = x_c.at_byte_offset((uint32_t)(ea.wc >= 0 ? ea.wc : 0) << 3);
    if (ea.wc < 0) p.x0 = 0.0;
    p.ad = 0.0; p.mi = 0.0;
    if constexpr (DIAG) { p.ad = buf_f64(make_rsrc32(aff_d + r0, ext), lr * 8); p.mi = buf_f64(make_rsrc32(minv + r0, ext), lr * 8); }
    return p;
  }
  __device__ __forceinline__ void store(int32_t r0, int nr, int lr, double acc, const Pre& p) const {
    const double t4 = p.b + (-1.0) * acc;
    const int ext = nr * 8;
    if constexpr (DIAG) {
      // one F smooth with diagonal A_ff and diagonal inverse (src/FC_Smooth.F90:541-559)
      double t3 = p.ad * p.x0;
      t3 = t4 + (-1.0) * t3;
      buf_store_f64(make_rsrc32(x_f + r0, ext), lr * 8, p.x0 + p.mi * t3);
    } else {
      buf_store_f64(make_rsrc32(t4_out + r0, ext), lr * 8, t4);
      buf_store_f64(make_rsrc32(x_f + r0, ext), lr * 8, p.x0);
    }
  }
};

template <bool DIST, bool DIAG>
__global__ __launch_bounds__(SPMV_BLOCK) void k_rb_fc_first(RbView A, XVec<DIST> x_c, EpiFcFirst<DIST, DIAG> epi) {
  __shared__ double lds_all[SPMV_WAVES][LDS_SLOTS];
  rb_block<true, DIST>(A, x_c, lds_all[threadIdx.x >> 6], epi);
}
template <bool DIST, bool DIAG>
__global__ __launch_bounds__(SPMV_BLOCK) void k_rb_fc_first_g(RbView A, XVec<DIST> x_c, EpiFcFirst<DIST, DIAG> epi) {
  __shared__ double lds_all[SPMV_WAVES][LDS_SLOTS];
  rb_block<false, DIST>(A, x_c, lds_all[threadIdx.x >> 6], epi);
}

// grid: one wave per block, four waves per workgroup, a multiple of 8 workgroups (every XCD class gets
// ceil(its blocks / 4) workgroups)
static unsigned grid_for(int64_t nblocks) {
  const int64_t per_class = (nblocks + 7) / 8;                       // blocks of the largest eighth
  int64_t wgs = 8 * ((per_class + SPMV_WAVES - 1) / SPMV_WAVES);
  if (wgs < 8) wgs = 8;
  if (wgs > 0x7fffffffll) throw std::runtime_error("matrix has too many row blocks for one launch");
  return (unsigned)wgs;
}

static void check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e) + " launching " + what);
}

template <bool MB, bool DIST, int F>
static void launch_spmv_f(const RbView& v, const XVec<DIST>& xv, const SpmvEpi& e, hipStream_t s) {
  const EpiAxpby<F> epi{e};
  const dim3 g(grid_for(v.nblocks)), b(SPMV_BLOCK);
  if constexpr (!DIST && (F == 0 || F == 1 || F == 9)) {      // the epilogues the setup's Arnoldi / Newton apply use
    if (e.setup_phase) {
      if constexpr (MB) hipLaunchKernelGGL((k_rb_spmv_setup<F>), g, b, 0, s, v, xv, epi);
      else hipLaunchKernelGGL((k_rb_spmv_setup_g<F>), g, b, 0, s, v, xv, epi);
      return;
    }
  }
  if constexpr (MB) hipLaunchKernelGGL((k_rb_spmv<DIST, F>), g, b, occ_pad(g_occ_spmv), s, v, xv, epi);
  else hipLaunchKernelGGL((k_rb_spmv_g<DIST, F>), g, b, 0, s, v, xv, epi);
}
template <bool MB, bool DIST>
static void launch_spmv_d(const RbView& v, const XVec<DIST>& xv, const SpmvEpi& e, hipStream_t s) {
  if (v.nblocks == 0) return;
  const int f = (e.z ? EPI_Z : 0) | (e.w ? EPI_W : 0) | (e.dscale ? EPI_D : 0) | (e.y2 ? EPI_Y : 0);
  switch (f) {
#define C(F) case F: launch_spmv_f<MB, DIST, F>(v, xv, e, s); break;
    C(0) C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15)
#undef C
  }
}

void launch_spmv(const DevMat& A, const double* x, const SpmvEpi& e, hipStream_t s, const double* x_ghost, int list, int part) {
  if (A.m == 0) return;
  if (!A.desc_ord) throw std::logic_error("launch_spmv: operator without a launch list");
  const RbView vm = view_of(A, list, true), vg = view_of(A, list, false);
  if (A.n == 0) x = A.val;     // no columns: blocks of empty rows still read element 0 of "x"
  if (A.n_ghost > 0) {
    // distributed operand: ghost columns start at A.n - A.n_ghost
    if (!x_ghost) throw std::runtime_error("launch_spmv: distributed operand without ghost values");
    const int32_t nl = (int32_t)(A.n - A.n_ghost);
    const XVec<true> xv{x, x_ghost - nl, nl};
    if (part != RB_GENERAL) launch_spmv_d<true, true>(vm, xv, e, s);
    if (part != RB_MB) launch_spmv_d<false, true>(vg, xv, e, s);
  } else {
    const XVec<false> xv{x, nullptr, 0};
    if (part != RB_GENERAL) launch_spmv_d<true, false>(vm, xv, e, s);
    if (part != RB_MB) launch_spmv_d<false, false>(vg, xv, e, s);
  }
  check_launch("k_rb_spmv");
}

template <bool MB, bool DIST>
static void launch_fc_first_d(const RbView& v, const XVec<DIST>& xv, const int32_t* wcol, const double* b_f, const double* aff_d,
                              const double* minv, double* t4_out, double* x_f, hipStream_t s) {
  if (v.nblocks == 0) return;
  const dim3 g(grid_for(v.nblocks)), b(SPMV_BLOCK);
  if (aff_d) {
    const EpiFcFirst<DIST, true> epi{wcol, b_f, xv, aff_d, minv, t4_out, x_f};
    if constexpr (MB) hipLaunchKernelGGL((k_rb_fc_first<DIST, true>), g, b, occ_pad(g_occ_fc), s, v, xv, epi);
    else hipLaunchKernelGGL((k_rb_fc_first_g<DIST, true>), g, b, 0, s, v, xv, epi);
  } else {
    const EpiFcFirst<DIST, false> epi{wcol, b_f, xv, aff_d, minv, t4_out, x_f};
    if constexpr (MB) hipLaunchKernelGGL((k_rb_fc_first<DIST, false>), g, b, occ_pad(g_occ_fc), s, v, xv, epi);
    else hipLaunchKernelGGL((k_rb_fc_first_g<DIST, false>), g, b, 0, s, v, xv, epi);
  }
}

void launch_fc_first(const DevMat& A_fc, const int32_t* wcol, const double* b_f, const double* x_c,
                     const double* aff_d, const double* minv, double* t4_out, double* x_f, hipStream_t s,
                     const double* x_ghost, int list, int part) {
  if (A_fc.m == 0) return;
  if (!A_fc.desc_ord) throw std::logic_error("launch_fc_first: operator without a launch list");
  if (A_fc.n == 0) x_c = A_fc.val;
  const RbView vm = view_of(A_fc, list, true), vg = view_of(A_fc, list, false);
  if (A_fc.n_ghost > 0) {
    if (!x_ghost) throw std::runtime_error("launch_fc_first: distributed operand without ghost values");
    const int32_t nl = (int32_t)(A_fc.n - A_fc.n_ghost);
    const XVec<true> xv{x_c, x_ghost - nl, nl};
    if (part != RB_GENERAL) launch_fc_first_d<true, true>(vm, xv, wcol, b_f, aff_d, minv, t4_out, x_f, s);
    if (part != RB_MB) launch_fc_first_d<false, true>(vg, xv, wcol, b_f, aff_d, minv, t4_out, x_f, s);
  } else {
    const XVec<false> xv{x_c, nullptr, 0};
    if (part != RB_GENERAL) launch_fc_first_d<true, false>(vm, xv, wcol, b_f, aff_d, minv, t4_out, x_f, s);
    if (part != RB_MB) launch_fc_first_d<false, false>(vg, xv, wcol, b_f, aff_d, minv, t4_out, x_f, s);
  }
  check_launch("k_rb_fc_first");
}


// ---------------------------------------------------------------------------------------
// CSR (device, any row order) -> row-block CSR (setup time)
// ---------------------------------------------------------------------------------------
constexpr int SCAN_CHUNK = 2048;   // elements per workgroup of the two-level scans

__global__ __launch_bounds__(BLOCK) void k_row_len(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ rowsrc,
                                                  int64_t* __restrict__ len) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < m; r += stride) {
    const int64_t src = rowsrc ? rowsrc[r] : r;
    len[r] = rp[src + 1] - rp[src];
  }
}

// chunk sums for a two-level exclusive scan
__global__ __launch_bounds__(BLOCK) void k_chunk_sum(int64_t n, const int64_t* __restrict__ in, int64_t* __restrict__ sums) {
  __shared__ int64_t red[BLOCK];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
  int64_t s = 0;
  for (int i = threadIdx.x; i < SCAN_CHUNK; i += BLOCK) { const int64_t k = base + i; if (k < n) s += in[k]; }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = BLOCK / 2; off > 0; off >>= 1) { if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off]; __syncthreads(); }
  if (threadIdx.x == 0) sums[blockIdx.x] = red[0];
}

// out[k] = offs[chunk] + exclusive prefix inside the chunk; out[n] is written by the host
__global__ __launch_bounds__(BLOCK) void k_chunk_scan(int64_t n, const int64_t* __restrict__ in, const int64_t* __restrict__ offs,
                                                     int64_t* __restrict__ out) {
  __shared__ int64_t tsum[BLOCK];
  const int64_t base = (int64_t)blockIdx.x * SCAN_CHUNK;
  constexpr int PER = SCAN_CHUNK / BLOCK;
  // each thread owns PER consecutive elements
  int64_t loc[PER]; int64_t s = 0;
#pragma unroll
  for (int j = 0; j < PER; j++) { const int64_t k = base + (int64_t)threadIdx.x * PER + j; loc[j] = k < n ? in[k] : 0; s += loc[j]; }
  tsum[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) { int64_t run = 0; for (int t = 0; t < BLOCK; t++) { int64_t c = tsum[t]; tsum[t] = run; run += c; } }
  __syncthreads();
  int64_t run = offs[blockIdx.x] + tsum[threadIdx.x];
#pragma unroll
  for (int j = 0; j < PER; j++) { const int64_t k = base + (int64_t)threadIdx.x * PER + j; if (k < n) out[k] = run; run += loc[j]; }
}

// exclusive scan of n int64 values on the device; returns the total
int64_t device_exclusive_scan(int64_t n, const int64_t* d_in, int64_t* d_out, hipStream_t s) {
  if (n == 0) return 0;
  const int64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
  int64_t* d_sums = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)nchunks);
  hipLaunchKernelGGL(k_chunk_sum, dim3((unsigned)nchunks), dim3(BLOCK), 0, s, n, d_in, d_sums);
  std::vector<int64_t> h(nchunks);
  HIP_CHECK(hipMemcpyAsync(h.data(), d_sums, sizeof(int64_t) * (size_t)nchunks, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  int64_t run = 0;
  for (int64_t i = 0; i < nchunks; i++) { int64_t c = h[i]; h[i] = run; run += c; }
  HIP_CHECK(hipMemcpyAsync(d_sums, h.data(), sizeof(int64_t) * (size_t)nchunks, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)nchunks), dim3(BLOCK), 0, s, n, d_in, d_sums, d_out);
  HIP_CHECK(hipStreamSynchronize(s));
  dev_free(d_sums);
  return run;
}

// copy (and remap) the entries of every destination row; a wave copies one row at a time with
// coalesced accesses - setup only
__global__ __launch_bounds__(BLOCK) void k_copy_rows(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                    const double* __restrict__ v, const int32_t* __restrict__ rowsrc,
                                                    const int32_t* __restrict__ colmap, int32_t n_mapped,
                                                    const int64_t* __restrict__ nrp,
                                                    int32_t* __restrict__ ocol, double* __restrict__ oval) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t r = wave; r < m; r += nwaves) {
    const int64_t src = rowsrc ? rowsrc[r] : r;
    const int64_t a = rp[src], len = rp[src + 1] - a, o = nrp[r];
    for (int64_t k = lane; k < len; k += WAVE) {
      int32_t c = ci[a + k];
      if (colmap && c < n_mapped) c = colmap[c];   // columns >= n_mapped (ghosts) keep their index
      ocol[o + k] = c; oval[o + k] = v[a + k];
    }
  }
}

// Greedy block partition inside chunks of CHUNK_ROWS rows (one thread per chunk; FILL = false counts the blocks
// of every chunk, FILL = true writes descriptors and row offsets): a block takes rows while it stays within
// BLK_NNZ non-zeros and BLK_ROWS rows; a row longer than BLK_NNZ is a block of its own.
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void k_chunk_blocks(int64_t m, const int64_t* __restrict__ nrp, int64_t* __restrict__ cnt,
                                                       const int64_t* __restrict__ first_blk, BlkDesc* __restrict__ blk,
                                                       uint16_t* __restrict__ roff) {
  const int64_t nchunks = (m + CHUNK_ROWS - 1) / CHUNK_ROWS;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t c = (int64_t)blockIdx.x * BLOCK + threadIdx.x; c < nchunks; c += stride) {
    const int64_t r_lo = c * CHUNK_ROWS, r_hi = min(r_lo + (int64_t)CHUNK_ROWS, m);
    int64_t nblk = 0, b = FILL ? first_blk[c] : 0;
    int64_t p_start = nrp[r_lo];
    int rows = 0;
    int64_t pr = p_start;                 // nrp[r]
    for (int64_t r = r_lo; r < r_hi; r++) {
      const int64_t pn = nrp[r + 1];
      const int64_t len = pn - pr;
      if (rows > 0 && (pn - p_start > BLK_NNZ || rows == BLK_ROWS)) { nblk++; rows = 0; p_start = pr; }
      if (rows == 0 && FILL) { BlkDesc& q = blk[b + nblk]; q.p0 = pr; q.row = (int32_t)r; q.cbase = -1; q.nb = 0; q.nr = 0; q.ip = 0; }
      if (FILL) roff[r] = (uint16_t)(len > BLK_NNZ ? 0 : (pr - p_start));
      rows++;
      if (len > BLK_NNZ) { nblk++; rows = 0; p_start = pn; }   // a long row closes its own block
      pr = pn;
    }
    if (rows > 0) nblk++;
    if (!FILL) cnt[c] = nblk;
  }
}

// Memory position of the value of logical slot s inside a block of n (<= 512) entries: every pair of full rounds
// (2i, 2i+1) is interleaved lane by lane, the rest stays in slot order (see the header; issue_stream reads it back)
__device__ __forceinline__ int val_slot(int s, int n) {
  const int npair = (n >> 6) >> 1;
  if (s >= 2 * WAVE * npair) return s;
  return (s >> 7) * (2 * WAVE) + 2 * (s & (WAVE - 1)) + ((s >> 6) & 1);
}

// One wave per block: decides the block's kind (dev.hpp: BlkDesc), writes its values in the interleaved placement and its
// index region IN PLACE over the block's columns (`col`, 4 bytes per entry: every kind needs at most that; all columns
// of a block of <= 512 entries are in LDS before the first write) and leaves the region's size in BlkDesc::ip for the
// compaction that follows.
//   packed: columns span < 2^COL_BITS: sorted by column (bitonic sort of (column << 16 | position) keys in LDS);
//   multi-base (mb_on; blocks of <= mb_max_rows rows, and blocks that cannot be packed): the windows are the distinct
//     values of column >> ob over the block's sorted columns, ob the largest of 14 ... 10 for which there are at most
//     2^(16 - ob) (<= 64) of them; a window's base is its smallest column;
//   raw: the rest (and rows longer than 512).
//   value dictionary (vdict_on; multi-base blocks of <= 512 entries): the block's values have at most MB_MAX_ND distinct bit
//     patterns and the dictionary form (8 nd + 512 bytes) is at most 5/8 of the plain values: dev.hpp.
__global__ __launch_bounds__(BLOCK) void k_block_pack(int64_t nblocks, int64_t nnz, int64_t m, BlkDesc* __restrict__ blk, int32_t* col,
                                                     double* __restrict__ val, int mb_on, int mb_max_rows, int vdict_on) {
  // (values: a block that gets a dictionary, or the packed form, rewrites its values IN PLACE - all of them in registers, eight
  // a lane, before the first store; every other block leaves them as they are and k_val_compact puts them into the
  // interleaved placement on its way to the final array)
  __shared__ unsigned long long keys_all[WAVES_PER_BLOCK][BLK_NNZ];
  double* const oval = val;
  __shared__ uint32_t tab_all[WAVES_PER_BLOCK][MB_MAX_WIN];
  __shared__ unsigned short word_all[WAVES_PER_BLOCK][BLK_NNZ];
  unsigned long long* keys = keys_all[threadIdx.x >> 6];
  uint32_t* tab = tab_all[threadIdx.x >> 6];
  unsigned short* word = word_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  auto lds_sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    const int64_t p0 = blk[b].p0, p1 = b + 1 < nblocks ? blk[b + 1].p0 : nnz;
    const int64_t nbl = p1 - p0;
    const int nrows = (int)((b + 1 < nblocks ? (int64_t)blk[b + 1].row : m) - blk[b].row);
    if (lane == 0) { blk[b].nb = (int32_t)nbl; blk[b].nr = nrows; blk[b].ip = 4 * nbl; }
    uint32_t* oidx = reinterpret_cast<uint32_t*>(col);
    if (nbl == 0 || nbl > BLK_NNZ) {
      // one long row (or an empty block): row-major order; k_val_compact gives the first 512 values the interleaved placement
      if (nbl == 0 || !mb_on) continue;      // raw: the absolute columns are in place already
      // multi-base form of a long row: its distinct windows (column >> ob) are collected in a small hash set in LDS (the row
      // is too long to sort here); a window's base is its smallest column, its number its rank among the row's windows
      uint32_t* wkey = reinterpret_cast<uint32_t*>(keys);          // [128] window or EMPTY
      uint32_t* wmin = wkey + 128;                                  // [128] smallest column of the window
      uint32_t* wid = wmin + 128;                                   // [128] rank of the window
      constexpr uint32_t EMPTY = 0xffffffffu;
      int ob = 0, nwin = 0;
      for (int t = 14; t >= 10 && ob == 0; t--) {
        for (int i = lane; i < 128; i += WAVE) { wkey[i] = EMPTY; wmin[i] = EMPTY; }
        lds_sync();
        bool over = false;
        for (int64_t k = p0 + lane; k < p1; k += WAVE) {
          const uint32_t c = (uint32_t)col[k], w = c >> t;
          uint32_t h = (w * 2654435761u) >> 25;
          int probes = 0;
          for (;;) {
            const uint32_t old = atomicCAS(&wkey[h], EMPTY, w);
            if (old == EMPTY || old == w) { atomicMin(&wmin[h], c); break; }
            h = (h + 1) & 127u;
            if (++probes == 128) { over = true; break; }
          }
        }
        lds_sync();
        int cnt = 0;
        for (int i = lane; i < 128; i += WAVE) cnt += wkey[i] != EMPTY ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, WAVE);
        if (!__any(over) && cnt <= (1 << (16 - t)) && cnt <= MB_MAX_WIN) { ob = t; nwin = cnt; }
      }
      if (!ob) continue;
      // ranks of the windows (deterministic: by value), table
      for (int i = lane; i < 128; i += WAVE) {
        const uint32_t me = wkey[i];
        int r = 0;
        if (me != EMPTY) for (int q = 0; q < 128; q++) r += wkey[q] < me ? 1 : 0;      // (EMPTY is the largest value)
        wid[i] = (uint32_t)r;
        if (me != EMPTY) tab[r] = wmin[i];
      }
      lds_sync();
      // words, written in place over the columns: chunk k + 1 is in registers before chunk k's words are written (the
      // words of chunk k end before the columns of chunk k + 2 begin; see the layout in dev.hpp)
      unsigned short* wout = reinterpret_cast<unsigned short*>(oidx + p0 + nwin);
      const int64_t nchunk = (nbl + WAVE - 1) / WAVE;
      uint32_t cur = lane < nbl ? (uint32_t)col[p0 + lane] : 0u;
      uint32_t tabreg = lane < nwin ? tab[lane] << 3 : 0u;
      for (int64_t ch = 0; ch < nchunk; ch++) {
        const int64_t i = ch * WAVE + lane, inext = i + WAVE;
        const uint32_t nxt = inext < nbl ? (uint32_t)col[p0 + inext] : 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (i < nbl) {
          const uint32_t w = cur >> ob;
          uint32_t h = (w * 2654435761u) >> 25;
          while (wkey[h] != w) h = (h + 1) & 127u;
          const uint32_t id = wid[h];
          wout[i < BLK_NNZ ? val_slot((int)i, BLK_NNZ) : i] = (unsigned short)((id << ob) | (cur - wmin[h]));
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cur = nxt;
      }
      // the table last (its bytes were columns of chunk 0)
      if (lane < nwin) oidx[p0 + lane] = tabreg;
      if (lane == 0) { blk[b].cbase = -2 - (ob | (nwin << MB_WIN_SHIFT)); blk[b].ip = 4 * ((int64_t)nwin + ((nbl + 1) >> 1)); }
      if ((nbl & 1) && lane == 0) wout[nbl] = 0;      // padding half of the last dword
      lds_sync();
      continue;
    }
    const int nb = (int)nbl;
    int P2 = WAVE; while (P2 < nb) P2 <<= 1;
    for (int i = lane; i < P2; i += WAVE)
      keys[i] = i < nb ? (((unsigned long long)(uint32_t)col[p0 + i]) << 16) | (unsigned long long)i : ~0ull;
    lds_sync();
    for (int size = 2; size <= P2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = lane; i < P2 / 2; i += WAVE) {
          const int a = 2 * i - (i & (stride - 1));
          const int bb = a + stride;
          const bool asc = (a & size) == 0;
          const unsigned long long ka = keys[a], kb = keys[bb];
          if ((ka > kb) == asc) { keys[a] = kb; keys[bb] = ka; }
        }
        lds_sync();
      }
    const int32_t cmin = (int32_t)(keys[0] >> 16), cmax = (int32_t)(keys[nb - 1] >> 16);
    const bool can_pack = (int64_t)cmax - (int64_t)cmin < (int64_t)(1u << COL_BITS);
    // multi-base?  count the windows for ob = 14 ... 10
    int ob = 0, nwin = 0;
    if (mb_on && (!can_pack || nrows <= mb_max_rows)) {
      for (int t = 14; t >= 10 && ob == 0; t--) {
        int cnt = 0;
        for (int i0 = 0; i0 < nb; i0 += WAVE) {
          const int i = i0 + lane;
          const bool fl = i < nb && (i == 0 || (((uint32_t)(keys[i] >> 16)) >> t) != (((uint32_t)(keys[i - 1] >> 16)) >> t));
          cnt += __popcll(__ballot(fl));
        }
        if (cnt <= (1 << (16 - t)) && cnt <= MB_MAX_WIN && 4 * cnt + 2 * nb <= 4 * nb) { ob = t; nwin = cnt; }
      }
    }
    if (ob) {
      // window of every sorted entry, its base; then the words at their row-major positions
      int run = 0;
      for (int i0 = 0; i0 < nb; i0 += WAVE) {
        const int i = i0 + lane;
        // (the keys of the round before carry their window in bits 48..: compare the column fields only)
        const bool fl = i < nb && (i == 0 || (((uint32_t)(keys[i] >> 16)) >> ob) != (((uint32_t)(keys[i - 1] >> 16)) >> ob));
        const unsigned long long bal = __ballot(fl);
        const int id = run + __popcll(bal & ((lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull)))) - 1;
        if (fl) tab[id] = (uint32_t)(keys[i] >> 16);
        if (i < nb) keys[i] = (keys[i] & 0xffffffffffffull) | ((unsigned long long)id << 48);   // columns are < 2^29: bits 48.. are free
        run += __popcll(bal);
      }
      lds_sync();
      for (int i = lane; i < nb; i += WAVE) {
        const unsigned long long k = keys[i];
        const int id = (int)(k >> 48), pos = (int)(k & 0xffffull);
        const uint32_t c = (uint32_t)((k >> 16) & 0xffffffffull);
        word[val_slot(pos, nb)] = (unsigned short)((id << ob) | (int)(c - tab[id]));
      }
      lds_sync();
      // region: table (byte offsets of the windows' first columns), then the words two per dword
      uint32_t* reg = oidx + p0;
      for (int k = lane; k < nwin; k += WAVE) reg[k] = tab[k] << 3;
      const int nd = (nb + 1) >> 1;
      for (int k = lane; k < nd; k += WAVE) {
        const uint32_t lo = word[2 * k], hi = 2 * k + 1 < nb ? word[2 * k + 1] : 0u;
        reg[nwin + k] = lo | (hi << 16);
      }
      // the values: plain, or - few distinct bit patterns - a dictionary.  The distinct patterns are collected in a hash set
      // (the keys array, free by now), counted (more than MB_MAX_ND: plain), sorted (ascending: the table does not depend
      // on the order of insertion) and every entry finds its index by bisection
      int ndv = 0;
      if (vdict_on && nb >= 2 * WAVE) {
        constexpr unsigned long long EMPTYV = ~0ull;
        double vreg[EPL];
        for (int u = 0; u < EPL; u++) vreg[u] = u * WAVE + lane < nb ? val[p0 + u * WAVE + lane] : 0.0;
        for (int i = lane; i < BLK_NNZ; i += WAVE) keys[i] = EMPTYV;
        lds_sync();
        bool bad = false;
        // (a block whose values are all different - every block of the deeper levels - is given up as soon as the set holds more
        // patterns than a dictionary may have: after five of its eight rounds, before the set fills up and the probes grow long)
        const int most = MB_MAX_ND < (5 * nb) / 8 - WAVE ? MB_MAX_ND : (5 * nb) / 8 - WAVE;
        int held = 0;
        for (int u = 0; u < EPL && held <= most; u++) {
          bool fresh = false;
          if (u * WAVE + lane < nb) {
            const unsigned long long v = (unsigned long long)__double_as_longlong(vreg[u]);
            if (v == EMPTYV) bad = true;
            else {
              uint32_t h = (uint32_t)((v * 0x9E3779B97F4A7C15ull) >> 55);      // 9 bits
              // (the set has as many slots as a block has entries, so every pattern finds one; the probe count is bounded all the same)
              for (int probes = 0;; probes++) {
                const unsigned long long old = atomicCAS(&keys[h], EMPTYV, v);
                if (old == EMPTYV) fresh = true;
                if (old == EMPTYV || old == v) break;
                if (probes == BLK_NNZ) { bad = true; break; }
                h = (h + 1) & (BLK_NNZ - 1);
              }
            }
          }
          held += __popcll(__ballot(fresh));
        }
        if (held > most) bad = true;
        lds_sync();
        int cnt = 0;
        for (int i = lane; i < BLK_NNZ; i += WAVE) cnt += keys[i] != EMPTYV ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, WAVE);
        if (!__any(bad) && cnt <= MB_MAX_ND && 8 * (cnt + WAVE) <= 5 * nb) {
          // ascending (bitonic over the 512 slots; EMPTYV is the largest pattern): the first cnt entries are the table
          for (int size = 2; size <= BLK_NNZ; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
              for (int i = lane; i < BLK_NNZ / 2; i += WAVE) {
                const int a = 2 * i - (i & (stride - 1));
                const int bb = a + stride;
                const bool asc = (a & size) == 0;
                const unsigned long long ka = keys[a], kb = keys[bb];
                if ((ka > kb) == asc) { keys[a] = kb; keys[bb] = ka; }
              }
              lds_sync();
            }
          ndv = cnt;
          unsigned long long* otab = reinterpret_cast<unsigned long long*>(oval + p0);
          for (int k = lane; k < ndv; k += WAVE) otab[k] = keys[k];
          // byte 8 l + u = index of entry 64 u + l (entries past the end: 0); assembled eight to a lane, one store each
          unsigned long long packed = 0;
          for (int u = 0; u < EPL; u++) {
            const int i = u * WAVE + lane;
            if (i < nb) {
              const unsigned long long v = (unsigned long long)__double_as_longlong(vreg[u]);
              int lo = 0, hi = ndv - 1;
              while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < v) lo = mid + 1; else hi = mid; }
              packed |= (unsigned long long)lo << (8 * u);
            }
          }
          otab[ndv + lane] = packed;
        }
        lds_sync();
      }
      if (lane == 0) { blk[b].cbase = -2 - (ob | (nwin << MB_WIN_SHIFT) | (ndv << MB_ND_SHIFT)); blk[b].ip = 4 * (int64_t)(nwin + nd); }
    } else if (can_pack && (!mb_on || nrows > mb_max_rows)) {
      // packed: the multi-base form is off, or the block has more rows than mb16_max_rows admits to it (the documented
      // meaning of that knob: such blocks keep the packed form of rounds 2-3 and run in the *_g kernels).  With the
      // multi-base form on, a block that was ADMITTED but fits no table stays raw: the lean kernels take it, and a launch
      // of the *_g kernels for a handful of one-entry blocks per operator costs more than their 4-byte words.
      // (in place: the values of this lane's sorted slots are gathered into registers first, every load of the wave before its
      // first store)
      double vsel[EPL];
      for (int u = 0; u < EPL; u++) { const int i = u * WAVE + lane; vsel[u] = i < nb ? val[p0 + (int)(keys[i] & 0xffffull)] : 0.0; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int u = 0; u < EPL; u++) {
        const int i = u * WAVE + lane;
        if (i >= nb) continue;
        const unsigned long long k = keys[i];
        const int pos = (int)(k & 0xffffull);
        oidx[p0 + i] = ((uint32_t)((int32_t)(k >> 16) - cmin) << (POS_BITS + 3)) | (uint32_t)(pos + 1);
        oval[p0 + val_slot(i, nb)] = vsel[u];
      }
      if (lane == 0) blk[b].cbase = cmin;
    }
    lds_sync();
  }
}

// the index regions back to back: ip_in[b] = size of block b's region (bytes, a multiple of 4), ip_out[b] = its offset in the
// compact stream; the region is copied from where k_block_pack left it (byte 4 * p0 of `src`)
__global__ __launch_bounds__(BLOCK) void k_region_sizes(int64_t nblocks, const BlkDesc* __restrict__ blk, int64_t* __restrict__ sz) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x; b < nblocks; b += stride) sz[b] = blk[b].ip;
}
__global__ __launch_bounds__(BLOCK) void k_idx_compact(int64_t nblocks, BlkDesc* __restrict__ blk, const int64_t* __restrict__ off,
                                                      const uint32_t* __restrict__ src, uint8_t* __restrict__ dst) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    const int64_t nd = blk[b].ip >> 2, o = off[b];
    const uint32_t* __restrict__ from = src + blk[b].p0;
    uint32_t* __restrict__ to = reinterpret_cast<uint32_t*>(dst + o);
    for (int64_t k = lane; k < nd; k += WAVE) to[k] = from[k];
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) blk[b].ip = o;
  }
}

// column of the k-th stored entry of a block (any kind; storage order, which is all a scan over the entries needs)
__device__ __forceinline__ int32_t stored_column(const BlkDesc& d, const uint8_t* __restrict__ idx, int64_t k) {
  const uint32_t* __restrict__ w = reinterpret_cast<const uint32_t*>(idx + d.ip);
  if (d.cbase >= 0) return d.cbase + (int32_t)(w[k] >> (POS_BITS + 3));
  if (d.cbase == -1) return (int32_t)w[k];
  const int meta = -(d.cbase + 2), ob = meta & MB_OB_MASK, nwin = (meta >> MB_WIN_SHIFT) & MB_WIN_MASK;
  const uint32_t q = (w[nwin + (k >> 1)] >> ((k & 1) * 16)) & 0xffffu;
  return (int32_t)((w[q >> ob] >> 3) + (q & ((1u << ob) - 1u)));
}

// flag[b] = 1 if block b references a ghost column (column >= n_local); one wave per block
__global__ __launch_bounds__(BLOCK) void k_boundary_flags(int64_t nblocks, const BlkDesc* __restrict__ blk, const uint8_t* __restrict__ idx,
                                                         int32_t n_local, uint8_t* __restrict__ flag) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    const BlkDesc d = blk[b];
    bool any = false;
    for (int64_t k = lane; k < d.nb; k += WAVE) any |= stored_column(d, idx, k) >= n_local;
    if (__any(any) && lane == 0) flag[b] = 1;
  }
}

void devmat_boundary_flags(const DevMat& A, std::vector<uint8_t>& flags, hipStream_t s) {
  flags.assign((size_t)A.nblocks, 0);
  if (A.nblocks == 0 || A.n_ghost == 0) return;
  uint8_t* d = (uint8_t*)dev_alloc((size_t)A.nblocks);
  HIP_CHECK(hipMemsetAsync(d, 0, (size_t)A.nblocks, s));
  const int g = (int)std::min<int64_t>((A.nblocks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 4096);
  hipLaunchKernelGGL(k_boundary_flags, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, A.nblocks, A.blk, A.idx, (int32_t)(A.n - A.n_ghost), d);
  HIP_CHECK(hipMemcpyAsync(flags.data(), d, (size_t)A.nblocks, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  dev_free(d);
}

__global__ __launch_bounds__(BLOCK) void k_desc_gather(int64_t n, const BlkDesc* __restrict__ blk, const int32_t* __restrict__ order, BlkDesc* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) out[i] = blk[order[i]];
}

// ---- launch lists: the blocks of the lean kernels first (stable), the rest after them
__global__ __launch_bounds__(BLOCK) void k_mb_flags(int64_t n, const BlkDesc* __restrict__ d, int64_t* __restrict__ is_mb, int64_t* __restrict__ nz_mb) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const bool mb = lean_block(d[i]);
    is_mb[i] = mb ? 1 : 0;
    if (nz_mb) nz_mb[i] = mb ? d[i].nb : 0;
  }
}
__global__ __launch_bounds__(BLOCK) void k_mb_scatter(int64_t n, int64_t n_mb, const BlkDesc* __restrict__ in, const int64_t* __restrict__ pos, BlkDesc* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const BlkDesc q = in[i];
    out[lean_block(q) ? pos[i] : n_mb + (i - pos[i])] = q;
  }
}
static int grid1d(int64_t n);
// bytes of the value array a product reads for each block (dev.hpp: value dictionary)
__global__ __launch_bounds__(BLOCK) void k_val_bytes(int64_t n, const BlkDesc* __restrict__ d, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const int nd = d[i].cbase <= -2 ? mb_nd(mb_meta(d[i])) : 0;
    out[i] = nd ? 8 * (int64_t)nd + 8 * WAVE : 8 * (int64_t)d[i].nb;
  }
}
// the value regions back to back: k_val_bytes gave each block's bytes; the region is copied from where k_block_pack left it
// (the block's first non-zero in `src`) to its offset in the compact array, and the descriptor follows.  Dictionary and
// packed blocks were rewritten in place and are copied as they are; the others are still in row-major order and take the
// interleaved placement here (val_slot; a long row: its first 512 entries)
__global__ __launch_bounds__(BLOCK) void k_val_compact(int64_t nblocks, BlkDesc* __restrict__ blk, const int64_t* __restrict__ sz, const int64_t* __restrict__ off,
                                                      const double* __restrict__ src, double* __restrict__ dst) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    const int64_t nd = sz[b] >> 3, o = off[b] >> 3, p0 = blk[b].p0;
    const int32_t cbase = blk[b].cbase;
    const bool as_is = cbase >= 0 || (cbase <= -2 && mb_nd(mb_meta(blk[b])) > 0);
    if (as_is) {
      for (int64_t k = lane; k < nd; k += WAVE) dst[o + k] = src[p0 + k];
    } else {
      const int n512 = (int)(nd < BLK_NNZ ? nd : BLK_NNZ);
      for (int k = lane; k < n512; k += WAVE) dst[o + val_slot(k, n512)] = src[p0 + k];
      for (int64_t k = BLK_NNZ + lane; k < nd; k += WAVE) dst[o + k] = src[p0 + k];
    }
    if (lane == 0) blk[b].p0 = o;
  }
}
// out (n descriptors, may not alias in) = in with the multi-base blocks first; returns their number, *nnz_mb their non-zeros
static int64_t partition_descs_device(const BlkDesc* in, int64_t n, BlkDesc* out, hipStream_t s, int64_t* nnz_mb) {
  if (nnz_mb) *nnz_mb = 0;
  if (n == 0) return 0;
  int64_t* fl = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)n);
  int64_t* pos = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)n);
  int64_t* nz = nnz_mb ? (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)n) : nullptr;
  hipLaunchKernelGGL(k_mb_flags, dim3(grid1d(n)), dim3(BLOCK), 0, s, n, in, fl, nz);
  const int64_t n_mb = device_exclusive_scan(n, fl, pos, s);
  hipLaunchKernelGGL(k_mb_scatter, dim3(grid1d(n)), dim3(BLOCK), 0, s, n, n_mb, in, pos, out);
  if (nnz_mb) *nnz_mb = device_exclusive_scan(n, nz, fl, s);
  HIP_CHECK(hipStreamSynchronize(s));
  dev_free(fl); dev_free(pos); dev_free(nz);
  return n_mb;
}
int64_t partition_descs_host(std::vector<BlkDesc>& v) {
  auto mid = std::stable_partition(v.begin(), v.end(), [](const BlkDesc& q) { return lean_block(q); });
  return (int64_t)(mid - v.begin());
}

BlkDesc* devmat_desc_list(const DevMat& A, const int32_t* order_host, int64_t n, hipStream_t s, int64_t* n_mb) {
  BlkDesc* out = (BlkDesc*)dev_alloc_exact(sizeof(BlkDesc) * (size_t)std::max<int64_t>(n, 1));
  *n_mb = 0;
  if (n == 0) return out;
  int32_t* d_ord = (int32_t*)dev_alloc(sizeof(int32_t) * (size_t)n);
  BlkDesc* tmp = (BlkDesc*)dev_alloc(sizeof(BlkDesc) * (size_t)n);
  HIP_CHECK(hipMemcpyAsync(d_ord, order_host, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, s));
  const int g = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, MAX_BLOCKS);
  hipLaunchKernelGGL(k_desc_gather, dim3(g), dim3(BLOCK), 0, s, n, A.blk, d_ord, tmp);
  *n_mb = partition_descs_device(tmp, n, out, s, nullptr);
  dev_free(d_ord); dev_free(tmp);
  return out;
}

static int grid1d(int64_t n) {
  int64_t b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

void devmat_from_csr_device(int64_t m, int64_t n, const int64_t* d_rp, const int32_t* d_ci, const double* d_v,
                            const int32_t* d_rowsrc, const int32_t* d_colmap, DevMat& out, hipStream_t s, int64_t n_ghost, bool cached) {
  devmat_free(out);
  out.m = m; out.n = n; out.n_ghost = n_ghost; out.cached = cached;
  auto alloc_out = [&](size_t bytes) { return cached ? dev_alloc(bytes) : dev_alloc_exact(bytes); };
  if (m == 0) return;
  if (m >= (1ll << 29) || n >= (1ll << 29)) throw std::runtime_error("matrix dimension exceeds the 2^29 rows / columns of the 32-bit byte offsets");
  // temporaries come from the block cache (device/devmem.hpp), the arrays of the matrix itself are exact allocations
  int64_t* d_len = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)m);
  int64_t* d_nrp = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)(m + 1));
  hipLaunchKernelGGL(k_row_len, dim3(grid1d(m)), dim3(BLOCK), 0, s, m, d_rp, d_rowsrc, d_len);
  out.nnz = device_exclusive_scan(m, d_len, d_nrp, s);
  HIP_CHECK(hipMemcpyAsync(d_nrp + m, &out.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  dev_free(d_len);
  // entries in the new row order (plain columns), then packed per block
  int32_t* tcol = (int32_t*)dev_alloc(sizeof(int32_t) * (size_t)(out.nnz + 1));
  double* tval = (double*)dev_alloc(sizeof(double) * (size_t)(out.nnz + 1));
  hipLaunchKernelGGL(k_copy_rows, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, m, d_rp, d_ci, d_v, d_rowsrc, d_colmap,
                     (int32_t)(n - n_ghost), d_nrp, tcol, tval);
  // row blocks: count per chunk, scan, fill
  const int64_t nchunks = (m + CHUNK_ROWS - 1) / CHUNK_ROWS;
  int64_t* d_cnt = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)nchunks);
  int64_t* d_first = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)nchunks);
  hipLaunchKernelGGL((k_chunk_blocks<false>), dim3(grid1d(nchunks)), dim3(BLOCK), 0, s, m, d_nrp, d_cnt, nullptr, nullptr, nullptr);
  out.nblocks = device_exclusive_scan(nchunks, d_cnt, d_first, s);
  if (out.nblocks >= (1ll << 31) - 8) throw std::runtime_error("matrix has too many row blocks");
  out.blk = (BlkDesc*)alloc_out(sizeof(BlkDesc) * (size_t)(out.nblocks + 1));
  out.roff = (uint16_t*)alloc_out(sizeof(uint16_t) * (size_t)m);
  hipLaunchKernelGGL((k_chunk_blocks<true>), dim3(grid1d(nchunks)), dim3(BLOCK), 0, s, m, d_nrp, d_cnt, d_first, out.blk, out.roff);
  const BlkDesc end{out.nnz, (int32_t)m, 0, 0, 0, 0};            // readable padding entry (empty block)
  HIP_CHECK(hipMemcpyAsync(out.blk + out.nblocks, &end, sizeof(BlkDesc), hipMemcpyHostToDevice, s));
  // (the values are rearranged in place in tval and then laid back to back in the final array)
  const int g = (int)std::min<int64_t>((out.nblocks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 4096);
  // kinds, values and index regions (in place over tcol), then the regions back to back
  hipLaunchKernelGGL(k_block_pack, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, out.nblocks, out.nnz, m, out.blk, tcol, tval, g_mb16, g_mb16_max_rows, cached ? 0 : g_vdict);      // (a cached conversion serves a few setup products: no dictionary search)
  int64_t* d_sz = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)out.nblocks);
  int64_t* d_off = (int64_t*)dev_alloc(sizeof(int64_t) * (size_t)out.nblocks);
  hipLaunchKernelGGL(k_region_sizes, dim3(grid1d(out.nblocks)), dim3(BLOCK), 0, s, out.nblocks, out.blk, d_sz);
  out.idx_bytes = device_exclusive_scan(out.nblocks, d_sz, d_off, s);
  out.idx = (uint8_t*)alloc_out((size_t)out.idx_bytes + 4);
  HIP_CHECK(hipMemsetAsync(out.idx + out.idx_bytes, 0, 4, s));
  hipLaunchKernelGGL(k_idx_compact, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, out.nblocks, out.blk, d_off, reinterpret_cast<const uint32_t*>(tcol), out.idx);
  // the launch list in storage order (DevPCAIR::upload replaces it with a spatial order where that pays)
  out.desc_ord = (BlkDesc*)alloc_out(sizeof(BlkDesc) * (size_t)out.nblocks);
  // the value regions back to back (a block with a dictionary keeps nd + 64 doubles, the others their non-zeros); BEFORE the
  // launch list is derived: the descriptors' p0 move
  hipLaunchKernelGGL(k_val_bytes, dim3(grid1d(out.nblocks)), dim3(BLOCK), 0, s, out.nblocks, out.blk, d_sz);
  out.val_bytes = device_exclusive_scan(out.nblocks, d_sz, d_off, s);
  out.val_len = out.val_bytes >> 3;
  out.val = (double*)alloc_out(sizeof(double) * (size_t)(out.val_len + 1));
  HIP_CHECK(hipMemsetAsync(out.val + out.val_len, 0, sizeof(double), s));
  hipLaunchKernelGGL(k_val_compact, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, out.nblocks, out.blk, d_sz, d_off, tval, out.val);
  { const BlkDesc end2{out.val_len, (int32_t)m, 0, 0, 0, 0}; HIP_CHECK(hipMemcpyAsync(out.blk + out.nblocks, &end2, sizeof(BlkDesc), hipMemcpyHostToDevice, s)); HIP_CHECK(hipStreamSynchronize(s)); }
  out.n_mb = partition_descs_device(out.blk, out.nblocks, out.desc_ord, s, &out.nnz_mb);
  HIP_CHECK(hipStreamSynchronize(s));
  check_launch("row-block conversion");
  dev_free(d_nrp); dev_free(d_cnt); dev_free(d_first); dev_free(tcol); dev_free(tval); dev_free(d_sz); dev_free(d_off);
}

void devmat_free(DevMat& A) {
  if (A.cached) { dev_free(A.blk); dev_free(A.roff); dev_free(A.idx); dev_free(A.val); dev_free(A.desc_ord); A.blk = nullptr; A.roff = nullptr; A.idx = nullptr; A.val = nullptr; A.desc_ord = nullptr; }
  dev_free_exact(A.blk);
  dev_free_exact(A.roff);
  dev_free_exact(A.idx);
  dev_free_exact(A.val);
  dev_free_exact(A.desc_int);
  dev_free_exact(A.desc_bnd);
  dev_free_exact(A.desc_ord);
  A = DevMat();
}

}  // namespace airhip
