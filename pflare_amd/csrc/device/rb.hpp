// Shared by the row-block SpMV kernel file (device code).
#pragma once
#include "dev.hpp"
#include "kcommon.hpp"
#include <cstdlib>

namespace airhip {

constexpr int BLK_NNZ = 512;        // non-zeros per row block (a longer row is a block of its own)
constexpr int BLK_ROWS = 128;       // rows per row block: two per lane
constexpr int EPL = BLK_NNZ / WAVE; // entries per lane
constexpr int CHUNK_ROWS = 512;     // the block partition is greedy inside chunks of this many rows
// index word of a packed block: ((column - cbase) * 8) << POS_BITS | (row-major position + 1); the gather's byte offset
// is one shift away, the staging slot one mask
constexpr int POS_BITS = 10;
constexpr uint32_t POS_MASK = (1u << POS_BITS) - 1u;
constexpr int COL_BITS = 32 - POS_BITS - 3;   // 19: a packed block's columns span less than 2^19
// products staged per wave: slot 0 is the dummy slot of padding lanes, slots 1 .. 512 the row-major positions; the row
// sums clamp their reads to slot 513
constexpr int LDS_SLOTS = BLK_NNZ + 2;

struct RbView {
  const BlkDesc* __restrict__ blk;        // the blocks this launch processes, in processing order
  const uint16_t* __restrict__ roff;
  const uint8_t* __restrict__ idx;
  const double* __restrict__ val;
  int32_t nblocks;                        // how many
  int32_t per_class;                      // ceil(nblocks / 8): the blocks of one XCD class (see my_block)
};
// the multi-base (mb = true) or the other part of one of A's launch lists
inline RbView view_of(const DevMat& A, int list, bool mb) {
  const BlkDesc* d = list == RB_INTERIOR ? A.desc_int : list == RB_BOUNDARY ? A.desc_bnd : A.desc_ord;
  const int64_t n = list == RB_INTERIOR ? A.n_int : list == RB_BOUNDARY ? A.n_bnd : A.nblocks;
  const int64_t nm = list == RB_INTERIOR ? A.n_mb_int : list == RB_BOUNDARY ? A.n_mb_bnd : A.n_mb;
  const int64_t cnt = mb ? nm : n - nm;
  return RbView{d ? d + (mb ? 0 : nm) : nullptr, A.roff, A.idx, A.val, (int32_t)cnt, (int32_t)((cnt + 7) / 8)};
}

// Input vector of an SpMV.  Single device: DIST = false, plain x[c].  Row-block distributed
// (one rank per GPU): columns < nl index the rank's own segment, columns >= nl the ghost values
// received from the neighbours (xg is pre-shifted by -nl).
template <bool DIST> struct XVec {
  const double* __restrict__ x; const double* __restrict__ xg; int32_t nl;
  __device__ __forceinline__ double operator[](int32_t c) const {
    if constexpr (DIST) return c < nl ? x[c] : xg[c]; else return x[c];
  }
  // the same vector seen from column c0 (wave-uniform): byte offsets then count from c0
  __device__ __forceinline__ XVec shifted(int32_t c0) const { return XVec{x + c0, xg + c0, nl - c0}; }
  // element at a 32-bit byte offset (a multiple of 8) from the uniform base: one saddr + voffset load on a single
  // device (operators have fewer than 2^29 columns, checked at conversion)
  __device__ __forceinline__ double at_byte_offset(uint32_t off) const {
    if constexpr (DIST) { const int32_t c = (int32_t)(off >> 3); return c < nl ? x[c] : xg[c]; }
    else return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(x) + off);
  }
};

// exclusive scan of n int64 values on the device; returns the total (spmv.hip)
int64_t device_exclusive_scan(int64_t n, const int64_t* d_in, int64_t* d_out, hipStream_t s);

}  // namespace airhip
