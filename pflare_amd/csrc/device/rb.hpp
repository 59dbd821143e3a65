// Shared by the row-block SpMV kernel files (device code).
#pragma once
#include "dev.hpp"
#include "kcommon.hpp"
#include <cstdlib>

namespace airhip {

constexpr int BLK_NNZ = 512;        // non-zeros per row block, untiled format (+ the tail of the last row)
constexpr int BLK_NNZ_TILED = 256;  // tiled format: blocks of <= 255 + row length non-zeros
constexpr int LDS_CAP = 1024;       // products staged per wave by the untiled kernels (8 KB)
constexpr int BLK_ROWS = 256;       // at most this many rows per block

constexpr int TILE_NNZ = 2048;      // a tile = the row blocks whose first non-zero lies in one 2048-bucket
constexpr int TILE_CAP = 2560;      // distinct columns staged in LDS per tile (20 KB); more -> "direct" tile
constexpr int TILE_SORT_CAP = 4096; // largest tile the builder sorts; larger -> "direct" tile
constexpr int TILE_LDS = 512;       // products staged per wave by the tiled kernels (4 KB)

struct RbView {
  int64_t m, nblocks, nnz;
  const int32_t* blk_row;
  const int64_t* blk_p0;
  const uint32_t* roff;
  const int32_t* col;
  const double* val;
  const uint16_t* slot;   // column-sorted blocks: position of the entry in row-major order inside its block
  int scalar_t;           // blocks with nnz <= scalar_t * rows skip the LDS stage (one lane per row)
  const int32_t* blk_list;   // optional: process only these blocks (interior / boundary split of a distributed operand)
  int64_t nlist;
};
// AIRHIP_SCALAR_T (default 0 = never): average row length up to which a block is handled one lane per row
// straight from global memory (was faster before the blocks were column-sorted; now the LDS path wins by 4 %)
inline int scalar_threshold() {
  static const int t = std::getenv("AIRHIP_SCALAR_T") ? std::atoi(std::getenv("AIRHIP_SCALAR_T")) : 0;
  return t;
}
inline RbView view_of(const DevMat& A) {
  return RbView{A.m, A.nblocks, A.nnz, A.blk_row, A.blk_p0, A.roff, A.col, A.val, A.slot, scalar_threshold(), nullptr, 0};
}

// Input vector of an SpMV.  Single device: DIST = false, plain x[c].  Row-block distributed
// (one rank per GPU): columns < nl index the rank's own segment, columns >= nl the ghost values
// received from the neighbours (xg is pre-shifted by -nl).
template <bool DIST> struct XVec {
  const double* __restrict__ x; const double* __restrict__ xg; int32_t nl;
  __device__ __forceinline__ double operator[](int32_t c) const {
    if constexpr (DIST) return c < nl ? x[c] : xg[c]; else return x[c];
  }
};


// exclusive scan of n int64 values on the device; returns the total (spmv.hip)
int64_t device_exclusive_scan(int64_t n, const int64_t* d_in, int64_t* d_out, hipStream_t s);

// tiled format (spmv_tile.hip)
bool tiles_enabled();
void devmat_build_tiles(DevMat& A, hipStream_t s);
// return false when A has no tiles (the caller then uses the untiled kernels)
bool launch_spmv_tiled(const DevMat& A, const double* x, const SpmvEpi& e, hipStream_t s, const double* x_ghost);
bool launch_fc_first_tiled(const DevMat& A_fc, const int32_t* wcol, const double* b_f, const double* x_c, const double* aff_d,
                           const double* minv, double* t4_out, double* x_f, hipStream_t s, const double* x_ghost);

}  // namespace airhip
