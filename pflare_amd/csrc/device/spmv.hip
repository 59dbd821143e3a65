// Row-block CSR SpMV kernels (the dominant kernels of the AIR V-cycle) and the device-side
// CSR -> row-block CSR conversion.
//
// Format and mapping (gfx950, wave64):
//   * the matrix stays in CSR row order (no padding; SELL-64 padded the AIR operators by 1.75x, and
//     length-sorted SELL-C-sigma destroyed the locality of the x gathers);
//   * rows are partitioned into blocks of <= BLK_NNZ non-zeros (+ the tail of the last row, at most
//     BLK_ROWS rows); one WAVE owns one block: phase 1 streams the block's values / column indices
//     with coalesced wave loads, gathers x and stages the products in LDS (8 KB per wave);
//     phase 2 lets lane l sum the LDS segment of row r0+l (+64, ...) in column order - the same
//     summation order as a sequential CSR loop - and apply the fused epilogue, so the z / w / out
//     vectors are read and written coalesced;
//   * inside a block that takes the LDS path the (column, value) pairs are stored SORTED BY COLUMN
//     and a 16-bit slot gives the row-major position where the product is staged: neighbouring
//     lanes gather neighbouring x entries (the gather is bound by distinct cache lines per wave
//     instruction, profiles/r01_spmv_experiments.md); results are unchanged bit for bit;
//   * blocks with <= 4 non-zeros per row on average skip the LDS stage (one lane per row); a block
//     holding a row longer than the LDS stage uses a wave-wide reduction per row;
//   * XCD-aware assignment: blocks b and b+8 of the grid share an XCD (observed round-robin
//     dispatch; a speed assumption only), so each residue class of workgroups walks a contiguous
//     range of row blocks and an XCD's private L2 sees a contiguous window of x.
// airhip_debug_set_variant selects timing variants (LDS stage size, non-temporal loads, unroll depth, a
// no-far-gather diagnostic); the flat-stream diagnostics of profiles/r01_spmv_experiments.md were removed
// after the measurements.
#include "dev.hpp"
#include "kcommon.hpp"
#include "rb.hpp"
#include <stdexcept>
#include <vector>
#include <algorithm>
#include <cstdlib>

namespace airhip {

struct BlockRange { int64_t first, last, stride; };

// XCD-aware, grid-strided assignment of row blocks to waves (gridDim.x is a multiple of 8).
__device__ __forceinline__ BlockRange my_blocks(int64_t nblocks) {
  const int g = blockIdx.x & 7;
  const int bi = blockIdx.x >> 3;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  BlockRange r;
  const int64_t lo = nblocks * g / 8;
  r.last = nblocks * (g + 1) / 8;
  r.stride = (int64_t)(gridDim.x >> 3) * WAVES_PER_BLOCK;
  r.first = lo + (int64_t)bi * WAVES_PER_BLOCK + wid;
  return r;
}

extern int g_spmv_variant;
static int grid_for_blocks(int64_t nblocks) {
  int64_t blocks = (nblocks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  // diagnostic: bits 16..23 of the variant override the grid cap in units of 64 workgroups
  const int64_t cap = ((g_spmv_variant >> 16) & 0xff) ? (int64_t)((g_spmv_variant >> 16) & 0xff) * 64 : MAX_BLOCKS;
  if (blocks > cap) blocks = cap;
  blocks = (blocks + 7) / 8 * 8;
  if (blocks < 8) blocks = 8;
  return (int)blocks;
}

// Calls epi(row, sum_j a_row,j x_j) once for every row of the matrix.
template <class T, bool NT>
__device__ __forceinline__ T ldg(const T* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}

// diagnostic only (airhip_debug_set_variant(8)): every gather hits the same 8 KB of x - shows how
// much of the kernel time is the x-vector gather
struct XVecNoGather {
  const double* __restrict__ x;
  __device__ __forceinline__ double operator[](int32_t c) const { return x[c & 1023]; }
};

// SORTED: 0 = row-major blocks, 1 = column-sorted blocks + 16-bit slot array, 2 = column-sorted blocks with
// the slot packed into the upper 10 bits of the column word (operators with <= 2^22 columns: 12 B/nnz again)
template <int CAP, bool NT, int UNR, int SORTED, class X, class Epi>
__device__ __forceinline__ void rb_rows(const RbView& A, const X x, double* __restrict__ lds, Epi epi) {
  const int lane = threadIdx.x & (WAVE - 1);
  const BlockRange br = my_blocks(A.blk_list ? A.nlist : A.nblocks);
  for (int64_t bi = br.first; bi < br.last; bi += br.stride) {
    const int64_t b = A.blk_list ? (int64_t)__builtin_amdgcn_readfirstlane(A.blk_list[bi]) : bi;
    const int32_t r0 = __builtin_amdgcn_readfirstlane(A.blk_row[b]);
    const int32_t r1 = __builtin_amdgcn_readfirstlane(A.blk_row[b + 1]);
    const int64_t p0 = uniform64(A.blk_p0[b]);
    const int64_t p1 = uniform64(A.blk_p0[b + 1]);
    const int nb = (int)(p1 - p0);
    const double* __restrict__ v = A.val + p0;
    const int32_t* __restrict__ c = A.col + p0;
    const uint16_t* __restrict__ sl = A.slot + p0;
    const int nr = r1 - r0;
    constexpr int32_t CMASK = SORTED == 2 ? 0x3fffff : -1;
    if (nb <= A.scalar_t * nr) {
      // very short rows (<= 4 non-zeros on average): one lane per row straight from global
      // memory - the block's ~6 KB are shared by neighbouring lanes through L1, and the
      // dependency chain is one round trip shorter than the LDS stage
      for (int32_t r = r0 + lane; r < r1; r += WAVE) {
        const int a = (int)A.roff[r];
        const int e = (r + 1 < r1) ? (int)A.roff[r + 1] : nb;
        double s = 0.0;
        for (int j = a; j < e; j++) s += ldg<double, NT>(v + j) * x[ldg<int32_t, NT>(c + j) & CMASK];
        epi(r, s);
      }
    } else if (nb <= CAP) {
      // phase 1: coalesced stream, products to LDS (UNR independent loads in flight per lane)
      int k = lane;
      for (; k + (UNR - 1) * WAVE < nb; k += UNR * WAVE) {
        int32_t cc[UNR]; double vv[UNR], xx[UNR]; int ss[UNR];
#pragma unroll
        for (int u = 0; u < UNR; u++) cc[u] = ldg<int32_t, NT>(c + k + u * WAVE);
#pragma unroll
        for (int u = 0; u < UNR; u++) vv[u] = ldg<double, NT>(v + k + u * WAVE);
#pragma unroll
        for (int u = 0; u < UNR; u++)
          ss[u] = SORTED == 2 ? (int)((uint32_t)cc[u] >> 22) : SORTED == 1 ? (int)ldg<uint16_t, NT>(sl + k + u * WAVE) : k + u * WAVE;
#pragma unroll
        for (int u = 0; u < UNR; u++) xx[u] = x[cc[u] & CMASK];
#pragma unroll
        for (int u = 0; u < UNR; u++) lds[ss[u]] = vv[u] * xx[u];
      }
      for (; k < nb; k += WAVE) {
        const int32_t cw = ldg<int32_t, NT>(c + k);
        lds[SORTED == 2 ? (int)((uint32_t)cw >> 22) : SORTED == 1 ? (int)sl[k] : k] = ldg<double, NT>(v + k) * x[cw & CMASK];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (nr > 16) {
        // phase 2: one lane per row, column order
        for (int32_t r = r0 + lane; r < r1; r += WAVE) {
          const int a = (int)A.roff[r];
          const int e = (r + 1 < r1) ? (int)A.roff[r + 1] : nb;
          double s = 0.0;
          for (int j = a; j < e; j++) s += lds[j];
          epi(r, s);
        }
      } else {
        // phase 2 for a few long rows: T = 64 / 2^ceil(log2 nr) lanes share a row
        int T = WAVE;
        while (WAVE / T < nr) T >>= 1;   // nr <= 16  ->  T >= 4
        const int g = lane / T, t = lane - g * T;
        const int32_t r = r0 + g;
        double s = 0.0;
        if (g < nr) {
          const int a = (int)A.roff[r];
          const int e = (r + 1 < r1) ? (int)A.roff[r + 1] : nb;
          for (int j = a + t; j < e; j += T) s += lds[j];
        }
        for (int off = T >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (g < nr && t == 0) epi(r, s);
      }
      __builtin_amdgcn_wave_barrier();   // LDS is reused by the next block of this wave
    } else {
      // long rows: the whole wave reduces one row at a time
      for (int32_t r = r0; r < r1; r++) {
        const int a = (int)A.roff[r];
        const int e = (r + 1 < r1) ? (int)A.roff[r + 1] : nb;
        // four independent (column, value, x) chains in flight per lane
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        int j = a + lane;
        for (; j + 3 * WAVE < e; j += 4 * WAVE) {
          const int32_t c0 = c[j] & CMASK, c1 = c[j + WAVE] & CMASK, c2 = c[j + 2 * WAVE] & CMASK, c3 = c[j + 3 * WAVE] & CMASK;
          const double v0 = v[j], v1 = v[j + WAVE], v2 = v[j + 2 * WAVE], v3 = v[j + 3 * WAVE];
          const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
          s0 = fma(v0, x0, s0); s1 = fma(v1, x1, s1); s2 = fma(v2, x2, s2); s3 = fma(v3, x3, s3);
        }
        for (; j < e; j += WAVE) s0 = fma(v[j], x[c[j] & CMASK], s0);
        double s = (s0 + s1) + (s2 + s3);
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (lane == 0) epi(r, s);
      }
    }
  }
}

template <int CAP, bool NT, int UNR, int SORTED, class X>
__global__ __launch_bounds__(BLOCK) void k_rb_spmv(RbView A, X x, SpmvEpi e) {
  __shared__ double lds_all[WAVES_PER_BLOCK][CAP];
  double* lds = lds_all[threadIdx.x >> 6];
  rb_rows<CAP, NT, UNR, SORTED>(A, x, lds, [&](int32_t row, double acc) {
    if (e.dscale) acc *= e.dscale[row];
    double o = e.alpha * acc;
    if (e.z) o += e.beta * e.z[row];
    if (e.w) o += e.gamma * e.w[row];
    e.out[row] = o;
    if (e.y2) {
      const double a = e.aux_out ? o : e.w2[row];
      e.y2[row] += e.delta * a;
    }
  });
}

// tuning variant (bit 0: 640-entry LDS stage instead of 1024, bit 1: non-temporal matrix loads,
// bit 2: 8 instead of 4 loads in flight); set through airhip_debug_set_variant
// AIRHIP_BLK=256: blocks of <= 255 + row length non-zeros with a 512-entry LDS stage (4 KB per wave ->
// twice the resident waves; rows longer than ~256 take the wave-per-row path).  Default 512 / 1024.
static int blk_from_env() {
  const char* e = std::getenv("AIRHIP_BLK");
  const int v = e ? std::atoi(e) : BLK_NNZ;
  return (v == 256 || v == 1024) ? v : BLK_NNZ;
}
int g_blk_nnz = blk_from_env();
// AIRHIP_CAP=768: 6-KB LDS stage per wave (6 instead of 5 workgroups per CU) with 512-nnz blocks
// AIRHIP_NT=1: non-temporal loads for the matrix stream (keeps L2 / Infinity Cache for the x vector)
static int g_nt = (std::getenv("AIRHIP_NT") && std::atoi(std::getenv("AIRHIP_NT")) != 0) ? 1 : 0;
static int g_cap768 = (std::getenv("AIRHIP_CAP") && std::atoi(std::getenv("AIRHIP_CAP")) == 768) ? 1 : 0;
int g_spmv_variant = std::getenv("AIRHIP_SPMV_VARIANT") ? std::atoi(std::getenv("AIRHIP_SPMV_VARIANT")) : 0;

void launch_spmv(const DevMat& A, const double* x, const SpmvEpi& e, hipStream_t s, const double* x_ghost,
                 const int32_t* blk_list, int64_t nlist) {
  if (A.m == 0 || (blk_list && nlist == 0)) return;
  if (!blk_list && g_spmv_variant == 0 && launch_spmv_tiled(A, x, e, s, x_ghost)) return;
  const dim3 grid(grid_for_blocks(blk_list ? nlist : A.nblocks)), block(BLOCK);
  RbView v = view_of(A);
  v.blk_list = blk_list; v.nlist = nlist;
  if (A.n_ghost > 0) {
    // distributed operand: ghost columns start at A.n - A.n_ghost
    if (!x_ghost) throw std::runtime_error("launch_spmv: distributed operand without ghost values");
    const int32_t nl = (int32_t)(A.n - A.n_ghost);
    const XVec<true> xv{x, x_ghost - nl, nl};
    if (A.packed) hipLaunchKernelGGL((k_rb_spmv<1024, false, 4, 2, XVec<true>>), grid, block, 0, s, v, xv, e);
    else if (A.slot) hipLaunchKernelGGL((k_rb_spmv<1024, false, 4, 1, XVec<true>>), grid, block, 0, s, v, xv, e);
    else hipLaunchKernelGGL((k_rb_spmv<1024, false, 4, 0, XVec<true>>), grid, block, 0, s, v, xv, e);
    return;
  }
  if (A.packed) {
    if (g_spmv_variant) throw std::runtime_error("kernel variants are not available for packed operators (set AIRHIP_NO_PACK=1)");
    hipLaunchKernelGGL((k_rb_spmv<1024, false, 4, 2, XVec<false>>), grid, block, 0, s, v, XVec<false>{x, nullptr, 0}, e);
    return;
  }
  if (g_spmv_variant & 8) {
    const XVecNoGather xd{x};
    hipLaunchKernelGGL((k_rb_spmv<1024, false, 4, 0, XVecNoGather>), grid, block, 0, s, v, xd, e);
    return;
  }
  const XVec<false> xv{x, nullptr, 0};
  if (A.slot) {   // column-sorted blocks: the production kernel
    if (g_blk_nnz == 256) hipLaunchKernelGGL((k_rb_spmv<512, false, 4, 1, XVec<false>>), grid, block, 0, s, v, xv, e);
    else if (g_blk_nnz == 1024) hipLaunchKernelGGL((k_rb_spmv<2048, false, 8, 1, XVec<false>>), grid, block, 0, s, v, xv, e);
    else if (g_cap768) hipLaunchKernelGGL((k_rb_spmv<768, false, 4, 1, XVec<false>>), grid, block, 0, s, v, xv, e);
    else if (g_nt) hipLaunchKernelGGL((k_rb_spmv<1024, true, 4, 1, XVec<false>>), grid, block, 0, s, v, xv, e);
    else hipLaunchKernelGGL((k_rb_spmv<1024, false, 4, 1, XVec<false>>), grid, block, 0, s, v, xv, e);
    return;
  }
#define RB_LAUNCH(CAP, NT, UNR) hipLaunchKernelGGL((k_rb_spmv<CAP, NT, UNR, 0, XVec<false>>), grid, block, 0, s, v, xv, e)
  switch (g_spmv_variant & 7) {
    case 0: RB_LAUNCH(1024, false, 4); break;
    case 1: RB_LAUNCH(640, false, 4); break;
    case 2: RB_LAUNCH(1024, true, 4); break;
    case 3: RB_LAUNCH(640, true, 4); break;
    case 4: RB_LAUNCH(1024, false, 8); break;
    case 5: RB_LAUNCH(640, false, 8); break;
    case 6: RB_LAUNCH(1024, true, 8); break;
    default: RB_LAUNCH(640, true, 8); break;
  }
#undef RB_LAUNCH
}

template <bool DIST, int SORTED, int CAP>
__global__ __launch_bounds__(BLOCK) void k_rb_fc_first(RbView A, const int32_t* __restrict__ wcol,
                                                      const double* __restrict__ b_f, XVec<DIST> x_c,
                                                      const double* __restrict__ aff_d, const double* __restrict__ minv,
                                                      double* __restrict__ t4_out, double* __restrict__ x_f) {
  __shared__ double lds_all[WAVES_PER_BLOCK][CAP];
  double* lds = lds_all[threadIdx.x >> 6];
  rb_rows<CAP, false, 4, SORTED>(A, x_c, lds, [&](int32_t row, double acc) {
    const double t4 = b_f[row] + (-1.0) * acc;
    const int32_t wc = wcol[row];
    const double x0 = wc >= 0 ? x_c[wc] : 0.0;
    if (aff_d) {
      // one F smooth with diagonal A_ff and diagonal inverse (src/FC_Smooth.F90:541-559)
      double t3 = aff_d[row] * x0;
      t3 = t4 + (-1.0) * t3;
      x_f[row] = x0 + minv[row] * t3;
    } else {
      t4_out[row] = t4;
      x_f[row] = x0;
    }
  });
}

void launch_fc_first(const DevMat& A_fc, const int32_t* wcol, const double* b_f, const double* x_c,
                     const double* aff_d, const double* minv, double* t4_out, double* x_f, hipStream_t s,
                     const double* x_ghost, const int32_t* blk_list, int64_t nlist) {
  if (A_fc.m == 0 || (blk_list && nlist == 0)) return;
  if (!blk_list && g_spmv_variant == 0 && launch_fc_first_tiled(A_fc, wcol, b_f, x_c, aff_d, minv, t4_out, x_f, s, x_ghost)) return;
  const dim3 grid(grid_for_blocks(blk_list ? nlist : A_fc.nblocks)), block(BLOCK);
  RbView vfc = view_of(A_fc);
  vfc.blk_list = blk_list; vfc.nlist = nlist;
  if (A_fc.n_ghost > 0) {
    if (!x_ghost) throw std::runtime_error("launch_fc_first: distributed operand without ghost values");
    const int32_t nl = (int32_t)(A_fc.n - A_fc.n_ghost);
    const XVec<true> xv{x_c, x_ghost - nl, nl};
    if (A_fc.packed) hipLaunchKernelGGL((k_rb_fc_first<true, 2, 1024>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else if (A_fc.slot) hipLaunchKernelGGL((k_rb_fc_first<true, 1, 1024>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else hipLaunchKernelGGL((k_rb_fc_first<true, 0, 1024>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
  } else {
    const XVec<false> xv{x_c, nullptr, 0};
    if (A_fc.packed) hipLaunchKernelGGL((k_rb_fc_first<false, 2, 1024>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else if (A_fc.slot && g_blk_nnz == 256) hipLaunchKernelGGL((k_rb_fc_first<false, 1, 512>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else if (A_fc.slot && g_cap768) hipLaunchKernelGGL((k_rb_fc_first<false, 1, 768>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else if (A_fc.slot && g_blk_nnz == 1024) hipLaunchKernelGGL((k_rb_fc_first<false, 1, 2048>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else if (A_fc.slot) hipLaunchKernelGGL((k_rb_fc_first<false, 1, 1024>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
    else hipLaunchKernelGGL((k_rb_fc_first<false, 0, 1024>), grid, block, 0, s, vfc, wcol, b_f, xv, aff_d, minv, t4_out, x_f);
  }
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
  int64_t* d_sums = nullptr;
  HIP_CHECK(hipMalloc(&d_sums, sizeof(int64_t) * (size_t)nchunks));
  hipLaunchKernelGGL(k_chunk_sum, dim3((unsigned)nchunks), dim3(BLOCK), 0, s, n, d_in, d_sums);
  std::vector<int64_t> h(nchunks);
  HIP_CHECK(hipMemcpyAsync(h.data(), d_sums, sizeof(int64_t) * (size_t)nchunks, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  int64_t run = 0;
  for (int64_t i = 0; i < nchunks; i++) { int64_t c = h[i]; h[i] = run; run += c; }
  HIP_CHECK(hipMemcpyAsync(d_sums, h.data(), sizeof(int64_t) * (size_t)nchunks, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_chunk_scan, dim3((unsigned)nchunks), dim3(BLOCK), 0, s, n, d_in, d_sums, d_out);
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d_sums);
  return run;
}

// copy (and remap) the entries of every destination row; one wave per 64 consecutive entries
// would need a row lookup, so rows are copied by lanes - setup only
__global__ __launch_bounds__(BLOCK) void k_copy_rows(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                    const double* __restrict__ v, const int32_t* __restrict__ rowsrc,
                                                    const int32_t* __restrict__ colmap, int32_t n_mapped,
                                                    const int64_t* __restrict__ nrp,
                                                    int32_t* __restrict__ ocol, double* __restrict__ oval) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  // a wave copies one row at a time with coalesced accesses
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

// flag[r] = 1 if row r starts a new block: its first non-zero lies in another BLK_NNZ bucket
// than the previous row's, or BLK_ROWS rows have passed (runs of empty rows, e.g. Dirichlet
// planes, must not pile up in one block)
__global__ __launch_bounds__(BLOCK) void k_block_flags(int64_t m, int blk_nnz, const int64_t* __restrict__ nrp, int64_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < m; r += stride)
    flag[r] = (r % BLK_ROWS == 0 || (nrp[r] / blk_nnz) != (nrp[r - 1] / blk_nnz)) ? 1 : 0;
}

__global__ __launch_bounds__(BLOCK) void k_block_build(int64_t m, const int64_t* __restrict__ nrp, const int64_t* __restrict__ flag,
                                                      const int64_t* __restrict__ blk_of, int32_t* __restrict__ blk_row,
                                                      int64_t* __restrict__ blk_p0) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < m; r += stride)
    if (flag[r]) { blk_row[blk_of[r]] = (int32_t)r; blk_p0[blk_of[r]] = nrp[r]; }
}

__global__ __launch_bounds__(BLOCK) void k_row_offsets(int64_t m, const int64_t* __restrict__ nrp, const int64_t* __restrict__ flag,
                                                      const int64_t* __restrict__ blk_of, const int64_t* __restrict__ blk_p0,
                                                      uint32_t* __restrict__ roff) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < m; r += stride) {
    const int64_t b = blk_of[r] + flag[r] - 1;   // exclusive scan of the flags: block of row r
    roff[r] = (uint32_t)(nrp[r] - blk_p0[b]);
  }
}

// Column-sort the entries of every block that takes the LDS path (4*rows < nnz <= LDS_CAP); other
// blocks are copied through with slot = identity.  One wave per block, bitonic sort of
// (column << 16 | position) keys in LDS.
__global__ __launch_bounds__(BLOCK) void k_block_sort(int64_t nblocks, const int32_t* __restrict__ blk_row, const int64_t* __restrict__ blk_p0,
                                                     const int32_t* __restrict__ col, const double* __restrict__ val,
                                                     int32_t* __restrict__ ocol, double* __restrict__ oval, uint16_t* __restrict__ slot,
                                                     int cap, int scalar_t) {
  __shared__ unsigned long long keys_all[WAVES_PER_BLOCK][2 * LDS_CAP];
  unsigned long long* keys = keys_all[threadIdx.x >> 6];
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    const int64_t p0 = blk_p0[b], p1 = blk_p0[b + 1];
    const int nb = (int)(p1 - p0);
    const int nr = blk_row[b + 1] - blk_row[b];
    if (p1 - p0 > cap || nb <= scalar_t * nr) {
      for (int64_t k = p0 + lane; k < p1; k += WAVE) { ocol[k] = col[k]; oval[k] = val[k]; slot[k] = (uint16_t)((k - p0) & 0xffff); }
      continue;
    }
    int P2 = WAVE; while (P2 < nb) P2 <<= 1;
    for (int i = lane; i < P2; i += WAVE)
      keys[i] = i < nb ? (((unsigned long long)(uint32_t)col[p0 + i]) << 16) | (unsigned long long)i : ~0ull;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int size = 2; size <= P2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = lane; i < P2 / 2; i += WAVE) {
          const int a = 2 * i - (i & (stride - 1));
          const int bb = a + stride;
          const bool asc = (a & size) == 0;
          const unsigned long long ka = keys[a], kb = keys[bb];
          if ((ka > kb) == asc) { keys[a] = kb; keys[bb] = ka; }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
    for (int i = lane; i < nb; i += WAVE) {
      const unsigned long long k = keys[i];
      const int pos = (int)(k & 0xffffull);
      ocol[p0 + i] = (int32_t)(k >> 16);
      oval[p0 + i] = val[p0 + pos];
      slot[p0 + i] = (uint16_t)pos;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// col |= slot << 22 for the entries of column-sorted blocks (operators with <= 2^22 columns)
__global__ __launch_bounds__(BLOCK) void k_pack_slot(int64_t nnz, int32_t* __restrict__ col, const uint16_t* __restrict__ slot) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t k = (int64_t)blockIdx.x * BLOCK + threadIdx.x; k < nnz; k += stride)
    col[k] = (int32_t)((uint32_t)col[k] | ((uint32_t)slot[k] << 22));
}

// blocks that were not sorted (longer than the LDS stage) keep slot = 0: their slot is never read
__global__ __launch_bounds__(BLOCK) void k_pack_unsorted_zero(int64_t nblocks, const int64_t* __restrict__ blk_p0, uint16_t* __restrict__ slot, int cap) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    const int64_t p0 = blk_p0[b], p1 = blk_p0[b + 1];
    if (p1 - p0 > cap) for (int64_t k = p0 + lane; k < p1; k += WAVE) slot[k] = 0;
  }
}

static bool pack_enabled() {
  static const bool on = !(std::getenv("AIRHIP_NO_PACK") && std::atoi(std::getenv("AIRHIP_NO_PACK")) != 0);
  return on;
}

// flag[b] = 1 if block b references a ghost column (column >= n_local); one wave per block
__global__ __launch_bounds__(BLOCK) void k_boundary_flags(int64_t nblocks, const int64_t* __restrict__ blk_p0, const int32_t* __restrict__ col,
                                                         int32_t n_local, int32_t cmask, uint8_t* __restrict__ flag) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t b = wave; b < nblocks; b += nwaves) {
    bool any = false;
    for (int64_t k = blk_p0[b] + lane; k < blk_p0[b + 1]; k += WAVE) any |= (col[k] & cmask) >= n_local;
    if (__any(any) && lane == 0) flag[b] = 1;
  }
}

void devmat_boundary_flags(const DevMat& A, std::vector<uint8_t>& flags, hipStream_t s) {
  flags.assign((size_t)A.nblocks, 0);
  if (A.nblocks == 0 || A.n_ghost == 0) return;
  uint8_t* d = nullptr;
  HIP_CHECK(hipMalloc(&d, (size_t)A.nblocks));
  HIP_CHECK(hipMemsetAsync(d, 0, (size_t)A.nblocks, s));
  const int g = (int)std::min<int64_t>((A.nblocks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 4096);
  // packed operators keep the column in the low 22 bits of the word (unsorted blocks: upper bits are zero)
  hipLaunchKernelGGL(k_boundary_flags, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, A.nblocks, A.blk_p0, A.col, (int32_t)(A.n - A.n_ghost),
                     A.packed ? 0x3fffff : -1, d);
  HIP_CHECK(hipMemcpyAsync(flags.data(), d, (size_t)A.nblocks, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d);
}

static bool sort_blocks_enabled() {
  static const bool on = !(std::getenv("AIRHIP_NO_BLOCK_SORT") && std::atoi(std::getenv("AIRHIP_NO_BLOCK_SORT")) != 0);
  return on;
}

static int grid1d(int64_t n) {
  int64_t b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

void devmat_from_csr_device(int64_t m, int64_t n, const int64_t* d_rp, const int32_t* d_ci, const double* d_v,
                            const int32_t* d_rowsrc, const int32_t* d_colmap, DevMat& out, hipStream_t s, int64_t n_ghost) {
  devmat_free(out);
  out.m = m; out.n = n; out.n_ghost = n_ghost;
  if (m == 0) return;
  if (m > 0x7fffffffll) throw std::runtime_error("matrix has too many rows for 32-bit row indices");
  int64_t *d_len = nullptr, *d_nrp = nullptr, *d_flag = nullptr, *d_blk_of = nullptr;
  HIP_CHECK(hipMalloc(&d_len, sizeof(int64_t) * (size_t)m));
  HIP_CHECK(hipMalloc(&d_nrp, sizeof(int64_t) * (size_t)(m + 1)));
  hipLaunchKernelGGL(k_row_len, dim3(grid1d(m)), dim3(BLOCK), 0, s, m, d_rp, d_rowsrc, d_len);
  out.nnz = device_exclusive_scan(m, d_len, d_nrp, s);
  HIP_CHECK(hipMemcpyAsync(d_nrp + m, &out.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMalloc(&out.col, sizeof(int32_t) * (size_t)(out.nnz > 0 ? out.nnz : 1)));
  HIP_CHECK(hipMalloc(&out.val, sizeof(double) * (size_t)(out.nnz > 0 ? out.nnz : 1)));
  hipLaunchKernelGGL(k_copy_rows, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, m, d_rp, d_ci, d_v, d_rowsrc, d_colmap,
                     (int32_t)(n - n_ghost), d_nrp, out.col, out.val);
  // row blocks
  d_flag = d_len;   // reuse
  HIP_CHECK(hipMalloc(&d_blk_of, sizeof(int64_t) * (size_t)m));
  hipLaunchKernelGGL(k_block_flags, dim3(grid1d(m)), dim3(BLOCK), 0, s, m, tiles_enabled() ? BLK_NNZ_TILED : g_blk_nnz, d_nrp, d_flag);
  out.nblocks = device_exclusive_scan(m, d_flag, d_blk_of, s);
  HIP_CHECK(hipMalloc(&out.blk_row, sizeof(int32_t) * (size_t)(out.nblocks + 1)));
  HIP_CHECK(hipMalloc(&out.blk_p0, sizeof(int64_t) * (size_t)(out.nblocks + 1)));
  HIP_CHECK(hipMalloc(&out.roff, sizeof(uint32_t) * (size_t)m));
  hipLaunchKernelGGL(k_block_build, dim3(grid1d(m)), dim3(BLOCK), 0, s, m, d_nrp, d_flag, d_blk_of, out.blk_row, out.blk_p0);
  const int32_t m32 = (int32_t)m;
  HIP_CHECK(hipMemcpyAsync(out.blk_row + out.nblocks, &m32, sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(out.blk_p0 + out.nblocks, &out.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(k_row_offsets, dim3(grid1d(m)), dim3(BLOCK), 0, s, m, d_nrp, d_flag, d_blk_of, out.blk_p0, out.roff);
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d_len); (void)hipFree(d_nrp); (void)hipFree(d_blk_of);
  if (sort_blocks_enabled() && !tiles_enabled() && out.nnz > 0) {
    int32_t* ncol = nullptr; double* nval = nullptr;
    HIP_CHECK(hipMalloc(&ncol, sizeof(int32_t) * (size_t)out.nnz));
    HIP_CHECK(hipMalloc(&nval, sizeof(double) * (size_t)out.nnz));
    HIP_CHECK(hipMalloc(&out.slot, sizeof(uint16_t) * (size_t)out.nnz));
    const int g = (int)std::min<int64_t>((out.nblocks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK, 4096);
    hipLaunchKernelGGL(k_block_sort, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, out.nblocks, out.blk_row, out.blk_p0, out.col, out.val, ncol, nval,
                       out.slot, g_blk_nnz == 256 ? 512 : (g_blk_nnz == 1024 ? 2 * LDS_CAP : (g_cap768 ? 768 : LDS_CAP)), scalar_threshold());
    HIP_CHECK(hipStreamSynchronize(s));
    (void)hipFree(out.col); (void)hipFree(out.val);
    out.col = ncol; out.val = nval;
    // unsorted blocks carry slot = position, which may exceed 10 bits: only default-shaped blocks are packed
    if (pack_enabled() && out.n <= (1 << 22) && g_blk_nnz == BLK_NNZ && !g_cap768 && scalar_threshold() == 0) {
      hipLaunchKernelGGL(k_pack_unsorted_zero, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, out.nblocks, out.blk_p0, out.slot, (int)LDS_CAP);
      hipLaunchKernelGGL(k_pack_slot, dim3(grid1d(out.nnz)), dim3(BLOCK), 0, s, out.nnz, out.col, out.slot);
      HIP_CHECK(hipStreamSynchronize(s));
      (void)hipFree(out.slot); out.slot = nullptr;
      out.packed = true;
    }
  }
  if (tiles_enabled()) devmat_build_tiles(out, s);
}

void devmat_free(DevMat& A) {
  if (A.blk_row) (void)hipFree(A.blk_row);
  if (A.blk_p0) (void)hipFree(A.blk_p0);
  if (A.roff) (void)hipFree(A.roff);
  if (A.col) (void)hipFree(A.col);
  if (A.val) (void)hipFree(A.val);
  if (A.slot) (void)hipFree(A.slot);
  if (A.list_int) (void)hipFree(A.list_int);
  if (A.list_bnd) (void)hipFree(A.list_bnd);
  if (A.tile_blk) (void)hipFree(A.tile_blk);
  if (A.tile_u0) (void)hipFree(A.tile_u0);
  if (A.ucol) (void)hipFree(A.ucol);
  if (A.lidx) (void)hipFree(A.lidx);
  A = DevMat();
}

}  // namespace airhip
