// Tiled row-block CSR SpMV: the x-vector entries a group of row blocks needs are staged in LDS.
//
// Measured on MI355X (profiles/): with one lane per non-zero, a 64-lane gather x[col]
// of these operators touches 20-33 distinct 128-B lines, and the per-CU vector cache processes
// lines, not bytes - the gather, not HBM, bounded the row-block kernel at ~2.5 TB/s while a flat
// stream of the same arrays runs at ~7 TB/s.  Consecutive rows share most of their columns, so:
//   * a TILE is the run of row blocks whose first non-zero lies in the same 2048-entry bucket
//     (<= ~2.5k non-zeros); the builder sorts the tile's columns, keeps the distinct ones
//     (ucol, nnz/distinct = 2-4 here) and stores for every non-zero the 16-bit position of its
//     column in that list (lidx);
//   * one WORKGROUP owns a tile: it gathers x[ucol[.]] once (sorted columns: ~10 per line) into
//     LDS, then each of its 4 waves processes row blocks as before (coalesced stream of values
//     and 2-byte indices, products staged in LDS, per-row sums in column order) reading x from
//     LDS.  Line touches per non-zero drop from ~0.5 to ~0.13 and the index stream from 4 to
//     2 (+ 4/reuse) bytes per non-zero;
//   * tiles that do not fit (a row longer than the sort buffer, or > TILE_CAP distinct columns)
//     are "direct": they gather from global memory through the 32-bit column array, which is kept.
#include "rb.hpp"
#include <cstdlib>
#include <stdexcept>
#include <vector>

namespace airhip {

// AIRHIP_TILE_MODE: 0 = untiled kernels with column-sorted blocks (default), 1 = one tile per row block, staged by the wave that owns the
// block (no workgroup barriers), 2 = one tile per ~2k non-zeros, staged by the workgroup
static int tile_mode() {
  static const int m = std::getenv("AIRHIP_TILE_MODE") ? std::atoi(std::getenv("AIRHIP_TILE_MODE"))
                       : 0;
  return m;
}
bool tiles_enabled() { return tile_mode() != 0; }

namespace {

struct RtView {
  RbView rb;
  int64_t ntiles;
  const int32_t* tile_blk;
  const int64_t* tile_u0;
  const int32_t* ucol;
  const uint16_t* lidx;
};
RtView tview(const DevMat& A) { return RtView{view_of(A), A.ntiles, A.tile_blk, A.tile_u0, A.ucol, A.lidx}; }

struct BlockRange { int64_t first, last, stride; };
// XCD-aware, grid-strided assignment of row blocks to waves (gridDim.x is a multiple of 8)
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
int block_grid(int64_t nblocks) {
  int64_t blocks = (nblocks + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  blocks = (blocks + 7) / 8 * 8;
  if (blocks < 8) blocks = 8;
  return (int)blocks;
}

int tile_grid(int64_t ntiles) {
  int64_t g = ntiles < 1 ? 1 : ntiles;
  if (g > 1024) g = 1024;             // 4 resident workgroups per CU x 256 CUs
  g = (g + 7) / 8 * 8;
  return (int)g;
}


// products -> LDS stage -> per-row sums in column order -> epilogue (shared by the tiled kernels)
template <int PF, int RC, class EpiT>
__device__ __forceinline__ void rw_finish(const RbView& A, const EpiT& epi, const typename EpiT::Pre& pre0, const double (&vv)[PF],
                                          const double (&xx)[PF], const uint32_t (&ro)[RC], double* __restrict__ lds, int32_t r0,
                                          int32_t r1, int nb, int nr, int lane) {
  using Pre = typename EpiT::Pre;
#pragma unroll
  for (int u = 0; u < PF; u++) { const int k = lane + u * WAVE; if (k < nb) lds[k] = vv[u] * xx[u]; }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (nr > 16) {
#pragma unroll
    for (int j = 0; j < RC; j++) {
      const int32_t r = r0 + lane + j * WAVE;
      uint32_t e = __shfl_down(ro[j], 1, WAVE);
      const uint32_t e0 = __shfl(ro[j + 1 < RC ? j + 1 : j], 0, WAVE);
      if (lane == WAVE - 1) e = e0;
      if (r < r1) {
        const int a = (int)ro[j];
        const int ee = (r + 1 < r1) ? (int)e : nb;
        double s = 0.0;
        for (int q = a; q < ee; q++) s += lds[q];
        if (j == 0) epi.post(r, s, pre0); else epi.post(r, s, epi.pre(r));
      }
    }
  } else {
    int TT = WAVE;
    while (WAVE / TT < nr) TT >>= 1;
    const int gg = lane / TT, tt = lane - gg * TT;
    const int a = (int)__shfl(ro[0], gg, WAVE);
    const int en = (int)__shfl(ro[0], gg + 1 < WAVE ? gg + 1 : gg, WAVE);
    const Pre pg = epi.shfl(pre0, gg);
    double s = 0.0;
    if (gg < nr) {
      const int e = (gg + 1 < nr) ? en : nb;
      for (int q = a + tt; q < e; q += TT) s += lds[q];
    }
    for (int off = TT >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
    if (gg < nr && tt == 0) epi.post(r0 + gg, s, pg);
  }
  __builtin_amdgcn_wave_barrier();    // the LDS stage is reused by the wave's next block
}

// ---------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------
// EpiT: struct with  Pre pre(row) const;  Pre shfl(Pre, src_lane) const;  void post(row, acc, Pre) const.
template <class X, class EpiT>
__device__ __forceinline__ void rt_rows(const RtView& T, const X x, double* __restrict__ xt, double* __restrict__ lds,
                                        const EpiT epi) {
  constexpr int PF = TILE_LDS / WAVE;       // 8 loads per lane cover the LDS stage
  constexpr int RC = BLK_ROWS / WAVE;
  using Pre = typename EpiT::Pre;
  const RbView& A = T.rb;
  const int tid = threadIdx.x;
  const int lane = tid & (WAVE - 1);
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t last_nz = A.nnz - 1;
  const int32_t last_row = (int32_t)(A.m - 1);
  // XCD-aware: the workgroups of one residue class walk a contiguous eighth of the tiles
  const int g = blockIdx.x & 7;
  const int64_t tlo = T.ntiles * g / 8, thi = T.ntiles * (g + 1) / 8;
  const int64_t tstride = gridDim.x >> 3;
  for (int64_t t = tlo + (blockIdx.x >> 3); t < thi; t += tstride) {
    int32_t b0, b1; int64_t u0, u1;
    {
      uint64_t bb, q0, q1;
      const int32_t* pb = T.tile_blk + t; const int64_t* pu = T.tile_u0 + t;
      asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dwordx2 %2, %4, 0x8\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(bb), "=&s"(q0), "=&s"(q1) : "s"(pb), "s"(pu) : "memory");
      b0 = (int32_t)(bb & 0xffffffffull); b1 = (int32_t)(bb >> 32); u0 = (int64_t)q0; u1 = (int64_t)q1;
    }
    const int U = (int)(u1 - u0);          // 0: direct tile
    __syncthreads();                       // the previous tile's x copy is no longer read
    for (int u = tid; u < U; u += BLOCK) xt[u] = x[T.ucol[u0 + u]];
    __syncthreads();
    for (int32_t b = b0 + wid; b < b1; b += WAVES_PER_BLOCK) {
      int32_t r0, r1; int64_t p0, p1;
      {
        uint64_t rr, q0, q1;
        const int32_t* pr = A.blk_row + b; const int64_t* pp = A.blk_p0 + b;
        asm volatile("s_load_dwordx2 %0, %3, 0x0\n\ts_load_dwordx2 %1, %4, 0x0\n\ts_load_dwordx2 %2, %4, 0x8\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(rr), "=&s"(q0), "=&s"(q1) : "s"(pr), "s"(pp) : "memory");
        r0 = (int32_t)(rr & 0xffffffffull); r1 = (int32_t)(rr >> 32); p0 = (int64_t)q0; p1 = (int64_t)q1;
      }
      const int nb = (int)(p1 - p0);
      const int nr = r1 - r0;
      const double* __restrict__ v = A.val + p0;
      if (nb <= TILE_LDS) {
        double vv[PF], xx[PF]; uint32_t ro[RC];
#pragma unroll
        for (int u = 0; u < PF; u++) vv[u] = A.val[min(p0 + lane + u * WAVE, last_nz)];
        if (U > 0) {
          uint16_t li[PF];
#pragma unroll
          for (int u = 0; u < PF; u++) li[u] = T.lidx[min(p0 + lane + u * WAVE, last_nz)];
#pragma unroll
          for (int j = 0; j < RC; j++) ro[j] = A.roff[min(r0 + lane + j * WAVE, last_row)];
#pragma unroll
          for (int u = 0; u < PF; u++) xx[u] = (lane + u * WAVE < nb) ? xt[li[u]] : 0.0;
        } else {
          int32_t cc[PF];
#pragma unroll
          for (int u = 0; u < PF; u++) cc[u] = A.col[min(p0 + lane + u * WAVE, last_nz)];
#pragma unroll
          for (int j = 0; j < RC; j++) ro[j] = A.roff[min(r0 + lane + j * WAVE, last_row)];
#pragma unroll
          for (int u = 0; u < PF; u++) xx[u] = x[cc[u]];
        }
        const Pre pre0 = epi.pre(min(r0 + lane, last_row));
#pragma unroll
        for (int u = 0; u < PF; u++) { const int k = lane + u * WAVE; if (k < nb) lds[k] = vv[u] * xx[u]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (nr > 16) {
#pragma unroll
          for (int j = 0; j < RC; j++) {
            const int32_t r = r0 + lane + j * WAVE;
            uint32_t e = __shfl_down(ro[j], 1, WAVE);
            const uint32_t e0 = __shfl(ro[j + 1 < RC ? j + 1 : j], 0, WAVE);
            if (lane == WAVE - 1) e = e0;
            if (r < r1) {
              const int a = (int)ro[j];
              const int ee = (r + 1 < r1) ? (int)e : nb;
              double s = 0.0;
              for (int q = a; q < ee; q++) s += lds[q];
              if (j == 0) epi.post(r, s, pre0); else epi.post(r, s, epi.pre(r));
            }
          }
        } else {
          int TT = WAVE;
          while (WAVE / TT < nr) TT >>= 1;
          const int gg = lane / TT, tt = lane - gg * TT;
          const int a = (int)__shfl(ro[0], gg, WAVE);
          const int en = (int)__shfl(ro[0], gg + 1 < WAVE ? gg + 1 : gg, WAVE);
          const Pre pg = epi.shfl(pre0, gg);
          double s = 0.0;
          if (gg < nr) {
            const int e = (gg + 1 < nr) ? en : nb;
            for (int q = a + tt; q < e; q += TT) s += lds[q];
          }
          for (int off = TT >> 1; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
          if (gg < nr && tt == 0) epi.post(r0 + gg, s, pg);
        }
        __builtin_amdgcn_wave_barrier();    // the LDS stage is reused by the wave's next block
      } else {
        // block holding a long row: the whole wave reduces one row at a time
        const uint16_t* __restrict__ li = T.lidx + p0;
        const int32_t* __restrict__ c = A.col + p0;
        for (int32_t r = r0; r < r1; r++) {
          const int a = (int)A.roff[r];
          const int e = (r + 1 < r1) ? (int)A.roff[r + 1] : nb;
          double s = 0.0;
          if (U > 0) { for (int q = a + lane; q < e; q += WAVE) s = fma(v[q], xt[li[q]], s); }
          else { for (int q = a + lane; q < e; q += WAVE) s = fma(v[q], x[c[q]], s); }
          for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
          if (lane == 0) epi.post(r, s, epi.pre(r));
        }
      }
    }
  }
}

struct SpmvEpiT {
  SpmvEpi e;
  struct Pre { double z, w, ds; };
  __device__ __forceinline__ Pre pre(int32_t row) const {
    Pre p;
    p.z = e.z ? e.z[row] : 0.0;
    p.w = e.w ? e.w[row] : 0.0;
    p.ds = e.dscale ? e.dscale[row] : 1.0;
    return p;
  }
  __device__ __forceinline__ Pre shfl(const Pre& p, int src) const {
    Pre q; q.z = __shfl(p.z, src, WAVE); q.w = __shfl(p.w, src, WAVE); q.ds = __shfl(p.ds, src, WAVE); return q;
  }
  __device__ __forceinline__ void post(int32_t row, double acc, const Pre& p) const {
    if (e.dscale) acc *= p.ds;
    double o = e.alpha * acc;
    if (e.z) o += e.beta * p.z;
    if (e.w) o += e.gamma * p.w;
    e.out[row] = o;
    if (e.y2) {
      const double a = e.aux_out ? o : e.w2[row];
      e.y2[row] += e.delta * a;
    }
  }
};

template <class X>
__global__ __launch_bounds__(BLOCK) void k_rt_spmv(RtView T, X x, SpmvEpi e) {
  __shared__ double xt[TILE_CAP];
  __shared__ double lds_all[WAVES_PER_BLOCK][TILE_LDS];
  rt_rows(T, x, xt, lds_all[threadIdx.x >> 6], SpmvEpiT{e});
}

// fused first stage of an F-point relaxation (see launch_fc_first in dev.hpp)
template <class X>
struct FcFirstEpiT {
  const int32_t* wcol; const double* b_f; X x_c; const double* aff_d; const double* minv; double* t4_out; double* x_f;
  struct Pre { double bf, x0, d, mi; };
  __device__ __forceinline__ Pre pre(int32_t row) const {
    Pre p;
    p.bf = b_f[row];
    const int32_t wc = wcol[row];
    p.x0 = wc >= 0 ? x_c[wc] : 0.0;
    p.d = aff_d ? aff_d[row] : 0.0;
    p.mi = aff_d ? minv[row] : 0.0;
    return p;
  }
  __device__ __forceinline__ Pre shfl(const Pre& p, int src) const {
    Pre q; q.bf = __shfl(p.bf, src, WAVE); q.x0 = __shfl(p.x0, src, WAVE); q.d = __shfl(p.d, src, WAVE); q.mi = __shfl(p.mi, src, WAVE);
    return q;
  }
  __device__ __forceinline__ void post(int32_t row, double acc, const Pre& p) const {
    const double t4 = p.bf + (-1.0) * acc;
    if (aff_d) {
      // one F smooth with diagonal A_ff and diagonal inverse (src/FC_Smooth.F90:541-559)
      double t3 = p.d * p.x0;
      t3 = t4 + (-1.0) * t3;
      x_f[row] = p.x0 + p.mi * t3;
    } else {
      t4_out[row] = t4;
      x_f[row] = p.x0;
    }
  }
};

template <class X>
__global__ __launch_bounds__(BLOCK) void k_rt_fc_first(RtView T, X x_c, const int32_t* __restrict__ wcol, const double* __restrict__ b_f,
                                                      const double* __restrict__ aff_d, const double* __restrict__ minv,
                                                      double* __restrict__ t4_out, double* __restrict__ x_f) {
  __shared__ double xt[TILE_CAP];
  __shared__ double lds_all[WAVES_PER_BLOCK][TILE_LDS];
  rt_rows(T, x_c, xt, lds_all[threadIdx.x >> 6], FcFirstEpiT<X>{wcol, b_f, x_c, aff_d, minv, t4_out, x_f});
}


// ---------------------------------------------------------------------------------------
// wave-level tiles: tile == row block.  One wave gathers the block's distinct columns (sorted:
// few lines per gather instruction) into its private LDS buffer, then forms the products from
// LDS.  buf: 1024 doubles per wave - x copy in [0,512) and products in [512,1024) for blocks of
// <= 512 non-zeros; x copy in [0,1024) for blocks of <= 1024 non-zeros that hold a long row.
// ---------------------------------------------------------------------------------------
constexpr int WT_LDS = 1024;
template <class X, class EpiT>
__device__ __forceinline__ void rw_rows(const RtView& T, const X x, double* __restrict__ buf, const EpiT epi) {
  constexpr int PF = 8;
  constexpr int RC = BLK_ROWS / WAVE;
  using Pre = typename EpiT::Pre;
  const RbView& A = T.rb;
  const int lane = threadIdx.x & (WAVE - 1);
  const BlockRange br = my_blocks(A.nblocks);
  const int64_t last_nz = A.nnz - 1;
  const int32_t last_row = (int32_t)(A.m - 1);
  double* __restrict__ xt = buf;
  double* __restrict__ lds = buf + 512;
  for (int64_t b = br.first; b < br.last; b += br.stride) {
    int32_t r0, r1; int64_t p0, p1, u0, u1;
    {
      uint64_t rr, q0, q1, w0, w1;
      const int32_t* pr = A.blk_row + b; const int64_t* pp = A.blk_p0 + b; const int64_t* pu = T.tile_u0 + b;
      asm volatile("s_load_dwordx2 %0, %5, 0x0\n\ts_load_dwordx2 %1, %6, 0x0\n\ts_load_dwordx2 %2, %6, 0x8\n\t"
                   "s_load_dwordx2 %3, %7, 0x0\n\ts_load_dwordx2 %4, %7, 0x8\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(rr), "=&s"(q0), "=&s"(q1), "=&s"(w0), "=&s"(w1) : "s"(pr), "s"(pp), "s"(pu) : "memory");
      r0 = (int32_t)(rr & 0xffffffffull); r1 = (int32_t)(rr >> 32); p0 = (int64_t)q0; p1 = (int64_t)q1;
      u0 = (int64_t)w0; u1 = (int64_t)w1;
    }
    const int nb = (int)(p1 - p0);
    const int nr = r1 - r0;
    const int U = (int)(u1 - u0);           // 0: direct block
    const double* __restrict__ v = A.val + p0;
    if (nb <= 512) {
      double vv[PF], xx[PF]; uint32_t ro[RC];
#pragma unroll
      for (int u = 0; u < PF; u++) vv[u] = A.val[min(p0 + lane + u * WAVE, last_nz)];
      if (U > 0) {
        uint16_t li[PF]; int32_t uc[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) li[u] = T.lidx[min(p0 + lane + u * WAVE, last_nz)];
#pragma unroll
        for (int u = 0; u < PF; u++) if (u * WAVE < U) uc[u] = T.ucol[min(u0 + lane + u * WAVE, u1 - 1)];
#pragma unroll
        for (int j = 0; j < RC; j++) ro[j] = A.roff[min(r0 + lane + j * WAVE, last_row)];
        const Pre pre_l = epi.pre(min(r0 + lane, last_row));
        double xg[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) if (u * WAVE < U) xg[u] = x[uc[u]];
#pragma unroll
        for (int u = 0; u < PF; u++) if (u * WAVE < U) { const int k = lane + u * WAVE; if (k < U) xt[k] = xg[u]; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int u = 0; u < PF; u++) xx[u] = (lane + u * WAVE < nb) ? xt[li[u]] : 0.0;
        rw_finish<PF, RC>(A, epi, pre_l, vv, xx, ro, lds, r0, r1, nb, nr, lane);
      } else {
        int32_t cc[PF];
#pragma unroll
        for (int u = 0; u < PF; u++) cc[u] = A.col[min(p0 + lane + u * WAVE, last_nz)];
#pragma unroll
        for (int j = 0; j < RC; j++) ro[j] = A.roff[min(r0 + lane + j * WAVE, last_row)];
        const Pre pre_l = epi.pre(min(r0 + lane, last_row));
#pragma unroll
        for (int u = 0; u < PF; u++) xx[u] = x[cc[u]];
        rw_finish<PF, RC>(A, epi, pre_l, vv, xx, ro, lds, r0, r1, nb, nr, lane);
      }
    } else {
      // block holding a long row: the whole wave reduces one row at a time
      const uint16_t* __restrict__ li = T.lidx + p0;
      const int32_t* __restrict__ c = A.col + p0;
      if (U > 0) {
        for (int k = lane; k < U; k += WAVE) xt[k] = x[T.ucol[u0 + k]];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
      for (int32_t r = r0; r < r1; r++) {
        const int a = (int)A.roff[r];
        const int e = (r + 1 < r1) ? (int)A.roff[r + 1] : nb;
        double s = 0.0;
        if (U > 0) { for (int q = a + lane; q < e; q += WAVE) s = fma(v[q], xt[li[q]], s); }
        else { for (int q = a + lane; q < e; q += WAVE) s = fma(v[q], x[c[q]], s); }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, WAVE);
        if (lane == 0) epi.post(r, s, epi.pre(r));
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <class X>
__global__ __launch_bounds__(BLOCK) void k_rw_spmv(RtView T, X x, SpmvEpi e) {
  __shared__ double buf_all[WAVES_PER_BLOCK][WT_LDS];
  rw_rows(T, x, buf_all[threadIdx.x >> 6], SpmvEpiT{e});
}

template <class X>
__global__ __launch_bounds__(BLOCK) void k_rw_fc_first(RtView T, X x_c, const int32_t* __restrict__ wcol, const double* __restrict__ b_f,
                                                      const double* __restrict__ aff_d, const double* __restrict__ minv,
                                                      double* __restrict__ t4_out, double* __restrict__ x_f) {
  __shared__ double buf_all[WAVES_PER_BLOCK][WT_LDS];
  rw_rows(T, x_c, buf_all[threadIdx.x >> 6], FcFirstEpiT<X>{wcol, b_f, x_c, aff_d, minv, t4_out, x_f});
}

// ---------------------------------------------------------------------------------------
// builder
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_tile_flags(int64_t nblocks, const int64_t* __restrict__ blk_p0, int per_block,
                                                     int64_t* __restrict__ flag) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x; b < nblocks; b += stride)
    flag[b] = (per_block || b == 0 || blk_p0[b] / TILE_NNZ != blk_p0[b - 1] / TILE_NNZ) ? 1 : 0;
}
__global__ __launch_bounds__(BLOCK) void k_tile_starts(int64_t nblocks, const int64_t* __restrict__ flag, const int64_t* __restrict__ tile_of,
                                                      int32_t* __restrict__ tile_blk) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t b = (int64_t)blockIdx.x * BLOCK + threadIdx.x; b < nblocks; b += stride)
    if (flag[b]) tile_blk[tile_of[b]] = (int32_t)b;
}

// One workgroup per tile: sort the tile's columns, count / emit the distinct ones and (FILL)
// the local index of every entry.  ucount[t] = 0 marks a direct tile.
template <bool FILL>
__global__ __launch_bounds__(BLOCK) void k_tile_unique(int64_t ntiles, const int32_t* __restrict__ tile_blk, const int64_t* __restrict__ blk_p0,
                                                      const int32_t* __restrict__ col, int64_t* __restrict__ ucount,
                                                      const int64_t* __restrict__ tile_u0, int32_t* __restrict__ ucol,
                                                      uint16_t* __restrict__ lidx, int sort_cap, int ucap) {
  __shared__ int32_t keys[TILE_SORT_CAP];
  __shared__ int32_t ukeys[TILE_CAP];
  __shared__ int scan[BLOCK];
  __shared__ int total_s;
  const int tid = threadIdx.x;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t p0 = blk_p0[tile_blk[t]], p1 = blk_p0[tile_blk[t + 1]];
    const int64_t n64 = p1 - p0;
    __syncthreads();
    if (n64 > sort_cap || n64 == 0) { if (!FILL && tid == 0) ucount[t] = 0; continue; }
    if (FILL && tile_u0[t + 1] == tile_u0[t]) continue;      // direct tile
    const int n = (int)n64;
    int P2 = BLOCK; while (P2 < n) P2 <<= 1;
    for (int i = tid; i < P2; i += BLOCK) keys[i] = i < n ? col[p0 + i] : 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= P2; size <<= 1)
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = tid; i < P2 / 2; i += BLOCK) {
          const int a = 2 * i - (i & (stride - 1));
          const int b = a + stride;
          const bool asc = (a & size) == 0;
          const int32_t ka = keys[a], kb = keys[b];
          if ((ka > kb) == asc) { keys[a] = kb; keys[b] = ka; }
        }
        __syncthreads();
      }
    // distinct keys: thread tid owns the contiguous segment [tid*per, (tid+1)*per)
    const int per = P2 / BLOCK;
    int cnt = 0;
    for (int j = 0; j < per; j++) {
      const int i = tid * per + j;
      if (i < n && (i == 0 || keys[i] != keys[i - 1])) cnt++;
    }
    scan[tid] = cnt;
    __syncthreads();
    if (tid == 0) { int run = 0; for (int q = 0; q < BLOCK; q++) { const int c = scan[q]; scan[q] = run; run += c; } total_s = run; }
    __syncthreads();
    const int Utot = total_s;
    if (!FILL) {
      if (tid == 0) ucount[t] = Utot > ucap ? 0 : Utot;
      continue;
    }
    int pos = scan[tid];
    const int64_t u0 = tile_u0[t];
    for (int j = 0; j < per; j++) {
      const int i = tid * per + j;
      if (i < n && (i == 0 || keys[i] != keys[i - 1])) { ukeys[pos] = keys[i]; ucol[u0 + pos] = keys[i]; pos++; }
    }
    __syncthreads();
    for (int k = tid; k < n; k += BLOCK) {
      const int32_t c = col[p0 + k];
      int lo = 0, hi = Utot - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (ukeys[mid] < c) lo = mid + 1; else hi = mid; }
      lidx[p0 + k] = (uint16_t)lo;
    }
  }
}

int grid1(int64_t n) {
  int64_t b = (n + BLOCK - 1) / BLOCK;
  if (b > MAX_BLOCKS) b = MAX_BLOCKS;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

void devmat_build_tiles(DevMat& A, hipStream_t s) {
  if (A.m == 0 || A.nnz == 0 || A.nblocks == 0) return;
  int64_t *d_flag = nullptr, *d_tile_of = nullptr, *d_ucount = nullptr;
  HIP_CHECK(hipMalloc(&d_flag, sizeof(int64_t) * (size_t)A.nblocks));
  HIP_CHECK(hipMalloc(&d_tile_of, sizeof(int64_t) * (size_t)A.nblocks));
  const bool wave_mode = tile_mode() == 1;
  const int sort_cap = wave_mode ? WT_LDS : TILE_SORT_CAP, ucap = wave_mode ? WT_LDS : TILE_CAP;
  hipLaunchKernelGGL(k_tile_flags, dim3(grid1(A.nblocks)), dim3(BLOCK), 0, s, A.nblocks, A.blk_p0, wave_mode ? 1 : 0, d_flag);
  A.ntiles = device_exclusive_scan(A.nblocks, d_flag, d_tile_of, s);
  HIP_CHECK(hipMalloc(&A.tile_blk, sizeof(int32_t) * (size_t)(A.ntiles + 1)));
  HIP_CHECK(hipMalloc(&A.tile_u0, sizeof(int64_t) * (size_t)(A.ntiles + 1)));
  hipLaunchKernelGGL(k_tile_starts, dim3(grid1(A.nblocks)), dim3(BLOCK), 0, s, A.nblocks, d_flag, d_tile_of, A.tile_blk);
  const int32_t nb32 = (int32_t)A.nblocks;
  HIP_CHECK(hipMemcpyAsync(A.tile_blk + A.ntiles, &nb32, sizeof(int32_t), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMalloc(&d_ucount, sizeof(int64_t) * (size_t)A.ntiles));
  const int g = (int)std::min<int64_t>(A.ntiles, 4096);
  hipLaunchKernelGGL(k_tile_unique<false>, dim3(g), dim3(BLOCK), 0, s, A.ntiles, A.tile_blk, A.blk_p0, A.col, d_ucount,
                     (const int64_t*)nullptr, (int32_t*)nullptr, (uint16_t*)nullptr, sort_cap, ucap);
  A.n_ucol = device_exclusive_scan(A.ntiles, d_ucount, A.tile_u0, s);
  HIP_CHECK(hipMemcpyAsync(A.tile_u0 + A.ntiles, &A.n_ucol, sizeof(int64_t), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMalloc(&A.ucol, sizeof(int32_t) * (size_t)std::max<int64_t>(A.n_ucol, 1)));
  HIP_CHECK(hipMalloc(&A.lidx, sizeof(uint16_t) * (size_t)A.nnz));
  hipLaunchKernelGGL(k_tile_unique<true>, dim3(g), dim3(BLOCK), 0, s, A.ntiles, A.tile_blk, A.blk_p0, A.col, d_ucount, A.tile_u0,
                     A.ucol, A.lidx, sort_cap, ucap);
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipFree(d_flag); (void)hipFree(d_tile_of); (void)hipFree(d_ucount);
}

// returns false when the matrix has no tiles (caller falls back to the untiled kernels)
bool launch_spmv_tiled(const DevMat& A, const double* x, const SpmvEpi& e, hipStream_t s, const double* x_ghost) {
  if (!A.tile_blk || A.ntiles == 0 || A.nnz == 0) return false;
  const bool wave_mode = tile_mode() == 1;
  const dim3 grid(wave_mode ? block_grid(A.nblocks) : tile_grid(A.ntiles)), block(BLOCK);
  if (A.n_ghost > 0) {
    if (!x_ghost) throw std::runtime_error("launch_spmv: distributed operand without ghost values");
    const int32_t nl = (int32_t)(A.n - A.n_ghost);
    const XVec<true> xv{x, x_ghost - nl, nl};
    if (wave_mode) hipLaunchKernelGGL(k_rw_spmv<XVec<true>>, grid, block, 0, s, tview(A), xv, e);
    else hipLaunchKernelGGL(k_rt_spmv<XVec<true>>, grid, block, 0, s, tview(A), xv, e);
  } else {
    const XVec<false> xv{x, nullptr, 0};
    if (wave_mode) hipLaunchKernelGGL(k_rw_spmv<XVec<false>>, grid, block, 0, s, tview(A), xv, e);
    else hipLaunchKernelGGL(k_rt_spmv<XVec<false>>, grid, block, 0, s, tview(A), xv, e);
  }
  return true;
}

bool launch_fc_first_tiled(const DevMat& A_fc, const int32_t* wcol, const double* b_f, const double* x_c, const double* aff_d,
                           const double* minv, double* t4_out, double* x_f, hipStream_t s, const double* x_ghost) {
  if (!A_fc.tile_blk || A_fc.ntiles == 0 || A_fc.nnz == 0) return false;
  const bool wave_mode = tile_mode() == 1;
  const dim3 grid(wave_mode ? block_grid(A_fc.nblocks) : tile_grid(A_fc.ntiles)), block(BLOCK);
  if (A_fc.n_ghost > 0) {
    if (!x_ghost) throw std::runtime_error("launch_fc_first: distributed operand without ghost values");
    const int32_t nl = (int32_t)(A_fc.n - A_fc.n_ghost);
    const XVec<true> xv{x_c, x_ghost - nl, nl};
    if (wave_mode) hipLaunchKernelGGL(k_rw_fc_first<XVec<true>>, grid, block, 0, s, tview(A_fc), xv, wcol, b_f, aff_d, minv, t4_out, x_f);
    else hipLaunchKernelGGL(k_rt_fc_first<XVec<true>>, grid, block, 0, s, tview(A_fc), xv, wcol, b_f, aff_d, minv, t4_out, x_f);
  } else {
    const XVec<false> xv{x_c, nullptr, 0};
    if (wave_mode) hipLaunchKernelGGL(k_rw_fc_first<XVec<false>>, grid, block, 0, s, tview(A_fc), xv, wcol, b_f, aff_d, minv, t4_out, x_f);
    else hipLaunchKernelGGL(k_rt_fc_first<XVec<false>>, grid, block, 0, s, tview(A_fc), xv, wcol, b_f, aff_d, minv, t4_out, x_f);
  }
  return true;
}

}  // namespace airhip
