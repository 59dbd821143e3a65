// Hand-written HIP kernels for the PCAIR solve phase on gfx950 (MI355X, wave64).
//
// Every kernel here is HBM-bandwidth bound (fp64 CSR-type SpMV: 2 flop per 12 B of matrix
// stream); none is shaped into a GEMM, MFMA is irrelevant.  What matters:
//   * coalesced streaming of the matrix: SELL-64 slices, column-major inside a slice;
//   * one wave owns one slice -> no cross-lane reduction, no LDS round trip;
//   * XCD-aware slice assignment: blocks b and b+8 share an XCD (observed round-robin
//     dispatch, a speed assumption only), so each residue class gets a contiguous range of
//     slices and each XCD's private L2 sees a contiguous window of the gathered vector;
//   * the vector updates of the reference (VecAXPY / VecAYPX / VecAXPBY /
//     VecPointwiseDivide: src/FC_Smooth.F90:537-556, src/Gmres_Poly.F90:1456-1480,
//     src/Gmres_Poly_Newton.F90:808-871) are fused into the SpMV epilogue.
#include "dev.hpp"
#include <stdexcept>

namespace airhip {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int MAX_BLOCKS = 2048;  // 256 CUs x 8 blocks

__device__ __forceinline__ int64_t uniform64(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (int64_t)(uint32_t)lo;
}

struct SliceRange { int64_t first, last, stride; };

// XCD-aware, grid-strided assignment of slices to waves (gridDim.x is a multiple of 8).
__device__ __forceinline__ SliceRange my_slices(int64_t nslices) {
  const int g = blockIdx.x & 7;
  const int bi = blockIdx.x >> 3;
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  SliceRange r;
  const int64_t lo = nslices * g / 8;
  r.last = nslices * (g + 1) / 8;
  r.stride = (int64_t)(gridDim.x >> 3) * WAVES_PER_BLOCK;
  r.first = lo + (int64_t)bi * WAVES_PER_BLOCK + wid;
  return r;
}

static int grid_for_slices(int64_t nslices) {
  int64_t blocks = (nslices + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK;
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  blocks = (blocks + 7) / 8 * 8;
  if (blocks < 8) blocks = 8;
  return (int)blocks;
}

struct SellView {
  int64_t m, nslices;
  const int64_t* slice_ptr;
  const int32_t* col;
  const double* val;
  const int32_t* rowidx;
};

static SellView view_of(const SellMat& A) { return SellView{A.m, A.nslices, A.slice_ptr, A.col, A.val, A.rowidx}; }

// partial row-sum of one slice for this lane: sum_k val[k][lane] * x[col[k][lane]], k ascending
__device__ __forceinline__ double slice_dot(const SellView& A, int64_t s, int lane, const double* __restrict__ x) {
  const int64_t base = uniform64(A.slice_ptr[s]);
  const int64_t end = uniform64(A.slice_ptr[s + 1]);
  const int width = (int)((end - base) >> 6);
  const double* __restrict__ v = A.val + base + lane;
  const int32_t* __restrict__ c = A.col + base + lane;
  double acc = 0.0;
  int k = 0;
  for (; k + 4 <= width; k += 4) {
    const int32_t c0 = c[(k + 0) * WAVE], c1 = c[(k + 1) * WAVE], c2 = c[(k + 2) * WAVE], c3 = c[(k + 3) * WAVE];
    const double v0 = v[(k + 0) * WAVE], v1 = v[(k + 1) * WAVE], v2 = v[(k + 2) * WAVE], v3 = v[(k + 3) * WAVE];
    const double x0 = x[c0], x1 = x[c1], x2 = x[c2], x3 = x[c3];
    acc = fma(v0, x0, acc); acc = fma(v1, x1, acc); acc = fma(v2, x2, acc); acc = fma(v3, x3, acc);
  }
  for (; k < width; k++) acc = fma(v[k * WAVE], x[c[k * WAVE]], acc);
  return acc;
}

// sum the T partial sums of a row (T consecutive lanes); every lane of the group gets the total
template <int T>
__device__ __forceinline__ double group_sum(double acc) {
#pragma unroll
  for (int off = T >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  return acc;
}

template <int T>
__global__ __launch_bounds__(BLOCK) void k_sell_spmv(SellView A, const double* __restrict__ x, SpmvEpi e) {
  constexpr int R = WAVE / T;
  const int lane = threadIdx.x & (WAVE - 1);
  const SliceRange r = my_slices(A.nslices);
  for (int64_t s = r.first; s < r.last; s += r.stride) {
    double acc = group_sum<T>(slice_dot(A, s, lane, x));
    if ((lane & (T - 1)) == 0) {
      const int32_t row = A.rowidx[s * R + lane / T];
      if (row >= 0) {
        if (e.dscale) acc *= e.dscale[row];
        double o = e.alpha * acc;
        if (e.z) o += e.beta * e.z[row];
        if (e.w) o += e.gamma * e.w[row];
        e.out[row] = o;
        if (e.y2) {
          const double a = e.aux_out ? o : e.w2[row];
          e.y2[row] += e.delta * a;
        }
      }
    }
  }
}

#define DISPATCH_T(T_, CALL)                      \
  switch (T_) {                                   \
    case 1: { constexpr int TT = 1; CALL; } break;   \
    case 2: { constexpr int TT = 2; CALL; } break;   \
    case 4: { constexpr int TT = 4; CALL; } break;   \
    case 8: { constexpr int TT = 8; CALL; } break;   \
    case 16: { constexpr int TT = 16; CALL; } break; \
    case 32: { constexpr int TT = 32; CALL; } break; \
    default: { constexpr int TT = 64; CALL; } break; \
  }

void launch_spmv(const SellMat& A, const double* x, const SpmvEpi& e, hipStream_t s) {
  if (A.m == 0) return;
  SellView v = view_of(A);
  const dim3 grid(grid_for_slices(A.nslices)), block(BLOCK);
  DISPATCH_T(A.T, hipLaunchKernelGGL(k_sell_spmv<TT>, grid, block, 0, s, v, x, e));
}

template <int T>
__global__ __launch_bounds__(BLOCK) void k_fc_first(SellView A, const int32_t* __restrict__ wcol,
                                                   const double* __restrict__ b_f, const double* __restrict__ x_c,
                                                   const double* __restrict__ aff_d, const double* __restrict__ minv,
                                                   double* __restrict__ t4_out, double* __restrict__ x_f) {
  constexpr int R = WAVE / T;
  const int lane = threadIdx.x & (WAVE - 1);
  const SliceRange r = my_slices(A.nslices);
  for (int64_t s = r.first; s < r.last; s += r.stride) {
    const double acc = group_sum<T>(slice_dot(A, s, lane, x_c));
    if ((lane & (T - 1)) != 0) continue;
    const int32_t row = A.rowidx[s * R + lane / T];
    if (row >= 0) {
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
    }
  }
}

void launch_fc_first(const SellMat& A_fc, const int32_t* wcol, const double* b_f, const double* x_c,
                     const double* aff_d, const double* minv, double* t4_out, double* x_f, hipStream_t s) {
  if (A_fc.m == 0) return;
  SellView v = view_of(A_fc);
  const dim3 grid(grid_for_slices(A_fc.nslices)), block(BLOCK);
  DISPATCH_T(A_fc.T, hipLaunchKernelGGL(k_fc_first<TT>, grid, block, 0, s, v, wcol, b_f, x_c, aff_d, minv, t4_out, x_f));
}

// ---------------------------------------------------------------------------------------
// Vector kernels
// ---------------------------------------------------------------------------------------
static int grid_for_n(int64_t n, int per_thread = 1) {
  int64_t blocks = (n + (int64_t)BLOCK * per_thread - 1) / ((int64_t)BLOCK * per_thread);
  if (blocks > MAX_BLOCKS) blocks = MAX_BLOCKS;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}

// x, y and out may alias each other (in-place updates)
__global__ __launch_bounds__(BLOCK) void k_axpby(int64_t n, double a, const double* x, double b,
                                                const double* y, double* out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double o = a * x[i];
    if (y) o += b * y[i];
    out[i] = o;
  }
}
void launch_axpby(int64_t n, double a, const double* x, double b, const double* y, double* out, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_axpby, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, a, x, b, y, out);
}

__global__ __launch_bounds__(BLOCK) void k_pmult(int64_t n, double alpha, const double* a, const double* b,
                                                double beta, const double* z, double* out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double o = alpha * (a[i] * b[i]);
    if (z) o += beta * z[i];
    out[i] = o;
  }
}
void launch_pointwise_mult(int64_t n, double alpha, const double* a, const double* b, double beta, const double* z, double* out, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_pmult, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, alpha, a, b, beta, z, out);
}

__global__ __launch_bounds__(BLOCK) void k_diag_resid_update(int64_t n, const double* __restrict__ d,
                                                            const double* __restrict__ minv,
                                                            const double* __restrict__ t4, double* __restrict__ x) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double t3 = d[i] * x[i];
    t3 = t4[i] + (-1.0) * t3;
    x[i] = x[i] + minv[i] * t3;
  }
}
void launch_diag_residual_update(int64_t n, const double* d, const double* minv, const double* t4, double* x, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_diag_resid_update, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, d, minv, t4, x);
}

__global__ __launch_bounds__(BLOCK) void k_gather(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ in,
                                                 double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const int32_t j = idx[i];
    out[i] = j >= 0 ? in[j] : 0.0;   // negative index: no entry (empty W row)
  }
}
void launch_gather(int64_t n, const int32_t* idx, const double* in, double* out, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_gather, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, idx, in, out);
}
__global__ __launch_bounds__(BLOCK) void k_scatter(int64_t n, const int32_t* __restrict__ idx, const double* __restrict__ in,
                                                  double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) out[idx[i]] = in[i];
}
void launch_scatter(int64_t n, const int32_t* idx, const double* in, double* out, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_scatter, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, idx, in, out);
}
__global__ __launch_bounds__(BLOCK) void k_fill(int64_t n, double v, double* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) out[i] = v;
}
void launch_fill(int64_t n, double v, double* out, hipStream_t s) {
  if (n == 0) return;
  hipLaunchKernelGGL(k_fill, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, v, out);
}

// ---------------------------------------------------------------------------------------
// Multi-dot / multi-axpy (outer Krylov: VecMDot + VecMAXPY of classical Gram-Schmidt)
// ---------------------------------------------------------------------------------------
constexpr int MDOT_BLOCKS = 1024;
int mdot_scratch_doubles(int nv) { return MDOT_BLOCKS * (nv > 0 ? nv : 1); }

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

template <int K>
__global__ __launch_bounds__(BLOCK) void k_mdot_partial(int64_t n, const double* __restrict__ w, const double* __restrict__ V,
                                                       int64_t ld, double* __restrict__ partial, int nv_total, int k0) {
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    const double wi = w[i];
#pragma unroll
    for (int k = 0; k < K; k++) acc[k] = fma(wi, V[(int64_t)(k0 + k) * ld + i], acc[k]);
  }
  __shared__ double red[WAVES_PER_BLOCK][K];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) {
    double v = wave_sum(acc[k]);
    if (lane == 0) red[wid][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double v = 0.0;
    for (int q = 0; q < WAVES_PER_BLOCK; q++) v += red[q][threadIdx.x];
    partial[(int64_t)blockIdx.x * nv_total + k0 + threadIdx.x] = v;
  }
}

__global__ void k_mdot_final(int nblocks, int nv, const double* __restrict__ partial, double* __restrict__ dots) {
  const int k = threadIdx.x;
  if (k < nv) {
    double v = 0.0;
    for (int b = 0; b < nblocks; b++) v += partial[(int64_t)b * nv + k];
    dots[k] = v;
  }
}

void launch_mdot(int64_t n, const double* w, const double* V, int64_t ld, int nv, double* scratch, double* dots, hipStream_t s) {
  if (nv <= 0) return;
  int blocks = grid_for_n(n, 4);
  if (blocks > MDOT_BLOCKS) blocks = MDOT_BLOCKS;
  int k0 = 0;
  while (k0 < nv) {
    int rem = nv - k0;
    if (rem >= 8) { hipLaunchKernelGGL(k_mdot_partial<8>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, scratch, nv, k0); k0 += 8; }
    else if (rem >= 4) { hipLaunchKernelGGL(k_mdot_partial<4>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, scratch, nv, k0); k0 += 4; }
    else if (rem >= 2) { hipLaunchKernelGGL(k_mdot_partial<2>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, scratch, nv, k0); k0 += 2; }
    else { hipLaunchKernelGGL(k_mdot_partial<1>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, scratch, nv, k0); k0 += 1; }
  }
  hipLaunchKernelGGL(k_mdot_final, dim3(1), dim3(64), 0, s, blocks, nv, scratch, dots);
}

template <int K>
__global__ __launch_bounds__(BLOCK) void k_maxpy(int64_t n, double* __restrict__ w, const double* __restrict__ V, int64_t ld,
                                                const double* __restrict__ coef, int k0) {
  double c[K];
#pragma unroll
  for (int k = 0; k < K; k++) c[k] = coef[k0 + k];
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
    double acc = w[i];
#pragma unroll
    for (int k = 0; k < K; k++) acc = fma(c[k], V[(int64_t)(k0 + k) * ld + i], acc);
    w[i] = acc;
  }
}

void launch_maxpy(int64_t n, double* w, const double* V, int64_t ld, int nv, const double* coef, hipStream_t s) {
  if (nv <= 0 || n == 0) return;
  const int blocks = grid_for_n(n);
  int k0 = 0;
  while (k0 < nv) {
    int rem = nv - k0;
    if (rem >= 8) { hipLaunchKernelGGL(k_maxpy<8>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, coef, k0); k0 += 8; }
    else if (rem >= 4) { hipLaunchKernelGGL(k_maxpy<4>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, coef, k0); k0 += 4; }
    else if (rem >= 2) { hipLaunchKernelGGL(k_maxpy<2>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, coef, k0); k0 += 2; }
    else { hipLaunchKernelGGL(k_maxpy<1>, dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, coef, k0); k0 += 1; }
  }
}

// ---------------------------------------------------------------------------------------
// CSR -> SELL-64-sigma conversion (setup time)
//   * rows are sorted by length (descending, stable) inside windows of SIGMA rows so that the
//     rows sharing a slice have similar lengths (padding 1.75x -> ~1.1x on AIR hierarchies);
//     rowidx records which row each slot holds, outputs are scattered through it;
//   * T lanes cooperate on a row (T chosen per matrix from the mean row length and the row
//     count), entries of a row are dealt round-robin to its lanes.
// ---------------------------------------------------------------------------------------
constexpr int SIGMA = 2048;

__global__ __launch_bounds__(BLOCK) void k_window_sort(int64_t m, const int64_t* __restrict__ rp,
                                                      const int32_t* __restrict__ rowsrc, int32_t* __restrict__ slot_row) {
  __shared__ uint32_t key[SIGMA];
  const int64_t w0 = (int64_t)blockIdx.x * SIGMA;
  constexpr uint32_t LMAX = (1u << 20) - 1;
  for (int i = threadIdx.x; i < SIGMA; i += BLOCK) {
    const int64_t row = w0 + i;
    uint32_t k = 0xFFFFFFFFu;
    if (row < m) {
      const int64_t src = rowsrc ? rowsrc[row] : row;
      int64_t len = rp[src + 1] - rp[src];
      if (len > LMAX) len = LMAX;
      k = ((LMAX - (uint32_t)len) << 11) | (uint32_t)i;
    }
    key[i] = k;
  }
  __syncthreads();
  for (int size = 2; size <= SIGMA; size <<= 1)
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int i = threadIdx.x; i < SIGMA / 2; i += BLOCK) {
        const int a = 2 * i - (i & (stride - 1));
        const int b = a + stride;
        const bool asc = (a & size) == 0;
        const uint32_t ka = key[a], kb = key[b];
        if ((ka > kb) == asc) { key[a] = kb; key[b] = ka; }
      }
      __syncthreads();
    }
  for (int i = threadIdx.x; i < SIGMA; i += BLOCK) {
    const uint32_t k = key[i];
    slot_row[w0 + i] = (k == 0xFFFFFFFFu) ? -1 : (int32_t)(w0 + (k & (SIGMA - 1)));
  }
}

__global__ __launch_bounds__(BLOCK) void k_slice_width(int T, int64_t nslices, const int64_t* __restrict__ rp,
                                                      const int32_t* __restrict__ rowsrc, const int32_t* __restrict__ slot_row,
                                                      int64_t* __restrict__ width) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t s = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int R = WAVE / T;
  const int32_t row = slot_row[s * R + lane / T];
  int len = 0;
  if (row >= 0) { const int64_t src = rowsrc ? rowsrc[row] : row; len = (int)((rp[src + 1] - rp[src] + T - 1) / T); }
  for (int off = 32; off > 0; off >>= 1) len = max(len, __shfl_xor(len, off, WAVE));
  if (lane == 0) width[s] = len;
}

__global__ __launch_bounds__(BLOCK) void k_sell_fill(int T, int64_t nslices, const int64_t* __restrict__ rp,
                                                    const int32_t* __restrict__ ci, const double* __restrict__ v,
                                                    const int32_t* __restrict__ rowsrc, const int32_t* __restrict__ colmap,
                                                    const int32_t* __restrict__ slot_row,
                                                    const int64_t* __restrict__ slice_ptr, int32_t* __restrict__ ocol,
                                                    double* __restrict__ oval) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t s = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (s >= nslices) return;
  const int64_t base = slice_ptr[s];
  const int width = (int)((slice_ptr[s + 1] - base) >> 6);
  const int R = WAVE / T;
  const int t = lane & (T - 1);
  const int32_t row = slot_row[s * R + lane / T];
  int64_t a = 0, len = 0;
  if (row >= 0) { const int64_t src = rowsrc ? rowsrc[row] : row; a = rp[src]; len = rp[src + 1] - a; }
  int32_t padcol = 0;
  if (len > 0) { padcol = ci[a]; if (colmap) padcol = colmap[padcol]; }
  for (int k = 0; k < width; k++) {
    const int64_t e = (int64_t)k * T + t;
    int32_t c = padcol; double x = 0.0;
    if (e < len) { c = ci[a + e]; if (colmap) c = colmap[c]; x = v[a + e]; }
    ocol[base + (int64_t)k * WAVE + lane] = c;
    oval[base + (int64_t)k * WAVE + lane] = x;
  }
}

static int choose_T(int64_t m, int64_t nnz) {
  const double mean = m > 0 ? (double)nnz / (double)m : 0.0;
  int T = 1;
  while (T < 64 && mean / T > 8.0) T *= 2;
  // few rows: spread each row over more lanes so that the launch still fills the chip
  while (T < 64 && (m * T) / 64 < 4096 && mean / T > 2.0) T *= 2;
  return T;
}

void sell_from_csr_device(int64_t m, int64_t n, const int64_t* d_rp, const int32_t* d_ci, const double* d_v,
                          const int32_t* d_rowsrc, const int32_t* d_colmap, SellMat& out, hipStream_t s) {
  sell_free(out);
  out.m = m; out.n = n;
  if (m == 0) { out.nnz = 0; out.padded = 0; out.nslices = 0; return; }
  int64_t nnz_end = 0;
  HIP_CHECK(hipMemcpy(&nnz_end, d_rp + m, sizeof(int64_t), hipMemcpyDeviceToHost));
  out.nnz = nnz_end;   // rowsrc is a permutation of the rows, so the true nnz is rp[m] either way
  out.T = choose_T(m, nnz_end);
  const int R = WAVE / out.T;
  const int64_t nwin = (m + SIGMA - 1) / SIGMA;
  const int64_t nslots = nwin * SIGMA;
  out.nslices = nslots / R;
  HIP_CHECK(hipMalloc(&out.rowidx, sizeof(int32_t) * (size_t)nslots));
  hipLaunchKernelGGL(k_window_sort, dim3((unsigned)nwin), dim3(BLOCK), 0, s, m, d_rp, d_rowsrc, out.rowidx);
  int64_t* d_width = nullptr;
  HIP_CHECK(hipMalloc(&d_width, sizeof(int64_t) * (out.nslices + 1)));
  const int blocks = (int)((out.nslices + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  hipLaunchKernelGGL(k_slice_width, dim3(blocks), dim3(BLOCK), 0, s, out.T, out.nslices, d_rp, d_rowsrc, out.rowidx, d_width);
  std::vector<int64_t> w(out.nslices + 1);
  HIP_CHECK(hipMemcpyAsync(w.data(), d_width, sizeof(int64_t) * out.nslices, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  int64_t run = 0;
  for (int64_t i = 0; i < out.nslices; i++) { int64_t c = w[i] * WAVE; w[i] = run; run += c; }
  w[out.nslices] = run;
  out.padded = run;
  out.slice_ptr = d_width;
  HIP_CHECK(hipMemcpyAsync(out.slice_ptr, w.data(), sizeof(int64_t) * (out.nslices + 1), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMalloc(&out.col, sizeof(int32_t) * (size_t)(run > 0 ? run : 1)));
  HIP_CHECK(hipMalloc(&out.val, sizeof(double) * (size_t)(run > 0 ? run : 1)));
  hipLaunchKernelGGL(k_sell_fill, dim3(blocks), dim3(BLOCK), 0, s, out.T, out.nslices, d_rp, d_ci, d_v, d_rowsrc, d_colmap,
                     out.rowidx, out.slice_ptr, out.col, out.val);
  HIP_CHECK(hipStreamSynchronize(s));
}

void sell_free(SellMat& A) {
  if (A.slice_ptr) (void)hipFree(A.slice_ptr);
  if (A.col) (void)hipFree(A.col);
  if (A.val) (void)hipFree(A.val);
  if (A.rowidx) (void)hipFree(A.rowidx);
  A = SellMat();
}

}  // namespace airhip
