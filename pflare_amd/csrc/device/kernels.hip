// Vector kernels of the PCAIR solve phase on gfx950 (MI355X, wave64): AXPBY-type updates that
// could not be fused into an SpMV epilogue, index gathers, and the multi-dot / multi-axpy of
// the outer Krylov method.  All of them are HBM-bandwidth bound streaming kernels.
// (The SpMV kernels, which dominate the V-cycle, live in spmv.hip.)
#include "dev.hpp"
#include "kcommon.hpp"
#include <stdexcept>

namespace airhip {

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

// second stage: one wave per dot product, lanes stride over the block partials, fixed-order wave
// reduction (deterministic)
__global__ __launch_bounds__(BLOCK) void k_mdot_final(int nblocks, int nv, const double* __restrict__ partial, double* __restrict__ dots) {
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
  for (int k = wid; k < nv; k += WAVES_PER_BLOCK) {
    double v = 0.0;
    for (int b = lane; b < nblocks; b += WAVE) v += partial[(int64_t)b * nv + k];
    v = wave_sum(v);
    if (lane == 0) dots[k] = v;
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
  hipLaunchKernelGGL(k_mdot_final, dim3(1), dim3(BLOCK), 0, s, blocks, nv, scratch, dots);
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

}  // namespace airhip
