// Vector kernels of the PCAIR solve phase on gfx950 (MI355X, wave64): AXPBY-type updates that
// could not be fused into an SpMV epilogue, index gathers, and the multi-dot / multi-axpy of
// the outer Krylov method.  All of them are HBM-bandwidth bound streaming kernels.
// (The SpMV kernels, which dominate the V-cycle, live in spmv.hip.)
#include "dev.hpp"
#include "kcommon.hpp"
#include <stdexcept>
#include <string>

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
// ---- the natural <-> hierarchical permutation of a level-1 vector, chunk by chunk of the locality key.
// A hierarchical vector is a row of sections [F_1 | F_2 | ... | coarsest], each in key order, so a plain gather over the
// positions walks the natural vector once PER SECTION and fetches every natural line again for each section that holds one of
// its points (1.8 TB/s on 20 bytes per element at 464^3).  Here a workgroup takes one chunk of the key range - in every section
// the positions whose keys fall into it, starts[c * nsec + s] ... starts[(c + 1) * nsec + s] - so the natural lines a chunk
// touches are used up within one workgroup's lifetime.  iperm[p] = natural index of hierarchical position p.
// k_keys_at: key_h[p] = key[iperm[p]]; k_chunk_starts: the first position of section s whose key is >= the key at position
// c * chunk of section 0 (binary search; row nchunks = the sections' ends).
__global__ __launch_bounds__(BLOCK) void k_keys_at(int64_t n, const int32_t* __restrict__ iperm, const int64_t* __restrict__ key, int64_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t p = (int64_t)blockIdx.x * BLOCK + threadIdx.x; p < n; p += stride) out[p] = key[iperm[p]];
}
__global__ __launch_bounds__(BLOCK) void k_chunk_starts(int64_t nchunks, int nsec, int64_t chunk, const int64_t* __restrict__ sec_off, const int64_t* __restrict__ key_h,
                                                       int32_t* __restrict__ starts) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK, total = (nchunks + 1) * nsec;
  for (int64_t t = (int64_t)blockIdx.x * BLOCK + threadIdx.x; t < total; t += stride) {
    const int64_t c = t / nsec; const int s = (int)(t % nsec);
    int64_t lo = sec_off[s], hi = sec_off[s + 1];
    if (c == nchunks) { starts[t] = (int32_t)hi; continue; }
    if (c == 0) { starts[t] = (int32_t)lo; continue; }
    const int64_t k = key_h[sec_off[0] + c * chunk];
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (key_h[mid] < k) lo = mid + 1; else hi = mid; }
    starts[t] = (int32_t)lo;
  }
}
template <bool TO_NATURAL>
__global__ __launch_bounds__(BLOCK) void k_permute_chunks(int nsec, const int32_t* __restrict__ starts, const int32_t* __restrict__ iperm,
                                                         const double* __restrict__ in, double* __restrict__ out) {
  const int32_t* s0 = starts + (int64_t)blockIdx.x * nsec;
  for (int s = 0; s < nsec; s++) {
    const int32_t a = s0[s], e = s0[nsec + s];
    for (int32_t p = a + (int32_t)threadIdx.x; p < e; p += BLOCK) {
      const int32_t nat = iperm[p];
      if (TO_NATURAL) out[nat] = in[p]; else out[p] = in[nat];
    }
  }
}
void launch_keys_at(int64_t n, const int32_t* iperm, const int64_t* key, int64_t* out, hipStream_t s) {
  if (n) hipLaunchKernelGGL(k_keys_at, dim3(grid_for_n(n)), dim3(BLOCK), 0, s, n, iperm, key, out);
}
void launch_chunk_starts(int64_t nchunks, int nsec, int64_t chunk, const int64_t* sec_off, const int64_t* key_h, int32_t* starts, hipStream_t s) {
  hipLaunchKernelGGL(k_chunk_starts, dim3(grid_for_n((nchunks + 1) * nsec)), dim3(BLOCK), 0, s, nchunks, nsec, chunk, sec_off, key_h, starts);
}
void launch_permute_chunks(int64_t nchunks, int nsec, const int32_t* starts, const int32_t* iperm, const double* in, double* out, bool to_natural, hipStream_t s) {
  if (nchunks == 0) return;
  if (to_natural) hipLaunchKernelGGL((k_permute_chunks<true>), dim3((unsigned)nchunks), dim3(BLOCK), 0, s, nsec, starts, iperm, in, out);
  else hipLaunchKernelGGL((k_permute_chunks<false>), dim3((unsigned)nchunks), dim3(BLOCK), 0, s, nsec, starts, iperm, in, out);
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
// Multi-dot / multi-axpy (outer Krylov: VecMDot + VecMAXPY of classical Gram-Schmidt; the setup's Arnoldi)
//
// One pass each: every vector is read exactly once per call.  Round 3 launched the basis in slices of 8 / 4 / 2 / 1
// vectors, each slice reading w (0.8 GB at 464^3) again, took the norm of the orthogonalised vector in a pass of its own
// and read 8 bytes per lane per load.  Here one kernel instance per number of vectors (1 ... 32, compile time, so that
// all loads of an element pair are in flight together - guards around single vectors would serialise them) reads
// 16 bytes per lane per load; the multi-axpy optionally leaves the partial sums of squares of its result behind, so
// the norm that classical Gram-Schmidt needs next costs no pass.  Reductions are two-stage and deterministic (fixed
// grid, fixed order).
// ---------------------------------------------------------------------------------------
constexpr int MDOT_BLOCKS = 2048;          // 256 CUs x 8 workgroups
constexpr int MDOT_MAXV = 32;              // vectors per launch
int mdot_scratch_doubles(int nv) { return MDOT_BLOCKS * (nv > 0 ? nv : 1); }

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  return v;
}

// block-level sums of K per-thread values -> partial[blockIdx * stride + k0 + k]
template <int K>
__device__ __forceinline__ void block_sums(const double (&acc)[K], double* __restrict__ partial, int stride, int k0) {
  __shared__ double red[WAVES_PER_BLOCK][K];
  const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; k++) {
    const double v = wave_sum(acc[k]);
    if (lane == 0) red[wid][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < K) {
    double v = 0.0;
    for (int q = 0; q < WAVES_PER_BLOCK; q++) v += red[q][threadIdx.x];
    partial[(int64_t)blockIdx.x * stride + k0 + threadIdx.x] = v;
  }
}

// partial[b * nv_total + k0 + k] = sum over block b's elements of w_i * V_{k0+k, i}.  VEC: two elements per lane and
// load (n even, every vector 16-byte aligned)
template <int K, bool VEC>
__global__ __launch_bounds__(BLOCK) void k_mdot_partial(int64_t n, const double* __restrict__ w, const double* __restrict__ V,
                                                       int64_t ld, double* __restrict__ partial, int nv_total, int k0) {
  double acc[K];
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = 0.0;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  const double* __restrict__ V0 = V + (int64_t)k0 * ld;
  if constexpr (VEC) {
    const int64_t n2 = n >> 1;
    const double2* __restrict__ w2 = reinterpret_cast<const double2*>(w);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += stride) {
      const double2 wi = w2[i];
      double2 v[K];
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const double2*>(V0 + (int64_t)k * ld)[i];
#pragma unroll
      for (int k = 0; k < K; k++) acc[k] = fma(wi.y, v[k].y, fma(wi.x, v[k].x, acc[k]));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
      const double wi = w[i];
#pragma unroll
      for (int k = 0; k < K; k++) acc[k] = fma(wi, V0[(int64_t)k * ld + i], acc[k]);
    }
  }
  block_sums<K>(acc, partial, nv_total, k0);
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

extern int g_stream_blocks_limit();
static int stream_blocks(int64_t n) {
  int blocks = grid_for_n(n, 2);
  const int lim = g_stream_blocks_limit();
  return blocks > lim ? lim : blocks;
}
static bool vec2_ok(int64_t n, int64_t ld, const void* a, const void* b) {
  return (n & 1) == 0 && (ld & 1) == 0 && (((uintptr_t)a | (uintptr_t)b) & 15) == 0;
}

template <bool VEC>
static void mdot_dispatch(int k, int blocks, hipStream_t s, int64_t n, const double* w, const double* V, int64_t ld, double* scratch, int nv, int k0) {
  switch (k) {
#define C(K) case K: hipLaunchKernelGGL((k_mdot_partial<K, VEC>), dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, scratch, nv, k0); break;
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
    C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32)
#undef C
  }
}

void launch_mdot(int64_t n, const double* w, const double* V, int64_t ld, int nv, double* scratch, double* dots, hipStream_t s) {
  if (nv <= 0) return;
  const int blocks = stream_blocks(n);
  const bool vec = vec2_ok(n, ld, w, V);
  for (int k0 = 0; k0 < nv; k0 += MDOT_MAXV) {
    const int k = nv - k0 < MDOT_MAXV ? nv - k0 : MDOT_MAXV;
    if (vec) mdot_dispatch<true>(k, blocks, s, n, w, V, ld, scratch, nv, k0);
    else mdot_dispatch<false>(k, blocks, s, n, w, V, ld, scratch, nv, k0);
  }
  hipLaunchKernelGGL(k_mdot_final, dim3(1), dim3(BLOCK), 0, s, blocks, nv, scratch, dots);
}

// w_i <- (ZERO ? 0 : w_i) + sum_k coef[k0 + k] V_{k0+k, i} (k ascending: the order of a sequence of axpys);
// NORM: the block's sum of squares of the result goes to partial[blockIdx]
template <int K, bool VEC, bool ZERO, bool NORM>
__global__ __launch_bounds__(BLOCK) void k_maxpy(int64_t n, double* __restrict__ w, const double* __restrict__ V, int64_t ld,
                                                const double* __restrict__ coef, int k0, double* __restrict__ partial) {
  double c[K];
#pragma unroll
  for (int k = 0; k < K; k++) c[k] = coef[k0 + k];
  const double* __restrict__ V0 = V + (int64_t)k0 * ld;
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  double ss[1] = {0.0};
  if constexpr (VEC) {
    const int64_t n2 = n >> 1;
    double2* __restrict__ w2 = reinterpret_cast<double2*>(w);
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n2; i += stride) {
      double2 a = ZERO ? make_double2(0.0, 0.0) : w2[i];
      double2 v[K];
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = reinterpret_cast<const double2*>(V0 + (int64_t)k * ld)[i];
#pragma unroll
      for (int k = 0; k < K; k++) { a.x = fma(c[k], v[k].x, a.x); a.y = fma(c[k], v[k].y, a.y); }
      w2[i] = a;
      if constexpr (NORM) ss[0] = fma(a.y, a.y, fma(a.x, a.x, ss[0]));
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < n; i += stride) {
      double a = ZERO ? 0.0 : w[i];
#pragma unroll
      for (int k = 0; k < K; k++) a = fma(c[k], V0[(int64_t)k * ld + i], a);
      w[i] = a;
      if constexpr (NORM) ss[0] = fma(a, a, ss[0]);
    }
  }
  if constexpr (NORM) block_sums<1>(ss, partial, 1, 0);
}

// Workgroups of the multi-dot / multi-axpy launches (airhip_set_tuning "krylov_blocks").  Measured at 464^3 over the 20 timed
// iterations of bench.py (profiles/r04_krylov_blocks.txt): 2048 workgroups (8 per CU) multi-dot 20.0 ms / multi-axpy 27.1 ms,
// 256 ... 768 workgroups 17.7 / 23.9-24.5 ms, 128 workgroups 26.0 / 28.3 ms - with K + 1 streams a gigabyte apart, fewer waves
// walking them keep the DRAM pages open longer; non-temporal stores of the result changed nothing.
static int g_stream_blocks = 512;
int kernels_set_tuning(const char* key, int value) {
  const std::string k(key);
  if (k == "krylov_blocks") { g_stream_blocks = value < 64 ? 64 : (value > MDOT_BLOCKS ? MDOT_BLOCKS : value); return 0; }
  return -1;
}
int g_stream_blocks_limit() { return g_stream_blocks; }
template <bool VEC, bool ZERO, bool NORM>
static void maxpy_dispatch(int k, int blocks, hipStream_t s, int64_t n, double* w, const double* V, int64_t ld, const double* coef, int k0, double* partial) {
  switch (k) {
#define C(K) case K: hipLaunchKernelGGL((k_maxpy<K, VEC, ZERO, NORM>), dim3(blocks), dim3(BLOCK), 0, s, n, w, V, ld, coef, k0, partial); break;
    C(1) C(2) C(3) C(4) C(5) C(6) C(7) C(8) C(9) C(10) C(11) C(12) C(13) C(14) C(15) C(16)
    C(17) C(18) C(19) C(20) C(21) C(22) C(23) C(24) C(25) C(26) C(27) C(28) C(29) C(30) C(31) C(32)
#undef C
  }
}

// w (+)= sum_k coef[k] V_k  (coef on device).  zero_first: w = sum ... (w is not read).  norm_scratch / norm_out (device,
// both or neither; scratch of mdot_scratch_doubles(1)): *norm_out = sum of squares of the result, at no extra pass.
void launch_maxpy(int64_t n, double* w, const double* V, int64_t ld, int nv, const double* coef, hipStream_t s,
                  bool zero_first, double* norm_scratch, double* norm_out) {
  if (n == 0) return;
  if (nv <= 0) {
    if (zero_first) launch_fill(n, 0.0, w, s);
    if (norm_out) launch_mdot(n, w, w, 0, 1, norm_scratch, norm_out, s);
    return;
  }
  const int blocks = stream_blocks(n);
  const bool vec = vec2_ok(n, ld, w, V);
  for (int k0 = 0; k0 < nv; k0 += MDOT_MAXV) {
    const int k = nv - k0 < MDOT_MAXV ? nv - k0 : MDOT_MAXV;
    const bool zero = zero_first && k0 == 0, norm = norm_out && k0 + k == nv;
    const int variant = (vec ? 4 : 0) | (zero ? 2 : 0) | (norm ? 1 : 0);
    switch (variant) {
      case 0: maxpy_dispatch<false, false, false>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      case 1: maxpy_dispatch<false, false, true>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      case 2: maxpy_dispatch<false, true, false>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      case 3: maxpy_dispatch<false, true, true>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      case 4: maxpy_dispatch<true, false, false>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      case 5: maxpy_dispatch<true, false, true>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      case 6: maxpy_dispatch<true, true, false>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
      default: maxpy_dispatch<true, true, true>(k, blocks, s, n, w, V, ld, coef, k0, norm_scratch); break;
    }
  }
  if (norm_out) hipLaunchKernelGGL(k_mdot_final, dim3(1), dim3(BLOCK), 0, s, blocks, 1, norm_scratch, norm_out);
}

}  // namespace airhip
