// Block-structured stream probe (gfx950): what limits a kernel of the shape of k_rb_spmv - one wave per block of 512
// (index word, value) pairs, four waves per workgroup, as many waves as blocks - when the stream really comes from HBM.
// Variants (compile-time): width of the value / index loads, one interleaved blob per block against two arrays,
// a dependent descriptor load in front, the x gather (window size = locality), the LDS staging + row sums.
//   hipcc -O3 --offload-arch=gfx950 tools/block_probe.hip -o /tmp/block_probe && /tmp/block_probe [GB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); std::exit(1); } } while (0)

constexpr int NNZ = 512, WAVE = 64, EPL = 8;

struct Desc { int64_t p0; int32_t row, cbase; };

struct Args {
  const uint32_t* idx; const double* val;     // two arrays (BLOB = 0)
  const char* blob;                            // BLOB = 1: per block [val 4 KB][idx 2 KB]
  const Desc* desc; const double* x; double* y; int64_t nblocks; int rows_per_block;
  int skew = 0;                                // class g starts g * skew blocks into its eighth (wraps around)
};

// VW: 2 = 8 x dwordx2 value loads, 4 = 4 x dwordx4 (pairs of rounds interleaved in memory)
// IW: 1 = 8 x dword index loads, 4 = 2 x dwordx4
template <int VW, int IW, int BLOB, int DESC, int GATHER, int LDS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_block(Args a) {
  __shared__ double lds_all[WAVES][NNZ + 2];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = blockIdx.x & 7;
  const int64_t bi = blockIdx.x >> 3;
  const int64_t lo = a.nblocks * g / 8, hi = a.nblocks * (g + 1) / 8;
  int64_t b = lo + bi * WAVES + wid;
  if (b >= hi) return;
  if (a.skew) b = lo + (b - lo + (int64_t)g * a.skew) % (hi - lo);
  int64_t p0 = b * NNZ;
  if (DESC) {
    const Desc __attribute__((address_space(4)))* dp = (const Desc __attribute__((address_space(4)))*)(uintptr_t)a.desc;
    p0 = dp[b].p0;
  }
  const uint32_t* ip; const double* vp;
  if (BLOB) { vp = (const double*)(a.blob + p0 * 12); ip = (const uint32_t*)(a.blob + p0 * 12 + NNZ * 8); }
  else { ip = a.idx + p0; vp = a.val + p0; }
  uint32_t ix[EPL]; double vv[EPL];
  if (IW == 1) {
#pragma unroll
    for (int u = 0; u < EPL; u++) ix[u] = ip[u * WAVE + lane];
  } else {
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const uint4 q = ((const uint4*)ip)[h * WAVE + lane];
      ix[4 * h] = q.x; ix[4 * h + 1] = q.y; ix[4 * h + 2] = q.z; ix[4 * h + 3] = q.w;
    }
  }
  if (VW == 2) {
#pragma unroll
    for (int u = 0; u < EPL; u++) vv[u] = vp[u * WAVE + lane];
  } else {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const double2 q = ((const double2*)vp)[i * WAVE + lane];
      vv[2 * i] = q.x; vv[2 * i + 1] = q.y;
    }
  }
  double acc = 0.0;
  if (GATHER) {
    double xx[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) xx[u] = *(const double*)((const char*)a.x + ((size_t)ix[u] << 3));
#pragma unroll
    for (int u = 0; u < EPL; u++) vv[u] *= xx[u];
  } else {
#pragma unroll
    for (int u = 0; u < EPL; u++) vv[u] += (double)ix[u];
  }
  if (LDS) {
    double* lds = lds_all[wid];
#pragma unroll
    for (int u = 0; u < EPL; u++) lds[u * WAVE + lane + 1] = vv[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int R = a.rows_per_block, L = NNZ / R;     // rows of equal length
    if (R > 32) { if (lane < R) { for (int j = 0; j < L; j++) acc += lds[lane * L + j + 1]; a.y[b * R + lane] = acc; } }
    else {
      const int T = 64 / R, gq = lane / T, t = lane % T;          // R is a power of two here
      for (int j = t; j < L; j += T) acc += lds[gq * L + j + 1];
      for (int off = T >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
      const double srow = __shfl(acc, (lane * T) & 63, 64);
      if (lane < R) a.y[b * R + lane] = srow;
    }
  } else {
#pragma unroll
    for (int u = 0; u < EPL; u++) acc += vv[u];
    a.y[b * 64 + lane] = acc;            // every lane stores: no load is dead
  }
}


// NB consecutive blocks per wave; the stream of block i+1 is requested before block i is processed (NB = 1: the
// one-block-per-wave structure again, same code path as a cross-check).  val 4 x dwordx4, idx 8 x dword, two arrays.
template <int NB, int IW, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_multi(Args a) {
  __shared__ double lds_all[WAVES][NNZ + 2];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int g = blockIdx.x & 7;
  const int64_t bi = blockIdx.x >> 3;
  const int64_t nsuper = a.nblocks / NB;
  const int64_t lo = nsuper * g / 8, hi = nsuper * (g + 1) / 8;
  const int64_t sb = lo + bi * WAVES + wid;
  if (sb >= hi) return;
  double* lds = lds_all[wid];
  uint32_t ix[2][EPL]; double vv[2][EPL];
  auto load = [&](int64_t b, int slot) {
    const uint32_t* ip = a.idx + b * NNZ; const double* vp = a.val + b * NNZ;
    if (IW == 1) {
#pragma unroll
      for (int u = 0; u < EPL; u++) ix[slot][u] = ip[u * WAVE + lane];
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const uint4 q = ((const uint4*)ip)[h * WAVE + lane];
        ix[slot][4 * h] = q.x; ix[slot][4 * h + 1] = q.y; ix[slot][4 * h + 2] = q.z; ix[slot][4 * h + 3] = q.w;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { const double2 q = ((const double2*)vp)[i * WAVE + lane]; vv[slot][2 * i] = q.x; vv[slot][2 * i + 1] = q.y; }
  };
  load(sb * NB, 0);
#pragma unroll
  for (int i = 0; i < NB; i++) {
    const int cur = i & 1;
    const int64_t b = sb * NB + i;
    if (i + 1 < NB) load(b + 1, cur ^ 1);
    double xx[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) xx[u] = *(const double*)((const char*)a.x + ((size_t)ix[cur][u] << 3));
#pragma unroll
    for (int u = 0; u < EPL; u++) lds[u * WAVE + lane + 1] = vv[cur][u] * xx[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int R = a.rows_per_block, L = NNZ / R;
    const int T = 64 / R, gq = lane / T, t = lane % T;
    double acc = 0.0;
    for (int j = t; j < L; j += T) acc += lds[gq * L + j + 1];
    for (int off = T >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const double srow = __shfl(acc, (lane * T) & 63, 64);
    if (lane < R) a.y[b * R + lane] = srow;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

// persistent waves: the grid is what fits on the chip (PER_CU workgroups per CU), wave w of the grid takes the blocks
// w, w + nwaves, ... ; the next block's stream is requested before the current one is processed
template <int IW, int WAVES, int MINW>
__global__ __launch_bounds__(WAVES * 64, MINW) void k_loop(Args a) {
  __shared__ double lds_all[WAVES][NNZ + 2];
  const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  double* lds = lds_all[wid];
  const int64_t nw = (int64_t)gridDim.x * WAVES;
  int64_t b = (int64_t)blockIdx.x * WAVES + wid;
  uint32_t ix[EPL], nix[EPL]; double vv[EPL], nvv[EPL];
  auto load = [&](int64_t bb, uint32_t (&qi)[EPL], double (&qv)[EPL]) {
    const uint32_t* ip = a.idx + bb * NNZ; const double* vp = a.val + bb * NNZ;
    if (IW == 1) {
#pragma unroll
      for (int u = 0; u < EPL; u++) qi[u] = ip[u * WAVE + lane];
    } else {
#pragma unroll
      for (int h = 0; h < 2; h++) { const uint4 q = ((const uint4*)ip)[h * WAVE + lane]; qi[4 * h] = q.x; qi[4 * h + 1] = q.y; qi[4 * h + 2] = q.z; qi[4 * h + 3] = q.w; }
    }
#pragma unroll
    for (int i = 0; i < 4; i++) { const double2 q = ((const double2*)vp)[i * WAVE + lane]; qv[2 * i] = q.x; qv[2 * i + 1] = q.y; }
  };
  if (b < a.nblocks) load(b, ix, vv);
  for (; b < a.nblocks; b += nw) {
    const int64_t nb = b + nw;
    if (nb < a.nblocks) load(nb, nix, nvv);
    double xx[EPL];
#pragma unroll
    for (int u = 0; u < EPL; u++) xx[u] = *(const double*)((const char*)a.x + ((size_t)ix[u] << 3));
#pragma unroll
    for (int u = 0; u < EPL; u++) lds[u * WAVE + lane + 1] = vv[u] * xx[u];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int R = a.rows_per_block, L = NNZ / R;
    const int T = 64 / R, gq = lane / T, t = lane % T;
    double acc = 0.0;
    for (int j = t; j < L; j += T) acc += lds[gq * L + j + 1];
    for (int off = T >> 1; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    const double srow = __shfl(acc, (lane * T) & 63, 64);
    if (lane < R) a.y[b * R + lane] = srow;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int u = 0; u < EPL; u++) { ix[u] = nix[u]; vv[u] = nvv[u]; }
  }
}

// plain streaming read (float4 per lane, grid-stride) for reference
__global__ __launch_bounds__(256) void k_stream(const double2* __restrict__ p, int64_t n, double* out) {
  double s = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double2 q = p[i]; s += q.x + q.y; }
  if (s == 1.2345e300) out[0] = s;
}

__global__ void k_blob_idx(const uint32_t* __restrict__ idx, char* __restrict__ blob, int64_t nnz) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nnz; i += (int64_t)gridDim.x * 256)
    ((uint32_t*)(blob + (i / NNZ) * NNZ * 12 + NNZ * 8))[i % NNZ] = idx[i];
}

template <class F>
static double time_ms(F f, int rep = 7) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f(); CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < rep; r++) {
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char** argv) {
  const double gb = argc > 1 ? std::atof(argv[1]) : 3.0;
  const int64_t nblocks = (int64_t)(gb * 1e9 / (NNZ * 12)) / 32 * 32;
  const int64_t nnz = nblocks * NNZ;
  std::printf("blocks %lld, stream %.2f GB\n", (long long)nblocks, nnz * 12e-9);
  uint32_t* idx; double* val; char* blob; Desc* desc; double* x; double* y;
  CHECK(hipMalloc(&idx, nnz * 4)); CHECK(hipMalloc(&val, nnz * 8)); CHECK(hipMalloc(&blob, nnz * 12)); CHECK(hipMalloc(&desc, (nblocks + 1) * sizeof(Desc)));
  const int64_t nx = nblocks * 64;                       // 64 rows per block at most
  CHECK(hipMalloc(&x, nx * 8)); CHECK(hipMalloc(&y, nx * 8));
  CHECK(hipMemset(val, 0, nnz * 8)); CHECK(hipMemset(x, 0, nx * 8));
  std::vector<Desc> hd(nblocks + 1);
  for (int64_t b = 0; b <= nblocks; b++) hd[b] = Desc{b * NNZ, (int32_t)(b * 16), 0};
  CHECK(hipMemcpy(desc, hd.data(), (nblocks + 1) * sizeof(Desc), hipMemcpyHostToDevice));
  double* out; CHECK(hipMalloc(&out, 64));
  auto fill_idx = [&](int window, int rows) {
    // window > 0: columns of block b pseudo-random inside a window of `window` columns around the block's rows.
    // window < 0: a stencil - row i has the columns i + off[k] with NNZ / rows fixed sorted offsets inside -window
    // columns (consecutive rows gather neighbouring entries, as the operators of the hierarchy do)
    std::vector<uint32_t> h((size_t)nnz);
    uint64_t s = 88172645463325252ull;
    if (window > 0) {
      for (int64_t b = 0; b < nblocks; b++) {
        const int64_t base = std::max<int64_t>(0, std::min<int64_t>(b * rows - window / 2, nx - window));
        for (int k = 0; k < NNZ; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[(size_t)(b * NNZ + k)] = (uint32_t)(base + (int64_t)(s % (uint64_t)window)); }
      }
    } else {
      const int L = NNZ / rows, W = -window;
      std::vector<int> off(L);
      for (int k = 0; k < L; k++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; off[k] = (int)(s % (uint64_t)W); }
      std::sort(off.begin(), off.end());
      for (int64_t b = 0; b < nblocks; b++)
        for (int r = 0; r < rows; r++)
          for (int k = 0; k < L; k++) {
            const int64_t c = std::min<int64_t>(nx - 1, b * rows + r + off[k]);
            h[(size_t)(b * NNZ + r * L + k)] = (uint32_t)c;
          }
    }
    CHECK(hipMemcpy(idx, h.data(), nnz * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_blob_idx, dim3(4096), dim3(256), 0, 0, idx, blob, nnz);   // blob copies of the index words
    CHECK(hipDeviceSynchronize());
  };
  CHECK(hipMemset(blob, 0, nnz * 12));
  const double stream_bytes = nnz * 12.0;
  {
    const double ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(256 * 16), dim3(256), 0, 0, (const double2*)val, nnz / 2, out); });
    std::printf("%-64s %8.3f ms  %7.1f GB/s\n", "plain streaming read of the value array (float4, grid-stride)", ms, nnz * 8.0 / ms * 1e-6);
  }
  auto run = [&](const char* name, auto kern, int waves, int rows) {
    Args a{idx, val, blob, desc, x, y, nblocks, rows};
    const int64_t per_class = (nblocks + 7) / 8;
    const unsigned grid = (unsigned)(8 * ((per_class + waves - 1) / waves));
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, a); });
    const double bytes = stream_bytes + nblocks * rows * 8.0;   // rows = doubles stored per block
    std::printf("%-64s %8.3f ms  %7.1f GB/s (stream + y)\n", name, ms, bytes / ms * 1e-6);
  };
  auto run_multi = [&](const char* name, auto kern, int waves, int nbper) {
    Args a{idx, val, blob, desc, x, y, nblocks, 16};
    const int64_t per_class = (nblocks / nbper + 7) / 8;
    const unsigned grid = (unsigned)(8 * ((per_class + waves - 1) / waves));
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(grid), dim3(waves * 64), 0, 0, a); });
    std::printf("%-64s %8.3f ms  %7.1f GB/s (stream + y)\n", name, ms, (stream_bytes + nblocks * 16 * 8.0) / ms * 1e-6);
  };
  auto run_loop = [&](const char* name, auto kern, int waves, int wg_per_cu) {
    Args a{idx, val, blob, desc, x, y, nblocks, 16};
    const double ms = time_ms([&] { hipLaunchKernelGGL(kern, dim3(256 * wg_per_cu), dim3(waves * 64), 0, 0, a); });
    std::printf("%-64s %8.3f ms  %7.1f GB/s (stream + y)\n", name, ms, (stream_bytes + nblocks * 16 * 8.0) / ms * 1e-6);
  };
  const bool only_placement = std::getenv("PROBE_PLACEMENT_ONLY") != nullptr;
  for (int window : {-256, -2048}) {
    if (only_placement) break;
    fill_idx(window, 16);
    std::printf("-- gather + LDS row sums, stencil inside %d columns, 16 rows per block: blocks per wave / persistent waves\n", -window);
    run("reference: one block per wave, val 4 x dwordx4, idx 8 x dword", k_block<4, 1, 0, 1, 1, 1, 4>, 4, 16);
    run_multi("k_multi NB=1 idx dword", k_multi<1, 1, 4, 1>, 4, 1);
    run_multi("k_multi NB=2 idx dword", k_multi<2, 1, 4, 1>, 4, 2);
    run_multi("k_multi NB=4 idx dword", k_multi<4, 1, 4, 1>, 4, 4);
    run_multi("k_multi NB=8 idx dword", k_multi<8, 1, 4, 1>, 4, 8);
    run_multi("k_multi NB=2 idx x4", k_multi<2, 4, 4, 1>, 4, 2);
    run_multi("k_multi NB=4 idx x4", k_multi<4, 4, 4, 1>, 4, 4);
    run_multi("k_multi NB=4 idx dword, 5 waves/SIMD forced", k_multi<4, 1, 4, 5>, 4, 4);
    run_multi("k_multi NB=4 idx dword, 6 waves/SIMD forced", k_multi<4, 1, 4, 6>, 4, 4);
    run_multi("k_multi NB=2 idx dword, 2 waves per workgroup", k_multi<2, 1, 2, 1>, 2, 2);
    run_multi("k_multi NB=4 idx dword, 8 waves per workgroup", k_multi<4, 1, 8, 1>, 8, 4);
    for (int wgcu : {3, 4, 5, 6, 8}) {
      char nm[96]; std::snprintf(nm, 96, "k_loop persistent, %d workgroups of 4 waves per CU, idx dword", wgcu);
      run_loop(nm, k_loop<1, 4, 1>, 4, wgcu);
    }
    run_loop("k_loop persistent, 5 wg/CU, idx x4", k_loop<4, 4, 1>, 4, 5);
    run_loop("k_loop persistent, 6 wg/CU, 6 waves/SIMD forced", k_loop<1, 4, 6>, 4, 6);
  }
  // ---- does the relative placement of the index and the value array matter?  (per-operator rates of the benchmark move by
  // +-8 % from run to run with the same kernel: the arrays land at other physical addresses each time)
  if (std::getenv("PROBE_SKIP_MALLOC")) fill_idx(-256, 16);
  else {
    fill_idx(-256, 16);
    char* big; const size_t slack = (size_t)64 << 20;
    CHECK(hipMalloc(&big, (size_t)nnz * 12 + 2 * slack));
    CHECK(hipMemcpy(big, val, (size_t)nnz * 8, hipMemcpyDeviceToDevice));
    std::printf("-- index array placed d bytes behind the end of the value array (one allocation), gather + LDS row sums\n");
    for (size_t d : {(size_t)0, (size_t)256, (size_t)1024, (size_t)4096, (size_t)16384, (size_t)65536, (size_t)262144, (size_t)1 << 20, ((size_t)2 << 20) + 4096, ((size_t)32 << 20) + 8192}) {
      uint32_t* ip = (uint32_t*)(big + (size_t)nnz * 8 + d);
      CHECK(hipMemcpy(ip, idx, (size_t)nnz * 4, hipMemcpyDeviceToDevice));
      Args a{ip, (const double*)big, blob, desc, x, y, nblocks, 16};
      const int64_t per_class = (nblocks + 7) / 8;
      const unsigned grid = (unsigned)(8 * ((per_class + 3) / 4));
      const double ms = time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 0, 1, 1, 1, 4>), dim3(grid), dim3(256), 0, 0, a); });
      std::printf("   d = %10zu   %8.3f ms  %7.1f GB/s\n", d, ms, (stream_bytes + nblocks * 16 * 8.0) / ms * 1e-6);
    }
    std::printf("-- the same arrays as separate allocations (new addresses each time): which = 0 both move, 1 only idx, 2 only val\n");
    for (int which = 0; which < 3; which++)
      for (int rep = 0; rep < 16; rep++) {
        uint32_t* i2 = idx; double* v2 = val; void* pad;
        CHECK(hipMalloc(&pad, ((size_t)(rep * 5 + 1) << 20) + (size_t)rep * 4096 * 37));
        if (which != 2) { CHECK(hipMalloc(&i2, (size_t)nnz * 4)); CHECK(hipMemcpy(i2, idx, (size_t)nnz * 4, hipMemcpyDeviceToDevice)); }
        if (which != 1) { CHECK(hipMalloc(&v2, (size_t)nnz * 8)); CHECK(hipMemcpy(v2, val, (size_t)nnz * 8, hipMemcpyDeviceToDevice)); }
        Args a{i2, v2, blob, desc, x, y, nblocks, 16};
        const int64_t per_class = (nblocks + 7) / 8;
        const unsigned grid = (unsigned)(8 * ((per_class + 3) / 4));
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 0, 1, 1, 1, 4>), dim3(grid), dim3(256), 0, 0, a); });
        std::printf("   which %d rep %2d  idx %p val %p   %8.3f ms  %7.1f GB/s\n", which, rep, (void*)i2, (void*)v2, ms, (stream_bytes + nblocks * 16 * 8.0) / ms * 1e-6);
        if (which != 2) CHECK(hipFree(i2));
        if (which != 1) CHECK(hipFree(v2));
        CHECK(hipFree(pad));
      }
    CHECK(hipFree(big));
  }
  // ---- the same through the virtual-memory interface: does the size of the physical pieces under the value / index arrays
  // matter?  (PROBE_VMM=1; every repetition takes fresh addresses and fresh pieces)
  if (std::getenv("PROBE_VMM")) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice; acc.location.id = 0; acc.flags = hipMemAccessFlagsProtReadWrite;
    auto vm_alloc = [&](size_t bytes, size_t piece, std::vector<hipMemGenericAllocationHandle_t>& hs) -> char* {
      const size_t total = (bytes + piece - 1) / piece * piece;
      void* q = nullptr; CHECK(hipMemAddressReserve(&q, total, 0, nullptr, 0));
      for (size_t o = 0; o < total; o += piece) {
        hipMemGenericAllocationHandle_t h; CHECK(hipMemCreate(&h, piece, &prop, 0));
        CHECK(hipMemMap((char*)q + o, piece, 0, h, 0)); CHECK(hipMemSetAccess((char*)q + o, piece, &acc, 1));
        hs.push_back(h);
      }
      return (char*)q;       // (never unmapped: the probe exits)
    };
    std::printf("-- value and index arrays on pieces of the virtual-memory interface (fresh addresses and pieces every time)\n");
    const bool order_only = std::getenv("PROBE_VMM")[0] == '2';      // PROBE_VMM=2: 256-MB pieces only, in the order the driver hands memory out
    for (size_t piece : {(size_t)2 << 20, (size_t)32 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)4 << 30})
      for (int rep = 0; rep < (order_only ? (std::getenv("PROBE_REPS") ? std::atoi(std::getenv("PROBE_REPS")) : 24) : 6); rep++) {
        if (order_only && piece != ((size_t)256 << 20)) break;
        std::vector<hipMemGenericAllocationHandle_t> hs;
        void* pad; CHECK(hipMalloc(&pad, ((size_t)(rep * 5 + 1) << 20) + (size_t)rep * 4096 * 37));
        double* v2 = (double*)vm_alloc((size_t)nnz * 8, piece, hs);
        uint32_t* i2 = (uint32_t*)vm_alloc((size_t)nnz * 4, piece, hs);
        CHECK(hipMemcpy(v2, val, (size_t)nnz * 8, hipMemcpyDeviceToDevice));
        CHECK(hipMemcpy(i2, idx, (size_t)nnz * 4, hipMemcpyDeviceToDevice));
        Args a{i2, v2, blob, desc, x, y, nblocks, 16};
        const int64_t per_class = (nblocks + 7) / 8;
        const unsigned grid = (unsigned)(8 * ((per_class + 3) / 4));
        const double ms = time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 0, 1, 1, 1, 4>), dim3(grid), dim3(256), 0, 0, a); });
        // the same arrays with the eight XCD streams started at different phases of their eighths
        double ms_skew[3]; int skews[3] = {1237, 4099, 17};
        for (int q = 0; q < 3; q++) { a.skew = skews[q]; ms_skew[q] = time_ms([&] { hipLaunchKernelGGL((k_block<4, 1, 0, 1, 1, 1, 4>), dim3(grid), dim3(256), 0, 0, a); }); }
        const double by = stream_bytes + nblocks * 16 * 8.0;
        std::printf("   piece %5zu MB rep %d   %8.3f ms  %7.1f GB/s   skew 1237: %7.1f  4099: %7.1f  17: %7.1f\n", piece >> 20, rep, ms, by / ms * 1e-6,
                    by / ms_skew[0] * 1e-6, by / ms_skew[1] * 1e-6, by / ms_skew[2] * 1e-6);
        CHECK(hipFree(pad));
      }
  }
  return 0;
}
