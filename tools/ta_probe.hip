// Vector-memory instruction cost probe (gfx950): how many CU cycles one wave64 load instruction occupies the
// address / L1 path, by access width and by the number of distinct 128-B lines a gather touches.  All data is
// L2-resident (small buffers, every workgroup reads the same window), so what is measured is the TA / TCP
// throughput of one CU, not HBM.   Build + run on the GPU box:
//   hipcc -O3 --offload-arch=gfx950 tools/ta_probe.hip -o /tmp/ta_probe && /tmp/ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); std::exit(1); } } while (0)

constexpr int ITER = 4096;

// coalesced loads of W bytes per lane from a 64 KB window (L1 / L2 resident)
template <class T>
__global__ __launch_bounds__(256) void k_coalesced(const T* __restrict__ buf, int window_elems, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T acc{};
  unsigned off = (wave * 64 + lane);
  for (int i = 0; i < ITER; i += 4) {
    T a = buf[(off + (i + 0) * 256) & (window_elems - 1)];
    T b = buf[(off + (i + 1) * 256) & (window_elems - 1)];
    T c = buf[(off + (i + 2) * 256) & (window_elems - 1)];
    T d = buf[(off + (i + 3) * 256) & (window_elems - 1)];
    if constexpr (sizeof(T) == 4) acc += a + b + c + d;
    else if constexpr (sizeof(T) == 8) acc += a + b + c + d;
    else { acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y; }
  }
  if constexpr (sizeof(T) <= 8) { if (acc == (T)123456789) out[0] = 1.0; }
  else { if (acc.x == 123456789.0) out[0] = 1.0; }
}

// gather of 8 B per lane; idx[] precomputed so that each wave instruction touches `lines` distinct 128-B lines
__global__ __launch_bounds__(256) void k_gather(const double* __restrict__ buf, const int* __restrict__ idx, int nidx, double* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc = 0.0;
  // the index stream itself is tiny and L1 resident: 64 ints per instruction, reused
  for (int i = 0; i < ITER; i += 4) {
    const int b = ((i + wave * 8) * 64) & (nidx - 1);
    const int i0 = idx[b + lane], i1 = idx[(b + 64) & (nidx - 1) + lane], i2 = idx[(b + 128) & (nidx - 1) + lane], i3 = idx[(b + 192) & (nidx - 1) + lane];
    acc += buf[i0] + buf[i1] + buf[i2] + buf[i3];
  }
  if (acc == 123456789.0) out[0] = 1.0;
}

template <class F>
static double time_ms(F f, int rep = 5) {
  hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  f();
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < rep; r++) {
    CHECK(hipEventRecord(a)); f(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const double ghz = prop.clockRate * 1e-6;
  std::printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, cus, ghz);
  const size_t bytes = 1 << 20;     // 1 MB buffer: L2 resident
  void* buf; CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 0, bytes));
  double* out; CHECK(hipMalloc(&out, 64));
  for (int wg_per_cu : {2, 4, 8}) {
    const int grid = cus * wg_per_cu;
    const double instr_per_cu = (double)wg_per_cu * 4 * ITER;   // wave instructions per CU
    auto report = [&](const char* name, double ms, double bytes_per_instr) {
      const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_cu;
      std::printf("  wg/cu=%d %-28s %7.2f cyc per wave-instr per CU  (%6.1f B/clk/CU)\n", wg_per_cu, name, cyc, bytes_per_instr / cyc);
    };
    report("coalesced dword (256 B)", time_ms([&] { hipLaunchKernelGGL(k_coalesced<int>, dim3(grid), dim3(256), 0, 0, (const int*)buf, 16384, out); }), 256);
    report("coalesced dwordx2 (512 B)", time_ms([&] { hipLaunchKernelGGL(k_coalesced<double>, dim3(grid), dim3(256), 0, 0, (const double*)buf, 8192, out); }), 512);
    report("coalesced dwordx4 (1 KB)", time_ms([&] { hipLaunchKernelGGL(k_coalesced<double2>, dim3(grid), dim3(256), 0, 0, (const double2*)buf, 4096, out); }), 1024);
    for (int lines : {1, 2, 4, 8, 16, 32, 64}) {
      // 64 lanes spread over `lines` lines (16 doubles per line), sorted
      const int nidx = 64 * 64;
      std::vector<int> h(nidx);
      for (int g = 0; g < 64; g++)
        for (int l = 0; l < 64; l++) {
          const int line = (g * 7 + (l * lines) / 64 * 3) % 512;          // distinct lines, a few lines apart
          h[g * 64 + l] = line * 16 + (l % 16);
        }
      int* d; CHECK(hipMalloc(&d, nidx * 4)); CHECK(hipMemcpy(d, h.data(), nidx * 4, hipMemcpyHostToDevice));
      char nm[64]; std::snprintf(nm, 64, "gather 8 B, %2d lines/instr", lines);
      // every gather instruction comes with one coalesced dword index load: subtract it below by eye
      report(nm, time_ms([&] { hipLaunchKernelGGL(k_gather, dim3(grid), dim3(256), 0, 0, (const double*)buf, d, nidx, out); }) / 2.0, 512);
      CHECK(hipFree(d));
    }
  }
  return 0;
}
