// vmm_probe.hip - can the setup arena be mapped piece by piece?  (diagnostic, not part of the library)
//
//   hipcc -O3 --offload-arch=gfx950 tools/vmm_probe.hip -o gpurun_out/vmm_probe && gpurun_out/vmm_probe [GB] [chunk_MB]
//
// Reserves a virtual range, backs it chunk by chunk (hipMemCreate + hipMemMap + hipMemSetAccess) and reports what a chunk
// costs, then streams through the range with a read+write kernel and does the same on a plain hipMalloc block of the same
// size: the two questions are (1) what does device memory cost per GB this way (a plain hipMalloc: 28 ms / GB, all of it in
// one call that holds the driver's lock) and (2) does a kernel run as fast on memory mapped like this.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void k_scale(double* p, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = p[i] * 1.0000001 + 1.0;
}

static double stream_gbs(double* p, size_t bytes) {
  const size_t n = bytes / 8;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k_scale, dim3(256 * 16), dim3(256), 0, 0, p, n);
  hipEventRecord(a, 0);
  for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k_scale, dim3(256 * 16), dim3(256), 0, 0, p, n);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return 3.0 * 2.0 * (double)bytes / (ms * 1e-3) / 1e9;
}

int main(int argc, char** argv) {
  const size_t total = (size_t)(argc > 1 ? std::atof(argv[1]) : 48.0) << 30;
  size_t chunk = (size_t)(argc > 2 ? std::atof(argv[2]) : 2048.0) << 20;
  int dev = 0; CK(hipSetDevice(dev));
  int vmm = 0; CK(hipDeviceGetAttribute(&vmm, hipDeviceAttributeVirtualMemoryManagementSupported, dev));
  std::printf("virtual memory management supported: %d\n", vmm);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  std::printf("recommended granularity: %zu bytes\n", gran);
  chunk = (chunk + gran - 1) / gran * gran;
  const size_t nchunks = total / chunk;
  { size_t f = 0, t = 0; CK(hipMemGetInfo(&f, &t)); std::printf("hipMemGetInfo at start: free %.1f GB of %.1f GB\n", f * 1e-9, t * 1e-9); }
  double t0 = now();
  void* base = nullptr;
  CK(hipMemAddressReserve(&base, nchunks * chunk, 0, nullptr, 0));
  std::printf("reserve %.1f GB: %.3f ms\n", nchunks * chunk * 1e-9, (now() - t0) * 1e3);
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  std::vector<hipMemGenericAllocationHandle_t> handles(nchunks);
  double t_create = 0, t_map = 0, t_acc = 0;
  const double t_all = now();
  for (size_t k = 0; k < nchunks; k++) {
    double a = now();
    CK(hipMemCreate(&handles[k], chunk, &prop, 0));
    double b = now();
    CK(hipMemMap((char*)base + k * chunk, chunk, 0, handles[k], 0));
    double c = now();
    CK(hipMemSetAccess((char*)base + k * chunk, chunk, &acc, 1));
    double d = now();
    t_create += b - a; t_map += c - b; t_acc += d - c;
  }
  { size_t f = 0, t = 0; CK(hipMemGetInfo(&f, &t)); std::printf("hipMemGetInfo after mapping: free %.1f GB of %.1f GB\n", f * 1e-9, t * 1e-9); }
  const double dt = now() - t_all;
  std::printf("%zu chunks of %.0f MB: %.3f s in all = %.1f ms / GB (create %.3f s, map %.3f s, set access %.3f s)\n", nchunks, chunk / 1048576.0, dt,
              dt * 1e3 / (nchunks * chunk * 1e-9), t_create, t_map, t_acc);
  // a kernel can use the first chunks while later ones are being mapped: here simply everything afterwards
  CK(hipMemset(base, 0, nchunks * chunk));
  CK(hipDeviceSynchronize());
  std::printf("stream (read + write) over the mapped range : %.0f GB/s\n", stream_gbs((double*)base, nchunks * chunk));
  t0 = now();
  void* plain = nullptr;
  CK(hipMalloc(&plain, nchunks * chunk));
  std::printf("plain hipMalloc of the same size: %.3f s = %.1f ms / GB\n", now() - t0, (now() - t0) * 1e3 / (nchunks * chunk * 1e-9));
  { size_t f = 0, t = 0; CK(hipMemGetInfo(&f, &t)); std::printf("hipMemGetInfo after the plain block too: free %.1f GB of %.1f GB\n", f * 1e-9, t * 1e-9); }
  CK(hipMemset(plain, 0, nchunks * chunk));
  CK(hipDeviceSynchronize());
  std::printf("stream (read + write) over the plain block  : %.0f GB/s\n", stream_gbs((double*)plain, nchunks * chunk));
  // a second plain block right after freeing the first: is memory the process has held before cheaper?
  CK(hipFree(plain));
  t0 = now();
  CK(hipMalloc(&plain, nchunks * chunk));
  std::printf("plain hipMalloc again after hipFree: %.3f s\n", now() - t0);
  CK(hipFree(plain));
  t0 = now();
  for (size_t k = 0; k < nchunks; k++) { CK(hipMemUnmap((char*)base + k * chunk, chunk)); CK(hipMemRelease(handles[k])); }
  CK(hipMemAddressFree(base, nchunks * chunk));
  std::printf("unmap + release + free: %.3f s\n", now() - t0);
  // does hipMemGetInfo follow blocks that come and go through this interface?
  for (int round = 0; round < 3; round++) {
    size_t f0 = 0, f1 = 0, f2 = 0, t = 0;
    CK(hipMemGetInfo(&f0, &t));
    CK(hipMemAddressReserve(&base, nchunks * chunk, 0, nullptr, 0));
    for (size_t k = 0; k < nchunks; k++) {
      CK(hipMemCreate(&handles[k], chunk, &prop, 0));
      CK(hipMemMap((char*)base + k * chunk, chunk, 0, handles[k], 0));
      CK(hipMemSetAccess((char*)base + k * chunk, chunk, &acc, 1));
    }
    CK(hipMemGetInfo(&f1, &t));
    for (size_t k = 0; k < nchunks; k++) { CK(hipMemUnmap((char*)base + k * chunk, chunk)); CK(hipMemRelease(handles[k])); }
    size_t f15 = 0; CK(hipMemGetInfo(&f15, &t));
    CK(hipMemAddressFree(base, nchunks * chunk));
    CK(hipMemGetInfo(&f2, &t));
    std::printf("round %d: free %.1f GB -> mapped %.1f GB -> unmapped + released %.1f GB -> address range freed %.1f GB\n", round, f0 * 1e-9, f1 * 1e-9, f15 * 1e-9, f2 * 1e-9);
  }
  return 0;
}
