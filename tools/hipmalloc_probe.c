// Cost of hipMalloc / hipFree of large blocks (GPU box).  gcc -O2 tools/hipmalloc_probe.c -o /tmp/hipmalloc_probe -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <time.h>
static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(void) {
  hipFree(0);
  const size_t sizes[] = {1ull << 30, 8ull << 30, 32ull << 30, 64ull << 30};
  for (int rep = 0; rep < 2; rep++)
    for (int k = 0; k < 4; k++) {
      void* p = NULL;
      double t0 = now(); hipError_t e = hipMalloc(&p, sizes[k]); double t1 = now();
      if (e != hipSuccess) { printf("hipMalloc %zu GiB failed\n", sizes[k] >> 30); continue; }
      hipMemset(p, 0, 1 << 20); hipDeviceSynchronize();
      double t2 = now(); hipFree(p); double t3 = now();
      printf("rep %d: %3zu GiB  hipMalloc %8.2f ms (%.1f GB/s)   hipFree %8.2f ms (%.1f GB/s)\n", rep, sizes[k] >> 30, (t1 - t0) * 1e3,
             sizes[k] / (t1 - t0) / 1e9, (t3 - t2) * 1e3, sizes[k] / (t3 - t2) / 1e9);
    }
  // two live blocks, freed in allocation order
  void *a = NULL, *b = NULL;
  double t0 = now(); hipMalloc(&a, 48ull << 30); hipMalloc(&b, 48ull << 30); double t1 = now();
  hipFree(a); hipFree(b); double t2 = now();
  printf("2 x 48 GiB: malloc %.2f ms, free %.2f ms\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
  return 0;
}
