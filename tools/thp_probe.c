// First-touch rate of freshly allocated host memory with and without transparent huge pages, and the rates of
// hipMemcpy from / to such memory (GPU box).   gcc -O2 -fopenmp tools/thp_probe.c -o /tmp/thp_probe -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -L/opt/rocm/lib -lamdhip64
#include <hip/hip_runtime_api.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>

static double now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

static void touch(char* p, size_t n) {
#pragma omp parallel for schedule(static)
  for (long long i = 0; i < (long long)(n >> 12); i++) p[(size_t)i << 12] = 1;
}

int main(void) {
  const size_t n = 8ull << 30;
  FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
  char line[128] = "?"; if (f) { fgets(line, sizeof line, f); fclose(f); }
  printf("THP enabled: %s threads %d\n", line, omp_get_max_threads());
  void* d = NULL;
  if (hipMalloc(&d, n) != hipSuccess) { printf("no device\n"); return 1; }
  for (int huge = 0; huge < 2; huge++) {
    char* p = NULL;
    if (posix_memalign((void**)&p, 2u << 20, n)) return 1;
    if (huge) madvise(p, n, MADV_HUGEPAGE);
    double t0 = now(); touch(p, n); double t1 = now();
    printf("%s: first touch %.2f GB/s", huge ? "MADV_HUGEPAGE" : "default      ", n / (t1 - t0) / 1e9);
    t0 = now(); hipMemcpy(d, p, n, hipMemcpyHostToDevice); t1 = now();
    printf(", H2D %.2f GB/s", n / (t1 - t0) / 1e9);
    t0 = now(); hipMemcpy(p, d, n, hipMemcpyDeviceToHost); t1 = now();
    printf(", D2H (touched) %.2f GB/s", n / (t1 - t0) / 1e9);
    free(p);
    if (posix_memalign((void**)&p, 2u << 20, n)) return 1;
    if (huge) madvise(p, n, MADV_HUGEPAGE);
    t0 = now(); hipMemcpy(p, d, n, hipMemcpyDeviceToHost); t1 = now();
    printf(", D2H (untouched) %.2f GB/s\n", n / (t1 - t0) / 1e9);
    free(p);
    // pinning in place: hipHostRegister of touched memory, copies to / from it, unregister
    if (posix_memalign((void**)&p, 2u << 20, n)) return 1;
    if (huge) madvise(p, n, MADV_HUGEPAGE);
    touch(p, n);
    t0 = now(); hipError_t e = hipHostRegister(p, n, hipHostRegisterDefault); t1 = now();
    printf("               hipHostRegister %s %.2f GB/s", e == hipSuccess ? "ok" : "FAILED", n / (t1 - t0) / 1e9);
    if (e == hipSuccess) {
      t0 = now(); hipMemcpy(d, p, n, hipMemcpyHostToDevice); t1 = now();
      printf(", H2D %.2f GB/s", n / (t1 - t0) / 1e9);
      t0 = now(); hipMemcpy(p, d, n, hipMemcpyDeviceToHost); t1 = now();
      printf(", D2H %.2f GB/s", n / (t1 - t0) / 1e9);
      t0 = now(); hipHostUnregister(p); t1 = now();
      printf(", unregister %.2f GB/s", n / (t1 - t0) / 1e9);
    }
    t0 = now(); free(p); t1 = now();
    printf(", free %.2f GB/s\n", n / (t1 - t0) / 1e9);
  }
  return 0;
}
