#include "staged_copy.hpp"
#include "dev.hpp"
#include "devmem.hpp"
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <omp.h>

namespace airhip {

namespace {

constexpr size_t CHUNK = 64ull << 20;      // 64 MB per bounce buffer
constexpr size_t SMALL = 8ull << 20;       // below this a plain copy is as fast

struct Stage {
  char* buf[2] = {nullptr, nullptr};
  hipStream_t s = nullptr;
  hipEvent_t ev[2] = {nullptr, nullptr};
  bool ok = false;
  Stage() {
    if (hipHostMalloc((void**)&buf[0], CHUNK) != hipSuccess || hipHostMalloc((void**)&buf[1], CHUNK) != hipSuccess) return;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return;
    if (hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) return;
    ok = true;
  }
};
Stage& stage() { static thread_local Stage st; return st; }

void par_memcpy(char* dst, const char* src, size_t n) {
  const size_t piece = 1ull << 20;
  const long long np = (long long)((n + piece - 1) / piece);
#pragma omp parallel for schedule(static)
  for (long long p = 0; p < np; p++) {
    const size_t o = (size_t)p * piece;
    std::memcpy(dst + o, src + o, std::min(piece, n - o));
  }
}

}  // namespace

// 0: large arrays pinned in place (falls back to the bounce buffers when the registration fails); 1: bounce buffers; 2: plain
// hipMemcpy.  Fixed at 0 since round 5 (the AIRHIP_HOST_COPY switch measured the three once: profiles/r02_thp_probe.txt).
static int copy_mode() { return 0; }

// Large arrays are pinned where they are for the duration of the copy: on the GPU box hipHostRegister of 2 MiB-paged
// memory runs at 600 GB/s (68 GB/s on 4 KiB pages) and the copy then moves at 57 GB/s either way, against 47 / 25 GB/s
// for a plain hipMemcpy from / to pageable memory and ~30 GB/s through the bounce buffers (profiles/r02_thp_probe.txt).
constexpr size_t PIN_MIN = 32ull << 20;
static bool pinned_copy(void* dst, const void* src, size_t bytes, bool to_device) {
  void* host = to_device ? const_cast<void*>(src) : dst;
  if (!to_device) {
    // a fresh destination: fault the pages in with all threads first (registering would do it with one)
    char* p = (char*)host;
    const long long pages = (long long)((bytes + 4095) >> 12);
#pragma omp parallel for schedule(static)
    for (long long k = 0; k < pages; k++) { volatile char* q = p + ((size_t)k << 12); *q = *q; }
  }
  if (hipHostRegister(host, bytes, hipHostRegisterDefault) != hipSuccess) { (void)hipGetLastError(); return false; }
  const hipError_t e = hipMemcpy(dst, src, bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost);
  (void)hipHostUnregister(host);
  HIP_CHECK(e);
  return true;
}

void staged_h2d(void* d_dst, const void* h_src, size_t bytes) {
  if (copy_mode() == 2) { HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); return; }
  if (copy_mode() == 0 && bytes >= PIN_MIN && pinned_copy(d_dst, h_src, bytes, true)) return;
  Stage& st = stage();
  if (bytes < SMALL || !st.ok) { HIP_CHECK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice)); return; }
  const char* src = (const char*)h_src; char* dst = (char*)d_dst;
  size_t off = 0; int k = 0;
  bool pending[2] = {false, false};
  while (off < bytes) {
    const size_t n = std::min(CHUNK, bytes - off);
    if (pending[k]) HIP_CHECK(hipEventSynchronize(st.ev[k]));      // the DMA that last used this buffer is done
    par_memcpy(st.buf[k], src + off, n);
    HIP_CHECK(hipMemcpyAsync(dst + off, st.buf[k], n, hipMemcpyHostToDevice, st.s));
    HIP_CHECK(hipEventRecord(st.ev[k], st.s));
    pending[k] = true;
    off += n; k ^= 1;
  }
  HIP_CHECK(hipStreamSynchronize(st.s));
}

void staged_d2h(void* h_dst, const void* d_src, size_t bytes) {
  if (copy_mode() == 2) { HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost)); return; }
  if (copy_mode() == 0 && bytes >= PIN_MIN && pinned_copy(h_dst, d_src, bytes, false)) return;
  Stage& st = stage();
  if (bytes < SMALL || !st.ok) { HIP_CHECK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost)); return; }
  char* dst = (char*)h_dst; const char* src = (const char*)d_src;
  // chunk c is copied by DMA into buffer c & 1 while the host drains chunk c - 1 from the other buffer
  const size_t nchunks = (bytes + CHUNK - 1) / CHUNK;
  auto issue = [&](size_t c) {
    const size_t o = c * CHUNK, n = std::min(CHUNK, bytes - o);
    HIP_CHECK(hipMemcpyAsync(st.buf[c & 1], src + o, n, hipMemcpyDeviceToHost, st.s));
    HIP_CHECK(hipEventRecord(st.ev[c & 1], st.s));
  };
  issue(0);
  for (size_t c = 0; c < nchunks; c++) {
    HIP_CHECK(hipEventSynchronize(st.ev[c & 1]));
    if (c + 1 < nchunks) issue(c + 1);
    const size_t o = c * CHUNK, n = std::min(CHUNK, bytes - o);
    par_memcpy(dst + o, st.buf[c & 1], n);
  }
}

}  // namespace airhip
