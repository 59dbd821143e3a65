#include "staged_copy.hpp"
#include "dev.hpp"
#include <cstring>
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

void staged_h2d(void* d_dst, const void* h_src, size_t bytes) {
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
