// Large host arrays of the setup: transparent huge pages, and a small cache of released blocks.
//
// Measured on the GPU box (profiles/r02_thp_probe.txt and the 464^3 setup): with THP in "madvise" mode the first
// touch of fresh 4 KiB pages runs at 16 GB/s over 16 threads and hipMemcpy from such memory at 19 GB/s, against
// 175 GB/s and 47 GB/s with 2 MiB pages; and giving a 26 GB level operator BACK to the kernel (munmap) takes 1.3 s
// even then - 9 s over the 14 levels.  So blocks of 64 MiB and more are aligned to 2 MiB and advised as huge pages, and
// when they are released they wait in a bounded cache for the next request they fit (the level operators shrink
// slowly from level to level, so each generation takes over the blocks of the one before last).
#include "common.hpp"
#include <sys/mman.h>
#include <cstdio>
#include <mutex>
#include <thread>
#include <unordered_map>

namespace airhip {

namespace {

constexpr size_t CACHE_MIN = 64u << 20;             // blocks below this go straight back to malloc

struct HostCache {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;            // big blocks handed out: true size
  std::vector<std::pair<void*, size_t>> idle;        // released, waiting for reuse
  size_t idle_bytes = 0;
  // statistics since the last trim (GB): requests served from the cache / by the system, blocks kept / given back
  double reused = 0, fresh = 0, kept = 0, returned = 0;
  size_t limit = [] {
    const char* e = std::getenv("AIRHIP_HOST_CACHE_GB");
    return (size_t)(e ? std::atof(e) : 32.0) << 30;
  }();
};
HostCache& cache() { static HostCache* c = new HostCache; return *c; }   // never destroyed: blocks may outlive static destruction

}  // namespace

void* alloc_host_array(size_t bytes) {
  if (bytes < 2 * HUGE_ALIGN) {
    void* p = std::aligned_alloc(64, (bytes + 63) / 64 * 64);
    if (!p) throw std::bad_alloc();
    return p;
  }
  bytes = (bytes + HUGE_ALIGN - 1) / HUGE_ALIGN * HUGE_ALIGN;
  HostCache& c = cache();
  if (bytes >= CACHE_MIN) {
    std::lock_guard<std::mutex> lock(c.mu);
    // smallest idle block that holds the request without wasting more than a third of itself
    int best = -1;
    for (int k = 0; k < (int)c.idle.size(); k++)
      if (c.idle[k].second >= bytes && c.idle[k].second / 3 * 2 <= bytes && (best < 0 || c.idle[k].second < c.idle[best].second)) best = k;
    if (best >= 0) {
      const auto blk = c.idle[best];
      c.idle.erase(c.idle.begin() + best);
      c.idle_bytes -= blk.second;
      c.live[blk.first] = blk.second;
      c.reused += blk.second * 1e-9;
      return blk.first;
    }
  }
  void* p = std::aligned_alloc(HUGE_ALIGN, bytes);
  if (!p) throw std::bad_alloc();
  (void)madvise(p, bytes, MADV_HUGEPAGE);            // advisory: failure leaves ordinary pages
  if (bytes >= CACHE_MIN) { std::lock_guard<std::mutex> lock(c.mu); c.live[p] = bytes; c.fresh += bytes * 1e-9; }
  return p;
}

void free_host_array(void* p) {
  if (!p) return;
  HostCache& c = cache();
  // tracked blocks are 2-MiB aligned: the many small arrays (thread-local scratch inside parallel regions) skip the lock
  if (((uintptr_t)p & (HUGE_ALIGN - 1)) == 0) {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      const size_t bytes = it->second;
      c.live.erase(it);
      if (c.idle_bytes + bytes <= c.limit) { c.idle.emplace_back(p, bytes); c.idle_bytes += bytes; c.kept += bytes * 1e-9; return; }
      c.returned += bytes * 1e-9;      // (a helper thread for these was tried: the unmapping then slows the page faults of the
    }                                  // working threads down by as much as it saves)
  }
  std::free(p);
}

std::string host_array_cache_report() {
  HostCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  char buf[256];
  std::snprintf(buf, sizeof buf, "large host arrays: %.1f GB from released blocks, %.1f GB fresh; %.1f GB of released blocks kept, %.1f GB returned at once, %.1f GB idle",
                c.reused, c.fresh, c.kept, c.returned, c.idle_bytes * 1e-9);
  return buf;
}

void trim_host_array_cache(bool in_background) {
  HostCache& c = cache();
  std::vector<std::pair<void*, size_t>> blocks;
  { std::lock_guard<std::mutex> lock(c.mu); blocks.swap(c.idle); c.idle_bytes = 0; c.reused = c.fresh = c.kept = c.returned = 0; }
  if (blocks.empty()) return;
  auto work = [blocks] { for (const auto& b : blocks) std::free(b.first); };
  if (in_background) std::thread(work).detach(); else work();
}

}  // namespace airhip
