#include "devmem.hpp"
#include "dev.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <thread>
#include <stdexcept>
#include <chrono>
#include <cstdio>
#include <exception>
#include <map>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace airhip {

namespace {

constexpr size_t CACHE_MIN = 1u << 20;      // smaller blocks are not worth keeping

struct DevCache {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;
  std::vector<std::pair<void*, size_t>> idle;
  size_t idle_bytes = 0;
  size_t limit = [] {
    const char* e = std::getenv("AIRHIP_DEVICE_CACHE_GB");
    return (size_t)(e ? std::atof(e) : 96.0) << 30;
  }();
};
DevCache& cache() { static DevCache* c = new DevCache; return *c; }

// ---- the setup arena (guarded by DevCache::mu)
constexpr size_t GRANULE = 2u << 20;
struct Arena {
  char* base = nullptr; size_t size = 0;
  std::map<size_t, size_t> free_by_off;             // offset -> length, neighbours merged
  std::unordered_map<void*, size_t> live;           // block -> length
  bool closing = false;                             // scope ended: release once the last block is back
  bool pending = false;                             // the helper thread is still asking the driver for the block
  size_t peak = 0, used = 0, misses = 0;
};
Arena& arena() { static Arena* a = new Arena; return *a; }

void* arena_alloc(Arena& a, size_t bytes) {
  if (!a.base || a.closing) return nullptr;
  const size_t need = (bytes + GRANULE - 1) / GRANULE * GRANULE;
  for (auto it = a.free_by_off.begin(); it != a.free_by_off.end(); ++it) {
    if (it->second < need) continue;
    const size_t off = it->first, len = it->second;
    a.free_by_off.erase(it);
    if (len > need) a.free_by_off[off + need] = len - need;
    a.live[a.base + off] = need;
    a.used += need; if (a.used > a.peak) a.peak = a.used;
    return a.base + off;
  }
  a.misses++;
  return nullptr;
}
bool arena_free(Arena& a, void* p) {          // true: the block was the arena's
  auto it = a.live.find(p);
  if (it == a.live.end()) return false;
  size_t off = (size_t)((char*)p - a.base), len = it->second;
  a.live.erase(it);
  a.used -= len;
  auto nxt = a.free_by_off.lower_bound(off);
  if (nxt != a.free_by_off.end() && off + len == nxt->first) { len += nxt->second; nxt = a.free_by_off.erase(nxt); }
  if (nxt != a.free_by_off.begin()) {
    auto prv = std::prev(nxt);
    if (prv->first + prv->second == off) { off = prv->first; len += prv->second; a.free_by_off.erase(prv); }
  }
  a.free_by_off[off] = len;
  return true;
}
void arena_release_if_done(Arena& a) {
  if (a.base && a.closing && a.live.empty()) {
    static const bool trace = std::getenv("AIRHIP_ALLOC_TRACE") != nullptr;
    if (trace) std::fprintf(stderr, "[devmem] arena of %.1f GB released: peak use %.1f GB, %zu requests did not fit\n", a.size * 1e-9, a.peak * 1e-9, a.misses);
    (void)hipFree(a.base);
    a = Arena();
  }
}

// journals of the DevScopes alive on this thread, innermost last
thread_local std::vector<std::unordered_set<void*>> t_scopes;
void note_alloc(void* p) { if (!t_scopes.empty()) t_scopes.back().insert(p); }
void note_free(void* p) {
  for (size_t k = t_scopes.size(); k-- > 0;) if (t_scopes[k].erase(p)) return;
}

}  // namespace

DevScope::DevScope(bool trim) : trim_(trim), exceptions_(std::uncaught_exceptions()) { t_scopes.emplace_back(); }
DevScope::~DevScope() {
  std::unordered_set<void*> mine;
  mine.swap(t_scopes.back());
  t_scopes.pop_back();
  if (std::uncaught_exceptions() > exceptions_) {
    for (void* p : mine) dev_free(p);                    // left by an exception: nobody holds these any more
  } else if (!t_scopes.empty()) {
    t_scopes.back().insert(mine.begin(), mine.end());    // results: the enclosing scope answers for them now
  }
  if (trim_) dev_cache_trim();
}

void* dev_alloc(size_t bytes) {
  if (bytes == 0) bytes = 1;
  DevCache& c = cache();
  if (bytes >= CACHE_MIN) {
    std::lock_guard<std::mutex> lock(c.mu);
    if (void* q = arena_alloc(arena(), bytes)) { note_alloc(q); return q; }
    int best = -1;          // smallest idle block that holds the request and is at most twice its size
    for (int k = 0; k < (int)c.idle.size(); k++)
      if (c.idle[k].second >= bytes && c.idle[k].second / 2 <= bytes && (best < 0 || c.idle[k].second < c.idle[best].second)) best = k;
    if (best >= 0) {
      const auto blk = c.idle[best];
      c.idle.erase(c.idle.begin() + best);
      c.idle_bytes -= blk.second;
      c.live[blk.first] = blk.second;
      note_alloc(blk.first);
      return blk.first;
    }
  }
  void* p = nullptr;
  static const bool trace = std::getenv("AIRHIP_ALLOC_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  hipError_t e = hipMalloc(&p, bytes);
  if (trace) {
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (dt > 0.05 || bytes >= ((size_t)1 << 30)) std::fprintf(stderr, "[devmem] hipMalloc %.2f GB took %.3f s (idle cache %.1f GB in %zu blocks)\n", bytes * 1e-9, dt, c.idle_bytes * 1e-9, c.idle.size());
  }
  if (e != hipSuccess) {                 // make room and try once more
    (void)hipGetLastError();
    dev_cache_trim();
    e = hipMalloc(&p, bytes);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw std::runtime_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
  }
  if (bytes >= CACHE_MIN) { std::lock_guard<std::mutex> lock(c.mu); c.live[p] = bytes; }
  note_alloc(p);
  return p;
}

void dev_free_exact(void* p) {
  if (!p) return;
  {
    DevCache& c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    Arena& a = arena();
    if (a.base && arena_free(a, p)) { arena_release_if_done(a); return; }
  }
  (void)hipFree(p);
}

void* dev_alloc_exact(size_t bytes) {
  if (bytes >= CACHE_MIN) {
    DevCache& c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    if (void* q = arena_alloc(arena(), bytes)) return q;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) { (void)hipGetLastError(); dev_cache_trim(); e = hipMalloc(&p, bytes ? bytes : 1); }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw std::runtime_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
  }
  return p;
}

void dev_free(void* p) {
  if (!p) return;
  note_free(p);
  DevCache& c = cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    Arena& a = arena();
    if (a.base && arena_free(a, p)) { arena_release_if_done(a); return; }
    auto it = c.live.find(p);
    if (it != c.live.end()) {
      const size_t bytes = it->second;
      c.live.erase(it);
      if (c.idle_bytes + bytes <= c.limit) { c.idle.emplace_back(p, bytes); c.idle_bytes += bytes; return; }
    }
  }
  (void)hipFree(p);
}

void dev_cache_trim() {
  DevCache& c = cache();
  std::vector<std::pair<void*, size_t>> blocks;
  { std::lock_guard<std::mutex> lock(c.mu); blocks.swap(c.idle); c.idle_bytes = 0; }
  for (const auto& b : blocks) (void)hipFree(b.first);
}

size_t dev_arena_bytes_for(double nnz) {
  size_t want = (size_t)(300.0 * std::max(nnz, 0.0));
  if (const char* e = std::getenv("AIRHIP_DEVICE_ARENA_GB")) want = (size_t)(std::atof(e) * 1e9);
  if (want < ((size_t)256 << 20)) return 0;            // small problems: the cache is all they need
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
  return std::min(want, (size_t)(0.45 * (double)free_b)) / GRANULE * GRANULE;
}

struct DevArena::Impl { std::thread worker; };

DevArena::DevArena(size_t bytes) {
  if (bytes == 0) return;
  DevCache& c = cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    Arena& a = arena();
    if (a.base || a.pending) return;                   // an enclosing call's arena is in use (or on its way)
    a.pending = true;
  }
  int dev = 0;
  (void)hipGetDevice(&dev);
  impl_ = new Impl;
  impl_->worker = std::thread([bytes, dev] {
    static const bool trace = std::getenv("AIRHIP_ALLOC_TRACE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    void* p = nullptr;
    const bool ok = hipSetDevice(dev) == hipSuccess && hipMalloc(&p, bytes) == hipSuccess;
    if (!ok) (void)hipGetLastError();                  // no arena: the cache path serves everything
    if (trace) std::fprintf(stderr, "[devmem] arena of %.1f GB %s in %.3f s (helper thread)\n", bytes * 1e-9, ok ? "allocated" : "REFUSED",
                            std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    DevCache& cc = cache();
    std::lock_guard<std::mutex> lock(cc.mu);
    Arena& a = arena();
    a.pending = false;
    if (ok) { a.base = (char*)p; a.size = bytes; a.free_by_off.clear(); a.free_by_off[0] = bytes; a.closing = false; }
  });
}
void DevArena::wait() { if (impl_ && impl_->worker.joinable()) impl_->worker.join(); }

DevArena::~DevArena() {
  if (!impl_) return;
  if (impl_->worker.joinable()) impl_->worker.join();
  delete impl_;
  DevCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  Arena& a = arena();
  if (a.base) { a.closing = true; arena_release_if_done(a); }
}

size_t dev_cache_idle_bytes() {
  DevCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  return c.idle_bytes;
}

}  // namespace airhip
