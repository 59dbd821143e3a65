#include "devmem.hpp"
#include "dev.hpp"
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <exception>
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
  hipError_t e = hipMalloc(&p, bytes);
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

void* dev_alloc_exact(size_t bytes) {
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

size_t dev_cache_idle_bytes() {
  DevCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  return c.idle_bytes;
}

}  // namespace airhip
