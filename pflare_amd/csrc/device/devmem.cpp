#include "devmem.hpp"
#include "dev.hpp"
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <chrono>
#include <cstdio>
#include <exception>
#include <unordered_map>
#include <unordered_set>
#include <vector>
#include <functional>
#include <atomic>
#include <string>
#include <csignal>
#include <cstring>
#include <unistd.h>
#include <execinfo.h>

namespace airhip {

namespace {
std::atomic<const char*> g_stage{"(none)"};
double trace_clock() {
  static const auto t0 = std::chrono::steady_clock::now();
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
extern "C" void airhip_abort_note(int sig) {
  const char* s = g_stage.load(std::memory_order_relaxed);
  const char head[] = "[airhip] fatal signal while in stage: ";
  (void)!write(2, head, sizeof head - 1);
  (void)!write(2, s, std::strlen(s));
  (void)!write(2, "\n", 1);
  void* frames[48];
  const int n = backtrace(frames, 48);
  backtrace_symbols_fd(frames, n, 2);
  std::signal(sig, SIG_DFL);
  std::raise(sig);
}
}  // namespace

void dev_stage(const char* what) {
  static const bool on = [] {
    const char* e = std::getenv("AIRHIP_FAULT_STAGE");
    if (!(e && e[0] == '1')) return false;
    std::signal(SIGABRT, airhip_abort_note);
    return true;
  }();
  if (on) g_stage.store(what, std::memory_order_relaxed);
}

namespace {

constexpr size_t CACHE_MIN = 1u << 20;      // smaller blocks are not worth keeping

struct DevCache {
  std::mutex mu;
  std::unordered_map<void*, size_t> live;
  std::vector<std::pair<void*, size_t>> idle;
  size_t idle_bytes = 0;
  size_t limit = [] {
    const char* e = std::getenv("AIRHIP_DEVICE_CACHE_GB");
    return (size_t)(e ? std::atof(e) : 128.0) << 30;      // (profiles/r04_setup_cache.txt)
  }();
};
DevCache& cache() { static DevCache* c = new DevCache; return *c; }

hipError_t raw_alloc(void** p, size_t bytes) { return hipMalloc(p, bytes); }
void raw_free(void* p) { (void)hipFree(p); }

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

namespace {
std::mutex g_pressure_mu;
std::vector<std::pair<int, std::function<size_t()>>> g_pressure;
int g_pressure_next = 0;
// asks the registered owners of optional device memory to let go of it; the bytes they freed
size_t relieve_pressure() {
  std::vector<std::function<size_t()>> fs;
  { std::lock_guard<std::mutex> lock(g_pressure_mu); for (auto& f : g_pressure) fs.push_back(f.second); }
  size_t got = 0;
  for (auto& f : fs) got += f();
  static const bool trace = std::getenv("AIRHIP_ALLOC_TRACE") != nullptr;
  if (trace && !fs.empty()) std::fprintf(stderr, "[devmem %.3f] out of device memory: %.1f GB of optional copies dropped\n", trace_clock(), got * 1e-9);
  return got;
}
}  // namespace
int dev_add_pressure_handler(std::function<size_t()> f) {
  std::lock_guard<std::mutex> lock(g_pressure_mu);
  g_pressure.emplace_back(++g_pressure_next, std::move(f));
  return g_pressure_next;
}
void dev_remove_pressure_handler(int id) {
  std::lock_guard<std::mutex> lock(g_pressure_mu);
  for (size_t k = 0; k < g_pressure.size(); k++) if (g_pressure[k].first == id) { g_pressure.erase(g_pressure.begin() + (long)k); return; }
}

static std::string memory_state() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) (void)hipGetLastError();
  DevCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  size_t lb = 0;
  for (auto& kv : c.live) lb += kv.second;
  return "driver reports " + std::to_string(free_b >> 20) + " MiB free of " + std::to_string(total_b >> 20) + ", block cache holds " +
         std::to_string(c.idle_bytes >> 20) + " MiB idle and " + std::to_string(lb >> 20) + " MiB in use";
}
static void* dev_alloc_once(size_t bytes, bool& failed);
void* dev_alloc(size_t bytes) {
  bool failed = false;
  void* p = dev_alloc_once(bytes, failed);
  if (failed) {
    (void)relieve_pressure();
    p = dev_alloc_once(bytes, failed);
  }
  if (failed) throw std::runtime_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: out of memory (" + memory_state() + ")");
  return p;
}
static void* dev_alloc_once(size_t bytes, bool& failed) {
  failed = false;
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
  static const bool trace = std::getenv("AIRHIP_ALLOC_TRACE") != nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  hipError_t e = raw_alloc(&p, bytes);
  if (trace) {
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
    if (dt > 0.05 || bytes >= ((size_t)1 << 30)) std::fprintf(stderr, "[devmem %.3f] hipMalloc %.2f GB took %.3f s (idle cache %.1f GB in %zu blocks)\n", trace_clock(), bytes * 1e-9, dt, c.idle_bytes * 1e-9, c.idle.size());
  }
  if (e != hipSuccess) {                 // make room and try once more
    (void)hipGetLastError();
    dev_cache_trim();
    e = raw_alloc(&p, bytes);
  }
  if (e != hipSuccess) { (void)hipGetLastError(); failed = true; return nullptr; }
  if (bytes >= CACHE_MIN) { std::lock_guard<std::mutex> lock(c.mu); c.live[p] = bytes; }
  note_alloc(p);
  return p;
}

void dev_free_exact(void* p) {
  if (p) raw_free(p);
}

void* dev_alloc_exact(size_t bytes) {
  if (bytes >= CACHE_MIN) {
    // A released block of the setup that fits closely (at most 1/8 larger) becomes the long-lived array: it is a plain
    // hipMalloc block (hipFree takes it back like any other) whose memory the driver does not have to hand out afresh -
    // 7-57 ms per GB for the 106 GB of a 464^3 hierarchy.
    DevCache& c = cache();
    std::lock_guard<std::mutex> lock(c.mu);
    int best = -1;
    for (int k = 0; k < (int)c.idle.size(); k++)
      if (c.idle[k].second >= bytes && c.idle[k].second - bytes <= bytes / 8 && (best < 0 || c.idle[k].second < c.idle[best].second)) best = k;
    if (best >= 0) {
      void* q = c.idle[best].first;
      c.idle_bytes -= c.idle[best].second;
      c.idle.erase(c.idle.begin() + best);
      return q;
    }
  }
  void* p = nullptr;
  hipError_t e = raw_alloc(&p, bytes ? bytes : 1);
  if (e != hipSuccess) { (void)hipGetLastError(); dev_cache_trim(); e = raw_alloc(&p, bytes ? bytes : 1); }
  if (e != hipSuccess) { (void)hipGetLastError(); (void)relieve_pressure(); e = raw_alloc(&p, bytes ? bytes : 1); }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    throw std::runtime_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e) + " (" + memory_state() + ")");
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
  raw_free(p);
}

void dev_cache_trim() {
  DevCache& c = cache();
  std::vector<std::pair<void*, size_t>> blocks;
  { std::lock_guard<std::mutex> lock(c.mu); blocks.swap(c.idle); c.idle_bytes = 0; }
  for (const auto& b : blocks) raw_free(b.first);
}

// gives idle blocks back, smallest first, until at most `keep_bytes` of them are left
void dev_cache_trim_to(size_t keep_bytes) {
  DevCache& c = cache();
  std::vector<std::pair<void*, size_t>> drop;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    std::sort(c.idle.begin(), c.idle.end(), [](const std::pair<void*, size_t>& a, const std::pair<void*, size_t>& b) { return a.second > b.second; });
    while (!c.idle.empty() && c.idle_bytes > keep_bytes) { drop.push_back(c.idle.back()); c.idle_bytes -= c.idle.back().second; c.idle.pop_back(); }
  }
  for (const auto& b : drop) raw_free(b.first);
}

size_t dev_available_bytes() {
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) { (void)hipGetLastError(); return 0; }
  DevCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  return free_b + c.idle_bytes;
}

size_t dev_cache_idle_bytes() {
  DevCache& c = cache();
  std::lock_guard<std::mutex> lock(c.mu);
  return c.idle_bytes;
}

}  // namespace airhip
