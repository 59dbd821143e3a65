// Device memory of the setup phase: a cache of released blocks in front of hipMalloc / hipFree.
//
// hipMalloc and hipFree of multi-GB blocks normally return in 0.3 ms, but every so often a hipMalloc that follows the
// release of large blocks stalls for 3.7-3.9 s (profiles/r02_hipmalloc_probe.txt; in the 464^3 setup two or three
// products per run paid it).  The setup allocates and frees operands, pools and results of similar sizes level after
// level, so released blocks simply wait here for the next request they fit; dev_cache_trim() hands everything back
// (the end of the setup, before the hierarchy is uploaded).
#pragma once
#include <cstddef>

namespace airhip {
void* dev_alloc(size_t bytes);           // throws std::runtime_error when the device is out of memory even after a trim
void dev_free(void* p);                  // any pointer dev_alloc returned (nullptr is ignored)
// long-lived arrays: a plain hipMalloc of exactly that size (released with hipFree), after a trim if the device is full
void* dev_alloc_exact(size_t bytes);
void dev_cache_trim();
size_t dev_cache_idle_bytes();

// Exception safety of the setup code, which holds raw dev_alloc pointers: while a DevScope is alive on a thread every
// block that thread takes with dev_alloc is noted (dev_free strikes it again).  When the scope is left by an exception the
// blocks still noted go back with dev_free - nothing stays in the cache's live table for ever; when it is left normally
// they pass to the enclosing scope (results that outlive the function).  trim = true also empties the cache on the way
// out, normal or not (the end of a setup call: later hipMalloc calls of the application must find the memory).
class DevScope {
 public:
  explicit DevScope(bool trim = false);
  ~DevScope();
  DevScope(const DevScope&) = delete;
  DevScope& operator=(const DevScope&) = delete;
 private:
  bool trim_;
  int exceptions_;
};
}  // namespace airhip
