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
}  // namespace airhip
