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
// long-lived arrays (the hierarchy): a plain hipMalloc of exactly that size, after a trim if the device is full - or, while a
// setup's DevArena is alive, a block of that arena (memory the setup has already paid the first-touch cost for; the arena then
// lives as long as its last block).  Release with dev_free_exact.
void* dev_alloc_exact(size_t bytes);
void dev_free_exact(void* p);
void dev_cache_trim();
size_t dev_cache_idle_bytes();

// One arena for the temporaries of a setup call.  Measured at 464^3 (AIRHIP_ALLOC_TRACE=1): a hipMalloc of fresh device
// memory costs 28 ms per GB (0.85 s for the 30-GB pool of a product) and now and then a second or two more, and because the
// operators grow from level to level the block cache above rarely holds a block large enough - 24 such calls added up to
// 9.6 s of a 29-s setup.  A DevArena takes ONE block when the setup starts; while it is alive dev_alloc carves blocks of
// 1 MiB and more out of it (first fit, 2-MiB granules, neighbours merged again on dev_free) and only falls back to the
// cache / hipMalloc when the arena has no room.  The arena goes back to the driver when the scope ends and its last block has
// been freed.  bytes = 0 disables it (AIRHIP_DEVICE_ARENA_GB overrides the caller's size; 0 = off).
class DevArena {
 public:
  // The block is requested on a helper thread (a fresh 119-GB block takes the driver about 3 s, which the first level of
  // the setup covers); until it is there dev_alloc serves requests the old way.
  explicit DevArena(size_t bytes);
  ~DevArena();
  void wait();      // until the helper thread has the block (a caller with nothing to do in the meantime)
  DevArena(const DevArena&) = delete;
  DevArena& operator=(const DevArena&) = delete;
 private:
  struct Impl;
  Impl* impl_ = nullptr;
};
// what a setup of an operator with nnz non-zeros asks for: ~300 bytes per non-zero, at most 45 % of the free device memory
size_t dev_arena_bytes_for(double nnz);

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
