// Device memory of the setup phase: a cache of released blocks in front of hipMalloc / hipFree.
//
// hipMalloc and hipFree of multi-GB blocks normally return in 0.3 ms, but every so often a hipMalloc that follows the
// release of large blocks stalls for 3.7-3.9 s (profiles/r02_hipmalloc_probe.txt; in the 464^3 setup two or three
// products per run paid it).  The setup allocates and frees operands, pools and results of similar sizes level after
// level, so released blocks simply wait here for the next request they fit; dev_cache_trim() hands everything back
// (the end of the setup, before the hierarchy is uploaded).
#pragma once
#include <cstddef>
#include <functional>

namespace airhip {
void* dev_alloc(size_t bytes);           // throws std::runtime_error when the device is out of memory even after a trim
void dev_free(void* p);                  // any pointer dev_alloc returned (nullptr is ignored)
// long-lived arrays (the hierarchy): a released block of the setup that fits closely (at most 1/8 larger), else a plain
// hipMalloc of exactly that size, after a trim if the device is full.  Release with dev_free_exact.
void* dev_alloc_exact(size_t bytes);
void dev_free_exact(void* p);
void dev_cache_trim();
// the same down to `keep_bytes` of idle blocks (the largest stay).  The upload of a hierarchy calls it first: measured at 464^3
// (profiles/r04_setup_cache.txt) the upload's hipMalloc calls of fresh memory take 3.9 / 4.2 / 5.5-6.6 / 7.7 s with 48 / 64 / 96 /
// 128 GB of the device held idle by the cache, while the setup before it wants 64-128 GB there (20-21 s against 25 s with 48)
void dev_cache_trim_to(size_t keep_bytes);
// Owners of device memory that is only an optimisation (the setup's kept copies of finished operators) register a function
// that frees it and returns the bytes; dev_alloc / dev_alloc_exact call them when the device is full and try once more.
int dev_add_pressure_handler(std::function<size_t()> f);
void dev_remove_pressure_handler(int id);
// free device memory as an allocation through dev_alloc sees it: what the driver reports plus the cache's idle blocks
size_t dev_available_bytes();
size_t dev_cache_idle_bytes();

// (Rounds 3-4 also had setup arenas on the virtual-memory interface - address ranges reserved per setup and backed piece by
// piece.  They were removed in round 5: round 3's records hold host segmentation faults and one GPU memory-access fault on
// that path whose mechanism was never established, and all they bought was ~2 s of a setup that the metric excludes;
// DESIGN.md section 4 keeps the post-mortem.)

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

// Where the library is, for the post-mortem of a GPU memory fault (the runtime aborts the process from its own thread; no
// exception, no error code): the long device steps of setup and upload note their name here, and with AIRHIP_FAULT_STAGE=1
// a SIGABRT handler prints the last one noted.  `what` must be a string literal (the pointer is kept).
void dev_stage(const char* what);
}  // namespace airhip
