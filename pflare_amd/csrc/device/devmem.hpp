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
// long-lived arrays (the hierarchy): a plain hipMalloc of exactly that size, after a trim if the device is full - or, while a
// setup's DevArena is alive, a block of that arena (memory the setup has already paid the first-touch cost for; the arena then
// lives as long as its last block).  Release with dev_free_exact.
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
// free device memory as an allocation through dev_alloc sees it: what the driver reports plus the cache's idle blocks plus
// the backed pieces of the arena that no block uses
size_t dev_available_bytes();
size_t dev_cache_idle_bytes();

// Arenas for a setup call.  Measured at 464^3 (AIRHIP_ALLOC_TRACE=1): a hipMalloc of fresh device memory costs 7 - 57 ms
// per GB from one call to the next and now and then seconds more, and because the operators grow from level to level the
// block cache above rarely holds a block large enough - 24 such calls added up to 9.6 s of a 29-s setup.  A DevArena
// reserves address space (hipMemAddressReserve; no memory yet), one range per class of blocks:
//   DEV_TEMP  what dev_alloc hands out by default: the temporaries of the setup (and, after recycle_temporaries(), of the
//             upload) - the range goes back as soon as they are gone;
//   DEV_KEPT  dev_alloc under a DevClassScope(DEV_KEPT): operators the setup keeps on the device until the upload has
//             converted them and the host mirror has fetched them;
//   DEV_HIER  dev_alloc_exact: the arrays that stay (the hierarchy).
// While the scope is alive blocks of 1 MiB and more are carved out of the class's range (first fit, 2-MiB granules,
// neighbours merged again on dev_free) and the range is backed with physical memory as far as the blocks reach
// (hipMemCreate + hipMemMap in 256-MB pieces: 0.6 - 2.5 ms per GB, tools/vmm_probe.hip).  A range is freed - memory and
// addresses - when it has been closed (end of the scope; recycle_temporaries) and its last block is gone; its addresses
// are then taken out of circulation (devmem.cpp: on this driver an address that is mapped a second time reads stale
// data), so a setup uses up address space: three times its limit, at most what the device holds each; the process stops
// making arenas after 64 TB and works through the cache / hipMalloc as it does without the virtual-memory interface
// (AIRHIP_DEVICE_VMM=0).
// bytes = the most the three may back together, 0 disables the arenas (AIRHIP_DEVICE_ARENA_GB sets the limit).
enum { DEV_TEMP = 0, DEV_KEPT = 1, DEV_HIER = 2 };
class DevClassScope {      // the class dev_alloc serves on this thread (DEV_TEMP outside any scope)
 public:
  explicit DevClassScope(int cls);
  ~DevClassScope();
  DevClassScope(const DevClassScope&) = delete;
  DevClassScope& operator=(const DevClassScope&) = delete;
 private:
  int prev_;
};
// ranges whose last block came back on a thread other than their maker's are freed by the maker's next call of this (the
// C API calls it on entry)
void dev_drain_releases();
int dev_check_covers();      // post-condition check of the freed ranges' covers (devmem.cpp); the number that had to be put back
int dev_class_of(void* p);      // the class of the arena a block came from, -1 for a block from the cache / hipMalloc
class DevArena {
 public:
  // front_bytes: how much of the temporaries' range a helper thread backs right away (devmem.cpp: DevArena::DevArena)
  explicit DevArena(size_t bytes, size_t front_bytes = 0);
  ~DevArena();
  void wait();      // until the helper thread is done
  // the temporaries are gone (the setup proper is over): their range goes back to the device and a fresh one serves the
  // temporaries of what follows (the upload of the hierarchy)
  // hier_front: how much of the range of the arrays that stay a helper thread backs right away
  void recycle_temporaries(size_t hier_front = 0);
  // what the destructor does, ahead of it (the caller is about to start a thread that must not run beside these frees)
  void finish();
  DevArena(const DevArena&) = delete;
  DevArena& operator=(const DevArena&) = delete;
 private:
  struct Impl;
  Impl* impl_ = nullptr;
  void start_helper(int cls, size_t ahead);
};
// the limit a setup of an operator with nnz non-zeros passes: about 1 kB per non-zero (at least 4 GB, at most 92 % of the
// free device memory); 0 = no arena, for setups that move less than 256 MB (~300 bytes per non-zero)
size_t dev_arena_bytes_for(double nnz);
// the front a setup of an operator with nnz non-zeros asks for: about what its temporaries need at their peak (350 bytes
// per non-zero)
inline size_t dev_arena_front_for(double nnz) { return (size_t)(350.0 * (nnz > 0 ? nnz : 0)); }

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
void dev_note_host_copy(void* host, size_t bytes);      // the host array a large copy is working on (post-mortem only)
}  // namespace airhip
