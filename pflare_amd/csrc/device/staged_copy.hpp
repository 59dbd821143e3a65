// Large host <-> device copies of pageable host memory (the setup moves every level operator once in each direction).
// Arrays of 32 MiB and more are pinned in place for the copy (hipHostRegister; 57 GB/s); if that is refused
// the copy goes through two pinned bounce buffers, the DMA of chunk k+1 overlapping a
// multi-threaded memcpy of chunk k (~30 GB/s); the third mode is a bare hipMemcpy (small arrays).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace airhip {
void staged_h2d(void* d_dst, const void* h_src, size_t bytes);
void staged_d2h(void* h_dst, const void* d_src, size_t bytes);
}
