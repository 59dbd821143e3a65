// Large host <-> device copies of pageable host memory through two pinned bounce buffers: the DMA of chunk k+1
// overlaps a multi-threaded memcpy of chunk k (a plain hipMemcpy of pageable memory ran at 13-15 GB/s on the test
// box; the setup moves every level operator once in each direction).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace airhip {
void staged_h2d(void* d_dst, const void* h_src, size_t bytes);
void staged_d2h(void* h_dst, const void* d_src, size_t bytes);
}
