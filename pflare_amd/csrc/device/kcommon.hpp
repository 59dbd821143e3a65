// Constants and helpers shared by the HIP kernel files (gfx950, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace airhip {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;
constexpr int MAX_BLOCKS = 2048;  // 256 CUs x 8 workgroups

// make a wave-uniform 64-bit value scalar (SGPR pair)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((int64_t)hi << 32) | (int64_t)(uint32_t)lo;
}

}  // namespace airhip
