// vmm_stress.hip - is device memory obtained through hipMemAddressReserve / hipMemCreate / hipMemMap safe to recycle?
// (diagnostic, not part of the library)
//
//   hipcc -O3 --offload-arch=gfx950 tools/vmm_stress.hip -o /tmp/vmm_stress && /tmp/vmm_stress MODE [rounds] [max_GB]
//
// MODE 0: every block is its own reservation, freed with hipMemUnmap + hipMemRelease + hipMemAddressFree (address ranges come
//         and go);  MODE 1: one reservation for the whole run, blocks are sub-ranges whose pieces are mapped on allocation and
//         unmapped on release (addresses are backed, given back and backed again);  MODE 2: one reservation, pieces are mapped
//         on first use and never unmapped;  MODE 3: plain hipMalloc / hipFree (control);  MODE 4: as MODE 0, but every freed
//         range is covered at once with an inaccessible CPU mapping (mmap PROT_NONE, MAP_FIXED_NOREPLACE) so that the address
//         range is never handed out again.
// A random sequence of allocations and releases; every block is filled by a host -> device copy of a pattern that depends on
// the block's serial number (hipMemcpy from pinned-in-place host memory, as the library's staged_h2d does), checked at once by a
// kernel, and checked again by a kernel and by a device -> host copy right before it is released.  Prints the first mismatches.
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(2); } } while (0)

constexpr size_t PIECE = 256ull << 20;

__host__ __device__ inline uint64_t pat(uint64_t serial, uint64_t i) { return (serial * 0x9E3779B97F4A7C15ull) ^ (i * 0xD6E8FEB86659FD93ull + 12345); }

__global__ void k_check(const uint64_t* p, size_t n, uint64_t serial, unsigned long long* bad) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) mine += p[i] != pat(serial, i);
  if (mine) atomicAdd(bad, mine);
}

struct Block { char* p; size_t bytes; uint64_t serial; std::vector<hipMemGenericAllocationHandle_t> h; size_t off; };

int main(int argc, char** argv) {
  const int mode = argc > 1 ? std::atoi(argv[1]) : 0;
  const int rounds = argc > 2 ? std::atoi(argv[2]) : 400;
  const double max_gb = argc > 3 ? std::atof(argv[3]) : 120.0;
  int dev = 0; CK(hipSetDevice(dev));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = dev;
  hipMemAccessDesc acc = {};
  acc.location.type = hipMemLocationTypeDevice; acc.location.id = dev; acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t free_b = 0, total_b = 0; CK(hipMemGetInfo(&free_b, &total_b));
  const size_t arena_bytes = total_b / PIECE * PIECE;
  char* arena = nullptr;
  std::map<size_t, size_t> free_by_off;               // modes 1 / 2: first fit in one reservation
  std::vector<hipMemGenericAllocationHandle_t> piece_h(arena_bytes / PIECE);
  std::vector<char> piece_mapped(arena_bytes / PIECE, 0);
  std::vector<int> piece_users(arena_bytes / PIECE, 0);
  if (mode == 1 || mode == 2) { void* q = nullptr; CK(hipMemAddressReserve(&q, arena_bytes, 0, nullptr, 0)); arena = (char*)q; free_by_off[0] = arena_bytes; }
  unsigned long long* d_bad = nullptr; CK(hipMalloc((void**)&d_bad, 8));
  std::mt19937_64 rng(12345);
  std::vector<Block> live;
  size_t live_bytes = 0; uint64_t serial = 0; long long mism = 0, allocs = 0, tomb_fail = 0, tomb_gone = 0;
  std::vector<std::pair<char*, size_t>> covers;
  // host staging buffer: pageable, registered for each copy like the library does
  const size_t host_max = 6ull << 30;
  uint64_t* host = (uint64_t*)std::malloc(host_max);
  auto check_kernel = [&](const Block& b, const char* when) {
    CK(hipMemset(d_bad, 0, 8));
    hipLaunchKernelGGL(k_check, dim3(2048), dim3(256), 0, 0, (const uint64_t*)b.p, b.bytes / 8, b.serial, d_bad);
    unsigned long long bad = 0; CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
    if (bad) { mism++; if (mism < 20) std::printf("  MISMATCH (%s, kernel): block %llu of %.3f GB at %p: %llu words differ\n", when, (unsigned long long)b.serial, b.bytes * 1e-9, (void*)b.p, bad); }
  };
  for (int r = 0; r < rounds; r++) {
    const bool do_alloc = live.empty() || (live_bytes < (size_t)(max_gb * 1e9) && (rng() % 100) < 60);
    if (do_alloc) {
      // sizes from 1 MB to 6 GB, log-uniform, not multiples of anything
      const double u = (double)(rng() % 100000) / 100000.0;
      size_t bytes = (size_t)(1e6 * std::pow(6000.0, u)) / 8 * 8 + 8;
      if (bytes > host_max) bytes = host_max;
      Block b; b.bytes = bytes; b.serial = ++serial; b.off = 0;
      const size_t rounded = (bytes + PIECE - 1) / PIECE * PIECE;
      if (mode == 0 || mode == 4) {
        void* q = nullptr; CK(hipMemAddressReserve(&q, rounded, 0, nullptr, 0)); b.p = (char*)q;
        for (size_t k = 0; k < rounded / PIECE; k++) {
          hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, PIECE, &prop, 0));
          CK(hipMemMap(b.p + k * PIECE, PIECE, 0, h, 0)); CK(hipMemSetAccess(b.p + k * PIECE, PIECE, &acc, 1));
          b.h.push_back(h);
        }
      } else if (mode == 1 || mode == 2) {
        const size_t need = (bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        bool ok = false;
        for (auto it = free_by_off.begin(); it != free_by_off.end(); ++it) if (it->second >= need) {
          const size_t off = it->first, len = it->second;
          free_by_off.erase(it); if (len > need) free_by_off[off + need] = len - need;
          b.off = off; b.p = arena + off; ok = true; break;
        }
        if (!ok) { std::printf("arena full\n"); return 3; }
        for (size_t k = b.off / PIECE; k <= (b.off + need - 1) / PIECE; k++) {
          if (!piece_mapped[k]) {
            CK(hipMemCreate(&piece_h[k], PIECE, &prop, 0));
            CK(hipMemMap(arena + k * PIECE, PIECE, 0, piece_h[k], 0)); CK(hipMemSetAccess(arena + k * PIECE, PIECE, &acc, 1));
            piece_mapped[k] = 1;
          }
          piece_users[k]++;
        }
        b.bytes = bytes;
      } else { void* q = nullptr; CK(hipMalloc(&q, bytes)); b.p = (char*)q; }
      for (size_t i = 0; i < bytes / 8; i++) host[i] = pat(b.serial, i);
      const bool pin = bytes >= (32u << 20) && hipHostRegister(host, bytes, hipHostRegisterDefault) == hipSuccess;
      CK(hipMemcpy(b.p, host, bytes, hipMemcpyHostToDevice));
      if (pin) CK(hipHostUnregister(host));
      check_kernel(b, "after the fill");
      live.push_back(b); live_bytes += bytes; allocs++;
    } else {
      const size_t k = rng() % live.size();
      Block b = live[k]; live[k] = live.back(); live.pop_back(); live_bytes -= b.bytes;
      check_kernel(b, "before the release");
      // and through the copy engine
      const bool pin = b.bytes >= (32u << 20) && hipHostRegister(host, b.bytes, hipHostRegisterDefault) == hipSuccess;
      CK(hipMemcpy(host, b.p, b.bytes, hipMemcpyDeviceToHost));
      if (pin) CK(hipHostUnregister(host));
      size_t bad = 0; for (size_t i = 0; i < b.bytes / 8; i++) bad += host[i] != pat(b.serial, i);
      if (bad) { mism++; if (mism < 20) std::printf("  MISMATCH (before the release, copy): block %llu of %.3f GB: %zu words differ\n", (unsigned long long)b.serial, b.bytes * 1e-9, bad); }
      CK(hipDeviceSynchronize());
      if (mode == 0 || mode == 4) {
        const size_t rounded = (b.bytes + PIECE - 1) / PIECE * PIECE;
        for (size_t q = 0; q < b.h.size(); q++) { CK(hipMemUnmap(b.p + q * PIECE, PIECE)); CK(hipMemRelease(b.h[q])); }
        CK(hipMemAddressFree(b.p, rounded));
        if (mode == 4) {
          void* t = mmap(b.p, rounded, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE, -1, 0);
          if (t != (void*)b.p) { tomb_fail++; if (t != MAP_FAILED) munmap(t, rounded); }
          else covers.push_back({b.p, rounded});
          // are the covers of the ranges freed earlier still there?  (a second cover that succeeds means the first one is gone)
          for (auto& c : covers) {
            void* t2 = mmap(c.first, c.second, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE | MAP_FIXED_NOREPLACE, -1, 0);
            if (t2 == (void*)c.first) tomb_gone++;
            else if (t2 != MAP_FAILED) munmap(t2, c.second);
          }
        }
      } else if (mode == 1 || mode == 2) {
        const size_t need = (b.bytes + (2u << 20) - 1) / (2u << 20) * (2u << 20);
        for (size_t q = b.off / PIECE; q <= (b.off + need - 1) / PIECE; q++) {
          if (--piece_users[q] == 0 && mode == 1) { CK(hipMemUnmap(arena + q * PIECE, PIECE)); CK(hipMemRelease(piece_h[q])); piece_mapped[q] = 0; }
        }
        size_t off = b.off, len = need;
        auto nxt = free_by_off.lower_bound(off);
        if (nxt != free_by_off.end() && off + len == nxt->first) { len += nxt->second; nxt = free_by_off.erase(nxt); }
        if (nxt != free_by_off.begin()) { auto prv = std::prev(nxt); if (prv->first + prv->second == off) { off = prv->first; len += prv->second; free_by_off.erase(prv); } }
        free_by_off[off] = len;
      } else CK(hipFree(b.p));
    }
  }
  std::printf("mode %d: %d steps, %lld allocations, %.1f GB live at the end, %lld blocks with mismatches", mode, rounds, allocs, live_bytes * 1e-9, mism);
  if (mode == 4) std::printf(", %lld freed ranges could not be covered, %lld covers found removed later", tomb_fail, tomb_gone);
  std::printf("\n");
  return mism ? 1 : 0;
}
