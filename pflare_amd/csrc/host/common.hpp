// Host-side containers and helpers shared by the PCAIR setup code.
//
// The setup phase of PCAIR (hierarchy construction) is host code in the reference as
// well (Fortran on PETSc Mat objects, src/AIR_MG_Setup.F90); this is its C++/OpenMP
// counterpart.  It produces the operands the HIP solve phase consumes.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <memory>
#include <string>
#include <stdexcept>
#include <algorithm>
#include <chrono>
#include <omp.h>

namespace airhip {

// Memory of the host arrays (host/hostmem.cpp): blocks of 4 MiB and more are 2 MiB-aligned and advised as transparent
// huge pages, blocks of 64 MiB and more wait in a bounded cache when they are released (returning tens of GB to the
// kernel costs seconds).  Release with free_host_array only (a block released with std::free would stay in the cache's
// table of live blocks under its old size).
constexpr size_t HUGE_ALIGN = 2u << 20;
void* alloc_host_array(size_t bytes);
void free_host_array(void* p);
// returns the cached blocks to the system (the end of a setup), optionally on a detached thread
void trim_host_array_cache(bool in_background);
std::string host_array_cache_report();   // since the last trim

// Uninitialised, 64-byte aligned array (std::vector would zero 5 GB arrays serially).
template <class T>
struct Arr {
  T* p = nullptr;
  int64_t n = 0;
  bool own = true;   // false: non-owning view of caller memory
  Arr() = default;
  explicit Arr(int64_t n_) { alloc(n_); }
  Arr(const Arr&) = delete;
  Arr& operator=(const Arr&) = delete;
  Arr(Arr&& o) noexcept : p(o.p), n(o.n), own(o.own) { o.p = nullptr; o.n = 0; o.own = true; }
  Arr& operator=(Arr&& o) noexcept {
    if (this != &o) { release(); p = o.p; n = o.n; own = o.own; o.p = nullptr; o.n = 0; o.own = true; }
    return *this;
  }
  ~Arr() { release(); }
  void release() { if (p && own) free_host_array(p); p = nullptr; n = 0; own = true; }
  void view(const T* ptr, int64_t n_) { release(); p = const_cast<T*>(ptr); n = n_; own = false; }
  void alloc(int64_t n_) {
    release();
    n = n_;
    if (n_ > 0) {
      p = (T*)alloc_host_array((size_t)n_ * sizeof(T));
    }
  }
  void fill(T v) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) p[i] = v;
  }
  T& operator[](int64_t i) { return p[i]; }
  const T& operator[](int64_t i) const { return p[i]; }
  T* data() { return p; }
  const T* data() const { return p; }
  int64_t size() const { return n; }
  Arr clone() const {
    Arr c(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) c.p[i] = p[i];
    return c;
  }
};

// CSR with 64-bit row pointers, 32-bit column indices, fp64 values; columns sorted
// ascending within a row (PETSc AIJ convention).
struct Csr {
  int64_t m = 0, n = 0;
  Arr<int64_t> rp;
  Arr<int32_t> ci;
  Arr<double> v;
  // Bookkeeping of the setup accelerator (host/spops.hpp: SetupAccel): device_tag names the copy of this operator the
  // accelerator holds (0 = none); stub_nnz >= 0 means the arrays are NOT on the host yet - only the sizes are - and
  // SetupAccel::materialize() brings them over when a host routine needs them.  Moves keep both, clone() drops both.
  uint64_t device_tag = 0;
  int64_t stub_nnz = -1;
  int64_t nnz() const { return stub_nnz >= 0 ? stub_nnz : (m > 0 && rp.p ? rp[m] : 0); }
  bool on_host() const { return stub_nnz < 0; }
  bool empty() const { return rp.p == nullptr; }
  Csr clone() const {
    if (!on_host()) throw std::logic_error("clone of an operator that is not on the host");
    Csr c; c.m = m; c.n = n; c.rp = rp.clone(); c.ci = ci.clone(); c.v = v.clone();
    return c;
  }
};

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// exclusive prefix sum of counts[0..m) into rp[0..m]
inline void exclusive_scan(const int64_t* counts, int64_t m, int64_t* rp) {
  int nt = omp_get_max_threads();
  std::vector<int64_t> part(nt + 1, 0);
#pragma omp parallel num_threads(nt)
  {
    int t = omp_get_thread_num();
    int64_t lo = m * t / nt, hi = m * (t + 1) / nt;
    int64_t s = 0;
    for (int64_t i = lo; i < hi; i++) s += counts[i];
    part[t + 1] = s;
#pragma omp barrier
#pragma omp single
    { for (int i = 0; i < nt; i++) part[i + 1] += part[i]; }
    int64_t run = part[t];
    for (int64_t i = lo; i < hi; i++) { int64_t c = counts[i]; rp[i] = run; run += c; }
  }
  rp[m] = part[nt];
}

// Deterministic parallel sum of term(i), i in [lo, hi): fixed 8192-element blocks are summed
// independently and the block partials are added in block order, so the result depends neither
// on the thread count nor on scheduling (an OpenMP reduction clause combines the per-thread
// partials in arrival order).  The setup must be reproducible bit for bit: 1-ulp changes of the
// polynomial coefficients flip ties in the next level's CF splitting.
template <class F>
inline double det_sum(int64_t lo, int64_t hi, F term) {
  const int64_t B = 8192;
  const int64_t nb = hi > lo ? (hi - lo + B - 1) / B : 0;
  if (nb <= 1) { double s = 0.0; for (int64_t i = lo; i < hi; i++) s += term(i); return s; }
  std::vector<double> part((size_t)nb);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < nb; b++) {
    const int64_t a = lo + b * B, e = std::min(hi, a + B);
    double s = 0.0;
    for (int64_t i = a; i < e; i++) s += term(i);
    part[(size_t)b] = s;
  }
  double s = 0.0;
  for (int64_t b = 0; b < nb; b++) s += part[(size_t)b];
  return s;
}

// ---------------------------------------------------------------------------------------
// Counter-based random numbers (documented replacement for the reference's Fortran
// random_number streams, src/Gmres_Poly.F90:189-208, src/PMISR_Module.F90:207-222).
// U(seed,i) = top 53 bits of splitmix64(splitmix64(seed) + i), scaled to [0,1).
// ---------------------------------------------------------------------------------------
inline uint64_t splitmix64(uint64_t x) {
  uint64_t z = x + 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
inline double uniform01(uint64_t key, uint64_t i) {
  return (double)(splitmix64(key + i) >> 11) * (1.0 / 9007199254740992.0);
}
inline uint64_t rng_key(uint64_t seed) { return splitmix64(seed); }

// Constants from src/Pflare_Parameters.F90:204-226 (single literals widened to double).
inline double f32(double v) { return (double)(float)v; }
const double TOL_ZERO = (double)1e-12f;
const double TOL_RCOND = (double)1e-12f;
const double TOL_ARNOLDI = (double)1e-14f;
const double TOL_LUCKY = 1e-30;
const double TOL_LEJA_PERTURB = (double)5e-8f;
const double TOL_AUTO_TRUNCATE = (double)1e-14f;

enum { F_POINT = -1, C_POINT = 1 };

// PCPFLAREINVType (include/pflare.h of the reference)
enum InvType { INV_POWER = 0, INV_ARNOLDI = 1, INV_NEWTON = 2, INV_NEUMANN = 3, INV_SAI = 4,
               INV_ISAI = 5, INV_WJACOBI = 6, INV_JACOBI = 7, INV_NEWTON_NO_EXTRA = 8 };

}  // namespace airhip
