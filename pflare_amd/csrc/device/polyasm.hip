// Fixed-sparsity assembly of a GMRES polynomial inverse on the device (reference: the CPU routine
// mat_mult_powers_share_sparsity_cpu src/Gmres_Poly.F90:920-1337 and its device twin src/Gmres_Polyk.kokkos.cxx):
//
//     M = c_0 I + c_1 A + sum_{t >= 2} c_t [A^t]_P ,      P = pattern(A)  (sparsity order 1, the default)
//
// where every power is truncated to P row by row:  row_i([A^t]_P) = sum_{j in P(i)} [A^(t-1)]_P(i,j) * A(j, P(i)).
// One WAVE builds one row: P(i) sits in LDS with a small hash map column -> position, the previous power's row is a
// vector over P(i); for j ascending the wave sweeps row j of A with its 64 lanes (distinct columns, so no two lanes
// hit the same position in one step) and adds prev_j * A(j,c) to the position of c if c is in P(i).  Every entry is
// therefore accumulated in exactly the order of the host code (host/poly.cpp: mat_mult_powers_share_sparsity), with
// separate multiply and add roundings: the result is bit-identical (tests/test_gpu_setup.py).
#include "dev.hpp"
#include "kcommon.hpp"
#include "rb.hpp"
#include "spgemm.hpp"
#include "polyasm.hpp"
#include "devmem.hpp"
#include <vector>
#include <stdexcept>
#include <algorithm>

namespace airhip {

namespace {

constexpr int MAXC = 16;                  // coefficients (polynomial order <= 15)
struct Coefs { double c[MAXC]; int n; };
constexpr int32_t EMPTYK = -1;

__device__ __forceinline__ uint32_t hcol(int32_t c) { return (uint32_t)c * 2654435761u; }

// rows per bin / diagonal presence: cnt[i] = len_i + (row i holds its diagonal ? 0 : 1)
__global__ __launch_bounds__(BLOCK) void k_poly_count(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                     int64_t* __restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i < m; i += stride) {
    int64_t lo = rp[i], hi = rp[i + 1];
    const int64_t e = hi;
    while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (ci[mid] < (int32_t)i) lo = mid + 1; else hi = mid; }
    const bool has = lo < e && ci[lo] == (int32_t)i;
    cnt[i] = (e - rp[i]) + (has ? 0 : 1);
  }
}

__device__ __forceinline__ int lane_i32(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ int64_t lane_i64(int64_t v, int l) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(v & 0xffffffffll), l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l);
  return (int64_t)(((uint64_t)hi << 32) | lo);
}

// CAP: longest row the bin handles; HT = 2 * CAP hash slots.  WPB waves per workgroup.
// The sweep over (j in P(i), 64-entry chunk of row j of A) is a scalar walk - lane e of a chunk of P(i) holds the
// extent of row cols[e] and prev[e], the current row's base / length / factor come through readlane - and PIPE steps
// form a batch: their loads are issued together, their columns are looked up in the hash map together (read-only
// probes), then the products are added with ds_add_f64 in step order, which is the order of the host code.
template <int CAP, int WPB>
__global__ __launch_bounds__(WPB * WAVE) void k_poly_rows(int64_t nrows, const int32_t* __restrict__ rows, const int64_t* __restrict__ rp,
                                                         const int32_t* __restrict__ ci, const double* __restrict__ v, Coefs cf,
                                                         const int64_t* __restrict__ orp, int32_t* __restrict__ oci, double* __restrict__ ov) {
#pragma clang fp contract(off)   // a*b + c rounds twice, as the host code does
  constexpr int HT = 2 * CAP;
  constexpr int PIPE = 8;
  __shared__ int32_t s_cols[WPB][CAP];
  __shared__ int32_t s_hkey[WPB][HT];
  __shared__ uint16_t s_hpos[WPB][HT];
  __shared__ int64_t s_rb0[WPB][CAP];
  __shared__ int32_t s_rlen[WPB][CAP];
  __shared__ double s_prev[WPB][CAP], s_cur[WPB][CAP + WAVE], s_acc[WPB][CAP];
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & (WAVE - 1);
  int32_t* cols = s_cols[wid]; int32_t* hkey = s_hkey[wid]; uint16_t* hpos = s_hpos[wid];
  int64_t* rb0 = s_rb0[wid]; int32_t* rlen = s_rlen[wid];
  double* prev = s_prev[wid]; double* cur = s_cur[wid]; double* acc = s_acc[wid];
  auto sync = [] { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  const int64_t nw = (int64_t)gridDim.x * WPB;
  for (int64_t w = (int64_t)blockIdx.x * WPB + wid; w < nrows; w += nw) {
    const int64_t i = rows[w];
    const int64_t p0 = uniform64(rp[i]);
    const int nc = (int)(uniform64(rp[i + 1]) - p0);
    // P(i), the extents of the rows it names, the previous power (= the row of A itself), the accumulator, the hash map
    for (int h = lane; h < HT; h += WAVE) hkey[h] = EMPTYK;
    for (int q = lane; q < nc; q += WAVE) {
      const int32_t c = ci[p0 + q];
      cols[q] = c; prev[q] = v[p0 + q]; acc[q] = 0.0;
      const int64_t b = rp[c];
      rb0[q] = b; rlen[q] = (int32_t)(rp[c + 1] - b);
    }
    sync();
    for (int q = lane; q < nc; q += WAVE) {
      const int32_t c = cols[q];
      uint32_t h = hcol(c) & (HT - 1);
      for (;;) {
        // the columns of a row are distinct: a plain compare-and-swap claims the slot
        const int32_t old = atomicCAS(&hkey[h], EMPTYK, c);
        if (old == EMPTYK) { hpos[h] = (uint16_t)q; break; }
        h = (h + 1) & (HT - 1);
      }
    }
    sync();
    for (int term = 2; term < cf.n; term++) {
      for (int q = lane; q < nc + WAVE; q += WAVE) cur[q] = 0.0;       // [nc, nc + 64): the dump slots of lanes without a hit
      sync();
      for (int jb = 0; jb < nc; jb += WAVE) {
        const int nk = nc - jb < WAVE ? nc - jb : WAVE;
        int64_t b0 = 0; int len = 0; double pj = 0.0;
        if (lane < nk) { b0 = rb0[jb + lane]; len = rlen[jb + lane]; pj = prev[jb + lane]; }
        const long long pj_bits = __double_as_longlong(pj);
        int e = 0, off = 0;
        int cur_len = lane_i32(len, 0);
        int64_t cur_b0 = lane_i64(b0, 0);
        double cur_p = __longlong_as_double(lane_i64(pj_bits, 0));
        while (e < nk) {
          int32_t cc[PIPE]; double x[PIPE], pv[PIPE]; bool act[PIPE]; uint32_t hh[PIPE]; int pos[PIPE];
#pragma unroll
          for (int d = 0; d < PIPE; d++) {
            act[d] = false; cc[d] = EMPTYK; x[d] = 0.0; pv[d] = 0.0;
            if (e < nk) {                                   // wave-uniform
              const uint32_t o = (uint32_t)(off + lane);
              act[d] = o < (uint32_t)cur_len;
              pv[d] = cur_p;
              if (act[d]) { cc[d] = (ci + cur_b0)[o]; x[d] = (v + cur_b0)[o]; }
              off += WAVE;
              if (off >= cur_len) {
                e++; off = 0;
                if (e < nk) { cur_len = lane_i32(len, e); cur_b0 = lane_i64(b0, e); cur_p = __longlong_as_double(lane_i64(pj_bits, e)); }
              }
            }
          }
#pragma unroll
          for (int d = 0; d < PIPE; d++) { x[d] = pv[d] * x[d]; hh[d] = hcol(cc[d]) & (HT - 1); pos[d] = nc + lane; }
          // read-only probes of all steps together; a lane without an entry carries EMPTYK, which ends its probe at once
          for (;;) {
            int32_t k[PIPE];
#pragma unroll
            for (int d = 0; d < PIPE; d++) k[d] = hkey[hh[d]];
            bool more = false;
#pragma unroll
            for (int d = 0; d < PIPE; d++) {
              const bool hit = k[d] == cc[d] && act[d];
              const bool miss = k[d] != cc[d] && k[d] != EMPTYK && act[d];
              if (hit) pos[d] = hpos[hh[d]];
              act[d] = act[d] && miss;                      // still looking
              hh[d] = miss ? ((hh[d] + 1) & (HT - 1)) : hh[d];
              more |= miss;
            }
            if (!__any(more)) break;
          }
#pragma unroll
          for (int d = 0; d < PIPE; d++)
            (void)__hip_atomic_fetch_add(&cur[pos[d]], pos[d] < nc ? x[d] : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
          sync();
        }
      }
      const double ct = cf.c[term];
      if (ct != 0.0) for (int q = lane; q < nc; q += WAVE) acc[q] += ct * cur[q];
      for (int q = lane; q < nc; q += WAVE) prev[q] = cur[q];
      sync();
    }
    // row of M: pattern P(i) plus the diagonal; value = A_ik * c_1 (+ c_0 on the diagonal) + acc
    const int64_t o = orp[i];
    const bool grows = (int)(orp[i + 1] - o) != nc;          // the diagonal is not in P(i)
    int d = 0;                                               // its insertion position
    if (grows) {
      for (int q = lane; q < nc; q += WAVE) d += cols[q] < (int32_t)i ? 1 : 0;
      for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off, WAVE);
    }
    for (int q = lane; q < nc; q += WAVE) {
      const int32_t c = cols[q];
      double t = v[p0 + q] * cf.c[1];
      if (c == (int32_t)i) t = t + cf.c[0];
      t = t + acc[q];
      const int64_t pos = o + q + ((grows && q >= d) ? 1 : 0);
      oci[pos] = c; ov[pos] = t;
    }
    if (grows && lane == 0) { oci[o + d] = (int32_t)i; ov[o + d] = 0.0 * cf.c[1] + cf.c[0]; }
    sync();
  }
}

template <class T> T* dalloc(int64_t n) { return (T*)dev_alloc(sizeof(T) * (size_t)(n > 0 ? n : 1)); }

}  // namespace

bool dev_poly_assemble(const DCsr& A, const double* c, int nterms, DCsr& M, hipStream_t s) {
  if (nterms < 3 || nterms > MAXC || A.m != A.n) return false;
  const int64_t m = A.m;
  // row lengths decide the bins; a row longer than the large bin's capacity sends the whole matrix back to the host
  std::vector<int64_t> rp(m + 1);
  HIP_CHECK(hipMemcpyAsync(rp.data(), A.rp, sizeof(int64_t) * (size_t)(m + 1), hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  // rows by length: the longer the row, the larger its wave's share of LDS and the fewer waves a CU holds (20 / 10 / 6 / 3)
  constexpr int NBIN = 4;
  const int cap[NBIN] = {128, 256, 512, 1024};
  std::vector<int32_t> bin[NBIN];
  bin[0].reserve(m);
  for (int64_t i = 0; i < m; i++) {
    const int64_t len = rp[i + 1] - rp[i];
    if (len > cap[NBIN - 1]) return false;
    int k = 0; while (len > cap[k]) k++;
    bin[k].push_back((int32_t)i);
  }
  Coefs cf; cf.n = nterms;
  for (int k = 0; k < MAXC; k++) cf.c[k] = k < nterms ? c[k] : 0.0;
  M = DCsr(); M.m = m; M.n = m;
  int64_t* cnt = dalloc<int64_t>(m);
  M.rp = dalloc<int64_t>(m + 1);
  const int g1 = (int)std::max<int64_t>(1, std::min<int64_t>((m + BLOCK - 1) / BLOCK, 65535));
  hipLaunchKernelGGL(k_poly_count, dim3(g1), dim3(BLOCK), 0, s, m, A.rp, A.ci, cnt);
  M.nnz = device_exclusive_scan(m, cnt, M.rp, s);
  HIP_CHECK(hipMemcpyAsync(M.rp + m, &M.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  dev_free(cnt);
  M.ci = dalloc<int32_t>(M.nnz); M.v = dalloc<double>(M.nnz);
  int32_t* d_rows[NBIN];
  for (int k = 0; k < NBIN; k++) {
    d_rows[k] = dalloc<int32_t>((int64_t)bin[k].size());
    if (!bin[k].empty()) HIP_CHECK(hipMemcpyAsync(d_rows[k], bin[k].data(), sizeof(int32_t) * bin[k].size(), hipMemcpyHostToDevice, s));
  }
  auto grid = [](size_t rows, int wpb, int per_cu) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>(((int64_t)rows + wpb - 1) / wpb, 256 * per_cu * 4))); };
  if (!bin[0].empty()) hipLaunchKernelGGL((k_poly_rows<128, 4>), grid(bin[0].size(), 4, 5), dim3(4 * WAVE), 0, s, (int64_t)bin[0].size(), d_rows[0], A.rp, A.ci, A.v, cf, M.rp, M.ci, M.v);
  if (!bin[1].empty()) hipLaunchKernelGGL((k_poly_rows<256, 2>), grid(bin[1].size(), 2, 5), dim3(2 * WAVE), 0, s, (int64_t)bin[1].size(), d_rows[1], A.rp, A.ci, A.v, cf, M.rp, M.ci, M.v);
  if (!bin[2].empty()) hipLaunchKernelGGL((k_poly_rows<512, 1>), grid(bin[2].size(), 1, 6), dim3(WAVE), 0, s, (int64_t)bin[2].size(), d_rows[2], A.rp, A.ci, A.v, cf, M.rp, M.ci, M.v);
  if (!bin[3].empty()) hipLaunchKernelGGL((k_poly_rows<1024, 1>), grid(bin[3].size(), 1, 3), dim3(WAVE), 0, s, (int64_t)bin[3].size(), d_rows[3], A.rp, A.ci, A.v, cf, M.rp, M.ci, M.v);
  HIP_CHECK(hipStreamSynchronize(s));
  const hipError_t e = hipGetLastError();
  for (int k = 0; k < NBIN; k++) dev_free(d_rows[k]);
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e) + " in the device polynomial assembly");
  return true;
}

}  // namespace airhip
