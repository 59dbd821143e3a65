// Sub-matrix extraction on the device; see extract.hpp.  One wave per source row: a counting pass, four exclusive scans,
// a filling pass in which the lanes of a 64-entry chunk find their places with ballots (the order inside a row is kept).
#include "extract.hpp"
#include "dev.hpp"
#include "kcommon.hpp"
#include "rb.hpp"
#include "devmem.hpp"
#include <stdexcept>
#include <algorithm>

namespace airhip {

namespace {

constexpr uint32_t FBIT = 0x80000000u;

struct Src { int64_t m; const int64_t* rp; const int32_t* ci; const double* v; };
struct Block { int64_t* rp; int32_t* ci; double* v; };      // rp doubles as the count array of the first pass

__global__ __launch_bounds__(BLOCK) void k_split_count(Src A, const uint32_t* __restrict__ code, Block ff, Block fc, Block cf, Block cc) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < A.m; i += nw) {
    const int64_t a0 = A.rp[i], a1 = A.rp[i + 1];
    int nf = 0;
    for (int64_t k = a0 + lane; k < a0 + ((a1 - a0 + WAVE - 1) / WAVE) * WAVE; k += WAVE) {
      const bool f = k < a1 && (code[A.ci[k]] & FBIT);
      nf += __popcll(__ballot(f));
    }
    if (lane == 0) {
      const uint32_t c = code[i];
      const int64_t r = c & ~FBIT, tot = a1 - a0;
      if (c & FBIT) { ff.rp[r] = nf; fc.rp[r] = tot - nf; }
      else { cf.rp[r] = nf; if (cc.rp) cc.rp[r] = tot - nf; }
    }
  }
}

__global__ __launch_bounds__(BLOCK) void k_split_fill(Src A, const uint32_t* __restrict__ code, Block ff, Block fc, Block cf, Block cc) {
  const int lane = threadIdx.x & (WAVE - 1);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < A.m; i += nw) {
    const int64_t a0 = A.rp[i], a1 = A.rp[i + 1];
    const uint32_t ci = code[i];
    const int64_t r = ci & ~FBIT;
    const Block& X = (ci & FBIT) ? ff : cf;      // F columns of this row
    const Block& Y = (ci & FBIT) ? fc : cc;      // C columns
    int64_t ox = X.rp[r], oy = Y.rp ? Y.rp[r] : 0;
    for (int64_t k = a0 + lane; k < a0 + ((a1 - a0 + WAVE - 1) / WAVE) * WAVE; k += WAVE) {
      const bool valid = k < a1;
      uint32_t c = 0; double v = 0.0;
      if (valid) { c = code[A.ci[k]]; v = A.v[k]; }
      const bool f = valid && (c & FBIT), g = valid && !(c & FBIT);
      const unsigned long long bf = __ballot(f), bg = __ballot(g);
      if (f) { const int64_t o = ox + __popcll(bf & lt); X.ci[o] = (int32_t)(c & ~FBIT); X.v[o] = v; }
      if (g && Y.rp) { const int64_t o = oy + __popcll(bg & lt); Y.ci[o] = (int32_t)c; Y.v[o] = v; }
      ox += __popcll(bf); oy += __popcll(bg);
    }
  }
}

template <class T> T* dalloc(int64_t n) { return (T*)dev_alloc(sizeof(T) * (size_t)(n > 0 ? n : 1)); }

// one wave per row of Z: its entries move over with their columns renumbered, the identity entry goes where its column
// C[j] belongs (F[.] ascends, so the entries below C[j] are a prefix of the row)
__global__ __launch_bounds__(BLOCK) void k_build_R(int64_t nc, const int64_t* __restrict__ zrp, const int32_t* __restrict__ zci,
                                                  const double* __restrict__ zv, const int32_t* __restrict__ F, const int32_t* __restrict__ C,
                                                  int64_t* __restrict__ rrp, int32_t* __restrict__ rci, double* __restrict__ rv) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t j = wave; j < nc; j += nw) {
    const int64_t a0 = zrp[j], a1 = zrp[j + 1], base = a0 + j;
    const int32_t cj = C[j];
    int below = 0;
    for (int64_t k0 = a0; k0 < a1; k0 += WAVE) {
      const int64_t k = k0 + lane;
      const bool valid = k < a1;
      const int32_t col = valid ? F[zci[k]] : 0;
      const bool lt = valid && col < cj;
      if (valid) { const int64_t o = base + (k - a0) + (lt ? 0 : 1); rci[o] = col; rv[o] = zv[k]; }
      below += __popcll(__ballot(lt));
    }
    if (lane == 0) { rrp[j] = base; rci[base + below] = cj; rv[base + below] = 1.0; if (j == nc - 1) rrp[nc] = a1 + nc; }
  }
}

__global__ __launch_bounds__(BLOCK) void k_p_count(int64_t n, const uint32_t* __restrict__ code, const int32_t* __restrict__ wcol, int64_t* __restrict__ cnt) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride) {
    const uint32_t c = code[r];
    cnt[r] = (c & FBIT) ? (wcol[c & ~FBIT] >= 0 ? 1 : 0) : 1;
  }
}
__global__ __launch_bounds__(BLOCK) void k_p_fill(int64_t n, const uint32_t* __restrict__ code, const int32_t* __restrict__ wcol,
                                                 const int64_t* __restrict__ rp, int32_t* __restrict__ ci, double* __restrict__ v) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < n; r += stride) {
    const uint32_t c = code[r];
    const int32_t col = (c & FBIT) ? wcol[c & ~FBIT] : (int32_t)c;
    if (col >= 0) { ci[rp[r]] = col; v[rp[r]] = 1.0; }
  }
}

// column of the largest |entry| of every row, the first one on ties, -1 for an empty row (host: one_point_W); one wave per row
__global__ __launch_bounds__(BLOCK) void k_one_point_w(int64_t m, const int64_t* __restrict__ rp, const int32_t* __restrict__ ci,
                                                      const double* __restrict__ v, int32_t* __restrict__ wcol) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < m; i += nw) {
    const int64_t a0 = rp[i], a1 = rp[i + 1];
    double best = -1.0; int64_t at = -1;               // this lane's entries in ascending position: the first maximum stays
    for (int64_t k = a0 + lane; k < a1; k += 64) { const double a = fabs(v[k]); if (a > best) { best = a; at = k; } }
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off, 64);
      const int64_t oa = __shfl_xor(at, off, 64);
      if (oa >= 0 && (at < 0 || ob > best || (ob == best && oa < at))) { best = ob; at = oa; }
    }
    if (lane == 0) wcol[i] = at >= 0 ? ci[at] : -1;
  }
}

}  // namespace

void dev_one_point_w(const DCsr& A_fc, int32_t* d_wcol, hipStream_t s) {
  if (A_fc.m == 0) return;
  const int g = (int)std::max<int64_t>(1, std::min<int64_t>((A_fc.m + 3) / 4, MAX_BLOCKS));
  hipLaunchKernelGGL(k_one_point_w, dim3(g), dim3(BLOCK), 0, s, A_fc.m, A_fc.rp, A_fc.ci, A_fc.v, d_wcol);
  HIP_CHECK(hipStreamSynchronize(s));
}

DCsr dev_build_R(const DCsr& Z, const int32_t* d_F, const int32_t* d_C, int64_t n, hipStream_t s) {
  DCsr R; R.m = Z.m; R.n = n; R.nnz = Z.nnz + Z.m;
  R.rp = dalloc<int64_t>(R.m + 1); R.ci = dalloc<int32_t>(R.nnz); R.v = dalloc<double>(R.nnz);
  if (R.m == 0) { const int64_t z = 0; HIP_CHECK(hipMemcpy(R.rp, &z, sizeof z, hipMemcpyHostToDevice)); return R; }
  hipLaunchKernelGGL(k_build_R, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, Z.m, Z.rp, Z.ci, Z.v, d_F, d_C, R.rp, R.ci, R.v);
  HIP_CHECK(hipStreamSynchronize(s));
  return R;
}

namespace {
__global__ __launch_bounds__(BLOCK) void k_cat_rp(int64_t m, const int64_t* __restrict__ a, const int64_t* __restrict__ b, int64_t* __restrict__ rp) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x; i <= m; i += stride) rp[i] = a[i] + b[i];
}
// one wave per row of [A | B]: A's entries, then B's with their columns moved by `shift`
__global__ __launch_bounds__(BLOCK) void k_cat_fill(Src A, Src B, int32_t shift, const int64_t* __restrict__ rp, int32_t* __restrict__ ci, double* __restrict__ v) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const int64_t nw = ((int64_t)gridDim.x * BLOCK) >> 6;
  for (int64_t i = wave; i < A.m; i += nw) {
    const int64_t a0 = A.rp[i], a1 = A.rp[i + 1], b0 = B.rp[i], b1 = B.rp[i + 1], o = rp[i];
    for (int64_t k = a0 + lane; k < a1; k += WAVE) { ci[o + (k - a0)] = A.ci[k]; v[o + (k - a0)] = A.v[k]; }
    const int64_t ob = o + (a1 - a0);
    for (int64_t k = b0 + lane; k < b1; k += WAVE) { ci[ob + (k - b0)] = B.ci[k] + shift; v[ob + (k - b0)] = B.v[k]; }
  }
}
// the code array of the index space [F points | C points] (dev_split_fc's convention)
__global__ __launch_bounds__(BLOCK) void k_cat_code(int64_t nf, int64_t nc, uint32_t* __restrict__ code) {
  const int64_t stride = (int64_t)gridDim.x * BLOCK;
  for (int64_t r = (int64_t)blockIdx.x * BLOCK + threadIdx.x; r < nf + nc; r += stride) code[r] = r < nf ? ((uint32_t)r | FBIT) : (uint32_t)(r - nf);
}
}  // namespace

DCsr dev_fused_first_operator(const DCsr& A_ff, const DCsr& A_fc, const int32_t* d_wcol, hipStream_t s) {
  const int64_t nf = A_fc.m, nc = A_fc.n;
  if (A_ff.m != nf || A_ff.n != nf) throw std::runtime_error("dev_fused_first_operator: A_ff / A_fc shape mismatch");
  DCsr cat; cat.m = nf; cat.n = nf + nc; cat.nnz = A_ff.nnz + A_fc.nnz;
  cat.rp = dalloc<int64_t>(nf + 1); cat.ci = dalloc<int32_t>(cat.nnz); cat.v = dalloc<double>(cat.nnz);
  const int g1 = (int)std::max<int64_t>(1, std::min<int64_t>((nf + 1 + BLOCK - 1) / BLOCK, MAX_BLOCKS));
  hipLaunchKernelGGL(k_cat_rp, dim3(g1), dim3(BLOCK), 0, s, nf, A_ff.rp, A_fc.rp, cat.rp);
  if (nf > 0) {
    const int g2 = (int)std::max<int64_t>(1, std::min<int64_t>((nf + 3) / 4, MAX_BLOCKS));
    hipLaunchKernelGGL(k_cat_fill, dim3(g2), dim3(BLOCK), 0, s, Src{nf, A_ff.rp, A_ff.ci, A_ff.v}, Src{nf, A_fc.rp, A_fc.ci, A_fc.v}, (int32_t)nf, cat.rp, cat.ci, cat.v);
  }
  uint32_t* code = dalloc<uint32_t>(nf + nc);
  const int g3 = (int)std::max<int64_t>(1, std::min<int64_t>((nf + nc + BLOCK - 1) / BLOCK, MAX_BLOCKS));
  hipLaunchKernelGGL(k_cat_code, dim3(g3), dim3(BLOCK), 0, s, nf, nc, code);
  DCsr P = dev_build_P_one_point(nf + nc, nc, code, d_wcol, s);
  dev_free(code);
  DCsr G = dev_spgemm(cat, P, false, 0.0, false, false, s);
  dcsr_free(cat); dcsr_free(P);
  return G;
}

DCsr dev_build_P_one_point(int64_t n, int64_t n_c, const uint32_t* d_code, const int32_t* d_wcol, hipStream_t s) {
  DCsr P; P.m = n; P.n = n_c;
  int64_t* cnt = dalloc<int64_t>(n);
  P.rp = dalloc<int64_t>(n + 1);
  const int g = (int)std::min<int64_t>((n + BLOCK - 1) / BLOCK, MAX_BLOCKS);
  hipLaunchKernelGGL(k_p_count, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, n, d_code, d_wcol, cnt);
  P.nnz = device_exclusive_scan(n, cnt, P.rp, s);
  HIP_CHECK(hipMemcpyAsync(P.rp + n, &P.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipStreamSynchronize(s));
  dev_free(cnt);
  P.ci = dalloc<int32_t>(P.nnz); P.v = dalloc<double>(P.nnz);
  hipLaunchKernelGGL(k_p_fill, dim3(g < 1 ? 1 : g), dim3(BLOCK), 0, s, n, d_code, d_wcol, P.rp, P.ci, P.v);
  HIP_CHECK(hipStreamSynchronize(s));
  return P;
}

void dev_split_fc(const DCsr& A, const uint32_t* d_code, int64_t n_f, int64_t n_c, DCsr& A_ff, DCsr& A_fc, DCsr& A_cf, DCsr* A_cc,
                  hipStream_t s) {
  if (A.m != A.n || n_f + n_c != A.m) throw std::runtime_error("dev_split_fc: shape mismatch");
  const Src src{A.m, A.rp, A.ci, A.v};
  struct Out { DCsr* d; int64_t m, n; int64_t* cnt; } outs[4] = {{&A_ff, n_f, n_f, nullptr}, {&A_fc, n_f, n_c, nullptr}, {&A_cf, n_c, n_f, nullptr},
                                                              {A_cc, n_c, n_c, nullptr}};
  Block blk[4] = {};
  for (int b = 0; b < 4; b++) {
    if (!outs[b].d) continue;
    *outs[b].d = DCsr(); outs[b].d->m = outs[b].m; outs[b].d->n = outs[b].n;
    outs[b].cnt = dalloc<int64_t>(outs[b].m);
    blk[b].rp = outs[b].cnt;
  }
  hipLaunchKernelGGL(k_split_count, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, src, d_code, blk[0], blk[1], blk[2], blk[3]);
  for (int b = 0; b < 4; b++) {
    if (!outs[b].d) continue;
    DCsr& X = *outs[b].d;
    X.rp = dalloc<int64_t>(X.m + 1);
    X.nnz = device_exclusive_scan(X.m, outs[b].cnt, X.rp, s);
    HIP_CHECK(hipMemcpyAsync(X.rp + X.m, &X.nnz, sizeof(int64_t), hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));                     // &X.nnz is read by the copy
    dev_free(outs[b].cnt);
    X.ci = dalloc<int32_t>(X.nnz); X.v = dalloc<double>(X.nnz);
    blk[b] = Block{X.rp, X.ci, X.v};
  }
  hipLaunchKernelGGL(k_split_fill, dim3(MAX_BLOCKS), dim3(BLOCK), 0, s, src, d_code, blk[0], blk[1], blk[2], blk[3]);
  HIP_CHECK(hipStreamSynchronize(s));
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(e) + " in the device sub-matrix extraction");
}

}  // namespace airhip
