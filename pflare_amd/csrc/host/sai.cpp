// Sparse approximate inverses by small dense solves: lAIR / SAI restrictors and the SAI / ISAI
// approximate inverses (src/SAI_Z.F90:24-645, 781-863).  For every row i with prescribed pattern J
//   incomplete (lAIR, ISAI):  z A_ff(J,J) = -A_cf(i,J)                       (square, LU)
//   complete   (SAI):         min || z A_ff(J,I) + A_cf(i,I) ||_2, I = columns seen by rows J   (QR)
// The reference switches to an inexact Krylov solve (rtol 1e-3) when a block is larger than 40; like its
// device path (calculate_and_build_sai_z_kokkos) this build always solves the blocks directly.
#include "air.hpp"
#include "dense.hpp"

namespace airhip {

Csr sai_z(const Csr& A_ff, const Csr* A_cf, const Csr& S, bool incomplete) {
  Csr Z; Z.m = S.m; Z.n = S.n;
  Z.rp.alloc(S.m + 1); Z.ci.alloc(S.nnz()); Z.v.alloc(S.nnz());
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i <= S.m; i++) Z.rp[i] = S.rp[i];
#pragma omp parallel for schedule(static)
  for (int64_t k = 0; k < S.nnz(); k++) { Z.ci[k] = S.ci[k]; Z.v[k] = 0.0; }
#pragma omp parallel
  {
    std::vector<int32_t> I;
    std::vector<double> jv, e;
#pragma omp for schedule(dynamic, 64)
    for (int64_t i = 0; i < S.m; i++) {
      const int64_t p0 = S.rp[i];
      const int js = (int)(S.rp[i + 1] - p0);
      if (js == 0) continue;                       // a C point without F neighbours
      const int32_t* J = S.ci.data() + p0;
      // the row of A_cf (or of -I) on the pattern
      jv.assign(js, 0.0);
      if (A_cf) {
        int64_t a = A_cf->rp[i], ae = A_cf->rp[i + 1]; int b = 0;
        while (a < ae && b < js) {
          if (A_cf->ci[a] == J[b]) { jv[b] = A_cf->v[a]; a++; b++; }
          else if (A_cf->ci[a] < J[b]) a++; else b++;
        }
      } else {
        for (int b = 0; b < js; b++) if (J[b] == i) jv[b] = -1.0;
      }
      // the shadow I
      if (incomplete) I.assign(J, J + js);
      else {
        I.clear();
        for (int b = 0; b < js; b++)
          for (int64_t k = A_ff.rp[J[b]]; k < A_ff.rp[J[b] + 1]; k++) I.push_back(A_ff.ci[k]);
        std::sort(I.begin(), I.end());
        I.erase(std::unique(I.begin(), I.end()), I.end());
      }
      const int is = (int)I.size();
      // B = A_ff(J, I)^T
      Dense B(is, js);
      for (int b = 0; b < js; b++) {
        int64_t a = A_ff.rp[J[b]], ae = A_ff.rp[J[b] + 1]; int q = 0;
        while (a < ae && q < is) {
          if (A_ff.ci[a] == I[q]) { B(q, b) = A_ff.v[a]; a++; q++; }
          else if (A_ff.ci[a] < I[q]) a++; else q++;
        }
      }
      e.assign(is, 0.0);
      { int q = 0; for (int b = 0; b < js && q < is;) { if (I[q] == J[b]) { e[q] = -jv[b]; q++; b++; } else if (I[q] < J[b]) q++; else b++; } }
      if (is < js) continue;                       // under-determined block (rows J without a diagonal): left at zero
      std::vector<double> z = incomplete ? lu_solve(B, e) : ls_qr(B, e);
      for (int b = 0; b < js; b++) Z.v[p0 + b] = z[b];
    }
  }
  return Z;
}

// calculate_and_build_sai (src/SAI_Z.F90:781-863): pattern of A^sparsity_order (the diagonal for order 0)
Csr sai_inverse(const Csr& A, int sparsity_order, bool incomplete) {
  if (sparsity_order == 0) {
    Arr<double> one(A.m); one.fill(1.0);
    Csr S = csr_from_diag(one.data(), A.m);
    return sai_z(A, nullptr, S, incomplete);
  }
  if (sparsity_order == 1) return sai_z(A, nullptr, A, incomplete);
  Csr Pw = spgemm(A, A);
  for (int o = 3; o <= sparsity_order; o++) Pw = spgemm(A, Pw);
  return sai_z(A, nullptr, Pw, incomplete);
}

}  // namespace airhip
