// Small dense linear algebra used by the polynomial-coefficient setup.  The reference
// calls LAPACK (gels, gelsd, geqrf, geev: src/Gmres_Poly.F90:245-304,681-773,
// src/Gmres_Poly_Newton.F90:491-569); no LAPACK is available to this build, so the few
// kernels needed are implemented here for the tiny (<= ~100 x 100) matrices involved.
#pragma once
#include "common.hpp"

namespace airhip {

// Column-major dense matrix helper
struct Dense {
  int m = 0, n = 0;
  std::vector<double> a;
  Dense() = default;
  Dense(int m_, int n_) : m(m_), n(n_), a((size_t)m_ * n_, 0.0) {}
  double& operator()(int i, int j) { return a[(size_t)j * m + i]; }
  double operator()(int i, int j) const { return a[(size_t)j * m + i]; }
};

// min ||A x - b||_2 for full-rank A (m >= n) by Householder QR  (LAPACK gels 'N')
std::vector<double> ls_qr(const Dense& A, const std::vector<double>& b);

// A x = b for square A by LU with partial pivoting (LAPACK gesv); a zero pivot leaves that unknown at 0
std::vector<double> lu_solve(const Dense& A, const std::vector<double>& b);

// inverse of a square matrix by one LU factorisation with partial pivoting (gesv with n right-hand sides)
Dense lu_inverse(const Dense& A);

// minimum-norm least squares by SVD, singular values <= rcond*s_max dropped
// (rcond < 0 -> machine epsilon)  (LAPACK gelsd)
std::vector<double> ls_minnorm_svd(const Dense& A, const std::vector<double>& b, double rcond);

// eigenvalues of an upper Hessenberg matrix (conjugate pairs adjacent, +imag first)
bool hessenberg_eig(Dense H, std::vector<double>& wr, std::vector<double>& wi);

// R factor (k x k, upper triangular) of the QR of the n x k matrix whose columns are
// cols[0..k) (overwritten).  (geqrf as used by src/TSQR.F90 in serial.)
Dense tall_qr_R(std::vector<Arr<double>>& cols, int64_t n);

}  // namespace airhip
