// Sparse building blocks of the host setup (OpenMP, row-parallel, deterministic).
#pragma once
#include <vector>
#include "common.hpp"

namespace airhip {

// tests/adv_1d.c:80-105
Csr gen_adv_1d(int64_t n);
// tests/adv_diff_fd.c:366-586; row = (k*N + j)*M + i restricted to the box
// [i0,i0+mi) x [j0,j0+mj) x [k0,k0+mk) when `perm` (natural -> output numbering, size M*N*P)
// is given (multi-GPU box numbering), natural otherwise.
struct FdOptions {
  int dim = 2;
  int64_t M = 11, N = 11, P = 11;
  double u = 1.0, v = 1.0, w = 1.0;
  bool unit_velocity = true;
  double alpha = 0.0;
  double L_x = 1.0, L_y = 1.0, L_z = 1.0;
  int adv_nondim = -1;  // -1: default (alpha == 0)
};
Csr gen_adv_diff_fd(const FdOptions& o, const int64_t* perm = nullptr);

// src/PETSc_Helper.F90:207-412
Csr remove_small(const Csr& A, double tol, int rel, bool lump, int drop_diag);

// src/SAbs.F90:16-145 reduced to what PMISR needs: the (optionally symmetrised) strength
// adjacency, diagonal removed, sorted columns.
void strength_graph(const Csr& A, double theta, bool symmetrize, Arr<int64_t>& rp, Arr<int32_t>& ci);

// Extract A[F,F], A[F,C], A[C,F] and optionally A[C,C] in one pass
// (MatCreateSubMatrix calls of src/AIR_MG_Setup.F90:405-451, src/AIR_Operators_Setup.F90:196-225).
// newidx[i] = position of i inside its own set (F or C), markers[i] = F_POINT / C_POINT.
void split_fc(const Csr& A, const int32_t* markers, const int32_t* newidx,
              const Arr<int32_t>& F, const Arr<int32_t>& C,
              Csr& A_ff, Csr& A_fc, Csr& A_cf, Csr* A_cc);

// C = A * B, row-wise Gustavson with the accumulation order of a sequential sparse
// accumulator (k ascending), sorted output columns.  Optionally negates the product and
// applies remove_small (relative to the row max, never dropping the diagonal, optional
// lumping) to each finished row before it is stored, which avoids materialising the large
// undropped product (Z = -A_cf M with r_drop, R A P with a_drop / a_lump).
struct SpgemmDrop { bool enabled = false; double tol = 0.0; bool lump = false; bool negate = false; };
Csr spgemm(const Csr& A, const Csr& B, const SpgemmDrop& drop = SpgemmDrop());

// Optional accelerator for the sparse products of the setup (implemented on the device in
// device/setup_accel.cpp).  Results must be bit-identical to spgemm() above.
struct SetupAccel {
  virtual ~SetupAccel() = default;
  virtual Csr spgemm(const Csr& A, const Csr& B, const SpgemmDrop& drop) = 0;
  // filter(R * (A * P)); the implementation may keep the result resident and recognise it when
  // it comes back as the A of the next level.  device_only: the caller can work with a result whose arrays stay on
  // the accelerator (Csr::stub_nnz >= 0) and will call materialize() before any host routine reads it.
  virtual Csr rap(const Csr& R, const Csr& A, const Csr& P, const SpgemmDrop& drop, bool device_only = false) = 0;
  // Z = filter(A * B) like spgemm(); the implementation may keep its copy of the result for rap_from_z
  virtual Csr spgemm_z(const Csr& A, const Csr& B, const SpgemmDrop& drop) { return spgemm(A, B, drop); }
  // filter(R * (A * P)) with R = [Z I] and the one-point P = [W; I] assembled where Z and the level operator already are
  // (no host copy of R / P, no upload of them); false when it declines (Z or A not resident) - the caller then builds R and P
  // and calls rap().  F / C: the level's index lists, wcol: the one-point prolongator's column per F point.
  virtual bool rap_from_z(const Csr& Z, const Arr<int32_t>& F, const Arr<int32_t>& C, const Arr<int32_t>& wcol, const Csr& A,
                          const SpgemmDrop& drop, Csr& coarse) {
    (void)Z; (void)F; (void)C; (void)wcol; (void)A; (void)drop; (void)coarse; return false;
  }
  // wcol[i] = column of the largest |entry| of row i of A_fc (air_setup.cpp one_point_W) where the accelerator holds the
  // block; false when it declines (the caller then needs A_fc on the host)
  virtual bool one_point_w(const Csr& A_fc, Arr<int32_t>& wcol) { (void)A_fc; (void)wcol; return false; }
  // brings the arrays of a device-only operator to the host (no-op for an operator that is there already)
  virtual void materialize(Csr& A) { if (!A.on_host()) throw std::logic_error("this accelerator holds no device-only operators"); }
  // the four F / C blocks of the level operator (host/spops.cpp: split_fc) extracted where the operator is; A_cc only
  // when asked for.  Returns false when it declines (the operator is not resident on the accelerator).
  virtual bool split_fc(const Csr& A, const Arr<int32_t>& markers, const Arr<int32_t>& newidx, int64_t n_f, int64_t n_c,
                        Csr& A_ff, Csr& A_fc, Csr& A_cf, Csr* A_cc) {
    (void)A; (void)markers; (void)newidx; (void)n_f; (void)n_c; (void)A_ff; (void)A_fc; (void)A_cf; (void)A_cc; return false;
  }
  // Auto-truncation test (src/AIR_MG_Setup.F90:158-256) evaluated by the accelerator: ||A M r - r|| / ||r|| for the
  // matrix-free Newton coarse solver M built from `seed`.  Returns false when it cannot do it (the host then does).
  // The value is only used to DECIDE (with a safety margin, see air_setup.cpp); the solver that ends up in the
  // hierarchy is always built by the host code, so the hierarchy stays bit-identical.
  virtual bool truncation_ratio(const Csr& A, int inverse_type, int poly_order, uint64_t seed, double& ratio) {
    (void)A; (void)inverse_type; (void)poly_order; (void)seed; (void)ratio; return false;
  }
  // CF splitting of the level operator by the accelerator (device/cfsplit.hip); markers must come out identical to
  // compute_cf_splitting.  Returns false when it declines (splitting types other than PMISR + DDC, small operators).
  virtual bool cf_splitting(const Csr& A, const struct AirOptions& o, Arr<int32_t>& markers) {
    (void)A; (void)o; (void)markers; return false;
  }
  // Assembled GMRES polynomial with the sparsity of A (sparsity order 1): M = c0 I + c1 A + sum c_t [A^t]_P
  // (host/poly.cpp::mat_mult_powers_share_sparsity).  Returns false when it declines.  final_values: the caller will not
  // change M afterwards (the accelerator may then keep its copy for the product that follows).
  virtual bool poly_assemble(const Csr& A, const std::vector<double>& c, Csr& M, bool final_values) { (void)A; (void)c; (void)M; (void)final_values; return false; }
  // Copies the accelerator kept of operators it produced or extracted (named by Csr::device_tag), for the upload of the
  // hierarchy that follows the setup on the same device: keep_copies(true, nnz of the top operator) asks for them (within
  // the accelerator's own memory budget), device_copy hands out the arrays of one (CSR, the host Csr's row order; false = none kept, the caller
  // copies the host arrays), drop_device_copy gives its memory back.
  virtual void keep_copies(bool on, double top_nnz = 0.0) { (void)on; (void)top_nnz; }
  virtual bool device_copy(uint64_t tag, int64_t m, int64_t nnz, const int64_t** rp, const int32_t** ci, const double** v) {
    (void)tag; (void)m; (void)nnz; (void)rp; (void)ci; (void)v; return false;
  }
  virtual void drop_device_copy(uint64_t tag) { (void)tag; }
  // defer_downloads(true): the caller can live with stubs (Csr::stub_nnz >= 0) for the operators no host routine of the
  // plain recipe reads and will fetch them with materialize() after the setup (the device copy is the only one until
  // then and is kept whatever happens); plain_recipe: set_air tells whether the options are that recipe.
  // release_copies: every kept copy whose operator is on the host goes back to the device memory pool.
  virtual void defer_downloads(bool on) { (void)on; }
  virtual void plain_recipe(bool on) { (void)on; }
  virtual void release_copies() {}
  // the setup proper is over: everything the accelerator still holds that is not a kept copy goes back to its memory pool
  virtual void release_temporaries() {}
  double t_transfer = 0.0, t_compute = 0.0, t_cfsplit = 0.0, t_poly = 0.0, t_extract = 0.0;
};
// the accelerator the polynomial assembly may use (set by setup_air for its duration; null = host only)
extern thread_local SetupAccel* g_setup_accel;

// The operator of the fused first F sweep with a one-point prolongator W (wcol[j] = W's column of F point j, -1 = none):
//   G = A_fc + A_ff W,   so that   b_f - A_fc x_c - A_ff (W x_c) = b_f - G x_c
// - the two products that open the reference's F-point relaxation (src/FC_Smooth.F90:533-548 after the prolongation
// src/AIR_MG_Setup.F90:1005) in one.  G is the F-row block of the A P product the setup forms for R A P; it is computed as
// [A_ff | A_fc] * [W; I] with spgemm(), so every entry is the Gustavson sum in the order of the row's entries (those of A_ff
// first) and the device twin (device/extract.hip: dev_fused_first_operator) gives the same bits.
Csr fused_first_operator(const Csr& A_ff, const Csr& A_fc, const int32_t* wcol);

void spmv(const Csr& A, const double* x, double* y);
// y = (A x) ./ d
void spmv_dscale(const Csr& A, const double* d, const double* x, double* y);

// src/PETSc_Helper.F90:930-1040
Csr copy_plus_diag(const Csr& A);
void get_diagonal(const Csr& A, double* d);
bool is_diag_only(const Csr& A);
Csr csr_from_diag(const double* d, int64_t n);
// scale columns: A(:,j) *= s[j]
void scale_cols(Csr& A, const double* s);
void scale_rows(Csr& A, const double* s);

}  // namespace airhip
