// PCAIR host-side state: options (the -pc_air_* subset), the hierarchy the solve phase
// reads (counterpart of air_multigrid_data, src/AIR_Data_Type.F90:284-360) and the
// setup entry points (counterpart of setup_air_pcmg, src/AIR_MG_Setup.F90:44-1231).
#pragma once
#include "common.hpp"
#include "spops.hpp"

namespace airhip {

// -pc_air_* options (src/AIR_Data_Type.F90:34-264).  Defaults reproduce the reference's,
// including single-precision literals widened to double (r_drop, a_drop, ddc_fraction).
struct AirOptions {
  int max_levels = 300;
  int64_t coarse_eq_limit = 6;
  int auto_truncate_start_level = -1;
  double auto_truncate_tol = (double)1e-14f;
  double strong_threshold = 0.5;
  int ddc_its = 1;
  double ddc_fraction = (double)0.1f;
  int max_luby_steps = -1;
  int smooth_order[10] = {2, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // "ff"
  bool diag_scale_polys = false;
  bool matrix_free_polys = false;
  bool one_point_classical_prolong = true;
  bool symmetric = false;
  int inverse_type = INV_ARNOLDI;
  int poly_order = 6;
  int inverse_sparsity_order = 1;
  int c_inverse_type = INV_ARNOLDI;
  int c_poly_order = 6;
  int c_inverse_sparsity_order = 1;
  int coarsest_inverse_type = INV_ARNOLDI;
  int coarsest_poly_order = 6;
  int coarsest_inverse_sparsity_order = 1;
  bool coarsest_matrix_free_polys = false;
  bool coarsest_diag_scale_polys = false;
  double r_drop = (double)0.01f;
  double a_drop = (double)1e-4f;
  bool a_lump = false;
  double strong_r_threshold = 0.0;   // -pc_air_strong_r_threshold: drop tolerance applied to A before Z is built
  int z_type = 0;                    // -pc_air_z_type: 0 product (AIRG), 1 lair, 2 lair_sai (src/Pflare_Parameters.F90:30-32)
  int lair_distance = 2;             // -pc_air_lair_distance
  int improve_z_its = 0, improve_w_its = 0;   // -pc_air_improve_z_its / -pc_air_improve_w_its
  // -pc_air_full_smoothing_up_and_down: multiplicative V-cycle with one sweep of x += M (b - A x) on ALL points down
  // and up instead of the F/C relaxation on the way up (src/AIR_MG_Setup.F90:1014-1079); M = approximate inverse
  // of the whole level operator, kept in inv_ff as the reference keeps it in inv_A_ff
  bool full_smoothing = false;
  int cf_splitting_type = 0;       // -pc_air_cf_splitting_type: 0 pmisr_ddc, 1 diag_dom, 2 cr (src/CF_Splitting.F90:19-25)
  int coarse_richardson_its = 0;  // -mg_coarse_ksp_type richardson -mg_coarse_ksp_max_it N
  bool coarse_lu = false;         // -mg_coarse_pc_type lu: exact coarsest solve (explicit dense inverse, small systems only)
  bool print_stats_timings = false;
  uint64_t seed = 1;              // comm_rank + 1 in the reference

  bool any_c_smooths() const { for (int s : smooth_order) if (s < 0) return true; return false; }
  // src/PCAIR_Interfaces.F90:1821-1882
  int set_smooth_type(const char* s);
};

// An approximate inverse as the solve phase applies it (counterpart of the assembled Mat /
// MatShell + mat_ctxtype pair, src/Matshell_Data_Type.F90:19-30).
struct InvOp {
  enum Kind { NONE = 0, ASSEMBLED = 1, DIAG = 2, HORNER = 3, NEWTON = 4 } kind = NONE;
  Csr mat;                       // ASSEMBLED
  Arr<double> d;                 // DIAG: the diagonal of the inverse
  std::vector<double> coeffs;    // HORNER: monomial coefficients, constant first
  std::vector<double> re, im;    // NEWTON: roots
  bool diag_scale = false;       // q(D^-1 A) D^-1
  bool neumann = false;          // scaled operator is I - D^-1 A
  int spmvs = 0;                 // SpMVs per application (matrix-free kinds)
};

struct Level {
  int64_t n = 0, n_f = 0, n_c = 0;
  Arr<int32_t> F, C;             // ascending level-local indices (IS_fine_index / IS_coarse_index)
  bool aff_diag = false;         // A_ff converted to MATDIAGONAL
  Arr<double> aff_d;             // its diagonal when aff_diag
  Csr A_ff, A_fc, A_cf, A_cc;    // A_cf kept for C smoothing / export; A_cc only with C smoothing
  InvOp inv_ff, inv_cc;
  Csr Z;                         // n_c x n_f (R = [Z I])
  bool one_point = true;
  Arr<int32_t> wcol;             // one-point W: coarse column per F row (-1: none)
  Csr W;                         // general W (n_f x n_c) when !one_point
  // Only in hierarchies that are cut for a multi-GPU run (and in the pieces): G = A_fc + A_ff W, the operator of the fused
  // first F sweep (host/spops.hpp: fused_first_operator), and - pieces only - W's columns in the numbering of G's column
  // space [local | ghosts of G].  On one device the upload builds G from the device copies of A_fc and A_ff instead.
  Csr G;
  Arr<int32_t> wcol_g;
  // the cycle may run its first F sweep on G: relaxation that starts with F sweeps, one-point W, sparse A_ff, an A_ff inverse that
  // is a linear operator applied as such (assembled, or a matrix-free polynomial in A_ff)
  bool fuses_first_sweep() const {
    const bool lin = inv_ff.kind == InvOp::ASSEMBLED || inv_ff.kind == InvOp::HORNER || inv_ff.kind == InvOp::NEWTON;
    return one_point && !aff_diag && smooth_order[0] > 0 && lin && A_fc.nnz() > 0 && A_ff.nnz() > 0 && !full;
  }
  int smooth_order[10] = {0};
  int64_t nnz_A = 0;             // nnz of the level operator (for complexities)
  bool full = false;             // full up / down smoothing: inv_ff is n x n and A_full holds the level operator
  Csr A_full;
};

struct Hierarchy {
  AirOptions opt;
  std::vector<std::unique_ptr<Level>> levels;   // levels.size() = no_levels - 1
  Csr A_top;                     // copy of the fine operator is NOT held; see pcair.cpp
  Csr coarse_A;                  // operator on the coarsest level
  InvOp coarse_inv;              // inv_A_ff(no_levels)
  bool single_level_jacobi = false;
  std::vector<double> timers;    // cumulative setup timers (src/Timers.F90), seconds
  int no_levels() const { return (int)levels.size() + 1; }
};

// CF splitting (src/CF_Splitting.F90:235-392, CF_PMISR_DDC)
void compute_cf_splitting(const Csr& A, const AirOptions& o, Arr<int32_t>& markers);

// Approximate inverse construction (src/Approx_Inverse_Setup.F90:207-500).
// A is either a CSR or (diag != nullptr) a MATDIAGONAL of size n.
struct OpView { const Csr* A = nullptr; const double* diag = nullptr; int64_t n = 0; };
void build_inverse(const OpView& A, int inverse_type, int poly_order, int sparsity_order,
                   bool matrix_free, bool diag_scale, uint64_t seed, InvOp& op, InvOp* assembled_too);

// lAIR / SAI rows by small dense solves (src/SAI_Z.F90): Z on the pattern of S; A_cf == nullptr stands for -I
Csr sai_z(const Csr& A_ff, const Csr* A_cf, const Csr& S, bool incomplete);
Csr sai_inverse(const Csr& A, int sparsity_order, bool incomplete);

// Pieces of the Newton-basis construction for an Arnoldi process that runs elsewhere (device auto-truncation test):
// the start vector of the Krylov space and the harmonic-Ritz / Leja roots from its Hessenberg matrix
// (column-major, (poly_order + 2) x (poly_order + 1)).
void krylov_start_vector(uint64_t seed, int64_t n, double* out);
void newton_roots_from_hessenberg(const double* H, double beta, int m, int poly_order, bool add_roots,
                                  std::vector<double>& re, std::vector<double>& im);

// Full setup.  A is consumed level by level (the caller's matrix is not modified).
std::unique_ptr<Hierarchy> setup_air(const Csr& A, const AirOptions& o, SetupAccel* accel = nullptr);

// y = op(x) on the host (used by the auto-truncation test of the setup only)
void host_apply_inv(const InvOp& op, const OpView& A, const double* x, double* y);

}  // namespace airhip
