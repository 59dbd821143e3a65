// Device-side data structures and kernel launchers of the PCAIR solve phase (gfx950).
//
// Layout decisions:
//  * matrices live in HBM as row-block CSR: CSR values / index words without padding, rows in vector
//    order, plus a partition of the rows into blocks of <= 512 non-zeros and <= 128 rows (a longer row is
//    a block of its own).  One wavefront owns exactly one block (spmv.hip).
//  * level vectors use a hierarchical F/C ordering: the level-l vector is [x_F | x_C] and
//    x_C *is* the level-(l+1) vector, so the level-l vector is a suffix of the level-1
//    vector.  Gathers/scatters (VecISCopy) and the identity blocks of R=[Z I], P=[W;I]
//    disappear from the solve phase.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include <string>
#include <stdexcept>

namespace airhip {

// One row block: a self-contained 32-byte descriptor, read with ONE scalar load (s_load_dwordx8).  p0: the block's first
// value (values are 8 bytes per non-zero for every block); ip: byte offset of the block's index region in DevMat::idx
// (a multiple of 4); nb / nr: its non-zeros and rows.  Three kinds of index region, told apart by cbase:
//   cbase >= 0   "packed": entries SORTED BY COLUMN, one 32-bit word each, ((column - cbase) * 8) << 10 | (row-major
//                position + 1) - blocks of many short rows whose columns span less than 2^19;
//   cbase == -1  "raw": row-major order, one 32-bit word = the absolute column (a single row longer than 512, or a block
//                that fits neither of the others);
//   cbase <= -2  "multi-base": row-major order, ONE 16-BIT WORD per entry, window << ob | offset, in front of them a table
//                of up to 64 windows (byte offset of the window's first column, 32 bits each); meta = -(cbase + 2) holds
//                ob (bits 0-4) and the number of windows (bits 5-).  10 bytes per non-zero + the table: blocks of few long
//                rows (the deep levels: a row's sorted columns are its row-major order anyway) and blocks whose columns
//                run through several sections of a hierarchical vector (A_fc), which were raw before round 4.
//                meta bits 12- (round 5): nd, the size of the block's VALUE DICTIONARY, 0 = plain values.  A multi-base block
//                of at most 512 entries whose values take nd <= 256 distinct bit patterns (the operators of the first levels
//                of a constant-coefficient problem: 1-5 distinct values on levels 1-2, ~150 per block on level 3) keeps, as
//                its region of the value array, the nd distinct values (ascending bit patterns) followed by 512 bytes of
//                indices - byte 8 l + u = the index of entry 64 u + l, so that a lane fetches its eight indices with one
//                8-byte load: 8 nd + 512 bytes of values per block instead of 8 nb.  Lossless: the kernels multiply the very
//                same doubles in the very same order.
struct alignas(32) BlkDesc { int64_t p0; int32_t row; int32_t cbase; int32_t nb; int32_t nr; int64_t ip; };
constexpr int MB_OB_MASK = 31, MB_WIN_SHIFT = 5, MB_WIN_MASK = 127, MB_MAX_WIN = 64, MB_ND_SHIFT = 12, MB_MAX_ND = 256;
__host__ __device__ inline int mb_meta(const BlkDesc& q) { return -(q.cbase + 2); }
__host__ __device__ inline int mb_nwin(int meta) { return (meta >> MB_WIN_SHIFT) & MB_WIN_MASK; }
__host__ __device__ inline int mb_nd(int meta) { return meta >> MB_ND_SHIFT; }
// the blocks the lean kernels process (spmv.hip): multi-base and raw blocks - everything but the packed form of rounds 2-3
__host__ __device__ inline bool lean_block(const BlkDesc& q) { return q.cbase <= -1; }

// Row-block CSR.  One wavefront streams a block's (index word, value) pairs with coalesced loads, gathers x,
// stages the products a_ij * x_j in LDS at their row-major positions and then sums every row's LDS segment
// (one lane per row, or T lanes per row when the block holds few long rows; a block of one row is summed in
// registers).  12 or 10 bytes per non-zero (BlkDesc); 2 bytes per row of block-relative row offsets.
struct DevMat {
  int64_t m = 0, n = 0;          // rows, cols (n includes the ghost columns of a distributed operand)
  int64_t n_ghost = 0;           // distributed operand: columns [n - n_ghost, n) are ghost values
  struct HaloPlan* halo = nullptr;   // how to obtain them (not owned)
  int64_t nnz = 0;
  int64_t nblocks = 0;
  BlkDesc* blk = nullptr;        // [nblocks] in storage order
  uint16_t* roff = nullptr;      // [m] start of row r relative to its block's first non-zero
  uint8_t* idx = nullptr;        // [idx_bytes + 4] the blocks' index regions back to back (BlkDesc::ip), one readable padding word
  int64_t idx_bytes = 0;
  double* val = nullptr;         // [val_len + 1] the blocks' value regions back to back (BlkDesc::p0): 8 bytes per non-zero, or - a block with
                                 // a value dictionary - nd values + 512 bytes of indices
  int64_t val_len = 0;           // doubles in val (= nnz when no block has a dictionary)
  // Launch lists: copies of the descriptors in processing order, the multi-base blocks FIRST (they run in the lean kernels,
  // the other kinds in the *_g kernels - spmv.hip): [0, n_mb*) multi-base, [n_mb*, n*) the rest, each part in order.
  //  desc_ord  all nblocks blocks.  Storage order, or - operators whose rows run through the sections of a hierarchical
  //            vector - the spatial order of their rows, so that the blocks of different sections that gather the same
  //            neighbourhood of x run close together in time (DevPCAIR::upload);
  //  desc_int / desc_bnd  distributed operand: the row blocks that touch no ghost column (interior) / at least one
  //            (boundary); the interior blocks run while the neighbour exchange is in flight
  BlkDesc* desc_ord = nullptr; int64_t n_mb = 0;
  BlkDesc* desc_int = nullptr; BlkDesc* desc_bnd = nullptr;
  int64_t n_int = 0, n_bnd = 0, n_mb_int = 0, n_mb_bnd = 0;
  int64_t nnz_mb = 0;            // non-zeros in multi-base blocks
  int64_t val_bytes = 0;         // bytes of the value array a product reads: 8 per non-zero, or 8 nd + 512 for a block with a value dictionary
  // transient operator (the setup's truncation test): its four arrays come from the block cache of device/devmem.hpp and go
  // back to it - a hipMalloc that follows the release of multi-GB blocks stalls for seconds every so often
  bool cached = false;
  bool valid() const { return blk != nullptr; }
  // algorithmic bytes of one SpMV with this operator: S(X) of SURVEY 8(d)
  double s_bytes() const { return 12.0 * nnz + 4.0 * (m + 1) + 8.0 * m + 8.0 * n; }
  // bytes the format actually holds
  double stored_bytes() const { return 8.0 * (val_len + 1) + (double)idx_bytes + 4.0 + 2.0 * m + 32.0 * nblocks * (desc_ord ? 2 : 1); }
  // share of s_bytes() that the lean kernels' launches carry
  double mb_share() const { return nnz > 0 ? (double)nnz_mb / (double)nnz : 1.0; }
};

// out_i = alpha * s_i + beta * z_i + gamma * w_i,  s = (A x)_i [ * dscale_i ]
// optional auxiliary update in the same pass:  y2_i += delta * (aux_out ? out_i : w2_i)
struct SpmvEpi {
  double alpha = 1.0;
  const double* z = nullptr; double beta = 0.0;
  const double* w = nullptr; double gamma = 0.0;
  const double* dscale = nullptr;     // multiply s by dscale_i (holds 1/d)
  double* out = nullptr;
  double* y2 = nullptr; const double* w2 = nullptr; double delta = 0.0; int aux_out = 0;
  // products of the setup phase (the device auto-truncation test) launch the same code under the kernel name
  // k_rb_spmv_setup, so that a kernel trace of a run separates them from the solve phase's k_rb_spmv
  int setup_phase = 0;
};

// which launch list (DevMat) and which of its two parts a launch covers
enum { RB_ALL = 0, RB_INTERIOR = 1, RB_BOUNDARY = 2 };
enum { RB_BOTH = 0, RB_MB = 1, RB_GENERAL = 2 };
// x_ghost: the received ghost values of a distributed operand (A.n_ghost entries), else null
void launch_spmv(const DevMat& A, const double* x, const SpmvEpi& e, hipStream_t s, const double* x_ghost = nullptr,
                 int list = RB_ALL, int part = RB_BOTH);

// Fused first stage of an F-point relaxation with a one-point prolongator:
//   acc = (A_fc x_c)_i ; t4 = b_f_i - acc ; x0 = x_c[wcol_i] (0 if wcol_i < 0)
//   diag mode (aff_d, minv != null):  x_f_i = x0 + minv_i * (t4 - aff_d_i * x0)
//   split mode                     :  t4_out_i = t4 ; x_f_i = x0
void launch_fc_first(const DevMat& A_fc, const int32_t* wcol, const double* b_f, const double* x_c,
                     const double* aff_d, const double* minv, double* t4_out, double* x_f, hipStream_t s,
                     const double* x_ghost = nullptr, int list = RB_ALL, int part = RB_BOTH);
// host flags (one per row block): does the block reference a ghost column?
void devmat_boundary_flags(const DevMat& A, std::vector<uint8_t>& flags, hipStream_t s);
// a launch list holding the blocks order[0 .. n) of A, multi-base blocks first (device memory, released with
// dev_free_exact); *n_mb = how many of them are multi-base
BlkDesc* devmat_desc_list(const DevMat& A, const int32_t* order_host, int64_t n, hipStream_t s, int64_t* n_mb);
// host side of the same: stable partition of descriptors, multi-base blocks first; returns their number
int64_t partition_descs_host(std::vector<BlkDesc>& v);

// vector kernels (n elements)
void launch_axpby(int64_t n, double a, const double* x, double b, const double* y, double* out, hipStream_t s);  // out = a x + b y
// out = alpha * (a .* b) + beta * z   (z may be null)
void launch_pointwise_mult(int64_t n, double alpha, const double* a, const double* b, double beta, const double* z, double* out, hipStream_t s);
void launch_diag_residual_update(int64_t n, const double* d, const double* minv, const double* t4, double* x, hipStream_t s); // x += minv.*(t4 - d.*x)
void launch_gather(int64_t n, const int32_t* idx, const double* in, double* out, hipStream_t s);   // out[i] = in[idx[i]]
void launch_scatter(int64_t n, const int32_t* idx, const double* in, double* out, hipStream_t s);  // out[idx[i]] = in[i]
void launch_fill(int64_t n, double v, double* out, hipStream_t s);
// natural <-> hierarchical permutation chunk by chunk of the locality key (kernels.hip): key_h[p] = key[iperm[p]]; the chunk
// table starts[(nchunks + 1) * nsec]; out[p] = in[iperm[p]] (to_natural = false) or out[iperm[p]] = in[p]
void launch_keys_at(int64_t n, const int32_t* iperm, const int64_t* key, int64_t* out, hipStream_t s);
void launch_chunk_starts(int64_t nchunks, int nsec, int64_t chunk, const int64_t* sec_off, const int64_t* key_h, int32_t* starts, hipStream_t s);
void launch_permute_chunks(int64_t nchunks, int nsec, const int32_t* starts, const int32_t* iperm, const double* in, double* out, bool to_natural, hipStream_t s);
// dots[k] = <w, V_k>, k < nv (V_k = V + k*ld); two-stage deterministic reduction using `scratch`
void launch_mdot(int64_t n, const double* w, const double* V, int64_t ld, int nv, double* scratch, double* dots, hipStream_t s);
// w += sum_k coef[k] V_k  (coef on device); zero_first: w = sum_k ... (w is not read); norm_scratch / norm_out (device):
// *norm_out = sum of squares of the result, computed in the same pass (scratch: mdot_scratch_doubles(1) doubles)
void launch_maxpy(int64_t n, double* w, const double* V, int64_t ld, int nv, const double* coef, hipStream_t s,
                  bool zero_first = false, double* norm_scratch = nullptr, double* norm_out = nullptr);
int mdot_scratch_doubles(int nv);

// CSR (device) -> row-block CSR conversion.  rowsrc (may be null): destination row r takes source
// row rowsrc[r]; colmap (may be null): column c becomes colmap[c].
void devmat_from_csr_device(int64_t m, int64_t n, const int64_t* d_rp, const int32_t* d_ci, const double* d_v,
                          const int32_t* d_rowsrc, const int32_t* d_colmap, DevMat& out, hipStream_t s,
                          int64_t n_ghost = 0,    // the last n_ghost columns are not remapped
                          bool cached = false);   // see DevMat::cached
void devmat_free(DevMat& A);

// performance knobs (not part of the stable ABI; airhip_set_tuning): see spmv.hip
int spmv_set_tuning(const char* key, int value);
int kernels_set_tuning(const char* key, int value);      // kernels.hip: experiments on the Krylov kernels

#define HIP_CHECK(expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      throw std::runtime_error(std::string("HIP error ") + hipGetErrorString(_e) + " at " +   \
                               __FILE__ + ":" + std::to_string(__LINE__));                    \
  } while (0)

}  // namespace airhip
