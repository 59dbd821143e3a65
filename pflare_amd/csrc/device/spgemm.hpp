// Device CSR container + SpGEMM with fused drop/lump used to accelerate the PCAIR setup
// (the Z = -A_cf*M and R*A*P products, src/AIR_Operators_Setup.F90:784-795, 1008-1069).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace airhip {

struct DCsr {
  int64_t m = 0, n = 0, nnz = 0;
  int64_t* rp = nullptr;
  int32_t* ci = nullptr;
  double* v = nullptr;
};

// rows each accumulator kind has worked on: registers (<= 64 candidates), LDS tables of 1k / 2k / 4k slots, global-memory
// table (a row that overflows a table is counted again by the next kind)
struct SpgemmStats { int64_t rows_reg = 0, rows_1k = 0, rows_2k = 0, rows_4k = 0, rows_global = 0; };

// test hook: cap of the output pool in entries (0 = automatic); a small cap forces the retry rounds
void spgemm_set_pool_limit(int64_t entries);
// test hook: entries a wave takes from the pool at a time (-1 = automatic: 2048 on large products, row by row on small ones)
void spgemm_set_pool_chunk(int entries);

void dcsr_free(DCsr& A);
DCsr dcsr_upload(int64_t m, int64_t n, const int64_t* rp, const int32_t* ci, const double* v);
void dcsr_download(const DCsr& d, int64_t* rp, int32_t* ci, double* v);

// C = filter(A * B).  With drop_enabled: entries with |c_ij| < tol * max_j |c_ij| are removed
// (the diagonal i==j is never removed; with lump the removed entries are added to it), negate
// flips the sign of the product first.  Rows come out sorted by column; values are bit-identical
// to the host spgemm (host/spops.cpp).
DCsr dev_spgemm(const DCsr& A, const DCsr& B, bool drop_enabled, double tol, bool lump, bool negate, hipStream_t s,
                SpgemmStats* stats = nullptr);

}  // namespace airhip
