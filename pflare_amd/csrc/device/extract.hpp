// Sub-matrix extraction on the device: the F / C blocks of a level operator that is resident there
// (the reference's MatCreateSubMatrix calls of src/AIR_MG_Setup.F90:405-451 and their device twins
// src/PETSc_Helperk.kokkos.cxx:1801-2477).
#pragma once
#include "spgemm.hpp"

namespace airhip {

// code[i] = position of row / column i inside its own set, with bit 31 set for an F point (device array of A.m words).
// A_cc is only built when `want_cc`.  Rows keep their order and their entries stay sorted (the positions are monotone
// inside each set), so the blocks equal host/spops.cpp: split_fc entry for entry.
void dev_split_fc(const DCsr& A, const uint32_t* d_code, int64_t n_f, int64_t n_c, DCsr& A_ff, DCsr& A_fc, DCsr& A_cf, DCsr* A_cc,
                  hipStream_t s);

// R = [Z I] in level ordering where Z already is (host twin: air_setup.cpp build_R; reference src/Grid_Transfer.F90:588-815):
// row j of R holds Z's row j with the columns F[.] and a one at column C[j], columns ascending.  d_F / d_C: the level's F and
// C index lists on the device.
DCsr dev_build_R(const DCsr& Z, const int32_t* d_F, const int32_t* d_C, int64_t n, hipStream_t s);
// P = [W; I] for the one-point classical prolongator (host twin: build_P; src/Grid_Transfer.F90:329-461): row r of P is a
// single one at column j for the level's j-th C point, at column wcol[i] for its i-th F point (no entry when wcol[i] < 0).
// d_code as in dev_split_fc.
DCsr dev_build_P_one_point(int64_t n, int64_t n_c, const uint32_t* d_code, const int32_t* d_wcol, hipStream_t s);
// the one-point classical prolongator's column per F point (host twin: air_setup.cpp one_point_W; src/Grid_Transfer.F90:
// 329-461): the column of the largest |entry| of the row of A_fc, the first one on ties, -1 for an empty row
void dev_one_point_w(const DCsr& A_fc, int32_t* d_wcol, hipStream_t s);
// G = A_fc + A_ff W for the one-point prolongator W: the operator of the fused first F sweep (host twin, with the reasons:
// host/spops.hpp fused_first_operator; same bits).  [A_ff | A_fc] and [W; I] are assembled on the device and multiplied by
// dev_spgemm.  d_wcol: W's column per F point on the device (-1 = none).
DCsr dev_fused_first_operator(const DCsr& A_ff, const DCsr& A_fc, const int32_t* d_wcol, hipStream_t s);

}  // namespace airhip
