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

}  // namespace airhip
