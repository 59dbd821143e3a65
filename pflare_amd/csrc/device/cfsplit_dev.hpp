// CF splitting on the device (device/cfsplit.hip)
#pragma once
#include "spgemm.hpp"
#include <cstdint>

namespace airhip {
// CF_PMISR_DDC splitting of the resident operator A; markers_host[i] = F_POINT (-1) / C_POINT (+1), identical to
// host/cfsplit.cpp::compute_cf_splitting.  seed_key = rng_key(seed).
void dev_cf_splitting_pmisr_ddc(const DCsr& A, double strong_threshold, bool symmetrize, int max_luby_steps, uint64_t seed_key,
                                int ddc_its, double ddc_fraction, int32_t* markers_host, hipStream_t s);
}
