// Fixed-sparsity polynomial assembly on the device (device/polyasm.hip)
#pragma once
#include "spgemm.hpp"

namespace airhip {
// M = c[0] I + c[1] A + sum_{t=2}^{nterms-1} c[t] [A^t]_pattern(A)   (sparsity order 1), bit-identical to
// host/poly.cpp::mat_mult_powers_share_sparsity.  Returns false (M untouched) for what it does not handle
// (fewer than 3 coefficients, more than 16, a row longer than 1024).
bool dev_poly_assemble(const DCsr& A, const double* c, int nterms, DCsr& M, hipStream_t s);
}
