// Device implementation of the SetupAccel interface (host/spops.hpp): the sparse products of
// the PCAIR setup run as HIP SpGEMMs (device/spgemm.hip).  The coarse operator produced by
// one level's R*A*P stays resident and is reused as the A of the next level, so each level
// operator crosses PCIe once (device -> host, for the host-side CF splitting / extraction).
#include "setup_accel.hpp"
#include "dev.hpp"
#include "spgemm.hpp"

namespace airhip {

namespace {


struct DevSetupAccel final : SetupAccel {
  hipStream_t stream = nullptr;
  DCsr resident;                 // last R*A*P result
  const void* resident_key = nullptr;  // host row-pointer array it was downloaded into
  int64_t min_nnz = 200000;      // smaller products stay on the host: launch + sync overheads dominate

  ~DevSetupAccel() override { dcsr_free(resident); }

  DCsr up(const Csr& A) {
    const double t0 = now_s();
    DCsr d = dcsr_upload(A.m, A.n, A.rp.data(), A.ci.data(), A.v.data());
    t_transfer += now_s() - t0;
    return d;
  }
  Csr down(const DCsr& d) {
    const double t0 = now_s();
    Csr C; C.m = d.m; C.n = d.n;
    C.rp.alloc(d.m + 1); C.ci.alloc(d.nnz); C.v.alloc(d.nnz);
    dcsr_download(d, C.rp.data(), C.ci.data(), C.v.data());
    t_transfer += now_s() - t0;
    return C;
  }

  Csr spgemm(const Csr& A, const Csr& B, const SpgemmDrop& drop) override {
    if (A.nnz() + B.nnz() < min_nnz) return airhip::spgemm(A, B, drop);
    DCsr dA = up(A), dB = up(B);
    const double t0 = now_s();
    DCsr dC = dev_spgemm(dA, dB, drop.enabled, drop.tol, drop.lump, drop.negate, stream);
    t_compute += now_s() - t0;
    dcsr_free(dA); dcsr_free(dB);
    Csr C = down(dC);
    dcsr_free(dC);
    return C;
  }

  Csr rap(const Csr& R, const Csr& A, const Csr& P, const SpgemmDrop& drop) override {
    if (A.nnz() < min_nnz) {
      dcsr_free(resident); resident_key = nullptr;
      Csr AP = airhip::spgemm(A, P);
      return airhip::spgemm(R, AP, drop);
    }
    DCsr dA;
    if (resident_key && resident_key == (const void*)A.rp.data() && resident.m == A.m && resident.nnz == A.nnz()) {
      dA = resident; resident = DCsr();
    } else {
      dcsr_free(resident);
      dA = up(A);
    }
    resident_key = nullptr;
    DCsr dP = up(P);
    double t0 = now_s();
    DCsr dAP = dev_spgemm(dA, dP, false, 0.0, false, false, stream);
    t_compute += now_s() - t0;
    dcsr_free(dA); dcsr_free(dP);
    DCsr dR = up(R);
    t0 = now_s();
    DCsr dC = dev_spgemm(dR, dAP, drop.enabled, drop.tol, drop.lump, drop.negate, stream);
    t_compute += now_s() - t0;
    dcsr_free(dR); dcsr_free(dAP);
    Csr C = down(dC);
    resident = dC;
    resident_key = (const void*)C.rp.data();   // Arr moves keep the pointer
    return C;
  }
};

}  // namespace

std::unique_ptr<SetupAccel> make_device_setup_accel(hipStream_t stream, int64_t min_nnz) {
  auto a = std::make_unique<DevSetupAccel>();
  a->stream = stream;
  if (min_nnz >= 0) a->min_nnz = min_nnz;
  return a;
}

}  // namespace airhip
