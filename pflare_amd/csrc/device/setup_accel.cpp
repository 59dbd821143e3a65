// Device implementation of the SetupAccel interface (host/spops.hpp): the sparse products of
// the PCAIR setup run as HIP SpGEMMs (device/spgemm.hip).  The coarse operator produced by
// one level's R*A*P stays resident: the next level's truncation test, CF splitting, F / C block
// extraction and R*A*P read it where it is, and it only crosses PCIe if a host routine asks for
// it (materialize; the coarsest operator always does).  What goes to the host per level are the
// blocks A_ff, A_fc, A_cf (A_cc only with C smoothing), the assembled inverse, Z and the counts.
#include "setup_accel.hpp"
#include "dev.hpp"
#include "spgemm.hpp"
#include "cfsplit_dev.hpp"
#include "polyasm.hpp"
#include "extract.hpp"
#include "devmem.hpp"
#include "staged_copy.hpp"
#include "../host/air.hpp"
#include <cmath>
#include <cstdlib>
#include <vector>
#include <unordered_map>
#include <unordered_set>
#include <mutex>

namespace airhip {

namespace {


struct DevSetupAccel final : SetupAccel {
  hipStream_t stream = nullptr;
  DCsr resident;                 // last R*A*P result (or the top-level operator)
  const void* resident_key = nullptr;  // host row-pointer array it was downloaded into / uploaded from
  uint64_t resident_tag = 0;     // Csr::device_tag of the operator it is
  DCsr res_ff, res_cf;           // blocks of the last extraction: the A of the polynomial assembly, the A of Z = -A_cf M
  uint64_t ff_tag = 0, cf_tag = 0;
  uint64_t next_tag = 0;
  DCsr res_z;                    // the last Z = -A_cf * M (spgemm_z): R = [Z I] is assembled from it for the R*A*P that follows
  uint64_t z_tag = 0;
  DCsr inverse;                  // last assembled polynomial inverse (the B of the Z = -A_cf * M product that follows)
  const void* inverse_key = nullptr;
  uint64_t inverse_tag = 0;      // Csr::device_tag of that inverse
  int64_t min_nnz = 200000;      // smaller products stay on the host: launch + sync overheads dominate
  DCsr res_fc;                   // the F-C block of the last extraction, until the one-point prolongator has been read off it
  uint64_t fc_tag = 0;

  // ---- copies kept for the upload of the hierarchy (SetupAccel::keep_copies): an operator the setup is done with is
  // retired, not freed, while the copies stay below 40 % of the device's memory.  With defer_downloads() on top, the
  // operators no host routine of the plain recipe reads (A_cf, the assembled inverse, Z) are not downloaded at all
  // during the setup: the caller gets stubs (Csr::stub_nnz >= 0) and fetches them with materialize() when it likes
  // (capi.cpp: a thread, after the upload) - until then the copy here is the only one (`host_missing`) and stays.
  bool keeping = false, defer_ok = false, plain = false;
  std::mutex mu_kept;            // kept / lent / host_missing: the mirror thread and the pressure handler
  std::unordered_map<uint64_t, DCsr> kept;
  std::unordered_map<uint64_t, DCsr> lent;      // copies handed out by device_copy: not dropped under memory pressure
  std::unordered_set<uint64_t> host_missing;
  size_t kept_bytes = 0, kept_limit = 0;
  int pressure_id = 0;
  static size_t bytes_of(const DCsr& d) { return sizeof(int64_t) * (size_t)(d.m + 1) + 12 * (size_t)d.nnz; }
  void keep_copies(bool on, double top_nnz) override {
    keeping = on;
    if (on && !pressure_id) pressure_id = dev_add_pressure_handler([this] { return drop_all_kept(); });
    if (on && !kept_limit) {
      // 36 % of the device (with the hierarchy's own arrays beside them during the upload that is nearly all of it), and never into what the setup itself will need: its temporaries peak at ~350 bytes per
      // non-zero of the top operator (464^3: 137 GB), 450 with a margin.  (The copies whose operators are not on the
      // host cannot be dropped when memory runs out, so the budget has to be safe beforehand.)
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double room = (double)dev_available_bytes() - 450.0 * std::max(top_nnz, 0.0);     // (the arena's helper is backing memory already)
        kept_limit = (size_t)std::max(0.0, std::min(0.36 * (double)total_b, room));
      } else (void)hipGetLastError();
    }
  }
  void defer_downloads(bool on) override { defer_ok = on; }
  void plain_recipe(bool on) override { plain = on; }
  // may an operator of this size stay on the device alone?
  bool can_defer(const DCsr& d) {
    const char* e = std::getenv("AIRHIP_DEFER_DOWNLOADS");      // (read every time: the tests switch it)
    const bool enabled = !(e && e[0] == '0');
    std::lock_guard<std::mutex> lk(mu_kept);
    return enabled && keeping && defer_ok && plain && kept_bytes + bytes_of(d) <= kept_limit;
  }
  static Csr stub_of(const DCsr& d, uint64_t tag) { Csr C; C.m = d.m; C.n = d.n; C.stub_nnz = d.nnz; C.device_tag = tag; return C; }
  size_t drop_all_kept() {          // the device is full: the copies are only a short cut - except the only ones
    std::unique_lock<std::mutex> lk(mu_kept, std::try_to_lock);
    if (!lk.owns_lock()) return 0;
    size_t b = 0;
    for (auto it = kept.begin(); it != kept.end();) {
      if (host_missing.count(it->first)) { ++it; continue; }
      b += bytes_of(it->second); dcsr_free(it->second); it = kept.erase(it);
    }
    kept_bytes -= b; keeping = false;
    return b;
  }
  void retire(DCsr& d, uint64_t tag) {
    if (!d.rp) return;
    std::lock_guard<std::mutex> lk(mu_kept);
    const bool only_copy = tag && host_missing.count(tag);
    if (tag && !kept.count(tag) && (only_copy || (keeping && kept_bytes + bytes_of(d) <= kept_limit))) { kept[tag] = d; kept_bytes += bytes_of(d); d = DCsr(); }
    else if (only_copy) throw std::logic_error("setup accelerator: two device copies of an operator that is not on the host");
    else dcsr_free(d);
  }
  // what still sits in the slots of the level loop (the last inverse - the coarse solver's has no product after it -, the
  // last Z, a C-F block no product came for) joins the kept copies
  void settle() {
    retire(inverse, inverse_tag); inverse_key = nullptr; inverse_tag = 0;
    retire(res_z, z_tag); z_tag = 0;
    retire(res_fc, fc_tag); fc_tag = 0;
    bool only; { std::lock_guard<std::mutex> lk(mu_kept); only = cf_tag && host_missing.count(cf_tag); }
    if (only) { retire(res_cf, cf_tag); cf_tag = 0; }
  }
  bool device_copy(uint64_t tag, int64_t m, int64_t nnz, const int64_t** rp, const int32_t** ci, const double** v) override {
    settle();
    std::lock_guard<std::mutex> lk(mu_kept);
    auto it = kept.find(tag);
    if (it == kept.end() || it->second.m != m || it->second.nnz != nnz) return false;
    *rp = it->second.rp; *ci = it->second.ci; *v = it->second.v;
    kept_bytes -= bytes_of(it->second);
    lent[tag] = it->second;
    kept.erase(it);
    return true;
  }
  void drop_device_copy(uint64_t tag) override {
    std::lock_guard<std::mutex> lk(mu_kept);
    for (auto* m : {&lent, &kept}) {
      auto it = m->find(tag);
      if (it == m->end()) continue;
      if (host_missing.count(tag)) {           // still the only copy: back to (or stays in) `kept`
        if (m == &lent) { kept[tag] = it->second; kept_bytes += bytes_of(it->second); lent.erase(it); }
        return;
      }
      if (m == &kept) kept_bytes -= bytes_of(it->second);
      dcsr_free(it->second);
      m->erase(it);
      return;
    }
  }

  // (the block cache is not trimmed here: the upload of the hierarchy that usually follows stages through it)
  ~DevSetupAccel() override {
    if (pressure_id) dev_remove_pressure_handler(pressure_id);
    dcsr_free(resident); dcsr_free(inverse); dcsr_free(res_ff); dcsr_free(res_cf); dcsr_free(res_z); dcsr_free(res_fc);
    for (auto& k : kept) dcsr_free(k.second);
    for (auto& k : lent) dcsr_free(k.second);
  }

  bool is_resident(const Csr& A) const {
    if (!resident.rp || resident.m != A.m || resident.nnz != A.nnz()) return false;
    if (A.device_tag && A.device_tag == resident_tag) return true;
    return A.on_host() && resident_key && resident_key == (const void*)A.rp.data();
  }
  void drop_resident() { dcsr_free(resident); resident_key = nullptr; resident_tag = 0; }

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

  Csr spgemm(const Csr& A, const Csr& B, const SpgemmDrop& drop) override { return product(A, B, drop, false); }
  Csr spgemm_z(const Csr& A, const Csr& B, const SpgemmDrop& drop) override { return product(A, B, drop, true); }

  Csr product(const Csr& A, const Csr& B, const SpgemmDrop& drop, bool keep) {
    retire(res_z, z_tag); z_tag = 0;
    if (A.on_host() && B.on_host() && A.nnz() + B.nnz() < min_nnz) return airhip::spgemm(A, B, drop);
    DCsr dA, dB;
    uint64_t a_tag = 0, b_tag = 0;
    if (A.device_tag && A.device_tag == cf_tag && res_cf.rp && res_cf.nnz == A.nnz()) { dA = res_cf; res_cf = DCsr(); a_tag = cf_tag; cf_tag = 0; }
    else {
      if (!A.on_host()) throw std::logic_error("setup accelerator: the device copy of a product's left operand is gone");
      dA = up(A);
    }
    const bool b_here = inverse.rp && inverse.m == B.m && inverse.nnz == B.nnz() &&
                        ((B.device_tag && B.device_tag == inverse_tag) || (B.on_host() && inverse_key && inverse_key == (const void*)B.rp.data()));
    if (b_here) { dB = inverse; inverse = DCsr(); b_tag = inverse_tag; }
    else {
      if (!B.on_host()) throw std::logic_error("setup accelerator: the device copy of a product's right operand is gone");
      dB = up(B);
    }
    retire(inverse, inverse_tag); inverse_key = nullptr; inverse_tag = 0;
    const double t0 = now_s();
    dev_stage("setup: product (Z = -A_cf * M or A * B)");
    DCsr dC = dev_spgemm(dA, dB, drop.enabled, drop.tol, drop.lump, drop.negate, stream, nullptr);
    t_compute += now_s() - t0;
    // the C-F block is only kept when the host has not got it yet; the inverse is an operator of the hierarchy
    { bool only; { std::lock_guard<std::mutex> lk(mu_kept); only = a_tag && host_missing.count(a_tag); } if (only) retire(dA, a_tag); else dcsr_free(dA); }
    retire(dB, b_tag);
    Csr C;
    if (keep && can_defer(dC)) {
      C = stub_of(dC, ++next_tag);
      std::lock_guard<std::mutex> lk(mu_kept); host_missing.insert(C.device_tag);
    } else {
      dev_stage("setup: product, download of the result");
      C = down(dC);
      if (keep) C.device_tag = ++next_tag;
    }
    if (keep) { res_z = dC; z_tag = C.device_tag; }
    else dcsr_free(dC);
    return C;
  }

  // CF splitting of the level operator where it already is (the previous level's R*A*P; the top level is uploaded
  // once and stays resident for its own R*A*P)
  bool cf_splitting(const Csr& A, const AirOptions& o, Arr<int32_t>& markers) override {
    if (o.cf_splitting_type != 0 || A.nnz() < min_nnz) return false;
    if (!is_resident(A)) { drop_resident(); resident = up(A); resident_key = (const void*)A.rp.data(); }
    const double t0 = now_s();
    markers.alloc(A.m);
    dev_stage("setup: CF splitting (PMISR + DDC)");
    dev_cf_splitting_pmisr_ddc(resident, o.strong_threshold, !o.symmetric, o.max_luby_steps, rng_key(o.seed), o.ddc_its,
                               o.ddc_fraction, markers.data(), stream);
    t_cfsplit += now_s() - t0;
    return true;
  }

  bool one_point_w(const Csr& A_fc, Arr<int32_t>& wcol) override {
    if (!(A_fc.device_tag && A_fc.device_tag == fc_tag && res_fc.rp && res_fc.m == A_fc.m && res_fc.nnz == A_fc.nnz())) return false;
    const double t0 = now_s();
    dev_stage("setup: one-point prolongator");
    wcol.alloc(A_fc.m);
    if (A_fc.m > 0) {
      int32_t* d_w = (int32_t*)dev_alloc(sizeof(int32_t) * (size_t)A_fc.m);
      dev_one_point_w(res_fc, d_w, stream);
      staged_d2h(wcol.data(), d_w, sizeof(int32_t) * (size_t)A_fc.m);
      dev_free(d_w);
    }
    retire(res_fc, fc_tag); fc_tag = 0;
    t_extract += now_s() - t0;
    return true;
  }

  bool poly_assemble(const Csr& A, const std::vector<double>& c, Csr& M, bool final_values) override {
    if (A.nnz() < min_nnz || c.size() < 3) return false;
    DCsr dA, dM;
    if (A.device_tag && A.device_tag == ff_tag && res_ff.rp && res_ff.nnz == A.nnz()) { dA = res_ff; res_ff = DCsr(); ff_tag = 0; }
    else dA = up(A);                       // (a fresh copy of the same host arrays: as good for the upload later)
    const double t0 = now_s();
    dev_stage("setup: polynomial assembly");
    bool ok;
    ok = dev_poly_assemble(dA, c.data(), (int)c.size(), dM, stream);
    t_poly += now_s() - t0;
    retire(dA, A.device_tag);              // A_ff itself is an operator of the hierarchy
    if (!ok) return false;
    retire(inverse, inverse_tag); inverse_key = nullptr; inverse_tag = 0;
    if (final_values && can_defer(dM)) {
      M = stub_of(dM, ++next_tag);
      { std::lock_guard<std::mutex> lk(mu_kept); host_missing.insert(M.device_tag); }
      inverse = dM; inverse_tag = M.device_tag;
      return true;
    }
    M = down(dM);
    if (final_values) { inverse = dM; inverse_key = (const void*)M.rp.data(); inverse_tag = M.device_tag = ++next_tag; }   // Arr moves keep the pointer
    else dcsr_free(dM);
    return true;
  }

  // Auto-truncation test on the device: Arnoldi (classical Gram-Schmidt with the multi-dot / multi-axpy kernels of
  // the Krylov solvers) -> harmonic-Ritz / Leja roots on the host (tiny) -> Newton-basis apply -> residual.
  bool truncation_ratio(const Csr& A, int inverse_type, int poly_order, uint64_t seed, double& ratio) override {
    if (A.nnz() < min_nnz) return false;                     // small levels: the host is as fast
    static const bool enabled = [] { const char* e = std::getenv("AIRHIP_DEVICE_TRUNC"); return !(e && e[0] == '0'); }();
    if (!enabled) return false;
    if (inverse_type != INV_NEWTON && inverse_type != INV_NEWTON_NO_EXTRA) return false;
    const int64_t n = A.m;
    const int p = poly_order;
    if (p < 1 || (int64_t)p + 1 > n) return false;
    struct Bufs {
      std::vector<void*> ptrs; DevMat M;
      ~Bufs() { for (void* q : ptrs) dev_free(q); devmat_free(M); }
      double* alloc(size_t count) { void* q = dev_alloc(sizeof(double) * std::max<size_t>(count, 1)); ptrs.push_back(q); return (double*)q; }
    } B;
    // the level operator: still resident from the previous level's R*A*P, or uploaded once
    {
      const bool have = is_resident(A);
      if (!have && !A.on_host()) return false;
      DCsr dA = have ? resident : up(A);
      dev_stage("setup: truncation test, row-block conversion");
      devmat_from_csr_device(dA.m, dA.n, dA.rp, dA.ci, dA.v, nullptr, nullptr, B.M, stream, 0, true);   // arrays from / to the block cache
      HIP_CHECK(hipStreamSynchronize(stream));
      if (!have) dcsr_free(dA);
    }
    const double t0 = now_s();
    dev_stage("setup: truncation test, Arnoldi");
    double* V = B.alloc((size_t)(p + 2) * (size_t)n);
    double* w = B.alloc((size_t)n);
    double* scratch = B.alloc((size_t)mdot_scratch_doubles(p + 2));
    double* d_small = B.alloc((size_t)(p + 4));
    std::vector<double> hv((size_t)n), h((size_t)p + 2), H((size_t)(p + 2) * (size_t)(p + 1), 0.0);
    auto norm_of = [&](const double* x) {
      launch_mdot(n, x, x, n, 1, scratch, d_small, stream);
      double v = 0.0;
      HIP_CHECK(hipMemcpyAsync(&v, d_small, sizeof(double), hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      return std::sqrt(v);
    };
    krylov_start_vector(seed, n, hv.data());
    HIP_CHECK(hipMemcpyAsync(V, hv.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream));
    const double beta = norm_of(V);
    if (!(beta > 0.0)) return false;
    launch_axpby(n, 1.0 / beta, V, 0.0, nullptr, V, stream);
    int m_used = p + 1;
    for (int m = 1; m <= p + 1; m++) {
      SpmvEpi e; e.setup_phase = 1; e.out = w;
      launch_spmv(B.M, V + (size_t)(m - 1) * (size_t)n, e, stream);
      launch_mdot(n, w, V, n, m, scratch, d_small, stream);
      HIP_CHECK(hipMemcpyAsync(h.data(), d_small, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
      for (int i = 0; i < m; i++) { H[(size_t)(m - 1) * (p + 2) + i] = h[i]; h[i] = -h[i]; }
      HIP_CHECK(hipMemcpyAsync(d_small, h.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice, stream));
      launch_maxpy(n, w, V, n, m, d_small, stream);
      const double hn = norm_of(w);
      H[(size_t)(m - 1) * (p + 2) + m] = hn;
      if (hn < TOL_LUCKY) { m_used = m; break; }
      launch_axpby(n, 1.0 / hn, w, 0.0, nullptr, V + (size_t)m * (size_t)n, stream);
    }
    std::vector<double> re, im;
    newton_roots_from_hessenberg(H.data(), beta, m_used, p, inverse_type == INV_NEWTON, re, im);
    const int nr = (int)re.size();
    if (nr == 0) return false;
    // y = q(A) x with the roots (petsc_newton, src/Gmres_Poly_Newton.F90:763-875); the Krylov basis is no longer needed
    double* x = V;                 // test vector
    double* y = V + (size_t)n;
    double* ta = V + 2 * (size_t)n;
    double* tb = V + 3 * (size_t)n;    // p >= 1 gives p + 2 >= 3 vectors; w is the fourth
    if (p + 2 < 4) tb = B.alloc((size_t)n);
    double* t2 = w;
    {
      const uint64_t key = rng_key(seed + 7919);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) hv[i] = uniform01(key, (uint64_t)i);
      HIP_CHECK(hipMemcpyAsync(x, hv.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, stream));
      HIP_CHECK(hipStreamSynchronize(stream));
    }
    dev_stage("setup: truncation test, Newton apply");
    launch_fill(n, 0.0, y, stream);
    const double* t = x;
    auto other = [&](const double* cur) { return cur == ta ? tb : ta; };
    int k = 0;
    while (k <= nr - 2) {
      if (im[k] == 0.0) {
        if (std::fabs(re[k]) < TOL_ZERO) { k++; continue; }
        const double a = 1.0 / re[k];
        launch_axpby(n, a, t, 1.0, y, y, stream);
        double* tn = other(t);
        SpmvEpi e; e.setup_phase = 1; e.out = tn; e.alpha = -a; e.z = t; e.beta = 1.0;
        launch_spmv(B.M, t, e, stream);
        t = tn; k++;
      } else {
        const double sq = re[k] * re[k] + im[k] * im[k];
        if (sq < TOL_ZERO) { k += 2; continue; }
        SpmvEpi e; e.setup_phase = 1; e.out = t2; e.alpha = -1.0; e.z = t; e.beta = 2.0 * re[k];
        launch_spmv(B.M, t, e, stream);
        launch_axpby(n, 1.0 / sq, t2, 1.0, y, y, stream);
        if (k <= nr - 3) {
          double* tn = other(t);
          SpmvEpi e2; e2.setup_phase = 1; e2.out = tn; e2.alpha = -1.0 / sq; e2.z = t; e2.beta = 1.0;
          launch_spmv(B.M, t2, e2, stream);
          t = tn;
        }
        k += 2;
      }
    }
    if (im[nr - 1] == 0.0 && std::fabs(re[nr - 1]) > TOL_ZERO) launch_axpby(n, 1.0 / re[nr - 1], t, 1.0, y, y, stream);
    // r = A y - x
    SpmvEpi e; e.setup_phase = 1; e.out = t2; e.alpha = 1.0; e.z = x; e.beta = -1.0;
    launch_spmv(B.M, y, e, stream);
    const double rn = norm_of(t2), xn = norm_of(x);
    t_compute += now_s() - t0;
    if (!(xn > 0.0) || !(rn == rn)) return false;
    ratio = rn / xn;
    return true;
  }

  Csr rap(const Csr& R, const Csr& A, const Csr& P, const SpgemmDrop& drop, bool device_only) override {
    if (A.nnz() < min_nnz) {
      if (!A.on_host()) throw std::logic_error("rap: a small operator that is not on the host");
      drop_resident();
      Csr AP = airhip::spgemm(A, P);
      return airhip::spgemm(R, AP, drop);
    }
    DCsr dA;
    if (is_resident(A)) { dA = resident; resident = DCsr(); }
    else {
      if (!A.on_host()) throw std::logic_error("rap: the device copy of the level operator is gone");
      dcsr_free(resident);
      dA = up(A);
    }
    resident_key = nullptr; resident_tag = 0;
    DCsr dP = up(P);
    double t0 = now_s();
    dev_stage("setup: rap, A * P");
    DCsr dAP = dev_spgemm(dA, dP, false, 0.0, false, false, stream);
    t_compute += now_s() - t0;
    dcsr_free(dA); dcsr_free(dP);
    DCsr dR = up(R);
    t0 = now_s();
    dev_stage("setup: rap, R * (A P)");
    DCsr dC = dev_spgemm(dR, dAP, drop.enabled, drop.tol, drop.lump, drop.negate, stream);
    dev_stage("setup: rap, after the products");
    t_compute += now_s() - t0;
    dcsr_free(dR); dcsr_free(dAP);
    Csr C;
    // a coarse operator below the device threshold goes to the host at once: the next level works there
    if (device_only && dC.nnz >= min_nnz) { C.m = dC.m; C.n = dC.n; C.stub_nnz = dC.nnz; }
    else C = down(dC);
    resident = dC;
    resident_tag = C.device_tag = ++next_tag;
    resident_key = C.on_host() ? (const void*)C.rp.data() : nullptr;   // Arr moves keep the pointer
    return C;
  }

  // R = [Z I] and the one-point P = [W; I] assembled on the device from the Z that spgemm_z left here, then the two
  // products of rap() - the host neither builds R and P nor uploads them (R alone is as large as Z)
  bool rap_from_z(const Csr& Z, const Arr<int32_t>& F, const Arr<int32_t>& C, const Arr<int32_t>& wcol, const Csr& A,
                  const SpgemmDrop& drop, Csr& coarse) override {
    static const bool enabled = [] { const char* e = std::getenv("AIRHIP_DEVICE_RP"); return !(e && e[0] == '0'); }();
    const bool have_z = enabled && res_z.rp && Z.device_tag && Z.device_tag == z_tag && res_z.m == Z.m && res_z.nnz == Z.nnz();
    if (!have_z || A.nnz() < min_nnz || !is_resident(A)) { retire(res_z, z_tag); z_tag = 0; return false; }
    const int64_t n = A.m, n_f = F.size(), n_c = C.size();
    // the index lists and the prolongator's columns: 8 bytes per row of the level
    double t0 = now_s();
    Arr<uint32_t> code(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n_f; i++) code[F[i]] = (uint32_t)i | 0x80000000u;
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < n_c; j++) code[C[j]] = (uint32_t)j;
    struct Tmp { void* p; explicit Tmp(size_t b) : p(dev_alloc(b ? b : 1)) {} ~Tmp() { dev_free(p); } };
    Tmp d_code(sizeof(uint32_t) * (size_t)n), d_F(sizeof(int32_t) * (size_t)n_f), d_C(sizeof(int32_t) * (size_t)n_c), d_w(sizeof(int32_t) * (size_t)n_f);
    staged_h2d(d_code.p, code.data(), sizeof(uint32_t) * (size_t)n);
    if (n_f) { staged_h2d(d_F.p, F.data(), sizeof(int32_t) * (size_t)n_f); staged_h2d(d_w.p, wcol.data(), sizeof(int32_t) * (size_t)n_f); }
    if (n_c) staged_h2d(d_C.p, C.data(), sizeof(int32_t) * (size_t)n_c);
    t_transfer += now_s() - t0;
    t0 = now_s();
    dev_stage("setup: rap_from_z, one-point P on the device");
    DCsr dP = dev_build_P_one_point(n, n_c, (const uint32_t*)d_code.p, (const int32_t*)d_w.p, stream);
    DCsr dA = resident; resident = DCsr(); resident_key = nullptr; resident_tag = 0;
    dev_stage("setup: rap_from_z, A * P");
    DCsr dAP = dev_spgemm(dA, dP, false, 0.0, false, false, stream);
    dcsr_free(dA); dcsr_free(dP);
    dev_stage("setup: rap_from_z, R = [Z I] on the device");
    DCsr dR = dev_build_R(res_z, (const int32_t*)d_F.p, (const int32_t*)d_C.p, n, stream);
    retire(res_z, z_tag); z_tag = 0;
    dev_stage("setup: rap_from_z, R * (A P)");
    DCsr dC = dev_spgemm(dR, dAP, drop.enabled, drop.tol, drop.lump, drop.negate, stream);
    dev_stage("setup: rap_from_z, after the products");
    t_compute += now_s() - t0;
    dcsr_free(dR); dcsr_free(dAP);
    coarse = Csr();
    if (dC.nnz >= min_nnz) { coarse.m = dC.m; coarse.n = dC.n; coarse.stub_nnz = dC.nnz; }
    else coarse = down(dC);
    resident = dC;
    resident_tag = coarse.device_tag = ++next_tag;
    resident_key = coarse.on_host() ? (const void*)coarse.rp.data() : nullptr;
    return true;
  }

  void materialize(Csr& A) override {
    if (A.on_host()) return;
    const uint64_t tag = A.device_tag;
    if (is_resident(A)) {
      A = down(resident);
      A.device_tag = tag;
      resident_key = (const void*)A.rp.data();
      return;
    }
    // an operator whose download was deferred: wherever its copy is at the moment (it stays there: the upload may
    // still want it; release_copies / drop_device_copy give it back)
    DCsr d;
    {
      std::lock_guard<std::mutex> lk(mu_kept);
      auto it = kept.find(tag);
      if (it != kept.end()) d = it->second;
      else if ((it = lent.find(tag)) != lent.end()) d = it->second;
    }
    if (!d.rp && tag && tag == z_tag) d = res_z;
    if (!d.rp && tag && tag == cf_tag) d = res_cf;
    if (!d.rp && tag && tag == inverse_tag) d = inverse;
    if (!d.rp && tag && tag == fc_tag) d = res_fc;
    if (!d.rp || d.m != A.m || d.nnz != A.nnz()) throw std::logic_error("materialize: the device copy of the operator is gone");
    A = down(d);
    A.device_tag = tag;
    std::lock_guard<std::mutex> lk(mu_kept);
    host_missing.erase(tag);
  }
  void release_temporaries() override {
    settle();
    drop_resident(); dcsr_free(res_ff); ff_tag = 0;
    bool only; { std::lock_guard<std::mutex> lk(mu_kept); only = cf_tag && host_missing.count(cf_tag); }
    if (!only) { dcsr_free(res_cf); cf_tag = 0; }
  }
  void release_copies() override {
    settle();
    std::lock_guard<std::mutex> lk(mu_kept);
    for (auto it = kept.begin(); it != kept.end();) {
      if (host_missing.count(it->first)) { ++it; continue; }
      kept_bytes -= bytes_of(it->second); dcsr_free(it->second); it = kept.erase(it);
    }
  }

  // the F / C blocks of the resident level operator
  bool split_fc(const Csr& A, const Arr<int32_t>& markers, const Arr<int32_t>& newidx, int64_t n_f, int64_t n_c,
                Csr& A_ff, Csr& A_fc, Csr& A_cf, Csr* A_cc) override {
    if (!is_resident(A)) return false;
    const int64_t n = A.m;
    Arr<uint32_t> code(n);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) code[i] = (uint32_t)newidx[i] | (markers[i] == F_POINT ? 0x80000000u : 0u);
    double t0 = now_s();
    uint32_t* d_code = (uint32_t*)dev_alloc(sizeof(uint32_t) * (size_t)n);
    staged_h2d(d_code, code.data(), sizeof(uint32_t) * (size_t)n);
    t_transfer += now_s() - t0;
    t0 = now_s();
    DCsr dff, dfc, dcf, dcc;
    dev_stage("setup: F / C block extraction");
    dev_split_fc(resident, d_code, n_f, n_c, dff, dfc, dcf, A_cc ? &dcc : nullptr, stream);
    dev_free(d_code);
    dev_stage("setup: F / C blocks, downloads");
    t_extract += now_s() - t0;
    // A_ff is read by the host at once (diagonal test, polynomial coefficients)
    A_ff = down(dff);
    const bool defer_cf = !A_cc && can_defer(dcf);       // only the Z product reads the C-F block
    // the F-C block: only the one-point prolongator reads it (one_point_w below does that here)
    retire(res_fc, fc_tag);
    if (!A_cc && can_defer(dfc)) { A_fc = stub_of(dfc, ++next_tag); std::lock_guard<std::mutex> lk(mu_kept); host_missing.insert(A_fc.device_tag); }
    else { A_fc = down(dfc); A_fc.device_tag = ++next_tag; }
    res_fc = dfc; fc_tag = A_fc.device_tag;
    if (!defer_cf) A_cf = down(dcf);
    if (A_cc) { *A_cc = down(dcc); dcsr_free(dcc); }
    // A_ff feeds the polynomial assembly, A_cf the Z product: they stay
    dcsr_free(res_ff);
    { bool only; { std::lock_guard<std::mutex> lk(mu_kept); only = cf_tag && host_missing.count(cf_tag); } if (only) retire(res_cf, cf_tag); else dcsr_free(res_cf); }   // a C-F block no product came for
    res_ff = dff; ff_tag = A_ff.device_tag = ++next_tag;
    if (defer_cf) { A_cf = stub_of(dcf, ++next_tag); std::lock_guard<std::mutex> lk(mu_kept); host_missing.insert(A_cf.device_tag); }
    else A_cf.device_tag = ++next_tag;
    res_cf = dcf; cf_tag = A_cf.device_tag;
    return true;
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
