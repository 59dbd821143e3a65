// Outer Krylov solvers on the device with PETSc semantics: KSPRICHARDSON (scale 1) and
// KSPGMRES (restart 30, classical Gram-Schmidt without refinement, Givens-rotated
// Hessenberg residual estimate, happy-breakdown tolerance 1e-30, KSPConvergedDefault with
// the zero-rhs / non-zero-guess special case).  These are the callers of PCApply
// (SURVEY.md 8a-a19); the reference leaves them to PETSc.
#include "pcair_dev.hpp"
#include <cmath>
#include <string>

namespace airhip {

namespace {
struct ConvState { double bnorm = 0, rnorm0 = 0, ttol = 0; };
// KSPConvergedDefault
int converged_default(ConvState& st, int it, double rnorm, const KspOptions& o) {
  if (it == 0) {
    double snorm = st.bnorm;
    if (snorm == 0.0) snorm = rnorm;   // zero RHS and non-zero initial guess
    st.rnorm0 = snorm;
    st.ttol = std::max(o.rtol * snorm, o.atol);
  }
  if (std::isnan(rnorm) || std::isinf(rnorm)) return -9;
  if (rnorm <= st.ttol) return rnorm < o.atol ? 3 : 2;
  if (rnorm >= o.dtol * st.rnorm0) return -4;
  return 0;
}
}  // namespace

DevKSP::DevKSP(int64_t n) : n_(n) {
  w_.alloc(n); t_.alloc(n);
  scratch_.alloc(mdot_scratch_doubles(64));
  ddots_.alloc(64); dcoef_.alloc(64); dnorm_.alloc(1);
}

// the timed sections of the solver's own kernels (KspOperator::prof_begin / prof_end; level -1 of the owner's statistics)
struct DevKSP::Timed {
  const KspOperator& op;
  Timed(const KspOperator& o, const char* name, double bytes, const char* kernel) : op(o) { if (op.prof_begin) op.prof_begin(name, bytes, kernel); }
  ~Timed() { if (op.prof_end) op.prof_end(); }
};

// sums of the local parts (device, n doubles at d) to the host; summed over the ranks when distributed
void DevKSP::fetch_dots(const double* d, double* host, int n, hipStream_t s) {
  if (comm_) { comm_->allreduce_sum(const_cast<double*>(d), host, n, s); return; }
  HIP_CHECK(hipMemcpyAsync(host, d, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
}

double DevKSP::norm2(const KspOperator& op, const double* v, hipStream_t s) {
  { Timed t(op, "Krylov norm", 8.0 * n_, "k_mdot_partial"); launch_mdot(n_, v, v, 0, 1, scratch_.p, ddots_.p, s); }
  double h = 0.0;
  fetch_dots(ddots_.p, &h, 1, s);
  return std::sqrt(h);
}
void DevKSP::scale(const KspOperator& op, double a, double* v, hipStream_t s) {
  Timed t(op, "Krylov scale", 16.0 * n_, "k_axpby");
  launch_axpby(n_, a, v, 0.0, nullptr, v, s);
}

KspResult DevKSP::solve(const KspOperator& op, const KspOptions& o, const double* d_b, double* d_x) {
  hipEvent_t e0, e1;
  HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventCreate(&e1));
  comm_ = op.comm;
  HIP_CHECK(hipStreamSynchronize(op.stream));
  HIP_CHECK(hipEventRecord(e0, op.stream));
  KspResult r = (o.type == 0) ? richardson(op, o, d_b, d_x) : gmres(op, o, d_b, d_x);
  HIP_CHECK(hipEventRecord(e1, op.stream));
  HIP_CHECK(hipEventSynchronize(e1));
  float ms = 0.f; HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
  r.solve_ms = ms;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return r;
}

static void op_mult(const KspOperator& op, const double* x, const SpmvEpi& e) {
  if (op.mult) op.mult(x, e); else launch_spmv(*op.A, x, e, op.stream);
}

// r = b - A x
static void residual(const KspOperator& op, const double* b, const double* x, double* r) {
  SpmvEpi e; e.out = r; e.alpha = -1.0; e.z = b; e.beta = 1.0;
  op_mult(op, x, e);
}

KspResult DevKSP::richardson(const KspOperator& op, const KspOptions& o, const double* b, double* x) {
  KspResult R;
  hipStream_t s = op.stream;
  const bool prec = (o.norm_type == 0);
  double* r = w_.p; double* z = t_.p;
  ConvState st;
  double bn = norm2(op, b, s);
  if (prec && bn != 0.0) { if (op.pc) { op.pc(b, z); R.pc_applies++; } else HIP_CHECK(hipMemcpyAsync(z, b, 8 * (size_t)n_, hipMemcpyDeviceToDevice, s)); bn = norm2(op, z, s); }
  st.bnorm = bn;
  auto apply_pc = [&](const double* in, double* out) {
    if (op.pc) { op.pc(in, out); R.pc_applies++; } else HIP_CHECK(hipMemcpyAsync(out, in, 8 * (size_t)n_, hipMemcpyDeviceToDevice, s));
  };
  residual(op, b, x, r);
  double rnorm = 0.0;
  int reason = 0, its = 0;
  for (int i = 0; i < o.max_it; i++) {
    if (prec) { apply_pc(r, z); rnorm = norm2(op, z, s); } else rnorm = norm2(op, r, s);
    R.history.push_back(rnorm);
    reason = converged_default(st, i, rnorm, o);
    if (reason) break;
    if (!prec) apply_pc(r, z);
    { Timed t(op, "Krylov x+=z", 24.0 * n_, "k_axpby"); launch_axpby(n_, 1.0, z, 1.0, x, x, s); }
    its++;
    residual(op, b, x, r);
  }
  if (!reason) {
    if (prec) { apply_pc(r, z); rnorm = norm2(op, z, s); } else rnorm = norm2(op, r, s);
    R.history.push_back(rnorm);
    if (its >= o.max_it) { reason = converged_default(st, its, rnorm, o); if (!reason) reason = -3; }
  }
  R.its = its; R.reason = reason; R.rnorm = rnorm;
  return R;
}

// GMRES keeps its basis UNNORMALISED when g_unnormalised is set (airhip_set_tuning "ksp_unnormalised"): u_k = sig_k v_k with
// sig_k = ||u_k|| known on the host.  Everything the method needs from the normalised basis follows on the host - the
// preconditioner and the operator are linear, so A M u_k = sig_k A M v_k; <w, v_j> = <w~, u_j> / (sig_k sig_j); w -= h_j v_j
// becomes w~ -= (<w~, u_j> / sig_j^2) u_j; h_{k+1,k} = ||w~|| / sig_k - and the pass that scaled every new basis vector
// (read + write of a vector per iteration, 0.3 ms of 2.7 at 464^3) goes.  Same algorithm, same Hessenberg matrix up to
// rounding (each entry differs by a few ulp from the normalised run, like any other order of the sums); sig_k shrinks with
// the residual (1e-10 at convergence), far from the range limits of a double.
static int g_unnormalised = 1;
// Right-preconditioned GMRES KEEPS THE PRECONDITIONED BASIS when g_keep_pc is set (airhip_set_tuning
// "ksp_keep_preconditioned", default 1): z_j = M v_j, which every Arnoldi step computes anyway on its way to A M v_j, is
// written into a second basis Z instead of a work vector, and the solution update is x += Z y - one multi-axpy - instead of
// PETSc's x += M (V y) (KSPGMRESBuildSoln -> KSPUnwindPreconditioner), which costs one more PCApply per restart cycle and
// per solve: 13 V-cycles instead of 12 for the 12 iterations of the 464^3 solve.  M is a fixed linear operator (the AIR
// V-cycle: fixed polynomials, no inner convergence test), so Z y = M (V y) exactly in exact arithmetic; the Arnoldi
// process, the Hessenberg matrix, every residual norm and the iteration count are untouched - only the last rounding of x
// differs (tests/test_gpu_ksp_variants.py).  This is what PETSc's KSPFGMRES stores; for a linear preconditioner the two
// methods generate the same iterates.  Price: restart more vectors of device memory (24 GB at 464^3); when that allocation
// fails the solver falls back to PETSc's form.  Left preconditioning never had the extra application.
static int g_keep_pc = 1;
int ksp_set_tuning(const char* key, int value) {
  if (std::string(key) == "ksp_unnormalised") { g_unnormalised = value != 0; return 0; }
  if (std::string(key) == "ksp_keep_preconditioned") { g_keep_pc = value != 0; return 0; }
  return -1;
}

KspResult DevKSP::gmres(const KspOperator& op, const KspOptions& o, const double* b, double* x) {
  KspResult R;
  const bool unn = g_unnormalised != 0;
  hipStream_t s = op.stream;
  const int m = o.restart;
  const bool right = (o.pc_side == 1);
  if (v_alloc_ < m + 1) { V_.alloc((int64_t)(m + 1) * n_); v_alloc_ = m + 1; }
  auto Vk = [&](int k) { return V_.p + (int64_t)k * n_; };
  // the preconditioned basis (see g_keep_pc): right preconditioning with a real preconditioner only
  bool keep = right && g_keep_pc != 0 && (bool)op.pc;
  if (keep && z_alloc_ < m) {
    try { Z_.alloc((int64_t)m * n_); z_alloc_ = m; }
    catch (const std::exception&) { (void)hipGetLastError(); Z_.release(); z_alloc_ = 0; keep = false; }      // no room: PETSc's form
  }
  if (comm_ && right && g_keep_pc != 0 && (bool)op.pc) {
    // distributed: the form of the solution update decides how many PCApply calls (collective exchanges) a solve makes,
    // so every rank must take the same one - all keep the basis, or none (a rank whose allocation failed decides for all)
    const double mine = keep ? 1.0 : 0.0;
    double all = 0.0;
    HIP_CHECK(hipMemcpyAsync(dnorm_.p, &mine, sizeof(double), hipMemcpyHostToDevice, s));
    fetch_dots(dnorm_.p, &all, 1, s);
    if (all < (double)comm_->size - 0.5) keep = false;
  }
  if (!keep && z_alloc_ > 0) { Z_.release(); z_alloc_ = 0; }
  auto Zk = [&](int k) { return Z_.p + (int64_t)k * n_; };
  R.pc_applies = 0;
  auto apply_pc = [&](const double* in, double* out) {
    if (op.pc) { op.pc(in, out); R.pc_applies++; } else HIP_CHECK(hipMemcpyAsync(out, in, 8 * (size_t)n_, hipMemcpyDeviceToDevice, s));
  };
  ConvState st;
  double bn = norm2(op, b, s);
  if (!right && bn != 0.0) { apply_pc(b, t_.p); bn = norm2(op, t_.p, s); }
  st.bnorm = bn;

  int its = 0, reason = 0;
  double rnorm = 0.0;
  const double haptol = 1e-30;
  bool first = true;
  std::vector<double> hh((size_t)(m + 2) * (m + 1)), cs(m + 1), sn(m + 1), grs(m + 2), lhh(m + 1), nrs(m + 1);
  std::vector<double> sig(m + 2, 1.0);      // norms of the stored basis vectors (all 1 when they are kept normalised)
  auto H = [&](int i, int j) -> double& { return hh[(size_t)j * (m + 2) + i]; };
  while (!reason) {
    // V0 = (B)(b - A x)
    if (right) residual(op, b, x, Vk(0));
    else { residual(op, b, x, w_.p); apply_pc(w_.p, Vk(0)); }
    double res = norm2(op, Vk(0), s);
    rnorm = res;
    if (first) {
      R.history.push_back(res);
      if (res == 0.0) { reason = 3; break; }
      reason = converged_default(st, its, res, o);
      first = false;
      if (reason) break;
    }
    if (unn) sig[0] = res; else { scale(op, 1.0 / res, Vk(0), s); sig[0] = 1.0; }
    std::fill(hh.begin(), hh.end(), 0.0);
    grs[0] = res;
    int it = 0;
    bool hapend = false;
    while (!reason && it < m && its < o.max_it) {
      double* w = Vk(it + 1);
      if (right) { double* z = keep ? Zk(it) : t_.p; apply_pc(Vk(it), z); SpmvEpi e; e.out = w; op_mult(op, z, e); }
      else { SpmvEpi e; e.out = t_.p; op_mult(op, Vk(it), e); apply_pc(t_.p, w); }
      // classical Gram-Schmidt: one multi-dot and one multi-axpy, each a single pass over w and the basis; the norm of
      // the orthogonalised vector rides along in the multi-axpy
      // (statistics per basis size: the bytes of a call site are fixed)
      const std::string tag = " (" + std::to_string(it + 1) + ")";
      { Timed t(op, ("Krylov multi-dot" + tag).c_str(), 8.0 * n_ * (it + 2), "k_mdot_partial"); launch_mdot(n_, w, V_.p, n_, it + 1, scratch_.p, ddots_.p, s); }
      fetch_dots(ddots_.p, lhh.data(), it + 1, s);
      for (int j = 0; j <= it; j++) { H(j, it) = lhh[j] / (sig[it] * sig[j]); nrs[j] = -lhh[j] / (sig[j] * sig[j]); }
      HIP_CHECK(hipMemcpyAsync(dcoef_.p, nrs.data(), sizeof(double) * (it + 1), hipMemcpyHostToDevice, s));
      { Timed t(op, ("Krylov multi-axpy + norm" + tag).c_str(), 8.0 * n_ * (it + 3), "k_maxpy"); launch_maxpy(n_, w, V_.p, n_, it + 1, dcoef_.p, s, false, scratch_.p, dnorm_.p); }
      double tt = 0.0;
      fetch_dots(dnorm_.p, &tt, 1, s);
      tt = std::sqrt(tt);                         // norm of the vector as stored
      if (unn) sig[it + 1] = tt != 0.0 ? tt : 1.0;
      else { if (tt != 0.0) scale(op, 1.0 / tt, w, s); sig[it + 1] = 1.0; }
      tt /= sig[it];                              // h_{it+1,it}: the norm the normalised method would have found
      H(it + 1, it) = tt;
      double hapbnd = grs[it] != 0.0 ? std::fabs(tt / grs[it]) : haptol;
      if (hapbnd > haptol) hapbnd = haptol;
      if (tt < hapbnd) hapend = true;
      // KSPGMRESUpdateHessenberg
      for (int j = 0; j < it; j++) {
        double t1 = H(j, it);
        H(j, it) = cs[j] * t1 + sn[j] * H(j + 1, it);
        H(j + 1, it) = -sn[j] * t1 + cs[j] * H(j + 1, it);
      }
      if (!hapend) {
        double t = std::sqrt(H(it, it) * H(it, it) + H(it + 1, it) * H(it + 1, it));
        if (t == 0.0) { reason = -5; break; }
        cs[it] = H(it, it) / t; sn[it] = H(it + 1, it) / t;
        grs[it + 1] = -sn[it] * grs[it];
        grs[it] = cs[it] * grs[it];
        H(it, it) = cs[it] * H(it, it) + sn[it] * H(it + 1, it);
        res = std::fabs(grs[it + 1]);
      } else res = 0.0;
      it++; its++;
      rnorm = res;
      R.history.push_back(res);
      reason = converged_default(st, its, res, o);
      if (hapend) { if (!reason) reason = -5; break; }
    }
    // KSPGMRESBuildSoln
    if (it > 0) {
      const int k = it - 1;
      std::fill(nrs.begin(), nrs.end(), 0.0);
      if (H(k, k) != 0.0) {
        nrs[k] = grs[k] / H(k, k);
        for (int ii = 1; ii <= k; ii++) {
          const int kk = k - ii;
          double tsum = grs[kk];
          for (int j = kk + 1; j <= k; j++) tsum -= H(kk, j) * nrs[j];
          nrs[kk] = tsum / H(kk, kk);
        }
      }
      for (int j = 0; j < it; j++) nrs[j] /= sig[j];      // coefficients of the stored (unnormalised) vectors
      HIP_CHECK(hipMemcpyAsync(dcoef_.p, nrs.data(), sizeof(double) * it, hipMemcpyHostToDevice, s));
      if (right && keep) {
        // x += Z y: the preconditioned vectors are at hand, no further PCApply
        Timed t(op, ("Krylov solution multi-axpy (" + std::to_string(it) + ")").c_str(), 8.0 * n_ * (it + 2), "k_maxpy");
        launch_maxpy(n_, x, Z_.p, n_, it, dcoef_.p, s);
      } else if (right) {
        { Timed t(op, ("Krylov solution multi-axpy (" + std::to_string(it) + ")").c_str(), 8.0 * n_ * (it + 1), "k_maxpy"); launch_maxpy(n_, w_.p, V_.p, n_, it, dcoef_.p, s, true); }
        apply_pc(w_.p, t_.p);
        Timed t(op, "Krylov x+=z", 24.0 * n_, "k_axpby");
        launch_axpby(n_, 1.0, t_.p, 1.0, x, x, s);
      } else { Timed t(op, ("Krylov solution multi-axpy (" + std::to_string(it) + ")").c_str(), 8.0 * n_ * (it + 2), "k_maxpy"); launch_maxpy(n_, x, V_.p, n_, it, dcoef_.p, s); }
      HIP_CHECK(hipStreamSynchronize(s));   // nrs (host) is reused by the next cycle
    }
    if (!reason && its >= o.max_it) reason = -3;
  }
  R.its = its; R.reason = reason; R.rnorm = rnorm;
  return R;
}

}  // namespace airhip
