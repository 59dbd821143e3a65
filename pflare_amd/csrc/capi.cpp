// extern "C" boundary of libairhip (see include/airhip.h).
#include "../../include/airhip.h"
#include "host/air.hpp"
#include "device/pcair_dev.hpp"
#include "device/setup_accel.hpp"
#include "device/spgemm.hpp"
#include "device/extract.hpp"
#include "device/comm.hpp"
#include "device/devmem.hpp"
#include "host/dist.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <mutex>
#include <exception>

using namespace airhip;

static thread_local std::string g_err;
const char* airhip_last_error(void) { return g_err.c_str(); }
int airhip_version(void) { return 100; }

struct TeamGuard { int prev; TeamGuard(); ~TeamGuard(); };      // host threads of one API call (defined below)
#define API_BEGIN try { TeamGuard team_guard_;
#define API_END                                                                      \
  }                                                                                  \
  catch (const std::bad_alloc&) { g_err = "out of host memory"; return AIRHIP_ERR_INTERNAL; } \
  catch (const std::exception& e) {                                                  \
    g_err = e.what();                                                                \
    return (g_err.find("HIP error") != std::string::npos) ? AIRHIP_ERR_HIP : AIRHIP_ERR_INTERNAL; \
  }                                                                                  \
  return AIRHIP_OK;
#define REQUIRE(cond, code, msg) do { if (!(cond)) { g_err = msg; return code; } } while (0)

int airhip_device_count(int* count) {
  API_BEGIN
  REQUIRE(count, AIRHIP_ERR_ARG, "null argument");
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *count = c;
  API_END
}
int airhip_set_device(int device) {
  API_BEGIN
  HIP_CHECK(hipSetDevice(device));
  API_END
}

// performance knobs of the kernels (see device/spmv.hip); declared in the header, not part of the stable ABI
int airhip_set_tuning(const char* key, int value) {
  API_BEGIN
  REQUIRE(key, AIRHIP_ERR_ARG, "null argument");
  if (std::string(key) == "spgemm_pool_entries") airhip::spgemm_set_pool_limit(value);
  else if (std::string(key) == "spgemm_pool_chunk") airhip::spgemm_set_pool_chunk(value);
  else if (std::string(key) == "block_order_min_blocks") airhip::set_block_order_min(value);
  else if (std::string(key) == "fc_fuse") airhip::set_fc_fuse(value != 0);
  else if (std::string(key) == "profile_chain") airhip::set_profile_chain(value != 0);
  else if (std::string(key) == "perm_chunks") airhip::set_perm_chunks(value != 0);
  else if (std::string(key) == "fuse_first_sweep") airhip::set_fuse_first_sweep(value);
  else if (std::string(key) == "fuse_first_sweep_run") airhip::set_fuse_first_sweep_run(value != 0);
  else if (std::string(key) == "fc_spatial_x") airhip::set_fc_spatial(value);
  else if (airhip::kernels_set_tuning(key, value) == 0) {}
  else if (airhip::ksp_set_tuning(key, value) == 0) {}
  else if (airhip::spmv_set_tuning(key, value) != 0) { g_err = std::string("unknown tuning key ") + key; return AIRHIP_ERR_ARG; }
  API_END
}

// ---- host threads.  OpenMP's default is one thread per CPU the process may be scheduled on; in a container
// with a CPU *quota* (cgroup cpu.max) that is the whole machine, and a team of 128+ threads on a 16-CPU share
// turns every parallel region of a small setup into a scheduler storm (measured: 16x slower with 64 threads on
// 8 CPUs).  So unless OMP_NUM_THREADS or airhip_set_num_threads says otherwise, the team is the CPU share,
// at most 32 (the measured optimum of the memory-bound setup phases), and small operators get fewer threads.
static int read_int_file(const char* path, long long* a, long long* b) {
  FILE* f = std::fopen(path, "r");
  if (!f) return 0;
  char t0[64] = {0}, t1[64] = {0};
  int n = std::fscanf(f, "%63s %63s", t0, t1);
  std::fclose(f);
  if (n < 1) return 0;
  if (std::strcmp(t0, "max") == 0) *a = -1; else *a = std::atoll(t0);
  if (n > 1 && b) *b = std::atoll(t1);
  return n;
}
static int cpu_share() {
  int n = omp_get_num_procs();                       // honours the affinity mask
  long long q = -1, per = 100000;
  if (read_int_file("/sys/fs/cgroup/cpu.max", &q, &per) >= 1 && q > 0 && per > 0)          // cgroup v2
    n = std::min<long long>(n, std::max<long long>(1, (q + per - 1) / per));
  else if (read_int_file("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", &q, nullptr) >= 1 && q > 0 &&
           read_int_file("/sys/fs/cgroup/cpu/cpu.cfs_period_us", &per, nullptr) >= 1 && per > 0)   // cgroup v1
    n = std::min<long long>(n, std::max<long long>(1, (q + per - 1) / per));
  return std::max(1, n);
}
// The library never changes the process-wide OpenMP setting behind the application's back (it may share the OpenMP
// runtime with PETSc and the user's own parallel regions): the team is chosen when an API call is entered and the
// caller's setting is restored when the call returns (TeamGuard in API_BEGIN).
static int g_team_override = 0;      // airhip_set_num_threads
static int library_team() {
  if (g_team_override > 0) return g_team_override;
  static const int team = [] {
    const char* e = std::getenv("OMP_NUM_THREADS");
    return (e && *e) ? omp_get_max_threads() : std::min(cpu_share(), 32);
  }();
  return team;
}
TeamGuard::TeamGuard() : prev(omp_get_max_threads()) { omp_set_num_threads(library_team()); }
TeamGuard::~TeamGuard() { omp_set_num_threads(prev); }
// fewer threads for small operators: one thread per ~32k non-zeros
struct ThreadScope {
  int prev;
  explicit ThreadScope(int64_t nnz) : prev(omp_get_max_threads()) {
    int want = (int)std::min<int64_t>(prev, std::max<int64_t>(1, nnz / 32768));
    omp_set_num_threads(want);
  }
  ~ThreadScope() { omp_set_num_threads(prev); }
};

int airhip_set_num_threads(int n, int* prev) {
  if (prev) *prev = library_team();
  if (n > 0) g_team_override = n;
  return AIRHIP_OK;
}

// ------------------------------------------------------------------------ host matrices
struct airhip_hostmat_s { Csr A; };

// Operators cross the boundary as caller memory: refuse anything the kernels would index out of range with
// (an out-of-bounds access on the device can take the whole node down).  Rows must hold strictly ascending
// column indices (the MatSeqAIJ convention the reference relies on).
static const char* check_csr(const airhip_csr* a, bool square) {
  if (!a) return "null matrix";
  if (a->m < 0 || a->n < 0) return "negative matrix dimension";
  if (square && a->m != a->n) return "operator must be square";
  if (a->m > INT32_MAX || a->n > INT32_MAX) return "dimension exceeds 32-bit indices";
  if (a->m == 0) return nullptr;
  if (!a->rowptr) return "null rowptr";
  if (a->rowptr[0] != 0) return "rowptr[0] must be 0";
  const int64_t m = a->m, n = a->n;
  int bad = 0;
#pragma omp parallel for schedule(static) reduction(max : bad)
  for (int64_t i = 0; i < m; i++) if (a->rowptr[i + 1] < a->rowptr[i]) bad = 1;
  if (bad) return "rowptr is not non-decreasing";
  if (a->rowptr[m] > 0 && (!a->col || !a->val)) return "null col / val";
#pragma omp parallel for schedule(static) reduction(max : bad)
  for (int64_t i = 0; i < m; i++) {
    int32_t prev = -1;
    for (int64_t k = a->rowptr[i]; k < a->rowptr[i + 1]; k++) {
      const int32_t c = a->col[k];
      if (c < 0 || c >= n) bad = std::max(bad, 2);
      else if (c <= prev) bad = std::max(bad, 1);
      prev = c;
    }
  }
  if (bad == 2) return "column index out of range";
  if (bad == 1) return "column indices of a row must be strictly ascending";
  return nullptr;
}
#define REQUIRE_CSR(a, square)                                                         \
  do { if (const char* _w = check_csr((a), (square))) { g_err = _w; return AIRHIP_ERR_ARG; } } while (0)

static void view_csr(const airhip_csr* a, Csr& out) {
  out.m = a->m; out.n = a->n;
  out.rp.view(a->rowptr, a->m + 1);
  int64_t nnz = a->rowptr[a->m];
  out.ci.view(a->col, nnz);
  out.v.view(a->val, nnz);
}
static Csr copy_csr(const airhip_csr* a) {
  Csr v; view_csr(a, v);
  return v.clone();
}
static void expose_csr(const Csr& A, airhip_csr* v) {
  v->m = A.m; v->n = A.n; v->rowptr = A.rp.data(); v->col = A.ci.data(); v->val = A.v.data();
}

int airhip_compute_cf_splitting(const airhip_csr* A, int symmetric, double strong_threshold, int max_luby_steps,
                                int cf_splitting_type, int ddc_its, double fraction_swap, int32_t* markers) {
  API_BEGIN
  REQUIRE(A && markers, AIRHIP_ERR_ARG, "null argument");
  REQUIRE_CSR(A, true);
  AirOptions o;
  o.symmetric = symmetric != 0; o.strong_threshold = strong_threshold; o.max_luby_steps = max_luby_steps;
  o.ddc_its = ddc_its; o.ddc_fraction = fraction_swap;
  if (cf_splitting_type == 0) o.cf_splitting_type = 0;
  else if (cf_splitting_type == 1) o.cf_splitting_type = 1;
  else if (cf_splitting_type == 6) o.cf_splitting_type = 2;
  else { g_err = "CF splitting type outside this build's scope (pmisr_ddc, diag_dom, cr)"; return AIRHIP_ERR_UNSUPPORTED; }
  Csr view; view_csr(A, view);
  ThreadScope threads(A->rowptr[A->m]);
  Arr<int32_t> m;
  compute_cf_splitting(view, o, m);
  for (int64_t i = 0; i < A->m; i++) markers[i] = m[i];
  API_END
}

// compute_diag_dom_submatrix (include/pflare.h:114, src/MatDiagDomSubmatrix.F90:17-76): the F-F block of a diag_dom CF
// splitting whose strong threshold is the ratio asked for - every row of it has sum |offdiag| / |diag| < max_dd_ratio
int airhip_compute_diag_dom_submatrix(const airhip_csr* A, double max_dd_ratio, airhip_hostmat* out) {
  API_BEGIN
  REQUIRE(A && out, AIRHIP_ERR_ARG, "null argument");
  REQUIRE_CSR(A, true);
  REQUIRE(max_dd_ratio > 0.0 && max_dd_ratio < 1.0, AIRHIP_ERR_ARG, "max_dd_ratio input to compute_diag_dom_submatrix must be (0.0, 1.0)");
  AirOptions o;
  o.symmetric = false; o.strong_threshold = max_dd_ratio; o.max_luby_steps = -1; o.ddc_its = 0; o.ddc_fraction = 0.0;
  o.cf_splitting_type = 1;
  Csr view; view_csr(A, view);
  ThreadScope threads(A->rowptr[A->m]);
  Arr<int32_t> m;
  compute_cf_splitting(view, o, m);
  const int64_t n = view.m;
  int64_t nf = 0;
  Arr<int32_t> newidx(n);
  for (int64_t i = 0; i < n; i++) newidx[i] = m[i] == F_POINT ? (int32_t)nf++ : -1;
  auto* h = new airhip_hostmat_s;
  Csr& S = h->A; S.m = S.n = nf; S.rp.alloc(nf + 1);
  S.rp[0] = 0;
  for (int64_t i = 0; i < n; i++) {
    if (m[i] != F_POINT) continue;
    int64_t c = 0;
    for (int64_t k = view.rp[i]; k < view.rp[i + 1]; k++) c += m[view.ci[k]] == F_POINT ? 1 : 0;
    S.rp[newidx[i] + 1] = S.rp[newidx[i]] + c;
  }
  S.ci.alloc(S.rp[nf]); S.v.alloc(S.rp[nf]);
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < n; i++) {
    if (m[i] != F_POINT) continue;
    int64_t o2 = S.rp[newidx[i]];
    for (int64_t k = view.rp[i]; k < view.rp[i + 1]; k++)
      if (m[view.ci[k]] == F_POINT) { S.ci[o2] = newidx[view.ci[k]]; S.v[o2] = view.v[k]; o2++; }
  }
  *out = h;
  API_END
}

// the same splitting computed by the device kernels (device/cfsplit.hip); PMISR + DDC only
int airhip_compute_cf_splitting_on_device(const airhip_csr* A, int symmetric, double strong_threshold, int max_luby_steps,
                                          int ddc_its, double fraction_swap, int32_t* markers) {
  API_BEGIN
  REQUIRE(A && markers, AIRHIP_ERR_ARG, "null argument");
  REQUIRE_CSR(A, true);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_err = "no HIP device"; return AIRHIP_ERR_HIP; }
  AirOptions o;
  o.symmetric = symmetric != 0; o.strong_threshold = strong_threshold; o.max_luby_steps = max_luby_steps;
  o.ddc_its = ddc_its; o.ddc_fraction = fraction_swap; o.cf_splitting_type = 0;
  Csr view; view_csr(A, view);
  auto accel = make_device_setup_accel(nullptr, 0);
  Arr<int32_t> m;
  REQUIRE(accel->cf_splitting(view, o, m), AIRHIP_ERR_STATE, "the device CF splitting declined this operator");
  for (int64_t i = 0; i < A->m; i++) markers[i] = m[i];
  accel.reset();
  dev_cache_trim();
  API_END
}

int airhip_gen_adv_1d(int64_t n, airhip_hostmat* out) {
  API_BEGIN
  REQUIRE(out && n > 0, AIRHIP_ERR_ARG, "bad argument");
  auto* h = new airhip_hostmat_s; h->A = gen_adv_1d(n); *out = h;
  API_END
}
int airhip_gen_adv_diff_fd(int dim, int64_t M, int64_t N, int64_t P, double u, double v, double w,
                           int unit_velocity, double alpha, double L_x, double L_y, double L_z,
                           int adv_nondim, airhip_hostmat* out) {
  API_BEGIN
  REQUIRE(out && (dim == 2 || dim == 3) && M > 1 && N > 1 && (dim == 2 || P > 1), AIRHIP_ERR_ARG,
          "Grid dimensions must be at least 2 points in each active direction");
  if (alpha != 0.0 && adv_nondim > 0) { g_err = "Non-dimensional advection only applies without diffusion"; return AIRHIP_ERR_ARG; }
  FdOptions o; o.dim = dim; o.M = M; o.N = N; o.P = P; o.u = u; o.v = v; o.w = w;
  o.unit_velocity = unit_velocity != 0; o.alpha = alpha; o.L_x = L_x; o.L_y = L_y; o.L_z = L_z; o.adv_nondim = adv_nondim;
  auto* h = new airhip_hostmat_s; h->A = gen_adv_diff_fd(o); *out = h;
  API_END
}
int airhip_hostmat_from_csr(const airhip_csr* A, airhip_hostmat* out) {
  API_BEGIN
  REQUIRE(A && out, AIRHIP_ERR_ARG, "null argument");
  REQUIRE_CSR(A, false);
  auto* h = new airhip_hostmat_s; h->A = copy_csr(A); *out = h;
  API_END
}
int airhip_hostmat_get(airhip_hostmat A, airhip_csr* view) {
  API_BEGIN
  REQUIRE(A && view, AIRHIP_ERR_ARG, "null argument");
  expose_csr(A->A, view);
  API_END
}
// Inspection hook (declared in the header, not part of the stable ABI): C = filter(A*B) with the host or the device SpGEMM of
// the setup; stats[5] = rows each accumulator kind worked on {registers, LDS 1k, LDS 2k, LDS 4k, global-memory table}.
int airhip_spgemm_filtered(const airhip_csr* A, const airhip_csr* B, int drop, double tol, int lump, int negate,
                                   int on_device, airhip_hostmat* out, int64_t* stats) {
  API_BEGIN
  REQUIRE(A && B && out && A->n == B->m, AIRHIP_ERR_ARG, "bad arguments");
  Csr a, b; view_csr(A, a); view_csr(B, b);
  SpgemmDrop d; d.enabled = drop != 0; d.tol = tol; d.lump = lump != 0; d.negate = negate != 0;
  auto* h = new airhip_hostmat_s;
  if (!on_device) h->A = spgemm(a, b, d);
  else {
    DCsr da = dcsr_upload(a.m, a.n, a.rp.data(), a.ci.data(), a.v.data());
    DCsr db = dcsr_upload(b.m, b.n, b.rp.data(), b.ci.data(), b.v.data());
    SpgemmStats st;
    DCsr dc = dev_spgemm(da, db, d.enabled, d.tol, d.lump, d.negate, nullptr, &st);
    h->A.m = dc.m; h->A.n = dc.n;
    h->A.rp.alloc(dc.m + 1); h->A.ci.alloc(dc.nnz); h->A.v.alloc(dc.nnz);
    dcsr_download(dc, h->A.rp.data(), h->A.ci.data(), h->A.v.data());
    dcsr_free(da); dcsr_free(db); dcsr_free(dc);
    dev_cache_trim();
    if (stats) { stats[0] = st.rows_reg; stats[1] = st.rows_1k; stats[2] = st.rows_2k; stats[3] = st.rows_4k; stats[4] = st.rows_global; }
  }
  *out = h;
  API_END
}
// Inspection hook: G = A_fc + A_ff W (one-point W given by wcol) built by the host code (host/spops.cpp) or by the device
// code (device/extract.hip) - the operator of the fused first F sweep
int airhip_fused_first_operator(const airhip_csr* A_ff, const airhip_csr* A_fc, const int32_t* wcol, int on_device, airhip_hostmat* out) {
  API_BEGIN
  REQUIRE(A_ff && A_fc && wcol && out && A_ff->m == A_fc->m && A_ff->n == A_ff->m, AIRHIP_ERR_ARG, "bad arguments");
  REQUIRE_CSR(A_ff, true); REQUIRE_CSR(A_fc, false);
  for (int64_t i = 0; i < A_fc->m; i++) REQUIRE(wcol[i] >= -1 && wcol[i] < A_fc->n, AIRHIP_ERR_ARG, "wcol out of range");
  Csr a, b; view_csr(A_ff, a); view_csr(A_fc, b);
  auto h = std::make_unique<airhip_hostmat_s>();
  if (!on_device) h->A = fused_first_operator(a, b, wcol);
  else {
    DevScope scope(true);
    DCsr da = dcsr_upload(a.m, a.n, a.rp.data(), a.ci.data(), a.v.data());
    DCsr db = dcsr_upload(b.m, b.n, b.rp.data(), b.ci.data(), b.v.data());
    int32_t* dw = (int32_t*)dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(a.m, 1));
    staged_h2d(dw, wcol, sizeof(int32_t) * (size_t)a.m);
    DCsr dg = dev_fused_first_operator(da, db, dw, nullptr);
    h->A.m = dg.m; h->A.n = dg.n;
    h->A.rp.alloc(dg.m + 1); h->A.ci.alloc(dg.nnz); h->A.v.alloc(dg.nnz);
    dcsr_download(dg, h->A.rp.data(), h->A.ci.data(), h->A.v.data());
    dcsr_free(da); dcsr_free(db); dcsr_free(dg); dev_free(dw);
  }
  *out = h.release();
  API_END
}
int airhip_hostmat_destroy(airhip_hostmat* A) {
  API_BEGIN
  if (A && *A) { delete *A; *A = nullptr; }
  API_END
}

// ------------------------------------------------------------------------------ options
static bool parse_bool(const char* v) {
  if (!v || !*v) return true;
  std::string s(v);
  return !(s == "0" || s == "false" || s == "no" || s == "False" || s == "FALSE");
}
static int parse_inv_type(const char* v) {
  std::string s(v);
  if (s == "power") return INV_POWER;
  if (s == "arnoldi") return INV_ARNOLDI;
  if (s == "newton") return INV_NEWTON;
  if (s == "newton_no_extra") return INV_NEWTON_NO_EXTRA;
  if (s == "neumann") return INV_NEUMANN;
  if (s == "wjacobi") return INV_WJACOBI;
  if (s == "jacobi") return INV_JACOBI;
  if (s == "sai") return INV_SAI;
  if (s == "isai") return INV_ISAI;
  char* end = nullptr; long k = std::strtol(v, &end, 10);
  if (end && *end == 0) return (int)k;
  return -1;
}

// Event pair that orders a private library stream after / before a caller's stream without host synchronisation.
struct StreamLink {
  hipEvent_t in = nullptr, out = nullptr;
  ~StreamLink() { if (in) (void)hipEventDestroy(in); if (out) (void)hipEventDestroy(out); }
  // everything queued on `caller` so far happens before what is queued on `lib` from now on
  void after_caller(hipStream_t lib, hipStream_t caller) {
    if (!in) HIP_CHECK(hipEventCreateWithFlags(&in, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(in, caller));
    HIP_CHECK(hipStreamWaitEvent(lib, in, 0));
  }
  // everything queued on `lib` so far happens before what is queued on `caller` from now on
  void before_caller(hipStream_t lib, hipStream_t caller) {
    if (!out) HIP_CHECK(hipEventCreateWithFlags(&out, hipEventDisableTiming));
    HIP_CHECK(hipEventRecord(out, lib));
    HIP_CHECK(hipStreamWaitEvent(caller, out, 0));
  }
};

struct airhip_pcair_s {
  AirOptions opt;
  std::unique_ptr<Hierarchy> H;
  Csr A_top;                      // owned copy of the operator (kept only until upload unless BYO)
  std::unique_ptr<DevPCAIR> dev;
  std::unique_ptr<DevKSP> ksp;
  DBuf<double> kb, kx;            // Krylov rhs / solution in hierarchical ordering
  double host_setup_s = 0, upload_s = 0;
  bool use_graph = false, profile = false;
  bool byo = false; int byo_levels = 0;
  double op_nnz_top = 0;
  bool setup_on_device = true;    // sparse products of the setup as HIP SpGEMMs
  int64_t setup_device_min_nnz = -1;
  bool device_upload = true;      // false: keep the hierarchy on the host only (it will be partitioned)
  std::vector<int64_t> hint;      // optional locality keys of the fine rows (device layout only)
  double accel_transfer_s = 0, accel_compute_s = 0;
  // ---- the host mirror of a hierarchy set up on the device: the operators whose download the setup deferred are fetched
  // by this thread after the upload (mirror_join() before anything reads pc->H's matrices)
  std::unique_ptr<SetupAccel> mirror_accel;
  std::thread mirror_thread;
  std::mutex mirror_mu;          // joining: the owner's calls and the device allocator's pressure handler
  int mirror_pressure_id = 0;
  std::exception_ptr mirror_error;
  double mirror_s = 0;
  double cplx[5] = {0, 0, 0, 0, 0};     // complexities, taken before the mirror starts
  bool cplx_valid = false;
  ~airhip_pcair_s() {
    if (mirror_pressure_id) dev_remove_pressure_handler(mirror_pressure_id);
    std::lock_guard<std::mutex> lock(mirror_mu);
    if (mirror_thread.joinable()) mirror_thread.join();
  }
  StreamLink link;                // orders the library stream against the caller's
};

int airhip_pcair_create(airhip_pcair* pc) {
  API_BEGIN
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  *pc = new airhip_pcair_s;
  API_END
}

int airhip_pcair_set_option(airhip_pcair pc, const char* name, const char* value) {
  API_BEGIN
  REQUIRE(pc && name, AIRHIP_ERR_ARG, "null argument");
  AirOptions& o = pc->opt;
  std::string k(name);
  if (k.rfind("-pc_air_", 0) == 0) k = k.substr(8);
  else if (k.rfind("pc_air_", 0) == 0) k = k.substr(7);
  const char* v = value ? value : "";
  auto I = [&]() { return std::atoi(v); };
  auto D = [&]() { return std::atof(v); };
  auto inv = [&](int& dst) -> int {
    int t = parse_inv_type(v);
    if (t < 0) { g_err = std::string("unknown inverse type ") + v; return AIRHIP_ERR_ARG; }
    dst = t; return 0;
  };
  int rc = 0;
  if (k == "max_levels") o.max_levels = I();
  else if (k == "coarse_eq_limit") o.coarse_eq_limit = I();
  else if (k == "auto_truncate_start_level") o.auto_truncate_start_level = I();
  else if (k == "auto_truncate_tol") o.auto_truncate_tol = D();
  else if (k == "strong_threshold") o.strong_threshold = D();
  else if (k == "ddc_its") o.ddc_its = I();
  else if (k == "ddc_fraction") o.ddc_fraction = D();
  else if (k == "max_luby_steps") o.max_luby_steps = I();
  else if (k == "smooth_type") { if (o.set_smooth_type(v) != 0) { g_err = "bad smooth_type"; return AIRHIP_ERR_ARG; } }
  else if (k == "diag_scale_polys") o.diag_scale_polys = parse_bool(v);
  else if (k == "matrix_free_polys") o.matrix_free_polys = parse_bool(v);
  else if (k == "one_point_classical_prolong") o.one_point_classical_prolong = parse_bool(v);
  else if (k == "inverse_type") { rc = inv(o.inverse_type); if (!rc) { /* C smoothing follows F unless set */ } }
  else if (k == "poly_order") o.poly_order = I();
  else if (k == "inverse_sparsity_order") o.inverse_sparsity_order = I();
  else if (k == "c_inverse_type") rc = inv(o.c_inverse_type);
  else if (k == "c_poly_order") o.c_poly_order = I();
  else if (k == "c_inverse_sparsity_order") o.c_inverse_sparsity_order = I();
  else if (k == "coarsest_inverse_type") rc = inv(o.coarsest_inverse_type);
  else if (k == "coarsest_poly_order") o.coarsest_poly_order = I();
  else if (k == "coarsest_inverse_sparsity_order") o.coarsest_inverse_sparsity_order = I();
  else if (k == "coarsest_matrix_free_polys") o.coarsest_matrix_free_polys = parse_bool(v);
  else if (k == "coarsest_diag_scale_polys") o.coarsest_diag_scale_polys = parse_bool(v);
  else if (k == "r_drop") o.r_drop = D();
  else if (k == "a_drop") o.a_drop = D();
  else if (k == "a_lump") o.a_lump = parse_bool(v);
  else if (k == "print_stats_timings") o.print_stats_timings = parse_bool(v);
  else if (k == "mg_coarse_ksp_max_it" || k == "-mg_coarse_ksp_max_it") o.coarse_richardson_its = I();
  else if (k == "mg_coarse_pc_type" || k == "-mg_coarse_pc_type") {
    if (std::string(v) == "lu") o.coarse_lu = true;
    else if (std::string(v) == "shell" || std::string(v) == "") o.coarse_lu = false;
    else { g_err = std::string("unsupported -mg_coarse_pc_type ") + v; return AIRHIP_ERR_UNSUPPORTED; }
  }
  else if (k == "seed") o.seed = (uint64_t)std::atoll(v);
  else if (k == "setup_on_device") pc->setup_on_device = parse_bool(v);
  else if (k == "setup_device_min_nnz") pc->setup_device_min_nnz = std::atoll(v);
  else if (k == "device_upload") pc->device_upload = parse_bool(v);

  else if (k == "processor_agglom" || k == "subcomm" || k == "coarsest_subcomm") { /* no-ops on one device */ }
  else if (k == "strong_r_threshold") o.strong_r_threshold = D();
  else if (k == "lair_distance") { o.lair_distance = I(); REQUIRE(o.lair_distance >= 1, AIRHIP_ERR_ARG, "lair_distance must be >= 1"); }
  else if (k == "z_type") {
    std::string z(v);
    if (z == "product" || z == "0") o.z_type = 0;
    else if (z == "lair" || z == "1") o.z_type = 1;
    else if (z == "lair_sai" || z == "2") o.z_type = 2;
    else { g_err = "unknown z_type " + z; return AIRHIP_ERR_ARG; }
  }
  else if (k == "improve_z_its") o.improve_z_its = I();
  else if (k == "improve_w_its") o.improve_w_its = I();
  else if (k == "full_smoothing_up_and_down") o.full_smoothing = parse_bool(v);
  else if (k == "cf_splitting_type") {
    std::string c(v);
    if (c == "pmisr_ddc" || c == "0") o.cf_splitting_type = 0;
    else if (c == "diag_dom" || c == "1") o.cf_splitting_type = 1;
    else if (c == "cr") o.cf_splitting_type = 2;
    else { g_err = "-pc_air_cf_splitting_type " + c + " is outside this build's scope (pmisr_ddc, diag_dom, cr)"; return AIRHIP_ERR_UNSUPPORTED; }
  }
  else if (k == "symmetric") o.symmetric = parse_bool(v);
  else if (k == "constrain_z" || k == "constrain_w" || k == "reuse_sparsity") {
    bool is_default = (std::string(v) == "0" || std::string(v) == "false" || std::string(v) == "" );
    if (!is_default) { g_err = "option -pc_air_" + k + " is outside this build's scope (SURVEY 8f)"; return AIRHIP_ERR_UNSUPPORTED; }
  } else { g_err = "unknown option -pc_air_" + k; return AIRHIP_ERR_ARG; }
  if (rc) return rc;
  API_END
}

// Waits for the host mirror (see airhip_pcair_s::mirror_thread); an error of the thread surfaces here
static void mirror_join(airhip_pcair pc) {
  if (!pc) return;
  if (pc->mirror_pressure_id) { dev_remove_pressure_handler(pc->mirror_pressure_id); pc->mirror_pressure_id = 0; }
  { std::lock_guard<std::mutex> lock(pc->mirror_mu); if (pc->mirror_thread.joinable()) pc->mirror_thread.join(); }
  pc->mirror_accel.reset();
  if (pc->mirror_error) { std::exception_ptr e = pc->mirror_error; pc->mirror_error = nullptr; std::rethrow_exception(e); }
}
// every operator of the hierarchy whose arrays are still on the device only
static void mirror_fetch(Hierarchy& H, SetupAccel& accel) {
  for (auto& lp : H.levels) {
    Level& l = *lp;
    for (Csr* c : {&l.A_ff, &l.A_fc, &l.A_cf, &l.A_cc, &l.Z, &l.W, &l.inv_ff.mat, &l.inv_cc.mat, &l.A_full})
      if (!c->on_host()) { const uint64_t tag = c->device_tag; accel.materialize(*c); accel.drop_device_copy(tag); }
  }
  if (!H.coarse_A.on_host()) accel.materialize(H.coarse_A);
  if (!H.coarse_inv.mat.on_host()) accel.materialize(H.coarse_inv.mat);
}
static bool hierarchy_has_stubs(const Hierarchy& H) {
  for (auto& lp : H.levels) {
    const Level& l = *lp;
    for (const Csr* c : {&l.A_ff, &l.A_fc, &l.A_cf, &l.A_cc, &l.Z, &l.W, &l.inv_ff.mat, &l.inv_cc.mat, &l.A_full}) if (!c->on_host()) return true;
  }
  return !H.coarse_A.on_host() || !H.coarse_inv.mat.on_host();
}

// AIRHIP_KEEP_DEVICE_COPIES=0: the upload copies every operator from the host again (the first version's behaviour)
static bool keep_device_copies() { static const bool on = [] { const char* e = std::getenv("AIRHIP_KEEP_DEVICE_COPIES"); return !(e && e[0] == '0'); }(); return on; }

static void finish_device(airhip_pcair pc, SetupAccel* accel = nullptr) {
  double t0 = now_s();
  pc->dev = std::make_unique<DevPCAIR>();
  pc->dev->set_device_source(accel);
  if (!pc->hint.empty()) {
    if ((int64_t)pc->hint.size() != pc->A_top.m) throw std::runtime_error("ordering hint: one key per row of the operator is required");
    pc->dev->set_ordering_hint(pc->hint.data(), (int64_t)pc->hint.size());
  }
  pc->dev->upload(*pc->H, pc->A_top);
  pc->dev->set_device_source(nullptr);
  pc->dev->set_use_graph(pc->use_graph);
  pc->dev->set_before_capture([pc]() { mirror_join(pc); });
  pc->dev->set_profile(pc->profile);
  pc->ksp.reset();
  pc->kb.release(); pc->kx.release();
  pc->op_nnz_top = (double)pc->A_top.nnz();
  pc->upload_s = now_s() - t0;
}

int airhip_pcair_setup(airhip_pcair pc, const airhip_csr* A) {
  API_BEGIN
  DevScope dev_scope(true);      // setup temporaries go back to the device also when the setup throws
  REQUIRE(pc && A && A->rowptr && A->m == A->n && A->m > 0, AIRHIP_ERR_ARG, "PCAIR needs a square CSR operator");
  REQUIRE_CSR(A, true);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device: the PCAIR solve phase has no CPU fallback"; return AIRHIP_ERR_HIP;
  }
  mirror_join(pc);
  pc->dev.reset(); pc->H.reset(); pc->cplx_valid = false; pc->mirror_s = 0.0;
  view_csr(A, pc->A_top);
  ThreadScope threads(A->rowptr[A->m]);
  double t0 = now_s();
  std::unique_ptr<SetupAccel> accel;
  {
    if (pc->setup_on_device) accel = make_device_setup_accel(nullptr, pc->setup_device_min_nnz);
    if (accel) accel->keep_copies(pc->device_upload && keep_device_copies(), (double)A->rowptr[A->m]);     // for finish_device below
    if (accel) accel->defer_downloads(pc->device_upload && keep_device_copies());  // fetched by the mirror thread afterwards
    pc->H = setup_air(pc->A_top, pc->opt, accel.get());
    if (accel) {
      pc->accel_transfer_s = accel->t_transfer; pc->accel_compute_s = accel->t_compute;
      if (pc->opt.print_stats_timings) {
        std::printf(" setup products on device: transfers %.2f s, SpGEMM %.2f s, CF splitting %.2f s, polynomial assembly %.2f s, block extraction %.2f s\n",
                    accel->t_transfer, accel->t_compute, accel->t_cfsplit, accel->t_poly, accel->t_extract);
        std::fflush(stdout);
      }
    }
  }
  pc->host_setup_s = now_s() - t0;
  pc->byo = false;
  pc->op_nnz_top = (double)pc->A_top.nnz();
  // the temporaries of the setup are gone before the hierarchy takes its own memory
  if (accel) accel->release_temporaries();
  if (pc->device_upload) finish_device(pc, accel.get());
  // the numbers print_stats and airhip_pcair_complexities report, while nothing else touches the hierarchy
  pc->cplx_valid = false;
  if (airhip_pcair_complexities(pc, &pc->cplx[0], &pc->cplx[1], &pc->cplx[2], &pc->cplx[3], &pc->cplx[4]) == 0) pc->cplx_valid = true;
  if (accel && hierarchy_has_stubs(*pc->H)) {
    // The operators whose download was deferred come to the host beside whatever the caller does next (the solve does
    // not need them); everything that reads the host hierarchy waits for this thread first (mirror_join).
    accel->release_copies();
    int dev = 0; (void)hipGetDevice(&dev);
    pc->mirror_accel = std::move(accel);
    pc->mirror_error = nullptr;
    pc->mirror_thread = std::thread([pc, dev] {
      const double t0 = now_s();
      try {
        if (hipSetDevice(dev) != hipSuccess) throw std::runtime_error("host mirror: hipSetDevice failed");
        mirror_fetch(*pc->H, *pc->mirror_accel);
      } catch (...) { pc->mirror_error = std::current_exception(); }
      pc->mirror_s = now_s() - t0;
    });
    // until the thread is through, its device copies (up to 40 % of the device) are still there: an allocation of the
    // library that finds the device full waits for it instead of failing
    pc->mirror_pressure_id = dev_add_pressure_handler([pc]() -> size_t {
      std::lock_guard<std::mutex> lock(pc->mirror_mu);
      if (!pc->mirror_thread.joinable() || std::this_thread::get_id() == pc->mirror_thread.get_id()) return 0;
      pc->mirror_thread.join();
      return 1;
    });
  }
  accel.reset();
  dev_cache_trim();
  if (pc->opt.print_stats_timings) { std::printf(" %s\n", host_array_cache_report().c_str()); std::fflush(stdout); }
  trim_host_array_cache(true);   // blocks released during the setup waited for reuse; after the upload, so that the unmapping does not get in its way
  pc->A_top = Csr();   // drop the view of the caller's arrays
  if (pc->opt.print_stats_timings) {
    // print_stats (src/AIR_MG_Stats.F90:386-416); the format tools/parse_pflare_output.py reads
    double g = 0, o = 0, c = 0, st = 0, by = 0;
    if (airhip_pcair_complexities(pc, &g, &o, &c, &st, &by) == 0) {
      std::printf(" ~~~~~~~~~~~~\n Grid complexity            : %.15g\n Operator complexity        : %.15g\n"
                  " Cycle complexity           : %.15g\n Storage complexity         : %.15g\n"
                  " Reuse storage complexity   : %.15g\n ~~~~~~~~~~~~\n", g, o, c, st, 0.0);
      std::fflush(stdout);
    }
  }
  API_END
}

// Host-only setup: builds the hierarchy without touching a device (used by CPU-side tests of
// the setup and by hierarchy export); apply / solve calls then fail with AIRHIP_ERR_STATE.
int airhip_pcair_setup_host_only(airhip_pcair pc, const airhip_csr* A) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && A && A->rowptr && A->m == A->n && A->m > 0, AIRHIP_ERR_ARG, "PCAIR needs a square CSR operator");
  REQUIRE_CSR(A, true);
  pc->dev.reset(); pc->H.reset();
  Csr view; view_csr(A, view);
  ThreadScope threads(A->rowptr[A->m]);
  double t0 = now_s();
  pc->H = setup_air(view, pc->opt);
  trim_host_array_cache(true);
  pc->host_setup_s = now_s() - t0;
  pc->op_nnz_top = (double)view.nnz();
  pc->byo = false;
  API_END
}

// ---- bring-your-own hierarchy
static void fill_inv(const airhip_invop* in, InvOp& out, int64_t n) {
  out = InvOp();
  if (!in) return;
  out.kind = (InvOp::Kind)in->kind;
  out.diag_scale = in->diag_scale != 0; out.neumann = in->neumann != 0;
  if (in->kind == AIRHIP_OP_ASSEMBLED) out.mat = copy_csr(&in->mat);
  else if (in->kind == AIRHIP_OP_DIAG) { out.d.alloc(n); std::memcpy(out.d.data(), in->diag, sizeof(double) * n); }
  else if (in->kind == AIRHIP_OP_HORNER) {
    out.coeffs.assign(in->coef, in->coef + in->ncoef);
    int nz = 0; for (int k = 0; k + 1 < in->ncoef; k++) if (in->coef[k] != 0.0) nz++;
    out.spmvs = nz;
  } else if (in->kind == AIRHIP_OP_NEWTON) {
    out.re.assign(in->coef, in->coef + in->ncoef);
    out.im.assign(in->coef_im, in->coef_im + in->ncoef);
    // SpMV count of petsc_newton (src/Gmres_Poly_Newton.F90:763-875)
    int i = 0, cnt = 0, nr = in->ncoef;
    while (i <= nr - 2) {
      if (out.im[i] == 0.0) { if (std::fabs(out.re[i]) < TOL_ZERO) { i++; continue; } cnt++; i++; }
      else { if (out.re[i] * out.re[i] + out.im[i] * out.im[i] < TOL_ZERO) { i += 2; continue; } cnt++; if (i <= nr - 3) cnt++; i += 2; }
    }
    out.spmvs = cnt;
  }
}

int airhip_pcair_begin_levels(airhip_pcair pc, int no_levels, const airhip_csr* A_top) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && no_levels >= 1 && A_top, AIRHIP_ERR_ARG, "bad argument");
  REQUIRE_CSR(A_top, true);
  pc->dev.reset();
  pc->H = std::make_unique<Hierarchy>();
  pc->H->opt = pc->opt;
  pc->H->levels.resize(no_levels - 1);
  pc->A_top = copy_csr(A_top);
  pc->byo = true; pc->byo_levels = no_levels;
  API_END
}

int airhip_pcair_set_level(airhip_pcair pc, int level, int64_t n, int64_t n_f, int64_t n_c,
                           const int32_t* F, const int32_t* C, const airhip_csr* A_ff, const double* aff_diag,
                           const airhip_csr* A_fc, const airhip_invop* inv_ff, const airhip_csr* Z,
                           const int32_t* wcol, const airhip_csr* W, const airhip_csr* A_cf,
                           const airhip_csr* A_cc, const airhip_invop* inv_cc, const int* smooth_order10) {
  API_BEGIN
  REQUIRE(pc && pc->byo && pc->H, AIRHIP_ERR_STATE, "call airhip_pcair_begin_levels first");
  REQUIRE(level >= 0 && level < pc->byo_levels - 1, AIRHIP_ERR_ARG, "level out of range");
  REQUIRE(F && C && A_fc && inv_ff && Z && (A_ff || aff_diag) && (wcol || W) && n == n_f + n_c && n_f >= 0 && n_c >= 0,
          AIRHIP_ERR_ARG, "missing level data");
  {
    // shapes and index ranges of everything the device will index with
    auto shaped = [&](const airhip_csr* a, int64_t m_, int64_t n_, const char* what) -> bool {
      if (!a) return true;
      const char* w = check_csr(a, false);
      if (!w && (a->m != m_ || a->n != n_)) w = "wrong shape";
      if (w) { g_err = std::string(what) + ": " + w; return false; }
      return true;
    };
    if (!shaped(A_ff, n_f, n_f, "A_ff") || !shaped(A_fc, n_f, n_c, "A_fc") || !shaped(Z, n_c, n_f, "Z") ||
        !shaped(W, n_f, n_c, "W") || !shaped(A_cf, n_c, n_f, "A_cf") || !shaped(A_cc, n_c, n_c, "A_cc"))
      return AIRHIP_ERR_ARG;
    if (inv_ff->kind == AIRHIP_OP_ASSEMBLED && !shaped(&inv_ff->mat, n_f, n_f, "inv_ff")) return AIRHIP_ERR_ARG;
    if (inv_cc && inv_cc->kind == AIRHIP_OP_ASSEMBLED && !shaped(&inv_cc->mat, n_c, n_c, "inv_cc")) return AIRHIP_ERR_ARG;
    for (int64_t i = 0; i < n_f; i++) REQUIRE(F[i] >= 0 && F[i] < n, AIRHIP_ERR_ARG, "F index out of range");
    for (int64_t i = 0; i < n_c; i++) REQUIRE(C[i] >= 0 && C[i] < n, AIRHIP_ERR_ARG, "C index out of range");
    if (wcol) for (int64_t i = 0; i < n_f; i++) REQUIRE(wcol[i] >= -1 && wcol[i] < n_c, AIRHIP_ERR_ARG, "wcol out of range");
  }
  auto Lp = std::make_unique<Level>();
  Level& L = *Lp;
  L.n = n; L.n_f = n_f; L.n_c = n_c;
  L.F.alloc(n_f); std::memcpy(L.F.data(), F, sizeof(int32_t) * n_f);
  L.C.alloc(n_c); std::memcpy(L.C.data(), C, sizeof(int32_t) * n_c);
  if (aff_diag) { L.aff_diag = true; L.aff_d.alloc(n_f); std::memcpy(L.aff_d.data(), aff_diag, sizeof(double) * n_f); }
  else L.A_ff = copy_csr(A_ff);
  L.A_fc = copy_csr(A_fc);
  L.Z = copy_csr(Z);
  fill_inv(inv_ff, L.inv_ff, n_f);
  if (wcol) { L.one_point = true; L.wcol.alloc(n_f); std::memcpy(L.wcol.data(), wcol, sizeof(int32_t) * n_f); }
  else { L.one_point = false; L.W = copy_csr(W); }
  bool any_c = false;
  for (int i = 0; i < 10; i++) { L.smooth_order[i] = smooth_order10 ? smooth_order10[i] : pc->opt.smooth_order[i]; if (L.smooth_order[i] < 0) any_c = true; }
  if (A_cf) L.A_cf = copy_csr(A_cf);
  if (any_c) {
    REQUIRE(A_cf && A_cc && inv_cc, AIRHIP_ERR_ARG, "C smoothing needs A_cf, A_cc and inv_cc");
    L.A_cc = copy_csr(A_cc);
    fill_inv(inv_cc, L.inv_cc, n_c);
    for (int i = 0; i < 10; i++) pc->H->opt.smooth_order[i] = L.smooth_order[i];   // any_c_smooths() of the hierarchy
  }
  L.nnz_A = (level == 0) ? pc->A_top.nnz() : 0;
  pc->H->levels[level] = std::move(Lp);
  API_END
}

int airhip_pcair_set_coarse(airhip_pcair pc, const airhip_csr* A_coarse, const airhip_invop* inv, int coarse_richardson_its) {
  API_BEGIN
  REQUIRE(pc && pc->byo && pc->H && A_coarse && inv, AIRHIP_ERR_STATE, "bad state / argument");
  REQUIRE_CSR(A_coarse, true);
  if (inv->kind == AIRHIP_OP_ASSEMBLED) {
    REQUIRE_CSR(&inv->mat, true);
    REQUIRE(inv->mat.m == A_coarse->m, AIRHIP_ERR_ARG, "coarse inverse has the wrong size");
  }
  pc->H->coarse_A = copy_csr(A_coarse);
  fill_inv(inv, pc->H->coarse_inv, A_coarse->m);
  pc->H->opt.coarse_richardson_its = coarse_richardson_its;
  API_END
}

int airhip_pcair_finalize(airhip_pcair pc) {
  API_BEGIN
  REQUIRE(pc && pc->byo && pc->H, AIRHIP_ERR_STATE, "nothing to finalize");
  for (auto& l : pc->H->levels) REQUIRE(l != nullptr, AIRHIP_ERR_STATE, "a level was not set");
  REQUIRE(!pc->H->coarse_A.empty(), AIRHIP_ERR_STATE, "coarse level was not set");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device: the PCAIR solve phase has no CPU fallback"; return AIRHIP_ERR_HIP;
  }
  finish_device(pc);
  API_END
}

int airhip_pcair_apply(airhip_pcair pc, const double* d_x, double* d_y) {
  API_BEGIN
  REQUIRE(pc && pc->dev, AIRHIP_ERR_STATE, "PCAIR is not set up on a device");
  REQUIRE(d_x && d_y, AIRHIP_ERR_ARG, "null vector");
  pc->link.after_caller(pc->dev->stream(), nullptr);      // inputs may come from the legacy default stream
  pc->dev->apply_natural(d_x, d_y);
  API_END
}
int airhip_pcair_apply_on_stream(airhip_pcair pc, const double* d_x, double* d_y, void* stream) {
  API_BEGIN
  REQUIRE(pc && pc->dev, AIRHIP_ERR_STATE, "PCAIR is not set up on a device");
  REQUIRE(d_x && d_y, AIRHIP_ERR_ARG, "null vector");
  hipStream_t caller = (hipStream_t)stream;
  pc->link.after_caller(pc->dev->stream(), caller);
  pc->dev->apply_natural(d_x, d_y);
  pc->link.before_caller(pc->dev->stream(), caller);
  API_END
}
int airhip_pcair_sync(airhip_pcair pc) {
  API_BEGIN
  REQUIRE(pc && pc->dev, AIRHIP_ERR_STATE, "PCAIR is not set up on a device");
  HIP_CHECK(hipStreamSynchronize(pc->dev->stream()));
  API_END
}
int airhip_pcair_destroy(airhip_pcair* pc) {
  API_BEGIN
  if (pc && *pc) {
    try { mirror_join(*pc); } catch (...) {}      // an error of the host mirror dies with the object
    delete *pc; *pc = nullptr;
  }
  API_END
}

int airhip_pcair_num_levels(airhip_pcair pc, int* no_levels) {
  API_BEGIN
  REQUIRE(pc && no_levels, AIRHIP_ERR_ARG, "null argument");
  if (pc->H) *no_levels = pc->H->no_levels();
  else if (pc->dev) *no_levels = pc->dev->no_levels();
  else { g_err = "not set up"; return AIRHIP_ERR_STATE; }
  API_END
}

int airhip_pcair_level_sizes(airhip_pcair pc, int level, int64_t* n, int64_t* n_f, int64_t* n_c, int* aff_diag,
                             int* one_point, int64_t* nnz_A) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && pc->H, AIRHIP_ERR_STATE, "host hierarchy not available");
  const int L = pc->H->no_levels();
  REQUIRE(level >= 0 && level < L, AIRHIP_ERR_ARG, "level out of range");
  if (level == L - 1) {
    if (n) *n = pc->H->coarse_A.m;
    if (n_f) *n_f = 0;
    if (n_c) *n_c = 0;
    if (aff_diag) *aff_diag = 0;
    if (one_point) *one_point = 0;
    if (nnz_A) *nnz_A = pc->H->coarse_A.nnz();
  } else {
    const Level& lv = *pc->H->levels[level];
    if (n) *n = lv.n;
    if (n_f) *n_f = lv.n_f;
    if (n_c) *n_c = lv.n_c;
    if (aff_diag) *aff_diag = lv.aff_diag;
    if (one_point) *one_point = lv.one_point;
    if (nnz_A) *nnz_A = lv.nnz_A;
  }
  API_END
}

static void expose_inv(const InvOp& in, airhip_invop* out) {
  std::memset(out, 0, sizeof(*out));
  out->kind = (int)in.kind; out->diag_scale = in.diag_scale; out->neumann = in.neumann;
  if (in.kind == InvOp::ASSEMBLED) expose_csr(in.mat, &out->mat);
  else if (in.kind == InvOp::DIAG) out->diag = in.d.data();
  else if (in.kind == InvOp::HORNER) { out->ncoef = (int)in.coeffs.size(); out->coef = in.coeffs.data(); }
  else if (in.kind == InvOp::NEWTON) { out->ncoef = (int)in.re.size(); out->coef = in.re.data(); out->coef_im = in.im.data(); }
}

int airhip_pcair_level_get(airhip_pcair pc, int level, const char* which, airhip_csr* csr, const int32_t** idx,
                           const double** vec, int64_t* len) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && pc->H && which, AIRHIP_ERR_STATE, "host hierarchy not available");
  REQUIRE(level >= 0 && level < pc->H->no_levels() - 1, AIRHIP_ERR_ARG, "level out of range");
  const Level& lv = *pc->H->levels[level];
  std::string w(which);
  auto give_csr = [&](const Csr& A) -> int { if (!csr || A.empty()) { g_err = "matrix not available: " + w; return AIRHIP_ERR_STATE; } expose_csr(A, csr); return 0; };
  auto give_idx = [&](const Arr<int32_t>& a) -> int { if (!idx || !len) { g_err = "null out"; return AIRHIP_ERR_ARG; } *idx = a.data(); *len = a.size(); return 0; };
  auto give_vec = [&](const Arr<double>& a) -> int { if (!vec || !len || !a.data()) { g_err = "vector not available: " + w; return AIRHIP_ERR_STATE; } *vec = a.data(); *len = a.size(); return 0; };
  int rc = 0;
  if (w == "F") rc = give_idx(lv.F);
  else if (w == "C") rc = give_idx(lv.C);
  else if (w == "wcol") rc = give_idx(lv.wcol);
  else if (w == "A_ff") rc = give_csr(lv.A_ff);
  else if (w == "A_fc") rc = give_csr(lv.A_fc);
  else if (w == "A_cf") rc = give_csr(lv.A_cf);
  else if (w == "A_cc") rc = give_csr(lv.A_cc);
  else if (w == "A_full") rc = give_csr(lv.A_full);     /* only with full up / down smoothing */
  else if (w == "Z") rc = give_csr(lv.Z);
  else if (w == "W") rc = give_csr(lv.W);
  else if (w == "M_ff") rc = give_csr(lv.inv_ff.mat);
  else if (w == "M_cc") rc = give_csr(lv.inv_cc.mat);
  else if (w == "aff_diag") rc = give_vec(lv.aff_d);
  else if (w == "M_ff_diag") rc = give_vec(lv.inv_ff.d);
  else if (w == "M_cc_diag") rc = give_vec(lv.inv_cc.d);
  else { g_err = "unknown item " + w; return AIRHIP_ERR_ARG; }
  if (rc) return rc;
  API_END
}

int airhip_pcair_level_inv(airhip_pcair pc, int level, int which_c, airhip_invop* inv) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && pc->H && inv, AIRHIP_ERR_STATE, "host hierarchy not available");
  const int L = pc->H->no_levels();
  REQUIRE(level >= 0 && level < L, AIRHIP_ERR_ARG, "level out of range");
  if (level == L - 1) expose_inv(pc->H->coarse_inv, inv);
  else expose_inv(which_c ? pc->H->levels[level]->inv_cc : pc->H->levels[level]->inv_ff, inv);
  API_END
}

int airhip_pcair_level_smooth_order(airhip_pcair pc, int level, int* so) {
  API_BEGIN
  REQUIRE(pc && pc->H && so, AIRHIP_ERR_STATE, "host hierarchy not available");
  REQUIRE(level >= 0 && level < pc->H->no_levels() - 1, AIRHIP_ERR_ARG, "level out of range");
  for (int i = 0; i < 10; i++) so[i] = pc->H->levels[level]->smooth_order[i];
  API_END
}

int airhip_pcair_coarse_get(airhip_pcair pc, airhip_csr* A) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && pc->H && A, AIRHIP_ERR_STATE, "host hierarchy not available");
  expose_csr(pc->H->coarse_A, A);
  API_END
}

int airhip_pcair_complexities(airhip_pcair pc, double* grid, double* op, double* cycle, double* storage, double* cyc_bytes) {
  API_BEGIN
  REQUIRE(pc && pc->H, AIRHIP_ERR_STATE, "host hierarchy not available");
  if (pc->cplx_valid) {      // taken at the end of the setup (the host mirror may still be filling the operators in)
    if (grid) *grid = pc->cplx[0];
    if (op) *op = pc->cplx[1];
    if (cycle) *cycle = pc->cplx[2];
    if (storage) *storage = pc->cplx[3];
    if (cyc_bytes) *cyc_bytes = pc->dev ? pc->dev->cycle_alg_bytes() : 0.0;
    return AIRHIP_OK;
  }
  const Hierarchy& H = *pc->H;
  // src/AIR_MG_Stats.F90:79-382
  double rows = 0, opn = 0, stor = 0, cyc = 0;
  auto inv_nnz = [](const InvOp& I, double n) { return I.kind == InvOp::ASSEMBLED ? (double)I.mat.nnz() : (I.kind == InvOp::DIAG ? n : 0.0); };
  for (auto& lp : H.levels) {
    const Level& l = *lp;
    rows += l.n; opn += l.nnz_A;
    double aff = l.aff_diag ? (double)l.n_f : (double)l.A_ff.nnz();
    double nnzW = 0;
    if (l.one_point) { for (int64_t i = 0; i < l.n_f; i++) nnzW += l.wcol[i] >= 0 ? 1 : 0; } else nnzW = (double)l.W.nnz();
    double Rn = (double)l.Z.nnz() + l.n_c, Pn = nnzW + l.n_c;
    if (l.full) {
      // full smoothing (AIR_MG_Stats.F90:315-330): level operator + its inverse stored; one sweep down (M b, then the
      // residual) and one up (residual, M r)
      const double an = (double)l.A_full.nnz();
      const bool mfl = l.inv_ff.kind == InvOp::HORNER || l.inv_ff.kind == InvOp::NEWTON;
      const double iw = mfl ? (l.inv_ff.kind == InvOp::HORNER ? (double)l.inv_ff.coeffs.size() : (double)l.inv_ff.spmvs + 1) * an
                            : inv_nnz(l.inv_ff, (double)l.n);
      stor += an + inv_nnz(l.inv_ff, (double)l.n) + Rn + Pn;
      cyc += 2.0 * (iw + an) + Rn + Pn;
      continue;
    }
    stor += aff + l.A_fc.nnz() + inv_nnz(l.inv_ff, l.n_f) + Rn + Pn;
    int g = 0, sf = 0, gc = 0, sc = 0;
    for (int s : l.smooth_order) { if (s > 0) { g++; sf += s; } else if (s < 0) { gc++; sc += -s; } }
    bool mf = l.inv_ff.kind == InvOp::HORNER || l.inv_ff.kind == InvOp::NEWTON;
    double invw = mf ? (l.inv_ff.kind == InvOp::HORNER ? (double)l.inv_ff.coeffs.size() : (double)l.inv_ff.spmvs + 1) * aff : inv_nnz(l.inv_ff, l.n_f);
    cyc += sf * (invw + aff) + g * (double)l.A_fc.nnz() + Rn + Pn;
    if (gc) {
      stor += l.A_cc.nnz() + inv_nnz(l.inv_cc, l.n_c) + l.A_cf.nnz();
      bool mfc = l.inv_cc.kind == InvOp::HORNER || l.inv_cc.kind == InvOp::NEWTON;
      double invc = mfc ? (double)l.inv_cc.coeffs.size() * l.A_cc.nnz() : inv_nnz(l.inv_cc, l.n_c);
      cyc += sc * (invc + (double)l.A_cc.nnz()) + gc * (double)l.A_cf.nnz();
    }
  }
  const double top_nnz = H.levels.empty() ? (double)H.coarse_A.nnz() : (double)H.levels[0]->nnz_A;
  const double top_rows = H.levels.empty() ? (double)H.coarse_A.m : (double)H.levels[0]->n;
  if (!H.levels.empty()) {
    rows += H.coarse_A.m;
    bool mf = H.coarse_inv.kind == InvOp::HORNER || H.coarse_inv.kind == InvOp::NEWTON;
    double cn = (double)H.coarse_A.nnz();
    opn += cn;
    if (mf) { stor += cn; cyc += (H.coarse_inv.kind == InvOp::HORNER ? (double)H.coarse_inv.coeffs.size() : (double)H.coarse_inv.spmvs + 1) * cn; }
    else { stor += inv_nnz(H.coarse_inv, H.coarse_A.m); cyc += inv_nnz(H.coarse_inv, H.coarse_A.m); if (H.opt.coarse_richardson_its > 1) stor += cn; }
  } else { rows = top_rows; opn = top_nnz; }
  if (grid) *grid = rows / top_rows;
  if (op) *op = opn / top_nnz;
  if (cycle) *cycle = cyc / top_nnz;
  if (storage) *storage = stor / top_nnz;
  if (cyc_bytes) *cyc_bytes = pc->dev ? pc->dev->cycle_alg_bytes() : 0.0;
  API_END
}

int airhip_pcair_set_ordering_hint(airhip_pcair pc, const int64_t* key, int64_t n) {
  API_BEGIN
  REQUIRE(pc && (key || n == 0) && n >= 0, AIRHIP_ERR_ARG, "bad ordering hint");
  pc->hint.assign(key, key + n);
  API_END
}
int airhip_pcair_cycle_alg_bytes(airhip_pcair pc, double* bytes) {
  API_BEGIN
  REQUIRE(pc && pc->dev && bytes, AIRHIP_ERR_STATE, "not set up on a device");
  *bytes = pc->dev->cycle_alg_bytes();
  API_END
}
int airhip_pcair_setup_times(airhip_pcair pc, double* host_setup_s, double* upload_s) {
  API_BEGIN
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  if (host_setup_s) *host_setup_s = pc->host_setup_s;
  if (upload_s) *upload_s = pc->upload_s;
  API_END
}
int airhip_pcair_host_mirror_wait(airhip_pcair pc, double* seconds) {
  API_BEGIN
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  mirror_join(pc);
  if (seconds) *seconds = pc->mirror_s;
  API_END
}
int airhip_pcair_release_host(airhip_pcair pc) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  pc->H.reset(); pc->A_top = Csr();
  API_END
}
int airhip_pcair_use_graph(airhip_pcair pc, int on) {
  API_BEGIN
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  pc->use_graph = on != 0;
  if (pc->dev) pc->dev->set_use_graph(pc->use_graph);
  API_END
}
int airhip_pcair_profile(airhip_pcair pc, int on) {
  API_BEGIN
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  pc->profile = on != 0;
  if (pc->dev) { if (!on) pc->dev->resolve_events(); pc->dev->set_profile(pc->profile); }
  API_END
}
int airhip_pcair_profile_reset(airhip_pcair pc) {
  API_BEGIN
  REQUIRE(pc && pc->dev, AIRHIP_ERR_STATE, "not set up");
  pc->dev->reset_stats();
  API_END
}
int airhip_pcair_profile_count(airhip_pcair pc, int* n) {
  API_BEGIN
  REQUIRE(pc && pc->dev && n, AIRHIP_ERR_STATE, "not set up");
  pc->dev->resolve_events();
  *n = (int)pc->dev->stats().size();
  API_END
}
int airhip_pcair_profile_get(airhip_pcair pc, int i, const char** name, int* level, double* alg_bytes, int64_t* calls, double* total_ms) {
  API_BEGIN
  REQUIRE(pc && pc->dev, AIRHIP_ERR_STATE, "not set up");
  const auto& st = pc->dev->stats();
  REQUIRE(i >= 0 && i < (int)st.size(), AIRHIP_ERR_ARG, "index out of range");
  if (name) *name = st[i].name.c_str();
  if (level) *level = st[i].level;
  if (alg_bytes) *alg_bytes = st[i].alg_bytes;
  if (calls) *calls = st[i].calls;
  if (total_ms) *total_ms = st[i].ms;
  API_END
}

int airhip_pcair_profile_kernel(airhip_pcair pc, int i, const char** kernel) {
  API_BEGIN
  REQUIRE(pc && pc->dev && kernel, AIRHIP_ERR_STATE, "not set up");
  const auto& st = pc->dev->stats();
  REQUIRE(i >= 0 && i < (int)st.size(), AIRHIP_ERR_ARG, "index out of range");
  *kernel = st[i].kernel;
  API_END
}

// ------------------------------------------------------------------------ distribution
struct airhip_archive_s { Archive ar; };
struct airhip_partition_s { std::unique_ptr<Partitioner> p; Csr A_view; };
struct airhip_comm_s { std::unique_ptr<Comm> c; };

int airhip_partition_create(airhip_pcair pc, const airhip_csr* A_top, int nranks, const int32_t* owner, airhip_partition* out) {
  API_BEGIN
  mirror_join(pc);
  REQUIRE(pc && pc->H && A_top && owner && out && nranks >= 1, AIRHIP_ERR_ARG, "partitioning needs a PCAIR with its host hierarchy");
  REQUIRE_CSR(A_top, true);
  auto* h = new airhip_partition_s;
  view_csr(A_top, h->A_view);
  REQUIRE(pc->hint.empty() || (int64_t)pc->hint.size() == A_top->m, AIRHIP_ERR_ARG, "ordering hint: one key per row of the operator is required");
  // the operators of the fused first F sweep (one device builds them at upload time from its own copies of A_fc / A_ff;
  // a cut hierarchy gets them here, cut like every other operand)
  if (fuse_first_sweep_mode() != 0) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); ndev = 0; }
    for (auto& lv : pc->H->levels) {
      if (!(lv->G.empty() && lv->fuses_first_sweep() && lv->A_fc.on_host() && lv->A_ff.on_host())) continue;
      // on this rank's device when there is one (the same bits as the host code, in a fraction of its time: the cut of a
      // 464^3 hierarchy builds 2.9e9 non-zeros of G), on the host otherwise
      bool done = false;
      if (ndev > 0 && lv->A_fc.nnz() + lv->A_ff.nnz() >= 2000000) {
        try {
          DevScope scope(true);
          DCsr da = dcsr_upload(lv->A_ff.m, lv->A_ff.n, lv->A_ff.rp.data(), lv->A_ff.ci.data(), lv->A_ff.v.data());
          DCsr db = dcsr_upload(lv->A_fc.m, lv->A_fc.n, lv->A_fc.rp.data(), lv->A_fc.ci.data(), lv->A_fc.v.data());
          int32_t* dw = (int32_t*)dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(lv->n_f, 1));
          staged_h2d(dw, lv->wcol.data(), sizeof(int32_t) * (size_t)lv->n_f);
          DCsr dg = dev_fused_first_operator(da, db, dw, nullptr);
          lv->G = Csr(); lv->G.m = dg.m; lv->G.n = dg.n;
          lv->G.rp.alloc(dg.m + 1); lv->G.ci.alloc(dg.nnz); lv->G.v.alloc(dg.nnz);
          dcsr_download(dg, lv->G.rp.data(), lv->G.ci.data(), lv->G.v.data());
          dcsr_free(da); dcsr_free(db); dcsr_free(dg); dev_free(dw);
          done = true;
        } catch (const std::exception&) { (void)hipGetLastError(); lv->G = Csr(); }      // (no room on the device: the host code)
      }
      if (!done) lv->G = fused_first_operator(lv->A_ff, lv->A_fc, lv->wcol.data());
    }
  }
  try { h->p = std::make_unique<Partitioner>(*pc->H, h->A_view, nranks, owner, pc->hint.empty() ? nullptr : pc->hint.data()); }
  catch (...) { delete h; throw; }
  *out = h;
  API_END
}
int airhip_partition_extract(airhip_partition p, int rank, airhip_archive* out) {
  API_BEGIN
  REQUIRE(p && out, AIRHIP_ERR_ARG, "null argument");
  auto* a = new airhip_archive_s;
  try { p->p->extract(rank, a->ar); } catch (...) { delete a; throw; }
  *out = a;
  API_END
}
int airhip_partition_destroy(airhip_partition* p) {
  API_BEGIN
  if (p && *p) { delete *p; *p = nullptr; }
  API_END
}
int airhip_archive_create(airhip_archive* out) {
  API_BEGIN
  REQUIRE(out, AIRHIP_ERR_ARG, "null argument");
  *out = new airhip_archive_s;
  API_END
}
int airhip_archive_add(airhip_archive ar, const char* name, int dtype, const void* data, int64_t count) {
  API_BEGIN
  REQUIRE(ar && name && count >= 0 && dtype >= 0 && dtype <= 2, AIRHIP_ERR_ARG, "bad archive item");
  ar->ar.add_view(name, dtype, data, count);
  API_END
}
int airhip_archive_count(airhip_archive ar, int* n) {
  API_BEGIN
  REQUIRE(ar && n, AIRHIP_ERR_ARG, "null argument");
  *n = (int)ar->ar.items.size();
  API_END
}
int airhip_archive_item(airhip_archive ar, int i, const char** name, int* dtype, const void** data, int64_t* count) {
  API_BEGIN
  REQUIRE(ar && i >= 0 && i < (int)ar->ar.items.size(), AIRHIP_ERR_ARG, "index out of range");
  const auto& it = ar->ar.items[i];
  if (name) *name = it.name.c_str();
  if (dtype) *dtype = it.dtype;
  if (data) *data = it.ptr;
  if (count) *count = it.count;
  API_END
}
int airhip_archive_destroy(airhip_archive* ar) {
  API_BEGIN
  if (ar && *ar) { delete *ar; *ar = nullptr; }
  API_END
}

int airhip_comm_rccl_unique_id(void* out128) {
  API_BEGIN
  REQUIRE(out128, AIRHIP_ERR_ARG, "null argument");
  rccl_unique_id(out128);
  API_END
}
int airhip_comm_create_rccl(int rank, int size, const void* unique_id128, airhip_comm* out) {
  API_BEGIN
  REQUIRE(out && unique_id128 && size >= 1 && rank >= 0 && rank < size, AIRHIP_ERR_ARG, "bad communicator arguments");
  auto* c = new airhip_comm_s;
  try { c->c = make_rccl_comm(rank, size, unique_id128); } catch (...) { delete c; throw; }
  *out = c;
  API_END
}
int airhip_comm_create_staged(int rank, int size, airhip_exchange_fn ex, airhip_allreduce_fn ar, void* user, airhip_comm* out) {
  API_BEGIN
  REQUIRE(out && ex && ar && size >= 1 && rank >= 0 && rank < size, AIRHIP_ERR_ARG, "bad communicator arguments");
  auto* c = new airhip_comm_s;
  c->c = make_staged_comm(rank, size, (exchange_cb)ex, (allreduce_cb)ar, user);
  *out = c;
  API_END
}
int airhip_comm_stats(airhip_comm c, int64_t* exchanges, int64_t* allreduces) {
  API_BEGIN
  REQUIRE(c, AIRHIP_ERR_ARG, "null argument");
  if (exchanges) *exchanges = c->c->n_exchanges;
  if (allreduces) *allreduces = c->c->n_allreduces;
  API_END
}
int airhip_comm_traffic(airhip_comm c, int64_t* bytes_sent, int64_t* messages_sent) {
  API_BEGIN
  REQUIRE(c, AIRHIP_ERR_ARG, "null argument");
  if (bytes_sent) *bytes_sent = c->c->bytes_sent;
  if (messages_sent) *messages_sent = c->c->messages_sent;
  API_END
}
int airhip_comm_selftest(airhip_comm c, int64_t count, int* ok) {
  API_BEGIN
  REQUIRE(c && ok, AIRHIP_ERR_ARG, "null argument");
  *ok = 0;
  *ok = comm_selftest(*c->c, count) ? 1 : 0;
  API_END
}
int airhip_comm_destroy(airhip_comm* c) {
  API_BEGIN
  if (c && *c) { delete *c; *c = nullptr; }
  API_END
}

// This rank's piece -> device.  The archive's arrays are only read during the call.
int airhip_pcair_load_distributed(airhip_pcair pc, airhip_archive piece, airhip_comm comm) {
  API_BEGIN
  mirror_join(pc);
  DevScope dev_scope(true);
  REQUIRE(pc && piece && comm, AIRHIP_ERR_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device: the PCAIR solve phase has no CPU fallback"; return AIRHIP_ERR_HIP;
  }
  pc->dev.reset(); pc->H.reset(); pc->ksp.reset(); pc->kb.release(); pc->kx.release();
  auto loc = std::make_unique<LocalHierarchy>();
  load_local(piece->ar, *loc);
  REQUIRE(loc->nranks == comm->c->size && loc->rank == comm->c->rank, AIRHIP_ERR_ARG, "piece does not belong to this rank");
  double t0 = now_s();
  pc->dev = std::make_unique<DevPCAIR>();
  if (loc->hint.size() > 0) pc->dev->set_ordering_hint(loc->hint.data(), loc->hint.size());
  pc->dev->upload(loc->H, loc->A_top, loc.get(), comm->c.get());
  pc->dev->set_use_graph(pc->use_graph);      // effective only with a transport whose exchange is device work (DevPCAIR::graph_ok)
  pc->dev->set_before_capture([pc]() { mirror_join(pc); });
  pc->dev->set_profile(pc->profile);
  pc->upload_s = now_s() - t0;
  pc->op_nnz_top = (double)loc->A_top.nnz();
  pc->byo = false;
  API_END
}

// ------------------------------------------------------------------------ PCPFLAREINV
// A PCPFLAREINV is a PCAIR hierarchy with no levels whose "coarse solver" is the
// approximate inverse of the whole operator (PCApply_PFLAREINV_c = one MatMult with it).
struct airhip_pcinv_s {
  int type = INV_ARNOLDI, poly_order = 6, sparsity_order = 1;
  bool matrix_free = false, diag_scale = false;
  uint64_t seed = 1;
  airhip_pcair_s inner;
};

int airhip_pcinv_create(airhip_pcinv* pc) {
  API_BEGIN
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  *pc = new airhip_pcinv_s;
  API_END
}
int airhip_pcinv_set_option(airhip_pcinv pc, const char* name, const char* value) {
  API_BEGIN
  REQUIRE(pc && name, AIRHIP_ERR_ARG, "null argument");
  std::string k(name);
  if (k.rfind("-pc_pflareinv_", 0) == 0) k = k.substr(14);
  else if (k.rfind("pc_pflareinv_", 0) == 0) k = k.substr(13);
  const char* v = value ? value : "";
  if (k == "type") {
    int t = parse_inv_type(v);
    if (t < 0) { g_err = std::string("unknown inverse type ") + v; return AIRHIP_ERR_ARG; }
    pc->type = t;
  } else if (k == "poly_order" || k == "order") pc->poly_order = std::atoi(v);
  else if (k == "sparsity_order") pc->sparsity_order = std::atoi(v);
  else if (k == "matrix_free") pc->matrix_free = parse_bool(v);
  else if (k == "diag_scale") pc->diag_scale = parse_bool(v);
  else if (k == "seed") pc->seed = (uint64_t)std::atoll(v);
  else if (k == "setup_on_device") pc->inner.setup_on_device = parse_bool(v);
  else if (k == "setup_device_min_nnz") pc->inner.setup_device_min_nnz = std::atoll(v);
  else { g_err = "unknown option -pc_pflareinv_" + k; return AIRHIP_ERR_ARG; }
  API_END
}

static void pcinv_build_host(airhip_pcinv pc, const Csr& A, bool device_assist = false) {
  // with a device: the fixed-sparsity polynomial assembly may run there (bit-identical; device/polyasm.hip)
  std::unique_ptr<SetupAccel> accel;
  if (device_assist && pc->inner.setup_on_device) accel = make_device_setup_accel(nullptr, pc->inner.setup_device_min_nnz);
  struct Scope { SetupAccel* prev; explicit Scope(SetupAccel* a) : prev(g_setup_accel) { g_setup_accel = a; } ~Scope() { g_setup_accel = prev; } } scope(accel.get());
  auto H = std::make_unique<Hierarchy>();
  OpView v; v.A = &A; v.n = A.m;
  build_inverse(v, pc->type, pc->poly_order, pc->sparsity_order, pc->matrix_free, pc->diag_scale, pc->seed, H->coarse_inv, nullptr);
  H->coarse_A = A.clone();
  pc->inner.H = std::move(H);
}

int airhip_pcinv_setup(airhip_pcinv pc, const airhip_csr* A) {
  API_BEGIN
  DevScope dev_scope(true);
  REQUIRE(pc && A && A->m == A->n && A->m > 0, AIRHIP_ERR_ARG, "PCPFLAREINV needs a square CSR operator");
  REQUIRE_CSR(A, true);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    g_err = "no HIP device: the PCPFLAREINV apply has no CPU fallback"; return AIRHIP_ERR_HIP;
  }
  Csr view; view_csr(A, view);
  ThreadScope threads(A->rowptr[A->m]);
  pcinv_build_host(pc, view, true);
  view_csr(A, pc->inner.A_top);
  finish_device(&pc->inner);
  dev_cache_trim();
  pc->inner.A_top = Csr();
  API_END
}
int airhip_pcinv_setup_host_only(airhip_pcinv pc, const airhip_csr* A) {
  API_BEGIN
  REQUIRE(pc && A && A->m == A->n && A->m > 0, AIRHIP_ERR_ARG, "PCPFLAREINV needs a square CSR operator");
  REQUIRE_CSR(A, true);
  Csr view; view_csr(A, view);
  ThreadScope threads(A->rowptr[A->m]);
  pcinv_build_host(pc, view);
  API_END
}
int airhip_pcinv_apply(airhip_pcinv pc, const double* d_x, double* d_y) {
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  return airhip_pcair_apply(&pc->inner, d_x, d_y);
}
// the zero-level PCAIR object a PCPFLAREINV is built on (borrowed: it lives and dies with pc)
int airhip_pcinv_as_pcair(airhip_pcinv pc, airhip_pcair* out) {
  API_BEGIN
  REQUIRE(pc && out, AIRHIP_ERR_ARG, "null argument");
  *out = &pc->inner;
  API_END
}
int airhip_pcinv_apply_on_stream(airhip_pcinv pc, const double* d_x, double* d_y, void* stream) {
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  return airhip_pcair_apply_on_stream(&pc->inner, d_x, d_y, stream);
}
int airhip_pcinv_sync(airhip_pcinv pc) {
  REQUIRE(pc, AIRHIP_ERR_ARG, "null argument");
  return airhip_pcair_sync(&pc->inner);
}
int airhip_pcinv_get_inv(airhip_pcinv pc, airhip_invop* inv) {
  API_BEGIN
  REQUIRE(pc && pc->inner.H && inv, AIRHIP_ERR_STATE, "not set up");
  expose_inv(pc->inner.H->coarse_inv, inv);
  API_END
}
int airhip_pcinv_destroy(airhip_pcinv* pc) {
  API_BEGIN
  if (pc && *pc) { delete *pc; *pc = nullptr; }
  API_END
}

// ------------------------------------------------------------------- device matrices
struct airhip_mat_s { DevMat A; hipStream_t stream = nullptr; int64_t n = 0; std::unique_ptr<DevKSP> ksp; StreamLink link; };

int airhip_mat_create(const airhip_csr* A, airhip_mat* out) {
  API_BEGIN
  DevScope dev_scope(true);
  REQUIRE(A && out, AIRHIP_ERR_ARG, "null argument");
  REQUIRE_CSR(A, false);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_err = "no HIP device"; return AIRHIP_ERR_HIP; }
  auto* m = new airhip_mat_s;
  HIP_CHECK(hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking));
  Csr view; view_csr(A, view);
  upload_mat(view, nullptr, nullptr, m->A, m->stream);
  dev_cache_trim();
  m->n = A->m;
  *out = m;
  API_END
}
int airhip_mat_mult(airhip_mat A, const double* d_x, double* d_y) {
  API_BEGIN
  REQUIRE(A && d_x && d_y, AIRHIP_ERR_ARG, "null argument");
  A->link.after_caller(A->stream, nullptr);
  SpmvEpi e; e.out = d_y;
  launch_spmv(A->A, d_x, e, A->stream);
  HIP_CHECK(hipStreamSynchronize(A->stream));
  API_END
}
int airhip_mat_mult_on_stream(airhip_mat A, const double* d_x, double* d_y, void* stream) {
  API_BEGIN
  REQUIRE(A && d_x && d_y, AIRHIP_ERR_ARG, "null argument");
  hipStream_t caller = (hipStream_t)stream;
  A->link.after_caller(A->stream, caller);
  SpmvEpi e; e.out = d_y;
  launch_spmv(A->A, d_x, e, A->stream);
  A->link.before_caller(A->stream, caller);
  API_END
}
int airhip_mat_blocks(airhip_mat A, int64_t* nblocks, int64_t* idx_bytes, int64_t* nnz, void* desc, void* idx, double* val, void* roff) {
  API_BEGIN
  REQUIRE(A && nblocks && idx_bytes && nnz, AIRHIP_ERR_ARG, "null argument");
  const DevMat& M = A->A;
  *nblocks = M.nblocks; *idx_bytes = M.idx_bytes; *nnz = M.val_len;      // (doubles of the value array: nnz unless blocks hold value dictionaries)
  if (desc) {
    REQUIRE(idx && val && roff, AIRHIP_ERR_ARG, "null argument");
    HIP_CHECK(hipStreamSynchronize(A->stream));
    if (M.nblocks) HIP_CHECK(hipMemcpy(desc, M.blk, sizeof(BlkDesc) * (size_t)M.nblocks, hipMemcpyDeviceToHost));
    if (M.idx_bytes) HIP_CHECK(hipMemcpy(idx, M.idx, (size_t)M.idx_bytes, hipMemcpyDeviceToHost));
    if (M.val_len) HIP_CHECK(hipMemcpy(val, M.val, sizeof(double) * (size_t)M.val_len, hipMemcpyDeviceToHost));
    if (M.m) HIP_CHECK(hipMemcpy(roff, M.roff, sizeof(uint16_t) * (size_t)M.m, hipMemcpyDeviceToHost));
  }
  API_END
}
int airhip_mat_destroy(airhip_mat* A) {
  API_BEGIN
  if (A && *A) { devmat_free((*A)->A); if ((*A)->stream) (void)hipStreamDestroy((*A)->stream); delete *A; *A = nullptr; }
  API_END
}

// ------------------------------------------------------------------------------- KSP
void airhip_ksp_default_options(airhip_ksp_options* o) {
  if (!o) return;
  o->type = 1; o->pc_side = 0; o->norm_type = 0; o->rtol = 1e-5; o->atol = 1e-50; o->dtol = 1e4; o->max_it = 10000; o->restart = 30;
}

int airhip_ksp_solve(const airhip_ksp_options* o, airhip_mat A, airhip_pcair pcair, airhip_pcinv pcinv,
                     const double* d_b, double* d_x, airhip_ksp_result* res, double* history, int history_cap,
                     int* history_len) {
  return airhip_ksp_solve_on_stream(o, A, pcair, pcinv, d_b, d_x, res, history, history_cap, history_len, nullptr);
}
int airhip_ksp_solve_on_stream(const airhip_ksp_options* o, airhip_mat A, airhip_pcair pcair, airhip_pcinv pcinv,
                               const double* d_b, double* d_x, airhip_ksp_result* res, double* history, int history_cap,
                               int* history_len, void* stream) {
  API_BEGIN
  hipStream_t caller = (hipStream_t)stream;
  REQUIRE(o && d_b && d_x && res, AIRHIP_ERR_ARG, "null argument");
  REQUIRE(!(pcair && pcinv), AIRHIP_ERR_ARG, "give at most one preconditioner");
  KspOptions ko; ko.type = o->type; ko.pc_side = o->pc_side; ko.norm_type = o->norm_type; ko.rtol = o->rtol;
  ko.atol = o->atol; ko.dtol = o->dtol; ko.max_it = o->max_it; ko.restart = o->restart;
  REQUIRE(ko.restart >= 1 && ko.restart <= 60, AIRHIP_ERR_ARG, "restart must be in [1,60]");
  KspResult r;
  airhip_pcair_s* pc = pcair ? pcair : (pcinv ? &pcinv->inner : nullptr);
  if (pc) {
    REQUIRE(pc->dev, AIRHIP_ERR_STATE, "preconditioner is not set up on a device");
    DevPCAIR& dev = *pc->dev;
    const int64_t n = dev.n();
    if (!pc->ksp) { pc->ksp = std::make_unique<DevKSP>(n); pc->kb.alloc(n); pc->kx.alloc(n); }
    pc->link.after_caller(dev.stream(), caller);
    dev.prof_begin("Krylov to hierarchical order", -1, 40.0 * n, "k_gather");
    dev.to_permuted(d_b, pc->kb.p);
    dev.to_permuted(d_x, pc->kx.p);
    dev.prof_end();
    KspOperator op; op.A = &dev.A_perm(); op.stream = dev.stream(); op.comm = dev.comm();
    op.prof_begin = [&dev](const char* name, double bytes, const char* kernel) { dev.prof_begin(name, -1, bytes, kernel); };
    op.prof_end = [&dev]() { dev.prof_end(); };
    op.pc = [&dev](const double* in, double* out) { dev.apply_permuted(in, out); };
    op.mult = [&dev](const double* x, const SpmvEpi& e) { dev.spmv(dev.A_perm(), x, e, "outer A*x (Krylov)", -1); };
    r = pc->ksp->solve(op, ko, pc->kb.p, pc->kx.p);
    dev.prof_begin("Krylov to natural order", -1, 20.0 * n, "k_gather");
    dev.to_natural(pc->kx.p, d_x);
    dev.prof_end();
    HIP_CHECK(hipStreamSynchronize(dev.stream()));
  } else {
    REQUIRE(A, AIRHIP_ERR_ARG, "an operator is required without a preconditioner");
    if (!A->ksp) A->ksp = std::make_unique<DevKSP>(A->n);
    A->link.after_caller(A->stream, caller);
    KspOperator op; op.A = &A->A; op.stream = A->stream;
    r = A->ksp->solve(op, ko, d_b, d_x);
    HIP_CHECK(hipStreamSynchronize(A->stream));
  }
  res->its = r.its; res->reason = r.reason; res->rnorm = r.rnorm; res->solve_ms = r.solve_ms; res->pc_applies = r.pc_applies; res->reserved = 0;
  int nh = (int)r.history.size();
  if (history && history_cap > 0) { int c = std::min(nh, history_cap); std::memcpy(history, r.history.data(), sizeof(double) * c); }
  if (history_len) *history_len = nh;
  API_END
}
