// Device-resident PCAIR hierarchy, the V-cycle driver and the outer Krylov solvers.
#pragma once
#include "dev.hpp"
#include "comm.hpp"
#include "staged_copy.hpp"
#include "devmem.hpp"
#include "../host/air.hpp"
#include "../host/dist.hpp"
#include <map>
#include <functional>
#include <memory>

namespace airhip {

// Device buffer with ownership
template <class T>
struct DBuf {
  T* p = nullptr; int64_t n = 0;
  DBuf() = default;
  DBuf(const DBuf&) = delete; DBuf& operator=(const DBuf&) = delete;
  DBuf(DBuf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
  DBuf& operator=(DBuf&& o) noexcept { if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; } return *this; }
  ~DBuf() { release(); }
  void release() { dev_free_exact(p); p = nullptr; n = 0; }
  void alloc(int64_t n_) { release(); n = n_; p = (T*)dev_alloc_exact(sizeof(T) * (size_t)(n_ > 0 ? n_ : 1)); }
  void upload(const T* h, int64_t n_) { alloc(n_); if (n_ > 0) staged_h2d(p, h, sizeof(T) * (size_t)n_); }
};

// test hook: operators with at least this many row blocks get a spatial processing order (DevMat::desc_ord); default 4096
void set_block_order_min(int64_t nblocks);
// diagnostic (airhip_set_tuning "fc_fuse"): 0 runs the first stage of an F relaxation as a gather (one-point prolongation)
// followed by a plain A_fc product instead of the fused k_rb_fc_first; same results, one more pass over the F vectors
void set_fc_fuse(bool on);
void set_profile_chain(bool on);      // pcair_dev.cpp: begin_kernel
void set_perm_chunks(bool on);        // pcair_dev.cpp: to_permuted / to_natural
// (airhip_set_tuning "fuse_first_sweep", acts at the next upload) 1 (default): on the levels whose relaxation starts with F
// sweeps, with a one-point prolongator, a sparse A_ff and an assembled A_ff inverse, the first sweep runs on
// G = A_fc + A_ff W: four products per level instead of five.  0: the reference's sequence of products.  2: as 1, and A_fc
// stays on the device everywhere (lets "fuse_first_sweep_run" switch between the two sequences on one hierarchy).
void set_fuse_first_sweep(int mode);
int fuse_first_sweep_mode();
void set_fuse_first_sweep_run(bool on);
// (airhip_set_tuning "fc_spatial_x", acts at the next upload) A_fc x_c gathers from a copy of x_c in which the level's C points
// lie in the order of their locality keys instead of section by section [F_{l+1} | F_{l+2} | ...]: the copy costs 20 bytes per
// C point and cycle, the blocks of A_fc then touch half as many lines of x.  Same results bit for bit (the order of the
// entries inside a row does not change).  Single device, one-point or general W; not for distributed hierarchies.
void set_fc_spatial(int mode);      // 0 off, 1 (default) levels where A_fc holds >= 25 non-zeros per C point, 2 every level

// Upload a host CSR and convert it to SELL-64 on the device (rows/cols optionally permuted).
void upload_mat(const Csr& A, const int32_t* d_rowsrc, const int32_t* d_colmap, DevMat& out, hipStream_t s, int64_t n_ghost = 0);

struct DevInv {
  InvOp::Kind kind = InvOp::NONE;
  int64_t n = 0;
  DevMat mat;                   // ASSEMBLED
  DBuf<double> d;                // DIAG
  std::vector<double> coeffs, re, im;
  bool diag_scale = false, neumann = false;
  int spmvs = 0;
  const DevMat* Aop = nullptr;  // operator of the matrix-free polynomial (not owned)
  DBuf<double> dinv;             // 1/diag(Aop) for the scaled variants
  DBuf<double> t1, t2, t3, rhs;  // matrix-free temporaries
  ~DevInv() { devmat_free(mat); }
  // algorithmic bytes of one application (T(M) of SURVEY 8d)
  double alg_bytes() const;
};

struct DevLevel {
  int64_t n = 0, n_f = 0, n_c = 0;
  bool aff_diag = false;
  DBuf<double> aff_d;
  DevMat A_ff, A_fc, A_cf, A_cc, Z, W;
  // G = A_fc + A_ff W (one-point W): the operator of the fused first F sweep, r0 = b_f - G x_c (pcair_dev.cpp: f_smooths,
  // host/spops.hpp: fused_first_operator).  Valid on the levels where the sweep is fused; A_fc is then only on the device
  // when a later smoothing group needs it (or the tuning key asks to keep it).
  DevMat G;
  bool one_point = true;
  DBuf<int32_t> wcol;
  DevInv inv_ff, inv_cc;
  int smooth_order[10] = {0};
  DBuf<double> t4, r, t4c, rc;
  // x_c in the spatial order of the level's C points (set_fc_spatial): A_fc and the one-point prolongator gather from this
  // copy; xs_src[k] = position in x_c of the point with spatial rank k
  DBuf<double> xs; DBuf<int32_t> xs_src;
  int64_t nnz_A = 0;
  // full up / down smoothing: the level operator in the level's hierarchical ordering (inv_ff is then n x n) and the
  // residual of the level, whose C part is the next level's right-hand side
  DevMat A_full;
  DBuf<double> rfull;
  ~DevLevel() { devmat_free(A_ff); devmat_free(A_fc); devmat_free(A_cf); devmat_free(A_cc); devmat_free(Z); devmat_free(W); devmat_free(A_full); devmat_free(G); }
};

// one record per call site; `kernel` names the HIP kernel function the site launches (what a
// rocprofv3 kernel trace aggregates by)
struct KernelStat { std::string name; int level; double alg_bytes; int64_t calls = 0; double ms = 0.0; const char* kernel = ""; };

struct PendingEvent { hipEvent_t a, b; int stat; bool a_shared; };      // a_shared: `a` is the entry before's `b`

class DevPCAIR {
 public:
  DevPCAIR();
  ~DevPCAIR();
  // Build the device hierarchy from a host hierarchy; A_top is the fine operator (natural order).
  // dist / comm: this rank's piece of a row-block distributed hierarchy and its transport.
  void upload(const Hierarchy& H, const Csr& A_top, const LocalHierarchy* dist = nullptr, Comm* comm = nullptr);
  Comm* comm() const { return comm_; }
  // Locality keys of the fine rows (e.g. a brick / Morton index of the grid point): rows with close
  // keys are stored close together on every level.  Affects the device layout only; call before upload.
  void set_ordering_hint(const int64_t* key, int64_t n) { hint_.assign(key, key + n); }
  // the accelerator that ran the setup on this device: upload() converts the operators it kept (SetupAccel::device_copy)
  // where they are instead of copying them from the host again.  Only read during upload().
  void set_device_source(SetupAccel* a) { source_ = a; }
  int64_t n() const { return n_; }
  int no_levels() const { return (int)levels_.size() + 1; }

  // z = M r in the hierarchical ordering (r, z: device vectors of size n; r is preserved)
  void apply_permuted(const double* r, double* z);
  // y = M x in natural ordering (PCApply_AIR semantics)
  void apply_natural(const double* d_x, double* d_y);
  // natural <-> hierarchical ordering
  void to_permuted(const double* nat, double* perm);
  void to_natural(const double* perm, double* nat);

  const DevMat& A_perm() const { return A_perm_; }
  hipStream_t stream() const { return stream_; }
  double cycle_alg_bytes() const { return cycle_bytes_; }
  double storage_bytes() const;

  void set_profile(bool on) { profile_ = on; }
  void reset_stats();
  void resolve_events();
  const std::vector<KernelStat>& stats() const { return stats_; }

  void set_use_graph(bool on) { want_graph_ = on; }
  // called right before the cycle is recorded into a graph: the owner waits there for everything of its own that may issue
  // HIP calls from another thread (the host-mirror thread of a setup: its copies and frees invalidate a capture in progress)
  void set_before_capture(std::function<void()> f) { before_capture_ = std::move(f); }
  std::vector<std::unique_ptr<DevLevel>> levels_;
  DevInv coarse_inv_;
  DevMat coarse_A_;
  int coarse_richardson_its_ = 0;
  std::vector<int64_t> off_;     // off_[l]: start of the level-l vector inside the level-1 vector

  // a timed section for kernels outside the cycle (the outer Krylov method's): HIP events under (name, level) while
  // profiling is on; `kernel` = the HIP kernel function, the row a kernel trace aggregates by
  void prof_begin(const char* name, int level, double bytes, const char* kernel) {
    if (!profile_) return;
    const int id = stat_id(name, level, bytes);
    stats_[id].kernel = kernel;
    begin_kernel(id);
  }
  void prof_end() { if (profile_) end_kernel(); }
  // SpMV on the library stream, timed under (name, level) when profiling is on
  void spmv(const DevMat& A, const double* x, const SpmvEpi& e, const char* name, int level, double bytes_override = -1);

 private:
  void vcycle();                                        // cb_ -> cx_
  void vcycle_full();                                   // -pc_air_full_smoothing_up_and_down
  void coarse_solve(const double* bL, double* xL);
  bool full_ = false;
  std::vector<DBuf<double>> xfull_;                     // full smoothing: solutions of levels >= 1
  void run_cycle();                                     // vcycle, through a HIP graph when enabled
  bool graph_ok() const;
  void f_smooths(DevLevel& L, int li, bool first, int its);
  void c_smooths(DevLevel& L, int li, bool first, int its);
  void apply_inv(DevInv& I, const double* r, double* x, bool add, int stat_level, const char* tag);
  int stat_id(const char* name, int level, double bytes);
  void begin_kernel(int stat);
  void end_kernel();
  void upload_inv(const InvOp& h, DevInv& d, const DevMat* Aop, const double* host_diag_of_op, const HaloHost* halo = nullptr,
                  const int32_t* d_rowsrc = nullptr, const int32_t* d_colmap = nullptr, const int32_t* pos_host = nullptr);
  SetupAccel* source_ = nullptr;
  std::vector<int64_t> hint_;    // optional locality keys of the fine rows (see set_ordering_hint)
  // distributed operands: halo plan creation and the exchange in front of a product
  void attach_halo(DevMat& M, const HaloHost* h, const int32_t* host_colmap);
  const double* halo_update(const DevMat& A, const double* x);
  Comm* comm_ = nullptr;
  std::vector<std::unique_ptr<HaloPlan>> plans_;
  // exchange / compute overlap: the exchange runs on its own stream while the interior row blocks (no ghost column)
  // are processed; the boundary blocks follow once the ghosts have arrived.  Default (overlap_mode_ = -1): ON for a
  // transport whose exchange is pure device work (RCCL: the send / receive group then costs its latency only where an
  // operand has too few interior blocks to cover it), OFF for the host-staged transport, whose exchange synchronises the
  // host anyway (measured slower there).  AIRHIP_OVERLAP=0|1 forces it; correctness is covered with the staged transport
  // by the 2- and 4-rank GPU tests.
  int overlap_mode_ = -1;
  bool overlap_ = false;
  int64_t overlap_min_blocks_ = 8192;   // interior row blocks (x 512 non-zeros) below which the plain path is used
  hipStream_t comm_stream_ = nullptr;
  hipEvent_t ev_pack_ = nullptr, ev_recv_ = nullptr;
  hipEvent_t exch_end_ = nullptr;       // profiling: end of the exchange in flight (begin_exchange -> wait_exchange)
  bool begin_exchange(const DevMat& A, const double* x);   // true: split launch (interior, wait_exchange, boundary)
  void wait_exchange();
  // level-0 rhs is only read and the solution of every level is a suffix of one buffer: PCApply in
  // hierarchical ordering therefore works straight on the caller's vectors (no copies)
  double* bvec(int l) { return l == 0 ? const_cast<double*>(b0_) : b_ptr_[l]; }
  double* xvec(int l) { return xbase_ + off_[l]; }
  const double* b0_ = nullptr;
  double* xbase_ = nullptr;

  int64_t n_ = 0;
  hipStream_t stream_ = nullptr;
  bool any_c_ = false;
  DBuf<double> cb_, cx_;          // level-1 rhs / solution in hierarchical ordering
  std::vector<DBuf<double>> b_sep_;   // separate rhs per level when C smoothing keeps b_c alive
  std::vector<double*> b_ptr_;
  DBuf<int32_t> perm_, iperm_;    // natural -> hierarchical and back (level 1)
  // the same permutation walked chunk by chunk of the locality key (kernels.hip: k_permute_chunks): chunk_starts_[c * nsec + s]
  DBuf<int32_t> chunk_starts_; int64_t perm_nchunks_ = 0; int perm_nsec_ = 0;
  DevMat A_perm_;
  DBuf<double> coarse_r_;
  double cycle_bytes_ = 0.0;
  // profiling
  bool profile_ = false;
  std::vector<KernelStat> stats_;
  std::map<std::string, int> stat_index_;
  std::vector<PendingEvent> pending_;
  std::vector<hipEvent_t> event_pool_;
  int cur_stat_ = -1; hipEvent_t cur_a_ = nullptr;
  // chained events (pcair_dev.cpp: begin_kernel): the stop event of one cycle kernel is the start event of the next
  bool cur_a_shared_ = false, chain_armed_ = false; hipEvent_t last_b_ = nullptr; int last_level_ = -1;
  // RAII timed section for the cycle's small vector kernels
  struct Aux { DevPCAIR* p; Aux(DevPCAIR* q, const char* name, int level, double bytes, const char* kernel) : p(q) { p->prof_begin(name, level, bytes, kernel); } ~Aux() { p->prof_end(); } };
  // graph
  bool want_graph_ = false;
  std::function<void()> before_capture_;
  hipGraphExec_t graph_exec_ = nullptr;
};

// ---------------------------------------------------------------------------------------
// Outer Krylov (PETSc KSPRICHARDSON / KSPGMRES semantics) on the device
// ---------------------------------------------------------------------------------------
struct KspOptions {
  int type = 1;            // 0 richardson, 1 gmres
  int pc_side = 0;         // 0 left, 1 right
  int norm_type = 0;       // richardson: 0 preconditioned, 1 unpreconditioned
  double rtol = 1e-5, atol = 1e-50, dtol = 1e4;
  int max_it = 10000;
  int restart = 30;
};
struct KspResult { int its = 0; int reason = 0; double rnorm = 0.0; double solve_ms = 0.0; std::vector<double> history; int pc_applies = 0; };

// Generic preconditioner callback in whatever ordering the operator uses.
struct KspOperator {
  const DevMat* A = nullptr;
  hipStream_t stream = nullptr;
  // z = M r ; null = identity
  std::function<void(const double*, double*)> pc;
  // optional replacement of launch_spmv(*A, ...) (lets the owner time the outer operator product;
  // mandatory for a distributed operator, whose product needs a halo exchange)
  std::function<void(const double*, const SpmvEpi&)> mult;
  Comm* comm = nullptr;         // distributed: dots and norms are summed over the ranks
  // optional: the owner's statistics time the solver's own kernels too (name, algorithmic bytes, kernel function)
  std::function<void(const char*, double, const char*)> prof_begin;
  std::function<void()> prof_end;
};

int ksp_set_tuning(const char* key, int value);      // ksp_dev.cpp
class DevKSP {
 public:
  explicit DevKSP(int64_t n);
  ~DevKSP() = default;
  KspResult solve(const KspOperator& op, const KspOptions& o, const double* d_b, double* d_x);
 private:
  struct Timed;
  double norm2(const KspOperator& op, const double* v, hipStream_t s);
  void scale(const KspOperator& op, double a, double* v, hipStream_t s);
  void fetch_dots(const double* d, double* host, int n, hipStream_t s);
  Comm* comm_ = nullptr;
  KspResult richardson(const KspOperator& op, const KspOptions& o, const double* b, double* x);
  KspResult gmres(const KspOperator& op, const KspOptions& o, const double* b, double* x);
  int64_t n_;
  DBuf<double> V_, Z_, w_, t_, scratch_, ddots_, dcoef_, dnorm_;      // Z_: the preconditioned basis (ksp_dev.cpp: g_keep_pc)
  int v_alloc_ = 0, z_alloc_ = 0;
};

}  // namespace airhip
