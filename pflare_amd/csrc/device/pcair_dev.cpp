// Device hierarchy upload, the AIR V-cycle (Kaskade: restrict the rhs down, coarse
// apply, prolong + FC-point relaxation up) and the matrix-free polynomial applies.
//
// Reference semantics followed: PCMG Kaskade wiring
// src/AIR_MG_Setup.F90:967-1156; mg_FC_point_richardson / f_smooths / c_smooths
// src/FC_Smooth.F90:421-640; petsc_horner src/Gmres_Poly.F90:1418-1484; petsc_newton
// src/Gmres_Poly_Newton.F90:763-875; Neumann shell src/Neumann_Poly.F90:19-55.
#include "pcair_dev.hpp"
#include "devmem.hpp"
#include "extract.hpp"
#include <parallel/algorithm>
#include <cstdio>

namespace airhip {

// where the upload of a hierarchy spends its time (printed with -pc_air_print_stats_timings)
static double g_t_h2d = 0.0, g_t_convert = 0.0, g_t_fuse = 0.0;

// the setup accelerator that built the hierarchy being uploaded (DevPCAIR::set_device_source), or null
static thread_local SetupAccel* t_source = nullptr;
static int64_t g_n_from_device = 0;

// The CSR arrays of an operator of the host hierarchy on the device, for as long as the upload needs them: the copy the
// setup left here (lent by the accelerator, handed back - or dropped - at the end) or a staged copy of the host arrays
// (blocks from the cache, back to it afterwards).
struct DevCsrLease {
  int64_t m = 0, n = 0, nnz = 0;
  const int64_t* rp = nullptr; const int32_t* ci = nullptr; const double* v = nullptr;
  uint64_t tag = 0;                          // != 0: the accelerator's copy
  void *s_rp = nullptr, *s_ci = nullptr, *s_v = nullptr;
  explicit DevCsrLease(const Csr& A) : m(A.m), n(A.n), nnz(A.nnz()) {
    if (t_source && A.device_tag && t_source->device_copy(A.device_tag, A.m, A.nnz(), &rp, &ci, &v)) { tag = A.device_tag; g_n_from_device++; return; }
    if (!A.on_host()) throw std::logic_error("upload of an operator that is neither on the host nor kept on the device");
    const double t0 = now_s();
    try {
      s_rp = dev_alloc(sizeof(int64_t) * (size_t)(m + 1)); s_ci = dev_alloc(sizeof(int32_t) * (size_t)std::max<int64_t>(nnz, 1));
      s_v = dev_alloc(sizeof(double) * (size_t)std::max<int64_t>(nnz, 1));
      staged_h2d(s_rp, A.rp.data(), sizeof(int64_t) * (size_t)(m + 1));
      if (nnz > 0) { staged_h2d(s_ci, A.ci.data(), sizeof(int32_t) * (size_t)nnz); staged_h2d(s_v, A.v.data(), sizeof(double) * (size_t)nnz); }
    } catch (...) { dev_free(s_rp); dev_free(s_ci); dev_free(s_v); throw; }
    rp = (const int64_t*)s_rp; ci = (const int32_t*)s_ci; v = (const double*)s_v;
    g_t_h2d += now_s() - t0;
  }
  ~DevCsrLease() {
    if (tag) t_source->drop_device_copy(tag);
    dev_free(s_rp); dev_free(s_ci); dev_free(s_v);
  }
  DevCsrLease(const DevCsrLease&) = delete; DevCsrLease& operator=(const DevCsrLease&) = delete;
  void convert(const int32_t* d_rowsrc, const int32_t* d_colmap, DevMat& out, hipStream_t s, int64_t n_ghost) const {
    const double t0 = now_s();
    devmat_from_csr_device(m, n, rp, ci, v, d_rowsrc, d_colmap, out, s, n_ghost);
    g_t_convert += now_s() - t0;
  }
};

void upload_mat(const Csr& A, const int32_t* d_rowsrc, const int32_t* d_colmap, DevMat& out, hipStream_t s, int64_t n_ghost) {
  DevCsrLease lease(A);
  lease.convert(d_rowsrc, d_colmap, out, s, n_ghost);
}

// Processing order of an operator's row blocks by the locality key of each block's first row (host arrays:
// dev_row_to_host[r] = host row stored as device row r, key[host row]).  See DevMat::desc_ord.
static int64_t g_block_order_min = 4096;      // smaller operators sit in the caches anyway
void set_block_order_min(int64_t nblocks) { g_block_order_min = nblocks > 0 ? nblocks : 1; }
static bool g_fc_fuse = true;
void set_fc_fuse(bool on) { g_fc_fuse = on; }
// (airhip_set_tuning "profile_chain") the per-kernel events of the profile: 1 = inside the cycle the stop event of a kernel
// is the start event of the next (half as many events in the stream), 0 = a pair of events per kernel
static bool g_profile_chain = true;
void set_profile_chain(bool on) { g_profile_chain = on; }
static bool g_perm_chunks = true;     // (airhip_set_tuning "perm_chunks", acts at the next launch) 0: the plain gathers
void set_perm_chunks(bool on) { g_perm_chunks = on; }
static int g_fuse_first = 1;          // pcair_dev.hpp: set_fuse_first_sweep
static bool g_fuse_first_run = true;
void set_fuse_first_sweep(int mode) { g_fuse_first = mode < 0 ? 0 : (mode > 2 ? 2 : mode); }
void set_fuse_first_sweep_run(bool on) { g_fuse_first_run = on; }
int fuse_first_sweep_mode() { return g_fuse_first; }
static int g_fc_spatial = 1;       // 0 off, 1 levels where it pays (upload), 2 every level (tests)
void set_fc_spatial(int mode) { g_fc_spatial = mode; }

static void order_blocks_by_key(DevMat& M, const int32_t* dev_row_to_host, const int64_t* key, hipStream_t s) {
  if (M.nblocks < g_block_order_min || M.cached) return;
  std::vector<BlkDesc> blk((size_t)M.nblocks);
  HIP_CHECK(hipMemcpyAsync(blk.data(), M.blk, sizeof(BlkDesc) * (size_t)M.nblocks, hipMemcpyDeviceToHost, s));
  HIP_CHECK(hipStreamSynchronize(s));
  std::vector<std::pair<int64_t, int32_t>> kb((size_t)M.nblocks);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < M.nblocks; b++) kb[(size_t)b] = {key[dev_row_to_host ? dev_row_to_host[blk[(size_t)b].row] : blk[(size_t)b].row], (int32_t)b};
  __gnu_parallel::sort(kb.begin(), kb.end());
  // the descriptors themselves in that order (a launch reads them front to back: no index list in between), multi-base
  // blocks first (DevMat: launch lists); replaces the storage-order list of the conversion
  std::vector<BlkDesc> ord((size_t)M.nblocks);
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < M.nblocks; b++) ord[(size_t)b] = blk[(size_t)kb[(size_t)b].second];
  M.n_mb = partition_descs_host(ord);
  HIP_CHECK(hipMemcpy(M.desc_ord, ord.data(), sizeof(BlkDesc) * (size_t)M.nblocks, hipMemcpyHostToDevice));
  if (M.desc_int && M.desc_bnd) {
    // distributed operand with the interior / boundary split: both lists in the same spatial order
    std::vector<uint8_t> fl;
    devmat_boundary_flags(M, fl, s);
    std::vector<BlkDesc> li, lb; li.reserve((size_t)M.n_int); lb.reserve((size_t)M.n_bnd);
    for (int64_t b = 0; b < M.nblocks; b++) { const int32_t q = kb[(size_t)b].second; (fl[(size_t)q] ? lb : li).push_back(blk[(size_t)q]); }
    M.n_mb_int = partition_descs_host(li); M.n_mb_bnd = partition_descs_host(lb);
    if (!li.empty()) HIP_CHECK(hipMemcpy(M.desc_int, li.data(), sizeof(BlkDesc) * li.size(), hipMemcpyHostToDevice));
    if (!lb.empty()) HIP_CHECK(hipMemcpy(M.desc_bnd, lb.data(), sizeof(BlkDesc) * lb.size(), hipMemcpyHostToDevice));
  }
}

double DevInv::alg_bytes() const {
  if (kind == InvOp::ASSEMBLED) return mat.s_bytes();
  if (kind == InvOp::DIAG) return 12.0 * n + 4.0 * (n + 1) + 8.0 * n + 8.0 * n;
  if (Aop) return spmvs * Aop->s_bytes();
  return 0.0;
}

DevPCAIR::DevPCAIR() {
  HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
  HIP_CHECK(hipEventCreateWithFlags(&ev_pack_, hipEventDisableTiming));
  HIP_CHECK(hipEventCreateWithFlags(&ev_recv_, hipEventDisableTiming));
  if (const char* e = std::getenv("AIRHIP_OVERLAP")) overlap_mode_ = std::atoi(e) != 0 ? 1 : 0;
  if (const char* e = std::getenv("AIRHIP_OVERLAP_MIN_BLOCKS")) overlap_min_blocks_ = std::atoll(e);
}

DevPCAIR::~DevPCAIR() {
  if (graph_exec_) (void)hipGraphExecDestroy(graph_exec_);
  for (auto e : event_pool_) (void)hipEventDestroy(e);
  for (auto& p : pending_) { if (!p.a_shared) (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
  devmat_free(A_perm_); devmat_free(coarse_A_);
  if (stream_) (void)hipStreamDestroy(stream_);
  if (comm_stream_) (void)hipStreamDestroy(comm_stream_);
  if (ev_pack_) (void)hipEventDestroy(ev_pack_);
  if (ev_recv_) (void)hipEventDestroy(ev_recv_);
}

// ---------------------------------------------------------------------------------------
// profiling helpers: HIP events on the launch stream around every kernel
// ---------------------------------------------------------------------------------------
int DevPCAIR::stat_id(const char* name, int level, double bytes) {
  std::string key = std::string(name) + "@" + std::to_string(level);
  auto it = stat_index_.find(key);
  if (it != stat_index_.end()) return it->second;
  KernelStat k; k.name = name; k.level = level; k.alg_bytes = bytes;
  stats_.push_back(k);
  stat_index_[key] = (int)stats_.size() - 1;
  return (int)stats_.size() - 1;
}
// Inside the cycle (statistics of level >= 0, one device) the kernels follow one another on the stream with nothing in
// between, so the stop event of one is the start event of the next: half as many events in the stream.  A kernel's time
// then includes the few microseconds between the end of the kernel before it and its own start - what a step pays for
// it.  The Krylov method's kernels (level -1) have host round trips between them and keep their own pair of events, as do
// the kernels of a distributed cycle (exchanges on a second stream in between).
void DevPCAIR::begin_kernel(int stat) {
  if (!profile_) return;
  cur_stat_ = stat;
  if (g_profile_chain && chain_armed_ && !comm_ && last_level_ >= 0 && stats_[stat].level >= 0) { cur_a_ = last_b_; cur_a_shared_ = true; return; }
  hipEvent_t a;
  if (!event_pool_.empty()) { a = event_pool_.back(); event_pool_.pop_back(); } else HIP_CHECK(hipEventCreate(&a));
  HIP_CHECK(hipEventRecord(a, stream_));
  cur_a_ = a; cur_a_shared_ = false;
}
void DevPCAIR::end_kernel() {
  if (!profile_) return;
  hipEvent_t b;
  if (!event_pool_.empty()) { b = event_pool_.back(); event_pool_.pop_back(); } else HIP_CHECK(hipEventCreate(&b));
  HIP_CHECK(hipEventRecord(b, stream_));
  pending_.push_back({cur_a_, b, cur_stat_, cur_a_shared_});
  last_b_ = b; last_level_ = stats_[cur_stat_].level; chain_armed_ = true;
}
void DevPCAIR::resolve_events() {
  if (pending_.empty()) return;
  HIP_CHECK(hipStreamSynchronize(stream_));
  if (comm_stream_) HIP_CHECK(hipStreamSynchronize(comm_stream_));      // (the exchange statistics pair events of both streams)
  for (auto& p : pending_) {
    float ms = 0.f;
    // a pair across the two streams may be "in the wrong order" (an exchange that ended before the interior blocks did:
    // nothing exposed) - a negative time or an error from the runtime both mean zero here
    if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) { (void)hipGetLastError(); ms = 0.f; }
    if (ms < 0.f) ms = 0.f;
    stats_[p.stat].calls++; stats_[p.stat].ms += ms;
    if (!p.a_shared) event_pool_.push_back(p.a);
    event_pool_.push_back(p.b);
  }
  pending_.clear();
  chain_armed_ = false; last_b_ = nullptr;
}
void DevPCAIR::reset_stats() { resolve_events(); for (auto& s : stats_) { s.calls = 0; s.ms = 0.0; } }

void DevPCAIR::spmv(const DevMat& A, const double* x, const SpmvEpi& e, const char* name, int level, double bytes_override) {
  int id = -1;
  if (profile_) { id = stat_id(name, level, bytes_override >= 0 ? bytes_override : A.s_bytes()); stats_[id].kernel = "k_rb_spmv"; }
  if (begin_exchange(A, x)) {
    if (profile_) begin_kernel(id);
    launch_spmv(A, x, e, stream_, A.halo->recvbuf, RB_INTERIOR);   // rows that read no ghost value
    wait_exchange();
    launch_spmv(A, x, e, stream_, A.halo->recvbuf, RB_BOUNDARY);
    if (profile_) end_kernel();
    return;
  }
  const double* xg = halo_update(A, x);
  if (!profile_) { launch_spmv(A, x, e, stream_, xg); return; }
  // times the product, not the exchange; the blocks outside the multi-base form run in a kernel of their own
  // (k_rb_spmv_g) and are accounted for under that name, with their share of the operator's bytes
  const double bytes = bytes_override >= 0 ? bytes_override : A.s_bytes(), share = A.mb_share();
  if (A.n_mb > 0) {
    stats_[id].alg_bytes = A.n_mb < A.nblocks ? bytes * share : bytes;
    begin_kernel(id);
    launch_spmv(A, x, e, stream_, xg, RB_ALL, RB_MB);
    end_kernel();
  }
  if (A.n_mb < A.nblocks) {
    const int idg = A.n_mb > 0 ? stat_id((std::string(name) + " [other blocks]").c_str(), level, bytes * (1.0 - share)) : id;
    stats_[idg].kernel = "k_rb_spmv_g";
    begin_kernel(idg);
    launch_spmv(A, x, e, stream_, xg, RB_ALL, RB_GENERAL);
    end_kernel();
  }
}

// Starts the neighbour exchange of a distributed operand on the communication stream.  Ordering: the
// pack kernel runs on the compute stream after everything that produced x; the exchange waits for it;
// the boundary launch waits for the exchange (wait_exchange); the NEXT pack is queued behind that
// boundary launch on the compute stream, so a plan's buffers are never rewritten while still in use.
bool DevPCAIR::begin_exchange(const DevMat& A, const double* x) {
  // worth it only when the interior part runs longer than an exchange takes (~20 us ~ 50 MB of matrix)
  if (!A.halo || !overlap_ || !A.desc_int || A.n_int < overlap_min_blocks_) return false;
  HaloPlan& p = *A.halo;
  if (p.n_send) launch_gather(p.n_send, p.send_idx, x, p.sendbuf, stream_);
  HIP_CHECK(hipEventRecord(ev_pack_, stream_));
  HIP_CHECK(hipStreamWaitEvent(comm_stream_, ev_pack_, 0));
  comm_->exchange(p, comm_stream_);
  HIP_CHECK(hipEventRecord(ev_recv_, comm_stream_));
  if (profile_) {      // when the exchange was over, for wait_exchange's "exposed" statistic
    hipEvent_t b;
    if (!event_pool_.empty()) { b = event_pool_.back(); event_pool_.pop_back(); } else HIP_CHECK(hipEventCreate(&b));
    HIP_CHECK(hipEventRecord(b, comm_stream_));
    exch_end_ = b;
  }
  return true;
}
// the boundary blocks wait for the ghosts.  Profiling: the time between the end of the interior blocks and the end of the
// exchange (zero when the exchange was hidden) goes to the statistic "halo exchange, exposed"
void DevPCAIR::wait_exchange() {
  if (profile_ && exch_end_) {
    hipEvent_t a;
    if (!event_pool_.empty()) { a = event_pool_.back(); event_pool_.pop_back(); } else HIP_CHECK(hipEventCreate(&a));
    HIP_CHECK(hipEventRecord(a, stream_));
    const int id = stat_id("halo exchange, exposed", -2, 0.0);
    stats_[id].kernel = "halo_exchange";
    pending_.push_back({a, exch_end_, id, false});
    exch_end_ = nullptr;
  }
  HIP_CHECK(hipStreamWaitEvent(stream_, ev_recv_, 0));
}

// ---------------------------------------------------------------------------------------
// distributed operands
// ---------------------------------------------------------------------------------------
void DevPCAIR::attach_halo(DevMat& M, const HaloHost* h, const int32_t* host_colmap) {
  if (!h || !comm_) return;   // also for an operand without local rows: its owner still serves the neighbours
  auto p = std::make_unique<HaloPlan>();
  p->peers.assign(h->peers.begin(), h->peers.end());
  p->send_off = h->send_off; p->recv_off = h->recv_off;
  if (p->send_off.empty()) { p->send_off.push_back(0); p->recv_off.push_back(0); }
  p->n_send = p->send_off.back(); p->n_ghost = h->n_ghost;
  std::vector<int32_t> idx((size_t)p->n_send);
  for (int64_t i = 0; i < p->n_send; i++) idx[i] = host_colmap ? host_colmap[h->send_idx[i]] : h->send_idx[i];
  HIP_CHECK(hipMalloc(&p->send_idx, sizeof(int32_t) * (size_t)std::max<int64_t>(p->n_send, 1)));
  HIP_CHECK(hipMalloc(&p->sendbuf, sizeof(double) * (size_t)std::max<int64_t>(p->n_send, 1)));
  HIP_CHECK(hipMalloc(&p->recvbuf, sizeof(double) * (size_t)std::max<int64_t>(p->n_ghost, 1)));
  if (p->n_send) HIP_CHECK(hipMemcpy(p->send_idx, idx.data(), sizeof(int32_t) * (size_t)p->n_send, hipMemcpyHostToDevice));
  M.halo = p.get();
  plans_.push_back(std::move(p));
  if (overlap_ && M.valid() && M.n_ghost > 0) {
    std::vector<uint8_t> fl;
    devmat_boundary_flags(M, fl, stream_);
    std::vector<int32_t> li, lb;
    for (int64_t b = 0; b < M.nblocks; b++) (fl[b] ? lb : li).push_back((int32_t)b);
    M.n_int = (int64_t)li.size(); M.n_bnd = (int64_t)lb.size();
    M.desc_int = devmat_desc_list(M, li.data(), M.n_int, stream_, &M.n_mb_int);
    M.desc_bnd = devmat_desc_list(M, lb.data(), M.n_bnd, stream_, &M.n_mb_bnd);
  }
}

// Every rank calls this for the same operand in the same order (the cycle is SPMD), also when it
// has nothing to send or receive itself.
const double* DevPCAIR::halo_update(const DevMat& A, const double* x) {
  if (!A.halo) return nullptr;
  HaloPlan& p = *A.halo;
  // (not overlapped: the whole exchange, pack kernel included, is exposed)
  hipEvent_t saved_a = cur_a_; const int saved_stat = cur_stat_; const bool saved_shared = cur_a_shared_;
  if (profile_) { const int id = stat_id("halo exchange, exposed", -2, 0.0); stats_[id].kernel = "halo_exchange"; begin_kernel(id); }
  if (p.n_send) launch_gather(p.n_send, p.send_idx, x, p.sendbuf, stream_);
  comm_->exchange(p, stream_);
  if (profile_) { end_kernel(); cur_a_ = saved_a; cur_stat_ = saved_stat; cur_a_shared_ = saved_shared; }
  return p.recvbuf;
}

// ---------------------------------------------------------------------------------------
// upload
// ---------------------------------------------------------------------------------------
// d_rowsrc / d_colmap / pos_host: placement of the operator's index space on the device (null = identity);
// host_diag_of_op is already in device order
void DevPCAIR::upload_inv(const InvOp& h, DevInv& d, const DevMat* Aop, const double* host_diag_of_op, const HaloHost* halo,
                          const int32_t* d_rowsrc, const int32_t* d_colmap, const int32_t* pos_host) {
  d.kind = h.kind; d.coeffs = h.coeffs; d.re = h.re; d.im = h.im;
  d.diag_scale = h.diag_scale; d.neumann = h.neumann; d.spmvs = h.spmvs;
  if (h.kind == InvOp::ASSEMBLED) {
    d.n = h.mat.m;
    upload_mat(h.mat, d_rowsrc, d_colmap, d.mat, stream_, halo ? halo->n_ghost : 0);
    attach_halo(d.mat, halo, pos_host);
  }
  else if (h.kind == InvOp::DIAG) {
    d.n = h.d.size();
    if (pos_host) {
      std::vector<double> dd(d.n);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < d.n; i++) dd[pos_host[i]] = h.d[i];
      d.d.upload(dd.data(), d.n);
    } else d.d.upload(h.d.data(), h.d.size());
  }
  else if (h.kind == InvOp::HORNER || h.kind == InvOp::NEWTON) {
    d.Aop = Aop; d.n = Aop->m;
    d.t1.alloc(d.n); d.t2.alloc(d.n); d.t3.alloc(d.n);
    if (h.diag_scale) {
      std::vector<double> di(d.n);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < d.n; i++) di[i] = 1.0 / host_diag_of_op[i];
      d.dinv.upload(di.data(), d.n);
      d.rhs.alloc(d.n);
    }
  }
}

void DevPCAIR::upload(const Hierarchy& H, const Csr& A_top, const LocalHierarchy* dist, Comm* comm) {
  dev_stage("upload of the hierarchy");
  dev_cache_trim_to((size_t)40 << 30);      // (devmem.hpp: what the setup left idle slows the allocations of the arrays that stay)
  const int L = H.no_levels();
  const double t_upload0 = now_s();
  g_t_h2d = g_t_convert = g_t_fuse = 0.0; g_n_from_device = 0;
  struct SourceScope { SetupAccel* prev; explicit SourceScope(SetupAccel* a) : prev(t_source) { t_source = a; } ~SourceScope() { t_source = prev; } } source_scope(dist ? nullptr : source_);
  comm_ = dist ? comm : nullptr;
  if (dist && !comm) throw std::runtime_error("a distributed hierarchy needs a communicator");
  overlap_ = comm_ && (overlap_mode_ < 0 ? comm_->device_side() : overlap_mode_ != 0);
  plans_.clear();
  auto ng = [&](const HaloHost* h) -> int64_t { return (dist && h) ? h->n_ghost : 0; };
  n_ = A_top.m;
  any_c_ = H.opt.any_c_smooths();
  full_ = H.opt.full_smoothing && !H.levels.empty() && H.levels[0]->full;
  if (full_ && dist) throw std::runtime_error("full up / down smoothing is not available for a distributed hierarchy");
  if (full_) any_c_ = false;
  coarse_richardson_its_ = H.opt.coarse_richardson_its;
  levels_.clear();
  off_.assign(L, 0);
  for (int l = 0; l + 1 < L; l++) off_[l + 1] = off_[l] + H.levels[l]->n_f;

  // Optional locality keys (hint_): rows with close keys are stored close together.  The key of a
  // coarse row is the key of its parent C point, so every level inherits the top-level locality.
  // fpos[l][i] = position of the level's i-th F point inside the F block (empty = identity).
  const bool use_key = !hint_.empty();
  if (use_key && (int64_t)hint_.size() != n_) throw std::runtime_error("ordering hint has the wrong length");
  std::vector<std::vector<int32_t>> fpos(L > 0 ? L - 1 : 0);
  std::vector<int32_t> cpos;       // coarsest level: natural index -> position (empty = identity)
  std::vector<std::vector<int64_t>> level_keys;   // locality key of every point of every level (level-local numbering)
  const bool order_blocks = true;      // (order_blocks_by_key itself skips operators below set_block_order_min's size: the switch)
  // level_keys[l][i]: locality key of point i of level l (level-local numbering) - the caller's hint on the top level, the
  // row's own index without one (the reference numbering is lexicographic, which is a locality key as well); a coarse
  // point inherits the key of its parent C point.  Used for the layout (with a hint) and for the processing order of
  // the row blocks of operators whose rows run through several sections (DevMat::desc_ord).
  {
    level_keys.resize(L);
    std::vector<int64_t> key;
    if (use_key) key = hint_;
    else { key.resize((size_t)n_); for (int64_t i = 0; i < n_; i++) key[(size_t)i] = i; }
    for (int l = 0; l + 1 < L; l++) {
      const Level& Lv = *H.levels[l];
      std::vector<int64_t> nxt((size_t)Lv.n_c);
#pragma omp parallel for schedule(static)
      for (int64_t j = 0; j < Lv.n_c; j++) nxt[(size_t)j] = key[Lv.C[j]];
      level_keys[l] = std::move(key);
      key = std::move(nxt);
    }
    level_keys[L - 1] = std::move(key);
  }
  if (use_key) {
    auto rank_by = [](const std::vector<int64_t>& k) {
      const int64_t n = (int64_t)k.size();
      std::vector<int32_t> ord(n), pos(n);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < n; i++) ord[i] = (int32_t)i;
      __gnu_parallel::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return k[a] < k[b]; });
#pragma omp parallel for schedule(static)
      for (int64_t r = 0; r < n; r++) pos[ord[r]] = (int32_t)r;
      return pos;
    };
    std::vector<int64_t> kf;
    for (int l = 0; l + 1 < L; l++) {
      const Level& Lv = *H.levels[l];
      kf.resize(Lv.n_f);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < Lv.n_f; i++) kf[i] = level_keys[l][Lv.F[i]];
      fpos[l] = rank_by(kf);
    }
    cpos = rank_by(level_keys[L - 1]);
  }
  // hierarchical permutations, coarsest level first: perm[l][i] = position of natural
  // index i of level l inside the level-l vector [F_l | level l+1 ...]
  std::vector<std::vector<int32_t>> perm(L);
  {
    const int64_t nL = H.coarse_A.m;
    perm[L - 1].resize(nL);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nL; i++) perm[L - 1][i] = use_key ? cpos[i] : (int32_t)i;
    for (int l = L - 2; l >= 0; l--) {
      const Level& Lv = *H.levels[l];
      perm[l].resize(Lv.n);
      int32_t* p = perm[l].data();
      const int32_t* pn = perm[l + 1].data();
      const int32_t* fp = use_key ? fpos[l].data() : nullptr;
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < Lv.n_f; i++) p[Lv.F[i]] = fp ? fp[i] : (int32_t)i;
#pragma omp parallel for schedule(static)
      for (int64_t j = 0; j < Lv.n_c; j++) p[Lv.C[j]] = (int32_t)(Lv.n_f + pn[j]);
    }
  }
  auto inverse = [](const std::vector<int32_t>& p) {
    std::vector<int32_t> ip(p.size());
    const int64_t n = (int64_t)p.size();
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++) ip[p[i]] = (int32_t)i;
    return ip;
  };

  const double t_order = now_s() - t_upload0;
  // top-level operator in hierarchical ordering (outer Krylov works in that ordering)
  {
    std::vector<int32_t> ip = inverse(perm[0]);
    perm_.upload(perm[0].data(), n_);
    iperm_.upload(ip.data(), n_);
    upload_mat(A_top, iperm_.p, perm_.p, A_perm_, stream_, ng(dist ? &dist->h_top : nullptr));
    attach_halo(A_perm_, dist ? &dist->h_top : nullptr, perm[0].data());
    if (order_blocks) order_blocks_by_key(A_perm_, ip.data(), level_keys[0].data(), stream_);
    // the permutation of level-1 vectors chunk by chunk of the key range (kernels.hip: k_permute_chunks): sections = the F
    // blocks of the levels and the coarsest level, each in key order; chunks of 1024 positions of the first section
    perm_nchunks_ = 0; perm_nsec_ = 0; chunk_starts_.release();
    constexpr int64_t PERM_CHUNK = 1024;
    if (L >= 2 && H.levels[0]->n_f >= 64 * PERM_CHUNK) {
      std::vector<int64_t> sec(L + 1);
      for (int l = 0; l < L; l++) sec[l] = off_[l];
      sec[L] = n_;
      // (a section's positions are in key order: rank_by with a hint, the ascending F / C lists without one)
      DBuf<int64_t> d_key, d_keyh, d_sec;
      d_key.upload(level_keys[0].data(), n_); d_keyh.alloc(n_); d_sec.upload(sec.data(), L + 1);
      const int64_t nch = (H.levels[0]->n_f + PERM_CHUNK - 1) / PERM_CHUNK;
      chunk_starts_.alloc((nch + 1) * L);
      launch_keys_at(n_, iperm_.p, d_key.p, d_keyh.p, stream_);
      launch_chunk_starts(nch, L, PERM_CHUNK, d_sec.p, d_keyh.p, chunk_starts_.p, stream_);
      HIP_CHECK(hipStreamSynchronize(stream_));
      perm_nchunks_ = nch; perm_nsec_ = L;
    }
  }

  cycle_bytes_ = 0.0;
  for (int l = 0; l + 1 < L; l++) {
    const Level& h = *H.levels[l];
    auto dl = std::make_unique<DevLevel>();
    DevLevel& d = *dl;
    d.n = h.n; d.n_f = h.n_f; d.n_c = h.n_c; d.aff_diag = h.aff_diag; d.one_point = h.one_point; d.nnz_A = h.nnz_A;
    for (int i = 0; i < 10; i++) d.smooth_order[i] = h.smooth_order[i];
    // permutation of the coarse (level l+1) numbering
    DBuf<int32_t> pc, ipc;
    std::vector<int32_t> ipn = inverse(perm[l + 1]);
    pc.upload(perm[l + 1].data(), h.n_c);
    ipc.upload(ipn.data(), h.n_c);

    // placement of the F points inside the F block (identity without locality keys)
    DBuf<int32_t> fpd, ifpd;
    const int32_t* fph = nullptr;     // host: F index -> position
    if (use_key) {
      std::vector<int32_t> ifp = inverse(fpos[l]);
      fpd.upload(fpos[l].data(), h.n_f);
      ifpd.upload(ifp.data(), h.n_f);
      fph = fpos[l].data();
    }
    const int32_t* Frow = use_key ? ifpd.p : nullptr;   // device row r takes host F row Frow[r]
    const int32_t* Fcol = use_key ? fpd.p : nullptr;    // host F column c becomes Fcol[c]
    auto fperm_vec = [&](const double* v) {             // host F-indexed vector -> device order
      std::vector<double> o(h.n_f);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < h.n_f; i++) o[fph ? fph[i] : i] = v[i];
      return o;
    };

    const LevelHalos* hl = dist ? &dist->halos[l] : nullptr;
    if (full_) {
      // level operator and its approximate inverse in the level's ordering [F block | level l+1 ...]
      DBuf<int32_t> pl, ipl;
      std::vector<int32_t> ipv = inverse(perm[l]);
      pl.upload(perm[l].data(), h.n); ipl.upload(ipv.data(), h.n);
      upload_mat(h.A_full, ipl.p, pl.p, d.A_full, stream_);
      if (order_blocks) order_blocks_by_key(d.A_full, ipv.data(), level_keys[l].data(), stream_);
      std::vector<double> dg;
      if ((h.inv_ff.kind == InvOp::HORNER || h.inv_ff.kind == InvOp::NEWTON) && h.inv_ff.diag_scale) {
        std::vector<double> nat(h.n); get_diagonal(h.A_full, nat.data());
        dg.resize(h.n);
        for (int64_t i = 0; i < h.n; i++) dg[perm[l][i]] = nat[i];
      }
      upload_inv(h.inv_ff, d.inv_ff, &d.A_full, dg.empty() ? nullptr : dg.data(), nullptr, ipl.p, pl.p, perm[l].data());
      d.inv_ff.n = h.n;
      if (order_blocks && d.inv_ff.kind == InvOp::ASSEMBLED) order_blocks_by_key(d.inv_ff.mat, ipv.data(), level_keys[l].data(), stream_);
      upload_mat(h.Z, ipc.p, Fcol, d.Z, stream_);
      if (order_blocks) order_blocks_by_key(d.Z, ipn.data(), level_keys[l + 1].data(), stream_);
      // the prolongator always as a matrix here (x_f += W e)
      d.one_point = false;
      if (h.one_point) {
        Csr Wc; Wc.m = h.n_f; Wc.n = h.n_c; Wc.rp.alloc(h.n_f + 1);
        Wc.rp[0] = 0;
        for (int64_t i = 0; i < h.n_f; i++) Wc.rp[i + 1] = Wc.rp[i] + (h.wcol[i] >= 0 ? 1 : 0);
        Wc.ci.alloc(Wc.rp[h.n_f]); Wc.v.alloc(Wc.rp[h.n_f]);
        for (int64_t i = 0; i < h.n_f; i++) if (h.wcol[i] >= 0) { Wc.ci[Wc.rp[i]] = h.wcol[i]; Wc.v[Wc.rp[i]] = 1.0; }
        upload_mat(Wc, Frow, pc.p, d.W, stream_);
      } else upload_mat(h.W, Frow, pc.p, d.W, stream_);
      d.rfull.alloc(h.n);
      auto S = [](double nnz, double m, double n) { return 12.0 * nnz + 4.0 * (m + 1) + 8.0 * m + 8.0 * n; };
      double nnzW = h.one_point ? 0.0 : (double)h.W.nnz();
      if (h.one_point) {
      int64_t cntw = 0;
#pragma omp parallel for schedule(static) reduction(+ : cntw)
      for (int64_t i = 0; i < h.n_f; i++) cntw += h.wcol[i] >= 0 ? 1 : 0;
      nnzW += (double)cntw;
    }
      cycle_bytes_ += S((double)h.Z.nnz() + h.n_c, h.n_c, h.n) + S(nnzW + h.n_c, h.n, h.n_c)
                    + 2.0 * (d.A_full.s_bytes() + d.inv_ff.alg_bytes());
      levels_.push_back(std::move(dl));
      continue;
    }
    // columns of A_fc (and of the one-point prolongator): positions in x_c = the level-(l+1) vector, or - single device,
    // set_fc_spatial - positions in the copy of x_c that f_smooths keeps in the spatial order of the C points
    std::vector<int32_t> csp;              // coarse index -> spatial rank (empty: not used)
    // Worth it where A_fc's rows are long: the copy moves 20 bytes per C point, the product streams ~10 bytes per non-zero and
    // gains ~8 % (464^3, profiles/r04_fc_spatial.txt: levels 5-10 gain 0.05-0.13 ms each for 0.01-0.06 ms of copy, levels 1-4
    // lose 0.03-0.2 ms each) - from 25 non-zeros of A_fc per C point on
    const bool fc_spatial = g_fc_spatial && !dist && (g_fc_spatial >= 2 || (h.n_c >= 4096 && h.A_fc.nnz() >= 25 * h.n_c));
    // The first F sweep on G = A_fc + A_ff W (set_fuse_first_sweep): relaxation that starts with F sweeps, one-point W,
    // sparse A_ff, an A_ff inverse applied as a linear operator (assembled or matrix-free).  One device: G is built below from the device copies of A_fc and A_ff.  Cut
    // hierarchy: EVERY rank must take the same decision (the exchanges are collective), so it is the cut that decides - a
    // piece either carries its rows of G (possibly none) or it does not; the predicate on a piece's own rows would differ
    const bool fuse_first = g_fuse_first && (dist ? (!h.G.empty() && h.one_point) : h.fuses_first_sweep());
    bool later_f_group = false;
    for (int i = 1; i < 10 && h.smooth_order[i] != 0; i++) later_f_group |= h.smooth_order[i] > 0;
    const bool need_fc = !fuse_first || g_fuse_first == 2 || later_f_group;
    DBuf<int32_t> cspd;
    if (fc_spatial) {
      const std::vector<int64_t>& kc = level_keys[l + 1];
      std::vector<int32_t> ord((size_t)h.n_c), src((size_t)h.n_c);
      csp.resize((size_t)h.n_c);
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < h.n_c; i++) ord[(size_t)i] = (int32_t)i;
      __gnu_parallel::sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return kc[(size_t)a] != kc[(size_t)b] ? kc[(size_t)a] < kc[(size_t)b] : a < b; });
      const int32_t* pn = perm[l + 1].data();
#pragma omp parallel for schedule(static)
      for (int64_t r = 0; r < h.n_c; r++) { csp[(size_t)ord[(size_t)r]] = (int32_t)r; src[(size_t)r] = pn[ord[(size_t)r]]; }
      d.xs_src.upload(src.data(), h.n_c);
      d.xs.alloc(h.n_c);
      cspd.upload(csp.data(), h.n_c);
    }
    if (fuse_first && dist) {
      upload_mat(h.G, Frow, pc.p, d.G, stream_, ng(&hl->G));
      attach_halo(d.G, &hl->G, perm[l + 1].data());
      if (need_fc) {
        upload_mat(h.A_fc, Frow, pc.p, d.A_fc, stream_, ng(&hl->A_fc));
        attach_halo(d.A_fc, &hl->A_fc, perm[l + 1].data());
      }
      upload_mat(h.A_ff, Frow, Fcol, d.A_ff, stream_, ng(&hl->A_ff));
    } else if (fuse_first) {
      // A_fc and A_ff as CSR on the device (the setup's copies, or staged), G built from them there, then all three
      // converted to row blocks - A_fc only where something still reads it
      const int32_t* fc_colmap = fc_spatial ? cspd.p : pc.p;
      DevCsrLease lfc(h.A_fc), lff(h.A_ff);
      {
        DBuf<int32_t> dw; dw.upload(h.wcol.data(), h.n_f);
        DCsr aff; aff.m = lff.m; aff.n = lff.n; aff.nnz = lff.nnz; aff.rp = const_cast<int64_t*>(lff.rp); aff.ci = const_cast<int32_t*>(lff.ci); aff.v = const_cast<double*>(lff.v);
        DCsr afc; afc.m = lfc.m; afc.n = lfc.n; afc.nnz = lfc.nnz; afc.rp = const_cast<int64_t*>(lfc.rp); afc.ci = const_cast<int32_t*>(lfc.ci); afc.v = const_cast<double*>(lfc.v);
        const double t0 = now_s();
        DCsr G = dev_fused_first_operator(aff, afc, dw.p, stream_);
        g_t_fuse += now_s() - t0;
        devmat_from_csr_device(G.m, G.n, G.rp, G.ci, G.v, Frow, fc_colmap, d.G, stream_, 0);
        dcsr_free(G);
      }
      if (need_fc) lfc.convert(Frow, fc_colmap, d.A_fc, stream_, 0);
      lff.convert(Frow, Fcol, d.A_ff, stream_, 0);
    } else if (fc_spatial) {
      upload_mat(h.A_fc, Frow, cspd.p, d.A_fc, stream_, 0);
    } else {
      upload_mat(h.A_fc, Frow, pc.p, d.A_fc, stream_, ng(hl ? &hl->A_fc : nullptr));   // rows F block, cols permuted coarse
      attach_halo(d.A_fc, hl ? &hl->A_fc : nullptr, perm[l + 1].data());
    }
    upload_mat(h.Z, ipc.p, Fcol, d.Z, stream_, ng(hl ? &hl->Z : nullptr));           // rows permuted coarse, cols F block
    attach_halo(d.Z, hl ? &hl->Z : nullptr, fph);
    if (order_blocks) order_blocks_by_key(d.Z, ipn.data(), level_keys[l + 1].data(), stream_);
    std::vector<double> affd_host;
    bool mf = (h.inv_ff.kind == InvOp::HORNER || h.inv_ff.kind == InvOp::NEWTON);
    if (h.aff_diag) {
      affd_host = fperm_vec(h.aff_d.data());
      d.aff_d.upload(affd_host.data(), h.n_f);
      if (mf) {  // matrix-free polynomial over a diagonal operator: keep one code path
        Csr D = csr_from_diag(affd_host.data(), h.n_f);
        upload_mat(D, nullptr, nullptr, d.A_ff, stream_);
      }
    } else {
      if (!fuse_first) upload_mat(h.A_ff, Frow, Fcol, d.A_ff, stream_, ng(hl ? &hl->A_ff : nullptr));
      attach_halo(d.A_ff, hl ? &hl->A_ff : nullptr, fph);
      if (mf && h.inv_ff.diag_scale) {
        std::vector<double> nat(h.n_f); get_diagonal(h.A_ff, nat.data());
        affd_host = fperm_vec(nat.data());
      }
    }
    upload_inv(h.inv_ff, d.inv_ff, &d.A_ff, affd_host.empty() ? nullptr : affd_host.data(), hl ? &hl->M_ff : nullptr, Frow, Fcol, fph);
    if (h.one_point) {
      std::vector<int32_t> wc(h.n_f);
      const int32_t* pn = perm[l + 1].data();
      // distributed: columns >= n_c are ghost columns of A_fc - of G on a level whose first sweep is fused - and keep their index
      const int32_t* wsrc = (dist && fuse_first) ? h.wcol_g.data() : h.wcol.data();
#pragma omp parallel for schedule(static)
      for (int64_t i = 0; i < h.n_f; i++) { const int32_t c = wsrc[i]; wc[fph ? fph[i] : i] = c < 0 ? -1 : (c < h.n_c ? (fc_spatial ? csp[(size_t)c] : pn[c]) : c); }
      d.wcol.upload(wc.data(), h.n_f);
    } else {
      upload_mat(h.W, Frow, pc.p, d.W, stream_, ng(hl ? &hl->W : nullptr));
      attach_halo(d.W, hl ? &hl->W : nullptr, perm[l + 1].data());
    }
    d.t4.alloc(h.n_f); d.r.alloc(h.n_f);
    if (any_c_) {
      upload_mat(h.A_cf, ipc.p, Fcol, d.A_cf, stream_, ng(hl ? &hl->A_cf : nullptr));
      attach_halo(d.A_cf, hl ? &hl->A_cf : nullptr, fph);
      if (order_blocks) order_blocks_by_key(d.A_cf, ipn.data(), level_keys[l + 1].data(), stream_);
      upload_mat(h.A_cc, ipc.p, pc.p, d.A_cc, stream_, ng(hl ? &hl->A_cc : nullptr));
      attach_halo(d.A_cc, hl ? &hl->A_cc : nullptr, perm[l + 1].data());
      if (order_blocks) order_blocks_by_key(d.A_cc, ipn.data(), level_keys[l + 1].data(), stream_);
      // inverse of A_cc lives in the permuted coarse numbering as well
      if (h.inv_cc.kind == InvOp::ASSEMBLED) {
        d.inv_cc.kind = InvOp::ASSEMBLED; d.inv_cc.n = h.n_c;
        upload_mat(h.inv_cc.mat, ipc.p, pc.p, d.inv_cc.mat, stream_, ng(hl ? &hl->M_cc : nullptr));
        attach_halo(d.inv_cc.mat, hl ? &hl->M_cc : nullptr, perm[l + 1].data());
        if (order_blocks) order_blocks_by_key(d.inv_cc.mat, ipn.data(), level_keys[l + 1].data(), stream_);
      } else if (h.inv_cc.kind == InvOp::DIAG) {
        std::vector<double> dd(h.n_c);
        for (int64_t j = 0; j < h.n_c; j++) dd[perm[l + 1][j]] = h.inv_cc.d[j];
        d.inv_cc.kind = InvOp::DIAG; d.inv_cc.n = h.n_c; d.inv_cc.d.upload(dd.data(), h.n_c);
      } else {
        std::vector<double> dg;
        if (h.inv_cc.diag_scale) {
          std::vector<double> nat(h.n_c); get_diagonal(h.A_cc, nat.data());
          dg.resize(h.n_c);
          for (int64_t j = 0; j < h.n_c; j++) dg[perm[l + 1][j]] = nat[j];
        }
        upload_inv(h.inv_cc, d.inv_cc, &d.A_cc, dg.empty() ? nullptr : dg.data());
      }
      d.t4c.alloc(h.n_c); d.rc.alloc(h.n_c);
    }
    // algorithmic bytes of this level, identity blocks of R/P included as the
    // reference counts them
    auto S = [](double nnz, double m, double n) { return 12.0 * nnz + 4.0 * (m + 1) + 8.0 * m + 8.0 * n; };
    double nnzW = h.one_point ? 0.0 : (double)h.W.nnz();
    if (h.one_point) {
      int64_t cntw = 0;
#pragma omp parallel for schedule(static) reduction(+ : cntw)
      for (int64_t i = 0; i < h.n_f; i++) cntw += h.wcol[i] >= 0 ? 1 : 0;
      nnzW += (double)cntw;
    }
    cycle_bytes_ += S((double)h.Z.nnz() + h.n_c, h.n_c, h.n) + S(nnzW + h.n_c, h.n, h.n_c);
    int g = 0, sf = 0, gc = 0, sc = 0;
    for (int i = 0; i < 10; i++) { int s = h.smooth_order[i]; if (s > 0) { g++; sf += s; } else if (s < 0) { gc++; sc += -s; } }
    double affB = h.aff_diag ? S(h.n_f, h.n_f, h.n_f) : d.A_ff.s_bytes();
    // (the reference's count whatever the device executes: on a fused level A_fc may not even be there)
    cycle_bytes_ += g * (d.A_fc.valid() ? d.A_fc.s_bytes() : S((double)h.A_fc.nnz(), h.n_f, h.n_c)) + sf * (affB + d.inv_ff.alg_bytes());
    if (any_c_) cycle_bytes_ += gc * d.A_cf.s_bytes() + sc * (d.A_cc.s_bytes() + d.inv_cc.alg_bytes());
    levels_.push_back(std::move(dl));
  }
  // coarse level
  {
    const bool mf = (H.coarse_inv.kind == InvOp::HORNER || H.coarse_inv.kind == InvOp::NEWTON);
    const int64_t nL = H.coarse_A.m;
    DBuf<int32_t> cpd, icpd;
    const int32_t* cph = nullptr;
    if (use_key) {
      std::vector<int32_t> icp = inverse(perm[L - 1]);
      cpd.upload(perm[L - 1].data(), nL); icpd.upload(icp.data(), nL);
      cph = perm[L - 1].data();
    }
    if (mf || coarse_richardson_its_ > 0) {
      upload_mat(H.coarse_A, use_key ? icpd.p : nullptr, use_key ? cpd.p : nullptr, coarse_A_, stream_, ng(dist ? &dist->h_coarse_A : nullptr));
      attach_halo(coarse_A_, dist ? &dist->h_coarse_A : nullptr, cph);
    }
    std::vector<double> dg;
    if (mf && H.coarse_inv.diag_scale) {
      std::vector<double> nat(nL); get_diagonal(H.coarse_A, nat.data());
      dg.resize(nL);
      for (int64_t i = 0; i < nL; i++) dg[cph ? cph[i] : i] = nat[i];
    }
    upload_inv(H.coarse_inv, coarse_inv_, &coarse_A_, dg.empty() ? nullptr : dg.data(), dist ? &dist->h_coarse_M : nullptr,
               use_key ? icpd.p : nullptr, use_key ? cpd.p : nullptr, cph);
    coarse_inv_.n = H.coarse_A.m;
    if (coarse_richardson_its_ > 0) coarse_r_.alloc(H.coarse_A.m);
    cycle_bytes_ += coarse_inv_.alg_bytes();
  }
  xfull_.clear();
  if (full_) {
    xfull_.resize(L);
    for (int l = 1; l < L; l++) xfull_[l].alloc(l < L - 1 ? H.levels[l]->n : H.coarse_A.m);
  }
  cb_.alloc(n_); cx_.alloc(n_);
  b_ptr_.assign(L, nullptr);
  b_sep_.clear(); b_sep_.resize(L);
  b_ptr_[0] = cb_.p;
  for (int l = 1; l < L; l++) {
    if (any_c_) { b_sep_[l].alloc(n_ - off_[l]); b_ptr_[l] = b_sep_[l].p; }
    else b_ptr_[l] = cb_.p + off_[l];
  }
  HIP_CHECK(hipStreamSynchronize(stream_));
  dev_cache_trim();       // the staging blocks
  if (H.opt.print_stats_timings) {
    std::printf(" hierarchy to the device: %.2f s (orderings %.2f s, host -> device copies of the operators %.2f s, conversion to row blocks %.2f s, fused first-sweep operators %.2f s; %lld operators were on the device already)\n",
                now_s() - t_upload0, t_order, g_t_h2d, g_t_convert, g_t_fuse, (long long)g_n_from_device);
    // what the index streams hold (12 bytes per non-zero = 8 of value + 4 of index in the packed / raw forms)
    double nz = 0, ib = 0, vb = 0;
    auto acc = [&](const DevMat& m) { if (m.valid()) { nz += (double)m.nnz; ib += (double)m.idx_bytes; vb += (double)m.val_bytes; } };
    acc(A_perm_); acc(coarse_A_); acc(coarse_inv_.mat);
    for (auto& l : levels_) for (const DevMat* m : {&l->A_ff, &l->A_fc, &l->G, &l->A_cf, &l->A_cc, &l->Z, &l->W, &l->inv_ff.mat, &l->inv_cc.mat, &l->A_full}) acc(*m);
    std::printf(" row blocks on the device: %.3g non-zeros, index regions %.2f bytes per non-zero, values %.2f bytes per non-zero (8 plain; value dictionaries where a block's values repeat), %.1f GB in all\n",
                nz, nz > 0 ? ib / nz : 0.0, nz > 0 ? vb / nz : 0.0, storage_bytes() * 1e-9);
    // how full the blocks are (512 entries = full): rows of 257 ... 511 entries cannot share a block
    std::printf(" non-zeros per row block (A_ff, M_ff, A_fc, Z) by level:");
    for (size_t l = 0; l < levels_.size(); l++) {
      auto f = [](const DevMat& m) { return m.valid() && m.nblocks > 0 ? (double)m.nnz / (double)m.nblocks : 0.0; };
      std::printf(" %zu: %.0f %.0f %.0f %.0f;", l + 1, f(levels_[l]->A_ff), f(levels_[l]->inv_ff.mat), f(levels_[l]->A_fc), f(levels_[l]->Z));
    }
    std::printf(" coarse: %.0f\n", coarse_A_.valid() && coarse_A_.nblocks > 0 ? (double)coarse_A_.nnz / (double)coarse_A_.nblocks : 0.0);
    std::fflush(stdout);
  }
}

double DevPCAIR::storage_bytes() const {
  double b = 0;
  auto sb = [&](const DevMat& m) { if (m.valid()) b += m.stored_bytes(); };
  sb(A_perm_); sb(coarse_A_); sb(coarse_inv_.mat);
  for (auto& l : levels_) { sb(l->A_ff); sb(l->A_fc); sb(l->G); sb(l->A_cf); sb(l->A_cc); sb(l->Z); sb(l->W); sb(l->inv_ff.mat); sb(l->inv_cc.mat); sb(l->A_full); }
  return b;
}

// ---------------------------------------------------------------------------------------
// approximate inverse application: x (+)= M r
// ---------------------------------------------------------------------------------------
void DevPCAIR::apply_inv(DevInv& I, const double* r, double* x, bool add, int lvl, const char* tag) {
  const int64_t n = I.n;
  if (I.kind == InvOp::ASSEMBLED) {
    SpmvEpi e; e.alpha = 1.0; e.out = x;
    if (add) { e.z = x; e.beta = 1.0; }
    spmv(I.mat, r, e, tag, lvl);
    return;
  }
  if (I.kind == InvOp::DIAG) {
    int id = -1;
    if (profile_) { id = stat_id(tag, lvl, I.alg_bytes()); stats_[id].kernel = "k_pmult"; begin_kernel(id); }
    launch_pointwise_mult(n, 1.0, I.d.p, r, 1.0, add ? x : nullptr, x, stream_);
    if (profile_) end_kernel();
    return;
  }
  const DevMat& A = *I.Aop;
  const double* rr = r;
  if (I.diag_scale) {   // MF_VEC_RHS = D^-1 x  (src/Gmres_Poly.F90:1406)
    Aux t(this, "polynomial rhs = D^-1 r", lvl, 24.0 * n, "k_pmult");
    launch_pointwise_mult(n, 1.0, I.dinv.p, r, 0.0, nullptr, I.rhs.p, stream_);
    rr = I.rhs.p;
  }
  if (I.kind == InvOp::HORNER) {
    const std::vector<double>& c = I.coeffs;
    const int m = (int)c.size() - 1;
    std::vector<int> steps;
    for (int k = m - 1; k >= 0; k--) if (c[k] != 0.0) steps.push_back(k);
    if (steps.empty()) { Aux t(this, "polynomial of order 0", lvl, 24.0 * n, "k_axpby"); launch_axpby(n, c[m], rr, 1.0, add ? x : nullptr, x, stream_); return; }
    const double* cur = rr;       // holds y / lead, with `lead` the scalar still to be applied
    double lead = c[m];
    for (size_t si = 0; si < steps.size(); si++) {
      const int k = steps[si];
      const bool last = (si + 1 == steps.size());
      double* out = last ? (add ? I.t3.p : x) : ((cur == I.t1.p) ? I.t2.p : I.t1.p);
      SpmvEpi e; e.out = out;
      // y_new = op(lead * cur) + c_k rr with op = A | D^-1 A | I - D^-1 A   (all linear)
      if (I.neumann) { e.alpha = -lead; e.dscale = I.dinv.p; e.z = cur; e.beta = lead; }
      else { e.alpha = lead; if (I.diag_scale) e.dscale = I.dinv.p; }
      e.w = rr; e.gamma = c[k];
      if (last && add) { e.y2 = x; e.delta = 1.0; e.aux_out = 1; }
      spmv(A, cur, e, tag, lvl);
      cur = out; lead = 1.0;
    }
    return;
  }
  // Newton basis (Loe et al. Alg. 3); y accumulates straight into x
  const std::vector<double>& re = I.re; const std::vector<double>& im = I.im;
  const int nr = (int)re.size();
  if (!add) { Aux ta(this, "Newton apply: x = 0", lvl, 8.0 * n, "k_fill"); launch_fill(n, 0.0, x, stream_); }
  const double* t = rr;
  auto next_t = [&]() { return (t == I.t1.p) ? I.t2.p : I.t1.p; };
  int i = 0;
  while (i <= nr - 2) {
    if (im[i] == 0.0) {
      if (std::fabs(re[i]) < TOL_ZERO) { i++; continue; }
      double* tn = next_t();
      SpmvEpi e; e.out = tn; e.alpha = -1.0 / re[i]; e.z = t; e.beta = 1.0;
      if (I.diag_scale) e.dscale = I.dinv.p;
      e.y2 = x; e.w2 = t; e.delta = 1.0 / re[i]; e.aux_out = 0;
      spmv(A, t, e, tag, lvl);
      t = tn; i++;
    } else {
      const double sq = re[i] * re[i] + im[i] * im[i];
      if (sq < TOL_ZERO) { i += 2; continue; }
      SpmvEpi e; e.out = I.t3.p; e.alpha = -1.0; e.z = t; e.beta = 2.0 * re[i];
      if (I.diag_scale) e.dscale = I.dinv.p;
      e.y2 = x; e.delta = 1.0 / sq; e.aux_out = 1;
      spmv(A, t, e, tag, lvl);
      if (i <= nr - 3) {
        double* tn = next_t();
        SpmvEpi e2; e2.out = tn; e2.alpha = -1.0 / sq; e2.z = t; e2.beta = 1.0;
        if (I.diag_scale) e2.dscale = I.dinv.p;
        spmv(A, I.t3.p, e2, tag, lvl);
        t = tn;
      }
      i += 2;
    }
  }
  if (im[nr - 1] == 0.0 && std::fabs(re[nr - 1]) > TOL_ZERO) {
    Aux ta(this, "Newton apply: last real root", lvl, 24.0 * n, "k_axpby");
    launch_axpby(n, 1.0 / re[nr - 1], t, 1.0, x, x, stream_);
  }
}

// ---------------------------------------------------------------------------------------
// FC relaxation
// ---------------------------------------------------------------------------------------
void DevPCAIR::f_smooths(DevLevel& L, int li, bool first, int its) {
  double* x_f = xvec(li); double* x_c = x_f + L.n_f;
  const double* b_f = bvec(li);
  const bool diag_inv = L.aff_diag && L.inv_ff.kind == InvOp::DIAG;
  const double* x_c_nat = x_c;       // x_c as the level-(l+1) vector: what a general W multiplies
  if (L.xs.p) {
    // x_c in the spatial order of the C points, for A_fc and the one-point prolongator
    prof_begin("x_c to spatial order", li, 20.0 * L.n_c, "k_gather");
    launch_gather(L.n_c, L.xs_src.p, x_c, L.xs.p, stream_);
    prof_end();
    x_c = L.xs.p;
  }
  if (first && L.G.valid() && (g_fuse_first_run || !L.A_fc.valid() || comm_)) {
    // The first sweep on G = A_fc + A_ff W (host/spops.hpp: fused_first_operator): with x0 = W x_c
    //   r0 = b_f - A_fc x_c - A_ff x0 = b_f - G x_c                      (one product instead of two)
    //   sweep k:  d = M r_k ; x += d ; r_{k+1} = r_k - A_ff d            (= t4 - A_ff x_{k+1}, the reference's residual)
    //   (two sweeps: x_2 = x_0 + M (r_0 + r_1), below)
    // - the reference's iterates x_1, x_2, ... (src/FC_Smooth.F90:541-559) with its + 1 + (its - 1) products instead of
    // 2 its + 1; sums re-associated, so the results agree to rounding, not bit for bit.
    if (g_fc_fuse || L.G.halo) {
      int id = -1;
      if (profile_) { id = stat_id("fc_first(G=A_fc+A_ff*W)", li, L.G.s_bytes()); stats_[id].kernel = "k_rb_fc_first"; }
      if (begin_exchange(L.G, x_c)) {
        if (profile_) begin_kernel(id);
        launch_fc_first(L.G, L.wcol.p, b_f, x_c, nullptr, nullptr, L.r.p, x_f, stream_, L.G.halo->recvbuf, RB_INTERIOR);
        wait_exchange();
        launch_fc_first(L.G, L.wcol.p, b_f, x_c, nullptr, nullptr, L.r.p, x_f, stream_, L.G.halo->recvbuf, RB_BOUNDARY);
      } else {
        const double* xg = halo_update(L.G, x_c);
        if (profile_) begin_kernel(id);
        launch_fc_first(L.G, L.wcol.p, b_f, x_c, nullptr, nullptr, L.r.p, x_f, stream_, xg);
      }
      if (profile_) end_kernel();
    } else {
      { Aux t(this, "prolong(one point)", li, 20.0 * L.n_f, "k_gather"); launch_gather(L.n_f, L.wcol.p, x_c, x_f, stream_); }
      SpmvEpi e; e.out = L.r.p; e.alpha = -1.0; e.z = b_f; e.beta = 1.0;
      spmv(L.G, x_c, e, "r=b_f-G*x_c", li);
    }
    if (its == 2) {
      // two sweeps (the default "ff"): x_2 = x_0 + M r_0 + M r_1 = x_0 + M (r_0 + r_1) with r_1 = r_0 - A_ff M r_0 - M is
      // linear -, so the first inverse product needs no update of x riding along and the last one is the plain x += M s
      apply_inv(L.inv_ff, L.r.p, L.t4.p, false, li, "d=M_ff*r");                          // d = M r_0 (assembled: one product; matrix-free: the polynomial)
      SpmvEpi e2; e2.out = L.r.p; e2.alpha = -1.0; e2.z = L.r.p; e2.beta = 2.0;      // s = r_0 + r_1 = 2 r_0 - A_ff d (row i reads and writes r_i only)
      spmv(L.A_ff, L.t4.p, e2, "s=2r-A_ff*d", li);
      apply_inv(L.inv_ff, L.r.p, x_f, true, li, "x_f+=M_ff*s");
      return;
    }
    for (int it = 0; it < its; it++) {
      if (it + 1 == its) { apply_inv(L.inv_ff, L.r.p, x_f, true, li, "x_f+=M_ff*r"); break; }
      if (L.inv_ff.kind == InvOp::ASSEMBLED) {
        SpmvEpi e; e.out = L.t4.p; e.y2 = x_f; e.delta = 1.0; e.aux_out = 1;           // d = M r, x_f += d
        spmv(L.inv_ff.mat, L.r.p, e, "d=M_ff*r,x_f+=d", li);
      } else {
        apply_inv(L.inv_ff, L.r.p, L.t4.p, false, li, "d=M_ff*r");
        Aux t(this, "x_f+=d", li, 24.0 * L.n_f, "k_axpby");
        launch_axpby(L.n_f, 1.0, L.t4.p, 1.0, x_f, x_f, stream_);
      }
      SpmvEpi e2; e2.out = L.r.p; e2.alpha = -1.0; e2.z = L.r.p; e2.beta = 1.0;      // r -= A_ff d (row i reads and writes r_i only)
      spmv(L.A_ff, L.t4.p, e2, "r-=A_ff*d", li);
    }
    return;
  }
  if (first && L.one_point && !g_fc_fuse && !L.A_fc.halo && !(diag_inv && its == 1)) {
    // diagnostic: the two steps apart
    { Aux t(this, "prolong(one point)", li, 20.0 * L.n_f, "k_gather"); launch_gather(L.n_f, L.wcol.p, x_c, x_f, stream_); }
    SpmvEpi e; e.out = L.t4.p; e.alpha = -1.0; e.z = b_f; e.beta = 1.0;
    spmv(L.A_fc, x_c, e, "t4=b_f-A_fc*x_c", li);
  } else if (first && L.one_point) {
    // prolongation x_f = W x_c fused with t4 = b_f - A_fc x_c (and the whole smooth when A_ff
    // and its inverse are diagonal and a single F smooth is requested)
    const bool full = diag_inv && its == 1;
    int id = -1;
    if (profile_) { id = stat_id(full ? "fc_first_diag(A_fc+W)" : "fc_first(A_fc+W)", li, L.A_fc.s_bytes()); stats_[id].kernel = "k_rb_fc_first"; }
    const double* ad = full ? L.aff_d.p : nullptr; const double* mi = full ? L.inv_ff.d.p : nullptr;
    if (begin_exchange(L.A_fc, x_c)) {
      if (profile_) begin_kernel(id);
      launch_fc_first(L.A_fc, L.wcol.p, b_f, x_c, ad, mi, L.t4.p, x_f, stream_, L.A_fc.halo->recvbuf, RB_INTERIOR);
      wait_exchange();
      launch_fc_first(L.A_fc, L.wcol.p, b_f, x_c, ad, mi, L.t4.p, x_f, stream_, L.A_fc.halo->recvbuf, RB_BOUNDARY);
    } else {
      const double* xg = halo_update(L.A_fc, x_c);
      if (profile_) begin_kernel(id);
      launch_fc_first(L.A_fc, L.wcol.p, b_f, x_c, ad, mi, L.t4.p, x_f, stream_, xg);
    }
    if (profile_) end_kernel();
    if (full) return;
  } else {
    if (first) { SpmvEpi e; e.out = x_f; spmv(L.W, x_c_nat, e, "prolong(W)", li); }
    SpmvEpi e; e.out = L.t4.p; e.alpha = -1.0; e.z = b_f; e.beta = 1.0;
    spmv(L.A_fc, x_c, e, "t4=b_f-A_fc*x_c", li);
  }
  for (int it = 0; it < its; it++) {
    if (diag_inv) {
      Aux t(this, "x_f+=M(t4-D*x_f) diagonal", li, 40.0 * L.n_f, "k_diag_resid_update");
      launch_diag_residual_update(L.n_f, L.aff_d.p, L.inv_ff.d.p, L.t4.p, x_f, stream_);
      continue;
    }
    if (L.aff_diag) { Aux t(this, "r=t4-D*x_f", li, 32.0 * L.n_f, "k_pmult"); launch_pointwise_mult(L.n_f, -1.0, L.aff_d.p, x_f, 1.0, L.t4.p, L.r.p, stream_); }
    else {
      SpmvEpi e; e.out = L.r.p; e.alpha = -1.0; e.z = L.t4.p; e.beta = 1.0;
      spmv(L.A_ff, x_f, e, "r=t4-A_ff*x_f", li);
    }
    apply_inv(L.inv_ff, L.r.p, x_f, true, li, "x_f+=M_ff*r");
  }
}

void DevPCAIR::c_smooths(DevLevel& L, int li, bool first, int its) {
  double* x_f = xvec(li); double* x_c = x_f + L.n_f;
  const double* b_c = bvec(li) + L.n_f;
  if (first) {   // prolongation has not been applied yet
    if (L.one_point) {
      if (L.A_fc.halo) throw std::runtime_error("a distributed cycle that starts with C smoothing needs a general W");
      const double* xw = x_c;
      if (L.xs.p) { Aux t(this, "x_c to spatial order", li, 20.0 * L.n_c, "k_gather"); launch_gather(L.n_c, L.xs_src.p, x_c, L.xs.p, stream_); xw = L.xs.p; }   // wcol counts in the spatial order
      Aux t(this, "prolong(one point)", li, 20.0 * L.n_f, "k_gather");
      launch_gather(L.n_f, L.wcol.p, xw, x_f, stream_);   // wcol < 0 -> 0 (no W entry)
    }
    else { SpmvEpi e; e.out = x_f; spmv(L.W, x_c, e, "prolong(W)", li); }
  }
  SpmvEpi e; e.out = L.t4c.p; e.alpha = -1.0; e.z = b_c; e.beta = 1.0;
  spmv(L.A_cf, x_f, e, "t4c=b_c-A_cf*x_f", li);
  for (int it = 0; it < its; it++) {
    SpmvEpi e2; e2.out = L.rc.p; e2.alpha = -1.0; e2.z = L.t4c.p; e2.beta = 1.0;
    spmv(L.A_cc, x_c, e2, "rc=t4c-A_cc*x_c", li);
    apply_inv(L.inv_cc, L.rc.p, x_c, true, li, "x_c+=M_cc*rc");
  }
}

// x_L = M_L b_L (KSPPREONLY) or N Richardson steps (-mg_coarse_ksp_type richardson)
void DevPCAIR::coarse_solve(const double* bL, double* xL) {
  const int L = no_levels();
  if (coarse_richardson_its_ <= 0) { apply_inv(coarse_inv_, bL, xL, false, L - 1, "coarse x=M*b"); return; }
  { Aux t(this, "coarse x = 0", L - 1, 8.0 * coarse_inv_.n, "k_fill"); launch_fill(coarse_inv_.n, 0.0, xL, stream_); }
  for (int it = 0; it < coarse_richardson_its_; it++) {
    const double* r = bL;
    if (it > 0) {
      SpmvEpi e; e.out = coarse_r_.p; e.alpha = -1.0; e.z = bL; e.beta = 1.0;
      spmv(coarse_A_, xL, e, "coarse r=b-A*x", L - 1);
      r = coarse_r_.p;
    }
    apply_inv(coarse_inv_, r, xL, true, L - 1, "coarse x+=M*r");
  }
}

// -pc_air_full_smoothing_up_and_down: PCMG multiplicative V(1,1) cycle, Richardson(1) + PCMAT(inv_A) on all points
// (src/AIR_MG_Setup.F90:1014-1079).  Level-l vectors are laid out [F block | level l+1 ...], so R = [Z I] turns the
// C part of the level's residual into the next right-hand side in place, and P = [W; I] adds the coarse solution
// to the C part and W e to the F part.
void DevPCAIR::vcycle_full() {
  const int L = no_levels();
  auto X = [&](int l) { return l == 0 ? xbase_ : xfull_[l].p; };
  auto B = [&](int l) -> const double* { return l == 0 ? b0_ : levels_[l - 1]->rfull.p + levels_[l - 1]->n_f; };
  for (int l = 0; l + 1 < L; l++) {
    DevLevel& Lv = *levels_[l];
    apply_inv(Lv.inv_ff, B(l), X(l), false, l, "down x=M*b");                       // zero initial guess
    SpmvEpi e; e.out = Lv.rfull.p; e.alpha = -1.0; e.z = B(l); e.beta = 1.0;
    spmv(Lv.A_full, X(l), e, "down r=b-A*x", l);
    SpmvEpi e2; e2.out = Lv.rfull.p + Lv.n_f; e2.alpha = 1.0; e2.z = Lv.rfull.p + Lv.n_f; e2.beta = 1.0;
    spmv(Lv.Z, Lv.rfull.p, e2, "restrict(r_c+=Z*r_f)", l);
  }
  coarse_solve(B(L - 1), X(L - 1));
  for (int l = L - 2; l >= 0; l--) {
    DevLevel& Lv = *levels_[l];
    { Aux t(this, "x_c+=e", l, 24.0 * Lv.n_c, "k_axpby"); launch_axpby(Lv.n_c, 1.0, X(l + 1), 1.0, X(l) + Lv.n_f, X(l) + Lv.n_f, stream_); }   // x_c += e
    SpmvEpi e; e.out = X(l); e.alpha = 1.0; e.z = X(l); e.beta = 1.0;
    spmv(Lv.W, X(l + 1), e, "prolong(x_f+=W*e)", l);
    SpmvEpi e2; e2.out = Lv.rfull.p; e2.alpha = -1.0; e2.z = B(l); e2.beta = 1.0;
    spmv(Lv.A_full, X(l), e2, "up r=b-A*x", l);
    apply_inv(Lv.inv_ff, Lv.rfull.p, X(l), true, l, "up x+=M*r");
  }
}

void DevPCAIR::vcycle() {
  dev_stage("solve: V-cycle");
  if (full_) { vcycle_full(); return; }
  const int L = no_levels();
  // DOWN: b_{l+1} = b_c + Z b_f (identity block of R=[Z I] is the in-place b_c)
  for (int l = 0; l + 1 < L; l++) {
    DevLevel& Lv = *levels_[l];
    SpmvEpi e; e.out = bvec(l + 1); e.alpha = 1.0; e.z = bvec(l) + Lv.n_f; e.beta = 1.0;
    spmv(Lv.Z, bvec(l), e, "restrict(b_c+=Z*b_f)", l);
  }
  coarse_solve(bvec(L - 1), xvec(L - 1));
  // UP: x_l = P x_{l+1} (x_c is in place), then FC relaxation
  for (int l = L - 2; l >= 0; l--) {
    DevLevel& Lv = *levels_[l];
    bool first = true;
    for (int i = 0; i < 10; i++) {
      int s = Lv.smooth_order[i];
      if (s == 0) break;
      if (s > 0) f_smooths(Lv, l, first, s); else c_smooths(Lv, l, first, -s);
      first = false;
    }
  }
}

// hipGraph capture of the cycle: single device, or a communicator whose exchange only enqueues device work (RCCL send /
// receive groups are capturable; the exchange stream joins the capture through the pack / receive events).  The
// host-staged transport synchronises the host inside the cycle and cannot be captured.
bool DevPCAIR::graph_ok() const { return want_graph_ && !profile_ && (!comm_ || comm_->device_side()); }

void DevPCAIR::run_cycle() {
  if (graph_ok()) {
    if (!graph_exec_) {
      hipGraph_t g = nullptr;
      // Found in round 4 with a captured PCApply right after setup(): the copies and frees of the library's own host-mirror
      // thread invalidate a capture in progress on this stream ("operation failed due to a previous error during capture"),
      // in thread-local and in relaxed mode alike.  So the owner first waits for that thread (before_capture_); relaxed mode
      // stays for the application's other threads.
      if (before_capture_) before_capture_();
      HIP_CHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeRelaxed));
      try { vcycle(); }
      catch (...) {
        // leave the stream usable: end (and discard) the broken capture, run without a graph from here on
        (void)hipStreamEndCapture(stream_, &g);
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        want_graph_ = false;
        throw;
      }
      HIP_CHECK(hipStreamEndCapture(stream_, &g));
      HIP_CHECK(hipGraphInstantiate(&graph_exec_, g, nullptr, nullptr, 0));
      HIP_CHECK(hipGraphDestroy(g));
    }
    HIP_CHECK(hipGraphLaunch(graph_exec_, stream_));
  } else vcycle();
}

void DevPCAIR::apply_permuted(const double* r, double* z) {
  if (r != z && !graph_ok()) {
    b0_ = r; xbase_ = z;          // r is preserved, z receives the cycle's output directly
    run_cycle();
    return;
  }
  // captured graph (fixed addresses) or aliased arguments: go through the internal buffers
  b0_ = cb_.p; xbase_ = cx_.p;
  HIP_CHECK(hipMemcpyAsync(cb_.p, r, sizeof(double) * (size_t)n_, hipMemcpyDeviceToDevice, stream_));
  run_cycle();
  HIP_CHECK(hipMemcpyAsync(z, cx_.p, sizeof(double) * (size_t)n_, hipMemcpyDeviceToDevice, stream_));
}

void DevPCAIR::to_permuted(const double* nat, double* perm) {
  if (perm_nchunks_ > 0 && g_perm_chunks) launch_permute_chunks(perm_nchunks_, perm_nsec_, chunk_starts_.p, iperm_.p, nat, perm, false, stream_);
  else launch_gather(n_, iperm_.p, nat, perm, stream_);
}
void DevPCAIR::to_natural(const double* perm, double* nat) {
  if (perm_nchunks_ > 0 && g_perm_chunks) launch_permute_chunks(perm_nchunks_, perm_nsec_, chunk_starts_.p, iperm_.p, perm, nat, true, stream_);
  else launch_gather(n_, perm_.p, perm, nat, stream_);
}

void DevPCAIR::apply_natural(const double* d_x, double* d_y) {
  b0_ = cb_.p; xbase_ = cx_.p;
  to_permuted(d_x, cb_.p);
  run_cycle();
  to_natural(cx_.p, d_y);
}

}  // namespace airhip
