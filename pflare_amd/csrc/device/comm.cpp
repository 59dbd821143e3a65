// Transports of the halo exchange; see comm.hpp.
#include "comm.hpp"
#include <dlfcn.h>
#include <cstring>
#include <stdexcept>

namespace airhip {

HaloPlan::~HaloPlan() {
  if (send_idx) (void)hipFree(send_idx);
  if (sendbuf) (void)hipFree(sendbuf);
  if (recvbuf) (void)hipFree(recvbuf);
}

namespace {

// ---------------------------------------------------------------------------------------
// host-staged transport
// ---------------------------------------------------------------------------------------
struct StagedComm final : Comm {
  exchange_cb ex = nullptr; allreduce_cb ar = nullptr; void* user = nullptr;
  double* hs = nullptr; double* hr = nullptr; int64_t cap_s = 0, cap_r = 0;   // pinned staging
  ~StagedComm() override { if (hs) (void)hipHostFree(hs); if (hr) (void)hipHostFree(hr); }
  void ensure(int64_t ns, int64_t nr) {
    if (ns > cap_s || !hs) { if (hs) (void)hipHostFree(hs); cap_s = ns + ns / 2 + 64; HIP_CHECK(hipHostMalloc(&hs, sizeof(double) * (size_t)cap_s)); }
    if (nr > cap_r || !hr) { if (hr) (void)hipHostFree(hr); cap_r = nr + nr / 2 + 64; HIP_CHECK(hipHostMalloc(&hr, sizeof(double) * (size_t)cap_r)); }
  }
  void exchange(const HaloPlan& p, hipStream_t s) override {
    n_exchanges++;
    bytes_sent += 8 * p.n_send;
    for (size_t k = 0; k + 1 < p.send_off.size(); k++) messages_sent += p.send_off[k + 1] > p.send_off[k] ? 1 : 0;
    ensure(p.n_send, p.n_ghost);
    if (p.n_send) HIP_CHECK(hipMemcpyAsync(hs, p.sendbuf, sizeof(double) * (size_t)p.n_send, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (ex(user, (int)p.peers.size(), p.peers.data(), hs, p.send_off.data(), hr, p.recv_off.data()) != 0)
      throw std::runtime_error("halo exchange callback failed");
    if (p.n_ghost) HIP_CHECK(hipMemcpyAsync(p.recvbuf, hr, sizeof(double) * (size_t)p.n_ghost, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));   // hr is reused by the next exchange
  }
  void allreduce_sum(double* d_vals, double* h_vals, int n, hipStream_t s) override {
    n_allreduces++;
    HIP_CHECK(hipMemcpyAsync(h_vals, d_vals, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    if (ar(user, h_vals, n) != 0) throw std::runtime_error("all-reduce callback failed");
    HIP_CHECK(hipMemcpyAsync(d_vals, h_vals, sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }
};

// ---------------------------------------------------------------------------------------
// RCCL transport (symbols resolved at run time)
// ---------------------------------------------------------------------------------------
struct UniqueId { char internal[128]; };
struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(UniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, UniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
constexpr int NCCL_DOUBLE = 8, NCCL_SUM = 0;

Rccl& rccl() {
  static Rccl r;
  if (r.lib) return r;
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  r.lib = dlopen(names[0], RTLD_NOW | RTLD_NOLOAD);          // the copy torch.distributed already loaded
  if (!r.lib) r.lib = dlopen(names[1], RTLD_NOW | RTLD_NOLOAD);
  for (int i = 0; i < 3 && !r.lib; i++) r.lib = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
  if (!r.lib) throw std::runtime_error(std::string("cannot load librccl: ") + dlerror());
  auto sym = [&](const char* n) { void* p = dlsym(r.lib, n); if (!p) throw std::runtime_error(std::string("librccl lacks ") + n); return p; };
  r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
  r.Send = (decltype(r.Send))sym("ncclSend");
  r.Recv = (decltype(r.Recv))sym("ncclRecv");
  r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
  r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  return r;
}
#define RCCL_CHECK(expr)                                                                            \
  do {                                                                                              \
    int _r = (expr);                                                                                \
    if (_r != 0) throw std::runtime_error(std::string("RCCL error: ") + rccl().GetErrorString(_r)); \
  } while (0)

struct RcclComm final : Comm {
  void* comm = nullptr;
  bool device_side() const override { return true; }
  ~RcclComm() override { if (comm) (void)rccl().CommDestroy(comm); }
  void exchange(const HaloPlan& p, hipStream_t s) override {
    n_exchanges++;
    bytes_sent += 8 * p.n_send;
    for (size_t k = 0; k + 1 < p.send_off.size(); k++) messages_sent += p.send_off[k + 1] > p.send_off[k] ? 1 : 0;
    Rccl& r = rccl();
    RCCL_CHECK(r.GroupStart());
    for (size_t k = 0; k < p.peers.size(); k++) {
      const int64_t ns = p.send_off[k + 1] - p.send_off[k], nr = p.recv_off[k + 1] - p.recv_off[k];
      if (ns) RCCL_CHECK(r.Send(p.sendbuf + p.send_off[k], (size_t)ns, NCCL_DOUBLE, p.peers[k], comm, s));
      if (nr) RCCL_CHECK(r.Recv(p.recvbuf + p.recv_off[k], (size_t)nr, NCCL_DOUBLE, p.peers[k], comm, s));
    }
    RCCL_CHECK(r.GroupEnd());
  }
  void allreduce_sum(double* d_vals, double* h_vals, int n, hipStream_t s) override {
    n_allreduces++;
    RCCL_CHECK(rccl().AllReduce(d_vals, d_vals, (size_t)n, NCCL_DOUBLE, NCCL_SUM, comm, s));
    HIP_CHECK(hipMemcpyAsync(h_vals, d_vals, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
  }
};

}  // namespace

std::unique_ptr<Comm> make_staged_comm(int rank, int size, exchange_cb ex, allreduce_cb ar, void* user) {
  if (!ex || !ar) throw std::runtime_error("staged comm needs both callbacks");
  auto c = std::make_unique<StagedComm>();
  c->rank = rank; c->size = size; c->ex = ex; c->ar = ar; c->user = user;
  return c;
}

void rccl_unique_id(void* out128) {
  UniqueId id;
  RCCL_CHECK(rccl().GetUniqueId(&id));
  std::memcpy(out128, &id, sizeof id);
}

std::unique_ptr<Comm> make_rccl_comm(int rank, int size, const void* unique_id) {
  auto c = std::make_unique<RcclComm>();
  c->rank = rank; c->size = size;
  UniqueId id; std::memcpy(&id, unique_id, sizeof id);
  RCCL_CHECK(rccl().CommInitRank(&c->comm, size, id, rank));
  return c;
}

// Every rank sends a distinct pattern to every other rank and checks what arrives, then checks an all-reduce.
bool comm_selftest(Comm& c, int64_t count) {
  if (count < 1) count = 1;
  HaloPlan p;
  for (int r = 0; r < c.size; r++) if (r != c.rank) p.peers.push_back(r);
  const int64_t np = (int64_t)p.peers.size();
  p.send_off.resize(np + 1); p.recv_off.resize(np + 1);
  for (int64_t k = 0; k <= np; k++) p.send_off[k] = p.recv_off[k] = k * count;
  p.n_send = p.n_ghost = np * count;
  hipStream_t s;
  HIP_CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto pattern = [](int from, int to, int64_t i) { return (double)from * 4096.0 + (double)to + (double)(i % 1000) * 1e-3; };
  std::vector<double> hs((size_t)std::max<int64_t>(1, p.n_send)), hr((size_t)std::max<int64_t>(1, p.n_ghost), -1.0);
  for (int64_t k = 0; k < np; k++)
    for (int64_t i = 0; i < count; i++) hs[k * count + i] = pattern(c.rank, p.peers[k], i);
  HIP_CHECK(hipMalloc(&p.sendbuf, hs.size() * sizeof(double)));
  HIP_CHECK(hipMalloc(&p.recvbuf, hr.size() * sizeof(double)));
  HIP_CHECK(hipMemcpyAsync(p.sendbuf, hs.data(), hs.size() * sizeof(double), hipMemcpyHostToDevice, s));
  HIP_CHECK(hipMemcpyAsync(p.recvbuf, hr.data(), hr.size() * sizeof(double), hipMemcpyHostToDevice, s));
  bool ok = true;
  for (int rep = 0; rep < 2 && np > 0; rep++) {          // twice: buffers are reused back to back in the cycle
    c.exchange(p, s);
    HIP_CHECK(hipMemcpyAsync(hr.data(), p.recvbuf, hr.size() * sizeof(double), hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
    for (int64_t k = 0; k < np; k++)
      for (int64_t i = 0; i < count; i++) ok = ok && hr[k * count + i] == pattern(p.peers[k], c.rank, i);
  }
  double* d = nullptr; double h[2] = {0, 0};
  HIP_CHECK(hipMalloc(&d, 2 * sizeof(double)));
  const double mine[2] = {(double)(c.rank + 1), 1.0};
  HIP_CHECK(hipMemcpyAsync(d, mine, sizeof mine, hipMemcpyHostToDevice, s));
  c.allreduce_sum(d, h, 2, s);
  ok = ok && h[0] == 0.5 * c.size * (c.size + 1) && h[1] == (double)c.size;
  (void)hipFree(d);
  HIP_CHECK(hipStreamSynchronize(s));
  (void)hipStreamDestroy(s);
  c.n_exchanges = c.n_allreduces = 0;
  c.bytes_sent = c.messages_sent = 0;
  return ok;
}

}  // namespace airhip
