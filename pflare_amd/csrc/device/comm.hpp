// Neighbour exchange + tiny all-reduce between the ranks of a row-block distributed hierarchy
// (one process per GPU).  Two transports:
//  * RcclComm  - RCCL send/recv groups and ncclAllReduce on the library stream, device buffers end
//                to end over xGMI (production path; librccl is loaded at run time so the library
//                has no link-time dependency and shares the copy torch.distributed already loaded);
//  * StagedComm - buffers staged through pinned host memory and handed to caller-supplied
//                callbacks (MPI in a host application, torch.distributed/gloo in the tests, where
//                several ranks may share one GPU, which RCCL refuses).
#pragma once
#include "dev.hpp"
#include <memory>
#include <vector>

namespace airhip {

// Halo plan of one distributed operand on the device.  Buffers are laid out peer after peer.
struct HaloPlan {
  std::vector<int> peers;
  std::vector<int64_t> send_off, recv_off;    // [npeers + 1]
  int32_t* send_idx = nullptr;                // device: positions inside the operand's input vector
  double* sendbuf = nullptr;                  // device [n_send]
  double* recvbuf = nullptr;                  // device [n_ghost]
  int64_t n_send = 0, n_ghost = 0;
  ~HaloPlan();
};

struct Comm {
  int rank = 0, size = 1;
  virtual ~Comm() = default;
  // true: exchange() only enqueues device work (no host synchronisation, no host callback) - it can run on a second
  // stream beside the interior row blocks and be captured in a hipGraph.  RCCL: yes; the host-staged transport: no.
  virtual bool device_side() const { return false; }
  // sendbuf -> peers, peers -> recvbuf; ordered after prior work on s, complete for later work on s
  virtual void exchange(const HaloPlan& p, hipStream_t s) = 0;
  // sum over ranks of n doubles at d_vals (in place); the result is also returned in h_vals and the
  // stream is synchronised
  virtual void allreduce_sum(double* d_vals, double* h_vals, int n, hipStream_t s) = 0;
  int64_t n_exchanges = 0, n_allreduces = 0;
  int64_t bytes_sent = 0, messages_sent = 0;     // halo traffic this rank has sent (8 bytes per ghost value)
};

typedef int (*exchange_cb)(void* user, int npeers, const int* peers, const double* send, const int64_t* send_off,
                           double* recv, const int64_t* recv_off);
typedef int (*allreduce_cb)(void* user, double* vals, int n);

std::unique_ptr<Comm> make_staged_comm(int rank, int size, exchange_cb ex, allreduce_cb ar, void* user);
// unique_id: 128 bytes from rccl_unique_id() on rank 0, broadcast by the caller
std::unique_ptr<Comm> make_rccl_comm(int rank, int size, const void* unique_id);
void rccl_unique_id(void* out128);
// collective: all-pairs exchange of `count` doubles with a known pattern + an all-reduce, verified
bool comm_selftest(Comm& c, int64_t count);

}  // namespace airhip
