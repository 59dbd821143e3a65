// Row-block distribution of a PCAIR hierarchy over the GPUs of a node.
//
// The reference distributes every level the PETSc MPIAIJ way: a rank owns a set of rows, coarse
// rows live with their parent C-points, and an operator is stored per rank as its owned rows with
// the columns split into owned and off-process ("ghost") ones that are fetched from the
// neighbours before each product (src/PMISR_Module.F90:174-180, src/Grid_Transfer.F90:656-668).
// Here the hierarchy is built once (globally, deterministic), then cut: for every rank the
// owned rows of every operand with columns renumbered to [local | ghost], plus the halo plan
// (who sends which of its entries to whom).  Pieces travel between processes as an Archive of
// named flat arrays.
#pragma once
#include "air.hpp"
#include <map>
#include <string>

namespace airhip {

// Halo plan of one distributed operand, in the numbering of its column space on this rank
// (level-local natural numbering; the device upload maps it to the hierarchical layout).
struct HaloHost {
  std::vector<int32_t> peers;                 // ascending
  std::vector<int64_t> send_off, recv_off;    // [npeers + 1]
  Arr<int32_t> send_idx;                      // [send_off.back()] local indices to pack, per peer
  int64_t n_ghost = 0;                        // = recv_off.back(); ghosts ordered by (owner, global id)
  bool active() const { return !peers.empty(); }
};

struct LevelHalos { HaloHost A_ff, A_fc, A_cf, A_cc, Z, W, M_ff, M_cc, G; };

// The piece of a hierarchy one rank owns.  Matrices have n = n_local_cols + n_ghost columns.
struct LocalHierarchy {
  Hierarchy H;                                // local rows of every operand
  Csr A_top;                                  // local rows of the fine operator
  std::vector<LevelHalos> halos;              // per level
  HaloHost h_top, h_coarse_A, h_coarse_M;
  int rank = 0, nranks = 1;
  int64_t n_global = 0;
  Arr<int32_t> gid;                           // global (natural) index of each local fine row, ascending
  Arr<int64_t> hint;                          // optional locality keys of the local fine rows
};

// Named flat arrays (the wire format of a LocalHierarchy)
struct Archive {
  struct Item { std::string name; int dtype; int64_t count; const void* ptr; std::shared_ptr<void> own; };
  std::vector<Item> items;
  std::map<std::string, int> index;
  enum { I32 = 0, I64 = 1, F64 = 2 };
  void add_copy(const std::string& name, int dtype, const void* p, int64_t count);
  void add_view(const std::string& name, int dtype, const void* p, int64_t count);
  const Item* find(const std::string& name) const;
};

// Cuts a global hierarchy.  owner[i] in [0, nranks) for every fine row i.
class Partitioner {
 public:
  Partitioner(const Hierarchy& G, const Csr& A_top, int nranks, const int32_t* owner, const int64_t* hint = nullptr);
  ~Partitioner();
  void extract(int rank, Archive& out) const;
 private:
  struct Impl;
  std::unique_ptr<Impl> impl_;
};

// Rebuilds the rank's piece from an archive (arrays are viewed, not copied: the archive must
// outlive the result).
void load_local(const Archive& ar, LocalHierarchy& out);

}  // namespace airhip
