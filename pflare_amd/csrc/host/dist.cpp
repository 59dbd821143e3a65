// Partitioning of a global PCAIR hierarchy into per-rank pieces with halo plans; see dist.hpp.
#include "dist.hpp"

namespace airhip {

// ---------------------------------------------------------------------------------------
// Archive
// ---------------------------------------------------------------------------------------
static size_t dtype_size(int dt) { return dt == Archive::I32 ? 4 : 8; }

void Archive::add_copy(const std::string& name, int dtype, const void* p, int64_t count) {
  Item it; it.name = name; it.dtype = dtype; it.count = count;
  const size_t bytes = dtype_size(dtype) * (size_t)std::max<int64_t>(count, 0);
  void* mem = std::malloc(bytes ? bytes : 1);
  if (!mem) throw std::bad_alloc();
  if (bytes) std::memcpy(mem, p, bytes);
  it.own = std::shared_ptr<void>(mem, std::free);
  it.ptr = mem;
  index[name] = (int)items.size();
  items.push_back(std::move(it));
}
void Archive::add_view(const std::string& name, int dtype, const void* p, int64_t count) {
  Item it; it.name = name; it.dtype = dtype; it.count = count; it.ptr = p;
  index[name] = (int)items.size();
  items.push_back(std::move(it));
}
const Archive::Item* Archive::find(const std::string& name) const {
  auto it = index.find(name);
  return it == index.end() ? nullptr : &items[it->second];
}

namespace {

// A row or column index space of some level: the full level, its F points or its C points.
struct Space {
  int64_t n = 0;
  std::vector<int32_t> own, loc;                 // owner rank and index within the owner's part
  std::vector<std::vector<int32_t>> members;     // per rank: global ids, ascending
  int64_t count(int r) const { return (int64_t)members[r].size(); }
};

void finish_space(Space& S, int nranks) {
  S.n = (int64_t)S.own.size();
  S.loc.resize(S.n);
  S.members.assign(nranks, {});
  std::vector<int64_t> cnt(nranks, 0);
  for (int64_t i = 0; i < S.n; i++) cnt[S.own[i]]++;
  for (int r = 0; r < nranks; r++) S.members[r].reserve(cnt[r]);
  for (int64_t i = 0; i < S.n; i++) {
    auto& m = S.members[S.own[i]];
    S.loc[i] = (int32_t)m.size();
    m.push_back((int32_t)i);
  }
}

// need[p * N + q]: sorted unique global column ids owned by p that rank q reads
struct OpHalo { std::vector<std::vector<int32_t>> need; };

OpHalo compute_halo(const Csr& X, const Space& R, const Space& C, int N) {
  OpHalo out; out.need.assign((size_t)N * N, {});
  if (N == 1 || X.empty()) return out;
  const int nt = omp_get_max_threads();
  std::vector<std::vector<uint64_t>> keys(nt);
#pragma omp parallel num_threads(nt)
  {
    auto& k = keys[omp_get_thread_num()];
#pragma omp for schedule(static)
    for (int64_t i = 0; i < X.m; i++) {
      const uint64_t q = (uint64_t)R.own[i];
      for (int64_t j = X.rp[i]; j < X.rp[i + 1]; j++) {
        const int32_t c = X.ci[j];
        const uint64_t p = (uint64_t)C.own[c];
        if (p != q) k.push_back(((p * (uint64_t)N + q) << 32) | (uint32_t)c);
      }
    }
    std::sort(k.begin(), k.end());
    k.erase(std::unique(k.begin(), k.end()), k.end());
  }
  std::vector<uint64_t> all;
  size_t tot = 0; for (auto& k : keys) tot += k.size();
  all.reserve(tot);
  for (auto& k : keys) { all.insert(all.end(), k.begin(), k.end()); std::vector<uint64_t>().swap(k); }
  std::sort(all.begin(), all.end());
  all.erase(std::unique(all.begin(), all.end()), all.end());
  for (uint64_t key : all) out.need[(size_t)(key >> 32)].push_back((int32_t)(key & 0xffffffffu));
  return out;
}

struct OpRef { const Csr* X; const Space* R; const Space* C; OpHalo halo; };

// hands the array's memory over to the archive
template <class T>
void put_take(Archive& ar, const std::string& name, int dtype, Arr<T>& a, int64_t count) {
  if (!a.own || !a.p) { ar.add_copy(name, dtype, a.p, count); return; }
  Archive::Item it; it.name = name; it.dtype = dtype; it.count = count; it.ptr = a.p;
  it.own = std::shared_ptr<void>((void*)a.p, free_host_array);
  a.p = nullptr; a.n = 0;
  ar.index[name] = (int)ar.items.size();
  ar.items.push_back(std::move(it));
}
void put_csr(Archive& ar, const std::string& name, Csr& A, int64_t n_ghost) {
  int64_t shape[3] = {A.m, A.n, n_ghost};
  const int64_t nnz = A.nnz();
  ar.add_copy(name + ".shape", Archive::I64, shape, 3);
  put_take(ar, name + ".rp", Archive::I64, A.rp, A.m + 1);
  put_take(ar, name + ".ci", Archive::I32, A.ci, nnz);
  put_take(ar, name + ".v", Archive::F64, A.v, nnz);
}
void put_halo(Archive& ar, const std::string& name, const HaloHost& h) {
  ar.add_copy(name + ".halo.peers", Archive::I32, h.peers.data(), (int64_t)h.peers.size());
  ar.add_copy(name + ".halo.send_off", Archive::I64, h.send_off.data(), (int64_t)h.send_off.size());
  ar.add_copy(name + ".halo.recv_off", Archive::I64, h.recv_off.data(), (int64_t)h.recv_off.size());
  ar.add_copy(name + ".halo.send_idx", Archive::I32, h.send_idx.data(), h.send_idx.size());
}

// Owned rows of X for `rank`, columns renumbered to [local | ghost]; gmap (size >= C.n, all -1 on
// entry and on exit) is scratch; ghost_of (optional) receives the ghost index of selected
// global columns through the callback while gmap is populated.
template <class WithMap>
void extract_op(const OpRef& op, int rank, int N, std::vector<int32_t>& gmap, Csr& out, HaloHost& halo, WithMap with_map) {
  const Csr& X = *op.X; const Space& R = *op.R; const Space& C = *op.C;
  const std::vector<int32_t>& rows = R.members[rank];
  const int64_t m = (int64_t)rows.size();
  const int64_t nlc = C.count(rank);
  // ghost numbering: by owner, then by global id
  halo = HaloHost();
  int64_t ng = 0;
  std::vector<int64_t> nrecv(N, 0), nsend(N, 0);
  for (int p = 0; p < N; p++) {
    if (p == rank) continue;
    const auto& lst = op.halo.need[(size_t)p * N + rank];
    nrecv[p] = (int64_t)lst.size();
    for (int32_t c : lst) gmap[c] = (int32_t)ng++;
    nsend[p] = (int64_t)op.halo.need[(size_t)rank * N + p].size();
  }
  halo.n_ghost = ng;
  halo.send_off.push_back(0); halo.recv_off.push_back(0);
  int64_t ns = 0;
  for (int p = 0; p < N; p++) {
    if (nrecv[p] == 0 && nsend[p] == 0) continue;
    halo.peers.push_back(p);
    ns += nsend[p];
    halo.send_off.push_back(ns);
    halo.recv_off.push_back(halo.recv_off.back() + nrecv[p]);
  }
  halo.send_idx.alloc(ns);
  {
    int64_t o = 0;
    for (int p : halo.peers) for (int32_t c : op.halo.need[(size_t)rank * N + p]) halo.send_idx[o++] = C.loc[c];
  }
  // local CSR
  out = Csr(); out.m = m; out.n = nlc + ng;
  out.rp.alloc(m + 1);
  {
    Arr<int64_t> len(m);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < m; i++) len[i] = X.rp[rows[i] + 1] - X.rp[rows[i]];
    exclusive_scan(len.data(), m, out.rp.data());
  }
  out.ci.alloc(out.rp[m]); out.v.alloc(out.rp[m]);
#pragma omp parallel
  {
    std::vector<std::pair<int32_t, double>> gh;
#pragma omp for schedule(static)
    for (int64_t i = 0; i < m; i++) {
      const int64_t g = rows[i];
      int64_t o = out.rp[i];
      gh.clear();
      for (int64_t j = X.rp[g]; j < X.rp[g + 1]; j++) {
        const int32_t c = X.ci[j];
        if (C.own[c] == rank) { out.ci[o] = C.loc[c]; out.v[o] = X.v[j]; o++; }
        else gh.emplace_back((int32_t)(nlc + gmap[c]), X.v[j]);
      }
      if (!gh.empty()) {
        std::sort(gh.begin(), gh.end(), [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& b) { return a.first < b.first; });
        for (auto& e : gh) { out.ci[o] = e.first; out.v[o] = e.second; o++; }
      }
    }
  }
  with_map(nlc);
  for (int p = 0; p < N; p++) if (p != rank) for (int32_t c : op.halo.need[(size_t)p * N + rank]) gmap[c] = -1;
}

void put_inv_meta(Archive& ar, const std::string& name, const InvOp& I) {
  int64_t meta[4] = {(int64_t)I.kind, I.diag_scale ? 1 : 0, I.neumann ? 1 : 0, I.spmvs};
  ar.add_copy(name + ".meta", Archive::I64, meta, 4);
  ar.add_copy(name + ".coeffs", Archive::F64, I.coeffs.data(), (int64_t)I.coeffs.size());
  ar.add_copy(name + ".re", Archive::F64, I.re.data(), (int64_t)I.re.size());
  ar.add_copy(name + ".im", Archive::F64, I.im.data(), (int64_t)I.im.size());
}

}  // namespace

// ---------------------------------------------------------------------------------------
struct Partitioner::Impl {
  const Hierarchy* G = nullptr;
  const Csr* A_top = nullptr;
  const int64_t* hint = nullptr;
  int N = 1;
  std::vector<Space> full, fsp;        // full[l]: level l rows; fsp[l]: F points of level l (C points = full[l+1])
  struct LevelOps { OpRef A_ff, A_fc, A_cf, A_cc, Z, W, M_ff, M_cc, G; };
  std::vector<LevelOps> ops;
  OpRef top, coarse_A, coarse_M;
  int64_t max_n = 0;
};

Partitioner::~Partitioner() = default;

Partitioner::Partitioner(const Hierarchy& G, const Csr& A_top, int nranks, const int32_t* owner, const int64_t* hint) : impl_(new Impl) {
  if (G.opt.full_smoothing) throw std::runtime_error("full up / down smoothing is not available for a distributed hierarchy");
  Impl& I = *impl_;
  I.G = &G; I.A_top = &A_top; I.N = nranks; I.hint = hint;
  const int L = G.no_levels();
  I.full.resize(L); I.fsp.resize(L > 0 ? L - 1 : 0);
  I.full[0].own.assign(owner, owner + A_top.m);
  for (int64_t i = 0; i < A_top.m; i++)
    if (owner[i] < 0 || owner[i] >= nranks) throw std::runtime_error("partition: owner out of range");
  finish_space(I.full[0], nranks);
  for (int l = 0; l + 1 < L; l++) {
    const Level& Lv = *G.levels[l];
    Space& F = I.fsp[l]; Space& Cn = I.full[l + 1];
    F.own.resize(Lv.n_f); Cn.own.resize(Lv.n_c);
    const int32_t* own = I.full[l].own.data();
#pragma omp parallel for schedule(static)
    for (int64_t f = 0; f < Lv.n_f; f++) F.own[f] = own[Lv.F[f]];
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < Lv.n_c; c++) Cn.own[c] = own[Lv.C[c]];
    finish_space(F, nranks); finish_space(Cn, nranks);
  }
  for (auto& s : I.full) I.max_n = std::max(I.max_n, s.n);
  auto mk = [&](const Csr& X, const Space& R, const Space& C) {
    OpRef o; o.X = &X; o.R = &R; o.C = &C;
    if (!X.empty()) {
      if (X.m != R.n || X.n != C.n) throw std::runtime_error("partition: operand / space size mismatch");
      o.halo = compute_halo(X, R, C, nranks);
    } else o.halo.need.assign((size_t)nranks * nranks, {});
    return o;
  };
  I.ops.resize(L > 0 ? L - 1 : 0);
  for (int l = 0; l + 1 < L; l++) {
    const Level& Lv = *G.levels[l];
    const Space& F = I.fsp[l]; const Space& C = I.full[l + 1];
    auto& o = I.ops[l];
    o.A_ff = mk(Lv.A_ff, F, F); o.A_fc = mk(Lv.A_fc, F, C);
    static const Csr none;
    const bool any_c = G.opt.any_c_smooths();
    o.A_cf = mk(any_c ? Lv.A_cf : none, C, F); o.A_cc = mk(any_c ? Lv.A_cc : none, C, C);
    o.Z = mk(Lv.Z, C, F); o.W = mk(Lv.W, F, C);
    o.G = mk(Lv.G, F, C);
    o.M_ff = mk(Lv.inv_ff.mat, F, F); o.M_cc = mk(Lv.inv_cc.mat, C, C);
  }
  I.top = mk(A_top, I.full[0], I.full[0]);
  I.coarse_A = mk(G.coarse_A, I.full[L - 1], I.full[L - 1]);
  I.coarse_M = mk(G.coarse_inv.mat, I.full[L - 1], I.full[L - 1]);
}

void Partitioner::extract(int rank, Archive& ar) const {
  const Impl& I = *impl_;
  const Hierarchy& G = *I.G;
  const int N = I.N, L = G.no_levels();
  if (rank < 0 || rank >= N) throw std::runtime_error("partition: bad rank");
  std::vector<int32_t> gmap((size_t)I.max_n, -1);
  auto none = [](int64_t) {};
  auto emit = [&](const std::string& name, const OpRef& op) {
    if (op.X->empty()) return;
    Csr loc; HaloHost h;
    extract_op(op, rank, N, gmap, loc, h, none);
    put_csr(ar, name, loc, h.n_ghost);
    put_halo(ar, name, h);
  };
  {
    int64_t meta[16] = {0};
    meta[0] = L; meta[1] = rank; meta[2] = N; meta[3] = I.A_top->m; meta[4] = G.opt.coarse_richardson_its;
    meta[5] = G.single_level_jacobi ? 1 : 0;
    for (int i = 0; i < 10; i++) meta[6 + i] = G.opt.smooth_order[i];
    ar.add_copy("meta", Archive::I64, meta, 16);
    ar.add_copy("gid", Archive::I32, I.full[0].members[rank].data(), I.full[0].count(rank));
    if (I.hint) {
      const auto& mem = I.full[0].members[rank];
      std::vector<int64_t> k(mem.size());
      for (size_t q = 0; q < mem.size(); q++) k[q] = I.hint[mem[q]];
      ar.add_copy("hint", Archive::I64, k.data(), (int64_t)k.size());
    }
  }
  emit("top.A", I.top);
  for (int l = 0; l + 1 < L; l++) {
    const Level& Lv = *G.levels[l];
    const Space& S = I.full[l]; const Space& F = I.fsp[l]; const Space& C = I.full[l + 1];
    const std::string p = "L" + std::to_string(l);
    const int64_t nf = F.count(rank), nc = C.count(rank);
    int64_t meta[16] = {0};
    meta[0] = S.count(rank); meta[1] = nf; meta[2] = nc; meta[3] = Lv.aff_diag ? 1 : 0; meta[4] = Lv.one_point ? 1 : 0;
    meta[5] = Lv.nnz_A;
    for (int i = 0; i < 10; i++) meta[6 + i] = Lv.smooth_order[i];
    ar.add_copy(p + ".meta", Archive::I64, meta, 16);
    {
      // local positions of the F / C points inside the rank's level vector
      std::vector<int32_t> Fl((size_t)nf), Cl((size_t)nc);
      const auto& fm = F.members[rank]; const auto& cm = C.members[rank];
      for (int64_t k = 0; k < nf; k++) Fl[k] = S.loc[Lv.F[fm[k]]];
      for (int64_t k = 0; k < nc; k++) Cl[k] = S.loc[Lv.C[cm[k]]];
      ar.add_copy(p + ".F", Archive::I32, Fl.data(), nf);
      ar.add_copy(p + ".C", Archive::I32, Cl.data(), nc);
    }
    if (Lv.aff_diag) {
      std::vector<double> d((size_t)nf);
      const auto& fm = F.members[rank];
      for (int64_t k = 0; k < nf; k++) d[k] = Lv.aff_d[fm[k]];
      ar.add_copy(p + ".aff_d", Archive::F64, d.data(), nf);
    }
    emit(p + ".A_ff", I.ops[l].A_ff);
    // A_fc together with the one-point prolongator (its column is a column of the A_fc row)
    if (!Lv.A_fc.empty()) {
      Csr loc; HaloHost h;
      std::vector<int32_t> wc;
      extract_op(I.ops[l].A_fc, rank, N, gmap, loc, h, [&](int64_t nlc) {
        if (!Lv.one_point) return;
        const auto& fm = F.members[rank];
        wc.resize((size_t)nf);
        for (int64_t k = 0; k < nf; k++) {
          const int32_t c = Lv.wcol[fm[k]];
          wc[k] = c < 0 ? -1 : (C.own[c] == rank ? C.loc[c] : (int32_t)(nlc + gmap[c]));
        }
      });
      put_csr(ar, p + ".A_fc", loc, h.n_ghost);
      put_halo(ar, p + ".A_fc", h);
      if (Lv.one_point) ar.add_copy(p + ".wcol", Archive::I32, wc.data(), nf);
    }
    // the operator of the fused first F sweep, with W's columns once more in ITS column numbering (its ghosts are those of
    // A_fc and the W columns of the F points the rank's rows of A_ff reach)
    if (!Lv.G.empty()) {
      Csr loc; HaloHost h;
      std::vector<int32_t> wc;
      extract_op(I.ops[l].G, rank, N, gmap, loc, h, [&](int64_t nlc) {
        const auto& fm = F.members[rank];
        wc.resize((size_t)nf);
        for (int64_t k = 0; k < nf; k++) {
          const int32_t c = Lv.wcol[fm[k]];
          // (a W column of a local F point is a column of its row of G whenever the row of A_ff holds its diagonal; an F
          // point without it gets no prolongated value from a ghost it does not receive)
          wc[k] = c < 0 ? -1 : (C.own[c] == rank ? C.loc[c] : (gmap[c] >= 0 ? (int32_t)(nlc + gmap[c]) : -2));
        }
      });
      for (int32_t c : wc) if (c == -2) throw std::runtime_error("partition: a one-point prolongator column is not a column of the fused operator's row (A_ff without its diagonal)");
      put_csr(ar, p + ".G", loc, h.n_ghost);
      put_halo(ar, p + ".G", h);
      ar.add_copy(p + ".wcol_g", Archive::I32, wc.data(), nf);
    }
    if (G.opt.any_c_smooths()) {     // only the C-point relaxation reads these
      emit(p + ".A_cf", I.ops[l].A_cf);
      emit(p + ".A_cc", I.ops[l].A_cc);
    }
    emit(p + ".Z", I.ops[l].Z);
    emit(p + ".W", I.ops[l].W);
    auto inv = [&](const std::string& nm, const InvOp& Iv, const OpRef& mat, const Space& Sp) {
      put_inv_meta(ar, nm, Iv);
      if (Iv.kind == InvOp::ASSEMBLED) emit(nm + ".mat", mat);
      if (Iv.kind == InvOp::DIAG) {
        const auto& mem = Sp.members[rank];
        std::vector<double> d(mem.size());
        for (size_t k = 0; k < mem.size(); k++) d[k] = Iv.d[mem[k]];
        ar.add_copy(nm + ".d", Archive::F64, d.data(), (int64_t)d.size());
      }
    };
    inv(p + ".inv_ff", Lv.inv_ff, I.ops[l].M_ff, F);
    inv(p + ".inv_cc", Lv.inv_cc, I.ops[l].M_cc, C);
  }
  emit("coarse.A", I.coarse_A);
  {
    put_inv_meta(ar, "coarse.inv", G.coarse_inv);
    if (G.coarse_inv.kind == InvOp::ASSEMBLED) emit("coarse.inv.mat", I.coarse_M);
    if (G.coarse_inv.kind == InvOp::DIAG) {
      const auto& mem = I.full[L - 1].members[rank];
      std::vector<double> d(mem.size());
      for (size_t k = 0; k < mem.size(); k++) d[k] = G.coarse_inv.d[mem[k]];
      ar.add_copy("coarse.inv.d", Archive::F64, d.data(), (int64_t)d.size());
    }
  }
}

// ---------------------------------------------------------------------------------------
// archive -> LocalHierarchy
// ---------------------------------------------------------------------------------------
namespace {
const Archive::Item& need(const Archive& ar, const std::string& name, int dtype) {
  const Archive::Item* it = ar.find(name);
  if (!it) throw std::runtime_error("archive: missing item " + name);
  if (it->dtype != dtype) throw std::runtime_error("archive: wrong dtype for " + name);
  return *it;
}
bool get_csr(const Archive& ar, const std::string& name, Csr& A, HaloHost& h) {
  if (!ar.find(name + ".shape")) return false;
  const auto& sh = need(ar, name + ".shape", Archive::I64);
  const int64_t* s = (const int64_t*)sh.ptr;
  A = Csr(); A.m = s[0]; A.n = s[1];
  const auto& rp = need(ar, name + ".rp", Archive::I64);
  const auto& ci = need(ar, name + ".ci", Archive::I32);
  const auto& v = need(ar, name + ".v", Archive::F64);
  if (rp.count != A.m + 1 || ci.count != v.count) throw std::runtime_error("archive: inconsistent CSR " + name);
  A.rp.view((const int64_t*)rp.ptr, rp.count); A.ci.view((const int32_t*)ci.ptr, ci.count); A.v.view((const double*)v.ptr, v.count);
  h = HaloHost(); h.n_ghost = s[2];
  const auto& pe = need(ar, name + ".halo.peers", Archive::I32);
  const auto& so = need(ar, name + ".halo.send_off", Archive::I64);
  const auto& ro = need(ar, name + ".halo.recv_off", Archive::I64);
  const auto& si = need(ar, name + ".halo.send_idx", Archive::I32);
  h.peers.assign((const int32_t*)pe.ptr, (const int32_t*)pe.ptr + pe.count);
  h.send_off.assign((const int64_t*)so.ptr, (const int64_t*)so.ptr + so.count);
  h.recv_off.assign((const int64_t*)ro.ptr, (const int64_t*)ro.ptr + ro.count);
  h.send_idx.view((const int32_t*)si.ptr, si.count);
  if ((int64_t)h.send_off.size() != pe.count + 1 || (int64_t)h.recv_off.size() != pe.count + 1 ||
      h.send_off.back() != si.count || h.recv_off.back() != h.n_ghost || h.n_ghost > A.n)
    throw std::runtime_error("archive: inconsistent halo plan " + name);
  // the device kernels index with these numbers: validate them on the host
  {
    const int64_t nnz = ci.count, ncol = A.n, nloc = A.n - h.n_ghost;
    if (A.rp[0] != 0 || A.rp[A.m] != nnz) throw std::runtime_error("archive: bad row pointers in " + name);
    int64_t bad = 0;
#pragma omp parallel for schedule(static) reduction(+ : bad)
    for (int64_t k = 0; k < nnz; k++) bad += (A.ci[k] < 0 || A.ci[k] >= ncol) ? 1 : 0;
    for (int64_t k = 0; k < si.count; k++) bad += (h.send_idx[k] < 0 || h.send_idx[k] >= nloc) ? 1 : 0;
    if (bad) throw std::runtime_error("archive: column / send index out of range in " + name);
  }
  return true;
}
void get_inv(const Archive& ar, const std::string& name, InvOp& I, HaloHost& h) {
  const auto& m = need(ar, name + ".meta", Archive::I64);
  const int64_t* s = (const int64_t*)m.ptr;
  I = InvOp();
  I.kind = (InvOp::Kind)s[0]; I.diag_scale = s[1] != 0; I.neumann = s[2] != 0; I.spmvs = (int)s[3];
  auto vec = [&](const char* suffix, std::vector<double>& out) {
    const auto& it = need(ar, name + suffix, Archive::F64);
    out.assign((const double*)it.ptr, (const double*)it.ptr + it.count);
  };
  vec(".coeffs", I.coeffs); vec(".re", I.re); vec(".im", I.im);
  if (I.kind == InvOp::ASSEMBLED && !get_csr(ar, name + ".mat", I.mat, h)) throw std::runtime_error("archive: missing " + name + ".mat");
  if (I.kind == InvOp::DIAG) { const auto& d = need(ar, name + ".d", Archive::F64); I.d.view((const double*)d.ptr, d.count); }
}
}  // namespace

void load_local(const Archive& ar, LocalHierarchy& out) {
  const auto& m = need(ar, "meta", Archive::I64);
  const int64_t* s = (const int64_t*)m.ptr;
  const int L = (int)s[0];
  out.rank = (int)s[1]; out.nranks = (int)s[2]; out.n_global = s[3];
  out.H = Hierarchy();
  out.H.opt.coarse_richardson_its = (int)s[4];
  out.H.single_level_jacobi = s[5] != 0;
  for (int i = 0; i < 10; i++) out.H.opt.smooth_order[i] = (int)s[6 + i];
  const auto& gid = need(ar, "gid", Archive::I32);
  out.gid.view((const int32_t*)gid.ptr, gid.count);
  out.hint = Arr<int64_t>();
  if (const Archive::Item* hk = ar.find("hint")) {
    if (hk->dtype != Archive::I64 || hk->count != gid.count) throw std::runtime_error("archive: bad ordering hint");
    out.hint.view((const int64_t*)hk->ptr, hk->count);
  }
  if (!get_csr(ar, "top.A", out.A_top, out.h_top)) throw std::runtime_error("archive: missing top.A");
  out.halos.clear();
  out.halos.resize(L > 0 ? L - 1 : 0);
  for (int l = 0; l + 1 < L; l++) {
    const std::string p = "L" + std::to_string(l);
    auto Lp = std::make_unique<Level>();
    Level& Lv = *Lp;
    const auto& lm = need(ar, p + ".meta", Archive::I64);
    const int64_t* q = (const int64_t*)lm.ptr;
    Lv.n = q[0]; Lv.n_f = q[1]; Lv.n_c = q[2]; Lv.aff_diag = q[3] != 0; Lv.one_point = q[4] != 0; Lv.nnz_A = q[5];
    for (int i = 0; i < 10; i++) Lv.smooth_order[i] = (int)q[6 + i];
    const auto& F = need(ar, p + ".F", Archive::I32); const auto& C = need(ar, p + ".C", Archive::I32);
    Lv.F.view((const int32_t*)F.ptr, F.count); Lv.C.view((const int32_t*)C.ptr, C.count);
    if (Lv.aff_diag) { const auto& d = need(ar, p + ".aff_d", Archive::F64); Lv.aff_d.view((const double*)d.ptr, d.count); }
    LevelHalos& H = out.halos[l];
    get_csr(ar, p + ".A_ff", Lv.A_ff, H.A_ff);
    if (!get_csr(ar, p + ".A_fc", Lv.A_fc, H.A_fc)) throw std::runtime_error("archive: missing " + p + ".A_fc");
    get_csr(ar, p + ".A_cf", Lv.A_cf, H.A_cf);
    get_csr(ar, p + ".A_cc", Lv.A_cc, H.A_cc);
    if (!get_csr(ar, p + ".Z", Lv.Z, H.Z)) throw std::runtime_error("archive: missing " + p + ".Z");
    get_csr(ar, p + ".W", Lv.W, H.W);
    if (Lv.one_point) {
      const auto& w = need(ar, p + ".wcol", Archive::I32);
      Lv.wcol.view((const int32_t*)w.ptr, w.count);
      if (w.count != Lv.n_f) throw std::runtime_error("archive: wcol size mismatch in " + p);
      for (int64_t i = 0; i < w.count; i++)
        if (Lv.wcol[i] >= Lv.A_fc.n) throw std::runtime_error("archive: wcol out of range in " + p);
    }
    if (get_csr(ar, p + ".G", Lv.G, H.G)) {
      const auto& w = need(ar, p + ".wcol_g", Archive::I32);
      Lv.wcol_g.view((const int32_t*)w.ptr, w.count);
      if (w.count != Lv.n_f || Lv.G.m != Lv.n_f) throw std::runtime_error("archive: fused operator size mismatch in " + p);
      for (int64_t i = 0; i < w.count; i++)
        if (Lv.wcol_g[i] >= Lv.G.n || Lv.wcol_g[i] < -1) throw std::runtime_error("archive: wcol_g out of range in " + p);
    }
    if (Lv.F.size() != Lv.n_f || Lv.C.size() != Lv.n_c || Lv.n_f + Lv.n_c != Lv.n || Lv.A_fc.m != Lv.n_f || Lv.Z.m != Lv.n_c)
      throw std::runtime_error("archive: level size mismatch in " + p);
    get_inv(ar, p + ".inv_ff", Lv.inv_ff, H.M_ff);
    get_inv(ar, p + ".inv_cc", Lv.inv_cc, H.M_cc);
    out.H.levels.push_back(std::move(Lp));
  }
  if (!get_csr(ar, "coarse.A", out.H.coarse_A, out.h_coarse_A)) throw std::runtime_error("archive: missing coarse.A");
  get_inv(ar, "coarse.inv", out.H.coarse_inv, out.h_coarse_M);
}

}  // namespace airhip
