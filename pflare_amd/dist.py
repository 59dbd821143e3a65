"""Row-block distribution of a PCAIR hierarchy over the GPUs of one node.

One process per GPU.  The hierarchy is built once (rank `src`, deterministic setup), cut into
per-rank pieces by the library (csrc/host/dist.cpp: owned rows of every operand, columns
renumbered [local | ghost], halo plans) and shipped with torch.distributed; every rank then
uploads its piece.  Inside the V-cycle the only communication is the neighbour exchange in
front of each sparse product (RCCL send/recv groups on the library stream, or host-staged
callbacks when ranks share a GPU / run under gloo); the outer Krylov solver adds an all-reduce of
1..31 doubles per iteration.  This is the reference's MPIAIJ model (src/PMISR_Module.F90:174-180,
src/Grid_Transfer.F90:656-668,745-746) with the exchange moved onto the device.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import ALLREDUCE_FN, EXCHANGE_FN, check, lib
from .pc import PCAIR, as_csr

_NP = {0: np.int32, 1: np.int64, 2: np.float64}
_CODE = {np.dtype(np.int32): 0, np.dtype(np.int64): 1, np.dtype(np.float64): 2}


# ------------------------------------------------------------------------------------------
# ownership of the fine rows
# ------------------------------------------------------------------------------------------
def box_dims(nranks, shape):
    """Process grid (px, py, pz) for a structured grid: prime factors of nranks go, largest
    first, to the direction with the largest local extent (x first on ties) - 8 -> 2x2x2,
    4 -> 2x2x1, 2 -> 2x1x1 on a cube, as PETSc's DMDA does."""
    p = [1, 1, 1]
    n = int(nranks)
    f, fac = 2, []
    while n > 1:
        while n % f == 0:
            fac.append(f); n //= f
        f += 1
    for f in sorted(fac, reverse=True):
        ext = [shape[d] / p[d] for d in range(3)]
        d = max(range(3), key=lambda k: (ext[k], -k))
        p[d] *= f
    return tuple(p)


def box_owner(M, N=1, P=1, nranks=1):
    """owner[(k*N+j)*M+i] for the box decomposition of an M x N x P grid (row numbering of
    tests/adv_diff_fd.c)."""
    px, py, pz = box_dims(nranks, (M, N, P))
    bx = (np.arange(M, dtype=np.int64) * px) // M
    by = (np.arange(N, dtype=np.int64) * py) // N
    bz = (np.arange(P, dtype=np.int64) * pz) // P
    own = (bz[:, None, None] * py + by[None, :, None]) * px + bx[None, None, :]
    return np.ascontiguousarray(own.reshape(-1).astype(np.int32))


def point_work(pc, A_top=None, vector_work=True):
    """Per fine point, the non-zeros of one Krylov iteration's products in the rows that live with it: a level's rows stay
    with their parent C points (the reference's row ownership, src/Grid_Transfer.F90:745-746), so the rows of every level
    are rows of some fine point.  Counted: the restriction Z, the F-C product (A_fc, or the slightly larger G of the fused
    first sweep), A_ff and twice the A_ff inverse per level, ten products with the coarsest operator (the truncated recipe's
    polynomial; an assembled inverse counts once) and the outer operator.  vector_work: plus what a row costs in vectors, in
    units of one non-zero (10 bytes): 6 per row and level for the cycle's vectors (~60 bytes), 15 per fine row for the Krylov
    method's (GMRES(30), ~150 bytes per iteration on average).  This is the weight balanced_box_owner cuts by."""
    from .pc import as_csr
    L = pc.num_levels
    comp = None
    w = None
    for l in range(L - 1):
        lv = pc.export_level(l, copy=False)
        F = np.asarray(lv["F"], dtype=np.int64); C = np.asarray(lv["C"], dtype=np.int64)
        if comp is None:
            comp = np.arange(F.size + C.size, dtype=np.int64)
            w = np.zeros(comp.size, dtype=np.float64)
        rf = np.diff(np.asarray(lv["A_fc"].rowptr)).astype(np.float64)
        if "A_ff" in lv:
            rf += np.diff(np.asarray(lv["A_ff"].rowptr))
        else:
            rf += 1.0
        inv = lv.get("inv_ff")
        if inv is not None and inv.get("mat") is not None:
            rf += 2.0 * np.diff(np.asarray(inv["mat"].rowptr))
        else:
            rf += 2.0
        w[comp[F]] += rf + (6.0 if vector_work else 0.0)
        w[comp[C]] += np.diff(np.asarray(lv["Z"].rowptr)) + (6.0 if vector_work else 0.0)
        comp = comp[C]
    A_co, inv = pc.export_coarse(copy=False)
    if comp is None:
        comp = np.arange(A_co.shape[0], dtype=np.int64)
        w = np.zeros(comp.size, dtype=np.float64)
    reps = 1.0 if (inv is not None and inv.get("mat") is not None) else 10.0
    op = inv["mat"] if reps == 1.0 else A_co
    w[comp] += reps * np.diff(np.asarray(op.rowptr))
    if A_top is not None:
        w += np.diff(np.asarray(as_csr(A_top).rowptr))
    if vector_work:
        w += 15.0
    return w


def balanced_box_owner(weight, shape, nranks):
    """Box decomposition of an M x N x P grid (row (k*N+j)*M+i) into box_dims(nranks) boxes of EQUAL WEIGHT instead of equal
    size: recursive coordinate bisection along x, then y inside each x slab, then z inside each column - every rank still
    owns a box, so the halo surfaces stay those of a box decomposition.  Why: with equal boxes the downstream corner of the 3D
    upwind problem carries 1.85 x the cycle's non-zeros of the upstream corner (fill-in follows the characteristics):
    busiest / mean = 1.31-1.35 on 2x2x2 boxes, which caps the parallel efficiency at 75 %."""
    M, N, P = shape
    px, py, pz = box_dims(nranks, shape)
    w = np.asarray(weight, dtype=np.float64).reshape(P, N, M)

    def cuts(marg, parts):
        """boundaries 0 = c_0 < c_1 < ... < c_parts = len(marg) with equal cumulative weight (at least one plane each)"""
        c = np.concatenate([[0.0], np.cumsum(marg)])
        tot = c[-1]
        out = [0]
        for k in range(1, parts):
            j = int(np.searchsorted(c, tot * k / parts))
            if j > 0 and abs(c[j - 1] - tot * k / parts) < abs(c[j] - tot * k / parts):
                j -= 1                                  # the nearer of the two planes
            j = max(out[-1] + 1, min(j, len(marg) - (parts - k)))
            out.append(j)
        out.append(len(marg))
        return out
    own = np.empty((P, N, M), dtype=np.int32)
    cx = cuts(w.sum(axis=(0, 1)), px)
    for bx in range(px):
        sx = slice(cx[bx], cx[bx + 1])
        cy = cuts(w[:, :, sx].sum(axis=(0, 2)), py)
        for by in range(py):
            sy = slice(cy[by], cy[by + 1])
            cz = cuts(w[:, sy, sx].sum(axis=(1, 2)), pz)
            for bz in range(pz):
                own[cz[bz]:cz[bz + 1], sy, sx] = (bz * py + by) * px + bx
    return np.ascontiguousarray(own.reshape(-1))


def block_owner(n, nranks):
    """contiguous row blocks"""
    return np.ascontiguousarray(((np.arange(n, dtype=np.int64) * nranks) // n).astype(np.int32))


# ------------------------------------------------------------------------------------------
# archives (named flat arrays = the wire format of one rank's piece)
# ------------------------------------------------------------------------------------------
class Archive:
    def __init__(self, handle=None):
        self._keep = []
        if handle is None:
            handle = C.c_void_p()
            check(lib().airhip_archive_create(C.byref(handle)))
        self._h = handle

    def add(self, name, arr):
        arr = np.ascontiguousarray(arr)
        self._keep.append(arr)
        check(lib().airhip_archive_add(self._h, name.encode(), _CODE[arr.dtype], arr.ctypes.data, arr.size))

    def items(self):
        """(name, numpy view) pairs; views are valid while the archive lives"""
        n = C.c_int()
        check(lib().airhip_archive_count(self._h, C.byref(n)))
        for i in range(n.value):
            name = C.c_char_p(); dt = C.c_int(); ptr = C.c_void_p(); cnt = C.c_int64()
            check(lib().airhip_archive_item(self._h, i, C.byref(name), C.byref(dt), C.byref(ptr), C.byref(cnt)))
            if cnt.value == 0:
                a = np.zeros(0, _NP[dt.value])
            else:
                ct = {0: C.c_int32, 1: C.c_int64, 2: C.c_double}[dt.value]
                a = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ct)), shape=(cnt.value,))
            yield name.value.decode(), a

    def nbytes(self):
        return sum(a.nbytes for _, a in self.items())

    def close(self):
        if self._h:
            lib().airhip_archive_destroy(C.byref(self._h))
            self._h = None
        self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Partition:
    """Cuts the host hierarchy of a set-up PCAIR (csrc/host/dist.cpp)."""

    def __init__(self, pc: PCAIR, A_top, nranks, owner):
        self._A = as_csr(A_top)
        self._owner = np.ascontiguousarray(owner, dtype=np.int32)
        if self._owner.size != self._A.shape[0]:
            raise ValueError("owner must have one entry per fine row")
        self._h = C.c_void_p()
        check(lib().airhip_partition_create(pc._h, self._A.ref(), int(nranks),
                                            self._owner.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(self._h)))

    def extract(self, rank) -> Archive:
        h = C.c_void_p()
        check(lib().airhip_partition_extract(self._h, int(rank), C.byref(h)))
        return Archive(h)

    def close(self):
        if self._h:
            lib().airhip_partition_destroy(C.byref(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------
# transports
# ------------------------------------------------------------------------------------------
class StagedComm:
    """Halo exchange staged through the host and carried by torch.distributed point-to-point
    calls (gloo): for CPU-hosted ranks, several ranks on one GPU, or as a template for MPI."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.error = None
        self._ex = EXCHANGE_FN(self._exchange)
        self._ar = ALLREDUCE_FN(self._allreduce)
        self._h = C.c_void_p()
        check(lib().airhip_comm_create_staged(self.rank, self.size, self._ex, self._ar, None, C.byref(self._h)))

    def _global(self, r):
        return self.dist.get_global_rank(self.group, r) if self.group is not None else r

    def _exchange(self, user, npeers, peers, send, send_off, recv, recv_off):
        try:
            import torch
            dist = self.dist
            if npeers == 0:
                return 0
            ns, nr = send_off[npeers], recv_off[npeers]
            sb = torch.from_numpy(np.ctypeslib.as_array(send, shape=(ns,))) if ns else None
            rb = torch.from_numpy(np.ctypeslib.as_array(recv, shape=(nr,))) if nr else None
            ops = []
            for k in range(npeers):
                p = self._global(peers[k])
                if send_off[k + 1] > send_off[k]:
                    ops.append(dist.P2POp(dist.isend, sb[send_off[k]:send_off[k + 1]], p, self.group))
                if recv_off[k + 1] > recv_off[k]:
                    ops.append(dist.P2POp(dist.irecv, rb[recv_off[k]:recv_off[k + 1]], p, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            return 0
        except Exception as e:          # never let an exception cross the C boundary
            self.error = e
            return 1

    def _allreduce(self, user, vals, n):
        try:
            import torch
            t = torch.from_numpy(np.ctypeslib.as_array(vals, shape=(n,)))
            self.dist.all_reduce(t, group=self.group)
            return 0
        except Exception as e:
            self.error = e
            return 1

    def stats(self):
        a, b = C.c_int64(), C.c_int64()
        check(lib().airhip_comm_stats(self._h, C.byref(a), C.byref(b)))
        c, d = C.c_int64(), C.c_int64()
        check(lib().airhip_comm_traffic(self._h, C.byref(c), C.byref(d)))
        return dict(exchanges=a.value, allreduces=b.value, bytes_sent=c.value, messages_sent=d.value)

    def selftest(self, count=4096):
        """Collective: all-pairs exchange of a known pattern + an all-reduce; True if intact on this rank."""
        ok = C.c_int(0)
        check(lib().airhip_comm_selftest(self._h, count, C.byref(ok)))
        if getattr(self, "error", None) is not None:
            raise self.error
        return bool(ok.value)

    def close(self):
        if self._h:
            lib().airhip_comm_destroy(C.byref(self._h))
            self._h = None


class RcclComm:
    """RCCL send/recv + all-reduce issued by the library on its own stream (device buffers over
    xGMI).  The unique id is created on rank 0 and broadcast with torch.distributed."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        uid = np.zeros(128, dtype=np.uint8)
        if self.rank == 0:
            check(lib().airhip_comm_rccl_unique_id(uid.ctypes.data))
        box = [uid.tobytes()]
        dist.broadcast_object_list(box, src=0, group=group)
        uid = np.frombuffer(box[0], dtype=np.uint8).copy()
        self._h = C.c_void_p()
        check(lib().airhip_comm_create_rccl(self.rank, self.size, uid.ctypes.data, C.byref(self._h)))

    stats = StagedComm.stats
    selftest = StagedComm.selftest
    close = StagedComm.close


# ------------------------------------------------------------------------------------------
# shipping pieces
# ------------------------------------------------------------------------------------------
def _send_piece(ar: Archive, dst, meta_group, bulk_group, on_gpu):
    import torch
    import torch.distributed as dist
    items = list(ar.items())
    header = [(n, int(_CODE[a.dtype]), int(a.size)) for n, a in items]
    dist.send_object_list([header], dst=dst, group=meta_group)
    for _, a in items:
        if a.size == 0:
            continue
        t = torch.from_numpy(a)
        if on_gpu:
            t = t.cuda()
        dist.send(t, dst=dst, group=bulk_group)


def _recv_piece(src, meta_group, bulk_group, on_gpu) -> Archive:
    import torch
    import torch.distributed as dist
    box = [None]
    dist.recv_object_list(box, src=src, group=meta_group)
    ar = Archive()
    for name, code, count in box[0]:
        if count == 0:
            ar.add(name, np.zeros(0, _NP[code]))
            continue
        t = torch.empty(count, dtype={0: torch.int32, 1: torch.int64, 2: torch.float64}[code],
                        device="cuda" if on_gpu else "cpu")
        dist.recv(t, src=src, group=bulk_group)
        ar.add(name, t.cpu().numpy())
    return ar


def distribute(pc_global, A_top, owner, comm, src=0, meta_group=None, bulk_group=None, bulk_on_gpu=False,
               profile=False, log=None):
    """Collective.  Rank `src` passes its set-up PCAIR (host hierarchy present), the fine
    operator and owner[]; the others pass None.  Returns this rank's distributed PCAIR and the
    global indices of its rows."""
    import time
    import torch.distributed as dist
    rank, size = dist.get_rank(), dist.get_world_size()
    mine = None
    phases = dict(cut_plan_s=0.0, cut_extract_s=0.0, ship_s=0.0, upload_s=0.0)
    if rank == src:
        # the cut of rank r + 1 (OpenMP inside the library, the GIL released) runs on a worker thread while the piece of
        # rank r travels: at 464^3 the two were ~40 s one after the other (extrapolated from 6 s at 256^3)
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.time()
        part = Partition(pc_global, A_top, size, owner)
        phases["cut_plan_s"] = time.time() - t0

        def cut(r):
            t = time.time()
            a = part.extract(r)
            return a, time.time() - t
        with ThreadPoolExecutor(max_workers=1) as pool:
            nxt = pool.submit(cut, 0)
            for r in range(size):
                ar, dt = nxt.result()
                phases["cut_extract_s"] += dt
                if r + 1 < size:
                    nxt = pool.submit(cut, r + 1)
                if log:
                    log(f"piece of rank {r}: {ar.nbytes() / 1e9:.2f} GB (cut in {dt:.1f} s)")
                if r == src:
                    mine = ar
                else:
                    t = time.time()
                    _send_piece(ar, r, meta_group, bulk_group, bulk_on_gpu)
                    phases["ship_s"] += time.time() - t
                    ar.close()
        part.close()
    else:
        t = time.time()
        mine = _recv_piece(src, meta_group, bulk_group, bulk_on_gpu)
        phases["ship_s"] = time.time() - t      # includes the wait for the pieces of the ranks before this one
    gid = dict(mine.items())["gid"].copy()
    pc = PCAIR()
    if profile:
        pc.profile(True)
    t = time.time()
    check(lib().airhip_pcair_load_distributed(pc._h, mine._h, comm._h))
    phases["upload_s"] = time.time() - t
    mine.close()
    pc._comm = comm          # the communicator must outlive the PCAIR
    pc.distribute_phases = phases
    return pc, gid
