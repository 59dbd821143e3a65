"""Multi-GPU row of the scope table on the CPU: the library's partitioner + halo
plans + the torch.distributed shipping of the pieces, validated by running the distributed
V-cycle in numpy on the pieces (tests/dist_util.py) against the global oracle V-cycle -
once with all ranks in one process, once with world_size 2 over gloo."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CASES = {
    "adv2d_40": (dict(dim=2, M=40, N=40), {}, "box2d"),
    "adv3d_14_lump": (dict(dim=3, M=14, N=14, P=14), dict(a_lump=True), "box3d"),
    "adv3d_14_balanced": (dict(dim=3, M=14, N=14, P=14), dict(a_lump=True), "balanced3d"),      # boxes of equal work (bench.py, N > 1)
    "advdiff2d_30": (dict(dim=2, M=30, N=30, alpha=0.05), {}, "block"),
    "adv2d_30_mf_trunc": (dict(dim=2, M=30, N=30), dict(matrix_free_polys=True, max_levels=3,
                                                         coarsest_inverse_type="power", coarsest_poly_order=4,
                                                         coarsest_matrix_free_polys=True), "block"),
    # scattered ownership (every row's neighbours mostly live elsewhere), lAIR restrictor
    "adv2d_24_lair_random": (dict(dim=2, M=24, N=24), dict(z_type="lair", lair_distance=3), "random"),
    # Z from the strong-R-dropped operator, SAI smoother, compatible-relaxation splitting
    "diff2d_20_rthr_sai_cr_random": (dict(dim=2, M=20, N=20, u=0, v=0, alpha=1.0),
                                     dict(strong_r_threshold=0.05, inverse_type="sai", cf_splitting_type="cr",
                                          strong_threshold=0.5), "random"),
}


def test_balanced_boxes_are_boxes_of_equal_work():
    """bench.py's decomposition for N > 1: boxes of equal WORK (non-zeros of an iteration's products in the rows that live with
    a point).  Equal boxes leave the downstream corner of the 3D upwind problem with far more than the mean; the balanced
    ones are still boxes, differ in size, and the weight they are cut by IS the per-rank work the partitioner's pieces hold."""
    import pflare_amd as P
    from pflare_amd import dist as D
    from dist_util import Piece
    M = 28
    A = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
    pc = P.PCAIR(a_lump=True)
    pc.setup_host_only(A)
    w = D.point_work(pc, A)
    assert w.shape == (M ** 3,) and np.all(w >= 1.0)
    for nranks in (2, 4, 8, 3):
        eq = D.box_owner(M, M, M, nranks)
        bal = D.balanced_box_owner(w, (M, M, M), nranks)
        assert bal.shape == eq.shape and set(np.unique(bal)) == set(range(nranks))
        k, j, i = np.meshgrid(np.arange(M), np.arange(M), np.arange(M), indexing="ij")
        for r in range(nranks):                                  # every rank owns a box
            m = (bal == r).reshape(M, M, M)
            ext = [(a[m].min(), a[m].max()) for a in (k, j, i)]
            assert int(m.sum()) == int(np.prod([hi - lo + 1 for lo, hi in ext])), (nranks, r)
        pe = np.bincount(eq, weights=w, minlength=nranks); pb = np.bincount(bal, weights=w, minlength=nranks)
        assert pb.max() / pb.mean() <= 1.12 and pb.max() / pb.mean() <= pe.max() / pe.mean() + 1e-12, (nranks, pe, pb)
    # the weight is the work: per-rank non-zeros of the pieces' products on 8 balanced boxes
    part = D.Partition(pc, A, 8, bal)
    tot = np.zeros(8)
    L = pc.num_levels
    for r in range(8):
        ar = part.extract(r)
        q = Piece((k_, v.copy()) for k_, v in ar.items())
        ar.close()
        for l in range(L - 1):
            for nm, t in (("Z", 1), ("A_fc", 1), ("A_ff", 1), ("inv_ff.mat", 2)):
                if q.has(f"L{l}.{nm}"):
                    tot[r] += t * q.mat(f"L{l}.{nm}").nnz
            if not q.has(f"L{l}.A_ff"):
                tot[r] += q.level(l)["n_f"]                         # diagonal A_ff: one entry per row
            if not q.has(f"L{l}.inv_ff.mat"):
                tot[r] += 2 * q.level(l)["n_f"]
        kind = q.inv("coarse.inv")["kind"]
        tot[r] += (1 * q.mat("coarse.inv.mat").nnz) if kind == 1 else 10 * q.mat("coarse.A").nnz
        tot[r] += q.mat("top.A").nnz
    part.close()
    np.testing.assert_allclose(tot, np.bincount(bal, weights=D.point_work(pc, A, vector_work=False), minlength=8), rtol=1e-12)


def _owner(kind, gen, nranks):
    from pflare_amd import dist as D
    if kind == "box3d":
        return D.box_owner(gen["M"], gen["N"], gen["P"], nranks)
    if kind == "balanced3d":
        import pflare_amd as P
        A = P.HostMat.adv_diff_fd(**gen)
        pc = P.PCAIR(**CASES["adv3d_14_balanced"][1])
        pc.setup_host_only(A)
        return D.balanced_box_owner(D.point_work(pc, A), (gen["M"], gen["N"], gen["P"]), nranks)
    if kind == "box2d":
        return D.box_owner(gen["M"], gen["N"], 1, nranks)
    n = gen["M"] * gen["N"] * gen.get("P", 1)
    if kind == "random":
        return np.random.default_rng(17).integers(0, nranks, size=n).astype(np.int32)
    return D.block_owner(n, nranks)


def _global(case):
    import pflare_amd as P
    from oracle import air_ref as R, bridge
    gen, opts, kind = CASES[case]
    A = P.HostMat.adv_diff_fd(**gen)
    pc = P.PCAIR(**opts)
    pc.setup_host_only(A)
    H = bridge.hierarchy_from_product(pc, A)
    return A, pc, H, R


def test_box_owner_matches_dmda_convention():
    from pflare_amd import dist as D
    assert D.box_dims(8, (464, 464, 464)) == (2, 2, 2)
    assert D.box_dims(4, (464, 464, 464)) == (2, 2, 1)
    assert D.box_dims(2, (464, 464, 464)) == (2, 1, 1)
    own = D.box_owner(4, 4, 4, 8)
    assert own.shape == (64,) and sorted(np.bincount(own)) == [8] * 8
    assert own[0] == 0 and own[3] == 1 and own[4 * 3] == 2 and own[16 * 3] == 4


@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("nranks", [2, 3, 4])
def test_pieces_reproduce_global_vcycle_in_process(case, nranks):
    from pflare_amd import dist as D
    from dist_util import Cluster, Piece
    A, pc, H, R = _global(case)
    gen, _, kind = CASES[case]
    n = A.shape[0]
    owner = _owner(kind, gen, nranks)
    part = D.Partition(pc, A, nranks, owner)
    ars = [part.extract(r) for r in range(nranks)]
    pieces = [Piece((k, v.copy()) for k, v in a.items()) for a in ars]
    for r, q in enumerate(pieces):
        assert np.array_equal(q.gid, np.flatnonzero(owner == r))
    cl = Cluster(pieces)
    # a single distributed product first: outer operator
    rng = np.random.default_rng(0)
    x = rng.standard_normal(n)
    import scipy.sparse as sp
    a = A.csr()
    Ag = sp.csr_matrix((a.val, a.col, a.rowptr), shape=a.shape)
    ys = cl.spmv("top.A", {r: x[q.gid] for r, q in cl.p.items()})
    yref = Ag @ x
    for r, q in cl.p.items():
        np.testing.assert_allclose(ys[r], yref[q.gid], rtol=1e-13, atol=1e-13)
    # the whole cycle
    b = rng.standard_normal(n)
    zs = cl.vcycle({r: b[q.gid] for r, q in cl.p.items()})
    z = np.empty(n)
    for r, q in cl.p.items():
        z[q.gid] = zs[r]
    zref = R.vcycle(H, b)
    assert np.linalg.norm(z - zref) <= 1e-12 * np.linalg.norm(zref)
    # the pieces of the fused first sweep's operator were there and consistent on every level that fuses
    # (assembled A_ff inverse, sparse A_ff, F sweeps first: not the matrix-free case)
    want = sum(1 for l in range(pieces[0].L - 1) if not pieces[0].level(l)["aff_diag"] and pieces[0].inv(f"L{l}.inv_ff")["kind"] == 1)
    assert getattr(cl, "fused_levels_checked", 0) == want, (getattr(cl, "fused_levels_checked", 0), want)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, case, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import proc_util
    proc_util.init_rank_env(rank, world, port)
    import torch
    import torch.distributed as dist
    from pflare_amd import dist as D
    from dist_util import Cluster, Piece
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=proc_util.pg_timeout())
    try:
        gen, _, kind = CASES[case]
        n = gen["M"] * gen["N"] * gen.get("P", 1)
        rng = np.random.default_rng(1)
        b = rng.standard_normal(n)          # same on every rank
        if rank == 0:
            A, pc, H, R = _global(case)
            owner = _owner(kind, gen, world)
            part = D.Partition(pc, A, world, owner)
            mine = part.extract(0)
            for r in range(1, world):
                ar = part.extract(r)
                D._send_piece(ar, r, None, None, False)
                ar.close()
        else:
            mine = D._recv_piece(0, None, None, False)
        q = Piece((k, v.copy()) for k, v in mine.items())
        cl = Cluster([q], dist=dist)
        z = cl.vcycle({rank: b[q.gid]})[rank]
        # gather on rank 0
        if rank == 0:
            full = np.empty(n); full[q.gid] = z
            for r in range(1, world):
                box = [None]
                dist.recv_object_list(box, src=r)
                gid, zr = box[0]
                full[gid] = zr
            zref = R.vcycle(H, b)
            err = float(np.linalg.norm(full - zref) / np.linalg.norm(zref))
            with open(out, "w") as f:
                f.write(f"{err:.3e} {cl.nexch}\n")
        else:
            dist.send_object_list([(q.gid, z)], dst=0)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["adv2d_40", "adv3d_14_lump"])
def test_world_size_2_gloo_vcycle(case, tmp_path):
    import proc_util
    out = str(tmp_path / "result.txt")
    proc_util.spawn(_worker, (2, _free_port(), case, out), 2)
    err, nexch = open(out).read().split()
    assert float(err) <= 1e-12
    assert int(nexch) > 0
