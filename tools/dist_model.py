#!/usr/bin/env python3
"""What the distributed V-cycle costs per level, from the real per-rank pieces (CPU only, no GPU needed).

    python tools/dist_model.py [--size 96] [--ranks 8] [--scale-to 464] [--bw 5.2e12] [--latency 15e-6] [--link 50e9]

Builds the PCAIR hierarchy of the 3D advection benchmark with the host setup, cuts it into DMDA-style boxes with the
library's partitioner (exactly what bench.py ships to the ranks) and reports, per level and for the rank with the most
work:
  * the rows / non-zeros each sparse product of the cycle runs on and the ghost values it needs (= halo bytes), per peer;
  * a time model of one V-cycle at that size: kernel time = algorithmic bytes / bw + launch, exchange time = latency +
    bytes / link, exposed time of an exchange = what the interior blocks do not cover (overlap; only attempted when the
    interior blocks run for at least --min-interior) or all of it (no overlap);
  * with --scale-to N the same model with rows and non-zeros scaled by the volume ratio and ghosts by the surface ratio
    (an extrapolation to the benchmark size, which needs 183 GB of host memory to build; the level count is kept);
  * how much redundant work the two alternatives to "one exchange per product" would cost:
      - replication of the levels below level k on every rank (one all-gather instead of their exchanges): the share of
        the cycle's bytes those levels hold;
      - communication avoidance (ghost layers deep enough for the whole F relaxation of a level, 2 exchanges per level
        instead of 6): |N_4(owned)| / |owned| with N_k the distance-k neighbourhood in the pattern of A_ff / M_ff.
The numbers quoted in DESIGN.md section 6 come from this script (profiles/r03_dist_model_*.txt, r05_dist_model_*.txt)."""
import argparse
import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def S(nnz, m, n):
    return 12.0 * nnz + 4.0 * (m + 1) + 8.0 * m + 8.0 * n


def op_stats(q, name):
    if not q.has(name):
        return None
    A = q.mat(name)
    peers, so, ro, sidx = q.halo(name) if (name + ".halo.peers") in q.d else ([], [0], [0], [])
    return dict(m=A.shape[0], n=A.shape[1], nnz=A.nnz, ghost=int(ro[-1]) if len(ro) else 0, peers=len(peers))


def model(pieces, L, args, vol, surf, title):
    """one V-cycle on the busiest rank; returns the algorithmic bytes per level (all ranks)"""
    rows = []
    tot = dict(kernel=0.0, exch=0.0, exposed_overlap=0.0, exposed_plain=0.0, nexch=0, msgs=0, bytes=0.0)
    level_bytes = np.zeros(L)
    # the products of one cycle: round 5 runs the first F sweep on G = A_fc + A_ff W where the pieces carry it (G, M, A_ff, M);
    # the reference's sequence (A_fc, 2 x A_ff, 2 x M) elsewhere and with --unfused
    ops = []
    for l in range(L - 1):
        fused = (not args.unfused) and pieces[0].has(f"L{l}.G")
        seq = (("Z", 1), ("G", 1), ("A_ff", 1), ("inv_ff.mat", 2)) if fused else (("Z", 1), ("A_fc", 1), ("A_ff", 2), ("inv_ff.mat", 2))
        ops += [(l, f"L{l}.{nm}", times) for nm, times in seq]
    ops.append((L - 1, "coarse.A", 10))
    for l, name, times in ops:
        st = [op_stats(q, name) for q in pieces]
        if st[0] is None:
            continue
        w = max(st, key=lambda s: S(s["nnz"], s["m"], s["n"]))                  # the rank with the most work sets the pace
        by = S(w["nnz"] * vol, w["m"] * vol, w["n"] * vol)
        level_bytes[l] += times * sum(S(s["nnz"], s["m"], s["n"]) for s in st)
        t_k = by / args.bw + args.launch
        gh = max(s["ghost"] for s in st) * surf
        pe = max(s["peers"] for s in st)
        t_x = 0.0 if pe == 0 else args.latency + 8.0 * gh / max(pe, 1) / args.link
        frac_int = max(0.0, 1.0 - 3.0 * gh / max(w["m"] * vol, 1))               # rough share of rows in interior blocks
        t_int = by * frac_int / args.bw
        exposed = max(0.0, t_x - t_int) if t_int >= args.min_interior else t_x
        label = name.split(".", 1)[1].replace("inv_ff.mat", "M_ff") if name.startswith("L") else "coarse_A"
        rows.append((l + 1, label, times, w["m"] * vol, w["nnz"] * vol, gh, pe, t_k * 1e6, t_x * 1e6, exposed * 1e6))
        tot["kernel"] += times * t_k; tot["exch"] += times * t_x
        tot["exposed_overlap"] += times * exposed; tot["exposed_plain"] += times * t_x
        tot["nexch"] += times * (1 if pe else 0); tot["msgs"] += times * pe; tot["bytes"] += times * 8.0 * gh
    print(f"## {title}")
    print("level operand  x  rows/rank   nnz/rank  ghosts peers  kernel_us exchange_us exposed_us(overlap)")
    for r in rows:
        print(f"{r[0]:5d} {r[1]:8s} {r[2]:2d} {r[3]:10.0f} {r[4]:10.0f} {r[5]:7.0f} {r[6]:5d} {r[7]:10.1f} {r[8]:11.1f} {r[9]:10.1f}")
    print(f"# per V-cycle on the busiest rank: kernels {tot['kernel'] * 1e3:.3f} ms, {tot['nexch']} exchanges, {tot['msgs']} messages, "
          f"{tot['bytes'] / 1e6:.2f} MB of halo;  exchange time {tot['exch'] * 1e3:.3f} ms, of it exposed "
          f"{tot['exposed_overlap'] * 1e3:.3f} ms with overlap / {tot['exposed_plain'] * 1e3:.3f} ms without")
    eff_o = tot["kernel"] / (tot["kernel"] + tot["exposed_overlap"]); eff_p = tot["kernel"] / (tot["kernel"] + tot["exposed_plain"])
    print(f"# => cycle time {1e3 * (tot['kernel'] + tot['exposed_overlap']):.3f} ms with overlap ({eff_o:.0%} of it kernels), "
          f"{1e3 * (tot['kernel'] + tot['exposed_plain']):.3f} ms without ({eff_p:.0%})")
    return level_bytes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=96)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--scale-to", type=int, default=0)
    ap.add_argument("--bw", type=float, default=5.8e12, help="achieved HBM bandwidth of the SpMV kernels (B/s)")
    ap.add_argument("--latency", type=float, default=15e-6, help="one send / receive group, small messages (s)")
    ap.add_argument("--link", type=float, default=50e9, help="effective bandwidth to one peer (B/s)")
    ap.add_argument("--launch", type=float, default=4e-6, help="fixed cost of one kernel launch on the stream (s)")
    ap.add_argument("--min-interior", type=float, default=10e-6, help="interior kernel time below which no overlap is attempted (s)")
    ap.add_argument("--equal-boxes", action="store_true", help="DMDA-style boxes of equal size instead of the boxes of equal work bench.py cuts (round 5)")
    ap.add_argument("--unfused", action="store_true", help="model the reference's sequence of products (rounds 1-4) instead of the fused first sweep")
    args = ap.parse_args()
    import pflare_amd as P
    from pflare_amd import dist as D
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from dist_util import Piece

    M = args.size
    hm = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
    pc = P.PCAIR(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, coarsest_inverse_type="newton",
                 coarsest_matrix_free_polys=True, coarsest_poly_order=10)
    pc.setup_host_only(hm)
    L = pc.num_levels
    owner = D.box_owner(M, M, M, args.ranks) if args.equal_boxes else D.balanced_box_owner(D.point_work(pc, hm), (M, M, M), args.ranks)
    part = D.Partition(pc, hm, args.ranks, owner)
    pieces = []
    for r in range(args.ranks):
        ar = part.extract(r)
        pieces.append(Piece([(k, v.copy()) for k, v in ar.items()]))
        ar.close()
    part.close()
    print(f"# {M}^3 = {M**3} rows, {L} levels, {args.ranks} ranks in boxes {D.box_dims(args.ranks, (M, M, M))} of equal {'size' if args.equal_boxes else 'work'}"
          f" ({np.bincount(owner).min()} ... {np.bincount(owner).max()} rows per rank)")
    print(f"# model: kernels at {args.bw / 1e12:.1f} TB/s + {args.launch * 1e6:.0f} us per launch, exchange = {args.latency * 1e6:.0f} us + "
          f"bytes / {args.link / 1e9:.0f} GB/s per peer, overlap only above {args.min_interior * 1e6:.0f} us of interior work")
    level_bytes = model(pieces, L, args, 1.0, 1.0, f"as built: {M}^3")
    if args.scale_to:
        f = args.scale_to / M
        model(pieces, L, args, f ** 3, f ** 2, f"scaled to {args.scale_to}^3 (rows, non-zeros x {f ** 3:.1f}; ghosts x {f ** 2:.1f}; {L} levels kept)")

    # ---- alternative 1: replicate everything below level k on all ranks
    cum = np.cumsum(level_bytes[::-1])[::-1] / level_bytes.sum()
    print("# replication: share of the cycle's algorithmic bytes held by levels >= k (replicated work is multiplied by the rank count)")
    print("#   " + "  ".join(f"k={k + 1}: {cum[k]:.0%}" for k in range(L)))

    # ---- alternative 2: communication avoidance, ghost layers for the whole F relaxation of a level
    print("# communication avoidance: rows a rank would have to hold for 2 F smooths without an exchange in between,")
    print("#   |N_k(owned F rows)| / |owned F rows|, k = 1 .. 4 (pattern of A_ff + M_ff), busiest rank")
    cur_owner = np.asarray(owner)
    for l in range(L - 1):
        lv = pc.export_level(l, copy=False)
        F, Cc = lv["F"], lv["C"]
        if "A_ff" not in lv:
            cur_owner = cur_owner[Cc]; continue
        A_ff = lv["A_ff"]
        G = sp.csr_matrix((np.ones(A_ff.nnz, dtype=np.int8), A_ff.col, A_ff.rowptr), shape=A_ff.shape)
        inv = lv["inv_ff"]
        if inv.get("mat") is not None:
            Mf = inv["mat"]
            G = G + sp.csr_matrix((np.ones(Mf.nnz, dtype=np.int8), Mf.col, Mf.rowptr), shape=Mf.shape)
        G = G.tocsr(); G.data[:] = 1
        of = cur_owner[F]
        worst = []
        for r in range(args.ranks):
            v = (of == r).astype(np.int8)
            n0 = int(v.sum())
            if n0 == 0:
                continue
            grow, x = [], v.copy()
            for k in range(4):
                x = (((G.T @ x) > 0) | (x > 0)).astype(np.int8)    # the rows the set's rows read: what the next product needs
                grow.append(int(x.sum()) / n0)
            worst.append(grow)
        w = np.max(np.array(worst), axis=0)
        print(f"#   level {l + 1:2d} ({len(F)} F rows): " + "  ".join(f"N{k + 1} {w[k]:.2f}x" for k in range(4)))
        cur_owner = cur_owner[Cc]


if __name__ == "__main__":
    main()
