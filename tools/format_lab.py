#!/usr/bin/env python3
"""How would the row blocks of a hierarchy encode?  (CPU only; design aid for the index formats of csrc/device/spmv.hip)

Builds the bench recipe's hierarchy with the host setup, lays every operator out as the device does (hierarchical F/C
ordering, rows of every segment in the order of the 16^3-brick locality key, greedy row blocks of <= 512 non-zeros and
<= 128 rows inside chunks of 512 rows) and counts, per operator, how its blocks would fare under
  * the packed 32-bit word (column span of the block < 2^19), else raw;
  * 16-bit words `base id | offset` with a per-block table of column bases (greedy over the block's sorted columns).

usage: python tools/format_lab.py [M=96] [brick=16]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pflare_amd as P

RECIPE = dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1,
              coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10)


def brick_key(M, B):
    nbk = (M + B - 1) // B
    ax = np.arange(M, dtype=np.int64)
    bi, li = ax // B, ax % B
    bkey = (bi[:, None, None] * nbk + bi[None, :, None]) * nbk + bi[None, None, :]
    return (bkey * B ** 3 + (li[:, None, None] * B + li[None, :, None]) * B + li[None, None, :]).reshape(-1)


def rank_by(k):
    ordr = np.argsort(k, kind="stable")
    pos = np.empty_like(ordr)
    pos[ordr] = np.arange(k.size)
    return pos


def blocks_of(rp):
    """greedy partition as k_chunk_blocks: list of (row0, row1)"""
    m = rp.size - 1
    out = []
    for c0 in range(0, m, 512):
        c1 = min(c0 + 512, m)
        r = c0
        while r < c1:
            r0 = r; p0 = rp[r]
            if rp[r + 1] - p0 > 512:
                r += 1
            else:
                while r < c1 and r - r0 < 128 and rp[r + 1] - p0 <= 512 and rp[r + 1] - rp[r] <= 512:
                    r += 1
            out.append((r0, r))
    return out


def analyse(name, rp, ci, variants, sample=4000, rng=None):
    blks = blocks_of(rp)
    if len(blks) > sample:
        sel = rng.choice(len(blks), sample, replace=False)
        blks = [blks[i] for i in sel]
    tot = 0
    packed = 0
    fit = {v: 0 for v in variants}
    nbases = {v: [] for v in variants}
    lines = []
    rows = []
    dist = 0
    for r0, r1 in blks:
        c = ci[rp[r0]:rp[r1]]
        n = c.size
        if n == 0:
            continue
        tot += n
        rows.append(r1 - r0)
        u = np.unique(c)
        dist += u.size
        lines.append(np.unique(u >> 4).size)
        if n <= 512 and u[-1] - u[0] < (1 << 19):
            packed += n
        if n > 512:
            continue
        for (ib, ob) in variants:
            # greedy bases over the sorted distinct columns
            nb = 0; base = -1 << 40
            for x in u:
                if x - base >= (1 << ob):
                    base = x; nb += 1
            if nb <= (1 << ib):
                fit[(ib, ob)] += n
                nbases[(ib, ob)].append(nb)
    s = f"{name:14s} blocks~{len(blks):6d} rows/blk {np.mean(rows):6.1f} distinct/nnz {dist / tot:5.2f} lines/blk {np.mean(lines):6.1f} packed32 {packed / tot:5.2f}"
    for v in variants:
        nbm = np.mean(nbases[v]) if nbases[v] else 0
        s += f" | {v[0]}+{v[1]}: fit {fit[v] / tot:5.2f} bases {nbm:5.1f}"
    print(s, flush=True)


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rng = np.random.default_rng(1)
    hm = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
    pc = P.PCAIR(**RECIPE).setup_host_only(hm)
    L = pc.num_levels
    print(f"{M}^3: {L} levels", flush=True)
    key = brick_key(M, B)
    lv = [pc.export_level(l, copy=True) for l in range(L - 1)]
    cA, _ = pc.export_coarse(copy=True)
    keys = [key]
    for l in range(L - 1):
        keys.append(keys[-1][lv[l]["C"]])
    fpos = [rank_by(keys[l][lv[l]["F"]]) for l in range(L - 1)]
    perm = [None] * L
    perm[L - 1] = rank_by(keys[L - 1])
    for l in range(L - 2, -1, -1):
        d = lv[l]
        p = np.empty(d["n"], dtype=np.int64)
        p[d["F"]] = fpos[l]
        p[d["C"]] = d["n_f"] + perm[l + 1]
        perm[l] = p
    variants = [(4, 12), (5, 11), (6, 10), (5, 12), (6, 12), (7, 9)]

    def permuted(csr, rowpos, colpos):
        """rows stored in device order (device row r = host row with rowpos == r), columns mapped"""
        rp, ci = np.asarray(csr.rowptr, dtype=np.int64), np.asarray(csr.col, dtype=np.int64)
        m = rp.size - 1
        inv = np.empty(m, dtype=np.int64); inv[rowpos] = np.arange(m)
        lens = np.diff(rp)[inv]
        nrp = np.zeros(m + 1, dtype=np.int64); np.cumsum(lens, out=nrp[1:])
        src = np.repeat(rp[inv], lens) + (np.arange(nrp[-1]) - np.repeat(nrp[:-1], lens))
        nci = colpos[ci[src]]
        # rows sorted by column
        rowid = np.repeat(np.arange(m), lens)
        o = np.lexsort((nci, rowid))
        return nrp, nci[o]

    for l in range(L - 1):
        d = lv[l]
        print(f"--- level {l + 1}: n {d['n']} n_f {d['n_f']} n_c {d['n_c']}", flush=True)
        ops = []
        if "A_ff" in d:
            ops.append(("A_ff", d["A_ff"], fpos[l], fpos[l]))
        ops.append(("A_fc", d["A_fc"], fpos[l], perm[l + 1]))
        ops.append(("Z", d["Z"], perm[l + 1], fpos[l]))
        for nm, csr, rpz, cpz in ops:
            if csr.nnz == 0:
                continue
            nrp, nci = permuted(csr, rpz, cpz)
            analyse(f"L{l + 1} {nm}", nrp, nci, variants, rng=rng)
    nrp, nci = permuted(cA, perm[L - 1], perm[L - 1])
    print(f"--- coarse: n {cA.shape[0]}")
    analyse("coarse A", nrp, nci, variants, rng=rng)


if __name__ == "__main__":
    main()
