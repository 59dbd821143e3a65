"""Emulation of the row-block distributed AIR V-cycle on the per-rank pieces produced by the
library's partitioner (csrc/host/dist.cpp), in numpy/scipy.  TEST INFRASTRUCTURE: it follows the
same operation order as oracle/air_ref.py::vcycle and is used to validate the pieces and their
halo plans (i) with all ranks hosted in one process and (ii) with one rank per process over a
gloo process group."""
import numpy as np
import scipy.sparse as sp

OP_ASSEMBLED, OP_DIAG, OP_HORNER = 1, 2, 3


class Piece:
    def __init__(self, items):
        self.d = dict(items)
        m = self.d["meta"]
        self.L = int(m[0]); self.rank = int(m[1]); self.nranks = int(m[2]); self.n_global = int(m[3])
        self.gid = self.d["gid"]
        self._mats = {}

    def has(self, name):
        return name + ".shape" in self.d

    def mat(self, name):
        if name not in self._mats:
            sh = self.d[name + ".shape"]
            self._mats[name] = sp.csr_matrix((self.d[name + ".v"], self.d[name + ".ci"], self.d[name + ".rp"]),
                                             shape=(int(sh[0]), int(sh[1])))
        return self._mats[name]

    def halo(self, name):
        return (self.d[name + ".halo.peers"], self.d[name + ".halo.send_off"], self.d[name + ".halo.recv_off"],
                self.d[name + ".halo.send_idx"])

    def level(self, l):
        m = self.d[f"L{l}.meta"]
        return dict(n=int(m[0]), n_f=int(m[1]), n_c=int(m[2]), aff_diag=bool(m[3]), one_point=bool(m[4]),
                    smooth=[int(s) for s in m[6:16]])

    def inv(self, name):
        m = self.d[name + ".meta"]
        return dict(kind=int(m[0]), diag_scale=bool(m[1]), neumann=bool(m[2]), coeffs=self.d[name + ".coeffs"],
                    d=self.d.get(name + ".d"))


class Cluster:
    """The ranks hosted by this process (all of them, or one with `dist` carrying the rest)."""

    def __init__(self, pieces, dist=None, group=None):
        self.p = {q.rank: q for q in pieces}
        self.dist = dist
        self.group = group
        self.nexch = 0

    def ghosts(self, name, xs):
        """ghost values of operand `name` for every hosted rank; xs: rank -> local input vector"""
        self.nexch += 1
        out = {}
        ops, pending = [], []
        for r, q in self.p.items():
            peers, so, ro, sidx = q.halo(name)
            g = np.zeros(int(ro[-1]) if len(ro) else 0)
            out[r] = g
            for k, pr in enumerate(peers):
                pr = int(pr)
                if pr in self.p:       # hosted peer: read straight from its plan
                    pp, pso, _, psidx = self.p[pr].halo(name)
                    kk = list(pp).index(r)
                    seg = psidx[pso[kk]:pso[kk + 1]]
                    assert len(seg) == ro[k + 1] - ro[k], (name, r, pr)
                    g[ro[k]:ro[k + 1]] = xs[pr][seg]
                else:
                    import torch
                    if so[k + 1] > so[k]:
                        t = torch.from_numpy(np.ascontiguousarray(xs[r][sidx[so[k]:so[k + 1]]]))
                        ops.append(self.dist.P2POp(self.dist.isend, t, pr, self.group))
                    if ro[k + 1] > ro[k]:
                        t = torch.empty(int(ro[k + 1] - ro[k]), dtype=torch.float64)
                        ops.append(self.dist.P2POp(self.dist.irecv, t, pr, self.group))
                        pending.append((g, int(ro[k]), int(ro[k + 1]), t))
        if ops:
            for w in self.dist.batch_isend_irecv(ops):
                w.wait()
        for g, a, b, t in pending:
            g[a:b] = t.numpy()
        return out

    def spmv(self, name, xs):
        gh = self.ghosts(name, xs)
        return {r: q.mat(name) @ np.concatenate([xs[r], gh[r]]) for r, q in self.p.items()}

    def apply_inv(self, name, matname, rs):
        out = {}
        kinds = {q.inv(name)["kind"] for q in self.p.values()}
        assert len(kinds) == 1
        kind = kinds.pop()
        if kind == OP_ASSEMBLED:
            return self.spmv(name + ".mat", rs)
        if kind == OP_DIAG:
            return {r: q.inv(name)["d"] * rs[r] for r, q in self.p.items()}
        if kind == OP_HORNER:    # plain monomial Horner on the operator `matname`
            any_q = next(iter(self.p.values()))
            c = any_q.inv(name)["coeffs"]
            assert not any_q.inv(name)["diag_scale"]
            y = {r: c[-1] * rs[r] for r in self.p}
            diag_name = matname.replace(".A_ff", ".aff_d")
            for k in range(len(c) - 2, -1, -1):
                if any_q.has(matname):
                    ay = self.spmv(matname, y)
                else:                       # the operator is a MATDIAGONAL A_ff
                    ay = {r: self.p[r].d[diag_name] * y[r] for r in self.p}
                y = {r: ay[r] + c[k] * rs[r] for r in self.p}
            return y
        raise NotImplementedError(kind)

    def vcycle(self, bs0):
        any_q = next(iter(self.p.values()))
        L = any_q.L
        bs = [bs0]
        for l in range(L - 1):
            F = {r: q.d[f"L{l}.F"] for r, q in self.p.items()}
            Cc = {r: q.d[f"L{l}.C"] for r, q in self.p.items()}
            zb = self.spmv(f"L{l}.Z", {r: bs[l][r][F[r]] for r in self.p})
            bs.append({r: bs[l][r][Cc[r]] + zb[r] for r in self.p})
        x = self.apply_inv("coarse.inv", "coarse.A", bs[L - 1])
        for l in range(L - 2, -1, -1):
            lv = any_q.level(l)
            assert lv["one_point"] and all(s >= 0 for s in lv["smooth"]), "emulator: F smoothing with one-point W only"
            its = sum(lv["smooth"])
            F = {r: q.d[f"L{l}.F"] for r, q in self.p.items()}
            Cc = {r: q.d[f"L{l}.C"] for r, q in self.p.items()}
            gh = self.ghosts(f"L{l}.A_fc", x)
            xc_ext = {r: np.concatenate([x[r], gh[r]]) for r in self.p}
            b_f = {r: bs[l][r][F[r]] for r in self.p}
            t4 = {r: b_f[r] - self.p[r].mat(f"L{l}.A_fc") @ xc_ext[r] for r in self.p}
            x_f = {}
            for r, q in self.p.items():
                wc = q.d[f"L{l}.wcol"]
                x_f[r] = np.where(wc >= 0, xc_ext[r][np.maximum(wc, 0)], 0.0)
            if any_q.has(f"L{l}.G"):
                # the pieces of the fused first sweep's operator G = A_fc + A_ff W and their halo plan: with x0 = W x_c (taken
                # through G's own column numbering, wcol_g)   b_f - G x_c   must be the reference's   t4 - A_ff x0
                ghg = self.ghosts(f"L{l}.G", x)
                self.nexch -= 1                                   # (a check, not an exchange of the cycle)
                ax0 = self.spmv(f"L{l}.A_ff", x_f)
                self.nexch -= 1
                for r, q in self.p.items():
                    xg = np.concatenate([x[r], ghg[r]])
                    wg = q.d[f"L{l}.wcol_g"]
                    np.testing.assert_array_equal(np.where(wg >= 0, xg[np.maximum(wg, 0)], 0.0), x_f[r])
                    r0 = b_f[r] - q.mat(f"L{l}.G") @ xg
                    ref = t4[r] - ax0[r]
                    scale = max(np.linalg.norm(b_f[r]), np.linalg.norm(ref), 1e-300)
                    assert np.linalg.norm(r0 - ref) <= 1e-12 * scale, (l, r, np.linalg.norm(r0 - ref), scale)
                self.fused_levels_checked = getattr(self, "fused_levels_checked", 0) + 1
            for _ in range(its):
                if lv["aff_diag"]:
                    res = {r: t4[r] - self.p[r].d[f"L{l}.aff_d"] * x_f[r] for r in self.p}
                else:
                    ax = self.spmv(f"L{l}.A_ff", x_f)
                    res = {r: t4[r] - ax[r] for r in self.p}
                dx = self.apply_inv(f"L{l}.inv_ff", f"L{l}.A_ff", res)
                x_f = {r: x_f[r] + dx[r] for r in self.p}
            xn = {}
            for r in self.p:
                v = np.empty(lv_n(self.p[r], l))
                v[F[r]] = x_f[r]; v[Cc[r]] = x[r]
                xn[r] = v
            x = xn
        return x


def lv_n(piece, l):
    return int(piece.d[f"L{l}.meta"][0])
