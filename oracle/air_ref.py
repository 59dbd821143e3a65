"""CPU restatement of PFLARE's PCAIR / PCPFLAREINV algorithms (setup + solve phase).

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (pflare_amd/, bench.py's timed GPU
leg) may import this module; it is the checker used by tests/, by __graft_entry__.smoke()
and by scripts that generate tests/golden/ fixtures.

It is a plain numpy/scipy restatement of the reference's *algorithms*, written from a
reading of the Fortran/C sources (citations are `file:line` into the reference tree):

  generators      tests/adv_1d.c:80-105, tests/adv_diff_fd.c:366-586
  drop / lump     src/PETSc_Helper.F90:207-412  (remove_small_from_sparse_cpu)
  strength        src/SAbs.F90:16-145
  PMISR           src/PMISR_Module.F90:121-671
  DDC             src/DDC_Module.F90:221-481, src/MatDiagDom.F90:98-270
  CF driver       src/CF_Splitting.F90:235-392
  Arnoldi coeffs  src/Gmres_Poly.F90:139-548
  power coeffs    src/Gmres_Poly.F90:552-773 (+ src/TSQR.F90: serial = one geqrf)
  Newton roots    src/Gmres_Poly_Newton.F90:21-712
  assembly        src/Gmres_Poly.F90:920-1337, 1522-1928
  Neumann         src/Neumann_Poly.F90:19-215
  Jacobi          src/Weighted_Jacobi.F90:15-86
  one-point W     src/Grid_Transfer.F90:94-220 ; P=[W;I] :329-461 ; R=[Z I] :588-815
  level loop      src/AIR_MG_Setup.F90:138-640, 924-1156 ; src/AIR_Operators_Setup.F90
  FC smoothing    src/FC_Smooth.F90:29-49, 421-640
  Horner / Newton src/Gmres_Poly.F90:1341-1518 ; src/Gmres_Poly_Newton.F90:716-912
  Kaskade cycle   PETSc PCMGKCycle_Private as wired by src/AIR_MG_Setup.F90:967-1156
  outer Krylov    PETSc KSPRICHARDSON / KSPGMRES (CGS, restart 30) semantics

PARITY STATUS: the reference cannot be built or imported in this environment (needs PETSc
>= 3.25 + a Fortran compiler) and draws from compiler-specific `random_number` streams, so
bit-level parity with the reference's own hierarchy is UNPINNED.  This restatement is pinned
by the reference's own known-answer tests (tests/Makefile iteration bounds, exact-solver and
constant-diagonal properties) - see tests/test_oracle_reference_bounds.py.

The random streams are replaced by a documented counter-based generator (splitmix64) with
the same seeding convention (seed = rank + 1) and the same consumption order.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import scipy.linalg.lapack as lapack
import scipy.sparse as sp

F_POINT = -1
C_POINT = 1

# src/Pflare_Parameters.F90:204-226 (double branch).  Literals written `1e-12` in the
# Fortran are default-real (single) literals widened to double; `1d-13` are double.
_f32 = lambda v: float(np.float32(v))
TOL_ZERO = _f32(1e-12)
TOL_RCOND = _f32(1e-12)
TOL_ARNOLDI = _f32(1e-14)
TOL_LUCKY = 1e-30
TOL_LEJA_PERTURB = _f32(5e-8)
TOL_AUTO_TRUNCATE = _f32(1e-14)
EPS = float(np.finfo(np.float64).eps)

# PCPFLAREINVType (include/pflare.h)
INV_POWER, INV_ARNOLDI, INV_NEWTON, INV_NEUMANN, INV_SAI, INV_ISAI, INV_WJACOBI, INV_JACOBI, \
    INV_NEWTON_NO_EXTRA = 0, 1, 2, 3, 4, 5, 6, 7, 8
INV_NAMES = {"power": INV_POWER, "arnoldi": INV_ARNOLDI, "newton": INV_NEWTON,
             "neumann": INV_NEUMANN, "wjacobi": INV_WJACOBI, "jacobi": INV_JACOBI,
             "newton_no_extra": INV_NEWTON_NO_EXTRA, "sai": INV_SAI, "isai": INV_ISAI}


# ----------------------------------------------------------------------------------------
# Random numbers: counter-based splitmix64.  U(seed, i) in [0,1).
# ----------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(seed: int, start: int, count: int) -> np.ndarray:
    """count numbers U(seed, start..start+count-1), each = top 53 bits * 2^-53."""
    key = _splitmix64(np.array([seed], dtype=np.uint64))[0]
    idx = np.arange(start, start + count, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = _splitmix64(idx + key)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def box_muller(seed: int, n: int) -> np.ndarray:
    """src/Gmres_Poly.F90:189-208: random_data(n,2) filled column-major, then
    sqrt(-2 log u1) cos(2 pi u2)."""
    u1 = uniform(seed, 0, n)
    u2 = uniform(seed, n, n)
    u1 = np.where(u1 == 0.0, 1.0 / 9007199254740992.0, u1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)


# ----------------------------------------------------------------------------------------
# Synthetic operators
# ----------------------------------------------------------------------------------------
def _assemble(n, rows, cols, vals):
    rows = np.concatenate(rows)
    cols = np.concatenate(cols)
    vals = np.concatenate(vals)
    keep = vals != 0.0  # MAT_IGNORE_ZERO_ENTRIES (tests/adv_diff_fd.c:173)
    A = sp.coo_matrix((vals[keep], (rows[keep], cols[keep])), shape=(n, n)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    return A


def gen_adv_1d(n: int) -> sp.csr_matrix:
    """tests/adv_1d.c:80-105."""
    i = np.arange(1, n)
    rows = [np.array([0]), i, i]
    cols = [np.array([0]), i - 1, i]
    vals = [np.array([1.0]), -np.ones(n - 1), np.ones(n - 1)]
    return _assemble(n, rows, cols, vals)


def gen_adv_diff_fd(dim=2, M=11, N=11, P=11, u=None, v=None, w=None, alpha=0.0,
                    L_x=1.0, L_y=1.0, L_z=1.0, theta=None, adv_nondim=None, curved_velocity=False,
                    unit_velocity_opt=True) -> sp.csr_matrix:
    """tests/adv_diff_fd.c:188-257 (option handling) and :366-586 (ComputeMat).
    Row numbering is the natural (k*N + j)*M + i of a single-rank DMDA."""
    uu, vv, ww = 1.0, 1.0, 1.0
    if theta is not None:
        uu, vv, ww = math.cos(theta), math.sin(theta), 0.0
    unit_velocity = bool(unit_velocity_opt)      # -unit_velocity (default true), tests/adv_diff_fd.c:224-231
    if u is not None or v is not None or w is not None:
        unit_velocity = False
    if u is not None:
        uu = float(u)
    if v is not None:
        vv = float(v)
    if w is not None:
        ww = float(w)
    if adv_nondim is None:
        adv_nondim = (alpha == 0.0)
    if curved_velocity:
        return _gen_adv_diff_fd_curved(dim, M, N, P, alpha, L_x, L_y, L_z, adv_nondim, unit_velocity)
    vel = [uu, vv, ww]
    if unit_velocity:
        mag = math.sqrt(sum(vel[d] * vel[d] for d in range(dim)))
        if mag > 1e-12:
            for d in range(dim):
                vel[d] /= mag
    u_loc, v_loc, w_loc = vel

    Hx = L_x / (M - 1)
    Hy = L_y / (N - 1)
    Hz = L_z / (P - 1) if dim == 3 else 1.0
    rows, cols, vals = [], [], []

    if dim == 2:
        HxdHy, HydHx = Hx / Hy, Hy / Hx
        adv_x_scale, adv_y_scale = Hx, Hy
        if adv_nondim:
            adv_x_scale, adv_y_scale = 1.0, HydHx
        jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
        ii = ii.ravel(); jj = jj.ravel()
        rid = jj * M + ii
        inflow = (ii == 0) | (jj == 0)
        outflow = ((ii == M - 1) | (jj == N - 1)) & ~inflow
        interior = ~(inflow | outflow)

        def add(mask, di, dj, val):
            r = rid[mask]
            rows.append(r); cols.append((jj[mask] + dj) * M + (ii[mask] + di))
            vals.append(np.full(r.size, val))

        add(inflow, 0, 0, 1.0)
        if alpha == 0.0:
            adv_mask = outflow | interior
        else:
            add(outflow, 0, 0, 1.0)
            adv_mask = interior
            add(interior, 0, -1, -alpha * HxdHy)
            add(interior, -1, 0, -alpha * HydHx)
            add(interior, 0, 0, alpha * 2.0 * (HxdHy + HydHx))
            add(interior, 1, 0, -alpha * HydHx)
            add(interior, 0, 1, -alpha * HxdHy)
        if u_loc != 0.0 or v_loc != 0.0 or alpha == 0.0:
            add(adv_mask, 0, -1, -v_loc * adv_x_scale)
            add(adv_mask, -1, 0, -u_loc * adv_y_scale)
            add(adv_mask, 0, 0, u_loc * adv_y_scale + v_loc * adv_x_scale)
        return _assemble(M * N, rows, cols, vals)

    HyHz_Hx, HxHz_Hy, HxHy_Hz = Hy * Hz / Hx, Hx * Hz / Hy, Hx * Hy / Hz
    adv_yz, adv_xz, adv_xy = Hy * Hz, Hx * Hz, Hx * Hy
    if adv_nondim:
        adv_yz, adv_xz, adv_xy = HyHz_Hx / Hx, Hz / Hx, Hy / Hx
    kk, jj, ii = np.meshgrid(np.arange(P), np.arange(N), np.arange(M), indexing="ij")
    ii = ii.ravel(); jj = jj.ravel(); kk = kk.ravel()
    rid = (kk * N + jj) * M + ii
    inflow = (ii == 0) | (jj == 0) | (kk == 0)
    outflow = ((ii == M - 1) | (jj == N - 1) | (kk == P - 1)) & ~inflow
    interior = ~(inflow | outflow)

    def add3(mask, di, dj, dk, val):
        r = rid[mask]
        rows.append(r)
        cols.append(((kk[mask] + dk) * N + (jj[mask] + dj)) * M + (ii[mask] + di))
        vals.append(np.full(r.size, val))

    add3(inflow, 0, 0, 0, 1.0)
    if alpha == 0.0:
        adv_mask = outflow | interior
    else:
        add3(outflow, 0, 0, 0, 1.0)
        adv_mask = interior
        add3(interior, 0, 0, -1, -alpha * HxHy_Hz)
        add3(interior, 0, -1, 0, -alpha * HxHz_Hy)
        add3(interior, -1, 0, 0, -alpha * HyHz_Hx)
        add3(interior, 0, 0, 0, alpha * 2.0 * (HyHz_Hx + HxHz_Hy + HxHy_Hz))
        add3(interior, 1, 0, 0, -alpha * HyHz_Hx)
        add3(interior, 0, 1, 0, -alpha * HxHz_Hy)
        add3(interior, 0, 0, 1, -alpha * HxHy_Hz)
    if u_loc != 0.0 or v_loc != 0.0 or w_loc != 0.0 or alpha == 0.0:
        add3(adv_mask, 0, 0, -1, -w_loc * adv_xy)
        add3(adv_mask, 0, -1, 0, -v_loc * adv_xz)
        add3(adv_mask, -1, 0, 0, -u_loc * adv_yz)
        add3(adv_mask, 0, 0, 0, u_loc * adv_yz + v_loc * adv_xz + w_loc * adv_xy)
    return _assemble(M * N * P, rows, cols, vals)


def _gen_adv_diff_fd_curved(dim, M, N, P, alpha, L_x, L_y, L_z, adv_nondim, unit_velocity):
    """-curved_velocity (tests/adv_diff_fd.c:64-98): node-wise velocity u = y, v = 1 - x (2D) or u = v = z,
    w = 2 - x - y (3D), normalised per node with -unit_velocity (the default); same stencils as above."""
    Hx, Hy = L_x / (M - 1), L_y / (N - 1)
    Hz = L_z / (P - 1) if dim == 3 else 1.0
    rows, cols, vals = [], [], []
    if dim == 2:
        jj, ii = np.meshgrid(np.arange(N), np.arange(M), indexing="ij")
        ii = ii.ravel(); jj = jj.ravel(); kk = np.zeros_like(ii)
        vel = [jj * Hy, 1.0 - ii * Hx, np.zeros(ii.size)]
    else:
        kk, jj, ii = np.meshgrid(np.arange(P), np.arange(N), np.arange(M), indexing="ij")
        ii = ii.ravel(); jj = jj.ravel(); kk = kk.ravel()
        vel = [kk * Hz, kk * Hz, 2.0 - ii * Hx - jj * Hy]
    vel = [np.asarray(c, dtype=np.float64) for c in vel]
    if unit_velocity:
        mag = np.sqrt(sum(vel[d] ** 2 for d in range(dim)))
        ok = mag > 1e-12
        for d in range(dim):
            vel[d] = np.where(ok, vel[d] / np.where(ok, mag, 1.0), vel[d])
    ul, vl, wl = vel
    rid = (kk * N + jj) * M + ii

    def add(mask, di, dj, dk, val):
        r = rid[mask]
        rows.append(r)
        cols.append(((kk[mask] + dk) * N + (jj[mask] + dj)) * M + (ii[mask] + di))
        vals.append(np.full(r.size, val) if np.isscalar(val) else np.asarray(val)[mask])

    if dim == 2:
        HxdHy, HydHx = Hx / Hy, Hy / Hx
        ax, ay = (1.0, HydHx) if adv_nondim else (Hx, Hy)
        inflow = (ii == 0) | (jj == 0)
        outflow = ((ii == M - 1) | (jj == N - 1)) & ~inflow
        interior = ~(inflow | outflow)
        add(inflow, 0, 0, 0, 1.0)
        if alpha == 0.0:
            adv_mask = outflow | (interior & ((ul != 0.0) | (vl != 0.0)))
        else:
            add(outflow, 0, 0, 0, 1.0)
            adv_mask = interior & ((ul != 0.0) | (vl != 0.0))
            add(interior, 0, -1, 0, -alpha * HxdHy)
            add(interior, -1, 0, 0, -alpha * HydHx)
            add(interior, 0, 0, 0, alpha * 2.0 * (HxdHy + HydHx))
            add(interior, 1, 0, 0, -alpha * HydHx)
            add(interior, 0, 1, 0, -alpha * HxdHy)
        add(adv_mask, 0, -1, 0, -vl * ax)
        add(adv_mask, -1, 0, 0, -ul * ay)
        add(adv_mask, 0, 0, 0, ul * ay + vl * ax)
        return _assemble(M * N, rows, cols, vals)
    HyHz_Hx, HxHz_Hy, HxHy_Hz = Hy * Hz / Hx, Hx * Hz / Hy, Hx * Hy / Hz
    a_yz, a_xz, a_xy = (HyHz_Hx / Hx, Hz / Hx, Hy / Hx) if adv_nondim else (Hy * Hz, Hx * Hz, Hx * Hy)
    inflow = (ii == 0) | (jj == 0) | (kk == 0)
    outflow = ((ii == M - 1) | (jj == N - 1) | (kk == P - 1)) & ~inflow
    interior = ~(inflow | outflow)
    moving = (ul != 0.0) | (vl != 0.0) | (wl != 0.0)
    add(inflow, 0, 0, 0, 1.0)
    if alpha == 0.0:
        adv_mask = outflow | (interior & moving)
    else:
        add(outflow, 0, 0, 0, 1.0)
        adv_mask = interior & moving
        add(interior, 0, 0, -1, -alpha * HxHy_Hz)
        add(interior, 0, -1, 0, -alpha * HxHz_Hy)
        add(interior, -1, 0, 0, -alpha * HyHz_Hx)
        add(interior, 0, 0, 0, alpha * 2.0 * (HyHz_Hx + HxHz_Hy + HxHy_Hz))
        add(interior, 1, 0, 0, -alpha * HyHz_Hx)
        add(interior, 0, 1, 0, -alpha * HxHz_Hy)
        add(interior, 0, 0, 1, -alpha * HxHy_Hz)
    add(adv_mask, 0, 0, -1, -wl * a_xy)
    add(adv_mask, 0, -1, 0, -vl * a_xz)
    add(adv_mask, -1, 0, 0, -ul * a_yz)
    add(adv_mask, 0, 0, 0, ul * a_yz + vl * a_xz + wl * a_xy)
    return _assemble(M * N * P, rows, cols, vals)


# ----------------------------------------------------------------------------------------
# Sparse helpers
# ----------------------------------------------------------------------------------------
def _rows_of(A):
    return np.repeat(np.arange(A.shape[0]), np.diff(A.indptr))


def _row_max(absv, indptr, fill):
    """max over each CSR row segment; empty rows get `fill`."""
    n = indptr.size - 1
    out = np.full(n, fill, dtype=np.float64)
    nz = np.diff(indptr) > 0
    if absv.size:
        red = np.maximum.reduceat(absv, indptr[:-1][nz])
        out[nz] = red
    return out


def remove_small(A, tol, rel=0, lump=False, drop_diag=0):
    """src/PETSc_Helper.F90:207-412.
    rel: 1 relative to row max |.| incl. diagonal, -1 excl. diagonal, 0 absolute.
    drop_diag: 0 never drop, 1 allow, -1 always drop.  Entries with |v| >= tol_row are
    kept; smaller ones are lumped onto (row,row) if lump (or if they are the diagonal and
    drop_diag == 0).  The `col == row` test is on global indices even for rectangular
    input (quirk noted in SURVEY Appendix A.10)."""
    A = A.tocsr()
    rows = _rows_of(A)
    cols = A.indices
    vals = A.data
    absv = np.abs(vals)
    isdiag = cols == rows
    if rel == 1:
        rowtol = tol * _row_max(absv, A.indptr, 0.0)
    elif rel == -1:
        big = np.finfo(np.float64).max
        rowtol = tol * _row_max(np.where(isdiag, -big, absv), A.indptr, -big)
    else:
        rowtol = np.full(A.shape[0], tol)
    big_enough = absv >= rowtol[rows]
    keep = big_enough.copy()
    if drop_diag == -1:
        keep &= ~isdiag
    to_diag = (~big_enough) & (lump | ((drop_diag == 0) & isdiag))
    r = np.concatenate([rows[keep], rows[to_diag]])
    c = np.concatenate([cols[keep], rows[to_diag]])
    v = np.concatenate([vals[keep], vals[to_diag]])
    B = sp.coo_matrix((v, (r, c)), shape=A.shape).tocsr()
    B.sum_duplicates()
    B.sort_indices()
    return B


def generate_sabs(A, strong_threshold, symmetrize):
    """src/SAbs.F90:16-145 (square=False path)."""
    S = remove_small(A, strong_threshold, rel=-1, lump=False, drop_diag=-1)
    S.data[:] = 1.0
    if symmetrize:
        S = (S + S.T).tocsr()
        S.data[:] = 1.0
    S.sort_indices()
    return S


def pmisr(S, seed=1, max_luby_steps=-1):
    """src/PMISR_Module.F90:121-671 (serial; synchronous Luby rounds, smallest measure -> F;
    after max_luby_steps > 0 rounds whatever is undecided becomes C)."""
    n = S.shape[0]
    indptr, indices = S.indptr, S.indices
    measure = uniform(seed, 0, n) + np.diff(indptr)
    markers = np.zeros(n, dtype=np.int32)
    assigned = np.zeros(n, dtype=bool)
    lonely = np.abs(measure) < 1
    assigned[lonely] = True
    markers[lonely] = F_POINT
    ei = _rows_of(S)
    ej = indices
    loops = 0
    while not assigned.all():
        if max_luby_steps > 0 and loops == max_luby_steps:
            break
        loops += 1
        live = ~assigned[ei] & ~assigned[ej]
        li, lj = ei[live], ej[live]
        worse = (measure[li] > measure[lj]) | ((measure[li] == measure[lj]) & (li > lj))
        bad = np.zeros(n, dtype=bool)
        bad[li[worse]] = True
        in_set = ~assigned & ~bad
        markers[in_set] = F_POINT
        assigned[in_set] = True
        nb = in_set[ei]
        assigned[ej[nb]] = True
    markers[markers == 0] = C_POINT
    return markers


def diag_dom_ratio(A, markers):
    """src/MatDiagDom.F90:98-270: per F row, sum |a_ij| over F neighbours j != i, over |a_ii|."""
    F = np.flatnonzero(markers == F_POINT)
    rows = _rows_of(A)
    cols = A.indices
    absv = np.abs(A.data)
    sel = (markers[rows] == F_POINT) & (markers[cols] == F_POINT)
    r, c, v = rows[sel], cols[sel], absv[sel]
    n = A.shape[0]
    diag = np.zeros(n)
    isd = r == c
    diag[r[isd]] = v[isd]
    # sequential (column-order) accumulation of the off-diagonal sum, like the reference loop
    off = np.zeros(n)
    np.add.at(off, r[~isd], v[~isd])
    ratio = np.zeros(n)
    nzd = diag != 0.0
    ratio[nzd] = off[nzd] / diag[nzd]
    return F, ratio[F]


def ddc(A, markers, fraction_swap):
    """src/DDC_Module.F90:221-481 (fraction path; no max_dd_ratio trigger)."""
    if fraction_swap == 0.0:
        return markers
    F, ratio = diag_dom_ratio(A, markers)
    fine_size = F.size
    if fraction_swap < 0:
        search_size = fine_size
    else:
        search_size = int(float(fine_size) * fraction_swap)
    if search_size <= 0:
        return markers
    nb = 1000
    with np.errstate(invalid="ignore", over="ignore"):
        b = np.floor(ratio * nb) + 1
    b = np.where(np.isfinite(b), b, nb)
    b = np.minimum(b, nb).astype(np.int64)
    b[b < 0] = nb
    bins = np.bincount(b, minlength=nb + 1)
    if fraction_swap < 0:
        swap_val = -fraction_swap
    else:
        bin_sum = 0
        bb = nb
        for bb in range(nb, 0, -1):
            bin_sum += bins[bb]
            if bin_sum >= search_size:
                break
        swap_val = float(bb - 1) / float(nb)
    swap = ~((ratio == 0) | (ratio < swap_val))
    markers = markers.copy()
    markers[F[swap]] *= -1
    return markers


def _swap_independent_set(A, markers, cand, measure):
    """pmisr_existing_measure_implicit_transpose on A_ff + A_ff^T (src/PMISR_Module.F90): among the candidate F rows
    repeatedly take the local minima of `measure` (ties: the larger index loses) and block their neighbours; the rows
    taken become C points.  Returns (markers, number swapped)."""
    n = A.shape[0]
    assigned = np.ones(n, dtype=bool)
    assigned[cand] = False
    rows = _rows_of(A); cols = A.indices
    sel = (markers[rows] == F_POINT) & (markers[cols] == F_POINT) & (rows != cols)
    ei = np.concatenate([rows[sel], cols[sel]]); ej = np.concatenate([cols[sel], rows[sel]])
    swap = np.zeros(n, dtype=bool)
    while not assigned.all():
        live = ~assigned[ei] & ~assigned[ej]
        li, lj = ei[live], ej[live]
        worse = (measure[li] > measure[lj]) | ((measure[li] == measure[lj]) & (li > lj))
        bad = np.zeros(n, dtype=bool)
        bad[li[worse]] = True
        in_set = ~assigned & ~bad
        swap |= in_set
        assigned[in_set] = True
        assigned[ej[in_set[ei]]] = True
    markers = markers.copy()
    markers[swap] *= -1
    return markers, int(swap.sum())


CR_NU, CR_NU_POLY, CR_MAX_ITS, CR_CANDIDATE = 10, 1, 100, 0.1     # src/Pflare_Parameters.F90:44-62


def cr_pass(A, markers, target, inverse_type, poly_order, sparsity_order, diag_scale, seed):
    """One pass of compatible-relaxation coarsening (src/CR_Splitting.F90:36-336): relax A_ff e = 0 from a random e
    with the (always assembled) A_ff inverse AIRG would smooth with; if the rate (||e_nu|| / ||e_0||)^(1/nu) is above
    the target, promote an independent set of the rows where the error stayed largest."""
    F = np.flatnonzero(markers == F_POINT)
    if F.size == 0:
        return markers, 0.0, 0
    n = A.shape[0]
    Aff = A if F.size == n else submatrix(A, F, F)
    forced = np.zeros(F.size, dtype=bool)
    use_poly = inverse_type not in (INV_WJACOBI, INV_JACOBI) and F.size > poly_order + 1
    nu = CR_NU_POLY if use_poly else CR_NU
    if use_poly:
        M, _ = build_inverse(Aff, inverse_type, poly_order, sparsity_order, False, diag_scale, seed)
        apply_m = lambda v: apply_inv(M, v)
    else:
        d = np.asarray(Aff.diagonal(), dtype=np.float64).copy()
        forced = d == 0.0
        d[forced] = 1.0
        weight = 1.0
        if inverse_type != INV_JACOBI:
            sc = 1.0 / np.sqrt(np.abs(d))
            T = sp.diags(sc) @ Aff @ sp.diags(sc)
            norm_inf = float(np.max(np.abs(T).sum(axis=1))) if T.nnz else 0.0
            if norm_inf != 0.0:
                weight = 3.0 / (4.0 * norm_inf)
        minv = weight / d
        minv[forced] = 0.0
        apply_m = lambda v: minv * v
    rnd = uniform(seed + 15485863, 0, n)
    e = rnd[F].copy()
    norm_e0 = float(np.linalg.norm(e))
    for _ in range(nu):
        e = e - apply_m(Aff @ e)
    norm_enu = float(np.linalg.norm(e))
    rate = (norm_enu / norm_e0) ** (1.0 / nu) if norm_e0 != 0.0 else 0.0
    if not rate > target:
        return markers, rate, 0
    e_inf = float(np.max(np.abs(e)))
    cm = np.where(forced, 2.0, (np.abs(e) / e_inf) if e_inf != 0.0 else 0.0)
    is_cand = cm >= CR_CANDIDATE
    measure = np.full(n, np.inf)
    measure[F[is_cand]] = 10.0 - (cm[is_cand] - rnd[F[is_cand]] / 1e10)
    markers, nsw = _swap_independent_set(A, markers, F[is_cand], measure)
    return markers, rate, nsw


def ddc_diag_dom(A, markers, target, seed):
    """The max_dd_ratio path of ddc / ddc_cpu (src/DDC_Module.F90:56-82, 301-392): while some F row is less
    diagonally dominant in A_ff than `target`, swap to C an independent set (in A_ff + A_ff^T) of the worst of those
    rows - PMISR on the measure max(10, 2 max_ratio) - (ratio - rand * 1e-10), rows already below the target
    excluded.  Returns (markers, max ratio found before the swap)."""
    F, ratio_f = diag_dom_ratio(A, markers)
    max_ratio = float(ratio_f.max()) if F.size else 0.0
    if max_ratio < target:
        return markers, max_ratio
    n = A.shape[0]
    ratio = np.zeros(n); ratio[F] = ratio_f
    rnd = uniform(seed + 104729, 0, n)
    measure = np.full(n, np.inf)
    measure[F] = max(10.0, 2.0 * max_ratio) - (ratio[F] - rnd[F] / 1e10)
    markers, _ = _swap_independent_set(A, markers, F[ratio_f >= target], measure)
    return markers, max_ratio


def compute_cf_splitting(A, strong_threshold=0.5, ddc_its=1, ddc_fraction=_f32(0.1),
                         symmetric=False, seed=1, max_luby_steps=-1, cf_splitting_type="pmisr_ddc", cr=None):
    """src/CF_Splitting.F90:235-392 for CF_PMISR_DDC and CF_DIAG_DOM (the latter repeats the cleanup until every
    F row reaches the diagonal dominance ratio given by strong_threshold, :361-392)."""
    if cf_splitting_type == "cr":
        # CF_CR (src/CF_Splitting.F90:303-340): everything starts as F, strong_threshold is the target CR rate
        markers = np.full(A.shape[0], F_POINT, dtype=np.int32)
        it, po, so, ds = cr if cr is not None else (INV_ARNOLDI, 6, 1, False)
        for _ in range(CR_MAX_ITS):
            markers, rate, nsw = cr_pass(A, markers, strong_threshold, it, po, so, ds, seed)
            if rate <= strong_threshold or nsw == 0:
                break
        F = np.flatnonzero(markers == F_POINT).astype(np.int64)
        C = np.flatnonzero(markers != F_POINT).astype(np.int64)
        return F, C
    S = generate_sabs(A, strong_threshold, not symmetric)
    markers = pmisr(S, seed, max_luby_steps)
    if strong_threshold != 0.0 and cf_splitting_type == "diag_dom":
        while True:
            markers, achieved = ddc_diag_dom(A, markers, strong_threshold, seed)
            if achieved < strong_threshold:
                break
    elif strong_threshold != 0.0:
        for _ in range(ddc_its):
            markers = ddc(A, markers, ddc_fraction)
    F = np.flatnonzero(markers == F_POINT).astype(np.int64)
    C = np.flatnonzero(markers != F_POINT).astype(np.int64)
    return F, C


def submatrix(A, rows, cols):
    B = A[rows][:, cols].tocsr()
    B.sort_indices()
    return B


def copy_plus_diag(A):
    """src/PETSc_Helper.F90:930-1040: copy with zero diagonal entries added where absent."""
    n = A.shape[0]
    A = A.tocoo()
    r = np.concatenate([A.row, np.arange(n)])
    c = np.concatenate([A.col, np.arange(n)])
    v = np.concatenate([A.data, np.zeros(n)])
    B = sp.coo_matrix((v, (r, c)), shape=A.shape).tocsr()
    B.sum_duplicates()
    B.sort_indices()
    return B


def values_on_pattern(prod, pat):
    """values of `prod` at the positions of `pat` (0 where prod has no entry)."""
    prod = prod.tocsr(); prod.sort_indices()
    ncol = np.int64(pat.shape[1])
    kp = _rows_of(pat).astype(np.int64) * ncol + pat.indices
    kq = _rows_of(prod).astype(np.int64) * ncol + prod.indices
    idx = np.searchsorted(kq, kp)
    idx_c = np.minimum(idx, max(kq.size - 1, 0))
    hit = (idx < kq.size) & (kq[idx_c] == kp) if kq.size else np.zeros(kp.size, bool)
    out = np.zeros(kp.size)
    out[hit] = prod.data[idx_c[hit]]
    return out


# ----------------------------------------------------------------------------------------
# Polynomial approximate inverses
# ----------------------------------------------------------------------------------------
@dataclass
class InvOp:
    """An approximate inverse as the solve phase sees it.
    kind: 'assembled' (csr `mat`), 'diag' (vector `d`, a MATDIAGONAL), 'horner', 'newton'."""
    kind: str
    mat: sp.csr_matrix | None = None
    d: np.ndarray | None = None
    A: object = None               # operator the MF polynomial multiplies with (csr or DiagMat)
    coeffs: np.ndarray | None = None
    re: np.ndarray | None = None
    im: np.ndarray | None = None
    diag: np.ndarray | None = None  # MF_VEC_DIAG when diagonally scaled (q(D^-1 A) D^-1)
    neumann: bool = False           # scaled operator is I - D^-1 A instead of D^-1 A
    spmvs: int = 0                  # SpMVs one application executes (for algorithmic bytes)


class DiagMat:
    """MATDIAGONAL stand-in (src/AIR_MG_Setup.F90:481-496)."""

    def __init__(self, d):
        self.d = np.asarray(d, dtype=np.float64)
        self.shape = (self.d.size, self.d.size)

    def __matmul__(self, x):
        return self.d * x

    def diagonal(self):
        return self.d

    def tocsr(self):
        return sp.diags(self.d).tocsr()

    @property
    def nnz(self):
        return self.d.size


def _matvec(A, x):
    return A @ x


def arnoldi(matvec, r0, poly_order, lucky_tol, rel_tol=-1.0, compute_cn=False):
    """src/Gmres_Poly.F90:308-456.  Returns beta, H (order+2, order+1), m, C, y."""
    sub = poly_order + 1
    H = np.zeros((poly_order + 2, poly_order + 1))
    Cn = np.zeros((poly_order + 2, poly_order + 2)) if compute_cn else None
    y = np.zeros(poly_order + 1)
    V = [None] * (sub + 1)
    beta = float(np.linalg.norm(r0))
    V[0] = r0 / beta
    if compute_cn:
        Cn[0, 0] = 1.0 / beta
    m_out = sub
    for m in range(1, sub + 1):
        w = matvec(V[m - 1])
        if compute_cn:
            c_j = np.zeros(poly_order + 2)
            c_j[1:m + 1] = Cn[0:m, m - 1]
        h = np.array([float(np.dot(w, V[i])) for i in range(m)])
        H[0:m, m - 1] = h
        for i in range(m):
            w = w - h[i] * V[i]
        if compute_cn:
            for i in range(m):
                c_j[0:i + 1] -= Cn[0:i + 1, i] * H[i, m - 1]
        H[m, m - 1] = float(np.linalg.norm(w))
        if H[m, m - 1] < lucky_tol:
            if rel_tol > 0:
                y = _ls_solve_arnoldi(beta, m, H)
            m_out = m
            break
        V[m] = w / H[m, m - 1]
        if compute_cn:
            Cn[0:m + 1, m] = c_j[0:m + 1] / H[m, m - 1]
        if rel_tol > 0:
            y = _ls_solve_arnoldi(beta, m, H)
            g0 = H[0:m + 1, 0:m] @ y[0:m]
            g0[0] -= beta
            achieved = float(np.linalg.norm(g0[0:m])) / beta
            if achieved < rel_tol:
                m_out = m
                break
    return beta, H, m_out, Cn, y


def _ls_solve_arnoldi(beta, m, H):
    """src/Gmres_Poly.F90:245-304 (LAPACK gels on the (m+1) x m Hessenberg)."""
    g0 = np.zeros((m + 1, 1))
    g0[0, 0] = beta
    Hc = np.array(H[0:m + 1, 0:m], order="F", copy=True)
    _, x, info = lapack.dgels(Hc, g0)
    if info != 0:
        raise RuntimeError("LS solve failed")
    y = np.zeros(H.shape[1])
    y[0:m] = x[0:m, 0]
    return y


def coeffs_arnoldi(matvec, n, poly_order, seed=1):
    """src/Gmres_Poly.F90:460-548: monomial coefficients c = C(1:m,1:m) y."""
    r0 = box_muller(seed, n)
    beta, H, m, Cn, y = arnoldi(matvec, r0, poly_order, TOL_LUCKY, TOL_ARNOLDI, compute_cn=True)
    c = np.zeros(poly_order + 1)
    c[0:m] = Cn[0:m, 0:m] @ y[0:m]
    return c


def coeffs_power(matvec, n, poly_order, seed=1):
    """src/Gmres_Poly.F90:552-773: K_{m+1}, R of its QR (geqrf), min ||R(:,2:) g - R(1,1) e1||
    by gelsd with rcond=-1."""
    sub = poly_order + 1
    K = np.zeros((n, sub + 1), order="F")
    K[:, 0] = box_muller(seed, n)
    for i in range(sub):
        K[:, i + 1] = matvec(K[:, i])
    qr, tau, _, info = lapack.dgeqrf(K)
    R = np.zeros((sub + 1, sub + 1))
    k = min(n, sub + 1)
    R[0:k, :] = np.triu(qr[0:k, :])
    g0 = np.zeros(sub + 1)
    g0[0] = R[0, 0]
    x, s, rank, info = _gelsd(R[:, 1:], g0, -1.0)
    return x[0:sub].copy()


def _gelsd(a, b, rcond):
    """LAPACK gelsd (SVD-based minimum-norm least squares)."""
    a = np.array(a, order="F", dtype=np.float64, copy=True)
    m, n = a.shape
    b2 = np.zeros((max(m, n), 1))
    b2[0:len(b), 0] = b
    work, iwork, info = lapack.dgelsd_lwork(m, n, 1, rcond)
    x, s, rank, info = lapack.dgelsd(a, b2, int(work), int(iwork), rcond)
    if info != 0:
        raise RuntimeError("gelsd failed")
    return x[:, 0], s, rank, info


def modified_leja(re, im):
    """src/Gmres_Poly_Newton.F90:21-141.  Returns 0-based index permutation."""
    n = re.size
    mag = np.sqrt(re ** 2 + im ** 2)
    order = []
    done = np.zeros(n, dtype=bool)

    def push(i):
        order.append(i); done[i] = True
        if im[i] != 0.0:
            j = i + 1 if im[i] > 0 else i - 1
            order.append(j); done[j] = True

    push(int(np.argmax(mag)))
    while len(order) < n:
        best, best_i = -np.finfo(float).max, -1
        for i in range(n):
            if done[i]:
                continue
            val = 1.0
            with np.errstate(divide="ignore"):
                for k in order:
                    a = re[i] - re[k]; b = im[i] - im[k]
                    val += np.log10(np.sqrt(a * a + b * b))
            if val > best:
                best, best_i = val, i
        if best < 0:
            best_i = int(np.flatnonzero(~done)[0])
        push(best_i)
    return np.array(order, dtype=np.int64)


def cluster_eigenvalues_stable(re, im, rel_tol, abs_tol):
    """src/Gmres_Poly_Newton.F90:145-245."""
    n = re.size
    used = np.zeros(n, dtype=bool)
    rt, it = [], []
    for i in range(n):
        if used[i]:
            continue
        if re[i] == 0.0 and im[i] == 0.0:
            used[i] = True
            continue
        sr, si, cnt = re[i], im[i], 1
        used[i] = True
        mag_i = math.sqrt(re[i] ** 2 + im[i] ** 2)
        for j in range(i + 1, n):
            if used[j]:
                continue
            if re[j] == 0.0 and im[j] == 0.0:
                used[j] = True
                continue
            mag_j = math.sqrt(re[j] ** 2 + im[j] ** 2)
            dist = math.sqrt((re[j] - re[i]) ** 2 + (im[j] - im[i]) ** 2)
            scale = max(mag_i, mag_j, 1.0)
            if dist <= abs_tol + rel_tol * scale:
                sr += re[j]; si += im[j]; cnt += 1
                used[j] = True
        rt.append(sr / cnt); it.append(si / cnt)
    re_o = np.zeros(n); im_o = np.zeros(n)
    re_o[0:len(rt)] = rt; im_o[0:len(it)] = it
    return re_o, im_o


def compute_extra_roots(re, im):
    """src/Gmres_Poly_Newton.F90:250-374."""
    n = re.size
    pof = np.ones(n)
    extra = np.zeros(n, dtype=np.int64)
    overflow = np.zeros(n, dtype=np.int64)
    total_extra = 0
    for k in range(n):
        a, b = re[k], im[k]
        if b < 0:
            continue
        if abs(a) < TOL_ZERO:
            continue
        if a * a + b * b < TOL_ZERO:
            continue
        for i in range(n):
            if i == k:
                continue
            c, d = re[i], im[i]
            if abs(c) < TOL_ZERO:
                continue
            if c * c + d * d < TOL_ZERO:
                continue
            div_real = (a * c + b * d) / (c * c + d * d)
            div_imag = (b * c - a * d) / (c * c + d * d)
            div_mag = math.sqrt((1 - div_real) ** 2 + div_imag ** 2)
            with np.errstate(divide="ignore"):
                if math.log10(pof[k]) + (math.log10(div_mag) if div_mag > 0 else -math.inf) > 307:
                    overflow[k] += int(math.log10(pof[k]))
                    pof[k] = 1.0
            pof[k] *= div_mag
        lp = math.log10(pof[k]) if pof[k] > 0 else -math.inf
        if lp > 4 or overflow[k] != 0:
            extra[k] = int(math.ceil((lp + overflow[k] - 4.0) / 14.0))
            total_extra += extra[k]
            if b > 0:
                pof[k + 1] = -1
                total_extra += extra[k]
    re_o = np.zeros(n + total_extra); im_o = np.zeros(n + total_extra)
    re_o[0:n] = re; im_o[0:n] = im
    cnt = n
    for i in range(n):
        for _ in range(int(extra[i])):
            re_o[cnt] = re[i]; im_o[cnt] = im[i]
            if im[i] > 0:
                re_o[cnt + 1] = re[i]; im_o[cnt + 1] = -im[i]
            cnt += 1
            if im[i] > 0:
                cnt += 1
    return re_o, im_o


def roots_newton(matvec, n, poly_order, add_roots=True, seed=1):
    """src/Gmres_Poly_Newton.F90:379-712: harmonic Ritz values, clustered, extra roots,
    modified Leja order, zeros last.  Returns (re, im)."""
    r0 = box_muller(seed, n)
    beta, H, m, _, _ = arnoldi(matvec, r0, poly_order, TOL_LUCKY)
    p1 = poly_order + 1
    e_d = np.zeros(p1); e_d[poly_order] = 1.0
    HT = np.array(H.T[0:p1, 0:p1], order="F", copy=True)
    sol, _, _, _ = _gelsd(HT, e_d, TOL_RCOND)
    sol = sol[0:p1] * H[poly_order + 1, poly_order] ** 2
    Hs = np.array(H[0:p1, 0:p1], order="F", copy=True)
    Hs[:, poly_order] += sol
    wr, wi, _, _, info = lapack.dgeev(Hs, compute_vl=0, compute_vr=0)
    if info != 0:
        raise RuntimeError("geev failed")
    re = wr.copy(); im = wi.copy()
    H_norm = float(np.linalg.norm(Hs[0:m, 0:m]))  # note: H_n was overwritten by geev in the
    # reference (LAPACK destroys its input); norm2(H_n(1:m,1:m)) is therefore taken on the
    # Schur-form leftovers.  We cannot reproduce LAPACK's internal state bit-for-bit, so the
    # Frobenius norm of the modified Hessenberg is used; it only feeds clustering tolerances.
    rel_tol = math.sqrt(EPS)
    abs_tol = EPS * max(H_norm, beta)
    small = re ** 2 + im ** 2 < (abs_tol + rel_tol * H_norm) ** 2
    re[small] = 0.0; im[small] = 0.0
    re, im = cluster_eigenvalues_stable(re, im, rel_tol, abs_tol)
    nz = (re != 0.0) | (im != 0.0)
    numerical_order = int(nz.sum())
    if numerical_order == 0:
        return np.zeros(p1), np.zeros(p1)
    rt, it = re[nz].copy(), im[nz].copy()
    if add_roots:
        ra, ia = compute_extra_roots(rt, it)
        pr, pi_ = ra.copy(), ia.copy()
        counter = numerical_order
        for i in range(numerical_order):
            k = 0
            for j in range(counter, ra.size):
                if ra[j] == rt[i] and abs(ia[j]) == abs(it[i]):
                    k += 1
                    pr[j] = ra[j] + k * TOL_LEJA_PERTURB
        idx = modified_leja(pr, pi_)
        total = ra.size + (p1 - numerical_order)
        re_o = np.zeros(total); im_o = np.zeros(total)
        re_o[0:ra.size] = ra[idx]; im_o[0:ra.size] = ia[idx]
        return re_o, im_o
    idx = modified_leja(rt, it)
    re_o = np.zeros(p1); im_o = np.zeros(p1)
    re_o[0:numerical_order] = rt[idx]; im_o[0:numerical_order] = it[idx]
    return re_o, im_o


def mat_mult_powers_share_sparsity(A, coeffs, sparsity_order):
    """src/Gmres_Poly.F90:920-1337: sum_k c_k A^k with every power > sparsity_order
    truncated row-wise to the pattern of A^sparsity_order."""
    A = A.tocsr(); A.sort_indices()
    n = A.shape[0]
    if sparsity_order == 0:
        d = A.diagonal()
        inv = np.full(n, coeffs[0])
        pw = d.copy()
        order = coeffs.size - 1
        for k in range(1, order + 1):
            inv = inv + coeffs[k] * pw
            if k != order:
                pw = pw * d
        return InvOp("diag", d=inv)
    powers = [A]
    for _ in range(2, sparsity_order + 1):
        Pn = (A @ powers[-1]).tocsr(); Pn.sort_indices()
        # MatMatMult keeps entries that cancel to exactly zero; scipy prunes them
        powers.append(_keep_pattern(A, powers[-1], Pn))
    pat = powers[sparsity_order - 1]
    cmat = copy_plus_diag(pat)
    cmat.data *= coeffs[sparsity_order]
    for order in range(sparsity_order - 1, 0, -1):
        cmat.data += coeffs[order] * values_on_pattern(powers[order - 1], cmat)
    diag_pos = values_on_pattern(sp.identity(n, format="csr"), cmat)
    cmat.data += coeffs[0] * diag_pos
    prev = pat.copy()
    for term in range(sparsity_order + 1, coeffs.size):
        prod = (prev @ A).tocsr()
        newv = values_on_pattern(prod, pat)
        prev = sp.csr_matrix((newv, pat.indices.copy(), pat.indptr.copy()), shape=pat.shape)
        if coeffs[term] != 0.0:
            cmat.data += coeffs[term] * values_on_pattern(prev, cmat)
    return InvOp("assembled", mat=cmat)


def mat_newton_share_sparsity(A, re, im, sparsity_order):
    """Assembled Newton-basis polynomial with fixed sparsity
    (mat_mult_powers_share_sparsity_newton / build_gmres_polynomial_newton_inverse,
    src/Gmres_Poly_Newton.F90:1203-1990): the matrix form of `newton_apply` started from
    the identity, every product T*A truncated to pattern(A^sparsity_order) + I."""
    A = A.tocsr(); A.sort_indices()
    n = A.shape[0]
    pat = A
    for _ in range(2, sparsity_order + 1):
        pat = _pattern_product(A, pat)
    pat = copy_plus_diag(pat)
    pat.sort_indices()

    def on_pat(M):
        return sp.csr_matrix((values_on_pattern(M.tocsr(), pat), pat.indices, pat.indptr), shape=pat.shape)

    mv = lambda T: on_pat(T @ A)
    Y = newton_apply(mv, re, im, on_pat(sp.identity(n, format="csr")))
    return InvOp("assembled", mat=on_pat(Y))


def build_gmres_polynomial_inverse(A, poly_order, coeffs, sparsity_order, matrix_free,
                                   diag_scale):
    """src/Gmres_Poly.F90:1522-1813."""
    if matrix_free:
        nz = int(np.count_nonzero(coeffs[:-1])) if coeffs.size > 1 else 0
        if diag_scale:
            return InvOp("horner", A=A, coeffs=coeffs.copy(), diag=np.asarray(A.diagonal()).copy(),
                         spmvs=nz)
        return InvOp("horner", A=A, coeffs=coeffs.copy(), spmvs=nz)
    if isinstance(A, DiagMat):
        A = A.tocsr()
    if diag_scale:
        dinv = 1.0 / A.diagonal()
        As = sp.diags(dinv) @ A
        As = As.tocsr(); As.sort_indices()
    else:
        As = A
    if poly_order == 0:
        op = InvOp("diag", d=np.full(A.shape[0], coeffs[0]))
    elif poly_order == 1 and sparsity_order == 1:
        M = copy_plus_diag(As)
        M.data *= coeffs[1]
        M.data += coeffs[0] * values_on_pattern(sp.identity(A.shape[0], format="csr"), M)
        op = InvOp("assembled", mat=M)
    elif sparsity_order < poly_order:
        op = mat_mult_powers_share_sparsity(As, coeffs, sparsity_order)
    else:
        M = copy_plus_diag(As) * coeffs[1]
        pw = As
        for order in range(2, poly_order + 1):
            pw = (As @ pw).tocsr()
            if coeffs[order] == 0.0:
                continue
            M = M + coeffs[order] * pw
        M = (M + coeffs[0] * sp.identity(A.shape[0], format="csr")).tocsr()
        M.sort_indices()
        op = InvOp("assembled", mat=M)
    if diag_scale:
        if op.kind == "diag":
            op.d = op.d * dinv
        else:
            op.mat = (op.mat @ sp.diags(dinv)).tocsr(); op.mat.sort_indices()
    return op


def _pattern_product(A, B):
    """symbolic product: the pattern of A*B without numerical cancellation"""
    A1 = A.tocsr().copy(); A1.data = np.ones(A1.nnz)
    B1 = B.tocsr().copy(); B1.data = np.ones(B1.nnz)
    C = (A1 @ B1).tocsr(); C.sort_indices()
    return C


def _keep_pattern(A, B, prod):
    """the numeric product `prod` = +-A*B re-expressed on the symbolic pattern of A*B (scipy prunes entries
    that cancel to exactly zero, PETSc's MatMatMult keeps them)"""
    pat = _pattern_product(A, B)
    return sp.csr_matrix((values_on_pattern(prod, pat), pat.indices, pat.indptr), shape=pat.shape)


def sai_z(A_ff, A_cf, S, incomplete):
    """calculate_and_build_sai_z_cpu (src/SAI_Z.F90:24-645): for every row i with pattern J = S[i,:]
    incomplete (lAIR / ISAI): z A_ff[J,J] = -A_cf[i,J]  (square solve, :520-540);
    complete (SAI): min ||z A_ff[J,I] + A_cf[i,I]||, I = columns present in the rows J (:330-362, :575-600).
    A_cf None stands for -I (calculate_and_build_sai, :781-863).  Blocks are always solved directly
    (the reference switches to GMRES/LSQR with rtol 1e-3 above size 40, its device path never does)."""
    A_ff = A_ff.tocsr(); A_ff.sort_indices()
    S = S.tocsr(); S.sort_indices()
    if A_cf is not None:
        A_cf = A_cf.tocsr(); A_cf.sort_indices()
    out = np.zeros(S.nnz)
    for i in range(S.shape[0]):
        p0, p1 = S.indptr[i], S.indptr[i + 1]
        J = S.indices[p0:p1]
        if J.size == 0:
            continue
        if A_cf is not None:
            jv = np.asarray(A_cf[i, J].todense()).ravel()
        else:
            jv = np.where(J == i, -1.0, 0.0)
        rows = A_ff[J, :]
        if incomplete:
            I = J
        else:
            I = np.unique(rows.indices)
        if I.size < J.size:
            continue
        B = np.asarray(rows[:, I].todense()).T            # A_ff[J, I]^T
        e = np.zeros(I.size)
        pos = np.searchsorted(I, J)
        ok = (pos < I.size) & (I[np.minimum(pos, I.size - 1)] == J)
        e[pos[ok]] = -jv[ok]
        if incomplete:
            z = np.linalg.solve(B, e)
        else:
            z = np.linalg.lstsq(B, e, rcond=None)[0]
        out[p0:p1] = z
    return sp.csr_matrix((out, S.indices.copy(), S.indptr.copy()), shape=S.shape)


def sai_inverse(A, sparsity_order, incomplete):
    """calculate_and_build_sai (src/SAI_Z.F90:781-863)."""
    A = A.tocsr(); A.sort_indices()
    n = A.shape[0]
    if sparsity_order == 0:
        S = sp.identity(n, format="csr")
    else:
        S = A
        for _ in range(2, sparsity_order + 1):
            S = _pattern_product(A, S)          # mat_mat_symbolic: no numerical cancellation
    return sai_z(A, None, S, incomplete)


def clamp_poly(global_rows, poly_order, sparsity_order):
    """src/Gmres_Poly.F90:64-76."""
    if poly_order + 1 > global_rows:
        poly_order = int(global_rows - 1)
    if sparsity_order > poly_order:
        sparsity_order = poly_order
    return poly_order, sparsity_order


def build_inverse(A, inverse_type, poly_order, sparsity_order, matrix_free, diag_scale,
                  seed=1, need_assembled=False):
    """start_/finish_approximate_inverse (src/Approx_Inverse_Setup.F90:207-500).
    Returns (op, assembled_op_or_None): the second is the assembled version the grid
    transfers need when `op` is matrix-free (src/AIR_Operators_Setup.F90:446-466)."""
    n = A.shape[0]
    poly_order, sparsity_order = clamp_poly(n, poly_order, sparsity_order)
    Acsr = A if isinstance(A, DiagMat) else A.tocsr()
    d = np.asarray(A.diagonal(), dtype=np.float64)

    def mv_plain(x):
        return Acsr @ x

    def mv_scaled(x):
        return (Acsr @ x) / d

    is_gmres = inverse_type in (INV_POWER, INV_ARNOLDI, INV_NEWTON, INV_NEWTON_NO_EXTRA)
    mv = mv_scaled if (is_gmres and diag_scale) else mv_plain

    if inverse_type in (INV_POWER, INV_ARNOLDI):
        c = coeffs_power(mv, n, poly_order, seed) if inverse_type == INV_POWER \
            else coeffs_arnoldi(mv, n, poly_order, seed)
        op = build_gmres_polynomial_inverse(A, poly_order, c, sparsity_order, matrix_free, diag_scale)
        asm = None
        if matrix_free and need_assembled:
            asm = build_gmres_polynomial_inverse(A, poly_order, c, sparsity_order, False, diag_scale)
        return op, asm
    if inverse_type in (INV_NEWTON, INV_NEWTON_NO_EXTRA):
        re, im = roots_newton(mv, n, poly_order, inverse_type == INV_NEWTON, seed)

        def assembled():
            # src/Gmres_Poly_Newton.F90:2010-2098: polynomial of D^-1 A, then scaled by D^-1 on the right
            dinv = 1.0 / d if diag_scale else None
            if isinstance(A, DiagMat) or sparsity_order == 0:
                dd = np.ones(n) if diag_scale else d
                a = InvOp("diag", d=newton_apply(lambda v: dd * v, re, im, np.ones(n)))
                if diag_scale:
                    a.d = a.d * dinv
                return a
            if not diag_scale:
                return mat_newton_share_sparsity(Acsr, re, im, sparsity_order)
            As = (sp.diags(dinv) @ Acsr).tocsr(); As.sort_indices()
            a = mat_newton_share_sparsity(As, re, im, sparsity_order)
            a.mat = (a.mat @ sp.diags(dinv)).tocsr(); a.mat.sort_indices()
            return a

        if not matrix_free:
            return assembled(), None
        op = InvOp("newton", A=A, re=re, im=im, diag=d.copy() if diag_scale else None)
        op.spmvs = newton_spmv_count(re, im)
        return op, (assembled() if need_assembled else None)
    if inverse_type == INV_NEUMANN:
        c = np.ones(poly_order + 1)
        if matrix_free:
            op = InvOp("horner", A=A, coeffs=c, diag=d.copy(), neumann=True, spmvs=poly_order)
            asm = None
            if need_assembled:
                asm, _ = build_inverse(A, INV_NEUMANN, poly_order, sparsity_order, False, False, seed)
            return op, asm
        Ac = A.tocsr()
        dinv = 1.0 / d
        T = (sp.identity(n, format="csr") - sp.diags(dinv) @ Ac).tocsr()
        # keep A's pattern (MatDuplicate + scale + shift keep explicit entries)
        T = sp.csr_matrix((values_on_pattern(T, copy_plus_diag(Ac)), copy_plus_diag(Ac).indices,
                           copy_plus_diag(Ac).indptr), shape=Ac.shape)
        op = build_gmres_polynomial_inverse(T, poly_order, c, sparsity_order, False, False)
        if op.kind == "diag":
            op.d = op.d * dinv
        else:
            op.mat = (op.mat @ sp.diags(dinv)).tocsr(); op.mat.sort_indices()
        return op, None
    if inverse_type in (INV_SAI, INV_ISAI):
        Ac = A.tocsr()
        return InvOp("assembled", mat=sai_inverse(Ac, sparsity_order, inverse_type == INV_ISAI)), None
    if inverse_type in (INV_JACOBI, INV_WJACOBI):
        weight = 1.0
        if inverse_type == INV_WJACOBI:
            s = 1.0 / np.sqrt(np.abs(d))
            T = sp.diags(s) @ A.tocsr() @ sp.diags(s)
            norm_inf = float(np.max(np.abs(T).sum(axis=1)))
            weight = 3.0 / (4.0 * norm_inf)
        return InvOp("diag", d=weight / d), None
    raise NotImplementedError(f"inverse type {inverse_type}")


def newton_spmv_count(re, im):
    """SpMVs one petsc_newton application executes (src/Gmres_Poly_Newton.F90:763-875)."""
    n = re.size
    i = 0
    cnt = 0
    while i <= n - 2:
        if im[i] == 0.0:
            if abs(re[i]) < TOL_ZERO:
                i += 1
                continue
            cnt += 1
            i += 1
        else:
            if re[i] ** 2 + im[i] ** 2 < TOL_ZERO:
                i += 2
                continue
            cnt += 1
            if i <= n - 3:
                cnt += 1
            i += 2
    return cnt


# ----------------------------------------------------------------------------------------
# Matrix-free applies (solve phase)
# ----------------------------------------------------------------------------------------
def horner(mv, coeffs, x):
    """src/Gmres_Poly.F90:1418-1484."""
    y = coeffs[-1] * x
    for k in range(coeffs.size - 2, -1, -1):
        if coeffs[k] == 0.0:
            continue
        t = y.copy()
        y = mv(t)
        y = coeffs[k] * x + 1.0 * y
    return y


def newton_apply(mv, re, im, x):
    """src/Gmres_Poly_Newton.F90:763-875 (Loe et al. Alg. 3)."""
    t = x.copy()
    y = np.zeros_like(x) if isinstance(x, np.ndarray) else 0.0 * x
    n = re.size
    i = 0
    while i <= n - 2:
        if im[i] == 0.0:
            if abs(re[i]) < TOL_ZERO:
                i += 1
                continue
            y = y + (1.0 / re[i]) * t
            t2 = mv(t)
            t = t + (-1.0 / re[i]) * t2
            i += 1
        else:
            sq = re[i] ** 2 + im[i] ** 2
            if sq < TOL_ZERO:
                i += 2
                continue
            t2 = mv(t)
            t2 = (2 * re[i]) * t + (-1.0) * t2
            y = y + (1.0 / sq) * t2
            if i <= n - 3:
                t3 = mv(t2)
                t = t + (-1.0 / sq) * t3
            i += 2
    if im[n - 1] == 0.0:
        if abs(re[n - 1]) > TOL_ZERO:
            y = (1.0 / re[n - 1]) * t + 1.0 * y
    return y


def apply_inv(op: InvOp, x):
    if op.kind == "assembled":
        return op.mat @ x
    if op.kind == "diag":
        return op.d * x
    A = op.A
    if op.kind == "horner":
        if op.diag is None:
            return horner(lambda v: A @ v, op.coeffs, x)
        rhs = x / op.diag
        if op.neumann:
            mv = lambda v: 1.0 * v + (-1.0) * ((A @ v) / op.diag)
        else:
            mv = lambda v: (A @ v) / op.diag
        return horner(mv, op.coeffs, rhs)
    if op.kind == "newton":
        if op.diag is None:
            return newton_apply(lambda v: A @ v, op.re, op.im, x)
        rhs = x / op.diag
        return newton_apply(lambda v: (A @ v) / op.diag, op.re, op.im, rhs)
    raise ValueError(op.kind)


# ----------------------------------------------------------------------------------------
# Hierarchy
# ----------------------------------------------------------------------------------------
@dataclass
class AirOptions:
    """-pc_air_* subset (src/AIR_Data_Type.F90:34-264; defaults keep the reference's
    single-precision literals widened to double)."""
    max_levels: int = 300
    coarse_eq_limit: int = 6
    auto_truncate_start_level: int = -1
    auto_truncate_tol: float = TOL_AUTO_TRUNCATE
    strong_threshold: float = 0.5
    ddc_its: int = 1
    ddc_fraction: float = _f32(0.1)
    max_luby_steps: int = -1
    smooth_type: str = "ff"
    diag_scale_polys: bool = False
    matrix_free_polys: bool = False
    one_point_classical_prolong: bool = True
    symmetric: bool = False
    inverse_type: int = INV_ARNOLDI
    poly_order: int = 6
    inverse_sparsity_order: int = 1
    c_inverse_type: int = INV_ARNOLDI
    c_poly_order: int = 6
    c_inverse_sparsity_order: int = 1
    coarsest_inverse_type: int = INV_ARNOLDI
    coarsest_poly_order: int = 6
    coarsest_inverse_sparsity_order: int = 1
    coarsest_matrix_free_polys: bool = False
    coarsest_diag_scale_polys: bool = False
    r_drop: float = _f32(0.01)
    a_drop: float = _f32(1e-4)
    a_lump: bool = False
    strong_r_threshold: float = 0.0       # -pc_air_strong_r_threshold
    z_type: str = "product"               # -pc_air_z_type product | lair | lair_sai
    lair_distance: int = 2                # -pc_air_lair_distance
    improve_z_its: int = 0                # -pc_air_improve_z_its
    improve_w_its: int = 0                # -pc_air_improve_w_its
    full_smoothing_up_and_down: bool = False   # -pc_air_full_smoothing_up_and_down
    cf_splitting_type: str = "pmisr_ddc"       # -pc_air_cf_splitting_type pmisr_ddc | diag_dom | cr
    # -mg_coarse_ksp_type richardson -mg_coarse_ksp_max_it N (0 = preonly, the default)
    coarse_richardson_its: int = 0
    coarse_lu: bool = False               # -mg_coarse_pc_type lu
    seed: int = 1

    def smooth_order(self):
        """src/PCAIR_Interfaces.F90:1821-1882: 'ffc' -> [2,-1]."""
        out = []
        for ch in self.smooth_type.lower():
            val = 1 if ch == "f" else -1
            if out and (out[-1] > 0) == (val > 0):
                out[-1] += val
            else:
                out.append(val)
        return out


@dataclass
class Level:
    n: int
    F: np.ndarray
    C: np.ndarray
    A_ff: object
    A_fc: sp.csr_matrix
    A_cf: sp.csr_matrix | None
    A_cc: sp.csr_matrix | None
    inv_A_ff: InvOp
    inv_A_cc: InvOp | None
    Z: sp.csr_matrix
    W: sp.csr_matrix
    R: sp.csr_matrix
    P: sp.csr_matrix
    smooth_order: list
    aff_diag: bool
    # -pc_air_full_smoothing_up_and_down: the level operator; its approximate inverse is then kept in inv_A_ff
    # as the reference does (src/AIR_Operators_Setup.F90:113-119,376-395)
    A_full: object = None


@dataclass
class Hierarchy:
    levels: list = field(default_factory=list)
    coarse_A: sp.csr_matrix | None = None
    coarse_inv: InvOp | None = None
    coarse_richardson_its: int = 0
    A: sp.csr_matrix | None = None
    nnzs: list = field(default_factory=list)
    full_smoothing: bool = False

    @property
    def no_levels(self):
        return len(self.levels) + 1


def _inv_diag(op):
    """diagonal of an assembled approximate inverse (MatGetDiagonal, src/Grid_Transfer_Improve.F90:235-241)"""
    if op.kind == "assembled":
        return np.asarray(op.mat.diagonal())
    if op.kind == "diag":
        return op.d
    raise NotImplementedError("improve_z / improve_w need an assembled A_ff inverse")


def improve_z(Z, A_ff, A_cf, dm, its):
    """src/Grid_Transfer_Improve.F90:200-371: Z <- Z - ((Z A_ff + A_cf) diag(M)) on the pattern of Z."""
    Aff = A_ff.tocsr() if not isinstance(A_ff, DiagMat) else sp.diags(A_ff.diagonal()).tocsr()
    Z = Z.tocsr().copy(); Z.sort_indices()
    for _ in range(its):
        T = ((Z @ Aff + A_cf) @ sp.diags(dm)).tocsr()
        Z = sp.csr_matrix((Z.data - values_on_pattern(T, Z), Z.indices, Z.indptr), shape=Z.shape)
    return Z


def improve_w(W, A_ff, A_fc, dm, its):
    """src/Grid_Transfer_Improve.F90:21-196: W <- W - (diag(M) (A_ff W + A_fc)) on the pattern of W."""
    Aff = A_ff.tocsr() if not isinstance(A_ff, DiagMat) else sp.diags(A_ff.diagonal()).tocsr()
    W = W.tocsr().copy(); W.sort_indices()
    for _ in range(its):
        T = (sp.diags(dm) @ (Aff @ W + A_fc)).tocsr()
        W = sp.csr_matrix((W.data - values_on_pattern(T, W), W.indices, W.indptr), shape=W.shape)
    return W


def one_point_W(A_fc):
    """src/Grid_Transfer.F90:94-220: a single 1.0 at the first max-|.| column of each row."""
    n_f, n_c = A_fc.shape
    indptr, indices, data = A_fc.indptr, A_fc.indices, np.abs(A_fc.data)
    rows, cols = [], []
    lens = np.diff(indptr)
    nz = np.flatnonzero(lens > 0)
    if nz.size:
        rmax = np.maximum.reduceat(data, indptr[:-1][nz])
        rr = _rows_of(A_fc)
        rowmax = np.zeros(n_f); rowmax[nz] = rmax
        ismax = data == rowmax[rr]
        pos = np.flatnonzero(ismax)
        first = np.full(n_f, -1, dtype=np.int64)
        # first occurrence per row
        rpos = rr[pos]
        uniq, fi = np.unique(rpos, return_index=True)
        first[uniq] = pos[fi]
        rows = uniq
        cols = indices[first[uniq]]
    W = sp.coo_matrix((np.ones(len(rows)), (rows, cols)), shape=(n_f, n_c)).tocsr()
    W.sort_indices()
    return W


def build_R(Z, F, C, n):
    """R = [Z I] in level ordering (src/Grid_Transfer.F90:588-815)."""
    n_c = C.size
    Zc = Z.tocoo()
    r = np.concatenate([Zc.row, np.arange(n_c)])
    c = np.concatenate([F[Zc.col], C])
    v = np.concatenate([Zc.data, np.ones(n_c)])
    R = sp.coo_matrix((v, (r, c)), shape=(n_c, n)).tocsr()
    R.sort_indices()
    return R


def build_P(W, F, C, n):
    """P = [W; I] in level ordering (src/Grid_Transfer.F90:329-461)."""
    n_c = C.size
    Wc = W.tocoo()
    r = np.concatenate([F[Wc.row], C])
    c = np.concatenate([Wc.col, np.arange(n_c)])
    v = np.concatenate([Wc.data, np.ones(n_c)])
    P = sp.coo_matrix((v, (r, c)), shape=(n, n_c)).tocsr()
    P.sort_indices()
    return P


def _is_diag_only(A):
    """src/C_PETSc_Routines.c:324-403."""
    n = A.shape[0]
    if A.nnz != n:
        return False
    return bool(np.all(A.indices == np.arange(n)))


def _coarse_solver(A, o: AirOptions, final=True):
    if o.coarse_lu and final:
        # -mg_coarse_pc_type lu (tests/Makefile:144-147): exact solve of the coarsest operator
        return InvOp("assembled", mat=sp.csr_matrix(np.linalg.inv(A.toarray())))
    op, _ = build_inverse(A, o.coarsest_inverse_type, o.coarsest_poly_order,
                          o.coarsest_inverse_sparsity_order, o.coarsest_matrix_free_polys,
                          o.coarsest_diag_scale_polys, o.seed)
    return op


def setup_air(A, o: AirOptions) -> Hierarchy:
    """src/AIR_MG_Setup.F90:44-1231 (default AIRG path: PMISR+DDC, z_type product,
    no reuse, no processor agglomeration, no constraints)."""
    H = Hierarchy(A=A.tocsr(), coarse_richardson_its=o.coarse_richardson_its,
                  full_smoothing=o.full_smoothing_up_and_down)
    cur = A.tocsr(); cur.sort_indices()
    so = o.smooth_order()
    any_c = any(s < 0 for s in so)
    auto_truncated = False
    no_levels = o.max_levels
    for our_level in range(1, o.max_levels):
        n = cur.shape[0]
        if o.auto_truncate_start_level != -1 and our_level >= o.auto_truncate_start_level:
            cop = _coarse_solver(cur, o, final=False)
            rand = uniform(o.seed + 7919, 0, n)  # PetscRandom stream differs from the poly one
            norm_b = float(np.linalg.norm(rand))
            res = cur @ apply_inv(cop, rand) - rand
            if float(np.linalg.norm(res)) / norm_b < o.auto_truncate_tol:
                auto_truncated = True
                H.coarse_inv = cop
        if not auto_truncated:
            F, C = compute_cf_splitting(cur, o.strong_threshold, o.ddc_its, o.ddc_fraction,
                                        o.symmetric, o.seed, o.max_luby_steps, o.cf_splitting_type,
                                        cr=(o.inverse_type, o.poly_order, o.inverse_sparsity_order, o.diag_scale_polys))
            n_f, n_c = F.size, C.size
        else:
            n_f = n_c = 0
        cont = (not auto_truncated) and (n_c > o.coarse_eq_limit and n_f != 0)
        if not cont:
            no_levels = our_level
            break

        A_ff = submatrix(cur, F, F)
        smooth = list(so)
        inv_type_aff = o.inverse_type
        sparsity_aff = o.inverse_sparsity_order
        aff_diag = _is_diag_only(A_ff) if o.strong_threshold != 0.0 else True
        if aff_diag:
            A_ff_op = DiagMat(A_ff.diagonal())
            sparsity_aff = 0
            if inv_type_aff != INV_WJACOBI and o.poly_order > 2:
                smooth = [1 if s > 0 else s for s in smooth]
        else:
            A_ff_op = A_ff
        A_fc = submatrix(cur, F, C)
        A_cf = submatrix(cur, C, F)

        # strong R threshold (src/AIR_Operators_Setup.F90:46-100, 219-276): Z comes from a copy of A with
        # the small entries of each row removed (diagonal kept)
        product_z = o.z_type == "product"
        r_dropped = o.strong_r_threshold != 0.0
        if r_dropped:
            A_drop = remove_small(cur, o.strong_r_threshold, rel=1, drop_diag=0)
            A_ff_drop = submatrix(A_drop, F, F)
            A_cf_z = submatrix(A_drop, C, F)
            A_fc_w = submatrix(A_drop, F, C)
        else:
            A_ff_drop, A_cf_z, A_fc_w = A_ff, A_cf, A_fc

        # the smoother: A_ff^-1, or with full smoothing an approximate inverse of the whole level operator
        full = o.full_smoothing_up_and_down
        inv_ff, inv_ff_asm = build_inverse(cur if full else A_ff_op, inv_type_aff, o.poly_order, sparsity_aff,
                                           o.matrix_free_polys, o.diag_scale_polys, o.seed,
                                           need_assembled=product_z and not r_dropped and not full)
        if product_z and (r_dropped or full):
            # second, always assembled, inverse for the grid transfers (src/AIR_Operators_Setup.F90:413-438)
            inv_ff_asm, _ = build_inverse(A_ff_drop if r_dropped else A_ff_op, inv_type_aff, o.poly_order,
                                          sparsity_aff, False, o.diag_scale_polys, o.seed)
        inv_for_Z = inv_ff_asm if inv_ff_asm is not None else inv_ff
        if product_z and inv_for_Z.kind not in ("assembled", "diag"):
            raise NotImplementedError("grid transfers need an assembled A_ff inverse")

        A_cc = inv_cc = None
        if any_c and not full:
            A_cc = submatrix(cur, C, C)
            inv_cc, _ = build_inverse(A_cc, o.c_inverse_type, o.c_poly_order,
                                      o.c_inverse_sparsity_order, o.matrix_free_polys,
                                      o.diag_scale_polys, o.seed)

        if product_z:
            Mz = inv_for_Z.mat if inv_for_Z.kind == "assembled" else sp.diags(inv_for_Z.d).tocsr()
        if o.symmetric:
            W = one_point_W(A_fc)      # placeholder: replaced by Z^T once Z exists
        elif o.one_point_classical_prolong:
            W = one_point_W(A_fc)
        else:
            if not product_z:
                raise NotImplementedError("lAIR needs the one-point prolongator (AIR_Operators_Setup.F90:478-482)")
            W = (-(Mz @ A_fc_w)).tocsr(); W.sort_indices()
            if o.improve_w_its:
                W = improve_w(_keep_pattern(Mz, A_fc_w, W), A_ff_op, A_fc, _inv_diag(inv_ff), o.improve_w_its)
            W = remove_small(W, o.r_drop, rel=1)
        P = build_P(W, F, C, n)

        if product_z:
            Z = (A_cf_z @ Mz).tocsr()
            Z.data *= -1.0
            Z.sort_indices()
        else:
            # lAIR / SAI (src/AIR_Operators_Setup.F90:698-778)
            if o.lair_distance == 1:
                S = A_cf_z
            else:
                pw = A_ff_drop
                for _ in range(3, o.lair_distance + 1):
                    pw = _pattern_product(A_ff_drop, pw)
                S = _pattern_product(A_cf_z, pw)
            Z = sai_z(A_ff, A_cf, S, o.z_type == "lair")
        if o.improve_z_its:
            if product_z:
                Z = _keep_pattern(A_cf_z, Mz, Z)
            Z = improve_z(Z, A_ff_op, A_cf, _inv_diag(inv_ff), o.improve_z_its)
        Z = remove_small(Z, o.r_drop, rel=1)
        R = build_R(Z, F, C, n)
        if o.symmetric:
            # -pc_air_symmetric (src/AIR_Operators_Setup.F90:898-902): P = R^T, i.e. W = Z^T
            P = R.T.tocsr(); P.sort_indices()
            W = Z.T.tocsr(); W.sort_indices()

        AP = (cur @ P).tocsr()
        RAP = (R @ AP).tocsr(); RAP.sort_indices()
        coarse = remove_small(RAP, o.a_drop, rel=1, lump=o.a_lump)

        H.levels.append(Level(n=n, F=F, C=C, A_ff=A_ff_op, A_fc=A_fc, A_cf=A_cf, A_cc=A_cc,
                              inv_A_ff=inv_ff, inv_A_cc=inv_cc, Z=Z, W=W, R=R, P=P,
                              smooth_order=smooth, aff_diag=aff_diag, A_full=cur if full else None))
        H.nnzs.append(cur.nnz)
        cur = coarse
    H.coarse_A = cur
    H.nnzs.append(cur.nnz)
    if not auto_truncated or o.coarse_lu:
        H.coarse_inv = _coarse_solver(cur, o)
    if no_levels == 1 and not auto_truncated and not o.coarse_lu:
        # "Only a single level, defaulting to Jacobi PC" (src/AIR_MG_Setup.F90:1167-1174)
        H.coarse_inv = InvOp("diag", d=1.0 / cur.diagonal())
    return H


# ----------------------------------------------------------------------------------------
# Solve phase
# ----------------------------------------------------------------------------------------
def f_smooths(L: Level, b, x, first, its, state):
    """src/FC_Smooth.F90:499-568."""
    b_f = b[L.F]
    if first:
        state["x_f"] = x[L.F]
        state["x_c"] = x[L.C]
    t2 = L.A_fc @ state["x_c"]
    t4 = b_f + (-1.0) * t2
    for _ in range(its):
        t3 = L.A_ff @ state["x_f"]
        t3 = t4 + (-1.0) * t3
        t2 = apply_inv(L.inv_A_ff, t3)
        state["x_f"] = state["x_f"] + 1.0 * t2
    x[L.F] = state["x_f"]


def c_smooths(L: Level, b, x, first, its, state):
    """src/FC_Smooth.F90:572-640."""
    b_c = b[L.C]
    if first:
        state["x_f"] = x[L.F]
        state["x_c"] = x[L.C]
    t2 = L.A_cf @ state["x_f"]
    t4 = b_c + (-1.0) * t2
    for _ in range(its):
        t3 = L.A_cc @ state["x_c"]
        t3 = t4 + (-1.0) * t3
        t2 = apply_inv(L.inv_A_cc, t3)
        state["x_c"] = state["x_c"] + 1.0 * t2
    x[L.C] = state["x_c"]


def fc_point_richardson(L: Level, b, x):
    """src/FC_Smooth.F90:421-495."""
    state = {}
    first = True
    for s in L.smooth_order:
        if s == 0:
            break
        if s > 0:
            f_smooths(L, b, x, first, s, state)
        else:
            c_smooths(L, b, x, first, -s, state)
        first = False


def coarse_solve(H: Hierarchy, b):
    """KSPPREONLY + mg_coarse_shell_apply (src/FC_Smooth.F90:29-49), or N Richardson steps."""
    if H.coarse_richardson_its <= 0:
        return apply_inv(H.coarse_inv, b)
    x = np.zeros_like(b)
    for i in range(H.coarse_richardson_its):
        r = b if i == 0 else b - H.coarse_A @ x
        x = x + apply_inv(H.coarse_inv, r)
    return x


def vcycle_full(H: Hierarchy, b):
    """-pc_air_full_smoothing_up_and_down (src/AIR_MG_Setup.F90:1014-1079): PCMG multiplicative V-cycle, one
    Richardson(1) + PCMAT(inv_A) sweep on all points down (zero initial guess) and one up."""
    bs = [np.asarray(b, dtype=np.float64)]
    xs = []
    for L in H.levels:
        x = apply_inv(L.inv_A_ff, bs[-1])
        xs.append(x)
        bs.append(L.R @ (bs[-1] - L.A_full @ x))
    x = coarse_solve(H, bs[-1])
    for li in range(len(H.levels) - 1, -1, -1):
        L = H.levels[li]
        x = xs[li] + L.P @ x
        x = x + apply_inv(L.inv_A_ff, bs[li] - L.A_full @ x)
    return x


def vcycle(H: Hierarchy, b):
    """PC_MG_KASKADE cycle: restrict b down, coarse solve, interpolate +
    FC relax on the way up."""
    if H.full_smoothing:
        return vcycle_full(H, b)
    bs = [np.asarray(b, dtype=np.float64)]
    for L in H.levels:
        bs.append(L.R @ bs[-1])
    x = coarse_solve(H, bs[-1])
    for li in range(len(H.levels) - 1, -1, -1):
        L = H.levels[li]
        x = L.P @ x
        fc_point_richardson(L, bs[li], x)
    return x


def pcpflareinv_setup(A, inverse_type=INV_ARNOLDI, poly_order=6, sparsity_order=1,
                      matrix_free=False, seed=1):
    """src/PCPFLAREINV.c:689-783 -> calculate_and_build_approximate_inverse."""
    op, _ = build_inverse(A, inverse_type, poly_order, sparsity_order, matrix_free, False, seed)
    return op


# ----------------------------------------------------------------------------------------
# Outer Krylov (PETSc semantics)
# ----------------------------------------------------------------------------------------
def _converged_default(state, it, rnorm, rtol, atol, dtol):
    """KSPConvergedDefault incl. the zero-RHS / nonzero-guess special case."""
    if it == 0:
        snorm = state["bnorm"]
        if snorm == 0.0:
            snorm = rnorm
        state["rnorm0"] = snorm
        state["ttol"] = max(rtol * snorm, atol)
    if math.isnan(rnorm) or math.isinf(rnorm):
        return -9  # KSP_DIVERGED_NANORINF
    if rnorm <= state["ttol"]:
        return 3 if rnorm < atol else 2
    if rnorm >= dtol * state["rnorm0"]:
        return -4
    return 0


def ksp_richardson(A, pc, b, x0, rtol=1e-5, atol=1e-50, dtol=1e4, max_it=10000,
                   norm_type="preconditioned", history=None):
    """KSPSolve_Richardson (left PC, scale 1, nonzero initial guess)."""
    x = x0.copy()
    prec = norm_type == "preconditioned"
    if prec:
        bnorm = float(np.linalg.norm(pc(b))) if np.any(b) else 0.0
    else:
        bnorm = float(np.linalg.norm(b))
    st = {"bnorm": bnorm}
    its = 0
    reason = 0
    r = b - A @ x
    z = None
    for i in range(max_it):
        if prec:
            z = pc(r); rnorm = float(np.linalg.norm(z))
        else:
            rnorm = float(np.linalg.norm(r))
        if history is not None:
            history.append(rnorm)
        reason = _converged_default(st, i, rnorm, rtol, atol, dtol)
        if reason:
            break
        if not prec:
            z = pc(r)
        x = x + z
        its += 1
        r = b - A @ x
    if not reason:
        if prec:
            z = pc(r); rnorm = float(np.linalg.norm(z))
        else:
            rnorm = float(np.linalg.norm(r))
        if history is not None:
            history.append(rnorm)
        if its >= max_it:
            reason = _converged_default(st, its, rnorm, rtol, atol, dtol)
            if not reason:
                reason = -3
    return x, its, reason, rnorm


def ksp_gmres(A, pc, b, x0, rtol=1e-5, atol=1e-50, dtol=1e4, max_it=10000, restart=30,
              pc_side="left", history=None):
    """KSPSolve_GMRES: classical Gram-Schmidt without refinement, Givens-rotated
    Hessenberg residual estimate, restart 30, happy-breakdown tolerance 1e-30."""
    n = b.size
    x = x0.copy()
    right = pc_side == "right"
    if right:
        bnorm = float(np.linalg.norm(b))
    else:
        bnorm = float(np.linalg.norm(pc(b))) if np.any(b) else 0.0
    st = {"bnorm": bnorm}
    its = 0
    reason = 0
    haptol = 1e-30
    rnorm = 0.0
    first = True
    while not reason:
        r = b - A @ x
        if not right:
            r = pc(r)
        V = np.zeros((restart + 1, n))
        res = float(np.linalg.norm(r))
        rnorm = res
        if first:
            if history is not None:
                history.append(res)
            if res == 0.0:
                reason = 3
                break
            reason = _converged_default(st, its, res, rtol, atol, dtol)
            first = False
            if reason:
                break
        V[0] = r / res if res != 0 else r
        hh = np.zeros((restart + 2, restart + 1))
        cs = np.zeros(restart + 1); sn = np.zeros(restart + 1)
        grs = np.zeros(restart + 2)
        grs[0] = res
        it = 0
        hapend = False
        while not reason and it < restart and its < max_it:
            w = (A @ pc(V[it])) if right else pc(A @ V[it])
            lhh = V[0:it + 1] @ w
            w = w - lhh @ V[0:it + 1]
            hh[0:it + 1, it] = lhh
            tt = float(np.linalg.norm(w))
            if tt != 0.0:
                V[it + 1] = w / tt
            hh[it + 1, it] = tt
            hapbnd = abs(tt / grs[it]) if grs[it] != 0 else haptol
            if hapbnd > haptol:
                hapbnd = haptol
            if tt < hapbnd:
                hapend = True
            # KSPGMRESUpdateHessenberg
            for j in range(it):
                t1 = hh[j, it]
                hh[j, it] = cs[j] * t1 + sn[j] * hh[j + 1, it]
                hh[j + 1, it] = -sn[j] * t1 + cs[j] * hh[j + 1, it]
            if not hapend:
                t = math.sqrt(hh[it, it] ** 2 + hh[it + 1, it] ** 2)
                if t == 0.0:
                    reason = -5
                    break
                cs[it] = hh[it, it] / t
                sn[it] = hh[it + 1, it] / t
                grs[it + 1] = -sn[it] * grs[it]
                grs[it] = cs[it] * grs[it]
                hh[it, it] = cs[it] * hh[it, it] + sn[it] * hh[it + 1, it]
                res = abs(grs[it + 1])
            else:
                res = 0.0
            it += 1
            its += 1
            rnorm = res
            if history is not None:
                history.append(res)
            reason = _converged_default(st, its, res, rtol, atol, dtol)
            if hapend:
                if not reason:
                    reason = -5
                break
        # KSPGMRESBuildSoln
        if it > 0:
            k = it - 1
            nrs = np.zeros(it)
            if hh[k, k] != 0.0:
                nrs[k] = grs[k] / hh[k, k]
                for ii in range(1, k + 1):
                    kk = k - ii
                    tsum = grs[kk] - np.dot(hh[kk, kk + 1:k + 1], nrs[kk + 1:k + 1])
                    nrs[kk] = tsum / hh[kk, kk]
            upd = nrs @ V[0:it]
            x = x + (pc(upd) if right else upd)
        if not reason and its >= max_it:
            reason = -3
    return x, its, reason, rnorm


# ----------------------------------------------------------------------------------------
# Algorithmic work
# ----------------------------------------------------------------------------------------
def S_bytes(nnz, m, n):
    return 12 * nnz + 4 * (m + 1) + 8 * m + 8 * n


def _inv_bytes(op: InvOp):
    if op.kind == "assembled":
        return S_bytes(op.mat.nnz, *op.mat.shape)
    if op.kind == "diag":
        n = op.d.size
        return S_bytes(n, n, n)
    A = op.A
    return op.spmvs * S_bytes(A.nnz, A.shape[0], A.shape[1])


def cycle_bytes(H: Hierarchy):
    tot = 0
    for L in H.levels:
        tot += S_bytes(L.R.nnz, *L.R.shape) + S_bytes(L.P.nnz, *L.P.shape)
        g = sum(1 for s in L.smooth_order if s > 0)
        s_f = sum(s for s in L.smooth_order if s > 0)
        tot += g * S_bytes(L.A_fc.nnz, *L.A_fc.shape)
        tot += s_f * (S_bytes(L.A_ff.nnz, *L.A_ff.shape) + _inv_bytes(L.inv_A_ff))
        if L.A_cc is not None:
            gc = sum(1 for s in L.smooth_order if s < 0)
            s_c = sum(-s for s in L.smooth_order if s < 0)
            tot += gc * S_bytes(L.A_cf.nnz, *L.A_cf.shape)
            tot += s_c * (S_bytes(L.A_cc.nnz, *L.A_cc.shape) + _inv_bytes(L.inv_A_cc))
    tot += _inv_bytes(H.coarse_inv)
    return tot
