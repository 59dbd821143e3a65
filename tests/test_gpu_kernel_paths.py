"""Every path of the row-block SpMV kernel (csrc/device/spmv.hip), hit on purpose and checked against scipy
(airhip_mat_mult through the C ABI, <= 1e-13 relative: PFLARE_TOL_MATFREE_13, src/Pflare_Parameters.F90:209):

* packed blocks (column-sorted, block-relative 19-bit columns: rb.hpp COL_BITS), raw blocks (row-major, absolute columns)
  and multi-base blocks (row-major, 16-bit window | offset words, dev.hpp BlkDesc) - every test of this file runs under
  three conversion settings: the default (multi-base wherever it fits), multi-base off (the round-3 format) and multi-base
  for blocks of <= 4 rows and blocks that cannot be packed only; the three must agree BIT FOR BIT (same products, same order
  of additions) - the round-1 kernel switched formats on the operator width (2^22 columns) and the wide mode, the one
  the 464^3 benchmark runs, had no test;
* every straight-line variant of the block code (0 ... 8 full rounds of 64 entries, with and without a partial round; values
  of paired full rounds interleaved for 16-byte loads), on packed and on raw blocks;
* a row longer than a block (wave-per-row reduction), rows of exactly 512 / 513 entries;
* blocks of <= 32 rows (T lanes share a row) for every T, blocks of 128 rows (two rows per lane), empty rows;
* matrices of 1, 2, 3 ... blocks (waves without a block, partly filled workgroups) and of many thousands;
* the fused epilogues through a polynomial apply (Horner / Newton / Neumann use every epilogue field).
"""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

TOL = 1e-13


def lcg_vector(n):
    i = np.arange(n, dtype=np.uint64)
    return (((np.uint64(1103515245) * (i + np.uint64(12345))) % np.uint64(2 ** 31)).astype(np.float64)
            / float(2 ** 30) - 1.0)


def _check(A, torch, P, x=None):
    A = sp.csr_matrix(A)
    A.sum_duplicates(); A.sort_indices()
    m, n = A.shape
    x = lcg_vector(n) if x is None else x
    dm = P.DeviceMat(A)
    xd = torch.from_numpy(x).cuda()
    y = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
    dm.mult(xd, y)
    ref = A @ x
    scale = np.abs(A) @ np.abs(x)                    # row-wise bound on the rounding error
    err = np.abs(y.cpu().numpy() - ref)
    assert np.all(np.isfinite(y.cpu().numpy()))
    bad = err > TOL * np.maximum(scale, 1e-300)
    assert not bad.any(), (int(bad.sum()), m, float(err.max()))
    # a second product with the same object gives the same bits
    y2 = torch.empty_like(y); dm.mult(xd, y2)
    assert torch.equal(y, y2)
    return dm


def _rows(lengths, n, rng, local=None):
    """CSR with the given row lengths; columns random in [0, n) or, with local=w, inside a window of width w"""
    lengths = np.asarray(lengths, dtype=np.int64)
    m = lengths.size
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    cols = np.empty(indptr[-1], dtype=np.int64)
    for i in range(m):
        k = int(lengths[i])
        if k == 0:
            continue
        if local is None:
            c = rng.choice(n, size=k, replace=False) if k <= n else np.arange(n)
        else:
            lo = int(rng.integers(0, max(1, n - local)))
            c = lo + rng.choice(min(local, n), size=k, replace=False)
        cols[indptr[i]:indptr[i + 1]] = np.sort(c)
    vals = rng.standard_normal(indptr[-1]) * 10.0 ** rng.integers(-3, 3, size=indptr[-1])
    return sp.csr_matrix((vals, cols, indptr), shape=(m, n))


# "mb16-mixed-with-packed": blocks of at most 4 rows (and blocks that cannot be packed) go multi-base, the others keep the
# packed form of rounds 2-3: one operator then holds both kinds and both launch parts (the lean kernels and the *_g kernels)
# run for one product
FORMATS = {"default": (1, 128), "mb16-off": (0, 128), "mb16-mixed-with-packed": (1, 4)}


def _set_format(P, name):
    on, rows = FORMATS[name]
    assert P.lib().airhip_set_tuning(b"mb16", on) == 0
    assert P.lib().airhip_set_tuning(b"mb16_max_rows", rows) == 0


@pytest.fixture(scope="module", params=list(FORMATS))
def env(gpu, request):
    import torch
    import pflare_amd as P
    _set_format(P, request.param)
    yield torch, P
    _set_format(P, "default")


def test_packed_and_raw_blocks_on_a_wide_operator(env):
    """20 M columns: rows with local columns give packed blocks, rows with columns spread over the whole width
    give raw blocks (span >= 2^19), both kinds interleaved in one matrix"""
    torch, P = env
    rng = np.random.default_rng(1)
    n = 20_000_000
    m = 60_000
    lengths = rng.integers(1, 9, size=m)
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    cols = np.empty(indptr[-1], dtype=np.int64)
    wide = (np.arange(m) // 300) % 2 == 1            # alternate stretches of local / scattered rows
    for i in range(m):
        k = lengths[i]
        if wide[i]:
            c = rng.integers(0, n, size=k)
        else:
            lo = (i * 300) % (n - 5000)
            c = lo + rng.integers(0, 5000, size=k)
        cols[indptr[i]:indptr[i + 1]] = c
    vals = rng.standard_normal(indptr[-1])
    A = sp.csr_matrix((vals, cols, indptr), shape=(m, n))
    _check(A, torch, P)


@pytest.mark.parametrize("wide", [False, True], ids=["packed", "raw"])
def test_every_round_count_variant(env, wide):
    """one row per block (rows of 257 ... 512 entries cannot share a block; shorter ones are separated by an empty-free
    filler of 512): every number of full rounds 0 ... 8, with remainders 0, 1, 63"""
    torch, P = env
    rng = np.random.default_rng(7 + int(wide))
    n = 3_000_000 if wide else 40_000
    lengths = []
    for nf in range(0, 9):
        for rem in (0, 1, 31, 63):
            L = 64 * nf + rem
            if 0 < L <= 512:
                lengths += [L, 512]          # the 512-entry row after it closes the block
    lengths += [64, 64, 64, 64, 64, 64, 64, 64, 1, 448, 64, 449, 63]     # several rows per block as well
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    cols = np.empty(indptr[-1], dtype=np.int64)
    for i, L in enumerate(lengths):
        if wide:
            c = rng.choice(n, size=L, replace=False)                      # span >= 2^19 almost surely
        else:
            lo = (i * 97) % (n - 4000)
            c = lo + rng.choice(4000, size=L, replace=False)
        cols[indptr[i]:indptr[i + 1]] = np.sort(c)
    A = sp.csr_matrix((rng.standard_normal(indptr[-1]), cols, indptr), shape=(len(lengths), n))
    _check(A, torch, P)


def test_long_rows_and_block_size_edges(env):
    torch, P = env
    rng = np.random.default_rng(2)
    n = 30_000
    lengths = [3, 512, 4, 513, 5, 2000, 0, 0, 1, 511, 1, 1024, 7, 7, 7, 4097, 2]
    _check(_rows(lengths, n, rng), torch, P)
    # a matrix that is one long row, and one with only long rows
    _check(_rows([5000], n, rng), torch, P)
    _check(_rows([600, 700, 800, 513], n, rng), torch, P)


@pytest.mark.parametrize("per_row", [1, 2, 3, 4, 7, 15, 16, 17, 31, 33, 63, 65, 100, 128, 200, 256, 300, 511, 512])
def test_every_lanes_per_row_mode(env, per_row):
    """uniform row length L: blocks hold min(128, 512 // L) rows, so L sweeps T = 1 (two rows per lane) ... 64"""
    torch, P = env
    rng = np.random.default_rng(100 + per_row)
    m = 3 * 128 + 5
    _check(_rows([per_row] * m, 6000, rng, local=2500), torch, P)


def test_empty_rows_and_tiny_matrices(env):
    torch, P = env
    rng = np.random.default_rng(3)
    _check(_rows([0] * 700, 50, rng), torch, P)                       # only empty rows (blocks without entries)
    _check(_rows([0, 0, 5, 0, 0, 0, 9, 0], 40, rng), torch, P)
    _check(_rows([1], 1, rng), torch, P)
    _check(_rows([2], 7, rng), torch, P)
    lengths = np.where(rng.random(5000) < 0.7, 0, rng.integers(1, 30, size=5000))
    _check(_rows(lengths, 5000, rng, local=800), torch, P)


@pytest.mark.parametrize("nblocks", [1, 2, 3, 4, 5, 9])
def test_few_blocks(env, nblocks):
    """fewer blocks than the 8 x 4 waves of the smallest grid: most waves have no block"""
    torch, P = env
    rng = np.random.default_rng(200 + nblocks)
    _check(_rows([8] * (64 * nblocks), 3000, rng, local=1000), torch, P)


def test_many_blocks_random_structure(env):
    """a few hundred thousand blocks: every wave of the resident grid loops over many blocks"""
    torch, P = env
    rng = np.random.default_rng(4)
    m = 1_500_000
    lengths = rng.integers(0, 13, size=m)
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    rows = np.repeat(np.arange(m), lengths)
    # columns near the diagonal with a few far ones (duplicates are summed by _check)
    cols = (rows + rng.integers(-2000, 2000, size=rows.size)) % m
    far = rng.random(rows.size) < 0.05
    cols[far] = rng.integers(0, m, size=int(far.sum()))
    A = sp.csr_matrix((rng.standard_normal(rows.size), (rows, cols)), shape=(m, m))
    _check(A, torch, P)


def test_banded_operator_with_nan_free_result_for_inf_free_input(env):
    """values of x with large dynamic range; padding entries (past a block's end) must not leak into a row sum"""
    torch, P = env
    m = 40_000
    d = [np.full(m, 4.0), np.full(m - 1, -1.0), np.full(m - 1, -1.0), np.full(m - 200, -0.5)]
    A = sp.diags(d, [0, -1, 1, -200]).tocsr()
    x = lcg_vector(m) * 10.0 ** (np.arange(m) % 60 - 30)
    _check(A, torch, P, x=x)


@pytest.mark.parametrize("itype,order,diag_scale", [("arnoldi", 6, False), ("arnoldi", 4, True), ("neumann", 5, True),
                                                    ("newton", 8, False), ("newton", 5, True)])
def test_fused_epilogues_through_matrix_free_polynomials(env, itype, order, diag_scale):
    """Horner / Newton / Neumann applies use alpha, beta*z, gamma*w, dscale and the auxiliary y2 update of the SpMV
    epilogue; checked against the oracle's apply on the same coefficients (<= 4e-11: PFLARE_TOL_MATFREE_4EM11)"""
    torch, P = env
    from oracle import air_ref as R, bridge
    A = R.gen_adv_diff_fd(2, 40, 40, alpha=0.05)
    n = A.shape[0]
    pc = P.PCPFLAREINV(type=itype, poly_order=order, matrix_free=True, diag_scale=diag_scale).setup(A)
    op = bridge._inv_from_dict(pc.get_inverse(n), A)
    x = lcg_vector(n)
    y = torch.empty(n, dtype=torch.float64, device="cuda")
    pc.apply(torch.from_numpy(x).cuda(), y)
    ref = R.apply_inv(op, x)
    assert np.linalg.norm(y.cpu().numpy() - ref) <= 4e-11 * np.linalg.norm(ref)


def test_tuning_hook_refuses_unknown_keys(env):
    """airhip_set_tuning is declared in the header; an unknown key is refused with AIRHIP_ERR_ARG and an error text"""
    torch, P = env
    assert P.lib().airhip_set_tuning(b"no_such_key", 1) != 0
    assert b"no_such_key" in P.lib().airhip_last_error()


def _val_slot(s, n):
    """memory position of the value (and of the 16-bit word) of logical slot s in a block of n <= 512 entries: every pair of
    full rounds of 64 is interleaved lane by lane (spmv.hip: val_slot)"""
    npair = (n >> 6) >> 1
    s = np.asarray(s)
    inter = (s >> 7) * 128 + 2 * (s & 63) + ((s >> 6) & 1)
    return np.where(s >= 128 * npair, s, inter)


def decode_blocks(dm):
    """the CSR a DeviceMat holds, decoded from its row blocks on the host (every kind of index region, dev.hpp BlkDesc)"""
    desc, idx, val, roff = dm.blocks()
    m = dm.shape[0]
    indptr = np.zeros(m + 1, dtype=np.int64)
    cols_out, vals_out = [], []
    kinds = {"packed": 0, "raw": 0, "multi-base": 0, "dictionary": 0}
    q0 = 0              # non-zeros before the block (p0 is its place in the value array, where dictionary blocks are shorter)
    expect_p0 = 0
    for d in desc:
        p0, row, cbase, nb, nr, ip = (int(d[k]) for k in ("p0", "row", "cbase", "nb", "nr", "ip"))
        if nb == 0:
            continue
        assert ip % 4 == 0
        assert p0 == expect_p0, "the value regions are not back to back"
        n512 = min(nb, 512)
        slot = np.arange(nb)
        vpos = np.where(slot < n512, _val_slot(np.minimum(slot, n512 - 1), n512), slot)
        has_dict = cbase <= -2 and ((-(cbase + 2)) >> 12) > 0
        v = None if has_dict else val[p0 + vpos]             # value of logical slot s (a dictionary block: decoded below)
        if cbase >= 0:
            w = idx[ip:ip + 4 * nb].view("<u4")
            col = cbase + (w >> 13).astype(np.int64)
            pos = (w & 1023).astype(np.int64) - 1            # row-major position of slot s
            assert np.array_equal(np.sort(pos), np.arange(nb)), "positions of a packed block are not a permutation"
            assert np.all(np.diff(col) >= 0), "a packed block is not sorted by column"
            c_rm = np.empty(nb, dtype=np.int64); v_rm = np.empty(nb)
            c_rm[pos] = col; v_rm[pos] = v
            kinds["packed"] += 1
        elif cbase == -1:
            c_rm = idx[ip:ip + 4 * nb].view("<u4").astype(np.int64); v_rm = v
            kinds["raw"] += 1
        else:
            meta = -(cbase + 2); ob = meta & 31; nwin = (meta >> 5) & 127; ndv = meta >> 12
            assert 1 <= nwin <= 64 and nwin <= (1 << (16 - ob))      # (a long row may be multi-base as well)
            if ndv:
                # value dictionary (dev.hpp): ndv distinct values in ascending bit pattern order, then 512 bytes of indices,
                # byte 8 l + u = index of entry 64 u + l - the block's region of the value array is these ndv + 64 doubles
                assert nb <= 512 and 1 <= ndv <= 256 and 8 * (ndv + 64) <= 5 * nb
                table = val[p0:p0 + ndv]
                tb = table.view("<u8")
                assert np.all(np.diff(tb.astype(object)) > 0), "dictionary not in ascending bit pattern order"
                ib = val[p0 + ndv:p0 + ndv + 64].view("u1")
                e = np.arange(nb)
                vid = ib[(e & 63) * 8 + (e >> 6)]
                assert np.all(vid < ndv)
                pad = np.arange(nb, 512)
                assert np.all(ib[(pad & 63) * 8 + (pad >> 6)] == 0)
                v = table[vid]
                kinds["dictionary"] += 1
            tab = idx[ip:ip + 4 * nwin].view("<u4").astype(np.int64)
            assert np.all(tab % 8 == 0) and np.all(np.diff(tab) > 0)
            words = idx[ip + 4 * nwin:ip + 4 * nwin + 2 * nb].view("<u2").astype(np.int64)
            wd = words[vpos]                                 # word of logical slot s = row-major position s (placed like the values)
            assert np.all((wd >> ob) < nwin)
            c_rm = tab[wd >> ob] // 8 + (wd & ((1 << ob) - 1)); v_rm = v
            kinds["multi-base"] += 1
        # rows of the block
        ro = roff[row:row + nr].astype(np.int64)
        if nb > 512:
            assert nr == 1
            ro = np.zeros(1, dtype=np.int64)
        ends = np.concatenate([ro[1:], [nb]])
        assert ro[0] == 0 and np.all(ends >= ro)
        indptr[row + 1:row + nr + 1] = ends - ro
        cols_out.append((q0, c_rm)); vals_out.append(v_rm)
        q0 += nb
        expect_p0 += (ndv + 64) if (cbase <= -2 and ndv) else nb
    assert expect_p0 == val.size, (expect_p0, val.size)
    np.cumsum(indptr, out=indptr)
    nnz = int(indptr[-1])
    cols = np.zeros(nnz, dtype=np.int64); vals = np.zeros(nnz)
    for (p0, c), v in zip(cols_out, vals_out):
        cols[p0:p0 + c.size] = c; vals[p0:p0 + c.size] = v
    return indptr, cols, vals, kinds


def test_row_blocks_decode_to_the_matrix_that_went_in(env):
    """index work is bit-exact: the device's row blocks - packed, raw and multi-base regions, interleaved values, row
    offsets - decoded on the host give back the CSR entry for entry"""
    torch, P = env
    rng = np.random.default_rng(21)
    mats = [_rows([per_row] * 200, 70_000, rng, local=30_000) for per_row in (1, 3, 8, 40, 64, 65, 100, 127, 128, 129, 200, 256, 300, 511, 512)]
    mats += [_rows([per_row] * 150, 3_000_000, rng) for per_row in (2, 9, 70, 300)]
    mats.append(_rows(rng.integers(0, 600, size=1500), 500_000, rng, local=9000))
    mats.append(_rows([5, 513, 700, 2, 4097, 1, 512, 511, 0, 0, 3], 40_000, rng))
    # rows that touch a few far-apart clusters (the sections of a hierarchical vector): windows of every width
    m = 3000
    lengths = rng.integers(1, 60, size=m)
    indptr = np.concatenate([[0], np.cumsum(lengths)])
    cols = np.empty(indptr[-1], dtype=np.int64)
    for i in range(m):
        k = int(lengths[i])
        centres = np.array([0, 900_000, 1_400_000, 1_650_000, 1_780_000]) + (i // 4) * 37
        c = centres[rng.integers(0, 5, size=k)] + rng.integers(0, 3000, size=k)
        cols[indptr[i]:indptr[i + 1]] = np.sort(c)
    A5 = sp.csr_matrix((rng.standard_normal(indptr[-1]), cols, indptr), shape=(m, 2_000_000))
    A5.sum_duplicates(); A5.sort_indices()
    mats.append(A5)
    seen = {"packed": 0, "raw": 0, "multi-base": 0, "dictionary": 0}
    for A in mats:
        A = sp.csr_matrix(A); A.sum_duplicates(); A.sort_indices()
        dm = P.DeviceMat(A)
        ip, ci, v, kinds = decode_blocks(dm)
        assert np.array_equal(ip, A.indptr.astype(np.int64))
        assert np.array_equal(ci, A.indices.astype(np.int64))
        assert np.array_equal(v, A.data)
        for k in kinds:
            seen[k] += kinds[k]
    assert seen["packed"] + seen["raw"] + seen["multi-base"] > 0


def test_the_three_conversions_agree_bit_for_bit(gpu):
    """the same operators converted with multi-base blocks off / by default / everywhere: identical products.  Rows of
    every length class, local and scattered columns (windows of every width, blocks that cannot be packed), 1 ... 128 rows
    per block, long rows"""
    import torch
    import pflare_amd as P
    rng = np.random.default_rng(11)
    mats = []
    for per_row in (1, 3, 8, 40, 64, 100, 200, 256, 300, 511, 512):
        mats.append(_rows([per_row] * 300, 70_000, rng, local=30_000))        # windows of 2^10 ... 2^14 entries all occur
        mats.append(_rows([per_row] * 300, 3_000_000, rng))                   # span >= 2^19: multi-base or raw
    mats.append(_rows(rng.integers(0, 600, size=2000), 500_000, rng, local=9000))
    mats.append(_rows([5, 513, 700, 2, 4097, 1, 512, 511], 40_000, rng))
    try:
        outs = {}
        for name in FORMATS:
            _set_format(P, name)
            outs[name] = []
            for A in mats:
                dm = P.DeviceMat(A)
                x = torch.from_numpy(lcg_vector(A.shape[1])).cuda()
                y = torch.empty(A.shape[0], dtype=torch.float64, device="cuda")
                dm.mult(x, y)
                outs[name].append(y)
        for name in ("mb16-off", "mb16-mixed-with-packed"):
            for k, (a, b) in enumerate(zip(outs["default"], outs[name])):
                assert torch.equal(a, b), (name, k)
    finally:
        _set_format(P, "default")


def test_value_dictionary_blocks_give_the_plain_blocks_bits(gpu):
    """operators whose values repeat (the first levels of a constant-coefficient problem): the blocks that take the
    dictionary form decode to the matrix that went in, and the product is the plain-value product bit for bit; blocks with
    too many distinct values, small blocks and rows longer than a block keep their plain values"""
    import torch
    import pflare_amd as P
    rng = np.random.default_rng(5)
    lib = P.lib()

    def few_values(A, k):
        A = A.copy()
        pool = rng.standard_normal(k) * 10.0 ** rng.integers(-3, 3, size=k)
        if k > 2:
            pool[1] = -pool[0]; pool[2] = 0.0           # -x / x and an explicit zero among the values
        A.data = pool[rng.integers(0, k, size=A.nnz)]
        return A
    mats = []
    for per_row, k in ((4, 2), (3, 1), (8, 5), (40, 43), (64, 150), (100, 256), (100, 257), (200, 300), (511, 3), (512, 7), (1, 2)):
        mats.append(few_values(_rows([per_row] * 700, 70_000, rng, local=30_000), k))
    mats.append(few_values(_rows(rng.integers(0, 60, size=4000), 500_000, rng, local=9000), 17))          # ragged rows, empty rows
    mats.append(few_values(_rows([5, 513, 700, 2, 4097, 1, 512, 511, 0, 0, 3] * 20, 40_000, rng), 4))        # long rows: plain
    x_cache = {}
    try:
        got = {}
        for mode in (1, 0):
            assert lib.airhip_set_tuning(b"vdict", mode) == 0
            got[mode] = []
            for A in mats:
                dm = P.DeviceMat(A)
                if mode == 1:
                    indptr, cols, vals, kinds = decode_blocks(dm)
                    B = sp.csr_matrix((vals, cols, indptr), shape=A.shape)
                    assert (B != A).nnz == 0 and np.array_equal(B.indptr, A.indptr)
                    assert np.array_equal(np.asarray(vals).view("<u8"), A.data.view("<u8"))      # bit patterns (signed zeros too)
                    got.setdefault("kinds", []).append(kinds)
                n = A.shape[1]
                x = x_cache.setdefault(n, torch.from_numpy(lcg_vector(n)).cuda())
                y = torch.empty(A.shape[0], dtype=torch.float64, device="cuda")
                dm.mult(x, y)
                got[mode].append(y)
        for k, (a, b) in enumerate(zip(got[1], got[0])):
            assert torch.equal(a, b), k
        kinds = got["kinds"]
        assert kinds[0]["dictionary"] > 0 and kinds[0]["dictionary"] == kinds[0]["multi-base"]      # 2 values: every block
        assert kinds[7]["dictionary"] == 0            # 300 values over blocks of two 200-entry rows: ~220 distinct, 8 (220 + 64) > 5/8 of 3200 bytes
        assert kinds[10]["dictionary"] > 0            # one entry per row, two values: a 128-entry block pays already (528 against 1024 bytes)
        assert kinds[12]["multi-base"] + kinds[12]["raw"] > kinds[12]["dictionary"]      # rows longer than a block keep plain values
    finally:
        lib.airhip_set_tuning(b"vdict", 1)
