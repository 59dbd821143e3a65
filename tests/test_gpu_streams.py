"""Stream contract of the C ABI (include/airhip.h, "Stream contract"): the library's private stream is ordered after
the caller's stream on entry, and - for the *_on_stream entry points - the caller's stream after the library's on exit,
without host synchronisation.  Round 1 had neither: a vector filled asynchronously right before a call could be read
before the fill had run."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env(gpu):
    import torch
    import pflare_amd as P
    A = P.HostMat.adv_diff_fd(dim=3, M=40, N=40, P=40)
    pc = P.PCAIR(a_lump=True).setup(A)
    return torch, P, A, pc


def _busy(torch, stream, n=4096, reps=6):
    """queue ~tens of milliseconds of work on `stream` so that whatever follows it is still pending"""
    with torch.cuda.stream(stream):
        a = torch.ones((n, n), device="cuda", dtype=torch.float32)
        for _ in range(reps):
            a = (a @ a) * (1.0 / n)
    return a


def test_apply_reads_inputs_produced_on_the_callers_stream_and_hands_the_result_back(env):
    torch, P, A, pc = env
    n = A.shape[0]
    xs = torch.from_numpy(np.linspace(-1.0, 1.0, n)).cuda()
    ref = torch.empty_like(xs)
    pc.apply(xs, ref); torch.cuda.synchronize()
    side = torch.cuda.Stream()
    x = torch.zeros_like(xs); y = torch.full_like(xs, float("nan")); out = torch.empty_like(xs)
    torch.cuda.synchronize()
    _busy(torch, side)
    with torch.cuda.stream(side):
        x.copy_(xs, non_blocking=True)        # still pending behind the busy work when apply is called
        pc.apply(x, y)                        # airhip_pcair_apply_on_stream(current stream = side)
        out.copy_(y * 2.0, non_blocking=True)   # consumes y on the caller's stream, no host synchronisation in between
    side.synchronize()
    assert torch.equal(out, ref * 2.0)


def test_solve_with_an_asynchronously_filled_initial_guess(env):
    torch, P, A, pc = env
    n = A.shape[0]
    ksp = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50)
    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    x_ref = torch.ones_like(b)
    torch.cuda.synchronize()
    r_ref = ksp.solve(b, x_ref, pc=pc)
    side = torch.cuda.Stream()
    x = torch.full_like(b, 123.0)
    torch.cuda.synchronize()
    _busy(torch, side)
    with torch.cuda.stream(side):
        x.fill_(1.0)                          # queued behind the busy work
        r = ksp.solve(b, x, pc=pc)
    assert r["its"] == r_ref["its"] and r["reason"] == r_ref["reason"]
    assert torch.equal(x, x_ref)


def test_legacy_entry_points_order_after_the_default_stream(env):
    """airhip_pcair_apply (no stream argument) waits for the legacy default stream; torch's default current stream
    is that stream"""
    import ctypes as C
    torch, P, A, pc = env
    n = A.shape[0]
    xs = torch.from_numpy(np.cos(np.arange(n) * 0.01)).cuda()
    ref = torch.empty_like(xs); pc.apply(xs, ref); torch.cuda.synchronize()
    x = torch.zeros_like(xs); y = torch.empty_like(xs)
    torch.cuda.synchronize()
    _busy(torch, torch.cuda.current_stream())
    x.copy_(xs, non_blocking=True)
    P.check(P.lib().airhip_pcair_apply(pc._h, C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr())))
    P.check(P.lib().airhip_pcair_sync(pc._h))
    assert torch.equal(y, ref)


def test_mat_mult_on_stream(env):
    torch, P, A, pc = env
    n = A.shape[0]
    dm = P.DeviceMat(A)
    xs = torch.from_numpy(np.sin(np.arange(n) * 0.1)).cuda()
    ref = torch.empty_like(xs); dm.mult(xs, ref)
    side = torch.cuda.Stream()
    x = torch.zeros_like(xs); y = torch.empty_like(xs); out = torch.empty_like(xs)
    torch.cuda.synchronize()
    _busy(torch, side)
    with torch.cuda.stream(side):
        x.copy_(xs, non_blocking=True)
        dm.mult_async(x, y)
        out.copy_(y + 1.0, non_blocking=True)
    side.synchronize()
    assert torch.equal(out, ref + 1.0)


def test_chunked_permutation_is_the_plain_gather(gpu):
    """PCApply and KSPSolve take vectors in the caller's (natural) numbering; the permutation to the hierarchical layout and
    back runs chunk by chunk of the locality key (kernels.hip: k_permute_chunks) on operators large enough - bit for bit what
    the plain gathers give, with and without an ordering hint"""
    import numpy as np
    import torch
    import pflare_amd as P
    lib = P.lib()
    M = 64
    A = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
    n = A.shape[0]
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal(n)).cuda()
    try:
        for hinted in (False, True):
            pc = P.PCAIR(a_lump=True)
            if hinted:
                ax = np.arange(M, dtype=np.int64)
                key = ((ax[:, None, None] // 8) * 64 + (ax[None, :, None] // 8) * 8 + (ax[None, None, :] // 8)) * 512 + \
                      ((ax[:, None, None] % 8) * 64 + (ax[None, :, None] % 8) * 8 + (ax[None, None, :] % 8))
                pc.set_ordering_hint(key.reshape(-1))
            pc.setup(A)
            out = {}
            for mode in (1, 0):
                assert lib.airhip_set_tuning(b"perm_chunks", mode) == 0
                y = torch.empty_like(x)
                pc.apply(x, y)
                b = torch.zeros(n, dtype=torch.float64, device="cuda"); sol = torch.ones_like(b)
                r = P.KSP("gmres", pc_side="right", rtol=1e-10, atol=1e-50, max_it=30).solve(b, sol, pc=pc)
                out[mode] = (y, sol, r["history"])
            assert torch.equal(out[1][0], out[0][0]) and torch.equal(out[1][1], out[0][1]) and out[1][2] == out[0][2], hinted
    finally:
        lib.airhip_set_tuning(b"perm_chunks", 1)
