"""The MPI host layer (include/airhip_mpi.h, pflare_amd/csrc/mpi_dist.c) and its C driver examples/adv_diff_fd_mpi.c - the
shape of the reference's tests/adv_diff_fd.c run under `mpiexec -n 2` (tests/Makefile, run_tests_*_parallel).  On the CPU:
the library builds against the MPI of this image, exports what the header declares, and two ranks gather the operator, build
the hierarchy on rank 0, cut it and ship the pieces over MPI (host_only: no device).  On the GPU: two ranks sharing the one
device solve with the halo exchange carried by MPI_Isend / MPI_Irecv and reproduce the single-rank iteration count."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MPIEXEC = shutil.which("mpiexec") or "/opt/conda/bin/mpiexec"
HAVE_MPI = os.path.exists(MPIEXEC) and os.path.exists("/opt/conda/include/mpi.h")
pytestmark = pytest.mark.skipif(not HAVE_MPI, reason="no MPI installation in this image")


def _build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pflare_amd", "csrc"), "mpi"], check=True, timeout=300)
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "examples"), "adv_diff_fd_mpi"], check=True, timeout=300)
    return os.path.join(ROOT, "examples", "adv_diff_fd_mpi")


def _env():
    e = dict(os.environ)
    e.setdefault("OMP_NUM_THREADS", "2")        # two ranks share the CPUs of the test box
    return e


def test_mpi_library_exports_the_declared_symbols():
    _build()
    hdr = open(os.path.join(ROOT, "include", "airhip_mpi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(airhip_mpi_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 7
    out = subprocess.run(["nm", "-D", "--defined-only", os.path.join(ROOT, "pflare_amd", "libairhip_mpi.so")],
                         capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (airhip_mpi_[a-z0-9_]+)", out))
    assert set(declared) == exported, (set(declared) ^ exported)


@pytest.mark.parametrize("ranks", [2, 3])
def test_two_ranks_gather_cut_and_ship_over_mpi_without_a_device(ranks):
    exe = _build()
    r = subprocess.run([MPIEXEC, "-n", str(ranks), exe, "-dim", "3", "-da_grid_x", "12", "-pc_air_a_lump", "1", "-host_only", "1"],
                       capture_output=True, text=True, timeout=300, env=_env())
    assert r.returncode == 0, r.stdout + r.stderr
    m = re.search(r"ranks=(\d+) n=(\d+) levels=(\d+) pieces=(\d+) bytes", r.stdout)
    assert m and int(m.group(1)) == ranks and int(m.group(2)) == 12 ** 3 and int(m.group(3)) >= 3 and int(m.group(4)) > 0, r.stdout
    # the same hierarchy whatever the number of ranks (rank 0 builds it; the setup is bit-reproducible)
    r1 = subprocess.run([MPIEXEC, "-n", "1", exe, "-dim", "3", "-da_grid_x", "12", "-pc_air_a_lump", "1", "-host_only", "1"],
                        capture_output=True, text=True, timeout=300, env=_env())
    assert r1.returncode == 0 and re.search(r"levels=(\d+)", r1.stdout).group(1) == m.group(3)


def test_a_failing_setup_does_not_leave_the_other_ranks_waiting():
    exe = _build()
    r = subprocess.run([MPIEXEC, "-n", "2", exe, "-dim", "2", "-da_grid_x", "8", "-pc_air_no_such_option", "1", "-host_only", "1"],
                       capture_output=True, text=True, timeout=120, env=_env())
    assert r.returncode != 0 and "setup failed" in r.stderr


@pytest.mark.gpu
def test_two_mpi_ranks_on_one_gpu_reproduce_the_single_rank_solve(gpu):
    exe = _build()
    args = ["-dim", "3", "-da_grid_x", "20", "-pc_air_a_lump", "1", "-ksp_max_it", "8", "-same_device", "1"]
    outs = {}
    for n in (1, 2):
        r = subprocess.run([MPIEXEC, "-n", str(n), exe] + args, capture_output=True, text=True, timeout=300, env=_env())
        assert r.returncode == 0, r.stdout + r.stderr
        m = re.search(r"iterations=(\d+) reason=(-?\d+) rnorm=([0-9.e+-]+) .* exchanges=(\d+) messages=(\d+)", r.stdout)
        assert m, r.stdout
        outs[n] = (int(m.group(1)), int(m.group(2)), float(m.group(3)), int(m.group(4)), int(m.group(5)))
    assert outs[1][:2] == outs[2][:2], outs                       # identical iteration count and reason
    assert abs(outs[1][2] - outs[2][2]) <= 1e-6 * max(outs[1][2], 1e-300), outs
    assert outs[1][3] == 0 or outs[1][4] == 0                     # one rank: nothing travels
    assert outs[2][3] > 0 and outs[2][4] > 0                      # two ranks: the halo went over MPI
