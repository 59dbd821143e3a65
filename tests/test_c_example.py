"""The C ABI from plain C: examples/adv_1d.c (a driver shaped like the reference's tests/adv_1d.c) is
compiled with gcc against include/airhip.h and libairhip.so.  Without a GPU it must fail loudly (no CPU
fallback); on the GPU it must converge within the reference's bound (tests/Makefile:537-540: -ksp_max_it 2)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def _build():
    subprocess.run(["make", "-s", "-C", EX, "clean"], check=True)
    subprocess.run(["make", "-s", "-C", EX], check=True)
    return os.path.join(EX, "adv_1d")


def test_c_driver_builds_and_refuses_to_run_without_a_device():
    import torch
    exe = _build()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: covered by the gpu test")
    r = subprocess.run([exe, "-n", "200"], capture_output=True, text=True)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr and "no CPU fallback" in r.stderr


@pytest.mark.gpu
def test_c_driver_converges_within_the_reference_bound(gpu):
    exe = _build()
    r = subprocess.run([exe, "-n", "1000", "-ksp_max_it", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "iterations=" in r.stdout
    its = int(r.stdout.split("iterations=")[1].split()[0])
    assert its <= 2


@pytest.mark.gpu
def test_python_ex2_runs(gpu=None):
    """examples/ex2.py: the reference's python/ex2.py (2-D Laplacian, PCAIR configured through the pflare option API)"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "ex2.py"), "-m", "24", "-n", "20"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Norm of error" in r.stdout
