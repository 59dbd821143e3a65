"""Test harness rules (round-1 lesson: the driver's GPU run was killed at its limit without a trace):

* host threads are pinned before any native library loads - OpenMP's default of one thread per visible CPU
  oversubscribes a box whose CPU share is a cgroup quota (16x slower setups with 64 threads on 8 CPUs);
* every test is bounded (pytest.ini: pytest-timeout + faulthandler), every subprocess / spawned rank has a deadline
  (tests/proc_util.py);
* one flushed line per finished test goes to stderr and, when writable, to gpurun_out/pytest_progress.log, so a
  killed run names its position;
* collection order: parity tests first, heavy full-size configs next, subprocess and multi-process tests last
  (with -x, everything before the first failure still counts).
"""
import os
import sys
import time


def _cpu_share():
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except Exception:
        pass
    return max(1, n)


HOST_THREADS = min(16, _cpu_share())
os.environ.setdefault("OMP_NUM_THREADS", str(HOST_THREADS))
os.environ.setdefault("OPENBLAS_NUM_THREADS", str(min(8, HOST_THREADS)))
os.environ.setdefault("MKL_NUM_THREADS", str(min(8, HOST_THREADS)))

import pytest  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# parity proper first; anything that starts other processes last
_ORDER = ["test_gpu_parity.py", "test_gpu_kernel_paths.py", "test_golden.py", "test_gpu_streams.py", "test_gpu_setup.py", "test_capi.py",
          "test_reference_data.py", "test_gpu_baseline_configs.py", "test_gpu_large.py", "test_gpu_full_size.py"]
_LAST = ["test_c_example.py", "test_gpu_dist.py", "test_bench_contract.py"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def pytest_collection_modifyitems(config, items):
    def key(it):
        f = os.path.basename(str(it.fspath))
        if f in _ORDER:
            return (0, _ORDER.index(f))
        if f in _LAST:
            return (2, _LAST.index(f))
        return (1, 0)
    items.sort(key=key)          # stable: the order inside a file is kept


_T0 = time.time()
_progress = None


def _progress_file():
    global _progress
    if _progress is None:
        _progress = False
        d = os.path.join(ROOT, "gpurun_out")
        try:
            os.makedirs(d, exist_ok=True)
            _progress = open(os.path.join(d, "pytest_progress.log"), "a", buffering=1)
            _progress.write(f"# session pid {os.getpid()} at {time.strftime('%H:%M:%S')} threads {HOST_THREADS}\n")
        except Exception:
            _progress = False
    return _progress


def pytest_runtest_logstart(nodeid, location):
    f = _progress_file()
    if f:
        f.write(f"[{time.time() - _T0:7.1f}s] start  {nodeid}\n")


def pytest_runtest_logreport(report):
    if report.when != "call" and not (report.when == "setup" and report.outcome != "passed"):
        return
    line = f"[{time.time() - _T0:7.1f}s] {report.outcome:7s} {report.duration:6.1f}s {report.nodeid}\n"
    try:
        sys.__stderr__.write(line); sys.__stderr__.flush()
    except Exception:
        pass
    f = _progress_file()
    if f:
        f.write(line)


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _have_gpu():
        pytest.skip("no GPU visible")
    import torch
    return torch.device("cuda:0")
