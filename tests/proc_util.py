"""Bounded child processes for the tests: nothing a test starts may outlive its deadline."""
import os
import subprocess
import time


def run(cmd, timeout=120, **kw):
    """subprocess.run with a deadline (kills the child's process group on expiry)."""
    kw.setdefault("capture_output", True)
    kw.setdefault("text", True)
    return subprocess.run(cmd, timeout=timeout, **kw)


def rank_threads(world):
    """host threads of one spawned rank: the test session's share divided by the ranks"""
    total = int(os.environ.get("OMP_NUM_THREADS", "8") or 8)
    return max(1, total // max(1, world))


def spawn(fn, args, nprocs, deadline=180):
    """torch.multiprocessing.spawn joined with a deadline: on expiry the ranks (exact PIDs) are killed and the
    test fails instead of hanging in gloo's 30-minute default."""
    import torch.multiprocessing as mp
    ctx = mp.spawn(fn, args=args, nprocs=nprocs, join=False)
    t_end = time.time() + deadline
    while True:
        if ctx.join(timeout=2):
            return
        if time.time() > t_end:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
            for p in ctx.processes:
                p.join(10)
            raise TimeoutError(f"spawned ranks did not finish within {deadline}s")


def init_rank_env(rank, world, port):
    """environment of a spawned rank, set before torch / the native library load"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS=str(rank_threads(world)), OMP_WAIT_POLICY="PASSIVE",
                      OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")


def pg_timeout():
    import datetime
    return datetime.timedelta(seconds=120)
