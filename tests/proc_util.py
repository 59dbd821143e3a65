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


def free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _spawn_once(fn, args, nprocs, deadline):
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


def spawn(fn, args, nprocs, deadline=180, port_index=1):
    """torch.multiprocessing.spawn joined with a deadline: on expiry the ranks (exact PIDs) are killed and the
    test fails instead of hanging in gloo's 30-minute default.  args[port_index] is the rendezvous port (a port found free a
    moment ago): when the store cannot listen on it after all (EADDRINUSE - the system handed the port to an outgoing
    connection in between; seen once in the GPU suite, round 5) the ranks are started again on a fresh port, twice at most.
    Only that error is retried."""
    for attempt in range(3):
        try:
            return _spawn_once(fn, args, nprocs, deadline)
        except Exception as e:      # noqa: BLE001 - re-raised unless it is the port race
            if attempt == 2 or "EADDRINUSE" not in str(e) or port_index is None:
                raise
            args = tuple(free_port() if k == port_index else a for k, a in enumerate(args))


def init_rank_env(rank, world, port):
    """environment of a spawned rank, set before torch / the native library load"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      OMP_NUM_THREADS=str(rank_threads(world)), OMP_WAIT_POLICY="PASSIVE",
                      OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")


def pg_timeout():
    import datetime
    return datetime.timedelta(seconds=120)
