#!/usr/bin/env python3
"""bench.py - KSP iterations/sec of the MI355X-native PCAIR solve phase.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.

Workload at N=1 (BASELINE.json configs[2]): 3D upwind advection from tests/adv_diff_fd.c on a 464^3
grid (99 897 344 DoF, velocity (1,1,1)/sqrt(3), b = 0, x0 = 1), right-preconditioned GMRES(30) to
rtol 1e-10 (tests/Makefile:543-545 options) preconditioned by PCAIR with the reference's GPU
recipe (docs/gpus.md:34-38, docs/faq.md "Truncating the hierarchy"): -pc_air_a_lump, hierarchy
auto-truncation and a matrix-free Newton-basis GMRES polynomial as coarse solver.

A "step" is one Krylov iteration = 1 SpMV with A + 1 AIR V-cycle (PCApply) + the classical
Gram-Schmidt update.  Exactly K iterations are timed, spread over consecutive solves of the
reference problem (each solve restarts from x0 = 1 and runs to convergence or to the remaining
step budget, and ends with its solution update), bracketed by barrier + device synchronisation.
Inputs are resident in HBM before the timed region starts.

Solution update (round 5): the solver keeps the preconditioned basis z_j = M v_j it computes anyway
and updates x += Z y, where PETSc's KSPGMRES computes x += M (V y) with one more PCApply per solve
(csrc/device/ksp_dev.cpp: same Arnoldi process, residual norms and iteration count bit for bit).
`config.pc_applies_per_step` says how many V-cycles a step of the timed region cost, and
`config.iterations_per_s_petsc_solution_update` is the same measurement with PETSc's form
(airhip_set_tuning "ksp_keep_preconditioned" = 0: 22 V-cycles for the default 20 steps, not 20).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)

# defaults of the airhip_set_tuning keys (include/airhip.h): what --variants restores after each variant
TUNING_DEFAULTS = {"spgemm_pool_entries": 0, "spgemm_pool_chunk": -1, "block_order_min_blocks": 4096, "fc_fuse": 1, "occ_spmv": 6, "occ_fc": 6, "krylov_blocks": 512, "ksp_unnormalised": 1, "ksp_keep_preconditioned": 1, "fuse_first_sweep": 1, "fuse_first_sweep_run": 1, "profile_chain": 1, "perm_chunks": 1, "vdict": 1, "mb16": 1, "mb16_max_rows": 128, "fc_spatial_x": 1}

RECIPES = {
    # reference GPU recipe: lumping + auto truncation + Newton matrix-free coarse solver
    "trunc": dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1,
                  coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10),
    # same with the GMRES-polynomial F smoother applied matrix-free (-pc_air_matrix_free_polys)
    "trunc_mf": dict(a_lump=True, auto_truncate_start_level=1, auto_truncate_tol=1e-1, matrix_free_polys=True,
                     coarsest_inverse_type="newton", coarsest_matrix_free_polys=True, coarsest_poly_order=10),
    # tests/Makefile:544-545 as is (full hierarchy, assembled smoothers)
    "full": dict(a_lump=True),
}


def cpu_share():
    """CPUs this process may use: the affinity mask cut by the cgroup CPU quota (cpu.max / cfs_quota)"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, -(-int(q) // int(per))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=464, help="grid points per dimension (464 -> ~1e8 DoF)")
    ap.add_argument("--recipe", default="trunc", choices=sorted(RECIPES))
    ap.add_argument("--max-levels", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=5, help="Krylov iterations of one CPU baseline sample (1.5-2 s each at 464^3 on 16 threads: three samples = 25-30 s of CPU work)")
    ap.add_argument("--cpu-samples", type=int, default=3, help="CPU baseline samples (each --cpu-iters iterations from x0 = 1); the median is reported")
    ap.add_argument("--graph", type=int, default=0)
    ap.add_argument("--profile-kernels", type=int, default=1)
    ap.add_argument("--variants", default="", help="diagnostic: ';'-separated tunings to time on the same hierarchy, each a "
                    "','-separated list of key=value for airhip_set_tuning (keys: include/airhip.h)")
    ap.add_argument("--tuning", default="", help="diagnostic: ','-separated key=value for airhip_set_tuning BEFORE the setup "
                    "(e.g. mb16_max_rows=128: which row blocks are converted to the 16-bit multi-base form)")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--hint-brick", type=int, default=16,
                    help="locality hint for the device layout: grid points grouped in BxBxB bricks (0: none, -1: Morton "
                         "order of the bricks of size 4); the operator keeps the reference numbering")
    ap.add_argument("--hint-sub", type=int, default=0, help="diagnostic: with --hint-brick B, order the points of a brick in SxSxS sub-bricks")
    ap.add_argument("--hint-morton", type=int, default=0, help="diagnostic: with --hint-brick B > 0, number the bricks along a Morton curve instead of lexicographically")
    ap.add_argument("--brick", type=int, default=0, help="diagnostic: renumber the grid in BxBxB bricks before the setup (locality experiment; not the reference numbering)")
    ap.add_argument("--threads", type=int, default=0, help="host threads for setup + CPU baseline (0: cores / ranks, capped at 32 - measured best on the GPU box)")
    ap.add_argument("--device-trunc", type=int, default=1,
                    help="auto-truncation test on the device (1, default) or on the host (0); it changes the setup time "
                         "only, the hierarchy is identical (validated at 464^3 in round 2: 15 levels, 12 iterations)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a CHILD torch.distributed.run (nothing
        # here has touched the GPU yet - no torch import, no HIP call) and exit with its code
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    if "AIRHIP_DEVICE_TRUNC" not in os.environ:
        os.environ["AIRHIP_DEVICE_TRUNC"] = str(1 if args.device_trunc else 0)
    # a GPU memory fault aborts the process from the runtime's own thread: the library's SIGABRT note then names the step
    # (async-signal-safe, no SIGSEGV handler)
    os.environ.setdefault("AIRHIP_FAULT_STAGE", "1")

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the PCAIR solve phase has no CPU fallback")
    # rehearsal mode for a one-GPU box: AIRHIP_BENCH_COMM=staged runs all ranks on device 0 over gloo
    # with the host-staged halo transport (RCCL refuses two ranks on one device)
    staged = os.environ.get("AIRHIP_BENCH_COMM", "rccl") == "staged"
    if staged:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import pflare_amd as P
    P.lib().airhip_set_device(local_rank)
    for t_ in [t for t in args.tuning.split(",") if t.strip()]:
        if P.lib().airhip_set_tuning(t_.split("=")[0].strip().encode(), int(t_.split("=")[1])) != 0:
            raise SystemExit(f"unknown tuning {t_}")
    dist = meta = None
    # AIRHIP_BENCH_FORCE_DIST=1: run the distributed code path (RCCL communicator, partition, shipping,
    # collective timing) even with a single rank - a rehearsal of everything but the peer traffic
    dist_mode = world > 1 or os.environ.get("AIRHIP_BENCH_FORCE_DIST", "0") == "1"
    if dist_mode:
        # one process per GPU (torch.distributed.run): RCCL for the bulk data and the halo exchange,
        # a gloo group for small host-side metadata.  Rank 0 builds the hierarchy, so the others wait
        # for minutes: generous timeouts.
        import datetime
        import torch.distributed as dist
        for k_, v_ in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29577"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k_, v_)
        # the only long wait is rank 0's setup (about 4 minutes at 464^3): everything else must fail inside a lease
        t_setup_wait = datetime.timedelta(minutes=float(os.environ.get("AIRHIP_BENCH_SETUP_TIMEOUT_MIN", "15")))
        if staged:
            dist.init_process_group("gloo", timeout=t_setup_wait)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=t_setup_wait)
            meta = dist.new_group(backend="gloo", timeout=t_setup_wait)
    # host threads for the setup phase and the CPU baseline: this rank's share of the host
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # (in the distributed run rank 0 builds the hierarchy while the other ranks sleep in a receive, so it
    # may use the whole host up to the measured optimum of 32 threads)
    want = args.threads if args.threads > 0 else min(32, cores if rank == 0 else max(1, cores // max(1, args.gpus)))
    threads = int(os.environ.get("AIRHIP_NUM_THREADS", max(1, min(want, cores))))
    P.lib().airhip_set_num_threads(threads, None)
    log = (lambda *a: print(*a, file=sys.stderr, flush=True)) if (args.verbose and rank == 0) else (lambda *a: None)

    M = args.size
    opts = dict(RECIPES[args.recipe])
    if args.max_levels:
        opts["max_levels"] = args.max_levels
    n = M ** 3
    hm = pc = None
    info = None
    if rank == 0:
        t0 = time.time()
        hm = P.HostMat.adv_diff_fd(dim=3, M=M, N=M, P=M)
        nnzA = hm.nnz
        log(f"[bench] operator {M}^3: n={n} nnz={nnzA} generated in {time.time()-t0:.1f}s")
        if args.brick:
            import numpy as np
            import scipy.sparse as sp
            B = args.brick
            i, j, k = np.meshgrid(np.arange(M), np.arange(M), np.arange(M), indexing="ij")   # natural id = (k*M + j)*M + i
            nat = ((k * M + j) * M + i).ravel()
            nb = (M + B - 1) // B
            key = ((((k // B) * nb + (j // B)) * nb + (i // B)) * B ** 3 + ((k % B) * B + (j % B)) * B + (i % B)).ravel()
            order = nat[np.argsort(key, kind="stable")]          # new position -> natural id
            a = hm.csr()
            A = sp.csr_matrix((a.val, a.col, a.rowptr), shape=a.shape)
            A = A[order][:, order].tocsr(); A.sort_indices()
            hm = P.CsrArrays(A.indptr, A.indices, A.data, A.shape)
            log(f"[bench] renumbered in {B}^3 bricks")
        pc = P.PCAIR(**opts)
        if args.hint_brick == -1 and not args.brick:
            import numpy as np
            ax = np.arange(M, dtype=np.int64)

            def spread(v):          # interleave the bits of v with two zero bits
                r = np.zeros_like(v)
                for bit in range(10):
                    r |= ((v >> bit) & 1) << (3 * bit)
                return r
            sx = spread(ax)
            key = (sx[:, None, None] << 2) | (sx[None, :, None] << 1) | sx[None, None, :]     # [k, j, i]
            pc.set_ordering_hint(key.reshape(-1))
            del key
        if args.hint_brick > 0 and not args.brick:
            # what a DM would tell the preconditioner: the brick of each grid point.  Only the device
            # layout uses it (airhip_pcair_set_ordering_hint); operator and hierarchy are the reference's.
            import numpy as np
            B = args.hint_brick
            nbk = (M + B - 1) // B
            ax = np.arange(M, dtype=np.int64)
            bi, li = ax // B, ax % B
            if args.hint_morton:
                def spread3(v):
                    r = np.zeros_like(v)
                    for bit in range(10):
                        r |= ((v >> bit) & 1) << (3 * bit)
                    return r
                sb = spread3(bi)
                bkey = (sb[:, None, None] << 2) | (sb[None, :, None] << 1) | sb[None, None, :]
            else:
                bkey = (bi[:, None, None] * nbk + bi[None, :, None]) * nbk + bi[None, None, :]
            if args.hint_sub > 1 and B % args.hint_sub == 0:
                # diagnostic: S^3 sub-bricks inside every brick (two-level locality)
                S_ = args.hint_sub; q_ = B // S_
                sb, sl = li // S_, li % S_
                inner = ((sb[:, None, None] * q_ + sb[None, :, None]) * q_ + sb[None, None, :]) * S_ ** 3 + \
                        (sl[:, None, None] * S_ + sl[None, :, None]) * S_ + sl[None, None, :]
                key = bkey * B ** 3 + inner
            else:
                key = bkey * B ** 3 + (li[:, None, None] * B + li[None, :, None]) * B + li[None, None, :]      # [k, j, i]
            pc.set_ordering_hint(key.reshape(-1))
            del key
        if dist_mode:
            pc.set_option("device_upload", False)     # the hierarchy is cut and shipped below
        if args.verbose:
            pc.set_option("print_stats_timings", True)   # per-level progress lines (keeps long setups audible)
        t0 = time.time()
        pc.setup(hm)
        t_setup = time.time() - t0
        st = pc.setup_times()
        cx = pc.complexities()
        log(f"[bench] setup {t_setup:.1f}s (host {st['host_setup_s']:.1f}s, upload {st['upload_s']:.1f}s) levels={pc.num_levels} {cx}")
        if args.verbose:
            try:
                roll = {l.split(":")[0]: l.split(":")[1].strip() for l in open("/proc/self/smaps_rollup") if ":" in l}
                log(f"[bench] host memory after setup: Rss {roll.get('Rss')}, AnonHugePages {roll.get('AnonHugePages')}")
            except OSError:
                pass
        info = dict(nnzA=nnzA, t_setup=t_setup, st=st, cx=cx, levels=pc.num_levels)
    parallelism = "single GPU"
    comm = None
    transport = None            # "rccl" | "staged" (rehearsal on one GPU) | "staged-fallback" (RCCL failed its self-test)
    if dist_mode:
        from pflare_amd import dist as D
        px, py, pz = D.box_dims(world, (M, M, M))
        parallelism = (f"row-block: {px}x{py}x{pz} boxes of the {M}^3 grid with equal WORK (non-zeros of an iteration's products; equal "
                       f"boxes leave the downstream corner 1.85 x the upstream one's), one rank per GPU, hierarchy built on rank 0 "
                       f"and cut; RCCL halo exchange per sparse product, all-reduce for Krylov dots")
        # ranks > 0 get here minutes before rank 0 (which has been building the hierarchy): wait on the host
        # (gloo) rather than inside an RCCL collective that would spin on their GPUs for the whole setup
        dist.barrier(group=meta)
        if staged:
            comm = D.StagedComm()
            transport = "staged"
        else:
            transport = "rccl"
            # trust the RCCL transport only after a collective self-test (all-pairs pattern exchange + all-reduce
            # on device buffers); if any rank fails it, every rank falls back to the host-staged transport over
            # the gloo group - slower halo exchanges, same kernels, same results
            ok, why = 1, ""
            try:
                comm = D.RcclComm()
                ok = 1 if comm.selftest(1 << 16) else 0
                why = "" if ok else "pattern mismatch"
            except Exception as e:
                ok, why = 0, repr(e)
            flag = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=meta)
            if int(flag.item()) == 0:
                print(f"[bench] rank {rank}: RCCL transport failed its self-test ({why or 'on another rank'}); "
                      f"using the host-staged transport", file=sys.stderr, flush=True)
                comm = D.StagedComm(group=meta)
                transport = "staged-fallback"
                parallelism += " [RCCL self-test failed: host-staged halo transport over gloo]"
        t0 = time.time()
        owner = None
        if rank == 0:
            # boxes of equal work, not of equal size (pflare_amd/dist.py: balanced_box_owner); AIRHIP_BENCH_EQUAL_BOXES=1 for
            # the DMDA-style equal boxes
            import numpy as np
            equal = D.box_owner(M, M, M, world)
            if os.environ.get("AIRHIP_BENCH_EQUAL_BOXES", "0") == "1":
                owner = equal
            else:
                tw = time.time()
                wpt = D.point_work(pc, hm)
                owner = D.balanced_box_owner(wpt, (M, M, M), world)
                per = [np.bincount(o, weights=wpt, minlength=world) for o in (equal, owner)]
                info["balance"] = {"work_busiest_over_mean_equal_boxes": float(per[0].max() / per[0].mean()),
                                   "work_busiest_over_mean": float(per[1].max() / per[1].mean()),
                                   "rows_per_rank": [int(v) for v in np.bincount(owner, minlength=world)],
                                   "seconds": time.time() - tw}
                log(f"[bench] work-balanced boxes: {info['balance']}")
                del wpt, per
            del equal
        # the bulk arrays travel GPU -> GPU over RCCL only when RCCL has just proved itself; otherwise over gloo
        pcl, gid = D.distribute(pc, hm, owner, comm, src=0, meta_group=meta, bulk_group=None if transport == "rccl" else meta,
                                bulk_on_gpu=(transport == "rccl"), log=log)
        box = [info]
        dist.broadcast_object_list(box, src=0, group=meta)
        info = box[0]
        info["t_distribute"] = time.time() - t0
        info["phases"] = dict(pcl.distribute_phases)        # this rank's: rank 0 cuts and ships, the others receive
        log(f"[bench] partition + ship + upload: {info['t_distribute']:.1f}s {info['phases']}")
        if rank == 0:
            pc.release_host()
        pc = pcl
        hm = None
        n = int(gid.size)
        args.no_cpu_baseline = True      # the CPU leg belongs to rank 0 at N=1
    nnzA, t_setup, st, cx = info["nnzA"], info["t_setup"], info["st"], info["cx"]

    ksp_kw = dict(ksp_type="gmres", pc_side="right", rtol=1e-10, atol=1e-50)

    # ---- CPU baseline: the oracle's restatement of the same KSP iteration on the same hierarchy
    cpu = None
    if not args.no_cpu_baseline:
        import numpy as np
        from oracle import c_oracle
        share = cpu_share()
        cpu_threads = int(os.environ.get("AIRHIP_NUM_THREADS", 0)) or (args.threads if args.threads > 0 else max(1, min(share, 32)))
        cores = c_oracle.set_threads(cpu_threads)
        ch = c_oracle.CHierarchy(pc, copy=False)
        b0 = np.zeros(n); x1 = np.ones(n)
        rates, hist_c, its_c, dts = [], [], 0, []
        for _ in range(max(1, args.cpu_samples) if args.cpu_iters > 0 else 1):
            tc = time.time()
            _, its_c, _, _, hist_c = ch.ksp(hm, b0, x1, "gmres", "right", rtol=1e-300, atol=1e-300, max_it=args.cpu_iters)
            dt = time.time() - tc
            dts.append(dt); rates.append(its_c / dt if dt > 0 else 0.0)
        med = sorted(rates)[len(rates) // 2]
        cpu = {"value": med, "unit": "KSP iterations/s", "cores": cores, "kind": "port",
               "host_logical_cpus": os.cpu_count(), "cpu_share": share, "threads": cores,
               "samples": [round(r_, 4) for r_ in rates],
               "sample": f"median of {len(rates)} samples of {its_c} right-preconditioned GMRES iterations (SpMV + AIR V-cycle + CGS) "
                         f"of the same {M}^3 system on the same hierarchy, C/OpenMP restatement of the reference algorithm "
                         f"(oracle/air_oracle.c) on {cores} threads (host: {os.cpu_count()} logical CPUs, this process may use "
                         f"{share}), {sum(dts):.1f}s in all"}
        del ch
        log(f"[bench] cpu baseline: {cpu}")
    if not dist_mode:
        pc.release_host()
    hm = None

    def barrier():
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize()

    b = torch.zeros(n, dtype=torch.float64, device="cuda")
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    pc.use_graph(bool(args.graph))

    # ---- reference practice: a first solve to converge (also the iteration-count check)
    ksp = P.KSP(max_it=200, **ksp_kw)
    res0 = ksp.solve(b, x, pc=pc)
    log(f"[bench] converged solve: its={res0['its']} reason={res0['reason_name']} rnorm={res0['rnorm']:.3e} {res0['solve_ms']:.1f} ms")
    its_conv = max(res0["its"], 1)
    if cpu is not None and its_c > 0:
        # the oracle as the checker: residual norms of the CPU sample's iterations against the GPU solve's (same system,
        # same hierarchy, same Krylov method)
        k = min(len(hist_c), len(res0["history"]))
        d = max(abs(res0["history"][i] - hist_c[i]) / max(abs(hist_c[i]), 1e-300) for i in range(k)) if k else None
        cpu["check"] = {"residual_norms_compared": k, "max_rel_diff_gpu_vs_cpu": d}
        log(f"[bench] residual norms GPU vs CPU oracle over {k} entries: max relative difference {d}")

    applies = [0]

    def run_iterations(k):
        """exactly k Krylov iterations over consecutive solves of the reference problem"""
        done = 0
        ms = 0.0
        applies[0] = 0
        while done < k:
            x.fill_(1.0)
            r = P.KSP(max_it=k - done, **ksp_kw).solve(b, x, pc=pc)
            if r["its"] == 0:
                raise RuntimeError("solver made no progress")
            done += r["its"]
            ms += r["solve_ms"]
            applies[0] += r["pc_applies"]
        return ms

    if args.variants:
        base = {}
        for v in args.variants.split(";"):
            kv = dict((t.split("=")[0].strip(), int(t.split("=")[1])) for t in v.split(",") if t.strip())
            for k_, v_ in kv.items():
                base.setdefault(k_, None)
                check = P.lib().airhip_set_tuning(k_.encode(), v_)
                if check != 0:
                    raise SystemExit(f"unknown tuning {k_}={v_}")
            run_iterations(3)
            pc.profile(True); pc.profile_reset()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            run_iterations(args.steps)
            torch.cuda.synchronize(); w = time.perf_counter() - t0
            stv = [s_ for s_ in pc.profile_stats() if s_["calls"] > 0]
            pc.profile(False)
            tot = sum(s_["total_ms"] for s_ in stv if s_["level"] >= 0)
            sp_ms = sum(s_["total_ms"] for s_ in stv if s_["kernel"] == "k_rb_spmv")
            sp_by = sum(s_["alg_bytes"] * s_["calls"] for s_ in stv if s_["kernel"] == "k_rb_spmv")
            cyc = max(1, next((s_["calls"] for s_ in stv if s_["name"].startswith("restrict") and s_["level"] == 0), 1))
            print(f"[variant {v or 'default'}] {args.steps / w:8.2f} it/s  vcycle kernels {tot / cyc:8.3f} ms  "
                  f"{cx['cycle_alg_bytes'] / (tot / cyc * 1e-3) / 1e9:7.1f} GB/s  k_rb_spmv {sp_by / (sp_ms * 1e-3) / 1e9:7.1f} GB/s",
                  file=sys.stderr, flush=True)
            for s_ in sorted(stv, key=lambda q: -q["total_ms"]):
                a_ = s_["total_ms"] / s_["calls"]
                log(f" v {s_['name']:28s} L{s_['level']+1:<2d} calls={s_['calls']:4d} avg={a_:8.3f} ms  {s_['alg_bytes'] / (a_ * 1e-3) / 1e9:8.1f} GB/s")
            # back to the defaults (include/airhip.h, airhip_set_tuning) before the next variant
            for k_ in kv:
                P.lib().airhip_set_tuning(k_.encode(), TUNING_DEFAULTS.get(k_, 0))
    def graph_rehearsal():
        """AFTER the measurement (rank 0 has printed its line by the time it gets here): the same iterations with the
        distributed cycle captured in a hipGraph (RCCL send / receive groups inside the capture).  Never run with N > 1
        before the driver's scaling run, so it must not be able to cost the line; the result goes to stderr."""
        # OPT-IN (AIRHIP_BENCH_GRAPH_REHEARSAL=1): captured RCCL groups have only ever run at world size 1 here; a hard fault in a
        # first run on N devices would turn a good measurement into a failed process, so the driver's plain runs skip it
        if not (dist_mode and transport == "rccl" and not args.graph and os.environ.get("AIRHIP_BENCH_GRAPH_REHEARSAL", "0") == "1"):
            return
        # a hang in here (a first run of captured RCCL groups on N > 1 devices) must not keep the ranks alive: the line is out,
        # so after three minutes every rank ends itself - with a NON-ZERO exit code (3) and a marker line: a hang with device
        # work in flight is a failure that needs a root cause, not a pass (the measured line above it stays readable)
        import threading
        dog = threading.Timer(float(os.environ.get("AIRHIP_BENCH_GRAPH_REHEARSAL_TIMEOUT_S", "180")),
                              lambda: (print(f"[bench] rank {rank}: GRAPH_REHEARSAL_HUNG (captured RCCL exchange did not finish); exit code 3",
                                             file=sys.stderr, flush=True), os._exit(3)))
        dog.daemon = True
        dog.start()
        try:
            pc.use_graph(True)
            r_ = P.KSP(max_it=200, **ksp_kw).solve(b, x.fill_(1.0), pc=pc)
            barrier(); t_ = time.perf_counter()
            run_iterations(args.steps)
            barrier(); w_ = time.perf_counter() - t_
            if rank == 0:
                print(f"[bench] graph rehearsal (cycle captured in a hipGraph, RCCL exchanges inside): {args.steps / w_:.2f} iterations/s, "
                      f"converged solve its={r_['its']} rnorm={r_['rnorm']:.3e} (measured line: {args.steps / wall:.2f} iterations/s)",
                      file=sys.stderr, flush=True)
        except Exception as e:      # noqa: BLE001 - a rehearsal
            print(f"[bench] rank {rank}: graph rehearsal failed: {e!r}", file=sys.stderr, flush=True)
        finally:
            dog.cancel()
            pc.use_graph(False)

    run_iterations(args.warmup)
    if args.profile_kernels:
        pc.profile(True)
        pc.profile_reset()
    comm0 = comm.stats() if dist_mode else None
    barrier()
    t0 = time.perf_counter()
    ms_dev = run_iterations(args.steps)
    barrier()
    wall = time.perf_counter() - t0
    if dist_mode:      # MAX over ranks
        tw = torch.tensor([wall], dtype=torch.float64, device="cpu" if staged else "cuda")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    stats = pc.profile_stats() if args.profile_kernels else []
    applies_timed = applies[0]
    comm1 = comm.stats() if dist_mode else None      # the traffic counters at the end of the timed region
    # the same steps with PETSc's solution update x += M (V y) (one more PCApply per solve), instrumented like the timed
    # region: for information (config.iterations_per_s_petsc_solution_update)
    petsc_form = None
    if P.lib().airhip_set_tuning(b"ksp_keep_preconditioned", 0) == 0:
        try:
            barrier(); t1 = time.perf_counter()
            run_iterations(args.steps)
            barrier(); w1 = time.perf_counter() - t1
            if dist_mode:
                tw1 = torch.tensor([w1], dtype=torch.float64, device="cpu" if staged else "cuda")
                dist.all_reduce(tw1, op=dist.ReduceOp.MAX)
                w1 = float(tw1.item())
            petsc_form = {"value": args.steps / w1, "pc_applies_per_step": applies[0] / args.steps}
        finally:
            P.lib().airhip_set_tuning(b"ksp_keep_preconditioned", TUNING_DEFAULTS["ksp_keep_preconditioned"])
    pc.profile(False)
    unprofiled = None
    if dist_mode and args.profile_kernels:
        # N > 1: the per-kernel events of the timed region (they are what `roofline` is made of) cost the short per-rank
        # kernels more than the 2 % they cost one GPU - the same iterations once more without them, for information
        barrier(); t1 = time.perf_counter()
        run_iterations(args.steps)
        barrier(); w1 = time.perf_counter() - t1
        tw1 = torch.tensor([w1], dtype=torch.float64, device="cpu" if staged else "cuda")
        dist.all_reduce(tw1, op=dist.ReduceOp.MAX)
        unprofiled = args.steps / float(tw1.item())
    if dist_mode:
        info["comm"] = comm1      # (the passes after the timed region are not part of its traffic)
        info["comm_timed"] = {k: comm1[k] - comm0[k] for k in comm0}      # the timed iterations only
        cab = torch.tensor([pc.cycle_alg_bytes()], dtype=torch.float64, device="cpu" if staged else "cuda")
        dist.all_reduce(cab)
        cx = dict(cx); cx["cycle_alg_bytes"] = float(cab.item())
        # every rank's own account of the timed iterations, for rank 0's line: kernel time, exposed exchange time, traffic
        live_r = [s for s in stats if s["calls"] > 0]
        mine = {"rank": rank,
                "vcycle_kernel_ms_per_step": sum(s["total_ms"] for s in live_r if s["level"] >= 0) / args.steps,
                "outer_and_krylov_kernel_ms_per_step": sum(s["total_ms"] for s in live_r if s["level"] == -1) / args.steps,
                "exposed_exchange_ms_per_step": sum(s["total_ms"] for s in live_r if s["kernel"] == "halo_exchange") / args.steps,
                "halo_exchanges_per_step": info["comm_timed"]["exchanges"] / args.steps,
                "halo_bytes_sent_per_step": info["comm_timed"]["bytes_sent"] / args.steps,
                "device_ms_per_step": ms_dev / args.steps, "phases": info.get("phases")}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine, group=meta)
        info["per_rank"] = per_rank
        if rank != 0:
            graph_rehearsal()
            dist.barrier()
            dist.destroy_process_group()
            return

    value = args.steps / wall
    roofline = None
    if stats:
        exch = [s for s in stats if s["calls"] > 0 and s["kernel"] == "halo_exchange"]      # distributed: exposed exchange time, not a kernel
        live = [s for s in stats if s["calls"] > 0 and s["kernel"] != "halo_exchange"]
        # aggregate the call sites by HIP kernel function: this is the row a rocprofv3
        # --kernel-trace --stats summary shows (profiles/), so the two can be compared directly
        kern = {}
        for s in live:
            k = kern.setdefault(s["kernel"], dict(calls=0, total_ms=0.0, bytes=0.0))
            k["calls"] += s["calls"]; k["total_ms"] += s["total_ms"]; k["bytes"] += s["alg_bytes"] * s["calls"]
        kname, dom = max(kern.items(), key=lambda kv: kv[1]["total_ms"])
        avg_ms = dom["total_ms"] / dom["calls"]
        achieved = dom["bytes"] / (dom["total_ms"] * 1e-3) / 1e9
        tot_ms = sum(s["total_ms"] for s in live)
        site = max((s for s in live if s["kernel"] == kname), key=lambda s: s["total_ms"])
        site_avg = site["total_ms"] / site["calls"]
        # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate
        # rocprofv3 --pmc passes of this same command, corrected as MI355X_MICROARCH.md prescribes;
        # recorded in profiles/pmc_traffic.json by tools/pmc_traffic.py) - null if not measured
        traffic = traffic_src = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                ent = json.load(f).get(f"{kname}@{M}")
            if ent and world == 1:
                # NOT measured in this run (PMC passes need rocprofv3 around the process): replayed from the committed
                # measurement of the same kernel at the same size
                traffic, traffic_src = ent["bytes_per_launch"], "replayed from profiles/pmc_traffic.json: " + ent["source"]
        except Exception:
            pass
        roofline = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                    "kernel": kname, "avg_launch_ms": avg_ms,
                    "alg_bytes_per_launch": dom["bytes"] / dom["calls"], "calls": dom["calls"],
                    "share_of_profiled_kernel_time": dom["total_ms"] / tot_ms,
                    "heaviest_call_site": {"site": f"{site['name']} (level {site['level'] + 1})", "avg_launch_ms": site_avg,
                                           "alg_bytes": site["alg_bytes"],
                                           "achieved": site["alg_bytes"] / (site_avg * 1e-3) / 1e9}}
        # the other profiled kernel functions of the cycle, the same way (k_rb_fc_first is a quarter of the cycle)
        others = {}
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pmc_all = json.load(f)
        except Exception:
            pmc_all = {}
        for kn, kk in kern.items():
            if kn == kname or kk["bytes"] <= 0:
                continue
            ach = kk["bytes"] / (kk["total_ms"] * 1e-3) / 1e9
            ent = pmc_all.get(f"{kn}@{M}") if world == 1 else None
            others[kn] = {"achieved": ach, "frac": ach / HBM_PEAK_GBS, "avg_launch_ms": kk["total_ms"] / kk["calls"], "calls": kk["calls"],
                          "alg_bytes_per_launch": kk["bytes"] / kk["calls"], "traffic": ent["bytes_per_launch"] if ent else None,
                          "share_of_profiled_kernel_time": kk["total_ms"] / tot_ms}
        roofline["other_kernels"] = others
        # whole V-cycle: algorithmic bytes of one cycle / measured kernel time of one cycle
        cycles = max(1, next((s["calls"] for s in live if s["name"].startswith("restrict") and s["level"] == 0), 1))
        vc_ms = sum(s["total_ms"] for s in live if s["level"] >= 0)     # level -1 = outer Krylov operator product
        # two denominators, both per cycle: SURVEY 8(d)'s B_cycle (counts the identity blocks of R / P and their vector
        # terms, which the layout never executes) and the S(X) of the operands the cycle's kernels actually executed
        roofline["vcycle_alg_GBs"] = cx["cycle_alg_bytes"] / (vc_ms / cycles * 1e-3) / 1e9
        roofline["vcycle_exec_GBs"] = sum(s["alg_bytes"] * s["calls"] for s in live if s["level"] >= 0) / (vc_ms * 1e-3) / 1e9
        roofline["vcycle_kernel_ms"] = vc_ms / cycles
        # the rest of a step (level -1 = the Krylov method's own kernels): outer product, Gram-Schmidt, norms, scaling
        outer_ms = sum(s["total_ms"] for s in live if s["level"] < 0 and s["name"].startswith("outer"))
        kry = [s for s in live if s["level"] == -1 and s["name"].startswith("Krylov ") and not s["name"].startswith("Krylov to ")]
        kry_ms = sum(s["total_ms"] for s in kry)
        roofline["outer_product_ms_per_step"] = outer_ms / args.steps
        roofline["krylov_kernels_ms_per_step"] = kry_ms / args.steps
        roofline["krylov_kernels_GBs"] = (sum(s["alg_bytes"] * s["calls"] for s in kry) / (kry_ms * 1e-3) / 1e9) if kry_ms > 0 else None
        # complete accounting: kernels timed inside the solves / device time of the solves (the rest: host round trips
        # of the dot products, launch gaps)
        roofline["profiled_share_of_device_ms"] = (vc_ms + outer_ms + kry_ms) / ms_dev if ms_dev > 0 else None
        if exch:
            roofline["exposed_exchange_ms_per_step"] = sum(s["total_ms"] for s in exch) / args.steps
        if args.verbose:
            for s in sorted(live, key=lambda s: -s["total_ms"])[:60]:
                a = s["total_ms"] / s["calls"]
                log(f"   {s['name']:28s} L{s['level']+1:<2d} calls={s['calls']:4d} avg={a:8.3f} ms  "
                    f"{s['alg_bytes']/(a*1e-3)/1e9:8.1f} GB/s")
    host_mirror_s = 0.0
    if not dist_mode and rank == 0:
        try:
            host_mirror_s = pc.host_mirror_wait()
        except Exception as e:      # noqa: BLE001 - reported, the solve results stand
            log(f"[bench] host mirror failed: {e}")
            host_mirror_s = None
    out = {
        "metric": "KSP iterations/sec (right-preconditioned GMRES(30) + PCAIR V-cycle)",
        "value": value, "unit": "iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic", "transport": transport,
        "config": {"workload": f"3D upwind advection (tests/adv_diff_fd.c) {M}^3 = {M ** 3} DoF, nnz(A)={nnzA}, b=0, x0=1, "
                               f"GMRES(30) right PC rtol 1e-10, PCAIR recipe '{args.recipe}' {opts}",
                   "parallelism": parallelism,
                   "layout_hint": ("morton" if args.hint_brick == -1 else f"{args.hint_brick}^3 bricks" if args.hint_brick > 0 and not args.brick else "none"),
                   "levels": info["levels"], "iterations_to_rtol_1e-10": res0["its"],
                   "converged_reason": res0["reason_name"], "converged_solve_ms": res0["solve_ms"],
                   "cycle_complexity": cx["cycle"], "operator_complexity": cx["operator"],
                   "storage_complexity": cx["storage"], "cycle_alg_bytes": cx["cycle_alg_bytes"],
                   "setup_s": t_setup, "host_setup_s": st["host_setup_s"], "upload_s": st["upload_s"],
                   # the library's thread that completes the HOST copy of the hierarchy after setup() has returned
                   # (the solve does not read it; runs beside the warm-up solves): how long it ran
                   "host_mirror_s": host_mirror_s,
                   "device_ms_of_timed_solves": ms_dev, "hip_graph": bool(args.graph),
                   # V-cycles per step of the timed region (1.0: the preconditioned basis is kept, x += Z y) and the same
                   # steps with PETSc's solution update x += M (V y)
                   "pc_applies_per_step": applies_timed / args.steps,
                   "solution_update": "x += Z y with the preconditioned basis z_j = M v_j kept (same iterates as PETSc's "
                                      "x += M (V y), one PCApply less per solve; airhip_set_tuning ksp_keep_preconditioned)",
                   "iterations_per_s_petsc_solution_update": petsc_form["value"] if petsc_form else None,
                   "pc_applies_per_step_petsc_solution_update": petsc_form["pc_applies_per_step"] if petsc_form else None},
        "roofline": roofline, "cpu_baseline": cpu,
    }
    # the residual history of the converged solve against the committed single-GPU history (same hierarchy for every N: the
    # setup is bit-reproducible; differences are the summation orders of the distributed dot products and halo'ed rows)
    try:
        with open(os.path.join(ROOT, "tests", "golden", "bench_history_n1.json")) as f:
            fx = json.load(f).get(str(M))
        if fx and args.recipe == fx["recipe"]:
            h = res0["history"]; k = min(len(h), len(fx["history"]))
            out["config"]["history_check"] = {
                "fixture": "tests/golden/bench_history_n1.json", "entries": k, "iterations_fixture": fx["iterations"],
                "max_rel_diff": max(abs(h[i] - fx["history"][i]) / max(abs(fx["history"][i]), 1e-300) for i in range(k)) if k else None}
    except (OSError, ValueError, KeyError):
        pass
    if args.verbose:
        log(f"[bench] residual history of the converged solve: {json.dumps(res0['history'])}")
    if dist_mode:
        out["config"]["phases_s"] = {"setup_rank0": t_setup, **{f"{k}_rank0": v for k, v in (info.get("phases") or {}).items()}}
        out["config"]["per_rank"] = info.get("per_rank")
        # the model of tools/dist_model.py on this run's own counts (constants NOT measured: 15 us per send / receive group,
        # 50 GB/s per peer): kernels + exchanges one after the other, i.e. without overlap - an upper bound to compare with
        try:
            pr = info["per_rank"]
            kmax = max(r_["vcycle_kernel_ms_per_step"] + r_["outer_and_krylov_kernel_ms_per_step"] for r_ in pr)
            emax = max(r_["halo_exchanges_per_step"] * 0.015 + r_["halo_bytes_sent_per_step"] / 50e9 * 1e3 for r_ in pr)
            out["config"]["model"] = {"kernels_ms_per_step": kmax, "exchanges_ms_per_step_unhidden": emax,
                                      "predicted_ms_per_step_without_overlap": kmax + emax,
                                      "assumptions": "15 us per exchange + bytes / 50 GB/s (tools/dist_model.py); not measured"}
        except Exception:      # noqa: BLE001 - diagnostic only
            pass
        out["config"]["distribute_s"] = info["t_distribute"]
        out["config"]["balance"] = info.get("balance")
        if unprofiled is not None:
            out["config"]["iterations_per_s_without_kernel_events"] = unprofiled
        out["config"]["halo_exchanges_rank0"] = info["comm"]["exchanges"]
        out["config"]["allreduces_rank0"] = info["comm"]["allreduces"]
        ct = info["comm_timed"]
        out["config"]["per_iteration_rank0"] = {"halo_exchanges": ct["exchanges"] / args.steps, "halo_messages_sent": ct["messages_sent"] / args.steps,
                                                "halo_bytes_sent": ct["bytes_sent"] / args.steps, "allreduces": ct["allreduces"] / args.steps}
        if out["roofline"] is not None:
            out["roofline"]["note"] = "kernel timings of rank 0"
    print(json.dumps(out), flush=True)
    if dist_mode:
        graph_rehearsal()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
