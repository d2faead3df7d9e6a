#!/usr/bin/env python3
"""V-cycle throughput of the MI355X AMG solve path on BASELINE.json's metric
configuration: 3D Poisson 500^3 (125 M dof, 7-pt stencil) CSR fp64,
smoothed-aggregation hierarchy, Chebyshev(2) pre/post smoother, V(1,1) cycles.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--grid 500] [--smoother chebyshev]

--config selects the BASELINE.json configuration (default C3 = the metric's: 500^3, SA, Chebyshev(2)):
  C1  2D Poisson 500^2, ruge_stuben_solver, symmetric Gauss-Seidel (the README example)
  C2  2D Poisson 2000^2, SA, weighted Jacobi (omega = 4/3)
  C4  3D Poisson 500^3, SA, HYBRID Gauss-Seidel: --smoother hybrid_gs (default: multicolour inside a rank, Jacobi
      across ranks) or hybrid_gs_lex (lexicographic inside a rank); with --gpus N the rows are partitioned
  C5  anisotropic diffusion, P1 on a jittered Kuhn tetrahedral mesh, --grid 360 -> 46.7 M unknowns (the largest cube
      whose BSR 3x3 data array stays below 2^31 values: int32 addressing, as in the reference), block-SA, symmetric
      block Gauss-Seidel
Every line carries `roofline` (dominant kernel, bytes / live HIP-event time) and `cpu_baseline` (the C oracle timed
on this box's host cores, with the CPU model) and checks the first iterate against the oracle at the full size.

One "step" = one multilevel_solver.solve() iteration: a V-cycle + the residual
norm (pyamg/multilevel.py:454-461).  The hierarchy is resident in HBM before the
timed region (b and x are device vectors).  Prints ONE JSON line (rank 0).

N > 1: one process per GPU.  `python bench.py --gpus N` starts the N ranks itself (fresh child processes; the parent
never touches the GPU); under `torch.distributed.run` the ranks come from the launcher (RANK / WORLD_SIZE).  The SAME
500^3 problem is row-partitioned over the N ranks on every level: the whole partitioned cycle runs in
libamgcore_hip.so (hier.hip + comm.hip) -- halos pushed GPU-to-GPU through IPC-mapped arenas over xGMI ("peer"), or
grouped ncclSend/ncclRecv + ncclAllReduce ("rccl", AMG_DIST_TRANSPORT), one 8-byte all-reduce per step for the residual
norm -> "scaling": "strong" (total work fixed).  Rank 0 builds the hierarchy once and ships it to the other ranks
through /dev/shm.  `--replicas` runs N independent copies instead (no exchange).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec
CACHE_LOAD_SECONDS = None   # --cache hit: the hierarchy was loaded, not set up, in this run


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def setup_seconds(t_gen, t_setup):
    d = {"matrix": round(t_gen, 1), "hierarchy": None if t_setup is None else round(t_setup, 1)}
    if CACHE_LOAD_SECONDS is not None:
        d["hierarchy_loaded_from_cache"] = True
        d["load_seconds"] = round(CACHE_LOAD_SECONDS, 1)
    return d


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_hierarchy(grid, smoother, cache=None):
    import pyamg_amd
    from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
    cdir = os.path.join(cache, "poisson%d_%s" % (grid, smoother)) if cache else None
    if cdir and os.path.exists(os.path.join(cdir, "meta.json")):
        # --cache DIR: the hierarchy of an earlier run (multilevel_solver.save), memory-mapped -- same operators in
        # the same stored order, same smoother constants, hence the same iterates
        t0 = time.time()
        ml = pyamg_amd.multilevel_solver.load(cdir, mmap=True)
        log("[bench] hierarchy loaded from %s in %.1fs" % (cdir, time.time() - t0))
        log(repr(ml))
        global CACHE_LOAD_SECONDS
        CACHE_LOAD_SECONDS = time.time() - t0
        return ml, (0.0, None)
    # load the HIP library and create the device context first: ~1 s of runtime start-up that is not hierarchy setup
    from pyamg_amd import _lib as _amg_lib
    if _amg_lib.device_count() > 0:
        _amg_lib.lib().amg_dev_free(_amg_lib.lib().amg_dev_alloc(1))
    t0 = time.time()
    A = poisson((grid, grid, grid))
    t1 = time.time()
    np.random.seed(0)
    if smoother == "chebyshev":
        sm = ("chebyshev", {"degree": 2})
    elif smoother == "jacobi":
        sm = ("jacobi", {"omega": 4.0 / 3.0})
    elif smoother in ("gauss_seidel", "hybrid_gs_lex"):
        sm = ("gauss_seidel", {"sweep": "symmetric"})
    elif smoother == "hybrid_gs":
        # BASELINE configuration C4: multicolour Gauss-Seidel inside a rank (greedy colouring, relaxed with
        # gauss_seidel_indexed semantics, relaxation.h:395-430), Jacobi across ranks (halo frozen per directional sweep)
        sm = ("multicolor_gauss_seidel", {"sweep": "symmetric"})
    else:
        raise ValueError(smoother)
    ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
    t2 = time.time()
    log("[bench] poisson %.1fs, SA setup %.1fs" % (t1 - t0, t2 - t1))
    log(repr(ml))
    if cdir:
        t3 = time.time()
        ml.save(cdir)
        log("[bench] hierarchy saved to %s in %.1fs" % (cdir, time.time() - t3))
    return ml, (t1 - t0, t2 - t1)


def oracle_hierarchy(ml):
    """The CPU oracle (oracle/amg_oracle.c) on the same host arrays -- the timed CPU baseline."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    levels = []
    for lvl in ml.levels:
        L = {"A": lvl.A}
        if hasattr(lvl, "P"):
            L["P"], L["R"] = lvl.P, lvl.R
            L["pre"] = dict(lvl.presmoother.desc)
            L["post"] = dict(lvl.postsmoother.desc)
        levels.append(L)
    kind, M = ml.coarse_solver.device_form(ml.levels[-1].A)
    return oracle_lib.Hierarchy(levels, M if kind == "dense" else None, dup_prolong=True)


def cpu_baseline_of(ml, b, gpu_first):
    """One step (V-cycle + residual norm, from x0 = 0) of the same hierarchy and right-hand side with the C oracle on the
    host cores: the row-parallel loops over the cores this process may use, then the scalar port.  gpu_first: None, or
    a callable returning (iterate after one GPU cycle from x0 = 0, its residual norm) for the full-size parity check."""
    H = oracle_hierarchy(ml)
    A0c = ml.levels[0].A
    n = A0c.shape[0]
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    bsr = getattr(A0c, "blocksize", (1, 1)) != (1, 1) and A0c.format == "bsr"
    A0p, A0j = np.ascontiguousarray(A0c.indptr, dtype=np.intc), np.ascontiguousarray(A0c.indices, dtype=np.intc)
    A0x = np.ascontiguousarray(np.ravel(A0c.data))

    def cpu_step(threads):
        H.lib.oracle_set_threads(threads)
        xo = np.zeros(n)
        t0 = time.perf_counter()
        H.cycle(xo, b, "V")
        # + the residual norm that closes the step (multilevel.py:461)
        Ah = np.zeros(n)
        if bsr:
            R, Cc = A0c.blocksize
            H.lib.oracle_bsr_matvec(n // R, R, Cc, ip(A0p), ip(A0j), dp(A0x), dp(xo), dp(Ah))
        else:
            H.lib.oracle_csr_matvec(n, ip(A0p), ip(A0j), dp(A0x), dp(xo), dp(Ah))
        rn = H.lib.oracle_norm2(dp(b - Ah), n)
        return time.perf_counter() - t0, rn, xo

    cores = max(1, min(len(os.sched_getaffinity(0)), 16))      # a 1-GPU box's share of host cores
    t_par, rn_par, xo_par = cpu_step(cores)
    t_cpu, rn, xo = cpu_step(1)
    H.lib.oracle_set_threads(1)
    log("[bench] CPU oracle: 1 step in %.2fs on 1 thread, %.2fs on %d threads, residual %.6e" % (t_cpu, t_par, cores, rn))
    cpu = {"value": round(1.0 / t_par, 5), "unit": "V-cycle iterations/s", "cores": cores, "kind": "port",
           "cpu_model": cpu_model(), "host_cores_visible": len(os.sched_getaffinity(0)),
           "sample": "1 V-cycle + residual norm of the same hierarchy and RHS from x0=0 with the C oracle "
                     "(oracle/amg_oracle.c, -O3, row-parallel OpenMP loops: SpMV, Jacobi, Chebyshev and vector "
                     "updates -- Gauss-Seidel sweeps are sequential as in the reference; includes the reference's "
                     "discarded second P*coarse_x per level, multilevel.py:548)",
           "seconds": round(t_par, 3),
           "single_thread": {"value": round(1.0 / t_cpu, 5), "seconds": round(t_cpu, 3)},
           "thread_count_independent": bool(np.array_equal(xo, xo_par) and rn == rn_par)}
    if gpu_first is not None:
        xg, rg = gpu_first()
        cpu["first_step_iterate_bit_identical_to_gpu"] = bool(np.array_equal(xg, xo))
        cpu["first_step_residual_equals_gpu"] = bool(abs(rn - rg) <= 1e-12 * abs(rn))
        log("[bench] first iterate at full size bit-identical to the oracle's: %s" % cpu["first_step_iterate_bit_identical_to_gpu"])
    return cpu


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (this parent never touches the
    GPU and is never replaced), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    out, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    for ln in (out.decode() if out else "").splitlines():
        if ln.startswith("{"):               # rank 0's result line (anything else a library printed goes to stderr)
            sys.stdout.write(ln + "\n")
        elif ln.strip():
            sys.stderr.write(ln + "\n")
    sys.stdout.flush()
    if any(rcs):
        raise SystemExit("rank exit codes: %s" % rcs)


def partitioned_main(args, rank, local_rank, world, torch, dist):
    """strong scaling: one problem, rows of every level split over the ranks"""
    import shutil
    from pyamg_amd.distributed import (DistributedSolver, HipBackend, levels_from_ml, load_levels, save_levels,
                                       split_rows)
    host_group = dist.group.WORLD
    n = args.grid ** 3
    # where to ship the hierarchy: a memory-backed directory with room for ~0.32 KB per fine unknown
    need = int(330.0 * n) + (1 << 28)
    shared = None
    if rank == 0:
        for base in ("/dev/shm", os.environ.get("TMPDIR", "/tmp"), "/tmp", ROOT):
            try:
                if shutil.disk_usage(base).free > need:
                    shared = os.path.join(base, "amg_bench_%s" % os.environ.get("MASTER_PORT", "0"))
                    break
            except OSError:
                pass
        if shared is None:
            shared = os.path.join("/tmp", "amg_bench_%s" % os.environ.get("MASTER_PORT", "0"))
    box = [shared]
    dist.broadcast_object_list(box, src=0, group=host_group)
    shared = box[0]
    t_gen = t_setup = 0.0
    cpu = None
    if rank == 0:
        shutil.rmtree(shared, ignore_errors=True)
        ml, (t_gen, t_setup) = build_hierarchy(args.grid, args.smoother, args.cache)
        levels, coarse = levels_from_ml(ml)
        t0 = time.time()
        save_levels(shared, levels, coarse)
        np.random.seed(0)
        bfull = np.random.rand(n)
        np.save(os.path.join(shared, "b.npy"), bfull)
        log("[bench] hierarchy shipped to %s in %.1fs" % (shared, time.time() - t0))
        shape_info = [[int(L["A"].shape[0]), int(L["A"].nnz)] for L in levels]
        # (the CPU baseline is timed at N = 1 only: with N > 1 the other ranks would sit idle through 10+ s of oracle;
        #  --cpu-baseline-anyway asks for it)
        if args.cpu_baseline_anyway and not args.no_cpu_baseline:
            cpu = cpu_baseline_of(ml, bfull, None)
        del ml, levels, bfull
    dist.barrier(group=host_group)
    levels, coarse = load_levels(shared)
    t0 = time.time()
    bnd = split_rows(n, world)
    lo, hi = int(bnd[rank]), int(bnd[rank + 1])
    b = np.ascontiguousarray(np.load(os.path.join(shared, "b.npy"), mmap_mode="r")[lo:hi])
    # transport: the C++ engine with GPU-to-GPU pushes ("peer"), else the C++ engine on RCCL, else the Python-driven
    # cycle on torch.distributed; a transport is taken only if EVERY rank could set it up and one cycle ran clean
    order = [os.environ["AMG_DIST_TRANSPORT"]] if os.environ.get("AMG_DIST_TRANSPORT") else ["peer", "rccl", "python"]
    S, transport = None, None
    for cand in order:
        os.environ["AMG_DIST_TRANSPORT"] = cand
        ok = 1.0
        try:
            S = DistributedSolver(levels, coarse, HipBackend(local_rank), rank, world, group=host_group, host_group=host_group)
            S.solve(b, None, tol=0.0, maxiter=1, cycle="V", fixed=True)
        except Exception as e:   # noqa: BLE001
            log("[bench] rank %d: transport %s failed: %r" % (rank, cand, e))
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float64)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=host_group)
        if flag.item() > 0.5:
            transport = cand
            break
        # some rank could not use this transport: every rank tears down what IT built without barriers (the ranks whose
        # constructor failed have nothing to close), then all of them meet once before the next candidate
        if S is not None and ok > 0.5:
            S.close(collective=False)
        S = None
        dist.barrier(group=host_group)
    if S is None:
        raise SystemExit("no working inter-GPU transport")
    log("[bench] rank %d: rows %d..%d, halos per level %s, transport %s, partition+upload %.1fs" %
        (rank, lo, hi, [lv.n_halo for lv in S.lv], transport, time.time() - t0))
    dist.barrier(group=host_group)
    if rank == 0:
        shutil.rmtree(shared, ignore_errors=True)

    def sync_all():
        torch.cuda.synchronize(); dist.barrier(group=host_group); torch.cuda.synchronize()

    # warm-up from x0 = 0, then EXACTLY K timed steps continuing from the warm iterate (vectors resident in HBM)
    if S.native is not None:
        from pyamg_amd import _lib
        Lb, h, comm = S.native
        res = np.zeros(max(args.steps, args.warmup) + 2)
        nres = C.c_int(0)
        NO_EARLY_STOP, DEVICE_VECTORS, X0_ZERO = 2, 4, 1
        xw = np.zeros(hi - lo)
        _lib.check(Lb.amg_hier_solve(h, b.ctypes.data, xw.ctypes.data, 0.0, args.warmup, 0, _lib.dp(res), C.byref(nres),
                                     NO_EARLY_STOP | X0_ZERO))
        warm = res[:nres.value].copy()
        db, dx = Lb.amg_hier_dev_b(h), Lb.amg_hier_dev_x(h)
        sync_all()
        t0 = time.perf_counter()
        _lib.check(Lb.amg_hier_solve(h, db, dx, 0.0, args.steps, 0, _lib.dp(res), C.byref(nres), NO_EARLY_STOP | DEVICE_VECTORS))
        sync_all()
        wall = time.perf_counter() - t0
        _lib.check(Lb.amg_hier_comm_check(h))
        timed = res[:nres.value].copy()
        r0 = float(warm[0])
        ms_resid = C.c_double(0.0)
        _lib.check(Lb.amg_hier_time_spmv(h, 0, 0, 1, 20, C.byref(ms_resid)))
        ms_resid = ms_resid.value
        form = Lb.amg_hier_operator_form(h, 0)
        moved = Lb.amg_hier_operator_bytes(h, 0, 1)
        coded0 = Lb.amg_hier_value_index(h, 0, -1)
        overl = "interior rows overlap the exchange" if os.environ.get("AMG_DIST_OVERLAP", "1") != "0" else "plain exchange"
    else:
        S.set_problem(b, None)
        r0 = S.residual_norm()
        warm = [r0] + S.run_fixed(args.warmup, "V", x_zero=True)
        sync_all()
        t0 = time.perf_counter()
        timed = S.run_fixed(args.steps, "V", x_zero=False)
        sync_all()
        wall = time.perf_counter() - t0
        lv = S.lv[0]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        S.be.apply(lv.A, 2, lv.x, lv.b, None, lv.r, None, 0.0)
        e0.record()
        for _ in range(20):
            S.be.apply(lv.A, 2, lv.x, lv.b, None, lv.r, None, 0.0)
        e1.record(); torch.cuda.synchronize()
        ms_resid = e0.elapsed_time(e1) / 20
        form = S.operator_form(0)
        moved = None
        coded0 = 0
        overl = "python driver"
    tw = torch.tensor([wall], dtype=torch.float64)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX, group=host_group)
    wall = float(tw.item())
    if rank == 0:
        lv = S.lv[0]
        n_own = lv.n_own
        spmv_bytes = 12.0 * lv.nnzA + 4.0 * (n_own + 1) + 24.0 * n_own
        if moved is None:
            moved = spmv_bytes
        ach = moved / (ms_resid * 1e-3) / 1e9
        out = {
            "metric": "V-cycle iterations/sec (3D Poisson %d^3 fp64, SA-AMG, %s smoother)" % (args.grid, args.smoother),
            "value": round(args.steps / wall, 4), "unit": "V-cycle iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D Poisson %dx%dx%d (%.1fM dof) CSR fp64, smoothed aggregation (%d levels), "
                                   "%s pre/post smoother, V(1,1), b=rand seed 0" %
                                   (args.grid, args.grid, args.grid, n / 1e6, len(S.lv), args.smoother),
                       "parallelism": "rows of every level partitioned over %d ranks on %d GPU(s); transport %s (%s); "
                                      "one 8-byte all-reduce per residual norm" %
                                      (world, min(world, torch.cuda.device_count()), transport, overl),
                       "transport": transport,
                       "levels": shape_info, "halo_per_level_rank0": [lv_.n_halo for lv_ in S.lv],
                       "replicated_from_level": S.first_rep,
                       "setup_seconds": setup_seconds(t_gen, t_setup),
                       "residuals": [r0, float(warm[-1]), float(timed[-1])]},
            "roofline": {"bound": "hbm",
                         "kernel": {0: "csr_stream_kernel", 1: "csr_pattern_kernel", 2: "stencil_coded_kernel (one-byte value codes)" if coded0 else "stencil2_kernel"}.get(form, "csr_stream_kernel") +
                                   " (level-0 A-application on rank 0's row block)",
                         "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                         "traffic_source": "not measured in this run (PMC passes need rocprofv3 around a single rank)",
                         "bytes_per_launch": moved, "ms_per_launch": round(ms_resid, 4),
                         "csr_equivalent_bytes_per_launch": spmv_bytes,
                         "csr_equivalent_GBs": round(spmv_bytes / (ms_resid * 1e-3) / 1e9, 1)},
            "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    S.close()
    dist.barrier(group=host_group)
    dist.destroy_process_group()


def config_main(args, rank, local_rank, world, torch, dist, L):
    """BASELINE configurations C1, C2, C5 (and C4 on one GPU): one process per GPU, N > 1 = N independent replicas."""
    import pyamg_amd
    from pyamg_amd import _lib
    from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
    cfg, g = args.config, args.grid
    if _lib.device_count() > 0:
        L.amg_dev_free(L.amg_dev_alloc(1))        # runtime start-up is not setup time
    t0 = time.time()
    if cfg == "C1":
        from pyamg_amd.classical import ruge_stuben_solver
        A = poisson((g, g)); t1 = time.time()
        ml = ruge_stuben_solver(A)
        kind, sweeps = "gs", 2
        what = "2D Poisson %dx%d (%.2fM dof) CSR fp64, ruge_stuben_solver, symmetric Gauss-Seidel pre/post (exact sequential sweeps)" % (g, g, A.shape[0] / 1e6)
        kname = "gs_chain2_kernel (level-0 symmetric Gauss-Seidel application: 2 directional sweeps, chained dependency levels, LDS hand-off)"
    elif cfg == "C2":
        A = poisson((g, g)); t1 = time.time()
        np.random.seed(0)
        sm = ("jacobi", {"omega": 4.0 / 3.0})
        ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
        kind, sweeps = "jacobi", 1
        what = "2D Poisson %dx%d (%.1fM dof) CSR fp64, smoothed aggregation, weighted Jacobi (omega = 4/3 / rho) pre/post" % (g, g, A.shape[0] / 1e6)
        kname = None
    elif cfg == "C4":
        A = poisson((g, g, g)); t1 = time.time()
        np.random.seed(0)
        lex = args.smoother in ("hybrid_gs_lex", "gauss_seidel")
        sm = ("gauss_seidel", {"sweep": "symmetric"}) if lex else ("multicolor_gauss_seidel", {"sweep": "symmetric"})
        ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
        kind, sweeps = "gs", 2
        what = ("3D Poisson %dx%dx%d (%.1fM dof) CSR fp64, smoothed aggregation, hybrid Gauss-Seidel pre/post (%s inside the rank; one rank: no "
                "frozen halo)" % (g, g, g, A.shape[0] / 1e6, "lexicographic" if lex else "multicolour, gauss_seidel_indexed semantics"))
        kname = ("gs_flow_kernel (level-0 symmetric Gauss-Seidel application as one persistent dataflow launch)" if lex else
                 "csr_stream_kernel<GS> (level-0 multicolour Gauss-Seidel application: 2 directional sweeps, one launch per colour)")
    elif cfg == "C5":
        from pyamg_amd.gallery import tet_diffusion
        A = tet_diffusion(g, blocksize=3); t1 = time.time()
        np.random.seed(0)
        sm = ("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3})
        ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
        kind, sweeps = "block_gs", 2
        what = ("anisotropic diffusion, P1 on a jittered Kuhn tetrahedral mesh, %d^3 = %.1fM unknowns, BSR 3x3 (%.1fM blocks), "
                "block smoothed aggregation, symmetric block Gauss-Seidel pre/post" % (g, A.shape[0] / 1e6, len(A.indices) / 1e6))
        kname = "bgs_flow_kernel<3, 5, 3> (level-0 symmetric block Gauss-Seidel application: 2 directional sweeps as one persistent dataflow launch)"
    else:
        raise ValueError(cfg)
    t2 = time.time()
    t_matrix = t1 - t0
    log("[bench] %s: matrix %.1fs, setup %.1fs" % (cfg, t_matrix, t2 - t1))
    log(repr(ml))
    ml.device = local_rank
    n = A.shape[0]
    dev = ml.device_hierarchy()
    log("[bench] upload %.1fs, %.1f GB in HBM" % (time.time() - t2, dev.device_bytes() / 1e9))
    h = dev.h
    np.random.seed(0)
    b = np.random.rand(n)
    x = np.zeros(n)
    res = np.zeros(max(args.steps, args.warmup) + 2)
    nres = C.c_int(0)
    NO_EARLY_STOP, DEVICE_VECTORS, X0_ZERO = 2, 4, 1
    # warm-up: at least 12 steps for the sub-millisecond configurations -- an iteration is captured into a hipGraph the
    # second time its buffer state is seen (Jacobi swaps x / xalt level by level: several states), and a capture +
    # instantiation inside a 20 ms timed region would be most of it; the count actually run is what the line reports
    warmup = max(args.warmup, 12) if cfg in ("C1", "C2") else args.warmup
    res = np.zeros(max(args.steps, warmup) + 2)
    _lib.check(L.amg_hier_solve(h, b.ctypes.data, x.ctypes.data, 0.0, warmup, 0, _lib.dp(res), C.byref(nres), NO_EARLY_STOP | X0_ZERO))
    warm_res = res[:nres.value].copy()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    db, dx = L.amg_hier_dev_b(h), L.amg_hier_dev_x(h)
    if cfg in ("C1", "C2"):
        # ... and the same number of steps through the entry the timed region uses (vectors resident): its first call may
        # capture again, and one capture is a multiple of a 20 ms region (a C2 line read 1.04 ms per step against 0.40 ms
        # of device time when it did)
        _lib.check(L.amg_hier_solve(h, db, dx, 0.0, warmup, 0, _lib.dp(res), C.byref(nres), NO_EARLY_STOP | DEVICE_VECTORS))
        warmup *= 2
    sync_all()
    t0 = time.perf_counter()
    _lib.check(L.amg_hier_solve(h, db, dx, 0.0, args.steps, 0, _lib.dp(res), C.byref(nres), NO_EARLY_STOP | DEVICE_VECTORS))
    sync_all()
    wall = time.perf_counter() - t0
    ev_ms = L.amg_hier_last_solve_ms(h)
    timed_res = res[:nres.value].copy()
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    if rank == 0:
        A0 = ml.levels[0].A
        if kind == "block_gs":
            bs = A0.blocksize[0]
            nb, nblk = n // bs, len(A0.indices)
            per_sweep = 8.0 * nblk * bs * bs + 4.0 * nblk + 4.0 * (nb + 1) + 8.0 * n * bs + 3 * 8.0 * n     # SURVEY 8(d), BSR + Dinv
        else:
            per_sweep = 12.0 * A0.nnz + 4.0 * (n + 1) + 8.0 * n + 8.0 * n + 8.0 * n                          # bytes_spmv(A_0) + 8 n
        alg = sweeps * per_sweep
        ms_relax = dev.time_relax(0, 0, reps=5)           # HIP events on the hierarchy's stream, inside the library
        ms_resid = dev.time_spmv(0, 0, mode=1, reps=10)
        form = L.amg_hier_operator_form(h, 0)
        if kname is None:
            coded = L.amg_hier_value_index(h, 0, -1)
            kname = {0: "csr_stream_kernel<JACOBI>", 1: "csr_pattern_kernel<JACOBI>",
                     2: "stencil_coded_kernel<JACOBI>" if coded else "stencil2_kernel<JACOBI>"}[form] + " (level-0 weighted-Jacobi sweep%s)" % (
                         ", one-byte value codes: %d distinct values" % coded if coded else "")
        # `achieved` prices the launch at the bytes the kernel's storage form streams where that is fewer than the CSR
        # bytes of SURVEY 8(d) (a Jacobi sweep from the stencil form: the same operands as r = b - A x); the CSR-priced
        # figure stays beside it.  The Gauss-Seidel kernels stream level-ordered CSR / BSR copies: algorithmic = moved.
        priced, bytes_are = alg, "algorithmic (SURVEY 8d): %d directional sweep(s) of the level-0 smoother" % sweeps
        if kind == "jacobi" and form != 0:
            priced = float(L.amg_hier_operator_bytes(h, 0, 1))
            bytes_are = "bytes this kernel's storage form streams for one sweep (matrix copy + x, b, out)"
        ach = priced / (ms_relax * 1e-3) / 1e9
        cycle_bytes = dev.cycle_bytes("V")
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None,
                    "traffic_source": "not collected for this configuration",
                    "bytes_per_launch": priced, "bytes_are": bytes_are,
                    "csr_equivalent_bytes_per_launch": alg, "csr_equivalent_GBs": round(alg / (ms_relax * 1e-3) / 1e9, 1),
                    "ms_per_launch": round(ms_relax, 4),
                    "level0_residual_ms": round(ms_resid, 4),
                    "cycle_bytes": cycle_bytes,
                    "cycle_achieved_GBs": round(cycle_bytes * (args.steps / (ev_ms * 1e-3)) / 1e9, 1)}
        if kind != "block_gs":
            moved = L.amg_hier_operator_bytes(h, 0, 1)
            roofline["level0_residual_bytes_moved"] = moved
            roofline["level0_residual_moved_GBs"] = round(moved / (ms_resid * 1e-3) / 1e9, 1)
        cpu = None
        if not args.no_cpu_baseline:
            def gpu_first():
                xg = np.zeros(n)
                r1 = np.zeros(3); n1 = C.c_int(0)
                _lib.check(L.amg_hier_solve(h, b.ctypes.data, xg.ctypes.data, 0.0, 1, 0, _lib.dp(r1), C.byref(n1), NO_EARLY_STOP | X0_ZERO))
                return xg, float(r1[1])
            cpu = cpu_baseline_of(ml, b, gpu_first)
        out = {
            "metric": "V-cycle iterations/sec (BASELINE configuration %s)" % cfg,
            "value": round(world * args.steps / wall, 4), "unit": "V-cycle iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": warmup, "ms_per_step": round(1e3 * wall / args.steps, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s, %d levels, V(1,1), b=rand seed 0" % (cfg, what, len(ml.levels)),
                       "baseline_config": cfg,
                       "parallelism": "single GPU" if world == 1 else "%d replicas (no exchange)" % world,
                       "levels": [[int(l.A.shape[0]), int(l.A.nnz)] for l in ml.levels],
                       "setup_seconds": {"matrix": round(t_matrix, 1), "hierarchy": round(t2 - t1, 1)},
                       "hbm_resident_GB": round(dev.device_bytes() / 1e9, 2),
                       "device_event_ms_per_step": round(ev_ms / args.steps, 4),
                       "residuals": [float(warm_res[0]), float(warm_res[-1]), float(timed_res[-1])]},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)    # default 50: ~0.6 s timed at 500^3 (20 steps = 0.37 s was short);
                                                          # C1 200, C2 500: their steps are 6 / 0.4 ms, and a 20 ms region reads a single
                                                          # host-side stall of the same length as half the rate (seen on C2: 0.41 / 0.88 ms)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3", choices=["C1", "C2", "C3", "C4", "C5"], help="BASELINE.json configuration (default: the metric's)")
    ap.add_argument("--grid", type=int, default=None, help="grid points per axis (defaults: C1 500, C2 2000, C3 / C4 500, C5 360)")
    ap.add_argument("--smoother", default=None, help="C3: chebyshev (default) | jacobi | gauss_seidel; C4: hybrid_gs (default) | hybrid_gs_lex")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-anyway", action="store_true", help="N > 1: time the CPU oracle on rank 0 too (default: N = 1 only)")
    ap.add_argument("--no-value-index", action="store_true", help="skip the extra measurement of the step with 8-byte level-0 values (value index off)")
    ap.add_argument("--variant", type=int, default=None, help="CSR stream kernel load variant (0/1)")
    ap.add_argument("--xcd-chunk", type=int, default=None)
    ap.add_argument("--tile-target", type=int, default=None)
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of partitioning")
    ap.add_argument("--cache", default=None, help="directory of saved hierarchies: skip the setup on reruns")
    args = ap.parse_args()
    if args.grid is None:
        args.grid = {"C1": 500, "C2": 2000, "C3": 500, "C4": 500, "C5": 360}[args.config]
    if args.smoother is None:
        args.smoother = "hybrid_gs" if args.config == "C4" else "chebyshev"
    if args.steps is None:
        args.steps = {"C1": 200, "C2": 500}.get(args.config, 50)

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)           # before anything touches the GPU
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if torch.cuda.device_count() == 0:
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        ndev = max(torch.cuda.device_count(), 1)
        if world > ndev:
            log("[bench] %d ranks on %d GPU(s): ranks share devices (rehearsal)" % (world, ndev))
        local_rank = local_rank % ndev
        torch.cuda.set_device(local_rank)
        # the process group only carries setup metadata and barriers (host tensors); the halos and the all-reduce
        # of the solve travel GPU-to-GPU from the C++ engine (peer arenas over xGMI, or RCCL)
        # (gloo announces its connections on STDOUT from C++: keep the one-JSON-line contract by pointing fd 1 at
        # stderr while the group forms)
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    import pyamg_amd
    from pyamg_amd import _lib
    L = _lib.lib()
    if pyamg_amd.device_count() <= local_rank:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.variant is not None:
        L.amg_set_stream_variant(args.variant)
    if args.xcd_chunk is not None:
        L.amg_set_xcd_chunk(args.xcd_chunk)
    if args.tile_target is not None:
        L.amg_set_tile_target(args.tile_target)

    if args.config in ("C1", "C2", "C5") or (args.config == "C4" and world == 1):
        return config_main(args, rank, local_rank, world, torch, dist, L)
    if world > 1 and not args.replicas:
        return partitioned_main(args, rank, local_rank, world, torch, dist)

    ml, (t_gen, t_setup) = build_hierarchy(args.grid, args.smoother, args.cache)
    ml.device = local_rank
    n = ml.levels[0].A.shape[0]
    t0 = time.time()
    dev = ml.device_hierarchy()
    log("[bench] upload %.1fs, %.1f GB in HBM" % (time.time() - t0, dev.device_bytes() / 1e9))
    h = dev.h

    np.random.seed(0)
    b = np.random.rand(n)
    x = np.zeros(n)
    res = np.zeros(max(args.steps, args.warmup) + 2)
    nres = C.c_int(0)
    NO_EARLY_STOP, DEVICE_VECTORS, X0_ZERO = 2, 4, 1

    # warm-up from x0 = 0 with host vectors (leaves b and the iterate resident in HBM)
    _lib.check(L.amg_hier_solve(h, b.ctypes.data, x.ctypes.data, 0.0, args.warmup, 0, _lib.dp(res),
                                C.byref(nres), NO_EARLY_STOP | X0_ZERO))
    warm_res = res[:nres.value].copy()

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # timed: EXACTLY K steps, inputs resident (device pointers), continuing from the warm iterate
    db, dx = L.amg_hier_dev_b(h), L.amg_hier_dev_x(h)
    sync_all()
    t0 = time.perf_counter()
    _lib.check(L.amg_hier_solve(h, db, dx, 0.0, args.steps, 0, _lib.dp(res), C.byref(nres),
                                NO_EARLY_STOP | DEVICE_VECTORS))
    sync_all()
    wall = time.perf_counter() - t0
    ev_ms = L.amg_hier_last_solve_ms(h)
    timed_res = res[:nres.value].copy()
    if dist is not None:
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())

    cycle_bytes = dev.cycle_bytes("V")
    cycles_per_s = world * args.steps / wall

    out = None
    if rank == 0:
        A0 = ml.levels[0].A
        reps = 20
        NAMES = {0: "csr_stream_kernel", 1: "csr_pattern_kernel", 2: "stencil2_kernel", 3: "sell_kernel"}
        PMC_KEYS = {"sell_kernel": "level1_residual", "stencil_coded_kernel": "level0_coded", "stencil2_kernel": "level0_values"}

        def residual_kernel(lvl):
            """r = b - A x of level lvl as the cycle launches it: time, bytes of SURVEY 8(d), bytes its storage form streams"""
            Al = ml.levels[lvl].A
            nl = Al.shape[0]
            ms = dev.time_spmv(lvl, 0, mode=1, reps=reps)
            alg = 12.0 * Al.nnz + 4.0 * (nl + 1) + 8.0 * nl + 8.0 * nl + 8.0 * nl      # bytes_spmv(A) + 8 n
            form = L.amg_hier_operator_form(h, lvl)
            moved = L.amg_hier_operator_bytes(h, lvl, 1)
            coded = L.amg_hier_value_index(h, lvl, -1)
            name = NAMES[form] if not coded else "stencil_coded_kernel"
            # `achieved` / `frac` price the launch at the bytes this kernel's storage form actually streams (DESIGN.md
            # section 4) -- the figure bounded by the HBM peak; `csr_equivalent_GBs` prices it at the CSR bytes of
            # SURVEY.md 8(d) (12 B per stored entry), what the reference's csr_matvec would have to stream.
            return {"kernel": "%s<RESIDUAL> (level-%d A-application, r = b - A x%s)" %
                              (name, lvl, ", one-byte value codes: %d distinct values" % coded if coded else ""),
                    "form": form, "value_index_distinct_values": coded,
                    "ms_per_launch": round(ms, 4), "bytes_per_launch": moved,
                    "achieved": round(moved / (ms * 1e-3) / 1e9, 1), "frac": round(moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "csr_equivalent_bytes_per_launch": alg, "csr_equivalent_GBs": round(alg / (ms * 1e-3) / 1e9, 1),
                    "csr_equivalent_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        k0 = residual_kernel(0)
        k1 = residual_kernel(1) if len(ml.levels) > 2 else None
        # every level's A is applied the same number of times per cycle (smoother applications + the residual), so the
        # operator with the longer launch is the one the step spends most of its time in
        dom, other = (k1, k0) if (k1 is not None and k1["ms_per_launch"] > k0["ms_per_launch"]) else (k0, k1)
        # (the plain CSR / offset-pattern kernels can only be timed beside it when the hierarchy kept the CSR arrays:
        #  AMG_RELEASE_SOURCES=0; by default a hierarchy of this size releases them -- 22 GB of HBM)
        try:
            ms_resid_csr = dev.time_spmv(0, 0, mode=3, reps=reps) # level 0 through the plain CSR stream kernel
            ms_resid_pat = dev.time_spmv(0, 0, mode=5, reps=reps) # ... and through the offset-pattern kernel
        except Exception:       # noqa: BLE001
            ms_resid_csr = ms_resid_pat = None
        ms_matvec = dev.time_spmv(0, 0, mode=0, reps=reps)
        ms_P = dev.time_spmv(0, 1, mode=0, reps=reps)
        ms_R = dev.time_spmv(0, 2, mode=0, reps=reps)
        roofline = {"bound": "hbm", "kernel": dom["kernel"],
                    "achieved": dom["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"], "traffic": None,
                    "bytes_per_launch": dom["bytes_per_launch"], "ms_per_launch": dom["ms_per_launch"],
                    "csr_equivalent_bytes_per_launch": dom["csr_equivalent_bytes_per_launch"],
                    "csr_equivalent_GBs": dom["csr_equivalent_GBs"], "csr_equivalent_frac": dom["csr_equivalent_frac"],
                    "second_kernel": other,
                    "plain_csr_stream_ms_per_launch": None if ms_resid_csr is None else round(ms_resid_csr, 4),
                    "pattern_kernel_ms_per_launch": None if ms_resid_pat is None else round(ms_resid_pat, 4),
                    "hbm_resident_GB": round(dev.device_bytes() / 1e9, 2),
                    "csr_sources_released_GB": round(getattr(dev, "released_bytes", 0) / 1e9, 2),
                    "cycle_bytes": cycle_bytes,
                    "cycle_achieved_GBs": round(cycle_bytes * (args.steps / (ev_ms * 1e-3)) / 1e9, 1),
                    "cycle_bytes_moved": dev.cycle_bytes_moved("V"),
                    "cycle_moved_GBs": round(dev.cycle_bytes_moved("V") * (args.steps / (ev_ms * 1e-3)) / 1e9, 1),
                    "other_kernels_ms": {"A0_matvec": round(ms_matvec, 4), "P0_matvec": round(ms_P, 4),
                                         "R0_matvec": round(ms_R, 4)}}
        # The level-0 operator of this benchmark has constant coefficients (2 distinct values + the padding zero), so the
        # library stores its values as one-byte codes into a dictionary (automatic since r3, amg_hier_value_index; the
        # products use the same doubles: tests/test_gpu_parity.py::test_value_index_is_lossless_and_automatic).  What the
        # same step costs with the 8-byte values -- the r1/r2 headline, and what a variable-coefficient operator of the
        # same shape would get -- is timed here as an extra.
        value_index = None
        nd0 = L.amg_hier_value_index(h, 0, -1)
        if nd0 > 0 and not args.no_value_index:
            try:
                L.amg_hier_value_index(h, 0, 0)
                res_vi = np.zeros(args.steps + 2); n_vi = C.c_int(0)
                for _ in range(2):          # first pass warms up / captures, second is timed
                    _lib.check(L.amg_hier_solve(h, db, dx, 0.0, args.steps, 0, _lib.dp(res_vi), C.byref(n_vi),
                                                NO_EARLY_STOP | DEVICE_VECTORS))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                _lib.check(L.amg_hier_solve(h, db, dx, 0.0, args.steps, 0, _lib.dp(res_vi), C.byref(n_vi),
                                            NO_EARLY_STOP | DEVICE_VECTORS))
                torch.cuda.synchronize()
                w_vi = time.perf_counter() - t0
                kv = residual_kernel(0)
                value_index = {"level0_distinct_values": int(nd0),
                               "off": {"value": round(args.steps / w_vi, 4), "ms_per_step": round(1e3 * w_vi / args.steps, 4),
                                       "level0": kv, "cycle_bytes_moved": dev.cycle_bytes_moved("V")},
                               "note": "automatic and lossless (same doubles in every product, bit-identical iterates); "
                                       "`off` = the same step with level 0's 8-byte values, the r1/r2 headline configuration"}
            except Exception as e:      # noqa: BLE001 -- an extra must never take the bench line down
                value_index = {"error": repr(e)}
            L.amg_hier_value_index(h, 0, 1)
        elif nd0 > 0:
            value_index = {"level0_distinct_values": int(nd0)}
        roofline["value_index"] = value_index
        pmc = os.path.join(ROOT, "profiles", "r03_pmc_summary.json")
        if os.path.exists(pmc):
            pj = json.load(open(pmc))
            ents = pj.get("entries", {}) if pj.get("grid") == args.grid else {}
            ent = ents.get(PMC_KEYS.get(dom["kernel"].split("<")[0]))
            if ent is not None:
                # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel (profiles/README.md): a COMMITTED
                # measurement of the same launch, not taken in this run (counters need rocprofv3 around the process)
                roofline["traffic"] = ent["traffic_bytes"]
                roofline["traffic_source"] = "from_file: profiles/r03_pmc_summary.json entry %s (%s)" % (
                    PMC_KEYS[dom["kernel"].split("<")[0]], pj["source"])
            ent2 = ents.get(PMC_KEYS.get(other["kernel"].split("<")[0])) if other is not None else None
            if ent2 is not None:
                other["traffic"] = ent2["traffic_bytes"]
        cpu = None
        if not args.no_cpu_baseline:
            def gpu_first():
                xg = np.zeros(n)
                r1 = np.zeros(3); n1 = C.c_int(0)
                _lib.check(L.amg_hier_solve(h, b.ctypes.data, xg.ctypes.data, 0.0, 1, 0, _lib.dp(r1), C.byref(n1),
                                            NO_EARLY_STOP | X0_ZERO))
                return xg, float(r1[1])
            cpu = cpu_baseline_of(ml, b, gpu_first)
        out = {
            "metric": "V-cycle iterations/sec (3D Poisson %d^3 fp64, SA-AMG, %s smoother)" % (args.grid, args.smoother),
            "value": round(cycles_per_s, 4), "unit": "V-cycle iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 4),
            "higher_is_better": True, "scaling": "strong" if not args.replicas else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "3D Poisson %dx%dx%d (%.1fM dof, nnz %.1fM) CSR fp64, smoothed aggregation "
                                   "(%d levels), %s pre/post smoother, V(1,1), b=rand seed 0" %
                                   (args.grid, args.grid, args.grid, n / 1e6, A0.nnz / 1e6, len(ml.levels),
                                    "Chebyshev degree 2" if args.smoother == "chebyshev" else args.smoother),
                       "parallelism": "single GPU" if world == 1 else "%d replicas (no exchange)" % world,
                       "levels": [[int(l.A.shape[0]), int(l.A.nnz)] for l in ml.levels],
                       "setup_seconds": setup_seconds(t_gen, t_setup),
                       "device_event_ms_per_step": round(ev_ms / args.steps, 4),
                       "residuals": [float(warm_res[0]), float(warm_res[-1]), float(timed_res[-1])]},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
