#!/usr/bin/env python3
"""BASELINE configuration C5 at (near) full size: anisotropic diffusion, P1 on a jittered Kuhn tetrahedral mesh,
C5_GRID^3 unknowns (default 345^3 = 41.1 M; 360^3 = 46.7 M is the largest whose block data array stays below 2^31
values, the int32 addressing of the reference's kernels and of scipy's csr_tobsr) as BSR with 3x3 blocks, our own block-SA setup,
symmetric block Gauss-Seidel (relaxation.py:509-590) and block Jacobi (:430-506) smoothers on one MI355X.
Prints one JSON line per smoother: V-cycle rate, the level-0 smoother's achieved HBM GB/s at the ALGORITHMIC
bytes of SURVEY 8(d) (BSR: 8 nnz + 4 nnz / bs^2 + 4 (rows / bs + 1); + 8 n bs for Dinv; x, b, out: 8 n each),
and whether the first iterate equals the CPU oracle's bit for bit at this size."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib  # noqa: E402
import pyamg_amd  # noqa: E402
from pyamg_amd.aggregation import smoothed_aggregation_solver  # noqa: E402
from pyamg_amd.gallery import tet_diffusion  # noqa: E402


def log(msg):
    sys.stderr.write("[c5 %7.1fs] %s\n" % (time.time() - T0, msg))
    sys.stderr.flush()


def oracle_of(ml):
    levels = []
    for lvl in ml.levels:
        L = {"A": lvl.A}
        if hasattr(lvl, "P"):
            L.update(P=lvl.P, R=lvl.R, pre=dict(lvl.presmoother.desc), post=dict(lvl.postsmoother.desc))
        levels.append(L)
    kind, M = ml.coarse_solver.device_form(ml.levels[-1].A)
    return oracle_lib.Hierarchy(levels, M)


def smoother_bytes(A, sweeps):
    """algorithmic bytes of `sweeps` directional passes of a block smoother over A (BSR bs x bs)"""
    bs = A.blocksize[0]
    nb, nblk = A.shape[0] // bs, len(A.indices)
    n = A.shape[0]
    per_pass = 8.0 * nblk * bs * bs + 4.0 * nblk + 4.0 * (nb + 1) + 8.0 * n * bs + 3 * 8.0 * n
    return sweeps * per_pass


def run(name, ml, A, sweeps_per_application, steps, check_oracle):
    n = A.shape[0]
    np.random.seed(0)
    b = np.random.rand(n)
    dev = ml.device_hierarchy()
    log("%s: hierarchy in HBM (%.1f GB)" % (name, dev.device_bytes() / 1e9))
    x = np.zeros(n)
    dev.solve(b, x, 0.0, 2, "V", x0_zero=True, fixed=True)                  # warm-up
    x = np.zeros(n)
    res = dev.solve(b, x, 0.0, steps, "V", x0_zero=True, fixed=True)
    ms_step = dev.last_solve_ms() / steps
    ms_relax = dev.time_relax(0, 0, reps=3)
    ms_spmv = dev.time_spmv(0, 0, mode=1, reps=3)
    out = {"config": name, "unknowns": int(n), "blocks": int(len(A.indices)), "blocksize": 3,
           "levels": [[int(l.A.shape[0]), int(l.A.nnz)] for l in ml.levels],
           "ms_per_step": round(ms_step, 3), "vcycles_per_s": round(1e3 / ms_step, 3),
           "level0_smoother_ms": round(ms_relax, 3),
           "level0_smoother_GBs": round(smoother_bytes(A, sweeps_per_application) / (ms_relax * 1e-3) / 1e9, 1),
           "level0_residual_ms": round(ms_spmv, 3),
           "level0_residual_GBs": round((smoother_bytes(A, 1) - 8.0 * n * 3) / (ms_spmv * 1e-3) / 1e9, 1),
           "cycle_algorithmic_GBs": round(dev.cycle_bytes("V") / (ms_step * 1e-3) / 1e9, 1),
           "hbm_resident_GB": round(dev.device_bytes() / 1e9, 2),
           "residuals_first_last": [float(res[0]), float(res[-1])],
           # one application of the presmoother on every coarser level that has one (ms): what the cycle spends below level 0
           "coarse_smoother_ms": [round(dev.time_relax(l, 0, reps=3), 3) for l in range(1, len(ml.levels) - 1)]}
    out["roofline"] = {"bound": "hbm", "kernel": "bsell_kernel (level-0 smoother from the sliced block form; bsr_stream_kernel where a level has no slices)", "achieved": out["level0_smoother_GBs"],
                       "peak": 8000.0, "unit": "GB/s", "frac": round(out["level0_smoother_GBs"] / 8000.0, 4)}
    if check_oracle:
        x1 = np.zeros(n)
        dev.cycle(b, x1, "V", x0_zero=True)
        log("oracle cycle on the host ...")
        H = oracle_of(ml)
        xo = np.zeros(n)
        t0 = time.perf_counter()
        H.cycle(xo, b, "V")
        out["cpu_oracle_s_per_cycle"] = round(time.perf_counter() - t0, 2)
        out["first_iterate_bit_equal_to_oracle"] = bool(np.array_equal(x1, xo))
    print(json.dumps(out), flush=True)
    return out


T0 = time.time()
m = int(os.environ.get("C5_GRID", "345"))
steps = int(os.environ.get("C5_STEPS", "5"))
log("assembling %d^3 = %d unknowns" % (m, m ** 3))
A = tet_diffusion(m, blocksize=3)
log("A: %d block rows, %d blocks (%.1f GB)" % (A.shape[0] // 3, len(A.indices), A.data.nbytes / 1e9))
np.random.seed(0)
bgs = ("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3})
ml = smoothed_aggregation_solver(A, presmoother=bgs, postsmoother=bgs)
log("setup done: %s" % [int(l.A.shape[0]) for l in ml.levels])
tag = "C5: anisotropic diffusion, P1 on a jittered Kuhn tet mesh, %d^3 = %d unknowns, BSR bs=3, SA, " % (m, A.shape[0])
run(tag + "symmetric block Gauss-Seidel", ml, A, 2, steps, os.environ.get("C5_ORACLE", "1") != "0")
bj = ("block_jacobi", {"omega": 4.0 / 3.0, "blocksize": 3})
pyamg_amd.change_smoothers(ml, bj, bj)
run(tag + "block Jacobi (omega = 4/3 / rho)", ml, A, 1, steps, False)
log("done")
