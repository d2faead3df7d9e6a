#!/usr/bin/env python3
"""The other BASELINE.json configurations as measured parity cases (not the bench.py line):
C1 README example, C2 2D 2000^2 SA + Jacobi, C5-shaped bs=3 block Gauss-Seidel (reduced size).
Each prints the iteration rate on the GPU, the CPU oracle's time for one step and whether the
first iterate agrees with the oracle bit for bit."""
import os, sys, time, json
import numpy as np, scipy.sparse as sps
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
from pyamg_amd.classical import ruge_stuben_solver


def oracle_of(ml):
    levels = []
    for lvl in ml.levels:
        L = {"A": lvl.A}
        if hasattr(lvl, "P"):
            L.update(P=lvl.P, R=lvl.R, pre=dict(lvl.presmoother.desc), post=dict(lvl.postsmoother.desc))
        levels.append(L)
    kind, M = ml.coarse_solver.device_form(ml.levels[-1].A)
    return oracle_lib.Hierarchy(levels, M)


def run(name, ml, A, steps=20, tol_run=None):
    n = A.shape[0]
    np.random.seed(0); b = np.random.rand(n)
    dev = ml.device_hierarchy()
    x = np.zeros(n)
    dev.solve(b, x, 0.0, 3, "V", x0_zero=True, fixed=True)                 # warm-up (+ graph capture)
    x = np.zeros(n)
    t0 = time.perf_counter(); res = dev.solve(b, x, 0.0, steps, "V", x0_zero=True, fixed=True); wall = time.perf_counter() - t0
    ms_dev = dev.last_solve_ms() / steps
    H = oracle_of(ml)
    xo = np.zeros(n); t0 = time.perf_counter(); H.cycle(xo, b, "V"); tcpu = time.perf_counter() - t0
    x1 = np.zeros(n); dev.cycle(b, x1, "V", x0_zero=True)
    out = {"config": name, "unknowns": n, "levels": [[int(l.A.shape[0]), int(l.A.nnz)] for l in ml.levels],
           "gpu_ms_per_step_device": round(ms_dev, 4), "gpu_steps_per_s": round(1e3 / ms_dev, 2),
           "wall_ms_per_step_incl_pcie": round(1e3 * wall / steps, 3),
           "cycle_algorithmic_GBs": round(dev.cycle_bytes("V") / (ms_dev * 1e-3) / 1e9, 1),
           "cpu_oracle_s_per_cycle": round(tcpu, 3), "first_iterate_bit_equal_to_oracle": bool(np.array_equal(x1, xo)),
           "residuals": [float(res[0]), float(res[-1])]}
    if tol_run is not None:
        r2 = []; ml.solve(b, tol=tol_run, residuals=r2)
        out["iterations_to_tol_%g" % tol_run] = len(r2) - 1
        out["final_residual"] = r2[-1]
    print(json.dumps(out), flush=True)


which = sys.argv[1:] or ["C1", "C2", "C5"]
if "C1" in which:
    A = poisson((500, 500))
    run("C1 README: 2D Poisson 500x500, ruge_stuben_solver, symmetric Gauss-Seidel (exact, level-scheduled)",
        ruge_stuben_solver(A), A, tol_run=1e-10)
    ml = ruge_stuben_solver(A, presmoother=("multicolor_gauss_seidel", {"sweep": "symmetric"}),
                            postsmoother=("multicolor_gauss_seidel", {"sweep": "symmetric"}))
    run("C1 variant: same hierarchy, multicolour Gauss-Seidel (extension)", ml, A, tol_run=1e-10)
if "C2" in which:
    A = poisson((2000, 2000)); np.random.seed(0)
    sm = ("jacobi", {"omega": 4.0 / 3.0})
    run("C2: 2D Poisson 2000x2000, SA, weighted Jacobi omega=4/3", smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm), A)
if "C5" in which:
    from pyamg_amd.gallery import tet_diffusion
    m = int(os.environ.get("C5_GRID", "72"))
    t0 = time.time()
    A = tet_diffusion(m, blocksize=3)
    np.random.seed(0)
    sm = ("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3})
    ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm, max_coarse=300)
    sys.stderr.write("C5 build+setup %.1fs\n" % (time.time() - t0))
    run("C5-shaped (reduced): anisotropic diffusion, P1 on a jittered Kuhn tetrahedral mesh, %d^3 = %d unknowns, "
        "BSR bs=3, SA, symmetric block Gauss-Seidel" % (m, A.shape[0]), ml, A)
    sm = ("block_jacobi", {"omega": 4.0 / 3.0, "blocksize": 3})
    ml2 = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm, max_coarse=300)
    run("C5 variant: same problem, block Jacobi (omega = 4/3 / rho)", ml2, A)
