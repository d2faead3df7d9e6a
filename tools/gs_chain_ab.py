#!/usr/bin/env python3
"""A/B of the chained Gauss-Seidel sweep (one workgroup per run of narrow dependency levels) against one
launch per level, on hierarchies whose smoother is the exact sequential Gauss-Seidel."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
from pyamg_amd.classical import ruge_stuben_solver
L = _lib.lib()
cases = [("RS 2-D 500x500 (README)", lambda: ruge_stuben_solver(poisson((500, 500)))),
         ("SA 3-D 64^3", lambda: smoothed_aggregation_solver(poisson((64, 64, 64)), presmoother=("gauss_seidel", {"sweep": "symmetric"}), postsmoother=("gauss_seidel", {"sweep": "symmetric"}))),
         ("SA 3-D 128^3", lambda: smoothed_aggregation_solver(poisson((128, 128, 128)), presmoother=("gauss_seidel", {"sweep": "symmetric"}), postsmoother=("gauss_seidel", {"sweep": "symmetric"}))),
         ("SA 2-D 1000x1000", lambda: smoothed_aggregation_solver(poisson((1000, 1000)), presmoother=("gauss_seidel", {"sweep": "symmetric"}), postsmoother=("gauss_seidel", {"sweep": "symmetric"})))]
for name, build in cases:
    np.random.seed(0)
    ml = build()
    b = np.random.rand(ml.levels[0].A.shape[0])
    out = {}
    for on in (2, 1, 0, 2, 1, 0):
        L.amg_set_gs_chain(on)
        res = []
        ml.solve(b, tol=0.0, maxiter=3, residuals=res)
        t0 = time.perf_counter()
        x = ml.solve(b, tol=0.0, maxiter=10, residuals=res)
        out.setdefault(on, []).append((time.perf_counter() - t0) / 10 * 1e3)
        out[("x", on)] = x
    L.amg_set_gs_chain(2)
    print("%-26s LDS hand-off chain %s   first-generation chain %s   per-level launches %s ms/cycle   same bits: %s" %
          (name, ["%.2f" % v for v in out[2]], ["%.2f" % v for v in out[1]], ["%.2f" % v for v in out[0]],
           np.array_equal(out[("x", 1)], out[("x", 0)]) and np.array_equal(out[("x", 2)], out[("x", 0)])), flush=True)
