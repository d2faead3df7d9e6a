#!/usr/bin/env python3
"""A/B of the two-rows-per-lane stencil kernel (amg_set_stencil_pairs) on BASELINE configuration C2 (2-D Poisson
2000 x 2000, SA, weighted Jacobi) and on a 3-D 200^3 Chebyshev hierarchy: ms per cycle, same bits."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
L = _lib.lib()
cases = [("C2 2000^2 SA Jacobi", lambda: smoothed_aggregation_solver(poisson((2000, 2000)), presmoother=("jacobi", {"omega": 4.0 / 3.0}), postsmoother=("jacobi", {"omega": 4.0 / 3.0}))),
         ("200^3 SA Chebyshev(2)", lambda: smoothed_aggregation_solver(poisson((200, 200, 200)), presmoother=("chebyshev", {"degree": 2}), postsmoother=("chebyshev", {"degree": 2})))]
for name, build in cases:
    np.random.seed(0)
    ml = build()
    b = np.random.rand(ml.levels[0].A.shape[0])
    out = {}
    for on in (2, 0, 2, 0):
        L.amg_set_stencil_pairs(on)
        res = []
        ml.solve(b, tol=0.0, maxiter=5, residuals=res)
        t0 = time.perf_counter()
        x = ml.solve(b, tol=0.0, maxiter=50, residuals=res)
        out.setdefault(on, []).append("%.3f" % ((time.perf_counter() - t0) / 50 * 1e3))
        out[("x", on)] = x
    L.amg_set_stencil_pairs(1)
    print("%-24s two rows per lane %s   one row per lane %s ms/cycle   same bits: %s" % (name, out[2], out[0], np.array_equal(out[("x", 0)], out[("x", 2)])), flush=True)
