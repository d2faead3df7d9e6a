#!/usr/bin/env python3
"""A/B of the sliced form (SELL-64-sigma, amg_set_sell_form) against csr_stream_kernel on the operators without grid
structure of the 3-D Poisson SA hierarchy: A_1 (r = b - A x), R_0, P_0 launches and the whole cycle, same process."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
g = int(sys.argv[1]) if len(sys.argv) > 1 else 400
L = _lib.lib()
np.random.seed(0)
ml = smoothed_aggregation_solver(poisson((g, g, g)), presmoother=("chebyshev", {"degree": 2}), postsmoother=("chebyshev", {"degree": 2}))
b = np.random.rand(g ** 3)
xs = {}
for on in (1, 0, 1, 0):
    L.amg_set_sell_form(on)
    dev = ml.device_hierarchy()
    x = np.zeros(g ** 3)
    dev.solve(b, x, 0.0, 3, "V", x0_zero=True, fixed=True)
    tA = dev.time_spmv(1, 0, mode=1, reps=30); tR = dev.time_spmv(0, 2, mode=0, reps=30); tP = dev.time_spmv(0, 1, mode=0, reps=30)
    t0 = time.perf_counter(); dev.solve(b, x, 0.0, 30, "V", x0_zero=True, fixed=True); t = (time.perf_counter() - t0) / 30 * 1e3
    xs[on] = x
    print("%s  A_1 %.4f ms  R_0 %.4f ms  P_0 %.4f ms  cycle %.3f ms  (form of A_1: %d, %.1f GB moved per cycle)" %
          ("sliced" if on else "csr   ", tA, tR, tP, t, L.amg_hier_operator_form(dev.h, 1), dev.cycle_bytes_moved("V") / 1e9), flush=True)
L.amg_set_sell_form(1)
print("same bits:", np.array_equal(xs[0], xs[1]))
