#!/usr/bin/env python3
"""16-bit column codes of the sliced form on the metric configuration's irregular operators: A_1 residual and R_0 matvec
with and without them, same process.  usage: sell16_ab.py [g=400]"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
g = int(sys.argv[1]) if len(sys.argv) > 1 else 400
L = _lib.lib()
np.random.seed(0)
sm = ("chebyshev", {"degree": 2})
ml = smoothed_aggregation_solver(poisson((g, g, g)), presmoother=sm, postsmoother=sm)
b = np.random.rand(g ** 3)
for idx16 in (1, 0, 1, 0):
    L.amg_set_sell_index16(idx16)
    ml._invalidate_device()
    dev = ml.device_hierarchy()
    x = np.zeros(g ** 3)
    dev.solve(b, x, 0.0, 3, "V", x0_zero=True, fixed=True)
    t = [dev.time_spmv(1, 0, mode=1, reps=30), dev.time_spmv(0, 2, mode=0, reps=30), dev.time_spmv(0, 1, mode=0, reps=30)]
    x = np.zeros(g ** 3)
    res = dev.solve(b, x, 0.0, 30, "V", x0_zero=True, fixed=True)
    print("index16 %d: A_1 residual %.4f ms  R_0 %.4f ms  P_0 %.4f ms  cycle %.3f ms  residual %.17g  HBM %.2f GB" %
          (idx16, t[0], t[1], t[2], dev.last_solve_ms() / 30, res[-1], dev.device_bytes() / 1e9), flush=True)
