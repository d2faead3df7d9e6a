#!/usr/bin/env python3
"""A/B of differently compiled libamgcore_hip.so builds on one box: every library (AMGCORE_HIP_LIB) times A_1, P_0, R_0
and A_0 of the same SA hierarchy in a process of its own.  Usage: python tools/lib_ab.py GRID lib1.so lib2.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, numpy as np
sys.path.insert(0, %r)
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
g = int(sys.argv[1])
np.random.seed(0)
ml = smoothed_aggregation_solver(poisson((g, g, g)), presmoother=("chebyshev", {"degree": 2}), postsmoother=("chebyshev", {"degree": 2}))
dev = ml.device_hierarchy()
b = np.random.rand(g ** 3); x = np.zeros(g ** 3)
dev.solve(b, x, 0.0, 2, "V", x0_zero=True, fixed=True)
out = []
for rep in range(2):
    out.append([dev.time_spmv(1, 0, mode=1, reps=30), dev.time_spmv(0, 1, mode=0, reps=30), dev.time_spmv(0, 2, mode=0, reps=30), dev.time_spmv(0, 0, mode=1, reps=30)])
import time
res = []
t0 = time.perf_counter(); dev.solve(b, x, 0.0, 30, "V", x0_zero=True, fixed=True); t = (time.perf_counter() - t0) / 30 * 1e3
print("A_1 %%.4f %%.4f  P_0 %%.4f %%.4f  R_0 %%.4f %%.4f  A_0 %%.4f %%.4f ms   cycle %%.3f ms" %% (out[0][0], out[1][0], out[0][1], out[1][1], out[0][2], out[1][2], out[0][3], out[1][3], t))
''' % ROOT
grid = sys.argv[1]
for lib in sys.argv[2:] * 2:
    env = dict(os.environ, AMGCORE_HIP_LIB=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD, grid], env=env, capture_output=True, text=True)
    print("%-40s %s" % (os.path.basename(lib), (r.stdout.strip().splitlines() or [r.stderr.strip()[-300:]])[-1]), flush=True)
