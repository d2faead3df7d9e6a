#!/usr/bin/env python3
"""The README-style call at the metric size: ml.solve(b, tol=1e-8) on 3-D Poisson 500^3 (host vectors in,
host vector out), iterations / time to solution, with the V-cycle alone and as a PCG preconditioner."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 500
ml, (t_gen, t_setup) = bench.build_hierarchy(grid, "chebyshev")
n = ml.levels[0].A.shape[0]
np.random.seed(0)
b = np.random.rand(n)
ml.solve(b, tol=0.5, maxiter=2)                      # upload + warm-up
for accel in (None, "cg"):
    res = []
    t0 = time.perf_counter()
    x = ml.solve(b, tol=1e-8, maxiter=200, accel=accel, residuals=res)
    t = time.perf_counter() - t0
    A = ml.levels[0].A
    print("accel=%s: %d iterations, %.3f s wall (host vectors in/out), residual history %.3e -> %.3e, true ||b-Ax||/||b|| = %.3e"
          % (accel, len(res) - 1, t, res[0], res[-1], np.linalg.norm(b - A * x) / np.linalg.norm(b)), flush=True)
