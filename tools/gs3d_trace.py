#!/usr/bin/env python3
"""The reference's default setup in 3-D (smoothed aggregation, symmetric Gauss-Seidel pre/post) for a rocprofv3 kernel
trace: 10 cycles on a g^3 Poisson problem; prints the hierarchy, the dependency levels per operator and ms per cycle.
usage: gs3d_trace.py [g=128]"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
g = int(sys.argv[1]) if len(sys.argv) > 1 else 128
np.random.seed(0)
sm = ("gauss_seidel", {"sweep": "symmetric"})
ml = smoothed_aggregation_solver(poisson((g, g, g)), presmoother=sm, postsmoother=sm)
print([(l.A.shape[0], l.A.nnz) for l in ml.levels], flush=True)
b = np.random.rand(g ** 3)
res = []
ml.solve(b, tol=0.0, maxiter=3, residuals=res)
t0 = time.perf_counter()
ml.solve(b, tol=0.0, maxiter=10, residuals=res)
print("SA %d^3 symmetric Gauss-Seidel: %.2f ms per cycle" % (g, (time.perf_counter() - t0) / 10 * 1e3))
