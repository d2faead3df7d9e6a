#!/usr/bin/env python3
"""Host-side cost of one partitioned V-cycle step: DistributedSolver(world=1) on a problem so small that
the GPU work is negligible, so wall time per step ~ Python + ctypes + launch overhead of the driver."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
from pyamg_amd.distributed import DistributedSolver, HipBackend, levels_from_ml
import torch
g = int(sys.argv[1]) if len(sys.argv) > 1 else 40
A = poisson((g, g, g)); np.random.seed(0)
sm = ("chebyshev", {"degree": 2})
ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
levels, coarse = levels_from_ml(ml)
S = DistributedSolver(levels, coarse, HipBackend(0), 0, 1)
b = np.random.rand(A.shape[0])
S.set_problem(b, None)
S.run_fixed(5, "V", True)
torch.cuda.synchronize()
for steps in (50, 200):
    t0 = time.perf_counter(); S.run_fixed(steps, "V", False); torch.cuda.synchronize(); t = time.perf_counter() - t0
    print("grid %d^3, %d levels: %.1f us per step (host-bound)" % (g, len(levels), 1e6 * t / steps), flush=True)
res = []
x = ml.solve(b, tol=0.0, maxiter=200, residuals=res)
t0 = time.perf_counter(); x = ml.solve(b, tol=0.0, maxiter=200, residuals=res); t = time.perf_counter() - t0
print("resident C path (graph replay): %.1f us per step" % (1e6 * t / 200))
