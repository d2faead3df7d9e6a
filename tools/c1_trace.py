#!/usr/bin/env python3
"""BASELINE configuration C1 (Ruge-Stuben, 2-D Poisson 500 x 500, symmetric Gauss-Seidel) for a rocprofv3 kernel trace:
20 cycles; the per-kernel table shows what the chained sweeps of each width class cost per dependency level."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
from pyamg_amd.classical import ruge_stuben_solver
gen = int(sys.argv[1]) if len(sys.argv) > 1 else 2
_lib.lib().amg_set_gs_chain(gen)
np.random.seed(0)
ml = ruge_stuben_solver(poisson((500, 500)))
b = np.random.rand(250000)
res = []
ml.solve(b, tol=0.0, maxiter=3, residuals=res)
t0 = time.perf_counter()
ml.solve(b, tol=0.0, maxiter=20, residuals=res)
print("C1 generation %d: %.2f ms per cycle" % (gen, (time.perf_counter() - t0) / 20 * 1e3))
