#!/usr/bin/env python3
"""The irregular operators of the metric configuration's cycle -- A_1 (sliced form), R_0 (sliced form), P_0 (CSR stream
kernel) -- launched a few times each for PMC collection, with the calibration kernel of known bytes (norm stage 1: 8 n
read).  Prints the algorithmic bytes per launch next to the launch times.
Usage:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_ops.py 500"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
g = int(sys.argv[1]) if len(sys.argv) > 1 else 500
np.random.seed(0)
sm = ("chebyshev", {"degree": 2})
ml = smoothed_aggregation_solver(poisson((g, g, g)), presmoother=sm, postsmoother=sm)
dev = ml.device_hierarchy()
L = _lib.lib()
A1, P0, R0 = ml.levels[1].A, ml.levels[0].P, ml.levels[0].R
spmv = lambda M: 12.0 * M.nnz + 4.0 * (M.shape[0] + 1) + 8.0 * M.shape[1] + 8.0 * M.shape[0]
out = {"A_1 residual (r = b - A x)": (dev.time_spmv(1, 0, mode=1, reps=4), spmv(A1) + 8.0 * A1.shape[0]),
       "R_0 matvec": (dev.time_spmv(0, 2, mode=0, reps=4), spmv(R0)),
       "P_0 matvec": (dev.time_spmv(0, 1, mode=0, reps=4), spmv(P0))}
n = ml.levels[0].A.shape[0]
b = np.random.rand(n); x = np.zeros(n); res = np.zeros(4); nres = C.c_int()
_lib.check(L.amg_hier_solve(dev.h, b.ctypes.data, x.ctypes.data, 0.0, 0, 0, _lib.dp(res), C.byref(nres), 1))    # calibration: norms read 8 n
for k, (ms, by) in out.items():
    print("%-28s %.4f ms per launch, algorithmic %.4f GB -> %.0f GB/s" % (k, ms, by / 1e9, by / ms / 1e6))
print("shapes: A_1 %s nnz %d, R_0 %s nnz %d, P_0 %s nnz %d; calibration n = %d" % (A1.shape, A1.nnz, R0.shape, R0.nnz, P0.shape, P0.nnz, n))
