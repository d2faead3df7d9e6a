#!/usr/bin/env python3
"""A/B of the persistent, software-pipelined CSR stream kernel (amg_set_stream_pipe) on the operators of the
GRID^3 Poisson hierarchy that have no grid structure (A_1, P_0, R_0, ...) and on level 0 through the plain CSR
kernel: ms per launch and achieved GB/s at the algorithmic bytes 12 nnz + 4 (rows + 1) + 8 cols + 8 rows."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 400
A = poisson((grid, grid, grid))
np.random.seed(0)
sm = ("chebyshev", {"degree": 2})
ml = smoothed_aggregation_solver(A, presmoother=sm, postsmoother=sm)
dev = ml.device_hierarchy()
L = _lib.lib()
b = np.random.rand(A.shape[0]); x = np.zeros(A.shape[0])
dev.solve(b, x, 0.0, 2, "V", x0_zero=True, fixed=True)
ops = [("A0 (plain CSR kernel)", 0, 0, 3, ml.levels[0].A), ("A1", 1, 0, 1, ml.levels[1].A), ("P0", 0, 1, 0, ml.levels[0].P),
       ("R0", 0, 2, 0, ml.levels[0].R), ("A2", 2, 0, 1, ml.levels[2].A), ("P1", 1, 1, 0, ml.levels[1].P), ("R1", 1, 2, 0, ml.levels[1].R)]
out = {}
for on in (0, 1, 0, 1):
    L.amg_set_stream_pipe(on)
    for name, lvl, which, mode, M in ops:
        ms = dev.time_spmv(lvl, which, mode=mode, reps=20)
        by = 12.0 * M.nnz + 4.0 * (M.shape[0] + 1) + 8.0 * M.shape[1] + 8.0 * M.shape[0] + (8.0 * M.shape[0] if mode & 1 else 0.0)
        out.setdefault(name, {}).setdefault("pipe%d" % on, []).append((round(ms, 4), round(by / ms / 1e6, 0)))
for name, v in out.items():
    print("%-24s plain %s   pipelined %s" % (name, v["pipe0"], v["pipe1"]))
# whole step
for on in (0, 1):
    L.amg_set_stream_pipe(on)
    x = np.zeros(A.shape[0])
    dev.solve(b, x, 0.0, 3, "V", x0_zero=True, fixed=True)
    dev.solve(b, x, 0.0, 10, "V", x0_zero=False, fixed=True)
    print("pipe=%d: %.3f ms per step" % (on, dev.last_solve_ms() / 10))
