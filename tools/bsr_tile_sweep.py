#!/usr/bin/env python3
"""Sweep of the products-per-workgroup target (amg_set_tile_target) for the block kernels on the C5 operator
(tet-mesh diffusion, BSR 3x3): block Jacobi sweep, r = b - A x from the blocks, symmetric block Gauss-Seidel."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pyamg_amd
from pyamg_amd import _lib
from pyamg_amd.aggregation import smoothed_aggregation_solver
from pyamg_amd.gallery import tet_diffusion
m = int(sys.argv[1]) if len(sys.argv) > 1 else 150
A = tet_diffusion(m, blocksize=3)
np.random.seed(0)
bj = ("block_jacobi", {"omega": 4.0 / 3.0, "blocksize": 3})
ml = smoothed_aggregation_solver(A, presmoother=bj, postsmoother=("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3}))
L = _lib.lib()
n = A.shape[0]; nblk = len(A.indices); nb = n // 3
per_pass = 8.0 * nblk * 9 + 4.0 * nblk + 4.0 * (nb + 1) + 8.0 * n * 3 + 3 * 8.0 * n
b = np.random.rand(n)
for target in (1024, 2048, 3072, 4096, 6144, 8192, 2048):
    L.amg_set_tile_target(target)
    ml._invalidate_device()
    dev = ml.device_hierarchy()
    x = np.zeros(n); dev.solve(b, x, 0.0, 1, "V", x0_zero=True, fixed=True)
    tj = dev.time_relax(0, 0, reps=5); tg = dev.time_relax(0, 1, reps=3); ts = dev.time_spmv(0, 0, mode=1, reps=5)
    print("tile target %5d: block Jacobi %.3f ms = %6.0f GB/s | residual from blocks %.3f ms = %6.0f GB/s | symmetric block GS %.3f ms = %6.0f GB/s"
          % (target, tj, per_pass / tj / 1e6, ts, (per_pass - 24.0 * n) / ts / 1e6, tg, 2 * per_pass / tg / 1e6), flush=True)
