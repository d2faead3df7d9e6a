#!/usr/bin/env python3
"""A few block Jacobi sweeps and r = b - A x on the C5 operator (tet-mesh diffusion, BSR 3x3) through the flat entry
points, for rocprofv3 --pmc passes over bsr_stream_kernel (no hierarchy, no Gauss-Seidel launches)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import relaxation
from pyamg_amd.gallery import tet_diffusion
m = int(sys.argv[1]) if len(sys.argv) > 1 else 252
A = tet_diffusion(m, blocksize=3)
n = A.shape[0]
np.random.seed(0)
b = np.random.rand(n)
x = np.zeros(n)
t0 = time.time()
relaxation.block_jacobi(A, x, b, blocksize=3, iterations=4, omega=0.5)
print("block Jacobi x4 on %d unknowns (%d blocks): %.2fs incl. upload" % (n, len(A.indices), time.time() - t0))
