#!/usr/bin/env python3
"""Level-0 SpMV timing sweep over the CSR-stream kernel's tuning knobs (GPU box only)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
from pyamg_amd.util import _DeviceOperator
import ctypes as C

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 400
A = poisson((grid, grid, grid))
n = A.shape[0]
bytes_app = 12.0 * A.nnz + 4.0 * (n + 1) + 24.0 * n
L = _lib.lib()
op = _DeviceOperator(A)
_lib.check(L.amg_hier_finalize(op.h))
print("grid %d  n=%d nnz=%d  bytes/app=%.3f GB" % (grid, n, A.nnz, bytes_app / 1e9))
for variant in (0, 1):
    for chunk in (0, 16, 64):
        L.amg_set_stream_variant(variant); L.amg_set_xcd_chunk(chunk)
        ms = C.c_double()
        for mode in (0, 1):
            _lib.check(L.amg_hier_time_spmv(op.h, 0, 0, mode, 20, C.byref(ms)))
            print("variant %d chunk %3d mode %d: %.4f ms  %.0f GB/s" % (variant, chunk, mode, ms.value, bytes_app / ms.value / 1e6 if mode else (bytes_app - 8.0 * n) / ms.value / 1e6))
