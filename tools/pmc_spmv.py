#!/usr/bin/env python3
"""Level-0 A-application (r = b - A x) at 500^3 for PMC collection: a few launches of
csr_stream_kernel<SM_RESIDUAL> plus two calibration kernels of known byte counts
(scale: 8n read + 8n written; norm stage 1: 8n read), all 8 B/lane or 16 B/lane accesses.
Usage under rocprofv3:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_spmv.py 500
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
from pyamg_amd.util import _DeviceOperator

grid = int(sys.argv[1]) if len(sys.argv) > 1 else 500
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A = poisson((grid, grid, grid))
n = A.shape[0]
L = _lib.lib()
L.amg_set_stream_variant(variant)
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 0
L.amg_set_xcd_chunk(chunk)
period = int(sys.argv[4]) if len(sys.argv) > 4 else 1
L.amg_set_xcd_period(period)
op = _DeviceOperator(A)
_lib.check(L.amg_hier_finalize(op.h))
ms = C.c_double()
_lib.check(L.amg_hier_time_spmv(op.h, 0, 0, 1, 5, C.byref(ms)))          # as the cycle runs it (pattern form if detected)
ms_csr = C.c_double()
_lib.check(L.amg_hier_time_spmv(op.h, 0, 0, 3, 5, C.byref(ms_csr)))      # plain CSR stream kernel
ms_pat = C.c_double()
_lib.check(L.amg_hier_time_spmv(op.h, 0, 0, 5, 5, C.byref(ms_pat)))      # offset-pattern kernel
# calibration: norm (8n read) via the solve entry with maxiter 0 (norm(b), residual, norm)
b = np.random.rand(n); x = np.zeros(n); res = np.zeros(4); nres = C.c_int()
_lib.check(L.amg_hier_solve(op.h, b.ctypes.data, x.ctypes.data, 0.0, 0, 0, _lib.dp(res), C.byref(nres), 1))
print("chunk=%d period=%d" % (chunk, period), "n=%d nnz=%d algorithmic bytes per launch = %.0f ; %.4f ms per launch" %
      (n, A.nnz, 12.0 * A.nnz + 4.0 * (n + 1) + 24.0 * n, ms.value), "; plain CSR kernel %.4f ms ; pattern kernel %.4f ms" % (ms_csr.value, ms_pat.value))
