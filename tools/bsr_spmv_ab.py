#!/usr/bin/env python3
"""A/B of the BSR-native operator application against the CSR expansion on a large bs=3 operator."""
import sys, os, numpy as np, scipy.sparse as sp, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
from pyamg_amd.util import _DeviceOperator
n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
L = _lib.lib()
for bs in (2, 3, 6):
    Mb = np.eye(bs) * 4.0 + 0.3 * np.random.RandomState(bs).randn(bs, bs)
    A = sp.kron(poisson((n, n, n)), Mb).tobsr((bs, bs)); A.sort_indices()
    op = _DeviceOperator(A)
    _lib.check(L.amg_hier_finalize(op.h))
    out = {}
    for on in (2, 0, 2, 0):
        L.amg_set_bsr_spmv(on)
        ms = C.c_double()
        _lib.check(L.amg_hier_time_spmv(op.h, 0, 0, 1, 10, C.byref(ms)))
        out.setdefault(on, []).append(ms.value)
    L.amg_set_bsr_spmv(1)
    nnz = A.nnz
    print("bs=%d rows=%d nnz=%d: blocks %s ms (%.0f GB/s at 8 B/entry)   CSR expansion %s ms (%.0f GB/s at 12 B/entry)" %
          (bs, A.shape[0], nnz, ["%.3f" % v for v in out[2]], 8.0 * nnz / min(out[2]) / 1e6,
           ["%.3f" % v for v in out[0]], 12.0 * nnz / min(out[0]) / 1e6), flush=True)
    op.close()
