#!/usr/bin/env python3
"""Level-0 r = b - A x of the 3-D (and 2-D) Poisson operator in the stencil form, one row vs two rows per lane, by size."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
from pyamg_amd.util import _DeviceOperator
L = _lib.lib()
for dims in ((2000, 2000), (4000, 4000), (160,) * 3, (200,) * 3, (256,) * 3, (320,) * 3, (400,) * 3, (500,) * 3):
    A = poisson(dims)
    op = _DeviceOperator(A)
    _lib.check(L.amg_hier_finalize(op.h))
    out = []
    for on in (2, 0, 2, 0):
        L.amg_set_stencil_pairs(on)
        ms = C.c_double()
        _lib.check(L.amg_hier_time_spmv(op.h, 0, 0, 1, 20, C.byref(ms)))
        out.append(ms.value)
    L.amg_set_stencil_pairs(1)
    print("%-18s %11d rows: two rows per lane %.4f %.4f ms   one row per lane %.4f %.4f ms" % ("x".join(map(str, dims)), A.shape[0], out[0], out[2], out[1], out[3]), flush=True)
    op.close()
