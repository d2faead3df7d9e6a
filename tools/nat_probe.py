#!/usr/bin/env python3
"""Where the time of amg_hier_gs_natural goes (the setup's candidate improvement on the GPU): operator upload, then calls
with 1, 2 and 8 sweeps -- fixed cost (task cuts and order on the host, vectors over PCIe) vs cost per sweep.
usage: python tools/nat_probe.py [grid]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
from pyamg_amd.util import _DeviceOperator
g = int(sys.argv[1]) if len(sys.argv) > 1 else 500
A = poisson((g, g, g))
L = _lib.lib()
t = time.time(); op = _DeviceOperator(A); print("operator in HBM %.2f s" % (time.time() - t), flush=True)
x = np.ones(A.shape[0])
for dirs in ([0], [0, 1], [0, 1] * 4, [0, 1] * 4):
    t = time.time()
    rc = L.amg_hier_gs_natural(op.h, 0, x.ctypes.data, None, bytes(dirs), len(dirs))
    print("%d sweep(s): rc %d, %.3f s" % (len(dirs), rc, time.time() - t), flush=True)
op.close()
