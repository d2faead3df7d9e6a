#!/usr/bin/env python3
"""A/B of the 16-bit column codes on the irregular operators of the 500^3 hierarchy."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from pyamg_amd import _lib
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 500
_lib.lib().amg_set_index16(1)      # codes are built at upload only when switched on
ml, _ = bench.build_hierarchy(grid, "chebyshev")
dev = ml.device_hierarchy()
L = _lib.lib()
for lvl, which, name in ((1, 0, "A1 residual"), (0, 1, "P0 matvec"), (0, 2, "R0 matvec"), (1, 1, "P1"), (1, 2, "R1"), (2, 0, "A2")):
    r = {}
    for on in (1, 0, 1, 0):
        L.amg_set_index16(on)
        r.setdefault(on, []).append(dev.time_spmv(lvl, which, mode=1 if which == 0 else 0, reps=20))
    print("%-12s  index16 on: %s ms   off: %s ms" % (name, ["%.4f" % v for v in r[1]], ["%.4f" % v for v in r[0]]), flush=True)
