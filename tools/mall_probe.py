#!/usr/bin/env python3
"""Does the Infinity Cache keep a chunk of the level-0 operator between two passes?  Two operator
applications over the 500^3 level, (A) one after the other over all rows, (B) interleaved chunk by chunk
(pass 2 lags one chunk behind pass 1, as a dependent smoother / residual pair would have to)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyamg_amd.aggregation import poisson
from pyamg_amd.distributed import HipBackend
grid = int(sys.argv[1]) if len(sys.argv) > 1 else 500
A = poisson((grid, grid, grid))
n = A.shape[0]
be = HipBackend(0)
hm = be.mat(n, n, A.indptr.astype(np.intc), A.indices.astype(np.intc), A.data)
print("form", be.L.amg_mat_form(hm), flush=True)
x = torch.rand(n, dtype=torch.float64, device="cuda")
b = torch.rand(n, dtype=torch.float64, device="cuda")
r1 = torch.zeros(n, dtype=torch.float64, device="cuda")
r2 = torch.zeros(n, dtype=torch.float64, device="cuda")

def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

def full():
    be.apply(hm, 2, x, b, None, r1, None, 0.0)
    be.apply(hm, 2, x, b, None, r2, None, 0.0)

print("two full passes: %.3f ms" % timeit(full), flush=True)
for chunk in (262144, 524288, 1048576, 2097152, 4194304, 8388608, 16777216):
    bounds = list(range(0, n, chunk)) + [n]
    def chunked():
        for k in range(len(bounds) - 1):
            be.apply_rows(hm, 2, bounds[k], bounds[k + 1], x, b, None, r1, None, 0.0)
            if k > 0:
                be.apply_rows(hm, 2, bounds[k - 1], bounds[k], x, b, None, r2, None, 0.0)
        be.apply_rows(hm, 2, bounds[-2], bounds[-1], x, b, None, r2, None, 0.0)
    print("chunk %9d rows (%6.1f MB of values): %.3f ms  (%d launches)" % (chunk, chunk * 56 / 1e6, timeit(chunked), 2 * (len(bounds) - 1)), flush=True)
