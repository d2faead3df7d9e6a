#!/usr/bin/env python3
"""Does a second dependent pass over the level-0 operator get its matrix values from the 256 MB Infinity Cache when it
follows the first pass slab by slab?  Two chained residual passes r1 = b - A x, r2 = b - A r1 on the 7-point Poisson
operator of the benchmark (stencil form), (a) as two whole launches, (b) interleaved in slabs of S grid planes
(pass 2 follows pass 1 on the same slab one plane back, stream order carries the dependency; replayed from a graph).  Prints ms per pair and checks
the bits of r2.  usage: slab_probe.py [grid=500]"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson
G = int(sys.argv[1]) if len(sys.argv) > 1 else 500
L = _lib.lib()
A = poisson((G, G, G)).tocsr()
n = A.shape[0]
ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), A.data
m = L.amg_mat_create(0, n, n, ip(Ap), ip(Aj), dp(Ax))
assert m, _lib.last_error()
print("grid %d^3, %d rows, form %d" % (G, n, L.amg_mat_form(m)), flush=True)
del A, Ap, Aj, Ax
dev = torch.device("cuda:0")
torch.manual_seed(0)
x = torch.rand(n, dtype=torch.float64, device=dev)
b = torch.rand(n, dtype=torch.float64, device=dev)
r1 = torch.zeros(n, dtype=torch.float64, device=dev)
r2 = torch.zeros(n, dtype=torch.float64, device=dev)
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())


def rows(lo, hi, src, dst):
    lo, hi = max(lo, 0), min(hi, n)
    if hi > lo:
        rc = L.amg_mat_apply_rows(m, 2, lo, hi, P(src), P(b), None, P(dst), None, 0.0, 0.0, st())
        assert rc == 0, _lib.last_error()


def whole():
    rows(0, n, x, r1)
    rows(0, n, r1, r2)


def slabbed(S):
    plane = G * G
    step = S * plane
    lo = 0
    while lo < n:
        rows(lo, lo + step, x, r1)                            # pass 1 on slab s
        rows(lo - plane, lo + step - plane, r1, r2)           # pass 2 right behind it, one plane back
        lo += step
    rows(lo - plane, n, r1, r2)


def timed(f, reps=10):
    f(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()                                # replayed from a graph: no host time per launch
    with torch.cuda.graph(g):
        f()
    f = g.replay
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for pairs in (1, 0):
    L.amg_set_stencil_pairs(2 if pairs else 0)
    t = timed(whole)
    ref = r2.clone()
    print("two rows per lane %d: two whole launches %.3f ms" % (pairs, t), flush=True)
    for S in (1, 2, 3, 4, 6, 8, 12, 16, 32):
        r2.zero_()
        t = timed(lambda: slabbed(S))
        print("   slabs of %2d planes (%5.1f MB of values): %.3f ms   same bits %s" % (S, S * G * G * 56 / 1e6, t, bool(torch.equal(r2, ref))), flush=True)
L.amg_set_stencil_pairs(1)
L.amg_mat_destroy(m)
