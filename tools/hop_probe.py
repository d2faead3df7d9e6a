#!/usr/bin/env python3
"""The hand-off latency of the dataflow Gauss-Seidel sweep in isolation: a tridiagonal operator has one row per dependency
level, so a forward sweep is a chain of n hand-offs between waves (store -> visible -> polled -> row sum -> store).
Prints us per level for the dataflow sweep (several look-ahead settings) and for the one-workgroup LDS chain.
usage: hop_probe.py [n=20000]"""
import sys, os, ctypes, numpy as np, torch, scipy.sparse as sps
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = _lib.lib()
A = sps.diags([-1.0 * np.ones(n - 1), 2.5 * np.ones(n), -1.0 * np.ones(n - 1)], [-1, 0, 1], format="csr")
Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for flow, la in ((0, 0), (2, 1), (2, 4), (2, 64)):
    L.amg_set_gs_flow(flow); L.amg_set_gs_flow_lookahead(la)
    m = L.amg_mat_create(0, n, n, _lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax))
    _lib.check(L.amg_mat_build_gs(m, None, 0))
    nl = L.amg_mat_gs_levels(m)
    x = torch.zeros(n, dtype=torch.float64, device="cuda"); b = torch.rand(n, dtype=torch.float64, device="cuda")
    for _ in range(2): L.amg_mat_gs_sweep(m, P(x), P(b), 0, 0, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): L.amg_mat_gs_sweep(m, P(x), P(b), 0, 0, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print("%-34s %d levels: %9.1f us per sweep = %.3f us per level" % ("one-workgroup LDS chain" if flow == 0 else "dataflow, look-ahead %d (32+ waves)" % la, nl, us, us / nl), flush=True)
    L.amg_mat_destroy(m)
L.amg_set_gs_flow(1); L.amg_set_gs_flow_lookahead(0)
assert L.amg_gs_flow_status() == 0
