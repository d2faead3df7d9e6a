#!/usr/bin/env python3
"""What a dependency level of the exact Gauss-Seidel sweep costs on each operator of a 3-D smoothed-aggregation
hierarchy: every level's A as a stand-alone operator (amg_mat_*), forward sweeps timed with HIP events; chained sweep
(default) against one launch per level and against the dataflow sweep.  usage: gs_level_cost.py [g=128] [look-ahead=0]"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
import scipy.sparse as sps
g = int(sys.argv[1]) if len(sys.argv) > 1 else 128
LA = int(sys.argv[2]) if len(sys.argv) > 2 else 0
L = _lib.lib()
ml = smoothed_aggregation_solver(poisson((g, g, g)))
ip = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))
dp = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for lvl in ml.levels[:-1]:
    A = sps.csr_matrix(lvl.A); A.sort_indices()
    n = A.shape[0]
    Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)
    out = []
    for chain, flow in ((2, 0), (0, 0), (2, 2)):
        L.amg_set_gs_chain(chain); L.amg_set_gs_flow(flow); L.amg_set_gs_flow_lookahead(LA)
        m = L.amg_mat_create(0, n, n, ip(Ap), ip(Aj), dp(Ax))
        assert L.amg_mat_build_gs(m, None, 0) == 0
        nl = L.amg_mat_gs_levels(m)
        x = torch.zeros(n, dtype=torch.float64, device="cuda"); b = torch.rand(n, dtype=torch.float64, device="cuda")
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        for _ in range(3): L.amg_mat_gs_sweep(m, P(x), P(b), 0, 0, st)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): L.amg_mat_gs_sweep(m, P(x), P(b), 0, 0, st)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 10 * 1e3)
        L.amg_mat_destroy(m)
    L.amg_set_gs_chain(2); L.amg_set_gs_flow(1); L.amg_set_gs_flow_lookahead(0)
    assert L.amg_gs_flow_status() == 0
    print("%9d rows, %5.1f entries/row, %5d dependency levels (%.0f rows each): chained %8.1f us/sweep = %.2f us/level   launch per level %8.1f us = %.2f us/level   dataflow %8.1f us = %.2f us/level"
          % (n, A.nnz / n, nl, n / nl, out[0], out[0] / nl, out[1], out[1] / nl, out[2], out[2] / nl), flush=True)
