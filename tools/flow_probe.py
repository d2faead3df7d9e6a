#!/usr/bin/env python3
"""Dataflow Gauss-Seidel sweep on the operators of a 3-D smoothed-aggregation hierarchy for several look-ahead settings
(resident waves): us per sweep and per dependency level.  usage: flow_probe.py [g=128] [la,la,...]"""
import sys, os, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
import scipy.sparse as sps
g = int(sys.argv[1]) if len(sys.argv) > 1 else 128
las = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16]
L = _lib.lib()
ml = smoothed_aggregation_solver(poisson((g, g, g)))
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for lvl in ml.levels[:-1]:
    A = sps.csr_matrix(lvl.A); A.sort_indices()
    n = A.shape[0]
    Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)
    L.amg_set_gs_flow(2)
    m = L.amg_mat_create(0, n, n, _lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax))
    _lib.check(L.amg_mat_build_gs(m, None, 0))
    nl = L.amg_mat_gs_levels(m)
    x = torch.zeros(n, dtype=torch.float64, device="cuda"); b = torch.rand(n, dtype=torch.float64, device="cuda")
    out = []
    for la in las:
        L.amg_set_gs_flow_lookahead(la)
        for seq in (np.array([0], dtype=np.uint8), np.array([0, 1], dtype=np.uint8)):
            sp_ = seq.ctypes.data_as(ctypes.c_void_p)
            for _ in range(3): L.amg_mat_gs_sweeps(m, P(x), P(b), sp_, len(seq), 0, st)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): L.amg_mat_gs_sweeps(m, P(x), P(b), sp_, len(seq), 0, st)
            e1.record(); torch.cuda.synchronize()
            out.append("la %3d %s %7.1f us = %.2f us/level" % (la, "fwd    " if len(seq) == 1 else "fwd+bwd", e0.elapsed_time(e1) / 10 * 1e3, e0.elapsed_time(e1) / 10 * 1e3 / nl / len(seq)))
    assert L.amg_gs_flow_status() == 0
    L.amg_mat_destroy(m)
    print("%9d rows, %5.1f entries/row, %5d levels (%.0f rows each):\n   " % (n, A.nnz / n, nl, n / nl) + "\n   ".join(out), flush=True)
L.amg_set_gs_flow(1); L.amg_set_gs_flow_lookahead(0)
