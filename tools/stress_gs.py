#!/usr/bin/env python3
"""Randomised cross-check of the Gauss-Seidel sweep paths: chained sweeps (LDS hand-off / memory hand-off), long-row
chains, level launches with and without entry ranges in the kernel arguments, the dataflow sweep (one persistent launch,
three look-ahead settings) -- against one plain launch per level,
bit for bit, and against the sequential reference loop (scipy-free restatement below) on the small cases.
Matrices: banded, random sparse, 2-D / 3-D grid operators with random coefficients, Galerkin-like products (long rows),
rows without a diagonal entry, zero diagonals, empty rows.  usage: stress_gs.py [seed=0] [cases=60]"""
import sys, os, ctypes, numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib, relaxation
from pyamg_amd.aggregation import poisson
L = _lib.lib()


def sequential(A, x, b, sweep):
    Ap, Aj, Ax = A.indptr, A.indices, A.data
    n = A.shape[0]
    orders = {"forward": [range(n)], "backward": [range(n - 1, -1, -1)], "symmetric": [range(n), range(n - 1, -1, -1)]}[sweep]
    for order in orders:
        for i in order:
            rsum, diag = 0.0, 0.0
            for k in range(Ap[i], Ap[i + 1]):
                if Aj[k] == i: diag = Ax[k]
                else: rsum += Ax[k] * x[Aj[k]]
            if diag != 0.0: x[i] = (b[i] - rsum) / diag
    return x


def make(rng, kind):
    if kind == 0:      # banded
        n = int(rng.randint(50, 4000)); w = int(rng.randint(1, 12))
        A = sp.diags([rng.rand(n - abs(o)) - 0.5 for o in range(-w, w + 1)], list(range(-w, w + 1)), format="csr")
        A = A + sp.identity(n) * (2.0 * w)
    elif kind == 1:    # random sparse
        n = int(rng.randint(30, 3000)); d = rng.uniform(0.001, 0.05)
        A = sp.random(n, n, density=d, random_state=rng, format="csr") + sp.identity(n) * 5.0
    elif kind == 2:    # grid operator, random coefficients
        dims = tuple(int(v) for v in rng.randint(3, 28, size=int(rng.randint(2, 4))))
        A = poisson(dims).tocsr(); A.data = A.data * (1.0 + 0.3 * rng.rand(A.nnz))
    else:              # Galerkin-like: long rows, few rows per level
        dims = tuple(int(v) for v in rng.randint(6, 22, size=3))
        A0 = poisson(dims).tocsr(); n0 = A0.shape[0]; nc = max(n0 // int(rng.randint(4, 12)), 4)
        agg = rng.randint(0, nc, size=n0)
        T = sp.csr_matrix((1.0 + rng.rand(n0), (np.arange(n0), agg)), shape=(n0, nc))
        P = (sp.identity(n0) - 0.3 * A0) @ T
        A = (P.T @ A0 @ P).tocsr() + sp.identity(nc) * 1e-3
    A = sp.csr_matrix(A); A.sort_indices(); A.eliminate_zeros()
    n = A.shape[0]
    flavour = rng.randint(0, 5)
    if flavour == 1 and n > 10:      # a zero diagonal entry, a missing diagonal entry, an empty row
        Al = A.tolil()
        i, j, k = rng.choice(n, 3, replace=False)
        Al[i, i] = 0.0; Al[k, :] = 0.0
        A = sp.csr_matrix(Al); A.eliminate_zeros()
        Al = A.tolil(); Al[j, j] = 0.0; A = sp.csr_matrix(Al)     # explicit zero stays stored
        A.sort_indices()
    return A


def fused_sequences(A, b, rng, bsr1):
    """a whole sequence of directional sweeps in one call (what the in-cycle smoothers issue): dataflow launches of up
    to four sweeps against the level-scheduled paths, on device vectors"""
    n = A.shape[0]
    Ap, Aj, Ax = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data)
    seq = rng.randint(0, 2, size=int(rng.randint(1, 7))).astype(np.uint8)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    outs = []
    for flow, la in ((0, 0), (2, 0), (2, 2), (2, 200)):
        L.amg_set_gs_flow(flow); L.amg_set_gs_flow_lookahead(la)
        m = L.amg_mat_create(0, n, n, _lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax))
        _lib.check(L.amg_mat_build_gs(m, None, 0))
        x = torch.from_numpy(np.cos(np.arange(n, dtype=float))).cuda(); bd = torch.from_numpy(b).cuda()
        _lib.check(L.amg_mat_gs_sweeps(m, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(bd.data_ptr()), seq.ctypes.data_as(ctypes.c_void_p), len(seq), int(bsr1), st))
        torch.cuda.synchronize()
        outs.append(x.cpu().numpy())
        L.amg_mat_destroy(m)
    L.amg_set_gs_flow(1); L.amg_set_gs_flow_lookahead(0)
    return all(np.array_equal(outs[0], v) for v in outs[1:]) and L.amg_gs_flow_status() == 0, len(seq)


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rng = np.random.RandomState(seed)
    bad = 0
    for case in range(ncase):
        kind = int(rng.randint(0, 4))
        A = make(rng, kind)
        n = A.shape[0]
        b = rng.rand(n)
        sweep = ("forward", "backward", "symmetric")[int(rng.randint(0, 3))]
        M = A if rng.rand() < 0.6 else sp.bsr_matrix(A, blocksize=(1, 1))
        out = {}
        its = int(rng.randint(1, 4))
        for chain, hint, flow, la in ((0, 0, 0, 0), (2, 1, 0, 0), (1, 1, 0, 0), (0, 1, 0, 0), (2, 0, 0, 0), (2, 1, 2, 0), (2, 1, 2, 1), (2, 1, 2, 64)):
            L.amg_set_gs_chain(chain); L.amg_set_gs_level_hint(hint); L.amg_set_gs_flow(flow); L.amg_set_gs_flow_lookahead(la)
            x = np.cos(np.arange(n, dtype=float))
            relaxation.gauss_seidel(M, x, b, iterations=its, sweep=sweep)
            out[(chain, hint, flow, la)] = x
        L.amg_set_gs_chain(2); L.amg_set_gs_level_hint(1); L.amg_set_gs_flow(1); L.amg_set_gs_flow_lookahead(0)
        if L.amg_gs_flow_status() != 0:
            print("case %d: a dataflow sweep ran out of its time budget" % case); bad += 1
        ok = all(np.array_equal(out[(0, 0, 0, 0)], v) for v in out.values())
        ref = ""
        if n <= 1500 and not sp.isspmatrix_bsr(M):
            xs = np.cos(np.arange(n, dtype=float))
            for _ in range(its): xs = sequential(A, xs, b, sweep)
            same = np.array_equal(xs, out[(0, 0, 0, 0)])
            ok = ok and same
            ref = " sequential loop equal: %s" % same
        fused, nseq = fused_sequences(A, b, rng, sp.isspmatrix_bsr(M))
        ok = ok and fused
        bad += 0 if ok else 1
        print("case %3d kind %d n %6d nnz/row %5.1f %-9s x%d %-4s all paths equal: %s%s  fused sequence of %d sweeps equal: %s" % (case, kind, n, A.nnz / max(n, 1), sweep, its, "bsr1" if sp.isspmatrix_bsr(M) else "csr", ok, ref, nseq, fused), flush=True)
    print("FAILED: %d" % bad if bad else "all %d cases equal" % ncase)
    return 1 if bad else 0


sys.exit(main())
