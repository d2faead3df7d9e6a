#!/usr/bin/env python3
"""Randomised cross-check of the Gauss-Seidel sweep paths: chained sweeps (LDS hand-off / memory hand-off), long-row
chains, level launches with and without entry ranges in the kernel arguments -- against one plain launch per level,
bit for bit, and against the sequential reference loop (scipy-free restatement below) on the small cases.
Matrices: banded, random sparse, 2-D / 3-D grid operators with random coefficients, Galerkin-like products (long rows),
rows without a diagonal entry, zero diagonals, empty rows.  usage: stress_gs.py [seed=0] [cases=60]"""
import sys, os, numpy as np, scipy.sparse as sp
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
        for chain, hint in ((0, 0), (2, 1), (1, 1), (0, 1), (2, 0)):
            L.amg_set_gs_chain(chain); L.amg_set_gs_level_hint(hint)
            x = np.cos(np.arange(n, dtype=float))
            relaxation.gauss_seidel(M, x, b, iterations=2, sweep=sweep)
            out[(chain, hint)] = x
        L.amg_set_gs_chain(2); L.amg_set_gs_level_hint(1)
        ok = all(np.array_equal(out[(0, 0)], v) for v in out.values())
        ref = ""
        if n <= 1500 and not sp.isspmatrix_bsr(M):
            xs = sequential(A, np.cos(np.arange(n, dtype=float)), b, sweep)
            xs = sequential(A, xs, b, sweep)
            same = np.array_equal(xs, out[(0, 0)])
            ok = ok and same
            ref = " sequential loop equal: %s" % same
        bad += 0 if ok else 1
        print("case %3d kind %d n %6d nnz/row %5.1f %-9s %-4s all paths equal: %s%s" % (case, kind, n, A.nnz / max(n, 1), sweep, "bsr1" if sp.isspmatrix_bsr(M) else "csr", ok, ref), flush=True)
    print("FAILED: %d" % bad if bad else "all %d cases equal" % ncase)
    return 1 if bad else 0


sys.exit(main())
