#!/usr/bin/env python3
"""Randomised cross-check of block Gauss-Seidel (relaxation.h:756-810): the dataflow sweep (every (blocks per lane, lanes per
scalar row) instantiation: block rows of 1 .. 120 off-diagonal blocks; three look-ahead settings) against the level-scheduled
sweep and the CPU oracle, 2x2 and 3x3 blocks, forward and backward, random sparsity incl. empty block rows and block rows
without a diagonal block.  usage: stress_bgs.py [seed=0] [cases=40]"""
import sys, os, numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from pyamg_amd import _lib, amg_core
L = _lib.lib()
O = oracle_lib.load()


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rng = np.random.RandomState(seed)
    bad = 0
    for case in range(ncase):
        bs = int(rng.choice([2, 3]))
        nb = int(rng.randint(40, 2500))
        per_row = int(rng.choice([2, 5, 8, 12, 15, 16, 30, 60, 110]))
        dens = min(0.9, per_row / nb)
        S = sp.random(nb, nb, density=dens, random_state=rng, format="csr") + sp.identity(nb)
        S = sp.csr_matrix(S); S.sort_indices()
        if rng.rand() < 0.3 and nb > 10:                       # an empty block row and one without its diagonal block
            Sl = S.tolil(); i, j = rng.choice(nb, 2, replace=False); Sl[i, :] = 0.0; Sl[j, j] = 0.0; S = sp.csr_matrix(Sl); S.eliminate_zeros()
        blocks = rng.randn(S.nnz, bs, bs)
        A = sp.bsr_matrix((blocks, S.indices.astype(np.intc), S.indptr.astype(np.intc)), shape=(nb * bs, nb * bs))
        n = nb * bs
        b = rng.rand(n)
        Dinv = rng.randn(nb, bs, bs)
        Ap, Aj, Ax = A.indptr.astype(np.intc), A.indices.astype(np.intc), np.ravel(A.data).copy()
        ok = True
        for (rs, re, rt) in ((0, nb, 1), (nb - 1, -1, -1)):
            out = {}
            for flow, la in ((0, 0), (2, 0), (2, 1), (2, 30)):
                L.amg_set_gs_flow(flow); L.amg_set_gs_flow_lookahead(la)
                x = np.cos(np.arange(n, dtype=float))
                amg_core.block_gauss_seidel(Ap, Aj, Ax, x, b, np.ravel(Dinv), rs, re, rt, bs)
                out[(flow, la)] = x
            xo = np.cos(np.arange(n, dtype=float))
            O.oracle_block_gauss_seidel(oracle_lib.ip(Ap), oracle_lib.ip(Aj), oracle_lib.dp(Ax), oracle_lib.dp(xo), oracle_lib.dp(b),
                                        oracle_lib.dp(np.ravel(Dinv).copy()), rs, re, rt, bs)
            ok = ok and all(np.array_equal(xo, v) for v in out.values())
        L.amg_set_gs_flow(1); L.amg_set_gs_flow_lookahead(0)
        ok = ok and L.amg_gs_flow_status() == 0
        bad += 0 if ok else 1
        print("case %3d bs %d block rows %5d blocks/row %6.1f (longest %3d): all paths equal the oracle: %s" %
              (case, bs, nb, S.nnz / nb, int(np.diff(S.indptr).max()), ok), flush=True)
    print("FAILED: %d" % bad if bad else "all %d cases equal" % ncase)
    return 1 if bad else 0


sys.exit(main())
