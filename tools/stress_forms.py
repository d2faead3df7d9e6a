#!/usr/bin/env python3
"""Randomised cross-check of the three operator forms (CSR stream / offset-pattern / stencil) through the
device-pointer API: random grid shapes (1-D .. 3-D, size-1 axes, sizes that are not multiples of the
row-block size), variable coefficients, random [owned | halo]-style rectangular extensions, random row
ranges.  Every result must equal scipy's product bit for bit."""
import sys, os, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pyamg_amd import _lib
from pyamg_amd.distributed import HipBackend
def run(seed, ncase):
    rng = np.random.RandomState(seed)
    be = HipBackend(0)
    L = _lib.lib()
    forms = {0: 0, 1: 0, 2: 0}
    for case in range(ncase):
        nd = rng.randint(1, 4)
        dims = [int(rng.choice([1, 2, 3, 7, 16, 33, 50, 127])) for _ in range(nd)]
        while np.prod(dims) < 1100:
            dims[rng.randint(nd)] *= 2
        T = []
        for d in dims:
            k = min(d - 1, int(rng.choice([1, 1, 1, 2])))
            offs = list(range(-k, k + 1))
            T.append(sp.diags([np.ones(d - abs(o)) for o in offs], offs) if d > 1 else sp.identity(1))
        S = T[0]
        for t in T[1:]:
            S = sp.kron(S, t)
        S = sp.csr_matrix(S); S.sort_indices()
        n = S.shape[0]
        A = sp.csr_matrix((rng.randn(S.nnz), S.indices.copy(), S.indptr.copy()), shape=S.shape)
        if rng.rand() < 0.3:                       # drop random entries: more row patterns
            A.data[rng.rand(A.nnz) < 0.05] = 0.0
            A.eliminate_zeros()
        ncols = n
        if rng.rand() < 0.5:                       # halo-like extension: a few rows reference extra columns
            h = int(rng.randint(1, 200))
            ncols = n + h
            rows = np.concatenate((np.arange(min(h, n)), n - 1 - np.arange(min(h, n))))
            cols = n + rng.randint(0, h, size=len(rows))
            E = sp.csr_matrix((rng.randn(len(rows)), (rows, cols)), shape=(n, ncols))
            A = sp.csr_matrix(sp.hstack([A, sp.csr_matrix((n, h))]) + E)
            if rng.rand() < 0.5:
                A.sort_indices()
        A.indptr = A.indptr.astype(np.intc); A.indices = A.indices.astype(np.intc)
        hm = be.mat(n, ncols, A.indptr, A.indices, A.data)
        forms[L.amg_mat_form(hm)] += 1
        x = torch.from_numpy(rng.randn(ncols)).cuda()
        b = torch.from_numpy(rng.randn(n)).cuda()
        ref = A * x.cpu().numpy()
        for st in (1, 0):
            L.amg_set_stencil_form(st)
            y = torch.full((n,), np.nan, dtype=torch.float64, device="cuda")
            be.apply(hm, 0, x, None, None, y, None, 0.0)
            lo, hi = sorted(rng.randint(0, n + 1, size=2))
            z = torch.zeros(n, dtype=torch.float64, device="cuda")
            be.apply_rows(hm, 2, lo, hi, x, b, None, z, None, 0.0)
            torch.cuda.synchronize()
            assert np.array_equal(y.cpu().numpy(), ref), (case, dims, st, "matvec")
            zr = np.zeros(n); zr[lo:hi] = (b.cpu().numpy() - ref)[lo:hi]
            assert np.array_equal(z.cpu().numpy(), zr), (case, dims, st, "residual rows", lo, hi)
        L.amg_set_stencil_form(1)
    be.close()
    return forms


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    forms = run(seed, ncase)
    print("seed %d: %d cases ok; forms used: csr %d, pattern %d, stencil %d" % (seed, ncase, forms[0], forms[1], forms[2]))
