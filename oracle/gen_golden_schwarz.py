#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (development container only): tests/golden/schwarz.npz.

Runs the REFERENCE's overlapping Schwarz relaxation (its Python shim staged by ref_env.py on top of its
own native module built by oracle/Makefile) and records inputs and outputs as data:
  extract_subblocks / the gelss pseudo-inverses of schwarz_parameters / overlapping_schwarz_csr
  (forward, backward), relaxation.schwarz (symmetric, 2 iterations) on a seeded random system with
  the default subdomains (A's sparsity pattern) and with user subdomains, and the docstring example
  of relaxation.schwarz (10x10 Poisson, 10 iterations, ||b - A x|| = 0.126326160522).
Usage:  make -C oracle ref && python oracle/gen_golden_schwarz.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402
from gen_golden import random_system  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def main():
    pyamg = ref_env.stage()
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    import _amg_core as core
    from pyamg.relaxation import relaxation as rel
    from pyamg.util.linalg import norm
    out, cases = {}, []

    def rec(name, **arrs):
        cases.append(name)
        for k, v in arrs.items():
            out["%s__%s" % (name, k)] = np.asarray(v)

    rng = np.random.RandomState(4321)
    for tag, n, dens, user in (("pattern", 60, 0.08, False), ("user", 45, 0.12, True)):
        A = random_system(n, dens, seed=900 + n)
        x0 = rng.randn(n); b = rng.randn(n)
        if user:
            # overlapping windows of 5 consecutive unknowns, stride 3 (sorted, as the kernel requires)
            doms = [np.arange(s, min(s + 5, n)) for s in range(0, n - 1, 3)]
            Sp = np.concatenate(([0], np.cumsum([len(d) for d in doms]))).astype(np.intc)
            Sj = np.concatenate(doms).astype(np.intc)
        else:
            Sp = A.indptr.copy(); Sj = A.indices.copy()
        nsd = len(Sp) - 1
        Tp = np.zeros(nsd + 1, dtype=np.intc)
        Tp[1:] = np.cumsum((Sp[1:] - Sp[:-1]) ** 2)
        Tx_raw = np.zeros(Tp[-1])
        core.extract_subblocks(A.indptr, A.indices, A.data, Tx_raw, Tp, Sj, Sp, nsd, n)
        params = rel.schwarz_parameters(A, Sj if user else None, Sp if user else None)
        Tx = params[2].copy()
        assert np.array_equal(params[1], Sp) and np.array_equal(params[3], Tp)
        res = {}
        for nm, (rs, re, rt) in (("fwd", (0, nsd, 1)), ("bwd", (nsd - 1, -1, -1))):
            x = x0.copy()
            core.overlapping_schwarz_csr(A.indptr, A.indices, A.data, x, b, Tx, Tp, Sj, Sp, nsd, n, rs, re, rt)
            res[nm] = x
        xs = x0.copy()
        rel.schwarz(A, xs, b, iterations=2, subdomain=Sj if user else None, subdomain_ptr=Sp if user else None,
                    sweep="symmetric")
        rec("schwarz_" + tag, Ap=A.indptr, Aj=A.indices, Ax=A.data, x0=x0, b=b, Sj=Sj, Sp=Sp, Tp=Tp,
            Tx_raw=Tx_raw, Tx=Tx, x_fwd=res["fwd"], x_bwd=res["bwd"], x_sym2=xs)

    A = ref_env.poisson((10, 10))
    x = np.zeros((A.shape[0], 1)); b = np.ones((A.shape[0], 1))
    rel.schwarz(A, x, b, iterations=10)
    rec("schwarz_docstring", Ap=A.indptr, Aj=A.indices, Ax=A.data, x=np.ravel(x), resnorm=[norm(b - A * x)])
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "schwarz.npz"), **out)
    # whole hierarchies smoothed by Schwarz (the docstring's configuration, relaxation.py:233-241, on a
    # larger grid; and strength-based subdomains with a forward pre- / backward post-sweep)
    from gen_golden import gen_hier
    P = ref_env.poisson
    gen_hier(pyamg, "sa_schwarz_2d", P((24, 24)),
             lambda A, **kw: pyamg.smoothed_aggregation_solver(A, B=np.ones((A.shape[0], 1)), max_coarse=50, **kw),
             "schwarz", "schwarz", dict(tol=1e-8))
    gen_hier(pyamg, "sa_sbschwarz_3d", P((9, 9, 9)),
             lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=20, **kw),
             ("strength_based_schwarz", {"sweep": "forward", "iterations": 2}),
             ("strength_based_schwarz", {"sweep": "backward"}), dict(tol=1e-9))
    # the normal-equation smoothers as multigrid smoothers (smoothing.py:452-478)
    gen_hier(pyamg, "sa_gsne_2d", P((20, 20)),
             lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=20, **kw),
             ("gauss_seidel_ne", {"sweep": "symmetric", "omega": 1.0}),
             ("gauss_seidel_ne", {"sweep": "backward", "iterations": 2, "omega": 0.9}), dict(tol=1e-8, maxiter=40))
    gen_hier(pyamg, "sa_gsnr_2d", P((20, 20)),
             lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=20, **kw),
             ("gauss_seidel_nr", {"sweep": "symmetric"}),
             ("gauss_seidel_nr", {"sweep": "forward", "iterations": 2, "omega": 1.1}), dict(tol=1e-8, maxiter=40))
    gen_hier(pyamg, "sa_jacne_2d", P((20, 20)),
             lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=20, **kw),
             ("jacobi_ne", {"omega": 4.0 / 3.0, "iterations": 2}), ("jacobi_ne", {"omega": 4.0 / 3.0}),
             dict(tol=1e-8, maxiter=40))
    print("schwarz.npz: %d cases; docstring residual %.12f" % (len(cases), norm(b - A * x)))


if __name__ == "__main__":
    main()
