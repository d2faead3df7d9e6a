#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (development container only): generate tests/golden/*.npz.

Runs the REFERENCE itself -- its Python (scratch copy staged by ref_env.py) on
top of its own native module (oracle/_ref/_amg_core.so, built by oracle/Makefile
from /root/reference/pyamg/amg_core/amg_core_wrap.cxx) -- and records inputs
and outputs as data:

  tests/golden/kernels.npz       inputs/outputs of the reference's amg_core
                                 relaxation kernels on seeded random systems
  tests/golden/hier_<case>.npz   a whole hierarchy built by the reference's
                                 ruge_stuben_solver / smoothed_aggregation_solver
                                 (A_l, P_l, R_l, smoother constants, the cached
                                 coarse pseudo-inverse) + b, x0, the residual
                                 history and iterates of multilevel_solver.solve()

Nothing but numbers is written; no reference source text enters the repo.
Usage:  make -C oracle ref && python oracle/gen_golden.py
"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sps

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_env  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


# --------------------------------------------------------------------------- kernels
def random_system(n, density, seed, bs=1, spd_shift=True):
    rng = np.random.RandomState(seed)
    nb = n // bs
    S = sps.random(nb, nb, density=density, random_state=rng, format="csr")
    S = S + S.T + sps.identity(nb) * 0.0
    S = sps.csr_matrix(S)
    S.data[:] = 1.0
    if bs == 1:
        A = S.copy().astype(np.float64)
        A.data = rng.randn(A.nnz)
        A = A + sps.diags(np.abs(A).sum(axis=1).A1 + 1.0 if hasattr(np.abs(A).sum(axis=1), "A1")
                          else np.asarray(np.abs(A).sum(axis=1)).ravel() + 1.0)
        A = sps.csr_matrix(A)
    else:
        S = (S + sps.identity(nb)).tocsr()
        S.sort_indices()
        data = rng.randn(S.nnz, bs, bs)
        A = sps.bsr_matrix((data, S.indices.copy(), S.indptr.copy()), shape=(n, n))
        # make block diagonal dominant
        D = sps.block_diag([np.eye(bs) * (4.0 * bs * (S.indptr[i + 1] - S.indptr[i]))
                            for i in range(nb)], format="bsr")
        A = (A + D).tobsr((bs, bs))
    A.sort_indices()
    A.indices = A.indices.astype(np.intc)
    A.indptr = A.indptr.astype(np.intc)
    return A


def gen_kernels(core):
    out = {}
    cases = []

    def rec(name, **arrs):
        cases.append(name)
        for k, v in arrs.items():
            out["%s__%s" % (name, k)] = np.asarray(v)

    rng = np.random.RandomState(1234)
    for tag, n, dens in (("a", 37, 0.15), ("b", 200, 0.03)):
        A = random_system(n, dens, seed=n)
        if tag == "a":   # a zero diagonal exercises the silent skip (relaxation.h:58-60)
            A = A.tolil(); A[5, 5] = 0.0; A = sps.csr_matrix(A); A.eliminate_zeros()
            A.sort_indices(); A.indices = A.indices.astype(np.intc); A.indptr = A.indptr.astype(np.intc)
        x0 = rng.randn(n); b = rng.randn(n)
        base = dict(Ap=A.indptr, Aj=A.indices, Ax=A.data, x0=x0, b=b)
        for nm, (rs, re, rt) in (("fwd", (0, n, 1)), ("bwd", (n - 1, -1, -1)),
                                 ("part", (3, 3 + 2 * ((n - 7) // 2), 2))):
            x = x0.copy()
            core.gauss_seidel(A.indptr, A.indices, A.data, x, b, rs, re, rt)
            rec("gauss_seidel_%s_%s" % (tag, nm), sweep=[rs, re, rt], x=x, **base)
        for om in (1.0, 0.7):
            x = x0.copy(); temp = np.zeros(n)
            core.jacobi(A.indptr, A.indices, A.data, x, b, temp, 0, n, 1, np.array([om]))
            rec("jacobi_%s_om%g" % (tag, om), omega=[om], sweep=[0, n, 1], x=x, **base)
        Id = rng.permutation(n)[: n // 2].astype(np.intc)
        for nm, (rs, re, rt) in (("fwd", (0, len(Id), 1)), ("bwd", (len(Id) - 1, -1, -1))):
            x = x0.copy()
            core.gauss_seidel_indexed(A.indptr, A.indices, A.data, x, b, Id, rs, re, rt)
            rec("gauss_seidel_indexed_%s_%s" % (tag, nm), Id=Id, sweep=[rs, re, rt], x=x, **base)
        # normal-equation kernels
        Dne = 1.0 / np.asarray(A.multiply(A).sum(axis=1)).ravel()
        for nm, (rs, re, rt) in (("fwd", (0, n, 1)), ("bwd", (n - 1, -1, -1))):
            x = x0.copy()
            core.gauss_seidel_ne(A.indptr, A.indices, A.data, x, b, rs, re, rt, Dne, 0.9)
            rec("gauss_seidel_ne_%s_%s" % (tag, nm), Tx=Dne, omega=[0.9], sweep=[rs, re, rt], x=x, **base)
        Ac = sps.csc_matrix(A); Ac.sort_indices()
        Ac.indices = Ac.indices.astype(np.intc); Ac.indptr = Ac.indptr.astype(np.intc)
        Dnr = 1.0 / np.asarray(Ac.multiply(Ac).sum(axis=0)).ravel()
        for nm, (rs, re, rt) in (("fwd", (0, n, 1)), ("bwd", (n - 1, -1, -1))):
            x = x0.copy(); z = b - A * x0; z0 = z.copy()
            core.gauss_seidel_nr(Ac.indptr, Ac.indices, Ac.data, x, z, rs, re, rt, Dnr, 1.1)
            rec("gauss_seidel_nr_%s_%s" % (tag, nm), Ap=Ac.indptr, Aj=Ac.indices, Ax=Ac.data,
                x0=x0, z0=z0, Tx=Dnr, omega=[1.1], sweep=[rs, re, rt], x=x, z=z)
        x = x0.copy(); temp = np.zeros(n); delta = (b - A * x0) * Dne
        core.jacobi_ne(A.indptr, A.indices, A.data, x, b, delta, temp, 0, n, 1, np.array([0.8]))
        rec("jacobi_ne_%s" % tag, Tx=delta, omega=[0.8], sweep=[0, n, 1], x=x, **base)

    for bs in (1, 2, 3, 4):
        n = 24 * bs
        A = random_system(n, 0.2, seed=100 + bs, bs=bs) if bs > 1 else \
            sps.bsr_matrix(random_system(n, 0.2, seed=100), blocksize=(1, 1))
        A.sort_indices()
        Ap = A.indptr.astype(np.intc); Aj = A.indices.astype(np.intc); Ax = np.ravel(A.data).copy()
        nb = n // bs
        x0 = rng.randn(n); b = rng.randn(n)
        base = dict(Ap=Ap, Aj=Aj, Ax=Ax, x0=x0, b=b, blocksize=[bs])
        for nm, (rs, re, rt) in (("fwd", (0, nb, 1)), ("bwd", (nb - 1, -1, -1))):
            x = x0.copy()
            core.bsr_gauss_seidel(Ap, Aj, Ax, x, b, rs, re, rt, bs)
            rec("bsr_gauss_seidel_bs%d_%s" % (bs, nm), sweep=[rs, re, rt], x=x, **base)
        x = x0.copy(); temp = np.zeros(n)
        core.bsr_jacobi(Ap, Aj, Ax, x, b, temp, 0, nb, 1, bs, np.array([0.6]))
        rec("bsr_jacobi_bs%d" % bs, omega=[0.6], sweep=[0, nb, 1], x=x, **base)
        # block smoothers with the inverse diagonal blocks
        Acsr = sps.csr_matrix(A)
        Dinv = np.zeros((nb, bs, bs))
        Ad = Acsr.toarray()
        for i in range(nb):
            Dinv[i] = np.linalg.inv(Ad[i * bs:(i + 1) * bs, i * bs:(i + 1) * bs])
        x = x0.copy(); temp = np.zeros(n)
        core.block_jacobi(Ap, Aj, Ax, x, b, np.ravel(Dinv), temp, 0, nb, 1, np.array([0.8]), bs)
        rec("block_jacobi_bs%d" % bs, Dinv=np.ravel(Dinv), omega=[0.8], sweep=[0, nb, 1], x=x, **base)
        for nm, (rs, re, rt) in (("fwd", (0, nb, 1)), ("bwd", (nb - 1, -1, -1))):
            x = x0.copy()
            core.block_gauss_seidel(Ap, Aj, Ax, x, b, np.ravel(Dinv), rs, re, rt, bs)
            rec("block_gauss_seidel_bs%d_%s" % (bs, nm), Dinv=np.ravel(Dinv), sweep=[rs, re, rt],
                x=x, **base)

    # scipy SpMV (third-party arithmetic at the reference's call sites)
    A = random_system(150, 0.05, seed=7)
    x = rng.randn(150)
    rec("csr_matvec", Ap=A.indptr, Aj=A.indices, Ax=A.data, x=x, y=A * x, shape=A.shape)
    for (R, C) in ((2, 3), (3, 3), (1, 1)):
        nbr, nbc = 20, 17
        S = sps.random(nbr, nbc, density=0.3, random_state=np.random.RandomState(R * 10 + C), format="csr")
        S.sort_indices()
        data = rng.randn(S.nnz, R, C)
        B = sps.bsr_matrix((data, S.indices.astype(np.intc), S.indptr.astype(np.intc)),
                           shape=(nbr * R, nbc * C))
        x = rng.randn(nbc * C)
        rec("bsr_matvec_%dx%d" % (R, C), Ap=B.indptr, Aj=B.indices, Ax=np.ravel(B.data), x=x, y=B * x,
            shape=B.shape, blocksize=[R, C])

    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "kernels.npz"), **out)
    print("kernels.npz: %d cases" % len(cases))


# --------------------------------------------------------------------------- setup kernels
def gen_setup_kernels(core):
    """inputs/outputs of the reference's native SETUP kernels (amg_core) on seeded random graphs:
    standard_aggregation, classical_strength_of_connection, rs_cf_splitting,
    rs_direct_interpolation_pass1/2, fit_candidates"""
    out = {}
    cases = []

    def rec(name, **arrs):
        cases.append(name)
        for k, v in arrs.items():
            out["%s__%s" % (name, k)] = np.asarray(v)

    for tag, n, dens, seed in (("g40", 40, 0.08, 1), ("g300", 300, 0.012, 2), ("g1000", 1000, 0.004, 3),
                                ("path", 9, None, 0), ("iso", 12, 0.05, 5)):
        if dens is None:
            A = sps.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
        else:
            A = random_system(n, dens, seed=seed)
        A = sps.csr_matrix(A); A.sort_indices()
        Ap, Aj, Ax = A.indptr.astype(np.intc), A.indices.astype(np.intc), A.data.astype(np.float64)
        # aggregation on the pattern of A
        x = np.empty(n, dtype=np.intc); y = np.empty(n, dtype=np.intc)
        na = core.standard_aggregation(n, Ap, Aj, x, y)
        rec("standard_aggregation_" + tag, Ap=Ap, Aj=Aj, n=[n], agg=x, roots=y[:na], count=[na])
        # classical strength
        for theta in (0.0, 0.25, 0.6):
            Sp = np.empty_like(Ap); Sj = np.empty_like(Aj); Sx = np.empty_like(Ax)
            core.classical_strength_of_connection(n, theta, Ap, Aj, Ax, Sp, Sj, Sx)
            nnz = Sp[-1]
            rec("classical_strength_%s_%g" % (tag, theta), Ap=Ap, Aj=Aj, Ax=Ax, theta=[theta], Sp=Sp, Sj=Sj[:nnz], Sx=Sx[:nnz])
        # RS splitting + direct interpolation on the theta = 0.25 strength graph
        S = sps.csr_matrix((Sx[:nnz], Sj[:nnz], Sp), shape=A.shape)    # theta = 0.6 from the loop above
        Sc = S.tocoo(); m = Sc.row != Sc.col
        S0 = sps.coo_matrix((Sc.data[m], (Sc.row[m], Sc.col[m])), shape=S.shape).tocsr()
        T = S0.T.tocsr()
        spl = np.empty(n, dtype=np.intc)
        core.rs_cf_splitting(n, S0.indptr.astype(np.intc), S0.indices.astype(np.intc), T.indptr.astype(np.intc),
                             T.indices.astype(np.intc), spl)
        rec("rs_cf_splitting_" + tag, Sp=S0.indptr.astype(np.intc), Sj=S0.indices.astype(np.intc),
            Tp=T.indptr.astype(np.intc), Tj=T.indices.astype(np.intc), splitting=spl)
        C = S.copy(); C.data[:] = 1.0; C = sps.csr_matrix(C.multiply(A)); C.sort_indices()
        Cp, Cj, Cx = C.indptr.astype(np.intc), C.indices.astype(np.intc), C.data.astype(np.float64)
        Pp = np.empty(n + 1, dtype=np.intc)
        core.rs_direct_interpolation_pass1(n, Cp, Cj, spl, Pp)
        Pj = np.empty(Pp[-1], dtype=np.intc); Px = np.empty(Pp[-1], dtype=np.float64)
        core.rs_direct_interpolation_pass2(n, Ap, Aj, Ax, Cp, Cj, Cx, spl, Pp, Pj, Px)
        rec("rs_direct_interpolation_" + tag, Ap=Ap, Aj=Aj, Ax=Ax, Cp=Cp, Cj=Cj, Cx=Cx, splitting=spl, Pp=Pp, Pj=Pj, Px=Px)
    out["cases"] = np.array(cases)
    np.savez_compressed(os.path.join(OUT, "setup_kernels.npz"), **out)
    print("setup_kernels.npz: %d cases" % len(cases))


# --------------------------------------------------------------------------- hierarchies
def put_mat(out, key, M):
    if sps.isspmatrix_bsr(M):
        out[key + "_bs"] = np.array(M.blocksize, dtype=np.int64)
        out[key + "_data"] = np.ravel(M.data).copy()
    else:
        M = sps.csr_matrix(M)
        out[key + "_bs"] = np.array([0, 0], dtype=np.int64)
        out[key + "_data"] = M.data.copy()
    out[key + "_indptr"] = M.indptr.astype(np.intc)
    out[key + "_indices"] = M.indices.astype(np.intc)
    out[key + "_shape"] = np.array(M.shape, dtype=np.int64)


def closure_vars(fn):
    if fn.__closure__ is None:
        return {}
    return {n: c.cell_contents for n, c in zip(fn.__code__.co_freevars, fn.__closure__)}


def smoother_desc(out, key, spec, fn, lvl):
    """Record the constants the reference baked into the smoother closure
    (pyamg/relaxation/smoothing.py:320-515)."""
    name = spec[0] if isinstance(spec, tuple) else spec
    cv = closure_vars(fn)
    d = {"name": str(name)}
    for k in ("iterations", "sweep", "blocksize"):
        if k in cv:
            d[k] = cv[k]
    if str(name) in ("gmres", "cg", "cgne", "cgnr"):
        # smoothing.py:481-509: Krylov iterations as relaxation; tol / maxiter / restrt are the closure's constants
        d["method"] = str(name)
        d["name"] = "krylov"
        d["tol"] = float(cv["tol"])
        d["maxiter"] = cv["maxiter"]
        d["restrt"] = cv.get("restrt")
    if "omega" in cv:
        d["omega"] = float(np.ravel(cv["omega"])[0])
    if "coefficients" in cv:
        d["coefficients"] = [float(c) for c in np.ravel(cv["coefficients"])]
    if "Dinv" in cv and cv["Dinv"] is not None:
        out[key + "_Dinv"] = np.ravel(np.asarray(cv["Dinv"], dtype=np.float64)).copy()
        d["has_Dinv"] = True
    if str(name) in ("schwarz", "strength_based_schwarz"):
        # smoothing.py:335-348: subdomains and their inverted diagonal blocks
        d["name"] = "schwarz"
        for k in ("subdomain", "subdomain_ptr", "inv_subblock_ptr"):
            out[key + "_" + k] = np.asarray(cv[k], dtype=np.intc).copy()
        out[key + "_inv_subblock"] = np.asarray(cv["inv_subblock"], dtype=np.float64).copy()
        d["has_schwarz"] = True
    return d


def gen_hier(pyamg, name, A, build, pre, post, solve_kw, B=None, x0_random=False, seed=0):
    np.random.seed(seed)
    kw = dict(presmoother=pre, postsmoother=post)
    if B is not None:
        kw["B"] = B
    ml = build(A, **kw)
    n = A.shape[0]
    b = np.random.rand(n)
    x0 = np.random.rand(n) if x0_random else None
    iterates = []
    res = []
    x = ml.solve(b, x0=x0, residuals=res, callback=lambda xk: iterates.append(np.array(xk, copy=True)), **solve_kw)
    out = {}
    meta = {"name": name, "nlevels": len(ml.levels), "cycle": solve_kw.get("cycle", "V"),
            "tol": solve_kw.get("tol", 1e-5), "maxiter": solve_kw.get("maxiter", 100),
            "accel": solve_kw.get("accel"), "levels": []}
    pre_l = pre if isinstance(pre, list) else [pre]
    post_l = post if isinstance(post, list) else [post]
    for i, lvl in enumerate(ml.levels):
        put_mat(out, "A%d" % i, lvl.A)
        if i < len(ml.levels) - 1:
            put_mat(out, "P%d" % i, lvl.P)
            put_mat(out, "R%d" % i, lvl.R)
            ps = pre_l[min(i, len(pre_l) - 1)]
            qs = post_l[min(i, len(post_l) - 1)]
            meta["levels"].append({
                "pre": smoother_desc(out, "pre%d" % i, ps, lvl.presmoother, lvl),
                "post": smoother_desc(out, "post%d" % i, qs, lvl.postsmoother, lvl)})
    if B is not None:
        out["B0"] = np.asarray(B, dtype=np.float64)          # near-null-space candidates handed to the setup
    out["coarse_pinv"] = np.asarray(ml.coarse_solver.P, dtype=np.float64)
    out["b"] = b
    out["x0"] = np.zeros(n) if x0 is None else x0
    out["x"] = np.asarray(x)
    out["residuals"] = np.array(res)
    iterates = [np.asarray(v, dtype=np.float64) for v in iterates if np.ndim(v) == 1 and np.size(v) == n]
    if not iterates:                       # Krylov callbacks that only report residual norms
        iterates = [np.asarray(x)]
    out["x_iter1"] = iterates[0]
    out["x_iter2"] = iterates[1] if len(iterates) > 1 else iterates[0]
    out["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(OUT, "hier_%s.npz" % name)
    np.savez_compressed(path, **out)
    print("%-22s levels=%d iters=%d  r0=%.3e  rN=%.3e  %6.0f KB" %
          (name, len(ml.levels), len(res) - 1, res[0], res[-1], os.path.getsize(path) / 1024))


def main():
    os.makedirs(OUT, exist_ok=True)
    pyamg = ref_env.stage()
    sys.path.insert(0, os.path.join(HERE, "_ref"))
    import _amg_core as core
    gen_kernels(core)
    gen_setup_kernels(core)

    P = ref_env.poisson
    rs = lambda A, **kw: pyamg.ruge_stuben_solver(A, max_coarse=40, **kw)
    sa = lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=30, **kw)

    # C1-like: RS + symmetric GS (the README configuration, small grid)
    gen_hier(pyamg, "rs_gs_2d", P((40, 40)), rs, ("gauss_seidel", {"sweep": "symmetric"}),
             ("gauss_seidel", {"sweep": "symmetric"}), dict(tol=1e-10))
    # C2-like: SA + weighted Jacobi omega=4/3 (CSR level 0, BSR(1,1) coarse levels)
    gen_hier(pyamg, "sa_jacobi_2d", P((48, 48)), sa, ("jacobi", {"omega": 4.0 / 3.0}),
             ("jacobi", {"omega": 4.0 / 3.0}), dict(tol=1e-10))
    # C3-like: SA + Chebyshev degree 2
    gen_hier(pyamg, "sa_cheb2_3d", P((16, 16, 16)), sa, ("chebyshev", {"degree": 2}),
             ("chebyshev", {"degree": 2}), dict(tol=1e-10))
    # SA defaults: block_gauss_seidel symmetric -> plain GS (bs == 1), bsr_gauss_seidel on coarse levels
    gen_hier(pyamg, "sa_gs_3d", P((12, 12, 12)), sa,
             ("block_gauss_seidel", {"sweep": "symmetric"}),
             ("block_gauss_seidel", {"sweep": "symmetric"}), dict(tol=1e-10))
    # mixed smoothers, W cycle, non-zero initial guess
    gen_hier(pyamg, "sa_mixed_W_2d", P((30, 30)), sa,
             [("sor", {"omega": 1.2, "sweep": "backward"}), ("jacobi", {"omega": 1.0, "iterations": 2})],
             [("chebyshev", {"degree": 3, "iterations": 2}), ("richardson", {"omega": 1.0})],
             dict(tol=1e-9, cycle="W"), x0_random=True)
    # F cycle, forward GS / None
    gen_hier(pyamg, "rs_F_2d", P((32, 32)), lambda A, **kw: pyamg.ruge_stuben_solver(A, max_coarse=10, **kw),
             ("gauss_seidel", {"sweep": "forward", "iterations": 2}), None,
             dict(tol=1e-9, cycle="F"))
    # AMLI cycle
    gen_hier(pyamg, "sa_amli_2d", P((24, 24)), lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=10, **kw),
             ("jacobi", {"omega": 4.0 / 3.0}), ("jacobi", {"omega": 4.0 / 3.0}),
             dict(tol=1e-9, cycle="AMLI", maxiter=8))
    # BSR: 2D linear elasticity (bs 2 on level 0, bs 3 below), SA default block GS
    from pyamg.gallery import linear_elasticity
    A, B = linear_elasticity((12, 12))
    gen_hier(pyamg, "elas_bgs_2d", A, lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=10, **kw),
             ("block_gauss_seidel", {"sweep": "symmetric"}),
             ("block_gauss_seidel", {"sweep": "symmetric"}), dict(tol=1e-10), B=B)
    gen_hier(pyamg, "elas_bjac_2d", A, lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=10, **kw),
             ("block_jacobi", {"omega": 4.0 / 3.0}), ("jacobi", {"omega": 4.0 / 3.0}),
             dict(tol=1e-8), B=B)
    gen_hier(pyamg, "elas_gs_2d", A, lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=10, **kw),
             ("gauss_seidel", {"sweep": "symmetric"}), ("gauss_seidel", {"sweep": "backward"}),
             dict(tol=1e-8), B=B)
    # C5-like: native block size 3 on the fine level (3 coupled diffusion unknowns per node)
    M3 = np.array([[4.0, -1.0, 0.5], [-1.0, 3.0, -0.5], [0.5, -0.5, 2.0]])
    A3 = sps.kron(P((10, 10)), M3).tobsr((3, 3))
    A3.sort_indices()
    B3 = np.kron(np.ones((100, 1)), np.eye(3))
    gen_hier(pyamg, "bs3_bgs_2d", A3, lambda A, **kw: pyamg.smoothed_aggregation_solver(A, max_coarse=10, **kw),
             ("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3}),
             ("block_gauss_seidel", {"sweep": "symmetric", "blocksize": 3}), dict(tol=1e-10), B=B3)


if __name__ == "__main__":
    main()
