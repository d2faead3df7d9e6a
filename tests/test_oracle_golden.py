"""CPU: pin the oracle (our C restatement) against the reference.

 * kernels.npz   -- outputs of the reference's own _amg_core kernels: BIT-EXACT.
 * hier_*.npz    -- residual histories / iterates of the reference's own
                    multilevel_solver.solve(): <= 1e-12 relative (the reference's
                    norm and coarse solve go through BLAS, whose summation order
                    is unspecified; everything else is bit-exact).
 * known-answer vectors of pyamg/relaxation/tests/test_relaxation.py, literal.
"""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sps

import golden_io
import oracle_lib
from oracle_lib import dp, ip

KERNELS = golden_io.load_kernels()


def run_kernel(lib, name, c):
    Ap = np.ascontiguousarray(c["Ap"], dtype=np.intc)
    Aj = np.ascontiguousarray(c["Aj"], dtype=np.intc)
    Ax = np.ascontiguousarray(c["Ax"], dtype=np.float64)
    if name.startswith("csr_matvec"):
        y = np.zeros(int(c["shape"][0]))
        lib.oracle_csr_matvec(len(y), ip(Ap), ip(Aj), dp(Ax), dp(np.ascontiguousarray(c["x"])), dp(y))
        return {"y": y}
    if name.startswith("bsr_matvec"):
        R, Cc = (int(v) for v in c["blocksize"])
        y = np.zeros(int(c["shape"][0]))
        lib.oracle_bsr_matvec(len(y) // R, R, Cc, ip(Ap), ip(Aj), dp(Ax),
                              dp(np.ascontiguousarray(c["x"])), dp(y))
        return {"y": y}
    x = c["x0"].copy()
    rs, re, rt = (int(v) for v in c["sweep"])
    n = len(x)
    if name.startswith("gauss_seidel_indexed"):
        Id = np.ascontiguousarray(c["Id"], dtype=np.intc)
        lib.oracle_gauss_seidel_indexed(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), ip(Id), rs, re, rt)
    elif name.startswith("gauss_seidel_ne"):
        lib.oracle_gauss_seidel_ne(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), rs, re, rt,
                                   dp(c["Tx"].copy()), float(c["omega"][0]))
    elif name.startswith("gauss_seidel_nr"):
        z = c["z0"].copy()
        lib.oracle_gauss_seidel_nr(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(z), rs, re, rt,
                                   dp(c["Tx"].copy()), float(c["omega"][0]))
        return {"x": x, "z": z}
    elif name.startswith("gauss_seidel"):
        lib.oracle_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), rs, re, rt)
    elif name.startswith("jacobi_ne"):
        temp = np.zeros(n)
        lib.oracle_jacobi_ne(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), dp(c["Tx"].copy()),
                             dp(temp), rs, re, rt, dp(c["omega"].astype(np.float64)))
    elif name.startswith("jacobi"):
        temp = np.zeros(n)
        lib.oracle_jacobi(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), dp(temp), rs, re, rt,
                          dp(c["omega"].astype(np.float64)))
    elif name.startswith("bsr_gauss_seidel"):
        lib.oracle_bsr_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), rs, re, rt,
                                    int(c["blocksize"][0]))
    elif name.startswith("bsr_jacobi"):
        temp = np.zeros(n)
        lib.oracle_bsr_jacobi(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), dp(temp), rs, re, rt,
                              int(c["blocksize"][0]), dp(c["omega"].astype(np.float64)))
    elif name.startswith("block_jacobi"):
        temp = np.zeros(n)
        lib.oracle_block_jacobi(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), dp(c["Dinv"].copy()),
                                dp(temp), rs, re, rt, dp(c["omega"].astype(np.float64)),
                                int(c["blocksize"][0]))
    elif name.startswith("block_gauss_seidel"):
        lib.oracle_block_gauss_seidel(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()),
                                      dp(c["Dinv"].copy()), rs, re, rt, int(c["blocksize"][0]))
    else:
        raise KeyError(name)
    return {"x": x}


@pytest.mark.parametrize("name", sorted(KERNELS))
def test_kernel_bit_exact_vs_reference(oracle, name):
    c = KERNELS[name]
    out = run_kernel(oracle, name, c)
    for k, v in out.items():
        assert np.array_equal(v, c[k]), "%s: %s differs from the reference (max |d| = %g)" % (
            name, k, np.abs(v - c[k]).max())


def build_oracle_hier(g, **kw):
    return oracle_lib.Hierarchy(g["levels"], g["coarse_pinv"], **kw)


@pytest.mark.parametrize("case", golden_io.hier_cases())
def test_solve_matches_reference_history(case):
    g = golden_io.load_hier(case)
    m = g["meta"]
    H = build_oracle_hier(g)
    x, res = H.solve(g["b"], x0=g["x0"], tol=m["tol"], maxiter=m["maxiter"], cycle=m["cycle"])
    ref = g["residuals"]
    # 1e-12 relative to each residual norm (north_star) + a quarter of the fp64 evaluation floor
    golden_io.assert_history(res, ref, g["levels"][0]["A"], g["x"], g["b"])
    scale = np.linalg.norm(g["x"])
    assert np.linalg.norm(x - g["x"]) <= 1e-12 * scale
    # first iterate
    x1 = g["x0"].copy()
    H.cycle(x1, np.ascontiguousarray(g["b"]), m["cycle"])
    assert np.linalg.norm(x1 - g["x_iter1"]) <= 1e-13 * np.linalg.norm(g["x_iter1"])


def test_duplicate_prolongation_does_not_change_result():
    g = golden_io.load_hier("sa_jacobi_2d")
    xa, ra = build_oracle_hier(g).solve(g["b"], tol=1e-10)
    xb, rb = build_oracle_hier(g, dup_prolong=True).solve(g["b"], tol=1e-10)
    assert np.array_equal(xa, xb) and np.array_equal(ra, rb)


# ---------------------------------------------------------------------------
# Known-answer vectors of the reference's own tests, as literal data
# (pyamg/relaxation/tests/test_relaxation.py:139-188 jacobi, :291-353 gauss_seidel,
#  :355-402 gauss_seidel_indexed, :106-137 polynomial).
# ---------------------------------------------------------------------------
def poisson1d(n):
    return sps.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")


def _relax(oracle, A, desc, x, b):
    keep = []
    m = oracle_lib.make_mat(A, keep)
    s = oracle_lib.make_smoother(desc, A, keep)
    oracle.oracle_relax(C.byref(m), C.byref(s), dp(x), dp(b))


def test_kat_jacobi(oracle):
    # test_relaxation.py:139-188
    A = poisson1d(1); x = np.arange(1.0); b = np.zeros(1)
    _relax(oracle, A, {"name": "jacobi"}, x, b); assert np.allclose(x, [0])
    A = poisson1d(3); x = np.zeros(3); b = np.arange(3.0)
    _relax(oracle, A, {"name": "jacobi"}, x, b); assert np.allclose(x, [0.0, 0.5, 1.0])
    x = np.arange(3.0); b = np.zeros(3)
    _relax(oracle, A, {"name": "jacobi"}, x, b); assert np.allclose(x, [0.5, 1.0, 0.5])
    A = poisson1d(1); x = np.arange(1.0); b = np.array([10.0])
    _relax(oracle, A, {"name": "jacobi"}, x, b); assert np.allclose(x, [5])
    A = poisson1d(3); x = np.arange(3.0); b = np.array([10.0, 20, 30])
    _relax(oracle, A, {"name": "jacobi"}, x, b); assert np.allclose(x, [5.5, 11.0, 15.5])
    x = np.arange(3.0); x_copy = x.copy(); b = np.array([10.0, 20, 30])
    _relax(oracle, A, {"name": "jacobi", "omega": 1.0 / 3.0}, x, b)
    assert np.allclose(x, 2.0 / 3.0 * x_copy + 1.0 / 3.0 * np.array([5.5, 11.0, 15.5]))


def test_kat_gauss_seidel(oracle):
    # test_relaxation.py:291-353
    A = poisson1d(1); x = np.arange(1.0); b = np.zeros(1)
    _relax(oracle, A, {"name": "gauss_seidel"}, x, b); assert np.allclose(x, [0])
    A = poisson1d(3); x = np.arange(3.0); b = np.zeros(3)
    _relax(oracle, A, {"name": "gauss_seidel"}, x, b); assert np.allclose(x, [1.0 / 2.0, 5.0 / 4.0, 5.0 / 8.0])
    x = np.arange(3.0)
    _relax(oracle, A, {"name": "gauss_seidel", "sweep": "backward"}, x, b)
    assert np.allclose(x, [1.0 / 8.0, 1.0 / 4.0, 1.0 / 2.0])
    x = np.arange(3.0); b = np.array([10.0, 20, 30])
    _relax(oracle, A, {"name": "gauss_seidel"}, x, b)
    assert np.allclose(x, [11.0 / 2.0, 55.0 / 4, 175.0 / 8.0])
    # forward then backward == symmetric
    x = np.arange(3.0); y = x.copy()
    _relax(oracle, A, {"name": "gauss_seidel", "sweep": "symmetric"}, x, b)
    _relax(oracle, A, {"name": "gauss_seidel", "sweep": "forward"}, y, b)
    _relax(oracle, A, {"name": "gauss_seidel", "sweep": "backward"}, y, b)
    assert np.array_equal(x, y)


def test_kat_gauss_seidel_indexed(oracle):
    # test_relaxation.py:355-402
    A = poisson1d(3); b = np.zeros(3)
    x = np.arange(3.0)
    _relax(oracle, A, {"name": "gauss_seidel_indexed", "indices": [0, 1, 2]}, x, b)
    assert np.allclose(x, [1.0 / 2.0, 5.0 / 4.0, 5.0 / 8.0])
    x = np.arange(3.0)
    _relax(oracle, A, {"name": "gauss_seidel_indexed", "indices": [2, 1, 0]}, x, b)
    assert np.allclose(x, [1.0 / 8.0, 1.0 / 4.0, 1.0 / 2.0])
    x = np.arange(3.0)
    _relax(oracle, A, {"name": "gauss_seidel_indexed", "indices": [0, 1, 2], "sweep": "backward"}, x, b)
    assert np.allclose(x, [1.0 / 8.0, 1.0 / 4.0, 1.0 / 2.0])
    A = poisson1d(4); x = np.ones(4); b = np.zeros(4)
    _relax(oracle, A, {"name": "gauss_seidel_indexed", "indices": [0, 3]}, x, b)
    assert np.allclose(x, [1.0 / 2.0, 1.0, 1.0, 1.0 / 2.0])


def test_kat_polynomial(oracle):
    # test_relaxation.py:106-137
    A = poisson1d(10)
    np.random.seed(0)
    x0 = np.arange(10.0); b = np.zeros(10)
    x = x0.copy()
    _relax(oracle, A, {"name": "polynomial", "coefficients": [0.0]}, x, b)
    assert np.allclose(x, x0)
    x = x0.copy()
    _relax(oracle, A, {"name": "polynomial", "coefficients": [1.0]}, x, b)   # x += 1*(b - A x)
    assert np.allclose(x, x0 - A * x0)
    x = x0.copy(); b = np.arange(10.0) ** 2
    _relax(oracle, A, {"name": "polynomial", "coefficients": [1.0, 0.0]}, x, b)  # x += A r
    assert np.allclose(x, x0 + A * (b - A * x0))
    x = x0.copy()
    _relax(oracle, A, {"name": "polynomial", "coefficients": [0.2, -1.0]}, x, b)
    r = b - A * x0
    assert np.allclose(x, x0 + 0.2 * (A * r) - r)


def test_row_parallel_oracle_is_thread_count_independent():
    """bench.py's CPU baseline runs the oracle's row loops over all host cores; a row is still summed by
    one thread, left to right, so solves must agree bit for bit for any thread count."""
    lib = oracle_lib.load()
    for case in ("sa_cheb2_3d", "sa_jacobi_2d"):
        g = golden_io.load_hier(case)
        H = oracle_lib.Hierarchy(g["levels"], g["coarse_pinv"])
        out = []
        for threads in (1, 4):
            lib.oracle_set_threads(threads)
            out.append(H.solve(g["b"], tol=0.0, maxiter=4))
        lib.oracle_set_threads(1)
        assert np.array_equal(out[0][0], out[1][0])
        assert np.array_equal(out[0][1], out[1][1])


def _schwarz_cases():
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "schwarz.npz"), allow_pickle=False)
    return {str(n): {k.split("__", 1)[1]: z[k] for k in z.files if k.startswith(str(n) + "__")}
            for n in z["cases"]}


def test_oracle_schwarz_matches_reference_bit_for_bit():
    """extract_subblocks and overlapping_schwarz_csr (relaxation.h:836-1007) against outputs of the
    reference's own native module: default subdomains (A's pattern) and user subdomains; forward and
    backward sweeps; the symmetric 2-iteration shim result; the docstring example of relaxation.schwarz
    (||b - A x|| = 0.126326160522 after 10 iterations on the 10x10 Poisson problem)."""
    lib = oracle_lib.load()
    for name, c in _schwarz_cases().items():
        Ap, Aj, Ax = (np.ascontiguousarray(c[k]) for k in ("Ap", "Aj", "Ax"))
        n = len(Ap) - 1
        if name == "schwarz_docstring":
            Sp, Sj = Ap.copy(), Aj.copy()
            nsd = n
            Tp = np.zeros(nsd + 1, dtype=np.intc); Tp[1:] = np.cumsum((Sp[1:] - Sp[:-1]) ** 2)
            Tx = np.zeros(Tp[-1])
            lib.oracle_extract_subblocks(ip(Ap), ip(Aj), dp(Ax), dp(Tx), ip(Tp), ip(Sj), ip(Sp), nsd, n)
            for d in range(nsd):                       # the reference inverts with LAPACK gelss: not bit-pinned
                m = Sp[d + 1] - Sp[d]
                Tx[Tp[d]:Tp[d + 1]] = np.linalg.pinv(Tx[Tp[d]:Tp[d + 1]].reshape(m, m)).ravel()
            x = np.zeros(n); b = np.ones(n)
            for _ in range(10):
                lib.oracle_overlapping_schwarz_csr(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(b), dp(Tx), ip(Tp), ip(Sj),
                                                   ip(Sp), nsd, n, 0, nsd, 1)
            A = sps.csr_matrix((Ax, Aj, Ap), shape=(n, n))
            assert abs(np.linalg.norm(b - A * x) - 0.126326160522) < 5e-13
            assert np.allclose(x, c["x"], rtol=1e-12, atol=1e-14)
            continue
        Sj, Sp, Tp = (np.ascontiguousarray(c[k]) for k in ("Sj", "Sp", "Tp"))
        nsd = len(Sp) - 1
        Tx = np.full(Tp[-1], np.nan)
        lib.oracle_extract_subblocks(ip(Ap), ip(Aj), dp(Ax), dp(Tx), ip(Tp), ip(Sj), ip(Sp), nsd, n)
        assert np.array_equal(Tx, c["Tx_raw"]), name
        Tinv = np.ascontiguousarray(c["Tx"])
        for key, (rs, re, rt) in (("x_fwd", (0, nsd, 1)), ("x_bwd", (nsd - 1, -1, -1))):
            x = c["x0"].copy()
            lib.oracle_overlapping_schwarz_csr(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), dp(Tinv), ip(Tp),
                                               ip(Sj), ip(Sp), nsd, n, rs, re, rt)
            assert np.array_equal(x, c[key]), (name, key)
        x = c["x0"].copy()
        for _ in range(2):
            for (rs, re, rt) in ((0, nsd, 1), (nsd - 1, -1, -1)):
                lib.oracle_overlapping_schwarz_csr(ip(Ap), ip(Aj), dp(Ax), dp(x), dp(c["b"].copy()), dp(Tinv),
                                                   ip(Tp), ip(Sj), ip(Sp), nsd, n, rs, re, rt)
        assert np.array_equal(x, c["x_sym2"]), name


# ---------------------------------------------------------------------------
# Krylov-accelerated solves: the reference's own solve(accel=...) histories (oracle/gen_golden_r2.py) against
# the host restatements of tests/krylov_host.py preconditioned with the oracle's cycle -- this pins the
# restatements, which the device implementations are then compared with on the GPU.
# ---------------------------------------------------------------------------
import krylov_host  # noqa: E402


def _oracle_operators(g):
    H = build_oracle_hier(g)
    A = g["levels"][0]["A"]
    cyc = g["meta"]["cycle"]

    def M(r):
        z = np.zeros_like(r)
        H.cycle(z, np.ascontiguousarray(r), cyc)
        return z
    return (lambda v: A * v), M, H


@pytest.mark.parametrize("case", golden_io.accel_cases("cg"))
def test_host_cg_restatement_matches_reference_history(case):
    g = golden_io.load_hier(case)
    m = g["meta"]
    A, M, _keep = _oracle_operators(g)
    x, res, info = krylov_host.cg(A, M, g["b"], g["x0"], m["tol"], m["maxiter"])
    ref = g["residuals"]
    assert len(res) == len(ref)
    # the history is sqrt(<r, M r>): inner products through BLAS in the reference, numpy here
    assert np.allclose(res, ref, rtol=1e-9, atol=1e-13 * ref[0]), np.max(np.abs(np.array(res) - ref) / ref)
    assert np.linalg.norm(x - g["x"]) <= 1e-10 * np.linalg.norm(g["x"])
