"""ctypes binding of oracle/_build/libamg_oracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_build", "libamg_oracle.so")

c_int_p = C.POINTER(C.c_int)
c_dbl_p = C.POINTER(C.c_double)

FMT_CSR, FMT_BSR = 0, 1
SM = {None: 0, "None": 0, "jacobi": 1, "gauss_seidel": 2, "sor": 3, "polynomial": 4,
      "block_jacobi": 5, "block_gauss_seidel": 6, "gauss_seidel_indexed": 7,
      "gauss_seidel_ne": 8, "gauss_seidel_nr": 9, "jacobi_ne": 10, "schwarz": 11}
SWEEP = {"forward": 0, "backward": 1, "symmetric": 2}
CYCLE = {"V": 0, "W": 1, "F": 2, "AMLI": 3}


class Mat(C.Structure):
    _fields_ = [("fmt", C.c_int), ("nrows", C.c_int), ("ncols", C.c_int), ("R", C.c_int),
                ("C", C.c_int), ("Ap", c_int_p), ("Aj", c_int_p), ("Ax", c_dbl_p)]


class Smoother(C.Structure):
    _fields_ = [("kind", C.c_int), ("iterations", C.c_int), ("sweep", C.c_int),
                ("omega", C.c_double), ("ncoef", C.c_int), ("coef", c_dbl_p),
                ("blocksize", C.c_int), ("Dinv", c_dbl_p), ("indices", c_int_p),
                ("nindices", C.c_int), ("Aalt", C.POINTER(Mat)),
                ("Sj", c_int_p), ("Sp", c_int_p), ("Tp", c_int_p), ("Tx", c_dbl_p), ("nsdomains", C.c_int)]


def ip(a):
    return a.ctypes.data_as(c_int_p)


def dp(a):
    return a.ctypes.data_as(c_dbl_p)


def build():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_build/libamg_oracle.so"],
                   check=True)


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(
            os.path.join(ROOT, "oracle", "amg_oracle.c")):
        build()
    lib = C.CDLL(LIB)
    I, D = C.c_int, C.c_double
    sig = {
        "oracle_gauss_seidel": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I],
        "oracle_bsr_gauss_seidel": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, I],
        "oracle_jacobi": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, c_dbl_p],
        "oracle_bsr_jacobi": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, I, c_dbl_p],
        "oracle_gauss_seidel_indexed": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_int_p, I, I, I],
        "oracle_jacobi_ne": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, c_dbl_p],
        "oracle_gauss_seidel_ne": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, c_dbl_p, D],
        "oracle_gauss_seidel_nr": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, c_dbl_p, D],
        "oracle_block_jacobi": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, c_dbl_p, I],
        "oracle_block_gauss_seidel": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, I, I, I, I],
        "oracle_extract_subblocks": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_int_p, c_int_p, c_int_p, I, I],
        "oracle_overlapping_schwarz_csr": [c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_int_p, c_int_p,
                                           c_int_p, I, I, I, I, I],
        "oracle_csr_matvec": [I, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p],
        "oracle_bsr_matvec": [I, I, I, c_int_p, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p],
        "oracle_hier_destroy": [C.c_void_p],
        "oracle_hier_set_A": [C.c_void_p, I, C.POINTER(Mat)],
        "oracle_hier_set_PR": [C.c_void_p, I, C.POINTER(Mat), C.POINTER(Mat)],
        "oracle_hier_set_smoothers": [C.c_void_p, I, C.POINTER(Smoother), C.POINTER(Smoother)],
        "oracle_hier_set_coarse_dense": [C.c_void_p, c_dbl_p, I],
        "oracle_hier_set_duplicate_prolongation": [C.c_void_p, I],
        "oracle_relax": [C.POINTER(Mat), C.POINTER(Smoother), c_dbl_p, c_dbl_p],
        "oracle_cycle": [C.c_void_p, I, c_dbl_p, c_dbl_p, I],
    }
    for name, args in sig.items():
        f = getattr(lib, name)
        f.argtypes = args
        f.restype = None
    lib.oracle_set_threads.argtypes = [I]
    lib.oracle_set_threads.restype = None
    lib.oracle_get_threads.argtypes = []
    lib.oracle_get_threads.restype = I
    lib.oracle_norm2.argtypes = [c_dbl_p, C.c_long]
    lib.oracle_norm2.restype = D
    lib.oracle_hier_create.argtypes = [I]
    lib.oracle_hier_create.restype = C.c_void_p
    lib.oracle_solve.argtypes = [C.c_void_p, c_dbl_p, c_dbl_p, D, I, I, c_dbl_p]
    lib.oracle_solve.restype = I
    _lib = lib
    return lib


def make_mat(M, keep):
    """scipy csr/bsr matrix -> oracle_mat (arrays appended to `keep`)."""
    if sps.isspmatrix_bsr(M):
        fmt, (R, Cc) = FMT_BSR, M.blocksize
        data = np.ascontiguousarray(np.ravel(M.data), dtype=np.float64)
    else:
        M = sps.csr_matrix(M)
        fmt, R, Cc = FMT_CSR, 1, 1
        data = np.ascontiguousarray(M.data, dtype=np.float64)
    Ap = np.ascontiguousarray(M.indptr, dtype=np.intc)
    Aj = np.ascontiguousarray(M.indices, dtype=np.intc)
    keep.extend([Ap, Aj, data])
    return Mat(fmt, M.shape[0], M.shape[1], R, Cc, ip(Ap), ip(Aj), dp(data))


def make_smoother(desc, A, keep):
    """descriptor dict (see tests/golden_io.py) -> oracle_smoother."""
    s = Smoother()
    if desc is None or desc.get("name") in (None, "None"):
        s.kind = 0
        return s
    name = desc["name"]
    s.kind = SM[name]
    s.iterations = int(desc.get("iterations", 1))
    s.sweep = SWEEP[desc.get("sweep", "forward")]
    s.omega = float(desc.get("omega", 1.0))
    if "coefficients" in desc:
        co = np.ascontiguousarray(desc["coefficients"], dtype=np.float64)
        keep.append(co)
        s.ncoef, s.coef = len(co), dp(co)
    bs = int(desc.get("blocksize", 1))
    s.blocksize = bs
    if desc.get("Dinv") is not None:
        Dinv = np.ascontiguousarray(np.ravel(desc["Dinv"]), dtype=np.float64)
        keep.append(Dinv)
        s.Dinv = dp(Dinv)
    if desc.get("indices") is not None:
        idx = np.ascontiguousarray(desc["indices"], dtype=np.intc)
        keep.append(idx)
        s.indices, s.nindices = ip(idx), len(idx)
    if name in ("gauss_seidel_ne", "jacobi_ne", "gauss_seidel_nr"):
        # the shims act on lvl.Acsr / lvl.Acsc (smoothing.py:452-478)
        M = A.tocsc() if name == "gauss_seidel_nr" else A.tocsr()
        M.sort_indices() if name == "gauss_seidel_nr" else None
        Ap = np.ascontiguousarray(M.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(M.indices, dtype=np.intc)
        Ax = np.ascontiguousarray(M.data, dtype=np.float64)
        keep.extend([Ap, Aj, Ax])
        m = Mat(0, M.shape[0], M.shape[1], 1, 1, ip(Ap), ip(Aj), dp(Ax))
        keep.append(m)
        s.Aalt = C.pointer(m)
        if desc.get("Dinv") is None:
            # util/utils.py:526-588 get_diagonal(A, norm_eq=2 | 1, inv=True): squared entries times ones
            Ms = M.copy() if name != "gauss_seidel_nr" else M.T.tocsr()
            Ms.sort_indices()
            D = np.asarray((Ms.multiply(Ms)) * np.ones((Ms.shape[1],))).ravel()
            Dn = np.zeros_like(D)
            Dn[D != 0.0] = 1.0 / D[D != 0.0]
            keep.append(Dn)
            s.Dinv = dp(Dn)
    if name == "schwarz":
        arrs = [np.ascontiguousarray(desc[k], dtype=np.intc) for k in ("subdomain", "subdomain_ptr", "inv_subblock_ptr")]
        Tx = np.ascontiguousarray(desc["inv_subblock"], dtype=np.float64)
        keep.extend(arrs + [Tx])
        s.Sj, s.Sp, s.Tp, s.Tx = ip(arrs[0]), ip(arrs[1]), ip(arrs[2]), dp(Tx)
        s.nsdomains = len(arrs[1]) - 1
        Ac = A.tocsr()
        Ac.sort_indices()
        m = make_mat(Ac, keep)
        keep.append(m)
        s.Aalt = C.pointer(m)
    if name in ("block_jacobi", "block_gauss_seidel"):
        # the shim re-blocks A: relaxation.py:471,563  A = A.tobsr(blocksize=(bs,bs))
        Ab = A.tobsr(blocksize=(bs, bs))
        m = make_mat(Ab, keep)
        keep.append(m)
        s.Aalt = C.pointer(m)
    return s


class Hierarchy(object):
    """Owns an oracle_hier plus every numpy array it borrows."""

    def __init__(self, levels, coarse_pinv, dup_prolong=False):
        """levels: list of dicts {A, P, R, pre, post} (P/R/pre/post absent on the last)."""
        self.lib = load()
        self.keep = []
        self.h = self.lib.oracle_hier_create(len(levels))
        self.n = levels[0]["A"].shape[0]
        for i, L in enumerate(levels):
            A = make_mat(L["A"], self.keep)
            self.lib.oracle_hier_set_A(self.h, i, C.byref(A))
            if i < len(levels) - 1:
                P = make_mat(L["P"], self.keep)
                R = make_mat(L["R"], self.keep)
                self.lib.oracle_hier_set_PR(self.h, i, C.byref(P), C.byref(R))
                pre = make_smoother(L.get("pre"), L["A"], self.keep)
                post = make_smoother(L.get("post"), L["A"], self.keep)
                self.lib.oracle_hier_set_smoothers(self.h, i, C.byref(pre), C.byref(post))
        if coarse_pinv is not None:
            cp = np.ascontiguousarray(coarse_pinv, dtype=np.float64)
            self.keep.append(cp)
            self.lib.oracle_hier_set_coarse_dense(self.h, dp(cp), cp.shape[0])
        self.lib.oracle_hier_set_duplicate_prolongation(self.h, int(dup_prolong))

    def solve(self, b, x0=None, tol=1e-5, maxiter=100, cycle="V"):
        b = np.ascontiguousarray(b, dtype=np.float64)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=np.float64)
        res = np.zeros(maxiter + 1)
        k = self.lib.oracle_solve(self.h, dp(b), dp(x), tol, maxiter, CYCLE[cycle], dp(res))
        return x, res[:k].copy()

    def cycle(self, x, b, cycle="V"):
        self.lib.oracle_cycle(self.h, 0, dp(x), dp(b), CYCLE[cycle])

    def __del__(self):
        try:
            self.lib.oracle_hier_destroy(self.h)
        except Exception:
            pass
