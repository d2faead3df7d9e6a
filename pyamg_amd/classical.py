"""Classical (Ruge-Stuben) AMG setup on the CPU -- restates
/root/reference/pyamg/classical/classical.py:22-188 with its default pipeline:
classical strength (theta) -> RS first-pass C/F splitting -> direct interpolation ->
R = P^T -> Galerkin R*A*P, so that BASELINE configuration 1 (README example) can be
built where the reference cannot travel.  Other splittings (PMIS, CLJP) and strength
measures raise NotImplementedError.  Returns a pyamg_amd.multilevel_solver.
"""
import ctypes as C
from warnings import warn

import numpy as np
from scipy.sparse import SparseEfficiencyWarning, csr_matrix, isspmatrix_csr

from .aggregation import _dp, _ip, host_lib, symmetric_strength_of_connection
from .multilevel import multilevel_solver
from .smoothing import change_smoothers

__all__ = ["ruge_stuben_solver", "classical_strength_of_connection", "RS", "direct_interpolation"]

_sig_done = False


def _lib():
    global _sig_done
    L = host_lib()
    if not _sig_done:
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        L.amgsetup_classical_strength.argtypes = [C.c_int, C.c_double, ip, ip, dp, ip, ip, dp]
        L.amgsetup_classical_strength.restype = C.c_int
        L.amgsetup_rs_cf_splitting.argtypes = [C.c_int, ip, ip, ip, ip, ip]
        L.amgsetup_rs_cf_splitting.restype = None
        L.amgsetup_rs_direct_interpolation_pass1.argtypes = [C.c_int, ip, ip, ip, ip]
        L.amgsetup_rs_direct_interpolation_pass1.restype = C.c_int
        L.amgsetup_rs_direct_interpolation_pass2.argtypes = [C.c_int, ip, ip, dp, ip, ip, dp, ip, ip, ip, dp]
        L.amgsetup_rs_direct_interpolation_pass2.restype = None
        _sig_done = True
    return L


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.intc)


def classical_strength_of_connection(A, theta=0.0):
    """pyamg/strength.py:111-210, amg_core/ruge_stuben.h:46-99"""
    if not isspmatrix_csr(A):
        warn("Implicit conversion of A to csr", SparseEfficiencyWarning)
        A = csr_matrix(A)
    if theta < 0 or theta > 1:
        raise ValueError("expected theta in [0,1]")
    Ap, Aj, Ax = _i32(A.indptr), _i32(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
    Sp = np.empty_like(Ap); Sj = np.empty_like(Aj); Sx = np.empty_like(Ax)
    nnz = _lib().amgsetup_classical_strength(A.shape[0], float(theta), _ip(Ap), _ip(Aj), _dp(Ax), _ip(Sp), _ip(Sj),
                                             _dp(Sx))
    S = csr_matrix((Sx[:nnz], Sj[:nnz], Sp), shape=A.shape)
    S.data = np.abs(S.data)
    # scale_rows_by_largest_entry
    counts = np.diff(S.indptr)
    nz = counts > 0
    largest = np.ones(S.shape[0])
    if S.nnz:
        largest[nz] = np.maximum.reduceat(S.data, S.indptr[:-1][nz])
    largest[largest == 0] = 1.0
    S.data = S.data / np.repeat(largest, counts)
    return S


def remove_diagonal(S):
    """pyamg/util/utils.py:1775-1830"""
    if not isspmatrix_csr(S):
        raise TypeError("expected csr_matrix")
    if S.shape[0] != S.shape[1]:
        raise ValueError("expected square matrix, shape=%s" % (S.shape,))
    S = S.tocoo()
    mask = S.row != S.col
    from scipy.sparse import coo_matrix
    return coo_matrix((S.data[mask], (S.row[mask], S.col[mask])), shape=S.shape).tocsr()


def RS(S):
    """pyamg/classical/split.py:110-158: first pass of the Ruge-Stuben splitting (C=1, F=0)"""
    if not isspmatrix_csr(S):
        raise TypeError("expected csr_matrix")
    S = remove_diagonal(S)
    T = S.T.tocsr()
    splitting = np.empty(S.shape[0], dtype=np.intc)
    _lib().amgsetup_rs_cf_splitting(S.shape[0], _ip(_i32(S.indptr)), _ip(_i32(S.indices)), _ip(_i32(T.indptr)),
                                    _ip(_i32(T.indices)), _ip(splitting))
    return splitting


def direct_interpolation(A, Cm, splitting):
    """pyamg/classical/interpolate.py:13-75"""
    if not isspmatrix_csr(A):
        raise TypeError("expected csr_matrix for A")
    if not isspmatrix_csr(Cm):
        raise TypeError("expected csr_matrix for C")
    Cm = Cm.copy()
    Cm.data[:] = 1.0
    Cm = csr_matrix(Cm.multiply(A))
    n = A.shape[0]
    splitting = _i32(splitting)
    Cp, Cj, Cx = _i32(Cm.indptr), _i32(Cm.indices), np.ascontiguousarray(Cm.data, dtype=np.float64)
    Ap, Aj, Ax = _i32(A.indptr), _i32(A.indices), np.ascontiguousarray(A.data, dtype=np.float64)
    Pp = np.empty(n + 1, dtype=np.intc)
    nnz = _lib().amgsetup_rs_direct_interpolation_pass1(n, _ip(Cp), _ip(Cj), _ip(splitting), _ip(Pp))
    Pj = np.empty(nnz, dtype=np.intc)
    Px = np.empty(nnz, dtype=np.float64)
    _lib().amgsetup_rs_direct_interpolation_pass2(n, _ip(Ap), _ip(Aj), _dp(Ax), _ip(Cp), _ip(Cj), _dp(Cx),
                                                  _ip(splitting), _ip(Pp), _ip(Pj), _dp(Px))
    return csr_matrix((Px, Pj, Pp))


def unpack_arg(v):
    if isinstance(v, tuple):
        return v[0], v[1]
    return v, {}


def ruge_stuben_solver(A, strength=("classical", {"theta": 0.25}), CF="RS",
                       presmoother=("gauss_seidel", {"sweep": "symmetric"}),
                       postsmoother=("gauss_seidel", {"sweep": "symmetric"}),
                       max_levels=10, max_coarse=500, keep=False, **kwargs):
    """Create a multilevel solver using Classical AMG (pyamg/classical/classical.py:22-116)"""
    levels = [multilevel_solver.level()]
    if not isspmatrix_csr(A):
        try:
            A = csr_matrix(A)
            warn("Implicit conversion of A to CSR", SparseEfficiencyWarning)
        except Exception:
            raise TypeError("Argument A must have type csr_matrix, or be convertible to csr_matrix")
    A = A.astype(np.float64) if A.dtype != np.float64 else A
    if A.shape[0] != A.shape[1]:
        raise ValueError("expected square matrix")
    levels[-1].A = A
    while len(levels) < max_levels and levels[-1].A.shape[0] > max_coarse:
        extend_hierarchy(levels, strength, CF, keep)
    ml = multilevel_solver(levels, **kwargs)
    change_smoothers(ml, presmoother, postsmoother)
    return ml


def extend_hierarchy(levels, strength, CF, keep):
    """classical.py:120-188"""
    A = levels[-1].A
    fn, kwargs = unpack_arg(strength)
    if fn == "symmetric":
        Cm = symmetric_strength_of_connection(A, **kwargs)
    elif fn == "classical":
        Cm = classical_strength_of_connection(A, **kwargs)
    elif fn is None:
        Cm = A
    else:
        raise NotImplementedError("strength=%r is outside the restated setup" % (fn,))
    fn, kwargs = unpack_arg(CF)
    if fn == "RS":
        splitting = RS(Cm)
    else:
        raise NotImplementedError("C/F splitting %r is outside the restated setup" % (fn,))
    P = direct_interpolation(A, Cm, splitting)
    R = P.T.tocsr()
    if keep:
        levels[-1].C = Cm
        levels[-1].splitting = splitting
    levels[-1].P = P
    levels[-1].R = R
    levels.append(multilevel_solver.level())
    A = R * A * P
    levels[-1].A = A
