"""Smoothed-aggregation setup on the CPU -- the part of
/root/reference/pyamg/aggregation/aggregation.py the BASELINE configurations
use, restated so that a hierarchy can be built where the reference cannot
travel (the GPU box).  The hierarchy is built ONCE on the host and shipped to
HBM by ``multilevel_solver``; nothing here runs inside the cycle.

Supported subset (anything else raises NotImplementedError):
  strength   'symmetric' (any theta) | None | ('predefined', {'C': csr})
  aggregate  'standard' | ('predefined', {'AggOp': csr})
  smooth     ('jacobi', {'omega', 'degree'}) | None
  symmetry   'hermitian' | 'symmetric'
  improve_candidates  relaxation descriptors (run on the device) | None

Arithmetic follows the reference step by step (same scipy sparse products,
same RNG consumption in the spectral-radius estimates), so for a seeded run the
operators match the reference's to rounding -- tests/test_setup_golden.py pins
that against the captured hierarchies.
"""
import ctypes as C
import os
import time

import numpy as np
import scipy.sparse as sparse
from scipy.sparse import bsr_matrix, csr_matrix, isspmatrix_bsr, isspmatrix_csr

from .multilevel import multilevel_solver
from .smoothing import change_smoothers
from .util import (approximate_spectral_radius, approximate_spectral_radius_device, get_diagonal,
                   release_device_operator, scale_rows, use_device_for)

__all__ = ["smoothed_aggregation_solver", "standard_aggregation", "fit_candidates",
           "symmetric_strength_of_connection", "jacobi_prolongation_smoother"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_host = None


def host_lib():
    global _host
    if _host is None:
        path = os.path.join(_HERE, "lib", "libamgsetup_host.so")
        if not os.path.exists(path):
            raise ImportError("%s missing: run `make -C pyamg_amd/csrc`" % path)
        L = C.CDLL(path)
        ip, lp, dp = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_double)
        L.amgsetup_standard_aggregation.argtypes = [C.c_int, ip, ip, ip, ip]
        L.amgsetup_standard_aggregation.restype = C.c_int
        L.amgsetup_gauss_seidel.argtypes = [ip, ip, dp, dp, dp, C.c_int, C.c_int, C.c_int]
        L.amgsetup_gauss_seidel.restype = None
        L.amgsetup_block_gauss_seidel.argtypes = [ip, ip, dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.amgsetup_block_gauss_seidel.restype = None
        L.amgsetup_csr_diagonal_inv.argtypes = [C.c_int, lp, ip, dp, dp]
        L.amgsetup_csr_diagonal_inv.restype = None
        L.amgsetup_gauss_seidel_pipelined.argtypes = [ip, ip, dp, dp, dp, C.c_int, C.c_int, C.c_int]
        L.amgsetup_gauss_seidel_pipelined.restype = C.c_int
        L.amgsetup_block_gauss_seidel_pipelined.argtypes = [ip, ip, dp, dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_int]
        L.amgsetup_block_gauss_seidel_pipelined.restype = C.c_int
        L.amgsetup_csr_matmat_count.argtypes = [C.c_int, C.c_int, lp, ip, lp, ip, lp]
        L.amgsetup_csr_matmat_count.restype = C.c_int64
        L.amgsetup_csr_matmat_fill.argtypes = [C.c_int, C.c_int, lp, ip, dp, lp, ip, dp, lp, ip, dp]
        L.amgsetup_csr_matmat_fill.restype = C.c_int64
        L.amgsetup_csr_sort_indices.argtypes = [C.c_int, lp, ip, dp]
        L.amgsetup_csr_sort_indices.restype = None
        L.amgsetup_csr_transpose.argtypes = [C.c_int, C.c_int, lp, ip, dp, lp, ip, dp]
        L.amgsetup_csr_transpose.restype = None
        L.amgsetup_fit_candidates_scalar.argtypes = [C.c_int, ip, ip, dp, dp, dp, C.c_double]
        L.amgsetup_fit_candidates_scalar.restype = None
        L.amgsetup_poisson_nnz.argtypes = [C.c_int, C.c_int, C.c_int]
        L.amgsetup_poisson_nnz.restype = C.c_int64
        L.amgsetup_poisson.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, lp, ip, dp]
        L.amgsetup_poisson.restype = None
        L.amgsetup_tentative_scalar.argtypes = [C.c_int, C.c_int, ip, dp, C.c_double, lp, ip, dp, dp]
        L.amgsetup_tentative_scalar.restype = C.c_int64
        L.amgsetup_smooth_prolongator.argtypes = [C.c_int, C.c_int, lp, ip, dp, dp, C.c_double, lp, ip, dp,
                                                  lp, ip, dp]
        L.amgsetup_smooth_prolongator.restype = C.c_int64
        L.amgsetup_bsr_matmat_count.argtypes = [C.c_int, lp, ip, lp, ip, lp]
        L.amgsetup_bsr_matmat_count.restype = C.c_int64
        L.amgsetup_bsr_matmat_fill.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, lp, ip, dp, lp, ip, dp, lp, ip, dp]
        L.amgsetup_bsr_matmat_fill.restype = None
        L.amgsetup_bsr_transpose.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, lp, ip, dp, lp, ip, dp]
        L.amgsetup_bsr_transpose.restype = None
        L.amgsetup_smooth_prolongator_block.argtypes = [C.c_int, C.c_int, C.c_int, lp, ip, dp, dp, C.c_double, ip, dp, lp, ip,
                                                        dp, ip]
        L.amgsetup_smooth_prolongator_block.restype = C.c_int64
        L.amgsetup_kuhn_p1_diffusion.argtypes = [C.c_int, dp, dp, lp, ip, dp]
        L.amgsetup_kuhn_p1_diffusion.restype = C.c_int64
        L.amgsetup_greedy_coloring.argtypes = [C.c_int, ip, ip, ip]
        L.amgsetup_greedy_coloring.restype = C.c_int
        L.amgsetup_pattern_symmetric.argtypes = [C.c_int, ip, ip]
        L.amgsetup_pattern_symmetric.restype = C.c_int
        L.amgsetup_extract_subblocks.argtypes = [ip, ip, dp, dp, ip, ip, ip, C.c_int, C.c_int]
        L.amgsetup_extract_subblocks.restype = None
        L.amgsetup_num_threads.restype = C.c_int
        L.amgsetup_set_num_threads.argtypes = [C.c_int]
        L.amgsetup_set_num_threads.restype = None
        # Row-parallel setup kernels keep O(columns) scratch per thread: cap the team at the CPUs this
        # process may use (launchers such as torchrun export OMP_NUM_THREADS=1; a bare run on a big
        # host would otherwise start hundreds of threads).  AMG_SETUP_THREADS overrides.
        try:
            avail = len(os.sched_getaffinity(0))
        except AttributeError:
            avail = os.cpu_count() or 1
        want = int(os.environ.get("AMG_SETUP_THREADS", min(32, avail)))
        L.amgsetup_set_num_threads(max(1, want))
        _host = L
    return _host


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _lp(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def greedy_colouring(A):
    """colour of every row of the (symmetrised) pattern of A, first-fit in natural order"""
    M = A if (isspmatrix_csr(A) or isspmatrix_bsr(A)) else csr_matrix(A)
    if isspmatrix_bsr(M) and M.blocksize != (1, 1):
        M = M.tocsr()
    Ap = np.ascontiguousarray(M.indptr, dtype=np.intc)
    Aj = np.ascontiguousarray(M.indices, dtype=np.intc)
    if not host_lib().amgsetup_pattern_symmetric(M.shape[0], _ip(Ap), _ip(Aj)):
        # a colouring must respect both a_ij and a_ji (a structurally symmetric pattern -- the usual case -- needs no
        # symmetrised copy: first-fit only looks at neighbour SETS)
        S = csr_matrix((np.ones(len(M.indices), dtype=np.int8), M.indices, M.indptr), shape=M.shape)
        S = (S + S.T).tocsr()
        Ap = np.ascontiguousarray(S.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(S.indices, dtype=np.intc)
    colour = np.empty(M.shape[0], dtype=np.intc)
    ncol = host_lib().amgsetup_greedy_coloring(M.shape[0], _ip(Ap), _ip(Aj), _ip(colour))
    return colour, ncol


def unpack_arg(v):
    if isinstance(v, tuple):
        return v[0], v[1]
    return v, {}


def blocksize(A):
    return A.blocksize[0] if isspmatrix_bsr(A) else 1


# --------------------------------------------------------------------------- strength
def symmetric_strength_of_connection(A, theta=0):
    """pyamg/strength.py:213-318, amg_core/smoothed_aggregation.h:49-99:
    keep a_ij with |a_ij| >= theta*sqrt(|a_ii a_jj|) (the diagonal always), take
    magnitudes and scale each row by its largest entry."""
    if theta < 0:
        raise ValueError("expected a positive theta")
    if isspmatrix_csr(A):
        A = csr_matrix(A)
        n = A.shape[0]
        rows = np.repeat(np.arange(n), np.diff(A.indptr))
        isdiag = rows == A.indices
        d = np.zeros(n)
        np.add.at(d, rows[isdiag], A.data[isdiag])
        diags = np.abs(d)
        eps = (theta * theta) * diags
        keep = isdiag | (A.data * A.data >= eps[rows] * diags[A.indices])
        Sp = np.concatenate(([0], np.cumsum(np.bincount(rows[keep], minlength=n)))).astype(A.indptr.dtype)
        S = csr_matrix((A.data[keep], A.indices[keep], Sp), shape=A.shape)
    elif isspmatrix_bsr(A):
        M, N = A.shape
        R, Cb = A.blocksize
        if R != Cb:
            raise ValueError("matrix must have square blocks")
        if theta == 0:
            data = np.ones(len(A.indices), dtype=A.dtype)
            S = csr_matrix((data, A.indices.copy(), A.indptr.copy()), shape=(int(M / R), int(N / Cb)))
        else:
            data = (np.conjugate(A.data) * A.data).reshape(-1, R * Cb).sum(axis=1)
            Ab = csr_matrix((data, A.indices, A.indptr), shape=(int(M / R), int(N / Cb)))
            return symmetric_strength_of_connection(Ab, theta)
    else:
        raise TypeError("expected csr_matrix or bsr_matrix")
    S.data = np.abs(S.data)
    # scale_rows_by_largest_entry (util/utils.py)
    largest = np.maximum.reduceat(S.data, S.indptr[:-1][np.diff(S.indptr) > 0]) if S.nnz else np.array([])
    scale = np.ones(S.shape[0])
    scale[np.diff(S.indptr) > 0] = largest
    scale[scale == 0] = 1.0
    S.data = S.data / np.repeat(scale, np.diff(S.indptr))
    return S


# --------------------------------------------------------------------------- aggregation
def standard_aggregation(Cm):
    """pyamg/aggregation/aggregate.py:20-105 -> (AggOp, Cpts)"""
    if not isspmatrix_csr(Cm):
        raise TypeError("expected csr_matrix")
    if Cm.shape[0] != Cm.shape[1]:
        raise ValueError("expected square matrix")
    num_rows = Cm.shape[0]
    Ap = np.ascontiguousarray(Cm.indptr, dtype=np.intc)
    Aj = np.ascontiguousarray(Cm.indices, dtype=np.intc)
    Tj = np.empty(num_rows, dtype=np.intc)
    Cpts = np.empty(num_rows, dtype=np.intc)
    num_aggregates = host_lib().amgsetup_standard_aggregation(num_rows, _ip(Ap), _ip(Aj), _ip(Tj), _ip(Cpts))
    Cpts = Cpts[:num_aggregates]
    if num_aggregates == 0:
        return csr_matrix((num_rows, 1), dtype="int8"), np.array([], dtype=np.intc)
    shape = (num_rows, num_aggregates)
    if Tj.min() == -1:
        mask = Tj != -1
        row = np.arange(num_rows, dtype=np.intc)[mask]
        col = Tj[mask]
        data = np.ones(len(col), dtype="int8")
        return sparse.coo_matrix((data, (row, col)), shape=shape).tocsr(), Cpts
    Tp = np.arange(num_rows + 1, dtype=np.intc)
    Tx = np.ones(len(Tj), dtype="int8")
    return csr_matrix((Tx, Tj, Tp), shape=shape), Cpts


# --------------------------------------------------------------------------- tentative prolongator
def fit_candidates(AggOp, B, tol=1e-10):
    """pyamg/aggregation/tentative.py:19-166 / amg_core/smoothed_aggregation.h:323-500:
    per aggregate, modified Gram-Schmidt QR of the candidates restricted to it."""
    if not isspmatrix_csr(AggOp):
        raise TypeError("expected csr_matrix for argument AggOp")
    B = np.asarray(B)
    if B.dtype not in ["float32", "float64"]:
        B = np.asarray(B, dtype="float64")
    if len(B.shape) != 2:
        raise ValueError("expected 2d array for argument B")
    if B.shape[0] % AggOp.shape[0] != 0:
        raise ValueError("dimensions of AggOp %s and B %s are incompatible" % (AggOp.shape, B.shape))
    N_fine, N_coarse = AggOp.shape
    K1 = int(B.shape[0] / N_fine)
    K2 = B.shape[1]
    AggOp_csc = AggOp.tocsc()
    Ap, Ai = AggOp_csc.indptr, AggOp_csc.indices
    nnz = AggOp.nnz
    BS = K1 * K2
    Bb = B.reshape(-1, K1, K2)
    # copy blocks: Qx[ii] = B block of fine node Ai[ii]   (smoothed_aggregation.h:341-351)
    Qx, R = _aggregate_qr(Bb[Ai], Ap, N_coarse, K1, K2, tol)
    Q = bsr_matrix((Qx.swapaxes(1, 2).copy(), Ai, Ap), shape=(K2 * N_coarse, K1 * N_fine))
    Q = Q.T.tobsr()
    R = R.reshape(-1, K2)
    return Q, R


def _aggregate_qr(Qx, Ap, N_coarse, K1, K2, tol):
    """Per-aggregate modified Gram-Schmidt of the candidate blocks Qx (one K1 x K2 block per member, members of
    aggregate j at Ap[j]:Ap[j+1]) -> (Q blocks, R (N_coarse, K2, K2)); smoothed_aggregation.h:341-452."""
    Qx = np.array(Qx, dtype=np.float64)
    nnz = Qx.shape[0]
    R = np.zeros((N_coarse, K2, K2), dtype=np.float64)
    if K1 == 1 and K2 == 1:
        # scalar fast path (one candidate, scalar unknowns)
        q = np.empty(nnz, dtype=np.float64)
        Rv = np.zeros(N_coarse, dtype=np.float64)
        Bv = np.ascontiguousarray(Qx.ravel(), dtype=np.float64)
        Ap32 = np.ascontiguousarray(Ap, dtype=np.intc)
        Ai32 = np.arange(nnz, dtype=np.intc)
        host_lib().amgsetup_fit_candidates_scalar(N_coarse, _ip(Ap32), _ip(Ai32), _dp(Bv), _dp(q), _dp(Rv),
                                                  float(tol))
        R[:, 0, 0] = Rv
        return q.reshape(-1, 1, 1), R
    # general case: modified Gram-Schmidt per aggregate, batched over the aggregates that have the
    # same number of members; every sum runs over the rows in storage order, one row at a time,
    # exactly as the reference's scalar loops do (smoothed_aggregation.h:367-452)
    counts = np.diff(Ap)
    for m in np.unique(counts):
        if m == 0:
            continue
        aggs = np.nonzero(counts == m)[0]
        pos = (np.asarray(Ap)[aggs][:, None] + np.arange(m)[None, :]).astype(np.int64)      # (ng, m)
        blk = Qx[pos].reshape(len(aggs), m * K1, K2).copy()
        nrow = m * K1
        for bj in range(K2):
            norm_j = np.zeros(len(aggs))
            for rr in range(nrow):
                norm_j = norm_j + blk[:, rr, bj] * blk[:, rr, bj]
            norm_j = np.sqrt(norm_j)
            threshold_j = tol * norm_j
            for bi in range(bj):
                dot_prod = np.zeros(len(aggs))
                for rr in range(nrow):
                    dot_prod = dot_prod + blk[:, rr, bi] * blk[:, rr, bj]
                blk[:, :, bj] = blk[:, :, bj] - dot_prod[:, None] * blk[:, :, bi]
                R[aggs, bi, bj] = dot_prod
            norm_j = np.zeros(len(aggs))
            for rr in range(nrow):
                norm_j = norm_j + blk[:, rr, bj] * blk[:, rr, bj]
            norm_j = np.sqrt(norm_j)
            ok = norm_j > threshold_j
            scale = np.zeros(len(aggs))
            scale[ok] = 1.0 / norm_j[ok]
            R[aggs, bj, bj] = np.where(ok, norm_j, 0.0)
            blk[:, :, bj] = blk[:, :, bj] * scale[:, None]
        Qx[pos] = blk.reshape(len(aggs), m, K1, K2)
    return Qx, R


# --------------------------------------------------------------------------- prolongation smoothing
def jacobi_prolongation_smoother(S, T, Cm, B, omega=4.0 / 3.0, degree=1, filter=False,
                                 weighting="diagonal"):
    """pyamg/aggregation/smooth.py:67-210 (weighting 'diagonal' / 'local', no filtering)."""
    if filter:
        raise NotImplementedError("filtered prolongation smoothing is outside the restated setup")
    if weighting == "block":
        if isspmatrix_csr(S) or (isspmatrix_bsr(S) and S.blocksize[0] == 1):
            weighting = "diagonal"
    if weighting == "diagonal":
        D_inv = get_diagonal(S, inv=True)
        D_inv_S = scale_rows(S, D_inv, copy=True)
        D_inv_S = (omega / approximate_spectral_radius(D_inv_S)) * D_inv_S
    elif weighting == "local":
        D = np.abs(S) * np.ones((S.shape[0], 1), dtype=S.dtype)
        D_inv = np.zeros_like(D)
        D_inv[D != 0] = 1.0 / np.abs(D[D != 0])
        D_inv_S = scale_rows(S, D_inv, copy=True)
        D_inv_S = omega * D_inv_S
    else:
        raise NotImplementedError("weighting=%r" % weighting)
    P = T
    for i in range(degree):
        P = P - (D_inv_S * P)
    return P


# --------------------------------------------------------------------------- driver
def _levelize_sa(to_levelize, max_levels, max_coarse):
    """util/utils.py:1872-1953"""
    if isinstance(to_levelize, tuple):
        if to_levelize[0] == "predefined":
            to_levelize = [to_levelize]
            max_levels = 2
            max_coarse = 0
        else:
            to_levelize = [to_levelize for i in range(max_levels - 1)]
    elif isinstance(to_levelize, str):
        if to_levelize == "predefined":
            raise ValueError("predefined to_levelize requires a user-provided CSR matrix")
        to_levelize = [to_levelize for i in range(max_levels - 1)]
    elif isinstance(to_levelize, list):
        if isinstance(to_levelize[-1], tuple) and (to_levelize[-1][0] == "predefined"):
            max_levels = len(to_levelize) + 1
            max_coarse = 0
        elif len(to_levelize) < max_levels - 1:
            to_levelize = to_levelize + [to_levelize[-1]] * (max_levels - 1 - len(to_levelize))
    elif to_levelize is None:
        to_levelize = [(None, {}) for i in range(max_levels - 1)]
    else:
        raise ValueError("invalid to_levelize")
    return max_levels, max_coarse, to_levelize


def _levelize_smooth(to_levelize, max_levels):
    """util/utils.py:1956-2006"""
    if isinstance(to_levelize, tuple) or isinstance(to_levelize, str):
        return [to_levelize for i in range(max_levels)]
    if isinstance(to_levelize, list):
        if len(to_levelize) < max_levels:
            return to_levelize + [to_levelize[-1]] * (max_levels - len(to_levelize))
        return list(to_levelize)
    if to_levelize is None:
        return [(None, {}) for i in range(max_levels)]
    return to_levelize


def _improve(method, A, B):
    """relaxation_as_linear_operator(method, A, 0) * B  (util/utils.py:1129-1204,
    aggregation.py:313-320): relax A x = 0 from each candidate column.  Setup runs on
    the CPU: Gauss-Seidel sweeps use the host restatement of relaxation.h:34-62."""
    from . import smoothing
    fn, kwargs = unpack_arg(method)
    lvl = multilevel_solver.level()
    lvl.A = A
    desc = getattr(smoothing, "setup_" + str(fn))(lvl, **kwargs).desc
    n = A.shape[0]
    b = np.zeros(n)
    out = np.empty_like(B)
    L = host_lib()
    its = int(desc.get("iterations", 1))
    sw = desc.get("sweep", "forward")
    if desc["name"] == "gauss_seidel" and (isspmatrix_csr(A) or A.blocksize == (1, 1)):
        if isspmatrix_bsr(A):
            raise NotImplementedError("candidate improvement on BSR(1,1) levels")
        Ap = np.ascontiguousarray(A.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(A.indices, dtype=np.intc)
        Ax = np.ascontiguousarray(A.data, dtype=np.float64)

        def sweep(x, reverse):
            if reverse:
                L.amgsetup_gauss_seidel(_ip(Ap), _ip(Aj), _dp(Ax), _dp(x), _dp(b), n - 1, -1, -1)
            else:
                L.amgsetup_gauss_seidel(_ip(Ap), _ip(Aj), _dp(Ax), _dp(x), _dp(b), 0, n, 1)
    elif desc["name"] == "block_gauss_seidel":
        bs = int(desc["blocksize"])
        Ab = A.tobsr(blocksize=(bs, bs))                      # relaxation.py:563
        Ap = np.ascontiguousarray(Ab.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(Ab.indices, dtype=np.intc)
        Ax = np.ascontiguousarray(np.ravel(Ab.data), dtype=np.float64)
        Dinv = np.ascontiguousarray(np.ravel(desc["Dinv"]), dtype=np.float64)
        nb = n // bs

        def sweep(x, reverse):
            if reverse:
                L.amgsetup_block_gauss_seidel(_ip(Ap), _ip(Aj), _dp(Ax), _dp(x), _dp(b), _dp(Dinv), nb - 1, -1, -1, bs)
            else:
                L.amgsetup_block_gauss_seidel(_ip(Ap), _ip(Aj), _dp(Ax), _dp(x), _dp(b), _dp(Dinv), 0, nb, 1, bs)
    else:
        raise NotImplementedError("improve_candidates=%r on this matrix is outside the restated setup" % (fn,))
    sweep_code = {"forward": 0, "backward": 1, "symmetric": 2}.get(sw)

    def relax_column(j, pipelined_ok=False):
        x = np.array(B[:, j], dtype=np.float64, order="C")
        ran = 0
        if pipelined_ok and sweep_code is not None:
            # all sweeps of this column by several host threads that trail each other chunk by chunk
            # (setup_host.cpp: pipelined_sweep) -- the sequential result; 0 = the operator does not qualify
            if desc["name"] == "gauss_seidel":
                ran = L.amgsetup_gauss_seidel_pipelined(_ip(Ap), _ip(Aj), _dp(Ax), _dp(x), _dp(b), n, sweep_code, its)
            else:
                ran = L.amgsetup_block_gauss_seidel_pipelined(_ip(Ap), _ip(Aj), _dp(Ax), _dp(x), _dp(b), _dp(Dinv),
                                                              nb, bs, sweep_code, its)
        if not ran:
            for it in range(its):
                if sw in ("forward", "symmetric"):
                    sweep(x, False)
                if sw in ("backward", "symmetric"):
                    sweep(x, True)
        out[:, j] = x
        return ran

    if desc["name"] == "gauss_seidel" and sweep_code is not None and use_device_for(A) \
            and os.environ.get("AMG_SETUP_DEVICE_GS", "1") != "0":
        # r3: the sweeps on the GPU, in the operator's own row order from its CSR arrays (csrc/gsflow.hip: gs_natural_kernel;
        # the operator is in HBM for the spectral-radius estimate anyway) -- the sequential result bit for bit.  Operators
        # with rows of more than 8 entries (coarse levels) keep the host sweeps below.
        from . import _lib
        from .util import device_operator
        dirs = bytes(({0: [0], 1: [1], 2: [0, 1]}[sweep_code]) * its)
        op = device_operator(A)
        cols = []
        for j in range(B.shape[1]):
            x = np.array(B[:, j], dtype=np.float64, order="C")
            rc = _lib.lib().amg_hier_gs_natural(op.h, 0, x.ctypes.data, None, dirs, len(dirs))
            if rc != 0:
                cols = None
                if rc != _lib.AMG_ENOTIMPL:
                    # e.g. the persistent kernel gave up waiting because another process holds part of the GPU: the host
                    # sweeps below produce the same numbers (x is only written on success)
                    import warnings
                    warnings.warn("candidate improvement on the GPU failed (%s); using the host sweeps"
                                  % _lib.lib().amg_last_error().decode(), RuntimeWarning)
                break
            cols.append(x)
        if cols is not None:
            for j, x in enumerate(cols):
                out[:, j] = x
            return out
    if n > 100000 and os.environ.get("AMG_SETUP_PIPELINED_GS", "1") != "0":
        # first column through the pipelined sweep; if the operator qualifies, so do the others
        if relax_column(0, True):
            for j in range(1, B.shape[1]):
                relax_column(j, True)
            return out
        first = 1
    else:
        first = 0
    if first >= B.shape[1]:
        return out
    if B.shape[1] - first > 1 and n > 100000:
        # the candidates are relaxed independently (each sweep is sequential): one host thread per column
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(B.shape[1], 8)) as pool:
            list(pool.map(relax_column, range(first, B.shape[1])))
    else:
        for j in range(first, B.shape[1]):
            relax_column(j)
    return out


def smoothed_aggregation_solver(A, B=None, BH=None, symmetry="hermitian", strength="symmetric",
                                aggregate="standard", smooth=("jacobi", {"omega": 4.0 / 3.0}),
                                presmoother=("block_gauss_seidel", {"sweep": "symmetric"}),
                                postsmoother=("block_gauss_seidel", {"sweep": "symmetric"}),
                                improve_candidates=[("block_gauss_seidel", {"sweep": "symmetric",
                                                                            "iterations": 4}), None],
                                max_levels=10, max_coarse=500, diagonal_dominance=False, keep=False,
                                fast=True, **kwargs):
    """Create a multilevel solver using Smoothed Aggregation (SA)
    (pyamg/aggregation/aggregation.py:30-290); returns a pyamg_amd.multilevel_solver."""
    if not (isspmatrix_csr(A) or isspmatrix_bsr(A)):
        try:
            A = csr_matrix(A)
        except Exception:
            raise TypeError("Argument A must have type csr_matrix or bsr_matrix, or be convertible to csr_matrix")
    A = A.astype(np.float64) if A.dtype != np.float64 else A
    if symmetry not in ("symmetric", "hermitian"):
        raise NotImplementedError("symmetry=%r is outside the restated setup" % (symmetry,))
    if diagonal_dominance:
        raise NotImplementedError("diagonal_dominance is outside the restated setup")
    A.symmetry = symmetry
    if A.shape[0] != A.shape[1]:
        raise ValueError("expected square matrix")
    if B is None:
        if blocksize(A) == 1:
            B = np.ones((A.shape[0], 1), dtype=A.dtype)         # = the kron below for 1x1 blocks, without its temporaries
        else:
            B = np.kron(np.ones((int(A.shape[0] / blocksize(A)), 1), dtype=A.dtype), np.eye(blocksize(A)))
    else:
        B = np.asarray(B, dtype=A.dtype)
        if len(B.shape) == 1:
            B = B.reshape(-1, 1)
        if B.shape[0] != A.shape[0]:
            raise ValueError("The near null-space modes B have incorrect dimensions for matrix A")

    max_levels, max_coarse, strength = _levelize_sa(strength, max_levels, max_coarse)
    max_levels, max_coarse, aggregate = _levelize_sa(aggregate, max_levels, max_coarse)
    improve_candidates = _levelize_smooth(list(improve_candidates) if isinstance(improve_candidates, list)
                                          else improve_candidates, max_levels)
    smooth = _levelize_smooth(smooth, max_levels)

    verbose = os.environ.get("AMG_SETUP_VERBOSE", "0") != "0"
    t0 = time.perf_counter()
    levels = [multilevel_solver.level()]
    levels[-1].A = A
    levels[-1].B = B
    while len(levels) < max_levels and int(levels[-1].A.shape[0] / blocksize(levels[-1].A)) > max_coarse:
        extend_hierarchy(levels, strength, aggregate, smooth, improve_candidates, keep or not fast)
    t1 = time.perf_counter()
    ml = multilevel_solver(levels, **kwargs)
    t2 = time.perf_counter()
    change_smoothers(ml, presmoother, postsmoother)
    if verbose:
        print("[setup] levels %.2fs, multilevel_solver (coarse solver) %.2fs, smoothers %.2fs"
              % (t1 - t0, t2 - t1, time.perf_counter() - t2), flush=True)
    return ml


def poisson(grid, format="csr"):
    """gallery.poisson(grid) (pyamg/gallery/laplacian.py:14-69) for 1-3 dimensions, generated
    natively: 2d on the diagonal, -1 off it, last axis fastest, sorted int32 indices."""
    grid = tuple(int(g) for g in grid)
    if not 1 <= len(grid) <= 3 or min(grid) < 1:
        raise ValueError("invalid grid shape: %s" % str(grid))
    g3 = (1,) * (3 - len(grid)) + grid
    L = host_lib()
    n = g3[0] * g3[1] * g3[2]
    nnz = L.amgsetup_poisson_nnz(*g3)
    if nnz >= 2 ** 31:
        raise ValueError("matrix exceeds int32 indices")
    Ap = np.empty(n + 1, dtype=np.int64)
    Aj = np.empty(nnz, dtype=np.intc)
    Ax = np.empty(nnz, dtype=np.float64)
    L.amgsetup_poisson(g3[0], g3[1], g3[2], 2.0 * len(grid), _lp(Ap), _ip(Aj), _dp(Ax))
    A = csr_matrix((Ax, Aj, Ap.astype(np.intc)), shape=(n, n))
    A.has_sorted_indices = True
    return A.asformat(format)


def _csr_arrays64(M):
    """(indptr int64, indices int32, data f64) of a CSR / BSR(1,1) matrix, without copies when possible"""
    data = M.data.reshape(-1) if isspmatrix_bsr(M) else M.data
    return (np.ascontiguousarray(M.indptr, dtype=np.int64), np.ascontiguousarray(M.indices, dtype=np.intc),
            np.ascontiguousarray(data, dtype=np.float64))


def _matmat(Aa, Ba, shape):
    """C = A*B with scipy's csr_matmat arithmetic and output order, row-parallel on the host."""
    (Ap, Aj, Ax), (Bp, Bj, Bx) = Aa, Ba
    L = host_lib()
    n_row, n_col = shape
    Cp = np.empty(n_row + 1, dtype=np.int64)
    nnz = L.amgsetup_csr_matmat_count(n_row, n_col, _lp(Ap), _ip(Aj), _lp(Bp), _ip(Bj), _lp(Cp))
    Cj = np.empty(nnz, dtype=np.intc)
    Cx = np.empty(nnz, dtype=np.float64)
    nnz2 = L.amgsetup_csr_matmat_fill(n_row, n_col, _lp(Ap), _ip(Aj), _dp(Ax), _lp(Bp), _ip(Bj), _dp(Bx),
                                      _lp(Cp), _ip(Cj), _dp(Cx))
    if nnz2 != nnz:
        Cj = Cj[:nnz2].copy()
        Cx = Cx[:nnz2].copy()
    return Cp, Cj, Cx


def _as_bsr11(arrs, shape):
    Cp, Cj, Cx = arrs
    if Cp[-1] >= 2 ** 31:
        raise ValueError("operator exceeds int32 indices")
    M = bsr_matrix((Cx.reshape(-1, 1, 1), Cj, Cp.astype(np.intc)), shape=shape, copy=False)
    return M


def _scalar_fast_path_ok(A, B, strength_l, aggregate_l, smooth_l):
    if not (isspmatrix_csr(A) or (isspmatrix_bsr(A) and A.blocksize == (1, 1))):
        return False
    return _default_options(B, strength_l, aggregate_l, smooth_l)


def _default_options(B, strength_l, aggregate_l, smooth_l):
    """one candidate, symmetric strength with theta = 0, standard aggregation, one Jacobi smoothing step"""
    if B.shape[1] != 1:
        return False
    fn, kw = unpack_arg(strength_l)
    if fn != "symmetric" or kw.get("theta", 0) != 0:
        return False
    fn, kw = unpack_arg(aggregate_l)
    if fn != "standard" or kw:
        return False
    fn, kw = unpack_arg(smooth_l)
    if fn != "jacobi" or kw.get("degree", 1) != 1 or kw.get("filter", False) or \
            kw.get("weighting", "diagonal") not in ("diagonal", "block"):
        return False
    return True


def _extend_scalar(levels, smooth_l, keep, rho_fn):
    """extend_hierarchy for scalar problems with one candidate (the BASELINE Poisson
    configurations), on flat arrays with the host helpers: same arithmetic as the generic
    path below, sized for 10^8 unknowns."""
    A = levels[-1].A
    B = levels[-1].B
    L = host_lib()
    n = A.shape[0]
    verbose = os.environ.get("AMG_SETUP_VERBOSE", "0") != "0"
    _t = [time.perf_counter()]

    def lap(what):
        if verbose:
            now = time.perf_counter()
            print("[setup] level %d (%d rows) %-22s %6.2fs" % (len(levels) - 1, n, what, now - _t[0]), flush=True)
            _t[0] = now
    Ap, Aj, Ax = _csr_arrays64(A)
    Ap32 = np.ascontiguousarray(A.indptr, dtype=np.intc)
    # rho(D^-1 A) does not depend on the aggregation: when the rows are already sorted (level 0 of the gallery
    # operators; get_diagonal's in-place sort below is then a no-op and the aggregation reads the same order either
    # way) the estimate -- upload of A, Arnoldi on the GPU -- runs in a thread of its own beside aggregation and
    # tentative prolongator.  Neither of those draws random numbers, so the estimate's single np.random.rand draw is the
    # same draw.
    early = None
    if n >= 1000000 and rho_fn is _rho_D_inv_A_host and use_device_for(A) and getattr(A, "has_sorted_indices", False) \
            and os.environ.get("AMG_SETUP_OVERLAP_RHO", "1") != "0":
        import threading
        box = {}

        def estimate():
            try:
                d = np.empty(n, dtype=np.float64)
                L.amgsetup_csr_diagonal_inv(n, _lp(Ap), _ip(Aj), _dp(Ax), _dp(d))
                box["D_inv"] = d
                box["rho"] = rho_fn(A, d)
            except BaseException as e:      # noqa: BLE001 -- re-raised in the main thread
                box["error"] = e
        early = (threading.Thread(target=estimate), box)
        early[0].start()
    # strength with theta = 0 keeps every entry: the aggregation only reads the pattern
    agg = np.empty(n, dtype=np.intc)
    cpts = np.empty(n, dtype=np.intc)
    n_agg = L.amgsetup_standard_aggregation(n, _ip(Ap32), _ip(Aj), _ip(agg), _ip(cpts))
    del cpts
    lap("aggregation")
    if n_agg == 0:
        raise ValueError("aggregation produced no aggregates")
    # tentative prolongator
    Bv = np.ascontiguousarray(B.ravel(), dtype=np.float64)
    Tp = np.empty(n + 1, dtype=np.int64)
    Tj = np.empty(n, dtype=np.intc)
    Tx = np.empty(n, dtype=np.float64)
    Bc = np.empty(n_agg, dtype=np.float64)
    tnnz = L.amgsetup_tentative_scalar(n, n_agg, _ip(agg), _dp(Bv), 1e-10, _lp(Tp), _ip(Tj), _dp(Tx), _dp(Bc))
    Tj, Tx = Tj[:tnnz], Tx[:tnnz]
    # Jacobi prolongation smoothing: rho(D^-1 A), then P = T - (omega/rho) D^-1 A T
    fn, kw = unpack_arg(smooth_l)
    omega = kw.get("omega", 4.0 / 3.0)
    lap("tentative prolongator")
    # get_diagonal(A, inv=True), row-parallel on the flat arrays -- including its side effect: the reference sorts
    # A's rows in place here (util/utils.py:566), so everything from here on sees the sorted order
    if early is not None:
        early[0].join()
        if "error" in early[1]:
            raise early[1]["error"]
        D_inv, rho = early[1]["D_inv"], early[1]["rho"]
        lap("rho(D^-1 A) (overlapped)")
    else:
        if Aj.ctypes.data == A.indices.ctypes.data and Ax.ctypes.data == A.data.ctypes.data and Ax.size == A.data.size \
                and Aj.flags.writeable and Ax.flags.writeable:
            L.amgsetup_csr_sort_indices(n, _lp(Ap), _ip(Aj), _dp(Ax))
            A.has_sorted_indices = True
        else:
            A.sort_indices()
            Ap, Aj, Ax = _csr_arrays64(A)
        D_inv = np.empty(n, dtype=np.float64)
        L.amgsetup_csr_diagonal_inv(n, _lp(Ap), _ip(Aj), _dp(Ax), _dp(D_inv))
        lap("diagonal")
        rho = rho_fn(A, D_inv)
        lap("rho(D^-1 A)")
    w = omega / rho
    Pp = np.empty(n + 1, dtype=np.int64)
    dnull = C.POINTER(C.c_double)()
    inull = C.POINTER(C.c_int)()
    pnnz = L.amgsetup_smooth_prolongator(n, n_agg, _lp(Ap), _ip(Aj), _dp(Ax), _dp(D_inv), float(w), _lp(Tp),
                                         _ip(Tj), _dp(Tx), _lp(Pp), inull, dnull)
    Pj = np.empty(pnnz, dtype=np.intc)
    Px = np.empty(pnnz, dtype=np.float64)
    L.amgsetup_smooth_prolongator(n, n_agg, _lp(Ap), _ip(Aj), _dp(Ax), _dp(D_inv), float(w), _lp(Tp),
                                  _ip(Tj), _dp(Tx), _lp(Pp), _ip(Pj), _dp(Px))
    # R = P^H (real: transpose)
    Rp = np.empty(n_agg + 1, dtype=np.int64)
    Rj = np.empty(pnnz, dtype=np.intc)
    Rx = np.empty(pnnz, dtype=np.float64)
    lap("smoothed prolongator")
    L.amgsetup_csr_transpose(n, n_agg, _lp(Pp), _ip(Pj), _dp(Px), _lp(Rp), _ip(Rj), _dp(Rx))
    lap("R = P^T")
    # Galerkin product (R*A)*P: on the GPU when A already sits in HBM (the spectral-radius estimate uploaded it),
    # else row-parallel on the host -- the same products in the same order either way
    Ac = None
    if os.environ.get("AMG_SETUP_DEVICE_GALERKIN", "1") != "0":
        from .util import galerkin_device
        Ac = galerkin_device(A, (Rp, Rj, Rx), (Pp, Pj, Px), n_agg)
        if Ac is not None:
            lap("(R*A)*P on the device")
    if Ac is None:
        RA = _matmat((Rp, Rj, Rx), (Ap, Aj, Ax), (n_agg, n))
        lap("R*A")
        Ac = _matmat(RA, (Pp, Pj, Px), (n_agg, n_agg))
        del RA
        lap("(R*A)*P")
    P = _as_bsr11((Pp, Pj, Px), (n, n_agg))
    R = _as_bsr11((Rp, Rj, Rx), (n_agg, n))
    Anew = _as_bsr11(Ac, (n_agg, n_agg))
    if keep:
        levels[-1].AggOp = agg
    levels[-1].P = P
    levels[-1].R = R
    levels.append(multilevel_solver.level())
    Anew.symmetry = A.symmetry
    levels[-1].A = Anew
    levels[-1].B = Bc.reshape(-1, 1)
    lap("wrap as BSR(1,1)")


def _block_fast_path_ok(A, B, strength_l, aggregate_l, smooth_l):
    """square-block BSR operator (any number of candidates), default strength / aggregation / smoothing"""
    if not (isspmatrix_bsr(A) and A.blocksize[0] == A.blocksize[1] and 1 < A.blocksize[0] <= 16):
        return False
    return _default_options(np.empty((1, 1)), strength_l, aggregate_l, smooth_l)


def _bsr_matmat(Aa, Ba, n_brow, R, N, Cc):
    """C = A*B for BSR arrays with R x N and N x Cc blocks: scipy's bsr_matmat arithmetic and block order"""
    (Ap, Aj, Ax), (Bp, Bj, Bx) = Aa, Ba
    L = host_lib()
    Cp = np.empty(n_brow + 1, dtype=np.int64)
    nb = L.amgsetup_bsr_matmat_count(n_brow, _lp(Ap), _ip(Aj), _lp(Bp), _ip(Bj), _lp(Cp))
    Cj = np.empty(nb, dtype=np.intc)
    Cx = np.empty(nb * R * Cc, dtype=np.float64)
    L.amgsetup_bsr_matmat_fill(n_brow, R, N, Cc, _lp(Ap), _ip(Aj), _dp(Ax), _lp(Bp), _ip(Bj), _dp(Bx), _lp(Cp), _ip(Cj), _dp(Cx))
    return Cp, Cj, Cx


def _extend_block(levels, smooth_l, keep, rho_fn):
    """extend_hierarchy for a BSR(bs, bs) operator (BASELINE configuration C5: 3x3 blocks, the default bs
    candidates) on flat arrays with the host helpers -- the arithmetic and the stored block order of the generic
    scipy path (bsr_matmat, bsr_minus_bsr, bsr_transpose), row-parallel and sized for 5*10^7 unknowns."""
    A = levels[-1].A
    B = np.asarray(levels[-1].B, dtype=np.float64)
    L = host_lib()
    bs = A.blocksize[0]
    K = B.shape[1]
    n = A.shape[0]
    nb = n // bs
    verbose = os.environ.get("AMG_SETUP_VERBOSE", "0") != "0"
    _t = [time.perf_counter()]

    def lap(what):
        if verbose:
            now = time.perf_counter()
            print("[setup] level %d (%d rows, bs %d) %-22s %6.2fs" % (len(levels) - 1, n, bs, what, now - _t[0]), flush=True)
            _t[0] = now
    Ap = np.ascontiguousarray(A.indptr, dtype=np.int64)
    Ap32 = np.ascontiguousarray(A.indptr, dtype=np.intc)
    Aj = np.ascontiguousarray(A.indices, dtype=np.intc)
    Ax = np.ascontiguousarray(A.data.reshape(-1), dtype=np.float64)
    # strength with theta = 0 keeps every block (strength.py:283-287); the aggregation reads the block pattern
    agg = np.empty(nb, dtype=np.intc)
    cpts = np.empty(nb, dtype=np.intc)
    n_agg = L.amgsetup_standard_aggregation(nb, _ip(Ap32), _ip(Aj), _ip(agg), _ip(cpts))
    del cpts
    lap("aggregation")
    if n_agg == 0:
        raise ValueError("aggregation produced no aggregates")
    # tentative prolongator: per-aggregate QR of the candidates (fit_candidates' arithmetic), as one
    # bs x K block per aggregated node
    Tx, Bc = _fit_candidates_flat(agg, n_agg, B, bs)
    lap("tentative prolongator")
    fn, kw = unpack_arg(smooth_l)
    omega = kw.get("omega", 4.0 / 3.0)
    D_inv = get_diagonal(A, inv=True)
    rho = rho_fn(A, D_inv)
    lap("rho(D^-1 A)")
    w = omega / rho
    Pp = np.empty(nb + 1, dtype=np.int64)
    xs = C.c_int(0)
    inull, dnull = C.POINTER(C.c_int)(), C.POINTER(C.c_double)()
    pb = L.amgsetup_smooth_prolongator_block(nb, bs, K, _lp(Ap), _ip(Aj), _dp(Ax), _dp(D_inv), float(w), _ip(agg), _dp(Tx),
                                             _lp(Pp), inull, dnull, C.byref(xs))
    Pj = np.empty(pb, dtype=np.intc)
    Px = np.empty(pb * bs * K, dtype=np.float64)
    L.amgsetup_smooth_prolongator_block(nb, bs, K, _lp(Ap), _ip(Aj), _dp(Ax), _dp(D_inv), float(w), _ip(agg), _dp(Tx),
                                        _lp(Pp), _ip(Pj), _dp(Px), C.byref(xs))
    del Tx
    lap("smoothed prolongator")
    Rp = np.empty(n_agg + 1, dtype=np.int64)
    Rj = np.empty(pb, dtype=np.intc)
    Rx = np.empty(pb * bs * K, dtype=np.float64)
    L.amgsetup_bsr_transpose(nb, n_agg, bs, K, _lp(Pp), _ip(Pj), _dp(Px), _lp(Rp), _ip(Rj), _dp(Rx))
    lap("R = P^T")
    RA = _bsr_matmat((Rp, Rj, Rx), (Ap, Aj, Ax), n_agg, K, bs, bs)
    lap("R*A")
    Ac = _bsr_matmat(RA, (Pp, Pj, Px), n_agg, K, bs, K)
    del RA
    lap("(R*A)*P")
    if max(Pp[-1], Ac[0][-1]) >= 2 ** 31:
        raise ValueError("operator exceeds int32 indices")
    P = bsr_matrix((Px.reshape(-1, bs, K), Pj, Pp.astype(np.intc)), shape=(n, n_agg * K), copy=False)
    R = bsr_matrix((Rx.reshape(-1, K, bs), Rj, Rp.astype(np.intc)), shape=(n_agg * K, n), copy=False)
    Anew = bsr_matrix((Ac[2].reshape(-1, K, K), Ac[1], Ac[0].astype(np.intc)), shape=(n_agg * K, n_agg * K), copy=False)
    if keep:
        levels[-1].AggOp = agg
    levels[-1].P = P
    levels[-1].R = R
    levels.append(multilevel_solver.level())
    Anew.symmetry = A.symmetry
    levels[-1].A = Anew
    levels[-1].B = Bc


def _fit_candidates_flat(agg, n_agg, B, K1, tol=1e-10):
    """fit_candidates (tentative.py:19-166) on the aggregate id of every node instead of AggOp:
    -> (Tx: (n_nodes, K1, K2) blocks of the tentative prolongator, zero for unaggregated nodes; R: (n_agg*K2, K2))."""
    n_nodes = len(agg)
    K2 = B.shape[1]
    member = np.nonzero(agg >= 0)[0]
    order = member[np.argsort(agg[member], kind="stable")]          # CSC order of AggOp: by aggregate, ascending node
    counts = np.bincount(agg[member], minlength=n_agg)
    Ap = np.concatenate(([0], np.cumsum(counts)))
    Qx, R = _aggregate_qr(B.reshape(-1, K1, K2)[order], Ap, n_agg, K1, K2, tol)
    Tx = np.zeros((n_nodes, K1, K2), dtype=np.float64)
    Tx[order] = Qx
    return np.ascontiguousarray(Tx.reshape(-1)), R.reshape(-1, K2)


def _rho_D_inv_A_host(A, D_inv):
    """approximate_spectral_radius(scale_rows(A, D_inv)) as the reference does it (smooth.py:169-171);
    operators beyond util.DEVICE_RHO_MIN_ROWS run the Arnoldi iterations on the GPU."""
    if use_device_for(A):
        return approximate_spectral_radius_device(A, D_inv)
    return approximate_spectral_radius(scale_rows(A, D_inv, copy=True))


def extend_hierarchy(levels, strength, aggregate, smooth, improve_candidates, keep=True, rho_fn=None):
    """aggregation.py:293-435"""
    A = levels[-1].A
    B = levels[-1].B

    li = len(levels) - 1
    fn, kwargs = unpack_arg(improve_candidates[li])
    # (running the candidate improvement beside the aggregation was tried: both are bound by host memory bandwidth and
    # the pair took longer than one after the other)
    if fn is not None:
        B = _improve((fn, kwargs), A, B)
        levels[-1].B = B

    if not keep and _scalar_fast_path_ok(A, B, strength[li], aggregate[li], smooth[li]):
        return _extend_scalar(levels, smooth[li], keep, rho_fn or _rho_D_inv_A_host)
    if not keep and _block_fast_path_ok(A, B, strength[li], aggregate[li], smooth[li]):
        return _extend_block(levels, smooth[li], keep, rho_fn or _rho_D_inv_A_host)

    fn, kwargs = unpack_arg(strength[len(levels) - 1])
    if fn == "symmetric":
        Cm = symmetric_strength_of_connection(A, **kwargs)
    elif fn == "predefined":
        Cm = kwargs["C"].tocsr()
    elif fn is None:
        Cm = A.tocsr()
    else:
        raise NotImplementedError("strength=%r is outside the restated setup" % (fn,))

    fn, kwargs = unpack_arg(aggregate[len(levels) - 1])
    if fn == "standard":
        AggOp = standard_aggregation(Cm, **kwargs)[0]
    elif fn == "predefined":
        AggOp = kwargs["AggOp"].tocsr()
    else:
        raise NotImplementedError("aggregate=%r is outside the restated setup" % (fn,))

    T, B = fit_candidates(AggOp, B)

    fn, kwargs = unpack_arg(smooth[len(levels) - 1])
    if fn == "jacobi":
        P = jacobi_prolongation_smoother(A, T, Cm, B, **kwargs)
    elif fn is None:
        P = T
    else:
        raise NotImplementedError("smooth=%r is outside the restated setup" % (fn,))

    symmetry = A.symmetry
    R = P.conj().T.asformat(P.format) if symmetry == "hermitian" else P.T.asformat(P.format)

    if keep:
        levels[-1].C = Cm
        levels[-1].AggOp = AggOp
        levels[-1].T = T
    levels[-1].P = P
    levels[-1].R = R

    levels.append(multilevel_solver.level())
    A = R * A * P
    A.symmetry = symmetry
    levels[-1].A = A
    levels[-1].B = B
