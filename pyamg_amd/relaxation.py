"""Relaxation methods -- host-side mirror of
/root/reference/pyamg/relaxation/relaxation.py with the native kernels replaced
by the HIP library (``pyamg_amd.amg_core``).

Same function names, argument meaning, in-place semantics and error behaviour
(``make_system`` checks, relaxation.py:21-105).  Each call here stages its
arrays through PCIe -- it is the drop-in for code that calls the reference's
relaxation functions directly.  The multigrid cycle does NOT come through
here: ``multilevel_solver.solve`` keeps the whole hierarchy resident in HBM.
"""
from warnings import warn

import numpy as np
from scipy import sparse

from . import amg_core
from .util import get_block_diag, get_diagonal, type_prep

__all__ = ["sor", "gauss_seidel", "jacobi", "polynomial", "jacobi_ne", "gauss_seidel_ne",
           "gauss_seidel_nr", "gauss_seidel_indexed", "block_jacobi", "block_gauss_seidel",
           "make_system", "schwarz", "schwarz_parameters"]


def make_system(A, x, b, formats=None):
    """relaxation.py:21-105"""
    if formats is None:
        pass
    elif formats == ["csr"]:
        if sparse.isspmatrix_csr(A):
            pass
        elif sparse.isspmatrix_bsr(A):
            A = A.tocsr()
        else:
            warn("implicit conversion to CSR", sparse.SparseEfficiencyWarning)
            A = sparse.csr_matrix(A)
    else:
        if sparse.isspmatrix(A) and A.format in formats:
            pass
        else:
            A = sparse.csr_matrix(A).asformat(formats[0])

    if not isinstance(x, np.ndarray):
        raise ValueError("expected numpy array for argument x")
    if not isinstance(b, np.ndarray):
        raise ValueError("expected numpy array for argument b")
    M, N = A.shape
    if M != N:
        raise ValueError("expected square matrix")
    if x.shape not in [(M,), (M, 1)]:
        raise ValueError("x has invalid dimensions")
    if b.shape not in [(M,), (M, 1)]:
        raise ValueError("b has invalid dimensions")
    if A.dtype != x.dtype or A.dtype != b.dtype:
        raise TypeError("arguments A, x, and b must have the same dtype")
    if not x.flags.carray:
        raise ValueError("x must be contiguous in memory")
    x = np.ravel(x)
    b = np.ravel(b)
    return A, x, b


def _ptr(A):
    return (np.ascontiguousarray(A.indptr, dtype=np.intc), np.ascontiguousarray(A.indices, dtype=np.intc))


def _spmv(A, x):
    """``A * x`` (scipy csr/bsr matvec arithmetic) on the device."""
    y = np.zeros(A.shape[0], dtype=np.float64)
    Ap, Aj = _ptr(A)
    x = np.ascontiguousarray(np.ravel(x), dtype=np.float64)
    if sparse.isspmatrix_bsr(A):
        R, C = A.blocksize
        amg_core.bsr_matvec(A.shape[0] // R, A.shape[1] // C, R, C, Ap, Aj,
                            np.ascontiguousarray(np.ravel(A.data)), x, y)
    else:
        A = sparse.csr_matrix(A)
        Ap, Aj = _ptr(A)
        amg_core.csr_matvec(A.shape[0], A.shape[1], Ap, Aj, np.ascontiguousarray(A.data), x, y)
    return y


def _bvec(b):
    return np.ascontiguousarray(b)


def sor(A, x, b, omega, iterations=1, sweep="forward"):
    """relaxation.py:108-169"""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    x_old = np.empty_like(x)
    for i in range(iterations):
        x_old[:] = x
        gauss_seidel(A, x, b, iterations=1, sweep=sweep)
        x *= omega
        x_old *= (1 - omega)
        x += x_old


def gauss_seidel(A, x, b, iterations=1, sweep="forward"):
    """relaxation.py:280-354"""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    if sparse.isspmatrix_csr(A):
        blocksize = 1
    else:
        R, C = A.blocksize
        if R != C:
            raise ValueError("BSR blocks must be square")
        blocksize = R
    if sweep == "forward":
        row_start, row_stop, row_step = 0, int(len(x) / blocksize), 1
    elif sweep == "backward":
        row_start, row_stop, row_step = int(len(x) / blocksize) - 1, -1, -1
    elif sweep == "symmetric":
        for it in range(iterations):
            gauss_seidel(A, x, b, iterations=1, sweep="forward")
            gauss_seidel(A, x, b, iterations=1, sweep="backward")
        return
    else:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    if sparse.isspmatrix_csr(A):
        for it in range(iterations):
            amg_core.gauss_seidel(Ap, Aj, np.ascontiguousarray(A.data), x, b, row_start, row_stop, row_step)
    else:
        for it in range(iterations):
            amg_core.bsr_gauss_seidel(Ap, Aj, np.ascontiguousarray(np.ravel(A.data)), x, b, row_start,
                                      row_stop, row_step, R)


def jacobi(A, x, b, iterations=1, omega=1.0):
    """relaxation.py:357-427"""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    sweep = slice(None)
    (row_start, row_stop, row_step) = sweep.indices(A.shape[0])
    if (row_stop - row_start) * row_step <= 0:
        return
    temp = np.empty_like(x)
    [omega] = type_prep(A.dtype, [omega])
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    if sparse.isspmatrix_csr(A):
        for it in range(iterations):
            amg_core.jacobi(Ap, Aj, np.ascontiguousarray(A.data), x, b, temp, row_start, row_stop, row_step, omega)
    else:
        R, C = A.blocksize
        if R != C:
            raise ValueError("BSR blocks must be square")
        row_start = int(row_start / R)
        row_stop = int(row_stop / R)
        for it in range(iterations):
            amg_core.bsr_jacobi(Ap, Aj, np.ascontiguousarray(np.ravel(A.data)), x, b, temp, row_start,
                                row_stop, row_step, R, omega)


def block_jacobi(A, x, b, Dinv=None, blocksize=1, iterations=1, omega=1.0):
    """relaxation.py:430-506"""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    A = A.tobsr(blocksize=(blocksize, blocksize))
    if Dinv is None:
        Dinv = get_block_diag(A, blocksize=blocksize, inv_flag=True)
    elif Dinv.shape[0] != int(A.shape[0] / blocksize):
        raise ValueError("Dinv and A have incompatible dimensions")
    elif (Dinv.shape[1] != blocksize) or (Dinv.shape[2] != blocksize):
        raise ValueError("Dinv and blocksize are incompatible")
    sweep = slice(None)
    (row_start, row_stop, row_step) = sweep.indices(int(A.shape[0] / blocksize))
    if (row_stop - row_start) * row_step <= 0:
        return
    temp = np.empty_like(x)
    [omega] = type_prep(A.dtype, [omega])
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    for it in range(iterations):
        amg_core.block_jacobi(Ap, Aj, np.ascontiguousarray(np.ravel(A.data)), x, b,
                              np.ascontiguousarray(np.ravel(Dinv)), temp, row_start, row_stop, row_step,
                              omega, blocksize)


def block_gauss_seidel(A, x, b, iterations=1, sweep="forward", blocksize=1, Dinv=None):
    """relaxation.py:509-590"""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    A = A.tobsr(blocksize=(blocksize, blocksize))
    if Dinv is None:
        Dinv = get_block_diag(A, blocksize=blocksize, inv_flag=True)
    elif Dinv.shape[0] != int(A.shape[0] / blocksize):
        raise ValueError("Dinv and A have incompatible dimensions")
    elif (Dinv.shape[1] != blocksize) or (Dinv.shape[2] != blocksize):
        raise ValueError("Dinv and blocksize are incompatible")
    if sweep == "forward":
        row_start, row_stop, row_step = 0, int(len(x) / blocksize), 1
    elif sweep == "backward":
        row_start, row_stop, row_step = int(len(x) / blocksize) - 1, -1, -1
    elif sweep == "symmetric":
        for it in range(iterations):
            block_gauss_seidel(A, x, b, iterations=1, sweep="forward", blocksize=blocksize, Dinv=Dinv)
            block_gauss_seidel(A, x, b, iterations=1, sweep="backward", blocksize=blocksize, Dinv=Dinv)
        return
    else:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    for it in range(iterations):
        amg_core.block_gauss_seidel(Ap, Aj, np.ascontiguousarray(np.ravel(A.data)), x, b,
                                    np.ascontiguousarray(np.ravel(Dinv)), row_start, row_stop, row_step,
                                    blocksize)


def polynomial(A, x, b, coefficients, iterations=1):
    """relaxation.py:593-668"""
    A, x, b = make_system(A, x, b, formats=None)
    for i in range(iterations):
        if not np.any(x):       # norm(x) == 0
            residual = b
        else:
            residual = (b - _spmv(A, x))
        h = coefficients[0] * residual
        for c in coefficients[1:]:
            h = c * residual + _spmv(A, h)
        x += h


def gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="forward"):
    """relaxation.py:671-741"""
    A, x, b = make_system(A, x, b, formats=["csr"])
    indices = np.asarray(indices, dtype="intc")
    if sweep == "forward":
        row_start, row_stop, row_step = 0, len(indices), 1
    elif sweep == "backward":
        row_start, row_stop, row_step = len(indices) - 1, -1, -1
    elif sweep == "symmetric":
        for it in range(iterations):
            gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="forward")
            gauss_seidel_indexed(A, x, b, indices, iterations=1, sweep="backward")
        return
    else:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    indices = np.ascontiguousarray(indices)
    for it in range(iterations):
        amg_core.gauss_seidel_indexed(Ap, Aj, np.ascontiguousarray(A.data), x, b, indices, row_start,
                                      row_stop, row_step)


def jacobi_ne(A, x, b, iterations=1, omega=1.0):
    """relaxation.py:744-818"""
    A, x, b = make_system(A, x, b, formats=["csr"])
    sweep = slice(None)
    (row_start, row_stop, row_step) = sweep.indices(A.shape[0])
    temp = np.zeros_like(x)
    Dinv = get_diagonal(A, norm_eq=2, inv=True)
    [omega] = type_prep(A.dtype, [omega])
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    for i in range(iterations):
        delta = (np.ravel(b - _spmv(A, x)) * np.ravel(Dinv)).astype(A.dtype)
        amg_core.jacobi_ne(Ap, Aj, np.ascontiguousarray(A.data), x, b, delta, temp, row_start, row_stop,
                           row_step, omega)


def gauss_seidel_ne(A, x, b, iterations=1, sweep="forward", omega=1.0, Dinv=None):
    """relaxation.py:821-908"""
    A, x, b = make_system(A, x, b, formats=["csr"])
    if Dinv is None:
        Dinv = np.ravel(get_diagonal(A, norm_eq=2, inv=True))
    if sweep == "forward":
        row_start, row_stop, row_step = 0, len(x), 1
    elif sweep == "backward":
        row_start, row_stop, row_step = len(x) - 1, -1, -1
    elif sweep == "symmetric":
        for it in range(iterations):
            gauss_seidel_ne(A, x, b, iterations=1, sweep="forward", omega=omega, Dinv=Dinv)
            gauss_seidel_ne(A, x, b, iterations=1, sweep="backward", omega=omega, Dinv=Dinv)
        return
    else:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Dinv = np.ascontiguousarray(Dinv, dtype=np.float64)
    for i in range(iterations):
        amg_core.gauss_seidel_ne(Ap, Aj, np.ascontiguousarray(A.data), x, b, row_start, row_stop, row_step,
                                 Dinv, omega)


def gauss_seidel_nr(A, x, b, iterations=1, sweep="forward", omega=1.0, Dinv=None):
    """relaxation.py:911-997"""
    A, x, b = make_system(A, x, b, formats=["csc"])
    if Dinv is None:
        Dinv = np.ravel(get_diagonal(A, norm_eq=1, inv=True))
    if sweep == "forward":
        col_start, col_stop, col_step = 0, len(x), 1
    elif sweep == "backward":
        col_start, col_stop, col_step = len(x) - 1, -1, -1
    elif sweep == "symmetric":
        for it in range(iterations):
            gauss_seidel_nr(A, x, b, iterations=1, sweep="forward", omega=omega, Dinv=Dinv)
            gauss_seidel_nr(A, x, b, iterations=1, sweep="backward", omega=omega, Dinv=Dinv)
        return
    else:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    # initial residual (relaxation.py:992).  scipy's csc_matvec adds the terms of each
    # output entry in ascending column order -- the same order as a CSR row with sorted
    # indices, so the device CSR kernel reproduces it bit for bit.
    Acsr = A.tocsr()
    Acsr.sort_indices()
    r = np.ascontiguousarray(b - _spmv(Acsr, x))
    Ap, Aj = _ptr(A)
    Dinv = np.ascontiguousarray(Dinv, dtype=np.float64)
    for i in range(iterations):
        amg_core.gauss_seidel_nr(Ap, Aj, np.ascontiguousarray(A.data), x, r, col_start, col_stop, col_step,
                                 Dinv, omega)


def schwarz(A, x, b, iterations=1, subdomain=None, subdomain_ptr=None, inv_subblock=None,
            inv_subblock_ptr=None, sweep="forward"):
    """Overlapping multiplicative Schwarz (relaxation.py:172-278).  Subdomains default to the rows'
    sparsity patterns; x is modified in place."""
    A, x, b = make_system(A, x, b, formats=["csr"])
    if subdomain is None and inv_subblock is not None:
        raise ValueError("inv_subblock must be None if subdomain is None")
    (subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr) = \
        schwarz_parameters(A, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr)
    nsd = subdomain_ptr.shape[0] - 1
    if sweep == "forward":
        row_start, row_stop, row_step = 0, nsd, 1
    elif sweep == "backward":
        row_start, row_stop, row_step = nsd - 1, -1, -1
    elif sweep == "symmetric":
        for _ in range(iterations):
            schwarz(A, x, b, iterations=1, subdomain=subdomain, subdomain_ptr=subdomain_ptr,
                    inv_subblock=inv_subblock, inv_subblock_ptr=inv_subblock_ptr, sweep="forward")
            schwarz(A, x, b, iterations=1, subdomain=subdomain, subdomain_ptr=subdomain_ptr,
                    inv_subblock=inv_subblock, inv_subblock_ptr=inv_subblock_ptr, sweep="backward")
        return
    else:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Ax = np.ascontiguousarray(A.data)
    for _ in range(iterations):
        amg_core.overlapping_schwarz_csr(Ap, Aj, Ax, x, b, inv_subblock, inv_subblock_ptr, subdomain,
                                         subdomain_ptr, nsd, A.shape[0], row_start, row_stop, row_step)


def schwarz_parameters(A, subdomain=None, subdomain_ptr=None, inv_subblock=None, inv_subblock_ptr=None):
    """relaxation.py:1011-1083: subdomains (default: A's sparsity pattern) and the pseudo-inverses of
    their diagonal blocks, cached on A as ``A.schwarz_parameters``.  The reference inverts block by
    block with LAPACK gelss (cond = eps*1e6); here the blocks of equal size are inverted together
    with numpy's batched SVD pseudo-inverse and the same cut-off."""
    if hasattr(A, "schwarz_parameters"):
        if subdomain is not None and subdomain_ptr is not None:
            if np.array(A.schwarz_parameters[0] == subdomain).all() and \
               np.array(A.schwarz_parameters[1] == subdomain_ptr).all():
                return A.schwarz_parameters
        else:
            return A.schwarz_parameters
    if subdomain is None or subdomain_ptr is None:
        subdomain_ptr = A.indptr.copy()
        subdomain = A.indices.copy()
    subdomain = np.ascontiguousarray(subdomain, dtype=np.intc)
    subdomain_ptr = np.ascontiguousarray(subdomain_ptr, dtype=np.intc)
    if inv_subblock is None or inv_subblock_ptr is None:
        nsd = subdomain_ptr.shape[0] - 1
        blocksize = (subdomain_ptr[1:] - subdomain_ptr[:-1]).astype(np.int64)
        ptr64 = np.zeros(nsd + 1, dtype=np.int64)
        ptr64[1:] = np.cumsum(blocksize * blocksize)
        if ptr64[-1] > np.iinfo(np.intc).max:
            raise ValueError("subdomain blocks need more than 2^31-1 entries")
        inv_subblock_ptr = ptr64.astype(np.intc)
        inv_subblock = np.zeros((int(ptr64[-1]),), dtype=np.float64)
        Ap, Aj = _ptr(A)
        amg_core.extract_subblocks(Ap, Aj, np.ascontiguousarray(A.data, dtype=np.float64), inv_subblock,
                                   inv_subblock_ptr, subdomain, subdomain_ptr, nsd, A.shape[0])
        cond = np.finfo(np.float64).eps * 1e6
        for m in np.unique(blocksize):
            m = int(m)
            if m == 0:
                continue
            which = np.nonzero(blocksize == m)[0]
            idx = ptr64[which][:, None] + np.arange(m * m)[None, :]
            blocks = inv_subblock[idx].reshape(len(which), m, m)
            inv_subblock[idx] = np.linalg.pinv(blocks, rcond=cond).reshape(len(which), m * m)
    else:
        inv_subblock = np.ascontiguousarray(inv_subblock, dtype=np.float64)
        inv_subblock_ptr = np.ascontiguousarray(inv_subblock_ptr, dtype=np.intc)
    A.schwarz_parameters = (subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr)
    return A.schwarz_parameters
