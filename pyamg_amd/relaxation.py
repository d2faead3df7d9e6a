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


_SWEEP_SIGNS = {"forward": (1,), "backward": (-1,), "symmetric": (1, -1)}


def _sweep_ranges(sweep, count):
    """The (start, stop, step) row ranges ONE iteration of a directional sweep visits, in order:
    a symmetric sweep is a forward range followed by a backward one (relaxation.py:334-343)."""
    signs = _SWEEP_SIGNS.get(sweep) if isinstance(sweep, str) else None
    if signs is None:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    return [(0, count, 1) if s > 0 else (count - 1, -1, -1) for s in signs]


def _to_accepted_format(A, formats):
    """The storage conversions the reference's shims apply before validating (relaxation.py:66-83)."""
    if formats is None:
        return A
    if sparse.isspmatrix(A) and A.format in formats:
        return A
    if formats == ["csr"]:
        if sparse.isspmatrix_bsr(A):
            return A.tocsr()
        warn("implicit conversion to CSR", sparse.SparseEfficiencyWarning)
        return sparse.csr_matrix(A)
    return sparse.csr_matrix(A).asformat(formats[0])


def make_system(A, x, b, formats=None):
    """Validate (A, x, b) for a relaxation call and return them with x, b flattened -- the error
    contract of relaxation.py:21-105 (exception types and messages are what
    relaxation/tests/test_relaxation.py:52-102 pins), stated as a table of requirements."""
    A = _to_accepted_format(A, formats)
    rows, cols = A.shape
    vec_shapes = ((rows,), (rows, 1))
    requirements = (
        (lambda: isinstance(x, np.ndarray), ValueError, "expected numpy array for argument x"),
        (lambda: isinstance(b, np.ndarray), ValueError, "expected numpy array for argument b"),
        (lambda: rows == cols, ValueError, "expected square matrix"),
        (lambda: x.shape in vec_shapes, ValueError, "x has invalid dimensions"),
        (lambda: b.shape in vec_shapes, ValueError, "b has invalid dimensions"),
        (lambda: A.dtype == x.dtype == b.dtype, TypeError, "arguments A, x, and b must have the same dtype"),
        (lambda: x.flags.carray, ValueError, "x must be contiguous in memory"),
    )
    for holds, exc, message in requirements:
        if not holds():
            raise exc(message)
    return A, np.ravel(x), np.ravel(b)


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


def _square_blocksize(A):
    """1 for CSR, R for BSR with square R x R blocks"""
    if sparse.isspmatrix_csr(A):
        return 1
    R, C = A.blocksize
    if R != C:
        raise ValueError("BSR blocks must be square")
    return R


def sor(A, x, b, omega, iterations=1, sweep="forward"):
    """Successive over-relaxation, x updated in place (relaxation.py:108-169): every iteration is a
    Gauss-Seidel pass blended with the iterate it started from, x <- omega*x_gs + (1-omega)*x_start,
    formed as two scalings and one addition."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    for _ in range(iterations):
        start = x.copy()
        gauss_seidel(A, x, b, iterations=1, sweep=sweep)
        np.multiply(x, omega, out=x)
        np.multiply(start, 1 - omega, out=start)
        np.add(x, start, out=x)


def gauss_seidel(A, x, b, iterations=1, sweep="forward"):
    """Gauss-Seidel relaxation, x updated in place (relaxation.py:280-354 over
    amg_core/relaxation.h:34-62 for CSR, :90-173 for BSR -- block rows are counted there)."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    bs = _square_blocksize(A)
    ranges = _sweep_ranges(sweep, len(x) // bs)
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Ax = np.ascontiguousarray(np.ravel(A.data))
    for _ in range(iterations):
        for start, stop, step in ranges:
            if bs == 1 and sparse.isspmatrix_csr(A):
                amg_core.gauss_seidel(Ap, Aj, Ax, x, b, start, stop, step)
            else:
                amg_core.bsr_gauss_seidel(Ap, Aj, Ax, x, b, start, stop, step, bs)


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
    Dinv = _block_inverse(A, blocksize, Dinv)
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


def _block_inverse(A, blocksize, Dinv):
    """Dinv as handed in (shape-checked) or the inverted diagonal blocks of A (relaxation.py:473-479)."""
    if Dinv is None:
        return get_block_diag(A, blocksize=blocksize, inv_flag=True)
    if Dinv.shape[0] != A.shape[0] // blocksize:
        raise ValueError("Dinv and A have incompatible dimensions")
    if Dinv.shape[1:] != (blocksize, blocksize):
        raise ValueError("Dinv and blocksize are incompatible")
    return Dinv


def block_gauss_seidel(A, x, b, iterations=1, sweep="forward", blocksize=1, Dinv=None):
    """Block Gauss-Seidel, x updated in place (relaxation.py:509-590 over relaxation.h:756-810):
    A is re-blocked to (blocksize, blocksize), block rows are swept in order and each one is
    solved with the stored inverse of its diagonal block."""
    A, x, b = make_system(A, x, b, formats=["csr", "bsr"])
    A = A.tobsr(blocksize=(blocksize, blocksize))
    Dinv = _block_inverse(A, blocksize, Dinv)
    ranges = _sweep_ranges(sweep, len(x) // blocksize)
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Ax = np.ascontiguousarray(np.ravel(A.data))
    Dflat = np.ascontiguousarray(np.ravel(Dinv))
    for _ in range(iterations):
        for start, stop, step in ranges:
            amg_core.block_gauss_seidel(Ap, Aj, Ax, x, b, Dflat, start, stop, step, blocksize)


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
    """Gauss-Seidel over the rows listed in `indices`, in list order (relaxation.py:671-741 over
    relaxation.h:395-430); the sweep ranges count positions of the list."""
    A, x, b = make_system(A, x, b, formats=["csr"])
    indices = np.ascontiguousarray(np.asarray(indices, dtype="intc"))
    ranges = _sweep_ranges(sweep, len(indices))
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Ax = np.ascontiguousarray(A.data)
    for _ in range(iterations):
        for start, stop, step in ranges:
            amg_core.gauss_seidel_indexed(Ap, Aj, Ax, x, b, indices, start, stop, step)


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
    """Gauss-Seidel on A A^H y = b with x = A^H y (Kaczmarz row projections), x updated in place
    (relaxation.py:821-908 over relaxation.h:529-560)."""
    A, x, b = make_system(A, x, b, formats=["csr"])
    if Dinv is None:
        Dinv = np.ravel(get_diagonal(A, norm_eq=2, inv=True))
    ranges = _sweep_ranges(sweep, len(x))
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Ax = np.ascontiguousarray(A.data)
    Dinv = np.ascontiguousarray(Dinv, dtype=np.float64)
    for _ in range(iterations):
        for start, stop, step in ranges:
            amg_core.gauss_seidel_ne(Ap, Aj, Ax, x, b, start, stop, step, Dinv, omega)


def gauss_seidel_nr(A, x, b, iterations=1, sweep="forward", omega=1.0, Dinv=None):
    """Gauss-Seidel on A^H A x = A^H b by columns of A, x updated in place (relaxation.py:911-997
    over relaxation.h:594-625).  The kernel keeps r = b - A x current while it sweeps; the
    reference forms r once per directional CALL, and its symmetric sweep is a forward call
    followed by a backward call -- so r is re-formed before every direction of a symmetric sweep
    and once for all iterations of a one-directional one."""
    A, x, b = make_system(A, x, b, formats=["csc"])
    if Dinv is None:
        Dinv = np.ravel(get_diagonal(A, norm_eq=1, inv=True))
    ranges = _sweep_ranges(sweep, len(x))
    # scipy's csc_matvec adds the terms of each output entry in ascending column order -- the
    # order of a CSR row with sorted indices, which the device CSR kernel reproduces bit for bit
    Acsr = A.tocsr()
    Acsr.sort_indices()
    Ap, Aj = _ptr(A)
    Ax = np.ascontiguousarray(A.data)
    Dinv = np.ascontiguousarray(Dinv, dtype=np.float64)

    def residual():
        return np.ascontiguousarray(b - _spmv(Acsr, x))

    if len(ranges) == 1:
        (start, stop, step), r = ranges[0], residual()
        for _ in range(iterations):
            amg_core.gauss_seidel_nr(Ap, Aj, Ax, x, r, start, stop, step, Dinv, omega)
        return
    for _ in range(iterations):
        for start, stop, step in ranges:
            amg_core.gauss_seidel_nr(Ap, Aj, Ax, x, residual(), start, stop, step, Dinv, omega)


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
    ranges = _sweep_ranges(sweep, nsd)
    Ap, Aj = _ptr(A)
    b = _bvec(b)
    Ax = np.ascontiguousarray(A.data)
    for _ in range(iterations):
        for start, stop, step in ranges:
            amg_core.overlapping_schwarz_csr(Ap, Aj, Ax, x, b, inv_subblock, inv_subblock_ptr, subdomain,
                                             subdomain_ptr, nsd, A.shape[0], start, stop, step)


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
