"""Drop-in for the hot-path part of ``pyamg.amg_core`` (the SWIG module of the
reference, /root/reference/pyamg/amg_core/amg_core.i:184-195), backed by HIP
kernels on MI355X through libamgcore_hip.so.

Same call signatures as the SWIG wrappers: numpy arrays are passed whole (one
argument per C++ ``(T*, int size)`` pair) and mutated in place; the return value
is None.  Like the SWIG overload dispatcher (amg_core_wrap.cxx:9209-9215) a call
with arrays of the wrong dtype raises ``NotImplementedError``; non-contiguous
arrays raise ``TypeError``.  Only float64 values / int32 indices are provided
(the BASELINE configurations); float32/complex raise NotImplementedError.
"""
import numpy as np

from . import _lib

__all__ = ["gauss_seidel", "bsr_gauss_seidel", "jacobi", "bsr_jacobi", "gauss_seidel_indexed",
           "jacobi_ne", "gauss_seidel_ne", "gauss_seidel_nr", "block_jacobi", "block_gauss_seidel",
           "csr_matvec", "bsr_matvec", "overlapping_schwarz_csr", "extract_subblocks"]


def _chk(name, a, dtype):
    if not isinstance(a, np.ndarray):
        raise TypeError("%s: numpy array expected" % name)
    if a.dtype != dtype:
        raise NotImplementedError(
            "Wrong number or type of arguments for overloaded function '%s' "
            "(float64 values / int32 indices only)" % name)
    if a.ndim != 1:
        raise ValueError("%s: array must have 1 dimension" % name)
    if not a.flags.c_contiguous:
        raise TypeError("%s: array must be contiguous" % name)
    return a


def _I(name, a):
    return _chk(name, a, np.intc)


def _D(name, a):
    return _chk(name, a, np.float64)


def _csr(name, Ap, Aj, Ax):
    Ap, Aj, Ax = _I(name, Ap), _I(name, Aj), _D(name, Ax)
    return [_lib.ip(Ap), len(Ap), _lib.ip(Aj), len(Aj), _lib.dp(Ax), len(Ax)]


def gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step):
    n = "gauss_seidel"
    x, b = _D(n, x), _D(n, b)
    _lib.check(_lib.lib().amgcore_gauss_seidel_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b),
                                                   len(b), int(row_start), int(row_stop), int(row_step)))


def bsr_gauss_seidel(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, blocksize):
    n = "bsr_gauss_seidel"
    x, b = _D(n, x), _D(n, b)
    _lib.check(_lib.lib().amgcore_bsr_gauss_seidel_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b),
                                                       len(b), int(row_start), int(row_stop), int(row_step),
                                                       int(blocksize)))


def jacobi(Ap, Aj, Ax, x, b, temp, row_start, row_stop, row_step, omega):
    n = "jacobi"
    x, b, temp, omega = _D(n, x), _D(n, b), _D(n, temp), _D(n, omega)
    _lib.check(_lib.lib().amgcore_jacobi_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b), len(b),
                                             _lib.dp(temp), len(temp), int(row_start), int(row_stop),
                                             int(row_step), _lib.dp(omega), len(omega)))


def bsr_jacobi(Ap, Aj, Ax, x, b, temp, row_start, row_stop, row_step, blocksize, omega):
    n = "bsr_jacobi"
    x, b, temp, omega = _D(n, x), _D(n, b), _D(n, temp), _D(n, omega)
    _lib.check(_lib.lib().amgcore_bsr_jacobi_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b), len(b),
                                                 _lib.dp(temp), len(temp), int(row_start), int(row_stop),
                                                 int(row_step), int(blocksize), _lib.dp(omega), len(omega)))


def gauss_seidel_indexed(Ap, Aj, Ax, x, b, Id, row_start, row_stop, row_step):
    n = "gauss_seidel_indexed"
    x, b, Id = _D(n, x), _D(n, b), _I(n, Id)
    _lib.check(_lib.lib().amgcore_gauss_seidel_indexed_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x),
                                                           _lib.dp(b), len(b), _lib.ip(Id), len(Id),
                                                           int(row_start), int(row_stop), int(row_step)))


def jacobi_ne(Ap, Aj, Ax, x, b, Tx, temp, row_start, row_stop, row_step, omega):
    n = "jacobi_ne"
    x, b, Tx, temp, omega = _D(n, x), _D(n, b), _D(n, Tx), _D(n, temp), _D(n, omega)
    _lib.check(_lib.lib().amgcore_jacobi_ne_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b), len(b),
                                                _lib.dp(Tx), len(Tx), _lib.dp(temp), len(temp),
                                                int(row_start), int(row_stop), int(row_step),
                                                _lib.dp(omega), len(omega)))


def gauss_seidel_ne(Ap, Aj, Ax, x, b, row_start, row_stop, row_step, Tx, omega):
    n = "gauss_seidel_ne"
    x, b, Tx = _D(n, x), _D(n, b), _D(n, Tx)
    _lib.check(_lib.lib().amgcore_gauss_seidel_ne_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b),
                                                      len(b), int(row_start), int(row_stop), int(row_step),
                                                      _lib.dp(Tx), len(Tx), float(omega)))


def gauss_seidel_nr(Ap, Aj, Ax, x, z, col_start, col_stop, col_step, Tx, omega):
    n = "gauss_seidel_nr"
    x, z, Tx = _D(n, x), _D(n, z), _D(n, Tx)
    _lib.check(_lib.lib().amgcore_gauss_seidel_nr_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(z),
                                                      len(z), int(col_start), int(col_stop), int(col_step),
                                                      _lib.dp(Tx), len(Tx), float(omega)))


def block_jacobi(Ap, Aj, Ax, x, b, Tx, temp, row_start, row_stop, row_step, omega, blocksize):
    n = "block_jacobi"
    x, b, Tx, temp, omega = _D(n, x), _D(n, b), _D(n, Tx), _D(n, temp), _D(n, omega)
    _lib.check(_lib.lib().amgcore_block_jacobi_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b),
                                                   len(b), _lib.dp(Tx), len(Tx), _lib.dp(temp), len(temp),
                                                   int(row_start), int(row_stop), int(row_step),
                                                   _lib.dp(omega), len(omega), int(blocksize)))


def block_gauss_seidel(Ap, Aj, Ax, x, b, Tx, row_start, row_stop, row_step, blocksize):
    n = "block_gauss_seidel"
    x, b, Tx = _D(n, x), _D(n, b), _D(n, Tx)
    _lib.check(_lib.lib().amgcore_block_gauss_seidel_f64(*_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b),
                                                         len(b), _lib.dp(Tx), len(Tx), int(row_start),
                                                         int(row_stop), int(row_step), int(blocksize)))


def csr_matvec(n_row, n_col, Ap, Aj, Ax, Xx, Yx):
    """scipy.sparse._sparsetools.csr_matvec signature: Yx += A * Xx."""
    n = "csr_matvec"
    Ap, Aj, Ax, Xx, Yx = _I(n, Ap), _I(n, Aj), _D(n, Ax), _D(n, Xx), _D(n, Yx)
    _lib.check(_lib.lib().amgcore_csr_matvec_f64(int(n_row), int(n_col), _lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax),
                                                 _lib.dp(Xx), _lib.dp(Yx)))


def bsr_matvec(n_brow, n_bcol, R, C, Ap, Aj, Ax, Xx, Yx):
    """scipy.sparse._sparsetools.bsr_matvec signature: Yx += A * Xx."""
    n = "bsr_matvec"
    Ap, Aj, Ax, Xx, Yx = _I(n, Ap), _I(n, Aj), _D(n, Ax), _D(n, Xx), _D(n, Yx)
    _lib.check(_lib.lib().amgcore_bsr_matvec_f64(int(n_brow), int(n_bcol), int(R), int(C), _lib.ip(Ap),
                                                 _lib.ip(Aj), _lib.dp(Ax), _lib.dp(Xx), _lib.dp(Yx)))


def overlapping_schwarz_csr(Ap, Aj, Ax, x, b, Tx, Tp, Sj, Sp, nsdomains, nrows, row_start, row_stop, row_step):
    """relaxation.h:935-1007: one sweep of multiplicative overlapping Schwarz (HIP, by dependency levels)"""
    n = "overlapping_schwarz_csr"
    x, b, Tx = _D(n, x), _D(n, b), _D(n, Tx)
    Tp, Sj, Sp = _I(n, Tp), _I(n, Sj), _I(n, Sp)
    _lib.check(_lib.lib().amgcore_overlapping_schwarz_csr_f64(
        *_csr(n, Ap, Aj, Ax), _lib.dp(x), len(x), _lib.dp(b), len(b), _lib.dp(Tx), len(Tx), _lib.ip(Tp), len(Tp),
        _lib.ip(Sj), len(Sj), _lib.ip(Sp), len(Sp), int(nsdomains), int(nrows), int(row_start), int(row_stop),
        int(row_step)))


def extract_subblocks(Ap, Aj, Ax, Tx, Tp, Sj, Sp, nsdomains, nrows):
    """relaxation.h:836-899 (setup helper of the Schwarz smoother; runs on the host)"""
    from .aggregation import host_lib
    n = "extract_subblocks"
    Ap, Aj, Ax, Tx = _I(n, Ap), _I(n, Aj), _D(n, Ax), _D(n, Tx)
    Tp, Sj, Sp = _I(n, Tp), _I(n, Sj), _I(n, Sp)
    if len(Sp) < nsdomains + 1 or len(Tp) < nsdomains + 1 or (nsdomains and len(Tx) < Tp[nsdomains]):
        raise ValueError("extract_subblocks: pointer arrays too short")
    host_lib().amgsetup_extract_subblocks(_lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax), _lib.dp(Tx), _lib.ip(Tp),
                                          _lib.ip(Sj), _lib.ip(Sp), int(nsdomains), int(nrows))
