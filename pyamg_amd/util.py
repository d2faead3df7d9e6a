"""Host-side helpers the relaxation shims and smoother setup need
(restated from /root/reference/pyamg/util/utils.py and util/linalg.py).

These produce per-level CONSTANTS (omega scaling, Chebyshev bounds, inverse
diagonal blocks) once at setup; nothing here runs inside the cycle.
"""
import numpy as np
import scipy.linalg
from scipy.sparse import bsr_matrix, csr_matrix, isspmatrix, isspmatrix_bsr, isspmatrix_csc, isspmatrix_csr

DEVICE_RHO_MIN_ROWS = 200000   # spectral-radius estimates of larger operators run on the GPU

__all__ = ["type_prep", "to_type", "get_diagonal", "get_block_diag", "scale_rows", "norm",
           "approximate_spectral_radius", "upcast"]


def upcast(*args):
    return np.result_type(*args)


def to_type(upcast_type, varlist):
    """util/utils.py:475-523"""
    out = list(varlist)
    for i, v in enumerate(out):
        if np.isscalar(v):
            out[i] = np.array([v], upcast_type)[0]
        else:
            try:
                if v.dtype != upcast_type:
                    out[i] = v.astype(upcast_type)
            except AttributeError:
                out[i] = np.asarray(v).astype(upcast_type)
    return out


def type_prep(upcast_type, varlist):
    """util/utils.py:431-472: scalars become length-1 arrays."""
    out = to_type(upcast_type, varlist)
    for i, v in enumerate(out):
        if np.isscalar(v):
            out[i] = np.array([v])
    return out


def norm(x, pnorm="2"):
    """util/linalg.py:17-58 (host vectors; device vectors use amgcore_norm2)."""
    x = np.ravel(x)
    if pnorm == "2":
        return np.sqrt(np.inner(x.conj(), x).real)
    if pnorm == "inf":
        return np.max(np.abs(x))
    raise ValueError("Only the 2-norm and infinity-norm are supported")


def get_diagonal(A, norm_eq=False, inv=False):
    """util/utils.py:526-588"""
    if not (isspmatrix_csr(A) or isspmatrix_csc(A) or isspmatrix_bsr(A)):
        A = csr_matrix(A)
    A.sort_indices()
    if norm_eq == 1:
        At = A.T
        D = (At.multiply(At.conjugate())) * np.ones((At.shape[0],))
    elif norm_eq == 2:
        D = (A.multiply(A.conjugate())) * np.ones((A.shape[0],))
    else:
        D = A.diagonal()
    D = np.asarray(D).ravel()
    if inv:
        Dinv = np.zeros_like(D)
        mask = (D != 0.0)
        Dinv[mask] = 1.0 / D[mask]
        return Dinv
    return D


def get_block_diag(A, blocksize, inv_flag=True):
    """util/utils.py:591-683.  The reference inverts the blocks with its own
    SVD routine (amg_core.pinv_array); LAPACK's pinv is used here -- the result
    is a setup constant handed to the smoother as Dinv."""
    if not isspmatrix(A):
        raise TypeError("Expected sparse matrix")
    if A.shape[0] != A.shape[1]:
        raise ValueError("Expected square matrix")
    if A.shape[0] % blocksize != 0:
        raise ValueError("blocksize and A.shape must be compatible")
    if not isspmatrix_bsr(A):
        A = bsr_matrix(A, blocksize=(blocksize, blocksize))
    if A.blocksize != (blocksize, blocksize):
        A = A.tobsr(blocksize=(blocksize, blocksize))
    if A.dtype != np.float64:
        A = A.astype(np.float64)
    nb = A.shape[0] // blocksize
    # the diagonal block of every block row (the last stored one if a row holds duplicates, as the
    # reference's loop leaves it), gathered in one pass
    block_diag = np.zeros((nb, blocksize, blocksize), dtype=A.dtype)
    brow = np.repeat(np.arange(nb, dtype=np.int64), np.diff(A.indptr))
    at = np.nonzero(A.indices == brow)[0]
    block_diag[brow[at]] = A.data[at]
    if inv_flag:
        # pseudo-inverse of every block (the reference: amg_core.pinv_array, an SVD per block), batched;
        # LAPACK runs without the GIL, so large inputs are cut into chunks for a few host threads
        step = 1 << 16
        chunks = [(lo, min(nb, lo + step)) for lo in range(0, nb, step)]

        def invert(c):
            block_diag[c[0]:c[1]] = np.linalg.pinv(block_diag[c[0]:c[1]])
        if len(chunks) > 4:
            import os
            from concurrent.futures import ThreadPoolExecutor
            try:
                workers = min(16, len(os.sched_getaffinity(0)))
            except AttributeError:
                workers = min(16, os.cpu_count() or 1)
            with ThreadPoolExecutor(max_workers=max(1, workers)) as pool:
                list(pool.map(invert, chunks))
        else:
            for c in chunks:
                invert(c)
    return block_diag


def scale_rows(A, v, copy=True):
    """util/utils.py:133-200 (CSR/BSR(1,1))"""
    v = np.ravel(v)
    if isspmatrix_bsr(A):
        R, C = A.blocksize
        A = bsr_matrix(A, copy=copy)
        per_block = np.repeat(v.reshape(-1, R), np.diff(A.indptr), axis=0)      # (nblocks, R)
        A.data = A.data * per_block[:, :, None]
        return A
    A = csr_matrix(A, copy=copy)
    A.data = A.data * np.repeat(v, np.diff(A.indptr))
    return A


def _approximate_eigenvalues(A, tol, maxiter, symmetric=None, initial_guess=None):
    """util/linalg.py:173-279 (non-symmetric Arnoldi branch, the only one the
    spectral-radius estimate uses, :353-355)."""
    from scipy.sparse.linalg import aslinearoperator
    A = aslinearoperator(A)
    eps = np.finfo(float).eps
    breakdown = eps * 1e6
    breakdown_flag = False
    if A.shape[0] != A.shape[1]:
        raise ValueError("expected square matrix")
    maxiter = min(A.shape[0], maxiter)
    if initial_guess is None:
        v0 = np.random.rand(A.shape[1], 1)
    else:
        v0 = initial_guess
    v0 = v0 / norm(v0)
    H = np.zeros((maxiter + 1, maxiter), dtype=np.result_type(v0.dtype, A.dtype))
    V = [v0]
    j = 0
    for j in range(maxiter):
        w = A * V[-1]
        for i, v in enumerate(V):
            H[i, j] = np.dot(np.conjugate(v.ravel()), w.ravel())
            w = w - H[i, j] * v
        H[j + 1, j] = norm(w)
        if H[j + 1, j] < breakdown:
            breakdown_flag = True
            if H[j + 1, j] != 0.0:
                w = w / H[j + 1, j]
            V.append(w)
            break
        w = w / H[j + 1, j]
        V.append(w)
    Eigs, Vects = scipy.linalg.eig(H[:j + 1, :j + 1], left=False, right=True)
    return (Vects, Eigs, H, V, breakdown_flag)


def _check_estimate_arguments(A, maxiter, restart):
    """the argument checks of util/linalg.py:346-352, shared by the host and the device estimate"""
    if maxiter < 1:
        raise ValueError("expected maxiter > 0")
    if restart < 0:
        raise ValueError("expected restart >= 0")
    if A.shape[0] != A.shape[1]:
        raise ValueError("expected square A")


def approximate_spectral_radius(A, tol=0.01, maxiter=15, restart=5, symmetric=None,
                                initial_guess=None, return_vector=False):
    """util/linalg.py:282-416.  Consumes the global numpy RNG exactly like the
    reference (one rand(n,1) per call), so seeded runs give the same rho."""
    if not hasattr(A, "rho") or return_vector:
        _check_estimate_arguments(A, maxiter, restart)
    if (not hasattr(A, "rho")) and (not return_vector) and initial_guess is None and isspmatrix(A) \
            and use_device_for(A):
        A.rho = approximate_spectral_radius_device(A, None, tol, maxiter, restart)
        return A.rho
    if not hasattr(A, "rho") or return_vector:
        if initial_guess is None:
            v0 = np.random.rand(A.shape[1], 1)
        else:
            v0 = np.array(initial_guess.reshape(-1, 1), dtype=A.dtype)
        for j in range(restart + 1):
            evect, ev, H, V, breakdown_flag = _approximate_eigenvalues(A, tol, maxiter, False,
                                                                        initial_guess=v0)
            nvecs = ev.shape[0]
            max_index = np.abs(ev).argmax()
            error = H[nvecs, nvecs - 1] * evect[-1, max_index]
            v0 = np.dot(np.hstack(V[:-1]), evect[:, max_index].reshape(-1, 1))
            if (np.abs(error) / np.abs(ev[max_index]) < tol) or breakdown_flag:
                break
        rho = np.abs(ev[max_index])
        if isspmatrix(A):
            A.rho = rho
        if return_vector:
            return (rho, v0)
        return rho
    return A.rho


# --------------------------------------------------------------------------- device-assisted setup
class _DeviceOperator(object):
    """A single stored operator in HBM (a one-level amg_hier) for setup-time Krylov work."""

    def __init__(self, A, device=0):
        from . import _lib
        from scipy.sparse import isspmatrix_bsr as _isbsr
        self._lib = _lib
        L = _lib.lib()
        self.L = L
        self.n = A.shape[0]
        self.h = L.amg_hier_create(1, int(device))
        if not self.h:
            raise _lib.AmgDeviceError(L.amg_last_error().decode())
        if _isbsr(A):
            fmt, (R, C) = 1, A.blocksize
            data = np.ascontiguousarray(np.ravel(A.data), dtype=np.float64)
        else:
            A = csr_matrix(A)
            fmt, R, C = 0, 1, 1
            data = np.ascontiguousarray(A.data, dtype=np.float64)
        Ap = np.ascontiguousarray(A.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(A.indices, dtype=np.intc)
        try:
            _lib.check(L.amg_hier_set_matrix(self.h, 0, 0, fmt, A.shape[0], A.shape[1], R, C, Ap.ctypes.data,
                                             Aj.ctypes.data, data.ctypes.data, 0))
        except Exception:
            self.close()
            raise

    def arnoldi(self, dinv, v0, maxiter, breakdown_tol):
        import ctypes as C
        _lib = self._lib
        maxiter = min(self.n, maxiter)
        H = np.zeros((maxiter + 1, maxiter), dtype=np.float64)
        steps, brk = C.c_int(0), C.c_int(0)
        v0 = np.ascontiguousarray(np.ravel(v0), dtype=np.float64)
        dptr = _lib.dp(np.ascontiguousarray(dinv, dtype=np.float64)) if dinv is not None else None
        _lib.check(self.L.amg_arnoldi(self.h, 0, dptr, _lib.dp(v0), int(maxiter), float(breakdown_tol),
                                      _lib.dp(H), C.byref(steps), C.byref(brk)))
        return H, steps.value, bool(brk.value)

    def combine(self, coef):
        _lib = self._lib
        coef = np.ascontiguousarray(coef, dtype=np.float64)
        v = np.empty(self.n, dtype=np.float64)
        _lib.check(self.L.amg_arnoldi_combine(self.h, _lib.dp(coef), len(coef), _lib.dp(v)))
        return v

    def free_workspace(self):
        if getattr(self, "h", None):
            self.L.amg_arnoldi_free(self.h)

    def device_bytes(self):
        return int(self.L.amg_hier_device_bytes(self.h)) if getattr(self, "h", None) else 0

    def close(self):
        if getattr(self, "h", None):
            self.L.amg_hier_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_operator(A):
    """Cached HBM copy of A for setup-time estimates (released by release_device_operator)."""
    op = getattr(A, "_amg_devop", None)
    if op is None or op.h is None:
        op = _DeviceOperator(A)
        A._amg_devop = op
    return op


def release_device_operator(A):
    op = getattr(A, "_amg_devop", None)
    if op is not None:
        op.close()
        try:
            del A._amg_devop
        except AttributeError:
            pass


def galerkin_device(A, R, P, n_coarse):
    """(R*A)*P on the GPU with scipy's csr_matmat arithmetic and output order (csrc/spgemm.hip), A being the HBM copy
    the spectral-radius estimate left behind (device_operator); R, P = (indptr int64, indices int32, data f64).
    Returns (Cp, Cj, Cx) or None when the device path does not apply (no HBM copy, operator not held as CSR, a row too
    long for the device tables) -- the caller then runs its host products."""
    import ctypes as C
    op = getattr(A, "_amg_devop", None)
    if op is None or not getattr(op, "h", None):
        return None
    from . import _lib
    L = _lib.lib()
    (Rp, Rj, Rx), (Pp, Pj, Px) = R, P
    Cp = np.empty(n_coarse + 1, dtype=np.int64)
    g = C.c_void_p()
    rc = L.amg_hier_galerkin(op.h, 0, int(n_coarse), Rp.ctypes.data, Rj.ctypes.data, Rx.ctypes.data,
                             Pp.ctypes.data, Pj.ctypes.data, Px.ctypes.data, Cp.ctypes.data, C.byref(g))
    if rc != 0:
        return None
    nnz = int(Cp[n_coarse])
    Cj = np.empty(nnz, dtype=np.intc)
    Cx = np.empty(nnz, dtype=np.float64)
    _lib.check(L.amg_galerkin_fetch(g, Cj.ctypes.data, Cx.ctypes.data))
    return Cp, Cj, Cx


def use_device_for(A):
    if A.shape[0] < DEVICE_RHO_MIN_ROWS:
        return False
    from . import _lib
    return _lib.device_count() > 0


def approximate_spectral_radius_device(A, dinv=None, tol=0.01, maxiter=15, restart=5):
    """approximate_spectral_radius (util/linalg.py:282-416) of diag(dinv)*A (dinv None: A) with the
    Arnoldi iterations on the GPU.  Same restart logic and the same single np.random.rand(n, 1)
    draw as the reference; dots/norms are device reductions, so rho agrees to rounding, not bitwise."""
    _check_estimate_arguments(A, maxiter, restart)
    import os, time
    verbose = os.environ.get("AMG_SETUP_VERBOSE", "0") != "0"
    t0 = time.perf_counter()
    n = A.shape[0]
    if getattr(A, "_amg_devop", None) is None and n >= 1000000:
        # the upload of A (and its structure analysis) and the random start vector are independent: side by side
        import threading
        box = {}

        def upload():
            try:
                box["op"] = device_operator(A)
            except BaseException as e:      # noqa: BLE001 -- re-raised below
                box["error"] = e
        th = threading.Thread(target=upload)
        th.start()
        v0 = np.random.rand(n, 1).ravel()
        th.join()
        if "error" in box:
            raise box["error"]
        op = box["op"]
        t1 = t2 = time.perf_counter()
    else:
        op = device_operator(A)
        t1 = time.perf_counter()
        v0 = np.random.rand(n, 1).ravel()
        t2 = time.perf_counter()
    breakdown_tol = np.finfo(float).eps * 1e6
    ev = None
    max_index = 0
    for j in range(restart + 1):
        H, m, breakdown = op.arnoldi(dinv, v0, maxiter, breakdown_tol)
        ev, evect = scipy.linalg.eig(H[:m, :m], left=False, right=True)
        max_index = np.abs(ev).argmax()
        error = H[m, m - 1] * evect[-1, max_index]
        if (np.abs(error) / np.abs(ev[max_index]) < tol) or breakdown:
            break
        coef = evect[:, max_index]
        if np.iscomplexobj(coef):
            if np.abs(coef.imag).max() > 1e-14 * np.abs(coef).max():
                raise NotImplementedError("complex Ritz vector in the device spectral-radius estimate")
            coef = coef.real
        v0 = op.combine(coef)
    op.free_workspace()          # the Krylov basis ((maxiter + 1) n doubles) is only needed during the estimate
    if verbose:
        print("[setup]   spectral radius (%d rows): operator in HBM %.2fs, random start %.2fs, %d Arnoldi pass(es) %.2fs"
              % (n, t1 - t0, t2 - t1, j + 1, time.perf_counter() - t2), flush=True)
    return float(np.abs(ev[max_index]))
