"""Generic AMG solver -- the ``multilevel_solver`` object API of the reference
(/root/reference/pyamg/multilevel.py) with the solve phase running on MI355X.

The object keeps the reference's attributes (``levels[i].A/P/R``,
``presmoother``/``postsmoother`` closures, ``coarse_solver``) as host-side
scipy matrices -- that is how setup code builds and inspects a hierarchy --
and mirrors them ONCE into HBM (operators, smoother constants, work vectors)
the first time ``solve`` runs.  ``solve``/``aspreconditioner`` then execute
multilevel.py:316-548 entirely on the device through libamgcore_hip.so.
"""
from warnings import warn

import os

import numpy as np
import scipy.linalg
import scipy.sparse as sparse

from . import _lib

__all__ = ["multilevel_solver", "coarse_grid_solver"]

_SM_KIND = {None: 0, "None": 0, "jacobi": 1, "gauss_seidel": 2, "sor": 3, "polynomial": 4,
            "block_jacobi": 5, "block_gauss_seidel": 6, "gauss_seidel_indexed": 7, "schwarz": 8,
            "gauss_seidel_ne": 9, "gauss_seidel_nr": 10, "jacobi_ne": 11}
# (name "krylov": a callback smoother, registered by _DeviceHierarchy._set_krylov_smoother)
_SWEEP = {"forward": 0, "backward": 1, "symmetric": 2}
_CYCLE = {"V": 0, "W": 1, "F": 2, "AMLI": 3}
_X0_ZERO, _NO_EARLY_STOP, _DEVICE_VECTORS = 1, 2, 4


def _desc_struct(desc, keep):
    """descriptor dict (smoothing.py closures' .desc) -> amg_smoother_desc"""
    d = _lib.SmootherDesc()
    name = None if desc is None else desc.get("name")
    if name not in _SM_KIND:
        raise NotImplementedError("smoother %r has no device implementation" % (name,))
    d.kind = _SM_KIND[name]
    if d.kind == 0:
        return d
    d.iterations = int(desc.get("iterations", 1))
    sweep = desc.get("sweep", "forward")
    if sweep not in _SWEEP:
        raise ValueError("valid sweep directions are 'forward', 'backward', and 'symmetric'")
    d.sweep = _SWEEP[sweep]
    d.omega = float(desc.get("omega", 1.0))
    if desc.get("coefficients") is not None:
        co = np.ascontiguousarray(desc["coefficients"], dtype=np.float64)
        keep.append(co)
        d.ncoef, d.coef = len(co), _lib.dp(co)
    d.blocksize = int(desc.get("blocksize", 1) or 1)
    if desc.get("Dinv") is not None:
        Dinv = np.ascontiguousarray(np.ravel(desc["Dinv"]), dtype=np.float64)
        keep.append(Dinv)
        d.Dinv = _lib.dp(Dinv)
    if desc.get("indices") is not None:
        idx = np.ascontiguousarray(desc["indices"], dtype=np.intc)
        keep.append(idx)
        d.indices, d.nindices = _lib.ip(idx), len(idx)
    if name == "schwarz":
        arrs = [np.ascontiguousarray(desc[k], dtype=np.intc) for k in ("subdomain", "subdomain_ptr", "inv_subblock_ptr")]
        Tx = np.ascontiguousarray(desc["inv_subblock"], dtype=np.float64)
        keep.extend(arrs + [Tx])
        d.Sj, d.Sp, d.Tp, d.Tx = _lib.ip(arrs[0]), _lib.ip(arrs[1]), _lib.ip(arrs[2]), _lib.dp(Tx)
        d.nsdomains = len(arrs[1]) - 1
    return d


class _DeviceHierarchy(object):
    """Owns an amg_hier handle (include/amgcore_hip.h section 2)."""

    def __init__(self, ml, device=0):
        L = _lib.lib()
        self.L = L
        levels = ml.levels
        self.n = levels[0].A.shape[0]
        self._shapes = {}
        self._callbacks = []             # ctypes callbacks must outlive the hierarchy
        self._callback_error = None
        self.h = L.amg_hier_create(len(levels), int(device))
        if not self.h:
            msg = L.amg_last_error().decode()
            raise _lib.AmgDeviceError(msg)
        try:
            self._build(ml)
        except Exception:
            L.amg_hier_destroy(self.h)
            self.h = None
            raise

    def _set_matrix(self, lvl, which, M):
        if M.dtype != np.float64:
            raise NotImplementedError("device hierarchy supports float64 operators only")
        if sparse.isspmatrix_bsr(M):
            fmt, (R, C) = 1, M.blocksize
            data = np.ascontiguousarray(np.ravel(M.data), dtype=np.float64)
        elif sparse.isspmatrix_csr(M):
            fmt, R, C = 0, 1, 1
            data = np.ascontiguousarray(M.data, dtype=np.float64)
        else:
            M = sparse.csr_matrix(M)
            fmt, R, C = 0, 1, 1
            data = np.ascontiguousarray(M.data, dtype=np.float64)
        Ap = np.ascontiguousarray(M.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(M.indices, dtype=np.intc)
        self._shapes[(lvl, which)] = M.shape
        _lib.check(self.L.amg_hier_set_matrix(self.h, lvl, which, fmt, M.shape[0], M.shape[1], R, C,
                                              Ap.ctypes.data, Aj.ctypes.data, data.ctypes.data, 0))

    def _set_smoother(self, lvl, which, fn, A):
        desc = getattr(fn, "desc", None)
        if fn is not None and desc is None:
            raise NotImplementedError(
                "level %d: smoother %r carries no device descriptor; use pyamg_amd.smoothing."
                "change_smoothers with one of the device smoothers" % (lvl, fn))
        if desc is not None and desc.get("name") == "krylov":
            return self._set_krylov_smoother(lvl, which, desc, A)
        keep = []
        d = _desc_struct(desc, keep)
        if which == 2:
            _lib.check(self.L.amg_hier_set_coarse_smoother(self.h, d))
        else:
            _lib.check(self.L.amg_hier_set_smoother(self.h, lvl, which, d))
        if d.kind in (5, 6):
            # the shim re-blocks A (relaxation.py:471,563): A.tobsr(blocksize=(bs, bs))
            bs = d.blocksize
            if not (sparse.isspmatrix_bsr(A) and A.blocksize == (bs, bs)):
                Ab = A.tobsr(blocksize=(bs, bs))
                Ap = np.ascontiguousarray(Ab.indptr, dtype=np.intc)
                Aj = np.ascontiguousarray(Ab.indices, dtype=np.intc)
                Ax = np.ascontiguousarray(np.ravel(Ab.data), dtype=np.float64)
                _lib.check(self.L.amg_hier_set_block_matrix(self.h, lvl, which, Ab.shape[0] // bs, bs,
                                                            _lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax)))
        if d.kind in (10, 11):
            # gauss_seidel_nr sweeps / jacobi_ne gathers through A by columns (lvl.Acsc, smoothing.py:472-478)
            self._set_aux(lvl, which, 0, A.tocsc())
        if d.kind == 10 and not A.tocsr().has_sorted_indices:
            # its starting residual is a CSC product: terms in ascending column order (relaxation.py:992)
            As = A.tocsr().copy()
            As.sort_indices()
            self._set_aux(lvl, which, 1, As)

    def _set_krylov_smoother(self, lvl, which, desc, A):
        """smoothing.py:481-509: a few iterations of an unpreconditioned Krylov method as the level's smoother, run by
        pyamg_amd.krylov on the level's resident vectors from a callback inside the cycle"""
        from . import krylov
        if which == 2:
            raise NotImplementedError("Krylov methods as coarse solver go through coarse_grid_solver('cg' ...)")
        method = desc["method"]
        tol, maxiter, restrt = float(desc["tol"]), desc["maxiter"], desc.get("restrt")
        aux = (which, 0) if method in ("cgne", "cgnr") else None
        owner = self

        def relax(user, level, x_dev, b_dev):
            try:
                with krylov.DeviceSpace(owner, level, cycle=None, aux=aux) as V:
                    krylov.run(method, V, b_dev, x_dev, tol, maxiter, restrt=restrt)
                return 0
            except Exception as e:          # noqa: BLE001 -- must not propagate through the C frames
                owner._callback_error = e
                return 1
        cb = _lib.RELAX_CALLBACK(relax)
        self._callbacks.append(cb)
        _lib.check(self.L.amg_hier_set_callback_smoother(self.h, lvl, which, cb, None))
        if aux is not None:
            self._set_aux(lvl, which, 0, sparse.csc_matrix(A))        # A by columns = A^T by rows

    def _set_coarse_callback(self, fn, Ac):
        """multilevel.py:642-692: Krylov names and callables as coarse solver -- host code on the coarsest level's few
        hundred unknowns, called from inside the cycle"""
        owner = self

        def solve(user, n, b_ptr, x_ptr):
            try:
                b = np.ctypeslib.as_array(b_ptr, shape=(n,)).copy()
                x = np.asarray(fn(Ac, b), dtype=np.float64).ravel()
                np.ctypeslib.as_array(x_ptr, shape=(n,))[:] = x
                return 0
            except Exception as e:          # noqa: BLE001
                owner._callback_error = e
                return 1
        cb = _lib.COARSE_CALLBACK(solve)
        self._callbacks.append(cb)
        _lib.check(self.L.amg_hier_set_coarse_callback(self.h, cb, None))

    def _set_aux(self, lvl, which, slot, M):
        M.sort_indices()
        Ap = np.ascontiguousarray(M.indptr, dtype=np.intc)
        Aj = np.ascontiguousarray(M.indices, dtype=np.intc)
        Ax = np.ascontiguousarray(M.data, dtype=np.float64)
        nmajor = len(Ap) - 1
        nminor = M.shape[0] if sparse.isspmatrix_csc(M) else M.shape[1]
        _lib.check(self.L.amg_hier_set_aux_matrix(self.h, lvl, which, slot, nmajor, nminor, _lib.ip(Ap), _lib.ip(Aj),
                                                  _lib.dp(Ax)))

    def _build(self, ml):
        levels = ml.levels
        for i, lvl in enumerate(levels):
            self._set_matrix(i, 0, lvl.A)
            if i < len(levels) - 1:
                self._set_matrix(i, 1, lvl.P)
                self._set_matrix(i, 2, lvl.R)
                self._set_smoother(i, 0, getattr(lvl, "presmoother", None), lvl.A)
                self._set_smoother(i, 1, getattr(lvl, "postsmoother", None), lvl.A)
        cs = ml.coarse_solver
        Ac = levels[-1].A
        kind, payload = cs.device_form(Ac)
        if kind == "dense":
            M = np.ascontiguousarray(payload, dtype=np.float64)
            _lib.check(self.L.amg_hier_set_coarse_dense(self.h, _lib.dp(M), M.shape[0]))
        elif kind == "smoother":
            self._set_smoother(len(levels) - 1, 2, payload, Ac)
        elif kind == "callback":
            self._set_coarse_callback(payload, Ac)
        elif kind != "none":
            raise NotImplementedError("coarse solver %s has no device implementation" % cs.name())
        _lib.check(self.L.amg_hier_finalize(self.h))
        # Large hierarchies: the CSR arrays of operators that are only ever applied from their stencil / sliced form
        # are not kept beside it (500^3 Chebyshev hierarchy: 56.6 -> 34 GB in HBM, same bits).  The host copies stay in
        # ml.levels, so change_smoothers / a rebuilt mirror have everything they need.  AMG_RELEASE_SOURCES=0 keeps
        # them (A/B runs of the other kernels on a live hierarchy), =1 releases at any size.
        rel = os.environ.get("AMG_RELEASE_SOURCES")
        big = levels[0].A.shape[0] >= 4000000
        self.released_bytes = 0
        if rel == "1" or (rel is None and big):
            self.released_bytes = int(self.L.amg_hier_release_sources(self.h))

    def _check(self, rc):
        err, self._callback_error = self._callback_error, None
        if err is not None:
            raise err
        _lib.check(rc)

    def solve(self, b, x, tol, maxiter, cycle, x0_zero=False, fixed=False):
        res = np.zeros(maxiter + 2, dtype=np.float64)
        nres = _lib.C.c_int(0)
        flags = (_X0_ZERO if x0_zero else 0) | (_NO_EARLY_STOP if fixed else 0)
        self._check(self.L.amg_hier_solve(self.h, b.ctypes.data, x.ctypes.data, float(tol), int(maxiter),
                                          _CYCLE[cycle], _lib.dp(res), _lib.C.byref(nres), flags))
        return res[:nres.value]

    def pcg(self, b, x, tol, maxiter, cycle, x0_zero=False):
        res = np.zeros(maxiter + 2, dtype=np.float64)
        nres, info = _lib.C.c_int(0), _lib.C.c_int(0)
        _lib.check(self.L.amg_hier_pcg(self.h, b.ctypes.data, x.ctypes.data, float(tol), int(maxiter),
                                       _CYCLE[cycle], _lib.dp(res), _lib.C.byref(nres), _lib.C.byref(info),
                                       _X0_ZERO if x0_zero else 0))
        return res[:nres.value], info.value

    def cycle(self, b, x, cycle, x0_zero=False):
        self._check(self.L.amg_hier_cycle(self.h, b.ctypes.data, x.ctypes.data, _CYCLE[cycle],
                                          _X0_ZERO if x0_zero else 0))

    def cycle_device(self, b_dev, x_dev, cycle):
        """x_dev = one cycle from a zero guess for the right-hand side b_dev (DEVICE pointers): the preconditioner
        M of multilevel.py:306-314 for the device-resident Krylov methods"""
        self._check(self.L.amg_hier_cycle(self.h, b_dev, x_dev, _CYCLE[cycle], _X0_ZERO | _DEVICE_VECTORS))

    def level_size(self, lvl):
        return self._shapes[(lvl, 0)][0]

    def relax(self, lvl, which, b, x):
        _lib.check(self.L.amg_hier_relax(self.h, lvl, which, _lib.dp(b), _lib.dp(x)))

    def matvec(self, lvl, which, x):
        rows, cols = self._shapes[(lvl, which)]
        x = np.ascontiguousarray(np.ravel(x), dtype=np.float64)
        if x.size != cols:
            raise ValueError("dimension mismatch")
        y = np.zeros(rows, dtype=np.float64)
        _lib.check(self.L.amg_hier_matvec(self.h, lvl, which, _lib.dp(x), _lib.dp(y)))
        return y

    def cycle_bytes(self, cycle="V"):
        return self.L.amg_hier_cycle_bytes(self.h, _CYCLE[cycle])

    def cycle_bytes_moved(self, cycle="V"):
        return self.L.amg_hier_cycle_bytes_moved(self.h, _CYCLE[cycle])

    def last_solve_ms(self):
        return self.L.amg_hier_last_solve_ms(self.h)

    def device_bytes(self):
        return self.L.amg_hier_device_bytes(self.h)

    def time_spmv(self, lvl, which, mode=0, reps=10):
        ms = _lib.C.c_double(0.0)
        _lib.check(self.L.amg_hier_time_spmv(self.h, lvl, which, mode, reps, _lib.C.byref(ms)))
        return ms.value

    def time_relax(self, lvl, which, reps=5):
        ms = _lib.C.c_double(0.0)
        _lib.check(self.L.amg_hier_time_relax(self.h, lvl, which, reps, _lib.C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "h", None):
            self.L.amg_hier_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class multilevel_solver:
    """Stores multigrid hierarchy and implements the multigrid cycle
    (multilevel.py:14-548).  Same attributes and methods as the reference."""

    class level:
        """One level: A, and (except on the coarsest) P, R, presmoother,
        postsmoother (multilevel.py:45-68)."""

        def __init__(self):
            pass

    def __init__(self, levels, coarse_solver="pinv2", device=0):
        self.levels = levels
        self.coarse_solver = coarse_grid_solver(coarse_solver)
        self.device = device
        self._dev = None
        for level in levels[:-1]:
            if not hasattr(level, "R"):
                level.R = level.P.conj().T.asformat(level.P.format)   # level.P.H (multilevel.py:154-156)

    # ------------------------------------------------------------------ reporting
    def _level_nnz(self):
        return [int(level.A.nnz) for level in self.levels]

    def __repr__(self):
        """Same text as the reference prints (multilevel.py:158-176): header lines, then one row
        per level with its share of the stored entries."""
        nnz = self._level_nnz()
        total = float(sum(nnz))
        lines = ["multilevel_solver",
                 "Number of Levels:     %d" % len(self.levels),
                 "Operator Complexity: %6.3f" % self.operator_complexity(),
                 "Grid Complexity:     %6.3f" % self.grid_complexity(),
                 "Coarse Solver:        %s" % self.coarse_solver.name(),
                 "  level   unknowns     nonzeros"]
        for i, level in enumerate(self.levels):
            lines.append("   %2d   %10d   %10d [%5.2f%%]" % (i, level.A.shape[1], nnz[i], 100 * nnz[i] / total))
        return "\n".join(lines) + "\n"

    def _level_visits(self, cycle):
        """How many times one cycle of the given type arrives at each level.  V goes down once;
        W arrives twice at every level below the first, except that the coarsest level is
        solved once per arrival at the level above it; F arrives once as F and once as V."""
        nlev = len(self.levels)
        visits = [0] * nlev

        def arrive(lvl, kind, times):
            visits[lvl] += times
            if lvl == nlev - 2:
                visits[nlev - 1] += times
            elif lvl < nlev - 2:
                if kind == "V":
                    arrive(lvl + 1, "V", times)
                elif kind == "W":
                    arrive(lvl + 1, "W", 2 * times)
                else:
                    arrive(lvl + 1, "F", times)
                    arrive(lvl + 1, "V", times)
        arrive(0, cycle, 1)
        return visits

    def cycle_complexity(self, cycle="V"):
        """Entries of the level operators touched by one cycle, relative to the finest operator:
        every arrival at a level costs two passes over A (pre- and post-smoothing), the coarsest
        level one (its solve) -- the values of multilevel.py:178-248."""
        cycle = str(cycle).upper()
        kind = {"V": "V", "W": "W", "AMLI": "W", "F": "F"}.get(cycle)
        if kind is None:
            raise TypeError("Unrecognized cycle type (%s)" % cycle)
        nnz = self._level_nnz()
        if len(nnz) == 1:
            return 1.0
        visits = self._level_visits(kind)
        work = sum(2 * v * z for v, z in zip(visits[:-1], nnz[:-1])) + visits[-1] * nnz[-1]
        return float(work) / float(nnz[0])

    def operator_complexity(self):
        """stored entries of all level operators over those of the finest (multilevel.py:250-259)"""
        nnz = self._level_nnz()
        return sum(nnz) / float(nnz[0])

    def grid_complexity(self):
        """unknowns of all levels over those of the finest (multilevel.py:261-269)"""
        sizes = [level.A.shape[0] for level in self.levels]
        return sum(sizes) / float(sizes[0])

    # ------------------------------------------------------------------ on-disk format
    def save(self, path):
        """Write the hierarchy (operators in stored order, smoother constants, the dense coarse operator) to the
        directory `path`; `multilevel_solver.load(path)` gives a solver with bit-identical iterates."""
        from .hierarchy_io import save_hierarchy
        return save_hierarchy(self, path)

    @staticmethod
    def load(path, mmap=False, device=0):
        from .hierarchy_io import load_hierarchy
        return load_hierarchy(path, mmap=mmap, device=device)

    # ------------------------------------------------------------------ device mirror
    def _invalidate_device(self):
        if self._dev is not None:
            self._dev.close()
        self._dev = None

    def device_hierarchy(self):
        """The HBM-resident mirror of this hierarchy (built on first use)."""
        if self._dev is None:
            self._dev = _DeviceHierarchy(self, self.device)
        return self._dev

    # ------------------------------------------------------------------ solve
    def psolve(self, b):
        return self.solve(b, maxiter=1)

    def aspreconditioner(self, cycle="V"):
        """multilevel.py:274-314"""
        from scipy.sparse.linalg import LinearOperator
        shape = self.levels[0].A.shape
        dtype = self.levels[0].A.dtype

        def matvec(b):
            return self.solve(b, maxiter=1, cycle=cycle, tol=1e-12)
        return LinearOperator(shape, matvec, dtype=dtype)

    def solve(self, b, x0=None, tol=1e-5, maxiter=100, cycle="V", accel=None, callback=None,
              residuals=None, return_residuals=False):
        """Main solution call to execute multigrid cycling (multilevel.py:316-471).

        Extension: b (and x0) may be torch CUDA tensors (float64, contiguous, on the hierarchy's device): the solve then
        never touches the host -- no PCIe copy of b / x0 in or x out (3 GB at 500^3) -- and returns a tensor.  Plain
        cycling and accel='cg' (the device PCG); same iterates and history as with host vectors."""
        if _is_device_tensor(b):
            return self._solve_device_tensors(b, x0, tol, maxiter, cycle, accel, callback, residuals)
        b = np.asarray(b)
        if x0 is None:
            x = np.zeros_like(b)
        else:
            x = np.array(x0)   # copy
        cycle = str(cycle).upper()
        if cycle not in _CYCLE:
            raise TypeError("Unrecognized cycle type (%s)" % cycle)
        if (cycle == "AMLI") and hasattr(self.levels[0].A, "symmetry"):
            if self.levels[0].A.symmetry != "hermitian":
                raise ValueError("AMLI cycles require symmetry to be hermitian")

        if accel is not None:
            return self._solve_accel(b, x0, tol, maxiter, cycle, accel, callback, residuals)

        if return_residuals:
            warn("return_residuals is deprecated.  Use residuals instead")
            residuals = []
        if residuals is None:
            residuals = []
        else:
            residuals[:] = []

        tp = np.result_type(b.dtype, x.dtype, self.levels[0].A.dtype)
        if tp != np.float64:
            if np.issubdtype(tp, np.floating) or np.issubdtype(tp, np.integer):
                tp = np.float64
            else:
                raise NotImplementedError("device solve supports real float64 systems only")
        shape = b.shape
        n = self.levels[0].A.shape[0]
        if b.size != n or x.size != n:
            raise ValueError("b and x0 must have %d entries" % n)
        b1 = np.ascontiguousarray(np.ravel(b), dtype=np.float64)
        x1 = np.ascontiguousarray(np.ravel(x), dtype=np.float64)
        x0_zero = not np.any(x1)
        dev = self.device_hierarchy()

        if callback is None:
            res = dev.solve(b1, x1, tol, maxiter, cycle, x0_zero=x0_zero)
            residuals.extend(float(r) for r in res)
        else:
            # one cycle per call so that callback(x) sees every iterate (multilevel.py:454-466)
            normb = float(np.sqrt(np.inner(b1, b1)))
            atol = tol * normb if normb != 0 else tol
            res = dev.solve(b1, x1, 0.0, 0, cycle, x0_zero=x0_zero)
            residuals.append(float(res[0]))
            while len(residuals) <= maxiter and residuals[-1] > atol:
                res = dev.solve(b1, x1, 0.0, 1, cycle, x0_zero=x0_zero, fixed=True)
                x0_zero = False
                residuals.append(float(res[-1]))
                callback(x1.reshape(shape))
        xout = x1.reshape(shape)
        if return_residuals:
            return xout, residuals
        return xout

    def _solve_device_tensors(self, b, x0, tol, maxiter, cycle, accel, callback, residuals):
        import torch
        cycle = str(cycle).upper()
        if cycle not in _CYCLE:
            raise TypeError("Unrecognized cycle type (%s)" % cycle)
        if callback is not None or accel not in (None, "cg"):
            raise NotImplementedError("device tensors: plain cycling and accel='cg' (no callback); pass host arrays otherwise")
        if accel == "cg" and cycle == "AMLI":
            raise ValueError("AMLI cycles require acceleration (accel) to be fgmres, or no acceleration")
        n = self.levels[0].A.shape[0]
        dev = self.device_hierarchy()
        for t in (b,) if x0 is None else (b, x0):
            if not _is_device_tensor(t) or t.dtype != torch.float64 or t.numel() != n or not t.is_contiguous() or t.device.index != self.device:
                raise ValueError("device tensors must be contiguous float64 CUDA tensors of %d entries on device %d" % (n, self.device))
        x = torch.zeros_like(b) if x0 is None else x0.clone()
        x0_zero = x0 is None
        torch.cuda.current_stream(b.device).synchronize()          # the library works on its own stream
        if maxiter is None:
            maxiter = int(1.3 * n) + 2 if accel == "cg" else 100
        res = np.zeros(maxiter + 2, dtype=np.float64)
        nres = _lib.C.c_int(0)
        flags = _DEVICE_VECTORS | (_X0_ZERO if x0_zero else 0)
        if accel == "cg":
            info = _lib.C.c_int(0)
            _lib.check(dev.L.amg_hier_pcg(dev.h, b.data_ptr(), x.data_ptr(), float(tol), int(maxiter), _CYCLE[cycle], _lib.dp(res),
                                          _lib.C.byref(nres), _lib.C.byref(info), flags))
            if info.value < 0:
                warn("Indefinite matrix or preconditioner detected in CG, aborting")
        else:
            dev._check(dev.L.amg_hier_solve(dev.h, b.data_ptr(), x.data_ptr(), float(tol), int(maxiter), _CYCLE[cycle], _lib.dp(res),
                                            _lib.C.byref(nres), flags))
        if residuals is not None:
            residuals[:] = [float(r) for r in res[:nres.value]]
        return x

    def _solve_accel(self, b, x0, tol, maxiter, cycle, accel, callback, residuals):
        """Krylov acceleration (multilevel.py:381-422): the cycle is the preconditioner M;
        the outer Krylov iteration is scipy's (the reference falls back to the same
        scipy.sparse.linalg interface, :404-422)."""
        if (accel != "fgmres") and (cycle == "AMLI"):
            raise ValueError("AMLI cycles require acceleration (accel) to be fgmres, or no acceleration")
        if accel == "cg" and callback is None:
            # device-resident PCG (pyamg/krylov/_cg.py semantics, preconditioner-norm history)
            n = self.levels[0].A.shape[0]
            b1 = np.ascontiguousarray(np.ravel(b), dtype=np.float64)
            x1 = np.zeros(n) if x0 is None else np.ascontiguousarray(np.ravel(np.array(x0)), dtype=np.float64)
            if maxiter is None:
                maxiter = int(1.3 * n) + 2
            res, info = self.device_hierarchy().pcg(b1, x1, tol, maxiter, cycle, x0_zero=not np.any(x1))
            if info < 0:
                warn("Indefinite matrix or preconditioner detected in CG, aborting")
            if residuals is not None:
                residuals[:] = [float(r) for r in res]
            return x1.reshape(np.asarray(b).shape)
        from . import krylov
        name = accel if isinstance(accel, str) else getattr(accel, "__name__", None)
        if name in krylov.METHODS and (isinstance(accel, str) or accel is krylov.METHODS[name]):
            # pyamg.krylov's own methods (multilevel.py:390-394), device-resident: vectors stay in HBM, the cycle is M
            n = self.levels[0].A.shape[0]
            b1 = np.ascontiguousarray(np.ravel(b), dtype=np.float64)
            x1 = np.zeros(n) if x0 is None else np.ascontiguousarray(np.ravel(np.array(x0)), dtype=np.float64)
            dev = self.device_hierarchy()
            aux = None
            if name in ("cgne", "cgnr"):
                dev._set_aux(0, 0, 0, sparse.csc_matrix(self.levels[0].A))
                aux = (0, 0)
            with krylov.DeviceSpace(dev, 0, cycle=cycle, aux=aux) as V:
                bd, xd = V.upload(b1), V.upload(x1)
                info = krylov.run(name, V, bd, xd, tol, maxiter, residuals=residuals, callback=callback)
                x1 = V.download(xd)
            if info < 0 and name == "cg":
                warn("Indefinite matrix or preconditioner detected in CG, aborting")
            return x1.reshape(np.asarray(b).shape)
        # anything else is scipy's (multilevel.py:395-396, 404-422): host vectors, the device cycle as M
        import scipy.sparse.linalg as spla
        if isinstance(accel, str):
            if not hasattr(spla, accel):
                raise ValueError("unknown accel method %r" % accel)
            accel = getattr(spla, accel)
        A = self.levels[0].A
        M = self.aspreconditioner(cycle=cycle)
        cb = callback
        if residuals is not None:
            residuals[:] = [float(np.linalg.norm(np.ravel(b) - A * np.ravel(np.zeros_like(b) if x0 is None else x0)))]

            def callback(x):
                if np.isscalar(x):
                    residuals.append(float(x))
                else:
                    residuals.append(float(np.linalg.norm(np.ravel(b) - A * np.ravel(x))))
                if cb is not None:
                    cb(x)
        try:
            return accel(A, b, x0=x0, rtol=tol, maxiter=maxiter, M=M, callback=callback)[0]
        except TypeError:
            return accel(A, b, x0=x0, tol=tol, maxiter=maxiter, M=M, callback=callback)[0]


def _is_device_tensor(v):
    return hasattr(v, "data_ptr") and hasattr(v, "is_cuda") and bool(v.is_cuda)


def coarse_grid_solver(solver):
    """Return a coarse grid solver suitable for multilevel_solver
    (multilevel.py:554-720).

    Dense methods ('pinv', 'pinv2', 'lu', 'cholesky', 'splu') are turned into
    one dense operator at setup and applied on the device; relaxation names run
    the corresponding device smoother from a zero guess (default 10
    iterations); None gives a zero correction.  ('dense', {'M': array}) supplies
    the operator directly.  Krylov names ('cg', 'gmres', 'bicgstab', ... and
    scipy's 'cgs', 'qmr', 'minres', 'bicg') and arbitrary callables act on the
    coarsest level's few hundred unknowns from a callback inside the cycle.
    """
    if isinstance(solver, _CoarseSolver):
        return solver
    return _CoarseSolver(solver)


class _CoarseSolver(object):
    _DENSE = ("pinv", "pinv2", "lu", "cholesky", "splu", "dense")
    _RELAX = ("gauss_seidel", "jacobi", "block_gauss_seidel", "block_jacobi", "richardson", "sor",
              "chebyshev")

    def __init__(self, solver):
        self.spec = solver
        if isinstance(solver, tuple):
            self.solver, self.kwargs = solver[0], dict(solver[1])
        else:
            self.solver, self.kwargs = solver, {}
        ok = self.solver is None or callable(self.solver) or self.solver in self._DENSE + self._RELAX + (
            "schwarz", "jacobi_ne", "gauss_seidel_ne", "gauss_seidel_nr", "bicg", "bicgstab", "cg", "cgs",
            "gmres", "qmr", "minres")
        if not ok:
            raise ValueError("unknown solver: %s" % self.solver)

    def device_form(self, A):
        """-> ('dense', M) | ('smoother', closure) | ('none', None)"""
        s = self.solver
        if s is None:
            return "none", None
        if s in self._DENSE:
            if not hasattr(self, "P"):
                if s == "dense":
                    self.P = np.asarray(self.kwargs["M"], dtype=np.float64)
                elif A.nnz == 0:
                    self.P = np.zeros(A.shape)
                elif s in ("pinv", "pinv2"):
                    self.P = scipy.linalg.pinv(A.toarray(), **self.kwargs)      # multilevel.py:608-612
                else:
                    # lu / cholesky / splu: exact inverse of the (nonsingular part of the) operator
                    self.P = scipy.linalg.pinv(A.toarray())
            return "dense", self.P
        if s in self._RELAX:
            from . import smoothing
            kw = dict(self.kwargs)
            if "iterations" not in kw:
                kw["iterations"] = 10                                           # multilevel.py:665-666
            lvl = multilevel_solver.level()
            lvl.A = A
            return "smoother", getattr(smoothing, "setup_" + str(s))(lvl, **kw)
        # Krylov names and callables (multilevel.py:642-660, 687-689): host code on the coarsest level, called from
        # inside the device cycle with the restricted right-hand side
        kw = dict(self.kwargs)
        if callable(s):
            return "callback", (lambda A_, b_: s(A_, b_, **kw))
        from . import krylov
        if "tol" not in kw:
            kw["tol"] = float(np.finfo(np.float64).eps * 1e6)                  # multilevel.py:649-657
        if s in krylov.METHODS:
            return "callback", (lambda A_, b_: krylov.solve_host(A_, b_, method=s, **kw))
        import scipy.sparse.linalg as spla

        def scipy_solve(A_, b_):
            k2 = dict(kw)
            k2["rtol"] = k2.pop("tol")
            return getattr(spla, s)(A_, b_, **k2)[0]
        return "callback", scipy_solve

    def __call__(self, A, b):
        """generic_solver.__call__ (multilevel.py:694-712): solve on the device."""
        b = np.asanyarray(b)
        if A.nnz == 0:
            return np.zeros(b.shape)
        lvl = multilevel_solver.level()
        lvl.A = sparse.csr_matrix(A) if not (sparse.isspmatrix_csr(A) or sparse.isspmatrix_bsr(A)) else A
        kind, payload = self.device_form(lvl.A)
        if kind == "callback":
            return np.asarray(payload(lvl.A, np.ravel(b))).reshape(b.shape)
        ml = multilevel_solver([lvl], coarse_solver=self)
        x = np.zeros(A.shape[0])
        ml.device_hierarchy().cycle(np.ascontiguousarray(np.ravel(b), dtype=np.float64), x, "V", x0_zero=True)
        ml._invalidate_device()
        return x.reshape(b.shape)

    def __repr__(self):
        return "coarse_grid_solver(" + repr(self.solver) + ")"

    def name(self):
        return repr(getattr(self, "solver_name", None) or self.solver)
