"""Krylov methods around the device-resident cycle -- the methods of /root/reference/pyamg/krylov that
`multilevel_solver.solve(accel=...)` (multilevel.py:381-422), the Krylov smoothers (relaxation/smoothing.py:481-509) and
the Krylov coarse solvers (multilevel.py:642-660) reach for.

Every vector lives in HBM: operator applications, preconditioner cycles and BLAS-1 updates are kernels on the hierarchy's
stream (include/amgcore_hip.h: amg_hier_apply, amg_hier_cycle with device vectors, amg_dev_*); what crosses PCIe per
iteration is a handful of scalars -- inner products, norms and, for the Householder GMRES variants, the few leading
entries of one vector that form the next Hessenberg column.  The iteration logic itself (what is computed from what,
stopping rules, residual-history semantics) follows the reference method by method; inner products are fixed-order device
reductions where the reference calls BLAS, so histories agree to rounding (pinned against histories the reference itself
produced: tests/golden/hier_accel_*.npz).
"""
import ctypes as C

import numpy as np
import scipy.linalg

from . import _lib

__all__ = ["cg", "fgmres", "gmres", "bicgstab", "cgne", "cgnr", "DeviceSpace"]

_H2D, _D2H, _D2D = 0, 1, 2


class DeviceSpace(object):
    """Vectors of one level of a device hierarchy plus the three operators a Krylov method needs:
    A (the level operator), AH (its transpose, when uploaded as the smoother slot's auxiliary operator) and
    M (one multigrid cycle from a zero guess -- level 0 only -- or the identity)."""

    def __init__(self, dev, lvl=0, cycle=None, aux=None):
        self.L = _lib.lib()
        self.dev = dev
        self.h = dev.h
        self.lvl = int(lvl)
        self.n = int(dev.level_size(lvl))
        self.stream = self.L.amg_hier_stream(self.h)
        self.scratch = self.L.amg_hier_scratch(self.h)
        if not self.scratch:
            raise _lib.AmgError(self.L.amg_last_error().decode())
        self.cycle = cycle                    # None: no preconditioner
        self.aux = aux                        # (which, slot) of the transpose operator, or None
        self._owned = []
        self._one = np.zeros(1)

    # -- storage
    def new(self, count=None):
        p = self.L.amg_dev_alloc(int(self.n if count is None else count))
        if not p:
            raise MemoryError(self.L.amg_last_error().decode())
        self._owned.append(p)
        return p

    def release(self):
        for p in self._owned:
            self.L.amg_dev_free(p)
        self._owned = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()

    def upload(self, host, dst=None):
        host = np.ascontiguousarray(np.ravel(host), dtype=np.float64)
        dst = self.new() if dst is None else dst
        _lib.check(self.L.amg_dev_copy(dst, host.ctypes.data, len(host), _H2D, self.stream))
        return dst

    def download(self, src, count=None, offset=0):
        count = self.n if count is None else count
        out = np.empty(count, dtype=np.float64)
        if count:
            _lib.check(self.L.amg_dev_copy(out.ctypes.data, src + 8 * offset, count, _D2H, self.stream))
        return out

    def poke(self, dst, offset, values):
        values = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64)
        _lib.check(self.L.amg_dev_copy(dst + 8 * offset, values.ctypes.data, len(values), _H2D, self.stream))

    def peek(self, src, offset):
        return float(self.download(src, 1, offset)[0])

    # -- BLAS-1 on (sub)vectors: `off` skips leading entries
    def copy(self, dst, src, off=0):
        if self.n - off > 0:
            _lib.check(self.L.amg_dev_copy(dst + 8 * off, src + 8 * off, self.n - off, _D2D, self.stream))

    def fill(self, x, value, off=0):
        if self.n - off > 0:
            _lib.check(self.L.amg_dev_fill(x + 8 * off, float(value), self.n - off, self.stream))

    def scale(self, out, x, c):                       # out = c * x
        _lib.check(self.L.amg_dev_scale(out, x, float(c), self.n, self.stream))

    def axpy(self, y, a, x):                          # y += a * x
        _lib.check(self.L.amg_dev_axmy(y, x, -float(a), self.n, self.stream))

    def xpby(self, p, beta, z):                       # p = beta * p + z
        _lib.check(self.L.amg_dev_scale_add(p, float(beta), z, self.n, self.stream))

    def sub(self, out, a, b):                         # out = a - b
        _lib.check(self.L.amg_dev_sub(out, a, b, self.n, self.stream))

    def dot(self, x, y):
        r = C.c_double(0.0)
        _lib.check(self.L.amg_dev_dot_host(x, y, self.n, self.scratch, C.byref(r), self.stream))
        return r.value

    def norm(self, x, off=0):
        if self.n - off <= 0:
            return 0.0
        r = C.c_double(0.0)
        _lib.check(self.L.amg_dev_norm_host(x + 8 * off, self.n - off, self.scratch, C.byref(r), self.stream))
        return r.value

    # -- operators
    def A(self, x, out):
        _lib.check(self.L.amg_hier_apply(self.h, self.lvl, 0, x, out))

    def AH(self, x, out):
        if self.aux is None:
            raise NotImplementedError("this method needs the transpose of the level operator")
        _lib.check(self.L.amg_hier_apply_aux(self.h, self.lvl, self.aux[0], self.aux[1], x, out))

    def M(self, r, out):
        if self.cycle is None:
            self.copy(out, r)
        else:
            self.dev.cycle_device(r, out, self.cycle)

    def residual(self, out, b, x, tmp):               # out = b - A x
        self.A(x, tmp)
        self.sub(out, b, tmp)


def _sign(v):
    return 1.0 if v == 0 else float(np.sign(v))


def _finish(V, x, x_host_out):
    if x_host_out is not None:
        x_host_out[:] = V.download(x)


# --------------------------------------------------------------------------- CG family
def cg(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """Preconditioned conjugate gradients (krylov/_cg.py:84-179): history in the preconditioner norm sqrt(<r, M r>),
    true residual every 8th iteration, tolerance relative to the first entry.  b, x: device vectors (x updated).
    -> info (0 converged, -1 indefinite operator / preconditioner, else the iteration count)"""
    if maxiter is None:
        maxiter = int(1.3 * V.n) + 2
    elif maxiter < 1:
        raise ValueError("Number of iterations must be positive")
    r, z, p, Ap = V.new(), V.new(), V.new(), V.new()
    V.residual(r, b, x, Ap)
    V.M(r, z)
    V.copy(p, z)
    rz = V.dot(r, z)
    normr = np.sqrt(rz)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    it = 0
    while True:
        V.A(p, Ap)
        rz_old = rz
        pAp = V.dot(Ap, p)
        if pAp < 0.0:
            return -1
        alpha = rz / pAp
        V.axpy(x, alpha, p)
        if (it % 8) and it > 0:
            V.axpy(r, -alpha, Ap)
        else:
            V.residual(r, b, x, z)
        V.M(r, z)
        rz = V.dot(r, z)
        if rz < 0.0:
            return -1
        V.xpby(p, rz / rz_old, z)
        it += 1
        normr = np.sqrt(rz)
        if residuals is not None:
            residuals.append(normr)
        if callback is not None:
            callback(V.download(x))
        if normr < tol:
            return 0
        if rz == 0.0:
            return -1
        if it == maxiter:
            return it


def cr(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """Preconditioned conjugate residuals (krylov/_cr.py:86-187): history sqrt(<z, z>) with z = M r; the true residual is
    recomputed at iterations 0, 8, 16, ... (`if mod(iter, 8) and iter > 0: r -= alpha Ap else: r = b - A x`)."""
    if maxiter is None:
        maxiter = int(1.3 * V.n) + 2
    elif maxiter < 1:
        raise ValueError("Number of iterations must be positive")
    r, z, p, Ap, Az = V.new(), V.new(), V.new(), V.new(), V.new()
    V.residual(r, b, x, Ap)
    V.M(r, z)
    V.copy(p, z)
    zz = V.dot(z, z)
    normr = np.sqrt(zz)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    it = 0
    V.A(z, Az)
    rAz = V.dot(r, Az)
    V.A(p, Ap)
    while True:
        rAz_old = rAz
        alpha = rAz / V.dot(Ap, Ap)
        V.axpy(x, alpha, p)
        if (it % 8) and it > 0:
            V.axpy(r, -alpha, Ap)
        else:
            V.residual(r, b, x, z)
        V.M(r, z)
        V.A(z, Az)
        rAz = V.dot(r, Az)
        beta = rAz / rAz_old
        V.xpby(p, beta, z)
        V.xpby(Ap, beta, Az)
        it += 1
        zz = V.dot(z, z)
        normr = np.sqrt(zz)
        if residuals is not None:
            residuals.append(normr)
        if callback is not None:
            callback(V.download(x))
        if normr < tol:
            return 0
        if zz == 0.0:
            return -1
        if it == maxiter:
            return it


def steepest_descent(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """Preconditioned steepest descent (krylov/_steepest_descent.py:83-165): history sqrt(<r, M r>); the reference
    recomputes r = b - A x whenever `mod(iter, 50)` is non-zero and updates it only at iterations 50, 100, ..."""
    if maxiter is None:
        maxiter = int(V.n)
    elif maxiter < 1:
        raise ValueError("Number of iterations must be positive")
    r, z, q = V.new(), V.new(), V.new()
    V.residual(r, b, x, q)
    V.M(r, z)
    rz = V.dot(r, z)
    normr = np.sqrt(rz)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    it = 0
    while True:
        it += 1
        V.A(z, q)
        zAz = V.dot(z, q)
        if zAz < 0.0:
            return -1
        alpha = rz / zAz
        V.axpy(x, alpha, z)
        if (it % 50) and it > 0:
            V.residual(r, b, x, z)
        else:
            V.axpy(r, -alpha, q)
        V.M(r, z)
        rz = V.dot(r, z)
        if rz < 0.0:
            return -1
        normr = np.sqrt(rz)
        if residuals is not None:
            residuals.append(normr)
        if callback is not None:
            callback(V.download(x))
        if normr < tol:
            return 0
        if rz == 0.0:
            return -1
        if it == maxiter:
            return it


def minimal_residual(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """Preconditioned minimal residual iteration (krylov/_minimal_residual.py:83-145): r = M (b - A x), p = M A r,
    alpha = <p, r> / <p, p>; history ||r||_2; same recompute pattern as steepest_descent."""
    if maxiter is None:
        maxiter = int(V.n)
    elif maxiter < 1:
        raise ValueError("Number of iterations must be positive")
    r, p, t, u = V.new(), V.new(), V.new(), V.new()
    V.residual(t, b, x, u)
    V.M(t, r)
    normr = V.norm(r)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    it = 0
    while True:
        it += 1
        V.A(r, t)
        V.M(t, p)
        rMAr = V.dot(p, r)
        if rMAr < 0.0:
            return -1
        alpha = rMAr / V.dot(p, p)
        V.axpy(x, alpha, r)
        if (it % 50) and it > 0:
            V.residual(t, b, x, u)
            V.M(t, r)
        else:
            V.axpy(r, -alpha, p)
        normr = V.norm(r)
        if residuals is not None:
            residuals.append(normr)
        if callback is not None:
            callback(V.download(x))
        if normr < tol:
            return 0
        if it == maxiter:
            return it


def cgne(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """CG on A A^H y = b, x = A^H y (krylov/_cgne.py:85-170): 2-norm history."""
    maxiter = _ne_maxiter(V.n, maxiter)
    r, z, p, t = V.new(), V.new(), V.new(), V.new()
    V.residual(r, b, x, t)
    normr = V.norm(r)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    V.M(r, z)
    V.AH(z, p)
    old_zr = V.dot(z, r)
    for it in range(maxiter):
        alpha = old_zr / V.dot(p, p)
        V.axpy(x, alpha, p)
        if (it % 8) and it > 0:
            V.A(p, t)
            V.axpy(r, -alpha, t)
        else:
            V.residual(r, b, x, t)
        V.M(r, z)
        new_zr = V.dot(z, r)
        beta = new_zr / old_zr
        old_zr = new_zr
        V.AH(z, t)
        V.xpby(p, beta, t)
        if callback is not None:
            callback(V.download(x))
        normr = V.norm(r)
        if residuals is not None:
            residuals.append(normr)
        if normr < tol:
            return 0
    return maxiter


def cgnr(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """CG on A^H A x = A^H b (krylov/_cgnr.py:85-178): 2-norm history of r = b - A x."""
    maxiter = _ne_maxiter(V.n, maxiter)
    r, rhat, z, p, w = V.new(), V.new(), V.new(), V.new(), V.new()
    V.residual(r, b, x, w)
    V.AH(r, rhat)
    normr = V.norm(r)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    V.M(rhat, z)
    V.copy(p, z)
    old_zr = V.dot(z, rhat)
    for it in range(maxiter):
        V.A(p, w)
        alpha = old_zr / V.dot(w, w)
        V.axpy(x, alpha, p)
        if (it % 8) and it > 0:
            V.axpy(r, -alpha, w)
        else:
            V.residual(r, b, x, w)
        V.AH(r, rhat)
        V.M(rhat, z)
        new_zr = V.dot(z, rhat)
        beta = new_zr / old_zr
        old_zr = new_zr
        V.xpby(p, beta, z)
        if callback is not None:
            callback(V.download(x))
        normr = V.norm(r)
        if residuals is not None:
            residuals.append(normr)
        if normr < tol:
            return 0
    return maxiter


def _ne_maxiter(n, maxiter):
    cap = int(np.ceil(1.3 * n)) + 2
    if maxiter is None:
        return cap
    if maxiter < 1:
        raise ValueError("Number of iterations must be positive")
    return cap if maxiter > 1.3 * n else int(maxiter)


def bicgstab(V, b, x, tol=1e-5, maxiter=None, residuals=None, callback=None):
    """Right-preconditioned BiCGStab (krylov/_bicgstab.py:80-150): 2-norm history."""
    if maxiter is None:
        maxiter = V.n + 5
    elif maxiter < 1:
        raise ValueError("Number of iterations must be positive")
    r, rstar, p, Mp, AMp, s_, Ms, AMs = (V.new() for _ in range(8))
    V.residual(r, b, x, Mp)
    normr = V.norm(r)
    if residuals is not None:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    V.copy(rstar, r)
    V.copy(p, r)
    rr_old = V.dot(rstar, r)
    it = 0
    while True:
        V.M(p, Mp)
        V.A(Mp, AMp)
        alpha = rr_old / V.dot(rstar, AMp)
        V.copy(s_, r)
        V.axpy(s_, -alpha, AMp)                       # s = r - alpha A M p
        V.M(s_, Ms)
        V.A(Ms, AMs)
        omega = V.dot(AMs, s_) / V.dot(AMs, AMs)
        V.axpy(x, alpha, Mp)
        V.axpy(x, omega, Ms)
        V.copy(r, s_)
        V.axpy(r, -omega, AMs)                        # r = s - omega A M s
        rr_new = V.dot(rstar, r)
        beta = (rr_new / rr_old) * (alpha / omega)
        rr_old = rr_new
        V.axpy(p, -omega, AMp)                        # p = r + beta (p - omega A M p)
        V.xpby(p, beta, r)
        it += 1
        normr = V.norm(r)
        if residuals is not None:
            residuals.append(normr)
        if callback is not None:
            callback(V.download(x))
        if normr < tol:
            return 0
        if it == maxiter:
            return it


# --------------------------------------------------------------------------- GMRES with Householder reflections
def _inner_limits(n, restrt, maxiter):
    """(outer, inner) iteration limits (krylov/_fgmres.py:136-155, _gmres_householder.py:130-149)"""
    if restrt:
        return (maxiter if maxiter else 1), min(int(restrt), n)
    if maxiter is None:
        maxiter = min(n, 40)
    return 1, min(int(maxiter), n)


def _solve_1x1(V, b, x):
    """a 1 x 1 system is solved directly: x = b / A[0, 0] (krylov/_fgmres.py:157-160, _gmres_householder.py:151-154)"""
    e, a = V.new(), V.new()
    V.fill(e, 1.0)
    V.A(e, a)
    V.copy(x, b)
    V.scale(x, x, 1.0 / V.peek(a, 0))
    return 0


def _reflect(V, v, W, j):
    """v <- (I - 2 w_j w_j^T) v  (amg_core/krylov.h:35-53: alpha = <w_j, v>; alpha *= -2; v += alpha w_j)"""
    V.axpy(v, -2.0 * V.dot(W[j], v), W[j])


def _hessenberg_step(V, v, W, inner, max_inner, Q, g, H):
    """The part of one (F)GMRES inner iteration after the operator has been applied and v holds
    P_inner ... P_0 (A ...) (krylov/_fgmres.py:219-262): the next reflector, then -- on the host, v has at most
    inner + 2 non-zero leading entries now -- the accumulated Givens rotations, the new rotation, the Hessenberg column."""
    n = V.n
    if inner != n - 1:
        if inner < max_inner - 1:
            # the reference starts every restart cycle from zeroed reflectors (W = zeros(...), _fgmres.py:195): after a
            # breakdown (alpha == 0) the next step must not find the previous cycle's vector here
            V.fill(W[inner + 1], 0.0)
        alpha = V.norm(v, off=inner + 1)
        if alpha != 0:
            alpha = _sign(V.peek(v, inner + 1)) * alpha
            if inner < max_inner - 1:
                w = W[inner + 1]
                V.copy(w, v, off=inner + 1)
                V.poke(w, inner + 1, V.peek(w, inner + 1) + alpha)
                V.scale(w, w, 1.0 / V.norm(w))
            V.poke(v, inner + 1, -alpha)
            V.fill(v, 0.0, off=inner + 2)
    head = V.download(v, min(n, inner + 2))
    for j in range(inner):                            # amg_core/krylov.h apply_givens: rotations 0 .. inner-1 in order
        c, s, ms, c2 = Q[4 * j:4 * j + 4]
        a, bb = head[j], head[j + 1]
        head[j] = c * a + s * bb
        head[j + 1] = ms * a + c2 * bb
    if inner != n - 1 and head[inner + 1] != 0:
        c, s = scipy.linalg.blas.drotg(head[inner], head[inner + 1])
        Q[4 * inner:4 * inner + 4] = (c, s, -s, c)
        g[inner:inner + 2] = (c * g[inner] + s * g[inner + 1], -s * g[inner] + c * g[inner + 1])
        head[inner] = c * head[inner] + s * head[inner + 1]
        head[inner + 1] = 0.0
    m = min(max_inner, len(head))
    H[:m, inner] = head[:m]


def fgmres(V, b, x, tol=1e-5, restrt=None, maxiter=None, residuals=None, callback=None):
    """Flexible GMRES, right preconditioning, Householder orthogonalisation (krylov/_fgmres.py:118-305); history:
    the 2-norm of the (true) residual, estimated through the rotated right-hand side inside a restart cycle."""
    n = V.n
    if n == 1:
        return _solve_1x1(V, b, x)
    max_outer, max_inner = _inner_limits(n, restrt, maxiter)
    r, v, t = V.new(), V.new(), V.new()
    V.residual(r, b, x, t)
    normr = V.norm(r)
    keep = residuals is not None
    if keep:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    W = [V.new() for _ in range(max_inner)]
    Z = [V.new() for _ in range(max_inner)]
    niter = 0
    for outer in range(max_outer):
        w = W[0]
        V.copy(w, r)
        beta = _sign(V.peek(w, 0)) * normr
        V.poke(w, 0, V.peek(w, 0) + beta)
        V.scale(w, w, 1.0 / V.norm(w))
        Q = np.zeros(4 * max_inner)
        H = np.zeros((max_inner, max_inner))
        g = np.zeros(n if n < 4096 else max_inner + 2)
        g[0] = -beta
        inner = 0
        for inner in range(max_inner):
            w = W[inner]
            V.scale(v, w, -2.0 * V.peek(w, inner))                 # v = P_inner e_inner ...
            V.poke(v, inner, V.peek(v, inner) + 1.0)
            for j in range(inner - 1, -1, -1):                    # ... = P_0 ... P_inner e_inner
                _reflect(V, v, W, j)
            V.M(v, Z[inner])
            V.A(Z[inner], v)
            for j in range(0, inner + 1):
                _reflect(V, v, W, j)
            _hessenberg_step(V, v, W, inner, max_inner, Q, g, H)
            if inner < max_inner - 1:
                normr = abs(g[inner + 1])
                if normr < tol:
                    break
                if callback is not None:
                    callback(normr)
                if keep:
                    residuals.append(normr)
            niter += 1
        y = scipy.linalg.solve(H[:inner + 1, :inner + 1], g[:inner + 1])
        V.fill(t, 0.0)                                             # update = Z[:, :inner+1] y
        for k in range(inner + 1):
            V.axpy(t, y[k], Z[k])
        V.axpy(x, 1.0, t)
        V.residual(r, b, x, v)
        normr = V.norm(r)
        if callback is not None:
            callback(normr)
        if keep:
            residuals.append(normr)
        if _stagnated(V, t, x):
            return -1
        if normr < tol:
            return 0
    return niter


def gmres(V, b, x, tol=1e-5, restrt=None, maxiter=None, residuals=None, callback=None):
    """GMRES with LEFT preconditioning and Householder orthogonalisation (krylov/_gmres_householder.py:107-268, the
    reference's default `orthog`); history: the norm of the preconditioned residual M (b - A x)."""
    n = V.n
    if n == 1:
        return _solve_1x1(V, b, x)
    max_outer, max_inner = _inner_limits(n, restrt, maxiter)
    r, v, t = V.new(), V.new(), V.new()
    V.residual(t, b, x, v)
    V.M(t, r)
    normr = V.norm(r)
    keep = residuals is not None
    if keep:
        residuals[:] = [normr]
    normb = V.norm(b) or 1.0
    if normr < tol * normb:
        return 0
    if normr != 0.0:
        tol = tol * normr
    W = [V.new() for _ in range(max_inner + 1)]
    niter = 0
    for outer in range(max_outer):
        w = W[0]
        V.copy(w, r)
        beta = _sign(V.peek(w, 0)) * normr
        V.poke(w, 0, V.peek(w, 0) + beta)
        V.scale(w, w, 1.0 / V.norm(w))
        Q = np.zeros(4 * max_inner)
        H = np.zeros((max_inner, max_inner))
        g = np.zeros(n if n < 4096 else max_inner + 2)
        g[0] = -beta
        inner = 0
        for inner in range(max_inner):
            w = W[inner]
            V.scale(v, w, -2.0 * V.peek(w, inner))
            V.poke(v, inner, V.peek(v, inner) + 1.0)
            for j in range(inner - 1, -1, -1):
                _reflect(V, v, W, j)
            V.A(v, t)
            V.M(t, v)
            for j in range(0, inner + 1):
                _reflect(V, v, W, j)
            _hessenberg_step(V, v, W, inner, max_inner, Q, g, H)
            niter += 1
            if inner < max_inner - 1:
                normr = abs(g[inner + 1])
                if normr < tol:
                    break
                if callback is not None:
                    callback(normr)
                if keep:
                    residuals.append(normr)
        y = scipy.linalg.solve(H[:inner + 1, :inner + 1], g[:inner + 1])
        # amg_core/krylov.h householder_hornerscheme: for j = inner .. 0: update[j] += y[j]; update <- P_j update
        V.fill(t, 0.0)
        for j in range(inner, -1, -1):
            V.poke(t, j, V.peek(t, j) + y[j])
            _reflect(V, t, W, j)
        V.axpy(x, 1.0, t)
        V.residual(v, b, x, r)
        V.M(v, r)
        normr = V.norm(r)
        if callback is not None:
            callback(normr)
        if keep:
            residuals.append(normr)
        if _stagnated(V, t, x):
            return -1
        if normr < tol:
            return 0
    return niter


def _stagnated(V, update, x):
    """max |update_i / x_i| over x_i != 0 below 1e-12 (krylov/_fgmres.py:293-297): checked on the host copy of the two
    vectors only when the update is tiny in norm to begin with, which is the only way the entrywise test can hold"""
    nu, nx = V.norm(update), V.norm(x)
    if nx == 0.0 or nu > 1e-10 * nx:
        return False
    u, xx = V.download(update), V.download(x)
    idx = xx != 0
    return bool(idx.any() and np.max(np.abs(u[idx] / xx[idx])) < 1e-12)


# --------------------------------------------------------------------------- convenience: host vectors in and out
METHODS = {"cg": cg, "fgmres": fgmres, "gmres": gmres, "bicgstab": bicgstab, "cgne": cgne, "cgnr": cgnr,
           "cr": cr, "steepest_descent": steepest_descent, "minimal_residual": minimal_residual}
_RESTARTED = ("fgmres", "gmres")


def run(method, V, b, x, tol, maxiter, restrt=None, residuals=None, callback=None):
    """dispatch by name with the arguments each method takes"""
    fn = METHODS[method]
    if method in _RESTARTED:
        return fn(V, b, x, tol=tol, restrt=restrt, maxiter=maxiter, residuals=residuals, callback=callback)
    return fn(V, b, x, tol=tol, maxiter=maxiter, residuals=residuals, callback=callback)


def solve_host(A, b, x0=None, method="cg", tol=1e-5, maxiter=None, restrt=None, residuals=None):
    """x ~ A^-1 b for a scipy operator and host vectors with the UNPRECONDITIONED method on the device (what the
    reference's Krylov smoothers and Krylov coarse solvers call, smoothing.py:481-509, multilevel.py:642-660)."""
    from scipy.sparse import csr_matrix, isspmatrix_bsr, isspmatrix_csr
    from .util import _DeviceOperator
    if not (isspmatrix_csr(A) or isspmatrix_bsr(A)):
        A = csr_matrix(A)
    op = _DeviceOperator(A)
    try:
        op.level_size = lambda lvl: A.shape[0]
        aux = None
        if method in ("cgne", "cgnr"):
            At = csr_matrix(A.T)
            At.sort_indices()
            Ap = np.ascontiguousarray(At.indptr, dtype=np.intc)
            Aj = np.ascontiguousarray(At.indices, dtype=np.intc)
            Ax = np.ascontiguousarray(At.data, dtype=np.float64)
            _lib.check(op.L.amg_hier_set_aux_matrix(op.h, 0, 0, 0, At.shape[0], At.shape[1], _lib.ip(Ap), _lib.ip(Aj), _lib.dp(Ax)))
            aux = (0, 0)
        with DeviceSpace(op, 0, cycle=None, aux=aux) as V:
            bd = V.upload(b)
            xd = V.upload(np.zeros(A.shape[0]) if x0 is None else x0)
            run(method, V, bd, xd, tol, maxiter, restrt=restrt, residuals=residuals)
            return V.download(xd)
    finally:
        op.close()
