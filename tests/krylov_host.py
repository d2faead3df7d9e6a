"""TEST INFRASTRUCTURE: host restatements of the reference's Krylov methods (pyamg/krylov), each citing the
lines it follows, written against two callables -- `A(v)` (operator) and `M(v)` (preconditioner).  They are
pinned by the histories the reference itself produced (tests/golden/hier_accel_*.npz) and then serve as the
order-of-operations oracle for the device implementations."""
import numpy as np


def cg(A, M, b, x0, tol, maxiter, recompute_every=8):
    """pyamg/krylov/_cg.py:84-179: preconditioned CG; the history is the PRECONDITIONER norm sqrt(<r, M r>);
    the true residual replaces the recurrence every 8th iteration; tol is relative to the first entry.
    -> (x, residuals, info)"""
    x = np.array(x0, dtype=np.float64)
    r = b - A(x)
    z = M(r)
    p = z.copy()
    rz = float(np.inner(r, z))
    res = [np.sqrt(rz)]
    normb = np.linalg.norm(b) or 1.0
    if res[0] < tol * normb:
        return x, res, 0
    if res[0] != 0.0:
        tol = tol * res[0]
    it = 0
    while True:
        Ap = A(p)
        rz_old = rz
        pAp = float(np.inner(Ap, p))
        if pAp < 0.0:
            return x, res, -1
        alpha = rz / pAp
        x += alpha * p
        if (it % recompute_every) and it > 0:
            r -= alpha * Ap
        else:
            r = b - A(x)
        z = M(r)
        rz = float(np.inner(r, z))
        if rz < 0.0:
            return x, res, -1
        p *= rz / rz_old
        p += z
        it += 1
        res.append(np.sqrt(rz))
        if res[-1] < tol:
            return x, res, 0
        if rz == 0.0:
            return x, res, -1
        if it == maxiter:
            return x, res, it
