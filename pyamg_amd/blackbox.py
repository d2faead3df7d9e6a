"""Solve an arbitrary system Ax = b with out-of-the-box settings -- the entry points of
/root/reference/pyamg/blackbox.py (`solver_configuration`, `solver`, `solve`) on the device path.

The reference configures energy-minimisation prolongation smoothing and evolution strength here;
those setup methods are outside this library's scope (out of scope), so the
configuration uses what the restated setup offers -- symmetric strength, standard aggregation,
Jacobi-smoothed prolongation -- with the reference's choice of relaxation (symmetric block
Gauss-Seidel) and Krylov acceleration (CG), both of which run on the GPU.  Non-Hermitian
operators are refused rather than solved with a configuration the reference would not use.
"""
import numpy as np
from scipy.sparse import csr_matrix, isspmatrix_bsr, isspmatrix_csr

__all__ = ["solve", "solver", "solver_configuration"]

# parity with the reference's blackbox is UNPINNED: no fixture of the reference covers it and the configuration differs
SUBSTITUTED = ("substituted configuration: symmetric strength + Jacobi prolongation smoothing where the reference's "
               "blackbox.py:56-158 uses evolution strength + energy minimisation; hierarchies and iterates differ from the reference's")


def _csr_or_bsr(A):
    """blackbox.py:15-53: accept CSR / BSR, convert anything else to CSR, insist on a square operator"""
    if not (isspmatrix_csr(A) or isspmatrix_bsr(A)):
        try:
            A = csr_matrix(A)
        except Exception:
            raise TypeError("Argument A must have type csr_matrix or bsr_matrix, or be convertible to csr_matrix")
    if A.shape[0] != A.shape[1]:
        raise TypeError("Argument A must be a square")
    return A.astype(np.float64) if A.dtype != np.float64 else A


def _is_symmetric(A, samples=3):
    """cheap probabilistic test: x^T A y == y^T A x for a few random pairs"""
    rng = np.random.RandomState(7)
    scale = abs(A).sum() / max(A.shape[0], 1) + 1e-300
    for _ in range(samples):
        x, y = rng.rand(A.shape[0]), rng.rand(A.shape[0])
        if abs(x.dot(A * y) - y.dot(A * x)) > 1e-10 * scale * A.shape[0]:
            return False
    return True


def solver_configuration(A, B=None, verb=True):
    """Keyword arguments for `smoothed_aggregation_solver` (blackbox.py:56-158)."""
    A = _csr_or_bsr(A)
    if not _is_symmetric(A):
        raise NotImplementedError("blackbox configuration: only symmetric operators have a device configuration")
    if verb:
        print("  Detected a Hermitian matrix")
        print("  " + SUBSTITUTED)
    bs = A.blocksize[0] if isspmatrix_bsr(A) else 1
    if B is None:
        B = np.kron(np.ones((A.shape[0] // bs, 1)), np.eye(bs))
    else:
        B = np.array(B, dtype=np.float64)
        B = B.reshape(-1, 1) if B.ndim == 1 else B
        if B.shape[0] != A.shape[0] or B.shape[1] == 0:
            raise TypeError("Invalid dimensions of B, B.shape[0] must equal A.shape[0]")
    relax = ("block_gauss_seidel", {"sweep": "symmetric", "iterations": 1})
    return {"note": SUBSTITUTED, "symmetry": "hermitian", "B": B, "BH": None, "strength": "symmetric", "aggregate": "standard",
            "smooth": ("jacobi", {"omega": 4.0 / 3.0}), "presmoother": relax, "postsmoother": relax,
            "max_levels": 15, "max_coarse": 500, "coarse_solver": "pinv", "keep": False}


def _setup_arguments(config):
    """the configuration without its annotation (which is not a setup argument)"""
    return {k: v for k, v in config.items() if k != "note"}


def solver(A, config):
    """A smoothed-aggregation hierarchy from a configuration dictionary (blackbox.py:161-216)."""
    from .aggregation import smoothed_aggregation_solver
    try:
        return smoothed_aggregation_solver(_csr_or_bsr(A), **_setup_arguments(config))
    except Exception:
        raise TypeError("Failed generating smoothed_aggregation_solver")


def solve(A, b, x0=None, tol=1e-5, maxiter=400, return_solver=False, existing_solver=None, verb=True,
          residuals=None):
    """x with ||b - A x|| reduced by `tol` (blackbox.py:219-330): build (or reuse) the hierarchy and run the
    Krylov-accelerated cycle on the device; x0 defaults to a random vector as in the reference."""
    A = _csr_or_bsr(A)
    ml = existing_solver
    if ml is None:
        ml = solver(A, solver_configuration(A, B=None, verb=verb))
    elif ml.levels[0].A.shape[0] != A.shape[0]:
        raise TypeError("Argument existing_solver must have level 0 matrix of same size as A")
    b = np.asarray(b, dtype=np.float64)
    if x0 is None:
        x0 = np.random.rand(A.shape[0])
    count = [0]

    def progress(_x):
        count[0] += 1
        print("    iteration %d" % count[0])
    if verb:
        print("    maxiter = %d" % maxiter)
    x = ml.solve(b, x0=x0, accel="cg", tol=tol, maxiter=maxiter, callback=progress if verb else None,
                 residuals=residuals)
    if verb:
        r0 = np.linalg.norm(np.ravel(b) - A * np.ravel(x0))
        rk = np.linalg.norm(np.ravel(b) - A * np.ravel(x))
        print("  Residual reduction ||r_k||/||r_0|| = %1.2e" % (rk / r0) if r0 != 0.0 else
              "  Residuals ||r_k||, ||r_0|| = %1.2e, %1.2e" % (rk, r0))
    x = np.asarray(x).reshape(b.shape)
    return (x, ml) if return_solver else x
