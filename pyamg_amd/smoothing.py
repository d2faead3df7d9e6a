"""Method to create pre- and post-smoothers on the levels of a multilevel_solver
-- mirror of /root/reference/pyamg/relaxation/smoothing.py.

``change_smoothers(ml, presmoother, postsmoother)`` accepts the reference's
descriptors: ``'name'``, ``('name', {opts})``, ``None`` or a per-level list.
Each ``setup_<name>(lvl, **opts)`` returns a closure ``smoother(A, x, b)`` like
the reference's; the closure additionally carries ``.desc``, the constants
(omega after rho scaling, Chebyshev coefficients, Dinv ...) that the
device-resident cycle uploads once.
"""
import numpy as np
import scipy.sparse as sparse

from . import relaxation
from .chebyshev import chebyshev_polynomial_coefficients
from .util import (approximate_spectral_radius, approximate_spectral_radius_device, get_block_diag,
                   get_diagonal, release_device_operator, scale_rows, use_device_for)

__all__ = ["change_smoothers", "rho_D_inv_A", "rho_block_D_inv_A"]

# names the device cycle implements; the rest of the reference's list
# (cg, gmres, cgne, cgnr: the Krylov smoothers) is outside the hot path
DEVICE_SMOOTHERS = ("gauss_seidel", "jacobi", "block_jacobi", "block_gauss_seidel", "richardson",
                    "sor", "chebyshev", "polynomial", "gauss_seidel_indexed", "schwarz",
                    "strength_based_schwarz", "jacobi_ne", "gauss_seidel_ne", "gauss_seidel_nr", "None")


def unpack_arg(v):
    if isinstance(v, tuple):
        return v[0], v[1]
    return v, {}


def _lookup(fn, which):
    try:
        return globals()["setup_" + str(fn)]
    except KeyError:
        raise NameError("invalid %s method: " % which, fn)


def change_smoothers(ml, presmoother, postsmoother):
    """smoothing.py:24-169"""
    if isinstance(presmoother, str) or isinstance(presmoother, tuple) or (presmoother is None):
        presmoother = [presmoother]
    elif not isinstance(presmoother, list):
        raise ValueError("Unrecognized presmoother")
    if isinstance(postsmoother, str) or isinstance(postsmoother, tuple) or (postsmoother is None):
        postsmoother = [postsmoother]
    elif not isinstance(postsmoother, list):
        raise ValueError("Unrecognized postsmoother")

    for side, spec in (("presmoother", presmoother), ("postsmoother", postsmoother)):
        i = 0
        setup, kwargs = None, {}
        for i in range(min(len(spec), len(ml.levels[:-1]))):
            fn, kwargs = unpack_arg(spec[i])
            setup = _lookup(fn, side)
            setattr(ml.levels[i], side, setup(ml.levels[i], **kwargs))
        for j in range(i + 1, len(ml.levels[:-1])):
            setattr(ml.levels[j], side, setup(ml.levels[j], **kwargs))
    # the HBM copies of the level operators that served the setup-time estimates (util.device_operator)
    # are not needed any more: the solve uploads its own mirror of the hierarchy
    for lvl in ml.levels:
        release_device_operator(lvl.A)
    if hasattr(ml, "_invalidate_device"):
        ml._invalidate_device()


def rho_D_inv_A(A):
    """smoothing.py:172-200"""
    if not hasattr(A, "rho_D_inv"):
        D_inv = get_diagonal(A, inv=True)
        if use_device_for(A):
            A.rho_D_inv = approximate_spectral_radius_device(A, D_inv)
        else:
            D_inv_A = scale_rows(A, D_inv, copy=True)
            A.rho_D_inv = approximate_spectral_radius(D_inv_A)
    return A.rho_D_inv


def rho_block_D_inv_A(A, Dinv):
    """smoothing.py:203-250"""
    if not hasattr(A, "rho_block_D_inv"):
        from scipy.sparse.linalg import LinearOperator
        blocksize = Dinv.shape[1]
        if Dinv.shape[1] != Dinv.shape[2]:
            raise ValueError("Dinv has incorrect dimensions")
        elif Dinv.shape[0] != int(A.shape[0] / blocksize):
            raise ValueError("Dinv and A have incompatible dimensions")
        Dm = sparse.bsr_matrix((Dinv, np.arange(Dinv.shape[0]), np.arange(Dinv.shape[0] + 1)), shape=A.shape)

        def matvec(x):
            return Dm * (A * x)
        D_inv_A = LinearOperator(A.shape, matvec, dtype=A.dtype)
        A.rho_block_D_inv = approximate_spectral_radius(D_inv_A)
    return A.rho_block_D_inv


def _with_desc(fn, **desc):
    fn.desc = desc
    return fn


def setup_gauss_seidel(lvl, iterations=1, sweep="forward"):
    def smoother(A, x, b):
        relaxation.gauss_seidel(A, x, b, iterations=iterations, sweep=sweep)
    return _with_desc(smoother, name="gauss_seidel", iterations=iterations, sweep=sweep)


def setup_jacobi(lvl, iterations=1, omega=1.0, withrho=True):
    if withrho:
        omega = omega / rho_D_inv_A(lvl.A)

    def smoother(A, x, b):
        relaxation.jacobi(A, x, b, iterations=iterations, omega=omega)
    return _with_desc(smoother, name="jacobi", iterations=iterations, omega=float(omega))


def _blocksize_of(lvl, blocksize, Dinv):
    if blocksize is None and Dinv is None:
        if sparse.isspmatrix_csr(lvl.A):
            blocksize = 1
        elif sparse.isspmatrix_bsr(lvl.A):
            blocksize = lvl.A.blocksize[0]
    elif blocksize is None:
        blocksize = Dinv.shape[1]
    return blocksize


def _block_inverses(A, blocksize):
    """get_block_diag(A, blocksize, inv_flag=True), kept on the operator: pre-smoother, post-smoother and the
    candidate improvement of the setup all ask for the same constants (the reference recomputes them)"""
    cache = getattr(A, "_amg_block_dinv", None)
    if cache is None:
        cache = {}
        try:
            A._amg_block_dinv = cache
        except AttributeError:
            pass
    if blocksize not in cache:
        cache[blocksize] = get_block_diag(A, blocksize=blocksize, inv_flag=True)
    return cache[blocksize]


def setup_block_jacobi(lvl, iterations=1, omega=1.0, Dinv=None, blocksize=None, withrho=True):
    blocksize = _blocksize_of(lvl, blocksize, Dinv)
    if blocksize == 1:
        return setup_jacobi(lvl, iterations=iterations, omega=omega, withrho=withrho)
    if Dinv is None:
        Dinv = _block_inverses(lvl.A, blocksize)
    if withrho:
        omega = omega / rho_block_D_inv_A(lvl.A, Dinv)

    def smoother(A, x, b):
        relaxation.block_jacobi(A, x, b, iterations=iterations, omega=omega, Dinv=Dinv, blocksize=blocksize)
    return _with_desc(smoother, name="block_jacobi", iterations=iterations, omega=float(omega),
                      Dinv=Dinv, blocksize=blocksize)


def setup_block_gauss_seidel(lvl, iterations=1, sweep="forward", Dinv=None, blocksize=None):
    blocksize = _blocksize_of(lvl, blocksize, Dinv)
    if blocksize == 1:
        return setup_gauss_seidel(lvl, iterations=iterations, sweep=sweep)
    if Dinv is None:
        Dinv = _block_inverses(lvl.A, blocksize)

    def smoother(A, x, b):
        relaxation.block_gauss_seidel(A, x, b, iterations=iterations, Dinv=Dinv, blocksize=blocksize,
                                      sweep=sweep)
    return _with_desc(smoother, name="block_gauss_seidel", iterations=iterations, sweep=sweep,
                      Dinv=Dinv, blocksize=blocksize)


def setup_richardson(lvl, iterations=1, omega=1.0):
    omega = omega / approximate_spectral_radius(lvl.A)

    def smoother(A, x, b):
        relaxation.polynomial(A, x, b, coefficients=[omega], iterations=iterations)
    return _with_desc(smoother, name="polynomial", iterations=iterations, coefficients=[float(omega)])


def setup_sor(lvl, omega=0.5, iterations=1, sweep="forward"):
    def smoother(A, x, b):
        relaxation.sor(A, x, b, omega=omega, iterations=iterations, sweep=sweep)
    return _with_desc(smoother, name="sor", iterations=iterations, sweep=sweep, omega=float(omega))


def setup_chebyshev(lvl, lower_bound=1.0 / 30.0, upper_bound=1.1, degree=3, iterations=1):
    rho = approximate_spectral_radius(lvl.A)
    a = rho * lower_bound
    b = rho * upper_bound
    coefficients = -chebyshev_polynomial_coefficients(a, b, degree)[:-1]
    return setup_polynomial(lvl, coefficients=coefficients, iterations=iterations)


def setup_polynomial(lvl, coefficients=None, iterations=1):
    """Extension: polynomial smoother with explicit Horner coefficients (what
    setup_chebyshev / setup_richardson reduce to, smoothing.py:422-449)."""
    coefficients = np.array(coefficients, dtype=float)

    def smoother(A, x, b):
        relaxation.polynomial(A, x, b, coefficients=coefficients, iterations=iterations)
    return _with_desc(smoother, name="polynomial", iterations=iterations,
                      coefficients=[float(c) for c in coefficients])


def setup_gauss_seidel_indexed(lvl, indices=None, iterations=1, sweep="forward"):
    """Extension: relaxation.gauss_seidel_indexed (relaxation.py:671-741) as a level smoother
    (e.g. multicolour or F/C-ordered Gauss-Seidel)."""
    indices = np.asarray(indices, dtype=np.intc)

    def smoother(A, x, b):
        relaxation.gauss_seidel_indexed(A, x, b, indices, iterations=iterations, sweep=sweep)
    return _with_desc(smoother, name="gauss_seidel_indexed", iterations=iterations, sweep=sweep,
                      indices=indices)


def setup_multicolor_gauss_seidel(lvl, iterations=1, sweep="forward"):
    """Extension: Gauss-Seidel in a multicolour ordering -- rows sorted by a
    greedy colouring, relaxed with the reference's gauss_seidel_indexed semantics
    (relaxation.py:671-741).  Different iterates from lexicographic gauss_seidel, but only as many
    dependency levels per sweep as colours (2 for a 7-point stencil), so it runs at SpMV speed."""
    from .aggregation import greedy_colouring
    colour, ncol = greedy_colouring(lvl.A)
    key = colour.astype(np.int16) if ncol < 32768 else colour     # 16-bit keys: numpy's stable sort is a radix sort there
    indices = np.argsort(key, kind="stable").astype(np.intc)
    sm = setup_gauss_seidel_indexed(lvl, indices=indices, iterations=iterations, sweep=sweep)
    sm.ncolours = ncol
    return sm


def setup_jacobi_ne(lvl, iterations=1, omega=1.0, withrho=True):
    """smoothing.py:452-460"""
    Acsr = lvl.A.tocsr()
    if withrho:
        omega = omega / rho_D_inv_A(Acsr) ** 2
    Dinv = np.ravel(get_diagonal(Acsr, norm_eq=2, inv=True))

    def smoother(A, x, b):
        relaxation.jacobi_ne(Acsr, x, b, iterations=iterations, omega=omega)
    return _with_desc(smoother, name="jacobi_ne", iterations=iterations, omega=float(omega), Dinv=Dinv)


def setup_gauss_seidel_ne(lvl, iterations=1, sweep="forward", omega=1.0):
    """smoothing.py:463-469"""
    Acsr = lvl.A.tocsr()
    Dinv = np.ravel(get_diagonal(Acsr, norm_eq=2, inv=True))

    def smoother(A, x, b):
        relaxation.gauss_seidel_ne(Acsr, x, b, iterations=iterations, sweep=sweep, omega=omega, Dinv=Dinv)
    return _with_desc(smoother, name="gauss_seidel_ne", iterations=iterations, sweep=sweep, omega=float(omega),
                      Dinv=Dinv)


def setup_gauss_seidel_nr(lvl, iterations=1, sweep="forward", omega=1.0):
    """smoothing.py:472-478"""
    Acsc = lvl.A.tocsc()
    Dinv = np.ravel(get_diagonal(Acsc, norm_eq=1, inv=True))

    def smoother(A, x, b):
        relaxation.gauss_seidel_nr(Acsc, x, b, iterations=iterations, sweep=sweep, omega=omega, Dinv=Dinv)
    return _with_desc(smoother, name="gauss_seidel_nr", iterations=iterations, sweep=sweep, omega=float(omega),
                      Dinv=Dinv)


def setup_schwarz(lvl, iterations=1, subdomain=None, subdomain_ptr=None, inv_subblock=None,
                  inv_subblock_ptr=None, sweep="symmetric"):
    """smoothing.py:335-348"""
    Acsr = lvl.A.tocsr()
    Acsr.sort_indices()
    lvl.Acsr = Acsr
    subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr = \
        relaxation.schwarz_parameters(Acsr, subdomain, subdomain_ptr, inv_subblock, inv_subblock_ptr)

    def smoother(A, x, b):
        relaxation.schwarz(Acsr, x, b, iterations=iterations, subdomain=subdomain, subdomain_ptr=subdomain_ptr,
                           inv_subblock=inv_subblock, inv_subblock_ptr=inv_subblock_ptr, sweep=sweep)
    return _with_desc(smoother, name="schwarz", iterations=iterations, sweep=sweep, subdomain=subdomain,
                      subdomain_ptr=subdomain_ptr, inv_subblock=inv_subblock, inv_subblock_ptr=inv_subblock_ptr)


def setup_strength_based_schwarz(lvl, iterations=1, sweep="symmetric"):
    """smoothing.py:351-363: subdomains from the strength-of-connection matrix of the level"""
    Cm = lvl.C.tocsr() if hasattr(lvl, "C") else lvl.A.tocsr()
    Cm.sort_indices()
    return setup_schwarz(lvl, iterations=iterations, subdomain=Cm.indices.copy(),
                         subdomain_ptr=Cm.indptr.copy(), sweep=sweep)


def _setup_krylov(method, lvl, tol, maxiter, restrt=None, M=None, callback=None, residuals=None):
    """smoothing.py:481-509: x <- method(A, b, x0=x, tol, maxiter) with no preconditioner as the level's relaxation;
    on the device the iterations run on the level's resident vectors (pyamg_amd.krylov, from a callback in the cycle)"""
    if M is not None or callback is not None or residuals is not None:
        raise NotImplementedError("Krylov smoothers run unpreconditioned and silent on the device")
    from . import krylov

    def smoother(A, x, b):
        x[:] = krylov.solve_host(A, b, x0=x, method=method, tol=tol, maxiter=maxiter, restrt=restrt).reshape(x.shape)
    return _with_desc(smoother, name="krylov", method=method, tol=float(tol), maxiter=maxiter, restrt=restrt)


def setup_gmres(lvl, tol=1e-12, maxiter=1, restrt=None, M=None, callback=None, residuals=None):
    return _setup_krylov("gmres", lvl, tol, maxiter, restrt, M, callback, residuals)


def setup_cg(lvl, tol=1e-12, maxiter=1, M=None, callback=None, residuals=None):
    return _setup_krylov("cg", lvl, tol, maxiter, None, M, callback, residuals)


def setup_cgne(lvl, tol=1e-12, maxiter=1, M=None, callback=None, residuals=None):
    return _setup_krylov("cgne", lvl, tol, maxiter, None, M, callback, residuals)


def setup_cgnr(lvl, tol=1e-12, maxiter=1, M=None, callback=None, residuals=None):
    return _setup_krylov("cgnr", lvl, tol, maxiter, None, M, callback, residuals)


def setup_None(lvl):
    def smoother(A, x, b):
        pass
    return _with_desc(smoother, name=None)


# --------------------------------------------------------------------------- descriptors back to smoothers
def spec_from_descriptor(desc):
    """The ('name', {options}) pair that re-creates a smoother from its descriptor (`fn.desc`) with every constant
    passed explicitly -- nothing is re-estimated (omega already carries its 1/rho, coefficients are final)."""
    if desc is None or desc.get("name") in (None, "None"):
        return None, {}
    name = desc["name"]
    it = int(desc.get("iterations", 1))
    sweep = desc.get("sweep", "forward")
    if name == "jacobi":
        return name, {"iterations": it, "omega": desc["omega"], "withrho": False}
    if name == "gauss_seidel":
        return name, {"iterations": it, "sweep": sweep}
    if name == "sor":
        return name, {"iterations": it, "omega": desc["omega"], "sweep": sweep}
    if name == "polynomial":
        return name, {"iterations": it, "coefficients": np.asarray(desc["coefficients"], dtype=float)}
    if name in ("block_jacobi", "block_gauss_seidel"):
        bs = int(desc["blocksize"])
        kw = {"iterations": it, "blocksize": bs, "Dinv": np.asarray(desc["Dinv"], dtype=float).reshape(-1, bs, bs)}
        if name == "block_jacobi":
            kw.update(omega=desc["omega"], withrho=False)
        else:
            kw["sweep"] = sweep
        return name, kw
    if name == "gauss_seidel_indexed":
        return name, {"iterations": it, "sweep": sweep, "indices": np.asarray(desc["indices"], dtype=np.intc)}
    if name in ("gauss_seidel_ne", "gauss_seidel_nr"):
        return name, {"iterations": it, "sweep": sweep, "omega": desc.get("omega", 1.0)}
    if name == "jacobi_ne":
        return name, {"iterations": it, "omega": desc["omega"], "withrho": False}
    if name == "krylov":
        kw = {"tol": desc["tol"], "maxiter": desc["maxiter"]}
        if desc["method"] == "gmres":
            kw["restrt"] = desc.get("restrt")
        return desc["method"], kw
    if name == "schwarz":
        return name, {"iterations": it, "sweep": desc.get("sweep", "symmetric"),
                      "subdomain": np.asarray(desc["subdomain"], dtype=np.intc),
                      "subdomain_ptr": np.asarray(desc["subdomain_ptr"], dtype=np.intc),
                      "inv_subblock": np.asarray(desc["inv_subblock"], dtype=float),
                      "inv_subblock_ptr": np.asarray(desc["inv_subblock_ptr"], dtype=np.intc)}
    raise KeyError("no smoother named %r" % (name,))


def smoother_from_descriptor(lvl, desc):
    """the level smoother a descriptor describes (setup_* closure with .desc), or the no-op smoother"""
    name, kw = spec_from_descriptor(desc)
    if name is None:
        return setup_None(lvl)
    return globals()["setup_" + name](lvl, **kw)
