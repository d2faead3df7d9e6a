"""TEST INFRASTRUCTURE (container-only): stage the Python reference so it imports.

This module is used ONLY by ``oracle/gen_golden.py`` in the development
container to produce the small golden fixtures under ``tests/golden/``.  It is
never imported by the product (``pyamg_amd``), by ``bench.py`` or by any test:
``/root/reference`` does not exist on the GPU box.

Recipe ("Workable oracle recipe"):
  1. copy ``/root/reference/pyamg`` to a scratch directory under ``/tmp``
     (nothing from the reference ever enters the repository);
  2. drop the reference's own native module, built by ``oracle/Makefile`` from
     ``/root/reference/pyamg/amg_core/amg_core_wrap.cxx`` into
     ``oracle/_ref/_amg_core.so``, next to the scratch ``amg_core`` package;
  3. run ``lib2to3 -f print -f import`` on the scratch copy (three files of the
     fork still use Python-2 syntax);
  4. install interpreter-level aliases for the numpy/scipy names the 2016 code
     base expects (``np.float``, ``sp.rand``, ``scipy.linalg.pinv2`` ...).
     These are aliases onto the *current* numpy/scipy implementations; no
     reference algorithm is replaced.

The result is the reference's own Python driving the reference's own C++.
"""
import builtins
import importlib
import os
import shutil
import subprocess
import sys
import types

REFERENCE = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REF_SO = os.path.join(HERE, "_ref", "_amg_core.so")
SCRATCH = os.environ.get("PYAMG_REF_SCRATCH", "/tmp/pyamg_ref_scratch")


def _install_aliases():
    import numpy as np
    import numpy.testing
    import scipy as sp
    import scipy.linalg
    import scipy.sparse
    import scipy.sparse.linalg

    builtins.basestring = str

    class _Tester(object):
        def test(self, *a, **k):
            raise RuntimeError("nose tester not available")
        bench = test
    numpy.testing.Tester = _Tester

    for name, val in (("float", float), ("int", int), ("complex", complex),
                      ("bool", bool), ("object", object),
                      ("longfloat", np.longdouble)):
        if name not in np.__dict__:
            setattr(np, name, val)
    if "rank" not in np.__dict__:
        np.rank = np.ndim
    if "deprecate" not in np.__dict__:
        def _deprecate(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f
        np.deprecate = _deprecate
    if "find_common_type" not in np.__dict__:
        np.find_common_type = lambda a, s: np.result_type(*(list(a) + list(s)))

    # scipy used to re-export the numpy namespace (sp.zeros, sp.rand, ...)
    for name in dir(np):
        if name.startswith("_"):
            continue
        if name not in sp.__dict__:
            try:
                setattr(sp, name, getattr(np, name))
            except Exception:
                pass
    sp.rand = np.random.rand
    sp.randn = np.random.randn
    sp.random = np.random
    sp.mat = np.asmatrix
    sp.sparse = scipy.sparse
    sp.linalg = scipy.linalg

    if not hasattr(scipy.linalg, "pinv2"):
        scipy.linalg.pinv2 = scipy.linalg.pinv

    def _fake(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    def upcast(*args):
        return np.result_type(*args)

    def isscalarlike(x):
        return np.isscalar(x) or (hasattr(x, "ndim") and x.ndim == 0)

    m = _fake("scipy.sparse.sputils", upcast=upcast, isscalarlike=isscalarlike)
    scipy.sparse.sputils = m

    def make_system(A, M, x0, b, xtype=None):
        # scipy <= 1.x's isolve.utils.make_system (the scipy helper, not reference code): linear operators,
        # flat float vectors, identity preconditioner when M is None
        from scipy.sparse.linalg import LinearOperator, aslinearoperator
        A_ = aslinearoperator(A)
        b = np.asarray(b, dtype=np.result_type(A_.dtype, np.asarray(b).dtype, np.float64)).reshape(-1)
        x = np.zeros_like(b) if x0 is None else np.array(x0, dtype=b.dtype).reshape(-1)
        if M is None:
            M_ = LinearOperator(A_.shape, matvec=lambda v: v, rmatvec=lambda v: v, dtype=b.dtype)
        else:
            M_ = aslinearoperator(M)
        return A_, M_, x, b, (lambda v: v)
    iso = _fake("scipy.sparse.linalg.isolve")
    isu = _fake("scipy.sparse.linalg.isolve.utils", make_system=make_system)
    iso.utils = isu
    for nm in ("cg", "gmres", "bicgstab", "cgs", "qmr", "minres"):
        if hasattr(scipy.sparse.linalg, nm):
            setattr(iso, nm, getattr(scipy.sparse.linalg, nm))
    scipy.sparse.linalg.isolve = iso
    scipy.linalg.calc_lwork = _fake("scipy.linalg.calc_lwork")

    # sparse .H (conjugate transpose) was removed from scipy
    for cls in (scipy.sparse.csr_matrix, scipy.sparse.bsr_matrix,
                scipy.sparse.csc_matrix, scipy.sparse.coo_matrix):
        base = cls
        if not hasattr(base, "H"):
            base.H = property(lambda self: self.conj().transpose())


def stage(force=False):
    """Create the scratch copy and return the imported reference package."""
    if "pyamg" in sys.modules and getattr(sys.modules["pyamg"], "__file__", "").startswith(SCRATCH):
        return sys.modules["pyamg"]
    if not os.path.isdir(REFERENCE):
        raise RuntimeError("reference tree not present (this only runs in the dev container)")
    if not os.path.exists(REF_SO):
        raise RuntimeError("build oracle/_ref first: make -C oracle ref")
    pkg = os.path.join(SCRATCH, "pyamg")
    if force or not os.path.isdir(pkg):
        shutil.rmtree(SCRATCH, ignore_errors=True)
        shutil.copytree(os.path.join(REFERENCE, "pyamg"), pkg,
                        ignore=shutil.ignore_patterns("dev", "*.pyc", "__pycache__"))
        shutil.copy(REF_SO, os.path.join(pkg, "amg_core", "_amg_core.so"))
        with open(os.path.join(pkg, "version.py"), "w") as f:
            f.write("version='3.0.2'\ngit_revision='scratch'\nshort_version=version\n"
                    "full_version=version\nrelease=False\n")
        with open(os.path.join(pkg, "__config__.py"), "w") as f:
            f.write("def show():\n    pass\n")
        subprocess.run([sys.executable, "-W", "ignore", "-m", "lib2to3", "-f", "print",
                        "-f", "import", "-w", "-n", pkg],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    _install_aliases()
    if SCRATCH not in sys.path:
        sys.path.insert(0, SCRATCH)
    import warnings
    warnings.filterwarnings("ignore")
    return importlib.import_module("pyamg")


def poisson(grid):
    """d-D Poisson as a Kronecker sum (the reference's gallery.stencil_grid is
    broken on numpy >= 1.23); same CSR as gallery.poisson: last grid
    axis fastest, sorted int32 column indices."""
    import numpy as np
    import scipy.sparse as sps
    A = None
    for n in grid:
        T = sps.diags([-np.ones(n - 1), 2 * np.ones(n), -np.ones(n - 1)], [-1, 0, 1], format="csr")
        if A is None:
            A = T
        else:
            A = sps.kron(A, sps.identity(n, format="csr"), format="csr") + \
                sps.kron(sps.identity(A.shape[0], format="csr"), T, format="csr")
    A = sps.csr_matrix(A)
    A.sort_indices()
    A.indices = A.indices.astype(np.intc)
    A.indptr = A.indptr.astype(np.intc)
    return A
