"""Load tests/golden/*.npz (fixtures captured from the reference by oracle/gen_golden.py)."""
import glob
import json
import os

import numpy as np
import scipy.sparse as sps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _all_cases():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "hier_*.npz")))


def hier_cases():
    """hierarchies solved by the stand-alone cycle iteration"""
    return [c for c in _all_cases() if not c.startswith(("accel_", "krylov_"))]


def krylov_smoother_cases():
    """hierarchies whose smoothers are Krylov iterations (smoothing.py:481-509): no oracle form, pinned by the
    reference's own histories"""
    return [c for c in _all_cases() if c.startswith("krylov_")]


def accel_cases(method=None):
    """hierarchies solved by the reference with Krylov acceleration (oracle/gen_golden_r2.py); the method is
    the second word of the name: accel_<method>_..."""
    return [c for c in _all_cases() if c.startswith("accel_") and (method is None or c.split("_")[1] == method)]


def get_mat(z, key):
    bs = tuple(int(v) for v in z[key + "_bs"])
    shape = tuple(int(v) for v in z[key + "_shape"])
    indptr, indices, data = z[key + "_indptr"], z[key + "_indices"], z[key + "_data"]
    if bs == (0, 0):
        return sps.csr_matrix((data, indices, indptr), shape=shape)
    return sps.bsr_matrix((data.reshape(-1, bs[0], bs[1]), indices, indptr), shape=shape)


def canonical(d):
    """Reduce a recorded closure description to the kernel family it calls
    (the same reductions pyamg/relaxation/smoothing.py:366-449 makes)."""
    name = d.get("name")
    if name in (None, "None"):
        return {"name": None}
    if name == "richardson":          # smoothing.py:422-428 -> polynomial([omega])
        return {"name": "polynomial", "coefficients": [d["omega"]],
                "iterations": d.get("iterations", 1)}
    if name == "chebyshev":           # smoothing.py:438-449
        return {"name": "polynomial", "coefficients": d["coefficients"],
                "iterations": d.get("iterations", 1)}
    if name in ("block_jacobi", "block_gauss_seidel") and d.get("Dinv") is None:
        d = dict(d)                   # blocksize 1 -> point smoother (smoothing.py:377-380,406-408)
        d["name"] = name[len("block_"):]
    return d


def load_hier(name):
    """-> dict(meta, levels=[{A,P,R,pre,post}], coarse_pinv, b, x0, x, residuals, x_iter1, x_iter2)"""
    z = np.load(os.path.join(GOLDEN, "hier_%s.npz" % name), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    levels = []
    for i in range(meta["nlevels"]):
        L = {"A": get_mat(z, "A%d" % i)}
        if i < meta["nlevels"] - 1:
            L["P"] = get_mat(z, "P%d" % i)
            L["R"] = get_mat(z, "R%d" % i)
            for side in ("pre", "post"):
                d = dict(meta["levels"][i][side])
                if d.pop("has_Dinv", False):
                    d["Dinv"] = z["%s%d_Dinv" % (side, i)]
                if d.pop("has_schwarz", False):
                    for k in ("subdomain", "subdomain_ptr", "inv_subblock", "inv_subblock_ptr"):
                        d[k] = z["%s%d_%s" % (side, i, k)]
                L[side] = canonical(d)
        levels.append(L)
    out = dict(meta=meta, levels=levels, coarse_pinv=z["coarse_pinv"])
    for k in ("b", "x0", "x", "residuals", "x_iter1", "x_iter2"):
        out[k] = z[k]
    if "B0" in z.files:
        out["B0"] = z["B0"]
    return out


def load_kernels():
    z = np.load(os.path.join(GOLDEN, "kernels.npz"), allow_pickle=False)
    cases = {}
    for name in z["cases"]:
        name = str(name)
        pre = name + "__"
        cases[name] = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    return cases


def evaluation_floor(A, x, b):
    """eps * (||b||_2 + ||A||_inf ||x||_2): the size of the rounding error any fp64 evaluation of b - A x carries,
    whatever the summation order"""
    eps = np.finfo(np.float64).eps
    Ainf = abs(sps.csr_matrix(A)).sum(axis=1).max()
    return eps * (np.linalg.norm(b) + Ainf * np.linalg.norm(x))


FLOOR_FACTOR = 0.25


def history_tolerance(A, x, b, ref):
    """Per-entry tolerance for comparing a residual-norm history with the reference's.

    north_star: 1e-12 relative.  The reference's coarse solve and norms go through BLAS, whose summation order
    is unspecified, so its iterates agree with a sequential-order evaluation only to rounding, and a residual
    b - A x_k evaluated from such iterates differs by a fraction of evaluation_floor() in ABSOLUTE terms
    however small r_k has become.  Measured over the 20 reference-built hierarchies (oracle vs reference):
    worst |r_k - ref_k| = 0.052 floors.  Hence
        tol_k = 1e-12 * ref_k + 0.25 * floor,
    five times the worst case seen (round 1 used 100 floors).  Entries above 1e12 floors are additionally held
    to the pure relative bound by assert_history()."""
    return 1e-12 * np.asarray(ref) + FLOOR_FACTOR * evaluation_floor(A, x, b)


def assert_history(res, ref, A, x, b):
    """the comparison every history test makes: entry-wise tolerance above, and 1e-12 RELATIVE alone wherever
    the residual is large enough for a relative statement to mean something (r_k > 1e12 floors)"""
    res, ref = np.asarray(res, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert len(res) == len(ref), "history length %d, reference %d" % (len(res), len(ref))
    tol = history_tolerance(A, x, b, ref)
    d = np.abs(res - ref)
    assert np.all(d <= tol), "worst |res - ref| / tol = %g" % np.max(d / tol)
    big = ref > 1e12 * evaluation_floor(A, x, b)
    assert np.all(d[big] <= 1e-12 * ref[big]), "relative deviation %g on a large residual" % np.max(d[big] / ref[big])
    return tol


def smoother_spec(d):
    """canonical recorded descriptor -> the ('name', {opts}) tuple change_smoothers takes,
    with the reference's constants passed explicitly (never re-estimated)."""
    name = d.get("name")
    if name is None:
        return None
    it = int(d.get("iterations", 1))
    if name == "jacobi":
        return ("jacobi", {"iterations": it, "omega": d["omega"], "withrho": False})
    if name == "gauss_seidel":
        return ("gauss_seidel", {"iterations": it, "sweep": d.get("sweep", "forward")})
    if name == "sor":
        return ("sor", {"iterations": it, "omega": d["omega"], "sweep": d.get("sweep", "forward")})
    if name == "polynomial":
        return ("polynomial", {"iterations": it, "coefficients": d["coefficients"]})
    bs = int(d.get("blocksize", 1))
    if name == "block_jacobi":
        return ("block_jacobi", {"iterations": it, "omega": d["omega"], "withrho": False, "blocksize": bs,
                                 "Dinv": np.asarray(d["Dinv"]).reshape(-1, bs, bs)})
    if name == "block_gauss_seidel":
        return ("block_gauss_seidel", {"iterations": it, "sweep": d.get("sweep", "forward"), "blocksize": bs,
                                       "Dinv": np.asarray(d["Dinv"]).reshape(-1, bs, bs)})
    if name == "gauss_seidel_ne" or name == "gauss_seidel_nr":
        return (name, {"iterations": it, "sweep": d.get("sweep", "forward"), "omega": d.get("omega", 1.0)})
    if name == "jacobi_ne":
        return ("jacobi_ne", {"iterations": it, "omega": d["omega"], "withrho": False})
    if name == "krylov":
        kw = {"tol": d["tol"], "maxiter": d["maxiter"]}
        if d["method"] == "gmres":
            kw["restrt"] = d.get("restrt")
        return (d["method"], kw)
    if name == "schwarz":
        return ("schwarz", {"iterations": it, "sweep": d.get("sweep", "symmetric"), "subdomain": d["subdomain"],
                            "subdomain_ptr": d["subdomain_ptr"], "inv_subblock": d["inv_subblock"],
                            "inv_subblock_ptr": d["inv_subblock_ptr"]})
    raise KeyError(name)


def build_ml(g):
    """pyamg_amd.multilevel_solver from a golden hierarchy (reference operators + constants)."""
    import pyamg_amd
    levels = []
    for L in g["levels"]:
        lvl = pyamg_amd.multilevel_solver.level()
        lvl.A = L["A"]
        if "P" in L:
            lvl.P, lvl.R = L["P"], L["R"]
        levels.append(lvl)
    ml = pyamg_amd.multilevel_solver(levels, coarse_solver=("dense", {"M": g["coarse_pinv"]}))
    pre = [smoother_spec(L["pre"]) for L in g["levels"][:-1]]
    post = [smoother_spec(L["post"]) for L in g["levels"][:-1]]
    pyamg_amd.change_smoothers(ml, pre, post)
    return ml
