"""Load tests/golden/*.npz (fixtures captured from the reference by oracle/gen_golden.py)."""
import glob
import json
import os

import numpy as np
import scipy.sparse as sps

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def hier_cases():
    return sorted(os.path.basename(p)[5:-4] for p in glob.glob(os.path.join(GOLDEN, "hier_*.npz")))


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


def history_tolerance(A, x, b, ref):
    """Per-entry tolerance for comparing residual-norm histories.

    north_star: 1e-12 relative.  A residual b - A*x evaluated in fp64 carries an
    absolute rounding floor of ~eps*(|b| + |A||x|) whatever the summation order
    (the reference's coarse solve and norm go through BLAS, order unspecified),
    so once r_k approaches that floor a purely relative bound is meaningless:
    tol_k = 1e-12 * r_k + 100*eps*(||b||_2 + ||A||_inf ||x||_2).
    """
    eps = np.finfo(np.float64).eps
    Ainf = abs(sps.csr_matrix(A)).sum(axis=1).max()
    floor = 100 * eps * (np.linalg.norm(b) + Ainf * np.linalg.norm(x))
    return 1e-12 * np.asarray(ref) + floor


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
