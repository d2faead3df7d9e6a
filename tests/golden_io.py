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
                L[side] = canonical(d)
        levels.append(L)
    out = dict(meta=meta, levels=levels, coarse_pinv=z["coarse_pinv"])
    for k in ("b", "x0", "x", "residuals", "x_iter1", "x_iter2"):
        out[k] = z[k]
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
