"""On-disk format of a multigrid hierarchy (the reference has none).

A directory:  meta.json  +  one .npy file per array (memory-mappable, no pickles):
    A<l>/P<l>/R<l>_{indptr,indices,data}.npy   level operators, CSR or BSR (block shape in meta.json; data is
                                               stored with its block dimensions), in their STORED order -- that
                                               order is the summation order of every operator application
    pre<l>/post<l>_<key>.npy                   array constants of the smoother descriptors (Dinv, coefficients,
                                               index lists, Schwarz subdomains and inverse blocks)
    coarse_dense.npy                           the dense coarse operator (pinv / lu / ... solvers), so that it is
                                               not recomputed (LAPACK may round differently from run to run)
Smoother constants (omega after the spectral-radius scaling, Chebyshev coefficients, inverse diagonal blocks)
are stored as computed: a loaded hierarchy re-estimates nothing and reproduces the saved one's iterates bit for
bit.  `pyamg_amd.distributed.save_levels / load_levels` ship the same files between the ranks of a node.
"""
import json
import os

import numpy as np
import scipy.sparse as sparse

FORMAT = "pyamg_amd-hierarchy-1"
_ARRAY_KEYS = ("Dinv", "coefficients", "indices", "subdomain", "subdomain_ptr", "inv_subblock", "inv_subblock_ptr")

__all__ = ["save_hierarchy", "load_hierarchy"]


def _put_matrix(path, tag, M):
    if sparse.isspmatrix_bsr(M):
        info = {"format": "bsr", "shape": [int(v) for v in M.shape], "blocksize": [int(v) for v in M.blocksize]}
    else:
        if not sparse.isspmatrix_csr(M):
            M = sparse.csr_matrix(M)
        info = {"format": "csr", "shape": [int(v) for v in M.shape]}
    np.save(os.path.join(path, tag + "_indptr.npy"), np.asarray(M.indptr))
    np.save(os.path.join(path, tag + "_indices.npy"), np.asarray(M.indices))
    np.save(os.path.join(path, tag + "_data.npy"), np.asarray(M.data))
    for attr in ("symmetry",):
        if hasattr(M, attr):
            info[attr] = getattr(M, attr)
    return info


def _get_matrix(path, tag, info, mmap):
    mode = "r" if mmap else None
    indptr = np.load(os.path.join(path, tag + "_indptr.npy"), mmap_mode=mode)
    indices = np.load(os.path.join(path, tag + "_indices.npy"), mmap_mode=mode)
    data = np.load(os.path.join(path, tag + "_data.npy"), mmap_mode=mode)
    shape = tuple(info["shape"])
    if info["format"] == "bsr":
        M = sparse.bsr_matrix((data, indices, indptr), shape=shape, blocksize=tuple(info["blocksize"]), copy=False)
    else:
        M = sparse.csr_matrix((data, indices, indptr), shape=shape, copy=False)
    if "symmetry" in info:
        M.symmetry = info["symmetry"]
    return M


def _put_descriptor(path, tag, desc):
    if desc is None:
        return None
    out = {}
    for k, v in desc.items():
        if k.startswith("_"):
            continue
        if k in _ARRAY_KEYS and v is not None:
            np.save(os.path.join(path, "%s_%s.npy" % (tag, k)), np.asarray(v))
            out[k] = {"array": True}
        elif isinstance(v, (np.floating, np.integer)):
            out[k] = v.item()
        else:
            out[k] = v
    return out


def _get_descriptor(path, tag, info):
    if info is None:
        return None
    desc = {}
    for k, v in info.items():
        if isinstance(v, dict) and v.get("array"):
            desc[k] = np.load(os.path.join(path, "%s_%s.npy" % (tag, k)))
        else:
            desc[k] = v
    return desc


def save_hierarchy(ml, path):
    """Write `ml` (a pyamg_amd.multilevel_solver) to the directory `path`."""
    os.makedirs(path, exist_ok=True)
    meta = {"format": FORMAT, "nlevels": len(ml.levels), "levels": []}
    for l, lvl in enumerate(ml.levels):
        m = {"A": _put_matrix(path, "A%d" % l, lvl.A)}
        if hasattr(lvl, "P"):
            m["P"] = _put_matrix(path, "P%d" % l, lvl.P)
            m["R"] = _put_matrix(path, "R%d" % l, lvl.R)
            for side, attr in (("pre", "presmoother"), ("post", "postsmoother")):
                fn = getattr(lvl, attr, None)
                desc = getattr(fn, "desc", None)
                if fn is not None and desc is None:
                    raise NotImplementedError("level %d: %s carries no descriptor and cannot be stored" % (l, attr))
                m[side] = _put_descriptor(path, "%s%d" % (side, l), desc)
        if getattr(lvl, "B", None) is not None:
            np.save(os.path.join(path, "B%d.npy" % l), np.asarray(lvl.B))
            m["B"] = True
        meta["levels"].append(m)
    cs = ml.coarse_solver
    spec = cs.spec
    kind, payload = cs.device_form(ml.levels[-1].A)
    cmeta = {"kind": kind}
    if kind == "dense":
        np.save(os.path.join(path, "coarse_dense.npy"), np.asarray(payload, dtype=np.float64))
        cmeta["name"] = cs.solver if isinstance(cs.solver, str) else "dense"
    elif kind == "smoother":
        cmeta["name"] = cs.solver
        cmeta["desc"] = _put_descriptor(path, "coarse", dict(payload.desc))
    else:
        cmeta["name"] = None
    meta["coarse_solver"] = cmeta
    with open(os.path.join(path, "meta.json"), "w") as f:
        json.dump(meta, f)
    return path


def load_hierarchy(path, mmap=False, device=0):
    """Read a directory written by save_hierarchy back into a pyamg_amd.multilevel_solver.  mmap=True maps the
    arrays instead of reading them (the device upload then streams them from the page cache)."""
    from . import smoothing
    from .multilevel import multilevel_solver
    with open(os.path.join(path, "meta.json")) as f:
        meta = json.load(f)
    if meta.get("format") != FORMAT:
        raise ValueError("%s is not a %s directory" % (path, FORMAT))
    levels = []
    for l, m in enumerate(meta["levels"]):
        lvl = multilevel_solver.level()
        lvl.A = _get_matrix(path, "A%d" % l, m["A"], mmap)
        if "P" in m:
            lvl.P = _get_matrix(path, "P%d" % l, m["P"], mmap)
            lvl.R = _get_matrix(path, "R%d" % l, m["R"], mmap)
        if m.get("B"):
            lvl.B = np.load(os.path.join(path, "B%d.npy" % l))
        levels.append(lvl)
    c = meta["coarse_solver"]
    if c["kind"] == "dense":
        M = np.load(os.path.join(path, "coarse_dense.npy"))
        ml = multilevel_solver(levels, coarse_solver=("dense", {"M": M}), device=device)
        ml.coarse_solver.solver_name = c.get("name")
    elif c["kind"] == "smoother":
        desc = _get_descriptor(path, "coarse", c["desc"])
        name, kw = smoothing.spec_from_descriptor(desc)
        ml = multilevel_solver(levels, coarse_solver=(name, kw), device=device)
    else:
        ml = multilevel_solver(levels, coarse_solver=None, device=device)
    for l, m in enumerate(meta["levels"]):
        if "P" not in m:
            continue
        for side, attr in (("pre", "presmoother"), ("post", "postsmoother")):
            desc = _get_descriptor(path, "%s%d" % (side, l), m.get(side))
            setattr(levels[l], attr, smoothing.smoother_from_descriptor(levels[l], desc))
    return ml
