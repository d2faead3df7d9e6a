#!/usr/bin/env python3
"""The operators of the metric configuration's cycle launched a few times each for PMC collection: level-0 r = b - A x from
the coded stencil form (one-byte value codes, the default for constant coefficients) and from the 8-byte-value stencil
form, A_1 (sliced form, residual mode), R_0 (sliced form), P_0 (CSR stream kernel), with the calibration kernel of known
bytes (norm stage 1: 8 n read).  Prints the bytes per launch (algorithmic per SURVEY 8d, and what the storage form in use
streams) next to the launch times, one JSON object on the last line.
Usage:  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d OUT -- python3 tools/pmc_ops.py 500
Then:   python tools/pmc_ops.py --summary FETCH.csv WRITE.csv OPS.json OUT.json   (profiles/rNN_pmc_summary.json)"""
import collections, csv, ctypes as C, json, os, sys
import numpy as np


def summary(fetch, write, ops, out):
    """per kernel: mean FETCH_SIZE / WRITE_SIZE of its dispatches -> traffic = (2 FETCH + WRITE) KB (gfx950: FETCH_SIZE
    counts 64-byte units as 32, MI355X_MICROARCH.md; checked against sumsq_stage1 which reads 8 n bytes)"""
    def per_kernel(path, counter):
        # mean over the dispatches of a kernel's LARGEST grid: the same instantiation also runs on coarser levels and in the setup
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]][int(r["Grid_Size"])].append(float(r["Counter_Value"]))
        out = {}
        for k, by_grid in acc.items():
            v = by_grid[max(by_grid)]
            out[k] = sum(v) / len(v)
        return out
    F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    info = json.loads([l for l in open(ops) if l.startswith("{")][-1])
    entries = {}
    for key, e in info["ops"].items():
        kf = [k for k in F if all(s in k for s in e["kernel_match"])]
        if not kf or kf[0] not in W:
            continue
        f, w = F[kf[0]], W[kf[0]]
        short = kf[0].replace("void ", "").replace("(anonymous namespace)::", "").replace("amg::", "").split("(")[0]
        entries[key] = {"kernel_name": short,
                        "what": e["what"], "fetch_size_kb": round(f, 2), "write_size_kb": round(w, 2),
                        "traffic_bytes": round((2.0 * f + w) * 1024.0, 2), "bytes_of_the_form": e["moved_bytes"],
                        "algorithmic_bytes": e["algorithmic_bytes"], "ms_per_launch_when_collected": e["ms"]}
    # calibration: the norm's first stage always launches the same grid; its largest reading is the level-0 vector (8 n bytes)
    cal = [max(float(r["Counter_Value"]) for r in csv.DictReader(open(fetch)) if r["Counter_Name"] == "FETCH_SIZE" and "sumsq_stage1" in r["Kernel_Name"])]
    s = {"grid": info["grid"], "fetch_correction": 2.0, "entries": entries,
         "note": "traffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B per launch; FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated in the "
                 "same pass: sumsq_stage1 reads %d B and reports %.2f KB); the counter includes Infinity-Cache hits, so this bounds HBM "
                 "traffic from above" % (8 * info["n"], cal[0] if cal else -1.0),
         "source": "%s, %s (tools/pmc_ops.py)" % (fetch, write)}
    json.dump(s, open(out, "w"), indent=1)
    print(json.dumps(s, indent=1))


if len(sys.argv) > 1 and sys.argv[1] == "--summary":
    summary(*sys.argv[2:6])
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pyamg_amd import _lib
from pyamg_amd.aggregation import poisson, smoothed_aggregation_solver
g = int(sys.argv[1]) if len(sys.argv) > 1 else 500
np.random.seed(0)
sm = ("chebyshev", {"degree": 2})
ml = smoothed_aggregation_solver(poisson((g, g, g)), presmoother=sm, postsmoother=sm)
dev = ml.device_hierarchy()
L = _lib.lib()
A0, A1, P0, R0 = ml.levels[0].A, ml.levels[1].A, ml.levels[0].P, ml.levels[0].R
spmv = lambda M: 12.0 * M.nnz + 4.0 * (M.shape[0] + 1) + 8.0 * M.shape[1] + 8.0 * M.shape[0]
ops = collections.OrderedDict()
nd = L.amg_hier_value_index(dev.h, 0, -1)
if nd > 0:
    ops["level0_coded"] = {"what": "level-0 r = b - A x, stencil form with one-byte value codes (%d distinct values)" % nd,
                           "kernel_match": ["stencil_coded_kernel<2>"], "ms": dev.time_spmv(0, 0, mode=1, reps=4),
                           "algorithmic_bytes": spmv(A0) + 8.0 * A0.shape[0], "moved_bytes": L.amg_hier_operator_bytes(dev.h, 0, 1)}
    L.amg_hier_value_index(dev.h, 0, 0)
ops["level0_values"] = {"what": "level-0 r = b - A x, stencil form with 8-byte values", "kernel_match": ["stencil2_kernel<2, 8>"],
                        "ms": dev.time_spmv(0, 0, mode=1, reps=4), "algorithmic_bytes": spmv(A0) + 8.0 * A0.shape[0],
                        "moved_bytes": L.amg_hier_operator_bytes(dev.h, 0, 1)}
if nd > 0:
    L.amg_hier_value_index(dev.h, 0, 1)
ops["level1_residual"] = {"what": "level-1 r = b - A x, sliced form (SELL-64, 16-bit column codes)", "kernel_match": ["sell_kernel<2"],
                          "ms": dev.time_spmv(1, 0, mode=1, reps=4), "algorithmic_bytes": spmv(A1) + 8.0 * A1.shape[0],
                          "moved_bytes": L.amg_hier_operator_bytes(dev.h, 1, 1)}
ops["R_0"] = {"what": "R_0 matvec, sliced form", "kernel_match": ["sell_kernel<0"], "ms": dev.time_spmv(0, 2, mode=0, reps=4),
              "algorithmic_bytes": spmv(R0), "moved_bytes": None}
ops["P_0"] = {"what": "P_0 matvec, CSR stream kernel", "kernel_match": ["csr_stream_kernel<0, 1>"], "ms": dev.time_spmv(0, 1, mode=0, reps=4),
              "algorithmic_bytes": spmv(P0), "moved_bytes": None}
n = A0.shape[0]
b = np.random.rand(n); x = np.zeros(n); res = np.zeros(4); nres = C.c_int()
_lib.check(L.amg_hier_solve(dev.h, b.ctypes.data, x.ctypes.data, 0.0, 0, 0, _lib.dp(res), C.byref(nres), 1))    # calibration: norms read 8 n
for k, e in ops.items():
    print("%-18s %.4f ms per launch, algorithmic %.4f GB -> %.0f GB/s%s" % (k, e["ms"], e["algorithmic_bytes"] / 1e9, e["algorithmic_bytes"] / e["ms"] / 1e6,
          "" if not e["moved_bytes"] else "; the form streams %.4f GB -> %.0f GB/s" % (e["moved_bytes"] / 1e9, e["moved_bytes"] / e["ms"] / 1e6)))
print("shapes: A_1 %s nnz %d, R_0 %s nnz %d, P_0 %s nnz %d; calibration n = %d" % (A1.shape, A1.nnz, R0.shape, R0.nnz, P0.shape, P0.nnz, n))
print(json.dumps({"grid": g, "n": n, "ops": ops}))
