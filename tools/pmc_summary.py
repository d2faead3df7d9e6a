#!/usr/bin/env python3
"""profiles/rNN_pmc_summary.json (what bench.py reports as roofline.traffic, labelled `from_file`) from the two
rocprofv3 --pmc passes over tools/pmc_spmv.py: per launch of the level-0 kernel of each storage form,
traffic = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE doubled per MI355X_MICROARCH.md, checked in the same
pass against sumsq_stage1, which reads 8 n bytes).
Usage: python tools/pmc_summary.py FETCH.csv WRITE.csv GRID OUT.json [BENCH.json: takes roofline.bytes_per_launch from it]"""
import csv, json, sys, collections


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def pick(d, *needles):
    for k, v in d.items():
        if all(s in k for s in needles):
            return k, v
    return None, None


fetch, write, grid, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
alg = None
if len(sys.argv) > 5:
    for line in open(sys.argv[5]):
        if line.startswith("{"):
            alg = json.loads(line)["roofline"]["bytes_per_launch"]
F, W = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
n = grid ** 3
nnz = 7 * n - 6 * grid * grid
forms = {}
for label, needles in (("stencil", ("stencil", "_kernel<2,")), ("csr_stream_kernel<2,1>", ("csr_stream_kernel<2, 1>",)),
                       ("csr_pattern_kernel<2>", ("csr_pattern_kernel<2>",))):
    kf, f = pick(F, *needles)
    kw, w = pick(W, *needles)
    if f is not None and w is not None:
        forms[label] = {"kernel_name": kf.split("(")[0].replace("void amg::", ""), "fetch_size_kb": round(f, 2), "write_size_kb": round(w, 2),
                        "traffic_bytes": round((2.0 * f + w) * 1024.0, 2)}
kc, cal = pick(F, "sumsq_stage1")
main = forms.pop("stencil")
summary = {"grid": grid, "form": 2,
           "kernel": main["kernel_name"] + " = level-0 r = b - A x in the stencil form (plane-periodic block->XCD mapping, the default)",
           "fetch_size_kb": main["fetch_size_kb"], "write_size_kb": main["write_size_kb"], "fetch_correction": 2.0,
           "traffic_bytes": main["traffic_bytes"],
           "algorithmic_bytes_of_this_form": alg if alg is not None else 8.0 * 7 * n + 1.0 * n + 24.0 * n,
           "other_forms": forms,
           "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (calibrated in the same pass: sumsq_stage1 reads %d B and reports %.2f KB); "
                   "the counter includes Infinity-Cache hits, so this bounds HBM traffic from above" % (8 * n, cal or -1.0),
           "source": "%s, %s" % (fetch, write)}
json.dump(summary, open(out, "w"), indent=1)
print(json.dumps(summary, indent=1))
