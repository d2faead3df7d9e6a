#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel stats of the TIMED solve region and the
timeline of one V-cycle (delimited by the residual-norm kernels)."""
import csv, sys, collections
path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
idx = [i for i, n in enumerate(names) if 'sum_sqrt_kernel' in n or 'sumsq_stage2' in n]
# the timed solve = the last (nsteps + 2) norm kernels before the SpMV timing loops: norm(b), r0, then nsteps
# find the run of nsteps+2 norms that ends last
end = idx[-1]
start = idx[-(nsteps + 1)]          # the initial-residual norm of the timed solve
sel = rows[start + 1:end + 1]
agg = collections.OrderedDict()
for r in sel:
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    k = (r['Kernel_Name'].replace('amg::', '').replace('(amg::StreamArgs, int, int)', '')[:48], r['Grid_Size_X'])
    a = agg.setdefault(k, [0, 0])
    a[0] += 1; a[1] += d
span = int(rows[end]['End_Timestamp']) - int(rows[start]['End_Timestamp'])
tot = sum(v[1] for v in agg.values())
print("timed region: %d kernels, span %.3f ms, sum of kernel time %.3f ms, per step %.3f ms" % (len(sel), span / 1e6, tot / 1e6, span / 1e6 / nsteps))
print("%-50s %10s %6s %10s %10s %6s" % ("kernel", "grid", "calls", "avg_us", "total_ms", "%"))
for (k, g), (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s %10s %6d %10.1f %10.3f %6.2f" % (k, g, c, d / c / 1e3, d / 1e6, 100.0 * d / tot))
