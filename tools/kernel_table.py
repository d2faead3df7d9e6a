#!/usr/bin/env python3
"""Per-kernel table (calls, total, average; by kernel name and by name + grid) of a `rocprofv3 --kernel-trace
--output-format csv` directory.  usage: kernel_table.py DIR [rows=14]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 14
by_name, by_grid = collections.defaultdict(lambda: [0, 0.0]), collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    name = r["Kernel_Name"][:64]
    for d, k in ((by_name, name), (by_grid, (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")))):
        d[k][0] += 1; d[k][1] += us
print("sum of kernel time %.1f us" % sum(v[1] for v in by_name.values()))
for d in (by_name, by_grid):
    for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:top]:
        print("%-90s calls %7d  total %10.1f us  avg %8.2f us" % (str(k), v[0], v[1], v[1] / v[0]))
    print()
