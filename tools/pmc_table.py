#!/usr/bin/env python3
"""Per-kernel mean of every counter in a `rocprofv3 --pmc ... --output-format csv` directory (counter_collection.csv).
usage: pmc_table.py DIR [name-filter]"""
import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"][:60]
    if flt and flt not in name: continue
    a = acc[name][r["Counter_Name"]]
    a[0] += 1; a[1] += float(r["Counter_Value"])
for name, cs in acc.items():
    print(name)
    for c, (n, v) in sorted(cs.items()):
        print("    %-28s %14.1f  (mean of %d dispatches)" % (c, v / n, n))
