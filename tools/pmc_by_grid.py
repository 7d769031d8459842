#!/usr/bin/env python3
"""Mean per-launch PMC values per (kernel, grid) from a rocprofv3 --pmc output directory: tools/pmc_by_grid.py <dir> [name filter]"""
import collections, csv, glob, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if flt not in r["Kernel_Name"]:
            continue
        k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k, {c: round(sum(x) / len(x), 1) for c, x in sorted(v.items())}, "launches", len(next(iter(v.values()))))
