#!/usr/bin/env python3
"""Mean per-launch PMC counter values of the learner kernels from a rocprofv3 --pmc output directory (counter_collection.csv)."""
import collections, csv, glob, sys

for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        if "mlp" in k or "adam" in k or "ppo" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}, "launches", len(next(iter(v.values()))))
