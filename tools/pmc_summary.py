#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel: tools/pmc_summary.py <dir> [<dir> ...] -> table kernel x counter (mean per launch)."""
import csv, glob, os, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        per = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            per[(r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][:48], r["Counter_Name"])] += float(r["Counter_Value"])
        for (_, k, c), v in per.items():
            acc[k][c].append(v)
ctrs = sorted({c for k in acc for c in acc[k]})
print("%-50s" % "kernel" + "".join("%18s" % c[:17] for c in ctrs))
for k in sorted(acc):
    print("%-50s" % k + "".join("%18.4g" % (sum(acc[k][c]) / max(1, len(acc[k][c]))) for c in ctrs))
