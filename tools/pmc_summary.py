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
keep = [k for k in sorted(acc) if k.startswith(("ctag::", "void ctag::"))]
for c in ctrs:
    print(c)
    for k in keep:
        v = acc[k].get(c)
        if v: print("    %-44s %14.5g" % (k.replace("void ", "")[:44], sum(v) / len(v)))
