import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        per[(r["Dispatch_Id"], k, r["Counter_Name"])] += float(r["Counter_Value"])
    for (_, k, c), v in per.items():
        acc[k][c].append(v)
for k in sorted(acc):
    if any(p in k for p in sys.argv[2:]):
        print(k, {c: round(sum(v) / len(v)) for c, v in acc[k].items()})
