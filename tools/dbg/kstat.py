import csv, glob, sys
pat = sys.argv[2:] or ["silhouette", "quad_edges_packed<8"]
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(p in r["Name"] for p in pat):
            print("   %-90s calls %4s avg %9.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
