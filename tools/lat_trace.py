#!/usr/bin/env python3
"""Timeline of one ctag_detect_u8 call on test.bmp from a rocprofv3 kernel trace: per kernel the median start offset within the call, duration
and the gap to the previous kernel's end.
  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lt -- python3 tools/lat_trace.py run
  analyse:  python3 tools/lat_trace.py gpurun_out/lt/.../*_kernel_trace.csv"""
import csv, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "run":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cylindertag_amd as ca
    import testkit as tk
    from ctag_testlib import read_bmp_gray, GOLDEN
    state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    det = tk.Detector(state, fs)
    img = read_bmp_gray(os.path.join(GOLDEN, "test.bmp")) if len(sys.argv) < 3 else tk.synth_frame_host(state, 0)[0]
    for _ in range(60): det.detect(img)
    sys.exit(0)
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "ctag::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
calls, cur = [], []
for r in rows:  # a call starts at its decimation kernel
    if "k_decimate" in r["Kernel_Name"] and cur:
        calls.append(cur); cur = []
    cur.append(r)
calls.append(cur)
calls = calls[10:]  # warm-up
n = statistics.mode(len(c) for c in calls)
calls = [c for c in calls if len(c) == n]
print("%d calls of %d kernels" % (len(calls), n))
tot = []
for k in range(n):
    name = calls[0][k]["Kernel_Name"].replace("ctag::", "").split("(")[0][:60]
    st = statistics.median(int(c[k]["Start_Timestamp"]) - int(c[0]["Start_Timestamp"]) for c in calls) / 1e3
    du = statistics.median(int(c[k]["End_Timestamp"]) - int(c[k]["Start_Timestamp"]) for c in calls) / 1e3
    gap = statistics.median(int(c[k]["Start_Timestamp"]) - max(int(x["End_Timestamp"]) for x in c[:k]) for c in calls) / 1e3 if k else 0.0
    print("%-62s start %7.1f us  dur %6.1f  gap after the latest earlier end %6.1f" % (name, st, du, gap))
span = statistics.median(max(int(x["End_Timestamp"]) for x in c) - int(c[0]["Start_Timestamp"]) for c in calls) / 1e3
period = statistics.median(int(b[0]["Start_Timestamp"]) - int(a[0]["Start_Timestamp"]) for a, b in zip(calls, calls[1:])) / 1e3
print("first kernel start -> last kernel end %.1f us; call period %.1f us" % (span, period))
