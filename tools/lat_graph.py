import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca, testkit as tk
from cylindertag_amd import capi
from ctag_testlib import read_bmp_gray, GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
img = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
for mode in (2, 0, 2, 0):
    det = tk.Detector(state, fs); det.set_option(capi.OPT_GRAPH, mode)
    for _ in range(20): det.detect(img)
    ts=[]
    for _ in range(300):
        t0=time.perf_counter(); det.detect(img); ts.append(time.perf_counter()-t0)
    ts.sort(); print("OPT_GRAPH", mode, "median %.4f p10 %.4f p90 %.4f ms" % (ts[150]*1e3, ts[30]*1e3, ts[270]*1e3)); det.close()
