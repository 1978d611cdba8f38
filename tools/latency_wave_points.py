"""Single-frame latency against CTAG_OPT_WAVE_POINTS (the boundary capacity above which a component gets a wave of its own in few-frame calls)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from ctag_testlib import read_bmp_gray, GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
syn = tk.synth_frame_host(state, 0)[0]
for wp in (96, 64, 48, 32, 24, 16, 8):
    det = tk.Detector(state, fs)
    det.set_option(capi.OPT_WAVE_POINTS, wp)
    out = []
    for img in (bmp, syn):
        for _ in range(5): r = det.detect(img)
        t0 = time.perf_counter(); n = 200
        for _ in range(n): r = det.detect(img)
        out.append((time.perf_counter() - t0) / n * 1e3)
    det.set_option(capi.OPT_TIMING, 1); det.detect(bmp); tm = det.timings(); det.set_option(capi.OPT_TIMING, 0)
    print("wave_points %3d: test.bmp %.3f ms, synthetic %.3f ms  quad_edges %.3f big %.3f welsch %.3f" % (wp, out[0], out[1], tm["quad_edges"], tm["quad_edges_big"], tm["welsch"]))
    det.close()
