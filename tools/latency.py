#!/usr/bin/env python3
"""Single-frame latency of ctag_detect_u8 (host frame in, host record out) -- what CylinderTag::detect() costs per call."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from ctag_testlib import read_bmp_gray, GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
syn = tk.synth_frame_host(state, 0)[0]
model = ca.Model(os.path.join(GOLDEN, "CTag_2f12c.model")); cam = ca.load_camera(os.path.join(GOLDEN, "cameraParams.yml"))
for name, img in (("test.bmp 1920x1200", bmp), ("synthetic 1920x1080", syn)):
    for _ in range(5): r = det.detect(img)
    t0 = time.perf_counter(); n = 200
    for _ in range(n): r = det.detect(img)
    dt = (time.perf_counter() - t0) / n
    det.set_option(capi.OPT_TIMING, 1); det.detect(img); tm = det.timings(); det.set_option(capi.OPT_TIMING, 0)
    print("%s: %.3f ms per detect() call (%d markers); kernel time %.3f ms %s" % (name, dt * 1e3, r["n_markers"], sum(tm.values()), {k: round(v, 3) for k, v in tm.items()}))
r = det.detect(bmp)
for _ in range(5): det.estimate_pose(r, model, cam)
t0 = time.perf_counter()
for _ in range(200): det.estimate_pose(r, model, cam)
print("estimate_pose (5 markers): %.3f ms per call" % ((time.perf_counter() - t0) / 200 * 1e3))
