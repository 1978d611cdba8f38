#!/usr/bin/env python3
"""One-off differential hunt (GPU box): random frames x random detect() parameters, GPU record vs oracle record, byte for byte.
usage: python tools/param_sweep.py [n_cases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import Oracle, read_bmp_gray, read_marker_file, GOLDEN
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
orc, det = Oracle(), tk.Detector(state, fs)
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)  # usage: param_sweep.py [n_cases] [seed]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(n):
    kind = rng.randint(0, 3)
    if kind == 0:
        img = tk.synth_frame_host(state, 2000 + case)[0]
    elif kind == 1:
        y0, x0 = rng.randint(0, 100), rng.randint(0, 200)
        img = np.ascontiguousarray(bmp[y0:y0 + rng.randint(500, 1100), x0:x0 + rng.randint(700, 1700)])
    else:
        img = tk.synth_frame_host(state, 3000 + case)[0]
        h, w = rng.randint(300, 1081), rng.randint(400, 1921)  # any size, odd ones included
        img = np.ascontiguousarray(img[:h, :w])
    tw = int(rng.choice([3, 4, 5, 5, 5, 6, 7, 9, 12]))
    subpix = bool(rng.rand() < 0.8)
    dist = int(rng.choice([1, 2, 3, 5, 5, 8, 12, 16, 20]))
    want = orc.detect_fast(img, state, fs, tw, subpix, dist)
    try:
        got = det.detect(img, tw, subpix, dist)
        same = got.tobytes() == want.tobytes()
    except ca.CtagError as e:
        same = (e.status == want["status"])
    if not same:
        bad += 1
        print("MISMATCH case", case, img.shape, "tw", tw, "subpix", subpix, "dist", dist, "status", want["status"])
print("cases", n, "mismatches", bad)
sys.exit(1 if bad else 0)
