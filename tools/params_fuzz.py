#!/usr/bin/env python3
"""One-off differential hunt (GPU box): random ctag_params (every field inside its valid range) x frames, through ctag_create_ex on the GPU
and ctago_set_params on the oracle: records byte for byte, one frame per call and as a batch.  usage: python tools/params_fuzz.py [n_sets]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import Oracle, read_bmp_gray, read_marker_file, GOLDEN
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
orc = Oracle()
rng = np.random.RandomState(77)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bad = differs = 0
base = {}
for s in range(n):
    p = ca.default_params()
    p.threshold_line = float(rng.uniform(0.8, 3.5)); p.threshold_expand = float(rng.uniform(0.5, 2.5))
    p.threshold_RAC = float(rng.uniform(0.03, 0.6)); p.threshold_angle = float(rng.uniform(1.0, 12.0)); p.threshold_vertical = float(rng.uniform(0.1, 1.2))
    for i in range(4):
        p.ID_cr_correspond[i] = float([1.47, 1.54, 1.61, 1.68][i] + rng.uniform(-0.03, 0.03))
        p.cr_covariance_left[i] = float(rng.uniform(0.01, 0.08)); p.cr_covariance_right[i] = float(rng.uniform(0.01, 0.08))
    p.dark_cap = float(rng.uniform(0.08, 0.48)); p.area_min = int(rng.randint(5, 120)); p.area_max_fraction = float(rng.uniform(0.002, 0.06))
    p.collinear_cost = float(rng.uniform(1.01, 2.5))
    frames = [bmp, tk.synth_frame_host(state, 4000 + s)[0]]
    batch = np.stack([tk.synth_frame_host(state, 5000 + 6 * s + j)[0] for j in range(6)])
    det = tk.Detector(state, fs, params=p)
    orc.set_params(p)
    try:
        for i, f in enumerate(frames):
            want = orc.detect_fast(f, state, fs)
            if det.detect(f).tobytes() != want.tobytes():
                bad += 1; print("MISMATCH set", s, "frame", i, flush=True)
        got = det.detect_batch(batch)
        for j in range(len(batch)):
            if got[j].tobytes() != orc.detect_fast(batch[j], state, fs).tobytes():
                bad += 1; print("MISMATCH set", s, "batch frame", j, flush=True)
    finally:
        det.close(); orc.set_params(None)
print("param sets", n, "records", n * 8, "mismatches", bad)
sys.exit(1 if bad else 0)
