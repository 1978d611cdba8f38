#!/usr/bin/env python3
"""One-frame call latency on content other than test.bmp: a 4K synthetic frame, the 1080p blob field (2666 blobs: through the any-frame workspace),
a frame without markers.  GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import GOLDEN
from clutter import blob_field
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
cases = {"4K synthetic": tk.synth_frame_host(state, 5, 2160, 3840)[0], "1080p blob field": blob_field(tk.synth_frame_host(state, 3)[0])[0],
         "1080p flat": np.full((1080, 1920), 180, np.uint8)}
for name, img in cases.items():
    try:
        for _ in range(3): det.detect(img)
    except ca.CtagError:
        pass
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        try:
            r = det.detect(img)
        except ca.CtagError as e:
            r = {"n_markers": -1}
        ts.append(time.perf_counter() - t0)
    ts.sort()
    print("%-18s median %.3f ms (markers %d)" % (name, ts[15] * 1e3, int(r["n_markers"])), flush=True)
