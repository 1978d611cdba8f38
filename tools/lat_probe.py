#!/usr/bin/env python3
"""One-frame latency probe: ms per ctag_detect_u8 on test.bmp and synthetic frame 0 with per-kernel HIP-event times; honours the
CTAG_* developer knobs of the environment (run once per setting: the library reads them once per process)."""
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
if os.environ.get("LAT_GRAPH"):  # replay the chain as a hipGraph (CTAG_OPT_GRAPH)
    det.set_option(capi.OPT_GRAPH, 1)
imgs = {"bmp": read_bmp_gray(os.path.join(GOLDEN, "test.bmp")), "syn": tk.synth_frame_host(state, 0)[0]}
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for name, img in imgs.items():
    for _ in range(5): det.detect(img)
    t0 = time.perf_counter()
    for _ in range(n): det.detect(img)
    dt = (time.perf_counter() - t0) / max(n, 1)
    det.set_option(capi.OPT_TIMING, 1); det.detect(img); tm = det.timings(); det.set_option(capi.OPT_TIMING, 0)
    print("%s %.3f ms/call kernels %.3f %s" % (name, dt * 1e3, sum(tm.values()), " ".join("%s=%.3f" % (k[:9], v) for k, v in tm.items() if v >= 0.02)), flush=True)
