#!/usr/bin/env python3
"""Throughput of device-resident BGR batches (ctag_detect_batch_bgr8_device: cvtColor(BGR2GRAY) of main.cpp:36,52-54 on the device in front of the
detection chain) against the same frames handed over as gray: frames/s and the ratio VERDICT r4 item 6 asks for.  GPU box."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cylindertag_amd as ca  # noqa: E402
import testkit as tk  # noqa: E402
from cylindertag_amd import capi  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
cols, rows = (int(v) for v in sys.argv[2].lower().split("x")) if len(sys.argv) > 2 else (1920, 1080)  # tools/bgr_rate.py 512 3840x2160
state, fs = ca.load_marker_file(os.path.join(ROOT, "tests", "golden", "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
det.set_option(capi.OPT_MAX_CHUNK, n)
gray = torch.empty((n, rows, cols), dtype=torch.uint8, device="cuda")
det.synth_frames_device(gray.data_ptr(), 0, n, rows, cols, cols, rows * cols)
torch.cuda.synchronize()
bgr = torch.empty((n, rows, cols, 3), dtype=torch.uint8, device="cuda")
for c in range(3):  # gray-valued BGR: 1868 + 9617 + 4899 = 16384, so the converted image is the gray batch itself and the records must be equal
    bgr[..., c] = gray
torch.cuda.synchronize()
out_g = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda")
out_b = torch.zeros_like(out_g)


def run_gray():
    det.detect_batch_device(gray.data_ptr(), n, rows, cols, cols, rows * cols, out_g.data_ptr())


def run_bgr():
    det.detect_batch_bgr_device(bgr.data_ptr(), n, rows, cols, cols * 3, rows * cols * 3, out_b.data_ptr())


res = {}
for name, fn in (("gray", run_gray), ("bgr", run_bgr), ("gray", run_gray), ("bgr", run_bgr)):
    fn()
    det.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    det.sync()
    res.setdefault(name, []).append(n * 5 / (time.perf_counter() - t0))
det.set_option(capi.OPT_TIMING, 1)
for name, fn in (("gray", run_gray), ("bgr", run_bgr)):
    fn()
    fn()
    t = det.timings()
    print("%-4s per-kernel ms (one stream): " % name + " ".join("%s=%.3f" % (k, v) for k, v in t.items() if v > 0.05))
det.set_option(capi.OPT_TIMING, 0)
same = bool(torch.equal(out_g, out_b))
g, b = max(res["gray"]), max(res["bgr"])
print("device-resident %d frames of %dx%d: gray %.1f K frames/s, BGR %.1f K frames/s (ratio %.3f), records equal: %s" % (n, cols, rows, g / 1e3, b / 1e3, b / g, same))
