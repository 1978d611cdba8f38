#!/usr/bin/env python3
"""Bounding-box shapes of the candidate components (synthetic batch): how many of the boundary kernel's 64-column lane groups a
component's rows actually fill.  GPU box."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
n = 32
frames = torch.empty((n, 1080, 1920), dtype=torch.uint8, device="cuda")
det.synth_frames_device(frames.data_ptr(), 0, n, 1080, 1920, 1920, 1080 * 1920)
out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda")
det.detect_batch_device(frames.data_ptr(), n, 1080, 1920, 1920, 1080 * 1920, out.data_ptr())
det.sync()
c = np.concatenate([det.debug(f, tk.DBG_CANDIDATES) for f in range(n)])
w = c[:, 3] - c[:, 1] + 1
h = c[:, 4] - c[:, 2] + 1
nb = c[:, 6]
print("candidates per frame %.1f; w mean %.1f median %d; h mean %.1f median %d; boundary points mean %.1f" % (len(c) / n, w.mean(), np.median(w), h.mean(), np.median(h), nb.mean()))
for lim in (8, 16, 24, 32, 48, 64, 128):
    print("  w <= %3d: %5.1f %% of components, %5.1f %% of the rows scanned" % (lim, 100.0 * (w <= lim).mean(), 100.0 * h[w <= lim].sum() / h.sum()))
print("  rows scanned per frame: %.0f; pixels in boxes per frame: %.0f" % (h.sum() / n, (w * h).sum() / n))
