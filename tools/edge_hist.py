#!/usr/bin/env python3
"""Histogram of the point counts of the edge clusters the Welsch fit receives (synthetic batch and test.bmp): which share of the
edges and of the point work falls into each length class.  GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402
import cylindertag_amd as ca  # noqa: E402
import testkit as tk  # noqa: E402
from ctag_testlib import GOLDEN, read_bmp_gray  # noqa: E402

state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
n = 64
frames = torch.empty((n, 1080, 1920), dtype=torch.uint8, device="cuda")
det.synth_frames_device(frames.data_ptr(), 0, n, 1080, 1920, 1920, 1080 * 1920)
out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda")
det.detect_batch_device(frames.data_ptr(), n, 1080, 1920, 1920, 1080 * 1920, out.data_ptr())
det.sync()
ns = np.concatenate([det.debug(f, tk.DBG_LINES) for f in range(n)])
print("synthetic: %d edges in %d frames (%.1f per frame), %d points (%.0f per frame)" % (len(ns), n, len(ns) / n, ns.sum(), ns.sum() / n))
edges = [0, 2, 10, 11, 12, 13, 16, 32, 48, 64, 96, 128, 192, 256, 100000]
for a, b in zip(edges[:-1], edges[1:]):
    m = (ns > a) & (ns <= b)
    print("  n in (%d, %d]: %5.1f %% of edges, %5.1f %% of points" % (a, b, 100.0 * m.mean(), 100.0 * ns[m].sum() / ns.sum()))
det.detect(read_bmp_gray(os.path.join(GOLDEN, "test.bmp")))
t = det.debug(0, tk.DBG_LINES)
print("test.bmp: %d edges, %d points, max %d; <=10: %d, <=16: %d, <=96: %d" % (len(t), t.sum(), t.max(), (t <= 10).sum(), (t <= 16).sum(), (t <= 96).sum()))
