#!/usr/bin/env python3
"""Side measurement (GPU box): throughput on camera content -- the 64-frame sequence derived from the reference's test.bmp
(cropped to 1920x1080, 5 physical markers in view), tiled to n frames resident in HBM.  Prints frames/s and the stage times."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from ctag_testlib import read_bmp_gray, read_marker_file, GOLDEN
from sequences import avi_substitute
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
seq = avi_substitute(read_bmp_gray(os.path.join(GOLDEN, "test.bmp")))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rows, cols = seq.shape[1:]
frames = torch.from_numpy(np.concatenate([seq] * (n // len(seq)))).cuda()
n = frames.shape[0]
det = tk.Detector(state, fs); det.set_option(capi.OPT_MAX_CHUNK, n)
out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda")
run = lambda: det.detect_batch_device(frames.data_ptr(), n, rows, cols, cols, rows * cols, out.data_ptr(), 5, True, 5)
run(); det.sync()
det.set_option(capi.OPT_TIMING, 1)
t0 = time.perf_counter()
for _ in range(3):
    run()
det.sync()
dt = (time.perf_counter() - t0) / 3
res = np.frombuffer(out.cpu().numpy().tobytes(), dtype=ca.RESULT_DT)
print("frames", n, "%dx%d" % (cols, rows), "frames/s %.0f" % (n / dt), "markers/frame %.2f" % res["n_markers"].mean(),
      {k: round(v, 2) for k, v in det.timings().items()})
