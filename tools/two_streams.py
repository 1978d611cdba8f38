#!/usr/bin/env python3
"""Experiment: does running two independent half-batches on two handles (two HIP streams, two workspaces) beat one full batch on one?
The compute kernels sit at 24-42 % of the SIMDs' issue roof (latency-bound at 2-5 waves per SIMD), so kernels of another
chunk could fill idle issue slots and kernel tails.  usage: python tools/two_streams.py [frames] [ways]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ways_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4]
state, fs = ca.load_marker_file(os.path.join(ROOT, "tests", "golden", "CTag_2f12c.marker"))
rows, cols = 1080, 1920
frames = torch.empty((n, rows, cols), dtype=torch.uint8, device="cuda")
d0 = tk.Detector(state, fs)
d0.synth_frames_device(frames.data_ptr(), 0, n, rows, cols, cols, rows * cols)
out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda")
ref = None
for ways in ways_list:
    dets = [tk.Detector(state, fs) for _ in range(ways)]
    m = n // ways
    for d in dets:
        d.set_option(capi.OPT_MAX_CHUNK, m)
    def step():
        for i, d in enumerate(dets):
            d.detect_batch_device(frames[i * m].data_ptr(), m, rows, cols, cols, rows * cols, out[i * m].data_ptr())
        for d in dets:
            d.sync()
    step(); step()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        step()
    dt = (time.perf_counter() - t0) / reps
    h = out.cpu().numpy().tobytes()
    ref = ref or h
    print("%d stream(s) x %d frames: %.2f ms per %d frames = %.0f frames/s, records %s" % (ways, m, dt * 1e3, n, n / dt, "identical" if h == ref else "DIFFER"), flush=True)
    for d in dets:
        d.close()
