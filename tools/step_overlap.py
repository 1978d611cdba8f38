#!/usr/bin/env python3
"""Experiment: consecutive steps (independent batches of `frames` frames) on ONE handle vs alternating between TWO handles (two streams,
two workspaces), so that the ramp-up / tail of a step's kernels overlaps the neighbouring step.  What a rank of an 8-GPU strong-
scaling job sees is 512-frame steps.  usage: python tools/step_overlap.py [frames] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
state, fs = ca.load_marker_file(os.path.join(ROOT, "tests", "golden", "CTag_2f12c.marker"))
rows, cols = 1080, 1920
dets = [tk.Detector(state, fs) for _ in range(3)]
frames = torch.empty((n, rows, cols), dtype=torch.uint8, device="cuda")
dets[0].synth_frames_device(frames.data_ptr(), 0, n, rows, cols, cols, rows * cols)
outs = [torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda") for _ in range(3)]
for d in dets:
    d.set_option(capi.OPT_MAX_CHUNK, n)
for ways in (1, 2, 3):
    def run(k):
        i = k % ways
        dets[i].detect_batch_device(frames.data_ptr(), n, rows, cols, cols, rows * cols, outs[i].data_ptr())
    for k in range(6):
        run(k)
    for d in dets:
        d.sync()
    t0 = time.perf_counter()
    for k in range(steps):
        run(k)
    for d in dets:
        d.sync()
    dt = time.perf_counter() - t0
    print("%d handle(s), %d-frame steps: %.3f ms per step = %.0f frames/s" % (ways, n, dt / steps * 1e3, n * steps / dt), flush=True)
