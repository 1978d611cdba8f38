#!/usr/bin/env python3
"""Determinism stress (GPU box): the same batch many times, and the same frames one per call (the few-frame path: second stream,
hipGraph replay, whole-wave boundary kernels, one-wave-per-restart Welsch fit) many times -- every result must hash the same.
usage: python tools/determinism.py [batch_frames] [repeats]"""
import hashlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import read_bmp_gray, GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
frames = torch.empty((n, 1080, 1920), dtype=torch.uint8, device="cuda")
det.synth_frames_device(frames.data_ptr(), 0, n, 1080, 1920, 1920, 1080 * 1920)
out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device="cuda")
hashes = set()
for _ in range(reps):
    out.zero_(); torch.cuda.synchronize()
    det.detect_batch_device(frames.data_ptr(), n, 1080, 1920, 1920, 1080 * 1920, out.data_ptr()); det.sync()
    hashes.add(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
print("batch of %d frames x %d runs: %d distinct result hashes" % (n, reps, len(hashes)))
batch = np.frombuffer(out.cpu().numpy().tobytes(), dtype=ca.RESULT_DT)
host = frames[:16].cpu().numpy()
bad = 0
for k in range(16):
    hs = set()
    for _ in range(reps * 5):
        hs.add(det.detect(host[k]).tobytes())
    bad += (len(hs) != 1) or (next(iter(hs)) != batch[k].tobytes())
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
hs = {det.detect(bmp).tobytes() for _ in range(reps * 10)}
print("16 frames x %d single calls each: %d frames differ from the batch or from themselves; test.bmp x %d calls: %d distinct" % (reps * 5, bad, reps * 10, len(hs)))
sys.exit(0 if len(hashes) == 1 and bad == 0 and len(hs) == 1 else 1)
