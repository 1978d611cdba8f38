#!/usr/bin/env python3
"""One-off differential hunt (GPU box): one BATCH of random shape frames through ctag_detect_batch_u8, every record vs the oracle.
usage: [CTAG_FUSED_SWEEP=2] [CTAG_SWEEP_SEED=9000] python tools/batch_sweep.py [frames] [rows cols]   (CTAG_FUSED_SWEEP=2: the fused sweep and the mask-based
silhouettes whatever the batch size, for the frame sizes that allow them)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from ctag_testlib import Oracle, read_marker_file, GOLDEN
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
src = open(os.path.join(ROOT, "tests", "test_gpu_parity.py")).read()
ns = {"ca": ca, "np": np, "tk": tk}
exec(src[src.index("def _random_shapes_frame"):src.index("def test_random_shapes_fuzz")], ns)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
rows, cols = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (720, 1152)
seed0 = int(os.environ.get("CTAG_SWEEP_SEED", "5000"))  # another population of shapes
frames = np.stack([ns["_random_shapes_frame"](state, seed0 + i, rows, cols) for i in range(n)])
orc, det = Oracle(), tk.Detector(state, fs)
for chunk in (1024, 37):
    det.set_option(capi.OPT_MAX_CHUNK, chunk)
    got = det.detect_batch(frames)
    bad = [i for i in range(n) if got[i].tobytes() != orc.detect_fast(frames[i], state, fs).tobytes()]
    print("chunk", chunk, "frames", n, "mismatches", bad[:10])
    if bad:
        sys.exit(1)
