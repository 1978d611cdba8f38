#!/usr/bin/env python3
"""Differential hunt (GPU box) on cluttered content: random mixtures of blob fields (random pitch, size, shapes), chevron textures, noise patches
and markers at random frame sizes, one frame per call and in small batches; GPU record vs oracle record byte for byte, flags == 0 wherever the
oracle's are.  Exercises the paths only cluttered frames take: k_candidates' global-memory sort, the counting sorts of k_pack / k_line_sort, the
any-frame workspace (its pools, the second CCL pass publishing into the large component pool).  usage: python tools/clutter_fuzz.py [n_cases]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import Oracle, read_marker_file, GOLDEN
from clutter import blob_field, chevron_texture
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
orc, det = Oracle(), tk.Detector(state, fs)
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)  # usage: clutter_fuzz.py [n_cases] [seed]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = limit = 0
sizes = [(1080, 1920), (1080, 1920), (720, 1280), (1200, 1920), (2160, 3840), (902, 1444), (600, 800)]
frames_kept = []
for case in range(n):
    rows, cols = sizes[rng.randint(0, len(sizes))]
    base = tk.synth_frame_host(state, 7000 + case, rows, cols)[0] if rng.rand() < 0.7 else np.full((rows, cols), int(rng.randint(150, 240)), np.uint8)
    kind = rng.randint(0, 4)
    if kind == 0:
        img = blob_field(base, pitch_x=int(rng.randint(22, 40)), pitch_y=int(rng.randint(20, 36)), size=(int(rng.randint(14, 20)), int(rng.randint(20, 26))),
                         seed=case, shapes=tuple(rng.choice(5, size=rng.randint(1, 4), replace=False).tolist()))[0]
    elif kind == 1:
        img = chevron_texture(base, pitch=int(rng.randint(7, 16)), band=int(rng.choice([40, 60, 84, 120])), arm=int(rng.randint(16, 60)), width=int(rng.randint(3, 6)))[0]
    elif kind == 2:  # noise patches: dense speckle beside everything else
        img = blob_field(base, seed=case)[0]
        for _ in range(rng.randint(1, 5)):
            y0, x0 = rng.randint(0, rows - 100), rng.randint(0, cols - 200)
            h, w = rng.randint(40, 100), rng.randint(80, 200)
            img[y0:y0 + h, x0:x0 + w] = np.clip(rng.normal(rng.randint(20, 90), rng.randint(5, 40), (h, w)), 0, 255).astype(np.uint8)
    else:  # both textures
        img = chevron_texture(blob_field(base, pitch_x=30, pitch_y=28, seed=case)[0], pitch=int(rng.randint(8, 14)))[0]
    img = np.ascontiguousarray(img)
    want = orc.detect_fast(img, state, fs)
    if want["status"] == -3:
        limit += 1  # the reference's own fixed arrays (> 1000 quads ...): both sides must say so, with the same flag
    try:
        got = det.detect(img)
        same = got.tobytes() == want.tobytes()
        info = (got["status"], got["flags"])
    except ca.CtagError as e:
        same = e.status == want["status"] == -3 and (want["flags"] & 7) != 0 and (want["flags"] & 8) == 0
        info = (e.status, None)
    if not same:
        bad += 1
        print("MISMATCH case", case, img.shape, "kind", kind, "gpu", info, "oracle", want["status"], want["flags"])
    if (rows, cols) == (1080, 1920) and len(frames_kept) < 24:
        frames_kept.append(img)
c = det.counters()
print("cases", n, "mismatches", bad, "reference-limit frames", limit, "frames through the any-frame workspace", c["reruns"])
if frames_kept:  # the same frames as one batch (pending frames inside a batch)
    batch = np.stack(frames_kept)
    got = det.detect_batch(batch)
    wantb, _ = orc.detect_many(batch, state, fs)
    badb = [i for i in range(len(batch)) if got[i].tobytes() != wantb[i].tobytes()]
    print("batch of", len(batch), "mismatches", badb)
    bad += len(badb)
sys.exit(1 if bad else 0)
