#!/usr/bin/env python3
"""One-off differential hunt (GPU box): ctag_pose_batch_device vs the CPU pose oracle on many seeds of synthetic detection
records (random poses, noise levels, feature-id patterns), byte for byte.  usage: python tools/pose_fuzz.py [n_seeds]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import GOLDEN
from pose_testlib import PoseOracle, make_camera, make_model_view, read_camera_yml, read_model_file, synth_pose_results
K, dist = read_camera_yml(os.path.join(GOLDEN, "cameraParams.yml"))
model = read_model_file(os.path.join(GOLDEN, "CTag_2f12c.model"))
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
M, cam = ca.Model(os.path.join(GOLDEN, "CTag_2f12c.model")), ca.load_camera(os.path.join(GOLDEN, "cameraParams.yml"))
po, cam_o, mv = PoseOracle(), make_camera(K, dist), make_model_view(model)
n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 16
bad = tot = 0
for seed in range(100, 100 + n_seeds):
    noise = [0.0, 0.05, 0.2, 1.0, 5.0][seed % 5]
    recs, _ = synth_pose_results(model, K, dist, 256, seed, noise_px=noise)
    d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), -1)).cuda()
    off = torch.zeros(len(recs) + 1, dtype=torch.int32, device="cuda")
    cap = max(1, int(recs["n_markers"].sum()))
    poses = torch.zeros(cap * ca.POSE_DT.itemsize, dtype=torch.uint8, device="cuda")
    det.pose_batch_device(d.data_ptr(), len(recs), M, cam, off.data_ptr(), poses.data_ptr(), cap)
    det.sync()
    offs = off.cpu().numpy()
    P = poses.cpu().numpy().view(ca.POSE_DT)[:offs[-1]]
    for f in range(len(recs)):
        want = po.pose_frame(recs[f], mv, cam_o, f)
        got = P[offs[f]:offs[f + 1]]
        tot += len(want)
        if got.tobytes() != want.tobytes():
            bad += 1
            if bad <= 5:
                print("MISMATCH seed", seed, "frame", f, got, want)
print("seeds", n_seeds, "markers", tot, "mismatching frames", bad)
sys.exit(1 if bad else 0)
