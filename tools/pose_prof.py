"""Phase profile of k_pose (library built with EXTRA=-DCTAG_POSE_PROF): cycles per phase summed over all markers.
usage (GPU box): CTAG_HIP_LIB=$PWD/cylindertag_amd/_var/prof/libctag_hip.so python tools/pose_prof.py [n_frames]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "tests")
sys.path.insert(0, ".")
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import GOLDEN
from pose_testlib import read_camera_yml, read_model_file, synth_pose_results

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K, dist = read_camera_yml(os.path.join(GOLDEN, "cameraParams.yml"))
model = read_model_file(os.path.join(GOLDEN, "CTag_2f12c.model"))
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
M = ca.Model(os.path.join(GOLDEN, "CTag_2f12c.model"))
cam = ca.load_camera(os.path.join(GOLDEN, "cameraParams.yml"))
recs, truth = synth_pose_results(model, K, dist, 512, 1)
recs = np.tile(recs, (n_frames + 511) // 512)[:n_frames]
d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), -1)).cuda()
off = torch.zeros(len(recs) + 1, dtype=torch.int32, device="cuda")
cap = int(recs["n_markers"].sum())
poses = torch.zeros(cap * ca.POSE_DT.itemsize, dtype=torch.uint8, device="cuda")
det.set_option(2, 1)
L = ca.load_library()
has_prof = hasattr(L, "ctag_pose_debug_prof")
buf = (C.c_ulonglong * 16)()
for rep in range(3):
    if has_prof:
        L.ctag_pose_debug_prof(buf, 1)
    det.pose_batch_device(d.data_ptr(), len(recs), M, cam, off.data_ptr(), poses.data_ptr(), cap)
    det.sync()
    print("markers %d  pose ms %.3f  (%.1f K markers/s)" % (cap, det.pose_last_ms(), cap / det.pose_last_ms()))
if has_prof:
    L.ctag_pose_debug_prof(buf, 0)
    names = ["setup+corr", "ctrl+alphas", "MtM", "jacobi", "null+L", "betas+GN", "R_and_t x3", "pick+rodrigues", "LM"]
    tot = sum(buf[:9])
    for i, nme in enumerate(names):
        print("%-16s %8.1f cycles/marker  %5.1f %%" % (nme, buf[i] / cap, 100.0 * buf[i] / tot))
