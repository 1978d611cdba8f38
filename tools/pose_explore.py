import os, sys, numpy as np, torch, time
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import *
from pose_testlib import *
K, dist = read_camera_yml(os.path.join(GOLDEN, "cameraParams.yml"))
model = read_model_file(os.path.join(GOLDEN, "CTag_2f12c.model"))
cam_o = make_camera(K, dist); mv = make_model_view(model); po = PoseOracle()
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
M = ca.Model(os.path.join(GOLDEN, "CTag_2f12c.model")); cam = ca.load_camera(os.path.join(GOLDEN, "cameraParams.yml"))
res = det.detect(read_bmp_gray(os.path.join(GOLDEN, "test.bmp")), 5, True, 5)
got = det.estimate_pose(res, M, cam)
want = po.pose_frame(res, mv, cam_o)
print("test.bmp identical:", got.tobytes() == want.tobytes())
for g, w in zip(got, want):
    print(g["status"], g["model_index"], g["n_points"], g["iterations"], w["iterations"], np.abs(g["rvec"]-w["rvec"]).max(), np.abs(g["tvec"]-w["tvec"]).max(), np.abs(g["rvec0"]-w["rvec0"]).max(), np.abs(g["tvec0"]-w["tvec0"]).max(), g["cost"], w["cost"])
# batch
recs, truth = synth_pose_results(model, K, dist, 512, 1)
d = torch.from_numpy(recs.view(np.uint8).reshape(len(recs), -1)).cuda()
off = torch.zeros(len(recs) + 1, dtype=torch.int32, device="cuda")
cap = len(recs) * 8
poses = torch.zeros(cap * ca.POSE_DT.itemsize, dtype=torch.uint8, device="cuda")
det.set_option(2, 1)
for rep in range(3):
    det.pose_batch_device(d.data_ptr(), len(recs), M, cam, off.data_ptr(), poses.data_ptr(), cap)
    det.sync()
    print("pose ms", det.pose_last_ms())
offs = off.cpu().numpy(); P = poses.cpu().numpy().view(ca.POSE_DT)[:offs[-1]]
print("total markers", offs[-1])
nbad = 0; nid = 0; mx = 0
t0 = time.time()
for f in range(len(recs)):
    w = po.pose_frame(recs[f], mv, cam_o, f)
    g = P[offs[f]:offs[f + 1]]
    assert len(w) == len(g)
    if g.tobytes() == w.tobytes(): nid += 1
    else:
        nbad += 1
        for a, b in zip(g, w):
            if a.tobytes() != b.tobytes():
                if nbad < 6: print(f, a, b)
                if a["status"] == 0 and b["status"] == 0: mx = max(mx, np.abs(a["rvec"]-b["rvec"]).max(), np.abs(a["tvec"]-b["tvec"]).max())
print("frames identical", nid, "different", nbad, "max diff", mx, "oracle s", time.time()-t0)
st = P["status"]; print("status hist", np.bincount(st), "iters", np.bincount(P["iterations"]))
# truth check
err = []
for f in range(len(recs)):
    for k, (mi, rv, tv) in enumerate(truth[f]):
        p = P[offs[f] + k]
        if p["status"] == 0: err.append(np.abs(p["tvec"] - tv).max())
print("tvec err vs truth: median", np.median(err), "max", np.max(err))
