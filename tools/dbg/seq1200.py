"""replays tests/test_fused_sweep_gpu.py::test_general_fused_build_on_batches_and_bgr's sequence several times and lists the frames that differ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cylindertag_amd as ca, testkit as tk
from cylindertag_amd import capi
from ctag_testlib import GOLDEN, Oracle, read_bmp_gray, read_marker_file
from test_gpu_parity import _colourise
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
orc = Oracle(); det = tk.Detector(state, fs, device=0)
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
base = np.stack([np.roll(bmp, 3 * k, axis=1) for k in range(8)])
want8, _ = orc.detect_many(base, state, fs)
bgr8 = np.stack([_colourise(base[k], k) for k in range(8)])
wantb = [orc.detect_fast(orc.bgr2gray(bgr8[k]), state, fs) for k in range(8)]
dev = torch.device("cuda:0")
n, m = 520, 64
frames = torch.from_numpy(base).to(dev).repeat(n // 8, 1, 1).contiguous()
bgr = torch.from_numpy(bgr8).to(dev).repeat(m // 8, 1, 1, 1).contiguous()
out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
outb = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
for rep in range(4):
    det.set_option(capi.OPT_FUSED_SWEEP, 1)
    det.detect_batch_device(frames.data_ptr(), n, 1200, 1920, 1920, 1200 * 1920, out.data_ptr()); det.sync()
    got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
    print("rep", rep, "gray 520 bad:", [k for k in range(n) if got[k].tobytes() != want8[k % 8].tobytes()][:10], flush=True)
    det.debug(n - 1, tk.DBG_MASK)
    det.set_option(capi.OPT_FUSED_SWEEP, 2)
    det.detect_batch_bgr_device(bgr.data_ptr(), m, 1200, 1920, 1920 * 3, 1200 * 1920 * 3, outb.data_ptr()); det.sync()
    got = np.frombuffer(outb.cpu().numpy().tobytes(), ca.RESULT_DT)
    bad = [k for k in range(m) if got[k].tobytes() != wantb[k % 8].tobytes()]
    print("rep", rep, "bgr 64 bad:", bad, flush=True)
    for k in bad[:2]:
        g, w_ = got[k], wantb[k % 8]
        nf = int(w_["n_features"])
        d = np.abs(g["features"]["corners"][:nf] - w_["features"]["corners"][:nf]).max()
        print("   frame", k, "status", g["status"], w_["status"], "nf", g["n_features"], nf, "max corner diff", d, "ids differ at", np.nonzero(g["features"]["id"][:nf] != w_["features"]["id"][:nf])[0][:5])
