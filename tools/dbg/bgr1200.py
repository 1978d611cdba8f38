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
bgr8 = np.stack([_colourise(base[k], k) for k in range(8)])
grays = [orc.bgr2gray(bgr8[k]) for k in range(8)]
want = [orc.detect_fast(g, state, fs) for g in grays]
dev = torch.device("cuda:0")
m = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bgr = torch.from_numpy(bgr8).to(dev).repeat(m // 8, 1, 1, 1).contiguous()
gray = torch.from_numpy(np.stack(grays)).to(dev).repeat(m // 8, 1, 1).contiguous()
out = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
def bad(got):
    return [k for k in range(m) if got[k].tobytes() != want[k % 8].tobytes()]
for mode in (2, 0):
    det.set_option(capi.OPT_FUSED_SWEEP, mode)
    for rep in range(2):
        out.zero_()
        det.detect_batch_bgr_device(bgr.data_ptr(), m, 1200, 1920, 1920 * 3, 1200 * 1920 * 3, out.data_ptr()); det.sync()
        print("bgr fused_mode", mode, "rep", rep, "bad frames", bad(np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)), flush=True)
        out.zero_()
        det.detect_batch_device(gray.data_ptr(), m, 1200, 1920, 1920, 1200 * 1920, out.data_ptr()); det.sync()
        print("gray of the same frames, fused_mode", mode, "rep", rep, "bad frames", bad(np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)), flush=True)
