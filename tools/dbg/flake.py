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
dev = torch.device("cuda:0")
first = sys.argv[1] if len(sys.argv) > 1 else "8k"
det.set_option(capi.OPT_FUSED_SWEEP, 2)
if first == "8k":
    det.detect(tk.synth_frame_host(state, 2, 4320, 7680)[0])
elif first == "4k":
    det.detect(tk.synth_frame_host(state, 2, 2160, 3840)[0])
det.set_option(capi.OPT_FUSED_SWEEP, 1)
for rows in (1200, 1080):
    src = bmp if rows == 1200 else bmp[60:1140]
    base = np.stack([np.roll(src, 3 * k, axis=1) for k in range(8)])
    bgr8 = np.stack([_colourise(base[k], k) for k in range(8)])
    grays = np.stack([orc.bgr2gray(bgr8[k]) for k in range(8)])
    wantb = [orc.detect_fast(grays[k], state, fs) for k in range(8)]
    m = 64
    bgr = torch.from_numpy(bgr8).to(dev).repeat(m // 8, 1, 1, 1).contiguous()
    gray = torch.from_numpy(grays).to(dev).repeat(m // 8, 1, 1).contiguous()
    out = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    def run(kind, direct, mode, reps=12):
        det.set_option(capi.OPT_FUSED_SWEEP, mode)
        det.set_option(capi.OPT_BGR_DIRECT, direct)
        fails = []
        for rep in range(reps):
            out.zero_()
            if kind == "bgr":
                det.detect_batch_bgr_device(bgr.data_ptr(), m, rows, 1920, 1920 * 3, rows * 1920 * 3, out.data_ptr())
            else:
                det.detect_batch_device(gray.data_ptr(), m, rows, 1920, 1920, rows * 1920, out.data_ptr())
            det.sync()
            got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
            bad = [k for k in range(m) if got[k].tobytes() != wantb[k % 8].tobytes()]
            if bad:
                fails.append((rep, bad[:6]))
        print("after %s: rows %d %s direct %d fused_mode %d: %d of %d runs differ %s" % (first, rows, kind, direct, mode, len(fails), reps, fails[:3]), flush=True)
    run("bgr", 1, 2)
    run("gray", 1, 2)
    run("bgr", 0, 2)
    run("bgr", 1, 0)
    run("bgr", 1, 2)
