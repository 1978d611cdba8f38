import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import cylindertag_amd as ca, testkit as tk
from cylindertag_amd import capi
from ctag_testlib import GOLDEN, Oracle, read_marker_file
from test_gpu_parity import _random_shapes_frame
state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
orc = Oracle(); det = tk.Detector(state, fs, device=0)
det.set_option(capi.OPT_FUSED_SWEEP, 2)
for (rows, cols, seeds) in ((1080, 1920, range(40, 48)), (720, 1280, range(40, 48)), (1200, 1920, range(40, 48)), (2160, 3840, range(70, 75))):
    frames = np.stack([_random_shapes_frame(state, s, rows, cols) for s in seeds])
    got = det.detect_batch(frames)
    for f in range(len(frames)):
        o = orc.detect(frames[f], state, fs)
        q = np.frombuffer(det.debug(f, tk.DBG_CAND_QUADS).tobytes(), np.float32).reshape(-1, 8)
        oq = o["candidate_quads"]
        bad = [i for i in range(len(oq)) if q[i].tobytes() != oq[i].tobytes()]
        print("%dx%d frame %d: %d candidates, %d differ" % (cols, rows, f, len(oq), len(bad)), flush=True)
        for i in bad[:6]:
            c = o["candidates"][i]  # label, area, x_min, y_min, x_max, y_max, has_quad, n_boundary
            print("    cand %d: box x %d..%d y %d..%d (w %d h %d, x_min&63 = %d) area %d nb %d   gpu %s   oracle %s" % (
                i, c[2], c[4], c[3], c[5], c[4] - c[2] + 1, c[5] - c[3] + 1, c[2] & 63, c[1], c[7], np.round(q[i][:4], 1), np.round(oq[i][:4], 1)), flush=True)
