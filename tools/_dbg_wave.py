import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import cylindertag_amd as ca
from cylindertag_amd import capi
from ctag_testlib import read_bmp_gray, GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = ca.Detector(state, fs)
img = ca.synth_frame_host(state, 0)[0]
out = {}
for wp in (100000, 1):
    det.set_option(capi.OPT_WAVE_POINTS, wp)
    r = det.detect(img)
    out[wp] = (det.debug(0, capi.DBG_CANDIDATES).copy(), det.debug(0, capi.DBG_CAND_QUADS).copy(), r.tobytes())
a, b = out[100000], out[1]
bad = np.nonzero((a[0][:, :7] != b[0][:, :7]).any(axis=1))[0]
print(os.environ.get("CTAG_HIP_LIB", "default"), "candidates", len(a[0]), "differ (area..n_boundary)", len(bad), "quads equal", a[1].tobytes() == b[1].tobytes(), "records equal", a[2] == b[2])
for i in bad[:6]:
    print(i, a[0][i], b[0][i])
