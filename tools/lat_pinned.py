import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cylindertag_amd as ca
import testkit as tk
from ctag_testlib import read_bmp_gray, GOLDEN
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
det = tk.Detector(state, fs)
bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
pin = ca.pinned_empty(bmp.shape, np.uint8); pin[...] = bmp
for name, img in (("pageable", bmp), ("pinned", pin)):
    for _ in range(20): det.detect(img)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); det.detect(img); ts.append(time.perf_counter() - t0)
    ts = np.sort(ts) * 1e3
    print(name, "median %.4f p10 %.4f p90 %.4f" % (ts[150], ts[30], ts[270]))
