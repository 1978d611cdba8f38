"""developer aid: where does the GPU half-resolution image differ from the oracle's on test.bmp?"""
import os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from ctag_testlib import *
state, fs = ca.load_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
img = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
det = tk.Detector(state, fs); orc = Oracle()
o = orc.resize_half(img)
det.detect(img)
h = det.debug(0, tk.DBG_HALF).reshape(o.shape)
d = np.argwhere(h != o)
print("mismatches", len(d), "of", o.size)
for y, x in d[:12]:
    src = img[2*y-1:2*y+3, 2*x-1:2*x+3].astype(np.int64)
    q = 19*(src[:,1]+src[:,2]) - 3*(src[:,0]+src[:,3])
    V = 19*(q[1]+q[2]) - 3*(q[0]+q[3])
    print(y, x, "gpu", h[y,x], "oracle", o[y,x], "V", V, "V/1024", V/1024, "rem", V & 1023)
