#!/usr/bin/env python3
"""Regenerates tests/golden/golden_v1.npz from the CPU oracle (oracle/ctag_oracle.cpp).

The reference ships no golden vectors and cannot be built here, so these fixtures pin the oracle's own outputs
(regression) and give the GPU tests committed expected values.  Inputs: the reference's test.bmp, the 64-frame
"test.avi substitute" derived from it (integer shifts + fixed-point gain, seed 2) and
synthetic frames 0..7 of the bench generator.  Run:  python tests/golden/make_golden.py"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ctag_testlib import GOLDEN, Oracle, read_bmp_gray, read_marker_file  # noqa: E402
from sequences import avi_substitute  # noqa: E402
import cylindertag_amd as ca  # noqa: E402
import testkit as tk  # noqa: E402


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    orc = Oracle()
    img = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    out = {}
    r = orc.detect(img, state, fs)
    out["bmp_result"] = np.array([r["result"]])
    out["bmp_hashes"] = np.array([sha(r["half"]), sha(r["binary"]), sha(r["labels"]), sha(r["candidates"]),
                                  sha(r["candidate_quads"]), sha(r["features"][2])])
    out["bmp_counts"] = np.array([len(r["areas"]), len(r["candidates"]), len(r["quads"]), len(r["features"][0])])
    seq = avi_substitute(img, 64)
    out["seq_results"] = np.array([orc.detect_fast(f, state, fs) for f in seq])
    syn = []
    for f in range(8):
        frame, truth = tk.synth_frame_host(state, f)
        syn.append(orc.detect_fast(frame, state, fs))
    out["synth_results"] = np.array(syn)
    np.savez_compressed(os.path.join(HERE, "golden_v1.npz"), **out)
    print("written", os.path.join(HERE, "golden_v1.npz"))
    print("test.bmp markers:", [int(m["marker_id"]) for m in r["result"]["markers"][:r["result"]["n_markers"]]])
    print("sequence markers per frame:", [int(x["n_markers"]) for x in out["seq_results"]])
    print("synthetic markers per frame:", [int(x["n_markers"]) for x in out["synth_results"]])


if __name__ == "__main__":
    main()
