"""Child process of tests/test_variant_builds_gpu.py: ONE build of the library (CTAG_HIP_LIB, set by the parent before anything imports the
binding) against the oracle with the matching switches.  Every record must equal the oracle's byte for byte: test.bmp and sequence frames one
per call (the few-frame kernels), a batch of sequence frames and of synthetic frames (the batch kernels), and -- for the resize switch -- frames
whose half width is not a multiple of 16, even and odd sized.  Prints one JSON line; exit code 1 on any difference."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--welsch", type=int, default=0)
    ap.add_argument("--lanes", type=int, default=8)
    a = ap.parse_args()
    import cylindertag_amd as ca
    import testkit as tk
    from cylindertag_amd import capi
    from ctag_testlib import GOLDEN, Oracle, read_bmp_gray, read_marker_file
    from sequences import avi_substitute
    state, fs = read_marker_file(os.path.join(GOLDEN, "CTag_2f12c.marker"))
    base, orc = Oracle(), Oracle()
    orc.set_variants(a.welsch, a.lanes)  # (process-wide: `base` answers with the switches too -- the defaults are asked for first, below)
    det = tk.Detector(state, fs, device=0)
    bmp = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    seq = avi_substitute(bmp, 24)
    syn = np.stack([tk.synth_frame_host(state, f)[0] for f in range(8)])
    odd = [tk.synth_frame_host(state, 3, rows=720, cols=1296)[0],   # half width 648 = 16 * 40 + 8: a vector body of 16 leaves 8 more columns to the scalar tail
           tk.synth_frame_host(state, 5, rows=301, cols=403)[0],    # general (odd-size) resize: half width 201
           tk.synth_frame_host(state, 6, rows=598, cols=1114)[0]]   # half width 557 = 16 * 34 + 13
    singles = [("test.bmp", bmp)] + [("sequence frame %d" % k, seq[k]) for k in range(6)] + [("odd frame %d" % k, f) for k, f in enumerate(odd)]
    orc.set_variants(0, 8)
    defaults = {name: base.detect_fast(f, state, fs).tobytes() for name, f in singles}
    defaults.update({"batch %d" % k: base.detect_fast(f, state, fs).tobytes() for k, f in enumerate(list(seq) + list(syn))})
    orc.set_variants(a.welsch, a.lanes)
    bad, differs_from_default = [], 0
    for name, f in singles:
        want = orc.detect_fast(f, state, fs).tobytes()
        differs_from_default += want != defaults[name]
        if det.detect(f, 5, True, 5).tobytes() != want:
            bad.append(name)
    for chunk in (1024, 5):
        det.set_option(capi.OPT_MAX_CHUNK, chunk)
        got = det.detect_batch(np.concatenate([seq, syn]))
        for k, f in enumerate(list(seq) + list(syn)):
            want = orc.detect_fast(f, state, fs).tobytes()
            if chunk == 1024:
                differs_from_default += want != defaults["batch %d" % k]
            if got[k].tobytes() != want:
                bad.append("batch frame %d (chunk %d)" % (k, chunk))
    det.close()
    print(json.dumps({"lib": os.environ.get("CTAG_HIP_LIB", "default"), "welsch_minerr_in_loop": a.welsch, "resize_simd_lanes": a.lanes,
                      "records": len(singles) + 2 * (len(seq) + len(syn)), "mismatches": bad[:10],
                      "oracle_records_that_differ_from_the_default_oracle": int(differs_from_default)}), flush=True)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
