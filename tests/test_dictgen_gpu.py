"""Round trip generator -> renderer -> detector on a dictionary that is NOT the reference's: a freshly generated 12-column
dictionary is planted into synthetic frames (ctag_synth, the reference generator's strip geometry) and must come back from
the HIP path -- decoded rows equal the planted ones and every record equals the CPU oracle's byte for byte."""
import os
import sys

import numpy as np
import pytest

import cylindertag_amd as ca
from ctag_testlib import ROOT, Oracle

sys.path.insert(0, os.path.join(ROOT, "tools"))
import dict_gen as dg  # noqa: E402

pytestmark = pytest.mark.gpu


def test_generated_dictionary_round_trip():
    code = dg.Generator(12, 2, seed=11).generate(30)
    assert code.shape == (30, 12) and dg.test_conflict(code, 2)
    det, orc = ca.Detector(code, 2, device=0), Oracle()
    exact = 0
    n = 24
    for f in range(n):
        frame, truth = ca.synth_frame_host(code, 700 + f)
        got = det.detect(frame, 5, True, 5)
        want = orc.detect_fast(frame, code, 2, 5, True, 5)
        assert got.tobytes() == want.tobytes(), f
        planted = sorted(int(x) for x in truth["dict_row"][:truth["n_markers"]])
        found = sorted(int(x) for x in got["markers"]["marker_id"][:got["n_markers"]])
        assert set(found) <= set(planted), f  # never a wrong id
        exact += planted == found
    assert exact >= int(0.85 * n)
    det.close()
