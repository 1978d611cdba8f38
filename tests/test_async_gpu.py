"""ctag_submit_u8 / ctag_collect (-m gpu): one frame per call with two frames in flight -- the reference's camera loop (main.cpp:44-61) with the
upload of frame k + 1 behind the detection of frame k.  Records must be the oracle's in submission order, also for a cluttered frame that
needs the any-frame workspace, and a third submit must be refused until a frame is collected."""
import numpy as np
import pytest

import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from clutter import blob_field
from sequences import avi_substitute
from test_gpu_parity import assert_same_record

pytestmark = pytest.mark.gpu


def test_submit_collect_pipeline(detector, oracle, dictionary, test_bmp):
    state, fs = dictionary
    seq = avi_substitute(test_bmp, 12)
    frames = ca.pinned_empty(seq.shape, np.uint8)
    frames[...] = seq
    want, _ = oracle.detect_many(seq, state, fs)
    got = []
    detector.submit(frames[0])
    for k in range(1, len(frames)):
        detector.submit(frames[k])           # frame k uploads while frame k - 1 is being detected
        got.append(detector.collect().copy())
    got.append(detector.collect().copy())
    for k in range(len(frames)):
        assert_same_record(got[k], want[k], "async frame %d" % k)
    with pytest.raises(ca.CtagError):
        detector.collect()                   # nothing in flight
    detector.submit(frames[0])
    detector.submit(frames[1])
    with pytest.raises(ca.CtagError):
        detector.submit(frames[2])           # two in flight already
    assert_same_record(detector.collect().copy(), want[0], "after refusal 0")
    assert_same_record(detector.collect().copy(), want[1], "after refusal 1")


def test_submit_collect_cluttered_frame_and_size_change(detector, oracle, dictionary):
    state, fs = dictionary
    a = tk.synth_frame_host(state, 11)[0]
    b = blob_field(tk.synth_frame_host(state, 3)[0])[0]      # > 2500 candidates: completed through the any-frame workspace inside collect
    c = np.ascontiguousarray(tk.synth_frame_host(state, 12, 720, 1280)[0])  # another size between two submits
    before = detector.counters()["reruns"]
    detector.submit(a)
    detector.submit(b)
    ra = detector.collect().copy()
    detector.submit(c)
    rb = detector.collect().copy()
    rc = detector.collect().copy()
    for name, img, r in (("plain", a, ra), ("cluttered", b, rb), ("720p", c, rc)):
        assert_same_record(r, oracle.detect_fast(img, state, fs), "async " + name)
    assert rb["flags"] == 0 and detector.counters()["reruns"] == before + 1
