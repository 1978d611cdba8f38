"""GPU parity of the mask-based silhouettes (-m gpu; round 6): in batches that took the fused sweep, k_silhouette_mask derives every packed component's
first / last pixel per row and column (corner_detector.cpp:184-232) from the threshold mask's row runs + one label probe per run, instead of testing every
label of the bounding box.  The fitted quads of every candidate (DBG_CAND_QUADS: what the silhouette, the ordered traversal and the edge clusters end in) and
the records must be the oracle's byte for byte -- on frames that are NOT markers: random shapes (boxes of other components inside a component's box, runs cut
by the box's edge, rings, L-shapes), wide bars (boxes over several mask words, runs across word boundaries), blob fields and textures, components across the
320-column / 30-row label tiles.  CTAG_OPT_FUSED_SWEEP = 2 makes a batch of any length take the fused sweep; the batches here have more than kLatencyFrames
frames, so the packed builds (not the whole-wave ones) run."""
import numpy as np
import pytest

import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from clutter import blob_field, chevron_texture
from test_gpu_parity import _random_shapes_frame, assert_same_record

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused(detector):
    detector.set_option(capi.OPT_FUSED_SWEEP, 2)
    yield detector
    detector.set_option(capi.OPT_FUSED_SWEEP, 1)


def _check_batch(det, oracle, state, fs, frames, what, every=1):
    got = det.detect_batch(frames)
    for f in range(0, len(frames), every):
        o = oracle.detect(frames[f], state, fs)
        assert det.debug(f, tk.DBG_MASK).reshape(o["binary"].shape).any() or not o["binary"].any()  # the fused sweep ran: a mask exists
        assert det.debug(f, tk.DBG_CAND_QUADS).tobytes() == o["candidate_quads"].tobytes(), "%s, frame %d: candidate quads" % (what, f)
        assert_same_record(got[f], o["result"], "%s, frame %d" % (what, f))
    return got


def test_random_shapes_in_fused_batches(fused, oracle, dictionary):
    state, fs = dictionary
    for (rows, cols) in ((1080, 1920), (720, 1280), (1200, 1920)):
        frames = np.stack([_random_shapes_frame(state, 40 + s, rows, cols) for s in range(8)])
        _check_batch(fused, oracle, state, fs, frames, "random shapes %dx%d" % (cols, rows))


def test_wide_bars_and_word_boundaries(fused, oracle, dictionary):
    """Bars 260-360 half-size pixels wide (boxes of five to seven mask words; a row of the bar is ONE run across all of them), thin diagonals whose rows hold
    short runs at moving offsets, combs whose teeth end inside other components' boxes, and bars that start / end exactly on 64-column word boundaries."""
    state, fs = dictionary
    rng = np.random.RandomState(78)
    frames = []
    for f in range(8):
        img = (rng.randint(0, 7, (1080, 1920)) + 190).astype(np.uint8)
        for k in range(20):
            y = 20 + k * 52 + rng.randint(0, 6)
            x = rng.randint(10, 400)
            img[y:y + rng.randint(10, 16), x:x + rng.randint(520, 720)] = 25 + (f + k) % 20
        # bars whose half-size extent is exactly [64 a, 64 b): full-resolution columns 128 a .. 128 b - 1 (the 2x decimation keeps such an edge within a pixel)
        img[30 + 52 * 19 + 20:30 + 52 * 19 + 34, 128 * 9:128 * 12] = 30
        yy, xx = np.mgrid[0:1080, 0:1920]
        d = yy - (100 + 40 * f + 0.22 * xx)
        img[(np.abs(d) < 4.0) & (xx > 900 + 10 * f) & (xx < 1700)] = 35  # a diagonal through the bars' boxes
        for t in range(12):  # a comb: teeth of 6 px every 20, joined by a spine
            img[700 + 3 * f:760 + 3 * f, 1200 + 40 * t:1212 + 40 * t] = 28
        img[756 + 3 * f:768 + 3 * f, 1200:1200 + 40 * 12] = 28
        frames.append(img)
    _check_batch(fused, oracle, state, fs, np.stack(frames), "wide bars")


def test_clutter_in_fused_batches(fused, oracle, dictionary):
    """Blob fields and a texture among marker frames, through the fused sweep: the cluttered frames exhaust the batch workspace's pools and are run again
    alone (not fused); their neighbours in the batch take the mask-based silhouettes."""
    state, fs = dictionary
    frames = np.stack([tk.synth_frame_host(state, 300 + f)[0] for f in range(12)])
    frames[2] = blob_field(frames[2], pitch_x=60, pitch_y=44)[0]   # ~700 blobs: fits the batch workspace, every blob a packed component
    frames[5] = blob_field(frames[5])[0]                            # > 2500: rerun
    frames[9] = chevron_texture(frames[9])[0]
    want, _ = oracle.detect_many(frames, state, fs)
    got = fused.detect_batch(frames)
    for f in range(12):
        assert_same_record(got[f], want[f], "clutter batch frame %d" % f)
    # the blob field that fits the batch workspace, in a batch of its own (the reruns above replaced what the probes look at): every blob a packed component
    few = np.stack([frames[2], frames[0], frames[2], frames[1], frames[2], frames[3]])
    _check_batch(fused, oracle, state, fs, few, "blob field in a fused batch", every=2)
    assert len(oracle.detect(frames[2], state, fs)["candidates"]) > 400


def test_mask_and_label_silhouettes_agree_at_4k(fused, oracle, dictionary):
    state, fs = dictionary
    frames = np.stack([_random_shapes_frame(state, 70 + s, 2160, 3840) for s in range(5)])
    _check_batch(fused, oracle, state, fs, frames, "random shapes 3840x2160", every=2)


def test_frames_without_components_in_fused_batches(fused, oracle, dictionary):
    """Blank frames (no component at all, hence no pack order entry), frames of one tiny component and marker frames in one batch."""
    state, fs = dictionary
    frames = np.stack([tk.synth_frame_host(state, 400 + f)[0] for f in range(10)])
    frames[1] = 230
    frames[4] = 0
    frames[7] = 200
    frames[7, 500:520, 900:960] = 20  # one dark bar
    want, _ = oracle.detect_many(frames, state, fs)
    assert want["status"][1] != 0 and want["status"][4] != 0
    for rep in range(2):  # twice: the second pass finds the first one's order entries in the workspace
        got = fused.detect_batch(frames[::-1] if rep else frames)
        for f in range(10):
            assert_same_record(got[f], want[9 - f] if rep else want[f], "blank-frame batch, pass %d, frame %d" % (rep, f))
