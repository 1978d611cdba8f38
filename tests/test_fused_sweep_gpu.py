"""GPU parity of the fused sweep (-m gpu): k_decimate_mask thresholds the half-size pixels where it computes them and hands K2 one bit
per pixel (no half-size image in HBM).  CTAG_OPT_FUSED_SWEEP = 2 forces that form for any number of frames of a size that allows it
(half size a multiple of 320 x 5); the mask must equal the oracle's adaptiveThreshold output (corner_detector.cpp:28-79)
bit for bit, and everything behind it the oracle's stage by stage."""
import numpy as np
import pytest

import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from clutter import blob_field
from test_gpu_parity import _stage_check, assert_same_record

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fused(detector):
    detector.set_option(capi.OPT_FUSED_SWEEP, 2)
    yield detector
    detector.set_option(capi.OPT_FUSED_SWEEP, 1)


def _mask_check(det, oracle, state, fs, img, what):
    o, r, lab = _stage_check(det, oracle, state, fs, img, what)
    mask = det.debug(0, tk.DBG_MASK).reshape(o["binary"].shape)
    assert (mask == (o["binary"] > 0)).all(), what + ": threshold mask"
    assert (det.debug(0, tk.DBG_HALF).reshape(o["half"].shape) == o["half"]).all(), what  # the stand-alone decimation, for the record
    return o, r


def test_fused_mask_equals_adaptive_threshold_1080p(fused, oracle, dictionary, test_bmp):
    state, fs = dictionary
    rng = np.random.RandomState(17)
    yy, xx = np.mgrid[0:1080, 0:1920]
    cases = [("test.bmp rows 60..1139", test_bmp[60:1140]),
             ("synthetic frame 2", tk.synth_frame_host(state, 2)[0]),
             ("blob field", blob_field(tk.synth_frame_host(state, 3)[0])[0]),
             # thresholds of every kind: ramps through the 0.3 cap (77) with noise, so that tile bounds come from the table, from the cap and from 0
             ("ramp + noise", np.clip(xx * (150.0 / 1920) + yy * (40.0 / 1080) + rng.randint(-30, 31, (1080, 1920)), 0, 255).astype(np.uint8)),
             ("dark noise", np.clip(rng.normal(40, 25, (1080, 1920)), 0, 255).astype(np.uint8)),
             ("checker of 5-pixel tiles", (((yy // 10 + xx // 10) & 1) * 120 + 20 + rng.randint(0, 3, (1080, 1920))).astype(np.uint8)),
             ("all dark", np.zeros((1080, 1920), np.uint8)), ("all bright", np.full((1080, 1920), 230, np.uint8))]
    for name, img in cases:
        _mask_check(fused, oracle, state, fs, np.ascontiguousarray(img), name)


def test_fused_mask_4k_two_waves_per_row(fused, oracle, dictionary):
    """3840x2160: two waves per half-size row; the tile column on either side of column 960 comes from the halo lanes."""
    state, fs = dictionary
    rng = np.random.RandomState(23)
    img = tk.synth_frame_host(state, 1, 2160, 3840)[0].copy()
    img[:, 1880:1960] = np.clip(rng.normal(60, 30, (2160, 80)), 0, 255).astype(np.uint8)  # texture across the seam between the waves
    _mask_check(fused, oracle, state, fs, img, "4K frame with a noise band across the wave seam")
    noise = np.clip(rng.normal(70, 40, (2160, 3840)), 0, 255).astype(np.uint8)
    _mask_check(fused, oracle, state, fs, noise, "4K noise")


def test_fused_and_two_kernel_sweeps_give_the_same_records(detector, oracle, dictionary, test_bmp):
    state, fs = dictionary
    frames = np.stack([tk.synth_frame_host(state, 100 + f)[0] for f in range(24)] + [test_bmp[60:1140]])
    want, _ = oracle.detect_many(frames, state, fs)
    try:
        for mode in (0, 2):
            detector.set_option(capi.OPT_FUSED_SWEEP, mode)
            got = detector.detect_batch(frames)
            for k in range(len(frames)):
                assert_same_record(got[k], want[k], "mode %d frame %d" % (mode, k))
    finally:
        detector.set_option(capi.OPT_FUSED_SWEEP, 1)


def test_fused_sweep_padded_rows_and_8k(fused, oracle, dictionary):
    """Rows padded beyond the frame width (row_stride 2048 for 1920 columns, device memory), and a 7680x4320 frame: four waves per half-size
    row, the two middle ones with halo lanes on both sides."""
    import torch
    state, fs = dictionary
    frames = np.stack([tk.synth_frame_host(state, 40 + f)[0] for f in range(6)])
    want, _ = oracle.detect_many(frames, state, fs)
    dev = torch.device("cuda:0")
    padded = torch.full((6, 1080, 2048), 77, dtype=torch.uint8, device=dev)
    padded[:, :, :1920] = torch.from_numpy(frames).to(dev)
    out = torch.zeros(6 * ca.RESULT_DT.itemsize, dtype=torch.uint8, device=dev)
    fused.detect_batch_device(padded.data_ptr(), 6, 1080, 1920, 2048, 1080 * 2048, out.data_ptr())
    fused.sync()
    got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
    for k in range(6):
        assert_same_record(got[k], want[k], "padded rows, frame %d" % k)
    assert (fused.debug(5, tk.DBG_MASK).reshape(540, 960) == (oracle.detect(frames[5], state, fs)["binary"] > 0)).all()
    big = tk.synth_frame_host(state, 2, 4320, 7680)[0].copy()
    rng = np.random.RandomState(5)
    for x0 in (1900, 3820, 5750):  # texture across each of the three seams between the four waves of a row (half-size columns 960, 1920, 2880)
        big[:, x0:x0 + 60] = np.clip(rng.normal(70, 35, (4320, 60)), 0, 255).astype(np.uint8)
    _mask_check(fused, oracle, state, fs, big, "8K frame, texture across the wave seams")


def test_fused_sweep_at_any_multiple_of_320_by_5(fused, oracle, dictionary, test_bmp):
    """Round 6: the run-time-band build of k_decimate_mask -- bands of whole threshold-tile rows handed out evenly, a last wave of fewer than 60 lanes --
    takes every frame whose half size is a multiple of 320 x 5: the reference's own test.bmp (1920x1200: four bands of 150 rows), 1280x720 (a 40-lane
    wave), 2560x1440 (a full wave + a 20-lane wave per row, a seam between them), 640x480, 3200x1800 (bands that do not divide evenly: 45 tile rows over..),
    and sizes whose tile rows do not divide by the band count.  Mask bit for bit, every stage behind it, the record."""
    state, fs = dictionary
    rng = np.random.RandomState(31)
    got_bmp = _mask_check(fused, oracle, state, fs, test_bmp, "test.bmp 1920x1200")[1]
    assert got_bmp["n_markers"] == 5
    for (w, h, seam) in ((1280, 720, None), (2560, 1440, 1920), (640, 480, None), (3200, 1800, 1920), (1920, 1210, None), (1280, 50, None), (640, 20, None),
                         (4480, 1090, 3840)):
        img = tk.synth_frame_host(state, 7, h, w)[0].copy() if h >= 400 else np.clip(rng.normal(80, 40, (h, w)), 0, 255).astype(np.uint8)
        if seam:  # texture across the seam between two waves of a row (half-size column 960 k)
            img[:, seam - 40:seam + 40] = np.clip(rng.normal(60, 30, (h, 80)), 0, 255).astype(np.uint8)
        _mask_check(fused, oracle, state, fs, img, "synthetic %dx%d" % (w, h))
        noise = np.clip(rng.normal(70, 40, (h, w)), 0, 255).astype(np.uint8)
        _mask_check(fused, oracle, state, fs, noise, "noise %dx%d" % (w, h))


def test_general_fused_build_on_batches_and_bgr(detector, oracle, dictionary, test_bmp):
    """Batches of 1920x1200 frames through the default rule (a batch takes the fused sweep by itself), gray and device-resident BGR (the direct form: no gray image)."""
    import torch
    from test_gpu_parity import _colourise
    state, fs = dictionary
    n = 520  # 520 frames x 1 wave x 4 bands >= 2048: a batch
    base = np.stack([np.roll(test_bmp, 3 * k, axis=1) for k in range(8)])
    want8, _ = oracle.detect_many(base, state, fs)
    dev = torch.device("cuda:0")
    frames = torch.from_numpy(base).to(dev).repeat(n // 8, 1, 1).contiguous()
    out = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()  # torch filled the frames on ITS stream; the library's streams do not wait for it
    detector.detect_batch_device(frames.data_ptr(), n, 1200, 1920, 1920, 1200 * 1920, out.data_ptr())
    detector.sync()
    got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
    for k in range(n):
        assert_same_record(got[k], want8[k % 8], "1920x1200 batch frame %d" % k)
    assert (detector.debug(n - 1, tk.DBG_MASK).reshape(600, 960) == (oracle.detect(base[(n - 1) % 8], state, fs)["binary"] > 0)).all()  # the fused form ran: a mask exists
    del frames
    bgr8 = np.stack([_colourise(base[k], k) for k in range(8)])
    wantb = [oracle.detect_fast(oracle.bgr2gray(bgr8[k]), state, fs) for k in range(8)]
    m = 64
    bgr = torch.from_numpy(bgr8).to(dev).repeat(m // 8, 1, 1, 1).contiguous()
    outb = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    detector.set_option(capi.OPT_FUSED_SWEEP, 2)
    try:
        detector.detect_batch_bgr_device(bgr.data_ptr(), m, 1200, 1920, 1920 * 3, 1200 * 1920 * 3, outb.data_ptr())
        detector.sync()
    finally:
        detector.set_option(capi.OPT_FUSED_SWEEP, 1)
    got = np.frombuffer(outb.cpu().numpy().tobytes(), ca.RESULT_DT)
    for k in range(m):
        assert_same_record(got[k], wantb[k % 8], "1920x1200 BGR direct frame %d" % k)
    with pytest.raises(ca.CtagError):
        detector.debug(0, tk.DBG_GRAY)  # no gray image exists
