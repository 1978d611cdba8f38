"""GPU parity on cluttered frames (-m gpu): content the reference processes like any other -- corner_detector.cpp:81-107 keeps every
component of 30 px .. 1 % of the half-size frame, :171-405 walks them all; its only hard limits are isVisited[1000] quads,
father[100] features and code[20] (header/corner_detector.h:124,143,152) -- but that needs more than the pools of the batch
workspace (cylindertag_amd/csrc/ctag_api.hip: make_caps: at 1080p 2048 candidates and 262 144 reserved cluster points, scaled
with the frame's area).  Such a frame is run again through the any-frame workspace; its record must be the oracle's, byte for byte,
with flags == 0 -- never CTAG_ERR_LIMIT, never a throw from CylinderTag::detect."""
import numpy as np
import pytest
import torch

import cylindertag_amd as ca
import testkit as tk
from cylindertag_amd import capi
from clutter import blob_field, chevron_texture, long_diagonal, tile_border_specks
from test_gpu_parity import _stage_check, assert_same_record

pytestmark = pytest.mark.gpu


def _reserved(cands):
    """Cluster points the boundary stage reserves for the candidates (k_quad.hip: pack_points + 64 each)."""
    w = cands[:, 4] - cands[:, 2] + 1
    h = cands[:, 5] - cands[:, 3] + 1
    return int((np.minimum(2 * (w + h), w * h) + 65).sum())


def test_thousands_of_blobs_1080p_every_stage(detector, oracle, dictionary):
    """> 2500 non-quad blobs of >= 30 px around four markers at 1080p: more candidates than the batch workspace holds (2048)."""
    state, fs = dictionary
    img, placed = blob_field(tk.synth_frame_host(state, 3)[0])
    before = detector.counters()["reruns"]
    o, r, _ = _stage_check(detector, oracle, state, fs, img, "1080p blob field")  # labels, candidates in OpenCV order, every fitted quad, the record
    assert len(o["candidates"]) >= 2500 and len(o["quads"]) <= 1000
    assert r["status"] == 0 and r["flags"] == 0 and r["n_markers"] == 4
    c = detector.counters()
    assert c["reruns"] == before + 1 and c["candidates"][1] == len(o["candidates"]) and c["quads"][1] == len(o["quads"])


def test_thousands_of_blobs_4k(detector, oracle, dictionary):
    """> 9000 blobs around four markers at 3840x2160: beyond the 8128 candidates a 4K batch workspace holds (and far beyond the 2048
    every frame size was held to before)."""
    state, fs = dictionary
    img, placed = blob_field(tk.synth_frame_host(state, 5, 2160, 3840)[0])
    o, r, _ = _stage_check(detector, oracle, state, fs, img, "4K blob field")
    assert len(o["candidates"]) >= 9000 and len(o["quads"]) <= 1000
    assert r["status"] == 0 and r["flags"] == 0 and r["n_markers"] == 4
    # 6000 blobs -- what used to be three times the cap -- now fit the 4K batch workspace itself: same answer, no second pass
    img6, _ = blob_field(tk.synth_frame_host(state, 5, 2160, 3840)[0], pitch_x=36, pitch_y=32)
    before = detector.counters()["reruns"]
    o6, r6, _ = _stage_check(detector, oracle, state, fs, img6, "4K, 6000 blobs")
    assert 6000 <= len(o6["candidates"]) <= 8128 and r6["flags"] == 0
    assert detector.counters()["reruns"] == before


def test_texture_that_reserves_more_cluster_points_than_the_pool(detector, oracle, dictionary):
    """Nested chevrons around four markers: fewer candidates than the candidate pool, > 262 144 reserved edge-cluster points."""
    state, fs = dictionary
    img, placed = chevron_texture(tk.synth_frame_host(state, 9)[0])
    o, r, _ = _stage_check(detector, oracle, state, fs, img, "chevron texture")
    assert len(o["candidates"]) < 2048 and _reserved(o["candidates"]) > 262144
    assert r["status"] == 0 and r["flags"] == 0 and r["n_markers"] == 4


def test_longest_possible_boundary_4k(detector, oracle, dictionary):
    """A thin band from corner to corner of a 4K frame.  A component's boundary is its silhouette -- first hits from the four sides
    (corner_detector.cpp:197-232) -- so it has at most 2 (w + h) <= 6000 points at half resolution: a boundary of more than 8192
    points cannot exist below 8K frames, and the whole-wave builds hold any that can (k_quad.hip: kWaveWordsMax)."""
    state, fs = dictionary
    img = long_diagonal()
    o, r, _ = _stage_check(detector, oracle, state, fs, img, "4K corner-to-corner band")
    assert o["candidates"].shape[0] == 1 and o["candidates"][0, 7] > 3000  # n_boundary
    assert r["flags"] == 0


def test_cluttered_frames_inside_batches(detector, oracle, dictionary):
    """Cluttered frames among ordinary ones: in a host-memory batch, and in a DEVICE-memory batch where they read CTAG_PENDING until
    the handle's next synchronisation point and are the oracle's records after it."""
    state, fs = dictionary
    frames = np.stack([tk.synth_frame_host(state, f)[0] for f in range(64)])
    frames[17] = blob_field(frames[17])[0]
    frames[40] = chevron_texture(frames[40])[0]
    frames[63] = blob_field(frames[63], seed=9)[0]
    want, _ = oracle.detect_many(frames, state, fs)
    assert (want["flags"] == 0).all() and (want["status"] == 0).all()
    got = detector.detect_batch(frames)
    for k in range(64):
        assert_same_record(got[k], want[k], "host batch, frame %d" % k)
    # device memory: the call returns at once; stream ordering alone shows the three frames as pending, ctag_sync completes them
    dev = torch.device("cuda:0")
    fr = torch.from_numpy(frames).to(dev)
    out = torch.zeros(64 * ca.RESULT_DT.itemsize, dtype=torch.uint8, device=dev)
    before = detector.counters()["reruns"]
    detector.detect_batch_device(fr.data_ptr(), 64, 1080, 1920, 1920, 1920 * 1080, out.data_ptr())
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipStreamSynchronize(C.c_void_p(detector.stream())) == 0  # NOT ctag_sync: what a caller relying on stream order sees
    early = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
    assert sorted(np.nonzero(early["status"] == capi.PENDING)[0]) == [17, 40, 63]
    for k in range(64):
        if k not in (17, 40, 63):
            assert_same_record(early[k], want[k], "device batch before ctag_sync, frame %d" % k)
    detector.sync()
    late = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
    for k in range(64):
        assert_same_record(late[k], want[k], "device batch after ctag_sync, frame %d" % k)
    assert detector.counters()["reruns"] == before + 3
    # BGR frames in device memory: the pending frame is converted again from the caller's colour frame
    bgr = np.repeat(frames[16:19, :, :, None], 3, axis=3)
    bgr_d = torch.from_numpy(np.ascontiguousarray(bgr)).to(dev)
    out3 = torch.zeros(3 * ca.RESULT_DT.itemsize, dtype=torch.uint8, device=dev)
    detector.detect_batch_bgr_device(bgr_d.data_ptr(), 3, 1080, 1920, 1920 * 3, 1920 * 1080 * 3, out3.data_ptr())
    detector.sync()
    got3 = np.frombuffer(out3.cpu().numpy().tobytes(), ca.RESULT_DT)
    for k in range(3):
        assert_same_record(got3[k], want[16 + k], "BGR device batch, frame %d" % (16 + k))


def test_ordinary_frames_after_a_frame_that_exhausted_the_component_pool(detector, oracle, dictionary):
    """Round-4 ADVICE (high): a frame whose second labelling pass runs out of component-pool entries (specks along every label tile's
    border: ~17 000 entries asked of 13 824) is rerun through the any-frame workspace -- but its tiles' labels stay in the BATCH workspace's
    slot, and the slot's tile_dirty bits must say so, or the next frames in that slot skip label blocks that still hold the texture's
    labels.  Batch 1 holds the texture in slot 3 (and a second one in slot 6), batch 2 ordinary marker frames in every slot: every
    record of both batches equals the oracle's, and slot 3's label image of batch 2 is the oracle's partition."""
    state, fs = dictionary
    dev = torch.device("cuda:0")
    specks = tile_border_specks(np.full((1080, 1920), 200, np.uint8))
    first = np.stack([tk.synth_frame_host(state, f)[0] for f in range(8)])
    first[3] = specks
    first[6] = tile_border_specks(np.full((1080, 1920), 170, np.uint8), ink=8)
    second = np.stack([tk.synth_frame_host(state, 20 + f)[0] for f in range(8)])
    want1, _ = oracle.detect_many(first, state, fs)
    want2, _ = oracle.detect_many(second, state, fs)
    assert want1["status"][3] == 1 and (want2["status"] == 0).all() and (want1["flags"] == 0).all()
    out = torch.zeros(8 * ca.RESULT_DT.itemsize, dtype=torch.uint8, device=dev)
    before = detector.counters()["reruns"]
    for rep in range(2):  # twice: the second round's texture lands on slots that held ordinary frames
        fr = torch.from_numpy(first).to(dev)
        detector.detect_batch_device(fr.data_ptr(), 8, 1080, 1920, 1920, 1920 * 1080, out.data_ptr())
        detector.sync()
        got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
        for k in range(8):
            assert_same_record(got[k], want1[k], "round %d, batch with the speck textures, frame %d" % (rep, k))
        fr = torch.from_numpy(second).to(dev)
        detector.detect_batch_device(fr.data_ptr(), 8, 1080, 1920, 1920, 1920 * 1080, out.data_ptr())
        detector.sync()
        got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
        for k in range(8):
            assert_same_record(got[k], want2[k], "round %d, ordinary batch behind the textures, frame %d" % (rep, k))
        for slot in (3, 6):
            o = oracle.detect(second[slot], state, fs)
            lab = detector.debug(slot, tk.DBG_LABELS).reshape(o["labels"].shape)
            assert ((lab != 0) == (o["binary"] > 0)).all(), "stale labels in slot %d" % slot
            pairs = np.unique(np.stack([o["labels"].ravel(), lab.ravel()], 1), axis=0)
            assert len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))
    assert detector.counters()["reruns"] >= before + 4  # the textures really took the any-frame pass (pool exhausted in the batch workspace)


def test_cpp_class_does_not_throw_on_a_cluttered_frame(detector, oracle, dictionary, tmp_path):
    """CylinderTag::detect (the drop-in class) on the 1080p blob field: four markers, no exception (CTAG_ERR_LIMIT used to become a throw
    inside the reference's loop, main.cpp:52-59, which has no try)."""
    import os
    import subprocess
    from ctag_testlib import GOLDEN, ROOT
    state, fs = dictionary
    img, _ = blob_field(tk.synth_frame_host(state, 3)[0])
    path = str(tmp_path / "clutter.bmp")
    _write_bmp8(path, img)
    demo = os.path.join(ROOT, "cylindertag_amd", "_build", "ctag_demo")
    out = subprocess.run([demo, os.path.join(GOLDEN, "CTag_2f12c.marker"), path], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    want = oracle.detect_fast(img, state, fs)
    assert int(want["n_markers"]) == 4 and "markers 4\n" in out.stdout and "error" not in out.stderr


def _write_bmp8(path, img):
    import struct
    h, w = img.shape
    rowbytes = (w + 3) & ~3
    pal = b"".join(struct.pack("<BBBB", i, i, i, 0) for i in range(256))
    data = b"".join(img[y].tobytes() + b"\0" * (rowbytes - w) for y in range(h - 1, -1, -1))
    off = 14 + 40 + 1024
    with open(path, "wb") as f:
        f.write(b"BM" + struct.pack("<IHHI", off + len(data), 0, 0, off))
        f.write(struct.pack("<IiiHHIIiiII", 40, w, h, 1, 8, 0, len(data), 2835, 2835, 256, 0))
        f.write(pal)
        f.write(data)


def test_gather_completes_pending_frames_without_waiting_in_begin(detector, oracle, dictionary):
    """ctag_gather_begin behind a device-memory batch that holds cluttered frames: it must not wait for the detection (the pipelined N > 1
    loop of bench.py enqueues the next step behind it), the ranks' pending counts travel with the packed sizes, and ctag_gather_end --
    on every rank -- completes the frames and packs again: the gathered list is the oracle's, no CTAG_PENDING left (world 1 here)."""
    import time
    state, fs = dictionary
    n = 48
    frames = np.stack([tk.synth_frame_host(state, 100 + f)[0] for f in range(n)])
    frames[11] = blob_field(frames[11])[0]
    frames[30] = chevron_texture(frames[30])[0]
    want, _ = oracle.detect_many(frames, state, fs)
    dev = torch.device("cuda:0")
    fr = torch.from_numpy(frames).to(dev)
    rec = torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    out = torch.full((n, ca.RESULT_DT.itemsize), 0xEE, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    detector.comm_init(capi.comm_unique_id(), 0, 1)
    try:
        # ordinary content first: nothing pending, and begin returns while the detection is still running
        plain = torch.from_numpy(np.ascontiguousarray(np.repeat(frames[:4], 64, axis=0))).to(dev)  # 256 frames: a few ms of GPU work
        prec = torch.zeros((256, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
        pout = torch.zeros_like(prec)
        torch.cuda.synchronize()
        detector.detect_batch_device(plain.data_ptr(), 256, 1080, 1920, 1920, 1920 * 1080, prec.data_ptr())
        detector.gather(prec.data_ptr(), 256, 256, pout.data_ptr())  # warm: workspaces and the gather's buffers allocated (hipMalloc waits for the device)
        detector.sync()
        t0 = time.perf_counter()
        detector.detect_batch_device(plain.data_ptr(), 256, 1080, 1920, 1920, 1920 * 1080, prec.data_ptr())
        t1 = time.perf_counter()
        detector.gather_begin(prec.data_ptr(), 256, 256)
        t2 = time.perf_counter()
        detector.sync()
        t3 = time.perf_counter()
        detector.gather_end(pout.data_ptr())
        detector.gather_wait()
        assert (pout.cpu().numpy() == prec.cpu().numpy()).all()
        assert (t2 - t1) < 0.5 * (t3 - t0), "gather_begin waited for the detection: %.3f ms of a %.3f ms step" % ((t2 - t1) * 1e3, (t3 - t0) * 1e3)
        # cluttered frames in the batch
        before = detector.counters()["reruns"]
        detector.detect_batch_device(fr.data_ptr(), n, 1080, 1920, 1920, 1920 * 1080, rec.data_ptr())
        detector.gather_begin(rec.data_ptr(), n, n)
        detector.gather_end(out.data_ptr())
        detector.gather_wait()
        detector.sync()
        got = np.frombuffer(out.cpu().numpy().tobytes(), ca.RESULT_DT)
        for k in range(n):
            assert_same_record(got[k], want[k], "gathered, frame %d" % k)
        assert detector.counters()["reruns"] == before + 2
    finally:
        detector.comm_destroy()
