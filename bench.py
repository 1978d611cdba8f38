#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X-native CylinderTag detect() front end on BASELINE.json's config 3:
a synthetic batch of 1920x1080 random-stripe frames, resident in HBM, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`, or plainly as
    above -- bench.py then starts that launcher itself as a child process, before anything touches a GPU: launch_ranks)

A "step" is one pass of the whole detect() path (resize -> threshold -> label -> quads -> features -> sub-pixel
refine -> decode) over the job's batch of frames.  Frames are independent, so with N GPUs every rank owns a contiguous
shard of the SAME 4096-frame batch (BASELINE config 4: strong scaling; `--scaling weak` gives every GPU its own 4096)
and the only collective is the final gather of the marker lists: the library's ctag_gather (packed shards, RCCL called
from the C ABI, include/ctag_gather.h), pipelined against the next step's detection.
Prints ONE JSON line on rank 0 (contract in the task statement), with extra objects:
  roofline        the threshold+label sweep (2*W*H algorithmic bytes per frame) against 8 TB/s HBM,
                  timed with HIP events on the library's own stream
  cpu_baseline    the CPU restatement (oracle/, "port") timed on the host cores on a bounded sample of the batch
  parity          the GPU records of the timed configuration compared byte for byte with the oracle's records of the
                  same frames (the ones the cpu_baseline leg computes anyway); a mismatch makes the exit code 1
  results_sha256  hash of the (gathered) result list in frame order: equal for N = 1, 2, 4, 8 in strong scaling
PyTorch is used only for device memory, synchronisation and the torch.distributed bootstrap.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ROWS, COLS = 1080, 1920
ALGO_BYTES_PER_FRAME = 2 * ROWS * COLS  # read W*H u8 once + write (W/2)(H/2) i32 labels once
HBM_PEAK_GBS = 8000.0                   # MI355X_MICROARCH.md: 8.0 TB/s spec
SWEEP_STAGES = ["decimate", "threshold_ccl", "seam_merge", "resolve", "candidates"]


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=4096, help="frames of the job's batch per step (BASELINE config 3: 4096); per GPU with --scaling weak")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the same --frames batch sharded over the GPUs (BASELINE config 4), weak = --frames per GPU")
    ap.add_argument("--chunk", type=int, default=4096, help="frames per pipeline pass (workspace size)")
    ap.add_argument("--markers", type=int, default=4)
    ap.add_argument("--cpu-frames", type=int, default=384, help="sample size of the one-thread CPU baseline (0 = skip CPU legs and parity)")
    ap.add_argument("--cpu-frames-per-thread", type=int, default=128, help="all-cores CPU baseline: frames per hardware thread")
    ap.add_argument("--host-frames", type=int, default=1024,
                    help="frames of the host-memory (PCIe-inclusive) side measurement, 0 = skip; never `value`")
    ap.add_argument("--pose-frames", type=int, default=1024,
                    help="frames of the detect()+estimatePose side measurement on camera content, 0 = skip; never `value`")
    ap.add_argument("--size", default="1920x1080", help="frame size WxH; the headline metric is 1920x1080 (other sizes are side measurements)")
    ap.add_argument("--no-subpix", action="store_true")
    ap.add_argument("--streams", type=int, default=0, choices=[0, 1, 2, 3, 4],
                    help="CTAG_OPT_STREAMS of the timed steps: 2 (the library's default) runs the halves of a chunk on two internal streams; 1 for profiler runs "
                         "(every launch then covers a whole chunk)")
    ap.add_argument("--latency-calls", type=int, default=200, help="single-frame ctag_detect_u8 calls of the latency side measurement, 0 = skip")
    ap.add_argument("--pipelined-steps", type=int, default=6, help="steps of the two-handle side measurement at N = 1, 0 = skip")
    ap.add_argument("--allow-torch-gather", action="store_true",
                    help="N > 1: if the library's RCCL gather (ctag_gather, the C ABI) cannot come up, fall back to torch.distributed's all_gather "
                         "of the fixed records instead of failing.  Without this flag such a run exits non-zero: the line must not hide a gather failure")
    return ap.parse_args()


def device_copy_rate(dev):
    """"also report against a measured device-copy kernel".  A plain device-to-device copy of 2 GiB (torch's copy
    kernel: every byte read once and written once), best of 5 after a warm-up: GB/s of read + written bytes -- what a streaming
    kernel can reach on this GPU, against the 8 TB/s specification the roofline fraction is priced at."""
    import torch
    n = 2 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.fill_(7)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def pipelined_steps_rate(det, state, fs, dev_index, frames_dev, n, chunk, subpix, out_buf, steps):
    """Side measurement (never `value`): the same steps alternating between TWO handles (streams, workspaces), so that the tail of a
    step's kernels -- a few long boundary / Welsch blocks -- overlaps the head of the next step.  This is how `--gpus N > 1` runs its
    steps; the N = 1 line stays on one handle so that its per-kernel HIP-event times are not stretched by a neighbouring stream."""
    import torch
    import testkit as tk
    from cylindertag_amd import capi
    det2 = tk.Detector(state, fs, device=dev_index)
    try:
        det2.set_option(capi.OPT_MAX_CHUNK, chunk)
        out2 = torch.empty_like(out_buf)
        pair = ((det, out_buf), (det2, out2))

        def run(k):
            d, o = pair[k % 2]
            d.detect_batch_device(frames_dev.data_ptr(), n, ROWS, COLS, COLS, ROWS * COLS, o.data_ptr(), 5, subpix, 5)
        for k in range(2):
            run(k)
        det.sync(); det2.sync()
        t0 = time.perf_counter()
        for k in range(steps):
            run(k)
        det.sync(); det2.sync()
        dt = time.perf_counter() - t0
        same = bool(torch.equal(out_buf[:n], out2[:n]))
        return {"value": round(n * steps / dt, 1), "unit": "frames/s", "handles": 2, "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
                "records_equal_between_handles": same,
                "note": "consecutive steps alternate between two handles (each running its chunks on two internal streams); the headline `value` is one handle"}
    finally:
        det2.close()


def host_stream_rate(det, frames_dev, m, subpix):
    """Side measurement (never `value`): the same frames handed over as HOST buffers through ctag_detect_batch_u8 --
    pinned memory, uploads double-buffered against detection, results downloaded.  Bounded by PCIe."""
    import cylindertag_amd as ca
    import testkit as tk
    host = ca.pinned_empty((m, ROWS, COLS), np.uint8)
    host[...] = frames_dev[:m].cpu().numpy()
    res = ca.pinned_empty((m,), ca.RESULT_DT)
    det.detect_batch(host, 5, subpix, 5, out=res)  # warm (allocates the slabs)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_batch(host, 5, subpix, 5, out=res)
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(m / dt, 1), "unit": "frames/s", "frames": m, "host_gb_per_s": round(m * ROWS * COLS / dt / 1e9, 2),
            "note": "ctag_detect_batch_u8: pinned host frames in, host results out, upload of sub-chunk k+1 overlapped "
                    "with detection of sub-chunk k"}


def host_stream_rate_bgr(det, frames_dev, m, subpix):
    """Side measurement (never `value`): BGR camera frames (main.cpp:52-54 hands cvtColor(BGR2GRAY) a colour frame) as pinned HOST
    buffers through ctag_detect_batch_bgr8 -- 3 bytes per pixel over PCIe, converted on the device."""
    import cylindertag_amd as ca
    host = ca.pinned_empty((m, ROWS, COLS, 3), np.uint8)
    g = frames_dev[:m].cpu().numpy()
    for c in range(3):
        host[..., c] = g  # gray-valued BGR: the converted image is the gray batch itself (1868 + 9617 + 4899 = 16384)
    del g
    res = ca.pinned_empty((m,), ca.RESULT_DT)
    det.detect_batch_bgr(host, 5, subpix, 5, out=res)  # warm
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_batch_bgr(host, 5, subpix, 5, out=res)
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(m / dt, 1), "unit": "frames/s", "frames": m, "host_gb_per_s": round(m * ROWS * COLS * 3 / dt / 1e9, 2),
            "frames_ok": int((res["status"] == 0).sum()),
            "note": "ctag_detect_batch_bgr8: pinned host BGR frames in (3 B/px), BGR2GRAY on the device, host results out"}


def bgr_resident_rate(det, frames_dev, m, subpix, gray_rate):
    """Side measurement (never `value`): the same frames as device-resident BGR (main.cpp:52-54 hands cvtColor(BGR2GRAY) a colour frame) through
    ctag_detect_batch_bgr8_device -- three bytes per pixel from HBM; frames of this size take the direct form (the decimation kernel and edgeRefine
    convert as they load, no gray image is written: CTAG_OPT_BGR_DIRECT)."""
    import torch
    import cylindertag_amd as ca
    g = frames_dev[:m]
    bgr = torch.empty((m, ROWS, COLS, 3), dtype=torch.uint8, device=g.device)
    for c in range(3):
        bgr[..., c] = g  # gray-valued BGR: the converted image is the gray batch itself (1868 + 9617 + 4899 = 16384)
    torch.cuda.synchronize()
    out = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=g.device)
    ref = torch.zeros_like(out)
    det.detect_batch_device(g.data_ptr(), m, ROWS, COLS, COLS, ROWS * COLS, ref.data_ptr(), 5, subpix, 5)
    det.detect_batch_bgr_device(bgr.data_ptr(), m, ROWS, COLS, COLS * 3, ROWS * COLS * 3, out.data_ptr(), 5, subpix, 5)
    det.sync()
    reps = 5
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_batch_device(g.data_ptr(), m, ROWS, COLS, COLS, ROWS * COLS, ref.data_ptr(), 5, subpix, 5)
    det.sync()
    tg = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps):
        det.detect_batch_bgr_device(bgr.data_ptr(), m, ROWS, COLS, COLS * 3, ROWS * COLS * 3, out.data_ptr(), 5, subpix, 5)
    det.sync()
    tb = (time.perf_counter() - t0) / reps
    # one device-resident frame per call: a call of a few BGR frames converts first and takes the short-band kernels, like a gray call of that size
    def one_frame(fn):
        ts = []
        for i in range(60):
            t1 = time.perf_counter()
            fn()
            det.sync()
            ts.append(time.perf_counter() - t1)
        return round(float(np.median(ts[10:])) * 1e3, 4)
    lat = {"gray_ms": one_frame(lambda: det.detect_batch_device(g.data_ptr(), 1, ROWS, COLS, COLS, ROWS * COLS, ref.data_ptr(), 5, subpix, 5)),
           "bgr_ms": one_frame(lambda: det.detect_batch_bgr_device(bgr.data_ptr(), 1, ROWS, COLS, COLS * 3, ROWS * COLS * 3, out.data_ptr(), 5, subpix, 5)),
           "note": "one device-resident frame per call + sync, median of 50"}
    det.detect_batch_device(g.data_ptr(), m, ROWS, COLS, COLS, ROWS * COLS, ref.data_ptr(), 5, subpix, 5)
    det.detect_batch_bgr_device(bgr.data_ptr(), m, ROWS, COLS, COLS * 3, ROWS * COLS * 3, out.data_ptr(), 5, subpix, 5)
    det.sync()
    return {"value": round(m / tb, 1), "unit": "frames/s", "frames": m, "gray_same_frames": round(m / tg, 1), "ratio_to_gray": round(tg / tb, 4),
            "records_equal_gray_path": bool(torch.equal(out, ref)), "one_frame_latency": lat,
            "note": "ctag_detect_batch_bgr8_device on device-resident BGR frames (3 B/px read from HBM; a 1080p frame's 6.2 MB against 2.1 MB of gray: "
                    "the bytes alone bound the ratio near 0.85)"}


def pose_side(det, m, dev):
    """Side measurement (never `value`; SURVEY.md 8(f) rank 2 / BASELINE config 5's estimatePose leg): camera content --
    the 64-frame sequence derived from the reference's test.bmp (config 2's test.avi substitute, 5 physical markers in
    view) tiled to m frames in HBM -- through detect() and then the GPU pose back end (EPnP + LM per marker against the
    reference's CTag_2f12c.model / cameraParams.yml), everything device-resident.  The CPU pose oracle is timed beside it."""
    import torch
    import cylindertag_amd as ca
    import testkit as tk
    from cylindertag_amd import capi
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ctag_testlib import GOLDEN, read_bmp_gray
    from pose_testlib import PoseOracle, make_camera, make_model_view, read_camera_yml, read_model_file
    from sequences import avi_substitute
    seq = avi_substitute(read_bmp_gray(os.path.join(GOLDEN, "test.bmp")))
    m = max(len(seq), m // len(seq) * len(seq))
    rows, cols = seq.shape[1:]
    frames = torch.from_numpy(np.concatenate([seq] * (m // len(seq)))).to(dev)
    model = ca.Model(os.path.join(GOLDEN, "CTag_2f12c.model"))
    cam = ca.load_camera(os.path.join(GOLDEN, "cameraParams.yml"))
    res = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    off = torch.zeros(m + 1, dtype=torch.int32, device=dev)
    cap = m * 8
    poses = torch.zeros(cap * ca.POSE_DT.itemsize, dtype=torch.uint8, device=dev)

    def run():
        det.detect_batch_device(frames.data_ptr(), m, rows, cols, cols, rows * cols, res.data_ptr(), 5, True, 5)
        det.pose_batch_device(res.data_ptr(), m, model, cam, off.data_ptr(), poses.data_ptr(), cap)

    run()
    det.sync()
    det.set_option(capi.OPT_TIMING, 1)
    reps = 3
    t0 = time.perf_counter()
    pose_ms = 0.0
    for _ in range(reps):
        run()
        pose_ms += det.pose_last_ms()
    det.sync()
    dt = (time.perf_counter() - t0) / reps
    det.set_option(capi.OPT_TIMING, 0)
    offs = off.cpu().numpy()
    P = poses.cpu().numpy().view(ca.POSE_DT)[:offs[-1]]
    ok = P[P["status"] == 0]
    rms = np.sqrt(2 * ok["cost"] / np.maximum(ok["n_points"], 1))
    # CPU pose oracle on the records of the first 64 frames (test infrastructure used as the timed baseline only)
    K, dist = read_camera_yml(os.path.join(GOLDEN, "cameraParams.yml"))
    mv = make_model_view(read_model_file(os.path.join(GOLDEN, "CTag_2f12c.model")))
    po, cam_o = PoseOracle(), make_camera(K, dist)
    recs = np.frombuffer(res[:64].cpu().numpy().tobytes(), dtype=ca.RESULT_DT)
    t0 = time.perf_counter()
    ncpu = sum(len(po.pose_frame(r, mv, cam_o, i)) for i, r in enumerate(recs))
    cpu_dt = time.perf_counter() - t0
    return {"workload": "test.bmp-derived %d-frame sequence tiled to %d frames of %dx%d in HBM; detect(img,5,true,5) then "
                        "estimatePose (EPnP + LM) with CTag_2f12c.model / cameraParams.yml" % (len(seq), m, cols, rows),
            "detect_plus_pose_frames_per_s": round(m / dt, 1), "pose_kernel_ms": round(pose_ms / reps, 3),
            "markers": int(offs[-1]), "poses_ok": int(len(ok)), "pose_markers_per_s": round(offs[-1] / (pose_ms / reps * 1e-3), 1),
            "reprojection_rms_px_median": round(float(np.median(rms)), 4), "lm_iterations_mean": round(float(ok["iterations"].mean()), 2),
            "cpu_pose_oracle_markers_per_s": round(ncpu / cpu_dt, 1), "cpu_cores": 1}


def pose_side_3d(det, state, m, dev, frames_cap):
    """Side measurement for frame sizes other than 1080p (BASELINE config 5: 3840x2160): ray-cast cylinders with PLANTED
    poses (ctag_synth3d_*: printed strips on cylinders, pinhole camera) -> detect(img,5,true,5) -> estimatePose on the GPU
    with the objects' 3-D corner lists, everything device-resident; the recovered poses are compared with the planted ones."""
    import torch
    import cylindertag_amd as ca
    import testkit as tk
    from cylindertag_amd import capi
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from pose_testlib import PoseOracle, make_camera, make_model_view, rodrigues
    m = max(1, min(m, frames_cap))
    f = 5200.0 * COLS / 3840.0
    K = np.array([[f, 0, COLS / 2.0], [0, f, ROWS / 2.0], [0, 0, 1]])
    model, corners = tk.synth3d_model(state)
    cam = ca.make_camera(K, np.zeros(5))
    frames = torch.empty((m, ROWS, COLS), dtype=torch.uint8, device=dev)
    det.synth3d_frames_device(frames.data_ptr(), 0, m, ROWS, COLS, COLS, ROWS * COLS, K)
    res = torch.zeros((m, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    off = torch.zeros(m + 1, dtype=torch.int32, device=dev)
    cap = m * 8
    poses = torch.zeros(cap * ca.POSE_DT.itemsize, dtype=torch.uint8, device=dev)

    def run():
        det.detect_batch_device(frames.data_ptr(), m, ROWS, COLS, COLS, ROWS * COLS, res.data_ptr(), 5, True, 5)
        det.pose_batch_device(res.data_ptr(), m, model, cam, off.data_ptr(), poses.data_ptr(), cap)

    run()
    det.sync()
    det.set_option(capi.OPT_TIMING, 1)
    reps, pose_ms = 3, 0.0
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
        pose_ms += det.pose_last_ms()
    det.sync()
    dt = (time.perf_counter() - t0) / reps
    det.set_option(capi.OPT_TIMING, 0)
    offs = off.cpu().numpy()
    P = poses.cpu().numpy().view(ca.POSE_DT)[:offs[-1]]
    recs = np.frombuffer(res.cpu().numpy().tobytes(), dtype=ca.RESULT_DT)
    ok = P[P["status"] == 0]
    # planted poses of the first frames (the host layout: no rendering needed beyond a 1-row image)
    ang, rel = [], []
    for fr in range(min(m, 64)):
        truth = tk.synth3d_frame_host(state, fr, K, rows=ROWS, cols=COLS)[1] if fr < 2 else None
        if truth is None:
            break
        for p in P[offs[fr]:offs[fr + 1]]:
            if p["status"] != 0:
                continue
            k = [i for i in range(truth["n_markers"]) if truth["dict_row"][i] == p["model_index"]]
            if not k:
                continue
            R, Rt = rodrigues(p["rvec"]), truth["R"][k[0]].reshape(3, 3)
            ang.append(float(np.degrees(np.arccos(np.clip((np.trace(R.T @ Rt) - 1) / 2, -1, 1)))))
            rel.append(float(np.linalg.norm(p["tvec"] - truth["t"][k[0]]) / np.linalg.norm(truth["t"][k[0]])))
    ids = np.arange(state.shape[0], dtype=np.int32)
    mv = make_model_view({"ids": ids, "size": state.shape[1], "base": np.zeros((len(ids), 3), np.float32),
                          "axis": np.zeros((len(ids), 3), np.float32), "corners": corners})
    po, cam_o = PoseOracle(), make_camera(K, np.zeros(5))
    t0 = time.perf_counter()
    want = [po.pose_frame(r, mv, cam_o, i) for i, r in enumerate(recs[:min(m, 64)])]
    cpu_dt = time.perf_counter() - t0
    ncpu = sum(len(w) for w in want)
    same = np.concatenate(want).tobytes() == P[:ncpu].tobytes() if ncpu else True
    rms = np.sqrt(2 * ok["cost"] / np.maximum(ok["n_points"], 1))
    return {"workload": "%d ray-cast %dx%d frames of 4 cylinders with planted poses in HBM; detect(img,5,true,5) then estimatePose (EPnP + LM) "
                        "with the objects' 3-D corner lists" % (m, COLS, ROWS),
            "detect_plus_pose_frames_per_s": round(m / dt, 1), "pose_kernel_ms": round(pose_ms / reps, 3), "markers": int(offs[-1]),
            "poses_ok": int(len(ok)), "reprojection_rms_px_median": round(float(np.median(rms)), 4) if len(ok) else None,
            "planted_pose_error": {"poses_compared": len(ang), "rotation_deg_max": round(max(ang), 4) if ang else None,
                                   "translation_rel_max": round(max(rel), 6) if rel else None},
            "gpu_poses_equal_cpu_pose_oracle": bool(same), "cpu_pose_oracle_markers_per_s": round(ncpu / cpu_dt, 1) if cpu_dt > 0 else None,
            "cpu_cores": 1}


ISSUE_KERNELS = {"edge_refine": "ctag::k_edge_refine", "welsch": "ctag::k_welsch", "quad_edges": "ctag::k_quad_edges_packed",
                 "threshold_ccl": "ctag::k_threshold_ccl<5"}


def issue_rooflines(stage_ms, n_frames):
    """Vector-instruction ISSUE roofline of the kernels that are not memory-bound (86 % of the step): wave-instructions of the
    kernel -- counted by rocprofv3 PMC passes, replayed from profiles/r*_pmc_instmix.json like roofline.traffic -- priced at
    the issue cost MEASURED per instruction class on this GPU (tools/ubench/valu_rate.hip, profiles/r03_valu_issue_cost.txt: 2.1
    SIMD cycles for f32 add / mul and simple integer / logic instructions, 4.2 for the rest -- fma, min / max, conversions, FP64
    add / mul / fma --, 8.2 / 16.1 for transcendentals; tools/pmc_instmix.py) on 1024 SIMDs at 2.4 GHz, against the kernel's
    time measured in THIS run."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_instmix.json")))
    if not cands or (ROWS, COLS) != (1080, 1920):
        return None
    prof = json.load(open(cands[-1]))
    out = {"source": "instruction counts replayed from %s (rocprofv3 --pmc, 1024-frame pass); times from this run's HIP events" % os.path.relpath(cands[-1], ROOT),
           "model": "issue_bound_ms = sum over instruction classes of count * measured SIMD cycles per wave-instruction (2.1 f32 add/mul and simple int, 4.2 others incl. FP64, 8.2 / 16.1 transcendental; int32 at 3.15) / (1024 SIMDs * 2.4 GHz); profiles/r03_valu_issue_cost.txt", "kernels": {}}
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import srcsha
        psha = (prof.get("sources_sha256") or {}).get("all")
        out["profile"] = {"file": os.path.relpath(cands[-1], ROOT), "git_head": prof.get("git_head"), "sources_sha256": psha,
                          "matches_this_tree": None if psha is None else psha == srcsha.sources_sha256()["all"]}
    except Exception:  # noqa: BLE001
        pass
    for stage, kname in ISSUE_KERNELS.items():
        hits = [v for k, v in prof["kernels"].items() if k == kname or k.startswith(kname + "<") or (kname.endswith("<5") and k.startswith(kname))]  # template arguments vary; several builds of a kernel
        if stage == "quad_edges":  # the packed build runs as two kernels (boundary, edge clusters) behind the mask-based silhouette scan (round 6): all belong to the stage
            hits = [v for k, v in prof["kernels"].items() if k.startswith(kname + "<8,") or k.startswith("ctag::k_silhouette_mask")]
            work = hits
        elif stage == "edge_refine":  # searches (k_edge_refine<1>), ordered sums (k_edge_refine_sums), lines and corners (k_edge_refine_tail), long quads: the stage is all of them
            work = [v for k, v in prof["kernels"].items() if k.startswith(kname + "<1") or k.startswith(kname + "_")]
        else:
            work = [max(hits, key=lambda v: v["wave_instructions"]["valu"])] if hits else []  # the build that does the work
        if not work or stage_ms.get(stage, 0) <= 0:
            continue
        cycles = sum(e["issue_model"]["issue_cycles"] for e in work)
        valu = sum(e["wave_instructions"]["valu"] for e in work)
        fp64 = sum(e["fp64_share_of_valu"] * e["wave_instructions"]["valu"] for e in work) / max(valu, 1)
        bound_ms = cycles / 1024.0 * n_frames / (1024 * 2.4e9) * 1e3
        out["kernels"]["k_" + stage if not stage.startswith("quad") else "k_quad_edges_packed"] = {
            "achieved_ms": round(stage_ms[stage], 3), "issue_bound_ms": round(bound_ms, 3), "frac": round(bound_ms / stage_ms[stage], 4),
            "valu_wave_instructions_per_frame": round(valu / 1024.0, 1), "fp64_share_of_valu": round(fp64, 4)}
    return out


def _native_oracle():
    """The timed baseline build of the oracle (-O3 -march=native -ffp-contract=off), compiled ON THIS HOST
    (native code must not travel between machines); falls back to the portable -O2 checker build when g++ is missing."""
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ctag_testlib import Oracle  # the oracle is test infrastructure: used here only as the timed CPU baseline / checker
    path = os.path.join(ROOT, "oracle", "_native", "libctag_oracle.so")
    try:
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
        return Oracle(path=path), "-O3 -march=native -ffp-contract=off, built on this host"
    except Exception as e:  # noqa: BLE001
        return Oracle(), "-O2 -ffp-contract=off (portable build; native build failed: %s)" % type(e).__name__


def _cpu_quota():
    """CPUs the container may actually use: cgroup v2 cpu.max / v1 cfs quota (the GPU box grants 16 of its host's 256
    hardware threads; threads beyond the quota only get throttled).  None = unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except Exception:  # noqa: BLE001
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except Exception:  # noqa: BLE001
        return None


def cpu_baseline(frames_dev, n_one, per_thread, state, fs, subpix):
    """Times the CPU restatement of detect() (oracle, kind 'port') on this host: one thread on the first n_one frames, then
    frame-parallel on every hardware thread (std::thread pool inside the oracle library, >= per_thread frames per
    thread).  Returns (json object, oracle records of the frames it ran) -- the records feed the parity check."""
    orc, build = _native_oracle()
    hw = int(orc.L.ctago_hardware_concurrency()) or (os.cpu_count() or 1)
    hw = min(hw, len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else hw
    quota = _cpu_quota()
    threads = max(1, min(hw, int(quota + 0.5))) if quota else hw  # more threads than granted CPUs only get throttled
    n_all = min(int(frames_dev.shape[0]), max(n_one, per_thread * threads))
    frames_host = frames_dev[:n_all].cpu().numpy()
    n_one = min(n_one, n_all)
    orc.detect_fast(frames_host[0], state, fs, 5, subpix, 5)  # warm
    t0 = time.perf_counter()
    rec_one, _ = orc.detect_many(frames_host[:n_one], state, fs, 5, subpix, 5, threads=1)
    dt = time.perf_counter() - t0
    out = {"value": n_one / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "first %d frames of the same synthetic batch, CPU restatement of detect() "
                     "(oracle/ctag_oracle.cpp, %s), 1 thread, %.1f s" % (n_one, build, dt)}
    records = rec_one
    if threads > 1:
        runs = []
        for _ in range(2):  # twice: the figure has to be stable
            t0 = time.perf_counter()
            rec_all, used = orc.detect_many(frames_host, state, fs, 5, subpix, 5, threads=threads)
            runs.append(n_all / (time.perf_counter() - t0))
        if rec_all[:n_one].tobytes() != rec_one.tobytes():
            raise AssertionError("frame-parallel oracle records differ from the one-thread ones")
        out["all_cores"] = {"value": max(runs), "unit": "frames/s", "cores": used, "hardware_concurrency": hw,
                            "cgroup_cpu_quota": quota,
                            "runs_frames_per_s": [round(r, 1) for r in runs],
                            "sample": "first %d frames (%.1f per thread), ctago_detect_many: std::thread pool inside the oracle library"
                                      % (n_all, n_all / used)}
        records = rec_all
    return out, records


def opencv_stage_probe(frames_host):
    """BASELINE.md's optional stage-level sanity baseline: if an OpenCV build is importable on this host, (1) COMPARE its
    primitives with the oracle's replicas (tests/cv2_pins.py: resize INTER_CUBIC, connectedComponentsWithStats BBDT label order,
    fitLine L2 / Welsch, fastAtan2 -- the [OCV-recall] items of SURVEY App. A) and print the mismatches, (2) time its own
    resize(INTER_CUBIC, 1/2) + connectedComponentsWithStats(8, CCL_BBDT) -- the two OpenCV primitives of the sweep
    (CylinderTag.cpp:79, corner_detector.cpp:82) -- on a few frames of the batch and say which version; otherwise say so.
    (This image and the GPU boxes of this pool have no OpenCV: the probe reports its absence.)"""
    try:
        import cv2
    except Exception as e:  # noqa: BLE001
        return {"available": False, "note": "no OpenCV importable on this host (%s): stage-level OpenCV baseline not measured, oracle primitives "
                                            "not compared (tests/cv2_pins.py)" % type(e).__name__}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ctag_testlib import GOLDEN, Oracle, read_bmp_gray
    import cv2_pins
    try:
        pins = cv2_pins.run_all(Oracle(), read_bmp_gray(os.path.join(GOLDEN, "test.bmp")))
    except Exception as e:  # noqa: BLE001
        pins = {"error": "%s: %s" % (type(e).__name__, e)}
    n = min(16, len(frames_host))
    t0 = time.perf_counter()
    for f in frames_host[:n]:
        half = cv2.resize(f, (f.shape[1] // 2, f.shape[0] // 2), None, 0.5, 0.5, cv2.INTER_CUBIC)
        binary = ((half.astype(np.float32) * np.float32(1.0 / 255)) < 0.3).astype(np.uint8) * 255  # stand-in mask: the threshold is the reference's own code
        cv2.connectedComponentsWithStatsWithAlgorithm(binary, 8, cv2.CV_32S, cv2.CCL_BBDT)
    dt = time.perf_counter() - t0
    return {"available": True, "version": cv2.__version__, "threads": cv2.getNumThreads(), "frames": n,
            "resize_plus_ccl_frames_per_s": round(n / dt, 1), "note": "OpenCV resize INTER_CUBIC + connectedComponentsWithStats(8, CCL_BBDT) only",
            "oracle_vs_opencv_primitives": pins}


def latency_side(det, state, fs, calls=200):
    """Side measurement (never `value`): the reference's actual use -- one frame per call (main.cpp:52-59) -- on the
    reference's test.bmp through ctag_detect_u8 (host frame in, host record out)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ctag_testlib import GOLDEN, read_bmp_gray
    img = read_bmp_gray(os.path.join(GOLDEN, "test.bmp"))
    for _ in range(20):
        det.detect(img, 5, True, 5)
    ts = []
    for _ in range(calls):
        t0 = time.perf_counter()
        det.detect(img, 5, True, 5)
        ts.append(time.perf_counter() - t0)
    ts = np.sort(np.array(ts)) * 1e3
    out = {"workload": "test.bmp 1920x1200, ctag_detect_u8(img,5,true,5), pageable host frame in, host record out, %d calls" % calls,
           "latency_ms_median": round(float(ts[len(ts) // 2]), 4), "latency_ms_p10": round(float(ts[len(ts) // 10]), 4),
           "latency_ms_p90": round(float(ts[len(ts) * 9 // 10]), 4)}
    # the same loop with two frames in flight (ctag_submit_u8 / ctag_collect: frame k + 1 uploads behind frame k's detection), pinned frames
    import cylindertag_amd as ca
    ring = ca.pinned_empty((2,) + img.shape, np.uint8)
    ring[0] = img
    ring[1] = img
    det.submit(ring[0])
    for k in range(20):
        det.submit(ring[(k + 1) % 2])
        det.collect()
    t0 = time.perf_counter()
    for k in range(calls):
        det.submit(ring[(k + 1) % 2])
        det.collect()
    dt = time.perf_counter() - t0
    det.collect()
    out["submit_collect_loop"] = {"frames_per_s": round(calls / dt, 1), "ms_per_frame": round(dt / calls * 1e3, 4), "frames_in_flight": 2,
                                  "note": "one frame per call, sustained: submit frame k + 1, collect frame k (pinned host frames)"}
    return out


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start the N ranks -- `python -m torch.distributed.run --nnodes=1
    --nproc-per-node N bench.py <the same arguments>`, one process per GPU -- as a CHILD of this process, which has not imported
    torch and never touches a GPU (a process that has initialised one must not exec another program on this pool).  Rank 0's JSON
    line goes to the inherited stdout; the launcher's exit code (non-zero as soon as one rank fails, and the launcher then ends the
    other ranks) is this process's.  The launcher stays in this process's group and is tied to its life, so nothing outlives a killed run."""
    import signal
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    def tie_to_parent():  # the launcher gets SIGTERM when this process goes away, however that happens (PR_SET_PDEATHSIG): no ranks left behind on the GPUs
        try:
            import ctypes
            ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, int(signal.SIGTERM), 0, 0, 0)
        except Exception:  # noqa: BLE001
            pass
    # (same session and process group as this process: whoever ends the group -- a driver's timeout -- ends the ranks too)
    child = subprocess.Popen(cmd, env=env, preexec_fn=tie_to_parent)
    def forward(sig, _frame):
        try:
            child.send_signal(sig)  # torch.distributed.run hands it on to its workers
        except ProcessLookupError:
            pass
    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, forward)
    rc = child.wait()
    return rc if rc >= 0 else 128 - rc


def main():
    global ROWS, COLS, ALGO_BYTES_PER_FRAME
    args = parse_args()
    COLS, ROWS = (int(v) for v in args.size.lower().split("x"))
    ALGO_BYTES_PER_FRAME = 2 * ROWS * COLS
    import hashlib
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus)  # before torch is imported: this process never touches a GPU
    import torch
    import cylindertag_amd as ca
    import testkit as tk
    from cylindertag_amd import capi
    from cylindertag_amd.dist import CommGather, shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d inside a job of WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the detection path has no CPU fallback")
    dev_index = local_rank % max(1, torch.cuda.device_count())  # identity on a node with one GPU per rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl == RCCL on ROCm.  CTAG_BENCH_BACKEND=gloo is a developer aid: it lets N ranks share ONE GPU (RCCL refuses that),
        # which exercises the sharding / double-buffering / hashing logic of this file; the gather then goes through host memory
        backend = os.environ.get("CTAG_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=backend)

    state, fs = ca.load_marker_file(os.path.join(ROOT, "tests", "golden", "CTag_2f12c.marker"))
    det = tk.Detector(state, fs, device=dev_index)
    if args.streams:  # 0: the library's default
        det.set_option(capi.OPT_STREAMS, args.streams)
    subpix = not args.no_subpix

    # ---- the job: n_total frames per step; this rank owns frames [lo, hi) of it
    strong = args.scaling == "strong" or world == 1
    n_total = args.frames if strong else args.frames * world
    lo, hi = shard_range(n_total, rank, world)
    n = hi - lo
    chunk = max(1, min(args.chunk, n))
    det.set_option(capi.OPT_MAX_CHUNK, chunk)
    frames = torch.empty((max(n, 1), ROWS, COLS), dtype=torch.uint8, device=dev)
    det.synth_frames_device(frames.data_ptr(), lo, n, ROWS, COLS, COLS, ROWS * COLS, markers=args.markers)
    rec_bytes = ca.RESULT_DT.itemsize
    local_bufs = [torch.zeros((max(n, 1), rec_bytes), dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    gather_impl, comm, gathered = "none (1 GPU)", None, None
    if world > 1:
        # the path's only exchange: final marker lists, through the library's own RCCL gather (packed shards, C ABI).  Should its
        # communicator fail to come up the run FAILS -- unless --allow-torch-gather (or the gloo developer backend, which has no
        # RCCL at all) asks for torch.distributed's all_gather of the fixed records, and then the line says so.
        gathered = [torch.zeros((n_total, rec_bytes), dtype=torch.uint8, device=dev) for _ in range(2)]
        host_gather = dist.get_backend() != "nccl"
        try:
            if host_gather:
                raise RuntimeError("backend %s: no RCCL communicator" % dist.get_backend())
            comm = CommGather(det, dist)
            gather_impl = "ctag_gather (C ABI -> ncclAllGather of packed shards)"
        except Exception as e:  # noqa: BLE001
            comm = None
            gather_impl = "torch.distributed.all_gather_into_tensor of fixed records (ctag_comm_init failed: %s)" % e
        flags = torch.tensor([1 if comm is not None else 0], dtype=torch.int32, device="cpu" if host_gather else dev)
        dist.all_reduce(flags, op=dist.ReduceOp.MIN)
        if int(flags.item()) == 0 and comm is not None:  # some rank fell back: all ranks must take the same path
            comm.close()
            comm = None
            gather_impl = "torch.distributed.all_gather_into_tensor of fixed records (another rank's ctag_comm_init failed)"
        if comm is None and not host_gather and not args.allow_torch_gather:
            if rank == 0:
                print("bench.py: the C-ABI gather is not the path in use: %s (pass --allow-torch-gather to measure the fallback)" % gather_impl,
                      file=sys.stderr, flush=True)
            dist.barrier()
            dist.destroy_process_group()
            return 3
        if host_gather:
            gather_impl = ("developer backend %s (ranks share a GPU; RCCL refuses that): ctag_pack_results kernel -> %s all_gather of the packed sizes and "
                           "shards through host memory -> ctag_gather_end's unpack kernels" % (dist.get_backend(), dist.get_backend()))
        elif comm is None and n * world != n_total:
            raise SystemExit("the torch.distributed fallback gather needs equal shards")
    # N > 1: consecutive steps alternate between two handles (two streams, two workspaces): a rank's shard is small (512 frames
    # of the 4096 at N = 8) and the tails of its kernels -- a few long boundary / Welsch blocks -- would otherwise idle most of
    # the GPU at the end of every kernel; the next step's kernels fill them (112 -> 149 K frames/s per GPU at 512
    # frames per step).  Both handles gather through ONE communicator (the second attaches to the first's: collectives of a
    # communicator run in issue order, so no rank can start two communicators' kernels in a different order than its peers).
    # CTAG_BENCH_PINGPONG=0 turns it off.
    dets, comms = [det], [comm]
    if world > 1 and os.environ.get("CTAG_BENCH_PINGPONG", "1") != "0":
        det2, comm2 = None, None
        try:
            det2 = tk.Detector(state, fs, device=dev_index)
            det2.set_option(capi.OPT_MAX_CHUNK, chunk)
            if comm is not None:
                comm2 = CommGather(det2, dist, share=comm)
        except Exception:  # noqa: BLE001
            det2 = None
        ok2 = torch.tensor([1 if det2 is not None else 0], dtype=torch.int32, device="cpu" if dist.get_backend() != "nccl" else dev)
        dist.all_reduce(ok2, op=dist.ReduceOp.MIN)  # all ranks take the same path
        if int(ok2.item()) == 1:
            dets.append(det2)
            comms.append(comm2)
        else:
            if comm2 is not None:
                comm2.close()
            if det2 is not None:
                det2.close()
    step_no = [0]
    pending = []  # gathers begun and not yet ended: (index of the gathered buffer)

    def finish_pending():
        while pending:
            k = pending.pop(0)
            if comm is not None:
                comms[k % len(comms)].end(gathered[k % 2])

    def step():
        k = step_no[0]
        step_no[0] += 1
        buf = local_bufs[k % len(local_bufs)]
        d = dets[k % len(dets)]
        d.detect_batch_device(frames.data_ptr(), n, ROWS, COLS, COLS, ROWS * COLS, buf.data_ptr(), 5, subpix, 5)
        if world > 1:
            if comm is not None:
                finish_pending()            # sizes of step k-1 are in (its detection ended while step k was being enqueued)
                comms[k % len(comms)].begin(buf[:n], n_total)  # pack + size exchange of step k behind its detection, on its handle's gather stream
                pending.append(k)
            else:
                d.sync()
                if host_gather:
                    host_packed_gather(d, buf, gathered[k % 2])
                else:
                    dist.all_gather_into_tensor(gathered[k % 2].view(-1), buf[:n].reshape(-1))

    host_stats = {}
    packed_dev = [None]

    def host_packed_gather(d, buf, out):
        """Developer backend (gloo: N ranks on ONE GPU, which RCCL refuses): the library's own protocol with the two ncclAllGather calls
        replaced by gloo collectives on host copies -- ctag_pack_results kernel -> sizes -> packed shards padded to the largest ->
        ctag_gather_end's segment table + unpack kernels (testkit hook on a caller-built buffer).  Uneven shards included."""
        cap = capi.packed_capacity(max(n, 1))
        if packed_dev[0] is None:
            packed_dev[0] = torch.empty(cap, dtype=torch.uint8, device=dev)
        nbytes = d.pack_results(buf.data_ptr(), n, packed_dev[0].data_ptr(), cap)  # waits for the size
        sizes = torch.zeros(world, dtype=torch.int64)
        dist.all_gather_into_tensor(sizes, torch.tensor([nbytes], dtype=torch.int64))
        width = (int(sizes.max()) + 255) & ~255
        send = torch.zeros(width, dtype=torch.uint8)
        send[:nbytes] = packed_dev[0][:nbytes].cpu()
        recv = torch.empty(world * width, dtype=torch.uint8)
        dist.all_gather_into_tensor(recv, send)
        recv_dev = recv.to(dev)
        d.unpack_gathered(recv_dev.data_ptr(), n_total, world, width, out.data_ptr())
        d.sync()
        host_stats.update(packed_local=int(nbytes), padded_per_rank=width, fixed_records_per_rank=n * rec_bytes)

    def fence():
        finish_pending()
        for c in comms:
            if c is not None:
                c.wait()
        for d in dets:
            d.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if os.environ.get("CTAG_BENCH_FAULT") == "rank1_exit" and rank == 1:  # test hook (tests/test_bench_gpu.py): a rank that dies mid-job
        os._exit(17)
    # The timed steps run as the library runs any device-memory batch: every chunk as two halves on two internal streams
    # (CTAG_OPT_STREAMS, default 2), no events.  Per-kernel device time comes from the SAME number of extra steps behind the timed
    # region with CTAG_OPT_TIMING on -- HIP events recorded by the library on ITS stream around every kernel (torch.cuda.Event would
    # only see torch's current stream), which keeps those steps on ONE stream: between two interleaved streams an event pair around
    # a kernel would time the neighbour's kernels as well.  (At N > 1 one extra step: it would serialise the detect / gather pipeline.)
    timed_with_events = False
    acc = {k: 0.0 for k in ca.STAGE_NAMES}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    single_stream_dt = None
    timed_records = None
    if world == 1:
        # the records the TIMED steps left (the one-stream steps below write the same buffer): what `results_sha256` hashes and `parity` compares
        timed_records = local_bufs[0][:n].clone()
        det.set_option(capi.OPT_TIMING, 1)
        step()  # warm: the one-stream workspace
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
            for k, v in det.timings().items():
                acc[k] += v
        fence()
        single_stream_dt = time.perf_counter() - t1
        det.set_option(capi.OPT_TIMING, 0)
        timed_with_events = True
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if dist.get_backend() != "nccl" else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    last = step_no[0] - 1
    if timed_with_events:
        stage_ms = {k: v / max(1, args.steps) for k, v in acc.items()}  # per step (n frames)
    else:
        det.set_option(capi.OPT_TIMING, 1)
        det.detect_batch_device(frames.data_ptr(), n, ROWS, COLS, COLS, ROWS * COLS, local_bufs[(last + 1) % 2].data_ptr(), 5, subpix, 5)
        det.sync()
        stage_ms = det.timings()
        det.set_option(capi.OPT_TIMING, 0)
    launches = (n + chunk - 1) // chunk
    stage_ms = dict(stage_ms)
    # what the frames held (ctag_get_counters: the last chunk of this rank's last step) -- SURVEY.md 5's metrics row
    try:
        frame_counters = det.counters()
    except Exception as e:  # noqa: BLE001
        frame_counters = {"error": "%s: %s" % (type(e).__name__, e)}
    # N > 1: what the exchange costs by itself -- one more gather of the last step's records, alone on the GPU, wall clock from
    # ctag_gather_begin to the end of ctag_gather_wait (max over ranks) -- and every rank's per-kernel times, so that a SCALE
    # record says where a step went
    gather_ms, rank_stage_ms = None, None
    if world > 1:
        cpu_side = dist.get_backend() != "nccl"
        if comm is not None:
            fence()
            g0 = time.perf_counter()
            comm.begin(local_bufs[last % len(local_bufs)][:n], n_total)
            comm.end(gathered[(last + 1) % 2])
            comm.wait()
            det.sync()
            g = torch.tensor([(time.perf_counter() - g0) * 1e3], dtype=torch.float64, device="cpu" if cpu_side else dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather_ms = round(float(g.item()), 3)
        elif dist.get_backend() != "nccl":
            fence()
            g0 = time.perf_counter()
            host_packed_gather(det, local_bufs[last % len(local_bufs)], gathered[(last + 1) % 2])
            g = torch.tensor([(time.perf_counter() - g0) * 1e3], dtype=torch.float64)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather_ms = round(float(g.item()), 3)
        mine = torch.tensor([stage_ms[k] for k in ca.STAGE_NAMES], dtype=torch.float64, device="cpu" if cpu_side else dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rank_stage_ms = [{k: round(float(v), 3) for k, v in zip(ca.STAGE_NAMES, t.tolist())} for t in allr]
    stage_ms["quad"] = sum(stage_ms[k] for k in capi.QUAD_STAGES)  # a4 edgeExtraction: the six kernels of the quad stage

    # ---- outcome of the last step: the job's result list in frame order (gathered when N > 1)
    # N = 1: of the last TIMED step (snapshot taken before the one-stream steps reused the buffer); N > 1: the gathered list of the last timed step
    final = gathered[last % 2] if world > 1 else timed_records
    res = np.frombuffer(final.cpu().numpy().tobytes(), dtype=ca.RESULT_DT)
    ok_frames = int((res["status"] == 0).sum())
    markers_found = int(res["n_markers"].sum())
    sha = hashlib.sha256(res.tobytes()).hexdigest()
    rc = 0
    one_stream_sha = None
    if world == 1:  # the one-stream steps (`stage_ms_per_step`, `roofline`) must have produced the very same records
        one_stream_sha = hashlib.sha256(local_bufs[0][:n].cpu().numpy().tobytes()).hexdigest()
        if one_stream_sha != sha:
            print("bench.py: the one-stream steps' records differ from the timed steps' (%s vs %s)" % (one_stream_sha[:16], sha[:16]), file=sys.stderr, flush=True)
            rc = 1

    if rank == 0:
        sweep_ms = sum(stage_ms[k] for k in SWEEP_STAGES)
        achieved = ALGO_BYTES_PER_FRAME * n / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        traffic, traffic_source, traffic_profile = None, None, None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import srcsha
        src_sha = srcsha.sources_sha256()
        import glob
        # per-round PMC passes, one file per frame size they were collected on (r05_pmc_traffic.json: 1080p; r05_4k_pmc_traffic.json: 3840x2160); latest round wins
        pat = {(1080, 1920): "r[0-9][0-9]_pmc_traffic.json", (2160, 3840): "r[0-9][0-9]_4k_pmc_traffic.json"}.get((ROWS, COLS))
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", pat))) if pat else []
        tpath = cands[-1] if cands else ""
        if os.path.exists(tpath):
            try:
                tprof = json.load(open(tpath))
                traffic = tprof.get("sweep_bytes_per_frame") * min(n, chunk)  # measured per frame
                psha = (tprof.get("sources_sha256") or {}).get("sweep")
                # the profile says which sweep sources it was collected on; a profile of other kernels than this tree's is flagged, not hidden
                traffic_profile = {"file": os.path.relpath(tpath, ROOT), "git_head": tprof.get("git_head"), "sweep_sources_sha256": psha,
                                   "this_tree_sweep_sources_sha256": src_sha["sweep"],
                                   "matches_this_tree": None if psha is None else psha == src_sha["sweep"]}
                traffic_source = "replayed from %s (separate rocprofv3 --pmc passes; not collected in this run)" % os.path.relpath(tpath, ROOT)
            except Exception:
                traffic = None
        try:
            copy_gbs = device_copy_rate(dev) if world == 1 else None
        except Exception:  # noqa: BLE001
            copy_gbs = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_source,
                    "traffic_profile": traffic_profile,
                    "measured_device_copy": None if not copy_gbs else {
                        "GB/s": round(copy_gbs, 1), "frac_of_copy": round(achieved / copy_gbs, 5),
                        "note": "plain device-to-device copy of 2 GiB (read + written bytes), best of 5, measured in this run: the rate a "
                                "streaming kernel reaches on this GPU; `frac` stays priced at the 8 TB/s specification"},
                    "kernel": "threshold+label sweep = " + "+".join("k_" + k for k in SWEEP_STAGES),
                    "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * min(n, chunk),
                    "avg_launch_ms": round(sweep_ms / launches, 4), "launches_per_step": launches,
                    "frames_per_launch": min(n, chunk),
                    "timing": "HIP events on the library's stream, " + ("%d one-stream steps right behind the timed region (the timed steps run the halves of a "
                                                                             "chunk on two streams, where an event pair would time the neighbour's kernels too)" % args.steps
                                                                             if timed_with_events else "one extra untimed step on rank 0 (N > 1)")}
        out = {"metric": "frames/sec detect() %dx%d" % (COLS, ROWS), "value": round(n_total * args.steps / dt, 2),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if strong else "weak",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "synthetic %dx%d random-stripe frames," % (COLS, ROWS) + " %d per step over %d GPU(s), %d planted "
                                      "CTag_2f12c markers each, detect(img,5,%s,5), inputs resident in HBM"
                                      % (n_total, world, args.markers, "true" if subpix else "false"),
                          "frames_per_step": n_total, "frames_per_gpu": n, "chunk": chunk, "parallelism": "frames sharded, dp%d" % world,
                          "gather": gather_impl,
                          "pipelining": ("steps alternate between %d handles (streams)" % len(dets)) if len(dets) > 1 else
                          ("one handle; the library runs the parts of a chunk on its internal streams (CTAG_OPT_STREAMS: library default)" if args.streams == 0 else
                           "one handle, CTAG_OPT_STREAMS = %d (--streams)" % args.streams)},
               "one_stream": None if single_stream_dt is None else {
                   "value": round(n_total * args.steps / single_stream_dt, 2), "unit": "frames/s", "ms_per_step": round(single_stream_dt / args.steps * 1e3, 3),
                   "note": "the same steps with CTAG_OPT_TIMING on (one stream, per-kernel HIP events read back every step): the steps `stage_ms_per_step` and "
                           "`roofline` are measured on"},
               "roofline": roofline,
               "issue_roofline": issue_rooflines(stage_ms, n),
               "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
               "frame_counters": frame_counters if "error" in frame_counters else {
                   "frames": frame_counters["frames"], "any_frame_reruns": frame_counters["reruns"],
                   **{k: {"mean": round(frame_counters[k][0], 2), "max": frame_counters[k][1]} for k in capi.COUNTER_NAMES},
                   "note": "per-frame counts of the last chunk of the last step (ctag_get_counters): components the label sweep published, "
                           "candidates (area in [30 px, 1 %]), quads, features, decoded markers"},
               "frames_ok": ok_frames, "markers_decoded_last_step": markers_found, "results_sha256": sha,
               "results_of": "the last TIMED step" + ("" if world > 1 else " (snapshot before the one-stream steps; their records hash %s)"
                                                      % ("the same" if one_stream_sha == sha else "DIFFERENTLY: " + str(one_stream_sha))),
               "sources_sha256": src_sha}
        if world > 1:
            out["gather_ms"] = gather_ms
            out["rank_stage_ms"] = rank_stage_ms
        if comm is not None:
            lb, pb = det.gather_last_bytes()
            out["config"]["gather_bytes"] = {"packed_local": lb, "padded_per_rank": pb, "fixed_records_per_rank": n * rec_bytes}
        elif host_stats:
            out["config"]["gather_bytes"] = dict(host_stats)
        # side legs: a failure in one of them is reported in its place and never costs the headline line
        def side(name, fn):
            try:
                out[name] = fn()
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

        oracle_records = [None]

        def cpu_leg():
            obj, recs = cpu_baseline(frames, min(args.cpu_frames, n), args.cpu_frames_per_thread, state, fs, subpix)
            oracle_records[0] = recs
            return obj

        if world == 1 and args.cpu_frames > 0:
            side("cpu_baseline", cpu_leg)
            recs = oracle_records[0]
            if recs is not None:  # parity of the timed configuration itself (chunk %d): GPU records vs oracle records, byte for byte
                bad = [int(f) for f in range(len(recs)) if recs[f].tobytes() != res[f].tobytes()]
                out["parity"] = {"frames_checked": len(recs), "mismatches": len(bad), "first_mismatching_frames": bad[:8],
                                 "bar": "byte-identical ctag_frame_result records (ids, order, float corners)"}
                if bad:
                    rc = 1
            else:
                out["parity"] = {"frames_checked": 0, "mismatches": None, "note": "cpu_baseline leg failed"}
                rc = 1
        else:
            out["cpu_baseline"] = None
            out["parity"] = None
        if world == 1 and args.cpu_frames > 0 and frames is not None:
            side("opencv_stage_probe", lambda: opencv_stage_probe(frames[:16].cpu().numpy()))
        if world == 1 and args.pipelined_steps > 0 and frames is not None:
            side("pipelined_two_handles", lambda: pipelined_steps_rate(det, state, fs, dev_index, frames, n, chunk, subpix, local_bufs[0], args.pipelined_steps))
        if world == 1 and args.host_frames > 0:
            side("pcie_inclusive", lambda: host_stream_rate(det, frames, min(args.host_frames, n), subpix))
            side("pcie_inclusive_bgr", lambda: host_stream_rate_bgr(det, frames, min(args.host_frames // 2, n), subpix))
        if world == 1 and args.host_frames > 0 and frames is not None:
            side("bgr_device_resident", lambda: bgr_resident_rate(det, frames, min(args.host_frames, n), subpix, out["value"]))
        if world == 1 and args.latency_calls > 0:
            side("single_frame_latency", lambda: latency_side(det, state, fs, args.latency_calls))
        if world == 1 and args.pose_frames > 0:
            del frames
            frames = None
            if (ROWS, COLS) == (1080, 1920):
                side("pose_side", lambda: pose_side(det, args.pose_frames, dev))
            else:
                side("pose_side", lambda: pose_side_3d(det, state, args.pose_frames, dev, 256))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        for c in reversed(comms):  # the attached handle lets go before the owner destroys the communicator
            if c is not None:
                c.close()
        dist.destroy_process_group()
    for d in dets:
        d.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
