#!/usr/bin/env python3
"""bench.py -- frames/s of the MI355X-native CylinderTag detect() front end on BASELINE.json's config 3:
a synthetic batch of 1920x1080 random-stripe frames, resident in HBM, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

A "step" is one pass of the whole detect() path (resize -> threshold -> label -> quads -> features -> sub-pixel
refine -> decode) over one batch of frames per GPU; frames are independent, so each rank owns its own shard of
the job (weak scaling) and the only collective is the final RCCL all-gather of the result records.
Prints ONE JSON line on rank 0 (contract in the task statement), with two extra objects:
  roofline      the threshold+label sweep (2*W*H algorithmic bytes per frame) against 8 TB/s HBM,
                timed with HIP events on the library's own stream
  cpu_baseline  the CPU restatement (oracle/, "port") timed on the host cores on a bounded sample of the batch
PyTorch is used only for device memory, synchronisation and torch.distributed.
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
    ap.add_argument("--frames", type=int, default=4096, help="frames per GPU per step (BASELINE config 3: 4096)")
    ap.add_argument("--chunk", type=int, default=4096, help="frames per pipeline pass (workspace size)")
    ap.add_argument("--markers", type=int, default=4)
    ap.add_argument("--cpu-frames", type=int, default=384, help="sample size of the CPU baseline (0 = skip)")
    ap.add_argument("--host-frames", type=int, default=1024,
                    help="frames of the host-memory (PCIe-inclusive) side measurement, 0 = skip; never `value`")
    ap.add_argument("--pose-frames", type=int, default=1024,
                    help="frames of the detect()+estimatePose side measurement on camera content, 0 = skip; never `value`")
    ap.add_argument("--size", default="1920x1080", help="frame size WxH; the headline metric is 1920x1080 (other sizes are side measurements)")
    ap.add_argument("--no-subpix", action="store_true")
    return ap.parse_args()


def host_stream_rate(det, frames_dev, m, subpix):
    """Side measurement (never `value`): the same frames handed over as HOST buffers through ctag_detect_batch_u8 --
    pinned memory, uploads double-buffered against detection, results downloaded.  Bounded by PCIe."""
    import cylindertag_amd as ca
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


def pose_side(det, m, dev):
    """Side measurement (never `value`; SURVEY.md 8(f) rank 2 / BASELINE config 5's estimatePose leg): camera content --
    the 64-frame sequence derived from the reference's test.bmp (config 2's test.avi substitute, 5 physical markers in
    view) tiled to m frames in HBM -- through detect() and then the GPU pose back end (EPnP + LM per marker against the
    reference's CTag_2f12c.model / cameraParams.yml), everything device-resident.  The CPU pose oracle is timed beside it."""
    import torch
    import cylindertag_amd as ca
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


def cpu_baseline(frames_host, state, fs, subpix):
    """Times the CPU restatement of detect() (oracle, kind 'port'), single thread, on the given frames."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from ctag_testlib import Oracle  # the oracle is test infrastructure: used here only as the timed CPU baseline
    orc = Oracle()
    n = frames_host.shape[0]
    orc.detect_fast(frames_host[0], state, fs, 5, subpix, 5)  # warm
    t0 = time.perf_counter()
    for i in range(n):
        orc.detect_fast(frames_host[i], state, fs, 5, subpix, 5)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "frames/s", "cores": 1, "kind": "port",
           "sample": "first %d frames of the same synthetic batch, CPU restatement of detect() "
                     "(oracle/ctag_oracle.cpp, -O2 -ffp-contract=off), 1 thread, %.1f s" % (n, dt)}
    # frame-parallel on every host core (SURVEY.md 8(d) baseline (ii)): the same frames, one oracle call per frame, a thread
    # per core (the calls run outside the GIL; results equal the single-thread ones)
    from concurrent.futures import ThreadPoolExecutor
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    if cores > 1:
        with ThreadPoolExecutor(cores) as ex:
            t0 = time.perf_counter()
            list(ex.map(lambda f: orc.detect_fast(f, state, fs, 5, subpix, 5), list(frames_host)))
            dt_all = time.perf_counter() - t0
        out["all_cores"] = {"value": n / dt_all, "unit": "frames/s", "cores": cores,
                            "sample": "the same %d frames, one thread per host core, %.1f s" % (n, dt_all)}
    return out


def main():
    global ROWS, COLS, ALGO_BYTES_PER_FRAME
    args = parse_args()
    COLS, ROWS = (int(v) for v in args.size.lower().split("x"))
    ALGO_BYTES_PER_FRAME = 2 * ROWS * COLS
    import torch
    import cylindertag_amd as ca
    from cylindertag_amd import capi

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`"
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the detection path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=dev)  # nccl == RCCL on ROCm

    state, fs = ca.load_marker_file(os.path.join(ROOT, "tests", "golden", "CTag_2f12c.marker"))
    det = ca.Detector(state, fs, device=local_rank)
    det.set_option(capi.OPT_MAX_CHUNK, args.chunk)
    n = args.frames
    subpix = not args.no_subpix

    # ---- synthetic shard of this rank, generated on the device (frames [rank*n, (rank+1)*n) of the job)
    frames = torch.empty((n, ROWS, COLS), dtype=torch.uint8, device=dev)
    det.synth_frames_device(frames.data_ptr(), rank * n, n, ROWS, COLS, COLS, ROWS * COLS, markers=args.markers)
    # two result buffers: the gather of step k (RCCL, its own stream) overlaps the detection of step k+1
    result_bufs = [torch.zeros((n, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev) for _ in range(2 if world > 1 else 1)]
    results = result_bufs[0]
    from cylindertag_amd.dist import gather_results_async
    pending = []
    step_no = [0]

    def step():
        buf = result_bufs[step_no[0] % len(result_bufs)]
        step_no[0] += 1
        if len(pending) == len(result_bufs):
            pending.pop(0).wait()  # the gather that read this buffer two steps ago
        det.detect_batch_device(frames.data_ptr(), n, ROWS, COLS, COLS, ROWS * COLS, buf.data_ptr(), 5, subpix, 5)
        if world > 1:
            det.sync()  # results are produced on the library's stream
            pending.append(gather_results_async(buf, world * n, dist))  # the path's only exchange: final marker lists (RCCL all-gather)

    def fence():
        while pending:
            pending.pop(0).wait()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    # Per-kernel device time: HIP events recorded by the library on ITS stream around every kernel of the timed steps
    # themselves (torch.cuda.Event would only see torch's current stream).  Reading them back costs one event
    # synchronisation per chunk, which is inside the timed region.
    det.set_option(capi.OPT_TIMING, 1)
    acc = {k: 0.0 for k in ca.STAGE_NAMES}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        for k, v in det.timings().items():
            acc[k] += v
    fence()
    dt = time.perf_counter() - t0
    det.set_option(capi.OPT_TIMING, 0)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    stage_ms = {k: v / max(1, args.steps) for k, v in acc.items()}  # per step (n frames)
    launches = (n + args.chunk - 1) // args.chunk

    # ---- sanity on the outcome of the last step
    results = result_bufs[(step_no[0] - 1) % len(result_bufs)]
    res = np.frombuffer(results.cpu().numpy().tobytes(), dtype=ca.RESULT_DT)
    ok_frames = int((res["status"] == 0).sum())
    markers_found = int(res["n_markers"].sum())

    if rank == 0:
        sweep_ms = sum(stage_ms[k] for k in SWEEP_STAGES)
        achieved = ALGO_BYTES_PER_FRAME * n / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
        traffic = None
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))  # per-round PMC passes; latest round wins
        tpath = cands[-1] if cands else ""
        if os.path.exists(tpath) and (ROWS, COLS) == (1080, 1920):  # the PMC passes were collected on the 1080p workload
            try:
                traffic = json.load(open(tpath)).get("sweep_bytes_per_frame") * min(n, args.chunk)  # measured per frame
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "kernel": "threshold+label sweep = " + "+".join("k_" + k for k in SWEEP_STAGES),
                    "algorithmic_bytes_per_launch": ALGO_BYTES_PER_FRAME * min(n, args.chunk),
                    "avg_launch_ms": round(sweep_ms / launches, 4), "launches_per_step": launches,
                    "frames_per_launch": min(n, args.chunk)}
        out = {"metric": "frames/sec detect() %dx%d" % (COLS, ROWS), "value": round(world * n * args.steps / dt, 2),
               "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "u8", "data": "synthetic",
               "config": {"workload": "synthetic %dx%d random-stripe frames," % (COLS, ROWS) + " %d per GPU per step, %d planted "
                                      "CTag_2f12c markers each, detect(img,5,%s,5), inputs resident in HBM"
                                      % (n, args.markers, "true" if subpix else "false"),
                          "frames_per_gpu": n, "chunk": args.chunk, "parallelism": "frames sharded, dp%d" % world},
               "roofline": roofline,
               "stage_ms_per_step": {k: round(v, 3) for k, v in stage_ms.items()},
               "frames_ok": ok_frames, "markers_decoded_last_step": markers_found}
        cpu_sample = frames[:min(args.cpu_frames, n)].cpu().numpy() if (world == 1 and args.cpu_frames > 0) else None
        # side legs: a failure in one of them is reported in its place and never costs the headline line
        def side(name, fn):
            try:
                out[name] = fn()
            except Exception as e:  # noqa: BLE001
                out[name] = {"error": "%s: %s" % (type(e).__name__, e)}

        if world == 1 and args.host_frames > 0:
            side("pcie_inclusive", lambda: host_stream_rate(det, frames, min(args.host_frames, n), subpix))
        if world == 1 and args.pose_frames > 0 and (ROWS, COLS) == (1080, 1920):
            del frames
            frames = None
            side("pose_side", lambda: pose_side(det, args.pose_frames, dev))
        if world == 1 and args.cpu_frames > 0:
            side("cpu_baseline", lambda: cpu_baseline(cpu_sample, state, fs, subpix))
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    det.close()


if __name__ == "__main__":
    main()
