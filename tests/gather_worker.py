"""Worker of test_two_rank_rccl_gather_when_two_gpus_are_present (launched by torch.distributed.run, one rank per GPU): every rank
detects ALL frames of a small job on its own GPU, then the ranks gather their shards through the library's RCCL gather
(include/ctag_gather.h) and every rank checks that the gathered list equals its own full list byte for byte -- uneven shards, a job
smaller than the world, an empty job, the two-phase form, and two handles sharing one communicator (bench.py's N > 1 pattern)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch
    import torch.distributed as dist
    import cylindertag_amd as ca
    import testkit as tk
    from cylindertag_amd.dist import CommGather, shard_range
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group(backend="nccl", device_id=dev)
    state, fs = ca.load_marker_file(os.path.join(ROOT, "tests", "golden", "CTag_2f12c.marker"))
    det, det2 = tk.Detector(state, fs, device=local), tk.Detector(state, fs, device=local)
    rows, cols, n_all = 1080, 1920, 37
    frames = torch.empty((n_all, rows, cols), dtype=torch.uint8, device=dev)
    det.synth_frames_device(frames.data_ptr(), 11, n_all, rows, cols, cols, rows * cols)
    frames[2] = 180  # an early return
    torch.cuda.synchronize()
    full = torch.zeros((n_all, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
    det.detect_batch_device(frames.data_ptr(), n_all, rows, cols, cols, rows * cols, full.data_ptr())
    det.sync()
    want = full.cpu().numpy()
    comm = CommGather(det, dist)
    if "--dead-peer" in sys.argv:
        # rank 1 never enters the collective -- it sits there, as a hung process would (were it to EXIT, the launcher would end rank 0 before rank 0
        # could show anything); rank 0's gather must come back with CTAG_ERR_HIP within CTAG_GATHER_TIMEOUT_MS instead of hanging, and the
        # process exits non-zero (the launcher then ends rank 1)
        import time
        dist.barrier()
        if rank == 1:
            time.sleep(60)
            os._exit(0)
        lo, hi = shard_range(n_all, rank, world)
        out = torch.zeros((n_all, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
        t0 = time.perf_counter()
        try:
            det.gather(full[lo:hi].contiguous().data_ptr(), hi - lo, n_all, out.data_ptr())
        except Exception as e:  # noqa: BLE001
            print("DEAD_PEER_DETECTED after %.1f s: %s" % (time.perf_counter() - t0, e), flush=True)
            os._exit(3)
        print("gather returned although rank 1 is gone", flush=True)
        os._exit(0)
    comm2 = CommGather(det2, dist, share=comm)  # the second handle gathers through the first one's communicator
    for n_total in (37, 1, 0, 36):
        lo, hi = shard_range(n_total, rank, world)
        local_rec = full[lo:hi].contiguous() if hi > lo else torch.zeros((1, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev)
        out = torch.full((max(n_total, 1), ca.RESULT_DT.itemsize), 0xEE, dtype=torch.uint8, device=dev)
        det.gather(local_rec.data_ptr(), hi - lo, n_total, out.data_ptr())
        assert (out.cpu().numpy()[:n_total] == want[:n_total]).all(), "rank %d: gather of %d frames differs" % (rank, n_total)
    # bench.py's pattern: steps alternate between the two handles, the gather of step k ends while step k + 1 detects
    dets, comms = [det, det2], [comm, comm2]
    lo, hi = shard_range(n_all, rank, world)
    recs = [torch.zeros((hi - lo, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev) for _ in range(2)]
    outs = [torch.zeros((n_all, ca.RESULT_DT.itemsize), dtype=torch.uint8, device=dev) for _ in range(2)]
    pending = None
    for k in range(6):
        d = dets[k % 2]
        d.detect_batch_device(frames[lo:hi].data_ptr(), hi - lo, rows, cols, cols, rows * cols, recs[k % 2].data_ptr())
        if pending is not None:
            comms[pending % 2].end(outs[pending % 2])
        comms[k % 2].begin(recs[k % 2], n_all)
        pending = k
    comms[pending % 2].end(outs[pending % 2])
    for c, d in zip(comms, dets):
        c.wait()
        d.sync()
    for o in outs:
        assert (o.cpu().numpy() == want).all(), "rank %d: pipelined gather differs" % rank
    dist.barrier()
    comm2.close()
    comm.close()
    det2.close()
    det.close()
    dist.destroy_process_group()
    if rank == 0:
        print("GATHER_WORKER_OK", flush=True)


if __name__ == "__main__":
    main()
