"""world_size-2 gloo test of the multi-GPU plumbing (cylindertag_amd/dist.py): contiguous frame shards, one
all-gather of result records, byte-identical to the single-process list in frame order."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, rec_bytes, ret):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from cylindertag_amd.dist import gather_results, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    # every frame's record is a deterministic function of its global index (stands in for the detector output)
    local = torch.stack([torch.full((rec_bytes,), (f * 7 + 3) % 251, dtype=torch.uint8) for f in range(lo, hi)]) if hi > lo \
        else torch.zeros((0, rec_bytes), dtype=torch.uint8)
    out = gather_results(local, n_total, dist)
    ret[rank] = out.numpy().copy()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7, 1])
def test_gather_two_ranks_matches_single_process(n_total):
    import torch.multiprocessing as mp
    world, rec = 2, 64
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(r, world, port, n_total, rec, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.stack([np.full((rec,), (f * 7 + 3) % 251, np.uint8) for f in range(n_total)])
    for r in range(world):
        assert ret[r].shape == want.shape and (ret[r] == want).all()


def _worker_pipelined(rank, world, port, n_local, rec_bytes, steps, ret):
    """bench.py's N>1 step loop: two result buffers, the gather of step k in flight while step k+1 is produced."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from cylindertag_amd.dist import gather_results_async
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    bufs = [torch.zeros((n_local, rec_bytes), dtype=torch.uint8) for _ in range(2)]
    pending, got = [], []
    for k in range(steps):
        buf = bufs[k % 2]
        if len(pending) == 2:
            got.append(pending.pop(0).wait().numpy().copy())
        for f in range(n_local):  # "detection" of step k: record = f(global frame index, step)
            buf[f] = (rank * n_local + f) * 5 + k * 17 + 1
        pending.append(gather_results_async(buf, world * n_local, dist))
    while pending:
        got.append(pending.pop(0).wait().numpy().copy())
    ret[rank] = np.stack(got)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gathers_two_ranks():
    import torch.multiprocessing as mp
    world, n_local, rec, steps = 2, 3, 32, 5
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker_pipelined, args=(r, world, port, n_local, rec, steps, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.stack([np.stack([np.full((rec,), (f * 5 + k * 17 + 1) % 256, np.uint8) for f in range(world * n_local)])
                     for k in range(steps)])
    for r in range(world):
        assert ret[r].shape == want.shape and (ret[r] == want).all()


def test_shard_ranges_cover_all_frames():
    from cylindertag_amd.dist import shard_range
    for n in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
