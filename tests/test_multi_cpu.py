"""world_size-2 gloo tests of the multi-GPU protocol (include/ctag_gather.h, host form in cylindertag_amd/dist.py):
contiguous frame shards, packed shards (record heads + used marker / feature records), all-gather of the sizes then of
the padded shards, byte-identical to the single-process list in frame order.  The records are
the committed golden DETECTOR records (tests/golden/golden_v1.npz), not synthetic byte patterns."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_NPZ = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _golden_records(n_total):
    """n_total records in a fixed order: the 64 sequence records, test.bmp, the 8 synthetic ones, plus early-return records."""
    g = np.load(GOLDEN_NPZ)
    recs = np.concatenate([g["seq_results"], g["bmp_result"], g["synth_results"]])
    extra = np.zeros(3, recs.dtype)
    extra["status"] = [1, 2, -3]     # "No corner detected!", "No feature detected!", CTAG_ERR_LIMIT
    extra["flags"] = [0, 0, 8]
    recs = np.concatenate([recs, extra])
    reps = (n_total + len(recs) - 1) // len(recs)
    return np.concatenate([recs] * reps)[:n_total]


def _worker(rank, world, port, n_total, ret):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from cylindertag_amd.dist import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(n_total, rank, world)
    local = _golden_records(n_total)[lo:hi]  # stands in for this rank's detector output: the records of ITS frames
    stats = {}
    out = gather_records(local, n_total, dist, stats)
    ret[rank] = (out.copy(), stats)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [64, 76, 7, 1])
def test_gather_two_ranks_equals_the_golden_list(n_total):
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_worker, args=(r, world, port, n_total, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = _golden_records(n_total)
    want_bytes = np.ascontiguousarray(want).view(np.uint8).reshape(n_total, -1)
    for r in range(world):
        got, stats = ret[r]
        assert got.shape == want_bytes.shape and (got == want_bytes).all()
        assert got.view(want.dtype).ravel().tobytes() == want.tobytes()
        if n_total >= 64:  # what travels is a fraction of the fixed 11 616-byte records (counts + used records)
            assert stats["padded_bytes"] * 2 < stats["fixed_record_bytes"]


def test_pack_unpack_round_trip_and_layout():
    from cylindertag_amd.dist import HEAD, pack_records, unpack_records
    recs = _golden_records(76)
    packed = pack_records(recs)
    n = len(recs)
    assert packed[:4].view(np.int32)[0] == n and packed[8:16].view(np.int64)[0] == packed.size
    heads = packed[HEAD:HEAD + 16 * n].view(np.int32).reshape(n, 4)
    assert (heads[:, 0] == recs["status"]).all() and (heads[:, 1] == recs["n_markers"]).all() and (heads[:, 2] == recs["n_features"]).all()
    assert packed.size == HEAD + 16 * n + int((16 * recs["n_markers"] + 100 * recs["n_features"]).sum())
    back = unpack_records(packed)
    assert back.view(recs.dtype).ravel().tobytes() == recs.tobytes()
    # an empty shard is a bare header
    assert pack_records(recs[:0]).size == HEAD and unpack_records(pack_records(recs[:0])).shape == (0, recs.dtype.itemsize)


def test_shard_ranges_cover_all_frames():
    from cylindertag_amd.dist import shard_range
    from cylindertag_amd import capi
    for n in (0, 1, 7, 4096, 4099):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
            assert spans == [capi.shard_range(n, r, world) for r in range(world)]  # the C ABI's rule (ctag_shard_range)


def _dying_worker(rank, world, port, n_total, ret):
    import datetime
    import time
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from cylindertag_amd.dist import gather_records, shard_range
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=5))
    dist.barrier()
    if rank == 1:
        os._exit(17)  # a crashed peer: no teardown, never enters the exchange
    lo, hi = shard_range(n_total, rank, world)
    t0 = time.perf_counter()
    try:
        gather_records(_golden_records(n_total)[lo:hi], n_total, dist)
        ret[rank] = ("returned", time.perf_counter() - t0)
    except Exception as e:  # noqa: BLE001
        ret[rank] = ("raised %s" % type(e).__name__, time.perf_counter() - t0)
    os._exit(3)


def test_gather_with_a_dead_peer_fails_within_the_deadline():
    """a rank that dies must make the others fail, not hang.  Host form of the protocol (gloo): rank 1 exits before the
    exchange, rank 0's gather raises within the group's deadline and its process exits non-zero.  (The device form's deadline --
    ctag_gather_set_timeout / CTAG_GATHER_TIMEOUT_MS, ncclCommAbort -- is covered under -m gpu: test_gather_deadline_aborts_instead_of_hanging.)"""
    import torch.multiprocessing as mp
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    procs = [mp.Process(target=_dying_worker, args=(r, 2, port, 64, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode is not None, "a rank hangs behind a dead peer"
    assert procs[1].exitcode == 17 and procs[0].exitcode == 3
    what, dt = ret[0]
    assert what.startswith("raised") and dt < 30, ret[0]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher around it (the driver's command shape): bench.py starts torch.distributed.run
    itself, as a child, before anything touches a GPU.  On this GPU-less machine the RANKS then refuse ("needs a GPU": the path has no
    CPU fallback) and the launcher's non-zero exit code comes back -- the failure is the ranks', not a launcher check."""
    import subprocess
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--frames", "8"],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""))
    assert p.returncode != 0
    assert 1 <= p.stderr.count("bench.py needs a GPU") <= 2, p.stderr[-3000:]  # (the launcher ends the second rank as soon as the first has failed: one message or two)
    assert "needs `python -m torch.distributed.run" not in p.stderr  # round 4's launcher check is gone
