"""bench.py itself, on the GPU box: the one-GPU line carries the objects the contract asks for and verifies what it times, and
the N > 1 logic (shards, double-buffered gathers, the hash of the gathered list) gives the same result list as N = 1 -- run
with two ranks sharing the one GPU through the developer backend (RCCL refuses two ranks on one device, so the gather goes
through gloo and host memory there; the RCCL gather itself is covered by test_packed_shards_and_rccl_gather_single_rank)."""
import json
import os
import socket
import subprocess
import sys

import pytest

from ctag_testlib import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _line(out):
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def test_one_gpu_line_and_two_rank_hash():
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "192", "--host-frames", "0",
            "--pose-frames", "0", "--latency-calls", "0"]
    p = subprocess.run(base + ["--cpu-frames", "24", "--cpu-frames-per-thread", "2"], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    one = _line(p.stdout)
    assert one["n_gpus"] == 1 and one["unit"] == "frames/s" and one["frames_ok"] == 192
    assert one["parity"]["mismatches"] == 0 and one["parity"]["frames_checked"] >= 24
    assert one["roofline"]["bound"] == "hbm" and 0.0 < one["roofline"]["frac"] < 1.0 and one["roofline"]["traffic_source"]
    assert one["cpu_baseline"]["kind"] == "port" and one["cpu_baseline"]["cores"] == 1 and one["cpu_baseline"]["value"] > 0
    assert set(one["issue_roofline"]["kernels"]) >= {"k_edge_refine", "k_welsch", "k_quad_edges_packed"}
    # N = 2 the way the driver starts it: `python bench.py --gpus 2 ...`, no launcher around it -- bench.py starts torch.distributed.run
    # itself as a child process before anything touches the GPU (launch_ranks) and hands back the launcher's exit code
    env = dict(os.environ, CTAG_BENCH_BACKEND="gloo")
    p = subprocess.run(base + ["--gpus", "2", "--cpu-frames", "0"], capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    two = _line(p.stdout)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["config"]["frames_per_gpu"] == 96 and two["frames_ok"] == 192
    assert two["results_sha256"] == one["results_sha256"]  # the gathered list equals the one-GPU list
    assert two["config"]["pipelining"].startswith("steps alternate between 2 handles")
    # ... and under an explicit launcher (the other command shape of the contract)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + base[1:] + ["--gpus", "2", "--cpu-frames", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    assert _line(p.stdout)["results_sha256"] == one["results_sha256"]


def test_a_rank_that_dies_ends_the_job_non_zero():
    """A rank that fails must end the whole job with a non-zero exit code, promptly: rank 1 is made to leave after the warm-up
    (CTAG_BENCH_FAULT=rank1_exit, a test hook), the launcher ends rank 0 and `python bench.py --gpus 2` returns non-zero."""
    import time
    env = dict(os.environ, CTAG_BENCH_BACKEND="gloo", CTAG_BENCH_FAULT="rank1_exit")
    t0 = time.perf_counter()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--frames", "64", "--cpu-frames", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and time.perf_counter() - t0 < 300
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]  # no line from a job that lost a rank


@pytest.mark.parametrize("frames", [256, 250])
def test_eight_ranks_give_the_one_gpu_list(frames):
    """BASELINE config 4's shape before it meets eight GPUs: `python bench.py --gpus 8` (self-launched, eight ranks sharing this box's one
    GPU through the developer backend) on 256 frames and on 250 (uneven shards: two ranks own 32 frames, six 31).  Rank 0's gathered
    list must hash like the one-GPU list of the same frames, and the line must carry what a SCALE record is read for: `roofline`,
    `gather_ms`, every rank's per-kernel times and the gather in use.  What the developer backend replaces is the two ncclAllGather
    calls only: shards are packed by ctag_pack_results and unpacked by ctag_gather_end's kernels."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", str(frames), "--host-frames", "0",
            "--pose-frames", "0", "--latency-calls", "0", "--cpu-frames", "0", "--pipelined-steps", "0"]
    p = subprocess.run(base, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    one = _line(p.stdout)
    assert one["frames_ok"] == frames and one["results_of"].startswith("the last TIMED step") and "the same" in one["results_of"]
    env = dict(os.environ, CTAG_BENCH_BACKEND="gloo", OMP_NUM_THREADS="1")
    p = subprocess.run(base + ["--gpus", "8"], capture_output=True, text=True, timeout=1500, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    eight = _line(p.stdout)
    assert eight["n_gpus"] == 8 and eight["scaling"] == "strong" and eight["frames_ok"] == frames
    assert eight["config"]["frames_per_step"] == frames and eight["config"]["frames_per_gpu"] == (frames + 7) // 8
    assert eight["results_sha256"] == one["results_sha256"]
    assert eight["roofline"]["bound"] == "hbm" and eight["roofline"]["frac"] > 0 and eight["roofline"]["frames_per_launch"] == (frames + 7) // 8
    assert eight["gather_ms"] is not None and eight["gather_ms"] > 0
    assert len(eight["rank_stage_ms"]) == 8 and all(r["edge_refine"] > 0 and r["decimate"] > 0 for r in eight["rank_stage_ms"])
    assert "ctag_pack_results" in eight["config"]["gather"]  # on RCCL: "ctag_gather (C ABI -> ncclAllGather of packed shards)"
    gb = eight["config"]["gather_bytes"]
    assert 0 < gb["packed_local"] <= gb["padded_per_rank"] < gb["fixed_records_per_rank"]
