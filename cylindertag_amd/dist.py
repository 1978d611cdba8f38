"""Multi-GPU plumbing: one process per GPU, frames sharded contiguously, one all-gather of the result records.

Frames are independent (the reference clears all per-frame state, CylinderTag.cpp:73-76), so the data path has no
collective; the only exchange is the final gather of marker lists (north_star).  backend "nccl" is RCCL on ROCm;
the same code runs on "gloo" CPU tensors, which is how the CPU test-suite covers it."""
import numpy as np


def shard_range(n_frames, rank, world):
    """Contiguous frame range [lo, hi) owned by `rank` (first ranks take the remainder)."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class _Gather:
    """Handle of one (possibly still running) gather: wait() returns the [n_total, record_bytes] tensor in frame order."""

    def __init__(self, out, work, counts, width, local):
        self.out, self.work, self.counts, self.width, self.local = out, work, counts, width, local

    def wait(self):
        import torch
        if self.work is None:
            return self.local
        self.work.wait()
        if self.out.is_cuda:  # nccl: wait() only orders the current stream behind the collective
            torch.cuda.current_stream(self.out.device).synchronize()
        parts = [self.out[r * self.width:r * self.width + (hi - lo)] for r, (lo, hi) in enumerate(self.counts)]
        return torch.cat(parts, 0)


def gather_results_async(local, n_total, dist=None):
    """Starts the all-gather of per-rank result records (torch uint8 tensor [n_local, record_bytes]) and returns a
    handle; the collective runs while the caller enqueues the next batch.  `local` must stay untouched until wait().

    Ranks may own different counts (shard_range); shards are padded to the largest one for the collective and
    trimmed afterwards."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return _Gather(None, None, None, 0, local)
    world = dist.get_world_size()
    counts = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in counts)
    padded = local
    if local.shape[0] < width:
        pad = torch.zeros((width - local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    out = torch.empty((world * width, local.shape[1]), dtype=local.dtype, device=local.device)
    work = dist.all_gather_into_tensor(out, padded.contiguous(), async_op=True)
    return _Gather(out, work, counts, width, local)


def gather_results(local, n_total, dist=None):
    """Blocking form: returns a tensor [n_total, record_bytes] identical on every rank."""
    return gather_results_async(local, n_total, dist).wait()


def records_from_tensor(t, dtype):
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=dtype)
