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


def gather_results(local, n_total, dist=None):
    """all-gather per-rank result records (torch uint8 tensor [n_local, record_bytes]) into frame order.

    Ranks may own different counts (shard_range); shards are padded to the largest one for the collective and
    trimmed afterwards.  Returns a tensor [n_total, record_bytes] identical on every rank."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    counts = [shard_range(n_total, r, world) for r in range(world)]
    width = max(hi - lo for lo, hi in counts)
    padded = local
    if local.shape[0] < width:
        pad = torch.zeros((width - local.shape[0], local.shape[1]), dtype=local.dtype, device=local.device)
        padded = torch.cat([local, pad], 0)
    out = torch.empty((world * width, local.shape[1]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous())
    parts = [out[r * width:r * width + (hi - lo)] for r, (lo, hi) in enumerate(counts)]
    return torch.cat(parts, 0)


def records_from_tensor(t, dtype):
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=dtype)
