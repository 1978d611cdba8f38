"""Multi-GPU plumbing around include/ctag_gather.h: one process per GPU, frames sharded contiguously, ONE exchange at the
end -- the gather of the detected marker lists (north_star).

Frames are independent (the reference clears all per-frame state, CylinderTag.cpp:73-76), so the data path has no
collective.  On GPUs the gather is the library's own (`ctag_gather*`: pack kernel -> ncclAllGather of the packed sizes ->
ncclAllGather of the packed shards -> unpack kernel, RCCL called directly from the C ABI); `CommGather` below only
bootstraps its communicator through torch.distributed.  The numpy functions restate the packed-shard format on the
host: they are what the world_size-2 gloo tests run (no GPU there), and the GPU tests check the kernels against them."""
import numpy as np

HEAD = 16  # shard header, and the head {status, n_markers, n_features, flags} of a record
MAX_MARKERS = MAX_FEATURES = 100
MARKER_BYTES, FEATURE_BYTES, RECORD_BYTES = 16, 100, 11616
_MARKERS_OFF, _FEATURES_OFF = HEAD, HEAD + MAX_MARKERS * MARKER_BYTES


def shard_range(n_frames, rank, world):
    """Contiguous frame range [lo, hi) owned by `rank` (first ranks take the remainder) == ctag_shard_range."""
    base, rem = divmod(n_frames, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _as_bytes(records):
    a = np.ascontiguousarray(records)
    a = a.view(np.uint8).reshape(-1, RECORD_BYTES)
    return a


def pack_records(records):
    """n fixed records (any array whose rows are the 11 616 record bytes) -> packed shard (uint8 array), the layout of
    include/ctag_gather.h: header {n, 0, total_bytes}, n record heads, then per frame its used marker and feature records."""
    a = _as_bytes(records)
    n = a.shape[0]
    heads = a[:, :HEAD].copy().view(np.int32).reshape(n, 4)
    nm = np.clip(heads[:, 1], 0, MAX_MARKERS)
    nf = np.clip(heads[:, 2], 0, MAX_FEATURES)
    parts = [np.zeros(HEAD, np.uint8), heads.view(np.uint8).ravel()]
    for f in range(n):
        parts.append(a[f, _MARKERS_OFF:_MARKERS_OFF + int(nm[f]) * MARKER_BYTES])
        parts.append(a[f, _FEATURES_OFF:_FEATURES_OFF + int(nf[f]) * FEATURE_BYTES])
    out = np.concatenate(parts)
    hd = out[:HEAD].view(np.int32)
    hd[0] = n
    out[8:16].view(np.int64)[0] = out.size
    return out


def unpack_records(packed, n=None):
    """packed shard -> uint8 array [n, 11 616] of fixed records (unused bytes zero, as the detector writes them)."""
    packed = np.ascontiguousarray(packed, dtype=np.uint8)
    n_in = int(packed[:4].view(np.int32)[0])
    n = n_in if n is None else n
    assert n == n_in, "shard holds %d frames, expected %d" % (n_in, n)
    heads = packed[HEAD:HEAD + n * HEAD].view(np.int32).reshape(n, 4)
    out = np.zeros((n, RECORD_BYTES), np.uint8)
    p = HEAD + n * HEAD
    for f in range(n):
        nm, nf = int(np.clip(heads[f, 1], 0, MAX_MARKERS)), int(np.clip(heads[f, 2], 0, MAX_FEATURES))
        out[f, :HEAD] = heads[f].view(np.uint8)
        out[f, _MARKERS_OFF:_MARKERS_OFF + nm * MARKER_BYTES] = packed[p:p + nm * MARKER_BYTES]
        p += nm * MARKER_BYTES
        out[f, _FEATURES_OFF:_FEATURES_OFF + nf * FEATURE_BYTES] = packed[p:p + nf * FEATURE_BYTES]
        p += nf * FEATURE_BYTES
    assert p == int(packed[8:16].view(np.int64)[0]), "packed size mismatch"
    return out


def gather_records(local, n_total, dist=None, stats=None):
    """Host form of ctag_gather (same protocol, torch.distributed collectives on CPU tensors -- the gloo tests): `local` =
    this rank's records; returns uint8 [n_total, 11 616], identical on every rank.  `stats` (dict) receives the byte counts."""
    import torch
    a = _as_bytes(local)
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return unpack_records(pack_records(a))
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_total, rank, world)
    assert hi - lo == a.shape[0], "rank %d owns frames [%d, %d), got %d records" % (rank, lo, hi, a.shape[0])
    packed = pack_records(a)
    sizes = torch.zeros(world, dtype=torch.int64)
    dist.all_gather_into_tensor(sizes, torch.tensor([packed.size], dtype=torch.int64))  # exchange 1: world x 8 bytes
    width = (int(sizes.max()) + 255) & ~255
    send = torch.zeros(width, dtype=torch.uint8)
    send[:packed.size] = torch.from_numpy(packed)
    recv = torch.empty(world * width, dtype=torch.uint8)
    dist.all_gather_into_tensor(recv, send)                                              # exchange 2: the packed shards
    if stats is not None:
        stats.update(local_bytes=int(packed.size), padded_bytes=width, fixed_record_bytes=a.shape[0] * RECORD_BYTES)
    buf = recv.numpy()
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(n_total, r, world)
        parts.append(unpack_records(buf[r * width:r * width + int(sizes[r])], rhi - rlo))
    return np.concatenate(parts, 0)


class CommGather:
    """The library's RCCL gather (include/ctag_gather.h) for one Detector: __init__ bootstraps the communicator (rank 0's
    ncclUniqueId travels through the already-initialised torch.distributed group, the only thing torch does here);
    begin/end/wait map 1:1 onto ctag_gather_begin/_end/_wait.  `share` = another CommGather of this process: the detector
    then gathers through THAT communicator (ctag_comm_native -> ctag_comm_attach) instead of creating a second one -- a process
    holds one library communicator however many handles it pipelines."""

    def __init__(self, det, dist, share=None):
        import torch
        from . import capi
        self.det, self.rank, self.world = det, dist.get_rank(), dist.get_world_size()
        if share is not None:
            native = share.det.comm_native()
            if not native:
                raise RuntimeError("the shared CommGather holds no communicator")
            det.comm_attach(native, self.rank, self.world)
            return
        ident = [capi.comm_unique_id() if self.rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        det.comm_init(ident[0], self.rank, self.world)
        torch.cuda.synchronize()

    def begin(self, local, n_total):
        self.det.gather_begin(local.data_ptr(), local.shape[0], n_total)

    def end(self, out):
        self.det.gather_end(out.data_ptr())

    def wait(self):
        self.det.gather_wait()

    def close(self):
        self.det.comm_destroy()


def records_from_tensor(t, dtype):
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=dtype)
