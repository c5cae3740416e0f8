"""Multi-GPU: frames are independent, so a batch shards contiguously over ranks and the result is
gathered on the host.  One process per GPU (torch.distributed); no data-path collective on the
device -- the numpy-out contract of the reference (tools.py:13-35 returns a host array) means an
RCCL all-gather over xGMI would move bytes that have to cross PCIe to the host anyway.

The gather itself uses torch.distributed on HOST tensors (gloo), so the same code is exercised by
the world_size-2 CPU tests in tests/test_sharding.py.
"""
import numpy as np


def shard_range(n_frames, rank, world_size):
    """Contiguous split of the batch dimension: rank r owns frames [lo, hi).  The first
    n_frames % world_size ranks get one extra frame."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size %r/%r" % (rank, world_size))
    q, r = divmod(n_frames, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def gather_frames(local, n_frames, group=None, dst=None):
    """Host-side gather of per-rank frame slabs into one [n_frames, ...] array.

    local: numpy array [hi-lo, ...] for this rank's shard_range.  With dst=None every rank gets the
    full array (all_gather); otherwise only rank `dst` does and the others return None."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        if local.shape[0] != n_frames:
            raise ValueError("single process: local must hold all frames")
        return local
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    tail = local.shape[1:]
    sizes = [shard_range(n_frames, r, world) for r in range(world)]
    lo, hi = sizes[rank]
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d: expected %d frames, got %d" % (rank, hi - lo, local.shape[0]))
    # collectives want equal-sized slabs: pad every shard to the largest one (they differ by at
    # most one frame) and trim after the gather
    cap = max(b - a for a, b in sizes)
    mine = torch.zeros((cap,) + tail, dtype=torch.from_numpy(local[:0]).dtype)
    mine[: hi - lo] = torch.from_numpy(np.ascontiguousarray(local))
    bufs = [torch.empty_like(mine) for _ in sizes]
    if dst is None:
        dist.all_gather(bufs, mine, group=group)
    else:
        dist.gather(mine, bufs if rank == dst else None, dst=dst, group=group)
        if rank != dst:
            return None
    return torch.cat([buf[: b - a] for buf, (a, b) in zip(bufs, sizes)], dim=0).numpy()


def fill_sharded(x, src_thr=0.1, val_thr=0.1, metric="l1_cv", want=("depth", "dt", "index"),
                 group=None, dst=None, compute=None):
    """Every rank passes the same full batch x [B,H,W] (or at least its own shard's rows valid);
    each computes frames shard_range(B, rank, world) on its own GPU and the results are gathered on
    the host.  `compute(x_shard, src_thr, val_thr, want) -> dict` defaults to the HIP operator on
    the rank's current device; tests inject a checker there."""
    import torch.distributed as dist

    B = x.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_range(B, rank, world)
    if compute is None:
        from . import device as _device

        op = _device.default_op(metric)
        compute = lambda xs, s, v, w: op.run_numpy(xs, s, v, w)
    if hi > lo:
        local = compute(np.ascontiguousarray(x[lo:hi]), src_thr, val_thr, want)
    else:
        H, W = x.shape[1:]
        local = {k: np.empty((0, H, W), np.int32 if k == "index" else np.float32) for k in want}
    out = {}
    for k in want:
        out[k] = gather_frames(local[k], B, group=group, dst=dst)
    return out
