"""Multi-GPU: frames are independent, so a batch shards contiguously over ranks (one process per GPU) and
the result lands in ONE host buffer.  No data-path collective: the numpy-out contract of the reference
(tools.py:13-35 returns a host array) means an RCCL all-gather over xGMI would move bytes that have to cross
PCIe to the host anyway.

Host gather = every rank's device-to-host copy goes straight into ITS slice of one shared host slab:
  * the slab is a POSIX shared-memory segment per output (depth, dt, index, status) that rank 0 of the group
    creates and the other ranks attach to by name (the name travels over the group's store as a broadcast);
  * every rank pins its mapping with cudaHostRegister when a GPU is present, so the copy is one asynchronous
    DMA from the rank's GPU into the final place -- no staging buffer, no concatenation, no TCP;
  * one barrier later the whole batch is in the slab on every rank (same physical pages), and the arrays that
    are returned ARE the slab: no host copy behind the DMA.  "Every call returns arrays of its own" (tools.py:29-35
    builds fresh ones) is kept by rotation: a slab is written again only after every rank has dropped the arrays
    it was handed from it (one small all-reduce at the start of a call agrees on a free slab; if the callers
    still hold them all, a new one is made).
torch.distributed is used for control only (slab names, that all-reduce, the barrier, the error flags), so the
world_size-2 gloo tests in tests/test_sharding.py exercise exactly this code on the CPU.

Errors: the reference raises IndexError once for the whole batch (tools.py:26).  Here every rank finishes its
shard, records a per-frame status in the slab, joins the barrier (in a `finally`: also when its own shard
failed in any other way -- a HIP error, a bad shape, out of memory -- which it records as FRAME_SHARD_FAILED),
and then EVERY rank raises: the same IndexError for the first bad global frame, or a RuntimeError naming the
rank whose shard failed.  No rank is left waiting in a collective.
"""
import atexit
import os
import weakref
from multiprocessing import shared_memory

import numpy as np

_DTYPES = {"depth": np.float32, "dt": np.float32, "index": np.int32, "status": np.int32}
FRAME_INDEX_ERROR = 1
FRAME_SHARD_FAILED = 1 << 30  # this frame's shard raised something other than the reference's IndexError


def shard_range(n_frames, rank, world_size):
    """Contiguous split of the batch dimension: rank r owns frames [lo, hi).  The first
    n_frames % world_size ranks get one extra frame."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size %r/%r" % (rank, world_size))
    q, r = divmod(n_frames, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _dist(group):
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(group), dist.get_world_size(group)
    return None, 0, 1


class HostSlab:
    """One host buffer [n_frames, ...] per name, shared by the ranks of a group.

    Rank 0 creates the segments, the others attach; `view(name)` is a fresh numpy array over the whole batch and
    `tensors[name]` the torch view of the same memory (pinned when a GPU is present, for async D2H).
    `dtypes` maps names to numpy dtypes (default: the operator's outputs)."""

    def __init__(self, n_frames, frame_shape, names, group=None, pin=None, dtypes=None):
        import torch

        dist, rank, world = _dist(group)
        self.group, self.rank, self.world = group, rank, world
        self.names = tuple(names) + ("status",)
        self.dtypes = dict(_DTYPES, **(dtypes or {}))
        self.shapes = {k: ((n_frames,) if k == "status" else (n_frames,) + tuple(frame_shape)) for k in self.names}
        sizes = {k: max(int(np.prod(self.shapes[k])) * np.dtype(self.dtypes[k]).itemsize, 1) for k in self.names}
        self._shm = {}
        if rank == 0:
            for k in self.names:
                self._shm[k] = shared_memory.SharedMemory(create=True, size=sizes[k])
        if dist is not None:
            box = [{k: s.name for k, s in self._shm.items()}] if rank == 0 else [None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            if rank != 0:
                for k in self.names:
                    self._shm[k] = shared_memory.SharedMemory(name=box[0][k])
                    # (Python < 3.13 registers an ATTACHED segment with this process's resource tracker as well, which then tries
                    # to unlink rank 0's segment at exit and warns that it is gone: only the creator owns it)
                    try:
                        from multiprocessing import resource_tracker

                        resource_tracker.unregister(self._shm[k]._name, "shared_memory")
                    except Exception:
                        pass
        self.arrays, self.tensors, self._registered = {}, {}, []
        self._handed = []  # weak references to the arrays handed to the caller from this slab
        if pin is None:
            pin = torch.cuda.is_available()
        for k in self.names:
            a = np.ndarray(self.shapes[k], dtype=self.dtypes[k], buffer=self._shm[k].buf)
            self.arrays[k] = a
            t = torch.from_numpy(a)
            if pin and a.nbytes:
                # page-lock this process's mapping of the segment: the D2H copy becomes a DMA into the final place
                rc = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), a.nbytes, 0)
                if int(rc) == 0:
                    self._registered.append(t.data_ptr())
            self.tensors[k] = t

    def view(self, name):
        """A fresh ndarray over the slab's buffer for the caller.  Every view the caller derives from it keeps it alive
        (numpy collapses their base to it), so `busy()` tells whether the caller can still see this slab."""
        a = np.ndarray(self.shapes[name], dtype=self.dtypes[name], buffer=self._shm[name].buf)
        self._handed.append(weakref.ref(a))
        return a

    def busy(self):
        self._handed = [r for r in self._handed if r() is not None]
        return bool(self._handed)

    def barrier(self):
        dist, _, _ = _dist(self.group)
        if dist is not None:
            dist.barrier(group=self.group)

    def close(self, sync=True):
        """Every rank: unpin and unmap.  Rank 0 also removes the segments' names (the other ranks' mappings stay valid until
        they unmap).  sync=False skips the barrier: for error paths and interpreter exit, where the ranks may not all get here."""
        import torch

        for p in self._registered:
            torch.cuda.cudart().cudaHostUnregister(p)
        self._registered = []
        self.arrays, self.tensors = {}, {}
        if sync:
            self.barrier()  # nobody unlinks while another rank still reads
        for s in self._shm.values():
            try:
                s.close()
            except BufferError:
                pass  # the caller still holds an array over it: the mapping lives on with that array
            if self.rank == 0:
                try:
                    s.unlink()
                except FileNotFoundError:
                    pass
        self._shm = {}


# Creating and page-locking the shared segments costs ~0.1 s for a KITTI batch of 64 frames -- far more than the fill and the
# copies.  Slabs are therefore kept and reused while batch shape, outputs and group stay the same (every rank makes the same
# sequence of calls, so all of them keep or replace them together).
_slab_cache = {"key": None, "slabs": []}
_MAX_CACHED = 4


_retired = []  # slabs that left the cache while a caller still held their arrays


def _drop_cached_slabs(sync=True):
    slabs, _slab_cache["slabs"], _slab_cache["key"] = _slab_cache["slabs"], [], None
    for slab in slabs:
        if slab.busy():  # still visible to the caller: unmapped later (or at exit, with the process)
            _retired.append(slab)
            if sync:
                slab.barrier()  # (keeps the collective sequence of close(sync=True) on every rank)
            continue
        try:
            slab.close(sync=sync)
        except Exception:
            pass


def _at_exit():
    """The names of every segment this process created go (the mappings go with the process)."""
    _drop_cached_slabs(sync=False)
    for slab in _retired:
        for s in slab._shm.values():
            if slab.rank == 0:
                try:
                    s.unlink()
                except (FileNotFoundError, OSError):
                    pass


atexit.register(_at_exit)


def _free_slab(n_frames, frame_shape, names, group, dtypes=None):
    """A slab no rank's caller can see any more (collective: every rank calls it with the same arguments).  The ranks agree
    through one small all-reduce; if every cached slab is still held somewhere a new one is made."""
    import torch

    key = (n_frames, tuple(frame_shape), tuple(names), id(group), tuple(sorted((dtypes or {}).items(), key=str)))
    for z in [z for z in _retired if not z.busy()]:  # (rank-local: only this process's mapping and pinning go)
        _retired.remove(z)
        z.close(sync=False)
    if _slab_cache["key"] != key:
        _drop_cached_slabs()
        _slab_cache["key"] = key
    slabs = _slab_cache["slabs"]
    dist, _, world = _dist(group)
    free = [0 if s.busy() else 1 for s in slabs] + [1] * (_MAX_CACHED - len(slabs))
    if dist is not None and world > 1:
        t = torch.tensor(free, dtype=torch.int32)
        if dist.get_backend(group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        free = [int(v) for v in t.cpu().tolist()]
    for i, s in enumerate(slabs):
        if free[i]:
            return s
    if len(slabs) >= _MAX_CACHED:
        # the callers hold every slab: the oldest leaves the cache but stays mapped until this rank's caller has dropped the
        # arrays it was handed from it (unmapping under a live array would leave it pointing into freed address space)
        _retired.append(slabs.pop(0))
    slab = HostSlab(n_frames, frame_shape, names, group=group, dtypes=dtypes)
    slabs.append(slab)
    return slab


def release_host_slab():
    """Unpin and remove the shared host slabs fill_sharded keeps between calls (collective: every rank calls it)."""
    _drop_cached_slabs()


def gather_frames(local, n_frames, group=None, dst=None):
    """Host-side gather of per-rank frame slabs into one [n_frames, ...] array through a shared slab (any dtype).

    local: numpy array [hi-lo, ...] for this rank's shard_range.  With dst=None every rank gets the
    full array; otherwise only rank `dst` does and the others return None."""
    dist, rank, world = _dist(group)
    local = np.asarray(local)
    if dist is None:
        if local.shape[0] != n_frames:
            raise ValueError("single process: local must hold all frames")
        return local
    lo, hi = shard_range(n_frames, rank, world)
    if local.shape[0] != hi - lo:
        raise ValueError("rank %d: expected %d frames, got %d" % (rank, hi - lo, local.shape[0]))
    slab = HostSlab(n_frames, local.shape[1:], ("data",), group=group, pin=False, dtypes={"data": local.dtype})
    try:
        slab.arrays["data"][lo:hi] = local
        slab.barrier()
        out = slab.arrays["data"].copy() if (dst is None or rank == dst) else None
    finally:
        slab.close(sync=True)  # (barrier inside: nobody unlinks while another rank still copies)
    return out


def select_device(local_rank=None):
    """One process per GPU: rank r of the node computes on cuda:(LOCAL_RANK % device_count).  Returns the device; the
    process-wide current device is left alone."""
    import torch

    if not torch.cuda.is_available():
        return None
    if local_rank is None:
        local_rank = int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0")))
    return torch.device("cuda", local_rank % torch.cuda.device_count())


def fill_sharded(x, src_thr=0.1, val_thr=0.1, metric="l1_cv", want=("depth", "dt", "index"),
                 group=None, dst=None, compute=None, timings=None):
    """Every rank passes the same full batch x [B,H,W] (or at least its own shard's frames valid); each computes
    frames shard_range(B, rank, world) on its own GPU (LOCAL_RANK picks it) and copies the results from the device
    straight into its slice of the shared host slab.  Returns dict name -> [B,H,W] array on every rank (dst=None)
    or only on rank `dst` (None elsewhere); the arrays are the slab itself (see the module docstring).  Raises
    IndexError on EVERY rank if any frame of the batch hits numpy's IndexError in depth_list[label_list-1]
    (tools.py:26) and depth is wanted, RuntimeError on every rank if a shard failed in any other way.

    compute(x_shard, src_thr, val_thr, want) -> dict of numpy arrays (plus optional "status") replaces the HIP
    operator; tests inject a checker there.  timings: optional dict that receives "compute_ms" / "gather_ms"."""
    import time

    B, H, W = x.shape
    dist, rank, world = _dist(group)
    lo, hi = shard_range(B, rank, world)
    slab = _free_slab(B, (H, W), tuple(want), group)
    t0 = time.perf_counter()
    t1 = None
    failure = None
    try:
        try:
            if compute is None:
                import torch

                from . import device as _device

                dev = select_device()
                with torch.cuda.device(dev):
                    op = _device.default_op(metric)
                    if hi > lo:
                        xd = op.upload(np.asarray(x)[lo:hi])  # pinned staging by a few threads, the DMA chunk by chunk behind them
                        res = op.run(xd, src_thr, val_thr, want)
                        t1 = time.perf_counter()
                        for k in slab.names:  # device -> this rank's slice of the slab, asynchronously, all outputs in flight
                            slab.tensors[k][lo:hi].copy_(res[k], non_blocking=True)
                        torch.cuda.current_stream(dev).synchronize()
            elif hi > lo:
                local = compute(np.ascontiguousarray(x[lo:hi]), src_thr, val_thr, want)
                t1 = time.perf_counter()
                for k in want:
                    slab.arrays[k][lo:hi] = local[k]
                slab.arrays["status"][lo:hi] = local.get("status", 0)
        except BaseException as e:  # recorded for every rank to see; re-raised below, after the barrier
            failure = e
            if hi > lo and slab.arrays:
                slab.arrays["status"][lo:hi] = FRAME_SHARD_FAILED
    finally:
        if t1 is None:
            t1 = time.perf_counter()
        slab.barrier()  # the whole batch -- or every shard's verdict -- is in the slab; every rank gets here
    t2 = time.perf_counter()
    if timings is not None:
        timings["compute_ms"] = 1e3 * (t1 - t0)
        timings["gather_ms"] = 1e3 * (t2 - t1)
    status = slab.arrays["status"]
    failed = np.nonzero(status & FRAME_SHARD_FAILED)[0]
    if failure is not None or len(failed):
        _drop_cached_slabs(sync=False)  # (every rank takes this branch: the status is shared)
        if failure is not None:
            raise failure
        bad_rank = next(r for r in range(world) if shard_range(B, r, world)[0] <= int(failed[0]) < shard_range(B, r, world)[1])
        raise RuntimeError("fill_sharded: the shard of rank %d (frames from %d) failed; see that rank's exception" % (bad_rank, int(failed[0])))
    bad = np.nonzero(status & FRAME_INDEX_ERROR)[0] if "depth" in want else ()
    if len(bad):
        raise IndexError("frame %d: index out of bounds in depth_list[label_list-1] "
                         "(value list shorter than a label, or empty with label 0)" % int(bad[0]))
    if dst is None or rank == dst:
        return {k: slab.view(k) for k in want}
    return None
