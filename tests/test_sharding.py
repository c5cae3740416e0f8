"""N>1 path on the CPU: two gloo ranks shard a batch, each fills its frames, the host gather
reassembles them in order.  The per-rank compute is injected (the oracle stands in for the GPU
here -- in tests only), so what is under test is shard_range + gather_frames + fill_sharded: the shared slab, its rotation
(returned arrays are never written again while anyone holds them), and the error paths (the reference's IndexError and any
other failure of one rank's shard reach every rank; nobody waits in a collective)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range(pkg):
    for n in (0, 1, 7, 32, 33, 256):
        for world in (1, 2, 3, 8):
            spans = [pkg.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert pkg.shard_range(256, 3, 8) == (96, 128)
    with pytest.raises(ValueError):
        pkg.shard_range(8, 2, 2)


def _worker(rank, world, port, n_frames, q):
    sys.path.insert(0, ROOT)
    import importlib

    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pkg = importlib.import_module("distancetransform-depthcompletion_amd")
    from oracle import oracle as O

    rng = np.random.default_rng(0)  # same batch on every rank
    x = np.where(rng.random((n_frames, 24, 40)) < 0.06, rng.uniform(1, 80, (n_frames, 24, 40)), 0).astype(np.float32)

    def compute(xs, st, vt, want):
        depth, dt, idx, status = O.fill_batch(xs, st, vt)
        return {"depth": depth, "dt": dt, "index": idx, "status": status}

    full = pkg.fill_sharded(x, compute=compute)  # all_gather form
    root_only = pkg.fill_sharded(x, compute=compute, dst=0)
    depth, dt, idx, _ = O.fill_batch(x)
    ok = all(np.array_equal(full[k], v) for k, v in (("depth", depth), ("dt", dt), ("index", idx)))
    if rank == 0:
        ok = ok and np.array_equal(root_only["index"], idx)
    else:
        ok = ok and root_only is None
    # a frame that makes numpy raise IndexError (no source, empty value list) in the LAST rank's shard: every rank must
    # raise, none may hang in a collective (the reference raises once for the whole batch, tools.py:26)
    xb = x.copy()
    xb[n_frames - 1] = 0
    try:
        pkg.fill_sharded(xb, compute=compute)
        ok = False
    except IndexError as e:
        ok = ok and ("frame %d" % (n_frames - 1)) in str(e)
    only_dt = pkg.fill_sharded(xb, compute=compute, want=("dt",))  # no depth wanted: no error, as in the reference's nearest_point
    ok = ok and np.array_equal(only_dt["dt"], O.fill_batch(xb)[1])
    # the arrays that come back ARE the shared slab: a later call must not write into what an earlier one returned (the
    # reference returns fresh arrays, tools.py:29-35) -- a slab is reused only after every rank has dropped its arrays
    first = pkg.fill_sharded(x, compute=compute)
    keep = {k: v.copy() for k, v in first.items()}
    x2 = np.roll(x, 1, axis=0)
    second = pkg.fill_sharded(x2, compute=compute)
    third = pkg.fill_sharded(x2, compute=compute)
    ok = ok and all(np.array_equal(first[k], keep[k]) for k in keep)
    ok = ok and np.array_equal(second["index"], O.fill_batch(x2)[2]) and np.array_equal(third["index"], second["index"])
    row = first["depth"][0]  # a view keeps its slab busy on its own
    del first, second, third
    again = [pkg.fill_sharded(x2, compute=compute) for _ in range(6)]
    ok = ok and np.array_equal(row, keep["depth"][0]) and all(np.array_equal(a["index"], O.fill_batch(x2)[2]) for a in again)
    del again
    # a shard that fails with something other than the reference's IndexError (a HIP error, a bad shape, no memory): the
    # failing rank raises its own exception, every other rank a RuntimeError -- and none waits in a collective
    def broken(xs, st, vt, want):
        if rank == world - 1:
            raise ValueError("injected shard failure")
        return compute(xs, st, vt, want)

    try:
        pkg.fill_sharded(x, compute=broken)
        ok = False
    except ValueError as e:
        ok = ok and rank == world - 1 and "injected" in str(e)
    except RuntimeError as e:
        ok = ok and rank != world - 1 and "rank %d" % (world - 1) in str(e)
    ok = ok and np.array_equal(pkg.fill_sharded(x, compute=compute)["index"], idx)  # and the next call works again
    # gather_frames under the group: any dtype, every rank or one
    lo, hi = pkg.shard_range(n_frames, rank, world)
    for dt_ in (np.int32, np.float32, np.float64, np.uint16):
        data = (np.arange(n_frames * 6).reshape(n_frames, 2, 3) * 3).astype(dt_)
        got = pkg.gather_frames(data[lo:hi], n_frames)
        ok = ok and got.dtype == dt_ and np.array_equal(got, data)
        got0 = pkg.gather_frames(data[lo:hi], n_frames, dst=0)
        ok = ok and ((got0 is None) if rank else np.array_equal(got0, data))
    leftover = [f for f in os.listdir("/dev/shm") if f.startswith("psm_")] if rank == 0 else []
    pkg.release_host_slab()
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 8])
def test_two_rank_gloo_gather(n_frames):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_frames, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == {0: True, 1: True}
