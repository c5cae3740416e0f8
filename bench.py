#!/usr/bin/env python3
"""bench.py -- DT + NN-fill frames/s at 352x1216 on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (dtfill_batch through the C ABI) over one batch of
B=32 synthetic KITTI-shaped frames per GPU (BASELINE.json configs[1]; 5 % valid pixels, depths on
KITTI's k/256 grid), inputs resident in HBM before the timed region, all three outputs (filled
depth, distance map, NN index map) written to HBM.  N>1: one process per GPU, each with its own 32
frames (frames are independent: no data-path collective; "scaling": "weak"), timing bracketed by a
barrier + synchronize, MAX over ranks.  `python bench.py --gpus N` without a launcher starts the N
ranks itself (torch.distributed.run as a child process, before this process touches a GPU).
Untimed, in this order: a spin-up until the pass time has settled (300 to 3000 passes, `spinup_passes` in the line: the GPU
runs its first few hundred passes after an idle spell slower), the W warm-up steps, then exactly K timed steps.

Prints ONE JSON line on rank 0, with
  roofline      the slowest kernel of the pass, timed with HIP events on the launch stream:
                achieved = its algorithmic bytes (its bytes per pixel x the pixels it processed in this
                pass, dtfill_pass_stats) / its mean duration, vs 8 TB/s HBM peak; `kernels` holds the
                same figure for every kernel of the pass, `pass_frac` the whole pass at 16 B/pixel;
                traffic = HBM bytes per launch of that kernel from the committed rocprofv3 PMC
                passes (profiles/traffic.json, keyed by workload)
  cpu_baseline  the CPU restatement of the reference path (oracle/, kind "port") timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1 only):
                all cores (value) and one thread (single_thread)
  workloads     (N=1) the other single-GPU configurations of BASELINE.json through the same code:
                scan-line KITTI (LiDAR-like: empty sky, ring rows), NYU 480x640 B=64 (config 4),
                2048^2 1 % B=16 (the per-GPU share of config 5), each with its own roofline
  repeat        the timed loop repeated (median / min / max of ms per step)
  end_to_end    what a reference-style caller sees: numpy in / numpy out through
                DT_complete_batch, PCIe included (never `value`), beside the measured pinned
                host <-> device copy rates of the box and the bound they put on one synchronous call
  sharded       (N>1, or --sharded) fill_sharded: per-rank D2H straight into one shared pinned
                host slab; compute and gather time
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PKG = "distancetransform-depthcompletion_amd"
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_PIXEL = 16   # read f32 depth, write f32 depth + f32 distance + i32 index
EXTRA_WORKLOADS = ("kitti_b32_scanline", "nyu_b64", "synth2048_b16")
XSTEPS = 50  # timed passes of the extra workloads (behind the spin-up and 10 warm-up passes)


def host_cores():
    """CPUs this process may actually use (affinity / cgroup quota), not every core of the host."""
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(cores, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return min(cores, 64)


def cpu_baseline(x, cpu_seconds=20.0):
    """Oracle (CPU port of the reference path) over frames of x: one frame per task on a thread pool over
    all host cores (ctypes releases the GIL; OpenCV's labels transform is serial per frame, so frames are
    the unit of parallelism), and on one thread."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O

    O.lib()
    cores = host_cores()
    B = x.shape[0]
    t0 = time.perf_counter()
    O.fill_batch(x[:1])
    per_frame = time.perf_counter() - t0
    # one thread: ~a third of the budget
    n1 = int(max(2, cpu_seconds / 3 / max(per_frame, 1e-4)))
    t0 = time.perf_counter()
    for i in range(n1):
        O.fill_batch(x[i % B : i % B + 1])
    dt1 = time.perf_counter() - t0
    n = int(max(cores, cpu_seconds * 2 / 3 / max(per_frame, 1e-4) * min(cores, 4)))
    frames = [x[i % B : i % B + 1] for i in range(n)]
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        list(ex.map(lambda f: O.fill_batch(f), frames))
    dt = time.perf_counter() - t0
    return {
        "value": round(n / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
        "single_thread": round(n1 / dt1, 2),
        "sample": "%d frames %dx%d (the bench batch, cycled), C restatement of cv2 L1/5x5 labels transform + "
                  "tools.py glue, one frame per thread, %d threads, %.1f s; single_thread: %d frames, %.1f s"
                  % (n, x.shape[1], x.shape[2], cores, dt, n1, dt1),
    }


def load_traffic():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except (OSError, ValueError):
        return {}


def measure(op, torch, x, steps, warmup, path, barrier, reduce_max, repeats=0, workload=None, traffic=None):
    """W untimed passes, then exactly `steps` timed ones between barrier + synchronize; per-kernel HIP-event
    times from a few instrumented passes.  Returns (line fields, roofline dict)."""
    B, H, W = x.shape
    # Spin-up, before the contract's W warm-up steps and just as untimed: the first few hundred passes after an idle spell run
    # slower (clocks, and whatever the process before left behind: 96 us per pass measured behind 10 passes, 90.7 behind 300
    # and ever after, once 125 us behind 300 right after a profiler run), and the driver's "--warmup 5 --steps 20" would time
    # exactly those.  At least 300 passes, then chunks of 100 until two in a row agree within 1.5 %, at most 3000 (0.3 s); the
    # count is reported as "spinup_passes".  DTFILL_BENCH_NO_SPINUP=1 switches it off.
    spinup, prev = 0, None
    while spinup < 3000 and not os.environ.get("DTFILL_BENCH_NO_SPINUP"):
        barrier()
        t0 = time.perf_counter()
        for _ in range(100):
            op.run(x, path=path)
        barrier()
        dt_ = reduce_max(time.perf_counter() - t0)
        spinup += 100
        if spinup >= 300 and prev is not None and abs(dt_ - prev) <= 0.015 * prev:
            break
        prev = dt_
    for _ in range(warmup):
        op.run(x, path=path)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        op.run(x, path=path)
    barrier()
    elapsed = reduce_max(time.perf_counter() - t0)
    rep = []
    for _ in range(repeats):
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            op.run(x, path=path)
        barrier()
        rep.append(1e3 * reduce_max(time.perf_counter() - t0) / steps)
    acc = {}
    reps = 10
    for _ in range(reps):
        op.run(x, timed=True, path=path)
        for k, v in op.last_kernel_ms.items():
            acc[k] = acc.get(k, 0.0) + v / reps
    st_ = op.run(x, path=path)["status"]
    status_bad = int(((st_ & 1) != 0).sum().item())
    general_frames = int(((st_ & 2) != 0).sum().item())
    px = op.pass_stats()  # which kernel family owned how many pixels of that pass
    torch.cuda.synchronize()
    ms_per_step = 1e3 * elapsed / steps
    # a kernel whose blocks all exit at once still costs its dispatch (~6 us between two events): not a candidate
    live = {k: v for k, v in acc.items() if v > 0}
    algo_bytes = BYTES_PER_PIXEL * B * H * W
    # every kernel is credited with the algorithmic bytes of the pixels IT processed (bytes per pixel of
    # each kernel x the pixels dtfill_pass_stats says it owned), not with the whole batch
    win = max((k for k in live if k.startswith("k_l2win")), key=lambda k: live[k], default=None)
    # (the tiles of the frames with a handful of sources -- k_pts, 12 B/px: they read a source list, not the frame -- ride in
    # k_fin's launch)
    per_px = {"k_mask": [(4.3, px["all"])], "k_frame": [], "k_fused": [(16.0, px["window"])],
              "k_colT": [(0.38, px["colt"]), (12.0, px["sky"])],  # (k_sky's blocks ride in k_colT's launch: 12 B per sky pixel)
              "k_rows": [(8.85, px["anydist"])], "k_fin": [(20.6, px["anydist"]), (12.0, px["points"])], "k_tiesx": [],
              "k_l2far": [], "k_l2env": [(20.25, px["anydist"] + px["points"])]}
    if win:
        per_px[win] = [(16.13, px["window"])]
    kern = {}
    for k, ms in live.items():
        parts = per_px.get(k, [])
        nbytes, n = sum(bpp * m for bpp, m in parts), sum(m for _, m in parts)
        kern[k] = {"ms": round(ms, 4), "algorithmic_bytes": int(nbytes), "pixels": int(n),
                   "frac": round(nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
    dom = max(live, key=live.get)
    achieved = kern[dom]["algorithmic_bytes"] / (live[dom] * 1e-3) / 1e9
    tr = None
    t = (traffic or {}).get(workload or "", {})  # keys of profiles/traffic.json: <workload> or <workload>_l2
    if t.get("batch") == B and dom in t.get("kernels", {}):
        k = t["kernels"][dom]
        tr = k["fetch_bytes"] + k["write_bytes"]
    roof = {
        "bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": tr, "algorithmic_bytes": kern[dom]["algorithmic_bytes"],
        "pass_algorithmic_bytes": algo_bytes,
        "kernel_ms": {k: round(v, 4) for k, v in live.items()},
        "kernels": kern, "pixels_by_path": px,
        "pass_achieved_GBs": round(algo_bytes / (ms_per_step * 1e-3) / 1e9, 1),
        "pass_frac": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "pass_traffic": t.get("pass_bytes") if t.get("batch") == B else None,
    }
    out = {"ms_per_step": round(ms_per_step, 4), "elapsed": elapsed, "spinup_passes": spinup, "frames_with_index_error": status_bad,
           "frames_on_general_path": general_frames}
    if rep:
        srt = sorted(rep)
        out["repeat"] = {"n": len(rep), "median_ms_per_step": round(srt[len(srt) // 2], 4),
                         "min_ms_per_step": round(srt[0], 4), "max_ms_per_step": round(srt[-1], 4)}
    return out, roof


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10, help="untimed passes right before the timed ones (behind the spin-up, see measure())")
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default: the workload's)")
    ap.add_argument("--workload", default="kitti_b32")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline workload only (no workloads / end_to_end keys)")
    ap.add_argument("--sharded", action="store_true", help="also time fill_sharded (host slab gather) at N=1")
    ap.add_argument("--path", default="auto", choices=["auto", "general", "fused"])
    ap.add_argument("--metric", default="l1_cv", choices=["l1_cv", "l2"],
                    help="l1_cv = the reference's cv2 transform (headline); l2 = exact Euclidean")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        # no launcher: start the ranks ourselves, BEFORE anything here touches a GPU (never exec from a GPU process)
        import socket

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != args.gpus:
        sys.exit("bench.py: WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import numpy as np
    import torch

    dist = None
    if world > 1 or os.environ.get("DTFILL_BENCH_FORCE_DIST"):  # the latter: exercise the RCCL path with one rank
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # one rank per GPU over RCCL.  DTFILL_BENCH_BACKEND=gloo is a rehearsal mode for a box with fewer GPUs
        # than ranks (the ranks then share devices and the barrier / max run over gloo on the host).
        backend = os.environ.get("DTFILL_BENCH_BACKEND", "nccl")
        ndev = torch.cuda.device_count()
        if backend != "gloo" and local_rank >= ndev:
            sys.exit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, ndev))
        torch.cuda.set_device(local_rank % ndev)
        # the collective libraries print their connection chatter on the C-level stdout: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group("gloo")
            dist.barrier()  # the first collective sets the connections up
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", torch.cuda.current_device())

    pkg = importlib.import_module(PKG)
    synth = importlib.import_module(PKG + ".synth")
    op = pkg.device.DtFill(device=dev, metric=args.metric)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if dist is None:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    traffic = load_traffic()
    cfg = synth.CONFIGS[args.workload]
    B = args.batch or cfg["B"]
    xh = synth.make(args.workload, B=B, seed=cfg["kwargs"]["seed"] + rank)
    _, H, W = xh.shape
    x = torch.from_numpy(xh).to(dev)
    sfx = "_l2" if args.metric == "l2" else ""
    res, roof = measure(op, torch, x, args.steps, args.warmup, args.path, barrier, reduce_max, repeats=5,
                        workload=args.workload + sfx, traffic=traffic)

    sharded = None
    if world > 1 or args.sharded:
        # the numpy-out contract on N GPUs: every rank computes its shard of ONE batch of world * B frames and copies it
        # from its GPU straight into its slice of the shared pinned host slab
        xs = np.concatenate([synth.make(args.workload, B=B, seed=cfg["kwargs"]["seed"] + r) for r in range(world)])
        tm = {}
        pkg.fill_sharded(xs, metric=args.metric, dst=0, timings=tm)  # warm (allocations, pinning)
        t0 = time.perf_counter()
        pkg.fill_sharded(xs, metric=args.metric, dst=0, timings=tm)
        total = reduce_max(time.perf_counter() - t0)
        sharded = {"frames": int(xs.shape[0]), "total_ms": round(1e3 * total, 2),
                   "h2d_compute_ms": round(reduce_max(tm["compute_ms"]), 2),
                   "d2h_into_slab_ms": round(reduce_max(tm["gather_ms"]), 2),
                   "frames_per_s_end_to_end": round(xs.shape[0] / total, 1)}
        # ... and with the one output DT_complete_batch returns (tools.py:13-35), for comparison with `end_to_end`
        pkg.fill_sharded(xs, metric=args.metric, dst=0, want=("depth",))
        t0 = time.perf_counter()
        pkg.fill_sharded(xs, metric=args.metric, dst=0, want=("depth",))
        sharded["depth_only_frames_per_s"] = round(xs.shape[0] / reduce_max(time.perf_counter() - t0), 1)

    if rank == 0:
        frames = world * B * args.steps
        line = {
            "metric": "DT+NN-fill frames/sec at 352x1216; achieved HBM GB/s vs peak",
            "value": round(frames / res["elapsed"], 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "spinup_passes": res["spinup_passes"],
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",  # packed integer keys / 32-pixel bit planes; int32 index, f32 depth + distance only at the stores
            "data": "synthetic",
            "config": {
                "workload": "%s: B=%d frames/GPU of %dx%d, %s" % (args.workload, B, H, W, json.dumps(cfg["kwargs"])),
                "metric_mode": args.metric, "outputs": "depth+dt+index", "frames_per_gpu": B,
                "parallelism": "frame-sharded x%d, no collective" % world,
            },
            "roofline": roof,
            "repeat": res.get("repeat"),
            "frames_with_index_error": res["frames_with_index_error"],
            "frames_on_general_path": res["frames_on_general_path"],
        }
        if sharded:
            line["sharded"] = sharded
        if world == 1 and not args.no_extras and args.metric == "l1_cv" and args.workload == "kitti_b32":
            wl = {}
            for name in EXTRA_WORKLOADS:
                c = synth.CONFIGS[name]
                xw = torch.from_numpy(synth.make(name)).to(dev)
                r, rf = measure(op, torch, xw, XSTEPS, 10, args.path, barrier, reduce_max, workload=name, traffic=traffic)
                wl[name] = {"value": round(c["B"] * XSTEPS / r["elapsed"], 1), "unit": "frames/s", "frames": c["B"],
                            "shape": [c["H"], c["W"]], "ms_per_step": r["ms_per_step"], "roofline": rf,
                            "frames_on_general_path": r["frames_on_general_path"]}
                del xw
            line["workloads"] = wl
            # the Euclidean mode (north_star's transform; the reference itself only calls the L1 one) on the same batches
            op2 = pkg.device.DtFill(device=dev, metric="l2")
            l2 = {}
            for name in (args.workload,) + tuple(EXTRA_WORKLOADS):
                c = synth.CONFIGS[name]
                xw = x if name == args.workload else torch.from_numpy(synth.make(name)).to(dev)
                r, rf = measure(op2, torch, xw, XSTEPS, 10, "auto", barrier, reduce_max, workload=name + "_l2", traffic=traffic)
                l2[name] = {"value": round(xw.shape[0] * XSTEPS / r["elapsed"], 1), "unit": "frames/s", "frames": int(xw.shape[0]),
                            "ms_per_step": r["ms_per_step"], "roofline": rf}
                del xw
            line["l2"] = l2
            del op2
            # numpy in / numpy out through the reference-named function (H2D + pass + D2H; returns the pinned DMA targets)
            link = pkg.device.host_link_bandwidth()
            e2e = {}
            for nb in (1, B):
                batch = xh[:nb, :, :, None]
                pkg.DT_complete_batch(batch)
                n = 20 if nb == 1 else 8
                runs = []  # five runs of n calls, the median reported (a host hiccup -- the staging threads share the box's
                for _ in range(5):  # cores with whatever else runs there -- once read 8.5 k for 12.5 k frames/s)
                    t0 = time.perf_counter()
                    for _ in range(n):
                        pkg.DT_complete_batch(batch)
                    runs.append(nb * n / (time.perf_counter() - t0))
                e2e["B%d" % nb] = round(sorted(runs)[2], 1)
                e2e["B%d_runs" % nb] = [round(v, 1) for v in runs]
            # what one synchronous call cannot beat: its input over the link, the pass, its output (depth only) back
            fb = 4 * H * W
            bound = 1.0 / (fb / (link["h2d"] * 1e9) + res["ms_per_step"] * 1e-3 / B + fb / (link["d2h"] * 1e9))
            line["end_to_end"] = {"what": "DT_complete_batch numpy->numpy incl. PCIe, frames/s", **e2e,
                                  "link_GBs": {k: round(v, 1) for k, v in link.items()},
                                  "bound_frames_per_s": round(bound, 1), "B%d_of_bound" % B: round(e2e["B%d" % B] / bound, 3)}
        if world == 1 and not args.no_cpu_baseline and args.metric == "l1_cv":
            line["cpu_baseline"] = cpu_baseline(xh)
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
