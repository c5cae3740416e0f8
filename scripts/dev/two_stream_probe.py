"""Probe: one 32-frame KITTI batch as ONE pass on one stream vs TWO half-batches on two streams (fork/join with
events).  Tells whether overlapping the halves (k_mask of one under k_fused of the other, de-phased tiles) pays."""
import importlib
import sys
import torch

sys.path.insert(0, ".")
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")

dev = torch.device("cuda:0")
x = torch.from_numpy(synth.make("kitti_b32", seed=0)).to(dev)
B = x.shape[0]
nsplit = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ops = [pkg.DtFill(dev) for _ in range(nsplit)]
one = pkg.DtFill(dev)
streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
parts = [x[i * B // nsplit:(i + 1) * B // nsplit].contiguous() for i in range(nsplit)]
main = torch.cuda.current_stream(dev)


def single():
    one.run(x, 0.0, 0.0)


def split():
    ev = torch.cuda.Event()
    ev.record(main)
    for s, op, p in zip(streams, ops, parts):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            op.run(p, 0.0, 0.0)
        e2 = torch.cuda.Event()
        e2.record(s)
        main.wait_event(e2)


def timeit(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def graphed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g.replay


def split_in_capture():
    cur = torch.cuda.current_stream(dev)
    ev = torch.cuda.Event()
    ev.record(cur)
    for s, op, p in zip(streams, ops, parts):
        s.wait_event(ev)
        with torch.cuda.stream(s):
            op.run(p, 0.0, 0.0)
        e2 = torch.cuda.Event()
        e2.record(s)
        cur.wait_event(e2)


g_single, g_split = graphed(single), graphed(split_in_capture)
for name, fn in (("single", single), ("g_single", g_single), ("g_split%d" % nsplit, g_split), ("g_single", g_single),
                 ("g_split%d" % nsplit, g_split)):
    ms = timeit(fn)
    print(f"{name:8s} {ms*1e3:7.1f} us/pass  {B/ms*1e3:9.0f} frames/s")
