"""Does replaying the pass from a HIP graph beat launching its kernels one by one?  (GPU box)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
for metric in ("l1_cv", "l2"):
    op = pkg.device.DtFill(device="cuda:0", metric=metric)
    x = torch.from_numpy(synth.make("kitti_b32")).cuda()
    for _ in range(10):
        op.run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        op.run(x)
    torch.cuda.synchronize()
    plain = (time.perf_counter() - t0) / 200
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        op.run(x)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        op.run(x)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 200
    print(metric, "plain %.2f us  graph %.2f us" % (plain * 1e6, graph * 1e6))
