"""Development probe: what one pass costs the host to enqueue (seven launches through ctypes) against what it costs the GPU."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x = torch.from_numpy(synth.make("kitti_b32")).cuda()
op = pkg.device.DtFill(device="cuda:0")
for _ in range(500): op.run(x)
torch.cuda.synchronize()
for n in (20, 200, 1000):
    t0 = time.perf_counter()
    for _ in range(n): op.run(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("n=%4d  enqueue %.1f us/pass   total %.1f us/pass" % (n, 1e6 * (t1 - t0) / n, 1e6 * (t2 - t0) / n))
