import importlib, sys
sys.path.insert(0, ".")
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x = synth.make("kitti_b32_scanline", seed=0)
rng = np.random.default_rng(0)
xs = x.copy()
for b in range(x.shape[0]):
    for _ in range(5):
        xs[b, rng.integers(0, 90), rng.integers(0, x.shape[2])] = 9.0
op = pkg.DtFill("cuda:0")
for name, arr in (("clean sky", x), ("5 stray points per frame", xs)):
    xd = torch.from_numpy(arr).cuda()
    for _ in range(5): op.run(xd)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): op.run(xd)
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us/pass" % (e0.elapsed_time(e1) / 50 * 1e3))
