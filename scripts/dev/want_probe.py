"""k_fused time as a function of which outputs are requested (what the three stores and the gather cost)."""
import importlib, sys
sys.path.insert(0, ".")
import torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x = torch.from_numpy(synth.make("kitti_b32", seed=0)).cuda()
op = pkg.DtFill("cuda:0")
for want in (("depth", "dt", "index"), ("depth", "dt"), ("depth",), ("dt",), ("index",), ("dt", "index")):
    ts = []
    for _ in range(12):
        op.run(x, 0.0, 0.0, want, timed=True, path="fused")
        ts.append(op.last_kernel_ms["k_fused"])
    ts = sorted(ts[2:])
    print("%-28s k_fused %.1f us (median of 10)" % ("+".join(want), 1e3 * ts[len(ts) // 2]))
