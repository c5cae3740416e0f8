"""A short headline run for counter collection: python scripts/dev/bench_headline_short.py [workload]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x = torch.from_numpy(synth.make(sys.argv[1] if len(sys.argv) > 1 else "kitti_b32")).cuda()
op = pkg.device.DtFill(device="cuda:0")
for _ in range(8): op.run(x)
torch.cuda.synchronize()
