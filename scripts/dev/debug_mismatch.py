"""Debug helper: where does the HIP path differ from the oracle? (run on the GPU box)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
from oracle import oracle as O
name = sys.argv[1] if len(sys.argv) > 1 else "kitti_b32"
path = sys.argv[2] if len(sys.argv) > 2 else "auto"
x = synth.make(name, B=2)
op = pkg.device.DtFill(device="cuda:0")
res = op.run(torch.from_numpy(x).cuda(), path=path); torch.cuda.synchronize()
depth, dt, lbl, st = O.fill_batch(x)
got = res["index"].cpu().numpy(); gdt = res["dt"].cpu().numpy()
bad = np.argwhere(got != lbl)
print("status", res["status"].cpu().numpy(), "dt mismatches", (gdt != dt).sum(), "label mismatches", len(bad))
H, W = x.shape[1:]
nty, ntx = -(-H // 88), -(-W // 160); TH, TW = -(-H // nty), -(-W // ntx)
for b, i, j in bad[:25]:
    print("frame", b, "px", (i, j), "tile", (i // TH, j // TW), "in-tile", (i % TH, j % TW), "d", dt[b, i, j], "got", got[b, i, j], "want", lbl[b, i, j])
if len(bad):
    ii = bad[:, 1] % TH; jj = bad[:, 2] % TW
    print("in-tile row hist", np.bincount(ii, minlength=TH)); print("in-tile col hist", np.bincount(jj, minlength=TW))
