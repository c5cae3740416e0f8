"""Per-kernel times of one workload for every subset of the outputs (which gathers / stores cost what)."""
import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('distancetransform-depthcompletion_amd')
synth = importlib.import_module('distancetransform-depthcompletion_amd.synth')
op = pkg.device.DtFill(device='cuda:0')
for name in sys.argv[1:]:
    x = torch.from_numpy(synth.make(name)).to('cuda:0')
    for want in (("depth", "dt", "index"), ("dt", "index"), ("dt", "depth"), ("dt",), ("index",), ("depth",)):
        acc = {}
        for _ in range(3): op.run(x, want=want)
        for _ in range(5):
            op.run(x, want=want, timed=True)
            for k, v in op.last_kernel_ms.items(): acc[k] = acc.get(k, 0) + v / 5
        print(name, want, {k: round(v * 1e3, 1) for k, v in acc.items() if k in ('k_rows', 'k_fin', 'k_colT', 'k_fused')}, flush=True)
