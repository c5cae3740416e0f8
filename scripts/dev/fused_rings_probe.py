"""Dev probe: k_fused's time on ring-structured frames against iid frames of the same density (run on the GPU box)."""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
op = pkg.device.DtFill(device="cuda:0")
rng = np.random.default_rng(0)
def rings(B, H, W, top, step, p):
    a = np.where(rng.random((B, H, W)) < p, rng.uniform(1.0, 80.0, (B, H, W)), 0.0).astype(np.float32)
    keep = np.zeros(H, bool); keep[top::step] = True
    a[:, ~keep] = 0
    return a
cases = {
    "iid 5%": synth.kitti_iid(32, 0.05),
    "iid 6.25%": synth.kitti_iid(32, 0.0625),
    "rings top0 step4 p.25": rings(32, 352, 1216, 0, 4, 0.25),
    "rings top0 step2 p.125": rings(32, 352, 1216, 0, 2, 0.125),
    "rings top0 step8 p.5": rings(32, 352, 1216, 0, 8, 0.5),
    "rings top100 step4 p.25 (scanline)": rings(32, 352, 1216, 100, 4, 0.25),
    "rows0 p.0625 cols every 4th": None,
}
a = np.where(rng.random((32, 352, 1216)) < 0.25, rng.uniform(1.0, 80.0, (32, 352, 1216)), 0.0).astype(np.float32)
keep = np.zeros(1216, bool); keep[::4] = True; a[:, :, ~keep] = 0
cases["rows0 p.0625 cols every 4th"] = a
for name, x in cases.items():
    xd = torch.from_numpy(x).to("cuda:0")
    for _ in range(3): op.run(xd)
    acc = {}
    for _ in range(10):
        op.run(xd, timed=True)
        for k, v in op.last_kernel_ms.items(): acc[k] = acc.get(k, 0) + v / 10
    st = op.run(xd)["status"].cpu().numpy()
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20): op.run(xd)
    t1.record(); torch.cuda.synchronize()
    print("%-40s pass %.1f us  general %d/32  %s" % (name, 1e3 * t0.elapsed_time(t1) / 20, int(((st & 2) != 0).sum()),
          {k: round(1e3 * v, 1) for k, v in acc.items()}), flush=True)
