"""Development probe: the first failing case of test_seeded_random_vs_oracle, with the mismatching pixels."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
from oracle import oracle as O
op = pkg.device.DtFill(device="cuda:0")
rng = np.random.default_rng(2024)
for t in range(40):
    B = int(rng.integers(1, 4)); H, W = int(rng.integers(1, 90)), int(rng.integers(1, 200))
    p = rng.choice([0.003, 0.02, 0.05, 0.3, 0.8])
    x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
    if t % 5 == 0: x[:, : H // 2] = 0
    if t % 7 == 0: x[:, :, W // 3:] = 0
    depth, dt, lbl, st = O.fill_batch(x)
    r = op.run(torch.from_numpy(x).cuda(), path="auto"); torch.cuda.synchronize()
    gi = r["index"].cpu().numpy(); gd = r["dt"].cpu().numpy()
    if not np.array_equal(gi, lbl):
        bad = np.argwhere(gi != lbl)
        print("case", t, (B, H, W), p, "status", r["status"].cpu().numpy(), "n bad", len(bad), "dt bad", int((gd != dt).sum()))
        print("rows of bad px:", sorted(set(bad[:, 1].tolist())), "cols min/max", bad[:, 2].min(), bad[:, 2].max())
        for b_, i, j in bad[:12]: print("  px", (i, j), "d", dt[b_, i, j], "got", gi[b_, i, j], "want", lbl[b_, i, j])
        break
