"""Randomised soak: HIP path (auto + general, and l2; the fused outlier filter on both paths, with negative values; the depth
epilogue; every subset of the outputs) against the oracle on many random frames.
Usage on the GPU box: python scripts/soak.py [n_cases] [seed]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
op = pkg.device.DtFill(device="cuda:0"); op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
bad = 0; t0 = time.time(); npx = 0
for t in range(n):
    kind = t % 6
    if kind == 0:   # KITTI-size, densities around the fused halos' limits
        B, H, W = 2, 352, 1216; p = float(rng.choice([0.008, 0.015, 0.03, 0.05, 0.1]))
    elif kind == 1:
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 500)), int(rng.integers(1, 900)); p = float(rng.choice([0.002, 0.02, 0.06, 0.3]))
    elif kind == 2:  # tile-seam sizes
        B, H, W = 1, int(rng.choice([87, 88, 89, 95, 96, 97, 176, 192])), int(rng.choice([151, 152, 153, 159, 160, 161, 304, 320])); p = 0.05
    elif kind == 3:
        B, H, W = 3, 240, 320; p = float(rng.choice([0.001, 0.005, 0.05]))
    elif kind == 4:  # a handful of points: long distances, long tie chains (diagonal pairs make whole quadrants tie)
        B, H, W = 2, int(rng.integers(40, 400)), int(rng.integers(40, 700)); p = 0.0
    else:            # LiDAR rings: sources on every 4th row below an empty sky
        B, H, W = 2, 352, int(rng.choice([1216, 1242, 640])); p = 0.25
    x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
    if kind == 4:
        for bb in range(B):
            for _ in range(int(rng.integers(1, 30))):
                i, j = int(rng.integers(0, H)), int(rng.integers(0, W))
                x[bb, i, j] = rng.uniform(1, 80)
                if rng.random() < 0.5:  # a partner on an exact diagonal
                    k = int(rng.integers(1, 40)); sgn = int(rng.choice([-1, 1]))
                    if 0 <= i + k < H and 0 <= j + sgn * k < W: x[bb, i + k, j + sgn * k] = rng.uniform(1, 80)
    if kind == 5:
        rows = np.zeros(H, bool); rows[int(rng.integers(60, 140))::4] = True
        x *= rows[None, :, None]
    if rng.random() < 0.4:
        a, b_ = sorted(rng.integers(0, H + 1, 2)); x[:, a:b_] *= (rng.random((B, 1, 1)) < 0.5)
    if rng.random() < 0.3:
        a, b_ = sorted(rng.integers(0, W + 1, 2)); x[:, :, a:b_] = 0
    if rng.random() < 0.3 and H > 120:  # LiDAR sky: an empty band on top (band mode and its ways out)
        x[:, :int(rng.integers(20, H - 40))] *= (rng.random((B, 1, 1)) < 0.7)
    depth, dt, lbl, st = O.fill_batch(x)
    xd = torch.from_numpy(x).cuda()
    for path in ("auto", "general"):
        r = op.run(xd, path=path); torch.cuda.synchronize()
        ok = (np.array_equal(r["dt"].cpu().numpy(), dt) and np.array_equal(r["index"].cpu().numpy(), lbl)
              and np.array_equal(r["status"].cpu().numpy() & 1, st)
              and np.array_equal(r["depth"].cpu().numpy()[st == 0], depth[st == 0], equal_nan=True))
        if not ok:
            bad += 1; print("MISMATCH case", t, path, (B, H, W), p, "labels differ:", int((r["index"].cpu().numpy() != lbl).sum()))
            gi = r["index"].cpu().numpy(); gd = r["dt"].cpu().numpy()
            print("   status", r["status"].cpu().numpy(), "dt differ:", int((gd != dt).sum()))
            for bb, ii, jj in np.argwhere(gi != lbl)[:10]:
                print("   frame", bb, "px", (ii, jj), "d", dt[bb, ii, jj], "got", gi[bb, ii, jj], "want", lbl[bb, ii, jj])
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez_compressed("gpurun_out/mismatch_%d_%s.npz" % (t, path), x=x, got=gi, want=lbl, dt=dt)
    if t % 2 == 0:  # the Euclidean mode: window kernels + far list / envelope rows (auto), every row by the envelope search (general)
        d2, dt2, idx2, st2 = O.fill_batch(x, metric="l2")
        for path in ("auto", "general"):
            r = op2.run(xd, path=path); torch.cuda.synchronize()
            if not (np.array_equal(r["index"].cpu().numpy(), idx2) and np.allclose(r["dt"].cpu().numpy(), dt2, rtol=1e-6, atol=0)
                    and np.array_equal(r["depth"].cpu().numpy()[st2 == 0], d2[st2 == 0], equal_nan=True)):
                bad += 1; print("L2 MISMATCH case", t, path, (B, H, W), p, "labels differ:", int((r["index"].cpu().numpy() != idx2).sum()))
                os.makedirs("gpurun_out", exist_ok=True)
                np.savez_compressed("gpurun_out/mismatch_l2_%d_%s.npz" % (t, path), x=x, got=r["index"].cpu().numpy(), want=idx2)
    if t % 7 == 0 and H >= 4 and W >= 4:  # outlier_removal() in front of the predicates == the two passes composed
        xo = x.copy()
        if t % 14 == 0:  # negative values: the exhaustive second launch (every pixel a candidate)
            xo[rng.random(x.shape) < 0.002] = -rng.uniform(1, 40)
        xf = np.stack([O.outlier_removal(f) for f in xo]).astype(np.float32)
        depth_f, dt_f, lbl_f, st_f = O.fill_batch(xf)
        for path in ("auto", "general"):
            r = op.run(torch.from_numpy(xo).cuda(), outlier_removal=True, path=path); torch.cuda.synchronize()
            if not (np.array_equal(r["index"].cpu().numpy(), lbl_f) and np.array_equal(r["dt"].cpu().numpy(), dt_f)
                    and np.array_equal(r["depth"].cpu().numpy()[st_f == 0], depth_f[st_f == 0], equal_nan=True)):
                bad += 1; print("FUSED OUTLIER MISMATCH case", t, path, (B, H, W), p)
    if t % 5 == 0:  # the drivers' post-fill steps folded into the depth stores (demo.py:292-293, eval_NYU.py:205): rows from r0, floored
        r0e = int(rng.integers(0, H)); fl = float(rng.choice([0.9, 5.0]))
        r = op.run(xd, want=("depth",), depth_rows_from=r0e, depth_floor=fl); torch.cuda.synchronize()
        ok = st == 0
        if not np.array_equal(r["depth"].cpu().numpy()[ok], O.depth_floor(depth[:, r0e:], fl)[ok], equal_nan=True):
            bad += 1; print("EPILOGUE MISMATCH case", t, (B, H, W), p, r0e, fl)
    if t % 3 == 0:  # every subset of the outputs (k_sky takes its base rows' distances from a scratch map when none is wanted)
        for want in (("depth",), ("index",), ("dt",), ("depth", "index")):
            r = op.run(xd, want=want); torch.cuda.synchronize()
            for k, ref in (("depth", depth), ("dt", dt), ("index", lbl)):
                if k in want:
                    g = r[k].cpu().numpy()
                    if not (np.array_equal(g[st == 0], ref[st == 0], equal_nan=True) if k == "depth" else np.array_equal(g, ref)):
                        bad += 1; print("OPTIONAL OUTPUT MISMATCH case", t, want, k, (B, H, W), p)
    npx += B * H * W
    if t % 25 == 0: print("case", t, "bad", bad, "%.0fs" % (time.time() - t0), flush=True)
print("soak done:", n, "cases,", npx, "pixels, mismatches:", bad)
sys.exit(1 if bad else 0)
