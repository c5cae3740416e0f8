"""Synthetic inputs of the shapes and sparsity patterns the reference's loaders produce.

There is no dataset in the container, so bench.py and the tests draw frames from these generators
(definitions fixed in SURVEY.md section 8d):
  kitti_iid       352x1216, Bernoulli(p) valid mask, depths on KITTI's k/256 grid in [1,80] m
                  (data_read.py:81-99: uint16 PNG / 256)
  kitti_scanline  rows 0..99 empty, valid pixels only on every 4th row (Velodyne-like rings)
  nyu_pattern     the reference's own sampling: Mask[randint(H-12,n)+6, randint(W-16,n)+8] = 1
                  (data_read.py:360-364), n = 200 by default (eval_NYU.py:40)
  iid             generic HxW Bernoulli mask, depths U(lo,hi)
All return float32 [B,H,W] with zeros at invalid pixels.
"""
import numpy as np

KITTI_HW = (352, 1216)
NYU_HW = (480, 640)


def _kitti_depths(rng, shape):
    return (np.round(rng.uniform(1.0, 80.0, size=shape) * 256.0) / 256.0).astype(np.float32)


def kitti_iid(B, p=0.05, seed=0, hw=KITTI_HW):
    rng = np.random.default_rng(seed)
    H, W = hw
    mask = rng.random((B, H, W)) < p
    return np.where(mask, _kitti_depths(rng, (B, H, W)), np.float32(0)).astype(np.float32)


def kitti_scanline(B, seed=0, hw=KITTI_HW, empty_rows=100, row_step=4, p=0.25):
    rng = np.random.default_rng(seed)
    H, W = hw
    mask = rng.random((B, H, W)) < p
    rows = np.zeros(H, bool)
    rows[empty_rows::row_step] = True
    mask &= rows[None, :, None]
    return np.where(mask, _kitti_depths(rng, (B, H, W)), np.float32(0)).astype(np.float32)


def nyu_pattern(B, n=200, seed=0, hw=NYU_HW, lo=1.0, hi=10.0):
    rng = np.random.default_rng(seed)
    H, W = hw
    x = np.zeros((B, H, W), np.float32)
    for b in range(B):
        r = rng.integers(0, H - 12, size=n) + 6
        c = rng.integers(0, W - 16, size=n) + 8
        x[b, r, c] = rng.uniform(lo, hi, size=n).astype(np.float32)
    return x


def iid(B, H, W, p, seed=0, lo=1.0, hi=80.0):
    rng = np.random.default_rng(seed)
    mask = rng.random((B, H, W)) < p
    vals = rng.uniform(lo, hi, size=(B, H, W)).astype(np.float32)
    return np.where(mask, vals, np.float32(0)).astype(np.float32)


# BASELINE.json configs (index = position in BASELINE.json "configs")
CONFIGS = {
    "kitti_b1": dict(gen="kitti_iid", B=1, H=352, W=1216, kwargs=dict(p=0.05, seed=0)),
    "kitti_b32": dict(gen="kitti_iid", B=32, H=352, W=1216, kwargs=dict(p=0.05, seed=0)),
    "kitti_b32_scanline": dict(gen="kitti_scanline", B=32, H=352, W=1216, kwargs=dict(seed=0)),
    "nyu_b64": dict(gen="nyu_pattern", B=64, H=480, W=640, kwargs=dict(n=200, seed=0)),
    "synth2048_b16": dict(gen="iid", B=16, H=2048, W=2048, kwargs=dict(p=0.01, seed=2)),
    # the reference's other real shapes: the KITTI frame as eval_NYU.py:157 feeds it (rows 96: of 352 x 1216 -> 256 x 1216) and
    # NYU after the loader's resize to 240 x 320 (data_read.py:350-352) with the sampling pattern of data_read.py:360-364
    "kitti_crop256": dict(gen="kitti_scanline", B=4, H=256, W=1216, kwargs=dict(seed=3, hw=(256, 1216), empty_rows=4)),
    "nyu_240x320": dict(gen="nyu_pattern", B=4, H=240, W=320, kwargs=dict(n=200, seed=4, hw=(240, 320))),
}


def make(name, B=None, seed=None):
    cfg = CONFIGS[name]
    kw = dict(cfg["kwargs"])
    if seed is not None:
        kw["seed"] = seed
    B = cfg["B"] if B is None else B
    if cfg["gen"] == "kitti_iid":
        return kitti_iid(B, **kw)
    if cfg["gen"] == "kitti_scanline":
        return kitti_scanline(B, **kw)
    if cfg["gen"] == "nyu_pattern":
        return nyu_pattern(B, **kw)
    return iid(B, cfg["H"], cfg["W"], **kw)
