"""the drivers' post-fill steps (crop, depth floor, PNG quantisation) and evaluation.py's metrics.
CPU part: the oracle restatements against hand-worked numbers and the accumulator classes; GPU part: the HIP
kernels (through the C ABI) against the oracle.  Bars: crop / floor / uint16 bit-exact; metrics within 1e-5
relative (the device accumulates the float32 terms in float64, numpy in float32 pairwise -- the terms themselves
are the same float32 values), counts exact."""
import warnings

import numpy as np
import pytest

METRIC_KEYS = ("mse", "rmse", "mae", "irmse", "imae", "delta1", "delta2", "delta3", "count")
RTOL = 1e-5


def depth_pair(rng, shape, scale, miss=0.6):
    gt = (rng.random(shape) * scale + 0.5).astype(np.float32)
    pred = (gt * (1.0 + 0.1 * rng.standard_normal(shape))).astype(np.float32)
    gt[rng.random(shape) < miss] = 0.0           # sparse ground truth
    pred[rng.random(shape) < 0.02] = 0.005       # a few predictions under the 0.01 gate
    return pred, gt


# ---------------------------------------------------------------- CPU: oracle + host logic

def test_oracle_metrics_hand_worked(oracle):
    out = np.float32([[2.0, 4.0, 0.0, 5.0]])
    tgt = np.float32([[1.0, 4.0, 3.0, 0.0]])  # valid: the first two elements
    k = oracle.evaluate_kitti(out, tgt)
    assert k["count"] == 2 and k["mse"] == pytest.approx(1000.0 ** 2 / 2) and k["mae"] == pytest.approx(500.0)
    assert k["rmse"] == pytest.approx(np.sqrt(5e5)) and k["imae"] == pytest.approx(250.0)  # |1/0.002 - 1/0.001| / 2
    assert k["irmse"] == pytest.approx(np.sqrt(500.0 ** 2 / 2)) and k["delta1"] == 0.0
    n = oracle.evaluate_nyu(out, tgt)
    assert n["mse"] == pytest.approx(0.5) and n["mae"] == pytest.approx(0.5)  # relative: (1/1 + 0/4) / 2
    assert n["imae"] == pytest.approx(0.25) and n["delta1"] == 0.5 and n["delta2"] == 0.5 and n["delta3"] == 0.5
    out2 = np.float32([[1.9, 4.0]])
    assert oracle.evaluate_nyu(out2, tgt[:, :2])["delta3"] == 1.0  # 1.9 < 1.953125


def test_oracle_post_steps(oracle):
    d = np.float32([[0.2, 0.9, 0.95, 50.0, 200.0]])
    f = oracle.depth_floor(d)
    assert f.dtype == np.float32 and f[0, 0] == np.float32(0.9) and f[0, 3] == (np.float32(50.0) - np.float32(0.9)) + np.float32(0.9)
    img = oracle.depth_to_png16(np.vstack([d, d * 0.5]), pad_top=3)
    assert img.shape == (5, 5) and img.dtype == np.uint16
    assert (img[:4] == img[0]).all() and img[0, 4] == 25600 and img[0, 0] == int(np.float32(0.9) * np.float32(256.0))
    b = np.arange(2 * 100 * 4, dtype=np.float32).reshape(2, 100, 4, 1)
    assert oracle.kitti_rows(b).shape == (2, 4, 4, 1) and oracle.kitti_rows(b)[0, 0, 0, 0] == 96 * 4
    assert oracle.nyu_eval_crop(np.zeros((1, 240, 320, 1), np.float32)).shape == (228, 304)


def test_result_accumulators(pkg):
    for cls in (pkg.Result, pkg.Result_NYU):
        r = cls()
        r.update(*[2.0] * 14)
        r.update(*[4.0] * 14, photometric=1.0)
        r.finalize()
        assert r.count == 2.0 and r.rmse == 3.0 and r.delta3 == 3.0 and r.photometric == 0.5
        r.set_to_worst()
        assert r.rmse == np.inf and r.delta1 == 0


# ---------------------------------------------------------------- GPU: parity through the C ABI

def metrics_rows(pkg, pred, gt, kind):
    import torch

    dev = "cuda:0"
    rows = pkg.device.metrics_device(torch.from_numpy(pred).to(dev), torch.from_numpy(gt).to(dev), kind).cpu().numpy()
    return [dict(zip(METRIC_KEYS, r.tolist())) for r in rows]


def assert_metrics_close(got, want):
    assert got["count"] == want["count"]
    for k in METRIC_KEYS[:-1]:
        if np.isnan(want[k]):
            assert np.isnan(got[k]), k
        else:
            assert got[k] == pytest.approx(want[k], rel=RTOL, abs=0.0), k


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["kitti", "nyu"])
def test_metrics_vs_oracle(pkg, oracle, kind, gpu_op):
    rng = np.random.default_rng(5)
    ref = oracle.evaluate_kitti if kind == "kitti" else oracle.evaluate_nyu
    for shape, scale in (((3, 352, 1216), 80.0), ((4, 228, 304), 10.0), ((2, 1, 7), 5.0), ((1, 1, 1), 3.0), ((5, 37, 1001), 60.0)):
        pred, gt = depth_pair(rng, shape, scale)
        got = metrics_rows(pkg, pred, gt, kind)
        for b in range(shape[0]):
            assert_metrics_close(got[b], ref(pred[b], gt[b]))
    # no valid element at all: numpy's mean of an empty array is NaN
    pred, gt = depth_pair(rng, (2, 16, 16), 5.0)
    gt[1] = 0.0
    got = metrics_rows(pkg, pred, gt, kind)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert_metrics_close(got[1], ref(pred[1], gt[1]))
    assert got[1]["count"] == 0 and np.isnan(got[1]["rmse"])
    # bit-reproducible run to run (fixed summation order)
    pred, gt = depth_pair(rng, (2, 352, 1216), 80.0)
    assert metrics_rows(pkg, pred, gt, kind) == metrics_rows(pkg, pred, gt, kind)


@pytest.mark.gpu
def test_result_classes_vs_oracle(pkg, oracle, gpu_op):
    rng = np.random.default_rng(6)
    pred, gt = depth_pair(rng, (352, 1216), 80.0)
    r = pkg.Result()
    r.evaluate(pred, gt, photometric=2)
    want = oracle.evaluate_kitti(pred, gt)
    for k in ("mse", "rmse", "mae", "irmse", "imae"):
        assert getattr(r, k) == pytest.approx(want[k], rel=RTOL), k
    assert r.delta1 == 0 and r.photometric == 2.0
    pred, gt = depth_pair(rng, (1, 228, 304, 1), 10.0)
    rn = pkg.Result_NYU()
    rn.evaluate(np.squeeze(pred), np.squeeze(gt))
    want = oracle.evaluate_nyu(pred, gt)
    for k in METRIC_KEYS[:-1]:
        assert getattr(rn, k) == pytest.approx(want[k], rel=RTOL), k
    with pytest.raises(IndexError):
        rn.evaluate(pred[:, :10], gt)


@pytest.mark.gpu
def test_crop_floor_png16_vs_oracle(pkg, oracle, gpu_op):
    import torch

    rng = np.random.default_rng(7)
    x = (rng.random((3, 352, 1216)) * 90.0).astype(np.float32)
    x[rng.random(x.shape) < 0.3] *= 0.01  # plenty of values under the 0.9 floor
    x[0, 0, :5] = [0.0, 0.9, 0.90000004, 100.0, 120.0]
    xd = torch.from_numpy(x).to("cuda:0")
    dev = pkg.device
    assert np.array_equal(dev.crop_floor_device(xd, rows=(96, 352)).cpu().numpy(), x[:, 96:])
    assert np.array_equal(dev.crop_floor_device(xd, floor=0.9).cpu().numpy(), oracle.depth_floor(x))
    assert np.array_equal(dev.crop_floor_device(xd, rows=(6, 234), cols=(8, 312), floor=0.9).cpu().numpy(),
                          oracle.depth_floor(x[:, 6:234, 8:312]))
    assert np.array_equal(dev.crop_floor_device(xd, rows=(351, 352), cols=(1215, 1216)).cpu().numpy(), x[:, 351:, 1215:])
    img = dev.png16_device(xd).cpu().numpy()
    assert img.dtype == np.uint16 and img.shape == (3, 448, 1216)
    for b in range(3):
        assert np.array_equal(img[b], oracle.depth_to_png16(x[b]))
    # no floor, no padding: floor 0 in the oracle is max(d, 0), which the clip to [0, 100] does anyway
    assert np.array_equal(dev.png16_device(xd[:1, :5, :7].contiguous(), pad_top=0, floor=None).cpu().numpy()[0],
                          oracle.depth_to_png16(x[0, :5, :7], pad_top=0, floor=0.0))
    # the numpy-facing mirrors
    assert np.array_equal(pkg.kitti_rows(x[..., None]), oracle.kitti_rows(x[..., None]))
    nyu = x[:1, :240, :320, None]
    assert np.array_equal(pkg.nyu_eval_crop(nyu), oracle.nyu_eval_crop(nyu))
    assert np.array_equal(pkg.depth_floor(x[:1, :, :, None]), np.squeeze(oracle.depth_floor(x[:1])))
    assert np.array_equal(pkg.depth_to_png16(x[1][None, :, :, None]), oracle.depth_to_png16(x[1]))
    with pytest.raises(ValueError):
        dev.crop_floor_device(xd, rows=(10, 10))
    with pytest.raises(pkg.DtfillError):
        pkg._lib.check(pkg._lib.load().dtfill_metrics(None, None, 1, 1, 0, None, None, 0, None))
