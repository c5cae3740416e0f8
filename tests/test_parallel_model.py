"""CPU check of the mathematics behind dtfill.hip: the scan + local-rule formulation
(tests/parallel_model.py) must reproduce the sequential two-pass chamfer restated in oracle/
bit for bit -- distances AND labels -- on random, structured and full-size frames."""
import importlib

import numpy as np
import pytest

import parallel_model as PM


def _check(O, x, src_thr=0.1):
    dt0, l0 = O.nearest_point(x, src_thr)
    dt1, l1 = PM.nearest_point(x, src_thr)
    assert np.array_equal(dt0, dt1)
    assert np.array_equal(l0, l1)


def test_small_random(oracle):
    rng = np.random.default_rng(123)
    for t in range(600):
        H = int(rng.integers(2, 24))
        W = int(rng.integers(2, 30))
        p = rng.choice([0.01, 0.03, 0.1, 0.3, 0.6])
        x = np.where(rng.random((H, W)) < p, 5.0, 0.0).astype(np.float32)
        if t % 7 == 0:
            x[: H // 2] = 0
        if t % 11 == 0:
            x[:, W // 2 :] = 0
        _check(oracle, x)


def test_edge_frames(oracle):
    z = np.zeros((9, 13), np.float32)
    _check(oracle, z)  # no source at all
    one = z.copy()
    for pos in [(0, 0), (0, 12), (8, 0), (8, 12), (4, 6)]:
        one[:] = 0
        one[pos] = 3.0
        _check(oracle, one)
    _check(oracle, np.full((9, 13), 2.0, np.float32))  # all sources


@pytest.mark.parametrize("name", ["kitti_b1", "kitti_b32_scanline", "nyu_b64"])
def test_full_size(oracle, pkg, name):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.make(name, B=1)[0]
    _check(oracle, x)
