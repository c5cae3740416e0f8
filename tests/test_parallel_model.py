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


def test_level_form_matches_oracle(oracle):
    """The level-synchronous form (what k_fused runs) on whole frames, incl. tie-heavy lattices."""
    rng = np.random.default_rng(77)
    frames = []
    for _ in range(150):
        H, W = int(rng.integers(1, 28)), int(rng.integers(1, 34))
        frames.append(np.where(rng.random((H, W)) < rng.choice([0.02, 0.1, 0.4]), 5.0, 0.0).astype(np.float32))
    lat = np.zeros((25, 31), np.float32)
    lat[::6, ::5] = 2.0
    frames.append(lat)
    for x in frames:
        dt0, l0 = oracle.cv_distance_transform_with_labels((x < 0.9).astype(np.uint8))
        dt1, l1 = PM.nearest_point_levels(x)
        assert np.array_equal(dt0, dt1) and np.array_equal(l0, l1)


def test_hypothesis_arbitrary_masks(oracle):
    """Property test: for ARBITRARY source masks (not just i.i.d. ones) both parallel formulations equal
    the sequential two-pass chamfer, distances and labels."""
    hyp = pytest.importorskip("hypothesis")
    from hypothesis import given, settings, strategies as st
    from hypothesis.extra import numpy as hnp

    @settings(max_examples=150, deadline=None, derandomize=True)
    @given(hnp.arrays(np.bool_, st.tuples(st.integers(1, 14), st.integers(1, 18))))
    def check(src):
        x = np.where(src, np.float32(3.0), np.float32(0.0))
        dt0, l0 = oracle.cv_distance_transform_with_labels((~src).astype(np.uint8))
        dt1, l1 = PM.nearest_point(x)
        dt2, l2 = PM.nearest_point_levels(x)
        assert np.array_equal(dt0, dt1) and np.array_equal(l0, l1)
        assert np.array_equal(dt0, dt2) and np.array_equal(l0, l2)

    check()


def test_l2_window_form_matches_the_exact_transform(oracle):
    """k_l2win's formulation (tests/parallel_model.py::l2_window): wherever a source lies within R, the packed-key minimum
    over the window rows gives the exact squared distance and the canonical nearest source."""
    rng = np.random.default_rng(5)
    for t in range(30):
        H, W = int(rng.integers(2, 60)), int(rng.integers(2, 90))
        p = float(rng.choice([0.01, 0.05, 0.2, 0.6]))
        src = rng.random((H, W)) < p
        if t % 5 == 0:
            src[:] = False
            src[::3, ::3] = True  # lattice: ties everywhere
        if not src.any():
            continue
        d2o, nearo = oracle.edt_l2((~src).astype(np.uint8))  # the mask is cv2-style: 0 = source
        for R in (7, 15):
            d2, near, decided = PM.l2_window(src, R)
            assert np.array_equal(decided, d2o <= R * R)
            assert np.array_equal(d2[decided], d2o[decided]) and np.array_equal(near[decided], nearo[decided])


def test_l2_envelope_search_matches_the_exact_transform(oracle):
    """k_l2env's formulation (tests/parallel_model.py::l2_envelope): the level-by-level owner search over the lower envelope
    gives the exact squared distance and the canonical nearest source, at ~log2(W) column evaluations per pixel."""
    rng = np.random.default_rng(6)
    for t in range(24):
        H, W = int(rng.integers(1, 40)), int(rng.integers(1, 70))
        p = float(rng.choice([0.004, 0.02, 0.1, 0.5]))
        src = rng.random((H, W)) < p
        if t % 6 == 0:
            src[:] = False
            src[::4, ::5] = True  # lattice: ties everywhere
        if t % 6 == 1:
            src[:] = False
            src[rng.integers(0, H), rng.integers(0, W)] = True  # a single source
        if not src.any():
            continue
        d2o, nearo = oracle.edt_l2((~src).astype(np.uint8))
        d2, near, evals = PM.l2_envelope(src)
        assert np.array_equal(d2, d2o) and np.array_equal(near, nearo)
        assert evals <= 2 * np.log2(max(W, 2)) + 4


def test_sky_rows_follow_from_the_two_rows_beneath(oracle):
    """k_sky's formulation: the rows above the first source row, recomputed from rows r0 and r0 + 1 of the oracle's own result,
    equal the oracle's.  Ring rows under a sky of several heights, stray first rows, a single source row at the very
    bottom (no row r0 + 1), widths 1 and 2, diagonal partners right under the sky (ties along lines)."""
    import parallel_model as P

    rng = np.random.default_rng(7)
    n = 0
    for trial in range(300):
        H, W = int(rng.integers(2, 70)), int(rng.integers(1, 90))
        top = int(rng.integers(1, H))
        x = np.where(rng.random((H, W)) < rng.choice([0.02, 0.1, 0.3, 0.7]), rng.uniform(1, 80, (H, W)), 0).astype(np.float32)
        x[:top] = 0
        if trial % 3 == 0:
            keep = np.zeros(H, bool)
            keep[top::int(rng.integers(2, 6))] = True
            x[~keep] = 0
        if trial % 5 == 0 and top + 2 < H and W > 6:
            x[top] = 0
            x[top, 2], x[top + 2, 4] = 3.0, 4.0
        if not (x >= 0.9).any():
            x[H - 1, W // 2] = 5.0
        dt, lbl = oracle.nearest_point(x) if min(H, W) > 1 else oracle.fill_batch(x[None])[1:3]
        dt, lbl = np.asarray(dt).reshape(H, W), np.asarray(lbl).reshape(H, W)
        junk_dt, junk_lbl = dt.copy(), lbl.copy()
        r0 = int(np.flatnonzero((dt == 0).any(axis=1))[0])
        junk_dt[:r0], junk_lbl[:r0] = -1, -7  # the model must not look at the sky rows it is given
        d2, l2 = P.sky_rows(junk_dt, junk_lbl)
        assert np.array_equal(d2, dt) and np.array_equal(l2, lbl), (trial, H, W, r0)
        d3, l3 = P.sky_rows_closed_form(junk_dt, junk_lbl)  # what the kernel evaluates: a hop count per pixel, no row loop
        assert np.array_equal(d3, dt) and np.array_equal(l3, lbl), (trial, H, W, r0)
        n += r0 > 0
    assert n > 200
