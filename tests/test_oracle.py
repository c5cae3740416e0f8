"""The CPU oracle against its committed golden vectors and against independent checkers.

PARITY UNPINNED (see oracle/dtfill_oracle.c): there is no cv2 here and the reference holds no
recorded outputs of this path, so the independent anchors are scipy's taxicab transform (equal
distances), the brute-force nearest-source search (every label is a true nearest source), and an
opportunistic comparison with a real cv2 on whatever machine has one."""
import importlib
import os
import subprocess

import numpy as np
import pytest

from helpers import digest, load_cases

CASES, DIGESTS = load_cases()


@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_cases(oracle, name):
    c = CASES[name]
    st, vt = float(c["thr"][0]), float(c["thr"][1])
    depth, dt, lbl, status = oracle.fill_batch(c["x"][None], st, vt)
    assert np.array_equal(dt[0], c["dt"])
    assert np.array_equal(lbl[0], c["lbl"])
    assert np.array_equal(status, c["status"])
    if status[0] == 0:
        assert np.array_equal(depth[0], c["depth"], equal_nan=True)


def test_hand_case_values(oracle):
    """the 5x7 worked example, written out."""
    c = CASES["hand5x7"]
    want_dt = np.array([[3, 2, 3, 2, 1, 0, 1], [2, 1, 2, 3, 2, 1, 2], [1, 0, 1, 2, 2, 2, 3],
                        [2, 1, 2, 2, 1, 2, 3], [3, 2, 2, 1, 0, 1, 2]], np.float32)
    want_lbl = np.array([[2, 2, 2, 1, 1, 1, 1], [2, 2, 2, 1, 1, 1, 1], [2, 2, 2, 2, 3, 1, 1],
                         [2, 2, 2, 3, 3, 3, 3], [2, 2, 3, 3, 3, 3, 3]], np.int32)
    dt, lbl = oracle.nearest_point(c["x"])
    assert np.array_equal(dt, want_dt) and np.array_equal(lbl, want_lbl)
    assert np.array_equal(oracle.Distance_Transform(c["x"], 0.1), np.choose(want_lbl - 1, [10.0, 20.0, 30.0]))


@pytest.mark.parametrize("cfg", sorted(DIGESTS))
def test_full_size_digests(oracle, pkg, cfg):
    synth = importlib.import_module(pkg.__name__ + ".synth")
    d = DIGESTS[cfg]
    x = synth.make(cfg, B=d["B"])
    assert digest(x) == d["x"], "synthetic generator changed"
    depth, dt, lbl, status = oracle.fill_batch(x)
    assert digest(dt) == d["dt"] and digest(lbl) == d["lbl"] and digest(depth) == d["depth"]
    assert status.tolist() == d["status"]


def test_against_scipy_and_bruteforce(oracle):
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(7)
    for _ in range(60):
        H, W = int(rng.integers(2, 40)), int(rng.integers(2, 40))
        src = rng.random((H, W)) < rng.choice([0.02, 0.1, 0.4])
        if not src.any():
            src[H // 2, W // 2] = True
        mask = (~src).astype(np.uint8)
        dist, lab = oracle.cv_distance_transform_with_labels(mask)
        assert np.array_equal(dist, ndimage.distance_transform_cdt(mask, metric="taxicab").astype(np.float32))
        bd, _ = oracle.brute_nearest(mask, 1)
        assert np.array_equal(dist.astype(np.int32), bd)
        pos = np.argwhere(src)
        ii, jj = np.indices((H, W))
        assert np.array_equal(np.abs(ii - pos[lab - 1, 0]) + np.abs(jj - pos[lab - 1, 1]), bd)


def test_glue_semantics(oracle):
    """The literal numpy glue (tools.py:13-35 / eval_NYU.py:120-133) and the all-C glue agree,
    including numpy's negative-index wrap and IndexError."""
    # label 0 with a non-empty value list -> last value
    c = CASES["nosource_values"]
    out = oracle.DT_complete_batch(c["x"][None, :, :, None])
    assert (out == np.float32(0.7)).all() and out.shape == (1, 6, 9, 1) and out.dtype == np.float32
    # label 0 with an empty value list -> IndexError
    with pytest.raises(IndexError):
        oracle.DT_complete_batch(CASES["nosource_novalue"]["x"][None, :, :, None])
    assert CASES["nosource_novalue"]["status"][0] == 1
    # misaligned enumeration really shifts the labels' depths
    c = CASES["misaligned"]
    src_depth = c["x"][np.unravel_index(np.flatnonzero(c["x"].ravel() >= 0.9)[c["lbl"] - 1], c["x"].shape)]
    assert (src_depth != c["depth"]).mean() > 0.5
    # eval_NYU.py:125 squeezes a one-element value list to 0-d and then fails to index it
    with pytest.raises(IndexError):
        oracle.Distance_Transform(CASES["single0"]["x"], 0.1)
    assert oracle.DT_complete_batch(CASES["single0"]["x"][None, :, :, None]).min() == 7.5


def test_l2_oracle_against_bruteforce(oracle):
    rng = np.random.default_rng(11)
    for _ in range(40):
        H, W = int(rng.integers(1, 30)), int(rng.integers(1, 30))
        mask = (rng.random((H, W)) > rng.choice([0.03, 0.2])).astype(np.uint8)
        d0, n0 = oracle.brute_nearest(mask, 2)
        d1, n1 = oracle.edt_l2(mask)
        assert np.array_equal(d0, d1) and np.array_equal(n0, n1)


def test_l2_fill_against_scipy_edt(oracle):
    """The l2 fill oracle against scipy's exact Euclidean transform: equal distances; the chosen
    source is at that distance (indices are compared modulo ties -- the canonical tie-break itself
    is pinned by brute force in test_l2_oracle_against_bruteforce)."""
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(5)
    x = np.where(rng.random((3, 50, 70)) < 0.04, rng.uniform(1, 80, (3, 50, 70)), 0).astype(np.float32)
    depth, dt, idx, status = oracle.fill_batch(x, metric="l2")
    ii, jj = np.indices(x.shape[1:])
    for b in range(3):
        src = x[b] >= 0.9
        e = ndimage.distance_transform_edt(~src)
        assert np.allclose(dt[b], e, rtol=1e-6)
        pos = np.argwhere(src)
        d2 = (ii - pos[idx[b] - 1, 0]) ** 2 + (jj - pos[idx[b] - 1, 1]) ** 2
        assert np.array_equal(d2, np.round(e ** 2).astype(np.int64))
        assert np.array_equal(depth[b], x[b][pos[idx[b] - 1, 0], pos[idx[b] - 1, 1]])
    empty = np.zeros((1, 4, 5), np.float32)
    depth, dt, idx, status = oracle.fill_batch(empty, metric="l2")
    assert np.isinf(dt).all() and (idx == 0).all() and status[0] == 1


def test_outlier_removal_oracle(oracle):
    """The restated filter against an independent formulation (scipy.ndimage.correlate, mirror border =
    BORDER_REFLECT_101) -- float64 there, so only pixels whose decision is not within rounding of the
    threshold are compared -- and against real cv2 when a machine has it."""
    ndimage = pytest.importorskip("scipy.ndimage")
    rng = np.random.default_rng(2)
    x = np.where(rng.random((60, 90)) < 0.3, rng.uniform(1, 80, (60, 90)), 0).astype(np.float32)
    x[10, 10] = 79.0
    k = np.array([[abs(i - 3) + abs(j - 3) <= 3 for j in range(7)] for i in range(7)], np.float64)
    s = ndimage.correlate(x.astype(np.float64), k, mode="mirror")
    c = ndimage.correlate((x > 0.1).astype(np.float64), k, mode="mirror")
    margin = x - s / (c + 0.00001) - 1.0
    want = np.where(margin > 0, 0, x)
    got = oracle.outlier_removal(x)
    sure = np.abs(margin) > 1e-3
    assert np.array_equal(got[sure], want[sure].astype(np.float32)) and (got[10, 10] == 0)
    try:
        import cv2
    except ImportError:
        return
    ref = x * (1 - ((x - cv2.filter2D(x, -1, k.astype(np.uint8)) / (cv2.filter2D((x > 0.1).astype(np.float64), -1, k.astype(np.uint8)) + 0.00001)) > 1.0))
    assert np.array_equal(got, ref.astype(np.float32))


def test_real_cv2_if_present(oracle):
    """Opportunistic pin: on a machine that has OpenCV, the restatement must equal it."""
    cv2 = pytest.importorskip("cv2")
    rng = np.random.default_rng(3)
    for _ in range(20):
        mask = (rng.random((37, 53)) > 0.07).astype(np.uint8)
        dt, lbl = cv2.distanceTransformWithLabels(mask, cv2.DIST_L1, 5, labelType=cv2.DIST_LABEL_PIXEL)
        d0, l0 = oracle.cv_distance_transform_with_labels(mask)
        assert np.array_equal(dt, d0) and np.array_equal(lbl, l0)


def test_oracle_sanitized():
    """the CPU restatement under AddressSanitizer + UBSan (host build only -- the GPU pool has no sanitizer
    for device code).  oracle/asan_main.c drives every entry point over awkward shapes; any report fails the run."""
    import shutil
    import subprocess

    here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
    if shutil.which("gcc") is None and shutil.which("cc") is None:
        pytest.skip("no C compiler")
    subprocess.check_call(["make", "-s", "-C", here, "oracle_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    env.pop("LD_PRELOAD", None)
    r = subprocess.run([os.path.join(here, "oracle_asan")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "oracle_asan ok" in r.stdout and "runtime error" not in r.stderr and "ERROR" not in r.stderr, r.stderr[-2000:]


def test_l2_oracle_against_the_scipy_pinned_fixtures(oracle, pkg):
    """tests/golden/l2_cases.npz was written from scipy.ndimage.distance_transform_edt (squared distances exact; canonical
    tie by brute force): the l2 oracle must reproduce it, and the full-size digests of the two KITTI shapes."""
    import importlib

    from helpers import digest, labels_from_nearest, load_l2_cases

    cases, digests = load_l2_cases()
    for name, c in cases.items():
        src = ~((np.float32(1.0) - c["x"]) > np.float32(0.1))
        d2, near = oracle.edt_l2((~src).astype(np.uint8))
        if src.any():
            assert np.array_equal(d2, c["d2"]) and np.array_equal(near, c["near"]), name
        depth, dt, idx, status = oracle.fill_batch(c["x"][None], metric="l2")
        assert np.array_equal(idx[0], labels_from_nearest(c["x"], c["near"])), name
    synth = importlib.import_module(pkg.__name__ + ".synth")
    for cfg in ("kitti_b32_scanline", "nyu_240x320"):
        d = digests[cfg]
        x = synth.make(cfg, B=d["B"])
        assert digest(x) == d["x"]
        depth, dt, idx, status = oracle.fill_batch(x, metric="l2")
        assert digest(idx) == d["lbl"] and digest(depth) == d["depth"]
        assert digest(np.round(dt.astype(np.float64) ** 2).astype(np.int32)) == d["d2"]
