"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors.  Bar: bit-exact distance map, label map and gathered depth (integer / index work;
the float outputs are exact copies or integer-valued, so the float tolerance is 0)."""
import importlib
import os

import numpy as np
import pytest

from helpers import digest, labels_from_nearest, load_cases, load_l2_cases

pytestmark = pytest.mark.gpu
CASES, DIGESTS = load_cases()
L2_CASES, L2_DIGESTS = load_l2_cases()


PATHS = ("auto", "general")  # every comparison runs through the fused+fallback pass AND the general kernels alone


def run(op, x, st=0.1, vt=0.1, want=("depth", "dt", "index"), path="auto"):
    """One pass through the C ABI.  Returns numpy outputs; "status" keeps only the IndexError bit,
    "general" says which frames took the any-distance kernels."""
    import torch

    xd = torch.from_numpy(np.ascontiguousarray(x, np.float32)).to("cuda:0")
    res = op.run(xd, st, vt, want, path=path)
    torch.cuda.synchronize()
    out = {k: v.cpu().numpy() for k, v in res.items()}
    out["general"] = (out["status"] & 2) != 0
    out["status"] = out["status"] & 1
    return out


def assert_equal_to_oracle(oracle, op, x, st=0.1, vt=0.1, paths=PATHS):
    depth, dt, lbl, status = oracle.fill_batch(x, st, vt)
    for path in paths:
        got = run(op, x, st, vt, path=path)
        assert np.array_equal(got["dt"], dt), "%s: distance map differs" % path
        assert np.array_equal(got["index"], lbl), "%s: label map differs: %d px" % (path, (got["index"] != lbl).sum())
        assert np.array_equal(got["status"], status), path
        ok = status == 0
        assert np.array_equal(got["depth"][ok], depth[ok], equal_nan=True), "%s: filled depth differs" % path


def test_native_library_loaded(gpu_op, pkg):
    import torch

    assert torch.cuda.get_device_name(0)
    maps = open("/proc/self/maps").read()
    assert "libdtfill.so" in maps, "the HIP extension is not the code that ran"


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name", sorted(CASES))
def test_golden_cases(gpu_op, name, path):
    c = CASES[name]
    got = run(gpu_op, c["x"][None], float(c["thr"][0]), float(c["thr"][1]), path=path)
    assert np.array_equal(got["dt"][0], c["dt"])
    assert np.array_equal(got["index"][0], c["lbl"])
    assert np.array_equal(got["status"], c["status"])
    if c["status"][0] == 0:
        assert np.array_equal(got["depth"][0], c["depth"], equal_nan=True)
    if path == "general":
        assert got["general"].all()


def test_golden_cases_as_one_batch_per_shape(gpu_op):
    """Frames of equal shape stacked into one batch: exercises B > 1 and per-frame status."""
    by_shape = {}
    for name, c in CASES.items():
        if tuple(c["thr"]) == (np.float32(0.1), np.float32(0.1)):
            by_shape.setdefault(c["x"].shape, []).append(c)
    for shape, cs in by_shape.items():
        x = np.stack([c["x"] for c in cs])
        got = run(gpu_op, x)
        for b, c in enumerate(cs):
            assert np.array_equal(got["dt"][b], c["dt"]) and np.array_equal(got["index"][b], c["lbl"])
            assert got["status"][b] == c["status"][0]
            if c["status"][0] == 0:
                assert np.array_equal(got["depth"][b], c["depth"], equal_nan=True)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("cfg", sorted(DIGESTS))
def test_full_size_digests(gpu_op, pkg, cfg, path):
    """BASELINE.json's shapes: outputs hashed against the oracle's committed sha256."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    d = DIGESTS[cfg]
    x = synth.make(cfg, B=d["B"])
    assert digest(x) == d["x"]
    got = run(gpu_op, x, path=path)
    assert digest(got["dt"]) == d["dt"]
    assert digest(got["index"]) == d["lbl"]
    assert digest(got["depth"]) == d["depth"]
    assert got["status"].tolist() == d["status"]


def test_seeded_random_vs_oracle(gpu_op, oracle):
    rng = np.random.default_rng(2024)
    for t in range(40):
        B = int(rng.integers(1, 4))
        H, W = int(rng.integers(1, 90)), int(rng.integers(1, 200))
        p = rng.choice([0.003, 0.02, 0.05, 0.3, 0.8])
        x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
        if t % 5 == 0:
            x[:, : H // 2] = 0
        if t % 7 == 0:
            x[:, :, W // 3 :] = 0
        assert_equal_to_oracle(oracle, gpu_op, x)


def test_thresholds_and_misalignment_vs_oracle(gpu_op, oracle):
    """eval_NYU.py's (0.001, 0.1) threshold pair with depths from 0.7 m: source and value
    enumerations differ, the value list must be materialised."""
    rng = np.random.default_rng(99)
    x = np.where(rng.random((3, 120, 160)) < 0.02, rng.uniform(0.7, 10, (3, 120, 160)), 0).astype(np.float32)
    x[0, 0, :5] = [0.5, 0.95, 0.9991, 0.3, 0.999]
    assert_equal_to_oracle(oracle, gpu_op, x, 0.001, 0.1)
    assert_equal_to_oracle(oracle, gpu_op, x, 0.1, 0.1)
    assert_equal_to_oracle(oracle, gpu_op, x, 0.1, 0.6)  # fewer values than sources -> IndexError frames


def test_sparse_long_chains_vs_oracle(gpu_op, oracle, pkg):
    """Few sources in a big frame: distances in the hundreds, long parent chains."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    for n in (1, 3, 20):
        x = synth.nyu_pattern(2, n=n, seed=n)
        assert_equal_to_oracle(oracle, gpu_op, x)
    x = np.zeros((1, 300, 500), np.float32)
    x[0, 299, 0] = 2.0
    assert_equal_to_oracle(oracle, gpu_op, x)
    x[0, 0, 499] = 3.0
    assert_equal_to_oracle(oracle, gpu_op, x)


def test_batch32_kitti_properties(gpu_op, oracle, pkg):
    """BASELINE config 2 at full size (B=32): size-independent properties on every frame, and a
    full oracle comparison on a few of them."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.make("kitti_b32")
    got = run(gpu_op, x)
    B, H, W = x.shape
    ii, jj = np.indices((H, W))
    for b in range(B):
        src = x[b] >= 0.9
        pos = np.argwhere(src)
        lbl = got["index"][b]
        assert lbl.min() >= 1 and lbl.max() <= len(pos)
        # every label is a true L1-nearest source and dt is that distance
        l1 = np.abs(ii - pos[lbl - 1, 0]) + np.abs(jj - pos[lbl - 1, 1])
        assert np.array_equal(l1.astype(np.float32), got["dt"][b])
        # sources keep their own rank and depth; the fill is the labelled source's depth
        assert np.array_equal(lbl[src], np.arange(1, len(pos) + 1))
        assert np.array_equal(got["depth"][b], x[b][pos[lbl - 1, 0], pos[lbl - 1, 1]])
    # idempotence: a filled frame has every pixel valid, so filling it again changes nothing
    again = run(gpu_op, got["depth"][:2])
    assert np.array_equal(again["depth"], got["depth"][:2]) and (again["dt"] == 0).all()
    for b in (0, 13, 31):
        depth, dt, lbl, status = oracle.fill_batch(x[b : b + 1])
        assert np.array_equal(got["dt"][b], dt[0]) and np.array_equal(got["index"][b], lbl[0])
        assert np.array_equal(got["depth"][b], depth[0])


def test_optional_outputs_and_reuse(gpu_op, oracle):
    rng = np.random.default_rng(5)
    x = np.where(rng.random((2, 50, 70)) < 0.05, rng.uniform(1, 80, (2, 50, 70)), 0).astype(np.float32)
    depth, dt, lbl, _ = oracle.fill_batch(x)
    for want in (("depth",), ("dt",), ("index",), ("dt", "index")):
        got = run(gpu_op, x, want=want)
        if "depth" in want:
            assert np.array_equal(got["depth"], depth)
        if "dt" in want:
            assert np.array_equal(got["dt"], dt)
        if "index" in want:
            assert np.array_equal(got["index"], lbl)
    # frames with a handful of sources: the row kernel stores the finals, the tile kernel only rewrites the tie pixels -- with
    # every subset of outputs, next to a dense frame and one whose value list is misaligned (which goes the long way)
    x = np.zeros((4, 120, 300), np.float32)
    for b, n in enumerate((1, 7, 60)):
        pos = rng.choice(120 * 300, n, replace=False)
        x[b].flat[pos] = rng.uniform(1, 80, n)
    x[1, 40, 40] = x[1, 70, 70] = x[1, 20, 90] = x[1, 50, 60] = 3.0  # diagonal partners: whole regions of tie pixels
    x[2, 5, :9] = 0.5                                                 # values that are not sources
    x[3] = np.where(rng.random((120, 300)) < 0.06, rng.uniform(1, 80, (120, 300)), 0)
    depth, dt, lbl, st = oracle.fill_batch(x)
    assert not st.any()
    for want in (("depth", "dt", "index"), ("depth",), ("dt",), ("index",), ("depth", "index")):
        for path in PATHS:
            got = run(gpu_op, x, want=want, path=path)
            for k, ref in (("depth", depth), ("dt", dt), ("index", lbl)):
                if k in want:
                    assert np.array_equal(got[k], ref), (want, path, k)


def test_reference_named_functions(pkg, oracle):
    """The drop-in layer: same names / shapes / dtypes / exceptions as tools.py and eval_NYU.py."""
    c = CASES["rand_p0.05"]
    dt, lbl = pkg.nearest_point(c["x"])
    assert dt.dtype == np.float32 and lbl.dtype == np.int32
    assert np.array_equal(dt, c["dt"]) and np.array_equal(lbl, c["lbl"])
    batch = np.stack([c["x"], c["x"][::-1].copy()])[..., None]
    out = pkg.DT_complete_batch(batch)
    assert out.shape == batch.shape and out.dtype == np.float32
    assert np.array_equal(out, oracle.DT_complete_batch(batch))
    one = pkg.Distance_Transform(c["x"][None, :, :, None].astype(np.float64))
    assert one.dtype == np.float64 and np.array_equal(one, oracle.Distance_Transform(c["x"].astype(np.float64)))
    with pytest.raises(IndexError):
        pkg.DT_complete_batch(np.zeros((1, 6, 9, 1), np.float32))
    with pytest.raises(IndexError):  # eval_NYU.py:125's 0-d squeeze of a one-element value list
        pkg.Distance_Transform(CASES["single0"]["x"], 0.1)
    with pytest.raises(TypeError):
        pkg.nearest_point(np.full((4, 4), 0.1, np.float64))


def test_fused_kernel_alone_decides_dense_frames(gpu_op, oracle, pkg):
    """The LDS-tile kernel by itself (general kernels skipped): on the 5 %-valid KITTI workload no
    pixel is farther than the halo from a source, so nothing may be flagged and everything must
    already be exact; a frame with an empty band must be flagged, and only that frame."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.make("kitti_b32", B=4)
    depth, dt, lbl, _ = oracle.fill_batch(x)
    got = run(gpu_op, x, path="fused")
    assert not got["general"].any()
    assert np.array_equal(got["dt"], dt) and np.array_equal(got["index"], lbl) and np.array_equal(got["depth"], depth)
    # an empty band of 10 rows: still within the halo
    x[1, 100:110] = 0
    # an empty band of 90 rows: distances up to ~45 -> only the any-distance kernels can decide it
    x[2, 100:190] = 0
    depth2, dt2, lbl2, _ = oracle.fill_batch(x)
    assert dt2[1].max() <= 16 < dt2[2].max()
    got = run(gpu_op, x, path="fused")
    assert got["general"].tolist() == [False, False, True, False]
    for b in (0, 1, 3):
        assert np.array_equal(got["index"][b], lbl2[b]) and np.array_equal(got["depth"][b], depth2[b])
        assert np.array_equal(got["dt"][b], dt2[b])
    auto = run(gpu_op, x)
    assert auto["general"].tolist() == [False, False, True, False]
    assert np.array_equal(auto["dt"], dt2) and np.array_equal(auto["index"], lbl2) and np.array_equal(auto["depth"], depth2)


def test_tile_seams_and_halo_boundary(gpu_op, oracle):
    """Distances exactly at / one past the halos (16 and 32), sources on tile seams, odd frame sizes
    that make ragged last tiles."""
    for H, W in [(88, 152), (89, 153), (176, 304), (100, 321), (33, 40), (17, 500)]:
        x = np.zeros((3, H, W), np.float32)
        x[0, ::17, ::17] = 2.0   # lattice with max L1 distance 16: exactly the halo
        x[1, ::18, ::17] = 3.0   # ... 17: one past it in places
        x[2, H // 2, :] = 4.0    # a full row of sources: vertical distances up to H/2
        x[2, :, W // 2] = 5.0
        assert_equal_to_oracle(oracle, gpu_op, x)
        y = np.zeros((2, H, W), np.float32)
        y[0, ::33, ::33] = 2.0   # max L1 distance 32: exactly the second halo
        y[1, ::34, ::33] = 3.0   # ... 33
        assert_equal_to_oracle(oracle, gpu_op, y)
    rng = np.random.default_rng(8)
    x = np.where(rng.random((2, 352, 1216)) < 0.012, rng.uniform(1, 80, (2, 352, 1216)), 0).astype(np.float32)
    assert_equal_to_oracle(oracle, gpu_op, x)  # 1.2 %: a mix of decided and flagged frames


def test_l2_metric_vs_oracle(pkg, oracle):
    """The `l2` mode (exact Euclidean transform, ties -> smallest raster index of the source):
    index map and squared distances bit-exact (the float distance is sqrtf of an exact integer,
    compared at rtol 1e-6, inside the 1e-5 BASELINE.json's north_star allows), depth gathered
    through the same value-list glue."""
    import torch

    op = pkg.device.DtFill(device="cuda:0", metric="l2")
    rng = np.random.default_rng(77)
    frames = []
    for t in range(12):
        B, H, W = int(rng.integers(1, 3)), int(rng.integers(1, 120)), int(rng.integers(1, 300))
        p = rng.choice([0.002, 0.02, 0.05, 0.4])
        x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
        if t == 3:
            x[:] = 0  # no source at all
        if t == 4:
            x[0, 0, :3] = [0.5, 0.3, 0.85]  # values that are not sources: misaligned enumerations
        frames.append(x)
    lat = np.zeros((1, 60, 90), np.float32)
    lat[0, ::10, ::10] = 3.0  # lattice: every kind of tie
    frames.append(lat)
    for x in frames:
        depth, dt, idx, status = oracle.fill_batch(x, metric="l2")
        res = op.run(torch.from_numpy(x).to("cuda:0"))
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in res.items()}
        assert np.array_equal(got["index"], idx), "l2 index map differs: %d px" % (got["index"] != idx).sum()
        assert np.allclose(got["dt"], dt, rtol=1e-6, atol=0), "l2 distance differs"
        assert np.array_equal(np.isinf(got["dt"]), np.isinf(dt))
        assert np.array_equal(got["status"] & 1, status)
        ok = status == 0
        assert np.array_equal(got["depth"][ok], depth[ok], equal_nan=True)


def _run_l2(pkg, x, path="auto"):
    import torch

    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    res = op2.run(torch.from_numpy(np.ascontiguousarray(x, np.float32)).to("cuda:0"), path=path)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in res.items()}


@pytest.mark.parametrize("path", ("auto", "general"))
def test_l2_scipy_pinned_cases(pkg, path):
    """The l2 mode against fixtures written from scipy.ndimage.distance_transform_edt (tests/golden/make_golden_l2.py; north_star:
    index map bit-exact, distance within 1e-5 relative): squared distances are scipy's, the index is the canonical one
    (smallest raster index among the equidistant sources, pinned by brute force when the fixtures were made).  Both kernel
    families (window + far list; column distances + envelope search / tile search)."""
    for name, c in L2_CASES.items():
        got = _run_l2(pkg, c["x"][None], path)
        if (c["near"] < 0).all():
            assert np.isinf(got["dt"]).all() and (got["index"] == 0).all(), name
            continue
        assert np.array_equal(got["index"][0], labels_from_nearest(c["x"], c["near"])), name
        want = np.sqrt(c["d2"].astype(np.float64))
        assert np.allclose(got["dt"][0], want, rtol=1e-5, atol=0), name  # tolerance of BASELINE.json's north_star
        src = c["x"] >= 0.9
        assert np.array_equal(got["depth"][0], c["x"].ravel()[c["near"].ravel()].reshape(c["x"].shape)) or not src.any(), name


@pytest.mark.parametrize("cfg", sorted(L2_DIGESTS))
def test_l2_full_size_digests_from_scipy(pkg, cfg):
    """BASELINE.json's shapes (and the reference's two other real shapes) in the l2 mode: squared distances and canonical
    labels hashed against the digests the scipy-checked generator committed."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    d = L2_DIGESTS[cfg]
    x = synth.make(cfg, B=d["B"])
    assert digest(x) == d["x"]
    got = _run_l2(pkg, x)
    assert digest(got["index"]) == d["lbl"]
    assert digest(got["depth"]) == d["depth"]
    assert digest(np.round(got["dt"].astype(np.float64) ** 2).astype(np.int32)) == d["d2"]


def test_l2_full_size_properties(pkg, oracle):
    """KITTI-size batch in l2 mode: the index is a true Euclidean-nearest source (checked against
    the exact transform of a few frames) and one full frame equals the oracle."""
    import torch

    synth = importlib.import_module(pkg.__name__ + ".synth")
    op = pkg.device.DtFill(device="cuda:0", metric="l2")
    x = synth.make("kitti_b32", B=4)
    res = op.run(torch.from_numpy(x).to("cuda:0"))
    torch.cuda.synchronize()
    idx = res["index"].cpu().numpy()
    dt = res["dt"].cpu().numpy()
    depth, dt0, idx0, _ = oracle.fill_batch(x[:1], metric="l2")
    assert np.array_equal(idx[0], idx0[0]) and np.allclose(dt[0], dt0[0], rtol=1e-6)
    assert np.array_equal(res["depth"].cpu().numpy()[0], depth[0])
    ii, jj = np.indices(x.shape[1:])
    for b in range(1, 4):
        pos = np.argwhere(x[b] >= 0.9)
        d2 = (ii - pos[idx[b] - 1, 0]) ** 2 + (jj - pos[idx[b] - 1, 1]) ** 2
        assert np.allclose(np.sqrt(d2.astype(np.float32)), dt[b], rtol=1e-6)


def test_rows_handed_on_by_the_window_kernel(gpu_op, oracle):
    """A dense frame keeps the window kernel's results wherever every pixel of a row is near a source; the rows with a pixel
    farther than the halo are redone by the any-distance kernels (k_fused's rowflag).  Frames whose undecided rows are a
    sky on top, a band in the middle, scattered holes; tie chains that start in a redone row and end in a kept one (sources
    on exact diagonals just inside the dense part), that cross tiles, that run longer than k_fin follows them; one batch
    mixing such frames with plain dense and sparse ones; the depth epilogue on top (then the whole frame is redone)."""
    import torch

    rng = np.random.default_rng(77)

    def dense(B, H, W, p):
        return np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)

    for trial in range(6):
        B, H, W = 4, int(rng.choice([200, 352, 417])), int(rng.choice([640, 1216, 1000]))
        x = dense(B, H, W, float(rng.choice([0.04, 0.08, 0.15])))
        top = int(rng.integers(40, 130))
        x[0, :top] = 0                                   # sky
        x[1, top:top + int(rng.integers(35, 90))] = 0    # a band without sources in the middle
        for _ in range(6):                               # holes
            i, j = int(rng.integers(0, H - 60)), int(rng.integers(0, W - 80))
            x[2, i:i + int(rng.integers(34, 60)), j:j + int(rng.integers(34, 80))] = 0
        x[3, :top] = 0
        for _ in range(12):                              # a few sources in the sky, half of them with a diagonal partner
            i, j = int(rng.integers(0, top)), int(rng.integers(0, W))
            x[3, i, j] = 5.0
            k = int(rng.integers(1, 30))
            if rng.random() < 0.5 and i + k < H and j + k < W:
                x[3, i + k, j + k] = 6.0
        # sources on exact diagonals at the edge of the dense part: chains of tie pixels that run from the sky into kept rows
        for j in range(20, W - 60, 97):
            x[0, top:top + 3, j - 20:j + 60] = 0
            x[0, top, j] = 7.0
            x[0, top + 2, j + 2] = 8.0
        assert_equal_to_oracle(oracle, gpu_op, x)
    x = dense(3, 352, 1216, 0.05)
    x[1, :120] = 0
    x[2] = dense(1, 352, 1216, 0.0005)[0]
    assert_equal_to_oracle(oracle, gpu_op, x)
    got = run(gpu_op, x)
    assert got["general"].tolist() == [False, True, True]
    # the depth epilogue: rows from 96 on, floored
    want = oracle.fill_batch(x)[0]
    res = gpu_op.run(torch.from_numpy(x).to("cuda:0"), want=("depth",), depth_rows_from=96, depth_floor=0.9)
    torch.cuda.synchronize()
    assert np.array_equal(res["depth"].cpu().numpy(), oracle.depth_floor(want[:, 96:], 0.9))


def assert_l2_equal_to_oracle(oracle, op2, x, st=0.1, vt=0.1, paths=("auto", "general")):
    """l2 mode through both kernel families (window + far list for dense frames, column distances + row search for
    the others): index bit-exact, distance sqrtf of the exact integer, depth through the value-list glue."""
    import torch

    depth, dt, idx, status = oracle.fill_batch(x, st, vt, metric="l2")
    for path in paths:
        res = op2.run(torch.from_numpy(x).to("cuda:0"), st, vt, path=path)
        torch.cuda.synchronize()
        got = {k: v.cpu().numpy() for k, v in res.items()}
        bad = got["index"] != idx
        assert not bad.any(), "l2 %s: index differs at %d px, first %s" % (path, bad.sum(), np.argwhere(bad)[:3].tolist())
        assert np.allclose(got["dt"], dt, rtol=1e-6, atol=0), "l2 %s: distance differs" % path
        assert np.array_equal(np.isinf(got["dt"]), np.isinf(dt))
        assert np.array_equal(got["status"] & 1, status)
        ok = status == 0
        assert np.array_equal(got["depth"][ok], depth[ok], equal_nan=True), "l2 %s: depth differs" % path


def test_l2_window_kernel_and_far_list(pkg, oracle):
    """The dense-frame l2 kernels: windows of 15 x 15 (route 16) and 31 x 31 (route 32) with the far list behind them.
    Densities either side of both routing thresholds, holes that leave thousands of pixels without a source in their
    window, lattices (every pixel a tie), shapes that are not multiples of the tile, frames of different routes in one
    batch, misaligned value lists."""
    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    rng = np.random.default_rng(2025)
    for (B, H, W, p) in [(2, 97, 300, 0.05), (1, 64, 256, 0.03), (2, 33, 513, 0.3), (1, 130, 1216, 0.012), (2, 75, 700, 0.008),
                         (1, 352, 1216, 0.05), (3, 40, 65, 0.1), (1, 31, 255, 0.9), (1, 200, 257, 0.02)]:
        x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
        assert_l2_equal_to_oracle(oracle, op2, x)
    # holes: a block without sources in the middle of a dense frame, touching / not touching the border
    x = np.where(rng.random((3, 160, 600)) < 0.06, rng.uniform(0.95, 80, (3, 160, 600)), 0).astype(np.float32)
    x[0, 40:120, 200:420] = 0
    x[1, :, 500:] = 0
    x[2, 100:, :90] = 0
    assert_l2_equal_to_oracle(oracle, op2, x)
    # lattices of period 2..9: dense enough for the window kernel, every pixel between sources is a tie
    for per in (2, 3, 4, 6, 9):
        lat = np.zeros((1, 70, 330), np.float32)
        lat[0, ::per, ::per] = 5.0
        lat[0, 3::per * 2, 1::per * 3] = 7.0
        assert_l2_equal_to_oracle(oracle, op2, lat)
    # one batch, three routes: 5 % (window 15), 1.2 % (window 31), 0.05 % (row search); and misaligned enumerations
    x = np.zeros((3, 128, 640), np.float32)
    for b, p in enumerate((0.05, 0.012, 0.0005)):
        x[b] = np.where(rng.random((128, 640)) < p, rng.uniform(0.95, 80, (128, 640)), 0)
    assert_l2_equal_to_oracle(oracle, op2, x)
    x[0, 5, :40] = 0.5  # values that are not sources
    assert_l2_equal_to_oracle(oracle, op2, x)
    assert_l2_equal_to_oracle(oracle, op2, np.where(x > 0, rng.uniform(0.7, 10, x.shape), 0).astype(np.float32), 0.001, 0.1)


def test_l2_sparse_shapes(pkg, oracle):
    """l2 on the sparse configurations: NYU 480 x 640 with 20 and 200 samples, 2048 x 2048 at 1 %."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    rng = np.random.default_rng(99)
    for n in (20, 200):
        x = np.zeros((2, 480, 640), np.float32)
        for b in range(2):
            pos = rng.choice(480 * 640, n, replace=False)
            x[b].flat[pos] = rng.uniform(0.95, 10, n)
        assert_l2_equal_to_oracle(oracle, op2, x)
    x = synth.make("synth2048_b16", B=1)
    assert_l2_equal_to_oracle(oracle, op2, x)


def test_l2_frames_with_a_handful_of_sources(pkg, oracle):
    """l2, at most 512 sources in the frame (the NYU sampling patterns): every 32 x 32 tile takes the minimum over the sources
    within (distance from its centre to the nearest source) + its diagonal.  512 / 513 sources (the routing boundary), all
    sources in one corner (every tile keeps every source), one source, two sources on a diagonal (ties along a line), frames
    smaller than a tile, a width that is no multiple of 32, values that are not sources, dense and sparse frames in one batch."""
    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    rng = np.random.default_rng(314)

    def pts(H, W, n, box=None):
        f = np.zeros((H, W), np.float32)
        r0, r1, c0, c1 = box or (0, H, 0, W)
        pos = rng.choice((r1 - r0) * (c1 - c0), n, replace=False)
        f[r0 + pos // (c1 - c0), c0 + pos % (c1 - c0)] = rng.uniform(0.95, 10, n)
        return f

    for n in (1, 2, 37, 512, 513):
        assert_l2_equal_to_oracle(oracle, op2, np.stack([pts(480, 640, n), pts(480, 640, max(1, n // 2))]))
    assert_l2_equal_to_oracle(oracle, op2, pts(300, 700, 400, box=(0, 40, 0, 60))[None])   # a cluster in one corner
    assert_l2_equal_to_oracle(oracle, op2, pts(200, 333, 50, box=(150, 200, 300, 333))[None])
    x = np.zeros((2, 90, 130), np.float32)
    x[0, 10, 10] = x[0, 50, 50] = 3.0     # every pixel of the anti-diagonal band between them is a tie
    x[1, 0, 129] = x[1, 89, 0] = 4.0
    assert_l2_equal_to_oracle(oracle, op2, x)
    for (H, W) in [(5, 7), (31, 33), (33, 31), (1, 100), (100, 1)]:
        assert_l2_equal_to_oracle(oracle, op2, pts(H, W, min(3, H * W))[None])
    x = np.stack([pts(128, 640, 60), pts(128, 640, 30)])
    x[0, 5, :40] = 0.5                    # values that are not sources: misaligned enumerations
    assert_l2_equal_to_oracle(oracle, op2, x)
    mix = np.zeros((3, 128, 640), np.float32)
    mix[0] = pts(128, 640, 100)
    mix[1] = np.where(rng.random((128, 640)) < 0.05, rng.uniform(0.95, 80, (128, 640)), 0)
    mix[2] = np.where(rng.random((128, 640)) < 0.004, rng.uniform(0.95, 80, (128, 640)), 0)
    assert_l2_equal_to_oracle(oracle, op2, mix)


def test_l1_frames_with_a_handful_of_sources(gpu_op, oracle):
    """l1_cv, at most 512 sources in a frame too thin for a window (the NYU sampling patterns): k_pts -- per tile the sources
    whose cells reach it, pruned by dominance on the box's corners, packed-key minima per pixel, k_fin's rule for the tie
    pixels.  512 / 513 sources (the routing boundary), clusters (more than 64 candidates per wave: the slow lists; more than
    96 in one band of rows: back to the any-distance kernels), one source, diagonal partners (ties along lines, chains that
    cross tiles and go to k_tiesx), regular lattices (many-way ties), frames smaller than a tile, widths that are no multiple
    of 4 / 32, sources on the frame's edges, values that are not sources, every route in one batch, optional outputs."""
    rng = np.random.default_rng(2718)

    def pts(H, W, n, box=None, lo=0.95):
        f = np.zeros((H, W), np.float32)
        r0, r1, c0, c1 = box or (0, H, 0, W)
        pos = rng.choice((r1 - r0) * (c1 - c0), n, replace=False)
        f[r0 + pos // (c1 - c0), c0 + pos % (c1 - c0)] = rng.uniform(lo, 10, n)
        return f

    for n in (1, 2, 37, 200, 512, 513):
        assert_equal_to_oracle(oracle, gpu_op, np.stack([pts(480, 640, n), pts(480, 640, max(1, n // 2))]))
    assert_equal_to_oracle(oracle, gpu_op, pts(300, 700, 90, box=(100, 130, 200, 260))[None])   # 90 sources in one band, close together
    assert_equal_to_oracle(oracle, gpu_op, pts(300, 700, 400, box=(0, 40, 0, 60))[None])       # a crowd in a corner: not k_pts's
    assert_equal_to_oracle(oracle, gpu_op, pts(200, 333, 50, box=(150, 200, 300, 333))[None])
    assert_equal_to_oracle(oracle, gpu_op, pts(480, 640, 300, box=(0, 480, 600, 640))[None])   # a strip on the right edge
    x = np.zeros((4, 90, 130), np.float32)
    x[0, 10, 10] = x[0, 50, 50] = 3.0     # every pixel of the anti-diagonal band between them is a tie
    x[1, 0, 129] = x[1, 89, 0] = 4.0
    x[2, ::15, ::13] = 2.0                # a lattice: three- and four-way ties
    x[3, 0, :] = 0
    x[3, 0, 0] = x[3, 0, 129] = x[3, 89, 0] = x[3, 89, 129] = 5.0   # the four corners
    assert_equal_to_oracle(oracle, gpu_op, x)
    x = np.zeros((2, 352, 1216), np.float32)
    x[0, 100, 100] = x[0, 300, 300] = x[0, 20, 1000] = x[0, 340, 1100] = 3.0   # long diagonal tie lines across many tiles
    x[1, 5, 5] = 1.0
    assert_equal_to_oracle(oracle, gpu_op, x)
    for (H, W) in [(5, 7), (31, 33), (33, 31), (1, 100), (100, 1), (64, 257), (37, 1023)]:
        assert_equal_to_oracle(oracle, gpu_op, pts(H, W, min(3, H * W))[None])
    for (H, W, n) in [(200, 300, 40), (480, 600, 150), (130, 129, 12), (65, 640, 30)]:  # shapes that take the 64 x 128 tiles, ragged either way
        assert_equal_to_oracle(oracle, gpu_op, np.stack([pts(H, W, n), pts(H, W, 2 * n)]))
    # the widest frame whose keys still hold distance and column (H + W <= 8100): sources all over, and one in a corner
    assert_equal_to_oracle(oracle, gpu_op, pts(30, 8060, 120)[None], paths=("auto",))
    x = np.zeros((1, 40, 8060), np.float32)
    x[0, 39, 8059] = 2.0
    assert_equal_to_oracle(oracle, gpu_op, x, paths=("auto",))
    x = np.stack([pts(128, 640, 60), pts(128, 640, 30)])
    x[0, 5, :40] = 0.5                    # values that are not sources: misaligned enumerations
    assert_equal_to_oracle(oracle, gpu_op, x)
    assert_equal_to_oracle(oracle, gpu_op, np.stack([pts(128, 640, 60, lo=0.7), pts(128, 640, 30, lo=0.7)]), 0.001, 0.1)  # eval_NYU.py's thresholds
    mix = np.zeros((4, 128, 640), np.float32)
    mix[0] = pts(128, 640, 100)
    mix[1] = np.where(rng.random((128, 640)) < 0.05, rng.uniform(0.95, 80, (128, 640)), 0)
    mix[2] = np.where(rng.random((128, 640)) < 0.004, rng.uniform(0.95, 80, (128, 640)), 0)
    mix[3] = pts(128, 640, 40)
    mix[3, :60] = 0                       # a sky over a handful of sources
    assert_equal_to_oracle(oracle, gpu_op, mix)
    depth, dt, lbl, status = oracle.fill_batch(mix)
    for want in (("index",), ("dt",), ("depth",), ("depth", "dt")):
        got = run(gpu_op, mix, want=want)
        for k, ref in (("index", lbl), ("dt", dt), ("depth", depth)):
            if k in want:
                assert np.array_equal(got[k], ref), (want, k)


def test_extreme_shapes_vs_oracle(gpu_op, oracle):
    """Shapes that stress the index arithmetic: rows wider than 4096 pixels (more than 64 bit words
    per row), tall thin frames, the largest supported H+W, and a batch with many small frames."""
    rng = np.random.default_rng(31)
    for B, H, W, p in [(1, 3, 5000, 0.01), (1, 5000, 3, 0.01), (1, 40, 8150, 0.002), (1, 8100, 90, 0.002),
                       (70, 17, 23, 0.1), (2, 129, 4097, 0.03)]:
        x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
        assert_equal_to_oracle(oracle, gpu_op, x)
    # one source in a corner of the largest frame: distances up to H + W - 2 = 8189
    x = np.zeros((1, 100, 8091), np.float32)
    x[0, 99, 8090] = 2.0
    assert_equal_to_oracle(oracle, gpu_op, x, paths=("auto",))


def test_l2_extreme_shapes_vs_oracle(pkg, oracle):
    """The shapes of test_extreme_shapes_vs_oracle in the Euclidean mode: rows wider than 4096 pixels (the envelope search's
    LDS beyond 48 KB, one wave per block), frames taller than 4096 rows (more than 128 bands per column), the largest supported
    H + W with one source in a corner (distances up to 8189 in both families: far list and envelope), many small frames."""
    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    rng = np.random.default_rng(32)
    for B, H, W, p in [(1, 3, 5000, 0.01), (1, 5000, 3, 0.01), (1, 40, 8150, 0.002), (1, 8100, 90, 0.002),
                       (70, 17, 23, 0.1), (2, 129, 4097, 0.03), (1, 300, 2000, 0.0008)]:
        x = np.where(rng.random((B, H, W)) < p, rng.uniform(0.95, 80, (B, H, W)), 0).astype(np.float32)
        assert_l2_equal_to_oracle(oracle, op2, x)
    x = np.zeros((1, 100, 8091), np.float32)
    x[0, 99, 8090] = 2.0
    assert_l2_equal_to_oracle(oracle, op2, x)
    x = np.where(rng.random((1, 60, 6000)) < 0.05, rng.uniform(0.95, 80, (1, 60, 6000)), 0).astype(np.float32)
    x[0, :, 1000:5000] = 0  # a dense frame with a hole 4000 pixels wide: the far list and rows of far pixels at large distances
    assert_l2_equal_to_oracle(oracle, op2, x)


def test_random_fuzz_both_metrics(pkg, gpu_op, oracle):
    """Seeded fuzz over shapes, densities, thresholds and structured holes (both metrics)."""
    import torch

    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    rng = np.random.default_rng(4242)
    for t in range(25):
        B, H, W = int(rng.integers(1, 4)), int(rng.integers(1, 200)), int(rng.integers(1, 400))
        p = float(rng.choice([0.001, 0.01, 0.05, 0.2, 0.7]))
        lo = float(rng.choice([0.05, 0.5, 0.95]))
        x = np.where(rng.random((B, H, W)) < p, rng.uniform(lo, 80, (B, H, W)), 0).astype(np.float32)
        if t % 3 == 0:
            r0, r1 = sorted(rng.integers(0, H + 1, 2))
            x[:, r0:r1] = 0
        if t % 4 == 0:
            c0, c1 = sorted(rng.integers(0, W + 1, 2))
            x[:, :, c0:c1] = 0
        st, vt = (0.001, 0.1) if t % 5 == 0 else (0.1, 0.1)
        assert_equal_to_oracle(oracle, gpu_op, x, st, vt)
        depth, dt, idx, status = oracle.fill_batch(x, st, vt, metric="l2")
        res = op2.run(torch.from_numpy(x).to("cuda:0"), st, vt)
        torch.cuda.synchronize()
        assert np.array_equal(res["index"].cpu().numpy(), idx)
        assert np.allclose(res["dt"].cpu().numpy(), dt, rtol=1e-6, atol=0)


def test_outlier_removal_vs_oracle(pkg, oracle):
    """data_read.py:103-128.  The result is the input with some pixels zeroed, so parity is
    exact equality (same float32 accumulation order, float64 decision)."""
    rng = np.random.default_rng(12)
    for H, W in [(352, 1216), (7, 9), (4, 4), (33, 130)]:
        x = np.where(rng.random((H, W)) < 0.3, np.round(rng.uniform(1, 80, (H, W)) * 256) / 256, 0).astype(np.float32)
        # plant outliers: isolated near values in front of far neighbourhoods and vice versa
        x[rng.integers(0, H, 40), rng.integers(0, W, 40)] = 75.0
        got = pkg.outlier_removal(x[None, :, :, None])
        want = oracle.outlier_removal(x)
        assert got.dtype == np.float32 and got.shape == (H, W)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))  # bit patterns: the sign of a removed zero counts
        assert (want != x).any() or H < 8


def test_outlier_removal_special_values(pkg, oracle):
    """The kernel only gathers the 25 taps for pixels that can be removed (v > 1.0) unless a negative value is around:
    negative values, values in (0, 1], NaN and inf must give the composed numpy answer all the same."""
    rng = np.random.default_rng(13)
    H, W = 90, 200
    base = np.where(rng.random((H, W)) < 0.2, np.round(rng.uniform(0.2, 60, (H, W)) * 64) / 64, 0).astype(np.float32)
    cases = {"plain": base.copy()}
    neg = base.copy()
    neg[rng.integers(0, H, 30), rng.integers(0, W, 30)] = -np.round(rng.uniform(1, 50, 30) * 8).astype(np.float32) / 8
    cases["negative"] = neg  # a negative neighbourhood mean removes pixels with v <= 1.0 too
    small = base.copy()
    small[small > 0] = np.minimum(small[small > 0], 1.0)
    small[10, 10] = -40.0
    cases["small values next to a negative"] = small
    nan = base.copy()
    nan[5, 7] = np.nan
    nan[50, 100] = np.inf
    cases["nan / inf"] = nan
    for name, x in cases.items():
        got = pkg.outlier_removal(x[None, :, :, None])
        want = oracle.outlier_removal(x)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), name  # bit patterns (NaN payloads, -0.0 of a removed negative)
    removed_neg = (cases["negative"] < 0) & (oracle.outlier_removal(cases["negative"]) == 0)
    assert not removed_neg.any() or np.signbit(oracle.outlier_removal(cases["negative"])[removed_neg]).all()
    assert (cases["small values next to a negative"] != oracle.outlier_removal(cases["small values next to a negative"])).any()


def test_outlier_removal_fused_into_the_predicates(pkg, gpu_op, oracle):
    """SURVEY 8f-2 / data_read.py:168-169: DTFILL_FLAG_OUTLIER_REMOVAL = outlier_removal() then the fill, without the
    filtered map.  Bit-exact with the composed oracle calls: both kernel families, both metrics, odd widths, frames with
    negative values (the exhaustive second launch) next to frames without, misaligned value lists, the shim."""
    import torch

    rng = np.random.default_rng(14)
    op2 = pkg.device.DtFill(device="cuda:0", metric="l2")
    for (B, H, W, p) in [(2, 352, 1216, 0.05), (3, 61, 131, 0.1), (2, 40, 2300, 0.02), (1, 4, 4, 0.5), (2, 100, 257, 0.004)]:
        x = np.where(rng.random((B, H, W)) < p, np.round(rng.uniform(0.95, 80, (B, H, W)) * 64) / 64, 0).astype(np.float32)
        n = max(1, int(0.002 * H * W))
        for b in range(B):
            x[b, rng.integers(0, H, n), rng.integers(0, W, n)] = 79.0  # far points in front of near neighbourhoods
        if B > 1:
            x[1, rng.integers(0, H), rng.integers(0, W)] = -30.0       # frame 1 takes the exhaustive launch
        xf = np.stack([oracle.outlier_removal(f) for f in x])
        assert (xf != x).any() or H < 8
        for metric, op in (("l1_cv", gpu_op), ("l2", op2)):
            depth, dt, idx, status = oracle.fill_batch(xf, metric=metric)
            for path in ("auto", "general"):
                res = op.run(torch.from_numpy(x).to("cuda:0"), path=path, outlier_removal=True)
                torch.cuda.synchronize()
                got = {k: v.cpu().numpy() for k, v in res.items()}
                assert np.array_equal(got["index"], idx), (metric, path, B, H, W)
                assert np.allclose(got["dt"], dt, rtol=1e-6, atol=0) if metric == "l2" else np.array_equal(got["dt"], dt)
                ok = status == 0
                assert np.array_equal(got["status"] & 1, status) and np.array_equal(got["depth"][ok], depth[ok], equal_nan=True)
    # values that are not sources (misaligned enumerations) + the reference-named entry point
    x = np.where(rng.random((2, 70, 300)) < 0.1, np.round(rng.uniform(0.2, 40, (2, 70, 300)) * 64) / 64, 0).astype(np.float32)
    xf = np.stack([oracle.outlier_removal(f) for f in x])
    want = oracle.fill_batch(xf)[0]
    got = pkg.DT_complete_batch(x[..., None], if_removal=True)
    assert np.array_equal(got[..., 0], want)
    assert np.array_equal(got, pkg.DT_complete_batch(np.stack([pkg.outlier_removal(f[None, :, :, None]) for f in x])[..., None]))


def test_concurrent_streams_and_threads(pkg, oracle):
    """include/dtfill.h: no global state, every call ordered on its stream only.  Four operators (own
    workspaces) driven from four host threads on four streams at once, several rounds, different inputs."""
    import threading

    import torch

    rng = np.random.default_rng(9)
    xs = [np.where(rng.random((3, 200, 300)) < p, rng.uniform(1, 80, (3, 200, 300)), 0).astype(np.float32)
          for p in (0.05, 0.01, 0.002, 0.2)]
    want = [oracle.fill_batch(x) for x in xs]
    ops = [pkg.device.DtFill(device="cuda:0") for _ in xs]
    streams = [torch.cuda.Stream(device="cuda:0") for _ in xs]
    errors = []

    def work(k):
        try:
            xd = torch.from_numpy(xs[k]).to("cuda:0")
            torch.cuda.synchronize()
            for _ in range(10):
                with torch.cuda.stream(streams[k]):
                    res = ops[k].run(xd)
                    got = {n: v.clone() for n, v in res.items()}
                streams[k].synchronize()
                depth, dt, lbl, st = want[k]
                assert np.array_equal(got["index"].cpu().numpy(), lbl)
                assert np.array_equal(got["dt"].cpu().numpy(), dt)
                assert np.array_equal(got["depth"].cpu().numpy(), depth)
        except Exception as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(xs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_generate_multi_channel_vs_oracle(pkg, oracle):
    """net.py:83-122.  Same float32 operations in the same order as the restated oracle, so
    equality is exact (against a real TF the sum order is the only freedom; tolerance 1e-6 relative there)."""
    rng = np.random.default_rng(21)
    for (B, H, W), ts, sn in [((2, 64, 200), 7, 4), ((1, 256, 1216), 7, 4), ((1, 9, 11), 5, 3), ((3, 5, 4), 3, 2)]:
        x = np.where(rng.random((B, H, W, 1)) < 0.05, rng.uniform(1, 80, (B, H, W, 1)), 0).astype(np.float32)
        m = (x > 0.1).astype(np.float32)
        got = pkg.generate_multi_channel(x, m, ts, sn)
        want = oracle.generate_multi_channel(x, m, ts, sn)
        assert len(got) == 4
        for g, w in zip(got, want):
            assert (g is None) == (w is None)
            if g is not None:
                assert g.shape == x.shape and g.dtype == np.float32
                assert np.array_equal(g, w)
    # masks the fast 0 / 1 evaluation must not be used for (any float is a legal weight), masks that disagree with the data
    # (masked zeros, unmasked values: "no masked tap" sums whatever is there), negative zero, a dense first step
    x = np.where(rng.random((2, 70, 150, 1)) < 0.08, rng.uniform(1, 80, (2, 70, 150, 1)), 0).astype(np.float32)
    for m in (np.where(x > 0, rng.choice([0.5, 1.0, 2.0, 3.5], x.shape), 0).astype(np.float32),
              (rng.random(x.shape) < 0.05).astype(np.float32),
              np.where(rng.random(x.shape) < 0.3, np.float32(-0.0), (x > 0.1).astype(np.float32)).astype(np.float32),
              np.ones_like(x)):
        xx = x.copy()
        xx[0, 3, 5, 0] = -0.0
        got, want = pkg.generate_multi_channel(xx, m, 7, 4), oracle.generate_multi_channel(xx, m, 7, 4)
        for g, w in zip(got, want):
            assert np.array_equal(g, w) and np.array_equal(np.signbit(g), np.signbit(w))
    # values around the next step's mask threshold (a step carries a pixel's value on through the later steps only while it
    # stays above 0.001; the pixel right at the edge drops out one step later), tiny and huge values, a NaN
    x = np.where(rng.random((2, 50, 140, 1)) < 0.06, rng.uniform(0.00099, 0.00101, (2, 50, 140, 1)), 0).astype(np.float32)
    x[0, 10, 10, 0] = np.float32(0.001) * np.float32(1.000001)
    x[0, 20, 100, 0] = np.float32(0.001) * np.float32(1.000002)
    x[1, 5, 5, 0] = 1e-30
    x[1, 30, 60, 0] = 3e38
    x[1, 40, 20, 0] = np.nan
    for m in ((x > 0).astype(np.float32), np.isfinite(x).astype(np.float32) * (x != 0)):
        got, want = pkg.generate_multi_channel(x, m.astype(np.float32), 7, 4), oracle.generate_multi_channel(x, m.astype(np.float32), 7, 4)
        for g, w in zip(got, want):
            assert np.array_equal(g, w, equal_nan=True) and np.array_equal(np.signbit(g), np.signbit(w))
    for sn in (2, 3):
        got, want = pkg.generate_multi_channel(x, (x > 0).astype(np.float32), 7, sn), oracle.generate_multi_channel(x, (x > 0).astype(np.float32), 7, sn)
        for g, w in zip(got, want):
            assert (g is None) == (w is None) and (g is None or np.array_equal(g, w, equal_nan=True))


def test_rows_handed_on_for_every_tile_height(gpu_op, oracle):
    """The window kernel walks its tile in groups of eight rows per lane and looks for the rows it has to hand on (a pixel
    farther than the halo from every source) by the same slots: every height modulo 8, a band of columns without sources wide
    enough that some rows -- not all -- hold such a pixel, and the same under an empty top half (found by the seeded random test
    when the hand-over still scanned the pixels in raster order)."""
    rng = np.random.default_rng(5150)
    for H in list(range(1, 20)) + [23, 24, 25, 87, 88, 89, 95, 96, 97]:
        W = int(rng.integers(120, 200))
        x = np.where(rng.random((2, H, W)) < 0.03, rng.uniform(0.95, 80, (2, H, W)), 0).astype(np.float32)
        x[:, :, W // 3 :] = 0                                  # distances up to 2 W / 3 on the right
        x[1, rng.integers(0, H), W - 1 - int(rng.integers(0, 30))] = 7.0   # one source in the band: only some rows stay far
        if H > 4:
            x[0, : H // 2] = 0
        assert_equal_to_oracle(oracle, gpu_op, x)


def test_shape_errors(gpu_op, pkg):
    import torch

    with pytest.raises(pkg.DtfillError):
        gpu_op.run(torch.zeros((1, 5000, 5000), device="cuda:0"))
    with pytest.raises(ValueError):
        gpu_op.run(torch.zeros((5, 5), device="cuda:0"))


def test_single_undecided_pixel_in_every_byte_lane(gpu_op, oracle):
    """Exactly ONE pixel of the frame lies beyond a fused stage's halo (everything on and past an anti-diagonal
    is a source, the corner pixel is R away from it), for the four positions the pixel can have inside the
    4-pixel groups the kernel un-slices.  The stage must notice that single pixel and hand the frame on: a
    lost "undecided" code in one byte lane once let it through with a garbage label (found by scripts/soak.py)."""
    for R, widths, general in ((16, (157, 158, 159, 160), False), (17, (157, 158, 159, 160), True), (33, (125, 126, 127, 128), True)):
        for W in widths:
            H = 40
            i, j = np.mgrid[0:H, 0:W]
            x = np.where(i + (W - 1 - j) >= R, 5.0 + 0.01 * ((i * 7 + j * 3) % 97), 0.0).astype(np.float32)[None]
            depth, dt, lbl, status = oracle.fill_batch(x, 0.1, 0.1)
            assert dt[0, 0, W - 1] == R and (dt[0] > R - 1).sum() == 1
            assert bool(dt[0].max() > 16) == general
            got = run(gpu_op, x, path="auto")
            assert np.array_equal(got["dt"], dt) and np.array_equal(got["index"], lbl), (R, W)
            assert np.array_equal(got["depth"], depth) and bool(got["general"][0]) == general, (R, W)
            # and mirrored: the top-left corner
            xm = np.ascontiguousarray(x[:, :, ::-1])
            depth, dt, lbl, status = oracle.fill_batch(xm, 0.1, 0.1)
            got = run(gpu_op, xm, path="auto")
            assert np.array_equal(got["dt"], dt) and np.array_equal(got["index"], lbl), (R, W, "mirrored")


def test_tie_regions_and_long_tie_chains(gpu_op, oracle):
    """The any-distance kernels decide a pixel with ONE nearest source without a chain; tie pixels hop until they
    stand on such a pixel.  Inputs that stress exactly that: two sources on a diagonal (a whole quadrant is
    equidistant from both: chains of tie pixels that run for hundreds of hops, far beyond k_ties' window, so
    k_tiesx finishes them), lattices (every pixel between four sources ties), a single source (no ties at all),
    sources only in one row / one column, and two far-apart points."""
    frames = []
    a = np.zeros((300, 420), np.float32); a[250, 60] = 1.5; a[246, 64] = 2.5; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[20, 300] = 1.5; a[24, 304] = 2.5; a[150, 100] = 3.5; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[::25, ::30] = 4.0; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[123, 321] = 4.0; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[299, ::7] = 5.0; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[::9, 0] = 6.0; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[0, 0] = 6.0; a[299, 419] = 7.0; frames.append(a)
    a = np.zeros((300, 420), np.float32); a[100, 100] = 6.0; a[100, 140] = 7.0; a[140, 100] = 8.0; a[140, 140] = 9.0
    frames.append(a)
    x = np.stack(frames)
    assert_equal_to_oracle(oracle, gpu_op, x, paths=("general", "auto"))
    for want in (("depth",), ("index",), ("dt",), ("depth", "index")):
        depth, dt, lbl, _ = oracle.fill_batch(x)
        got = run(gpu_op, x, want=want, path="general")
        for k, ref in (("depth", depth), ("dt", dt), ("index", lbl)):
            if k in want:
                assert np.array_equal(got[k], ref), (want, k)
    # widths that are not a multiple of 8 (scalar loads / stores) and of 32 (ragged last plane word)
    rng = np.random.default_rng(31)
    for H, W in [(70, 70), (33, 97), (64, 130), (100, 1030)]:
        y = np.where(rng.random((3, H, W)) < 0.004, rng.uniform(1, 80, (3, H, W)), 0).astype(np.float32)
        y[2, H // 3, W // 3] = 2.0
        y[2, H // 3 - 5, W // 3 + 5] = 3.0
        assert_equal_to_oracle(oracle, gpu_op, y, paths=("general",))


def test_input_pointer_alignment_does_not_matter(gpu_op, oracle):
    """The 16-byte-load mask kernel is only used for 16-byte aligned frames with W % 4 == 0; a frame batch that
    starts 4 bytes into an allocation takes the scalar-load kernel and must give the same maps."""
    import torch

    rng = np.random.default_rng(21)
    B, H, W = 2, 96, 320
    x = np.where(rng.random((B, H, W)) < 0.05, rng.uniform(1.0, 80.0, (B, H, W)), 0.0).astype(np.float32)
    depth, dt, lbl, status = oracle.fill_batch(x, 0.1, 0.1)
    flat = torch.zeros(B * H * W + 3, dtype=torch.float32, device="cuda:0")
    for off in (0, 1, 2, 3):
        view = flat[off:off + B * H * W].view(B, H, W)
        view.copy_(torch.from_numpy(x))
        assert view.data_ptr() % 16 == (flat.data_ptr() + 4 * off) % 16
        res = gpu_op.run(view, 0.1, 0.1)
        torch.cuda.synchronize()
        assert np.array_equal(res["index"].cpu().numpy(), lbl) and np.array_equal(res["dt"].cpu().numpy(), dt), off
        assert np.array_equal(res["depth"].cpu().numpy(), depth), off


def test_empty_bands_in_dense_frames(gpu_op, oracle):
    """An empty band (the sky of a LiDAR frame) on top of, below or inside an otherwise dense frame, holes of
    several sizes below it, bands of several heights around the window kernel's halo (16): whichever kernel family
    takes the frame, the maps are the oracle's.  Also with differing thresholds (value list gather), next to
    dense frames in one batch, and with each subset of the outputs."""
    rng = np.random.default_rng(77)
    H, W = 352, 640

    def dense(p=0.08):
        return np.where(rng.random((H, W)) < p, rng.uniform(1.0, 80.0, (H, W)), 0.0).astype(np.float32)

    frames = []
    a = dense(); a[:100] = 0; frames.append(a)                                   # plain band mode
    a = dense(); a[:60] = 0; a[200:240, 300:345] = 0; frames.append(a)           # + a hole with 16 < d <= 32: stage 2
    a = dense(); a[:60] = 0; a[180:260, 250:340] = 0; frames.append(a)           # + a hole with d > 32: full general
    a = dense(); a[:300] = 0; frames.append(a)                                   # band + margin does not fit: full general
    a = dense(); a[250:] = 0; frames.append(a)                                   # band at the bottom
    a = dense(); a[100:200] = 0; frames.append(a)                                # band in the middle
    a = dense(); a[:40] = 0; frames.append(a)                                    # short bands
    a = dense(); a[:33] = 0; frames.append(a)
    a = dense(); a[:20] = 0; frames.append(a)                                    # 20 rows: the fused stage 2 decides it
    x = np.stack(frames)
    assert_equal_to_oracle(oracle, gpu_op, x)
    got = run(gpu_op, x, path="auto")
    assert got["general"][:8].all()  # every one of them holds a pixel farther than 16 from all sources
    # a band with thresholds that make the value list differ from the source list
    assert_equal_to_oracle(oracle, gpu_op, x[:2], st=0.1, vt=30.0)
    # one such frame between dense frames (two kernel families in one batch)
    mix = np.stack([dense(), frames[0], dense(0.3)])
    assert_equal_to_oracle(oracle, gpu_op, mix)
    # optional outputs
    depth, dt, lbl, status = oracle.fill_batch(mix, 0.1, 0.1)
    for want in (("index",), ("dt",), ("depth",), ("depth", "dt")):
        got = run(gpu_op, mix, want=want)
        for k, ref in (("index", lbl), ("dt", dt), ("depth", depth)):
            if k in want:
                assert np.array_equal(got[k], ref), (want, k)


def test_stray_points_in_the_sky(gpu_op, oracle):
    """The band need not be perfectly empty: a few sources inside it."""
    rng = np.random.default_rng(78)
    H, W = 352, 640
    frames = []
    for stray in (1, 3, 10):
        a = np.where(rng.random((H, W)) < 0.08, rng.uniform(1.0, 80.0, (H, W)), 0.0).astype(np.float32)
        a[:120] = 0
        for _ in range(stray):
            a[rng.integers(0, 115), rng.integers(0, W)] = 7.5
        frames.append(a)
    x = np.stack(frames)
    assert_equal_to_oracle(oracle, gpu_op, x)
    assert run(gpu_op, x, path="auto")["general"].all()


def test_rows_marked_up_front_for_the_any_distance_kernels(gpu_op, oracle, pkg):
    """LiDAR-like frames (empty sky, ring rows): k_frame hands the rows that are too far from every row with a source to the
    any-distance kernels before the window kernel runs, and the window kernel takes the rest of the frame (its tiles shrink to
    their unmarked rows; a tile row left with a few rows is marked whole).  Sky heights around the marking distances and the
    tile heights, bands at the bottom and in the middle, ring spacings, both halos, shapes that do not divide into tiles,
    stray sources inside the sky, differing thresholds -- all bit-exact, and such frames report the general-path bit."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(91)
    x = synth.make("kitti_b32_scanline", B=3)
    assert_equal_to_oracle(oracle, gpu_op, x)
    assert run(gpu_op, x)["general"].all()

    def rings(H, W, top, step, p, bottom=0):
        a = np.where(rng.random((H, W)) < p, rng.uniform(1.0, 80.0, (H, W)), 0.0).astype(np.float32)
        keep = np.zeros(H, bool)
        keep[top:H - bottom:step] = True
        a[~keep] = 0
        return a

    for H, W in ((352, 1216), (417, 1000), (200, 640), (700, 300)):  # (700 rows: k_sky as a launch of its own)
        frames = []
        for top in (9, 17, 40, 87, 88, 100, H - 12):
            frames.append(rings(H, W, top, 4, 0.25))
        frames.append(rings(H, W, 60, 2, 0.15, bottom=70))   # sky on top and an empty band at the bottom
        frames.append(rings(H, W, 30, 8, 0.5))               # rings 8 apart
        frames.append(rings(H, W, 50, 12, 0.6))              # 12 apart: the rows between two rings are near the marking distance
        a = rings(H, W, 0, 1, 0.05); a[90:150] = 0; frames.append(a)     # a band in the middle of an iid frame
        a = rings(H, W, 0, 1, 0.012); a[:70] = 0; frames.append(a)       # thin: halo 32, its own marking distance
        a = rings(H, W, 0, 1, 0.012); a[120:175] = 0; frames.append(a)
        a = rings(H, W, 100, 4, 0.25)                                    # stray sources in the sky, some on a diagonal
        for _ in range(8):
            i, j = int(rng.integers(0, 95)), int(rng.integers(0, W - 40))
            a[i, j] = 3.0
            if rng.random() < 0.5:
                a[i + 5, j + 5] = 4.0
        frames.append(a)
        a = rings(H, W, min(100, H - 40), 4, 0.25)                      # the first source row holds ONE stray source, the rings start
        a[min(70, H - 70)] = 0                                           # 30 rows further down: the window kernel cannot finish
        a[min(70, H - 70), W // 3] = 2.5                                 # that row, so it calls the sky off
        a[:min(70, H - 70)] = 0
        frames.append(a)
        x = np.stack(frames)
        assert_equal_to_oracle(oracle, gpu_op, x)
        assert_equal_to_oracle(oracle, gpu_op, x[:4], st=0.1, vt=30.0)  # value list differs from the source list


# ------------------------------------------------------------------------------------------------
# BASELINE.json's configurations at the sizes and variants SURVEY 8d states
# ------------------------------------------------------------------------------------------------
def test_config1_one_kitti_frame_through_the_reference_functions(pkg, oracle):
    """Config 1 (plumbing shape): ONE 352x1216 frame through nearest_point / DT_complete_batch /
    Distance_Transform, numpy in / numpy out, as demo.py:289-290 and eval_NYU.py:195 call them."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.make("kitti_b1")[0]
    dt, lbl = pkg.nearest_point(x)
    dt0, lbl0 = oracle.nearest_point(x)
    assert dt.shape == (352, 1216) and np.array_equal(dt, dt0) and np.array_equal(lbl, lbl0)
    batch = x[None, :, :, None]
    out = pkg.DT_complete_batch(batch)
    assert out.shape == (1, 352, 1216, 1) and out.dtype == np.float32
    assert np.array_equal(out, oracle.DT_complete_batch(batch))
    one = pkg.Distance_Transform(batch, 0.1)  # the notebooks' thresholds (0.1 / 0.1)
    assert one.shape == (352, 1216) and np.array_equal(one, oracle.Distance_Transform(batch, 0.1))
    # a LiDAR-like frame (empty sky, ring rows) takes the other kernel family through the same functions
    xs = synth.make("kitti_b32_scanline", B=1)[0]
    assert np.array_equal(pkg.Distance_Transform(xs[None, :, :, None], 0.1), oracle.Distance_Transform(xs, 0.1))


@pytest.mark.parametrize("n", [20, 50, 200, 500])
def test_config4_nyu_b64_every_sampling_rate(gpu_op, oracle, pkg, n):
    """Config 4: B=64 frames 480x640 with the reference's sampling pattern (data_read.py:360-364) at the rates the
    drivers use (eval_NYU.py:40: 200; train.py:182-189: 20 / 50 / 500), every frame against the oracle."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.nyu_pattern(64, n=n, seed=100 + n)
    depth, dt, lbl, status = oracle.fill_batch(x)
    got = run(gpu_op, x)
    assert np.array_equal(got["dt"], dt) and np.array_equal(got["index"], lbl)
    assert np.array_equal(got["depth"], depth) and not got["status"].any() and not status.any()
    assert got["general"].all()  # 20 .. 500 points in 307200 pixels: never the window kernel


def test_config4_nyu_misaligned_thresholds_full_size(gpu_op, oracle, pkg):
    """eval_NYU.py's own thresholds on NYU-range depths: sources are x >= 0.999 (eval_NYU.py:115) but the value list
    is x > 0.1 (:124), and depths start near 0.7 m -- every value in (0.1, 0.999) shifts the labels that follow it
    (SURVEY 0 fact 3).  480x640, B=8, against the oracle."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.nyu_pattern(8, n=200, seed=9, lo=0.7, hi=10.0)
    assert ((x > 0.1) & (x < 0.999)).any()
    assert_equal_to_oracle(oracle, gpu_op, x, st=0.001, vt=0.1)
    one = x[3]
    assert np.array_equal(pkg.Distance_Transform(one[None, :, :, None]), oracle.Distance_Transform(one))  # defaults: 0.001 / 0.1


def test_config5_synth2048_b16_properties(gpu_op, oracle, pkg):
    """Config 5's per-GPU share at full size (B=16 frames of 2048x2048, 1 % valid): size-independent properties on
    every frame, the full oracle comparison on two of them."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.make("synth2048_b16")
    got = run(gpu_op, x)
    B, H, W = x.shape
    ii, jj = np.indices((H, W), dtype=np.int32)
    for b in range(B):
        src = x[b] >= 0.9
        pos = np.argwhere(src).astype(np.int32)
        lbl = got["index"][b]
        assert lbl.min() >= 1 and lbl.max() <= len(pos)
        l1 = np.abs(ii - pos[lbl - 1, 0]) + np.abs(jj - pos[lbl - 1, 1])  # every label is a true L1-nearest source
        assert np.array_equal(l1.astype(np.float32), got["dt"][b])
        assert np.array_equal(lbl[src], np.arange(1, len(pos) + 1))
        assert np.array_equal(got["depth"][b], x[b][pos[lbl - 1, 0], pos[lbl - 1, 1]])
    for b in (0, 15):
        depth, dt, lbl, status = oracle.fill_batch(x[b : b + 1])
        assert np.array_equal(got["dt"][b], dt[0]) and np.array_equal(got["index"][b], lbl[0])
        assert np.array_equal(got["depth"][b], depth[0])


def test_fill_sharded_on_the_gpu(pkg, oracle):
    """The multi-GPU entry point with the real operator: one rank here (the group of one), so the shard is the whole
    batch and the device-to-host copies land in the shared pinned slab; two ranks when the box has two GPUs."""
    import torch

    rng = np.random.default_rng(77)
    x = np.where(rng.random((5, 96, 320)) < 0.05, rng.uniform(1, 80, (5, 96, 320)), 0).astype(np.float32)
    x[2, :60] = 0  # one frame for the any-distance kernels
    depth, dt, lbl, _ = oracle.fill_batch(x)
    tm = {}
    out = pkg.fill_sharded(x, timings=tm)
    assert np.array_equal(out["depth"], depth) and np.array_equal(out["dt"], dt) and np.array_equal(out["index"], lbl)
    assert tm["compute_ms"] > 0 and tm["gather_ms"] >= 0
    xb = x.copy()
    xb[4] = 0  # numpy's IndexError in depth_list[label_list-1] (tools.py:26): raised for the batch
    with pytest.raises(IndexError):
        pkg.fill_sharded(xb)
    assert np.array_equal(pkg.fill_sharded(xb, want=("dt",))["dt"], oracle.fill_batch(xb)[1])
    if torch.cuda.device_count() >= 2:
        import socket
        import subprocess
        import sys

        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sharded_worker.py")
        rc = subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                              "--master-addr", "127.0.0.1", "--master-port", str(port), worker])
        assert rc == 0


def test_depth_epilogue_fused_into_the_stores(gpu_op, oracle, pkg):
    """the drivers' next lines after the fill -- rows 96: (demo.py:292-293) and the depth floor
    relu(d - 0.9) + 0.9 (eval_NYU.py:205) -- folded into the depth stores of both kernel families and of k_tiesx;
    bit-exact with the composed oracle calls."""
    import torch

    synth = importlib.import_module(pkg.__name__ + ".synth")
    rng = np.random.default_rng(5)
    xs = [synth.make("kitti_b32", B=3), synth.make("kitti_b32_scanline", B=3)]
    a = np.zeros((2, 300, 420), np.float32)  # diagonal pairs: chains that cross tiles and the cut row (k_tiesx)
    a[0, 250, 60] = 0.95; a[0, 246, 64] = 2.5; a[1, 20, 300] = 0.95; a[1, 24, 304] = 2.5; a[1, 150, 100] = 0.92
    xs.append(a)
    xs.append(np.where(rng.random((2, 97, 130)) < 0.01, rng.uniform(0.9, 3.0, (2, 97, 130)), 0).astype(np.float32))
    for x in xs:
        depth, dt, lbl, status = oracle.fill_batch(x)
        assert not status.any()
        for r0, fl in ((96, None), (0, 0.9), (96, 0.9), (17, 1.5)):
            want = depth[:, r0:]
            if fl is not None:
                want = oracle.depth_floor(want, fl)
            for path in PATHS:
                xd = torch.from_numpy(x).to("cuda:0")
                res = gpu_op.run(xd, path=path, depth_rows_from=r0, depth_floor=fl)
                torch.cuda.synchronize()
                assert tuple(res["depth"].shape) == want.shape
                assert np.array_equal(res["depth"].cpu().numpy(), want), (x.shape, r0, fl, path)
                assert np.array_equal(res["dt"].cpu().numpy(), dt) and np.array_equal(res["index"].cpu().numpy(), lbl)
    # through the reference-named functions
    batch = xs[0][:2, :, :, None]
    assert np.array_equal(pkg.DT_complete_batch(batch, first_row=96), oracle.kitti_rows(oracle.DT_complete_batch(batch)))
    one = xs[1][0]
    assert np.array_equal(pkg.Distance_Transform(one, 0.1, floor=0.9), oracle.depth_floor(oracle.Distance_Transform(one, 0.1)))


def test_reference_functions_return_arrays_of_their_own(pkg, oracle):
    """numpy in / numpy out through the shim: the arrays that come back are the page-locked DMA targets themselves, and a
    later call never writes into what an earlier one returned (the reference builds fresh arrays, tools.py:29-35).  Also the
    cropped + floored form at a real batch size (demo.py:292-293 with B = 8: several frames per transfer chunk)."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.kitti_iid(8, p=0.05, seed=5)
    a = pkg.DT_complete_batch(x[..., None])
    keep = a.copy()
    x2 = synth.kitti_scanline(8, seed=6)
    outs = [pkg.DT_complete_batch(x2[..., None]) for _ in range(5)]  # more calls than any fixed ring of buffers
    assert np.array_equal(a, keep) and a.dtype == np.float32 and a.shape == (8, 352, 1216, 1)
    want2 = oracle.DT_complete_batch(x2[..., None])
    assert all(np.array_equal(o, want2) for o in outs)
    row = outs[0][3, 100]  # a slice keeps its buffer alive on its own
    ref_row = row.copy()
    del outs, a
    for _ in range(4):
        pkg.DT_complete_batch(x[..., None])
    assert np.array_equal(row, ref_row)
    got = pkg.DT_complete_batch(x2[..., None], first_row=96, floor=0.9)
    assert got.shape == (8, 256, 1216, 1)
    assert np.array_equal(got, oracle.depth_floor(want2[:, 96:], 0.9))
    dt, lbl = pkg.nearest_point(x2[0])
    d0, l0 = oracle.nearest_point(x2[0])
    assert np.array_equal(dt, d0) and np.array_equal(lbl, l0)


def test_config3_kitti_b256_through_fill_sharded(pkg, oracle):
    """BASELINE.json config 3 as ONE batch: 256 KITTI frames through fill_sharded (here the group of one: every frame is this
    GPU's; on 8 GPUs each rank takes 32 of them and the same slab is filled from 8 devices).  Properties at full size --
    sources keep their own depth and distance 0, every filled pixel carries the depth of the source its label names, the
    distance is the L1 distance to that source, no frame reports an error -- and three frames against the oracle."""
    synth = importlib.import_module(pkg.__name__ + ".synth")
    x = synth.kitti_iid(256, p=0.05, seed=1)
    tm = {}
    out = pkg.fill_sharded(x, timings=tm)
    assert all(out[k].shape == (256, 352, 1216) for k in ("depth", "dt", "index"))
    src = x >= 0.9
    assert np.array_equal(out["depth"][src], x[src]) and (out["dt"][src] == 0).all() and (out["dt"][~src] > 0).all()
    ii, jj = np.indices((352, 1216))
    for b in (0, 100, 255):
        pos = np.argwhere(src[b])
        lab = out["index"][b]
        assert lab.min() >= 1 and lab.max() <= len(pos)
        si, sj = pos[lab - 1, 0], pos[lab - 1, 1]
        assert np.array_equal(np.abs(ii - si) + np.abs(jj - sj), out["dt"][b].astype(np.int64))
        assert np.array_equal(out["depth"][b], x[b][si, sj])
    depth, dt, lbl, status = oracle.fill_batch(x[[3, 131, 254]])
    for k, b in enumerate((3, 131, 254)):
        assert np.array_equal(out["depth"][b], depth[k]) and np.array_equal(out["dt"][b], dt[k]) and np.array_equal(out["index"][b], lbl[k])
    pkg.release_host_slab()
