"""Generates tests/golden/{cases.npz,digests.json}.  Run from the repo root:
    python tests/golden/make_golden.py

PARITY UNPINNED: cv2 is not installed here and the reference ships no recorded outputs for this
path, so these vectors come from the CPU restatement in oracle/ (see the header of
oracle/dtfill_oracle.c), cross-checked at generation time against scipy's taxicab distances and
the brute-force nearest-source search.  They pin the restatement against regressions and travel
to the GPU box, where the HIP path is compared with them.

Case (hand5x7) is the worked example of SURVEY.md section 8c(1).
"""
import hashlib
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
HERE = os.path.dirname(os.path.abspath(__file__))


def crosscheck(x, src_thr, dt, lbl):
    from scipy import ndimage

    src = ~((np.float32(1.0) - x) > np.float32(src_thr))
    if src.any():
        cdt = ndimage.distance_transform_cdt(~src, metric="taxicab")
        assert np.array_equal(cdt.astype(np.float32), dt), "scipy taxicab distance mismatch"
        pos = np.argwhere(src)
        ii, jj = np.indices(x.shape)
        si, sj = pos[lbl - 1, 0], pos[lbl - 1, 1]
        assert np.array_equal(np.abs(ii - si) + np.abs(jj - sj), dt.astype(np.int64)), "label is not a nearest source"
    else:
        assert (dt == 8192.0).all() and (lbl == 0).all()


def small_cases():
    cases = {}
    x = np.zeros((5, 7), np.float32)
    x[0, 5], x[2, 1], x[4, 4] = 10, 20, 30
    cases["hand5x7"] = (x, 0.1, 0.1)
    cases["nosource_novalue"] = (np.zeros((6, 9), np.float32), 0.1, 0.1)
    # no source (all x < 0.9) but values exist (x > 0.1): label 0 -> depth_list[-1] = last value
    x = np.zeros((6, 9), np.float32)
    x[1, 2], x[4, 7] = 0.5, 0.7
    cases["nosource_values"] = (x, 0.1, 0.1)
    for k, pos in enumerate([(0, 0), (0, 8), (5, 0), (5, 8), (2, 4)]):
        x = np.zeros((6, 9), np.float32)
        x[pos] = 7.5
        cases["single%d" % k] = (x, 0.1, 0.1)
    cases["allsource"] = (np.arange(1, 55, dtype=np.float32).reshape(6, 9), 0.1, 0.1)
    # misalignment: values in (val_thr, 1-src_thr) are in the value list but are
    # not sources, so every later label addresses a shifted entry
    rng = np.random.default_rng(5)
    x = np.where(rng.random((12, 17)) < 0.15, rng.uniform(1, 9, (12, 17)), 0).astype(np.float32)
    x[0, 3], x[0, 9], x[5, 5] = 0.5, 0.3, 0.85
    cases["misaligned"] = (x, 0.1, 0.1)
    # eval_NYU thresholds (src 0.001, val 0.1) with depths from 0.7
    x = np.where(rng.random((14, 19)) < 0.1, rng.uniform(0.7, 10, (14, 19)), 0).astype(np.float32)
    cases["nyu_thresholds"] = (x, 0.001, 0.1)
    # NaN is a source (the mask compare is False) but not a value
    x = np.where(rng.random((8, 11)) < 0.2, rng.uniform(1, 9, (8, 11)), 0).astype(np.float32)
    x[7, 10] = np.nan
    cases["nan_source"] = (x, 0.1, 0.1)
    # ragged sizes around the 64-pixel word / wave width
    for H, W in [(1, 1), (1, 70), (70, 1), (3, 64), (3, 65), (2, 63), (17, 129), (33, 200)]:
        x = np.where(rng.random((H, W)) < 0.08, rng.uniform(1, 80, (H, W)), 0).astype(np.float32)
        if not (x > 0.9).any():
            x[H // 2, W // 2] = 4.0
        cases["ragged_%dx%d" % (H, W)] = (x, 0.1, 0.1)
    # tie-heavy regular lattices
    x = np.zeros((24, 31), np.float32)
    x[::6, ::6] = 3.0
    cases["lattice6"] = (x, 0.1, 0.1)
    x = np.zeros((24, 31), np.float32)
    x[3::5, 2::7] = 3.0
    cases["lattice5x7"] = (x, 0.1, 0.1)
    # random densities
    for k, p in enumerate([0.005, 0.02, 0.05, 0.2, 0.5, 0.9]):
        H, W = 40 + 3 * k, 70 - 5 * k
        x = np.where(rng.random((H, W)) < p, np.round(rng.uniform(1, 80, (H, W)) * 256) / 256, 0).astype(np.float32)
        cases["rand_p%g" % p] = (x, 0.1, 0.1)
    return cases


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    out = {}
    names = []
    for name, (x, st, vt) in small_cases().items():
        depth, dt2, lbl2, status = O.fill_batch(x[None], st, vt)
        dt, lbl = dt2[0], lbl2[0]
        crosscheck(x, st, dt, lbl)
        if min(x.shape) == 1:  # the reference's np.squeeze cannot express 1-pixel-wide frames
            names.append(name)
            out.update({name + "/x": x, name + "/thr": np.array([st, vt], np.float32), name + "/dt": dt,
                        name + "/lbl": lbl, name + "/depth": depth[0], name + "/status": status})
            continue
        dtn, lbln = O.nearest_point(x, st)
        assert np.array_equal(dt, dtn) and np.array_equal(lbl, lbln)
        # the literal numpy glue must agree with the C glue, including the IndexError cases
        # (DT_complete_batch's form: Distance_Transform additionally squeezes a one-element value
        # list to 0-d, eval_NYU.py:125, and then raises -- covered in tests/test_oracle.py)
        try:
            ref = O.DT_complete_batch(x[None, :, :, None], st, vt)[0, :, :, 0]
            assert status[0] == 0 and np.array_equal(ref, depth[0], equal_nan=True), name
        except IndexError:
            assert status[0] == 1, name
        names.append(name)
        out[name + "/x"] = x
        out[name + "/thr"] = np.array([st, vt], np.float32)
        out[name + "/dt"] = dt
        out[name + "/lbl"] = lbl
        out[name + "/depth"] = depth[0]
        out[name + "/status"] = status
    np.savez_compressed(os.path.join(HERE, "cases.npz"), **out)

    # full-size frames: (generator, seed) -> sha256 of the oracle outputs
    dig = {}
    for cfg, B in [("kitti_b32", 2), ("kitti_b32_scanline", 1), ("nyu_b64", 2), ("synth2048_b16", 1), ("kitti_crop256", 2),
                   ("nyu_240x320", 2)]:
        x = synth.make(cfg, B=B)
        depth, dt, lbl, status = O.fill_batch(x)
        for b in range(B):
            crosscheck(x[b], 0.1, dt[b], lbl[b])
        dig[cfg] = dict(B=B, shape=list(x.shape), x=digest(x), dt=digest(dt), lbl=digest(lbl),
                        depth=digest(depth), status=status.tolist())
    json.dump(dict(cases=names, digests=dig), open(os.path.join(HERE, "digests.json"), "w"), indent=1)
    print("wrote", len(names), "cases;", {k: v["shape"] for k, v in dig.items()})


if __name__ == "__main__":
    main()
