import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_cases():
    z = np.load(os.path.join(GOLDEN, "cases.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(GOLDEN, "digests.json")))
    cases = {}
    for name in meta["cases"]:
        cases[name] = {k: z[name + "/" + k] for k in ("x", "thr", "dt", "lbl", "depth", "status")}
    return cases, meta["digests"]


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def load_l2_cases():
    """The scipy-pinned fixtures of the l2 mode (tests/golden/make_golden_l2.py): x, squared distances, canonical nearest
    source (raster index, -1 without sources)."""
    z = np.load(os.path.join(GOLDEN, "l2_cases.npz"), allow_pickle=False)
    meta = json.load(open(os.path.join(GOLDEN, "l2_digests.json")))
    return {name: {k: z[name + "/" + k] for k in ("x", "d2", "near")} for name in meta["cases"]}, meta["digests"]


def labels_from_nearest(x, near, src_thr=0.1):
    """1-based raster rank of the source at raster index `near` (0 where near < 0): the label the l2 mode reports."""
    src = ~((np.float32(1.0) - x) > np.float32(src_thr))
    rank = (np.cumsum(src.ravel()) * src.ravel()).astype(np.int32)
    return np.where(near.ravel() >= 0, rank[np.maximum(near.ravel(), 0)], 0).reshape(x.shape).astype(np.int32)
