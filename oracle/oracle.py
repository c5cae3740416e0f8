"""ctypes front-end of the CPU oracle + numpy restatement of the reference glue.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never from the product package.  PARITY UNPINNED: see the
header of dtfill_oracle.c (cv2 is absent here; the chamfer is restated from OpenCV 3.4).

Reference lines restated here (numpy glue, kept literally so that numpy itself supplies the
negative-index / IndexError behaviour):
  nearest_point       solution_DeepNet/tools.py:7-10, eval_NYU.py:114-117
  DT_complete_batch   solution_DeepNet/tools.py:13-35 (demo.py:84-106)
  Distance_Transform  solution_DeepNet/eval_NYU.py:120-133
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so with gcc (idempotent)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "dtfill_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = build()
        L = ctypes.CDLL(so)
        vp, ci, cf = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
        L.cvdt_l1_labels.argtypes = [vp, ci, ci, vp, vp]
        L.cvdt_l1_labels.restype = None
        L.oracle_nearest_point.argtypes = [vp, ci, ci, cf, vp, vp]
        L.oracle_nearest_point.restype = None
        L.oracle_fill_frame.argtypes = [vp, ci, ci, cf, cf, vp, vp, vp]
        L.oracle_fill_frame.restype = ci
        L.oracle_fill_batch.argtypes = [vp, ci, ci, ci, cf, cf, vp, vp, vp, vp]
        L.oracle_fill_batch.restype = ci
        L.oracle_fill_batch_l2.argtypes = [vp, ci, ci, ci, cf, cf, vp, vp, vp, vp]
        L.oracle_fill_batch_l2.restype = ci
        L.brute_nearest.argtypes = [vp, ci, ci, ci, vp, vp]
        L.brute_nearest.restype = None
        L.edt_l2_labels.argtypes = [vp, ci, ci, vp, vp]
        L.edt_l2_labels.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def cv_distance_transform_with_labels(mask):
    """cv2.distanceTransformWithLabels(mask, DIST_L1, 5, labelType=DIST_LABEL_PIXEL) restated."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    assert mask.ndim == 2
    H, W = mask.shape
    dist = np.empty((H, W), np.float32)
    lab = np.empty((H, W), np.int32)
    lib().cvdt_l1_labels(_p(mask), H, W, _p(dist), _p(lab))
    return dist, lab


def nearest_point(refined_lidar, src_thr=0.1):
    """tools.py:7-10 (src_thr=0.1) / eval_NYU.py:114-117 (src_thr=0.001)."""
    value_mask = np.asarray(1.0 - np.squeeze(refined_lidar) > src_thr).astype(np.uint8)
    dt, lbl = cv_distance_transform_with_labels(value_mask)
    return dt, lbl


def DT_complete_batch(lidar_batch, src_thr=0.1, val_thr=0.1):
    """tools.py:13-35.  The reference hard-codes 352x1216 in its reshapes (tools.py:25,27);
    H, W are taken from the input here so the same restatement serves every config."""
    batch_size = np.shape(lidar_batch)[0]
    new_batch = []
    for i in range(batch_size):
        lidar_single = lidar_batch[i, :, :, 0]
        h, w = np.squeeze(lidar_single).shape
        dt, lbl = nearest_point(lidar_single, src_thr)
        with_value = np.squeeze(lidar_single) > val_thr
        depth_list = np.squeeze(lidar_single)[with_value]
        label_list = np.reshape(lbl, [1, w * h])
        depth_list_all = depth_list[label_list - 1]
        depth_map = np.reshape(depth_list_all, (h, w))
        new_batch.append(depth_map)
    new_batch = np.asarray(new_batch)
    new_batch = np.expand_dims(new_batch, axis=-1)
    return new_batch.astype(np.float32)


def Distance_Transform(lidar, src_thr=0.001, val_thr=0.1):
    """eval_NYU.py:120-133 (src_thr=0.001 there, 0.1 in the notebooks)."""
    lidar = np.squeeze(lidar)
    height, width = np.shape(lidar)
    with_value = lidar > val_thr
    dt, lbl = nearest_point(lidar, src_thr)
    depth_list = np.squeeze(lidar[with_value])
    label_list = np.reshape(lbl, [1, height * width])
    depth_list_all = depth_list[label_list - 1]
    return np.reshape(depth_list_all, (height, width))


def fill_batch(x, src_thr=0.1, val_thr=0.1, metric="l1_cv"):
    """All-C batched path (used for bulk parity and the cpu_baseline timing).
    x: float32 [B,H,W].  Returns depth, dt, index(int32 labels), status(int32 [B]).
    metric "l1_cv" = the reference's cv2 transform; "l2" = exact Euclidean, canonical tie-break."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, H, W = x.shape
    depth = np.empty_like(x)
    dt = np.empty_like(x)
    idx = np.empty(x.shape, np.int32)
    status = np.zeros(B, np.int32)
    fn = lib().oracle_fill_batch if metric == "l1_cv" else lib().oracle_fill_batch_l2
    fn(_p(x), B, H, W, src_thr, val_thr, _p(depth), _p(dt), _p(idx), _p(status))
    return depth, dt, idx, status


def brute_nearest(mask, metric):
    """Exhaustive nearest source (tiny frames). metric 1=L1, 2=squared L2; ties -> smallest raster index."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = mask.shape
    d = np.empty((H, W), np.int32)
    near = np.empty((H, W), np.int32)
    lib().brute_nearest(_p(mask), H, W, metric, _p(d), _p(near))
    return d, near


def edt_l2(mask):
    """Exact squared-Euclidean transform + nearest index, canonical tie-break (the `l2` mode)."""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    H, W = mask.shape
    d = np.empty((H, W), np.int32)
    near = np.empty((H, W), np.int32)
    lib().edt_l2_labels(_p(mask), H, W, _p(d), _p(near))
    return d, near
