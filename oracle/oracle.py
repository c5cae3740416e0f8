"""ctypes front-end of the CPU oracle + numpy restatement of the reference glue.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py, never from the product package.  PARITY UNPINNED: see the
header of dtfill_oracle.c (cv2 is absent here; the chamfer is restated from OpenCV 3.4).

Reference lines restated here (the glue uses numpy's own fancy indexing, so numpy itself supplies the
negative-index / IndexError behaviour of the reference):
  nearest_point       solution_DeepNet/tools.py:7-10, eval_NYU.py:114-117
  DT_complete_batch   solution_DeepNet/tools.py:13-35 (demo.py:84-106)
  Distance_Transform  solution_DeepNet/eval_NYU.py:120-133
  evaluate_kitti / evaluate_nyu   evaluation.py:82-123 / :196-239 (Result.evaluate, Result_NYU.evaluate)
  depth_floor, kitti_rows, nyu_eval_crop, depth_to_png16   the drivers' inline post-fill steps
                      (demo.py:292-293, eval_NYU.py:202-205, test.py:133-148); tf.nn.relu / tf.clip_by_value
                      restated as np.maximum / np.clip on float32 (tensorflow is absent: unpinned)
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


def _source_mask(frame, src_thr):
    """uint8 mask handed to the transform: 1 = pixel to fill, 0 = source (tools.py:8, eval_NYU.py:115).
    Evaluated by numpy in the frame's own dtype, exactly as the reference does."""
    return np.asarray((1.0 - frame) > src_thr).astype(np.uint8)


def _gather_by_label(frame, labels, val_thr):
    """tools.py:22-27 / eval_NYU.py:124-131: compact the pixels above val_thr in raster order and index
    that list with label-1.  numpy's own fancy indexing supplies the reference's corner cases: label 0
    wraps to the last entry, an index past the list raises IndexError."""
    values = frame[frame > val_thr]
    picked = values[labels.ravel() - 1]
    return picked.reshape(frame.shape)


def nearest_point(refined_lidar, src_thr=0.1):
    """tools.py:7-10 (src_thr=0.1) / eval_NYU.py:114-117 (src_thr=0.001): (dt, lbl) of one frame."""
    frame = np.squeeze(refined_lidar)
    return cv_distance_transform_with_labels(_source_mask(frame, src_thr))


def DT_complete_batch(lidar_batch, src_thr=0.1, val_thr=0.1):
    """tools.py:13-35: channel 0 of every frame of a [B,H,W,C] batch is filled; float32 [B,H,W,1] back.
    (The reference hard-codes 352x1216 in its reshapes, tools.py:25,27; H, W come from the input here.)"""
    filled = []
    for frame in lidar_batch[..., 0]:
        frame = np.squeeze(frame)
        _, labels = nearest_point(frame, src_thr)
        filled.append(_gather_by_label(frame, labels, val_thr))
    return np.asarray(filled)[..., np.newaxis].astype(np.float32)


def Distance_Transform(lidar, src_thr=0.001, val_thr=0.1):
    """eval_NYU.py:120-133 (src_thr=0.001 there, 0.1 in the notebooks): one frame, result in the input's
    dtype.  eval_NYU.py:125 squeezes the value list, so a frame with exactly ONE value makes it 0-d and the
    gather on the next line raises IndexError -- reproduced."""
    frame = np.squeeze(lidar)
    if frame.ndim != 2:
        raise ValueError("Distance_Transform expects a frame squeezable to [H,W]")
    _, labels = nearest_point(frame, src_thr)
    values = np.squeeze(frame[frame > val_thr])
    return values[labels.reshape(1, -1) - 1].reshape(frame.shape)


def generate_multi_channel(lidar_data, lidar_mask, table_size=7, scale_num=4):
    """net.py:83-122 restated (the in-network windowed nearest fill every model class
    carries, weights from create_weight_matrix net.py:71-81).  TensorFlow is absent here: PARITY UNPINNED.
    One step: over the table_size^2 window (zero padding, extract_patches padding='SAME'), s = mask * w with
    w = table_size - |di| - |dj|; the output is the sum of the inputs at the positions where s equals its
    window maximum, divided by (1e-6 + their count) -- including the all-zero case, where every position
    (padding too) ties at 0.  The next step's mask is (output > 0.001).  float32 throughout; the window sum
    is accumulated in tap (row-major) order -- TF's reduction order is unspecified, so against a real TF
    this is exact only up to float32 summation order.
    lidar_data, lidar_mask: [B,H,W,1] (or [B,H,W]).  Returns (lidar_1, .., lidar_4) like the reference,
    None for the scales beyond scale_num."""
    data = np.asarray(lidar_data, np.float32)
    mask = np.asarray(lidar_mask, np.float32)
    squeeze_c = data.ndim == 4
    if squeeze_c:
        data, mask = data[..., 0], mask[..., 0]
    half = (table_size - 1) // 2
    outs = [np.asarray(lidar_data, np.float32)]
    for _ in range(scale_num - 1):
        B, H, W = data.shape
        pd = np.pad(data, ((0, 0), (half, half), (half, half)))
        pm = np.pad(mask, ((0, 0), (half, half), (half, half)))
        taps = [(i, j, np.float32(table_size - abs(i - half) - abs(j - half))) for i in range(table_size) for j in range(table_size)]
        mx = np.zeros((B, H, W), np.float32)
        for i, j, w in taps:
            mx = np.maximum(mx, pm[:, i : i + H, j : j + W] * w)
        acc = np.zeros((B, H, W), np.float32)
        cnt = np.zeros((B, H, W), np.float32)
        for i, j, w in taps:
            sel = (pm[:, i : i + H, j : j + W] * w) == mx
            acc = np.where(sel, (acc + pd[:, i : i + H, j : j + W]).astype(np.float32), acc)
            cnt = cnt + sel.astype(np.float32)
        data = (acc / (np.float32(0.000001) + cnt)).astype(np.float32)
        mask = (data > np.float32(0.001)).astype(np.float32)
        outs.append(data[..., None] if squeeze_c else data)
    return tuple(outs + [None] * (4 - len(outs)))


_DIAMOND7 = np.array([[abs(i - 3) + abs(j - 3) <= 3 for j in range(7)] for i in range(7)])


def outlier_removal(lidar):
    """data_read.py:103-128 restated (the optional step in front of the path,
    data_read.py:168-169).  Two cv2.filter2D correlations with the 7x7 diamond of ones (third-party
    OpenCV again, PARITY UNPINNED): default border BORDER_REFLECT_101; for the float32 frame OpenCV
    accumulates in float32 over the non-zero kernel taps in kernel row-major order; the valid-pixel
    count is computed on a float64 image (np.float), so it is exact; mean, difference and the > 1.0 test
    are float64 by numpy promotion; the result is the frame times (1 - flag): flagged pixels become a zero that
    keeps the pixel's sign, float32."""
    x = np.squeeze(np.asarray(lidar)).astype(np.float32)
    H, W = x.shape
    pad = np.pad(x, 3, mode="reflect")  # numpy 'reflect' == BORDER_REFLECT_101
    acc = np.zeros((H, W), np.float32)
    cnt = np.zeros((H, W), np.float64)
    for i in range(7):
        for j in range(7):
            if _DIAMOND7[i, j]:
                win = pad[i : i + H, j : j + W]
                acc = (acc + win).astype(np.float32)  # sequential float32 accumulation, row-major taps
                cnt += win > 0.1
    mean = acc / (cnt + 0.00001)  # float32 / float64 -> float64
    outlier = ((x - mean) > 1.0).astype(np.float64)  # (...).astype(np.float) in the reference
    # data_read.py:128 multiplies: a removed NEGATIVE pixel comes out as -0.0, not +0.0 (float32 * float64 -> float64 -> float32)
    with np.errstate(invalid="ignore"):
        return (x * (1 - outlier)).astype(np.float32)


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


# ---- post-fill steps of the drivers and the metrics (8f-3) ----------------------------

def depth_floor(d, floor=0.9):
    """relu(d - 0.9) + 0.9 on float32, one rounding per step (eval_NYU.py:205, test.py:133)."""
    d = np.asarray(d, np.float32)
    f = np.float32(floor)
    return (np.maximum(d - f, np.float32(0.0)) + f).astype(np.float32)


def kitti_rows(batch, first_row=96):
    """lidar_batch[:, 96:, :, :] (demo.py:292-293)."""
    return np.ascontiguousarray(np.asarray(batch)[:, first_row:])


def nyu_eval_crop(frame):
    """np.squeeze(x)[6:234, 8:312] (eval_NYU.py:202-203)."""
    return np.ascontiguousarray(np.squeeze(np.asarray(frame))[6:234, 8:312])


def depth_to_png16(depth, pad_top=96, floor=0.9, lo=0.0, hi=100.0, scale=256.0):
    """test.py:133-148: depth floor, clip, pad_top copies of row 0 on top, * 256, astype(uint16)."""
    d = depth_floor(np.squeeze(np.asarray(depth, np.float32)), floor)
    d = np.clip(d, np.float32(lo), np.float32(hi))
    top = np.tile(d[0, :], (pad_top, 1)).astype(np.float32)
    d = np.vstack((top, d)) if pad_top else d
    return (d * np.float32(scale)).astype(np.uint16)


def _both_valid(output, target):
    output, target = np.asarray(output), np.asarray(target)
    keep = np.logical_and(output > 0.01, target > 0.01)  # evaluation.py:85-87 / :199-201
    return output[keep], target[keep]


def evaluate_kitti(output, target):
    """Result.evaluate (evaluation.py:82-123): mm for mse/rmse/mae, 1/km for irmse/imae."""
    import math
    o, t = _both_valid(output, target)
    with np.errstate(all="ignore"):
        err = np.abs(1e3 * o - 1e3 * t)
        mse = np.mean(np.power(err, 2))
        ierr = np.abs((1e-3 * o) ** (-1) - (1e-3 * t) ** (-1))
        return {"mse": float(mse), "rmse": math.sqrt(mse) if mse == mse else float("nan"), "mae": float(np.mean(err)),
                "irmse": float(np.sqrt(np.mean(np.power(ierr, 2)))), "imae": float(np.mean(ierr)),
                "delta1": 0.0, "delta2": 0.0, "delta3": 0.0, "count": float(o.size)}


def evaluate_nyu(output, target):
    """Result_NYU.evaluate (evaluation.py:196-239): metres, RELATIVE mae, delta accuracies."""
    import math
    o, t = _both_valid(output, target)
    with np.errstate(all="ignore"):
        err = np.abs(o - t)
        mse = np.mean(np.power(err, 2))
        worst = np.maximum(o / t, t / o)
        ierr = np.abs(o ** (-1) - t ** (-1))
        imse = np.mean(np.power(ierr, 2))
        return {"mse": float(mse), "rmse": math.sqrt(mse) if mse == mse else float("nan"),
                "mae": float(np.mean(err / t)),
                "irmse": math.sqrt(imse) if imse == imse else float("nan"), "imae": float(np.mean(ierr)),
                "delta1": float(np.mean(worst < 1.25)), "delta2": float(np.mean(worst < 1.25 ** 2)),
                "delta3": float(np.mean(worst < 1.25 ** 3)), "count": float(o.size)}
