"""What the reference's drivers do with a filled frame, and its metrics (numpy in, numpy out, computed on the device).

  Result, Result_NYU   evaluation.py:11-123 / :125-239 -- same attributes and methods (evaluate, update,
                       finalize, set_to_worst); evaluate() runs on the current HIP device
  kitti_rows           lidar_batch[:, 96:, :, :]                  demo.py:292-293
  nyu_eval_crop        np.squeeze(x)[6:234, 8:312]                eval_NYU.py:202-203
  depth_floor          np.squeeze(tf.nn.relu(d - 0.9) + 0.9)      eval_NYU.py:205, test.py:133
  depth_to_png16       floor, clip(0, 100), 96 x row 0 on top, * 256, uint16   test.py:133-148
The drivers write these steps inline; the function names are ours, the arithmetic (float32, one rounding per
step) is theirs.
"""
import numpy as np

from . import device as _device
from .tools import _as_f32_frames


def _to_device(a, ndim_frame=2):
    import torch

    a = _as_f32_frames(a)
    lead = a.shape[:a.ndim - ndim_frame]
    a3 = np.ascontiguousarray(a.reshape((-1,) + a.shape[a.ndim - ndim_frame:]))
    return torch.from_numpy(a3).to(_device.default_op().device), lead


def kitti_rows(lidar_batch, first_row=96):
    """demo.py:292-293: lidar_batch[:, 96:, :, :] for a [B,H,W,1] (or [B,H,W]) batch."""
    a = _as_f32_frames(lidar_batch)
    chan = a.ndim == 4
    if chan and a.shape[-1] != 1 or a.ndim not in (3, 4):
        raise ValueError("kitti_rows expects [B,H,W,1] or [B,H,W]")
    x, _ = _to_device(a[..., 0] if chan else a)
    out = _device.crop_floor_device(x, rows=(first_row, x.shape[1])).cpu().numpy()
    return out[..., None] if chan else out


def nyu_eval_crop(frame, rows=(6, 234), cols=(8, 312)):
    """eval_NYU.py:202-203: np.squeeze(x)[6:234, 8:312]."""
    a = np.squeeze(_as_f32_frames(frame))
    if a.ndim != 2:
        raise ValueError("nyu_eval_crop expects an array squeezable to [H,W]")
    x, _ = _to_device(a)
    return _device.crop_floor_device(x, rows=rows, cols=cols).cpu().numpy()[0]


def depth_floor(depth, floor=0.9):
    """eval_NYU.py:205 / test.py:133: np.squeeze(relu(depth - 0.9) + 0.9), float32."""
    a = np.squeeze(_as_f32_frames(depth))
    if a.ndim < 2:
        raise ValueError("depth_floor expects at least one [H,W] frame")
    x, lead = _to_device(a)
    return _device.crop_floor_device(x, floor=floor).cpu().numpy().reshape(lead + a.shape[-2:])


def depth_to_png16(depth, pad_top=96, floor=0.9, lo=0.0, hi=100.0, scale=256.0):
    """test.py:133-148: the uint16 image the KITTI submission writer saves ([96 + H, W])."""
    a = np.squeeze(_as_f32_frames(depth))
    if a.ndim != 2:
        raise ValueError("depth_to_png16 expects an array squeezable to [H,W]")
    x, _ = _to_device(a)
    return _device.png16_device(x, pad_top, floor, lo, hi, scale).cpu().numpy()[0]


class _ResultBase(object):
    """The accumulator half of evaluation.py's classes (:12-80 / :126-194), unchanged in meaning."""
    _kind = None
    _fields = ("irmse", "imae", "mse", "rmse", "mae", "absrel", "squared_rel", "lg10", "delta1", "delta2", "delta3",
               "data_time", "gpu_time", "silog", "photometric")

    def __init__(self):
        for f in self._fields:
            setattr(self, f, 0)
        self.count = 0.0

    def set_to_worst(self):
        for f in ("irmse", "imae", "mse", "rmse", "mae", "absrel", "squared_rel", "lg10", "silog"):
            setattr(self, f, np.inf)
        for f in ("delta1", "delta2", "delta3", "data_time", "gpu_time"):
            setattr(self, f, 0)

    def finalize(self):
        for f in self._fields:
            setattr(self, f, getattr(self, f) / self.count)

    def update(self, irmse, imae, mse, rmse, mae, absrel, squared_rel, lg10, delta1, delta2, delta3, gpu_time,
               data_time, silog, photometric=0):
        self.count += 1.0
        for f, v in (("irmse", irmse), ("imae", imae), ("mse", mse), ("rmse", rmse), ("mae", mae), ("absrel", absrel),
                     ("squared_rel", squared_rel), ("lg10", lg10), ("delta1", delta1), ("delta2", delta2),
                     ("delta3", delta3), ("data_time", data_time), ("gpu_time", gpu_time), ("silog", silog),
                     ("photometric", photometric)):
            setattr(self, f, getattr(self, f) + v)

    def evaluate(self, output, target, photometric=0):
        """All elements of output / target (any equal shapes) count as ONE sample, as in the reference."""
        import torch

        o, t = _as_f32_frames(output), _as_f32_frames(target)
        if o.shape != t.shape:
            raise IndexError("boolean index did not match indexed array: output %s, target %s" % (o.shape, t.shape))
        dev = _device.default_op().device
        od = torch.from_numpy(np.ascontiguousarray(o.reshape(1, -1))).to(dev)
        td = torch.from_numpy(np.ascontiguousarray(t.reshape(1, -1))).to(dev)
        row = _device.metrics_device(od, td, self._kind).cpu().numpy()[0]
        m = dict(zip(_device._lib.METRICS_COLUMNS, row.tolist()))
        self.mse, self.rmse, self.mae, self.irmse, self.imae = m["mse"], m["rmse"], m["mae"], m["irmse"], m["imae"]
        if self._kind == "nyu":
            self.delta1, self.delta2, self.delta3 = m["delta1"], m["delta2"], m["delta3"]
        self.photometric = float(photometric)


class Result(_ResultBase):
    """evaluation.py:11-123 (KITTI: mm and 1/km)."""
    _kind = "kitti"


class Result_NYU(_ResultBase):
    """evaluation.py:125-239 (NYU: metres, relative mae, delta accuracies)."""
    _kind = "nyu"
