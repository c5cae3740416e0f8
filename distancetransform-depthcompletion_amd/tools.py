"""Drop-in mirror of the reference operator interface (numpy in, numpy out).

Same names, argument meaning and error behaviour as
  solution_DeepNet/tools.py:7-35      nearest_point, DT_complete_batch
  solution_DeepNet/eval_NYU.py:114-133 nearest_point (threshold 0.001), Distance_Transform
but the work is done by libdtfill.so on the current HIP device.  Differences from the reference,
all widening: DT_complete_batch accepts any HxW (the reference hard-codes 352x1216 in its
reshapes, tools.py:25,27), and the thresholds the reference writes as literals are keyword
arguments whose defaults are those literals.
"""
import numpy as np

from . import device as _device


def _as_f32_frames(a):
    """Input checking shared by the three functions: the kernels compute the predicates in
    float32, which is what numpy does for the float32 arrays the reference's loaders produce
    (data_read.py:81-99).  float64 input is accepted when it is exactly float32-representable."""
    a = np.asarray(a)
    if a.dtype == np.float32:
        return a
    if a.dtype.kind not in "fiub":
        raise TypeError("expected a real-valued array, got dtype %s" % a.dtype)
    a32 = a.astype(np.float32)
    if a.dtype.kind == "f" and not np.array_equal(a32.astype(a.dtype), a, equal_nan=True):
        raise TypeError(
            "float64 input that is not exactly representable in float32 is not supported: "
            "the source/value predicates (tools.py:8,22) are evaluated in float32 on the device"
        )
    return a32


def nearest_point(refined_lidar, src_thr=0.1):
    """tools.py:7-10.  Returns (dt float32 [H,W], lbl int32 [H,W]) exactly as
    cv2.distanceTransformWithLabels(uint8((1.0-x) > src_thr), DIST_L1, 5, DIST_LABEL_PIXEL)."""
    x = np.squeeze(_as_f32_frames(refined_lidar))
    if x.ndim != 2:
        raise ValueError("nearest_point expects an array squeezable to [H,W], got shape %s" % (np.shape(refined_lidar),))
    out = _device.default_op().run_numpy(x[None], src_thr=src_thr, val_thr=0.1, want=("dt", "index"))
    return out["dt"][0], out["index"][0]


def DT_complete_batch(lidar_batch, src_thr=0.1, val_thr=0.1, first_row=0, floor=None, if_removal=False):
    """tools.py:13-35.  lidar_batch [B,H,W,C>=1] (channel 0 is used, tools.py:18) ->
    float32 [B,H,W,1].  Raises IndexError like tools.py:26 when a frame's value list is too short.
    first_row / floor fold the caller's next lines into the same pass: demo.py:292-293
    (lidar_batch[:, 96:, :, :] -> first_row=96, result [B,H-96,W,1]) and the depth floor relu(d - 0.9) + 0.9.
    if_removal folds the loader's outlier_removal() (data_read.py:103-128, the flag of data_read.py:168) into the same pass:
    DT_complete_batch(x, if_removal=True) == DT_complete_batch(outlier_removal(x_i) for every frame)."""
    lb = _as_f32_frames(lidar_batch)
    if lb.ndim != 4:
        raise IndexError("too many indices for array: DT_complete_batch indexes lidar_batch[i,:,:,0]")
    x = lb[:, :, :, 0]
    out = _device.default_op().run_numpy(x, src_thr=src_thr, val_thr=val_thr, want=("depth",), depth_rows_from=first_row,
                                         depth_floor=floor, outlier_removal=if_removal)
    return np.expand_dims(out["depth"], axis=-1)  # already a fresh float32 array


def Distance_Transform(lidar, src_thr=0.001, val_thr=0.1, floor=None):
    """eval_NYU.py:120-133 (src_thr=0.001 as eval_NYU.py:115; the notebooks use 0.1).
    One frame squeezable to [H,W]; the result keeps the input's dtype like the reference.
    floor=0.9 folds eval_NYU.py:205 (relu(depth - 0.9) + 0.9, float32) into the same pass."""
    src = np.asarray(lidar)
    x = np.squeeze(_as_f32_frames(src))
    if x.ndim != 2:
        raise ValueError("not enough values to unpack: Distance_Transform expects a frame squeezable to [H,W]")
    if np.count_nonzero(x > np.float32(val_thr)) == 1:
        # eval_NYU.py:125 squeezes the value list; with exactly one valid pixel it becomes 0-d and
        # the gather on the next line raises -- kept, so callers see the reference's behaviour
        raise IndexError("too many indices for array: array is 0-dimensional, but 1 were indexed")
    out = _device.default_op().run_numpy(x[None], src_thr=src_thr, val_thr=val_thr, want=("depth",), depth_floor=floor)
    depth = out["depth"][0]
    return depth.astype(src.dtype) if src.dtype.kind == "f" else depth


def outlier_removal(lidar):
    """data_read.py:103-128 (the loader's optional filter in front of the fill, data_read.py:168-169):
    zero every pixel that exceeds the mean of the valid pixels in its 7x7 diamond by more than 1.0.
    Input squeezable to [H,W]; float32 [H,W] back, like the reference."""
    import torch

    x = np.squeeze(_as_f32_frames(lidar))
    if x.ndim != 2:
        raise ValueError("outlier_removal expects an array squeezable to [H,W], got shape %s" % (np.shape(lidar),))
    xd = torch.from_numpy(np.ascontiguousarray(x[None])).to(_device.default_op().device)
    return _device.outlier_removal_device(xd)[0].cpu().numpy()


def generate_multi_channel(lidar_data, lidar_mask, table_size=7, scale_num=4):
    """net.py:83-122 as a numpy function: lidar_data, lidar_mask [B,H,W,1] -> (lidar_1, .., lidar_4), each
    [B,H,W,1] float32, None beyond scale_num (the reference returns TF tensors of the same shapes)."""
    import torch

    d = _as_f32_frames(lidar_data)
    m = _as_f32_frames(lidar_mask)
    if d.ndim != 4 or d.shape[-1] != 1 or m.shape != d.shape:
        raise ValueError("generate_multi_channel expects lidar_data and lidar_mask of shape [B,H,W,1]")
    dev = _device.default_op().device
    dd = torch.from_numpy(np.ascontiguousarray(d[..., 0])).to(dev)
    mm = torch.from_numpy(np.ascontiguousarray(m[..., 0])).to(dev)
    outs = _device.generate_multi_channel_device(dd, mm, table_size, scale_num)
    return tuple(None if o is None else o.cpu().numpy()[..., None] for o in outs)
