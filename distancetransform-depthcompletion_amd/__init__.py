"""MI355X-native distance-transform + nearest-valid-depth fill operator.

Drop-in for the preprocessing path of placeforyiming/DistanceTransform-DepthCompletion
(solution_DeepNet/tools.py, demo.py:78-106, eval_NYU.py:114-133): same function names, numpy in /
numpy out, computed by hand-written gfx950 HIP kernels behind the C ABI of include/dtfill.h.

The directory name contains a '-', so import it with
    importlib.import_module("distancetransform-depthcompletion_amd")
or through the `dtfill_amd` alias module at the repository root.
"""
from . import _lib
from ._lib import METRICS, DtfillError, build, load
from .sharding import shard_range, gather_frames, fill_sharded, release_host_slab
from .tools import DT_complete_batch, Distance_Transform, generate_multi_channel, nearest_point, outlier_removal
from .postfill import Result, Result_NYU, depth_floor, depth_to_png16, kitti_rows, nyu_eval_crop


def __getattr__(name):
    # device.py imports torch; keep `import package` cheap for tooling that only needs the ABI
    if name in ("DtFill", "fill", "default_op", "device"):
        import importlib

        mod = importlib.import_module(__name__ + ".device")
        return mod if name == "device" else getattr(mod, name)
    raise AttributeError(name)


__all__ = [
    "nearest_point", "DT_complete_batch", "Distance_Transform", "outlier_removal", "generate_multi_channel", "fill", "DtFill",
    "shard_range", "gather_frames", "fill_sharded", "release_host_slab", "build", "load", "METRICS", "DtfillError",
    "Result", "Result_NYU", "depth_floor", "depth_to_png16", "kitti_rows", "nyu_eval_crop",
]
