"""ctypes binding of libdtfill.so (the C ABI declared in include/dtfill.h).

The HIP library is the product: there is no CPU fallback here.  If the shared object is missing
or cannot be loaded, every entry point raises -- loudly -- instead of computing something else.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO_PATH = os.path.join(CSRC, "libdtfill.so")

METRIC_L1_CV = 0
METRIC_L2 = 1
METRICS = {"l1_cv": METRIC_L1_CV, "l2": METRIC_L2}

FRAME_OK = 0
FRAME_INDEX_ERROR = 1  # bit
FRAME_GENERAL_PATH = 2  # bit, informational
FLAG_GENERAL_ONLY = 1
FLAG_FUSED_ONLY = 2
FLAG_OUTLIER_REMOVAL = 4  # outlier_removal() (data_read.py:103-128) in front of the predicates
PATHS = {"auto": 0, "general": FLAG_GENERAL_ONLY, "fused": FLAG_FUSED_ONLY}

# every symbol include/dtfill.h declares (tests/test_abi.py checks the .so exports exactly these)
SYMBOLS = (
    "dtfill_abi_version",
    "dtfill_strerror",
    "dtfill_workspace_bytes",
    "dtfill_batch",
    "dtfill_batch_flags",
    "dtfill_batch_epilogue",
    "dtfill_num_kernels",
    "dtfill_kernel_name",
    "dtfill_batch_timed",
    "dtfill_pass_stats",
    "dtfill_outlier_removal",
    "dtfill_generate_multi_channel",
    "dtfill_crop_floor",
    "dtfill_png16",
    "dtfill_metrics_workspace_bytes",
    "dtfill_metrics",
)
STATS = ("all", "window", "anydist", "sky", "points", "colt")  # DTFILL_STATS_*
METRICS_KITTI = 0
METRICS_NYU = 1
METRICS_COLUMNS = ("mse", "rmse", "mae", "irmse", "imae", "delta1", "delta2", "delta3", "count")

_lib = None


class DtfillError(RuntimeError):
    """A negative DTFILL_ERR_* return code from the C ABI."""

    def __init__(self, code, msg):
        super().__init__("dtfill error %d: %s" % (code, msg))
        self.code = code


def build(force=False):
    """Compile csrc/dtfill.hip for gfx950 with hipcc (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp"))]
    srcs.append(os.path.join(_HERE, "..", "include", "dtfill.h"))
    stale = force or not os.path.exists(SO_PATH) or os.path.getmtime(SO_PATH) < max(os.path.getmtime(f) for f in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", CSRC, "libdtfill.so"])
    return SO_PATH


def load():
    """Load libdtfill.so; raises ImportError if it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "libdtfill.so is missing at %s -- build it with __graft_entry__.build() or "
            "`make -C %s`; this package has no CPU fallback" % (SO_PATH, CSRC)
        )
    L = ctypes.CDLL(SO_PATH)
    vp, ci, cf, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
    L.dtfill_abi_version.argtypes = []
    L.dtfill_abi_version.restype = ci
    L.dtfill_strerror.argtypes = [ci]
    L.dtfill_strerror.restype = ctypes.c_char_p
    L.dtfill_workspace_bytes.argtypes = [ci, ci, ci, ci]
    L.dtfill_workspace_bytes.restype = sz
    L.dtfill_batch.argtypes = [vp, ci, ci, ci, cf, cf, ci, vp, vp, vp, vp, vp, sz, vp]
    L.dtfill_batch.restype = ci
    L.dtfill_batch_flags.argtypes = [vp, ci, ci, ci, cf, cf, ci, vp, vp, vp, vp, vp, sz, vp, ctypes.c_uint]
    L.dtfill_batch_flags.restype = ci
    L.dtfill_batch_epilogue.argtypes = [vp, ci, ci, ci, cf, cf, ci, vp, vp, vp, vp, vp, sz, vp, ctypes.c_uint, ci, ci, cf]
    L.dtfill_batch_epilogue.restype = ci
    L.dtfill_num_kernels.argtypes = [ci]
    L.dtfill_num_kernels.restype = ci
    L.dtfill_kernel_name.argtypes = [ci, ci]
    L.dtfill_kernel_name.restype = ctypes.c_char_p
    L.dtfill_batch_timed.argtypes = [vp, ci, ci, ci, cf, cf, ci, vp, vp, vp, vp, vp, sz, vp, ctypes.c_uint, vp]
    L.dtfill_batch_timed.restype = ci
    L.dtfill_pass_stats.argtypes = [vp, sz, ci, ci, ci, ci, vp, vp]
    L.dtfill_pass_stats.restype = ci
    L.dtfill_outlier_removal.argtypes = [vp, ci, ci, ci, vp, vp]
    L.dtfill_outlier_removal.restype = ci
    L.dtfill_generate_multi_channel.argtypes = [vp, vp, ci, ci, ci, ci, ci, vp, vp, vp, vp]
    L.dtfill_generate_multi_channel.restype = ci
    L.dtfill_crop_floor.argtypes = [vp, ci, ci, ci, ci, ci, ci, ci, ci, cf, vp, vp]
    L.dtfill_crop_floor.restype = ci
    L.dtfill_png16.argtypes = [vp, ci, ci, ci, ci, ci, cf, cf, cf, cf, vp, vp]
    L.dtfill_png16.restype = ci
    L.dtfill_metrics_workspace_bytes.argtypes = [ci]
    L.dtfill_metrics_workspace_bytes.restype = sz
    L.dtfill_metrics.argtypes = [vp, vp, ci, ctypes.c_longlong, ci, vp, vp, sz, vp]
    L.dtfill_metrics.restype = ci
    _lib = L
    return L


def check(code):
    if code != 0:
        raise DtfillError(code, load().dtfill_strerror(code).decode())
