"""Measures k_outlier (data_read.py:103-128 on the device): frames/s and achieved GB/s against the
8 B/pixel it has to move (read f32, write f32).  Run on the GPU box."""
import importlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x = torch.from_numpy(synth.make("kitti_b32")).cuda()
for _ in range(5):
    pkg.device.outlier_removal_device(x)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
K = 50
for _ in range(K):
    pkg.device.outlier_removal_device(x)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / K
B, H, W = x.shape
print(json.dumps({"op": "outlier_removal", "frames_per_s": round(B / ms * 1e3, 1), "ms_per_batch": round(ms, 4),
                  "achieved_GBs": round(8 * B * H * W / ms / 1e6, 1), "peak_GBs": 8000.0,
                  "frac": round(8 * B * H * W / ms / 1e6 / 8000.0, 4), "note": "includes torch.empty_like per call"}))

# the loader's order of things (data_read.py:168-169, then tools.py:13-35): filter, then fill -- as two passes and as one
op = pkg.device.DtFill(device="cuda:0")


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(); e0.record()
    for _ in range(K):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / K


two = timed(lambda: op.run(pkg.device.outlier_removal_device(x)))
one = timed(lambda: op.run(x, outlier_removal=True))
plain = timed(lambda: op.run(x))
print(json.dumps({"op": "outlier_removal + fill (kitti_b32, l1_cv)", "two_passes_ms": round(two, 4), "fused_ms": round(one, 4),
                  "fill_alone_ms": round(plain, 4), "fused_frames_per_s": round(B / one * 1e3, 1),
                  "two_passes_frames_per_s": round(B / two * 1e3, 1)}))
