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
