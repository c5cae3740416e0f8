"""What a reference-style caller sees: numpy in / numpy out through the drop-in functions (PCIe included),
and the device-resident latency of a single frame.  Run on the GPU box."""
import importlib, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x32 = synth.make("kitti_b32")
out = {}
for B in (1, 32):
    batch = x32[:B, :, :, None]
    for _ in range(3):
        pkg.DT_complete_batch(batch)
    t0 = time.perf_counter(); n = 20
    for _ in range(n):
        pkg.DT_complete_batch(batch)
    dt = (time.perf_counter() - t0) / n
    out["DT_complete_batch_numpy_B%d" % B] = {"ms_per_call": round(dt * 1e3, 3), "frames_per_s": round(B / dt, 1)}
op = pkg.device.DtFill(device="cuda:0")
for B in (1, 4, 32):
    xd = torch.from_numpy(x32[:B]).cuda()
    for _ in range(5):
        op.run(xd)
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 100
    for _ in range(n):
        op.run(xd)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    out["device_resident_B%d" % B] = {"ms_per_call": round(dt * 1e3, 4), "frames_per_s": round(B / dt, 1)}
print(json.dumps(out))
