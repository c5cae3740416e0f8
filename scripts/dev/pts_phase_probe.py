"""Development probe: where a k_pts tile's cycles go (needs a libdtfill.so built with -DPTS_PROF:
   make -C distancetransform-depthcompletion_amd/csrc -B HIPFLAGS='-O3 -std=c++17 --offload-arch=gfx950 -fPIC -DPTS_PROF').
Prints, per phase, the cycles summed over all waves and their share."""
import ctypes, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
pkg = importlib.import_module("distancetransform-depthcompletion_amd")
L = pkg._lib.load()
L.dtfill_pts_prof.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
synth = importlib.import_module("distancetransform-depthcompletion_amd.synth")
x = torch.from_numpy(synth.make(sys.argv[1] if len(sys.argv) > 1 else "nyu_b64")).cuda()
op = pkg.device.DtFill(device="cuda:0")
for _ in range(20): op.run(x)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 12)()
L.dtfill_pts_prof(buf, 1)
n = 50
for _ in range(n): op.run(x)
torch.cuda.synchronize()
L.dtfill_pts_prof(buf, 1)
names = ["candidates", "minima+stores", "planes", "barrier", "rule", "hops", "tail"]
tot = sum(buf[:7])
for k, nm in enumerate(names):
    print("%-14s %12d cycles/pass  %5.1f %%" % (nm, buf[k] // n, 100.0 * buf[k] / max(tot, 1)))
w = max(buf[8], 1)
print("waves/pass %d  candidates per wave: mean %.2f  rms %.2f  max %d" % (buf[8] // n, buf[9] / w, (buf[10] / w) ** 0.5, buf[11]))
