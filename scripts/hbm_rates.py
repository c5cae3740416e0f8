"""Achievable HBM rates on this GPU (fill = write only, copy = read + write, sum = read only), for reading the
roofline fractions in DESIGN.md against something measured rather than the 8 TB/s datasheet figure."""
import torch

def rate(fn, nbytes, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    return nbytes / ms / 1e6, ms

for mb in (55, 166, 512, 2048):
    n = mb * 1000 * 1000 // 4
    a = torch.empty(n, device="cuda", dtype=torch.float32)
    b = torch.empty(n, device="cuda", dtype=torch.float32)
    w, wms = rate(lambda: a.fill_(1.0), n * 4)
    c, cms = rate(lambda: b.copy_(a), 2 * n * 4)
    r, rms = rate(lambda: a.sum(), n * 4)
    print(f"{mb:5d} MB  fill {w:7.0f} GB/s ({wms*1e3:6.1f} us)  copy {c:7.0f} GB/s ({cms*1e3:6.1f} us)  sum {r:7.0f} GB/s ({rms*1e3:6.1f} us)")
