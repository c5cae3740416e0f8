import importlib, sys, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module('distancetransform-depthcompletion_amd')
from oracle import oracle as O
op = pkg.device.DtFill(device='cuda:0', metric='l2')
x = np.zeros((1, 9, 40), np.float32); x[0, 3, 7] = 2.0; x[0, 6, 30] = 3.0
res = op.run(torch.from_numpy(x).to('cuda:0'), want=("dt", "index"))
torch.cuda.synchronize()
depth, dt, idx, status = O.fill_batch(x, metric='l2')
g = res['dt'].cpu().numpy()[0]; gi = res['index'].cpu().numpy()[0]
np.set_printoptions(linewidth=250, precision=1, suppress=True)
print('oracle d2 row0', (dt[0, 0] ** 2).round().astype(int)[:40])
print('gpu    d2 row0', (g[0] ** 2).round()[:40])
print('oracle idx row0', idx[0, 0][:40]); print('gpu idx row0', gi[0][:40])
print('oracle d2 row8', (dt[0, 8] ** 2).round().astype(int)[:40])
print('gpu    d2 row8', (g[8] ** 2).round()[:40])
op1 = pkg.device.DtFill(device='cuda:0', metric='l1_cv')
r1 = op1.run(torch.from_numpy(x).to('cuda:0'), path='general')
torch.cuda.synchronize()
d1, t1, i1, s1 = O.fill_batch(x)
print('l1 general ok', np.array_equal(r1['dt'].cpu().numpy(), t1), np.array_equal(r1['index'].cpu().numpy(), i1))
# peek at the workspace of the l2 operator: srcbits words of row 3 and the ct entries of band 0
ws = op._ws[op._ws_off:]
B, H, W = x.shape; Wd = 1
src = ws[:B*H*Wd*8].view(torch.int64).cpu().numpy()
print('srcbits rows', [hex(int(v) & 0xFFFFFFFFFFFFFFFF) for v in src])
def al(n): return (n + 255) & ~255
NW = B*H*Wd; NR = B*H
off = al(NW*8)*2 + al(NW*2)*2 + al(NR*4)*4 + al(B*8*4) + al(B*4)*2
ctp = ((W+63)//64)*64 + 640
ct = ws[off:off + 1*ctp*8].view(torch.int32).cpu().numpy().reshape(-1, 2)
print('ct[0:12]', ct[:12].tolist())
print('ct[28:32]', ct[28:32].tolist())
