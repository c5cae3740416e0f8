"""numpy statement of the PARALLEL formulation the HIP kernels implement (test infrastructure).

The reference's transform is OpenCV's two raster sweeps.  dtfill.hip computes the same labels
without sweeping, from 1-D scans and a local rule (DESIGN.md "Why no raster sweep").  This module
spells that formulation out in numpy, one function per kernel, so that the mathematics can be
checked against the sequential oracle on the CPU, at full frame sizes, without a GPU
(tests/test_parallel_model.py).  It is not used by the product.
"""
import numpy as np

BIG = 1 << 20
FWD = [(-2, -1, 3), (-2, 1, 3), (-1, -2, 3), (-1, -1, 2), (-1, 0, 1), (-1, 1, 2), (-1, 2, 3), (0, -1, 1)]
BWD = [(2, 1, 3), (2, -1, 3), (1, 2, 3), (1, 1, 2), (1, 0, 1), (1, -1, 2), (1, -2, 3), (0, 1, 1)]


def colscan(src):
    """k_colscan: gu (distance to nearest source at-or-above), g = min(gu, gd)."""
    H, W = src.shape
    rows = np.arange(H)[:, None]
    last = np.maximum.accumulate(np.where(src, rows, -BIG), axis=0)
    gu = np.minimum(rows - last, BIG)
    nxt = np.minimum.accumulate(np.where(src, rows, BIG)[::-1], axis=0)[::-1]
    gd = np.minimum(nxt - rows, BIG)
    return gu, np.minimum(gu, gd)


def skew(gu):
    """k_skew: dB(i,j) = 3 + D(i-1,j+2), D(i,j) = min(E(i,j), 3 + D(i-1,j+2)) on columns 0..W."""
    H, W = gu.shape
    E = np.full((H, W + 3), BIG, np.int64)
    E[:, :W] = gu
    E[:, 1 : W + 1] = np.minimum(E[:, 1 : W + 1], np.where(gu < BIG, gu - 1, BIG))
    D = np.full((H, W + 3), BIG, np.int64)
    dB = np.full((H, W + 3), BIG, np.int64)
    for i in range(H):
        if i > 0:
            dB[i, : W + 1] = np.minimum(3 + D[i - 1, 2:], BIG)
        D[i] = np.minimum(E[i], dB[i])
    return dB[:, :W]


def rowscan(g, gu, dB):
    """k_rowscan: d (two-sided min-plus row scan of g), live = (dA == d) | (dB == d)."""
    H, W = g.shape
    idx = np.arange(W)
    a = np.minimum.accumulate(g - idx, axis=1) + idx
    b = np.minimum.accumulate((g + idx)[:, ::-1], axis=1)[:, ::-1] - idx
    d = np.minimum(a, b)
    dA = np.minimum.accumulate(gu - idx, axis=1) + idx
    none = d >= BIG // 2
    live = ((dA == d) | (dB == d)) & ~none
    return np.where(none, BIG, d), live


def parent(d, live):
    """k_parent: code 0..7 forward tap / 8..15 backward tap, 255 source, 254 none."""
    H, W = d.shape
    code = np.full((H, W), 254, np.int32)
    pad = 2
    dp = np.full((H + 4, W + 4), -BIG, np.int64)
    dp[pad : pad + H, pad : pad + W] = d
    lp = np.zeros((H + 4, W + 4), bool)
    lp[pad : pad + H, pad : pad + W] = live
    for base, taps, need_live in ((0, FWD, True), (8, BWD, False)):
        for t in range(7, -1, -1):  # descending: the first matching tap wins
            di, dj, w = taps[t]
            dn = dp[pad + di : pad + di + H, pad + dj : pad + dj + W]
            ln = lp[pad + di : pad + di + H, pad + dj : pad + dj + W]
            ok = dn + w == d
            if need_live:
                ok &= ln & live
            else:
                ok &= ~live
            code = np.where(ok, base + t, code)
    code = np.where(d == 0, 255, code)
    code = np.where(d >= BIG // 2, 254, code)
    return code


def resolve(code, src):
    """k_resolve: walk to the root; label = 1 + raster rank of the root source."""
    H, W = code.shape
    off_i = np.array([t[0] for t in FWD + BWD])
    off_j = np.array([t[1] for t in FWD + BWD])
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    ii, jj = ii.copy(), jj.copy()
    for _ in range(H + W):
        c = code[ii, jj]
        mv = c < 16
        if not mv.any():
            break
        cc = np.where(mv, c, 0)
        ii = np.where(mv, ii + off_i[cc], ii)
        jj = np.where(mv, jj + off_j[cc], jj)
    c = code[ii, jj]
    rank = np.cumsum(src.ravel()).reshape(H, W)
    label = np.where(c == 255, rank[ii, jj], 0).astype(np.int32)
    return label, ii * W + jj


def nearest_point(x, src_thr=0.1):
    """The whole parallel pipeline for one frame: (dt float32, lbl int32) as cv2 would return."""
    x = np.asarray(x, np.float32)
    src = ~((np.float32(1.0) - x) > np.float32(src_thr))
    gu, g = colscan(src)
    dB = skew(gu)
    d, live = rowscan(g, gu, dB)
    code = parent(d, live)
    label, _ = resolve(code, src)
    dt = np.where(d >= BIG // 2, np.float32(8192.0), d.astype(np.float32)).astype(np.float32)
    return dt, label


def nearest_point_levels(x, src_thr=0.1, max_level=None):
    """The LEVEL-SYNCHRONOUS form the fused kernels run, for one whole frame:
    E_t = pixels at distance t, L_t = the live ones; forward taps offer L_{t-w}, backward taps E_{t-w};
    first tap in cv2 order wins; roots are propagated level by level.  Returns (dt, lbl)."""
    x = np.asarray(x, np.float32)
    src = ~((np.float32(1.0) - x) > np.float32(src_thr))
    H, W = src.shape
    if not src.any():
        return np.full((H, W), 8192.0, np.float32), np.zeros((H, W), np.int32)

    def shift(P, di, dj):  # Q[q] = P[q + (di, dj)], zero outside
        Q = np.zeros_like(P)
        r0, r1 = max(0, -di), min(H, H - di)
        c0, c1 = max(0, -dj), min(W, W - dj)
        if r0 < r1 and c0 < c1:
            Q[r0:r1, c0:c1] = P[r0 + di : r1 + di, c0 + dj : c1 + dj]
        return Q

    Z = np.zeros((H, W), bool)
    D = src.copy()
    E, L = {0: src.copy()}, {0: src.copy()}
    rank = np.cumsum(src.ravel()).reshape(H, W)
    label = np.where(src, rank, 0).astype(np.int64)
    dist = np.zeros((H, W), np.int64)
    t = 0
    while not D.all():
        t += 1
        Et = (shift(D, 1, 0) | shift(D, -1, 0) | shift(D, 0, 1) | shift(D, 0, -1)) & ~D
        taken = Z.copy()
        new_label = label.copy()
        for di, dj, w in FWD:
            cand = shift(L.get(t - w, Z), di, dj) & Et
            sel = cand & ~taken
            taken |= cand
            new_label = np.where(sel, shift(label, di, dj), new_label)
        Lt = taken
        takenb = ~(Et & ~Lt)
        for di, dj, w in FWD:
            cand = shift(E.get(t - w, Z), -di, -dj)
            sel = cand & ~takenb
            takenb |= cand
            new_label = np.where(sel, shift(label, -di, -dj), new_label)
        label = new_label
        dist[Et] = t
        E[t], L[t] = Et, Lt
        D = D | Et
    return dist.astype(np.float32), label.astype(np.int32)
