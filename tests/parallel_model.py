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


# ------------------------------------------------------------------------------------------------
# The ARGMIN-SCAN form (what the any-distance kernels k_colT / k_rows / k_ties run).
#   * every chain ends on a NEAREST source of its start pixel (each hop lowers d by exactly the hop's L1
#     length), so a pixel with exactly ONE nearest source has that source's label -- no chain needed;
#   * the row scans carry, next to the distance, the smallest and the largest column that achieves it
#     (packed keys, plain integer min): one nearest source <=> kmin == kmax and that column is not tied
#     between its nearest source above and below;
#   * live(q) <=> some nearest source lies in q's forward cone <=> the LEFTMOST nearest source among the
#     sources at-or-above q's row (distance dU == d) has 3 * (its column - q's column) <= 2 d
#     (no knight-line scan);
#   * only the remaining "tie" pixels apply the 5x5 parent rule and hop -- until they stand on a pixel
#     with one nearest source (1.0 - 1.6 hops on average on the bench workloads).
# ------------------------------------------------------------------------------------------------
GBIG = 16383  # column distance when a column holds no source (d >= 8192 <=> the frame has no source)
KOFF = 8192
KSH = 15      # key = value << 15 | arg; arg = column << 2 | flags (min keys) or (8191 - column) << 2 (max keys)


def colscan_flags(src):
    """k_colT + the first lines of k_rows: g, gu (capped at GBIG), 'nearest is below', 'above and below tie'."""
    H, W = src.shape
    rows = np.arange(H)[:, None]
    last = np.maximum.accumulate(np.where(src, rows, -BIG), axis=0)
    gu = np.minimum(rows - last, GBIG)
    nxt = np.minimum.accumulate(np.where(src, rows, BIG)[::-1], axis=0)[::-1]
    gd = np.minimum(nxt - rows, GBIG)
    g = np.minimum(gu, gd)
    below = gd < gu
    coltie = (gd == gu) & (g > 0) & (g < GBIG)
    return g, gu, below, coltie


def _two_sided(key_of_k):
    """min over k of (value(k) + |j - k|) on packed keys; key_of_k[..., k] = (value(k) + KOFF - k) << KSH | arg."""
    W = key_of_k.shape[-1]
    idx = np.arange(W, dtype=np.int64)
    a = np.minimum.accumulate(key_of_k, axis=-1) + (idx << KSH)           # value a(j) + KOFF, arg of the winner
    b = np.minimum.accumulate((a + (idx << KSH))[..., ::-1], axis=-1)[..., ::-1] - (idx << KSH)
    assert b.max() < (1 << 31)
    return b


def rowscan_argmin(g, gu, below, coltie):
    """k_rows: d, uniq, (si, sj) of the unique nearest source, live."""
    H, W = g.shape
    k = np.arange(W, dtype=np.int64)
    flags = below.astype(np.int64) | (coltie.astype(np.int64) << 1)
    kmin_key = ((g - k + KOFF) << KSH) | (k << 2) | flags
    kmax_key = ((g - k + KOFF) << KSH) | ((8191 - k) << 2)
    ku_key = ((gu - k + KOFF) << KSH) | (k << 2)
    r1, r2, r3 = _two_sided(kmin_key), _two_sided(kmax_key), _two_sided(ku_key)
    d = (r1 >> KSH) - KOFF
    assert np.array_equal(d, (r2 >> KSH) - KOFF)
    kmin = (r1 >> 2) & 8191
    fl = r1 & 3
    kmax = 8191 - ((r2 >> 2) & 8191)
    dU = (r3 >> KSH) - KOFF
    kminU = (r3 >> 2) & 8191
    none = d >= 8192
    uniq = (kmin == kmax) & ((fl & 2) == 0) & ~none
    gk = d - np.abs(k[None, :] - kmin)
    ii = np.arange(H, dtype=np.int64)[:, None]
    si = np.where(fl & 1, ii + gk, ii - gk)
    live = (dU == d) & (3 * (kminU - k[None, :]) <= 2 * d) & ~none
    return np.where(none, BIG, d), uniq, si, kmin, live


def nearest_point_argmin(x, src_thr=0.1, return_stats=False):
    """The argmin-scan pipeline for one frame: (dt float32, lbl int32) as cv2 would return."""
    x = np.asarray(x, np.float32)
    src = ~((np.float32(1.0) - x) > np.float32(src_thr))
    H, W = src.shape
    g, gu, below, coltie = colscan_flags(src)
    d, uniq, si, sj, live = rowscan_argmin(g, gu, below, coltie)
    rank = np.cumsum(src.ravel()).reshape(H, W)
    label = np.zeros((H, W), np.int64)
    ok = uniq | src
    assert src[si[ok], sj[ok]].all()
    label[ok] = rank[si[ok], sj[ok]]
    code = parent(d, live)
    off_i = np.array([t[0] for t in FWD + BWD])
    off_j = np.array([t[1] for t in FWD + BWD])
    ti, tj = np.nonzero(~ok & (d < BIG // 2))
    ci, cj = ti.copy(), tj.copy()
    hops = 0
    open_ = np.ones(len(ti), bool)
    while open_.any():
        c = code[ci[open_], cj[open_]]
        assert (c < 16).all()
        ci[open_] += off_i[c]
        cj[open_] += off_j[c]
        open_[open_] = ~ok[ci[open_], cj[open_]]
        hops += 1
    label[ti, tj] = label[ci, cj]
    dt = np.where(d >= BIG // 2, np.float32(8192.0), d.astype(np.float32)).astype(np.float32)
    if return_stats:
        return dt, label.astype(np.int32), dict(ties=len(ti), hops=hops)
    return dt, label.astype(np.int32)


# ---- l2 mode, dense frames: the window formulation of k_l2win (dtfill_l2.hpp) --------------------------------

def l2_window(src, R):
    """Exact squared Euclidean distance and nearest source (smallest source row, then column) for every pixel that has a
    source within distance R, from the separable form d2 = min_dy dy^2 + hx(i + dy, j)^2 over the (2R+1) rows around the
    pixel, hx = horizontal distance to the row's nearest source, left preferred on a tie and capped at R.  The packed key
    (d2, row, side) is the one the kernel minimises.  Returns d2, near (raster index of the source), decided (bool)."""
    src = np.asarray(src, bool)
    H, W = src.shape
    jj = np.arange(W)
    big = 1 << 20
    # nearest source at or left / at or right of every pixel, per row
    left = np.where(src, jj[None, :], -big)
    left = np.maximum.accumulate(left, axis=1)
    right = np.where(src, jj[None, :], big)
    right = np.minimum.accumulate(right[:, ::-1], axis=1)[:, ::-1]
    dxl, dxr = jj[None, :] - left, right - jj[None, :]
    dx = np.minimum(dxl, dxr)
    side = (dxr < dxl).astype(np.int64)
    none = (R + 1) * (R + 1)
    e = np.where(dx <= R, dx * dx, none).astype(np.int64) << 6 | np.where(dx <= R, side, 0)
    ep = np.full((H + 2 * R, W), none << 6, np.int64)  # rows outside the frame hold nothing
    ep[R:R + H] = e
    best = np.full((H, W), np.iinfo(np.int64).max)
    for dyi in range(2 * R + 1):
        dy = dyi - R
        best = np.minimum(best, ep[dyi:dyi + H] + ((dy * dy) << 6 | dyi << 1))
    d2 = best >> 6
    decided = d2 <= R * R
    dy = ((best >> 1) & 31) - R
    dxw = np.sqrt(np.maximum(d2 - dy * dy, 0)).round().astype(np.int64)
    sc = np.where(best & 1, jj[None, :] + dxw, jj[None, :] - dxw)
    sr = np.arange(H)[:, None] + dy
    return d2, sr * W + sc, decided


# ---- l2 mode, any distance: the lower-envelope search of k_l2env (dtfill_l2.hpp) -----------------------------

def l2_envelope(src):
    """Exact squared Euclidean distance and nearest source (smallest source row, then column) by the two-pass scheme:
    vertical distance g per column (upper source on a tie), then per row the owner of every pixel among the parabolas
    g(k)^2 + (j-k)^2 -- found level by level: pixel 0 and W-1 over all columns, then midpoints, each between the owners of
    its solved neighbours (the owner never moves left as j moves right).  Returns d2, near, evaluations per pixel."""
    src = np.asarray(src, bool)
    H, W = src.shape
    big = 1 << 20
    ii = np.arange(H)[:, None]
    up = np.maximum.accumulate(np.where(src, ii, -big), axis=0)
    dn = np.minimum.accumulate(np.where(src, ii, big)[::-1], axis=0)[::-1]
    gu, gd = ii - up, dn - ii
    g = np.minimum(gu, gd)
    srow = np.where(gd < gu, dn, up)
    d2 = np.zeros((H, W), np.int64)
    near = np.zeros((H, W), np.int64)
    evals = 0
    for i in range(H):
        cols = np.nonzero(g[i] < big // 2)[0]
        cand = [(int(g[i, k]) ** 2, int(srow[i, k]), int(k)) for k in cols]
        own = {}

        def query(j, lo, hi):
            nonlocal evals
            evals += hi - lo + 1
            best = min(range(lo, hi + 1), key=lambda c: (cand[c][0] + (j - cand[c][2]) ** 2, cand[c][1], cand[c][2]))
            own[j] = best

        query(0, 0, len(cand) - 1)
        if W > 1:
            query(W - 1, 0, len(cand) - 1)
        s = 1
        while s < W - 1:
            s <<= 1
        s >>= 1
        while s >= 1:
            nq = ((W - 2) // s + 1) // 2 if W - 2 >= s else 0
            for m in range(nq):
                j = (2 * m + 1) * s
                query(j, own[j - s], own[min(j + s, W - 1)])
            s >>= 1
        for j in range(W):
            g2, sr, k = cand[own[j]]
            d2[i, j] = g2 + (j - k) ** 2
            near[i, j] = sr * W + k
    return d2, near, evals / (H * W)


def sky_rows(dt, lbl):
    """k_sky: the rows above the first source row r0, from rows r0 and r0 + 1 of a finished (dt, lbl) pair alone.
    Returns (dt, lbl) with rows [0, r0) recomputed (the input's are ignored), or the input unchanged when the frame has no
    sky (a source in row 0, or no source at all).  The statement of dtfill_sky.hpp: a step code per column for the rows
    up to r0 - 2, the five leading backward taps on the real distances for row r0 - 1."""
    H, W = dt.shape
    rows = np.flatnonzero((dt == 0).any(axis=1))
    if rows.size == 0 or rows[0] == 0:
        return dt, lbl
    r0 = int(rows[0])
    f = dt[r0].astype(np.int64)
    two = r0 + 1 < H
    g = dt[r0 + 1].astype(np.int64) if two else None
    dt, lbl = dt.copy(), lbl.copy()
    base = np.concatenate([lbl[r0], lbl[r0 + 1] if two else lbl[r0]])
    j = np.arange(W)
    key = {r0: j.copy()}
    # row r0 - 1
    D = f + 1
    k = j.copy()
    taken = np.zeros(W, bool)

    def offer(cond, val):
        nonlocal k, taken
        sel = cond & ~taken
        k = np.where(sel, val, k)
        taken |= cond

    pad = lambda a: np.concatenate([np.full(2, BIG), a, np.full(2, BIG)])
    sh = lambda a, s: pad(a)[2 + s : 2 + s + W]  # a[j + s], BIG outside the row
    if two:
        offer(sh(g, 1) + 3 == D, W + j + 1)
        offer(sh(g, -1) + 3 == D, W + j - 1)
    offer(sh(f, 2) + 3 == D, j + 2)
    offer(sh(f, 1) + 2 == D, j + 1)
    key[r0 - 1] = k
    step = np.where(sh(f, 1) + 1 == f, 1, np.where(sh(f, -1) + 1 == f, 2, 0))
    for i in range(r0 - 2, -1, -1):
        k2, k1 = key[i + 2], key[i + 1]
        key[i] = np.where(step == 1, np.concatenate([k2, [0]])[1 : W + 1], np.where(step == 2, np.concatenate([[0], k2])[:W], k1))
    for i in range(r0):
        dt[i] = (f + (r0 - i)).astype(dt.dtype)
        lbl[i] = base[key[i]]
    return dt, lbl


def sky_rows_closed_form(dt, lbl):
    """k_sky as the kernel computes it: no row-by-row propagation.  The step code of a column is the same in every row up to
    r0 - 2, a (+2,+1) column is followed by (+2,+1) columns up to a (+1,0) column (mirror image for (+2,-1)), and a (+1,0)
    column repeats row r0 - 1 all the way up.  So pixel (i, j) takes t = min(run length of its column, floor((r0 - i) / 2))
    hops of (+2, +-1) and lands in row i + 2 t: in row r0 on the base pixel itself, in any row above on what row r0 - 1 holds
    in that column."""
    H, W = dt.shape
    rows = np.flatnonzero((dt == 0).any(axis=1))
    if rows.size == 0 or rows[0] == 0:
        return dt, lbl
    r0 = int(rows[0])
    f = dt[r0].astype(np.int64)
    first_dt, first_lbl = sky_rows(dt, lbl)  # row r0 - 1 by the five taps (the model above), the rest is replaced below
    first = first_lbl[r0 - 1]
    pad = lambda a: np.concatenate([np.full(2, BIG), a, np.full(2, BIG)])
    sh = lambda a, s: pad(a)[2 + s : 2 + s + W]
    A = sh(f, 1) + 1 == f
    Bm = ~A & (sh(f, -1) + 1 == f)
    run = np.zeros(W, np.int64)  # consecutive A columns from j rightwards / B columns leftwards
    for j in range(W - 1, -1, -1):
        if A[j]:
            run[j] = 1 + (run[j + 1] if A[j + 1] else 0)
    for j in range(W):
        if Bm[j]:
            run[j] = 1 + (run[j - 1] if Bm[j - 1] else 0)
    direction = np.where(A, 1, np.where(Bm, -1, 0))
    out_dt, out_lbl = dt.copy(), lbl.copy()
    j = np.arange(W)
    for i in range(r0):
        t = np.zeros(W, np.int64) if i == r0 - 1 else np.minimum(run, (r0 - i) // 2)
        col = j + direction * t
        out_lbl[i] = np.where(i + 2 * t == r0, lbl[r0][col], first[col])
        out_dt[i] = (f + (r0 - i)).astype(dt.dtype)
    return out_dt, out_lbl
