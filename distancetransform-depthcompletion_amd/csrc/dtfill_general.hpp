// dtfill_general.hpp -- k_colscan, k_skew, k_rowscan, k_exit, k_final: full-frame scans, any distance
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ================================================================================================
// General path (any distance).  Every kernel returns at once for frames k_fused did not flag.
// ================================================================================================

constexpr int G_NCH = 16;  // row chunks (= waves per block) of the chunked column / knight-line scans

// k_colscan: 64 adjacent image columns per block, one wave per chunk of rows.
//   pass A: last / first source row of every column inside the chunk -> LDS
//   pass B: carry in the nearest source row above / below the chunk, then
//           down sweep: gu(i,j) = rows to the nearest source at or above (i,j);  up sweep: g = min(gu, gd).
// L2 = true (the `l2` metric): g carries in bit 15 whether that nearest source is BELOW the pixel (strictly
// nearer than the one above: on a vertical tie the upper source has the smaller raster index).
template <bool L2>
__global__ __launch_bounds__(64 * G_NCH) void k_colscan(const u64 *__restrict__ srcbits,
                                                        const int *__restrict__ fflag, const int *__restrict__ finfo,
                                                        int Hfull, int W, int Wd, u16 *__restrict__ gu,
                                                        u16 *__restrict__ g) {
    __shared__ int s_last[G_NCH][64], s_first[G_NCH][64];
    // ch is wave-uniform: as a scalar, the row loops and row addresses below run on the scalar unit
    const int b = blockIdx.y, wd = blockIdx.x, lane = threadIdx.x & 63, ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (fflag && !fflag[b]) return;
    // rows [0, H) of the frame are scanned: all of them, or -- band mode -- the top part the general kernels own
    // plus its margin (rows beyond count as source-free); the arrays keep the full-frame pitch
    const int H = finfo ? finfo[b * FI_STRIDE + FI_HG] : Hfull;
    const int CR = (H + G_NCH - 1) / G_NCH;
    const int j = wd * 64 + lane;
    const bool inb = j < W;
    const size_t fo = (size_t)b * Hfull * W;
    u16 *guf = gu + fo, *gf = g + fo;
    const u64 *sbf = srcbits + (size_t)b * Hfull * Wd + wd;
    const int i0 = min(ch * CR, H), i1 = min(i0 + CR, H);

    int last = -BIG, first = BIG;
    u32 bits = 0;  // fast path: the chunk's source bits of this lane's column
    if (CR <= 32) {
        // all of the chunk's words at once (wave-uniform addresses: 32 scalar loads in flight instead of one
        // load-and-wait per row)
#pragma unroll
        for (int kb = 0; kb < 32; kb += 16) {  // two batches of 16: 32 words at once would cost a block per CU in SGPRs
            u64 w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = sbf[(size_t)min(i0 + kb + k, max(i1 - 1, 0)) * Wd];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (i0 + kb + k < i1) bits |= (u32)((w[k] >> lane) & 1ull) << (kb + k);
        }
        if (bits) {
            last = i0 + 31 - __clz((int)bits);
            first = i0 + __ffs((int)bits) - 1;
        }
    } else {
        for (int ib = i0; ib < i1; ib += 16) {  // 16 rows' words per round trip (wave-uniform addresses: scalar loads)
            u64 w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = sbf[(size_t)min(ib + k, i1 - 1) * Wd];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const bool s = ib + k < i1 && ((w[k] >> lane) & 1ull);
                last = s ? ib + k : last;
                first = s ? min(first, ib + k) : first;
            }
        }
    }
    s_last[ch][lane] = last;
    s_first[ch][lane] = first;
    __syncthreads();
    int above = -BIG, below = BIG;
#pragma unroll
    for (int c = 0; c < G_NCH; ++c) {
        above = c < ch ? max(above, s_last[c][lane]) : above;
        below = c > ch ? min(below, s_first[c][lane]) : below;
    }
    int up = min(i0 - 1 - above, BIG);  // value "at row i0-1"
    int dn = min(below - i1, BIG);      // value "at row i1"
    if (CR <= 32) {
        // fast path (H <= 512): the chunk's source bits sit in one register, the from-below distances of its
        // rows in (statically indexed) registers; one store pass, nothing is re-read
        const int n = i1 - i0;
        int dnv[32];
#pragma unroll
        for (int k = 31; k >= 0; --k) {
            if (k < n) dn = (bits >> k) & 1u ? 0 : min(dn + 1, BIG);  // rows past the chunk end leave dn at "row i1"
            dnv[k] = dn;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k < n) {
                up = (bits >> k) & 1u ? 0 : min(up + 1, BIG);
                if (inb) {
                    const size_t o = (size_t)(i0 + k) * W + j;
                    guf[o] = st16(up);
                    if (L2) {
                        const int m = min(up, dnv[k]);
                        gf[o] = m >= 0x7FFF ? (u16)INF16 : (u16)(m | (dnv[k] < up ? 0x8000 : 0));
                    } else {
                        gf[o] = st16(min(up, dnv[k]));
                    }
                }
            }
        }
        return;
    }
    // any chunk height: the same two sweeps, 16 rows' words and (up sweep) 16 rows' gu values per round trip
    for (int ib = i0; ib < i1; ib += 16) {
        u64 w[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) w[k] = sbf[(size_t)min(ib + k, i1 - 1) * Wd];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = ib + k;
            if (i < i1) {  // wave-uniform
                const bool s = (w[k] >> lane) & 1ull;
                up = s ? 0 : min(up + 1, BIG);
                if (inb) guf[(size_t)i * W + j] = st16(up);
            }
        }
    }
    for (int it = i1; it > i0; it -= 16) {  // rows it-1 down to it-16
        u64 w[16];
        int uu[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = max(it - 1 - k, i0);
            w[k] = sbf[(size_t)i * Wd];
            uu[k] = inb ? ld16(guf + (size_t)i * W + j) : BIG;  // this lane's own stores of the down sweep
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int i = it - 1 - k;
            if (i < i0) continue;  // wave-uniform
            const bool s = (w[k] >> lane) & 1ull;
            dn = s ? 0 : min(dn + 1, BIG);
            if (inb) {
                const int u = uu[k];
                if (L2) {
                    const int m = min(u, dn);
                    gf[(size_t)i * W + j] = m >= 0x7FFF ? (u16)INF16 : (u16)(m | (dn < u ? 0x8000 : 0));
                } else {
                    gf[(size_t)i * W + j] = st16(min(u, dn));
                }
            }
        }
    }
}


// k_skew: knight lines u = j + 2 i over the extended column range j in [0, W] (column W is virtual:
// E(i,W) = gu(i,W-1) - 1).  64 adjacent lines per block, one wave per chunk of the rows those lines
// cross; all lanes of a wave sit in the same image row at each step, so the gu reads and dB writes
// of a step are contiguous.  D(i) = min(E(i), 3 + D(i-1)) is scanned per chunk from "infinity"
// (pass A), the true value at each chunk start follows from the chunk ends (LDS), and pass B rescans
// from it and stores dB = 3 + D(previous row).
// rows [c0, c1) of the lane's knight line.  8 rows at a time: their 16 loads are unconditional (clamped
// addresses, the predicates are applied afterwards with selects), so they are all in flight together.
// Returns D after row c1-1; if dBf, stores dB = 3 + D(previous row).
__device__ __forceinline__ int skew_run(const u16 *__restrict__ guf, u16 *__restrict__ dBf, int W, int nU, int u,
                                        int c0, int c1, int D) {
    const bool lane_on = u < nU;
    for (int ib = c0; ib < c1; ib += 8) {
        int ga[8], gb[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = min(ib + t, c1 - 1);
            const int j = u - 2 * i;
            const u16 *row = guf + (size_t)i * W;
            ga[t] = row[min(max(j, 0), W - 1)];
            gb[t] = row[min(max(j - 1, 0), W - 1)];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = ib + t;
            const int j = u - 2 * i;
            const bool on = lane_on && i < c1 && j >= 0 && j <= W;
            const int dbv = min(D + 3, BIG);  // 3 + D(i-1, j+2)
            const int a = ga[t] == INF16 ? BIG : ga[t], bq = gb[t] == INF16 ? BIG : gb[t] - 1;
            int e = (on && j < W) ? a : BIG;
            e = (on && j >= 1) ? min(e, bq) : e;
            if (dBf && on && j < W) dBf[(size_t)i * W + j] = st16(dbv);
            D = i < c1 ? (on ? min(e, dbv) : BIG) : D;
        }
    }
    return D;
}

__global__ __launch_bounds__(64 * G_NCH) void k_skew(const u16 *__restrict__ gu, const int *__restrict__ fflag,
                                                     const int *__restrict__ finfo, int Hfull, int W,
                                                     u16 *__restrict__ dB) {
    __shared__ int s_end[G_NCH][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar
    if (!fflag[b]) return;
    const int H = finfo[b * FI_STRIDE + FI_HG];  // rows scanned (band mode: fewer than the frame has)
    const int nU = W + 2 * (H - 1) + 1;
    const int u0 = blockIdx.x * 64;
    if (u0 >= nU) return;
    const int u = u0 + lane;
    const int u1 = min(u0 + 63, nU - 1);
    const size_t fo = (size_t)b * Hfull * W;
    const u16 *guf = gu + fo;
    u16 *dBf = dB + fo;

    const int i_lo = max(0, (u0 - W + 1) / 2);  // first row any lane of this block is inside [0, W]
    const int i_hi = min(H - 1, u1 / 2);
    const int CR = (i_hi - i_lo + 1 + G_NCH - 1) / G_NCH;
    const int c0 = min(i_lo + ch * CR, i_hi + 1), c1 = min(c0 + CR, i_hi + 1);
    int D = skew_run(guf, nullptr, W, nU, u, c0, c1, BIG);
    s_end[ch][lane] = D;
    __syncthreads();
    // D just before row c0: chain the chunk ends (a line is "on" for one contiguous row range, and an
    // "off" row resets D to BIG exactly as in the local scans)
    int K = BIG;
    for (int c = 0; c < ch; ++c) {
        const int len = min(i_lo + (c + 1) * CR, i_hi + 1) - min(i_lo + c * CR, i_hi + 1);
        K = min(s_end[c][lane], min(K + 3 * len, BIG));
        // a line that was off at the end of chunk c has s_end == BIG and, being contiguous, was never on
        // before: K + 3 len stays >= BIG only if K was BIG -- which holds, because any earlier on-rows
        // would make the line on at the end of chunk c as well (it leaves the image only at its last row)
    }
    skew_run(guf, dBf, W, nU, u, c0, c1, K);
}

// k_rowscan: one wave per image row, 8 consecutive pixels per lane (one 16-byte load per array), 512
// pixels per segment.  With a(j) = min_{k<=j} g(k) + (j-k) and dA likewise from gu (left-to-right), then
// d(j) = min_{k>=j} a(k) + (k-j) over a itself (right-to-left; a <= g and a(k)+(k-j) is a real path length,
// so this equals the two-sided minimum over g):  live = (dA == d) or (dB == d).
// Inside a lane the scans are sequential (8 steps); across lanes ONE wave scan of the lane totals per
// segment and quantity; across segments a wave-uniform carry.  The left-to-right results wait in a
// per-wave LDS row buffer.
__device__ __forceinline__ int wave_excl_prefix_min(int v, int lane) {  // min over lanes < lane (BIG for lane 0)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        if (lane >= off) v = min(v, t);
    }
    const int e = __shfl_up(v, 1);
    return lane == 0 ? BIG : e;
}
__device__ __forceinline__ int wave_excl_suffix_min(int v, int lane) {  // min over lanes > lane (BIG for lane 63)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_down(v, off);
        if (lane + off < 64) v = min(v, t);
    }
    const int e = __shfl_down(v, 1);
    return lane == 63 ? BIG : e;
}

// 8 consecutive uint16 of a row starting at element idx0 (multiple of 8); vectorised when the row is 16-byte aligned
__device__ __forceinline__ void load8(const u16 *__restrict__ row, int idx0, int W, bool vec, int (&v)[8]) {
    if (vec && idx0 + 8 <= W) {
        const uint4 q = *reinterpret_cast<const uint4 *>(row + idx0);
        v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
        v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = idx0 + q < W ? (int)row[idx0 + q] : INF16;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = v[q] == INF16 ? BIG : v[q];
}

// the same 8 values as one raw 16-byte load (rows 16-byte aligned, W % 8 == 0: a lane's 8 pixels are inside the row
// or outside together) -- kept raw so that the NEXT segment's load can be in flight while this one is worked on
__device__ __forceinline__ uint4 load8_raw(const u16 *__restrict__ row, int idx0, int W) {
    return idx0 < W ? *reinterpret_cast<const uint4 *>(row + idx0) : make_uint4(~0u, ~0u, ~0u, ~0u);
}
__device__ __forceinline__ void unpack8(const uint4 q, int (&v)[8]) {
    v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
    v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
#pragma unroll
    for (int q2 = 0; q2 < 8; ++q2) v[q2] = v[q2] == INF16 ? BIG : v[q2];
}

// Output: the float distance map (the last consumer of the full d) and four bit planes for k_exit, one byte
// per 8 pixels and plane: d & 1, d >> 1 & 1, d >> 2 & 1, live (plane p of row i starts at
// planes + p * plane_bytes + (b * H + i) * Wp; Wp = bytes per row, a multiple of 8).
__global__ __launch_bounds__(256) void k_rowscan(const u16 *__restrict__ g, const u16 *__restrict__ gu,
                                                 const u16 *__restrict__ dB, const int *__restrict__ fflag,
                                                 const int *__restrict__ finfo, int H, int W, int nseg, int Wp,
                                                 u8 *__restrict__ planes, size_t plane_bytes,
                                                 float *__restrict__ out_dt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // scalar row index
    const int i = blockIdx.x * (blockDim.x >> 6) + wave, b = blockIdx.y;
    if (!fflag[b] || i >= finfo[b * FI_STRIDE + FI_HG]) return;  // wave-uniform; no block-level barrier below
    if (i >= finfo[b * FI_STRIDE + FI_RW]) out_dt = nullptr;    // band mode: the margin rows' distances are the fused stage's to write
    int *s_a = reinterpret_cast<int *>(smem) + (size_t)wave * nseg * 512;  // a | (dA == a) << 24, lane-private slots
    const size_t ro = ((size_t)b * H + i) * W;
    const u16 *grow = g + ro, *gurow = gu + ro, *dBrow = dB + ro;
    const bool vec = (W & 7) == 0;  // rows start 16-byte aligned (the arrays are 256-byte aligned)

    int carry_a = BIG, carry_dA = BIG;  // min of (value - index) over everything left of the segment
    uint4 rg = make_uint4(0, 0, 0, 0), ru = rg;  // prefetched raw segment (vec rows only)
    if (vec) {
        rg = load8_raw(grow, lane * 8, W);
        ru = load8_raw(gurow, lane * 8, W);
    }
    for (int sg = 0; sg < nseg; ++sg) {
        const int idx0 = sg * 512 + lane * 8;
        int gv[8], uv[8];
        if (vec) {
            const uint4 cg = rg, cu = ru;
            if (sg + 1 < nseg) {  // next segment's loads fly during this segment's scans
                rg = load8_raw(grow, idx0 + 512, W);
                ru = load8_raw(gurow, idx0 + 512, W);
            }
            unpack8(cg, gv);
            unpack8(cu, uv);
        } else {
            load8(grow, idx0, W, vec, gv);
            load8(gurow, idx0, W, vec, uv);
        }
        int ma = BIG, md = BIG, la[8], ld[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ma = min(ma, gv[q] - (idx0 + q));
            md = min(md, uv[q] - (idx0 + q));
            la[q] = ma;
            ld[q] = md;
        }
        const int ea = min(wave_excl_prefix_min(ma, lane), carry_a);
        const int ed = min(wave_excl_prefix_min(md, lane), carry_dA);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int a = min(min(la[q], ea) + idx0 + q, BIG);
            const int dA = min(min(ld[q], ed) + idx0 + q, BIG);
            s_a[sg * 512 + q * 64 + lane] = a | (dA == a ? 1 << 24 : 0);  // [q][lane]: conflict-free, lane-private
        }
        carry_a = __shfl(min(ma, ea), 63);
        carry_dA = __shfl(min(md, ed), 63);
    }
    int carry_b = BIG;  // min of (a + index) over everything right of the segment
    uint4 rb = make_uint4(0, 0, 0, 0);
    if (vec) rb = load8_raw(dBrow, (nseg - 1) * 512 + lane * 8, W);
    for (int sg = nseg - 1; sg >= 0; --sg) {
        const int idx0 = sg * 512 + lane * 8;
        int av[8], fl[8], ms = BIG, ls[8];
#pragma unroll
        for (int q = 7; q >= 0; --q) {
            const int v = s_a[sg * 512 + q * 64 + lane];
            av[q] = v & 0xFFFFFF;
            fl[q] = v >> 24;
            ms = min(ms, av[q] + idx0 + q);
            ls[q] = ms;
        }
        const int es = min(wave_excl_suffix_min(ms, lane), carry_b);
        int dbv[8];
        if (vec) {
            const uint4 cb = rb;
            if (sg > 0) rb = load8_raw(dBrow, idx0 - 512, W);
            unpack8(cb, dbv);
        } else {
            load8(dBrow, idx0, W, vec, dbv);
        }
        u32 p0 = 0, p1 = 0, p2 = 0, pl = 0;
        float fd[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int d = min(ls[q], es) - (idx0 + q);
            const bool live = (fl[q] && av[q] == d) || dbv[q] == d;
            const bool inw = idx0 + q < W;
            p0 |= (inw ? (u32)d & 1u : 0u) << q;
            p1 |= (inw ? ((u32)d >> 1) & 1u : 0u) << q;
            p2 |= (inw ? ((u32)d >> 2) & 1u : 0u) << q;
            pl |= ((inw && live) ? 1u : 0u) << q;
            fd[q] = d >= DL_NONE ? 8192.0f : (float)d;  // no source in the frame: float(INIT_DIST0 * 2^-16) == 8192.0f
        }
        carry_b = __shfl(min(ms, es), 0);
        const int byte = idx0 >> 3;
        if (byte < Wp) {
            u8 *pb = planes + ((size_t)b * H + i) * Wp + byte;
            pb[0] = (u8)p0;
            pb[plane_bytes] = (u8)p1;
            pb[2 * plane_bytes] = (u8)p2;
            pb[3 * plane_bytes] = (u8)pl;
        }
        if (out_dt) {
            if (vec && (reinterpret_cast<size_t>(out_dt) & 15) == 0 && idx0 + 8 <= W) {  // W % 8 == 0: aligned float rows
                float4 *o = reinterpret_cast<float4 *>(out_dt + ro + idx0);
                o[0] = make_float4(fd[0], fd[1], fd[2], fd[3]);
                o[1] = make_float4(fd[4], fd[5], fd[6], fd[7]);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (idx0 + q < W) out_dt[ro + idx0 + q] = fd[q];
            }
        }
    }
}

constexpr int G_PPT = 16;  // pixels per thread in k_final (keeps its no-op grid small)

// k_exit: one block per 128 x 128 tile.  Loads the tile's bit planes (d mod 8, live -- from k_rowscan -- and
// the source bits) with a 2-cell halo into LDS, applies the 5x5 parent rule BIT-SLICED, 32 pixels per
// operation, and resolves the chains inside the tile by pointer doubling in LDS (every cell does the same work
// each round: no divergent walks, and the number of rounds is log of the longest in-tile chain, whatever
// the distances are).  A cell is terminal if it is a source, has no parent, or its parent lies outside the
// tile.  Result per pixel: an exit pointer
//   bit 31 set : the chain's root source, pixel index in the low bits
//   0x7FFFFFFF : no source in the frame
//   otherwise  : pixel index (another tile) where the chain continues
//
// Parent rule on d mod 8: a tap (di,dj) of weight w = |di|+|dj| <= 3 matches iff d(r) + w == d(q); d is the
// exact L1 distance, so |d(r) - d(q)| <= w and d(r) + w - d(q) lies in [0, 6]: it is 0 iff it is 0 mod 8.
constexpr int X_T = 128;             // tile edge
constexpr int X_NT = 1024;           // threads per block
constexpr int X_P = X_T + 4;         // rows of the plane tile (2-cell halo above and below)
constexpr int X_CP = X_T + 4;        // s_code row pitch in bytes: 33 dwords (odd) -> the un-slice stores spread over the banks
constexpr u32 X_ROOT = 0x80000000u;  // exit pointer: resolved to a root
constexpr u32 X_NONE = 0x7FFFFFFFu;  // exit pointer: frame without sources
constexpr int X_NPL = 6;             // planes in LDS: d bit 0, 1, 2, live, source, in-image
constexpr int X_RW = 7;              // words per plane row in LDS: image words c0/32 - 1 .. c0/32 + 4, + 1 pad (odd stride)

// bits of the pixels (column + DJ) of the word `cur` of a row (prev / next = the words left / right of it)
template <int DJ>
__device__ __forceinline__ u32 xshift(u32 prev, u32 cur, u32 next) {
    if (DJ == 0) return cur;
    if (DJ > 0) return __builtin_amdgcn_alignbit(next, cur, DJ);
    return __builtin_amdgcn_alignbit(cur, prev, 32 + DJ);
}

// One tap of the parent rule for 32 pixels.  The candidate r = q + (DI, DJ) has the planes row[...] (three
// words each: left, this, right).  FWD: r must be live.  First match wins: sel = match & ~taken.
template <int DJ, int WGT, bool FWD, int CODE>
__device__ __forceinline__ void exit_tap(const u32 (&a0)[3], const u32 (&a1)[3], const u32 (&a2)[3], const u32 (&lv)[3],
                                         const u32 (&vd)[3], u32 b0, u32 b1, u32 b2, u32 &taken, u32 (&T)[4]) {
    const u32 x0 = xshift<DJ>(a0[0], a0[1], a0[2]), x1 = xshift<DJ>(a1[0], a1[1], a1[2]),
              x2 = xshift<DJ>(a2[0], a2[1], a2[2]);
    u32 s0, s1, s2;  // (d(r) + WGT) mod 8
    if (WGT == 1) {
        s0 = ~x0; s1 = x1 ^ x0; s2 = x2 ^ (x1 & x0);
    } else if (WGT == 2) {
        s0 = x0; s1 = ~x1; s2 = x2 ^ x1;
    } else {
        s0 = ~x0; s1 = ~(x1 ^ x0); s2 = x2 ^ (x1 | x0);  // +1 then +2: carry into bit 2 iff x1 | x0
    }
    u32 m = ~((s0 ^ b0) | (s1 ^ b1) | (s2 ^ b2)) & xshift<DJ>(vd[0], vd[1], vd[2]);
    if (FWD) m &= xshift<DJ>(lv[0], lv[1], lv[2]);
    const u32 sel = m & ~taken;
    taken |= m;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (CODE & (1 << j)) T[j] |= sel;
}

__global__ __launch_bounds__(X_NT) void k_exit(const u8 *__restrict__ planes, size_t plane_bytes, int Wp,
                                              const u64 *__restrict__ srcbits, int Wd,
                                              const int *__restrict__ finfo, const int *__restrict__ fflag, int Hfull,
                                              int W, int tiles_x, u32 *__restrict__ exitp, int stop_after) {
    __shared__ __attribute__((aligned(16))) u16 s_big[X_T * X_T];  // first the planes (22 KB), later the pointers
    __shared__ __attribute__((aligned(16))) u8 s_code[X_T * X_CP];
    static_assert(X_NPL * X_P * X_RW * 4 <= X_T * X_T * 2, "the planes must fit under the pointers");
    const int b = blockIdx.y;
    if (!fflag[b]) return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * X_T, c0 = tx * X_T;
    const int H = finfo[b * FI_STRIDE + FI_HG];  // rows that exist for this pass (band mode: fewer than the frame has)
    if (r0 >= H) return;
    const size_t fo = (size_t)b * Hfull * W;
    const int tid = threadIdx.x;
    u32 *s_pl = reinterpret_cast<u32 *>(s_big);  // [plane][row][X_RW]
    const bool has_src = finfo[b * FI_STRIDE + FI_NSRC] != 0;

    // plane tile: rows r0-2 .. r0+129, image words c0/32 - 1 .. c0/32 + 4 (32-bit views of the byte / u64 arrays).
    // All of a thread's loads are issued before its first LDS store (one memory round trip, not five).
    {
        constexpr int NITEM = X_NPL * X_P * 6, PER = (NITEM + X_NT - 1) / X_NT;
        u32 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = tid + u * X_NT;
            const int pl = k / (X_P * 6), rem = k - pl * (X_P * 6);
            const int r = rem / 6, w = rem - r * 6;
            const int gi = r0 + r - 2, wi = (c0 >> 5) - 1 + w;
            const bool in = k < NITEM && gi >= 0 && gi < H && wi >= 0 && wi * 32 < W;
            const int gic = min(max(gi, 0), H - 1), wic = min(max(wi, 0), Wd * 2 - 1);  // clamped: unconditional loads
            const u32 *src = pl < 4 ? reinterpret_cast<const u32 *>(planes + min(pl, 3) * plane_bytes + ((size_t)b * Hfull + gic) * Wp)
                                    : reinterpret_cast<const u32 *>(srcbits + ((size_t)b * Hfull + gic) * Wd);
            const u32 ld = src[wic];
            const int up = min(W - wi * 32, 32);  // in-image columns of this word: [0, up)
            const u32 inimg = !in ? 0u : (up >= 32 ? 0xFFFFFFFFu : ((1u << up) - 1u));
            v[u] = pl == 5 ? inimg : (ld & inimg);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = tid + u * X_NT;
            if (k < NITEM) {
                const int pl = k / (X_P * 6), rem = k - pl * (X_P * 6);
                const int r = rem / 6, w = rem - r * 6;
                s_pl[(pl * X_P + r) * X_RW + w] = v[u];
            }
        }
    }
    __syncthreads();
    if (stop_after == 0) return;  // timing-only (DTFILL_EXIT_STOP)
    // parent rule: thread (row, word) handles 32 pixels; code = tap t (forward, live pixels) or 8 | t (the
    // negated tap, the others), the format tap_decode reads
    u32 T[4] = {0, 0, 0, 0}, found = 0, qsrc = 0, qin = 0;
    const int pr = tid >> 2, pw = tid & 3;  // tile row, tile word (tid < 512)
    if (tid < X_T * 4) {
        auto ld3 = [&](int pl, int row, u32 (&o)[3]) {  // words pw-1, pw, pw+1 of tile = LDS words pw, pw+1, pw+2
            const u32 *p = s_pl + (pl * X_P + row) * X_RW + pw;
            o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
        };
        const int qrow = pr + 2;
        const u32 *qp = s_pl + qrow * X_RW + pw + 1;
        const u32 b0 = qp[0], b1 = qp[X_P * X_RW], b2 = qp[2 * X_P * X_RW], qlive = qp[3 * X_P * X_RW];
        qsrc = qp[4 * X_P * X_RW];
        qin = qp[5 * X_P * X_RW];
        u32 takenF = ~(qin & ~qsrc & qlive), takenB = ~(qin & ~qsrc & ~qlive);
        u32 a0[3], a1[3], a2[3], lv[3], vd[3];
        // cv2 tap order: (-2,-1) (-2,+1) (-1,-2) (-1,-1) (-1,0) (-1,+1) (-1,+2) (0,-1); backward = the negated
        // offsets in the same order.  The two chains are independent (live / non-live pixels), each keeps its order.
        ld3(0, qrow - 2, a0); ld3(1, qrow - 2, a1); ld3(2, qrow - 2, a2); ld3(3, qrow - 2, lv); ld3(5, qrow - 2, vd);
        exit_tap<-1, 3, true, 0>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        exit_tap<+1, 3, true, 1>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        ld3(0, qrow - 1, a0); ld3(1, qrow - 1, a1); ld3(2, qrow - 1, a2); ld3(3, qrow - 1, lv); ld3(5, qrow - 1, vd);
        exit_tap<-2, 3, true, 2>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        exit_tap<-1, 2, true, 3>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        exit_tap<0, 1, true, 4>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        exit_tap<+1, 2, true, 5>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        exit_tap<+2, 3, true, 6>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, T);
        u32 z0[3], z1[3], z2[3], zv[3];  // this row: last forward tap now, last backward tap at the end
        ld3(0, qrow, z0); ld3(1, qrow, z1); ld3(2, qrow, z2); ld3(3, qrow, lv); ld3(5, qrow, zv);
        exit_tap<-1, 1, true, 7>(z0, z1, z2, lv, zv, b0, b1, b2, takenF, T);
        ld3(0, qrow + 2, a0); ld3(1, qrow + 2, a1); ld3(2, qrow + 2, a2); ld3(5, qrow + 2, vd);
        exit_tap<+1, 3, false, 8 | 0>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        exit_tap<-1, 3, false, 8 | 1>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        ld3(0, qrow + 1, a0); ld3(1, qrow + 1, a1); ld3(2, qrow + 1, a2); ld3(5, qrow + 1, vd);
        exit_tap<+2, 3, false, 8 | 2>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        exit_tap<+1, 2, false, 8 | 3>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        exit_tap<0, 1, false, 8 | 4>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        exit_tap<-1, 2, false, 8 | 5>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        exit_tap<-2, 3, false, 8 | 6>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, T);
        exit_tap<+1, 1, false, 8 | 7>(z0, z1, z2, lv, zv, b0, b1, b2, takenB, T);
        const u32 open = qin & ~qsrc;
        found = has_src ? ((takenF & open & qlive) | (takenB & open & ~qlive)) : 0u;
        if (!has_src) qsrc = T[0] = T[1] = T[2] = T[3] = 0;  // frame without sources: every cell PAR_NONE
    }
    __syncthreads();  // everybody has read the planes it needs (s_code is separate memory, written next)
    if (tid < X_T * 4) {
        // un-slice: 4 pixels per step; byte = tap code, PAR_NONE (0xFE) without a parent, PAR_SRC (0xFF) for a source
        u32 *crow = reinterpret_cast<u32 *>(s_code + pr * X_CP + pw * 32);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            u32 v = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) v |= (__umul24((T[j] >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u) << j;
            const u32 none = __umul24((~(found | qsrc) >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u;
            const u32 src = __umul24((qsrc >> (4 * q)) & 0xFu, 0x00204081u) & 0x01010101u;
            crow[q] = v | none * (u32)PAR_NONE | src * (u32)PAR_SRC;
        }
    }
    __syncthreads();
    if (stop_after == 1) return;
    u16 *s_ptr = s_big;  // the planes are dead
    for (int k = tid; k < X_T * X_T; k += X_NT) {
        const int r = k >> 7, c = k & (X_T - 1);
        const int code = s_code[r * X_CP + c];
        int di, dj;
        tap_decode(code, di, dj);
        const u32 nr = (u32)(r + di), nc = (u32)(c + dj);
        const bool inside = code < 16 && nr < (u32)X_T && nc < (u32)X_T;
        s_ptr[k] = inside ? (u16)(nr * X_T + nc) : (u16)(k | 0x8000);
    }
    __syncthreads();
    if (stop_after == 2) return;
    // pointer doubling.  Each thread owns cells tid + 256 j and keeps the still-open ones as bits, so late
    // rounds only touch what is left; two jumps per round.  Any pointer value read here is an ancestor of the
    // cell (other threads only ever replace a pointer by a farther ancestor): races just speed things up.
    {
        u64 open = 0;
#pragma unroll 8
        for (int j = 0; j < X_T * X_T / X_NT; ++j) open |= (u64)(!(s_ptr[tid + X_NT * j] & 0x8000)) << j;
        for (int round = 0; round < 16; ++round) {  // 4^16 > any in-tile chain
            u64 m = open;
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int k = tid + X_NT * j;
                int q = s_ptr[s_ptr[k]];  // s_ptr[k] has no flag: k is open
                if (!(q & 0x8000)) q = s_ptr[q];
                s_ptr[k] = (u16)q;
                if (q & 0x8000) open &= ~(1ull << j);
            }
            if (!__syncthreads_or(open != 0)) break;
        }
    }
    if (stop_after == 3) return;
    for (int k = tid; k < X_T * X_T; k += X_NT) {
        const int r = k >> 7, c = k & (X_T - 1);
        const int gi = r0 + r, gj = c0 + c;
        if (gi >= H || gj >= W) continue;
        const int t = s_ptr[k] & 0x3FFF;  // terminal cell of k's in-tile chain
        const int code = s_code[(t >> 7) * X_CP + (t & (X_T - 1))];
        const int tr = r0 + (t >> 7), tc = c0 + (t & (X_T - 1));
        u32 e;
        if (code == PAR_SRC) {
            e = X_ROOT | (u32)(tr * W + tc);
        } else if (code >= 16) {
            e = X_NONE;
        } else {
            int di, dj;
            tap_decode(code, di, dj);
            e = (u32)min(max((tr + di) * W + tc + dj, 0), H * W - 1);  // clamp: a logic error must not become a wild access
        }
        exitp[fo + (size_t)gi * W + gj] = e;
    }
}

// k_final: follow the exit pointers from tile to tile (a chain crosses few tiles), then label, gather, store.
__global__ __launch_bounds__(256) void k_final(
    const float *__restrict__ x, const u32 *__restrict__ exitp,
    const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
    const int *__restrict__ finfo, const float *__restrict__ vlist, const int *__restrict__ fflag, int H,
    int W, int Wd, float *__restrict__ out_depth, int32_t *__restrict__ out_index,
    int *__restrict__ frame_status) {
    const int b = blockIdx.y;
    if (!fflag[b]) return;
    const size_t fo = (size_t)b * H * W;
    const u32 *ef = exitp + fo;
    const int N1 = finfo[b * FI_STRIDE + FI_RW] * W;  // band mode: only the rows the general kernels own
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    constexpr int FB = 8;  // pixels per lane whose loads are in flight together
    for (int pb = blockIdx.x * (256 * G_PPT) + threadIdx.x; pb < min(N1, (int)(blockIdx.x + 1) * 256 * G_PPT);
         pb += 256 * FB) {
        u32 e[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) e[u] = ef[min(pb + 256 * u, N1 - 1)];
        for (int hop = 0; hop < MAX_HW_SUM; ++hop) {  // tile-to-tile hops, all FB chains in lock-step
            bool open = false;
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const bool mv = !(e[u] & X_ROOT) && e[u] != X_NONE;
                const u32 nx = ef[mv ? e[u] : 0u];
                e[u] = mv ? nx : e[u];
                open |= mv;
            }
            if (!__any(open)) break;
        }
        int label[FB], q[FB];
        u32 base[FB];
        u64 word[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const bool root = e[u] & X_ROOT;
            q[u] = root ? (int)(e[u] & ~X_ROOT) : 0;
            const int i = q[u] / W;
            const size_t w = ((size_t)b * H + i) * Wd + ((q[u] - i * W) >> 6);
            base[u] = rowbase_s[(size_t)b * H + i] + wpre_s[w];
            word[u] = srcbits[w];
        }
        float val[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = q[u] / W;
            label[u] = (e[u] & X_ROOT) ? source_rank(base[u], word[u], q[u] - i * W) : 0;
            const int p = pb + 256 * u;
            val[u] = (out_depth && p < N1) ? gather_depth(x + fo, vlist + fo, label[u], q[u], nval, misaligned, frame_status + b)
                                          : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int p = pb + 256 * u;
            if (p >= N1) continue;
            if (out_index) out_index[fo + p] = label[u];
            if (out_depth) out_depth[fo + p] = val[u];
        }
    }
}
