// dtfill_rows.hpp -- k_colT, k_rows, k_ties, k_tiesx: the any-distance path of the l1_cv pass (argmin scans)
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ================================================================================================
// Any distance, any density.  Rests on three facts (numpy statement: tests/parallel_model.py,
// nearest_point_argmin; checked bit-for-bit against the sequential oracle):
//   1. every hop of a parent chain lowers d by exactly the hop's L1 length, so a chain ends on a NEAREST source
//      of its start pixel: a pixel with exactly one nearest source has that source's label -- no chain;
//   2. the two-sided min-plus row scan can carry the column of its winner in the low bits of the key
//      (plain integer min): with one scan that prefers the smallest and one that prefers the largest
//      column, "one nearest source" <=> kmin == kmax and that column is not tied above / below;
//   3. live(q) <=> some nearest source lies in q's forward cone <=> the LEFTMOST nearest source among those
//      at-or-above q's row (the scan over gu finds it) has 3 (col - q.col) <= 2 d.  No knight-line scan.
// Only "tie" pixels (3 % - 22 % on the bench workloads) apply the 5x5 parent rule, and they hop only
// until they stand on a pixel with one nearest source (1.0 - 1.6 hops on average).
//
//   k_colT   srcbits -> per 32-row band and column: the band's source bits of that column (one word) and
//            the distance from the band's first / last row to the nearest source above / below the band.
//            0.25 B/px; everything a row needs to know about its columns.
//   k_rows   one block per image row: column distances from the band words, six packed-key scans
//            (kmin / kmax / upper-sources, left and right), d, the nearest source, label, depth gather,
//            the three output stores; five bit planes (d mod 8, live, tie) for k_ties.
//   k_ties   one block per 64 x 128 tile: bit-sliced 5x5 parent rule on the planes, tie pixels hop through
//            the tile's window in LDS and copy label + depth of the pixel they end on.
//   k_tiesx  the few tie pixels whose hops left the window: the same rule, evaluated per hop from the
//            planes in global memory (any chain length).
// ================================================================================================

constexpr int GBIG = 16383;  // column distance when the column has no source (d >= 8192 <=> frame without sources)
constexpr int K_OFF = 8192;  // left keys carry value - column + K_OFF
constexpr int K_SH = 15;     // key = value << 15 | arg; arg = column << 2 | flags, or (8191 - column) << 2
constexpr u32 K_IDENT = 0x3FFFFFFFu;  // larger than any real key, small enough to survive the +- (column << 15)
constexpr int PL_D0 = 0, PL_D1 = 1, PL_D2 = 2, PL_LIVE = 3, PL_TIE = 4, PL_UNRES = 5, PL_N = 6;

// ------------------------------------------------------------------------------------------------
// k_colT: 64 adjacent columns per block, one wave per 32-row band (bands beyond the block's waves: loop).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_colT(const u64 *__restrict__ srcbits, const int *__restrict__ fflag, int H,
                                               int W, int Wd, int nb, u32 *__restrict__ ctT, u16 *__restrict__ ctU,
                                               u16 *__restrict__ ctD) {
    extern __shared__ u16 s_lf[];  // [nb][64] last source row of the band, [nb][64] first (0xFFFF: none)
    const int b = blockIdx.y, wd = blockIdx.x, lane = threadIdx.x & 63;
    const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    if (fflag && !fflag[b]) return;
    const int j = wd * 64 + lane;
    const u64 *sbf = srcbits + (size_t)b * H * Wd + wd;
    u16 *s_last = s_lf, *s_first = s_lf + nb * 64;
    for (int band = ch; band < nb; band += nwv) {
        const int i0 = band * 32, i1 = min(i0 + 32, H);
        u32 bits = 0;
#pragma unroll
        for (int kb = 0; kb < 32; kb += 16) {  // wave-uniform addresses: scalar loads, 16 in flight
            u64 w[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) w[k] = sbf[(size_t)min(i0 + kb + k, i1 - 1) * Wd];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (i0 + kb + k < i1) bits |= (u32)((w[k] >> lane) & 1ull) << (kb + k);
        }
        if (j < W) ctT[((size_t)b * nb + band) * W + j] = bits;
        s_last[band * 64 + lane] = bits ? (u16)(i0 + 31 - __clz((int)bits)) : (u16)0xFFFF;
        s_first[band * 64 + lane] = bits ? (u16)(i0 + __ffs((int)bits) - 1) : (u16)0xFFFF;
    }
    __syncthreads();
    if (ch == 0) {  // distance from the band's FIRST row to the nearest source above the band
        int run = -1;
        for (int band = 0; band < nb; ++band) {
            const int up = run < 0 ? GBIG : min(band * 32 - run, GBIG);
            if (j < W) ctU[((size_t)b * nb + band) * W + j] = (u16)up;
            const int l = s_last[band * 64 + lane];
            run = l != 0xFFFF ? l : run;
        }
    } else if (ch == 1) {  // distance from the band's LAST row to the nearest source below the band
        int run = -1;
        for (int band = nb - 1; band >= 0; --band) {
            const int dn = run < 0 ? GBIG : min(run - (band * 32 + 31), GBIG);
            if (j < W) ctD[((size_t)b * nb + band) * W + j] = (u16)dn;
            const int f = s_first[band * 64 + lane];
            run = f != 0xFFFF ? f : run;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// wave scans of packed keys (unsigned min) on the DPP network: row_shr / row_shl inside a row of 16 lanes,
// row_bcast (prefix) or three readlanes (suffix) across the rows, wave_shr / wave_shl for the exclusive shift
// ------------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ u32 dpp_min(u32 v) {  // lanes the pattern does not feed keep v
    return min(v, (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROWMASK, 0xF, false));
}
// excl = min over the lanes before this one (K_IDENT for lane 0); total = min over the wave (uniform)
__device__ __forceinline__ void wave_prefix_min(u32 v, u32 &excl, u32 &total) {
    v = dpp_min<0x111, 0xF>(v);
    v = dpp_min<0x112, 0xF>(v);
    v = dpp_min<0x114, 0xF>(v);
    v = dpp_min<0x118, 0xF>(v);
    v = dpp_min<0x142, 0xA>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_min<0x143, 0xC>(v);  // row_bcast:31 into rows 2 and 3
    total = (u32)__builtin_amdgcn_readlane((int)v, 63);
    excl = (u32)__builtin_amdgcn_update_dpp((int)K_IDENT, (int)v, 0x138, 0xF, 0xF, false);  // wave_shr:1
}
// excl = min over the lanes after this one (K_IDENT for lane 63)
__device__ __forceinline__ void wave_suffix_min(u32 v, int lane, u32 &excl, u32 &total) {
    v = dpp_min<0x101, 0xF>(v);
    v = dpp_min<0x102, 0xF>(v);
    v = dpp_min<0x104, 0xF>(v);
    v = dpp_min<0x108, 0xF>(v);
    const u32 t1 = (u32)__builtin_amdgcn_readlane((int)v, 16), t2 = (u32)__builtin_amdgcn_readlane((int)v, 32),
              t3 = (u32)__builtin_amdgcn_readlane((int)v, 48);
    const u32 s23 = min(t2, t3), s123 = min(t1, s23);
    const int row = lane >> 4;
    v = min(v, row == 0 ? s123 : row == 1 ? s23 : row == 2 ? t3 : K_IDENT);
    total = (u32)__builtin_amdgcn_readlane((int)v, 0);
    excl = (u32)__builtin_amdgcn_update_dpp((int)K_IDENT, (int)v, 0x130, 0xF, 0xF, false);  // wave_shl:1
}

// ------------------------------------------------------------------------------------------------
// k_rows: one block per image row, one wave per 512 columns, 8 consecutive columns per lane.
// VEC: W % 8 == 0 (16-byte loads of the band words / carries, 16-byte stores of the outputs if they are aligned)
// ------------------------------------------------------------------------------------------------
constexpr int R_MAXWV = 16;  // W <= 8191 -> at most 16 waves of 512 columns

template <bool VEC>
__global__ __launch_bounds__(1024) void k_rows(
    const float *__restrict__ x, const u32 *__restrict__ ctT, const u16 *__restrict__ ctU, const u16 *__restrict__ ctD,
    const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
    const int *__restrict__ finfo, const float *__restrict__ vlist, const int *__restrict__ fflag, int H, int W, int Wd,
    int nb, int Wp, u8 *__restrict__ planes, size_t plane_bytes, float *__restrict__ out_depth,
    float *__restrict__ out_dt, int32_t *__restrict__ out_index, int *__restrict__ frame_status, int ovec) {
    __shared__ u32 s_tot[R_MAXWV][6];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    const int i = blockIdx.x, b = blockIdx.y;
    if (!fflag[b]) return;  // block-uniform
    const int band = i >> 5, r = i & 31;
    const int idx0 = wv * 512 + lane * 8;
    const size_t cbase = ((size_t)b * nb + band) * W;

    // ---- the band words and carries of this lane's 8 columns -> column distances
    u32 T[8];
    int U[8], Dn[8];
    if (VEC) {
        if (idx0 < W) {  // W % 8 == 0: a lane's 8 columns are inside or outside together
            const uint4 t0 = *reinterpret_cast<const uint4 *>(ctT + cbase + idx0);
            const uint4 t1 = *reinterpret_cast<const uint4 *>(ctT + cbase + idx0 + 4);
            const uint4 u = *reinterpret_cast<const uint4 *>(ctU + cbase + idx0);
            const uint4 dd = *reinterpret_cast<const uint4 *>(ctD + cbase + idx0);
            T[0] = t0.x; T[1] = t0.y; T[2] = t0.z; T[3] = t0.w; T[4] = t1.x; T[5] = t1.y; T[6] = t1.z; T[7] = t1.w;
            U[0] = u.x & 0xFFFF; U[1] = u.x >> 16; U[2] = u.y & 0xFFFF; U[3] = u.y >> 16;
            U[4] = u.z & 0xFFFF; U[5] = u.z >> 16; U[6] = u.w & 0xFFFF; U[7] = u.w >> 16;
            Dn[0] = dd.x & 0xFFFF; Dn[1] = dd.x >> 16; Dn[2] = dd.y & 0xFFFF; Dn[3] = dd.y >> 16;
            Dn[4] = dd.z & 0xFFFF; Dn[5] = dd.z >> 16; Dn[6] = dd.w & 0xFFFF; Dn[7] = dd.w >> 16;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                T[q] = 0;
                U[q] = Dn[q] = GBIG;
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const bool in = idx0 + q < W;
            T[q] = in ? ctT[cbase + idx0 + q] : 0u;
            U[q] = in ? (int)ctU[cbase + idx0 + q] : GBIG;
            Dn[q] = in ? (int)ctD[cbase + idx0 + q] : GBIG;
        }
    }
    // six keys per column: {all sources, prefer the smallest column | all, prefer the largest | sources at or above
    // this row} x {left scan (value - column), right scan (value + column)}
    u32 Lmin[8], Lmax[8], LU[8], Rmin[8], Rmax[8], RU[8];
    {
        const u32 upmask = (2u << r) - 1u;  // band rows 0..r (r = 31: all)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const u32 ma = T[q] & upmask, mb = T[q] >> r;
            const int gu = ma ? r - (31 - __clz((int)ma)) : min(r + U[q], GBIG);
            const int gd = mb ? __ffs((int)mb) - 1 : min(31 - r + Dn[q], GBIG);
            const int g = min(gu, gd);
            // flags of the column: bit 0 = its nearest source is below this row, bit 1 = above and below tie
            const u32 fl = (gd < gu ? 1u : 0u) | ((gd == gu && g > 0 && g < GBIG) ? 2u : 0u);
            const int k = idx0 + q;
            const u32 amin = (u32)k << 2 | fl, amax = (u32)(8191 - k) << 2, au = (u32)k << 2;
            Lmin[q] = (u32)(g - k + K_OFF) << K_SH | amin;
            Lmax[q] = (u32)(g - k + K_OFF) << K_SH | amax;
            LU[q] = (u32)(gu - k + K_OFF) << K_SH | au;
            Rmin[q] = (u32)(g + k) << K_SH | amin;
            Rmax[q] = (u32)(g + k) << K_SH | amax;
            RU[q] = (u32)(gu + k) << K_SH | au;
        }
    }
    // inclusive scans inside the lane
#pragma unroll
    for (int q = 1; q < 8; ++q) {
        Lmin[q] = min(Lmin[q], Lmin[q - 1]);
        Lmax[q] = min(Lmax[q], Lmax[q - 1]);
        LU[q] = min(LU[q], LU[q - 1]);
    }
#pragma unroll
    for (int q = 6; q >= 0; --q) {
        Rmin[q] = min(Rmin[q], Rmin[q + 1]);
        Rmax[q] = min(Rmax[q], Rmax[q + 1]);
        RU[q] = min(RU[q], RU[q + 1]);
    }
    // across the lanes, then across the waves of the row
    u32 e[6], tot[6];
    wave_prefix_min(Lmin[7], e[0], tot[0]);
    wave_prefix_min(Lmax[7], e[1], tot[1]);
    wave_prefix_min(LU[7], e[2], tot[2]);
    wave_suffix_min(Rmin[0], lane, e[3], tot[3]);
    wave_suffix_min(Rmax[0], lane, e[4], tot[4]);
    wave_suffix_min(RU[0], lane, e[5], tot[5]);
    if (nwv > 1) {  // block-uniform
        if (lane < 6) {
            u32 t = tot[0];
#pragma unroll
            for (int s = 1; s < 6; ++s) t = lane == s ? tot[s] : t;
            s_tot[wv][lane] = t;
        }
        __syncthreads();
        for (int w2 = 0; w2 < nwv; ++w2) {
            if (w2 < wv) {
#pragma unroll
                for (int s = 0; s < 3; ++s) e[s] = min(e[s], s_tot[w2][s]);
            } else if (w2 > wv) {
#pragma unroll
                for (int s = 3; s < 6; ++s) e[s] = min(e[s], s_tot[w2][s]);
            }
        }
    }

    // ---- per pixel: distance, nearest source, flags; label, depth; stores
    const size_t fo = (size_t)b * H * W;
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const size_t ro = fo + (size_t)i * W + idx0;
    float fd[8], fv[8];
    int lab[8];
    u32 p0 = 0, p1 = 0, p2 = 0, pl = 0, pt = 0;
    u32 spix[8], wrd[8];  // frame offset of the nearest source (kmin's), its word index in the bit arrays
    int dd_[8], kk[8];
    bool nn[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = idx0 + q;
        const u32 jl = (u32)j << K_SH;
        const u32 bmin = min(min(Lmin[q], e[0]) + jl - ((u32)K_OFF << K_SH), min(Rmin[q], e[3]) - jl);
        const u32 bmax = min(min(Lmax[q], e[1]) + jl - ((u32)K_OFF << K_SH), min(Rmax[q], e[4]) - jl);
        const u32 bu = min(min(LU[q], e[2]) + jl - ((u32)K_OFF << K_SH), min(RU[q], e[5]) - jl);
        const int d = (int)(bmin >> K_SH), kmin = (int)(bmin >> 2) & 8191, kmax = 8191 - ((int)(bmax >> 2) & 8191);
        const int dU = (int)(bu >> K_SH), kU = (int)(bu >> 2) & 8191;
        const bool none = d >= MAX_HW_SUM;  // no source in the frame
        const bool uniq = kmin == kmax && !(bmin & 2u);
        const bool live = dU == d && 3 * (kU - j) <= 2 * d;
        const bool tie = !uniq && d != 0 && !none;
        const bool inw = j < W;
        p0 |= (inw ? (u32)d & 1u : 0u) << q;
        p1 |= (inw ? ((u32)d >> 1) & 1u : 0u) << q;
        p2 |= (inw ? ((u32)d >> 2) & 1u : 0u) << q;
        pl |= ((inw && live && !none) ? 1u : 0u) << q;
        pt |= ((inw && tie) ? 1u : 0u) << q;
        fd[q] = none ? 8192.0f : (float)d;  // float(INIT_DIST0 * 2^-16) == 8192.0f
        // the nearest source in column kmin (for a tie pixel: one of its nearest sources; k_ties overwrites it)
        const int gk = d - abs(j - kmin);
        const int si = min(max((bmin & 1u) ? i + gk : i - gk, 0), H - 1);  // clamps: never taken on a correct frame
        const int sj = min(kmin, W - 1);
        spix[q] = (u32)(si * W + sj);
        wrd[q] = (u32)(si * Wd + (sj >> 6));
        dd_[q] = si;
        kk[q] = sj;
        nn[q] = none;
    }
    if (out_depth || out_index) {
        u32 base[8];
        u64 word[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const size_t w = (size_t)b * H * Wd + wrd[q];
            base[q] = rowbase_s[(size_t)b * H + dd_[q]] + wpre_s[w];
            word[q] = srcbits[w];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            lab[q] = nn[q] ? 0 : source_rank(base[q], word[q], kk[q]);
            fv[q] = (out_depth && idx0 + q < W)
                        ? gather_depth(x + fo, vlist + fo, lab[q], (int)spix[q], nval, misaligned, frame_status + b)
                        : 0.0f;
        }
    }
    {
        const int byte = idx0 >> 3;
        if (byte < Wp) {
            u8 *pb = planes + ((size_t)b * H + i) * Wp + byte;
            pb[PL_D0 * plane_bytes] = (u8)p0;
            pb[PL_D1 * plane_bytes] = (u8)p1;
            pb[PL_D2 * plane_bytes] = (u8)p2;
            pb[PL_LIVE * plane_bytes] = (u8)pl;
            pb[PL_TIE * plane_bytes] = (u8)pt;
        }
    }
    if (VEC && ovec) {
        if (idx0 < W) {
            if (out_dt) {
                float4 *o = reinterpret_cast<float4 *>(out_dt + ro);
                o[0] = make_float4(fd[0], fd[1], fd[2], fd[3]);
                o[1] = make_float4(fd[4], fd[5], fd[6], fd[7]);
            }
            if (out_index) {
                int4 *o = reinterpret_cast<int4 *>(out_index + ro);
                o[0] = make_int4(lab[0], lab[1], lab[2], lab[3]);
                o[1] = make_int4(lab[4], lab[5], lab[6], lab[7]);
            }
            if (out_depth) {
                float4 *o = reinterpret_cast<float4 *>(out_depth + ro);
                o[0] = make_float4(fv[0], fv[1], fv[2], fv[3]);
                o[1] = make_float4(fv[4], fv[5], fv[6], fv[7]);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            if (idx0 + q >= W) continue;
            if (out_dt) out_dt[ro + q] = fd[q];
            if (out_index) out_index[ro + q] = lab[q];
            if (out_depth) out_depth[ro + q] = fv[q];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_ties: one block (256 threads) per 64 x 128 tile.  Window = tile + 16 rows above / below and one 32-pixel
// word left / right, as bit planes in LDS.  The 5x5 parent rule runs bit-sliced (32 pixels per operation, on
// d mod 8: a tap of weight w <= 3 matches iff d(r) + w == d(q), and |d(r) - d(q)| <= w makes that exact mod 8)
// for the words that hold a tie pixel; the tie pixels of the tile then hop through the window until they stand
// on a pixel that is not a tie pixel (one nearest source, or a source) and copy its label and depth, which
// k_rows has written.  A hop that would need the rule of a pixel outside the window's safe region (2 cells
// inside its edge) hands the pixel to k_tiesx through the "unresolved" plane.
// ------------------------------------------------------------------------------------------------
constexpr int Q_TH = 64, Q_TW = 128, Q_MR = 16;
constexpr int Q_WR = Q_TH + 2 * Q_MR;  // window rows
constexpr int Q_NW = Q_TW / 32 + 2;    // window words per row
constexpr int Q_RS = 9;                // LDS row pitch in words: zero pad, 6 words, zero pad, (odd stride)
constexpr int Q_NPL = 6;               // planes in LDS: d bit 0, 1, 2, live, tie, in-image
constexpr int Q_NT = 256;
static_assert(Q_TH * Q_TW / 32 == Q_NT, "one tile word per thread");

template <int DJ>
__device__ __forceinline__ u32 qshift(const u32 (&a)[3]) {  // bits of the pixels (column + DJ) of the word a[1]
    if (DJ == 0) return a[1];
    if (DJ > 0) return __builtin_amdgcn_alignbit(a[2], a[1], DJ);
    return __builtin_amdgcn_alignbit(a[1], a[0], 32 + DJ);
}
// One tap of the parent rule for 32 pixels; the candidate r = q + (row of the arrays, DJ).  FWD: r must be live.
template <int DJ, int WGT, bool FWD, int CODE>
__device__ __forceinline__ void rule_tap(const u32 (&a0)[3], const u32 (&a1)[3], const u32 (&a2)[3], const u32 (&lv)[3],
                                         const u32 (&vd)[3], u32 b0, u32 b1, u32 b2, u32 &taken, u32 (&C)[4]) {
    const u32 x0 = qshift<DJ>(a0), x1 = qshift<DJ>(a1), x2 = qshift<DJ>(a2);
    u32 s0, s1, s2;  // (d(r) + WGT) mod 8
    if (WGT == 1) {
        s0 = ~x0; s1 = x1 ^ x0; s2 = x2 ^ (x1 & x0);
    } else if (WGT == 2) {
        s0 = x0; s1 = ~x1; s2 = x2 ^ x1;
    } else {
        s0 = ~x0; s1 = ~(x1 ^ x0); s2 = x2 ^ (x1 | x0);
    }
    u32 m = ~((s0 ^ b0) | (s1 ^ b1) | (s2 ^ b2)) & qshift<DJ>(vd);
    if (FWD) m &= qshift<DJ>(lv);
    const u32 sel = m & ~taken;
    taken |= m;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (CODE & (1 << j)) C[j] |= sel;
}

__global__ __launch_bounds__(Q_NT) void k_ties(const u8 *__restrict__ planes, size_t plane_bytes, int Wp,
                                               const int *__restrict__ fflag, int H, int W, int tiles_x,
                                               float *__restrict__ out_depth,
                                               int32_t *__restrict__ out_index, u8 *__restrict__ unres) {
    __shared__ u32 s_pl[Q_NPL][Q_WR][Q_RS];
    __shared__ u32 s_code[4][Q_WR][Q_RS];
    __shared__ u16 s_list[Q_TH * Q_TW];
    __shared__ u32 s_unres[Q_NT];
    __shared__ u32 s_cnt[Q_NT / 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (!fflag[b]) return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * Q_TH, c0 = tx * Q_TW;
    const int wr0 = r0 - Q_MR, ww0 = (c0 >> 5) - 1;  // image row / image word column of window cell (0, 0)
    const size_t rowb = (size_t)b * H;
    // this thread's tile word
    const int trow = tid >> 2, tw = tid & 3;
    const int gi = r0 + trow, gw = (c0 >> 5) + tw;  // image row, image word (32 px) column
    const bool tin = gi < H && gw * 4 < Wp;  // the word exists in the planes (its pixels beyond W are zero bits)
    u32 mytie = 0;
    if (tin) mytie = *reinterpret_cast<const u32 *>(planes + PL_TIE * plane_bytes + (rowb + gi) * Wp + 4 * gw);
    u32 *ures = reinterpret_cast<u32 *>(unres + (rowb + min(gi, H - 1)) * Wp) + gw;
    if (!__syncthreads_or(mytie != 0)) {  // nothing to do in this tile
        if (tin) *ures = 0;
        return;
    }
    // ---- the window's planes -> LDS (all of a thread's loads are issued before its first LDS store)
    {
        constexpr int NITEM = Q_NPL * Q_WR * Q_NW, PER = (NITEM + Q_NT - 1) / Q_NT;
        u32 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = tid + u * Q_NT;
            const int pl = k / (Q_WR * Q_NW), rem = k - pl * (Q_WR * Q_NW);
            const int rr = rem / Q_NW, w = rem - rr * Q_NW;
            const int i = wr0 + rr, wi = ww0 + w;
            const bool in = k < NITEM && i >= 0 && i < H && wi >= 0 && wi * 32 < W;
            const int ic = min(max(i, 0), H - 1), wic = min(max(wi, 0), (Wp >> 2) - 1);  // clamped: unconditional loads
            const u32 ld = reinterpret_cast<const u32 *>(planes + min(pl, 4) * plane_bytes + (rowb + ic) * Wp)[wic];
            const int up = min(W - wi * 32, 32);  // in-image columns of this word: [0, up)
            const u32 inimg = !in ? 0u : (up >= 32 ? 0xFFFFFFFFu : ((1u << up) - 1u));
            v[u] = pl == 5 ? inimg : (ld & inimg);
        }
        for (int k = tid; k < Q_NPL * Q_WR; k += Q_NT) {  // the pad words
            (&s_pl[0][0][0])[k * Q_RS] = 0;
            (&s_pl[0][0][0])[k * Q_RS + Q_NW + 1] = 0;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = tid + u * Q_NT;
            if (k < NITEM) {
                const int pl = k / (Q_WR * Q_NW), rem = k - pl * (Q_WR * Q_NW);
                const int rr = rem / Q_NW, w = rem - rr * Q_NW;
                s_pl[pl][rr][w + 1] = v[u];
            }
        }
    }
    s_unres[tid] = 0;
    __syncthreads();
    // ---- parent rule, bit-sliced, for the window words (rows 2 .. Q_WR-3) that hold a tie pixel.
    // code = tap t (forward, live pixels) or 8 | t (the negated tap, the others): the format tap_decode reads
    for (int it = tid; it < (Q_WR - 4) * Q_NW; it += Q_NT) {
        const int qrow = 2 + it / Q_NW, pw = it - (qrow - 2) * Q_NW;  // window row, window word
        const u32 qtie = s_pl[4][qrow][pw + 1];
        if (!qtie) continue;
        auto ld3 = [&](int p, int row, u32 (&o)[3]) {
            const u32 *s = &s_pl[p][row][pw];
            o[0] = s[0]; o[1] = s[1]; o[2] = s[2];
        };
        const u32 b0 = s_pl[0][qrow][pw + 1], b1 = s_pl[1][qrow][pw + 1], b2 = s_pl[2][qrow][pw + 1];
        const u32 qlive = s_pl[3][qrow][pw + 1];
        u32 takenF = ~(qtie & qlive), takenB = ~(qtie & ~qlive);
        u32 C[4] = {0, 0, 0, 0};
        u32 a0[3], a1[3], a2[3], lv[3], vd[3];
        ld3(0, qrow - 2, a0); ld3(1, qrow - 2, a1); ld3(2, qrow - 2, a2); ld3(3, qrow - 2, lv); ld3(5, qrow - 2, vd);
        rule_tap<-1, 3, true, 0>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        rule_tap<+1, 3, true, 1>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        ld3(0, qrow - 1, a0); ld3(1, qrow - 1, a1); ld3(2, qrow - 1, a2); ld3(3, qrow - 1, lv); ld3(5, qrow - 1, vd);
        rule_tap<-2, 3, true, 2>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        rule_tap<-1, 2, true, 3>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        rule_tap<0, 1, true, 4>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        rule_tap<+1, 2, true, 5>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        rule_tap<+2, 3, true, 6>(a0, a1, a2, lv, vd, b0, b1, b2, takenF, C);
        u32 z0[3], z1[3], z2[3], zv[3];  // this row: last forward tap now, last backward tap at the end
        ld3(0, qrow, z0); ld3(1, qrow, z1); ld3(2, qrow, z2); ld3(3, qrow, lv); ld3(5, qrow, zv);
        rule_tap<-1, 1, true, 7>(z0, z1, z2, lv, zv, b0, b1, b2, takenF, C);
        ld3(0, qrow + 2, a0); ld3(1, qrow + 2, a1); ld3(2, qrow + 2, a2); ld3(5, qrow + 2, vd);
        rule_tap<+1, 3, false, 8 | 0>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        rule_tap<-1, 3, false, 8 | 1>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        ld3(0, qrow + 1, a0); ld3(1, qrow + 1, a1); ld3(2, qrow + 1, a2); ld3(5, qrow + 1, vd);
        rule_tap<+2, 3, false, 8 | 2>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        rule_tap<+1, 2, false, 8 | 3>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        rule_tap<0, 1, false, 8 | 4>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        rule_tap<-1, 2, false, 8 | 5>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        rule_tap<-2, 3, false, 8 | 6>(a0, a1, a2, lv, vd, b0, b1, b2, takenB, C);
        rule_tap<+1, 1, false, 8 | 7>(z0, z1, z2, lv, zv, b0, b1, b2, takenB, C);
#pragma unroll
        for (int j = 0; j < 4; ++j) s_code[j][qrow][pw + 1] = C[j];
    }
    // ---- the tile's tie pixels as a list (window row << 8 | window column), so that the hops are spread evenly
    int n_all;
    {
        const int cnt = __popc(mytie);
        int incl = cnt;
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_cnt[wave] = (u32)incl;
        __syncthreads();  // also: s_code is complete
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < Q_NT / 64; ++w) {
            pre += w < wave ? (int)s_cnt[w] : 0;
            all += (int)s_cnt[w];
        }
        n_all = all;
        int o = pre + incl - cnt;
        u32 m = mytie;
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            s_list[o++] = (u16)((Q_MR + trow) << 8 | (32 + tw * 32 + bit));
        }
    }
    __syncthreads();
    // ---- hops
    const size_t fo = (size_t)b * H * W;
    for (int e = tid; e < n_all; e += Q_NT) {
        const int start = s_list[e];
        int wr = start >> 8, wc = start & 255;
        bool solved = false;
        for (int hop = 0; hop < Q_WR + Q_NW * 32; ++hop) {  // d falls with every hop: the bound is never reached
            const int wi = 1 + (wc >> 5), bit = wc & 31;
            const int code = (int)((s_code[0][wr][wi] >> bit) & 1u) | (int)((s_code[1][wr][wi] >> bit) & 1u) << 1 |
                             (int)((s_code[2][wr][wi] >> bit) & 1u) << 2 | (int)((s_code[3][wr][wi] >> bit) & 1u) << 3;
            int di, dj;
            tap_decode(code, di, dj);
            wr += di;
            wc += dj;
            if (!((s_pl[4][wr][1 + (wc >> 5)] >> (wc & 31)) & 1u)) {  // not a tie pixel: the chain's label is its label
                solved = true;
                break;
            }
            if (wr < 2 || wr >= Q_WR - 2 || wc < 2 || wc >= Q_NW * 32 - 2) break;  // its rule was not evaluated here
        }
        const int q = (r0 + (start >> 8) - Q_MR) * W + c0 - 32 + (start & 255);
        const int ei = wr0 + wr, ej = ww0 * 32 + wc;  // where the hops ended: always inside the image on consistent planes
        if (solved && ei >= 0 && ei < H && ej >= 0 && ej < W) {
            const int p = ei * W + ej;
            if (out_index) out_index[fo + q] = out_index[fo + p];
            if (out_depth) out_depth[fo + q] = out_depth[fo + p];
        } else {
            const int sw = ((start >> 8) - Q_MR) * 4 + (((start & 255) - 32) >> 5);
            atomicOr(&s_unres[sw], 1u << (start & 31));
        }
    }
    __syncthreads();
    if (tin) *ures = s_unres[tid];
}

// ------------------------------------------------------------------------------------------------
// k_tiesx: the tie pixels k_ties could not finish inside its window (one thread per 32-pixel word of the
// "unresolved" plane; normally every word is zero).  Same rule, evaluated per hop from the planes in global
// memory: any chain length, no assumption about the window.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tiesx(const u8 *__restrict__ planes, size_t plane_bytes, int Wp,
                                               const int *__restrict__ fflag, int H, int W,
                                               float *__restrict__ out_depth, int32_t *__restrict__ out_index) {
    const int b = blockIdx.y;
    if (!fflag[b]) return;
    const int wpr = Wp >> 2;  // 32-pixel words per plane row
    const int widx = blockIdx.x * 256 + threadIdx.x;
    if (widx >= H * wpr) return;
    const size_t rowb = (size_t)b * H;
    u32 m = *reinterpret_cast<const u32 *>(planes + PL_UNRES * plane_bytes + rowb * Wp + 4 * (size_t)widx);
    if (!m) return;
    const int qi = widx / wpr, qw = widx - qi * wpr;
    const size_t fo = (size_t)b * H * W;
    auto bit = [&](int pl, int i, int j) -> u32 {
        return (planes[pl * plane_bytes + (rowb + i) * Wp + (j >> 3)] >> (j & 7)) & 1u;
    };
    auto dmod = [&](int i, int j) -> int {
        return (int)(bit(PL_D0, i, j) | bit(PL_D1, i, j) << 1 | bit(PL_D2, i, j) << 2);
    };
    while (m) {
        const int qb = __ffs((int)m) - 1;
        m &= m - 1;
        int pi = qi, pj = qw * 32 + qb;
        for (int hop = 0; hop < MAX_HW_SUM && bit(PL_TIE, pi, pj); ++hop) {  // d falls with every hop
            const int dq = dmod(pi, pj);
            const bool fwd = bit(PL_LIVE, pi, pj);
            int t = 0;
            for (; t < 8; ++t) {
                int di, dj;
                tap_decode(fwd ? t : 8 | t, di, dj);
                const int ri = pi + di, rj = pj + dj;
                if (ri < 0 || ri >= H || rj < 0 || rj >= W) continue;
                const int w = abs(di) + abs(dj);
                if (((dmod(ri, rj) + w) & 7) != dq) continue;
                if (fwd && !bit(PL_LIVE, ri, rj)) continue;
                pi = ri;
                pj = rj;
                break;
            }
            if (t == 8) break;  // cannot happen on consistent planes; never spin
        }
        const int q = qi * W + qw * 32 + qb, p = pi * W + pj;
        if (out_index) out_index[fo + q] = out_index[fo + p];
        if (out_depth) out_depth[fo + q] = out_depth[fo + p];
    }
}
