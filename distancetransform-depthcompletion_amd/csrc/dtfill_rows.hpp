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
// Output per band and column: {source bits of the band's 32 rows in this column, distance from the band's first row
// to the nearest source above the band | distance from its last row to the nearest source below << 16}; the
// rows are CTP columns long (W rounded up to 64, plus slack for the last lane of k_rows), the columns beyond W hold
// "no source".
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline int ct_pitch(int W) { return ((W + 63) / 64) * 64 + 640; }

__global__ __launch_bounds__(1024) void k_colT(const u64 *__restrict__ srcbits, const int *__restrict__ fflag, int H,
                                               int W, int Wd, int nb, int CTP, uint2 *__restrict__ ct) {
    extern __shared__ u16 s_lf[];  // [nb][64] last source row of the band, [nb][64] first (0xFFFF: none)
    const int b = blockIdx.y, wd = blockIdx.x, lane = threadIdx.x & 63;
    const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    if (fflag && !fflag[b]) return;
    const int j = wd * 64 + lane;
    const bool real = wd < Wd;  // block-uniform: word columns beyond the image only write the "no source" padding
    const u64 *sbf = srcbits + (size_t)b * H * Wd + min(wd, Wd - 1);
    u16 *s_last = s_lf, *s_first = s_lf + nb * 64;
    for (int band = ch; band < nb; band += nwv) {
        const int i0 = band * 32, i1 = min(i0 + 32, H);
        u32 bits = 0;
        if (real) {
#pragma unroll
            for (int kb = 0; kb < 32; kb += 16) {  // wave-uniform addresses: scalar loads, 16 in flight
                u64 w[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) w[k] = sbf[(size_t)min(i0 + kb + k, i1 - 1) * Wd];
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (i0 + kb + k < i1) bits |= (u32)((w[k] >> lane) & 1ull) << (kb + k);
            }
        }
        if (j < CTP) ct[((size_t)b * nb + band) * CTP + j].x = bits;
        s_last[band * 64 + lane] = bits ? (u16)(i0 + 31 - __clz((int)bits)) : (u16)0xFFFF;
        s_first[band * 64 + lane] = bits ? (u16)(i0 + __ffs((int)bits) - 1) : (u16)0xFFFF;
    }
    __syncthreads();
    if (ch == 0) {  // distance from the band's FIRST row to the nearest source above the band (low half)
        int run = -1;
        for (int band = 0; band < nb; ++band) {
            const int up = run < 0 ? GBIG : min(band * 32 - run, GBIG);
            if (j < CTP) reinterpret_cast<u16 *>(&ct[((size_t)b * nb + band) * CTP + j].y)[0] = (u16)up;
            const int l = s_last[band * 64 + lane];
            run = l != 0xFFFF ? l : run;
        }
    } else if (ch == 1) {  // distance from the band's LAST row to the nearest source below the band (high half)
        int run = -1;
        for (int band = nb - 1; band >= 0; --band) {
            const int dn = run < 0 ? GBIG : min(run - (band * 32 + 31), GBIG);
            if (j < CTP) reinterpret_cast<u16 *>(&ct[((size_t)b * nb + band) * CTP + j].y)[1] = (u16)dn;
            const int f = s_first[band * 64 + lane];
            run = f != 0xFFFF ? f : run;
        }
    }
}

__device__ __forceinline__ u32 ffbh_u32(u32 v) {  // position of the highest set bit from the top; 0xFFFFFFFF for 0
    u32 r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
__device__ __forceinline__ u32 ffbl_b32(u32 v) {  // position of the lowest set bit; 0xFFFFFFFF for 0
    u32 r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// uniform base + 32-bit BYTE offset: the form that needs no 64-bit vector address arithmetic
template <typename T>
__device__ __forceinline__ T ld_off(const void *base, u32 byte_off) {
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_off(void *base, u32 byte_off, T v) {
    *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_off) = v;
}
__device__ __forceinline__ u32 min3u(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// ------------------------------------------------------------------------------------------------
// wave scans of packed keys (unsigned min) on the DPP network, three keys at a time: v_min_u32 with a DPP source
// (row_shr / row_shl inside a row of 16 lanes; lanes the pattern does not feed keep their value), row_bcast (prefix)
// or three readlanes (suffix) across the rows, wave_shr / wave_shl for the exclusive shift.  The three keys are
// interleaved so that a register written by one step is read two instructions later (the DPP read-after-VALU-write
// hazard needs two wait states; the s_nop covers the instruction before the block).
// ------------------------------------------------------------------------------------------------
#define DPP3(ctrl)                                        \
    "v_min_u32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t" \
    "v_min_u32_dpp %1, %1, %1 " ctrl " bank_mask:0xf\n\t" \
    "v_min_u32_dpp %2, %2, %2 " ctrl " bank_mask:0xf\n\t"

// in: per-lane values; out: e* = min over the lanes BEFORE this one (K_IDENT for lane 0), t* = min over the wave
__device__ __forceinline__ void wave_prefix_min3(u32 a, u32 b, u32 c, u32 &ea, u32 &eb, u32 &ec, u32 &ta, u32 &tb,
                                                 u32 &tc) {
    asm volatile("s_nop 1\n\t" DPP3("row_shr:1 row_mask:0xf") DPP3("row_shr:2 row_mask:0xf") DPP3("row_shr:4 row_mask:0xf")
                     DPP3("row_shr:8 row_mask:0xf") DPP3("row_bcast:15 row_mask:0xa") DPP3("row_bcast:31 row_mask:0xc") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c));
    ta = (u32)__builtin_amdgcn_readlane((int)a, 63);
    tb = (u32)__builtin_amdgcn_readlane((int)b, 63);
    tc = (u32)__builtin_amdgcn_readlane((int)c, 63);
    ea = (u32)__builtin_amdgcn_update_dpp((int)K_IDENT, (int)a, 0x138, 0xF, 0xF, false);  // wave_shr:1
    eb = (u32)__builtin_amdgcn_update_dpp((int)K_IDENT, (int)b, 0x138, 0xF, 0xF, false);
    ec = (u32)__builtin_amdgcn_update_dpp((int)K_IDENT, (int)c, 0x138, 0xF, 0xF, false);
}
// e* = min over the lanes AFTER this one (K_IDENT for lane 63)
__device__ __forceinline__ void wave_suffix_min3(u32 a, u32 b, u32 c, int lane, u32 &ea, u32 &eb, u32 &ec, u32 &ta,
                                                 u32 &tb, u32 &tc) {
    asm volatile("s_nop 1\n\t" DPP3("row_shl:1 row_mask:0xf") DPP3("row_shl:2 row_mask:0xf") DPP3("row_shl:4 row_mask:0xf")
                     DPP3("row_shl:8 row_mask:0xf") "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c));
    const int row = lane >> 4;
    auto across = [&](u32 v, u32 &e, u32 &t) {
        const u32 t1 = (u32)__builtin_amdgcn_readlane((int)v, 16), t2 = (u32)__builtin_amdgcn_readlane((int)v, 32),
                  t3 = (u32)__builtin_amdgcn_readlane((int)v, 48);
        const u32 s23 = min(t2, t3), s123 = min(t1, s23);
        v = min(v, row == 0 ? s123 : row == 1 ? s23 : row == 2 ? t3 : K_IDENT);
        t = (u32)__builtin_amdgcn_readlane((int)v, 0);
        e = (u32)__builtin_amdgcn_update_dpp((int)K_IDENT, (int)v, 0x130, 0xF, 0xF, false);  // wave_shl:1
    };
    across(a, ea, ta);
    across(b, eb, tb);
    across(c, ec, tc);
}

// ------------------------------------------------------------------------------------------------
// k_rows<PPL>: one block per image row, one wave per 64 * PPL columns, PPL consecutive columns per lane (the host
// picks PPL = 8 or 10, whichever leaves fewer idle lanes: 1216 = 2 x 640 - 64, 640 = 1 x 640, 2048 = 4 x 512).
// Bit planes for k_ties are assembled in LDS (a lane's PPL bits straddle words when PPL is not 8) and leave as
// whole 32-pixel words.
// ------------------------------------------------------------------------------------------------
constexpr int R_MAXWV = 16;    // W <= 8191 -> at most 16 waves of 512 columns
constexpr int R_MAXPW = 256;   // 32-pixel words per row: W <= 8191

template <int PPL, int MAXT>  // MAXT: 256 (rows of up to 4 waves; 3 waves per SIMD) or 1024 (any row the shape limit allows)
__global__ __launch_bounds__(MAXT, MAXT == 256 ? 4 : 1) void k_rows(
    const float *__restrict__ x, const uint2 *__restrict__ ct, int CTP, const u64 *__restrict__ srcbits,
    const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s, const int *__restrict__ finfo,
    const float *__restrict__ vlist, const int *__restrict__ fflag, int H, int W, int Wd, int nb, int Wp,
    u8 *__restrict__ planes, size_t plane_bytes, float *__restrict__ out_depth, float *__restrict__ out_dt,
    int32_t *__restrict__ out_index, int *__restrict__ frame_status, int ovec) {
    static_assert(PPL == 8 || PPL == 10, "loads and stores below are written for 8 or 10 columns per lane");
    __shared__ u32 s_tot[R_MAXWV][6];
    __shared__ u32 s_bits[5][R_MAXPW + 1];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    const int i = blockIdx.x, b = blockIdx.y;
    if (!fflag[b]) return;  // block-uniform
    const int band = i >> 5, r = i & 31;
    const int idx0 = (wv * 64 + lane) * PPL;
    const int wpr = Wp >> 2;
    for (int k = threadIdx.x; k < 5 * (R_MAXPW + 1); k += blockDim.x) (&s_bits[0][0])[k] = 0;  // (the row's words + 1 would do)

    // ---- the band words and carries of this lane's columns (16-byte loads; the padding holds "no source")
    u32 T[PPL], UD[PPL];
    {
        const uint4 *src = reinterpret_cast<const uint4 *>(ct + ((size_t)b * nb + band) * CTP + idx0);
#pragma unroll
        for (int q = 0; q < PPL; q += 2) {
            const uint4 v = src[q >> 1];
            T[q] = v.x; UD[q] = v.y; T[q + 1] = v.z; UD[q + 1] = v.w;
        }
    }
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const u32 fo = (u32)b * (u32)(H * W);  // < 2^31 (shape_ok)
    __syncthreads();  // s_bits is zero
    // six keys per column: {all sources, prefer the smallest column | all, prefer the largest | sources at or above
    // this row} x {left scan (value - column), right scan (value + column)}.  Branch-free: v_ffbh / v_ffbl give
    // 0xFFFFFFFF for "no bit", which loses the unsigned min against the carried distance.
    u32 Lmin[PPL], Lmax[PPL], LU[PPL], Rmin[PPL], Rmax[PPL], RU[PPL];
    {
        const u32 upmask = (2u << r) - 1u;  // band rows 0..r (r = 31: all)
        const u32 kbase = (u32)(K_OFF - idx0);
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const u32 gu = min(ffbh_u32(T[q] & upmask) + (u32)(r - 31), (UD[q] & 0xFFFFu) + (u32)r);
            const u32 gd = min(ffbl_b32(T[q] >> r), (UD[q] >> 16) + (u32)(31 - r));
            const u32 g = min(gu, gd);
            // flags of the column: bit 0 = its nearest source is below this row, bit 1 = above and below tie
            // (gu == gd != 0; two "no source" distances r + GBIG and 31 - r + GBIG are never equal)
            const u32 t = gd - gu;
            const u32 fl = (t >> 31) | (t == 0 ? min(g, 1u) << 1 : 0u);
            const u32 au = (u32)(idx0 + q) << 2;
            const u32 amin = au | fl, amax = 0x7FFCu - au;
            const u32 vl = g + kbase - (u32)q, vu = gu + kbase - (u32)q;
            const u32 t2 = (au << 14) - ((u32)K_OFF << K_SH);  // (2 k - K_OFF) << 15: left key -> right key
            Lmin[q] = vl << K_SH | amin;
            Lmax[q] = vl << K_SH | amax;
            LU[q] = vu << K_SH | au;
            Rmin[q] = Lmin[q] + t2;
            Rmax[q] = Lmax[q] + t2;
            RU[q] = LU[q] + t2;
        }
    }
    // the lane's minima (three-operand min), across the lanes, then across the waves of the row
    u32 e[6], tot[6];
    {
        auto lane_min = [](const u32 (&k)[PPL]) {
            u32 m = min3u(k[0], k[1], k[2]);
#pragma unroll
            for (int q = 3; q + 1 < PPL; q += 2) m = min3u(m, k[q], k[q + 1]);
            if ((PPL & 1) == 0) m = min(m, k[PPL - 1]);
            return m;
        };
        wave_prefix_min3(lane_min(Lmin), lane_min(Lmax), lane_min(LU), e[0], e[1], e[2], tot[0], tot[1], tot[2]);
        wave_suffix_min3(lane_min(Rmin), lane_min(Rmax), lane_min(RU), lane, e[3], e[4], e[5], tot[3], tot[4], tot[5]);
    }
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
    // inclusive scans inside the lane, seeded with everything before / after the lane
    Lmin[0] = min(Lmin[0], e[0]);
    Lmax[0] = min(Lmax[0], e[1]);
    LU[0] = min(LU[0], e[2]);
    Rmin[PPL - 1] = min(Rmin[PPL - 1], e[3]);
    Rmax[PPL - 1] = min(Rmax[PPL - 1], e[4]);
    RU[PPL - 1] = min(RU[PPL - 1], e[5]);
#pragma unroll
    for (int q = 1; q < PPL; ++q) {
        Lmin[q] = min(Lmin[q], Lmin[q - 1]);
        Lmax[q] = min(Lmax[q], Lmax[q - 1]);
        LU[q] = min(LU[q], LU[q - 1]);
    }
#pragma unroll
    for (int q = PPL - 2; q >= 0; --q) {
        Rmin[q] = min(Rmin[q], Rmin[q + 1]);
        Rmax[q] = min(Rmax[q], Rmax[q + 1]);
        RU[q] = min(RU[q], RU[q + 1]);
    }

    // ---- per pixel: distance, nearest source, flags
    float fd[PPL];
    u32 spix[PPL], wrd[PPL], srow[PPL], scol[PPL];  // the nearest source (kmin's): frame offset, bit word, row, column
    u32 acc012 = 0, acc34 = 0;  // bit planes of the lane's pixels: plane p of pixel q at bit 10 p + q
    u32 nonebits = 0;
    const u32 jbase = ((u32)idx0 << K_SH) - ((u32)K_OFF << K_SH);
#pragma unroll
    for (int q = 0; q < PPL; ++q) {
        const int j = idx0 + q;
        const u32 cl = jbase + ((u32)q << K_SH), cr = (u32)j << K_SH;
        const u32 bmin = min(Lmin[q] + cl, Rmin[q] - cr);
        const u32 bmax = min(Lmax[q] + cl, Rmax[q] - cr);
        const u32 bu = min(LU[q] + cl, RU[q] - cr);
        const u32 d = bmin >> K_SH, kmin = (bmin >> 2) & 8191u;
        const bool none = bmin >= ((u32)MAX_HW_SUM << K_SH);  // no source in the frame
        // one nearest source: the smallest and the largest column that reach d are the same (the column field of bmax
        // is the complement of the one of bmin), and that column is not tied above / below (bit 1; bmax has 0 there)
        const bool uniq = ((bmin ^ bmax) & 0x7FFEu) == 0x7FFCu;
        // live: the leftmost nearest source at or above this row exists (its distance is d) and lies in the forward cone
        const u32 ku = (bu >> 2) & 8191u;
        const bool live = (((bu ^ bmin) >> K_SH) == 0) & ((int)(3u * ku) <= (int)(2u * d + 3u * (u32)j)) & !none;  // & not &&: no branches
        const bool tie = !uniq & (d != 0) & !none;
        const u32 inm = (u32)((j - W) >> 31);  // all ones inside the image (a mask, not a branch)
        const u32 b5 = ((d & 7u) | (live ? 8u : 0u) | (tie ? 16u : 0u)) & inm;  // d mod 8 | live << 3 | tie << 4
        // bit p of b5 -> bit 10 p + q: copies of the low three / the high two bits at offsets 0, 9, 18 (24-bit multiply)
        acc012 |= (__umul24(b5 & 7u, 0x040201u) & 0x100401u) << q;
        acc34 |= (__umul24(b5 >> 3, 0x000201u) & 0x000401u) << q;
        nonebits |= none ? 1u << q : 0u;
        fd[q] = none ? 8192.0f : (float)d;  // float(INIT_DIST0 * 2^-16) == 8192.0f
        // the nearest source in column kmin (for a tie pixel: one of its nearest sources; k_ties overwrites it)
        const int gk = (int)d - (int)__builtin_amdgcn_sad_u16((u32)j, kmin, 0u);  // d - |j - kmin| (both < 2^13)
        const int sg = (int)((bmin & 1u) << 1) - 1;                              // below: +1, above: -1
        const int si = min(max(i + sg * gk, 0), H - 1);  // clamps: never taken on a correct frame
        const u32 sj = min(kmin, (u32)(W - 1));
        srow[q] = (u32)si;
        scol[q] = sj;
        spix[q] = (u32)si * (u32)W + sj;
        wrd[q] = (u32)si * (u32)Wd + (sj >> 6);
    }
    // ---- bit planes: a lane's PPL bits of each plane are ORed into the row's words in LDS
    if (idx0 < W) {
        const int w0 = idx0 >> 5, sh = idx0 & 31;
        constexpr u32 PM = (1u << PPL) - 1u;
#pragma unroll
        for (int p = 0; p < 5; ++p) {
            const u32 bits = (p < 3 ? acc012 >> (10 * p) : acc34 >> (10 * (p - 3))) & PM;
            if (bits) {
                atomicOr(&s_bits[p][w0], bits << sh);
                if (sh + PPL > 32) atomicOr(&s_bits[p][w0 + 1], bits >> (32 - sh));
            }
        }
    }
    // ---- label, depth: all loads of the lane's pixels are issued together
    int lab[PPL];
    float fv[PPL];
    if (out_depth || out_index) {
        // frame bases are block-uniform (scalar registers); per pixel only 32-bit byte offsets
        const u64 *sb_f = srcbits + (size_t)b * H * Wd;
        const u16 *wp_f = wpre_s + (size_t)b * H * Wd;
        const u32 *rb_f = rowbase_s + (size_t)b * H;
        const float *x_f = x + fo;
        u32 base[PPL];
        u64 word[PPL];
        bool bad = false;
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            base[q] = ld_off<u32>(rb_f, srow[q] << 2) + ld_off<u16>(wp_f, wrd[q] << 1);
            word[q] = ld_off<u64>(sb_f, wrd[q] << 3);
        }
        if (!misaligned) {  // block-uniform: the label-th value IS x at the source pixel
#pragma unroll
            for (int q = 0; q < PPL; ++q) fv[q] = ld_off<float>(x_f, spix[q] << 2);
        }
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            const bool none = (nonebits >> q) & 1u;
            lab[q] = none ? 0 : source_rank(base[q], word[q], (int)scol[q]);  // a select, not a branch
        }
        if (misaligned || nonebits) {  // rare: value list gather with numpy's index rules (tools.py:26)
#pragma unroll
            for (int q = 0; q < PPL; ++q) {
                int idx = lab[q] - 1;
                if (idx < 0) idx += nval;  // numpy: index -1 wraps to the last element
                const bool oob = idx < 0 || idx >= nval;
                bad |= oob && idx0 + q < W;
                // label 0 with aligned masks: nval == nsrc == 0 -> out of bounds; so an in-bounds gather reads the value list
                fv[q] = oob ? nanf("") : (misaligned ? vlist[fo + idx] : x_f[spix[q]]);
            }
        }
        if (bad && out_depth) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
    }
    // ---- stores (frame bases scalar, 32-bit byte offsets)
    const bool full = idx0 + PPL <= W;
    const u32 rob = ((u32)i * (u32)W + (u32)idx0) << 2;
    float *dt_f = out_dt ? out_dt + fo : nullptr, *dp_f = out_depth ? out_depth + fo : nullptr;
    int32_t *ix_f = out_index ? out_index + fo : nullptr;
    if (ovec && full) {
        if (PPL == 8) {
            if (dt_f) {
                st_off(dt_f, rob, make_float4(fd[0], fd[1], fd[2], fd[3]));
                st_off(dt_f, rob + 16, make_float4(fd[4], fd[5], fd[6], fd[7]));
            }
            if (ix_f) {
                st_off(ix_f, rob, make_int4(lab[0], lab[1], lab[2], lab[3]));
                st_off(ix_f, rob + 16, make_int4(lab[4], lab[5], lab[6], lab[7]));
            }
            if (dp_f) {
                st_off(dp_f, rob, make_float4(fv[0], fv[1], fv[2], fv[3]));
                st_off(dp_f, rob + 16, make_float4(fv[4], fv[5], fv[6], fv[7]));
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPL; q += 2) {
                if (dt_f) st_off(dt_f, rob + 4 * q, make_float2(fd[q], fd[q + 1]));
                if (ix_f) st_off(ix_f, rob + 4 * q, make_int2(lab[q], lab[q + 1]));
                if (dp_f) st_off(dp_f, rob + 4 * q, make_float2(fv[q], fv[q + 1]));
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (idx0 + q >= W) continue;
            if (dt_f) st_off(dt_f, rob + 4 * q, fd[q]);
            if (ix_f) st_off(ix_f, rob + 4 * q, lab[q]);
            if (dp_f) st_off(dp_f, rob + 4 * q, fv[q]);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 5 * wpr; k += blockDim.x) {  // the row's plane words leave as whole words
        const int p = k / wpr, w = k - p * wpr;
        reinterpret_cast<u32 *>(planes + p * plane_bytes + ((size_t)b * H + i) * Wp)[w] = s_bits[p][w];
    }
}

// ------------------------------------------------------------------------------------------------
// k_ties: one block (256 threads) per 32 x 256 tile, one thread per 32-pixel word of it.  The tile's planes (d mod 8,
// live, tie) with a 2-cell ring go to LDS once (every plane word is read from memory once per block, in runs of
// 40 bytes).  A thread whose word holds a tie pixel applies the 5x5 parent rule BIT-SLICED, 32 pixels per
// operation, on d mod 8: a tap of weight w <= 3 matches iff d(r) + w == d(q), and |d(r) - d(q)| <= w makes that
// exact mod 8.  The tile's tie pixels, as a list, then hop through the tile's code planes in LDS until they stand
// on a pixel that is not a tie pixel (one nearest source, or a source) and copy that pixel's label and depth, which
// k_rows has written.  A chain that leaves the tile while still on tie pixels records where it goes on (xptr[q] =
// that pixel, and q's bit in the "unresolved" plane, every word of which is written here) and is listed for k_tiesx.
// ------------------------------------------------------------------------------------------------
constexpr int Q_TH = 32, Q_TW = 256;
constexpr int Q_NT = 256;
constexpr int Q_WW = Q_TW / 32;  // tile words per row
constexpr int Q_EB = 8;          // list entries per thread whose copies are in flight together
constexpr int Q_RS = Q_WW + 3;   // LDS row pitch in words: image words c0/32 - 1 .. c0/32 + Q_WW, + 1 (odd: 11)
static_assert(Q_TH * Q_WW == Q_NT, "one tile word per thread");

template <int DJ>
__device__ __forceinline__ u32 qshift(const u32 (&a)[3]) {  // bits of the pixels (column + DJ) of the word a[1]
    if (DJ == 0) return a[1];
    if (DJ > 0) return __builtin_amdgcn_alignbit(a[2], a[1], DJ);
    return __builtin_amdgcn_alignbit(a[1], a[0], 32 + DJ);
}
// One tap of the parent rule for 32 pixels; the candidate r = q + (row of the arrays, DJ).  FWD: r must be live.
// code = tap t (forward, live pixels) or 8 | t (the negated tap, the others): the format tap_decode reads
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
                                               float *out_depth, int32_t *out_index, int *__restrict__ finfo,
                                               u32 *__restrict__ xlist, u32 *__restrict__ xptr, u8 *__restrict__ unres) {
    __shared__ u32 s_pl[6][Q_TH + 4][Q_RS];  // d bit 0, 1, 2, live, tie, in-image; rows r0-2 .. r0+TH+1
    __shared__ u32 s_code[4][Q_TH][Q_WW + 1];
    __shared__ u16 s_list[Q_TH * Q_TW];
    __shared__ u32 s_cnt[Q_NT / 64];
    __shared__ u32 s_unres[Q_NT];
    const int b = blockIdx.y, tid = threadIdx.x;
    if (!fflag[b]) return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * Q_TH, c0 = tx * Q_TW;
    const int wpr = Wp >> 2;  // 32-pixel words per plane row
    const size_t rowb = (size_t)b * H;
    // this thread's tile word
    const int trow = tid / Q_WW, tw = tid % Q_WW;
    const int gi = r0 + trow, gw = (c0 >> 5) + tw;  // image row, image word (32 px) column
    const bool tin = gi < H && gw < wpr;            // the word exists in the planes (its pixels beyond W are zero bits)
    u32 *ures = reinterpret_cast<u32 *>(unres + (rowb + min(gi, H - 1)) * Wp) + gw;
    // ---- the window's planes -> LDS: (TH + 4) rows x (WW + 2) words, item = (row, word), all five planes of an item
    // by the same thread (all loads of a thread are issued before its first LDS store)
    {
        constexpr int NIT = (Q_TH + 4) * (Q_WW + 2), PER = (NIT + Q_NT - 1) / Q_NT;
        u32 v[PER][6];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = tid + u * Q_NT;
            const int rr = k / (Q_WW + 2), w = k - rr * (Q_WW + 2);
            const int i = r0 - 2 + rr, wi = (c0 >> 5) - 1 + w;
            const bool in = k < NIT && i >= 0 && i < H && wi >= 0 && wi < wpr;
            const size_t ro = (rowb + min(max(i, 0), H - 1)) * Wp;
            const int wic = min(max(wi, 0), wpr - 1);  // clamped: unconditional loads
            const int up = min(W - wi * 32, 32);       // in-image columns of this word: [0, up)
            const u32 inimg = (!in || up <= 0) ? 0u : (up >= 32 ? 0xFFFFFFFFu : ((1u << up) - 1u));
#pragma unroll
            for (int p = 0; p < 5; ++p) v[u][p] = reinterpret_cast<const u32 *>(planes + p * plane_bytes + ro)[wic] & inimg;
            v[u][5] = inimg;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int k = tid + u * Q_NT;
            if (k < NIT) {
                const int rr = k / (Q_WW + 2), w = k - rr * (Q_WW + 2);
#pragma unroll
                for (int p = 0; p < 6; ++p) s_pl[p][rr][w] = v[u][p];
            }
        }
    }
    s_unres[tid] = 0;
    __syncthreads();
    const u32 mytie = s_pl[4][trow + 2][tw + 1];
    if (!__syncthreads_or(mytie != 0)) {  // no tie pixel in this tile
        if (tin) *ures = 0;
        return;
    }
    // ---- parent rule for this word (planes from LDS: rows trow .. trow + 4 of the window, words tw .. tw + 2)
    if (mytie) {
        auto ld3 = [&](int p, int row, u32 (&o)[3]) {
            const u32 *sp = &s_pl[p][row][tw];
            o[0] = sp[0]; o[1] = sp[1]; o[2] = sp[2];
        };
        const int qrow = trow + 2;
        const u32 b0 = s_pl[0][qrow][tw + 1], b1 = s_pl[1][qrow][tw + 1], b2 = s_pl[2][qrow][tw + 1];
        const u32 qlive = s_pl[3][qrow][tw + 1];
        u32 takenF = ~(mytie & qlive), takenB = ~(mytie & ~qlive);
        u32 C[4] = {0, 0, 0, 0};
        u32 a0[3], a1[3], a2[3], lv[3], vd[3];
        // cv2 tap order: (-2,-1) (-2,+1) (-1,-2) (-1,-1) (-1,0) (-1,+1) (-1,+2) (0,-1); backward = the negated offsets in
        // the same order.  The two chains are independent (live / non-live pixels), each keeps its order.
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
        for (int j = 0; j < 4; ++j) s_code[j][trow][tw] = C[j];
    }
    // ---- the tile's tie pixels as a list (tile row << 8 | tile column), so that the hops are spread evenly
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
            s_list[o++] = (u16)(trow << 8 | (tw * 32 + bit));
        }
    }
    __syncthreads();
    // ---- hops inside the tile, Q_EB list entries per thread at a time: first all the hops (LDS only), then all the
    // copies (independent loads), then the stores.
    // reads: pixels that are NOT tie pixels; writes: tie pixels -- the two sets never meet inside this kernel
    const int32_t *__restrict__ rd_i = out_index;
    const float *__restrict__ rd_f = out_depth;
    int32_t *__restrict__ wr_i = out_index;
    float *__restrict__ wr_f = out_depth;
    const size_t fo = (size_t)b * H * W;
    for (int e0 = 0; e0 < n_all; e0 += Q_EB * Q_NT) {  // block-uniform trip count
        int qoff[Q_EB], poff[Q_EB];
        u32 okm = 0, givem = 0;
#pragma unroll
        for (int u = 0; u < Q_EB; ++u) {
            const int e = e0 + u * Q_NT + tid;
            const bool have = e < n_all;
            bool solved = false;
            const int start = have ? (int)s_list[e] : 0;
            int r = start >> 8, c = start & 255;  // tile coordinates
            for (int hop = 0; have && hop < Q_TH + Q_TW; ++hop) {  // d falls with every hop: the bound is never reached
                const int w = c >> 5, bt = c & 31;
                const int code = (int)((s_code[0][r][w] >> bt) & 1u) | (int)((s_code[1][r][w] >> bt) & 1u) << 1 |
                                 (int)((s_code[2][r][w] >> bt) & 1u) << 2 | (int)((s_code[3][r][w] >> bt) & 1u) << 3;
                int di, dj;
                tap_decode(code, di, dj);
                r += di;
                c += dj;
                if (!((s_pl[4][r + 2][(c + 32) >> 5] >> ((c + 32) & 31)) & 1u)) {  // not a tie pixel: its label is the chain's
                    solved = true;
                    break;
                }
                if (r < 0 || r >= Q_TH || c < 0 || c >= Q_TW) break;  // a tie pixel of another tile: no code here
            }
            const int ei = min(max(r0 + r, 0), H - 1), ej = min(max(c0 + c, 0), W - 1);  // inside the image on consistent planes
            qoff[u] = have ? (r0 + (start >> 8)) * W + c0 + (start & 255) : 0;
            poff[u] = have ? ei * W + ej : 0;
            okm |= (have && solved) ? 1u << u : 0u;
            givem |= (have && !solved) ? 1u << u : 0u;
            if (have && !solved) atomicOr(&s_unres[(start >> 8) * Q_WW + ((start & 255) >> 5)], 1u << (start & 31));
        }
        if (out_index) {
            int32_t v[Q_EB];
#pragma unroll
            for (int u = 0; u < Q_EB; ++u) v[u] = rd_i[fo + poff[u]];
#pragma unroll
            for (int u = 0; u < Q_EB; ++u)
                if ((okm >> u) & 1u) wr_i[fo + qoff[u]] = v[u];
        }
        if (out_depth) {
            float v[Q_EB];
#pragma unroll
            for (int u = 0; u < Q_EB; ++u) v[u] = rd_f[fo + poff[u]];
#pragma unroll
            for (int u = 0; u < Q_EB; ++u)
                if ((okm >> u) & 1u) wr_f[fo + qoff[u]] = v[u];
        }
        // the others go on in another tile: k_tiesx follows them from (ei, ej).  One slot of the frame's list per pixel,
        // one atomic per wave.
        if (__any(givem != 0)) {
            const int lane = tid & 63;
            const int cnt = __popc(givem);
            int incl = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int t = __shfl_up(incl, off);
                if (lane >= off) incl += t;
            }
            int base = 0;
            if (lane == 63) base = atomicAdd(&finfo[b * FI_STRIDE + FI_NUNRES], incl);
            int o = __shfl(base, 63) + incl - cnt;
#pragma unroll
            for (int u = 0; u < Q_EB; ++u)
                if ((givem >> u) & 1u) {
                    xptr[fo + qoff[u]] = (u32)poff[u];
                    xlist[fo + o++] = (u32)qoff[u];
                }
        }
    }
    __syncthreads();
    if (tin) *ures = s_unres[tid];
}

// ------------------------------------------------------------------------------------------------
// k_tiesx: the tie pixels whose chain left their tile while still on tie pixels (k_ties listed them per frame; the
// list is short or empty).  One thread per listed pixel.  xptr[q] is the pixel where q's chain goes on; that pixel was
// either finished by its own tile (its bit in the "unresolved" plane is clear: copy its label and depth) or it is
// on the list itself and has its own xptr -- follow it.  A step is one memory round trip (the bit and the pointer are
// loaded together) and crosses a whole tile, whatever the chain's length inside the tiles.  Only memory that this
// kernel does not write is used to steer (xptr, the plane); labels and depths are read from finished pixels only.
// ------------------------------------------------------------------------------------------------
constexpr int XL_BLOCKS = 16;  // blocks per frame (grid-stride over the list)

__global__ __launch_bounds__(256) void k_tiesx(const u8 *__restrict__ unres, int Wp, const int *__restrict__ fflag,
                                               const int *__restrict__ finfo, const u32 *__restrict__ xlist,
                                               const u32 *__restrict__ xptr, int H, int W, float *out_depth,
                                               int32_t *out_index) {
    const int b = blockIdx.y;
    if (!fflag[b]) return;
    const int n = finfo[b * FI_STRIDE + FI_NUNRES];
    const size_t rowb = (size_t)b * H;
    const size_t fo = (size_t)b * H * W;
    const int32_t *__restrict__ rd_i = out_index;  // reads: finished pixels; writes: listed pixels
    const float *__restrict__ rd_f = out_depth;
    int32_t *__restrict__ wr_i = out_index;
    float *__restrict__ wr_f = out_depth;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += XL_BLOCKS * 256) {
        const u32 q = xlist[fo + e];
        u32 p = xptr[fo + q];
        for (int step = 0; step < H * W; ++step) {  // every step ends on a pixel nearer to the sources
            const u32 pi = p / (u32)W, pj = p - pi * (u32)W;
            const u32 open = (unres[(rowb + pi) * Wp + (pj >> 3)] >> (pj & 7u)) & 1u;
            const u32 nx = xptr[fo + p];  // meaningful only if open
            if (!open) break;
            p = min(nx, (u32)(H * W - 1));
        }
        if (out_index) wr_i[fo + q] = rd_i[fo + p];
        if (out_depth) wr_f[fo + q] = rd_f[fo + p];
    }
}
