// dtfill_rows.hpp -- k_colT, k_rows, k_fin, k_tiesx: the any-distance path of the l1_cv pass (argmin scans)
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
// Which rows: all of a frame k_frame routes here (fflag 2), or the rows flagged 1 of a window-kernel frame (fflag 1: rows
// too far from every source row, marked by k_frame up front; rows in which k_fused met a pixel beyond its halo; the sky's
// rows when k_fused had to call k_sky off).  A frame of k_pts (fflag 3) has its tiles in k_fin's launch (dtfill_pts.hpp) and
// passes through k_tiesx.
//
//   k_colT   srcbits -> per 32-row band and column: the band's source bits of that column (one word) and
//            the distance from the band's first / last row to the nearest source above / below the band:
//            0.25 B/px, everything a row needs to know about its columns; and the rank records (label_from_rec) k_fin turns a source into its label with.
//            k_sky's blocks (dtfill_sky.hpp) ride behind the column blocks of this launch.
//   k_rows   one block per image row: column distances from the band words, six packed-key scans
//            (kmin / kmax / upper-sources, left and right): d, the nearest source in column kmin (spix), "one nearest
//            source", live; stores the distance map, spix, and five bit planes (d mod 8, live, tie) for k_fin.
//   k_fin    one block per 32 x 256 tile: bit-sliced 5x5 parent rule on the planes for the words that hold a tie pixel,
//            step bytes, up to Q_HOPS hops per tie pixel inside the tile; every pixel takes the source of the pixel that ends
//            its chain; label (rank record + popcount) and depth (gather) of every pixel, written once as whole lines.  A chain
//            that leaves the tile on tie pixels, or is longer, is listed for k_tiesx with the pixel where it goes on.
//   k_tiesx  the listed pixels: follows the recorded links (one memory round trip per link, whatever the chain's length
//            inside the tiles) to a finished pixel and copies its label and depth.
// ================================================================================================

constexpr int GBIG = 16383;  // column distance when the column has no source (d >= 8192 <=> frame without sources)
constexpr int K_OFF = 8192;  // left keys carry value - column + K_OFF
constexpr int K_SH = 15;     // key = value << 15 | arg; arg = column << 2 | flags, or (8191 - column) << 2
constexpr u32 SPIX_NONE = 0xFFFFFFFFu;  // k_rows -> k_fin: no source in the frame
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
constexpr int COLT_KEEP_NB = 128;
__host__ __device__ inline size_t colT_lds(int nb) { return (size_t)nb * 64 * (2 * sizeof(u16) + (nb <= COLT_KEEP_NB ? sizeof(u32) : 0)); }

// s_lf: [nb][64] last source row of the band, [nb][64] first (0xFFFF: none); wd: the block's 64-pixel word column
__device__ __forceinline__ void colT_block(const u64 *__restrict__ srcbits, const int *__restrict__ fflag, int H,
                                           int W, int Wd, int nb, int CTP, uint2 *__restrict__ ct,
                                           const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                           uint4 *__restrict__ rec, int wd, int b, u16 *s_lf, u64 (*s_rowword)[64]) {
    const int lane = threadIdx.x & 63;
    const int ch = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    if (fflag && (!fflag[b] || fflag[b] == 3)) return;  // 3: k_pts's frame
    const int j = wd * 64 + lane;
    const bool real = wd < Wd;  // block-uniform: word columns beyond the image only write the "no source" padding
    u16 *s_last = s_lf, *s_first = s_lf + nb * 64;
    // up to 4096 rows the column's bits of every band fit into LDS next to them (colT_lds()): whole elements are stored
    const bool keep_t = nb <= COLT_KEEP_NB;
    u32 *s_t = reinterpret_cast<u32 *>(s_lf + 2 * nb * 64);
    // A wave takes 64 consecutive rows = two bands at a time; lane = row: ONE load per row brings its word (and what its
    // labels need), 64 ballots transpose the 64 x 64 bits, lane = column afterwards.
    for (int band = 2 * ch; band < nb; band += 2 * nwv) {
        const int row = band * 32 + lane;
        u64 w = 0;
        u32 base = 0;
        if (real && row < H) {
            const size_t wi = ((size_t)b * H + row) * Wd + wd;
            w = srcbits[wi];
            if (rec) base = rowbase_s[(size_t)b * H + row] + wpre_s[wi];
        }
        // transpose the 64 x 64 bits through LDS: every lane reads the 64 row words (one broadcast read each) and keeps
        // its column's bit of each -- lane = column afterwards
        s_rowword[ch][lane] = w;
        u32 t_lo = 0, t_hi = 0;  // the column's source bits of band / band + 1
#pragma unroll
        for (int rr = 0; rr < 32; ++rr) {
            t_lo |= (u32)((s_rowword[ch][rr] >> lane) & 1ull) << rr;
            t_hi |= (u32)((s_rowword[ch][rr + 32] >> lane) & 1ull) << rr;
        }
        if (rec && real && row < H) {
            // the rank record of this row's word (common.hpp: label_from_rec): lanes = consecutive rows -> one 1 KB run per store
            rec[((size_t)b * Wd + wd) * H + row] = make_uint4((u32)w, base, (u32)(w >> 32), base + (u32)__popc((u32)w));
        }
        if (keep_t) {  // the bits wait in LDS for their {up, down}: one 8-byte store per element at the end
            s_t[band * 64 + lane] = t_lo;
            if (band + 1 < nb) s_t[(band + 1) * 64 + lane] = t_hi;
        } else if (j < CTP) {
            ct[((size_t)b * nb + band) * CTP + j].x = t_lo;
            if (band + 1 < nb) ct[((size_t)b * nb + band + 1) * CTP + j].x = t_hi;
        }
        const int i0 = band * 32;
        s_last[band * 64 + lane] = t_lo ? (u16)(i0 + 31 - __clz((int)t_lo)) : (u16)0xFFFF;
        s_first[band * 64 + lane] = t_lo ? (u16)(i0 + __ffs((int)t_lo) - 1) : (u16)0xFFFF;
        if (band + 1 < nb) {
            s_last[(band + 1) * 64 + lane] = t_hi ? (u16)(i0 + 63 - __clz((int)t_hi)) : (u16)0xFFFF;
            s_first[(band + 1) * 64 + lane] = t_hi ? (u16)(i0 + 32 + __ffs((int)t_hi) - 1) : (u16)0xFFFF;
        }
    }
    __syncthreads();
    // The two carries are sequential over the bands: wave 0 walks down (distance from the band's FIRST row to the nearest
    // source above the band), wave 1 walks up (from its LAST row to the nearest source below).  Each overwrites the LDS entry
    // it has just consumed with its result, and after a barrier every (band, column) leaves as ONE 4-byte store of
    // {up, down} -- not as two 2-byte stores by two waves at different times, which tripled the kernel's HBM traffic.
    if (ch == 0) {
        int run = -1;
        for (int band = 0; band < nb; ++band) {
            const int up = run < 0 ? GBIG : min(band * 32 - run, GBIG);
            const int l = s_last[band * 64 + lane];
            s_last[band * 64 + lane] = (u16)up;
            run = l != 0xFFFF ? l : run;
        }
    } else if (ch == 1) {
        int run = -1;
        for (int band = nb - 1; band >= 0; --band) {
            const int dn = run < 0 ? GBIG : min(run - (band * 32 + 31), GBIG);
            const int f = s_first[band * 64 + lane];
            s_first[band * 64 + lane] = (u16)dn;
            run = f != 0xFFFF ? f : run;
        }
    }
    __syncthreads();
    if (j < CTP)
        for (int band = ch; band < nb; band += nwv) {
            const u32 ud = (u32)s_last[band * 64 + lane] | (u32)s_first[band * 64 + lane] << 16;
            if (keep_t)
                ct[((size_t)b * nb + band) * CTP + j] = make_uint2(s_t[band * 64 + lane], ud);
            else
                ct[((size_t)b * nb + band) * CTP + j].y = ud;
        }
}

struct SkyArgs {  // k_sky's blocks ride behind k_colT's (l1_cv with row flags): see dtfill_sky.hpp
    const int *finfo;
    const float *dt_src;
    float *out_dt, *out_depth;
    int32_t *out_index;
    int nstrips, nblocks;  // blocks = strips x row groups; 0: none
};
__device__ __forceinline__ void sky_body(unsigned char *s_sky, const int *__restrict__ finfo, int H, int W, const float *dt_src, float *out_dt,
                                         float *out_depth, int32_t *out_index, int strip, int rowgroup, int b);

__global__ __launch_bounds__(1024) void k_colT(const u64 *__restrict__ srcbits, const int *__restrict__ fflag, int H,
                                               int W, int Wd, int nb, int CTP, uint2 *__restrict__ ct,
                                               const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                               uint4 *__restrict__ rec, int ncolblocks, const SkyArgs sky, int frames_x) {
    extern __shared__ __attribute__((aligned(16))) u16 s_lf[];
    __shared__ u64 s_rowword[16][64];
    // frames_x: frames along grid x, a frame's blocks along y -- with a batch of a multiple of eight frames a frame's blocks then run
    // on one XCD: the sky's blocks read the two rows the window kernel left there, the column blocks the bit words (frames of up
    // to 512 rows: 17.9 against 21.5 us on the scan-line batch; the 1024-thread blocks of a 2048-row frame want all XCDs: 32 against
    // 60 us the other way round)
    const int b = frames_x ? blockIdx.x : blockIdx.y, blk = frames_x ? blockIdx.y : blockIdx.x;
    if (blk >= ncolblocks) {  // (block-uniform)
        const int k = blk - ncolblocks;
        sky_body(reinterpret_cast<unsigned char *>(s_lf), sky.finfo, H, W, sky.dt_src, sky.out_dt, sky.out_depth, sky.out_index, k % sky.nstrips,
                 k / sky.nstrips, b);
        return;
    }
    colT_block(srcbits, fflag, H, W, Wd, nb, CTP, ct, wpre_s, rowbase_s, rec, blk, b, s_lf, s_rowword);
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
// the same as a streaming store: for outputs nothing in the pass reads again (vector types go out component-wise: the
// builtin takes scalars and native vectors only)
template <typename T>
__device__ __forceinline__ void st_off_nt(void *base, u32 byte_off, T v) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    char *p = reinterpret_cast<char *>(base) + byte_off;
    if constexpr (sizeof(T) == 16) {
        f4 t;
        __builtin_memcpy(&t, &v, 16);
        __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(p));
    } else if constexpr (sizeof(T) == 8) {
        f2 t;
        __builtin_memcpy(&t, &v, 8);
        __builtin_nontemporal_store(t, reinterpret_cast<f2 *>(p));
    } else {
        float t;
        __builtin_memcpy(&t, &v, 4);
        __builtin_nontemporal_store(t, reinterpret_cast<float *>(p));
    }
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
// fflag[b] == 1: only the rows k_fused marked in rowflag are redone ("core" rows).  Their chains run at most Q_HOPS hops of
// at most two rows inside k_fin, every step byte on the way needs the planes of two more rows: k_rows computes the rows
// within R_MARGIN of a core row; everything further away keeps k_fused's results.
constexpr int R_MARGIN = 10;

constexpr int R_RPB = 4;  // rows per block (one after the other, a grid apart) where a row is a single wave's: a frame that is none of
                          // these kernels' then costs a quarter of the block dispatches (NYU: 30 720 blocks of 64 threads, 3.4 us)

template <int PPL>
__device__ __forceinline__ void rows_body(
    const uint2 *__restrict__ ct, int CTP, int ff, int H, int W, int nb, int Wp,
    u8 *__restrict__ planes, size_t plane_bytes, float *__restrict__ out_dt, u32 *__restrict__ spix_out, int ovec,
    const u32 *__restrict__ rowflag, const int *__restrict__ finfo_sky, int i, int b, u32 (*__restrict__ s_tot)[6],
    u32 (*__restrict__ s_bits)[R_MAXPW + 1]) {
    static_assert(PPL == 8 || PPL == 10, "loads and stores below are written for 8 or 10 columns per lane");
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwv = blockDim.x >> 6;
    if (ff == 1) {    // only near a row k_fused could not finish
        const int rr = i - R_MARGIN + lane;
        const int sky_live = finfo_sky[b * FI_STRIDE + FI_SKY];
        if (!__any(lane <= 2 * R_MARGIN && rr >= 0 && rr < H && row_is_anydist(rowflag[(size_t)b * H + rr], sky_live))) return;
    }
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
    u32 spix[PPL];  // the nearest source in column kmin as row << 16 | column; SPIX_NONE: the frame has no source
    u32 acc012 = 0, acc34 = 0;  // bit planes of the lane's pixels: plane p of pixel q at bit 10 p + q
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
        fd[q] = none ? 8192.0f : (float)d;  // float(INIT_DIST0 * 2^-16) == 8192.0f
        // the nearest source in column kmin (for a tie pixel: one of its nearest sources; k_fin takes another pixel's)
        const int gk = (int)d - (int)__builtin_amdgcn_sad_u16((u32)j, kmin, 0u);  // d - |j - kmin| (both < 2^13)
        const int sg = (int)((bmin & 1u) << 1) - 1;                              // below: +1, above: -1
        const int si = min(max(i + sg * gk, 0), H - 1);  // clamps: never taken on a correct frame
        const u32 sj = min(kmin, (u32)(W - 1));
        spix[q] = none ? SPIX_NONE : (u32)si << 16 | sj;  // the source pixel: row << 16 | column
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
    // ---- stores (frame bases scalar, 32-bit byte offsets)
    const bool full = idx0 + PPL <= W;
    const u32 rob = ((u32)i * (u32)W + (u32)idx0) << 2;
    float *dt_f = out_dt ? out_dt + fo : nullptr;
    u32 *sp_f = spix_out ? spix_out + fo : nullptr;
    // ovec bit 1: the distance map (never read again in this pass) leaves through LDS as whole 128-byte lines with streaming
    // stores after the barrier below -- a lane's own 32 or 40 bytes are only part of a line, and the L2 would have to merge them
    extern __shared__ __attribute__((aligned(16))) float s_dtrow[];
    const bool dt_lds = (ovec & 2) && dt_f;
    if (dt_lds) {
#pragma unroll
        for (int q = 0; q < PPL; ++q)
            if (idx0 + q < W) s_dtrow[idx0 + q] = fd[q];
        dt_f = nullptr;  // (the stores below skip it)
    }
    if ((ovec & 1) && full) {
        if (PPL == 8) {
            if (dt_f) {
                st_off(dt_f, rob, make_float4(fd[0], fd[1], fd[2], fd[3]));
                st_off(dt_f, rob + 16, make_float4(fd[4], fd[5], fd[6], fd[7]));
            }
            if (sp_f) {
                st_off(sp_f, rob, make_uint4(spix[0], spix[1], spix[2], spix[3]));
                st_off(sp_f, rob + 16, make_uint4(spix[4], spix[5], spix[6], spix[7]));
            }
        } else {
#pragma unroll
            for (int q = 0; q < PPL; q += 2) {
                if (dt_f) st_off(dt_f, rob + 4 * q, make_float2(fd[q], fd[q + 1]));
                if (sp_f) st_off(sp_f, rob + 4 * q, make_uint2(spix[q], spix[q + 1]));
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < PPL; ++q) {
            if (idx0 + q >= W) continue;
            if (dt_f) st_off(dt_f, rob + 4 * q, fd[q]);
            if (sp_f) st_off(sp_f, rob + 4 * q, spix[q]);
        }
    }
    __syncthreads();
    if (dt_lds) {
        float *drow = out_dt + fo + (size_t)i * W;
        for (int k = threadIdx.x * 4; k < W; k += blockDim.x * 4)  // W % 32 == 0
            st_off_nt(drow, (u32)k << 2, *reinterpret_cast<const float4 *>(&s_dtrow[k]));
    }
    for (int k = threadIdx.x; k < 5 * wpr; k += blockDim.x) {  // the row's plane words leave as whole words
        const int p = k / wpr, w = k - p * wpr;
        reinterpret_cast<u32 *>(planes + p * plane_bytes + ((size_t)b * H + i) * Wp)[w] = s_bits[p][w];
    }
}

template <int PPL, int MAXT, bool MULTI>  // MAXT: 256 (rows of up to 4 waves; 3 waves per SIMD) or 1024 (any row the shape limit allows)
__global__ __launch_bounds__(MAXT, MAXT == 256 ? 4 : 1) void k_rows(  // MULTI: rpb rows per block (the loop costs 17 VGPR: its own instance)
   
    const uint2 *__restrict__ ct, int CTP, const int *__restrict__ fflag, int H, int W, int nb, int Wp,
    u8 *__restrict__ planes, size_t plane_bytes, float *__restrict__ out_dt, u32 *__restrict__ spix_out, int ovec,
    const u32 *__restrict__ rowflag, const int *__restrict__ finfo_sky, int rpb) {
    __shared__ u32 s_tot[R_MAXWV][6];
    __shared__ u32 s_bits[5][R_MAXPW + 1];
    const int b = blockIdx.y;
    const int ff = fflag[b];
    if (!ff || ff == 3) return;  // block-uniform (3: k_pts's frame)
    if (!MULTI) {
        rows_body<PPL>(ct, CTP, ff, H, W, nb, Wp, planes, plane_bytes, out_dt, spix_out, ovec, rowflag, finfo_sky, (int)blockIdx.x, b, s_tot, s_bits);
        return;
    }
    // rows blockIdx.x, + gridDim.x, ...: rpb of them, a grid apart so that the rows around a handed-on row stay spread over the
    // blocks
#pragma unroll 1
    for (int k = 0; k < rpb; ++k) {
        const int i = (int)blockIdx.x + k * (int)gridDim.x;
        if (i >= H) break;
        rows_body<PPL>(ct, CTP, ff, H, W, nb, Wp, planes, plane_bytes, out_dt, spix_out, ovec, rowflag, finfo_sky, i, b, s_tot, s_bits);
        __syncthreads();  // the next row reuses the block's LDS
    }
}

// ------------------------------------------------------------------------------------------------
// k_fin: one block (256 threads) per 32 x 256 tile, one thread per 32-pixel word of it: tie pixels get their source,
// then every pixel of the tile gets its label and depth, written as whole 128-byte runs (nothing is patched in
// memory afterwards except the few pixels of k_tiesx).
//   1. The tile's planes (d mod 8, live, tie) with a 2-cell ring go to LDS once.
//   2. A thread whose word holds a tie pixel applies the 5x5 parent rule BIT-SLICED, 32 pixels per operation, on
//      d mod 8: a tap of weight w <= 3 matches iff d(r) + w == d(q), and |d(r) - d(q)| <= w makes that exact mod 8.
//   3. Its tie pixels hop through the tile's code planes in LDS until they stand on a pixel that is not a tie pixel
//      (one nearest source, or a source) and take over that pixel's source (k_rows left every pixel's nearest
//      source in column kmin in spix).  A chain that leaves the tile while still on tie pixels records where it
//      goes on (xptr, the "unresolved" plane, the frame's list) for k_tiesx.
//   4. label = 1 + raster rank of the source (cv2's label init), depth = depth_list[label - 1] with numpy's index
//      rules (tools.py:24-26); three loads + a gather per pixel, eight pixels' worth in flight per thread.
// ------------------------------------------------------------------------------------------------
constexpr int Q_TH = 32, Q_TW = 256;
constexpr int Q_NT = 256;
constexpr int Q_WW = Q_TW / 32;  // tile words per row
constexpr int Q_HOPS = 4;        // hops a tie pixel takes inside k_fin before it is handed to k_tiesx
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

// the pixels (of 32) whose code planes C spell tap CODE: their step code (di + 2) << 3 | (dj + 2) goes into the planes E
template <int CODE>
__device__ __forceinline__ void step_tap(const u32 (&C)[4], u32 mytie, u32 (&E)[6]) {
    constexpr int t = CODE & 7;
    constexpr int DI = (CODE & 8) ? -TAP_DI(t) : TAP_DI(t), DJ = (CODE & 8) ? -TAP_DJ(t) : TAP_DJ(t);
    u32 sel = mytie;
#pragma unroll
    for (int j = 0; j < 4; ++j) sel &= (CODE & (1 << j)) ? C[j] : ~C[j];
    constexpr int ENC = (DI + 2) << 3 | (DJ + 2);
#pragma unroll
    for (int j = 0; j < 6; ++j)
        if (ENC & (1 << j)) E[j] |= sel;
}

// LDS of a k_fin block (carved from the kernel's buffer: the kernel, in dtfill_pts.hpp, also runs k_pts's tiles)
constexpr size_t FIN_OFF_BYTE = sizeof(u32) * 6 * (Q_TH + 4) * Q_RS, FIN_OFF_UNRES = FIN_OFF_BYTE + Q_TH * Q_TW,
                 FIN_OFF_CNT = FIN_OFF_UNRES + sizeof(u32) * Q_NT, FIN_LDS_OWN = FIN_OFF_CNT + sizeof(u32) * (Q_NT / 64);
static_assert(FIN_OFF_BYTE % 4 == 0, "the step bytes are read as words");

__device__ __forceinline__ void fin_body(
    unsigned char *__restrict__ s_raw,
    const u8 *__restrict__ planes, size_t plane_bytes, int Wp, const int *__restrict__ fflag, int H, int W, int Wd,
    int tiles_x, const u32 *__restrict__ spix_ws, const float *__restrict__ x, const uint4 *__restrict__ rec,
    const float *__restrict__ vlist,
    float *__restrict__ out_depth, int32_t *__restrict__ out_index, int *__restrict__ frame_status,
    int *__restrict__ finfo, u32 *__restrict__ xlist, u32 *__restrict__ xptr, u8 *__restrict__ unres, int vec,
    const DepthEpilogue ep, float *__restrict__ dscratch, const u32 *__restrict__ rowflag) {
    u32(*s_pl)[Q_TH + 4][Q_RS] = reinterpret_cast<u32(*)[Q_TH + 4][Q_RS]>(s_raw);  // d bit 0, 1, 2, live, tie, in-image; rows r0-2 .. r0+TH+1
    u8(*s_byte)[Q_TW] = reinterpret_cast<u8(*)[Q_TW]>(s_raw + FIN_OFF_BYTE);  // per tile pixel: step to its parent ((di + 2) << 3 | (dj + 2); 18 = none)
    u32 *s_unres = reinterpret_cast<u32 *>(s_raw + FIN_OFF_UNRES);            // per tile word: tie pixels that k_tiesx finishes
    u32 *s_cnt = reinterpret_cast<u32 *>(s_raw + FIN_OFF_CNT);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int ff = fflag[b];
    if (!ff) return;  // (3, k_pts's frame: the kernel has taken that turn)
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * Q_TH, c0 = tx * Q_TW;
    // the tile's rows that are redone: all of them, or (ff == 1) those k_fused marked; the others keep k_fused's results
    u32 coremask = 0xFFFFFFFFu;
    if (ff == 1) {
        const int l = tid & 63;
        const int sky_live = finfo[b * FI_STRIDE + FI_SKY];
        coremask = (u32)__ballot(l < Q_TH && r0 + l < H && row_is_anydist(rowflag[(size_t)b * H + min(r0 + l, H - 1)], sky_live));
        if (!coremask) return;  // block-uniform: every wave computes the same mask
    }
    const int wpr = Wp >> 2;  // 32-pixel words per plane row
    const size_t rowb = (size_t)b * H;
    const u32 fo = (u32)b * (u32)(H * W);
    const u32 *sp_f = spix_ws + fo;
    // this thread's tile word
    const int trow = tid / Q_WW, tw = tid % Q_WW;
    const int gi = r0 + trow, gw = (c0 >> 5) + tw;  // image row, image word (32 px) column
    const bool tin = gi < H && gw < wpr;            // the word exists in the planes (its pixels beyond W are zero bits)
    // phase 4 mapping: a wave takes tile rows ewave, ewave + waves, ..., a lane four consecutive pixels of the row
    const int elane = tid & 63, ewave = tid >> 6;
    const int lw = elane >> 3, lb = (elane & 7) * 4;  // the lane's word in the row, first bit in it
    const int col = c0 + elane * 4;
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
    __syncthreads();
    const u32 mytie = s_pl[4][trow + 2][tw + 1];
    s_unres[tid] = 0;
    // block-uniform: some tie pixel in this tile (a word per wave: __syncthreads_or would take 256 bytes of LDS of its own, the
    // sixth of five blocks' worth on a CU)
    if ((tid & 63) == 0) s_cnt[tid >> 6] = 0;
    if (mytie) s_cnt[tid >> 6] = 1;  // (same-wave stores to one address: any of them lands)
    __syncthreads();
    bool any_tie = false;
#pragma unroll
    for (int w = 0; w < Q_NT / 64; ++w) any_tie = any_tie || s_cnt[w] != 0;
    if (any_tie) {
        u32 C[4] = {0, 0, 0, 0};
        u32 E[6] = {0, ~0u, 0, 0, ~0u, 0};  // bit planes of the step code (di + 2) << 3 | (dj + 2); 18 = (0, 0) where no tie pixel
        // ---- parent rule for this word (planes from LDS: rows trow .. trow + 4 of the window, words tw .. tw + 2)
        if (mytie) {
            auto ld3 = [&](int p, int row, u32 (&o)[3]) {
                const u32 *q3 = &s_pl[p][row][tw];
                o[0] = q3[0]; o[1] = q3[1]; o[2] = q3[2];
            };
            const int qrow = trow + 2;
            const u32 b0 = s_pl[0][qrow][tw + 1], b1 = s_pl[1][qrow][tw + 1], b2 = s_pl[2][qrow][tw + 1];
            const u32 qlive = s_pl[3][qrow][tw + 1];
            u32 takenF = ~(mytie & qlive), takenB = ~(mytie & ~qlive);
            u32 a0[3], a1[3], a2[3], lv[3], vd[3];
            // cv2 tap order: (-2,-1) (-2,+1) (-1,-2) (-1,-1) (-1,0) (-1,+1) (-1,+2) (0,-1); backward = the negated offsets
            // in the same order.  The two chains are independent (live / non-live pixels), each keeps its order.
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
            // every tie pixel's step code, still bit-sliced: tap t was chosen where the code planes spell t
            {
                E[1] = E[4] = ~mytie;
                step_tap<0>(C, mytie, E);
                step_tap<1>(C, mytie, E);
                step_tap<2>(C, mytie, E);
                step_tap<3>(C, mytie, E);
                step_tap<4>(C, mytie, E);
                step_tap<5>(C, mytie, E);
                step_tap<6>(C, mytie, E);
                step_tap<7>(C, mytie, E);
                step_tap<8>(C, mytie, E);
                step_tap<9>(C, mytie, E);
                step_tap<10>(C, mytie, E);
                step_tap<11>(C, mytie, E);
                step_tap<12>(C, mytie, E);
                step_tap<13>(C, mytie, E);
                step_tap<14>(C, mytie, E);
                step_tap<15>(C, mytie, E);
            }
        }
        // un-slice into one byte per pixel, four pixels per step: ((nibble * 0x00204081) & 0x01010101) spreads a nibble's
        // bits to the low bits of four bytes (24-bit multiplies)
        {
            u32 *brow = reinterpret_cast<u32 *>(&s_byte[trow][tw * 32]);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                u32 v = 0;
#pragma unroll
                for (int j = 0; j < 6; ++j) v |= (mul_u24_opaque((E[j] >> (4 * g)) & 0xFu, 0x00204081u) & 0x01010101u) << j;
                brow[g] = v;
            }
        }
    }
    __syncthreads();  // s_byte is complete (tiles without a tie pixel: not written, not read)
    // ---- 4. label, depth, stores.  A wave takes tile rows ewave, ewave + waves, ...; a lane four consecutive pixels.
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const float *x_f = x + fo, *vl_f = vlist + fo;
    const uint2 *rec_f = reinterpret_cast<const uint2 *>(rec + (size_t)b * Wd * H);
    // the depth output may drop the first ep.row0 rows (frames are then H - row0 rows apart); the depths of the dropped
    // rows go to a scratch frame instead, where k_tiesx finds them if a handed-on chain ends there
    float *dp_f = out_depth ? out_depth + (size_t)b * (H - ep.row0) * W : nullptr;
    float *ds_f = dscratch + fo;
    const u32 dcrop = (u32)(ep.row0 * W) << 2;
    int32_t *ix_f = out_index ? out_index + fo : nullptr;
    bool bad = false;
    constexpr int NRW = Q_TH / (Q_NT / 64);  // rows per wave (8)
    // Every pixel takes the source k_rows found for the pixel that ends its chain: itself unless it is a tie pixel; a tie
    // pixel hops along the step bytes until it stands on a pixel that is not a tie pixel (ONE hop for nine in ten).  A
    // chain that leaves the tile while still on tie pixels, or is still on one after Q_HOPS hops, is handed to k_tiesx
    // with the pixel where it goes on (that pixel's own chain is shorter: k_tiesx follows such links to their end).
    u32 esp[NRW][4];
    u32 umask = 0;  // bit 4 * it + u: handed to k_tiesx
    auto is_tie = [&](int r, int c) -> bool {  // tile coordinates, ring included
        return (s_pl[4][r + 2][(c + 32) >> 5] >> ((c + 32) & 31)) & 1u;
    };
#pragma unroll
    for (int it = 0; it < NRW; ++it) {
        const int rr = ewave + (Q_NT / 64) * it;
        const int i = r0 + rr;
        const u32 inm = ((coremask >> rr) & 1u) ? (s_pl[5][rr + 2][lw + 1] >> lb) & 15u : 0u;  // a row that is not redone: as if outside
        const u32 b4 = any_tie ? *reinterpret_cast<const u32 *>(&s_byte[rr][elane * 4]) : 0x12121212u;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 bb = (b4 >> (8 * u)) & 63u;
            int er = rr + (int)(bb >> 3) - 2, ec = elane * 4 + u + (int)(bb & 7u) - 2;  // one hop (none: step (0, 0))
            if (bb != 18u && ((inm >> u) & 1u) && is_tie(er, ec)) {  // rare: a chain of more than one hop
                bool open = true;
                for (int hop = 1; hop < Q_HOPS; ++hop) {
                    if (er < 0 || er >= Q_TH || ec < 0 || ec >= Q_TW) break;  // a tie pixel of another tile: no step here
                    const u32 b2 = s_byte[er][ec] & 63u;
                    er += (int)(b2 >> 3) - 2;
                    ec += (int)(b2 & 7u) - 2;
                    if (!is_tie(er, ec)) {
                        open = false;
                        break;
                    }
                }
                if (open) {  // (er, ec) is a tie pixel: keeps k_rows' source until k_tiesx overwrites label and depth
                    umask |= 1u << (4 * it + u);
                    const int ei = min(max(r0 + er, 0), H - 1), ej = min(max(c0 + ec, 0), W - 1);
                    xptr[fo + (u32)(i * W + col + u)] = (u32)(ei * W + ej);
                    atomicOr(&s_unres[rr * Q_WW + lw], 1u << (lb + u));
                    er = rr;
                    ec = elane * 4 + u;
                }
            }
            const u32 off = (u32)(min(max(r0 + er, 0), H - 1) * W + min(max(c0 + ec, 0), W - 1));
            esp[it][u] = ((inm >> u) & 1u) ? ld_off<u32>(sp_f, off << 2) : SPIX_NONE;
        }
    }
    if (any_tie) {
        // the handed-on pixels join the frame's list: block-wide count, ONE atomic, then every thread writes its own
        const int cu = __popc(umask);
        int incl = cu;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (elane >= off) incl += t;
        }
        if (elane == 63) s_cnt[ewave] = (u32)incl;
        __syncthreads();  // also: s_unres is complete
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < Q_NT / 64; ++w) {
            pre += w < ewave ? (int)s_cnt[w] : 0;
            all += (int)s_cnt[w];
        }
        if (all) {  // block-uniform
            __syncthreads();
            if (tid == 0) s_cnt[0] = (u32)atomicAdd(&finfo[b * FI_STRIDE + FI_NUNRES], all);
            __syncthreads();
            u32 o = s_cnt[0] + (u32)(pre + incl - cu);
            u32 m = umask;
            while (m) {
                const int k = __ffs((int)m) - 1;
                m &= m - 1;
                xlist[fo + o++] = (u32)((r0 + ewave + (Q_NT / 64) * (k >> 2)) * W + col + (k & 3));
            }
        }
    }
    if (tin) reinterpret_cast<u32 *>(unres + (rowb + gi) * Wp)[gw] = s_unres[tid];
    // label from the source's rank record, depth = depth_list[label - 1] = x[source] when the masks agree: two gathers per pixel,
    // all 32 pixels of the lane in flight together
    int lab[NRW][4];
    float val[NRW][4];
    u32 nonem = 0;
#pragma unroll
    for (int it = 0; it < NRW; ++it)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 v = esp[it][u];
            const bool none = v == SPIX_NONE;
            nonem |= none ? 1u << (4 * it + u) : 0u;
            // v = source row << 16 | column (the clamps only make sure that a logic error could never become a wild access)
            const int si = none ? 0 : min((int)(v >> 16), H - 1), sj = none ? 0 : min((int)(v & 0xFFFFu), W - 1);
            lab[it][u] = ix_f || misaligned ? label_from_rec(rec_f, H, si, sj) : 0;
            val[it][u] = dp_f ? ld_off<float>(x_f, (u32)(__umul24((u32)si, (u32)W) + (u32)sj) << 2) : 0.0f;
        }
#pragma unroll
    for (int it = 0; it < NRW; ++it) {
        const int rr = ewave + (Q_NT / 64) * it;
        const u32 inm = ((coremask >> rr) & 1u) ? (s_pl[5][rr + 2][lw + 1] >> lb) & 15u : 0u;  // in-image bits of the four pixels
        const u32 pixb = ((u32)min(r0 + rr, H - 1) * (u32)W + (u32)col) << 2;
#pragma unroll
        for (int u = 0; u < 4; ++u) lab[it][u] = ((nonem >> (4 * it + u)) & 1u) ? 0 : lab[it][u];
        if (dp_f && (misaligned || ((nonem >> (4 * it)) & 15u))) {  // rare: numpy's index rules (tools.py:26)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool none = (nonem >> (4 * it + u)) & 1u;
                int idx = lab[it][u] - 1;
                if (idx < 0) idx += nval;  // numpy: index -1 wraps to the last element
                const bool oob = idx < 0 || idx >= nval;
                bad |= oob && ((inm >> u) & 1u);
                // label 0 with agreeing masks means nval == nsrc == 0: out of bounds; so an in-bounds gather of a
                // source-less frame reads the value list
                val[it][u] = oob ? nanf("") : ((misaligned || none) ? vl_f[idx] : val[it][u]);
            }
        }
        const bool kept = pixb >= dcrop;  // wave-uniform: a wave holds one row
        if (dp_f && kept) {
#pragma unroll
            for (int u = 0; u < 4; ++u) val[it][u] = depth_epilogue(val[it][u], ep);
        }
        float *dst = kept ? dp_f : ds_f;
        const u32 dpix = kept ? pixb - dcrop : pixb;
        if (vec && inm == 15u) {
            if (ix_f) st_off_nt(ix_f, pixb, make_int4(lab[it][0], lab[it][1], lab[it][2], lab[it][3]));
            if (dp_f) st_off_nt(dst, dpix, make_float4(val[it][0], val[it][1], val[it][2], val[it][3]));
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!((inm >> u) & 1u)) continue;
                if (ix_f) st_off(ix_f, pixb + 4 * u, lab[it][u]);
                if (dp_f) st_off(dst, dpix + 4 * u, val[it][u]);
            }
        }
    }
    if (bad) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
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
                                               int32_t *out_index, const DepthEpilogue ep,
                                               const float *__restrict__ dscratch, const u32 *__restrict__ rowflag) {
    const int b = blockIdx.y;
    const int ff = fflag[b];
    if (!ff) return;
    const int n = finfo[b * FI_STRIDE + FI_NUNRES];
    const size_t rowb = (size_t)b * H;
    const size_t fo = (size_t)b * H * W;
    const int32_t *__restrict__ rd_i = out_index;  // reads: finished pixels; writes: listed pixels
    int32_t *__restrict__ wr_i = out_index;
    // depth output: rows [ep.row0, H) of every frame; the finished pixels of the dropped rows keep theirs in dscratch
    const size_t fod = (size_t)b * (H - ep.row0) * W;
    const u32 dcrop = (u32)(ep.row0 * W);
    const float *__restrict__ rd_f = out_depth;
    float *__restrict__ wr_f = out_depth;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < n; e += XL_BLOCKS * 256) {
        const u32 q = xlist[fo + e];
        u32 p = xptr[fo + q];
        for (int step = 0; step < H * W; ++step) {  // every step ends on a pixel nearer to the sources
            const u32 pi = p / (u32)W, pj = p - pi * (u32)W;
            // a row that was not redone holds k_fused's finished pixels (and no "unresolved" bits of this pass)
            const bool redone = ff != 1 || row_is_anydist(rowflag[rowb + pi], finfo[b * FI_STRIDE + FI_SKY]);
            const u32 open = (unres[(rowb + pi) * Wp + (pj >> 3)] >> (pj & 7u)) & 1u;
            const u32 nx = xptr[fo + p];  // meaningful only if open
            if (!redone || !open) break;
            p = min(nx, (u32)(H * W - 1));
        }
        if (out_index) wr_i[fo + q] = rd_i[fo + p];
        if (out_depth && q >= dcrop)
            wr_f[fod + q - dcrop] = p >= dcrop ? rd_f[fod + p - dcrop] : depth_epilogue(dscratch[fo + p], ep);
    }
}
