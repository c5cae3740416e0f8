// dtfill_fused.hpp -- k_fused<FR>: the bit-sliced LDS-window kernel of the l1_cv pass
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_fused: one workgroup (4 waves) per window = tile (<= 96 x 160) + halo FR, bit-sliced.
//
// Lane (r, half) owns half of window row r as three 32-bit words per bit plane (bit c%32 of word c/32 = window
// column c).  Level-synchronous form of the identity of DESIGN.md section 2 (checked in tests/parallel_model.py):
// with E_t = {d == t} and L_t = live pixels of E_t (E_0 = L_0 = sources),
//   E_t = dilate4(D_{t-1}) & ~D_{t-1} & in-image
//   forward tap T = (di,dj,w) offers   shift(L_{t-w}, di, dj)   to the pixels of E_t; L_t = those offered any
//   backward tap (negated offset)      candidates of level t-w for the pixels of E_t \ L_t
//   the FIRST tap in cv2 order wins (taken-mask chain); the winning step is recorded in six "code planes"
//   holding the bits of enc = (di+2)<<3 | (dj+2)  (sources: enc 18 = step (0,0)).
// The level loop runs the forward taps (they define L_t) and ORs E_t into three planes of d mod 8; the backward
// taps run ONCE after it for all levels, on those planes (bwd_tap).  A horizontal shift of a row is one
// v_alignbit per word; rows r-2..r+2 of the previous three levels come from a 4-slot LDS ring (one barrier per
// level).  Levels stop at FR, when a level is empty, or when every tile pixel is decided.
// Then every lane un-slices its row, 4 pixels per step, into the byte array s_par (2 * enc; F_NONE =
// undecided), which reuses the ring's memory, and the tile pixels walk to their sources in lock-step:
// the byte is the byte offset of the step's s_par displacement in a 64-entry int16 table (s_tab), so a
// hop is two LDS reads and one add; d is |drow| + |dcol| to the root.
// LDS: 28.9 KB ring/s_par + 8 KB rank records.
// ------------------------------------------------------------------------------------------------
constexpr int F_WHM = 128;  // window rows
constexpr int F_WWM = 192;  // window columns = 6 words
constexpr int F_NT = 256;   // lane = (row, half): waves 0-1 own words 0..2 of rows 0..127, waves 2-3 words 3..5
constexpr int F_P = 196;               // s_par row pitch: 49 dwords (odd) -> lane-per-row dword stores are conflict-free
constexpr int F_NWD = 6;               // 32-bit words per window row
constexpr int F_HW = 3;                // words per lane
constexpr int F_RS = 7;                // ring row stride in words: 6 + one zero pad (odd: conflict-free; the pad is
                                       // also the zero "word -1" of the next row and "word 6" of this one)
constexpr int F_RROWS = F_WHM + 4;     // ring rows: 2 zero rows above and below the window
constexpr int F_RPLANE = F_RROWS * F_RS;
constexpr int F_RING = 1 + 4 * 2 * F_RPLANE;  // one leading zero word, then [slot][plane E/L][row][7]
constexpr int F_EB = 8;                // tile pixels per lane walked in lock-step
// byte code of a decided pixel: 2 * enc (a source: 2 * 18 = step (0,0))
constexpr int F_NONE = 2 * 63;         // byte code of an undecided pixel: table entry 63, also step (0,0)
static_assert(F_WWM == 32 * F_NWD, "window width must be six words");
static_assert(F_RING * 4 >= F_WHM * F_P + 256 * 8, "s_par and the un-slicing table must fit in the ring's memory");
static_assert((F_WHM * F_P) % 8 == 0 && F_WHM * F_P >= 4 * (1 + 4 * F_RPLANE), "the table is 8-byte aligned and lies behind the planes of P1b");

#define ENC_F(t) (((TAP_DI(t) + 2) << 3) | (TAP_DJ(t) + 2))

// a[0..4] = the lane's three words a[1..3] with their left / right neighbour words; word i (0..2) of the
// row shifted so that result[c] = row[c + DJ]
template <int DJ>
__device__ __forceinline__ u32 hshift(const u32 (&a)[5], int i) {
    if (DJ == 0) return a[i + 1];
    if (DJ > 0) return __builtin_amdgcn_alignbit(a[i + 2], a[i + 1], DJ);
    return __builtin_amdgcn_alignbit(a[i + 1], a[i], 32 + DJ);
}

// one tap of the first-match chain: cand = shift(src, DJ); winners get the bits of ENC in the code planes
template <int DJ, int ENC>
__device__ __forceinline__ void tap_step(const u32 (&src)[5], u32 (&taken)[F_HW], u32 (&C)[6][F_HW]) {
#pragma unroll
    for (int i = 0; i < F_HW; ++i) {
        const u32 cand = hshift<DJ>(src, i);
        const u32 sel = cand & ~taken[i];
        taken[i] |= cand;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (ENC & (1 << j)) C[j][i] |= sel;
    }
}

// one BACKWARD tap, evaluated after the level loop for all levels at once: the candidate r = q + (row, DJ) matches
// iff it is decided and d(r) + WGT == d(q).  Only d mod 8 is kept (planes a0..a2 / b0..b2): d is an exact L1
// distance field, so |d(r) - d(q)| <= WGT <= 3 and d(r) + WGT - d(q) lies in [0, 6]: zero iff zero mod 8.
template <int DJ, int WGT, int ENC>
__device__ __forceinline__ void bwd_tap(const u32 (&a0)[5], const u32 (&a1)[5], const u32 (&a2)[5], const u32 (&dv)[5],
                                        const u32 (&b0)[F_HW], const u32 (&b1)[F_HW], const u32 (&b2)[F_HW],
                                        u32 (&taken)[F_HW], u32 (&C)[6][F_HW]) {
#pragma unroll
    for (int i = 0; i < F_HW; ++i) {
        const u32 x0 = hshift<DJ>(a0, i), x1 = hshift<DJ>(a1, i), x2 = hshift<DJ>(a2, i);
        u32 s0, s1, s2;  // (d(r) + WGT) mod 8
        if (WGT == 1) {
            s0 = ~x0; s1 = x1 ^ x0; s2 = x2 ^ (x1 & x0);
        } else if (WGT == 2) {
            s0 = x0; s1 = ~x1; s2 = x2 ^ x1;
        } else {
            s0 = ~x0; s1 = ~(x1 ^ x0); s2 = x2 ^ (x1 | x0);
        }
        const u32 m = ~((s0 ^ b0[i]) | (s1 ^ b1[i]) | (s2 ^ b2[i])) & hshift<DJ>(dv, i);
        const u32 sel = m & ~taken[i];
        taken[i] |= m;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (ENC & (1 << j)) C[j][i] |= sel;
    }
}

// 24-bit multiply the optimiser cannot see through: written as a plain product it folds the following shift
// into the constant, which then no longer fits 24 bits and becomes a quarter-rate v_mul_lo_u32
__device__ __forceinline__ u32 mul_u24_opaque(u32 a, u32 b) {
    u32 r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// the lane's three words of a ring row plus one neighbour word on each side (pads / other half)
__device__ __forceinline__ void ring_load5(const u32 *__restrict__ ring, int slot, int plane, int row, int wb,
                                           u32 (&a)[5]) {
    const u32 *p = ring + 1 + (slot * 2 + plane) * F_RPLANE + row * F_RS + wb - 1;
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = p[i];
}
__device__ __forceinline__ void ring_store3(u32 *__restrict__ ring, int slot, int plane, int row, int wb,
                                            const u32 (&w)[F_HW]) {
    u32 *p = ring + 1 + (slot * 2 + plane) * F_RPLANE + row * F_RS + wb;
#pragma unroll
    for (int i = 0; i < F_HW; ++i) p[i] = w[i];
}

// P3 of the fused pass as a function of the staged window (s_par byte codes, s_rw rank half words): tile
// pixels walk to their sources in lock-step, d, rank -> label, gather, store.  Returns whether some tile pixel
// was undecided.  NT = threads of the calling block.
template <int FR, int NT, bool EPI, bool STREAM>
__device__ __forceinline__ bool fused_walk_epilogue(
    const u8 *__restrict__ s_par, const short *__restrict__ s_tab, const uint2 *__restrict__ s_rw, int b, int H,
    int W, int th, int tw, int r0, int c0, int wr0, int wc0, int sh, const float *__restrict__ x,
    const float *__restrict__ vlist,
    int *__restrict__ finfo, float *__restrict__ out_depth, float *__restrict__ out_dt,
    int32_t *__restrict__ out_index, int *__restrict__ frame_status, const DepthEpilogue ep, int *__restrict__ fflag) {
    // ---- P3: tile pixels: walk to the source, d, rank -> label, gather, store.  Each lane walks F_EB
    // pixels in lock-step (their LDS reads are independent, so the hop latencies overlap) and then has
    // F_EB global gathers in flight together.
    // Frame bases are block-uniform (scalar registers); per pixel only 32-bit in-frame offsets are computed.
    const size_t fo = (size_t)b * H * W;
    const int nval = finfo[b * FI_STRIDE + FI_NVAL];
    const int misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const float *gbase = misaligned ? vlist + fo : x + fo;
    // the depth output may drop the first ep.row0 rows (then frames are H - row0 rows apart)
    float *od = out_depth ? out_depth + (size_t)b * (H - ep.row0) * W : nullptr, *ot = out_dt ? out_dt + fo : nullptr;
    const u32 dcrop = (u32)(ep.row0 * W) << 2;  // bytes of a frame's dropped rows
    constexpr bool plain = !EPI;  // the plain pass is compiled without the epilogue
    int32_t *oi = out_index ? out_index + fo : nullptr;
    const char *tab = reinterpret_cast<const char *>(s_tab);
    const int src_base = wr0 * W + wc0;  // frame offset of window cell (0,0) (may be negative)
    const u32 last_px = (u32)(H * W - 1);
    bool overflow = false;
    // A lane walks F_EB pixels of ONE column, rows 8 g .. 8 g + 7 of the tile: slot L = sb + tid is (row group g, column) =
    // divmod(L, tw), so the index arithmetic is done once per slot and the F_EB pixels differ by a row pitch each; a wave's store
    // is still a run of 64 consecutive pixels of a row (two runs where the slots wrap into the next group).
    // Software pipeline: the gathers of one batch stay in flight during the walk of the next batch (LDS only);
    // a batch's stores are issued just before the next batch's gathers.
    const int nslots = ((th + F_EB - 1) / F_EB) * tw;
    const float inv_tw = 1.0f / (float)tw;
    const u32 w4 = (u32)W << 2;
    int p_lab[F_EB], p_dd[F_EB];  // the batch whose gathers are in flight
    u32 p_opix[F_EB];
    float p_val[F_EB];
    u32 p_ok = 0;
    auto retire = [&]() {
#pragma unroll
        for (int e = 0; e < F_EB; ++e) {
            if (!((p_ok >> e) & 1u)) continue;
            // p_opix is a BYTE offset (< 2^32: a frame has < 2^26 pixels): scalar base + 32-bit vector offset
            // STREAM (rows of whole 128-byte lines: W % 32 == 0, aligned outputs, tile columns in whole lines): a wave's run of
            // consecutive tile pixels is whole lines, and nothing of the outputs is read again in this pass -- streaming stores
            // keep them from pushing the inputs out of the caches.  Otherwise the L2 has to merge the partial lines: plain stores.
            auto put = [](auto *p, auto v) {
                if constexpr (STREAM)
                    __builtin_nontemporal_store(v, p);
                else
                    *p = v;
            };
            if (oi) put(reinterpret_cast<int32_t *>(reinterpret_cast<char *>(oi) + p_opix[e]), p_lab[e]);
            if (ot) put(reinterpret_cast<float *>(reinterpret_cast<char *>(ot) + p_opix[e]), (float)p_dd[e]);
            if (od && plain) {
                put(reinterpret_cast<float *>(reinterpret_cast<char *>(od) + p_opix[e]), p_val[e]);
            } else if (od && p_opix[e] >= dcrop) {  // block-uniform choice: the plain pass pays nothing for the epilogue
                *reinterpret_cast<float *>(reinterpret_cast<char *>(od) + (p_opix[e] - dcrop)) = depth_epilogue(p_val[e], ep);
            }
        }
    };
    for (int sb = 0; sb < nslots; sb += NT) {
        int pos[F_EB], code[F_EB];  // pos = row * F_P + col: byte index in s_par
        u32 home[F_EB];             // window row << 16 | window column of the walker's own pixel
        u32 opix[F_EB];
        u32 ok = 0;
        {
            const int L = sb + (int)threadIdx.x, Lc = min(L, nslots - 1);
            // L / tw: (L + 0.5) / tw is at least 0.5 / tw away from an integer, far above float rounding
            // for L < 2^14, tw < 2^8
            const int g = (int)(((float)Lc + 0.5f) * inv_tw);
            const int tc = Lc - __mul24(g, tw), trb = g * F_EB;
            const u32 home0 = (u32)(FR + trb) << 16 | (u32)(FR + tc);
            const u32 opix0 = (u32)(__mul24(r0 + trb, W) + c0 + tc) << 2;  // byte offset; 24-bit multiplies are full rate (H, W < 8192)
            const int pos0 = __mul24(FR + trb, F_P) + FR + tc;
            const int nv = L < nslots ? th - trb : 0;  // the slot's rows inside the tile (the rest: still cells of the window, read and dropped)
#pragma unroll
            for (int e = 0; e < F_EB; ++e) {
                const bool valid = e < nv;
                home[e] = home0 + ((u32)e << 16);
                opix[e] = opix0 + (u32)e * w4;
                pos[e] = pos0 + e * F_P;
                code[e] = s_par[pos[e]];
                ok |= (valid && code[e] != F_NONE) ? (1u << e) : 0u;
                // undecidable here: the any-distance kernels take the pixel's row (found again below, off the common path)
                overflow |= valid && code[e] == F_NONE;
            }
        }
        // Unconditional hops: sources and undecided cells carry the step (0,0), so a walker that has
        // arrived just stays.  No selects, no divergent control flow -- the reads of the F_EB walkers
        // overlap.  Two hops between "everybody arrived?" checks.
        for (int hop = 0; hop < FR + 2; hop += 2) {
            int step[F_EB];
#pragma unroll
            for (int e = 0; e < F_EB; ++e) step[e] = *reinterpret_cast<const short *>(tab + code[e]);
            // everybody arrived? -- asked right after the (cheap, broadcast) table read, so the last trip
            // through the loop costs those 8 reads and not two more full hops
            int moving = 0;
#pragma unroll
            for (int e = 0; e < F_EB; ++e) moving |= step[e];
            if (!__any(moving != 0)) break;
#pragma unroll
            for (int e = 0; e < F_EB; ++e) pos[e] += step[e];
#pragma unroll
            for (int e = 0; e < F_EB; ++e) code[e] = s_par[pos[e]];
#pragma unroll
            for (int e = 0; e < F_EB; ++e) step[e] = *reinterpret_cast<const short *>(tab + code[e]);
#pragma unroll
            for (int e = 0; e < F_EB; ++e) pos[e] += step[e];
#pragma unroll
            for (int e = 0; e < F_EB; ++e) code[e] = s_par[pos[e]];
        }
        int lab[F_EB], dd[F_EB];
        u32 goff[F_EB];
        bool bad = false;
#pragma unroll
        for (int e = 0; e < F_EB; ++e) {
            // a decided chain ends on a source inside the in-image window
            // pos / F_P by a 24-bit multiply: 196 * 5350 / 2^20 = 1 + 2.3e-5, pos / 196 < 128, so the product is at most 0.003 above
            // the quotient, whose fraction is at most 195 / 196: same integer part (three full-rate instructions with the remainder;
            // the integer division is a quarter-rate v_mul_hi_u32)
            static_assert(F_P == 196 && F_WHM * F_P < (1 << 15), "the reciprocal below is F_P's");
            const int pr_ = (int)(__umul24((u32)pos[e], 5350u) >> 20), pc_ = pos[e] - __mul24(pr_, F_P);
            // L1 distance to the nearest source IS d: |drow| + |dcol| of the two 16-bit halves in one instruction
            dd[e] = (int)__builtin_amdgcn_sad_u16((u32)pr_ << 16 | (u32)pc_, home[e], 0u);
            const int bitpos = sh + pc_;  // bit index in the row's image-aligned bit string, from word w0
            const uint2 rw = s_rw[pr_ * 8 + (bitpos >> 5)];  // {32 source bits, sources before them in the frame}
            lab[e] = (int)rw.y + __popc(rw.x & ((1u << (bitpos & 31)) - 1u)) + 1;
            // depth_list[label-1] (tools.py:26).  A decided pixel has label >= 1, so the numpy wrap of index -1
            // cannot occur here.  Masks agree (block-uniform): the label-th value is x at the source, and
            // label <= nsrc = nval.  Masks differ: an index past the value list is numpy's IndexError.
            // The min only makes sure that a logic error could never become a wild global access (an LDS index
            // out of range reads garbage at worst).
            if (!misaligned) {
                goff[e] = min((u32)(src_base + __mul24(pr_, W) + pc_), last_px) << 2;  // byte offset
            } else {
                const int idx = lab[e] - 1;
                const bool oob = idx >= nval;
                bad |= ((ok >> e) & 1u) && oob;
                goff[e] = (oob ? 0u : min((u32)idx, last_px)) << 2;
            }
        }
        if (bad && out_depth) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
        retire();  // the previous batch: its gathers had the whole walk above to arrive
#pragma unroll
        for (int e = 0; e < F_EB; ++e) {
            p_val[e] = *reinterpret_cast<const float *>(reinterpret_cast<const char *>(gbase) + goff[e]);
            p_lab[e] = lab[e];
            p_dd[e] = dd[e];
            p_opix[e] = opix[e];
        }
        p_ok = ok;
    }
    retire();
    const bool any = __any(overflow);  // wave-uniform
    // The rows with an undecided pixel are redone by the any-distance kernels (every pixel of them; the decided ones get the
    // same values again).  With a depth epilogue the whole frame is (its chains may end on finished pixels whose stored
    // depth is already cropped / floored).  Rare, so a wave that met one looks for its rows only now (s_par is still there).
    if (any && !EPI) {
        u32 *rowflag = rowflag_of(fflag, (int)gridDim.x);  // the workspace keeps the row flags right behind the frame flags (gridDim.x = frames)
        for (int L = threadIdx.x; L < nslots; L += NT) {  // the slots this thread walked above
            const int g = L / tw, tc = L - g * tw;
            for (int tr = g * F_EB; tr < min(g * F_EB + F_EB, th); ++tr)
                if (s_par[__mul24(FR + tr, F_P) + FR + tc] == F_NONE) {
                    rowflag[(size_t)b * H + r0 + tr] = 1u;  // same-value race
                    // one of the two rows k_sky starts from: the any-distance kernels take the sky's rows as well
                    const int s0 = finfo[b * FI_STRIDE + FI_SKY0];
                    if (s0 > 0 && (r0 + tr == s0 || r0 + tr == s0 + 1)) finfo[b * FI_STRIDE + FI_SKY] = 0;
                }
        }
    }
    return any;
}

// FR = halo = largest distance the window can decide.
// Geometry of one halo's tiling (host side computes it; the window is always F_WHM x F_WWM)
struct FusedTiles {
    int TH, TW, tiles_x, ntiles, nty;  // TH: the tile rows' height when they split the whole frame (the largest it gets)
};

template <int FR, bool EPI, bool STREAM>
__device__ __forceinline__ void fused_body(bool premarked,
    const float *__restrict__ x, const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s,
    const u32 *__restrict__ rowbase_s, int *__restrict__ finfo, const float *__restrict__ vlist,
    int H, int W, int Wd, int nty, int TW, int tiles_x, float *__restrict__ out_depth,
    float *__restrict__ out_dt, int32_t *__restrict__ out_index,
    int *__restrict__ fflag, int *__restrict__ frame_status, const DepthEpilogue ep, u32 *__restrict__ s_ring, uint2 *__restrict__ s_rw,
    short *__restrict__ s_tab, u32 (*__restrict__ s_any)[F_NT / 64]) {
    const int tid = threadIdx.x;
    // (frames along x, tiles along y: the tiles beyond a frame's own tiling, which exit, are dispatched after every working block)
    const int b = blockIdx.x;
    const int ty = blockIdx.y / tiles_x, tx = blockIdx.y - ty * tiles_x;
    // the tile rows split the rows from FI_TR0 on evenly (k_frame: 0, or the first source row under a sky)
    const int tbase = finfo[b * FI_STRIDE + FI_TR0];
    const int TH = (H - tbase + nty - 1) / nty;
    int r0 = tbase + ty * TH;
    const int c0 = tx * TW;
    int th = min(TH, H - r0);
    const int tw = min(TW, W - c0);
    if (th <= 0) return;  // block-uniform
    if (premarked) {
        // k_frame handed rows of this frame to the any-distance kernels up front (the empty sky): the tile shrinks to the span
        // of its rows that are still this kernel's; a tile without any is done.  (A row another block marks meanwhile is
        // redone whole as well: whether this block still stores its part of it does not matter.)
        const u32 *rowflag = rowflag_of(fflag, (int)gridDim.x) + (size_t)b * H;
        if (tid == 0) {
            s_any[0][0] = 0xFFFFFFFFu;
            s_any[0][1] = 0u;
        }
        __syncthreads();
        if (tid < th && rowflag[r0 + tid] == 0u) {
            atomicMin(&s_any[0][0], (u32)tid);
            atomicMax(&s_any[0][1], (u32)tid + 1u);
        }
        __syncthreads();
        const u32 lo = s_any[0][0], hi = s_any[0][1];
        if (hi == 0u) return;  // block-uniform
        r0 += (int)lo;
        th = (int)(hi - lo);
    }
    const int wr0 = r0 - FR, wc0 = c0 - FR;  // image coords of window cell (0,0)
    const int WH = th + 2 * FR, WW = tw + 2 * FR;
    const int ca = max(0, -wc0), cb = min(WW, W - wc0);  // in-image window columns [ca, cb)
    const int w0 = wc0 >> 6;                             // first image word column the window touches (-1 if wc0 < 0)
    const int sh = wc0 - 64 * w0;                        // window column 0 is bit sh of image word w0

    // ---- P0: this lane's half row: image-aligned words -> LDS (for the ranks), window-aligned planes -> registers
    const int r = tid & (F_WHM - 1);  // window row of this lane
    const int hf = tid >> 7;          // which half of the row (wave-uniform)
    const int wb = F_HW * hf;         // first of the lane's three words
    u32 M[F_HW], D[F_HW], TM[F_HW];  // in-image mask, decided pixels, tile pixels (of the lane's three words)
    {
        const int gi = wr0 + r;
        const bool rowin = r < WH && gi >= 0 && gi < H;
        // the lane's 96 window columns start at bit sh + 96 hf of the row's image-aligned bit string;
        // three image words (192 bits) starting at word (sh + 96 hf) / 64 cover them
        const int bit0 = sh + 96 * hf;
        const int kw = bit0 >> 6;  // wave-uniform
        u32 g[7];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kk = kw + k;  // 0..3 relative to w0
            const int w = w0 + kk;
            u64 sb = 0;
            u32 rk = 0;
            if (rowin && kk < 4 && w >= 0 && w < Wd) {
                const size_t wi = ((size_t)b * H + gi) * Wd + w;
                sb = srcbits[wi];
                rk = rowbase_s[(size_t)b * H + gi] + wpre_s[wi];
            }
            g[2 * k] = (u32)sb;
            g[2 * k + 1] = (u32)(sb >> 32);
            // each of the four image words of a row is stored once: half 0 stores its words kk = 0, 1 (and 2
            // if half 1 starts later), half 1 the rest
            const bool mine = hf == 0 ? kk < 2 : (kk >= 2 && kk < 4);
            if (mine) {
                s_rw[r * 8 + 2 * kk] = make_uint2((u32)sb, rk);
                s_rw[r * 8 + 2 * kk + 1] = make_uint2((u32)(sb >> 32), rk + (u32)__popc((u32)sb));
            }
        }
        g[6] = 0;
        const int s6 = bit0 & 63;
        const bool hi = s6 & 32;  // wave-uniform
        const int s5 = s6 & 31;
#pragma unroll
        for (int i = 0; i < F_HW; ++i) {
            const u32 lo_w = hi ? g[i + 1] : g[i], hi_w = hi ? g[i + 2] : g[i + 1];
            const u32 word = s5 ? __builtin_amdgcn_alignbit(hi_w, lo_w, s5) : lo_w;
            // in-image columns of window word wb + i: [max(ca, 32 (wb+i)), min(cb, 32 (wb+i) + 32))
            const int lo = max(ca - 32 * (wb + i), 0), up = min(cb - 32 * (wb + i), 32);
            u32 m = 0;
            if (rowin && up > lo) m = (up >= 32 ? 0xFFFFFFFFu : ((1u << up) - 1u)) & ~((1u << lo) - 1u);
            M[i] = m;
            D[i] = word & m;
            // tile columns of window word wb + i: [FR, FR + tw) -- rows [FR, FR + th)
            const int tlo = max(FR - 32 * (wb + i), 0), tup = min(FR + tw - 32 * (wb + i), 32);
            u32 tm = 0;
            if (r >= FR && r < FR + th && tup > tlo) tm = (tup >= 32 ? 0xFFFFFFFFu : ((1u << tup) - 1u)) & ~((1u << tlo) - 1u);
            TM[i] = tm;
        }
    }
    // level 0: E_0 = L_0 = sources; zero the rest of the ring (levels "-1,-2,-3", the guard rows, the pads)
    for (int k = tid; k < F_RING; k += F_NT) s_ring[k] = 0;
    if (tid < 64) {
        const int hi = tid >> 3, lo = tid & 7;
        s_tab[tid] = (hi <= 4 && lo <= 4) ? (short)((hi - 2) * F_P + (lo - 2)) : (short)0;
    }
    __syncthreads();
    ring_store3(s_ring, 0, 0, r + 2, wb, D);
    ring_store3(s_ring, 0, 1, r + 2, wb, D);
    u32 C[6][F_HW];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < F_HW; ++i) C[j][i] = ((18 >> j) & 1) ? D[i] : 0u;  // sources: enc 18
    u32 Dup[F_HW], Ddn[F_HW];
    u32 Dl = 0, Dr = 0;  // D's neighbour words left / right of the lane's three (other half or nothing)
#pragma unroll
    for (int i = 0; i < F_HW; ++i) Dup[i] = Ddn[i] = 0;
    __syncthreads();

    // ---- P1: levels.  Per level: E_t (dilation), L_t (forward taps, winners recorded), d mod 8 planes.
    u32 P0[F_HW], P1[F_HW], P2[F_HW];  // bits 0..2 of the level at which a pixel was decided (sources: 0)
#pragma unroll
    for (int i = 0; i < F_HW; ++i) P0[i] = P1[i] = P2[i] = 0;
    for (int t = 1; t <= FR; ++t) {
        const int s1 = (t - 1) & 3, s2 = (t - 2) & 3, s3 = (t - 3) & 3, sw = t & 3;
        u32 nb[5], e1[5], l1[5], taken[F_HW], Et[F_HW], Lt[F_HW];
        // dilation of D_{t-1}: left/right in registers (+ the neighbour words), up/down through E_{t-1}
        ring_load5(s_ring, s1, 0, r + 2, wb, e1);  // E_{t-1}, this row (also the last tap's source)
        Dl |= e1[0];
        Dr |= e1[4];
        ring_load5(s_ring, s1, 0, r + 1, wb, nb);  // E_{t-1}, row r-1
#pragma unroll
        for (int i = 0; i < F_HW; ++i) Dup[i] |= nb[i + 1];
        u32 e1d[5];
        ring_load5(s_ring, s1, 0, r + 3, wb, e1d);  // E_{t-1}, row r+1
        bool nonempty = false;
        {
            const u32 dd[5] = {Dl, D[0], D[1], D[2], Dr};
#pragma unroll
            for (int i = 0; i < F_HW; ++i) {
                Ddn[i] |= e1d[i + 1];
                const u32 dil = hshift<1>(dd, i) | hshift<-1>(dd, i) | Dup[i] | Ddn[i];
                Et[i] = dil & ~D[i] & M[i];
                taken[i] = ~Et[i];
                nonempty |= Et[i] != 0;
            }
        }
        // forward taps in cv2 order; the candidates are live pixels of levels t-3, t-2, t-1
        ring_load5(s_ring, s3, 1, r + 0, wb, nb);  // L_{t-3}, row r-2
        tap_step<-1, ENC_F(0)>(nb, taken, C);
        tap_step<1, ENC_F(1)>(nb, taken, C);
        {
            u32 l3[5], l2[5];
            ring_load5(s_ring, s3, 1, r + 1, wb, l3);  // L_{t-3}, row r-1
            ring_load5(s_ring, s2, 1, r + 1, wb, l2);  // L_{t-2}, row r-1
            ring_load5(s_ring, s1, 1, r + 1, wb, nb);  // L_{t-1}, row r-1
            tap_step<-2, ENC_F(2)>(l3, taken, C);
            tap_step<-1, ENC_F(3)>(l2, taken, C);
            tap_step<0, ENC_F(4)>(nb, taken, C);
            tap_step<1, ENC_F(5)>(l2, taken, C);
            tap_step<2, ENC_F(6)>(l3, taken, C);
        }
        ring_load5(s_ring, s1, 1, r + 2, wb, l1);  // L_{t-1}, this row
        tap_step<-1, ENC_F(7)>(l1, taken, C);
        {
            const u32 m0 = (t & 1) ? 0xFFFFFFFFu : 0u, m1 = (t & 2) ? 0xFFFFFFFFu : 0u, m2 = (t & 4) ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (int i = 0; i < F_HW; ++i) {
                Lt[i] = taken[i] & Et[i];
                P0[i] |= Et[i] & m0;
                P1[i] |= Et[i] & m1;
                P2[i] |= Et[i] & m2;
            }
        }
#pragma unroll
        for (int i = 0; i < F_HW; ++i) D[i] |= Et[i];
        ring_store3(s_ring, sw, 0, r + 2, wb, Et);
        ring_store3(s_ring, sw, 1, r + 2, wb, Lt);
        // Go on while level t produced something (else nothing farther exists either) AND some TILE pixel is
        // still undecided (a chain only runs through smaller distances, so the halo beyond that is not needed;
        // halo pixels near the window edge see fewer sources and would drag the loop towards FR).  One flag
        // word per wave, read after the level's only barrier (__syncthreads_or costs three); double-buffered
        // by level parity: a wave can be at most one level ahead.
        {
            bool open = false;
#pragma unroll
            for (int i = 0; i < F_HW; ++i) open |= (TM[i] & ~D[i]) != 0;
            const u32 flags = (__any(nonempty) ? 1u : 0u) | (__any(open) ? 2u : 0u);
            if ((tid & 63) == 0) s_any[t & 1][tid >> 6] = flags;
            __syncthreads();
            u32 any = 0;
#pragma unroll
            for (int w = 0; w < F_NT / 64; ++w) any |= s_any[t & 1][w];
            if (any != 3u) break;
        }
    }
    // ---- P1b: the backward taps of ALL levels in one pass (pixels that are decided but not live).  The d mod 8
    // planes and D go through the ring's memory (slots 0 and 1; its guard rows and pad words are still zero).
    __syncthreads();  // every wave is out of the level loop: the ring is dead
    // (for P2, in the ring's tail that neither the planes below nor s_par use: a byte of a bit plane -> its eight bits in the
    // low bits of eight bytes)
    uint2 *s_lut = reinterpret_cast<uint2 *>(reinterpret_cast<u8 *>(s_ring) + F_WHM * F_P);
    s_lut[tid] = make_uint2((((u32)tid & 15u) * 0x00204081u) & 0x01010101u, (((u32)tid >> 4) * 0x00204081u) & 0x01010101u);
    ring_store3(s_ring, 0, 0, r + 2, wb, P0);
    ring_store3(s_ring, 0, 1, r + 2, wb, P1);
    ring_store3(s_ring, 1, 0, r + 2, wb, P2);
    ring_store3(s_ring, 1, 1, r + 2, wb, D);
    __syncthreads();
    {
        u32 taken[F_HW], a0[5], a1[5], a2[5], dv[5];
#pragma unroll
        for (int i = 0; i < F_HW; ++i) {
            // live = has a forward code or is a source (every forward enc and 18 are non-zero)
            const u32 live = C[0][i] | C[1][i] | C[2][i] | C[3][i] | C[4][i] | C[5][i];
            taken[i] = ~(D[i] & ~live);
        }
        auto row4 = [&](int row) {
            ring_load5(s_ring, 0, 0, row, wb, a0);
            ring_load5(s_ring, 0, 1, row, wb, a1);
            ring_load5(s_ring, 1, 0, row, wb, a2);
            ring_load5(s_ring, 1, 1, row, wb, dv);
        };
        // negated offsets of the cv2 taps, same order: (+2,+1) (+2,-1) (+1,+2) (+1,+1) (+1,0) (+1,-1) (+1,-2) (0,+1)
        row4(r + 4);
        bwd_tap<1, 3, 36 - ENC_F(0)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        bwd_tap<-1, 3, 36 - ENC_F(1)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        row4(r + 3);
        bwd_tap<2, 3, 36 - ENC_F(2)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        bwd_tap<1, 2, 36 - ENC_F(3)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        bwd_tap<0, 1, 36 - ENC_F(4)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        bwd_tap<-1, 2, 36 - ENC_F(5)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        bwd_tap<-2, 3, 36 - ENC_F(6)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
        row4(r + 2);
        bwd_tap<1, 1, 36 - ENC_F(7)>(a0, a1, a2, dv, P0, P1, P2, taken, C);
    }

    // ---- P2: un-slice the code planes of this half row into bytes (2 * enc), 8 pixels per step: a byte of a plane goes
    // through the table above (one 8-byte LDS read), two shift-ors put it into its bit of the eight bytes.  Undecided pixels
    // (not in D) get every plane bit: 2 * 63 = F_NONE.
    u8 *s_par = reinterpret_cast<u8 *>(s_ring);
    __syncthreads();  // the ring is dead for everybody before its memory becomes s_par
    {
        u32 *prow = reinterpret_cast<u32 *>(s_par + r * F_P) + wb * 8;
#pragma unroll
        for (int i = 0; i < F_HW; ++i) {
            u32 cj[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) cj[j] = C[j][i] | ~D[i];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                u32 lo = 0, hi = 0;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const uint2 sp = s_lut[(cj[j] >> (8 * q)) & 0xFFu];
                    lo |= sp.x << (j + 1);
                    hi |= sp.y << (j + 1);
                }
                prow[i * 8 + 2 * q] = lo;
                prow[i * 8 + 2 * q + 1] = hi;
            }
        }
    }
    __syncthreads();

    const bool overflow = fused_walk_epilogue<FR, F_NT, EPI, STREAM>(s_par, s_tab, s_rw, b, H, W, th, tw, r0, c0, wr0, wc0, sh, x, vlist,
                                                        finfo, out_depth, out_dt, out_index, frame_status, ep, fflag);
    if (overflow && (threadIdx.x & 63) == 0) {  // wave-uniform
        // 1: the rows marked in rowflag, 2: the whole frame.  Same-value race: every writer of a frame stores the same value,
        // the any-distance kernels read it after this kernel
        fflag[b] = EPI ? 2 : 1;
        atomicOr(frame_status + b, DTFILL_FRAME_GENERAL_PATH);
    }
}

// k_fused: one launch for both halos.  route[b] (k_frame): 16 / 32 = the halo that is expected to decide every pixel of
// frame b (source density, runs of source-free rows), 0 = the any-distance kernels take the frame.  The grid is
// sized for the halo-32 tiling (more, smaller tiles); blocks beyond a frame's own tiling exit.  A tile pixel that turns out
// to be farther than the halo from every source is not stored: its ROW is handed to the any-distance kernels (rowflag,
// fflag = 1; frame_status), which redo exactly those rows -- the empty sky of a LiDAR frame, a hole in a dense one.
// the block's LDS: one buffer, carved here
constexpr size_t F_OFF_RW = (sizeof(u32) * F_RING + 15) & ~(size_t)15, F_OFF_TAB = F_OFF_RW + sizeof(uint2) * F_WHM * 8, F_OFF_ANY = F_OFF_TAB + sizeof(short) * 64,
                 F_LDS_OWN = F_OFF_ANY + sizeof(u32) * 2 * (F_NT / 64);
constexpr size_t F_LDS = F_LDS_OWN;

template <bool STREAM>
__global__ __launch_bounds__(F_NT, 4) void k_fused(
    const float *__restrict__ x, const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s,
    const u32 *__restrict__ rowbase_s, int *__restrict__ finfo, const float *__restrict__ vlist,
    int H, int W, int Wd, FusedTiles t16, FusedTiles t32, float *__restrict__ out_depth,
    float *__restrict__ out_dt, int32_t *__restrict__ out_index,
    const int *__restrict__ route, int *__restrict__ fflag, int *__restrict__ frame_status, const DepthEpilogue ep) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[F_LDS];
    u32 *s_ring = reinterpret_cast<u32 *>(s_raw);  // later: s_par bytes
    // per window row, the eight image-aligned 32-pixel half words it touches: {source bits, sources before them
    // in frame raster order} -- one 8-byte LDS read and a 32-bit popcount per rank lookup
    uint2 *s_rw = reinterpret_cast<uint2 *>(s_raw + F_OFF_RW);
    short *s_tab = reinterpret_cast<short *>(s_raw + F_OFF_TAB);  // s_par displacement of the step enc (0 for the codes that are no step)
    u32(*s_any)[F_NT / 64] = reinterpret_cast<u32(*)[F_NT / 64]>(s_raw + F_OFF_ANY);  // per wave: did level t produce anything (double-buffered by level parity)
    const int rt = route[blockIdx.x];        // block-uniform
    if (rt == ROUTE_POINTS) return;  // a frame with a handful of sources: k_pts's tiles ride in k_fin's launch (dtfill_pts.hpp)
    const int r = rt > 0 ? (rt & 0xFF) : 0;
    const bool pre = rt > 0 && (rt & ROUTE_PREMARK);
    const bool epi = ep.row0 != 0 || ep.use_floor;  // uniform: the plain pass runs code compiled without the epilogue
#define FUSED_CALL(FR_, EPI_, T_)                                                                                        \
    fused_body<FR_, EPI_, STREAM && !(EPI_)>(pre, x, srcbits, wpre_s, rowbase_s, finfo, vlist, H, W, Wd, T_.nty, T_.TW, T_.tiles_x, out_depth, out_dt, \
                          out_index, fflag, frame_status, ep, s_ring, s_rw, s_tab, s_any)
    if (r == 16 && (int)blockIdx.y < t16.ntiles) {
        if (epi)
            FUSED_CALL(16, true, t16);
        else
            FUSED_CALL(16, false, t16);
    } else if (r == 32 && (int)blockIdx.y < t32.ntiles) {
        if (epi)
            FUSED_CALL(32, true, t32);
        else
            FUSED_CALL(32, false, t32);
    }
#undef FUSED_CALL
}
