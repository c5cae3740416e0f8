// dtfill_pts.hpp -- pts_body ("k_pts"): l1_cv frames with a handful of sources (the NYU sampling patterns), one kernel from the source
// list to the three outputs
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit), after dtfill_rows.hpp
// (the bit-sliced parent rule rule_tap, the tile geometry Q_* and fin_body, the other half of the k_fin kernel at the end of this
// file, are defined there).
#pragma once

// ------------------------------------------------------------------------------------------------
// A frame with at most PTS_MAX sources that is too thin for a window kernel (k_frame: ROUTE_POINTS; eval_NYU.py samples 200
// of 307 200 pixels, data_read.py:360-364) needs neither column pass nor row scans: a pixel's nearest sources are among the
// few whose cells reach its neighbourhood.  One block per 32 x 256 tile (k_fin's tile), four waves of 64 columns:
//   1. candidates of a wave's box (its 64 columns + 2, the tile's rows + 2): the nearest source of a box pixel p lies within
//      |p - c| + delta of p (delta = distance from the box's centre c to ITS nearest source), hence within delta + 2 rho of c
//      (rho = centre to corner, L1): the wave scans the frame's source list twice (delta, then the keepers);
//   2. pairwise dominance: |p - s| - |p - s'| is monotone in p.x and in p.y, so its minimum over the box sits on a corner --
//      s is nearer than s' nowhere in the box iff it is nowhere nearer on the four corners.  Strictly dominated sources are
//      no pixel's nearest (not even tied) and drop out: about one source per cell the box touches survives (5 - 8 of 200);
//   3. per pixel (lane = column, rows in registers) three packed-key minima over the survivors (v_sad_u16 is the L1
//      distance of two packed (row, column) pairs):
//        K1 = the smallest key d << 15 | list index << 6 | position in the wave's list, M2 = the second smallest (v_med3):
//        their distances differ iff the pixel has ONE nearest source (then that is its label: chains end on a nearest source);
//        K3 = min over the sources at or above the pixel's row of d << 13 | column: live(q) iff the leftmost nearest source
//        at or above q's row exists and has 3 (col - q.col) <= 2 d (the forward cone, as in k_rows);
//      d goes straight to the distance map, and so do label and depth of a pixel with one nearest source; d mod 8, live, tie and
//      "in the image" ride with the nearest source's list index in an LDS array over the box and are gathered into k_fin's
//      bit planes by a thread per word;
//   4. k_fin's second half on those planes: bit-sliced 5x5 parent rule for the tie pixels, hops through the step bytes,
//      every pixel takes the list index of the pixel its chain ends on: label = index + 1, depth = depth_list[label - 1]
//      (the source's own value when the masks agree).  A chain that leaves the tile while still on tie pixels is handed to
//      k_tiesx exactly as k_fin does.
// ------------------------------------------------------------------------------------------------
constexpr int PTS_MAX = L2_PTS_MAX;       // 512: the list index has 9 bits in the keys
constexpr int P_WB = 36;                  // rows of a wave's box: its 32 rows + 2 either side
constexpr int P_RC = 12;                  // rows per register chunk
static_assert(P_WB % P_RC == 0, "whole chunks");

// A block's tile is TH x TW pixels = (TH / 32) x (TW / 64) waves of 32 rows x 64 columns: 32 x 256 (k_fin's tile) or 64 x 128 --
// the host picks the one that wastes fewer waves on the frame's shape (a 640-wide frame is 2.5 tiles of 256 but 5 of 128).
template <int TH, int TW>
struct PtsGeom {
    static_assert(TH % 32 == 0 && TW % 64 == 0 && (TH / 32) * (TW / 64) == Q_NT / 64 && TH * (TW / 32) == Q_NT, "four waves, a thread per word");
    static constexpr int WW = TW / 32, RS = WW + 2, NR = TH + 4, SP = TW + 4, NWC = TW / 64, NPL = 5;
    // LDS of one block: under 32 KB, five blocks per CU (k_fin's own tiles need less)
    static constexpr size_t OFF_BYTE = (sizeof(u32) * NPL * NR * RS + 15) & ~(size_t)15, OFF_SRC = OFF_BYTE + sizeof(uint4) * TH * WW,
                            OFF_RC = (OFF_SRC + sizeof(u16) * NR * SP + 3) & ~(size_t)3, OFF_LIST = OFF_RC + sizeof(u32) * PTS_MAX,
                            OFF_CNT = OFF_LIST + Q_NT, LDS = OFF_CNT + sizeof(u32) * (Q_NT / 64);
    static_assert(SP % 2 == 0, "rows of the index array are 4-byte aligned");
    static_assert(sizeof(uint4) * TH * WW >= (Q_NT / 64) * PTS_MAX * sizeof(u16), "the candidate lists live in the step codes' place until the planes are done");
    static_assert(LDS <= 32 * 1024, "five blocks per CU");
};

#ifdef PTS_PROF  // development probe (scripts/dev/pts_phase_probe.py): cycles per phase, summed over the waves
__device__ unsigned long long g_pts_prof[12];
#define PTS_MARK(k_)                                                         \
    do {                                                                     \
        const unsigned long long t_ = __builtin_readcyclecounter();          \
        if ((threadIdx.x & 63) == 0) atomicAdd(&g_pts_prof[k_], t_ - tprev); \
        tprev = t_;                                                          \
    } while (0)
#else
#define PTS_MARK(k_)
#endif
__device__ __forceinline__ u32 med3u(u32 a, u32 b, u32 c) {
    u32 r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

template <int TH, int TW>
__device__ __forceinline__ void pts_body(unsigned char *__restrict__ s_raw, const float *__restrict__ x, const PtsSrc *__restrict__ ptslist, int H,
                                         int W, int Wp, int tiles_x, const float *__restrict__ vlist, float *__restrict__ out_depth,
                                         float *__restrict__ out_dt, int32_t *__restrict__ out_index, int *__restrict__ frame_status,
                                         int *__restrict__ finfo, u32 *__restrict__ xlist, u32 *__restrict__ xptr, u8 *__restrict__ unres) {
    using G = PtsGeom<TH, TW>;
    constexpr int WW = G::WW, RS = G::RS, NR = G::NR, SP = G::SP, NWC = G::NWC;
    u32(*s_pl)[NR][RS] = reinterpret_cast<u32(*)[NR][RS]>(s_raw);      // d bit 0, 1, 2, live, tie; rows r0-2 .. r0+TH+1; word 0 / WW+1: the ring's
    uint4 *s_code = reinterpret_cast<uint4 *>(s_raw + G::OFF_BYTE);      // per tile word: the four bit planes of its tie pixels' tap codes (tap_decode); before that: the waves' candidate lists
    u16(*s_src)[SP] = reinterpret_cast<u16(*)[SP]>(s_raw + G::OFF_SRC);  // per box pixel: list index of its nearest source | plane bits << 9
    u32 *s_rc = reinterpret_cast<u32 *>(s_raw + G::OFF_RC);              // the frame's sources: row << 16 | column; later their depths
    u8 *s_list = s_raw + G::OFF_LIST;                                     // the listed words (phase 4)
    u32 *s_cnt = reinterpret_cast<u32 *>(s_raw + G::OFF_CNT);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef PTS_PROF
    unsigned long long tprev = __builtin_readcyclecounter();
#endif
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * TH, c0 = tx * TW;
    const int wr = wave / NWC, wcol = wave - wr * NWC;  // the wave's place in the tile
    const int r0w = r0 + 32 * wr;                        // its first row
    const int nsrc = finfo[b * FI_STRIDE + FI_NSRC];
    const PtsSrc *sl = ptslist + (size_t)b * PTS_MAX;
    const u32 fo = (u32)b * (u32)(H * W);
    const int wpr = Wp >> 2;
    u32 myv[PTS_MAX / Q_NT];  // depth_list[k] of the sources this thread stages (bit pattern; NaN past the value list)
#pragma unroll
    for (int q = 0; q < PTS_MAX / Q_NT; ++q) {
        const int k = tid + q * Q_NT;
        myv[q] = 0u;
        if (k < nsrc) {
            s_rc[k] = sl[k].rc;
            myv[q] = __float_as_uint(sl[k].v);
        }
    }
    __syncthreads();
    // a wave whose 64 columns lie beyond the image (the last tile column of a 640-wide frame: half of it) has no pixel to decide
    const bool wave_idle = c0 + 64 * wcol >= W || r0w >= H;
    // ---- 1 + 2. this wave's candidates: box = rows r0 - 2 .. r0 + 33, columns wc0 - 2 .. wc0 + 65 (the edge waves' ring columns).
    // The list holds indices into the frame's list, in raster order (the frame's list is, and every compaction keeps it).
    const int wc0 = c0 + 64 * wcol;
    u16 *wc = reinterpret_cast<u16 *>(s_raw + G::OFF_BYTE) + wave * PTS_MAX;
    int nw = 0;  // wave-uniform
    if (!wave_idle) {
        // doubled coordinates: the centre sits on a half pixel
        const int cy2 = 2 * r0w + 32 - 1, cx2 = 2 * wc0 + 63;
        constexpr int RHO2 = (32 + 3) + (64 + 3);  // centre to corner, doubled
        u32 dmin = 0xFFFFFFFFu;
        for (int k = lane; k < nsrc; k += 64) {
            const u32 rc = s_rc[k];
            dmin = min(dmin, (u32)(abs(2 * (int)(rc >> 16) - cy2) + abs(2 * (int)(rc & 0xFFFFu) - cx2)));
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) dmin = min(dmin, (u32)__shfl_xor((int)dmin, o));
        const u32 reach2 = dmin + 2u * RHO2;
        int n = 0;
        for (int k0 = 0; k0 < nsrc; k0 += 64) {
            const int k = k0 + lane;
            bool keep = false;
            if (k < nsrc) {
                const u32 rc = s_rc[k];
                keep = (u32)(abs(2 * (int)(rc >> 16) - cy2) + abs(2 * (int)(rc & 0xFFFFu) - cx2)) <= reach2;
            }
            const u64 bal = __ballot(keep);
            if (keep) wc[n + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)k;
            n += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        // dominance on the box's corners (coordinates may lie outside the image: the box is only larger for it); packed
        // corners: v_sad_u16 wants both halves non-negative, so rows and columns are offset by 4.  The survivors are
        // compacted in place: whatever a later chunk still finds in the front part of the list is a real source of the frame,
        // and being dominated by any real source is reason enough to go.
        const u32 off = 4u << 16 | 4u;
        const u32 ca = (u32)(r0w - 2 + 4) << 16 | (u32)(wc0 - 2 + 4), cb = (u32)(r0w - 2 + 4) << 16 | (u32)(wc0 + 65 + 4),
                  cc = (u32)(r0w + 32 + 1 + 4) << 16 | (u32)(wc0 - 2 + 4), cd = (u32)(r0w + 32 + 1 + 4) << 16 | (u32)(wc0 + 65 + 4);
        for (int k0 = 0; k0 < n; k0 += 64) {
            const int k = k0 + lane;
            const u32 me = wc[min(k, n - 1)];
            const u32 s = s_rc[me] + off;
            const u32 da = __builtin_amdgcn_sad_u16(s, ca, 0u), db = __builtin_amdgcn_sad_u16(s, cb, 0u),
                      dc = __builtin_amdgcn_sad_u16(s, cc, 0u), dd = __builtin_amdgcn_sad_u16(s, cd, 0u);
            bool dom = false;
            for (int q = 0; q < n; ++q) {
                const u32 t = s_rc[wc[q]] + off;  // broadcast reads
                dom |= (__builtin_amdgcn_sad_u16(t, ca, 0u) < da) & (__builtin_amdgcn_sad_u16(t, cb, 0u) < db) &
                       (__builtin_amdgcn_sad_u16(t, cc, 0u) < dc) & (__builtin_amdgcn_sad_u16(t, cd, 0u) < dd);
            }
            const bool keep = k < n && !dom;
            const u64 bal = __ballot(keep);
            if (keep) wc[nw + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)me;
            nw += __popcll(bal);
            __builtin_amdgcn_wave_barrier();
        }
    }
    PTS_MARK(0);
#ifdef PTS_PROF
    if (lane == 0 && !wave_idle) {
        atomicAdd(&g_pts_prof[8], 1ull);
        atomicAdd(&g_pts_prof[9], (unsigned long long)nw);
        atomicAdd(&g_pts_prof[10], (unsigned long long)nw * nw);
        atomicMax(&g_pts_prof[11], (unsigned long long)nw);
    }
#endif
    // ---- 3. per pixel: the minima over the wave's candidates.  Lane = column wc0 + lane, P_RC rows at a time in registers.
    //   K1 = the smallest key d << 15 | index << 6 | position, M2 = the second smallest (v_med3 of the two and the newcomer): one nearest
    //   source iff their distances differ; K3 over the candidates at or above the row: the list is in raster order, so those
    //   are a prefix of it -- all rows of a chunk share the candidates above its first row (no test), none has those below its
    //   last row (no K3 at all), only the few inside the chunk's rows are tested row by row.
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    float *dt_f = out_dt ? out_dt + fo : nullptr, *dp_f = out_depth ? out_depth + fo : nullptr;
    int32_t *ix_f = out_index ? out_index + fo : nullptr;
    bool bad = false;
    {
        const int j = wc0 + lane;
        const bool jin = j < W;
        // Up to 64 candidates (the rule: a handful) sit one per lane in registers -- list index, position, depth -- and reach
        // the loop by v_readlane, the pixel's winner comes back by ds_bpermute: no memory access inside the loop or behind it.
        // More than 64 (sources crowding around one box) take the same loop through the lists in LDS and memory.
        auto run = [&](auto fast_t) {
            constexpr bool FAST = decltype(fast_t)::value;
            const u32 myidx = lane < nw ? (u32)wc[lane] : 0u;
            const u32 myrc = s_rc[myidx];
            // the depth goes out with the label when it can come from the candidates' registers: not from a list of more than
            // 64, not when the masks disagree (depth_list is then not the sources' own values) -- those waves' pixels take
            // theirs from the depth list in LDS once the tile's planes are done (below)
            const bool depth_now = FAST && dp_f && !misaligned;
            const float myval = (depth_now && lane < nw) ? sl[myidx].v : 0.0f;
            const int j3 = 3 * j;
            for (int rb = 0; rb < P_WB; rb += P_RC) {
                u32 K1[P_RC], M2[P_RC], K3[P_RC];
#pragma unroll
                for (int u = 0; u < P_RC; ++u) K1[u] = M2[u] = K3[u] = 0xFFFFFFFFu;
                const int ib = r0w - 2 + rb;  // image row of the chunk's first row (may be negative: such rows are masked below)
                const u32 qb = (u32)(ib + 4) << 16 | (u32)(j + 4);
                // nA candidates lie at or above the chunk's first row, nB at or above its last
                int nA = 0, nB = 0;
                for (int k0 = 0; k0 < nw; k0 += 64) {
                    const int sr = k0 + lane < nw ? (int)((k0 ? s_rc[wc[k0 + lane]] : myrc) >> 16) : 0x7FFF;
                    nA += __popcll(__ballot(sr <= ib));
                    nB += __popcll(__ballot(sr <= ib + P_RC - 1));
                }
                if constexpr (FAST) {
                    // keys d << 19 | column << 6 | position in the wave's list.  A column of pixels sees a source above it one farther
                    // per row down: the minimum (and the second smallest) over the sources at or above the chunk's first row is taken
                    // ONCE, on that row, and goes down the chunk by adding 1 << 19 per row; likewise upwards from the chunk's last row
                    // for the sources below it.  Only the sources on the chunk's own rows meet every row one by one.  The "down"
                    // minimum is K3 as it stands: its ties go to the smaller column.
                    u32 kd = 0xFFFFFFFFu, md = 0xFFFFFFFFu, ku = 0xFFFFFFFFu, mu = 0xFFFFFFFFu;
                    const u32 qe = qb + ((u32)(P_RC - 1) << 16);
                    for (int c = 0; c < nA; ++c) {
                        const u32 rc = (u32)__builtin_amdgcn_readlane((int)myrc, c);
                        const u32 key = __builtin_amdgcn_sad_u16(qb, rc + (4u << 16 | 4u), 0u) << 19 | ((rc & 0xFFFFu) << 6 | (u32)c);
                        md = med3u(kd, md, key);
                        kd = min(kd, key);
                    }
                    for (int c = nB; c < nw; ++c) {
                        const u32 rc = (u32)__builtin_amdgcn_readlane((int)myrc, c);
                        const u32 key = __builtin_amdgcn_sad_u16(qe, rc + (4u << 16 | 4u), 0u) << 19 | ((rc & 0xFFFFu) << 6 | (u32)c);
                        mu = med3u(ku, mu, key);
                        ku = min(ku, key);
                    }
#pragma unroll
                    for (int u = 0; u < P_RC; ++u) {
                        // (saturating: "no such source" stays what it is)
                        const u32 a = __builtin_elementwise_add_sat(kd, (u32)u << 19), a2 = __builtin_elementwise_add_sat(md, (u32)u << 19);
                        const u32 e = __builtin_elementwise_add_sat(ku, (u32)(P_RC - 1 - u) << 19),
                                  e2 = __builtin_elementwise_add_sat(mu, (u32)(P_RC - 1 - u) << 19);
                        K1[u] = min(a, e);
                        M2[u] = min(min(max(a, e), a2), e2);
                        K3[u] = a;
                    }
                    for (int c = nA; c < nB; ++c) {
                        const u32 rc = (u32)__builtin_amdgcn_readlane((int)myrc, c);
                        const u32 sp = rc + (4u << 16 | 4u), low = (rc & 0xFFFFu) << 6 | (u32)c;
                        const int sr = (int)(rc >> 16);
#pragma unroll
                        for (int u = 0; u < P_RC; ++u) {
                            const u32 key = __builtin_amdgcn_sad_u16(qb + ((u32)u << 16), sp, 0u) << 19 | low;
                            M2[u] = med3u(K1[u], M2[u], key);
                            K1[u] = min(K1[u], key);
                            if (sr <= ib + u) K3[u] = min(K3[u], key);  // wave-uniform
                        }
                    }
                } else {
                    auto body = [&](int c, auto mode) {
                        // the low 15 bits of the keys: list index << 6
                        u32 idx = (u32)__builtin_amdgcn_readfirstlane((int)wc[c]);
                        const u32 rc = (u32)__builtin_amdgcn_readfirstlane((int)s_rc[idx]);
                        idx <<= 6;
                        const u32 sp = rc + (4u << 16 | 4u), colkey = rc & 0xFFFFu;
                        const int sr = (int)(rc >> 16);
#pragma unroll
                        for (int u = 0; u < P_RC; ++u) {
                            const u32 d = __builtin_amdgcn_sad_u16(qb + ((u32)u << 16), sp, 0u);
                            const u32 key = d << 15 | idx;
                            M2[u] = med3u(K1[u], M2[u], key);  // K1 <= M2: the second smallest of the three
                            K1[u] = min(K1[u], key);
                            if (decltype(mode)::value == 0) {
                                K3[u] = min(K3[u], d << 13 | colkey);
                            } else if (decltype(mode)::value == 1) {
                                if (sr <= ib + u) K3[u] = min(K3[u], d << 13 | colkey);  // wave-uniform
                            }
                        }
                    };
                    for (int c = 0; c < nA; ++c) body(c, std::integral_constant<int, 0>{});
                    for (int c = nA; c < nB; ++c) body(c, std::integral_constant<int, 1>{});
                    for (int c = nB; c < nw; ++c) body(c, std::integral_constant<int, 2>{});
                }
                // the chunk's rows in three passes over the registers, so that the lookups of all rows are in flight together:
                // plane bits; the winners' list indices and depths (ds_bpermute from the lanes that hold the candidates); stores
                constexpr int DS = FAST ? 19 : 15;  // the distance's place in the keys
                u32 cb[P_RC];                       // d mod 8 | live << 3 | tie << 4 | in-image << 5, at bit 9 (all zero outside the image)
#pragma unroll
                for (int u = 0; u < P_RC; ++u) {
                    const u32 k1 = K1[u], k3 = K3[u], d = k1 >> DS;
                    // (a source pixel, d = 0, is no tie pixel: no second source shares its pixel, M2's distance is larger)
                    const bool tie = (M2[u] >> DS) == d;
                    const u32 col3 = FAST ? (k3 >> 6) & 8191u : k3 & 8191u;
                    const bool live = ((k3 >> (FAST ? 19 : 13)) == d) & (3 * (int)col3 <= (int)(2u * d) + j3);
                    const bool rin = (unsigned)(ib + u) < (unsigned)H;  // wave-uniform
                    cb[u] = (rin && jin) ? ((d & 7u) | (live ? 8u : 0u) | (tie ? 16u : 0u) | 32u) << 9 : 0u;
                }
                u32 i1[P_RC];   // the winner's list index: in the key, or with the candidate in lane k1 & 63
                float val[P_RC];
#pragma unroll
                for (int u = 0; u < P_RC; ++u) {
                    i1[u] = FAST ? (u32)__shfl((int)myidx, (int)(K1[u] & 63u)) : (K1[u] >> 6) & 511u;
                    val[u] = depth_now ? __shfl(myval, (int)(K1[u] & 63u)) : 0.0f;
                }
#pragma unroll
                for (int u = 0; u < P_RC; ++u) {
                    const int row = rb + u, i = ib + u;  // wave-uniform
                    // the pixel's plane bits ride with its index
                    s_src[32 * wr + row][2 + 64 * wcol + lane] = (u16)(cb[u] ? (i1[u] | cb[u]) : 0u);
                    if ((unsigned)i < (unsigned)H && row >= 2 && row < 34 && jin) {  // the wave's own 32 rows
                        // the tile's own pixels: the distance now; label and depth too unless a chain has to be followed (phase 4)
                        const u32 ob = (u32)(__umul24((u32)i, (u32)W) + (u32)j) << 2;
                        if (dt_f) st_off_nt(dt_f, ob, (float)(K1[u] >> DS));
                        if (!((cb[u] >> 13) & 1u)) {
                            if (ix_f) st_off_nt(ix_f, ob, (int32_t)i1[u] + 1);
                            if (depth_now) st_off_nt(dp_f, ob, val[u]);
                        }
                    }
                }
            }
        };
        if (wave_idle) {
            // no plane bits: outside the image.  (Its first four rows are the last four of the wave above, which writes them.)
            for (int row = wr ? 4 : 0; row < P_WB; ++row) s_src[32 * wr + row][2 + 64 * wcol + lane] = 0;
        } else if (nw <= 64 && H + W <= 8100)  // (13 bits of distance, 13 of column in the fast keys)
            run(std::true_type{});
        else
            run(std::false_type{});
        // the ring's columns: c0 - 2, c0 - 1 (the waves of the first wave column) and c0 + TW, c0 + TW + 1 (of the last); lane = box row
        if (wave_idle && wcol == NWC - 1) {
            if (lane >= (wr ? 4 : 0) && lane < P_WB) s_src[32 * wr + lane][TW + 2] = s_src[32 * wr + lane][TW + 3] = 0;
        } else if (wave_idle) {
            if (wcol == 0 && lane >= (wr ? 4 : 0) && lane < P_WB) s_src[32 * wr + lane][0] = s_src[32 * wr + lane][1] = 0;
        } else {
          for (int side = 0; side < 2; ++side) {  // the tile's left ring columns (first wave column), its right ones (last)
            if (side == 0 ? wcol != 0 : wcol != NWC - 1) continue;
            const int jb = side == 0 ? c0 - 2 : c0 + TW;
            const int row = lane, i = r0w - 2 + row;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int jj = jb + e;
                u32 k1 = 0xFFFFFFFFu, m2 = 0xFFFFFFFFu, k3 = 0xFFFFFFFFu;
                const u32 q = (u32)(i + 4) << 16 | (u32)(jj + 4);
                for (int c = 0; c < nw; ++c) {
                    const u32 idx = wc[c], rc = s_rc[idx];
                    const u32 d = __builtin_amdgcn_sad_u16(q, rc + (4u << 16 | 4u), 0u);
                    const u32 key = d << 9 | idx;
                    m2 = med3u(k1, m2, key);
                    k1 = min(k1, key);
                    if ((int)(rc >> 16) <= i) k3 = min(k3, d << 13 | (rc & 0xFFFFu));
                }
                const bool in = row < P_WB && i >= 0 && i < H && jj >= 0 && jj < W;
                const u32 d = k1 >> 9, i1 = k1 & 511u;
                const bool tie = ((m2 >> 9) == d) & (d != 0u);
                const bool live = ((k3 >> 13) == d) & (3 * (int)(k3 & 8191u) <= (int)(2u * d) + 3 * jj);
                const u32 code = in ? ((d & 7u) | (live ? 8u : 0u) | (tie ? 16u : 0u) | 32u) : 0u;
                if (row < P_WB) s_src[32 * wr + row][side == 0 ? e : TW + 2 + e] = (u16)(i1 | code << 9);
            }
          }
        }
    }
    PTS_MARK(1);
    __syncthreads();  // every box pixel's index and plane bits are in s_src; the candidate lists (in s_code's place) are dead
    // from here on s_rc holds depth_list[k] instead of the sources' positions (phase 4 looks depths up there)
    {
        const int nvl = finfo[b * FI_STRIDE + FI_NVAL], mis = finfo[b * FI_STRIDE + FI_MISALIGNED];
#pragma unroll
        for (int q = 0; q < PTS_MAX / Q_NT; ++q) {
            const int k = tid + q * Q_NT;
            if (k < nsrc) s_rc[k] = !mis ? myv[q] : k < nvl ? __float_as_uint(vlist[fo + k]) : 0x7FC00000u;
        }
    }
    // ---- the bit planes k_fin's rule reads, 32 pixels per word: a thread per (box row, word) gathers bit p of 32 codes
    for (int it = tid; it < NR * (WW + 2); it += Q_NT) {
        const int row = it / (WW + 2), w = it - row * (WW + 2);  // word 0 / WW + 1: the ring's (two pixels each)
        u32 pl[5] = {0u, 0u, 0u, 0u, 0u};  // (bit 5 of a code, "in the image", follows from the geometry: inw() below)
        if (w >= 1 && w <= WW) {
            const u32 *q4 = reinterpret_cast<const u32 *>(&s_src[row][2 + 32 * (w - 1)]);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                // four pixels: the high bytes of their entries side by side (code bit p is bit 1 + p of a byte), then per plane the
                // four bits 8 apart are folded next to each other (y | y << 7, | << 14: bits 21 .. 24 of the shifted word)
                const u32 X = __builtin_amdgcn_perm(q4[2 * g + 1], q4[2 * g], 0x07050301u);
#pragma unroll
                for (int p = 0; p < 5; ++p) {
                    u32 y = X & (0x01010101u << (1 + p));
                    y |= y << 7;
                    y |= y << 14;
                    pl[p] |= ((y >> (22 + p)) & 15u) << (4 * g);
                }
            }
        } else {
            const int cb = w == 0 ? 0 : TW + 2, sh = w == 0 ? 30 : 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const u32 v = s_src[row][cb + e];
#pragma unroll
                for (int p = 0; p < 5; ++p) pl[p] |= ((v >> (9 + p)) & 1u) << (sh + e);
            }
        }
#pragma unroll
        for (int p = 0; p < 5; ++p) s_pl[p][row][w] = pl[p];
    }
    PTS_MARK(2);
    __syncthreads();
    PTS_MARK(3);
    // the depths phase 3 left out (see depth_now there): depth_list[label - 1] (tools.py:26) from LDS, for the wave's pixels that
    // are no tie pixels (those follow their chain below)
    if (dp_f && !wave_idle && (misaligned || nw > 64 || H + W > 8100)) {  // wave-uniform
        const int j = wc0 + lane;
        for (int row = 0; row < 32; ++row) {
            const int i = r0w + row;
            if (i >= H) break;
            const u32 v = s_src[32 * wr + row + 2][2 + 64 * wcol + lane];
            if (j < W && !((v >> 13) & 1u)) {
                const u32 idx = v & 511u;
                bad |= misaligned && (int)idx >= nval;  // (the list holds NaN there)
                dp_f[i * W + j] = __uint_as_float(s_rc[idx]);
            }
        }
    }
    // ---- 4. the tie pixels: the tile's 32-pixel words that hold one are listed, one thread per listed word (the words
    // without one -- most, in a frame with a handful of sources -- cost nothing, and whole waves drop out): k_fin's bit-sliced
    // parent rule, step bytes, hops
    int trow = tid / WW, tw = tid % WW;
    int total = 0;  // listed words (block-uniform)
    bool has = false;
    {
        const int gi0 = r0 + trow, gw0 = (c0 >> 5) + tw;
        const u32 own = s_pl[4][trow + 2][tw + 1];
        if (gi0 < H && gw0 < wpr && !own) reinterpret_cast<u32 *>(unres + ((size_t)b * H + gi0) * Wp)[gw0] = 0u;  // nothing open in this word
        const u64 tb = __ballot(own != 0u);
        if (lane == 0) s_cnt[wave] = (u32)__popcll(tb);
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int w = 0; w < Q_NT / 64; ++w) {
            base += w < wave ? (int)s_cnt[w] : 0;
            total += (int)s_cnt[w];
        }
        if (own) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(tb >> 32), __builtin_amdgcn_mbcnt_lo((u32)tb, 0u))] = (u8)tid;
        __syncthreads();  // (also: every thread has read the per-wave counts, s_cnt is free for the list append below)
        has = tid < total;
        const int mine = has ? (int)s_list[tid] : 0;
        trow = mine / WW;
        tw = mine % WW;
    }
    const bool any_tie = total != 0;
    const int gi = r0 + trow, gw = (c0 >> 5) + tw;
    const bool tin = has && gi < H && gw < wpr;
    const u32 mytie = has ? s_pl[4][trow + 2][tw + 1] : 0u;
    u32 umask = 0;  // the word's tie pixels that k_tiesx finishes
    if (any_tie) {
        if (mytie) {
            u32 C[4] = {0, 0, 0, 0};
            // the pixels of box row `row`, plane word `widx` that lie inside the image (word 0 is the left ring's: columns c0 - 32 ..)
            auto inw = [&](int row, int widx) -> u32 {
                const int i = r0 - 2 + row, j0 = c0 - 32 + 32 * widx;  // the word's first column
                if (i < 0 || i >= H) return 0u;
                const int lo = max(-j0, 0), up = min(W - j0, 32);
                return up <= lo ? 0u : (up >= 32 ? 0xFFFFFFFFu : ((1u << up) - 1u)) & ~((1u << lo) - 1u);
            };
            auto ld3 = [&](int p, int row, u32 (&o)[3]) {
                if (p == 5) {
                    o[0] = inw(row, tw); o[1] = inw(row, tw + 1); o[2] = inw(row, tw + 2);
                    return;
                }
                const u32 *q3 = &s_pl[p][row][tw];
                o[0] = q3[0]; o[1] = q3[1]; o[2] = q3[2];
            };
            const int qrow = trow + 2;
            const u32 b0 = s_pl[0][qrow][tw + 1], b1 = s_pl[1][qrow][tw + 1], b2 = s_pl[2][qrow][tw + 1];
            const u32 qlive = s_pl[3][qrow][tw + 1];
            u32 takenF = ~(mytie & qlive), takenB = ~(mytie & ~qlive);
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
            u32 z0[3], z1[3], z2[3], zv[3];
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
            // the tap codes stay bit-sliced: a hop reads its four bits out of the word's planes (only tie pixels' codes are ever
            // read: a chain stops on the first pixel that is none)
            s_code[trow * WW + tw] = make_uint4(C[0], C[1], C[2], C[3]);
        }
        PTS_MARK(4);
        __syncthreads();
        // every tie pixel of this word hops along the tap codes until it stands on a pixel that is no tie pixel and takes that
        // pixel's source; a chain that leaves the tile while still on tie pixels, or runs longer than Q_HOPS, goes to k_tiesx
        auto is_tie = [&](int r, int c) -> bool { return (s_pl[4][r + 2][(c + 32) >> 5] >> ((c + 32) & 31)) & 1u; };
        u32 m = mytie;  // (a tie bit is only ever set inside the image)
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            const int pc = tw * 32 + bit;
            int er = trow, ec = pc;
            bool open = true;
            for (int hop = 0; hop < Q_HOPS; ++hop) {
                if (er < 0 || er >= TH || ec < 0 || ec >= TW) break;  // a tie pixel of another tile: no step here
                const uint4 cw4 = s_code[er * WW + (ec >> 5)];
                const int sb = ec & 31;
                int di, dj;
                tap_decode((int)(((cw4.x >> sb) & 1u) | ((cw4.y >> sb) & 1u) << 1 | ((cw4.z >> sb) & 1u) << 2 | ((cw4.w >> sb) & 1u) << 3), di, dj);
                er += di;
                ec += dj;
                if (!is_tie(er, ec)) {
                    open = false;
                    break;
                }
            }
            const u32 pix = (u32)(gi * W + c0 + pc);
            if (open) {
                umask |= 1u << bit;
                const int ei = min(max(r0 + er, 0), H - 1), ej = min(max(c0 + ec, 0), W - 1);
                xptr[fo + pix] = (u32)(ei * W + ej);
            } else {
                const u32 idx = s_src[min(max(er, -2), TH + 1) + 2][min(max(ec, -2), TW + 1) + 2] & 511u;
                if (ix_f) ix_f[pix] = (int32_t)idx + 1;
                bad |= misaligned && (int)idx >= nval;
                if (dp_f) dp_f[pix] = __uint_as_float(s_rc[idx]);
            }
        }
        PTS_MARK(5);
        // the handed-on pixels join the frame's list: block-wide count, ONE atomic, then every thread writes its own
        const int cu = __popc(umask);
        int incl = cu;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) s_cnt[wave] = (u32)incl;
        __syncthreads();
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < Q_NT / 64; ++w) {
            pre += w < wave ? (int)s_cnt[w] : 0;
            all += (int)s_cnt[w];
        }
        if (all) {  // block-uniform
            __syncthreads();
            if (tid == 0) s_cnt[0] = (u32)atomicAdd(&finfo[b * FI_STRIDE + FI_NUNRES], all);
            __syncthreads();
            u32 o = s_cnt[0] + (u32)(pre + incl - cu);
            u32 mm = umask;
            while (mm) {
                const int k = __ffs((int)mm) - 1;
                mm &= mm - 1;
                xlist[fo + o++] = (u32)(gi * W + c0 + tw * 32 + k);
            }
        }
    }
    if (tin) reinterpret_cast<u32 *>(unres + ((size_t)b * H + gi) * Wp)[gw] = umask;
    PTS_MARK(6);
    if (bad && out_depth) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
}

// ------------------------------------------------------------------------------------------------
// The k_fin launch: a block is a tile of fin_body (frames on the any-distance path: fflag 1 / 2) or a tile of pts_body (fflag 3).
// One buffer of LDS, carved by either; 32 KB and 92 registers keep five blocks on a CU.
// ------------------------------------------------------------------------------------------------
struct PtsArgs {  // what a k_pts tile needs beyond fin_body's arguments
    const PtsSrc *ptslist;
    float *out_dt;
    int tiles_x, ntiles;
    int tall;        // tiles of 64 x 128 instead of 32 x 256
    int fin_ntiles;  // fin_body's tiles (0: neither depth nor labels are wanted, the launch is k_pts's alone)
};
constexpr size_t cmax(size_t a, size_t b) { return a > b ? a : b; }
constexpr size_t FIN_LDS = cmax(FIN_LDS_OWN, cmax(PtsGeom<32, 256>::LDS, PtsGeom<64, 128>::LDS));
static_assert(FIN_LDS <= 32 * 1024, "five blocks per CU");

__global__ __launch_bounds__(Q_NT, 4) void k_fin(
    const u8 *__restrict__ planes, size_t plane_bytes, int Wp, const int *__restrict__ fflag, int H, int W, int Wd,
    int tiles_x, const u32 *__restrict__ spix_ws, const float *__restrict__ x, const uint4 *__restrict__ rec,
    const float *__restrict__ vlist,
    float *__restrict__ out_depth, int32_t *__restrict__ out_index, int *__restrict__ frame_status,
    int *__restrict__ finfo, u32 *__restrict__ xlist, u32 *__restrict__ xptr, u8 *__restrict__ unres, int vec,
    const DepthEpilogue ep, float *__restrict__ dscratch, const u32 *__restrict__ rowflag, const PtsArgs pa) {
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[FIN_LDS];
    if (fflag[blockIdx.y] == 3) {  // (block-uniform)
        if ((int)blockIdx.x < pa.ntiles) {
            if (pa.tall)
                pts_body<64, 128>(s_raw, x, pa.ptslist, H, W, Wp, pa.tiles_x, vlist, out_depth, pa.out_dt, out_index, frame_status, finfo, xlist, xptr, unres);
            else
                pts_body<32, 256>(s_raw, x, pa.ptslist, H, W, Wp, pa.tiles_x, vlist, out_depth, pa.out_dt, out_index, frame_status, finfo, xlist, xptr, unres);
        }
        return;
    }
    if ((int)blockIdx.x < pa.fin_ntiles)
        fin_body(s_raw, planes, plane_bytes, Wp, fflag, H, W, Wd, tiles_x, spix_ws, x, rec, vlist, out_depth, out_index, frame_status, finfo, xlist, xptr,
                 unres, vec, ep, dscratch, rowflag);
}
