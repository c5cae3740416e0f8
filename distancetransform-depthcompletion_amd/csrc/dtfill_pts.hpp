// dtfill_pts.hpp -- pts_body ("k_pts"): l1_cv frames with a handful of sources (the NYU sampling patterns), one kernel from the source
// list to the three outputs
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit), after dtfill_rows.hpp
// (the bit-sliced parent rule rule_tap / step_tap and the tile geometry Q_* are k_fin's).
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
    static constexpr int WW = TW / 32, RS = WW + 3, NR = TH + 4, SP = TW + 4, NWC = TW / 64;
    // LDS of one block, carved from the window kernel's buffer (the tiles of such frames ride in k_fused's launch)
    static constexpr size_t OFF_BYTE = (sizeof(u32) * 6 * NR * RS + 15) & ~(size_t)15, OFF_SRC = OFF_BYTE + TH * TW,
                            OFF_RC = (OFF_SRC + sizeof(u16) * NR * SP + 3) & ~(size_t)3, OFF_LIST = OFF_RC + sizeof(u32) * PTS_MAX,
                            OFF_CNT = OFF_LIST + Q_NT, LDS = OFF_CNT + sizeof(u32) * (Q_NT / 64);
    static_assert(SP % 2 == 0, "rows of the index array are 4-byte aligned");
    static_assert(LDS <= F_LDS, "k_fused's buffer holds a k_pts block");
    static_assert((size_t)TH * TW >= (Q_NT / 64) * PTS_MAX * sizeof(u16), "the candidate lists live in s_byte until the planes are done");
};

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
    u32(*s_pl)[NR][RS] = reinterpret_cast<u32(*)[NR][RS]>(s_raw);      // d bit 0, 1, 2, live, tie, in-image; rows r0-2 .. r0+TH+1; word 0 / WW+1: the ring's
    u8(*s_byte)[TW] = reinterpret_cast<u8(*)[TW]>(s_raw + G::OFF_BYTE);  // per tile pixel: step to its parent; before that: the waves' candidate lists
    u16(*s_src)[SP] = reinterpret_cast<u16(*)[SP]>(s_raw + G::OFF_SRC);  // per box pixel: list index of its nearest source | plane bits << 9
    u32 *s_rc = reinterpret_cast<u32 *>(s_raw + G::OFF_RC);              // the frame's sources: row << 16 | column; later their depths
    u8 *s_list = s_raw + G::OFF_LIST;                                     // the listed words (phase 4)
    u32 *s_cnt = reinterpret_cast<u32 *>(s_raw + G::OFF_CNT);
    const int b = blockIdx.y, tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    // ---- 3. per pixel: the minima over the wave's candidates.  Lane = column wc0 + lane, P_RC rows at a time in registers.
    //   K1 = the smallest key d << 15 | index << 6 | position, M2 = the second smallest (v_med3 of the two and the newcomer): one nearest
    //   source iff their distances differ; K3 over the candidates at or above the row: the list is in raster order, so those
    //   are a prefix of it -- all rows of a chunk share the candidates above its first row (no test), none has those below its
    //   last row (no K3 at all), only the few inside the chunk's rows are tested row by row.
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const float *vl_f = vlist + fo;
    float *dt_f = out_dt ? out_dt + fo : nullptr, *dp_f = out_depth ? out_depth + fo : nullptr;
    int32_t *ix_f = out_index ? out_index + fo : nullptr;
    bool bad = false;
    auto depth_of = [&](u32 idx, bool want) -> float {  // depth_list[label - 1] (tools.py:26) for label = idx + 1
        if (misaligned) {  // block-uniform; rare
            const bool oob = (int)idx >= nval;
            bad |= want && oob;
            return oob ? nanf("") : vl_f[idx];
        }
        return sl[min(idx, (u32)(PTS_MAX - 1))].v;  // masks agree: the label-th value IS the source pixel's own depth
    };
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
            const float myval = (FAST && dp_f && !misaligned && lane < nw) ? sl[myidx].v : 0.0f;
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
                auto body = [&](int c, auto mode) {
                    // the low 15 bits of the keys: list index << 6 | position in the wave's list (FAST; both in the list's order)
                    u32 idx, rc;
                    if (FAST) {
                        idx = (u32)__builtin_amdgcn_readlane((int)myidx, c) << 6 | (u32)c;
                        rc = (u32)__builtin_amdgcn_readlane((int)myrc, c);
                    } else {
                        idx = (u32)__builtin_amdgcn_readfirstlane((int)wc[c]);
                        rc = (u32)__builtin_amdgcn_readfirstlane((int)s_rc[idx]);
                        idx <<= 6;
                    }
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
#pragma unroll
                for (int u = 0; u < P_RC; ++u) {
                    const int row = rb + u, i = ib + u;  // wave-uniform
                    const bool rin = i >= 0 && i < H;
                    const u32 d = K1[u] >> 15, i1 = (K1[u] >> 6) & 511u, pos = K1[u] & 63u;
                    const bool tie = ((M2[u] >> 15) == d) & (d != 0u);
                    const bool live = ((K3[u] >> 13) == d) & (3 * (int)(K3[u] & 8191u) <= (int)(2u * d) + 3 * j);
                    // the pixel's plane bits ride with its index: d mod 8 | live << 3 | tie << 4 | in-image << 5 (all zero outside)
                    const u32 code = (rin && jin) ? ((d & 7u) | (live ? 8u : 0u) | (tie ? 16u : 0u) | 32u) : 0u;
                    s_src[32 * wr + row][2 + 64 * wcol + lane] = (u16)(i1 | code << 9);
                    if (row >= 2 && row < 34) {  // the wave's own 32 rows
                        // the tile's own pixels: the distance now; label and depth too unless a chain has to be followed (phase 4)
                        float val = 0.0f;
                        if (dp_f) val = (FAST && !misaligned) ? __shfl(myval, (int)pos) : 0.0f;
                        if (rin && jin) {
                            const u32 ob = (u32)(__umul24((u32)i, (u32)W) + (u32)j) << 2;
                            if (dt_f) st_off_nt(dt_f, ob, (float)d);
                            if (!tie) {
                                if (ix_f) st_off_nt(ix_f, ob, (int32_t)i1 + 1);
                                if (dp_f) st_off_nt(dp_f, ob, (FAST && !misaligned) ? val : depth_of(i1, true));
                            }
                        }
                    }
                }
            }
        };
        if (wave_idle) {
            // no plane bits: outside the image.  (Its first four rows are the last four of the wave above, which writes them.)
            for (int row = wr ? 4 : 0; row < P_WB; ++row) s_src[32 * wr + row][2 + 64 * wcol + lane] = 0;
        } else if (nw <= 64)
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
    __syncthreads();  // every box pixel's index and plane bits are in s_src; the candidate lists (in s_byte) are dead
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
        u32 pl[6] = {0u, 0u, 0u, 0u, 0u, 0u};
        if (w >= 1 && w <= WW) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const u32 *q4 = reinterpret_cast<const u32 *>(&s_src[row][2 + 32 * (w - 1) + 8 * g]);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const u32 v = q4[h];  // two pixels: bits 9..14 and 25..30
#pragma unroll
                    for (int p = 0; p < 6; ++p) pl[p] |= (((v >> (9 + p)) & 1u) | ((v >> (24 + p)) & 2u)) << (8 * g + 2 * h);
                }
            }
        } else {
            const int cb = w == 0 ? 0 : TW + 2, sh = w == 0 ? 30 : 0;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const u32 v = s_src[row][cb + e];
#pragma unroll
                for (int p = 0; p < 6; ++p) pl[p] |= ((v >> (9 + p)) & 1u) << (sh + e);
            }
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) s_pl[p][row][w] = pl[p];
    }
    __syncthreads();
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
            u32 E[6] = {0, ~0u, 0, 0, ~0u, 0};
            auto ld3 = [&](int p, int row, u32 (&o)[3]) {
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
            E[1] = E[4] = ~mytie;
            step_tap<0>(C, mytie, E); step_tap<1>(C, mytie, E); step_tap<2>(C, mytie, E); step_tap<3>(C, mytie, E);
            step_tap<4>(C, mytie, E); step_tap<5>(C, mytie, E); step_tap<6>(C, mytie, E); step_tap<7>(C, mytie, E);
            step_tap<8>(C, mytie, E); step_tap<9>(C, mytie, E); step_tap<10>(C, mytie, E); step_tap<11>(C, mytie, E);
            step_tap<12>(C, mytie, E); step_tap<13>(C, mytie, E); step_tap<14>(C, mytie, E); step_tap<15>(C, mytie, E);
            // only tie pixels' bytes are ever read (a chain stops on the first pixel that is none): words without one write nothing
            u32 *brow = reinterpret_cast<u32 *>(&s_byte[trow][tw * 32]);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                u32 v = 0;
#pragma unroll
                for (int jq = 0; jq < 6; ++jq) v |= (mul_u24_opaque((E[jq] >> (4 * g)) & 0xFu, 0x00204081u) & 0x01010101u) << jq;
                brow[g] = v;
            }
        }
        __syncthreads();
        // every tie pixel of this word hops along the step bytes until it stands on a pixel that is no tie pixel and takes that
        // pixel's source; a chain that leaves the tile while still on tie pixels, or runs longer than Q_HOPS, goes to k_tiesx
        auto is_tie = [&](int r, int c) -> bool { return (s_pl[4][r + 2][(c + 32) >> 5] >> ((c + 32) & 31)) & 1u; };
        u32 m = mytie & s_pl[5][trow + 2][tw + 1];
        while (m) {
            const int bit = __ffs((int)m) - 1;
            m &= m - 1;
            const int pc = tw * 32 + bit;
            int er = trow, ec = pc;
            bool open = true;
            for (int hop = 0; hop < Q_HOPS; ++hop) {
                if (er < 0 || er >= TH || ec < 0 || ec >= TW) break;  // a tie pixel of another tile: no step here
                const u32 bb = s_byte[er][ec] & 63u;
                er += (int)(bb >> 3) - 2;
                ec += (int)(bb & 7u) - 2;
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
    if (bad && out_depth) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
}
