// dtfill_l2.hpp -- the exact Euclidean transform (l2 metric): k_l2win (dense frames), k_l2rest (far pixels, rows of far pixels, sparse frames)
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------------
// l2 metric, dense frames: k_l2win<R>.  The squared distance separates, d2(i,j) = min_y (y-i)^2 + hx(y,j)^2 with hx the
// horizontal distance to the nearest source of row y, and a source within Euclidean distance R of a pixel lies in the
// (2R+1)^2 window around it.  Per tile of 32 x 256 pixels (one thread per column):
//   phase 1  hx(y,j) for the tile's rows and R rows either side, straight from the source BIT words: the 32 bits around
//            a pixel come out of the row's words with one v_alignbit, the nearest set bit left / right with
//            v_ffbh / v_ffbl.  hx^2 goes to LDS (a byte per pixel for R <= 10), "that source is right of the pixel" to a
//            bit plane (left wins a tie: the smaller column).
//   phase 2  per pixel the minimum over the 2R+1 rows of (hx^2 + dy^2) << DB | (dy + R): an unsigned min whose low bits
//            carry the row, so ties resolve to the smallest source row, then (phase 1) the smallest column -- the smallest
//            raster index, the order brute force gives.  Four rows per thread share their LDS reads; the depth gathers
//            of one group of four are in flight while the next is computed.
//   A pixel whose minimum is <= R^2 is exact (everything that near was inside the window): rank -> label, gather, store.
//   The others (no source within R) go on the frame's list for k_l2far, one wave per pixel.
// R = 10 for the frames k_frame routes 16 (at 5 % density one pixel in 10^7 has no source that near), R = 15 for route 32;
// frames with route 0 are left to k_colT + k_l2env, and so are the rows of a window-kernel frame with many far pixels.
// ------------------------------------------------------------------------------------------------
constexpr int W2_TH = 32, W2_TW = 256;
// a row with this many far pixels (an eighth of it: the empty sky of a LiDAR frame, not the scattered voids of a uniform one)
// is redone whole by k_l2env instead of pixel by pixel (k_l2far)
__device__ __forceinline__ u32 w2_row_t(int W) { return (u32)max(32, W >> 3); }

// ------------------------------------------------------------------------------------------------
// A pixel with no source inside its window ("far"), one HALF WAVE per pixel: lane l takes the rows i - (base + l) and
// i + (base + l), finds in each the nearest source left and right of the pixel's column by scanning the row's bit words,
// and the half reduces (d2, source row << 16 | source column) to its minimum; base advances by 32 until base^2 exceeds the
// best squared distance.  Any distance, exact, canonical ties (smallest source row, then column).
// ------------------------------------------------------------------------------------------------
// One row of the search, in two steps so that the lanes can agree on a bound in between: FIRST only the bit word that holds
// column j (one load; most pixels find their nearest source's neighbourhood there), REST the words left and right of it as far
// as the bound allows.  need: bit 0 / 1 = the left / right side of this row is still open after FIRST.
template <bool FIRST>
__device__ __forceinline__ void l2far_row(const u64 *__restrict__ row, int Wd, int y, int j, u32 dy2, u32 &bestd2, u32 &bestrc,
                                          u32 &need) {
    const int wj = j >> 6;
    auto offer = [&](int col) {
        const u32 d2 = dy2 + (u32)((j - col) * (j - col));
        const u32 rc = (u32)y << 16 | (u32)col;
        if (d2 < bestd2 || (d2 == bestd2 && rc < bestrc)) {
            bestd2 = d2;
            bestrc = rc;
        }
    };
    if (FIRST) {
        const u64 w0 = row[wj];
        const u64 lb = w0 & (~0ull >> (63 - (j & 63))), rb = w0 & ((~0ull << (j & 63)) << 1);
        need = 0u;
        if (lb)
            offer(wj * 64 + 63 - __clzll((long long)lb));
        else
            need |= 1u;
        if (rb)
            offer(wj * 64 + __ffsll((long long)rb) - 1);
        else
            need |= 2u;
        return;
    }
    if (need & 1u) {  // left of the word of column j
        for (int w = wj - 1; w >= 0; --w) {
            const u32 dm = (u32)(j - (w * 64 + 63));
            if (dy2 + dm * dm > bestd2) break;
            const u64 bits = row[w];
            if (bits) {
                offer(w * 64 + 63 - __clzll((long long)bits));
                break;
            }
        }
    }
    if (need & 2u) {  // right of it
        for (int w = wj + 1; w < Wd; ++w) {
            const u32 dm = (u32)(w * 64 - j);
            if (dy2 + dm * dm > bestd2) break;
            const u64 bits = row[w];
            if (bits) {
                offer(w * 64 + __ffsll((long long)bits) - 1);
                break;
            }
        }
    }
}

// Two pixels per wave: lanes 0-31 search for pixel (i, j) of their half, lanes 32-63 for theirs (the two may be the same
// pixel); lane l of a half takes the rows i - (base + l) and i + (base + l), base advancing by 32.  The first lane of each
// half stores the three outputs.
__device__ __forceinline__ void l2far_pixel(const float *__restrict__ x, const u64 *__restrict__ srcbits,
                                            const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                            const float *__restrict__ vlist, int b, int H, int W, int Wd, int i, int j,
                                            int nval, int misaligned, float *__restrict__ out_depth,
                                            float *__restrict__ out_dt, int32_t *__restrict__ out_index,
                                            int *__restrict__ frame_status) {
    const int hl = threadIdx.x & 31;
    const size_t fo = (size_t)b * H * W;
    u32 bestd2 = 0xFFFFFFFFu, bestrc = 0xFFFFFFFFu;
    for (int base = 0; base < H; base += 32) {       // a routed frame has sources: the loop ends with a finite best
        // base^2 == bestd2 may still hold a smaller source row; the wave goes on while either half has rows to look at
        if (!__any((u32)(base * base) <= bestd2)) break;
        const int dy = base + hl;
        const u32 dy2 = (u32)(dy * dy);
        const u64 *rup = srcbits + ((size_t)b * H + max(i - dy, 0)) * Wd, *rdn = srcbits + ((size_t)b * H + min(i + dy, H - 1)) * Wd;
        const bool up = i - dy >= 0 && dy2 <= bestd2, dn = dy > 0 && i + dy < H && dy2 <= bestd2;
        auto agree = [&]() {  // minimum of (d2, row << 16 | column) over the half wave
            u32 m = bestd2;
#pragma unroll
            for (int o = 16; o; o >>= 1) m = min(m, (u32)__shfl_xor((int)m, o));
            u32 r = bestd2 == m ? bestrc : 0xFFFFFFFFu;
#pragma unroll
            for (int o = 16; o; o >>= 1) r = min(r, (u32)__shfl_xor((int)r, o));
            bestd2 = m;
            bestrc = r;
        };
        u32 nu = 0u, nd = 0u;
        if (up) l2far_row<true>(rup, Wd, i - dy, j, dy2, bestd2, bestrc, nu);
        if (dn) l2far_row<true>(rdn, Wd, i + dy, j, dy2, bestd2, bestrc, nd);
        agree();  // the bound of the half: most lanes have nothing left to look at
        if (up && dy2 <= bestd2) l2far_row<false>(rup, Wd, i - dy, j, dy2, bestd2, bestrc, nu);
        if (dn && dy2 <= bestd2) l2far_row<false>(rdn, Wd, i + dy, j, dy2, bestd2, bestrc, nd);
        agree();
    }
    if (hl == 0) {
        const int p = i * W + j;
        int label = 0, q = p;
        float dist = INFINITY;
        if (bestd2 != 0xFFFFFFFFu) {
            const int srow = (int)(bestrc >> 16), scol = (int)(bestrc & 0xFFFF);
            q = srow * W + scol;
            const size_t w = ((size_t)b * H + srow) * Wd + (scol >> 6);
            label = source_rank(rowbase_s[(size_t)b * H + srow] + wpre_s[w], srcbits[w], scol);
            dist = sqrtf((float)bestd2);
        }
        if (out_index) out_index[fo + p] = label;
        if (out_dt) out_dt[fo + p] = dist;
        if (out_depth) out_depth[fo + p] = gather_depth(x + fo, vlist + fo, label, q, nval, misaligned, frame_status + b);
    }
}

template <int R>
struct L2Win {
    static_assert(R >= 1 && R <= 15, "the 32-bit window holds 2R + 1 <= 31 columns");
    static constexpr int NR = W2_TH + 2 * R;                     // rows of the tile + halo
    static constexpr int DB = R <= 7 ? 4 : 5;                    // bits of the row field of the key
    static constexpr u32 NONE = (u32)((R + 1) * (R + 1));        // "no source within R in this row": worse than any decided result
    static constexpr bool BYTE = NONE <= 255;                    // R <= 10: the entries fit a byte
    static constexpr int ESZ = BYTE ? 1 : 2;
    // LDS: entries [NR][256], row bits for the windows [NR][4] x 16 B, the words 4tx-1 .. 4tx+4 [NR][6] x 8 B, the number of
    // sources before each of them [NR][6] x 4 B, the side bits [NR][4] x 8 B, sqrt table [R^2 + 1]
    static constexpr size_t OFF_X = (size_t)NR * W2_TW * ESZ, OFF_W = OFF_X + (size_t)NR * 64, OFF_BASE = OFF_W + (size_t)NR * 48,
                            OFF_SIDE = OFF_BASE + (size_t)NR * 24, OFF_SQRT = OFF_SIDE + (size_t)NR * 32,
                            LDS = OFF_SQRT + (size_t)(R * R + 1) * 4;
};

template <int R>
__global__ __launch_bounds__(256) void k_l2win(const float *__restrict__ x, const u64 *__restrict__ srcbits,
                                               const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                               int *__restrict__ finfo, const float *__restrict__ vlist, u32 *__restrict__ xlist,
                                               const int *__restrict__ route, int want_route, u32 *__restrict__ rowfar,
                                               int *__restrict__ fflag2, int H, int W, int Wd, int tiles_x,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    using C = L2Win<R>;
    using entry_t = typename std::conditional<C::BYTE, u8, u16>::type;
    constexpr int NR = C::NR, DB = C::DB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = blockIdx.y;
    if (route[b] != want_route) {
        // a frame with a handful of sources (ROUTE_POINTS): this launch's blocks, idle for it, write the list of its sources in
        // raster order (list index = label - 1; pixel offsets) for l2pts_tile -- each block a share of the frame's bit words
        if (R == W2_R32 && route[b] == ROUTE_POINTS) {  // (compiled into the route-32 instance only: the other one is the hot one)
            u32 *srclist = xlist;  // such a frame has no far list: its slice of that buffer holds the source list
            const int nwords = H * Wd, per = (nwords + (int)gridDim.x - 1) / (int)gridDim.x;
            u32 *sl = srclist + (size_t)b * H * W;
            for (int w = (int)blockIdx.x * per + (int)threadIdx.x; w < min(nwords, ((int)blockIdx.x + 1) * per); w += 256) {
                u64 sb = srcbits[(size_t)b * nwords + w];
                if (!sb) continue;
                const int i = w / Wd, j0 = (w - i * Wd) * 64;
                u32 k = rowbase_s[(size_t)b * H + i] + wpre_s[(size_t)b * nwords + w];
                while (sb) {
                    const int bit = __ffsll((long long)sb) - 1;
                    sb &= sb - 1;
                    sl[k++] = (u32)(i * W + j0 + bit);
                }
            }
        }
        return;
    }
    const int ty0 = blockIdx.x / tiles_x;
    // rows k_frame (too far from every row with a source: the sky) or other blocks (too many far pixels) have handed to the
    // row search: nothing of them is computed, stored or counted here; a tile that holds no other row is done
    u64 rowgone = 0ull;
    if (fflag2[b]) {  // (block-uniform; a frame without such rows does not pay for the look)
        const int l = threadIdx.x & 63, row = ty0 * W2_TH + l;
        rowgone = __ballot(l < W2_TH && (row >= H || rowfar[(size_t)b * H + min(row, H - 1)] >= w2_row_t(W)));
        if ((u32)rowgone == 0xFFFFFFFFu) return;  // block-uniform: every wave computes the same mask
    }
    entry_t *s_h = reinterpret_cast<entry_t *>(lds);
    uint4 *s_x = reinterpret_cast<uint4 *>(lds + C::OFF_X);
    u64 *s_w = reinterpret_cast<u64 *>(lds + C::OFF_W);
    u32 *s_base = reinterpret_cast<u32 *>(lds + C::OFF_BASE);
    u64 *s_side = reinterpret_cast<u64 *>(lds + C::OFF_SIDE);
    float *s_sqrt = reinterpret_cast<float *>(lds + C::OFF_SQRT);
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int y0 = ty * W2_TH, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t fo = (size_t)b * H * W;
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    for (int t = tid; t < NR * 6; t += 256) {
        const int yy = t / 6, c = t - yy * 6;
        const int y = y0 - R + yy, wd = tx * 4 - 1 + c;
        u64 w = 0;
        u32 base = 0;
        if (y >= 0 && y < H && wd >= 0 && wd < Wd) {
            const size_t wi = ((size_t)b * H + y) * Wd + wd;
            w = srcbits[wi];
            base = rowbase_s[(size_t)b * H + y] + wpre_s[wi];
        }
        s_w[t] = w;
        s_base[t] = base;
    }
    if (tid <= R * R) s_sqrt[tid] = sqrtf((float)tid);  // exact square roots of the distances a window can decide
    __syncthreads();
    for (int t = tid; t < NR * 4; t += 256) {
        const int yy = t >> 2, c = t & 3;
        const u64 prv = s_w[yy * 6 + c], cur = s_w[yy * 6 + c + 1], nxt = s_w[yy * 6 + c + 2];
        const u64 lo = (prv >> (64 - R)) | (cur << R), hi = (cur >> (64 - R)) | (nxt << R);
        s_x[t] = make_uint4((u32)lo, (u32)(lo >> 32), (u32)hi, 0u);
    }
    __syncthreads();
    // phase 1: this thread's column, every row of the tile + halo (four rows at a time: their LDS reads are in flight together).
    // entry = hx^2 (NONE beyond R); which side the row's nearest source is on goes into a bit plane (left on a tie)
    const bool upper = lane >= 32;
    for (int yy = 0; yy < NR; yy += 4) {  // NR is even, not always a multiple of 4
        uint4 xx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xx[u] = s_x[min(yy + u, NR - 1) * 4 + wv];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const u32 a = upper ? xx[u].y : xx[u].x, bb = upper ? xx[u].z : xx[u].y;
            const u32 win = __builtin_amdgcn_alignbit(bb, a, (u32)lane);  // row bits of columns j-R .. j-R+31; the pixel is bit R
            const u32 dxl = ffbh_u32(win << (31 - R));                 // nearest source at or left of the pixel (<= R, or none)
            const u32 dxr = ffbl_b32(win >> R);                        // ... at or right of it
            const u32 dx = min(dxl, dxr);
            const u64 right = __ballot(dxr < dxl);
            if (yy + u < NR) {
                s_h[(yy + u) * W2_TW + tid] = (entry_t)(dx <= (u32)R ? __umul24(dx, dx) : C::NONE);
                if (lane == 0) s_side[(yy + u) * 4 + wv] = right;
            }
        }
    }
    __syncthreads();
    const int j = tx * W2_TW + tid;
    const bool inw = j < W;
    const float *gsrc = misaligned ? vlist + fo : x + fo;
    bool index_error = false;
    // phase 2, four rows per step, software-pipelined: the depth gathers of step g are in flight while step g + 1 is
    // computed; a step's stores follow one step later.  key = (hx^2 + dy^2) << DB | (dy + R)
    struct Step {
        int label[4];
        float dist[4], dep[4];
        bool live[4], gok[4];
    };
    auto compute = [&](int t0, Step &S) {
        u32 e[2 * R + 4];
#pragma unroll
        for (int k = 0; k < 2 * R + 4; ++k) e[k] = s_h[(t0 + k) * W2_TW + tid];
        u32 best[4];
        bool far[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            u32 m = 0xFFFFFFFFu;
#pragma unroll
            for (int dyi = 0; dyi + 1 < 2 * R + 1; dyi += 2)
                m = min3u(m, (e[u + dyi] << DB) + (u32)(((dyi - R) * (dyi - R)) << DB | dyi),
                          (e[u + dyi + 1] << DB) + (u32)(((dyi + 1 - R) * (dyi + 1 - R)) << DB | (dyi + 1)));
            best[u] = min(m, (e[u + 2 * R] << DB) + (u32)((R * R) << DB | (2 * R)));
            far[u] = inw & (y0 + t0 + u < H) & ((best[u] >> DB) > (u32)(R * R)) & !((rowgone >> (t0 + u)) & 1ull);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // a pixel with no source within R: counted per row; a few go on k_l2far's list, 32 or more of
            const u64 fm = __ballot(far[u]);  // a row make k_l2env redo the row (the empty sky of a LiDAR frame)
            if (fm) {
                const u32 n = (u32)__popcll(fm);
                u32 base = 0xFFFFFFFFu;  // lane 0: where the wave's pixels go on the list, or "the row is k_l2env's by now"
                if (lane == 0) {
                    const u32 before = atomicAdd(&rowfar[(size_t)b * H + y0 + t0 + u], n), t = w2_row_t(W);
                    if (before < t && before + n >= t) fflag2[b] = 1;  // the frame needs k_colT's column distances
                    if (before + n < t) base = (u32)atomicAdd(&finfo[b * FI_STRIDE + FI_NUNRES], (int)n);
                }
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                if (far[u] && base != 0xFFFFFFFFu)
                    xlist[fo + base + __builtin_amdgcn_mbcnt_hi((u32)(fm >> 32), __builtin_amdgcn_mbcnt_lo((u32)fm, 0u))] =
                        (u32)((y0 + t0 + u) * W + j);
            }
        }
        int goff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = y0 + t0 + u;
            S.live[u] = inw & (i < H) & !far[u] & !((rowgone >> (t0 + u)) & 1ull);
            const u32 d2 = min(best[u] >> DB, (u32)(R * R));  // clamped for the table (a far pixel stores nothing)
            const int dyi = (int)(best[u] & ((1u << DB) - 1u)), dy = dyi - R;
            const int dx2 = max((int)d2 - dy * dy, 0);
            S.dist[u] = s_sqrt[d2];
            const int dx = (int)s_sqrt[dx2];
            const int yy = t0 + u + dyi;
            const int sc = ((s_side[yy * 4 + wv] >> lane) & 1ull) ? j + dx : j - dx;
            const int c = min(max((sc >> 6) - (tx * 4 - 1), 0), 5);
            S.label[u] = source_rank(s_base[yy * 6 + c], s_w[yy * 6 + c], sc);
            // depth_list[label - 1] with numpy's index semantics (tools.py:26): the label is >= 1 here
            S.gok[u] = S.label[u] - 1 < nval;
            goff[u] = misaligned ? S.label[u] - 1 : (i + dy) * W + sc;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) S.dep[u] = (out_depth && S.live[u] && S.gok[u]) ? gsrc[goff[u]] : nanf("");
    };
    auto store = [&](int t0, const Step &S) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = fo + (size_t)(y0 + t0 + u) * W + j;
            if (S.live[u]) {
                if (out_index) __builtin_nontemporal_store(S.label[u], &out_index[o]);
                if (out_dt) __builtin_nontemporal_store(S.dist[u], &out_dt[o]);
                if (out_depth) __builtin_nontemporal_store(S.dep[u], &out_depth[o]);
                index_error |= !S.gok[u];
            }
        }
    };
    Step cur, prev;
    compute(0, prev);
#pragma unroll
    for (int t0 = 4; t0 < W2_TH; t0 += 4) {
        compute(t0, cur);
        store(t0 - 4, prev);
        prev = cur;
    }
    store(W2_TH - 4, prev);
    if (out_depth && index_error) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
}

// ------------------------------------------------------------------------------------------------
// The far pixels k_l2win put on the frame's list, one wave per pixel (l2far_pixel); run by k_l2far.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l2far_list(const float *__restrict__ x, const u64 *__restrict__ srcbits,
                                           const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                           const int *__restrict__ finfo, const float *__restrict__ vlist,
                                           const u32 *__restrict__ xlist, const u32 *__restrict__ rowfar, int b, int blk, int nblk,
                                           int H, int W, int Wd, float *__restrict__ out_depth, float *__restrict__ out_dt,
                                           int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    const int n = finfo[b * FI_STRIDE + FI_NUNRES];
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const int lane = threadIdx.x & 63;
    const u32 t = w2_row_t(W);
    // the list is dealt out to the waves entry by entry (neighbours on the list are neighbours in the frame and cost alike); a
    // wave looks 64 of its entries up at a time (one lane each), drops those whose row has been handed to the envelope search
    // since, and searches for the rest one after the other, the whole wave per pixel
    const int wpb = blockDim.x >> 6, nwaves = nblk * wpb, w0 = blk * wpb + (threadIdx.x >> 6);
    for (int e0 = 0; w0 + e0 * nwaves < n; e0 += 64) {
        const int e = w0 + (e0 + lane) * nwaves;
        const int p = e < n ? (int)xlist[(size_t)b * H * W + e] : -1;
        u64 keep = __ballot(p >= 0 && rowfar[(size_t)b * H + max(p, 0) / W] < t);
        while (keep) {  // two pixels at a time, one per half wave (the last one of an odd count twice)
            const int l0 = __ffsll((long long)keep) - 1;
            keep &= keep - 1;
            const int l1 = keep ? __ffsll((long long)keep) - 1 : l0;
            keep &= keep - 1;  // (0 stays 0)
            const int pp = __shfl(p, lane < 32 ? l0 : l1), i = pp / W;
            l2far_pixel(x, srcbits, wpre_s, rowbase_s, vlist, b, H, W, Wd, i, pp - i * W, nval, misaligned, out_depth, out_dt, out_index,
                        frame_status);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// l2 metric, every other frame: k_l2env -- the row pass of the two-pass Euclidean transform (Felzenszwalb / Meijster) as a
// search over the lower envelope of parabolas.  With g(k) the vertical distance from row i to the nearest source of column
// k (k_colT's band words), pixel j takes min_k g(k)^2 + (j-k)^2: the lowest of the parabolas rooted at the columns.  Two
// parabolas of the same curvature cross once, so the column that owns pixel j never moves left as j moves right, whatever
// rule breaks the ties.  That monotonicity replaces the sequential stack: solve j = 0 and j = W-1 over all columns, then
// the midpoints level by level -- the owner of a midpoint lies between the owners of its solved neighbours -- so a level
// looks at every column about once: O(W log W) per row, any distance, no data-dependent radius.
//   * one wave per row; the columns that hold a source are compacted into LDS as {g^2, source row << 16 | column};
//   * a level's queries are shared out over the wave: 64 / queries lanes per query (candidates strided over the lanes,
//     minimum by wave shuffles), one lane per query once there are 64 or more;
//   * keys compare as (d2, source row, column): ties go to the smallest raster index of the source, as brute force does.
// ------------------------------------------------------------------------------------------------
struct L2Cand {
    u32 hi, lo;  // squared distance; source row << 16 | position in the compacted column list (same order as the columns)
};
__device__ __forceinline__ void l2env_offer(L2Cand &best, const uint2 cv, int cc, int j) {
    const int dk = j - (int)(cv.y & 0xFFFFu);
    const u32 hi = cv.x + (u32)(dk * dk), lo = (cv.y & 0xFFFF0000u) | (u32)cc;
    if (hi < best.hi || (hi == best.hi && lo < best.lo)) {
        best.hi = hi;
        best.lo = lo;
    }
}
__device__ __forceinline__ void l2env_merge(L2Cand &best, int off) {
    const u32 oh = (u32)__shfl_xor((int)best.hi, off), ol = (u32)__shfl_xor((int)best.lo, off);
    if (oh < best.hi || (oh == best.hi && ol < best.lo)) {
        best.hi = oh;
        best.lo = ol;
    }
}

__host__ __device__ constexpr size_t l2env_lds(int W) { return ((size_t)W * 10 + 15) & ~(size_t)15; }  // LDS of one row's search
// NW waves per image row.  k_l2env uses NW = 1: the per-level bookkeeping is per wave, so one wave per row costs the fewest
// instructions (a 256-thread block per row measured 1.8 x the time on whole sparse frames); NW = 4 shortens a single row's
// chain and pays when only a few rows are searched.
template <int NW>
__device__ __forceinline__ void l2env_sync() {
    if (NW == 1)
        __builtin_amdgcn_wave_barrier();  // one wave: its LDS operations complete in order
    else
        __syncthreads();
}
template <int NW>
__device__ __forceinline__ void l2env_row(const float *__restrict__ x, const uint2 *__restrict__ ct, int CTP, int nb,
                                          const uint4 *__restrict__ rec, int Wd, const int *__restrict__ finfo,
                                          const float *__restrict__ vlist, int H, int W, int b, int i,
                                          float *__restrict__ out_depth, float *__restrict__ out_dt,
                                          int32_t *__restrict__ out_index, int *__restrict__ frame_status,
                                          unsigned char *s_env, u32 *s_wcnt) {
    constexpr int NT = 64 * NW;
    uint2 *s_c = reinterpret_cast<uint2 *>(s_env);               // [W] the columns with a source, in column order: {g^2, row << 16 | column}
    u16 *s_own = reinterpret_cast<u16 *>(s_env + (size_t)W * 8);  // [W] owner of every solved pixel (index into s_c)
    const int tid = NW == 1 ? (int)(threadIdx.x & 63) : (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t fo = (size_t)b * H * W;
    int c = 0;  // uniform over the row's waves
    {
        const int band = i >> 5, r = i & 31;
        const uint2 *crow = ct + ((size_t)b * nb + band) * CTP;
        const u32 upmask = (2u << r) - 1u;
        for (int k0 = 0; k0 < W; k0 += NT) {
            const int k = k0 + tid;
            bool has = false;
            uint2 cv = make_uint2(0u, 0u);
            if (k < W) {
                const uint2 w = crow[k];
                const u32 gu = min(ffbh_u32(w.x & upmask) + (u32)(r - 31), (w.y & 0xFFFFu) + (u32)r);
                const u32 gd = min(ffbl_b32(w.x >> r), (w.y >> 16) + (u32)(31 - r));
                const u32 m = min(gu, gd);  // on a vertical tie the upper source (the smaller raster index)
                has = m < (u32)MAX_HW_SUM;
                cv = make_uint2(m * m, (u32)(gd < gu ? i + (int)m : i - (int)m) << 16 | (u32)k);
            }
            const u64 bal = __ballot(has);
            int off = c, tot = __popcll(bal);
            if (NW > 1) {
                if (lane == 0) s_wcnt[wv] = (u32)tot;
                l2env_sync<NW>();
                tot = 0;
                for (int w = 0; w < NW; ++w) {
                    if (w < wv) off += (int)s_wcnt[w];
                    tot += (int)s_wcnt[w];
                }
            }
            if (has) s_c[off + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = cv;
            c += tot;
            if (NW > 1) l2env_sync<NW>();
        }
    }
    l2env_sync<NW>();
    if (c > 0) {
        // a level: nq queries j = (2m + 1) s, each between the owners of its solved neighbours; G lanes per query
        auto level = [&](int nq, int s, bool ends) {
            int lg = 6;  // log2 G: G = threads / (nq rounded up to a power of two), at most a wave, at least one lane per query
            while (lg > 0 && (nq << lg) > NT) --lg;
            const int G = 1 << lg, per = NT >> lg, gl = lane & (G - 1);
            for (int m0 = 0; m0 < nq; m0 += per) {
                const int m = m0 + (tid >> lg);
                const bool act = m < nq;
                const int j = ends ? (m ? W - 1 : 0) : (2 * m + 1) * s;
                L2Cand best = {0xFFFFFFFFu, 0xFFFFFFFFu};
                if (act) {
                    const int lo = ends ? 0 : (int)s_own[j - s], hi = ends ? c - 1 : (int)s_own[min(j + s, W - 1)];
                    int cc = lo + gl;
                    for (; cc + 3 * G <= hi; cc += 4 * G) {  // four candidates per step: their LDS reads are in flight together
                        const uint2 c0 = s_c[cc], c1 = s_c[cc + G], c2 = s_c[cc + 2 * G], c3 = s_c[cc + 3 * G];
                        l2env_offer(best, c0, cc, j);
                        l2env_offer(best, c1, cc + G, j);
                        l2env_offer(best, c2, cc + 2 * G, j);
                        l2env_offer(best, c3, cc + 3 * G, j);
                    }
                    for (; cc <= hi; cc += G) l2env_offer(best, s_c[cc], cc, j);
                }
                for (int o = G >> 1; o; o >>= 1) l2env_merge(best, o);
                if (act && gl == 0) s_own[j] = (u16)(best.lo & 0xFFFFu);
            }
            l2env_sync<NW>();
        };
        level(W > 1 ? 2 : 1, 0, true);  // pixels 0 and W-1 over every column
        int s = 1;
        while (s < W - 1) s <<= 1;  // s >= W - 1: the first stride below it has its odd multiples inside (0, W-1)
        for (s >>= 1; s >= 1; s >>= 1) {
            const int nq = W - 2 >= s ? ((W - 2) / s + 1) / 2 : 0;  // odd multiples of s in [s, W-2]
            if (nq) level(nq, s, false);
        }
    }
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const float *gsrc = misaligned ? vlist + fo : x + fo;
    bool index_error = false;
    for (int j0 = 0; j0 < W; j0 += 4 * NT) {  // four pixels per lane and step: their gathers are in flight together
        int q[4], label[4], sr[4], sc[4];
        float dist[4], dep[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            sr[u] = sc[u] = 0;
            const int j = j0 + NT * u + tid;
            in[u] = j < W;
            q[u] = i * W + min(j, W - 1);
            dist[u] = INFINITY;
            label[u] = 0;
            if (c > 0) {
                const uint2 cv = s_c[s_own[min(j, W - 1)]];
                const int scol = (int)(cv.y & 0xFFFFu), dk = min(j, W - 1) - scol;
                q[u] = (int)(cv.y >> 16) * W + scol;
                sr[u] = (int)(cv.y >> 16);
                sc[u] = scol;
                dist[u] = sqrtf((float)(cv.x + (u32)(dk * dk)));
            }
        }
        if (c > 0 && (out_index || out_depth)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) label[u] = label_from_rec(reinterpret_cast<const uint2 *>(rec + (size_t)b * Wd * H), H, sr[u], sc[u]);  // k_colT wrote the rank records
        }
        if (out_depth) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // depth_list[label - 1] with numpy's index semantics (tools.py:26)
                int idx = label[u] - 1;
                if (idx < 0) idx += nval;
                const bool ok = idx >= 0 && idx < nval;
                dep[u] = ok ? gsrc[misaligned ? idx : q[u]] : nanf("");
                index_error |= in[u] && !ok;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const size_t o = fo + (size_t)i * W + j0 + NT * u + tid;
            if (in[u]) {  // whole 256-byte runs per store instruction, never read again in this pass: streaming stores
                if (out_index) __builtin_nontemporal_store((int32_t)label[u], &out_index[o]);
                if (out_dt) __builtin_nontemporal_store(dist[u], &out_dt[o]);
                if (out_depth) __builtin_nontemporal_store(dep[u], &out_depth[o]);
            }
        }
    }
    if (out_depth && index_error) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
}

// ------------------------------------------------------------------------------------------------
// A row of far pixels of a window-kernel frame -- the empty sky of a LiDAR frame: every column has its source a long way down,
// the nearest of them all differ little.  There the owner of pixel j is a column NEAR j, and the search over the lower
// envelope (above: a few thousand instructions of bookkeeping per row) is replaced by a window in the COLUMN distances:
//     best(j) = min over |k - j| <= R of g(k)^2 + (j - k)^2,
// exact as soon as best(j) < gmin^2 + (R + 1)^2 (gmin = the row's smallest column distance: no column outside the window can
// do better, or as well).  R starts at L2S_R and grows by L2S_R for the pixels that fail the test (none to speak of on a sky
// over ring rows); a window that has grown over the whole row is exact by exhaustion.  Keys compare as (d2, source row,
// column): ties go to the smallest raster index of the source, as brute force gives.  One wave per row; the columns' {g^2,
// source row << 16 | column} are staged in LDS with a margin of "no column" on either side.
// ------------------------------------------------------------------------------------------------
constexpr int L2S_R = 16;
__device__ __forceinline__ void l2sky_row(const float *__restrict__ x, const uint2 *__restrict__ ct, int CTP, int nb,
                                          const uint4 *__restrict__ rec, int Wd, const int *__restrict__ finfo,
                                          const float *__restrict__ vlist, int H, int W, int b, int i, float *__restrict__ out_depth,
                                          float *__restrict__ out_dt, int32_t *__restrict__ out_index, int *__restrict__ frame_status,
                                          unsigned char *s_env) {
    uint2 *s_col = reinterpret_cast<uint2 *>(s_env) + L2S_R;  // [-L2S_R, W + L2S_R): {g^2, source row << 16 | column}; 0x7FFFFFFF: no source / outside the row
    const int lane = threadIdx.x & 63;
    const size_t fo = (size_t)b * H * W;
    u32 gmin = 0xFFFFFFFFu;
    {
        const int band = i >> 5, r = i & 31;
        const uint2 *crow = ct + ((size_t)b * nb + band) * CTP;
        const u32 upmask = (2u << r) - 1u;
        if (lane < 2 * L2S_R) s_col[lane < L2S_R ? lane - L2S_R : W + lane - L2S_R] = make_uint2(0x7FFFFFFFu, 0xFFFFFFFFu);
        for (int k = lane; k < W; k += 64) {
            const uint2 w = crow[k];
            const u32 gu = min(ffbh_u32(w.x & upmask) + (u32)(r - 31), (w.y & 0xFFFFu) + (u32)r);
            const u32 gd = min(ffbl_b32(w.x >> r), (w.y >> 16) + (u32)(31 - r));
            const u32 m = min(gu, gd);  // on a vertical tie the upper source (the smaller raster index)
            const bool has = m < (u32)MAX_HW_SUM;
            s_col[k] = make_uint2(has ? m * m : 0x7FFFFFFFu, (u32)(gd < gu ? i + (int)m : i - (int)m) << 16 | (u32)k);
            gmin = min(gmin, has ? m : 0xFFFFFFFFu);
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) gmin = min(gmin, (u32)__shfl_xor((int)gmin, o));
    }
    __builtin_amdgcn_wave_barrier();
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const float *gsrc = misaligned ? vlist + fo : x + fo;
    const bool any = gmin != 0xFFFFFFFFu;  // the frame has a source at all (a routed frame does)
    const u32 gmin2 = any ? gmin * gmin : 0u;
    bool index_error = false;
    auto search = [&](int j) -> unsigned long long {
        const bool in = j < W;
        unsigned long long best = ~0ull;
        // first the 2 L2S_R + 1 columns around the pixel, all lanes together, no bounds (the margins hold "no column": a key
        // that loses against every real one); 0x7FFFFFFF + (j - k)^2 stays below 2^32
        const int jc = min(j, W - 1);
        uint2 cv[2 * L2S_R + 1];
#pragma unroll
        for (int t = 0; t <= 2 * L2S_R; ++t) cv[t] = s_col[jc - L2S_R + t];
#pragma unroll
        for (int t = 0; t <= 2 * L2S_R; ++t) {
            const unsigned long long key = (unsigned long long)(cv[t].x + (u32)((t - L2S_R) * (t - L2S_R))) << 32 | cv[t].y;
            best = key < best ? key : best;
        }
        // then, for the pixels that are not sure yet, ring by ring; a lane that has its answer (or has seen the whole row)
        // stops taking part
        int R = L2S_R;
        bool open = in && (u32)(best >> 32) >= gmin2 + (u32)((R + 1) * (R + 1)) && (j - R > 0 || j + R < W - 1);
        while (__any(open)) {
            if (open) {
                for (int k = max(j - R - L2S_R, 0); k <= min(j - R - 1, W - 1); ++k) {
                    const uint2 c2 = s_col[k];
                    const unsigned long long key = (unsigned long long)(c2.x + (u32)((j - k) * (j - k))) << 32 | c2.y;
                    best = key < best ? key : best;
                }
                for (int k = max(j + R + 1, 0); k <= min(j + R + L2S_R, W - 1); ++k) {
                    const uint2 c2 = s_col[k];
                    const unsigned long long key = (unsigned long long)(c2.x + (u32)((j - k) * (j - k))) << 32 | c2.y;
                    best = key < best ? key : best;
                }
                R += L2S_R;
                open = (u32)(best >> 32) >= gmin2 + (u32)((R + 1) * (R + 1)) && (j - R > 0 || j + R < W - 1);
            }
        }
        return best;
    };
    // four groups of 64 pixels per step: their label and depth gathers are in flight together
    for (int j0 = 0; j0 < W; j0 += 256) {
        unsigned long long best[4];
        int q[4], label[4];
        float dep[4];
        bool in[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = j0 + 64 * u + lane;
            in[u] = j < W;
            best[u] = (any && j0 + 64 * u < W) ? search(j) : ~0ull;  // (the group test is wave-uniform)
            q[u] = any ? (int)((u32)best[u] >> 16) * W + (int)((u32)best[u] & 0xFFFFu) : 0;
            label[u] = 0;
            dep[u] = 0.0f;
        }
        if (any && (out_index || out_depth)) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (in[u]) label[u] = label_from_rec(reinterpret_cast<const uint2 *>(rec + (size_t)b * Wd * H), H, (int)((u32)best[u] >> 16), (int)((u32)best[u] & 0xFFFFu));  // k_colT wrote the rank records
        }
        if (out_depth) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // depth_list[label - 1] with numpy's index semantics (tools.py:26)
                int idx = label[u] - 1;
                if (idx < 0) idx += nval;
                const bool ok = idx >= 0 && idx < nval;
                dep[u] = (in[u] && ok) ? gsrc[misaligned ? idx : q[u]] : nanf("");
                index_error |= in[u] && !ok;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (!in[u]) continue;
            const size_t o = fo + (size_t)i * W + j0 + 64 * u + lane;
            if (out_index) __builtin_nontemporal_store((int32_t)label[u], &out_index[o]);
            if (out_dt) __builtin_nontemporal_store(any ? sqrtf((float)(u32)(best[u] >> 32)) : INFINITY, &out_dt[o]);
            if (out_depth) __builtin_nontemporal_store(dep[u], &out_depth[o]);
        }
    }
    if (out_depth && index_error) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
}

// ------------------------------------------------------------------------------------------------
// k_l2far: the far list of the window-kernel frames (l2far_list), one wave per listed pixel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_l2far(const float *__restrict__ x, const u64 *__restrict__ srcbits,
                                               const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                               const int *__restrict__ finfo, const float *__restrict__ vlist,
                                               const u32 *__restrict__ xlist, const int *__restrict__ route,
                                               const u32 *__restrict__ rowfar, int H, int W, int Wd,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    const int b = blockIdx.y;
    if (route[b] != 0)
        l2far_list(x, srcbits, wpre_s, rowbase_s, finfo, vlist, xlist, rowfar, b, (int)blockIdx.x, (int)gridDim.x, H, W, Wd, out_depth, out_dt,
                   out_index, frame_status);
}

// ------------------------------------------------------------------------------------------------
// l2, a handful of sources (route ROUTE_POINTS: at most L2_PTS_MAX in the frame -- the NYU sampling patterns): one wave per
// 32 x 32 tile.  The nearest source to a pixel p of the tile is within |p - c| + delta of p, delta = distance from the
// tile's centre c to ITS nearest source, so it lies within delta + 2 rho of c (rho = the tile's half diagonal): the wave finds
// delta over the frame's source list, compacts the sources inside that radius into LDS (a dozen of 200 in an NYU frame), and
// every pixel takes the minimum of (d2, source row, source column) over them -- exact, canonical ties, no column pass at all.
// ------------------------------------------------------------------------------------------------
constexpr int PT_T = 32;
__device__ __forceinline__ void l2pts_tile(const float *__restrict__ x, const u32 *__restrict__ srclist,
                                           const int *__restrict__ finfo, const float *__restrict__ vlist, int b, int H, int W,
                                           int ty, int tx, float *__restrict__ out_depth, float *__restrict__ out_dt,
                                           int32_t *__restrict__ out_index, int *__restrict__ frame_status, unsigned char *lds) {
    uint2 *s_cand = reinterpret_cast<uint2 *>(lds);  // {row << 16 | column, index in the source list = label - 1}
    const int lane = threadIdx.x & 63;
    const size_t fo = (size_t)b * H * W;
    const u32 *sl = srclist + fo;
    const int nsrc = finfo[b * FI_STRIDE + FI_NSRC];
    const int r0 = ty * PT_T, c0 = tx * PT_T;
    // distances from the centre in doubled coordinates (the centre is at a half pixel): exact integers below 2^30
    const int cy2 = 2 * r0 + PT_T - 1, cx2 = 2 * c0 + PT_T - 1;
    u32 dmin = 0xFFFFFFFFu;
    auto decode = [&](int k, u32 &rc) -> u32 {  // source k: row << 16 | column, squared distance from the centre (doubled)
        const int p = (int)sl[k], sr = p / W, sc = p - sr * W;
        const int dy = 2 * sr - cy2, dx = 2 * sc - cx2;
        rc = (u32)sr << 16 | (u32)sc;
        return (u32)(dy * dy + dx * dx);
    };
    for (int k = lane; k < nsrc; k += 64) {
        u32 rc;
        dmin = min(dmin, decode(k, rc));
    }
#pragma unroll
    for (int o = 32; o; o >>= 1) dmin = min(dmin, (u32)__shfl_xor((int)dmin, o));
    // |s - c| <= delta + 2 rho; doubled: 2 rho = (PT_T - 1) sqrt 2 = 43.9, twice that + a pixel of slack for the float arithmetic
    const float reach = sqrtf((float)dmin) + 2.0f * 1.4143f * (float)(PT_T - 1) + 2.0f;
    const float reach2 = reach * reach;
    int n = 0;  // wave-uniform
    for (int k0 = 0; k0 < nsrc; k0 += 64) {  // the list again (cache hits): keeping it in registers costs occupancy
        const int k = k0 + lane;
        u32 rc = 0u;
        const bool keep = k < nsrc && (float)decode(k, rc) <= reach2;
        const u64 bal = __ballot(keep);
        if (keep) s_cand[n + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = make_uint2(rc, (u32)k);
        n += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();
    // a lane: one column, 16 rows of the tile
    constexpr int NP = PT_T / 2;
    const int j = c0 + (lane & 31), i0 = r0 + (lane >> 5) * NP;
    unsigned long long best[NP];
    u32 bidx[NP];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        best[u] = ~0ull;
        bidx[u] = 0u;
    }
    for (int q = 0; q < n; ++q) {
        const uint2 cv = s_cand[q];  // the same address for every lane: a broadcast
        const int sr = (int)(cv.x >> 16), dc = (int)(cv.x & 0xFFFFu) - j;
        const u32 dc2 = (u32)(dc * dc);
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int dr = sr - (i0 + u);
            const unsigned long long key = (unsigned long long)((u32)(dr * dr) + dc2) << 32 | cv.x;
            if (key < best[u]) {
                best[u] = key;
                bidx[u] = cv.y;
            }
        }
    }
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    const float *gsrc = misaligned ? vlist + fo : x + fo;
    float dep[NP];
    bool index_error = false;
#pragma unroll
    for (int u = 0; u < NP; ++u) {  // depth_list[label - 1] (tools.py:26): the label is >= 1 here
        const u32 rc = (u32)best[u];
        const int q = (int)(rc >> 16) * W + (int)(rc & 0xFFFFu);
        const bool ok = (int)bidx[u] < nval;
        const bool in = i0 + u < H && j < W;
        dep[u] = (out_depth && in && ok) ? gsrc[misaligned ? (int)bidx[u] : q] : nanf("");
        index_error |= in && !ok;
    }
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        if (i0 + u < H && j < W) {
            const size_t o = fo + (size_t)(i0 + u) * W + j;
            if (out_index) __builtin_nontemporal_store((int32_t)bidx[u] + 1, &out_index[o]);
            if (out_dt) __builtin_nontemporal_store(sqrtf((float)(u32)(best[u] >> 32)), &out_dt[o]);
            if (out_depth) __builtin_nontemporal_store(dep[u], &out_depth[o]);
        }
    }
    if (out_depth && index_error) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
}

// ------------------------------------------------------------------------------------------------
// k_l2env: what the window kernels left, one WAVE per unit of work (the waves of a block share nothing but the launch).
// Blocks [0, nrowblk): rows -- every row of a route-0 frame, and the rows of a window-kernel frame in which k_l2win counted
// too many far pixels (l2env_row: one wave per row costs the fewest instructions -- the per-level bookkeeping is per wave --
// and leaves the most rows in flight per CU).  The blocks behind them: the 32 x 32 tiles of the frames with a handful of
// sources (l2pts_tile).
// ------------------------------------------------------------------------------------------------
template <int WPB>  // waves per block, each on a unit of its own: 4 while four rows' LDS fit 64 KB (W <= 1638), else 1
__global__ __launch_bounds__(64 * WPB, 5) void k_l2env(const float *__restrict__ x, const uint2 *__restrict__ ct, int CTP, int nb,
                                                    const uint4 *__restrict__ rec, int Wd, const int *__restrict__ finfo,
                                                    const float *__restrict__ vlist, const int *__restrict__ route,
                                                    const u32 *__restrict__ rowfar, const u32 *__restrict__ srclist, int H, int W,
                                                    int nrowblk, size_t wave_lds, int tpw, float *__restrict__ out_depth,
                                                    float *__restrict__ out_dt, int32_t *__restrict__ out_index,
                                                    int *__restrict__ frame_status) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_env[];
    // (frames along grid x: a frame's blocks on one XCD -- where k_colT's blocks of that frame left the column words -- when the
    // batch is a multiple of eight frames)
    const int b = blockIdx.x, blk = blockIdx.y, wv = threadIdx.x >> 6, r = route[b];
    unsigned char *lds = s_env + (size_t)wv * wave_lds;
    if (blk >= nrowblk) {
        if (r == ROUTE_POINTS) {
            const int tiles_x = (W + PT_T - 1) / PT_T, ntile = tiles_x * ((H + PT_T - 1) / PT_T);
            const int t0 = ((blk - nrowblk) * WPB + wv) * tpw;  // tpw tiles per wave (wide frames: fewer blocks)
            for (int t = t0; t < min(ntile, t0 + tpw); ++t) {
                l2pts_tile(x, srclist, finfo, vlist, b, H, W, t / tiles_x, t % tiles_x, out_depth, out_dt, out_index, frame_status, lds);
                __builtin_amdgcn_wave_barrier();  // the next tile reuses the candidate list
            }
        }
        return;
    }
    const int i = blk * WPB + wv;
    if (i >= H) return;
    if (r == 0)  // a sparse frame: few columns hold a source, the distances are large -- the envelope search
        l2env_row<1>(x, ct, CTP, nb, rec, Wd, finfo, vlist, H, W, b, i, out_depth, out_dt, out_index, frame_status, lds, nullptr);
    else if (r > 0 && rowfar[(size_t)b * H + i] >= w2_row_t(W))  // a row of far pixels of a dense frame (the sky): the window in column distances
        l2sky_row(x, ct, CTP, nb, rec, Wd, finfo, vlist, H, W, b, i, out_depth, out_dt, out_index, frame_status, lds);
}
