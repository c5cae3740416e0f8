// dtfill_l2.hpp -- the exact Euclidean transform (l2 metric): k_l2win + k_l2far (dense frames), k_l2row (the others)
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
// frames with route 0 are left to k_colT + k_l2row.
// ------------------------------------------------------------------------------------------------
constexpr int W2_TH = 32, W2_TW = 256;
constexpr int W2_R16 = 10, W2_R32 = 15;  // window radius for the frames k_frame routes 16 / 32

// ------------------------------------------------------------------------------------------------
// A pixel with no source inside its window ("far"), one WAVE per pixel: lane l takes the rows i - (base + l) and
// i + (base + l), finds in each the nearest source left and right of the pixel's column by scanning the row's bit words,
// and the wave reduces (d2, source row << 16 | source column) to its minimum; base advances by 64 until base^2 exceeds the
// best squared distance.  Any distance, exact, canonical ties (smallest source row, then column).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void l2far_row(const u64 *__restrict__ row, int Wd, int y, int j, u32 dy2, u32 &bestd2, u32 &bestrc) {
    const int wj = j >> 6;
    const u64 w0 = row[wj];
    auto offer = [&](int col) {
        const u32 d2 = dy2 + (u32)((j - col) * (j - col));
        const u32 rc = (u32)y << 16 | (u32)col;
        if (d2 < bestd2 || (d2 == bestd2 && rc < bestrc)) {
            bestd2 = d2;
            bestrc = rc;
        }
    };
    {  // at or left of column j
        int w = wj;
        u64 bits = w0 & (~0ull >> (63 - (j & 63)));
        for (;;) {
            if (bits) {
                offer(w * 64 + 63 - __clzll((long long)bits));
                break;
            }
            if (--w < 0) break;
            const u32 dm = (u32)(j - (w * 64 + 63));
            if (dy2 + dm * dm > bestd2) break;
            bits = row[w];
        }
    }
    {  // right of column j
        int w = wj;
        u64 bits = w0 & ((~0ull << (j & 63)) << 1);
        for (;;) {
            if (bits) {
                offer(w * 64 + __ffsll((long long)bits) - 1);
                break;
            }
            if (++w >= Wd) break;
            const u32 dm = (u32)(w * 64 - j);
            if (dy2 + dm * dm > bestd2) break;
            bits = row[w];
        }
    }
}

// the whole wave calls this with the same pixel p = i * W + j of frame b; lane 0 stores the three outputs
__device__ __forceinline__ void l2far_pixel(const float *__restrict__ x, const u64 *__restrict__ srcbits,
                                            const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                            const float *__restrict__ vlist, int b, int H, int W, int Wd, int i, int j,
                                            int nval, int misaligned, float *__restrict__ out_depth,
                                            float *__restrict__ out_dt, int32_t *__restrict__ out_index,
                                            int *__restrict__ frame_status) {
    const int lane = threadIdx.x & 63;
    const size_t fo = (size_t)b * H * W;
    u32 bestd2 = 0xFFFFFFFFu, bestrc = 0xFFFFFFFFu;
    for (int base = 0; base < H; base += 64) {       // a routed frame has sources: the loop ends with a finite best
        if ((u32)(base * base) > bestd2) break;      // base^2 == bestd2 may still hold a smaller source row
        const int dy = base + lane;
        const u32 dy2 = (u32)(dy * dy);
        if (i - dy >= 0 && dy2 <= bestd2) l2far_row(srcbits + ((size_t)b * H + (i - dy)) * Wd, Wd, i - dy, j, dy2, bestd2, bestrc);
        if (dy > 0 && i + dy < H && dy2 <= bestd2) l2far_row(srcbits + ((size_t)b * H + (i + dy)) * Wd, Wd, i + dy, j, dy2, bestd2, bestrc);
        u32 m = bestd2;  // wave minimum of (d2, row << 16 | column)
#pragma unroll
        for (int o = 32; o; o >>= 1) m = min(m, (u32)__shfl_xor((int)m, o));
        u32 r = bestd2 == m ? bestrc : 0xFFFFFFFFu;
#pragma unroll
        for (int o = 32; o; o >>= 1) r = min(r, (u32)__shfl_xor((int)r, o));
        bestd2 = m;
        bestrc = r;
    }
    if (lane == 0) {
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
                                               const int *__restrict__ route, int want_route, int H, int W, int Wd, int tiles_x,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    using C = L2Win<R>;
    using entry_t = typename std::conditional<C::BYTE, u8, u16>::type;
    constexpr int NR = C::NR, DB = C::DB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int b = blockIdx.y;
    if (route[b] != want_route) return;
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
            far[u] = inw & (y0 + t0 + u < H) & ((best[u] >> DB) > (u32)(R * R));
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // rare: a pixel with no source within R goes on k_l2far's list, one atomic per wave and row
            const u64 fm = __ballot(far[u]);
            if (fm) {
                u32 base = 0;
                if (lane == 0) base = (u32)atomicAdd(&finfo[b * FI_STRIDE + FI_NUNRES], __popcll(fm));
                base = (u32)__builtin_amdgcn_readfirstlane((int)base);
                if (far[u])
                    xlist[fo + base + __builtin_amdgcn_mbcnt_hi((u32)(fm >> 32), __builtin_amdgcn_mbcnt_lo((u32)fm, 0u))] =
                        (u32)((y0 + t0 + u) * W + j);
            }
        }
        int goff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = y0 + t0 + u;
            S.live[u] = inw & (i < H) & !far[u];
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
                if (out_index) out_index[o] = S.label[u];
                if (out_dt) out_dt[o] = S.dist[u];
                if (out_depth) out_depth[o] = S.dep[u];
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
// k_l2far: the far pixels k_l2win put on the frame's list, one wave per pixel (l2far_pixel).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_l2far(const float *__restrict__ x, const u64 *__restrict__ srcbits,
                                               const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
                                               const int *__restrict__ finfo, const float *__restrict__ vlist,
                                               const u32 *__restrict__ xlist, const int *__restrict__ route, int H, int W, int Wd,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    const int b = blockIdx.y;
    if (route[b] == 0) return;
    const int n = finfo[b * FI_STRIDE + FI_NUNRES];
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    for (int idx = blockIdx.x * 4 + (threadIdx.x >> 6); idx < n; idx += gridDim.x * 4) {  // wave-uniform
        const int p = (int)xlist[(size_t)b * H * W + idx];
        const int i = p / W;
        l2far_pixel(x, srcbits, wpre_s, rowbase_s, vlist, b, H, W, Wd, i, p - i * W, nval, misaligned, out_depth, out_dt, out_index,
                    frame_status);
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
//   * one block per row; the columns that hold a source are compacted into LDS as {g^2, source row << 16 | column};
//   * a level's queries are shared out over the block: 256 / queries lanes per query (candidates strided over the lanes,
//     minimum by wave shuffles), one lane per query once there are 256 or more;
//   * keys compare as (d2, source row, column): ties go to the smallest raster index of the source, as brute force does.
// ------------------------------------------------------------------------------------------------
struct L2Cand {
    u32 hi, lo, idx;  // squared distance, source row << 16 | column, position in the compacted column list
};
__device__ __forceinline__ void l2env_offer(L2Cand &best, const uint2 *__restrict__ s_c, int cc, int j) {
    const uint2 cv = s_c[cc];
    const int dk = j - (int)(cv.y & 0xFFFFu);
    const u32 hi = cv.x + (u32)(dk * dk);
    if (hi < best.hi || (hi == best.hi && cv.y < best.lo)) {
        best.hi = hi;
        best.lo = cv.y;
        best.idx = (u32)cc;
    }
}
__device__ __forceinline__ void l2env_merge(L2Cand &best, int off) {
    const u32 oh = (u32)__shfl_xor((int)best.hi, off), ol = (u32)__shfl_xor((int)best.lo, off), oi = (u32)__shfl_xor((int)best.idx, off);
    if (oh < best.hi || (oh == best.hi && ol < best.lo)) {
        best.hi = oh;
        best.lo = ol;
        best.idx = oi;
    }
}

constexpr u32 L2_NOSRC = 0x3FFFFFFFu;  // g^2 of a column without a source
__global__ __launch_bounds__(256) void k_l2env(const float *__restrict__ x, const uint2 *__restrict__ ct, int CTP, int nb,
                                               const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s,
                                               const u32 *__restrict__ rowbase_s, const int *__restrict__ finfo,
                                               const float *__restrict__ vlist, int H, int W, int Wd,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status,
                                               const int *__restrict__ route) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_env[];
    __shared__ u32 s_wcnt[4];
    __shared__ u32 s_red[4][3];
    const int b = blockIdx.y, i = blockIdx.x;
    if (route[b] != 0) return;  // k_l2win + k_l2far took the frame
    uint2 *s_c = reinterpret_cast<uint2 *>(s_env);               // [W] the columns with a source, in column order
    u16 *s_own = reinterpret_cast<u16 *>(s_env + (size_t)W * 8);  // [W] owner of every solved pixel (index into s_c)
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t fo = (size_t)b * H * W;
    // columns -> {g^2, source row << 16 | column}, compacted
    int c = 0;
    {
        const int band = i >> 5, r = i & 31;
        const uint2 *crow = ct + ((size_t)b * nb + band) * CTP;
        const u32 upmask = (2u << r) - 1u;
        for (int k0 = 0; k0 < W; k0 += 256) {
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
            if (lane == 0) s_wcnt[wv] = (u32)__popcll(bal);
            __syncthreads();
            int off = c;
            for (int w = 0; w < wv; ++w) off += (int)s_wcnt[w];
            if (has) s_c[off + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = cv;
            c += (int)(s_wcnt[0] + s_wcnt[1] + s_wcnt[2] + s_wcnt[3]);
            __syncthreads();
        }
    }
    if (c > 0) {
        // a query for the whole block: candidates strided over the 256 threads, wave shuffles, then the four wave minima
        auto block_query = [&](int j, int lo, int hi) {
            L2Cand best = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
            for (int cc = lo + tid; cc <= hi; cc += 256) l2env_offer(best, s_c, cc, j);
#pragma unroll
            for (int o = 32; o; o >>= 1) l2env_merge(best, o);
            if (lane == 0) {
                s_red[wv][0] = best.hi;
                s_red[wv][1] = best.lo;
                s_red[wv][2] = best.idx;
            }
            __syncthreads();
            if (tid == 0) {
                int w0 = 0;
                for (int w = 1; w < 4; ++w)
                    if (s_red[w][0] < s_red[w0][0] || (s_red[w][0] == s_red[w0][0] && s_red[w][1] < s_red[w0][1])) w0 = w;
                s_own[j] = (u16)s_red[w0][2];
            }
            __syncthreads();
        };
        block_query(0, 0, c - 1);
        if (W > 1) block_query(W - 1, 0, c - 1);
        int s = 1;
        while (s < W - 1) s <<= 1;  // s >= W - 1: the first stride below it has its odd multiples inside (0, W-1)
        for (s >>= 1; s >= 1; s >>= 1) {
            const int nq = W - 2 >= s ? ((W - 2) / s + 1) / 2 : 0;  // odd multiples of s in [s, W-2]
            if (nq <= 2) {
                for (int m = 0; m < nq; ++m) {
                    const int j = (2 * m + 1) * s;
                    block_query(j, (int)s_own[j - s], (int)s_own[min(j + s, W - 1)]);
                }
                continue;
            }
            int G = 64;  // lanes per query: 256 / (nq rounded up to a power of two), at most a wave, at least one
            while (G > 1 && G * nq > 256) G >>= 1;
            const int per = 256 / G, gl = tid & (G - 1);
            for (int m0 = 0; m0 < nq; m0 += per) {
                const int m = m0 + tid / G;
                const bool act = m < nq;
                const int j = (2 * m + 1) * s;
                L2Cand best = {0xFFFFFFFFu, 0xFFFFFFFFu, 0u};
                if (act) {
                    const int lo = (int)s_own[j - s], hi = (int)s_own[min(j + s, W - 1)];
                    for (int cc = lo + gl; cc <= hi; cc += G) l2env_offer(best, s_c, cc, j);
                }
                for (int o = G >> 1; o; o >>= 1) l2env_merge(best, o);
                if (act && gl == 0) s_own[j] = (u16)best.idx;
            }
            __syncthreads();
        }
    }
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    for (int j = tid; j < W; j += 256) {
        const int p = i * W + j;
        int label = 0, q = p;
        float dist = INFINITY;
        if (c > 0) {
            const uint2 cv = s_c[s_own[j]];
            const int srow = (int)(cv.y >> 16), scol = (int)(cv.y & 0xFFFFu);
            q = srow * W + scol;
            const size_t w = ((size_t)b * H + srow) * Wd + (scol >> 6);
            label = source_rank(rowbase_s[(size_t)b * H + srow] + wpre_s[w], srcbits[w], scol);
            dist = sqrtf((float)(cv.x + (u32)((j - scol) * (j - scol))));
        }
        if (out_index) out_index[fo + p] = label;
        if (out_dt) out_dt[fo + p] = dist;
        if (out_depth) out_depth[fo + p] = gather_depth(x + fo, vlist + fo, label, q, nval, misaligned, frame_status + b);
    }
}

