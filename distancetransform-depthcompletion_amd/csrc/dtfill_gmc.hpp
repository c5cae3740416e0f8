// dtfill_gmc.hpp -- k_gmc: one step of generate_multi_channel(), net.py:83-122
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// One step of the reference's windowed nearest fill: over the ts x ts window (zero padding, as
// tf.image.extract_patches(padding='SAME')), s = mask * w with w = ts - |di| - |dj| (net.py:71-81);
// out = (sum of the inputs at the positions where s equals the window maximum) / (1e-6 + their count)
// (net.py:91-93).  mask == nullptr: the mask is (data > 0.001f), what the reference feeds to the next step
// (net.py:95-96).  float32; the sum runs over the taps in row-major order, a new maximum restarts it, so
// the additions are exactly those of "sum over the selected taps in order".
// One block per 16 x 64 tile; data and mask tiles with a (ts-1)/2 halo in LDS.  (GM_TW = 128 fetches a third less -- a staged
// row segment of 134 floats straddles five 128-byte lines for its four, one of 70 four for two -- and takes the same time on the
// KITTI crop: the first step is bound by its instructions.  64 divides 1216.)
constexpr int GM_TH = 16, GM_TW = 64, GM_MAXHALF = 7;
constexpr int GM_NCW = GM_TW / 64;  // 64-column groups of a tile row: a wave works on one (row, group) at a time, a lane per column
static_assert(GM_TW % 64 == 0, "whole groups");

// table_size 7 (every model of the reference), compile-time unrolled, ONE pass over the 49 taps: a new window
// maximum restarts sum and count, an equal value extends them -- the additions performed are exactly "sum
// over the selected taps in tap order", as in the two-pass form below.  DERIVED: the mask is data > 0.001
// (steps 2 and 3), so only the data tile is staged.
//
// A pixel whose own value passes the next step's mask (v > 0.001) is that step's nearest masked tap itself: its next value is
// v / (1 + 1e-6), whatever its neighbours hold.  So every value written here is carried on through the later steps for as long
// as it stays above 0.001 (out_n1, out_n2: the outputs of the next two steps, or null), and a DERIVED launch only works on the
// pixels that did not pass (a tile without one ends after reading its own pixels).
template <bool DERIVED>
__global__ __launch_bounds__(256) void k_gmc7(const float *__restrict__ data, const float *__restrict__ mask, int H,
                                              int W, float *__restrict__ out, float *__restrict__ out_n1,
                                              float *__restrict__ out_n2) {
    constexpr int half = 3, PW = GM_TW + 6, PH = GM_TH + 6;
    if (DERIVED) {
        // the tile's own pixels first: nothing to do unless one of them failed the mask
        bool hard = false;
        for (int k = threadIdx.x; k < GM_TH * GM_TW; k += 256) {
            const int gi = blockIdx.y * GM_TH + k / GM_TW, gj = blockIdx.x * GM_TW + k % GM_TW;
            if (gi < H && gj < W) hard |= !(data[(size_t)blockIdx.z * H * W + (size_t)gi * W + gj] > 0.001f);
        }
        if (!__syncthreads_or(hard)) return;
    }
    __shared__ float s_d[PH * PW];
    __shared__ float s_m[DERIVED ? 1 : PH * PW];
    const int b = blockIdx.z, r0 = blockIdx.y * GM_TH, c0 = blockIdx.x * GM_TW;
    const size_t fo = (size_t)b * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ u32 s_mb[PH][2 * GM_NCW + 2];  // bit c of a row: the mask of padded column c is not zero
    bool odd = false;            // a mask value that is neither 0 nor 1 (only a caller's mask can hold one)
    {
        // every load of the thread first (independent: one round trip), then the tile and the mask bits
        constexpr int NR = (PH + 3) / 4;
        float lv[NR][GM_NCW + 1], lm[NR][GM_NCW + 1];
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            // (clamped addresses, unconditional loads: what lies outside the image is dropped below)
            const int gi = min(max(r0 + wave + 4 * q - half, 0), H - 1);
#pragma unroll
            for (int k = 0; k <= GM_NCW; ++k) {
                const int gj = min(max(c0 + lane + 64 * k - half, 0), W - 1);
                const u32 at = (u32)gi * (u32)W + (u32)gj;
                lv[q][k] = data[fo + at];
                lm[q][k] = DERIVED ? 0.0f : mask[fo + at];
            }
        }
#pragma unroll
        for (int q = 0; q < NR; ++q) {
            const int r = wave + 4 * q, gi = r0 + r - half;
            if (r >= PH) break;  // wave-uniform
#pragma unroll
            for (int k = 0; k <= GM_NCW; ++k) {
                const int c = lane + 64 * k, gj = c0 + c - half;
                const bool in = c < PW && gi >= 0 && gi < H && gj >= 0 && gj < W;
                const float v = in ? lv[q][k] : 0.0f;
                const float m = DERIVED ? (v > 0.001f ? 1.0f : 0.0f) : (in ? lm[q][k] : 0.0f);
                if (c < PW) {
                    s_d[r * PW + c] = v;
                    if (!DERIVED) s_m[r * PW + c] = m;
                }
                odd |= !(m == 0.0f || m == 1.0f);
                const u64 bal = __ballot(m != 0.0f);
                if (lane == 0) {  // (the last group holds the six columns beyond the tile's: the lanes past them ballot 0)
                    s_mb[r][2 * k] = (u32)bal;
                    s_mb[r][2 * k + 1] = (u32)(bal >> 32);
                }
            }
        }
    }
    const bool binary = !__syncthreads_or(odd);  // block-uniform; the tiles are staged
    // With a 0 / 1 mask: the L1 distance of every pixel's nearest masked tap, 64 pixels of a row at a time.  E_k of a padded row:
    // bit c = a masked tap within k columns of pixel c; "nearest tap within t" of an output row = the OR over its seven rows of
    // E_(t - |di|); seven cumulative words per row, kept as the three bit planes of n = 7 - distance (0: no masked tap).
    __shared__ u64 s_E[PH][GM_NCW][4], s_N[GM_TH][GM_NCW][3];
    if (binary) {
        if (threadIdx.x < PH * GM_NCW) {
            const int rr = threadIdx.x / GM_NCW, h = threadIdx.x % GM_NCW;  // padded row, column group
            const u64 lo = (u64)s_mb[rr][2 * h] | (u64)s_mb[rr][2 * h + 1] << 32, hi = (u64)s_mb[rr][2 * h + 2] | (u64)s_mb[rr][2 * h + 3] << 32;
            auto S = [&](int x) { return lo >> x | hi << (64 - x); };  // the mask bits of padded columns 64 h + c + x
            const u64 e0 = S(3), e1 = e0 | S(2) | S(4), e2 = e1 | S(1) | S(5), e3 = e2 | lo | S(6);
            s_E[rr][h][0] = e0; s_E[rr][h][1] = e1; s_E[rr][h][2] = e2; s_E[rr][h][3] = e3;
        }
        __syncthreads();
        if (threadIdx.x < GM_TH * GM_NCW) {
            const int r = threadIdx.x / GM_NCW, h = threadIdx.x % GM_NCW;  // output row r: padded rows r .. r + 6, centre r + 3
            auto E = [&](int k, int d) { return d ? s_E[r + 3 - d][h][k] | s_E[r + 3 + d][h][k] : s_E[r + 3][h][k]; };
            const u64 t0 = E(0, 0), t1 = E(1, 0) | E(0, 1), t2 = E(2, 0) | E(1, 1) | E(0, 2), t3 = E(3, 0) | E(2, 1) | E(1, 2) | E(0, 3),
                      t4 = E(3, 0) | E(3, 1) | E(2, 2) | E(1, 3), t5 = E(3, 0) | E(3, 1) | E(3, 2) | E(2, 3),
                      t6 = E(3, 0) | E(3, 1) | E(3, 2) | E(3, 3);
            s_N[r][h][2] = t3;
            s_N[r][h][1] = (t5 & ~t3) | t1;
            s_N[r][h][0] = (t6 & ~t5) | (t4 & ~t3) | (t2 & ~t1) | t0;
        }
        __syncthreads();
    }
    auto emit = [&](int r, int c, float acc, float cnt) {
        const size_t at = fo + (size_t)(r0 + r) * W + c0 + c;
        float v = __fdiv_rn(acc, __fadd_rn(0.000001f, cnt));
        out[at] = v;
        if (out_n1 && v > 0.001f) {
            v = __fdiv_rn(v, __fadd_rn(0.000001f, 1.0f));
            out_n1[at] = v;
            if (out_n2 && v > 0.001f) out_n2[at] = __fdiv_rn(v, __fadd_rn(0.000001f, 1.0f));
        }
    };
    // With a 0 / 1 mask the selected taps are the masked taps at the smallest L1 distance from the pixel (w = 7 - |di| - |dj|):
    // that distance from the rows' mask bits (nearest set bit left / right of the centre per row), then at most two taps per
    // row, added to +0 in row-major order as the reference's reduce_sum over the 49 products does (a lone -0.0 comes out as
    // +0.0).  No masked tap in the window: every tap ties at s = 0, all 49 are added.
    auto none49 = [&](int r, int c) {  // no masked tap in the window: every tap ties at s = 0
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) acc = __fadd_rn(acc, s_d[(r + i) * PW + c + j]);
        emit(r, c, acc, 49.0f);
    };
    // INLINE49: the "no masked tap" pixels are evaluated here; otherwise the caller has listed them (returns true for such a pixel)
    auto ring = [&](int r, int c, auto inline49) -> bool {
        constexpr bool INLINE49 = decltype(inline49)::value;
        u32 mbits[7];
        const int w = c >> 5;
#pragma unroll
        for (int i = 0; i < 7; ++i) mbits[i] = __builtin_amdgcn_alignbit(s_mb[r + i][w + 1], s_mb[r + i][w], (u32)c) & 0x7Fu;  // taps j = 0..6: bits 0..6
        const u64 *np = s_N[r][c >> 6];
        const int cb = c & 63;
        const u32 n = (u32)((np[0] >> cb) & 1ull) | (u32)((np[1] >> cb) & 1ull) << 1 | (u32)((np[2] >> cb) & 1ull) << 2;
        const u32 tmin = n ? 7u - n : 64u;
        float acc = 0.0f, cnt = 0.0f;
        if (tmin >= 64u) {
            if (INLINE49) none49(r, c);
            return true;
        }
        {
#pragma unroll
            for (int i = 0; i < 7; ++i) {
                const int dx = (int)tmin - (i < 3 ? 3 - i : i - 3);
                if (dx < 0 || dx > 3) continue;
                if ((mbits[i] >> (3 - dx)) & 1u) {
                    const float v = s_d[(r + i) * PW + c + 3 - dx];
                    acc = __fadd_rn(acc, v);
                    cnt += 1.0f;
                }
                if (dx > 0 && ((mbits[i] >> (3 + dx)) & 1u)) {
                    const float v = s_d[(r + i) * PW + c + 3 + dx];
                    acc = __fadd_rn(acc, v);
                    cnt += 1.0f;
                }
            }
        }
        emit(r, c, acc, cnt);
        return false;
    };
    auto full = [&](int r, int c) {
        float mx = 0.0f, acc = 0.0f, cnt = 0.0f;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                const float w = (float)(7 - (i < 3 ? 3 - i : i - 3) - (j < 3 ? 3 - j : j - 3));
                const float v = s_d[(r + i) * PW + c + j];
                const float m = DERIVED ? (v > 0.001f ? 1.0f : 0.0f) : s_m[(r + i) * PW + c + j];
                const float sv = m * w;
                const bool gt = sv > mx, eq = sv == mx;
                acc = gt ? __fadd_rn(0.0f, v) : (eq ? __fadd_rn(acc, v) : acc);  // a new maximum restarts the sum (from +0)
                cnt = gt ? 1.0f : (eq ? cnt + 1.0f : cnt);
                mx = gt ? sv : mx;
            }
        }
        emit(r, c, acc, cnt);
    };
    __shared__ u16 s_list[GM_TH * GM_TW];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    if (!DERIVED) {
        if (!binary) {
            for (int u = wave; u < GM_TH * GM_NCW; u += 4) {  // a wave per (row, column group)
                const int r = u / GM_NCW, c = 64 * (u % GM_NCW) + lane;
                if (r0 + r < H && c0 + c < W) full(r, c);
            }
            return;
        }
        // the pixels without a masked tap in their window (8 % at 5 % density: nearly every wave holds one) are listed and
        // summed afterwards, 64 of them per wave, instead of every wave walking through the 49 taps for its one or two
        for (int u = wave; u < GM_TH * GM_NCW; u += 4) {
            const int r = u / GM_NCW, c = 64 * (u % GM_NCW) + lane;
            const bool in = r0 + r < H && c0 + c < W;
            const bool none = in && ring(r, c, std::false_type{});
            const u64 bal = __ballot(none);
            int base = 0;
            if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
            base = __builtin_amdgcn_readfirstlane(base);
            if (none) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)(r * GM_TW + c);
        }
        __syncthreads();
        const int n49 = s_n;
        for (int t = threadIdx.x; t < n49; t += 256) {
            const int k = s_list[t];
            none49(k / GM_TW, k % GM_TW);
        }
        return;
    }
    for (int u = wave; u < GM_TH * GM_NCW; u += 4) {
        const int r = u / GM_NCW, c = 64 * (u % GM_NCW) + lane;
        const bool in = r0 + r < H && c0 + c < W;
        const float vc = s_d[(r + 3) * PW + c + 3];
        const bool easy = in && vc > 0.001f;  // (its value went out with the previous step's)
        const u64 bal = __ballot(in && !easy);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (in && !easy) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)(r * GM_TW + c);
    }
    __syncthreads();
    const int n = s_n;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int k = s_list[t];
        ring(k / GM_TW, k % GM_TW, std::true_type{});
    }
}

__global__ __launch_bounds__(256) void k_gmc(const float *__restrict__ data, const float *__restrict__ mask, int H,
                                             int W, int ts, float *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float s_gm[];
    const int half = (ts - 1) / 2;
    const int PW = GM_TW + 2 * half, PH = GM_TH + 2 * half;
    float *s_d = s_gm, *s_m = s_gm + PH * PW;
    const int b = blockIdx.z, r0 = blockIdx.y * GM_TH, c0 = blockIdx.x * GM_TW;
    const size_t fo = (size_t)b * H * W;
    for (int k = threadIdx.x; k < PH * PW; k += 256) {
        const int r = k / PW, c = k - r * PW;
        const int gi = r0 + r - half, gj = c0 + c - half;
        const bool in = gi >= 0 && gi < H && gj >= 0 && gj < W;
        const float v = in ? data[fo + (size_t)gi * W + gj] : 0.0f;
        s_d[k] = v;
        s_m[k] = in ? (mask ? mask[fo + (size_t)gi * W + gj] : (v > 0.001f ? 1.0f : 0.0f)) : 0.0f;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < GM_TH * GM_TW; k += 256) {
        const int r = k / GM_TW, c = k - r * GM_TW;
        const int gi = r0 + r, gj = c0 + c;
        if (gi >= H || gj >= W) continue;
        float mx = 0.0f, acc = 0.0f, cnt = 0.0f;
        // pass 1: the window maximum of mask * w (max over exact small products: order does not matter)
        for (int i = 0; i < ts; ++i)
            for (int j = 0; j < ts; ++j) {
                const float w = (float)(ts - abs(i - half) - abs(j - half));
                mx = fmaxf(mx, s_m[(r + i) * PW + c + j] * w);
            }
        // pass 2: sum and count of the positions that reach it, in tap order
        for (int i = 0; i < ts; ++i)
            for (int j = 0; j < ts; ++j) {
                const float w = (float)(ts - abs(i - half) - abs(j - half));
                const bool sel = s_m[(r + i) * PW + c + j] * w == mx;
                acc = sel ? __fadd_rn(acc, s_d[(r + i) * PW + c + j]) : acc;
                cnt += sel ? 1.0f : 0.0f;
            }
        out[fo + (size_t)gi * W + gj] = __fdiv_rn(acc, __fadd_rn(0.000001f, cnt));
    }
}
