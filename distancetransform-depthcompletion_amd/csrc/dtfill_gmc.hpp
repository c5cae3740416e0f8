// dtfill_gmc.hpp -- k_gmc: one step of generate_multi_channel(), net.py:83-122
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// One step of the reference's windowed nearest fill: over the ts x ts window (zero padding, as
// tf.image.extract_patches(padding='SAME')), s = mask * w with w = ts - |di| - |dj| (net.py:71-81);
// out = (sum of the inputs at the positions where s equals the window maximum) / (1e-6 + their count)
// (net.py:91-93).  mask == nullptr: the mask is (data > 0.001f), what the reference feeds to the next step
// (net.py:95-96).  float32; the sum runs over the taps in row-major order, a new maximum restarts it, so
// the additions are exactly those of "sum over the selected taps in order".
// One block per 16 x 64 tile; data and mask tiles with a (ts-1)/2 halo in LDS.
constexpr int GM_TH = 16, GM_TW = 64, GM_MAXHALF = 7;

// table_size 7 (every model of the reference), compile-time unrolled, ONE pass over the 49 taps: a new window
// maximum restarts sum and count, an equal value extends them -- the additions performed are exactly "sum
// over the selected taps in tap order", as in the two-pass form below.  DERIVED: the mask is data > 0.001
// (steps 2 and 3), so only the data tile is staged.
template <bool DERIVED>
__global__ __launch_bounds__(256) void k_gmc7(const float *__restrict__ data, const float *__restrict__ mask, int H,
                                              int W, float *__restrict__ out) {
    constexpr int half = 3, PW = GM_TW + 6, PH = GM_TH + 6;
    __shared__ float s_d[PH * PW];
    __shared__ float s_m[DERIVED ? 1 : PH * PW];
    const int b = blockIdx.z, r0 = blockIdx.y * GM_TH, c0 = blockIdx.x * GM_TW;
    const size_t fo = (size_t)b * H * W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ u32 s_mb[PH][4];  // bit c of a row: the mask of padded column c is not zero
    bool odd = false;            // a mask value that is neither 0 nor 1 (only a caller's mask can hold one)
    for (int r = wave; r < PH; r += 4) {
        const int gi = r0 + r - half;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int c = lane + 64 * k;
            const int gj = c0 + c - half;
            const bool in = c < PW && gi >= 0 && gi < H && gj >= 0 && gj < W;
            const float v = in ? data[fo + (size_t)gi * W + gj] : 0.0f;
            const float m = DERIVED ? (v > 0.001f ? 1.0f : 0.0f) : (in ? mask[fo + (size_t)gi * W + gj] : 0.0f);
            if (c < PW) {
                s_d[r * PW + c] = v;
                if (!DERIVED) s_m[r * PW + c] = m;
            }
            odd |= !(m == 0.0f || m == 1.0f);
            const u64 bal = __ballot(m != 0.0f);
            if (lane == 0) {
                if (k == 0) {
                    s_mb[r][0] = (u32)bal;
                    s_mb[r][1] = (u32)(bal >> 32);
                } else {
                    s_mb[r][2] = (u32)bal;
                    s_mb[r][3] = 0u;
                }
            }
        }
    }
    const bool binary = !__syncthreads_or(odd);  // block-uniform; the tiles are staged
    // With a 0 / 1 mask the selected taps are the masked taps at the smallest L1 distance from the pixel (w = 7 - |di| - |dj|):
    // that distance from the rows' mask bits (nearest set bit left / right of the centre per row), then at most two taps per
    // row, added to +0 in row-major order as the reference's reduce_sum over the 49 products does (a lone -0.0 comes out as
    // +0.0).  No masked tap in the window: every tap ties at s = 0, all 49 are added.
    auto ring = [&](int r, int c) {
        u32 mbits[7];
        u32 tmin = 64u;
        const int w = c >> 5;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const u32 mb = __builtin_amdgcn_alignbit(s_mb[r + i][w + 1], s_mb[r + i][w], (u32)c) & 0x7Fu;  // taps j = 0..6: bits 0..6
            mbits[i] = mb;
            const u32 hx = min(min(ffbh_u32((mb & 0xFu) << 28), ffbl_b32(mb >> 3)), 64u);
            tmin = min(tmin, hx + (u32)(i < 3 ? 3 - i : i - 3));
        }
        float acc = 0.0f, cnt = 0.0f;
        if (tmin >= 64u) {
#pragma unroll
            for (int i = 0; i < 7; ++i)
#pragma unroll
                for (int j = 0; j < 7; ++j) acc = __fadd_rn(acc, s_d[(r + i) * PW + c + j]);
            cnt = 49.0f;
        } else {
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
        out[fo + (size_t)(r0 + r) * W + c0 + c] = __fdiv_rn(acc, __fadd_rn(0.000001f, cnt));
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
        out[fo + (size_t)(r0 + r) * W + c0 + c] = __fdiv_rn(acc, __fadd_rn(0.000001f, cnt));
    };
    if (!DERIVED) {
        for (int r = wave; r < GM_TH; r += 4)
            if (r0 + r < H && c0 + lane < W) {
                if (binary)
                    ring(r, lane);
                else
                    full(r, lane);
            }
        return;
    }
    __shared__ u16 s_list[GM_TH * GM_TW];
    __shared__ int s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    for (int r = wave; r < GM_TH; r += 4) {
        const bool in = r0 + r < H && c0 + lane < W;
        const float vc = s_d[(r + 3) * PW + lane + 3];
        const bool easy = in && vc > 0.001f;
        if (easy) out[fo + (size_t)(r0 + r) * W + c0 + lane] = __fdiv_rn(vc, __fadd_rn(0.000001f, 1.0f));
        const u64 bal = __ballot(in && !easy);
        int base = 0;
        if (lane == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (in && !easy) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)(r * GM_TW + lane);
    }
    __syncthreads();
    const int n = s_n;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int k = s_list[t];
        ring(k / GM_TW, k % GM_TW);
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
