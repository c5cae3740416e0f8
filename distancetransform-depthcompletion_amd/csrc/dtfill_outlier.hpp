// dtfill_outlier.hpp -- outlier_removal() of data_read.py:103-128: stand-alone (k_outlier) and in front of the predicates (k_mask_o)
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------------
// outlier_removal() zeroes a pixel when (v - mean) > 1.0 with mean = sum / (count + 1e-5) over the 25 taps of the 7x7
// diamond (cv2.filter2D: reflect-101 border, float32 accumulation in kernel row-major order -- what OpenCV's direct
// filter does; mean, difference and test in float64 -- numpy's promotion).
// Only pixels with v > 1.0 can be zeroed as long as no tap is negative (then mean >= 0, or NaN and the test is false), so
// the 25 taps are only gathered for those (a few percent of a LiDAR frame).  A negative value anywhere switches to the
// exhaustive evaluation: per tile in k_outlier, per frame (a second launch) in k_mask_o.
// ------------------------------------------------------------------------------------------------
constexpr int O_TH = 16, O_TW = 128;

__device__ __forceinline__ int reflect101(int p, int n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}

// k_outlier: one block per 16 x 128 tile staged in LDS with a 3-cell halo; candidates (v > 1.0, or every pixel when the
// staged tile holds a negative value) are compacted so that every lane evaluates one.
__global__ __launch_bounds__(256) void k_outlier(const float *__restrict__ x, int H, int W,
                                                 float *__restrict__ out) {
    __shared__ float s_t[(O_TH + 6) * (O_TW + 6)];
    __shared__ u16 s_list[O_TH * O_TW];
    __shared__ int s_n, s_neg;
    const int b = blockIdx.z, r0 = blockIdx.y * O_TH, c0 = blockIdx.x * O_TW;
    const float *xf = x + (size_t)b * H * W;
    if (threadIdx.x == 0) s_n = s_neg = 0;
    __syncthreads();
    bool neg = false;
    {
        // every load of the thread first (independent: one round trip instead of seven), then the tile
        constexpr int NIT = (O_TH + 6) * (O_TW + 6), PER = (NIT + 255) / 256;
        float lv[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int k = min((int)threadIdx.x + 256 * q, NIT - 1);
            const int r = k / (O_TW + 6), c = k - r * (O_TW + 6);
            const int gi = reflect101(min(r0 + r - 3, H + 2), H), gj = reflect101(min(c0 + c - 3, W + 2), W);
            lv[q] = xf[(size_t)gi * W + gj];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int k = (int)threadIdx.x + 256 * q;
            if (k < NIT) {
                neg |= lv[q] < 0.0f;
                s_t[k] = lv[q];
            }
        }
    }
    if (__any(neg) && (threadIdx.x & 63) == 0) s_neg = 1;
    __syncthreads();
    const bool all = s_neg != 0;
    for (int k = threadIdx.x; k < O_TH * O_TW; k += 256) {  // the other pixels go out unchanged, candidates are listed
        const int r = k / O_TW, c = k - r * O_TW;
        const int gi = r0 + r, gj = c0 + c;
        const bool in = gi < H && gj < W;
        const float v = s_t[(r + 3) * (O_TW + 6) + c + 3];
        const bool cand = in && (all || v > 1.0f);
        if (in && !cand) out[(size_t)b * H * W + (size_t)gi * W + gj] = v;  // a candidate is stored by the lane that evaluates it
        const u64 bal = __ballot(cand);
        int base = 0;
        if ((threadIdx.x & 63) == 0 && bal) base = atomicAdd(&s_n, __popcll(bal));
        base = __builtin_amdgcn_readfirstlane(base);
        if (cand) s_list[base + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)k;
    }
    __syncthreads();
    const int n = s_n;
    for (int t = threadIdx.x; t < n; t += 256) {
        const int k = s_list[t], r = k / O_TW, c = k - r * O_TW;
        float acc = 0.0f;
        int cnt = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = 0; j < 7; ++j) {
                if ((i < 3 ? 3 - i : i - 3) + (j < 3 ? 3 - j : j - 3) > 3) continue;
                const float v = s_t[(r + i) * (O_TW + 6) + c + j];
                acc = __fadd_rn(acc, v);  // no contraction, no reassociation
                cnt += v > 0.1f;
            }
        }
        const float v = s_t[(r + 3) * (O_TW + 6) + c + 3];
        const double mean = (double)acc / ((double)cnt + 0.00001);
        out[(size_t)b * H * W + (size_t)(r0 + r) * W + c0 + c] = ((double)v - mean) > 1.0 ? v * 0.0f : v;  // data_read.py:128 multiplies by (1 - flag): a removed negative pixel is -0.0
    }
}

// the 25 taps of pixel (i, j) straight from the frame (neighbouring rows: cache hits); declared in dtfill_prepass.hpp for k_mask4
__device__ __forceinline__ bool outlier_at(const float *__restrict__ xf, int H, int W, int i, int j, float v) {
    float acc = 0.0f;
    int cnt = 0;
#pragma unroll
    for (int di = -3; di <= 3; ++di) {
        const float *row = xf + (size_t)reflect101(i + di, H) * W;
        const int span = 3 - (di < 0 ? -di : di);
#pragma unroll
        for (int dj = -3; dj <= 3; ++dj) {
            if (dj < -span || dj > span) continue;
            const float t = row[reflect101(j + dj, W)];
            acc = __fadd_rn(acc, t);
            cnt += t > 0.1f;
        }
    }
    const double mean = (double)acc / ((double)cnt + 0.00001);
    return ((double)v - mean) > 1.0;
}

// ------------------------------------------------------------------------------------------------
// k_mask_o: k_mask with outlier_removal() in front of the predicates (data_read.py:168-169: the loader filters the sparse
// map before anything else sees it) -- the filtered map is never written: a zeroed pixel only clears its predicate bits, and
// the fill gathers the surviving pixels' own values from x.  One wave per row: the row's candidates are listed in LDS, one
// lane each evaluates them, the dropped pixels' bits go into an LDS bit row, then the row is read again (cache) for the
// words as k_mask builds them.  ALL = false looks at v > 1.0 only and raises negflag[b] when it meets a negative value;
// the ALL = true launch then redoes exactly those frames, every pixel a candidate.
// ------------------------------------------------------------------------------------------------
constexpr int MO_SEG = 2048;  // candidates are listed per segment of this many pixels

template <bool ALL>
__global__ __launch_bounds__(256) void k_mask_o(const float *__restrict__ x, int H, int W, int Wd, float src_thr, float val_thr,
                                                u64 *__restrict__ srcbits, u64 *__restrict__ valbits, u16 *__restrict__ wpre_s,
                                                u16 *__restrict__ wpre_v, u32 *__restrict__ rowcnt_s, u32 *__restrict__ rowcnt_v,
                                                int *__restrict__ negflag) {
    __shared__ u16 s_list[4][MO_SEG];
    __shared__ u32 s_drop[4][MAX_HW_SUM / 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * 4 + wave, b = blockIdx.y;
    if (i >= H) return;
    if (ALL && !negflag[b]) return;
    const float *xf = x + (size_t)b * H * W;
    const float *row = xf + (size_t)i * W;
    for (int k = lane; k < (W + 31) / 32; k += 64) s_drop[wave][k] = 0u;
    bool neg = false;
    for (int seg0 = 0; seg0 < W; seg0 += MO_SEG) {
        int n = 0;  // wave-uniform
        const int seg1 = min(W, seg0 + MO_SEG);
        for (int j0 = seg0; j0 < seg1; j0 += 64) {
            const int j = j0 + lane;
            const float v = j < seg1 ? row[j] : 0.0f;
            neg |= v < 0.0f;
            const bool cand = j < seg1 && (ALL || v > 1.0f);
            const u64 bal = __ballot(cand);
            if (cand) s_list[wave][n + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)j;
            n += __popcll(bal);
        }
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < n; t += 64) {
            const int j = s_list[wave][t];
            if (outlier_at(xf, H, W, i, j, row[j])) atomicOr(&s_drop[wave][j >> 5], 1u << (j & 31));
        }
        __builtin_amdgcn_wave_barrier();
    }
    if (!ALL && __any(neg) && lane == 0) negflag[b] = 1;  // this frame is redone by the exhaustive launch
    // the words, as k_mask builds them, with the dropped pixels read as 0.0f
    u32 run_s = 0, run_v = 0, mis = 0;
    for (int k0 = 0; k0 < Wd; k0 += 64) {
        const int nk = min(64, Wd - k0);
        u64 ws = 0, wv = 0;
        for (int kb = 0; kb < nk; kb += M_KU) {
            float v[M_KU];
#pragma unroll
            for (int u = 0; u < M_KU; ++u) {
                const int j = (k0 + kb + u) * 64 + lane;
                v[u] = (kb + u < nk && j < W) ? row[j] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < M_KU; ++u) {
                const int k = kb + u;
                const int j = (k0 + k) * 64 + lane;
                const bool in = k < nk && j < W;
                const float f = (in && ((s_drop[wave][j >> 5] >> (j & 31)) & 1u)) ? 0.0f : v[u];
                const u64 sb = __ballot(in && !((1.0f - f) > src_thr));
                const u64 vb = __ballot(in && (f > val_thr));
                ws = lane == k ? sb : ws;
                wv = lane == k ? vb : wv;
            }
        }
        const u32 cs = __popcll(ws), cv = __popcll(wv);
        const u32 is = wave_incl_sum(cs, lane), iv = wave_incl_sum(cv, lane);
        mis |= __any(ws != wv) ? 1u : 0u;
        if (lane < nk) {
            const size_t wi = ((size_t)b * H + i) * Wd + k0 + lane;
            srcbits[wi] = ws;
            valbits[wi] = wv;
            wpre_s[wi] = (u16)(run_s + is - cs);
            wpre_v[wi] = (u16)(run_v + iv - cv);
        }
        run_s += __shfl(is, 63);
        run_v += __shfl(iv, 63);
    }
    if (lane == 0) {
        rowcnt_s[(size_t)b * H + i] = run_s;
        rowcnt_v[(size_t)b * H + i] = run_v | (mis ? 0x80000000u : 0u);
    }
}
