// dtfill_post.hpp -- what the reference's drivers do with a filled frame: crop / depth floor, PNG quantisation,
// error metrics.  Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace.
#pragma once

// depth floor of the drivers: np.squeeze(tf.nn.relu(d - 0.9) + 0.9) (eval_NYU.py:205, test.py:133).  Float32
// step by step -- (d - 0.9f) + 0.9f is NOT d in float32, so the two roundings are kept.
__device__ __forceinline__ float depth_floor(float d, float floor_) {
    return __fadd_rn(fmaxf(__fsub_rn(d, floor_), 0.0f), floor_);
}

// ------------------------------------------------------------------------------------------------
// k_crop_floor: out[b, i, j] = f(x[b, r0 + i, c0 + j]); f = depth floor if use_floor, identity otherwise.
// demo.py:292-293 (rows 96: of a KITTI frame), eval_NYU.py:202-205 ([6:234, 8:312] of an NYU frame; the
// depth floor on KITTI).  One lane per output pixel, rows of the crop are contiguous runs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_crop_floor(const float *__restrict__ x, int H, int W, int r0, int c0, int OH,
                                                    int OW, int use_floor, float floor_, float *__restrict__ out) {
    const int b = blockIdx.z, i = blockIdx.y;
    const float *src = x + ((size_t)b * H + r0 + i) * W + c0;
    float *dst = out + ((size_t)b * OH + i) * OW;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < OW; j += gridDim.x * 256) {
        const float v = src[j];
        dst[j] = use_floor ? depth_floor(v, floor_) : v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_png16: test.py:133-148.  depth floor, clip to [lo, hi], pad_top copies of the first row on top, * scale,
// C cast to uint16 (numpy astype).  out is [B, pad_top + H, W].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_png16(const float *__restrict__ x, int H, int W, int pad_top, int use_floor,
                                               float floor_, float lo, float hi, float scale,
                                               unsigned short *__restrict__ out) {
    const int b = blockIdx.z, oi = blockIdx.y;
    const int i = max(oi - pad_top, 0);  // rows above the frame repeat its first row (np.tile + np.vstack)
    const float *src = x + ((size_t)b * H + i) * W;
    unsigned short *dst = out + ((size_t)b * (H + pad_top) + oi) * W;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < W; j += gridDim.x * 256) {
        float v = src[j];
        if (use_floor) v = depth_floor(v, floor_);
        v = fminf(fmaxf(v, lo), hi);  // tf.clip_by_value
        dst[j] = (unsigned short)(int)__fmul_rn(v, scale);
    }
}

// ------------------------------------------------------------------------------------------------
// Metrics: evaluation.py:82-123 (Result.evaluate, KITTI: metres -> mm / km) and :196-239
// (Result_NYU.evaluate: no unit change, mae is RELATIVE, + the three delta accuracies).  Per element the
// float32 operations of the numpy expressions, one rounding each (no contraction); the means accumulate in
// float64 in a fixed order (numpy: float32 pairwise), so results are reproducible run to run.
// Stage 1: M_NB blocks per frame write partial sums; stage 2: one wave per frame adds them and finishes.
// ------------------------------------------------------------------------------------------------
constexpr int M_NB = 64;   // partial-sum blocks per frame
constexpr int M_NS = 8;    // sums: d^2, d, inv^2, inv, delta1, delta2, delta3, count
constexpr int M_NOUT = 9;  // mse, rmse, mae, irmse, imae, delta1, delta2, delta3, count

template <int KIND>  // 0 = KITTI (Result), 1 = NYU (Result_NYU)
__global__ __launch_bounds__(256) void k_metrics_part(const float *__restrict__ output, const float *__restrict__ target,
                                                      long long n, double *__restrict__ part) {
    const int b = blockIdx.y;
    const float *o_ = output + (size_t)b * n, *t_ = target + (size_t)b * n;
    double s[M_NS];
#pragma unroll
    for (int k = 0; k < M_NS; ++k) s[k] = 0.0;
    constexpr int U = 4;  // elements per lane in flight
    for (long long base = (long long)blockIdx.x * 256 * U; base < n; base += (long long)gridDim.x * 256 * U) {
        float o[U], t[U];
        bool in[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = base + u * 256 + threadIdx.x;
            in[u] = p < n;
            o[u] = in[u] ? o_[p] : 0.0f;
            t[u] = in[u] ? t_[p] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool valid = in[u] && o[u] > 0.01f && t[u] > 0.01f;  // evaluation.py:85-87 / :199-201
            // an invalid element computes on (1, 1): all its terms are 0 and it is not counted
            const float ov = valid ? o[u] : 1.0f, tv = valid ? t[u] : 1.0f;
            float diff, io, it, rel = 0.0f, ratio = 1.0f;
            if (KIND == 0) {
                diff = fabsf(__fsub_rn(__fmul_rn(1000.0f, ov), __fmul_rn(1000.0f, tv)));  // :89-92
                io = __frcp_rn(__fmul_rn(0.001f, ov));                                    // :115-116
                it = __frcp_rn(__fmul_rn(0.001f, tv));
            } else {
                diff = fabsf(__fsub_rn(ov, tv));  // :203-206
                rel = __fdiv_rn(diff, tv);        // :210 mae = mean(abs_diff / target)
                ratio = fmaxf(__fdiv_rn(ov, tv), __fdiv_rn(tv, ov));  // :217
                io = __frcp_rn(ov);               // :232-233
                it = __frcp_rn(tv);
            }
            const float idiff = fabsf(__fsub_rn(io, it));
            s[0] += (double)__fmul_rn(diff, diff);
            s[1] += (double)(KIND == 0 ? diff : rel);
            s[2] += (double)__fmul_rn(idiff, idiff);
            s[3] += (double)idiff;
            if (KIND == 1) {
                s[4] += (valid && ratio < 1.25f) ? 1.0 : 0.0;  // :218-220
                s[5] += (valid && ratio < 1.5625f) ? 1.0 : 0.0;
                s[6] += (valid && ratio < 1.953125f) ? 1.0 : 0.0;
            }
            s[7] += valid ? 1.0 : 0.0;
        }
    }
    // block sum in a fixed order: lanes (shuffle tree), then the four waves
    __shared__ double s_w[4][M_NS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < M_NS; ++k) {
        double v = s[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) s_w[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < M_NS)
        part[((size_t)b * gridDim.x + blockIdx.x) * M_NS + threadIdx.x] =
            ((s_w[0][threadIdx.x] + s_w[1][threadIdx.x]) + s_w[2][threadIdx.x]) + s_w[3][threadIdx.x];
}

template <int KIND>
__global__ __launch_bounds__(64) void k_metrics_final(const double *__restrict__ part, int nb, double *__restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double s[M_NS];
#pragma unroll
    for (int k = 0; k < M_NS; ++k) {
        double v = 0.0;
        for (int q = lane; q < nb; q += 64) v += part[((size_t)b * nb + q) * M_NS + k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_down(v, off);
        s[k] = v;
    }
    if (lane == 0) {
        const double cnt = s[7];  // 0 valid elements: 0 / 0 = NaN, numpy's mean of an empty array
        double *o = out + (size_t)b * M_NOUT;
        o[0] = s[0] / cnt;
        o[1] = sqrt(s[0] / cnt);
        o[2] = s[1] / cnt;
        o[3] = sqrt(s[2] / cnt);
        o[4] = s[3] / cnt;
        o[5] = KIND == 1 ? s[4] / cnt : 0.0;  // Result (KITTI) leaves the deltas at their initial 0
        o[6] = KIND == 1 ? s[5] / cnt : 0.0;
        o[7] = KIND == 1 ? s[6] / cnt : 0.0;
        o[8] = cnt;
    }
}


// ------------------------------------------------------------------------------------------------
// k_stats (dtfill_pass_stats): which kernel family owned how many pixels of the last pass, from the routing state the pass
// left in the workspace.  One block per frame.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_stats(const int *__restrict__ route, const int *__restrict__ fflag, const u32 *__restrict__ rowfar,
                                               const int *__restrict__ finfo, int H, int W, int l2, long long *__restrict__ out) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int rt = route[b], r = rt > 0 ? (rt & 0xFF) : rt, ff = fflag[b];
    const u32 t = w2_row_t(W);
    int win = 0, any = 0, sky = 0;
    for (int i = tid; i < H; i += 256) {
        const u32 f = rowfar[(size_t)b * H + i];
        if (l2) {
            if (r > 0) {
                if (f >= t) ++any; else ++win;
            } else if (r == 0)
                ++any;
        } else if (r > 0) {
            if (f == 2u && finfo[b * FI_STRIDE + FI_SKY] > 0) ++sky; else if (f == 0u) ++win; else ++any;
        } else if (r == 0) {
            if (f == 2u && finfo[b * FI_STRIDE + FI_SKY] > 0) ++sky; else ++any;
        }
    }
    __shared__ int s[3];
    if (tid < 3) s[tid] = 0;
    __syncthreads();
    atomicAdd(&s[0], win);
    atomicAdd(&s[1], any);
    atomicAdd(&s[2], sky);
    __syncthreads();
    if (tid == 0) {
        auto add = [&](int k, long long v) { if (v) atomicAdd(reinterpret_cast<unsigned long long *>(out + k), (unsigned long long)v); };
        add(DTFILL_STATS_ALL, (long long)H * W);
        add(DTFILL_STATS_WINDOW, (long long)s[0] * W);
        add(DTFILL_STATS_ANYDIST, (long long)s[1] * W);
        add(DTFILL_STATS_SKY, (long long)s[2] * W);
        add(DTFILL_STATS_POINTS, r == ROUTE_POINTS ? (long long)H * W : 0);
        add(DTFILL_STATS_COLT, (l2 ? (r == 0 || ff == 1) : (ff == 1 || ff == 2)) ? (long long)H * W : 0);
    }
}
