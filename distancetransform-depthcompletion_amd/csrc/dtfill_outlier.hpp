// dtfill_outlier.hpp -- k_outlier: outlier_removal() of data_read.py:103-128
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------------
// k_outlier: outlier_removal() of data_read.py:103-128.  One block per 16 x 64 tile staged in LDS with a
// 3-cell halo (reflect-101 at the image border, as cv2.filter2D's default); the 25 taps of the 7x7
// diamond are accumulated in float32 in kernel row-major order (what OpenCV's direct filter does), the
// valid count as an integer; mean, difference and the > 1.0 test in float64 (numpy's promotion).
// ------------------------------------------------------------------------------------------------
constexpr int O_TH = 16, O_TW = 64;

__device__ __forceinline__ int reflect101(int p, int n) {
    p = p < 0 ? -p : p;
    return p >= n ? 2 * n - 2 - p : p;
}

__global__ __launch_bounds__(256) void k_outlier(const float *__restrict__ x, int H, int W,
                                                 float *__restrict__ out) {
    __shared__ float s_t[(O_TH + 6) * (O_TW + 6)];
    const int b = blockIdx.z, r0 = blockIdx.y * O_TH, c0 = blockIdx.x * O_TW;
    const float *xf = x + (size_t)b * H * W;
    for (int k = threadIdx.x; k < (O_TH + 6) * (O_TW + 6); k += 256) {
        const int r = k / (O_TW + 6), c = k - r * (O_TW + 6);
        const int gi = reflect101(min(r0 + r - 3, H + 2), H), gj = reflect101(min(c0 + c - 3, W + 2), W);
        s_t[k] = xf[(size_t)gi * W + gj];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < O_TH * O_TW; k += 256) {
        const int r = k / O_TW, c = k - r * O_TW;
        const int gi = r0 + r, gj = c0 + c;
        if (gi >= H || gj >= W) continue;
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
        out[(size_t)b * H * W + (size_t)gi * W + gj] = ((double)v - mean) > 1.0 ? 0.0f : v;
    }
}
