// dtfill_sky.hpp -- k_sky: the rows above every source (the empty sky of a LiDAR frame), l1_cv
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------------
// Rows [0, r0) of a frame whose first source lies in row r0 (k_frame: FI_SKY, row flag 2).  Every source is below such a
// pixel q = (i, j), so (cv2's two sweeps, distanceTransformEx_5x5):
//   * the forward sweep leaves nothing there (no source above or left: q is not "live"), the backward sweep decides q from
//     its taps in the order (+2,+1,3) (+2,-1,3) (+1,+2,3) (+1,+1,2) (+1,0,1) ..., first minimum wins (strict '>');
//   * d(i, j) = d(i + 1, j) + 1 = f(j) + (r0 - i) with f = d(r0, .): the tap (+1,0) always reaches the minimum, so only the
//     four taps in front of it can win instead, and for i <= r0 - 2 (both tap rows still follow d = f + const) they reduce
//     to: (+2,+1) if f(j+1) = f(j) - 1, else (+2,-1) if f(j-1) = f(j) - 1, else (+1,0) -- a step code per COLUMN, the same
//     for every such row ((+1,+2) and (+1,+1) can only match when (+2,+1) already has: f is 1-Lipschitz);
//   * row r0 - 1 applies the five taps to the real distances of rows r0 and r0 + 1.
// label and depth of q are those of the pixel its chain reaches in rows r0 / r0 + 1, which the other kernels of the pass have
// finished.  No row-by-row propagation is needed: the step code of a column is the same in every row up to r0 - 2, a (+2,+1)
// column is followed by (+2,+1) columns up to a (+1,0) column (a (+2,-1) column cannot follow: f(j+1) = f(j) - 1 rules it
// out; mirror image for (+2,-1)), and a (+1,0) column repeats what row r0 - 1 holds all the way up.  So pixel (i, j) takes
//     t = min(run length of column j, floor((r0 - i) / 2))
// hops of (+2, +-1) and lands in row i + 2 t, column j +- t: in row r0 on that base pixel itself, in any row above on what row
// r0 - 1 holds in that column -- O(1) per pixel, every pixel independent (tests/parallel_model.py::sky_rows_closed_form).
// The blocks ride behind k_colT's in its launch (the first launch after the window kernel, whose rows r0 and r0 + 1 they read;
// should the window kernel have had to hand one of those on, it has called the sky off and the any-distance kernels take its rows).
// A block takes SKY_RG rows x SKY_SW columns of one frame's sky: it stages f, the base rows' (label, depth) pairs and row
// r0 - 1's pairs for its columns and the columns its hops can reach in LDS, then every thread stores its column's SKY_RG
// pixels.  Reads the base rows from the pass's own outputs (dt from the scratch when the caller wants none).
// ------------------------------------------------------------------------------------------------
constexpr int SKY_NT = 256;  // threads = columns of a block
constexpr int SKY_SW = 256;
constexpr int SKY_RG = 16;   // rows of a block
__host__ __device__ inline int sky_span_max(int H, int W) { return min(W, SKY_SW + 2 * (min(H, SKY_MAX) / 2) + 4); }
__host__ __device__ inline size_t sky_lds(int span) { return (size_t)((span + 7) & ~7) * (2 * sizeof(u16) + 3 * sizeof(uint2)); }

__device__ __forceinline__ void sky_body(unsigned char *s_sky, const int *__restrict__ finfo, int H, int W, const float *dt_src, float *out_dt,
                                         float *out_depth, int32_t *out_index, int strip, int rowgroup, int b) {
    const int tid = threadIdx.x, NT = blockDim.x;  // (at least SKY_NT threads: the host sees to it)
    const int r0 = finfo[b * FI_STRIDE + FI_SKY];
    const int i0 = rowgroup * SKY_RG;
    if (r0 <= 0 || r0 >= H || i0 >= r0) return;  // block-uniform: no sky in this frame (or called off), or not this far down
    const int i1 = min(i0 + SKY_RG, r0);
    const int T = (r0 - i0) / 2;  // the most hops a pixel of these rows takes
    const int c0 = strip * SKY_SW, c1 = min(W, c0 + SKY_SW);  // the columns this block stores
    // the columns it looks at: its own, the T columns either side a hop chain can end in, and what row r0 - 1's rule reads
    // around those (one column to the left, two to the right)
    const int lo = max(0, c0 - T - 1), hi = min(W, c1 + T + 3), n = hi - lo;
    const int Np = (n + 7) & ~7;
    uint2 *s_b0 = reinterpret_cast<uint2 *>(s_sky);  // [Np] {label, depth bits} of row r0
    uint2 *s_b1 = s_b0 + Np;                          // [Np] row r0 + 1
    uint2 *s_first = s_b1 + Np;                       // [Np] row r0 - 1
    u16 *s_f0 = reinterpret_cast<u16 *>(s_first + Np);  // [Np] d(r0, .)
    u16 *s_f1 = s_f0 + Np;                              // [Np] d(r0 + 1, .); 0xFFFF without such a row
    const size_t fo = (size_t)b * H * W;
    const bool two = r0 + 1 < H;
    for (int k = tid; k < n; k += NT) {
        const size_t o0 = fo + (size_t)r0 * W + lo + k, o1 = o0 + (two ? W : 0);
        s_f0[k] = (u16)dt_src[o0];
        s_f1[k] = two ? (u16)dt_src[o1] : (u16)0xFFFF;
        s_b0[k] = make_uint2(out_index ? (u32)out_index[o0] : 0u, out_depth ? __float_as_uint(out_depth[o0]) : 0u);
        s_b1[k] = make_uint2(out_index ? (u32)out_index[o1] : 0u, out_depth ? __float_as_uint(out_depth[o1]) : 0u);
    }
    __syncthreads();
    // row r0 - 1: the five leading backward taps on the real distances (a span edge inside the frame yields a stale entry that
    // no hop chain of this block can end in)
    for (int k = tid; k < n; k += NT) {
        const int D = (int)s_f0[k] + 1;
        const int fp1 = k + 1 < n ? (int)s_f0[k + 1] : BIG, fp2 = k + 2 < n ? (int)s_f0[k + 2] : BIG;
        const int gp1 = k + 1 < n ? (int)s_f1[k + 1] : BIG, gm1 = k >= 1 ? (int)s_f1[k - 1] : BIG;
        s_first[k] = gp1 + 3 == D ? s_b1[k + 1] : gm1 + 3 == D ? s_b1[k - 1] : fp2 + 3 == D ? s_b0[k + 2] : fp1 + 2 == D ? s_b0[k + 1] : s_b0[k];
    }
    // this thread's column: direction and length (capped at T) of its run of hopping columns
    const int j = c0 + tid, k = j - lo;
    int dir = 0, run = 0, f = 0;
    if (j < c1) {
        f = (int)s_f0[k];
        dir = (k + 1 < n && (int)s_f0[k + 1] + 1 == f) ? 1 : (k >= 1 && (int)s_f0[k - 1] + 1 == f) ? -1 : 0;
        if (dir) {
            int kk = k, fc = f;  // kk hops along while the next column is one nearer
            while (run < T && kk + dir >= 0 && kk + dir < n && (int)s_f0[kk + dir] + 1 == fc) {
                kk += dir;
                fc -= 1;
                ++run;
            }
        }
    }
    __syncthreads();
    if (j >= c1) return;
    for (int i = i0; i < i1; ++i) {
        const int t = i == r0 - 1 ? 0 : min(run, (r0 - i) >> 1);
        const int kk = k + dir * t;
        const uint2 v = (i + 2 * t == r0) ? s_b0[kk] : s_first[kk];
        const size_t o = fo + (size_t)i * W + j;
        if (out_dt) __builtin_nontemporal_store((float)(f + r0 - i), &out_dt[o]);
        if (out_index) __builtin_nontemporal_store((int32_t)v.x, &out_index[o]);
        if (out_depth) __builtin_nontemporal_store(__uint_as_float(v.y), &out_depth[o]);
    }
}

// k_sky's blocks as a launch of their own: for frames so tall that k_colT's blocks are far larger than SKY_NT threads (its
// launch would then start thousands of 1024-thread blocks for the sky units)
__global__ __launch_bounds__(SKY_NT) void k_sky(const int *__restrict__ finfo, int H, int W, const float *dt_src, float *out_dt, float *out_depth,
                                                int32_t *out_index, int nstrips) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_skyraw[];
    sky_body(s_skyraw, finfo, H, W, dt_src, out_dt, out_depth, out_index, (int)blockIdx.x % nstrips, (int)blockIdx.x / nstrips, (int)blockIdx.y);
}
