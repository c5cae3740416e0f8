// dtfill_l2.hpp -- k_l2row: exact Euclidean row search (l2 metric)
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

// ------------------------------------------------------------------------------------------------
// l2 metric: k_l2row.  One lane per pixel.  With g(i,k) the vertical distance to the nearest source of
// column k (from k_colT's band words), the exact squared Euclidean distance is min_k g(i,k)^2 + (j-k)^2; the columns are visited
// outward from j (r = |j-k| = 0,1,2,...) and the search stops once r^2 exceeds the best value, so the
// work per pixel is ~2 sqrt(d^2) candidates.  Ties go to the smallest raster index of the SOURCE
// (smaller row, then smaller column) -- the order brute force gives.  Then rank -> label, gather, store.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_l2row(const float *__restrict__ x, const uint2 *__restrict__ ct, int CTP, int nb,
                                               const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s,
                                               const u32 *__restrict__ rowbase_s, const int *__restrict__ finfo,
                                               const float *__restrict__ vlist, int H, int W, int Wd,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    // one block per image row: the row of g is staged in LDS once, every pixel's outward search reads it there
    extern __shared__ __attribute__((aligned(16))) u16 s_grow[];
    const int b = blockIdx.y, i = blockIdx.x;
    const size_t fo = (size_t)b * H * W;
    {
        // vertical distance to the nearest source of every column, from k_colT's band words (as in k_rows); bit 15:
        // that source is BELOW this row (strictly nearer than the one above: on a vertical tie the upper source has the
        // smaller raster index)
        const int band = i >> 5, r = i & 31;
        const uint2 *crow = ct + ((size_t)b * nb + band) * CTP;
        const u32 upmask = (2u << r) - 1u;
        for (int k = threadIdx.x; k < W; k += 256) {
            const uint2 c = crow[k];
            const u32 gu = min(ffbh_u32(c.x & upmask) + (u32)(r - 31), (c.y & 0xFFFFu) + (u32)r);
            const u32 gd = min(ffbl_b32(c.x >> r), (c.y >> 16) + (u32)(31 - r));
            const u32 m = min(gu, gd);
            s_grow[k] = m >= (u32)MAX_HW_SUM ? (u16)INF16 : (u16)(m | (gd < gu ? 0x8000u : 0u));  // >= 8192: no source in the column
        }
    }
    __syncthreads();
    const u16 *grow = s_grow;
    for (int j = threadIdx.x; j < W; j += 256) {
    const int p = i * W + j;
    // best candidate: key = (d2, source row, source column), lexicographic
    long long best = 0x7FFFFFFFFFFFFFFFll;
    int bestd2 = 0x7FFFFFFF;
    const int rmax = finfo[b * FI_STRIDE + FI_NSRC] ? W : 0;  // a frame without sources has nothing to search
    constexpr int RC = 4;  // radii per chunk: their 2*RC loads are issued together, then applied in order
    for (int r0 = 0; r0 < rmax; r0 += RC) {
        if ((long long)r0 * r0 > bestd2) break;  // r*r == bestd2 still matters: a same-row source ties on d2
        int v[RC][2];
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const int kl = j - (r0 + u), kr = j + (r0 + u);
            v[u][0] = kl >= 0 ? (int)grow[kl] : INF16;
            v[u][1] = (kr < W && r0 + u > 0) ? (int)grow[kr] : INF16;
        }
#pragma unroll
        for (int u = 0; u < RC; ++u) {
            const int r = r0 + u;
            if ((long long)r * r > bestd2) break;
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                const int vv = v[u][side];
                if (vv == INF16) continue;
                const int k = side ? j + r : j - r;
                const int gv = vv & 0x7FFF;
                const int srow = (vv & 0x8000) ? i + gv : i - gv;
                const int d2 = gv * gv + r * r;
                const long long key = ((long long)d2 << 32) | ((long long)srow << 16) | k;
                if (key < best) {
                    best = key;
                    bestd2 = d2;
                }
            }
        }
    }
    int label = 0, q = p;
    float dist = INFINITY;
    if (bestd2 != 0x7FFFFFFF) {
        const int srow = (int)((best >> 16) & 0xFFFF), scol = (int)(best & 0xFFFF);
        q = srow * W + scol;
        const size_t w = ((size_t)b * H + srow) * Wd + (scol >> 6);
        label = source_rank(rowbase_s[(size_t)b * H + srow] + wpre_s[w], srcbits[w], scol);
        dist = sqrtf((float)bestd2);
    }
    if (out_index) out_index[fo + p] = label;
    if (out_dt) out_dt[fo + p] = dist;
    if (out_depth)
        out_depth[fo + p] = gather_depth(x + fo, vlist + fo, label, q, finfo[b * FI_STRIDE + FI_NVAL],
                                         finfo[b * FI_STRIDE + FI_MISALIGNED], frame_status + b);
    }
}
