/*
 * dtfill_oracle.c -- CPU restatement of the reference's DT + nearest-valid-depth fill path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it, and
 * there only as the checker / the CPU number printed beside the GPU one.
 *
 * PARITY UNPINNED.  The arithmetic of the reference path lives in a third-party dependency
 * that is not under /root/reference: OpenCV `cv2.distanceTransformWithLabels`, pinned by
 * `pip3 install opencv-contrib-python==3.4.2.16` (install_dependency.sh:3-4), called at
 * solution_DeepNet/tools.py:9, demo.py:80, eval_NYU.py:116.  cv2 is not installed in this
 * image and the reference ships no tests, fixtures or recorded outputs for this path, so
 * this file restates the published algorithm of OpenCV 3.4 imgproc/distransform.cpp
 * (cv::distanceTransform labels branch -> distanceTransformEx_5x5) from knowledge of that
 * source; it could not be checked against a real cv2 here.  Independent cross-checks that
 * ARE run (tests/test_oracle.py): distances equal scipy.ndimage.distance_transform_cdt
 * (taxicab); every label is a true L1-nearest source; brute force on tiny frames.
 *
 * What is restated (reference file:line):
 *   cvdt_l1_labels      <- cv2.distanceTransformWithLabels(mask, DIST_L1, 5, DIST_LABEL_PIXEL)
 *                          as called at tools.py:9 (OpenCV 3.4: label init loop + 5x5 two-pass
 *                          chamfer, weights {1,2,3} in Q16 fixed point, strict '>' updates).
 *   oracle_nearest_point<- nearest_point(), tools.py:7-10 (src_thr 0.1) and
 *                          eval_NYU.py:114-117 (src_thr 0.001).
 *   oracle_fill_frame   <- the per-frame body of DT_complete_batch(), tools.py:16-27, and
 *                          Distance_Transform(), eval_NYU.py:120-133: value list compaction,
 *                          gather depth_list[label-1] with numpy negative-index semantics.
 *   edt_*               <- the `l2` mode BASELINE.json's north_star asks for (exact Euclidean
 *                          transform, canonical tie-break = smallest raster index); there is no
 *                          reference code for it -- brute force is its definition.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>

#define DT_SHIFT 16
#define DT_INIT (INT_MAX >> 2)

/* status codes of oracle_fill_frame (mirror of include/dtfill.h frame status) */
#define ORACLE_OK 0
#define ORACLE_INDEX_ERROR 1 /* numpy would raise IndexError in depth_list[label-1] */

/* ---- OpenCV 3.4 distanceTransformWithLabels(DIST_L1, 5, DIST_LABEL_PIXEL) ------------------
 * mask: uint8 [H,W], 0 = source ("zero pixel"), !=0 = pixel to fill.
 * dist: float32 [H,W]; labels: int32 [H,W] (1-based raster rank of the winning source, 0 = none).
 * Follows cv::distanceTransform (labels branch) + distanceTransformEx_5x5.                    */
void cvdt_l1_labels(const uint8_t *mask, int H, int W, float *dist, int32_t *labels)
{
    const int B = 2;
    const int step = W + 2 * B;
    const int HV = 1 << DT_SHIFT, DIAG = 2 << DT_SHIFT, LONG = 3 << DT_SHIFT;
    const float scale = 1.f / (1 << DT_SHIFT);
    int *temp = (int *)malloc(sizeof(int) * (size_t)(H + 2 * B) * step);
    int i, j;

    /* label init: labels.setTo(0, src); k=1; raster scan, every zero pixel gets k++ */
    {
        int k = 1;
        for (i = 0; i < H * W; i++)
            labels[i] = mask[i] == 0 ? k++ : 0;
    }
    /* initTopBottom: the border rows are INIT_DIST0 */
    for (i = 0; i < B; i++)
        for (j = 0; j < step; j++) {
            temp[i * step + j] = DT_INIT;
            temp[(H + B + i) * step + j] = DT_INIT;
        }

    /* forward pass */
    for (i = 0; i < H; i++) {
        const uint8_t *s = mask + (size_t)i * W;
        int *tmp = temp + (i + B) * step + B;
        int32_t *lls = labels + (size_t)i * W;
        for (j = 0; j < B; j++)
            tmp[-j - 1] = tmp[W + j] = DT_INIT;
        for (j = 0; j < W; j++) {
            if (!s[j]) {
                tmp[j] = 0;
            } else {
                int t0 = DT_INIT, t;
                int32_t l0 = 0;
#define FWD(off_t, off_l, w)            \
    t = tmp[j + (off_t)] + (w);         \
    if (t0 > t) {                       \
        t0 = t;                         \
        l0 = lls[j + (off_l)];          \
    }
                FWD(-step * 2 - 1, -W * 2 - 1, LONG)
                FWD(-step * 2 + 1, -W * 2 + 1, LONG)
                FWD(-step - 2, -W - 2, LONG)
                FWD(-step - 1, -W - 1, DIAG)
                FWD(-step, -W, HV)
                FWD(-step + 1, -W + 1, DIAG)
                FWD(-step + 2, -W + 2, LONG)
                FWD(-1, -1, HV)
#undef FWD
                tmp[j] = t0;
                lls[j] = l0;
            }
        }
    }
    /* backward pass */
    for (i = H - 1; i >= 0; i--) {
        float *d = dist + (size_t)i * W;
        int *tmp = temp + (i + B) * step + B;
        int32_t *lls = labels + (size_t)i * W;
        for (j = W - 1; j >= 0; j--) {
            int t0 = tmp[j];
            int32_t l0 = lls[j];
            if (t0 > HV) {
                int t;
#define BWD(off_t, off_l, w)            \
    t = tmp[j + (off_t)] + (w);         \
    if (t0 > t) {                       \
        t0 = t;                         \
        l0 = lls[j + (off_l)];          \
    }
                BWD(step * 2 + 1, W * 2 + 1, LONG)
                BWD(step * 2 - 1, W * 2 - 1, LONG)
                BWD(step + 2, W + 2, LONG)
                BWD(step + 1, W + 1, DIAG)
                BWD(step, W, HV)
                BWD(step - 1, W - 1, DIAG)
                BWD(step - 2, W - 2, LONG)
                BWD(1, 1, HV)
#undef BWD
                tmp[j] = t0;
                lls[j] = l0;
            }
            d[j] = (float)(t0 * scale);
        }
    }
    free(temp);
}

/* nearest_point(): tools.py:7-10 / eval_NYU.py:114-117.
 * value_mask = uint8((1.0 - x) > src_thr) evaluated in float32 exactly as numpy does for a
 * float32 array and Python-float scalars (NEP 50 weak scalars: 1.0 and the threshold are cast
 * to float32, the subtraction and the compare are float32).                                */
void oracle_nearest_point(const float *x, int H, int W, float src_thr, float *dt, int32_t *lbl)
{
    size_t n = (size_t)H * W, p;
    uint8_t *mask = (uint8_t *)malloc(n);
    for (p = 0; p < n; p++) {
        volatile float one_minus = 1.0f - x[p];
        mask[p] = (one_minus > src_thr) ? 1 : 0;
    }
    cvdt_l1_labels(mask, H, W, dt, lbl);
    free(mask);
}

/* One frame of DT_complete_batch (tools.py:16-27) / Distance_Transform (eval_NYU.py:120-133):
 *   with_value = x > val_thr ; depth_list = x[with_value] ; out = depth_list[lbl - 1]
 * numpy semantics of the gather: index -1 (label 0) wraps to the LAST element; an index outside
 * [-n, n) raises IndexError -> returns ORACLE_INDEX_ERROR (outputs then unspecified).
 * Any of out_depth / out_dt / out_lbl may be NULL.                                            */
int oracle_fill_frame(const float *x, int H, int W, float src_thr, float val_thr,
                      float *out_depth, float *out_dt, int32_t *out_lbl)
{
    size_t n = (size_t)H * W, p, nval = 0;
    float *dt = out_dt ? out_dt : (float *)malloc(n * sizeof(float));
    int32_t *lbl = out_lbl ? out_lbl : (int32_t *)malloc(n * sizeof(int32_t));
    float *vlist = (float *)malloc(n * sizeof(float) + 4);
    int rc = ORACLE_OK;

    oracle_nearest_point(x, H, W, src_thr, dt, lbl);
    for (p = 0; p < n; p++)
        if (x[p] > val_thr)
            vlist[nval++] = x[p];
    for (p = 0; p < n; p++) {
        long long idx = (long long)lbl[p] - 1;
        if (idx < 0)
            idx += (long long)nval;
        if (idx < 0 || idx >= (long long)nval) {
            rc = ORACLE_INDEX_ERROR;
            break;
        }
        if (out_depth)
            out_depth[p] = vlist[idx];
    }
    free(vlist);
    if (!out_dt)
        free(dt);
    if (!out_lbl)
        free(lbl);
    return rc;
}

/* Batched form, frames [B,H,W] contiguous; status[b] per frame. Returns number of failed frames. */
int oracle_fill_batch(const float *x, int B, int H, int W, float src_thr, float val_thr,
                      float *out_depth, float *out_dt, int32_t *out_lbl, int32_t *status)
{
    size_t n = (size_t)H * W;
    int b, bad = 0;
    for (b = 0; b < B; b++) {
        int rc = oracle_fill_frame(x + b * n, H, W, src_thr, val_thr,
                                   out_depth ? out_depth + b * n : NULL,
                                   out_dt ? out_dt + b * n : NULL,
                                   out_lbl ? out_lbl + b * n : NULL);
        if (status)
            status[b] = rc;
        bad += rc != ORACLE_OK;
    }
    return bad;
}

/* ---- the `l2` mode: exact Euclidean transform + the same value-list glue --------------------------
 * out_dt = sqrtf(squared distance) (+inf without sources); out_idx = 1-based raster rank of the nearest
 * source, ties -> smallest raster index (0 without sources); out_depth = depth_list[idx-1] with the
 * numpy index semantics of tools.py:26.  There is no reference code for this mode (the reference is
 * L1 only); BASELINE.json's north_star asks for it, brute force (brute_nearest) defines it.          */
void edt_l2_labels(const uint8_t *mask, int H, int W, int32_t *dist2, int32_t *near);
#include <math.h>
int oracle_fill_frame_l2(const float *x, int H, int W, float src_thr, float val_thr,
                         float *out_depth, float *out_dt, int32_t *out_idx)
{
    size_t n = (size_t)H * W, p, nval = 0;
    uint8_t *mask = (uint8_t *)calloc(n ? n : 1, 1);
    int32_t *d2 = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *near = (int32_t *)malloc(n * sizeof(int32_t));
    int32_t *rank = (int32_t *)malloc(n * sizeof(int32_t));
    float *vlist = (float *)malloc(n * sizeof(float) + 4);
    int rc = ORACLE_OK, k = 0;
    for (p = 0; p < n; p++) {
        volatile float one_minus = 1.0f - x[p];
        mask[p] = (one_minus > src_thr) ? 1 : 0;
        rank[p] = mask[p] ? 0 : ++k;
        if (x[p] > val_thr)
            vlist[nval++] = x[p];
    }
    edt_l2_labels(mask, H, W, d2, near);
    for (p = 0; p < n; p++) {
        int32_t lbl = near[p] < 0 ? 0 : rank[near[p]];
        long long idx = (long long)lbl - 1;
        if (out_idx) out_idx[p] = lbl;
        if (out_dt) out_dt[p] = near[p] < 0 ? INFINITY : sqrtf((float)d2[p]);
        if (idx < 0) idx += (long long)nval;
        if (idx < 0 || idx >= (long long)nval)
            rc = ORACLE_INDEX_ERROR;
        else if (out_depth)
            out_depth[p] = vlist[idx];
    }
    free(mask); free(d2); free(near); free(rank); free(vlist);
    return rc;
}

int oracle_fill_batch_l2(const float *x, int B, int H, int W, float src_thr, float val_thr,
                         float *out_depth, float *out_dt, int32_t *out_idx, int32_t *status)
{
    size_t n = (size_t)H * W;
    int b, bad = 0;
    for (b = 0; b < B; b++) {
        int rc = oracle_fill_frame_l2(x + b * n, H, W, src_thr, val_thr, out_depth ? out_depth + b * n : NULL,
                                      out_dt ? out_dt + b * n : NULL, out_idx ? out_idx + b * n : NULL);
        if (status) status[b] = rc;
        bad += rc != ORACLE_OK;
    }
    return bad;
}

/* ---- brute force nearest source, tiny frames only (O(N*K)) ----------------------------------
 * metric 1 = L1, 2 = squared L2.  Tie-break: smallest raster index of the source (canonical).
 * dist2: int32 distance (L1) or squared distance (L2); INT_MAX where no source exists.
 * near : raster index of the chosen source, -1 if none.                                       */
void brute_nearest(const uint8_t *mask, int H, int W, int metric, int32_t *dist2, int32_t *near)
{
    int n = H * W, p, q;
    for (p = 0; p < n; p++) {
        int pi = p / W, pj = p % W, best = INT_MAX, arg = -1;
        for (q = 0; q < n; q++) {
            if (mask[q])
                continue;
            int di = pi - q / W, dj = pj - q % W, dd;
            if (di < 0) di = -di;
            if (dj < 0) dj = -dj;
            dd = metric == 1 ? di + dj : di * di + dj * dj;
            if (dd < best) {
                best = dd;
                arg = q;
            }
        }
        dist2[p] = best;
        near[p] = arg;
    }
}

/* ---- exact Euclidean transform with canonical tie-break (the `l2` mode) ---------------------
 * Separable: per column nearest-source rows, then per row a scan over columns keeping, for
 * every column k, the best vertical candidate.  O(H*W*W) worst case but simple and obviously
 * right; pruned by the current best so it is fast on sparse-but-not-empty frames.
 * Tie-break = smallest raster index among sources at equal squared distance.                 */
void edt_l2_labels(const uint8_t *mask, int H, int W, int32_t *dist2, int32_t *near)
{
    int i, j, k;
    /* up[i*W+k] / dn[i*W+k]: row of nearest source at or above / at or below row i in column k */
    int32_t *up = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * W);
    int32_t *dn = (int32_t *)malloc(sizeof(int32_t) * (size_t)H * W);
    for (k = 0; k < W; k++) {
        int last = -1;
        for (i = 0; i < H; i++) {
            if (!mask[i * W + k]) last = i;
            up[i * W + k] = last;
        }
        last = -1;
        for (i = H - 1; i >= 0; i--) {
            if (!mask[i * W + k]) last = i;
            dn[i * W + k] = last;
        }
    }
    for (i = 0; i < H; i++) {
        for (j = 0; j < W; j++) {
            long long best = LLONG_MAX;
            int arg = -1, r;
            /* expand outward in |dj| so the loop can stop once dj*dj > best */
            for (r = 0; r < W; r++) {
                int side;
                if ((long long)r * r > best)
                    break;
                for (side = 0; side < 2; side++) {
                    k = side ? j + r : j - r;
                    if (k < 0 || k >= W || (side && r == 0))
                        continue;
                    int u = up[i * W + k], d = dn[i * W + k], c;
                    for (c = 0; c < 2; c++) {
                        int si = c ? d : u;
                        if (si < 0)
                            continue;
                        long long dd = (long long)(i - si) * (i - si) + (long long)r * r;
                        int q = si * W + k;
                        if (dd < best || (dd == best && q < arg)) {
                            best = dd;
                            arg = q;
                        }
                    }
                }
            }
            dist2[i * W + j] = arg < 0 ? INT_MAX : (int32_t)best;
            near[i * W + j] = arg;
        }
    }
    free(up);
    free(dn);
}
