/*
 * dtfill.h -- C ABI of libdtfill.so: the MI355X (gfx950) distance-transform + nearest-valid-depth
 * fill operator.  Plain pointers and sizes only; no torch / HIP types in the signatures
 * (`stream` is a hipStream_t passed as void*, NULL = the default stream).
 *
 * What each entry point replaces in the reference (placeforyiming/DistanceTransform-DepthCompletion):
 *
 *   dtfill_batch()            the per-frame body of DT_complete_batch(), solution_DeepNet/tools.py:13-35
 *                             (live copy demo.py:84-106), and Distance_Transform(),
 *                             solution_DeepNet/eval_NYU.py:120-133, for B frames at once:
 *                               nearest_point()                      tools.py:7-10 / eval_NYU.py:114-117
 *                                 value_mask = uint8((1.0 - x) > src_thr)
 *                                 cv2.distanceTransformWithLabels(value_mask, DIST_L1, 5, DIST_LABEL_PIXEL)
 *                               with_value = x > val_thr ; depth_list = x[with_value]   tools.py:22,24
 *                               out = depth_list[lbl - 1]                               tools.py:26
 *                             out_dt / out_index are the (dt, lbl) pair nearest_point() returns.
 *   dtfill_workspace_bytes()  the temporaries numpy/cv2 allocate implicitly (cv2's (H+4)x(W+4) int32
 *                             `temp`, the masks, depth_list); here the caller owns them.
 *   dtfill_strerror()         the numpy / cv2 exceptions the reference surfaces (tools.py:26 IndexError).
 *
 * All pointers are DEVICE pointers unless stated otherwise.  Frames are float32, C-contiguous
 * [B, H, W] (the shim slices channel 0 of the reference's [B,H,W,1] layout).  The library keeps
 * no global state, allocates nothing, and every call is ordered on `stream` only: it is safe to
 * call from several host threads on different streams / devices with different workspaces.
 */
#ifndef DTFILL_H
#define DTFILL_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DTFILL_ABI_VERSION 1

/* metric */
#define DTFILL_METRIC_L1_CV 0 /* reference parity: OpenCV L1 5x5 chamfer + its label tie-break order */
#define DTFILL_METRIC_L2    1 /* exact Euclidean transform, tie-break = smallest raster index of the source
                                 (not in the reference, which is L1 only; BASELINE.json's north_star asks for it) */

/* return codes */
#define DTFILL_OK               0
#define DTFILL_ERR_NULL        -1 /* x, workspace or every output is NULL */
#define DTFILL_ERR_SHAPE       -2 /* B,H,W < 1, B > 65535, B*H*W >= 2^31, or H+W-2 >= 8192 (cv2's Q16 INIT_DIST0 range) */
#define DTFILL_ERR_WORKSPACE   -3 /* ws_bytes < dtfill_workspace_bytes() or workspace not 256-B aligned */
#define DTFILL_ERR_METRIC      -4 /* unknown metric */
#define DTFILL_ERR_LAUNCH      -5 /* a HIP launch failed (hipGetLastError) */
#define DTFILL_ERR_NO_DEVICE   -6 /* no usable HIP device */

/* per-frame status written to frame_status[b] (device int32): a bit set */
#define DTFILL_FRAME_OK           0
#define DTFILL_FRAME_INDEX_ERROR  1 /* numpy would raise IndexError in depth_list[lbl-1] (tools.py:26):
                                       a label addresses past the value list, or label 0 (no source in
                                       the frame) with an empty value list.  out_depth of that frame is
                                       then unspecified; out_dt / out_index are still exact. */
#define DTFILL_FRAME_GENERAL_PATH 2 /* informational (l1_cv): the frame, or some of its rows, was computed outside the LDS
                                       window kernel -- by the any-distance kernels (sparse frame, rows with a pixel
                                       farther than the halo from every source), k_sky (the rows above every source)
                                       or k_pts (a handful of sources); results are identical. */

/* flags of dtfill_batch_flags(): path selection, for tests and benchmarks */
#define DTFILL_FLAG_GENERAL_ONLY 1u /* skip the window kernel, every frame takes the any-distance kernels */
#define DTFILL_FLAG_FUSED_ONLY   2u /* skip the any-distance kernels: frames that need them are left
                                       undefined and carry DTFILL_FRAME_GENERAL_PATH in their status */
/* ... and one that changes the result: the loader's outlier_removal() (data_read.py:103-128, applied to the sparse map
   before anything else sees it, data_read.py:168-169) in front of the predicates.  The pass then equals
   dtfill_outlier_removal() followed by dtfill_batch() on its output, without the filtered map ever being written:
   a removed pixel stops being a source / value, the surviving ones are gathered from x.  Needs H, W >= 4. */
#define DTFILL_FLAG_OUTLIER_REMOVAL 4u

int dtfill_abi_version(void);
const char *dtfill_strerror(int code);

/* Bytes of scratch dtfill_batch() needs for this shape (0 on a bad shape). 256-B aligned carve. */
size_t dtfill_workspace_bytes(int B, int H, int W, int metric);

/*
 * One pass of the hot path over B frames.
 *   x            float32 [B,H,W], never written.
 *   src_thr      a pixel is a SOURCE iff NOT((1.0f - x) > src_thr) in float32   (tools.py:8: 0.1,
 *                eval_NYU.py:115: 0.001).  NaN is therefore a source, as in numpy.
 *   val_thr      a pixel enters the value list iff x > val_thr                  (tools.py:22: 0.1)
 *   out_depth    float32 [B,H,W] filled depth  = depth_list[lbl-1]              (nullable)
 *   out_dt       float32 [B,H,W] distance map  (l1_cv: integer-valued L1, 8192.0 in a frame with
 *                no source; l2: sqrtf of the exact squared distance, +inf if no source) (nullable)
 *   out_index    int32 [B,H,W]: l1_cv: cv2's label (1-based raster rank of the nearest source under
 *                cv2's tie-break, 0 = no source); l2: the same 1-based rank under the canonical
 *                tie-break                                                      (nullable)
 *   frame_status int32 [B] (nullable): DTFILL_FRAME_*
 *   workspace    ws_bytes >= dtfill_workspace_bytes(B,H,W,metric), 256-B aligned
 * Returns DTFILL_OK or a negative DTFILL_ERR_*.  Asynchronous: outputs are valid after `stream`
 * has been synchronised.
 */
int dtfill_batch(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                 float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                 void *workspace, size_t ws_bytes, void *stream);

/* dtfill_batch() with explicit path selection (DTFILL_FLAG_*); flags = 0 is dtfill_batch(). */
int dtfill_batch_flags(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                       float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                       void *workspace, size_t ws_bytes, void *stream, unsigned flags);

/*
 * dtfill_batch_flags() with the drivers' post-fill steps folded into the depth stores (l1_cv only):
 *   depth_row0  out_depth holds rows [depth_row0, H) of every frame: float32 [B, H - depth_row0, W]
 *               (demo.py:292-293: lidar_batch = lidar_batch[:, 96:, :, :] right after DT_complete_batch)
 *   use_floor   out_depth = relu(d - floor_) + floor_ in float32, both roundings kept
 *               (eval_NYU.py:205, test.py:133: the 0.9 m depth floor)
 * out_dt / out_index stay whole frames.  Saves the separate dtfill_crop_floor() pass over the filled depth.
 * depth_row0 = 0 and use_floor = 0 is dtfill_batch_flags().  DTFILL_ERR_METRIC for the l2 metric with an epilogue.
 */
int dtfill_batch_epilogue(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                          float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                          void *workspace, size_t ws_bytes, void *stream, unsigned flags, int depth_row0,
                          int use_floor, float floor_);

/*
 * Same pass, instrumented for bench.py: records a HIP event on `stream` before and after every
 * kernel, synchronises, and returns each kernel's duration in milliseconds in kernel_ms (HOST
 * float[dtfill_num_kernels(metric)]; 0 for kernels the flags skip).  Not for production use (it blocks).
 */
int dtfill_num_kernels(int metric);
const char *dtfill_kernel_name(int metric, int k);
int dtfill_batch_timed(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                       float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                       void *workspace, size_t ws_bytes, void *stream, unsigned flags, float *kernel_ms);

/*
 * Which kernel family owned how many pixels in the LAST pass run in `workspace` (same B, H, W, metric) -- for bench.py's
 * per-kernel roofline figures: a kernel is credited with the bytes of the pixels it processed, not with the whole batch.
 * out_px: DEVICE int64[DTFILL_STATS_N], written on `stream`:
 *   [DTFILL_STATS_ALL]     B*H*W
 *   [DTFILL_STATS_WINDOW]  pixels of the rows the window kernel kept (l1_cv: k_fused; l2: k_l2win, rows it did not hand on)
 *   [DTFILL_STATS_ANYDIST] pixels of the rows the any-distance kernels stored (l1_cv: k_rows / k_fin; l2: k_l2env's rows)
 *   [DTFILL_STATS_SKY]     l1_cv: pixels of the rows k_sky stored
 *   [DTFILL_STATS_POINTS]  pixels of the frames with a handful of sources (l1_cv: k_pts; l2: k_l2env's tiles)
 *   [DTFILL_STATS_COLT]    pixels of the frames k_colT ran for
 */
#define DTFILL_STATS_ALL 0
#define DTFILL_STATS_WINDOW 1
#define DTFILL_STATS_ANYDIST 2
#define DTFILL_STATS_SKY 3
#define DTFILL_STATS_POINTS 4
#define DTFILL_STATS_COLT 5
#define DTFILL_STATS_N 6
int dtfill_pass_stats(const void *workspace, size_t ws_bytes, int B, int H, int W, int metric, long long *out_px, void *stream);

/*
 * outlier_removal() of the reference's loader, data_read.py:103-128 (applied in front of the path when
 * if_removal=True, data_read.py:168-169): two 7x7-diamond cv2.filter2D sums (values, valid counts;
 * BORDER_REFLECT_101), out = x where NOT (x - sum/(count + 1e-5) > 1.0), else 0.  The float32 sum is
 * accumulated over the taps in kernel row-major order, the mean and the test in float64, as
 * numpy/OpenCV do.  x, out: float32 [B,H,W] device pointers (out may not alias x).  Needs H, W >= 4.
 */
int dtfill_outlier_removal(const float *x, int B, int H, int W, float *out, void *stream);

/*
 * generate_multi_channel() of the reference's models, solution_DeepNet/net.py:83-122 (weights
 * create_weight_matrix, net.py:71-81): the in-network windowed nearest fill.  Step k: over the
 * table_size^2 window (zero padding), s = mask * (table_size - |di| - |dj|); out = sum of the inputs where s
 * equals its window maximum / (1e-6 + their count); the next step's mask is (out > 0.001).
 * data, mask: float32 [B,H,W] (the reference's [B,H,W,1]); out2/out3/out4: the reference's lidar_2..4
 * (those beyond scale_num may be NULL); lidar_1 is the input itself.  table_size odd, <= 15; scale_num 1..4.
 * float32; the window sum is accumulated in tap (row-major) order.
 */
int dtfill_generate_multi_channel(const float *data, const float *mask, int B, int H, int W, int table_size,
                                  int scale_num, float *out2, float *out3, float *out4, void *stream);

/*
 * What the reference's drivers do with a filled frame.
 *
 * dtfill_crop_floor: out[b, i, j] = f(x[b, r0+i, c0+j]) for rows [r0, r1), columns [c0, c1); f is the depth
 * floor relu(d - floor) + floor in float32, both roundings kept (eval_NYU.py:205, test.py:133: floor 0.9), when
 * use_floor != 0, the identity otherwise.  Replaces lidar_batch[:, 96:, :, :] (demo.py:292-293), the NYU
 * evaluation crop [6:234, 8:312] (eval_NYU.py:202-203) and the KITTI depth floor (eval_NYU.py:205).
 * x: float32 [B,H,W]; out: float32 [B, r1-r0, c1-c0], may not alias x.
 *
 * dtfill_png16: test.py:133-148 -- depth floor (if use_floor), clip to [lo, hi] (0, 100), pad_top (96) copies
 * of the first row on top, * scale (256), cast to uint16.  out: uint16 [B, pad_top+H, W].
 */
int dtfill_crop_floor(const float *x, int B, int H, int W, int r0, int r1, int c0, int c1, int use_floor, float floor_,
                      float *out, void *stream);
int dtfill_png16(const float *x, int B, int H, int W, int pad_top, int use_floor, float floor_, float lo, float hi,
                 float scale, uint16_t *out, void *stream);

/*
 * Error metrics of evaluation.py, one row per frame:
 *   DTFILL_METRICS_KITTI  Result.evaluate, evaluation.py:82-123 (metres -> mm for mse/rmse/mae, -> 1/km for
 *                         irmse/imae; the deltas stay 0 as in the reference);
 *   DTFILL_METRICS_NYU    Result_NYU.evaluate, evaluation.py:196-239 (no unit change, mae = mean(|d|/target),
 *                         delta1..3 = mean(max(o/t, t/o) < 1.25^k)).
 * Elements with output > 0.01 and target > 0.01 count.  Per element the float32 arithmetic of the numpy
 * expressions; the means accumulate in float64 in a fixed order (numpy: float32 pairwise sums), so a result is
 * reproducible and within ~1e-6 relative of numpy's.  No valid element: NaN (numpy's mean of nothing).
 * output, target: float32 [B, n] device pointers; out: float64 [B, DTFILL_METRICS_N] device pointer, columns
 * mse, rmse, mae, irmse, imae, delta1, delta2, delta3, count.  workspace: device scratch of at least
 * dtfill_metrics_workspace_bytes(B) bytes.
 */
#define DTFILL_METRICS_KITTI 0
#define DTFILL_METRICS_NYU 1
#define DTFILL_METRICS_N 9
size_t dtfill_metrics_workspace_bytes(int B);
int dtfill_metrics(const float *output, const float *target, int B, long long n, int kind, double *out, void *workspace,
                   size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DTFILL_H */
