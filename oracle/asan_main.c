/*
 * asan_main.c -- driver of the address/UB-sanitized HOST build of the oracle (make oracle_asan).
 *
 * TEST INFRASTRUCTURE ONLY (tests/test_oracle.py::test_oracle_sanitized).  The GPU pool offers no sanitizer for device
 * code, so the checker itself is the piece that can be run under one: every entry point of dtfill_oracle.c on seeded
 * random frames of awkward shapes (1 x 1, a single row, a single column, widths around the 5x5 mask's two-cell border),
 * without sources, with sources everywhere, with every optional output dropped in turn, and with thresholds that
 * misalign the value list (the IndexError path).  Prints one checksum line; a sanitizer report aborts with a non-zero
 * exit code.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void cvdt_l1_labels(const uint8_t *mask, int H, int W, float *dist, int32_t *labels);
void oracle_nearest_point(const float *x, int H, int W, float src_thr, float *dt, int32_t *lbl);
int oracle_fill_frame(const float *x, int H, int W, float src_thr, float val_thr, float *out_depth, float *out_dt,
                      int32_t *out_lbl);
int oracle_fill_batch(const float *x, int B, int H, int W, float src_thr, float val_thr, float *out_depth, float *out_dt,
                      int32_t *out_lbl, int32_t *status);
int oracle_fill_batch_l2(const float *x, int B, int H, int W, float src_thr, float val_thr, float *out_depth, float *out_dt,
                         int32_t *out_idx, int32_t *status);
void brute_nearest(const uint8_t *mask, int H, int W, int metric, int32_t *dist2, int32_t *near);
void edt_l2_labels(const uint8_t *mask, int H, int W, int32_t *dist2, int32_t *near);

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd(void)
{
    rng_state ^= rng_state << 13;
    rng_state ^= rng_state >> 7;
    rng_state ^= rng_state << 17;
    return (uint32_t)(rng_state >> 32);
}
static double unit(void) { return rnd() / 4294967296.0; }

static uint64_t sum;
static void mix(const void *p, size_t n)
{
    const unsigned char *c = (const unsigned char *)p;
    size_t i;
    for (i = 0; i < n; i++) sum = (sum ^ c[i]) * 1099511628211ull;
}

static void one_shape(int B, int H, int W, double density, float lo, float hi, float src_thr, float val_thr)
{
    size_t n = (size_t)H * W, p;
    float *x = (float *)malloc(sizeof(float) * n * B);
    float *depth = (float *)malloc(sizeof(float) * n * B), *dt = (float *)malloc(sizeof(float) * n * B);
    int32_t *lbl = (int32_t *)malloc(sizeof(int32_t) * n * B), *status = (int32_t *)malloc(sizeof(int32_t) * B);
    uint8_t *mask = (uint8_t *)malloc(n);
    int32_t *d2 = (int32_t *)malloc(sizeof(int32_t) * n), *near = (int32_t *)malloc(sizeof(int32_t) * n);
    int drop;
    for (p = 0; p < n * B; p++) x[p] = unit() < density ? (float)(lo + (hi - lo) * unit()) : 0.0f;
    for (p = 0; p < n; p++) mask[p] = x[p] >= 0.9f ? 0 : 1;
    cvdt_l1_labels(mask, H, W, dt, lbl);
    mix(dt, sizeof(float) * n);
    mix(lbl, sizeof(int32_t) * n);
    oracle_nearest_point(x, H, W, src_thr, dt, lbl);
    mix(lbl, sizeof(int32_t) * n);
    for (drop = 0; drop < 4; drop++) { /* every optional output dropped in turn */
        int rc = oracle_fill_frame(x, H, W, src_thr, val_thr, drop == 1 ? NULL : depth, drop == 2 ? NULL : dt,
                                   drop == 3 ? NULL : lbl);
        mix(&rc, sizeof rc);
        if (rc == 0 && drop != 1) mix(depth, sizeof(float) * n);
    }
    oracle_fill_batch(x, B, H, W, src_thr, val_thr, depth, dt, lbl, status);
    mix(status, sizeof(int32_t) * B);
    mix(dt, sizeof(float) * n * B);
    oracle_fill_batch_l2(x, B, H, W, src_thr, val_thr, depth, dt, lbl, status);
    mix(status, sizeof(int32_t) * B);
    mix(lbl, sizeof(int32_t) * n * B);
    edt_l2_labels(mask, H, W, d2, near);
    mix(d2, sizeof(int32_t) * n);
    if (n <= 4096) {
        brute_nearest(mask, H, W, 1, d2, near);
        mix(near, sizeof(int32_t) * n);
        brute_nearest(mask, H, W, 2, d2, near);
        mix(near, sizeof(int32_t) * n);
    }
    free(x); free(depth); free(dt); free(lbl); free(status); free(mask); free(d2); free(near);
}

int main(void)
{
    static const int shapes[][2] = {{1, 1}, {1, 2}, {2, 1}, {1, 37}, {41, 1}, {2, 2}, {3, 3}, {4, 5}, {5, 4},
                                    {5, 7}, {17, 33}, {64, 64}, {37, 130}, {96, 320}};
    static const double dens[] = {0.0, 0.003, 0.05, 0.5, 1.0};
    size_t s, d;
    for (s = 0; s < sizeof shapes / sizeof shapes[0]; s++)
        for (d = 0; d < sizeof dens / sizeof dens[0]; d++) {
            one_shape(2, shapes[s][0], shapes[s][1], dens[d], 1.0f, 80.0f, 0.1f, 0.1f);
            /* values inside (val_thr, 1 - src_thr): the two enumerations misalign, some frames raise IndexError */
            one_shape(2, shapes[s][0], shapes[s][1], dens[d], 0.2f, 3.0f, 0.001f, 0.1f);
            one_shape(1, shapes[s][0], shapes[s][1], dens[d], 1.0f, 80.0f, 0.1f, 30.0f);
        }
    printf("oracle_asan ok checksum %016llx\n", (unsigned long long)sum);
    return 0;
}
