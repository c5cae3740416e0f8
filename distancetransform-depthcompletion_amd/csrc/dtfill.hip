// dtfill.hip -- MI355X (gfx950) distance-transform + nearest-valid-depth fill.
//
// Replaces, for B frames at once, the reference's per-frame
//   nearest_point()      solution_DeepNet/tools.py:7-10   (cv2.distanceTransformWithLabels, L1, 5x5, LABEL_PIXEL)
//   DT_complete_batch()  solution_DeepNet/tools.py:13-35  (value-list compaction + depth_list[lbl-1] gather)
//   Distance_Transform() solution_DeepNet/eval_NYU.py:120-133
//
// The reference's arithmetic is OpenCV's two-pass 5x5 chamfer: two raster sweeps that are sequential
// in both image axes.  This file does NOT sweep.  It uses the following identity (derived in
// DESIGN.md section 2, checked bit-for-bit against the sequential restatement in oracle/):
//
//   d(q)      = exact L1 distance to the nearest source               (weights {1,2,3} == L1 norms of the taps)
//   live(q)   = "the forward sweep already reached the final value at q"
//             = some nearest source s of q lies in q's forward cone:  s above-or-left of q, or above-right
//               with (s.col - q.col) <= 2 (q.row - s.row)
//   parent(q) = live(q) ? first forward tap r (cv2 order) with live(r) and d(r)+w == d(q)
//                       : first backward tap r (cv2 order) with d(r)+w == d(q)
//   label(q)  = label(root of the parent chain) = 1 + raster rank of that source.
//
// Every hop lowers d by exactly the hop's L1 length, so a chain ends on a NEAREST source of its start pixel, and
// everything that decides label(q) lies inside the L1 ball of radius d(q) around q.
//
// Kernels of one l1_cv pass (seven launches):
//   k_mask     source / value bit words + per-row prefix popcounts                    reads x once
//   k_frame    per-frame row-count scan -> compaction ranks; frame facts; the row structure (first source row, rows too far
//              from every source row) and with it WHO takes which rows of the frame; value list when the source and value masks
//              of a frame differ; the source list of a frame that holds a handful
//   k_fused    dense frames (halo 16 or 32 per frame): one workgroup per (tile + halo) window, bit-sliced: the
//              level-synchronous form of the identity on bit planes held in registers, byte codes un-sliced into LDS,
//              lock-step chain walk, rank lookup, depth gather and the three output stores.  Hands a ROW on when one of its
//              pixels is farther than the halo from every source.
//   every other frame, and the handed-on rows (any distance, any density; dtfill_rows.hpp):
//   k_colT     per 32-row band and column: the band's source bits and the distances to the nearest source
//              above / below the band -- everything a row needs to know about its columns; and the label of
//              every source pixel, where k_fin can gather it
//              + k_sky's blocks (dtfill_sky.hpp) in the same launch: the rows above the first source row (the empty sky of a
//              LiDAR frame) in closed form from the two rows beneath
//   k_rows     packed-key min-plus row scans: d, the nearest source (smallest and largest column that reach d:
//              a pixel with ONE nearest source needs no chain), live; distance map + bit planes
//   k_fin      per tile: 5x5 parent rule bit-sliced for the remaining "tie" pixels, their chains (up to four hops through
//              the step bytes in LDS), label + depth of every pixel
//              + the tiles of the frames with a handful of sources ("k_pts", dtfill_pts.hpp) in the same launch: candidates
//              per wave box, dominance pruning, packed-key minima per pixel by one sweep down and one up, the same rule for
//              their tie pixels -- from the source list to the three outputs
//   k_tiesx    the few tie pixels whose chain crosses tiles or runs longer: follows the recorded links
// l2 pass (exact Euclidean, canonical tie-break; dtfill_l2.hpp): k_mask, k_frame, then
//   k_l2win<R> dense frames: (2R+1)^2 windows straight from the bit words, packed-key minimum over the window rows
//   k_l2far    the odd far pixel of a dense frame, half a wave each
//   k_colT     vertical distances per column, for sparse frames and for rows of far pixels (the empty sky)
//   k_l2env    sparse frames: lower envelope of parabolas searched by monotone bisection; rows of far pixels of a dense frame
//              (the sky): a window in the column distances; frames with a handful of sources: tiles over the source list
//
// No MFMA anywhere: this path is compare/min/index work (DESIGN.md "Roofline").

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include "../../include/dtfill.h"

typedef uint16_t u16;
typedef uint8_t u8;
typedef unsigned long long u64;
typedef uint32_t u32;

namespace {

#include "dtfill_common.hpp"
#include "dtfill_prepass.hpp"
#include "dtfill_fused.hpp"
#include "dtfill_rows.hpp"
#include "dtfill_sky.hpp"
#include "dtfill_pts.hpp"
#include "dtfill_l2.hpp"
#include "dtfill_outlier.hpp"
#include "dtfill_gmc.hpp"
#include "dtfill_post.hpp"

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Carve {
    u8 *planes;          // k_rows -> k_fin: bit planes d & 1, d >> 1 & 1, d >> 2 & 1, live, tie; k_fin -> k_tiesx: unresolved; Wd * 8 bytes per row
    size_t plane_bytes;  // bytes of one plane
    u64 *srcbits, *valbits;
    u16 *wpre_s, *wpre_v;
    u32 *rowcnt_s, *rowcnt_v, *rowbase_s, *rowbase_v;
    u32 *rowfar;         // per row: l1_cv: k_fused left a pixel undecided (-> k_rows, k_fin, k_tiesx); l2: far pixels (k_l2win -> k_l2far, k_l2env)
    uint2 *ct;           // k_colT -> k_rows: per 32-row band and column {the band's source bits of the column, distances
                         // from the band's first / last row to the nearest source above / below}; rows of ctp columns
    int nb, ctp;         // bands per frame, columns per row of ct
    u32 *xlist, *xptr;   // k_fin -> k_tiesx: the pixels whose chain left their tile, and where each goes on
    float *dscratch;     // k_fin -> k_tiesx: depths of the rows a depth epilogue drops from the output
    u32 *spix;           // k_rows -> k_fin: per pixel, the frame offset of its nearest source in column kmin
    uint4 *rec;          // k_colT -> k_fin, k_l2env: rank records, per 64-pixel word of a row (label_from_rec)
    int *finfo, *fflag2, *route, *status, *negflag;
    float *vlist;
    PtsSrc *ptslist;     // k_frame -> k_pts: the sources of a frame that has a handful (l1_cv, ROUTE_POINTS)
    size_t total;
};

Carve carve(void *ws, int B, int H, int W) {
    const size_t N = (size_t)B * H * W;
    const size_t Wd = (size_t)(W + 63) / 64;
    const size_t NW = (size_t)B * H * Wd;
    const size_t NR = (size_t)B * H;
    char *p = static_cast<char *>(ws);
    size_t off = 0;
    Carve c;
    auto take = [&](size_t bytes) {
        char *r = p ? p + off : nullptr;
        off += align256(bytes);
        return r;
    };
    c.srcbits = (u64 *)take(NW * 8);
    c.valbits = (u64 *)take(NW * 8);
    c.wpre_s = (u16 *)take(NW * 2);
    c.wpre_v = (u16 *)take(NW * 2);
    c.rowcnt_s = (u32 *)take(NR * 4);
    c.rowcnt_v = (u32 *)take(NR * 4);
    c.rowbase_s = (u32 *)take(NR * 4);
    c.rowbase_v = (u32 *)take(NR * 4);
    c.finfo = (int *)take((size_t)B * FI_STRIDE * 4);
    c.fflag2 = (int *)take((size_t)B * 4);
    c.rowfar = (u32 *)take(NR * 4);  // right behind fflag2: rowflag_of()
    c.route = (int *)take((size_t)B * 4);
    c.status = (int *)take((size_t)B * 4);
    c.negflag = (int *)take((size_t)B * 4);
    // any-distance path (touched only for frames the fused kernel does not take)
    c.nb = (H + 31) / 32;
    c.ctp = ct_pitch(W);
    c.ct = (uint2 *)take((size_t)B * c.nb * c.ctp * sizeof(uint2));
    c.rec = (uint4 *)take(NW * sizeof(uint4));
    c.spix = (u32 *)take(N * 4);
    c.dscratch = (float *)take(N * 4);
    c.xlist = (u32 *)take(N * 4);
    c.xptr = (u32 *)take(N * 4);
    c.plane_bytes = align256(NW * 8);
    c.planes = (u8 *)take(PL_N * c.plane_bytes);
    c.vlist = (float *)take(N * 4);
    c.ptslist = (PtsSrc *)take((size_t)B * L2_PTS_MAX * sizeof(PtsSrc));
    c.total = off;
    return c;
}

bool shape_ok(int B, int H, int W) {
    return B >= 1 && B <= 65535 && H >= 1 && W >= 1 && (long long)H + W - 2 < MAX_HW_SUM &&
           (long long)B * H * W < (1ll << 31);  // B is a grid dimension; pixel indices are 32-bit
}

constexpr int NK_L1 = 7;
const char *const kNamesL1[NK_L1] = {"k_mask", "k_frame", "k_fused", "k_colT", "k_rows", "k_fin", "k_tiesx"};

// k_mask4 when the rows can be read 16 bytes at a time, k_mask otherwise (same outputs); with DTFILL_FLAG_OUTLIER_REMOVAL the
// predicates see outlier_removal(x) (k_mask_o, then its exhaustive variant for the frames that hold a negative value)
void launch_mask(const float *x, int B, int H, int W, int Wd, float src_thr, float val_thr, const Carve &c, unsigned flags,
                 hipStream_t st) {
    const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    const dim3 g4((H + 3) / 4, B), g4m(B, (H + 3) / 4);  // (k_mask4: frames along x)
    // negflag ("this frame holds a negative value", raised by the first outlier launch, read by the second) starts clear whatever
    // the workspace held before
    if (flags & DTFILL_FLAG_OUTLIER_REMOVAL) (void)hipMemsetAsync(c.negflag, 0, (size_t)B * sizeof(int), st);
    if ((flags & DTFILL_FLAG_OUTLIER_REMOVAL) && vec) {
        k_mask4<1, 1><<<g4m, 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits, c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v, c.negflag);
        k_mask4<2, 1><<<g4m, 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits, c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v, c.negflag);
    } else if (flags & DTFILL_FLAG_OUTLIER_REMOVAL) {
        k_mask_o<false><<<g4, 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits, c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v,
                                            c.negflag);
        k_mask_o<true><<<g4, 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits, c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v,
                                           c.negflag);
    }
    else if (vec)
        k_mask4<0, 1><<<g4m, 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits, c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v,
                                          c.negflag);
    else
        k_mask<<<dim3((H + 4 * M_RPW - 1) / (4 * M_RPW), B), 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits,
                                                                          c.valbits, c.wpre_s, c.wpre_v, c.rowcnt_s,
                                                                          c.rowcnt_v);
}

int run_l1(const float *x, int B, int H, int W, float src_thr, float val_thr, float *out_depth,
           float *out_dt, int32_t *out_index, int32_t *frame_status, void *workspace, unsigned flags,
           hipStream_t st, hipEvent_t *ev, DepthEpilogue ep = DepthEpilogue{0, 0, 0.0f}) {
    const Carve c = carve(workspace, B, H, W);
    const int Wd = (W + 63) / 64;
    int *status = frame_status ? frame_status : c.status;
    const bool general_only = flags & DTFILL_FLAG_GENERAL_ONLY;
    const bool fused_only = flags & DTFILL_FLAG_FUSED_ONLY;
    int k = 0;
    bool ok = true;
    auto mark = [&]() {
        ok = ok && hipGetLastError() == hipSuccess;  // after every launch
        if (ev) (void)hipEventRecord(ev[k++], st);
    };
    mark();
    launch_mask(x, B, H, W, Wd, src_thr, val_thr, c, flags, st);
    mark();
    // geometry of the window kernel's two tilings (k_frame pre-marks whole tile rows)
    auto tiling = [&](int R) {
            const int THM = F_WHM - 2 * R, TWM = F_WWM - 2 * R;
            const int nty = (H + THM - 1) / THM, ntx = (W + TWM - 1) / TWM;
            FusedTiles t;
            t.TH = (H + nty - 1) / nty;  // even split
            t.TW = (W + ntx - 1) / ntx;
            // columns in whole 128-byte lines when the window allows it: the runs of a tile row a wave stores are then whole lines
            if (((t.TW + 31) & ~31) <= TWM) t.TW = (t.TW + 31) & ~31;
            t.tiles_x = ntx;
            t.nty = nty;
            t.ntiles = ntx * nty;
            return t;
        };
    const FusedTiles t16 = tiling(16), t32 = tiling(32);
    const bool epi = ep.row0 != 0 || ep.use_floor;
    // row flags (k_frame): the sky above the first source row goes to k_sky, rows too far from every source row to the
    // any-distance kernels, the rest of the frame to the window kernel.  Not with a depth epilogue (a handed-on row costs the
    // whole frame there) and not on the forced paths of the tests.
    const bool rowflags = !epi && !fused_only && !general_only;
    // k_sky takes the distances of its two base rows from the distance map: without one from the caller, the scratch frame
    float *const out_dt_caller = out_dt;
    if (rowflags && !out_dt) out_dt = c.dscratch;
    k_frame<<<B, 256, 0, st>>>(x, c.valbits, c.wpre_v, c.srcbits, c.wpre_s, c.ptslist, c.rowcnt_s, c.rowcnt_v, H, W, Wd, c.rowbase_s, c.rowbase_v,
                               c.finfo, c.vlist, c.fflag2, c.route, status, (general_only ? 1 : 0) | (rowflags ? 4 | 8 : 0), c.negflag, c.rowfar,
                               t16.nty, t32.nty);
    mark();
    if (!general_only) {
        // dense frames: the window kernel, halo 16 or 32 per frame (k_frame's route).  It hands rows on (fflag2, rowflag) when a
        // tile pixel turns out to be farther than the halo from every source.
        // streaming stores only where every run of tile pixels a wave stores is whole 128-byte lines
        auto line = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 127) == 0; };
        const bool stream = (W & 31) == 0 && (t16.TW & 31) == 0 && (t32.TW & 31) == 0 && line(out_depth) && line(out_dt) && line(out_index);
        const dim3 fg(B, max(t16.ntiles, t32.ntiles));  // frames along x: a frame's tiles beyond its own tiling (they exit) come last
        if (stream)
            k_fused<true><<<fg, F_NT, 0, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, H, W, Wd, t16, t32, out_depth, out_dt,
                                               out_index, c.route, c.fflag2, status, ep);
        else
            k_fused<false><<<fg, F_NT, 0, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, H, W, Wd, t16, t32, out_depth, out_dt,
                                                out_index, c.route, c.fflag2, status, ep);
    }
    mark();
    if (!fused_only) {
        // every other frame: argmin scans, any distance (dtfill_rows.hpp)
        const int nb = c.nb;
        int cw = min(16, max(2, (nb + 3) / 4));  // two iterations of two bands per wave: fewer, longer waves fit the CUs in one round
        // k_sky's blocks (the rows above the first source row, dtfill_sky.hpp) ride behind the column blocks
        SkyArgs sky = {c.finfo, out_dt, out_dt_caller, out_depth, out_index, (W + SKY_SW - 1) / SKY_SW, 0};
        size_t lds = colT_lds(nb);
        const bool sky_rides = rowflags && cw <= SKY_NT / 64;  // (taller frames: a launch of its own, below)
        if (sky_rides) {
            sky.nblocks = sky.nstrips * ((H + SKY_RG - 1) / SKY_RG);
            cw = SKY_NT / 64;
            lds = max(lds, sky_lds(sky_span_max(H, W)));
        }
        if (lds > 48 * 1024)  // (set per call: the attribute belongs to the current device)
            ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_colT), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        const int ncol = (c.ctp + 63) / 64;
        const int fx = cw <= 4;  // frames along grid x for frames of up to 512 rows (k_colT's comment)
        k_colT<<<fx ? dim3(B, ncol + sky.nblocks) : dim3(ncol + sky.nblocks, B), 64 * cw, lds, st>>>(
            c.srcbits, c.fflag2, H, W, Wd, nb, c.ctp, c.ct, c.wpre_s, c.rowbase_s, (out_depth || out_index) ? c.rec : nullptr, ncol, sky, fx);
        if (rowflags && !sky_rides)
            k_sky<<<dim3(sky.nstrips * ((H + SKY_RG - 1) / SKY_RG), B), SKY_NT, sky_lds(sky_span_max(H, W)), st>>>(
                c.finfo, H, W, out_dt, out_dt_caller, out_depth, out_index, sky.nstrips);
        mark();
        const int Wp = Wd * 8;
        // columns per lane: 8 or 10, whichever leaves fewer idle lanes in the row's last wave
        const int nw8 = (W + 511) / 512, nw10 = (W + 639) / 640;
        const bool ten = nw10 * 640 < nw8 * 512;
        auto aligned = [](const void *p, uintptr_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; };
        const int nwv = ten ? nw10 : nw8;
        const bool fin = out_depth || out_index;  // the distance map alone needs neither sources nor tie-breaks
        float *dt = out_dt;
        // bit 0: a lane's pixels can leave as vectors; bit 1: the distance map goes through LDS and leaves as whole lines
        const int ovec = (ten ? ((W & 1) == 0 && aligned(dt, 8)) : ((W & 3) == 0 && aligned(dt, 16))) |
                         (((W & 31) == 0 && dt && aligned(dt, 128)) ? 2 : 0);
        const size_t rlds = (ovec & 2) ? (size_t)W * sizeof(float) : 0;
        // rows per block: several where a row is one wave, or where there are very many rows (the exit of a frame that has none, or of
        // a row far from every handed-on row, costs a block dispatch per row)
        const int rpb = (nwv == 1 || (long long)H * B >= 16384) && nwv <= 4 ? R_RPB : 1;
        const dim3 grid((H + rpb - 1) / rpb, B);
#define LAUNCH_ROWS(PPL_, MAXT_, MULTI_)                                                                                        \
    k_rows<PPL_, MAXT_, MULTI_><<<grid, 64 * nwv, rlds, st>>>(c.ct, c.ctp, c.fflag2, H, W, nb, Wp, c.planes, c.plane_bytes, dt, \
                                                           fin ? c.spix : nullptr, ovec, c.rowfar, c.finfo, rpb)
        if (rpb > 1 && ten)
            LAUNCH_ROWS(10, 256, true);
        else if (rpb > 1)
            LAUNCH_ROWS(8, 256, true);
        else if (ten && nwv <= 4)
            LAUNCH_ROWS(10, 256, false);
        else if (ten)
            LAUNCH_ROWS(10, 1024, false);
        else if (nwv <= 4)
            LAUNCH_ROWS(8, 256, false);
        else
            LAUNCH_ROWS(8, 1024, false);
#undef LAUNCH_ROWS
        mark();
        // the frames with a handful of sources (k_frame: ROUTE_POINTS, fflag 3; only with the row flags) ride in k_fin's launch:
        // their tiles, from the source list to the outputs (dtfill_pts.hpp); k_tiesx finishes the chains that leave a tile
        PtsArgs pa;
        pa.ptslist = c.ptslist;
        pa.out_dt = out_dt;
        // 32 x 256 tiles or 64 x 128: whichever wastes fewer waves on this shape (640 columns are 2.5 tiles of 256 but 5 of 128)
        const int nwide = ((W + 255) / 256) * ((H + 31) / 32), ntall = ((W + 127) / 128) * ((H + 63) / 64);
        pa.tall = ntall < nwide;
        pa.tiles_x = pa.tall ? (W + 127) / 128 : (W + 255) / 256;
        pa.ntiles = rowflags ? (pa.tall ? ntall : nwide) : 0;
        const int ttx = (W + Q_TW - 1) / Q_TW, tty = (H + Q_TH - 1) / Q_TH;
        pa.fin_ntiles = fin ? ttx * tty : 0;
        if (fin || pa.ntiles) {
            const int vec = (W & 3) == 0 && aligned(out_depth, 16) && aligned(out_index, 16);
            k_fin<<<dim3(max(pa.fin_ntiles, pa.ntiles), B), Q_NT, 0, st>>>(c.planes, c.plane_bytes, Wp, c.fflag2, H, W, Wd, ttx, c.spix, x, c.rec,
                                                       c.vlist, out_depth, out_index, status, c.finfo,
                                                       c.xlist, c.xptr, c.planes + PL_UNRES * c.plane_bytes, vec, ep, c.dscratch, c.rowfar, pa);
            mark();
            if (fin)
                k_tiesx<<<dim3(XL_BLOCKS, B), 256, 0, st>>>(c.planes + PL_UNRES * c.plane_bytes, Wp, c.fflag2, c.finfo, c.xlist,
                                                            c.xptr, H, W, out_depth, out_index, ep, c.dscratch, c.rowfar);
            mark();
        } else {
            mark();
            mark();
        }
    } else {
        for (int t = 0; t < 4; ++t) mark();
    }
    return ok ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

constexpr int NK_L2 = 7;
const char *const kNamesL2[NK_L2] = {"k_mask", "k_frame", "k_l2win<10>", "k_l2win<15>", "k_l2far", "k_colT", "k_l2env"};

int run_l2(const float *x, int B, int H, int W, float src_thr, float val_thr, float *out_depth, float *out_dt,
           int32_t *out_index, int32_t *frame_status, void *workspace, unsigned flags, hipStream_t st, hipEvent_t *ev) {
    const Carve c = carve(workspace, B, H, W);
    const int Wd = (W + 63) / 64;
    int *status = frame_status ? frame_status : c.status;
    const bool general_only = flags & DTFILL_FLAG_GENERAL_ONLY;
    int k = 0;
    bool ok = true;
    auto mark = [&]() {
        ok = ok && hipGetLastError() == hipSuccess;  // after every launch
        if (ev) (void)hipEventRecord(ev[k++], st);
    };
    mark();
    launch_mask(x, B, H, W, Wd, src_thr, val_thr, c, flags, st);
    mark();
    k_frame<<<B, 256, 0, st>>>(x, c.valbits, c.wpre_v, c.srcbits, c.wpre_s, c.ptslist, c.rowcnt_s, c.rowcnt_v, H, W, Wd, c.rowbase_s, c.rowbase_v,
                               c.finfo, c.vlist, c.fflag2, c.route, status, (general_only ? 1 : 0) | 2, c.negflag, c.rowfar, 0, 0);
    mark();
    // dense frames (k_frame's route 16 / 32): windows of 15 x 15 / 31 x 31 around every pixel, the few pixels with no source
    // that near one by one
    const int ttx = (W + W2_TW - 1) / W2_TW, tty = (H + W2_TH - 1) / W2_TH;
    k_l2win<W2_R16><<<dim3(ttx * tty, B), 256, L2Win<W2_R16>::LDS, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, c.xlist,
                                                                         c.route, 16, c.rowfar, c.fflag2, H, W, Wd, ttx, out_depth, out_dt, out_index, status);
    mark();
    k_l2win<W2_R32><<<dim3(ttx * tty, B), 256, L2Win<W2_R32>::LDS, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, c.xlist,
                                                                         c.route, 32, c.rowfar, c.fflag2, H, W, Wd, ttx, out_depth, out_dt, out_index, status);
    mark();
    // one wave per listed pixel: enough blocks per frame for ~16 k waves in the batch
    k_l2far<<<dim3(min(256, max(16, 4096 / B)), B), 256, 0, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, c.xlist, c.route,
                                                                  c.rowfar, H, W, Wd, out_depth, out_dt, out_index, status);
    mark();
    // vertical distances per column (k_colT) for the frames that need them: route 0, or a row handed on by k_l2win
    {
        const int cw = min(16, max(2, (c.nb + 3) / 4));
        const size_t lds = colT_lds(c.nb);
        if (lds > 48 * 1024)
            ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_colT), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
        const int fx = cw <= 4, ncb = (c.ctp + 63) / 64;  // (frames along grid x for frames of up to 512 rows: k_colT's comment)
        k_colT<<<fx ? dim3(B, ncb) : dim3(ncb, B), 64 * cw, lds, st>>>(c.srcbits, c.fflag2, H, W, Wd, c.nb, c.ctp, c.ct, c.wpre_s, c.rowbase_s,
                                                                        (out_depth || out_index) ? c.rec : nullptr, ncb, SkyArgs{}, fx);
    }
    mark();
    {
        // the rows, one wave each; then the 32 x 32 tiles of the frames with a handful of sources, one wave each
        const size_t wave_lds = max(max(l2env_lds(W), (size_t)L2_PTS_MAX * 8), (size_t)(W + 2 * L2S_R) * 8);
        const int ntile = ((H + PT_T - 1) / PT_T) * ((W + PT_T - 1) / PT_T);
        if (4 * wave_lds <= 64 * 1024) {
            const int nrowblk = (H + 3) / 4;
            k_l2env<4><<<dim3(B, nrowblk + (ntile + 3) / 4), 256, 4 * wave_lds, st>>>(x, c.ct, c.ctp, c.nb, c.rec, Wd, c.finfo, c.vlist, c.route,
                                                                                  c.rowfar, c.xlist, H, W, nrowblk, wave_lds, 1, out_depth,
                                                                                  out_dt, out_index, status);
        } else {
            if (wave_lds > 48 * 1024)  // rows wider than ~4900 pixels (set per call: the attribute belongs to the current device)
                ok = ok && hipFuncSetAttribute(reinterpret_cast<const void *>(k_l2env<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)l2env_lds(8192)) == hipSuccess;
            k_l2env<1><<<dim3(B, H + (ntile + 7) / 8), 64, wave_lds, st>>>(x, c.ct, c.ctp, c.nb, c.rec, Wd, c.finfo, c.vlist, c.route, c.rowfar, c.xlist,
                                                               H, W, H, wave_lds, 8, out_depth, out_dt, out_index, status);
        }
    }
    mark();
    return ok ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

int check_args(const float *x, int B, int H, int W, int metric, float *out_depth, float *out_dt,
               int32_t *out_index, void *workspace, size_t ws_bytes, unsigned flags = 0u) {
    if (!x || !workspace || (!out_depth && !out_dt && !out_index)) return DTFILL_ERR_NULL;
    if (!shape_ok(B, H, W)) return DTFILL_ERR_SHAPE;
    if ((flags & DTFILL_FLAG_OUTLIER_REMOVAL) && (H < 4 || W < 4)) return DTFILL_ERR_SHAPE;  // the 7x7 filter's reflected border
    if (metric != DTFILL_METRIC_L1_CV && metric != DTFILL_METRIC_L2) return DTFILL_ERR_METRIC;
    if (ws_bytes < dtfill_workspace_bytes(B, H, W, metric) || ((uintptr_t)workspace & 255))
        return DTFILL_ERR_WORKSPACE;
    return DTFILL_OK;
}

}  // namespace

extern "C" {

int dtfill_abi_version(void) { return DTFILL_ABI_VERSION; }

const char *dtfill_strerror(int code) {
    switch (code) {
        case DTFILL_OK: return "ok";
        case DTFILL_ERR_NULL: return "null input, workspace, or no output requested";
        case DTFILL_ERR_SHAPE: return "bad shape: need 1 <= B <= 65535, H,W >= 1, H+W-2 < 8192 and B*H*W < 2^31";
        case DTFILL_ERR_WORKSPACE: return "workspace too small or not 256-byte aligned";
        case DTFILL_ERR_METRIC: return "unknown metric";
        case DTFILL_ERR_LAUNCH: return "HIP kernel launch failed";
        case DTFILL_ERR_NO_DEVICE: return "no usable HIP device";
        default: return "unknown dtfill error code";
    }
}

size_t dtfill_workspace_bytes(int B, int H, int W, int metric) {
    if (!shape_ok(B, H, W)) return 0;
    if (metric != DTFILL_METRIC_L1_CV && metric != DTFILL_METRIC_L2) return 0;
    return carve(nullptr, B, H, W).total;
}

int dtfill_batch_flags(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                       float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                       void *workspace, size_t ws_bytes, void *stream, unsigned flags) {
    int rc = check_args(x, B, H, W, metric, out_depth, out_dt, out_index, workspace, ws_bytes, flags);
    if (rc != DTFILL_OK) return rc;
    if (metric == DTFILL_METRIC_L2)
        return run_l2(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, flags,
                      static_cast<hipStream_t>(stream), nullptr);
    return run_l1(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, flags,
                  static_cast<hipStream_t>(stream), nullptr);
}

int dtfill_batch(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                 float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                 void *workspace, size_t ws_bytes, void *stream) {
    return dtfill_batch_flags(x, B, H, W, src_thr, val_thr, metric, out_depth, out_dt, out_index,
                              frame_status, workspace, ws_bytes, stream, 0u);
}

int dtfill_batch_epilogue(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                          float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                          void *workspace, size_t ws_bytes, void *stream, unsigned flags, int depth_row0, int use_floor,
                          float floor_) {
    int rc = check_args(x, B, H, W, metric, out_depth, out_dt, out_index, workspace, ws_bytes, flags);
    if (rc != DTFILL_OK) return rc;
    if (depth_row0 < 0 || depth_row0 >= H) return DTFILL_ERR_SHAPE;
    if (metric != DTFILL_METRIC_L1_CV) return (depth_row0 || use_floor) ? DTFILL_ERR_METRIC : dtfill_batch_flags(x, B, H, W, src_thr, val_thr, metric, out_depth, out_dt, out_index, frame_status, workspace, ws_bytes, stream, flags);
    return run_l1(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, flags,
                  static_cast<hipStream_t>(stream), nullptr, DepthEpilogue{depth_row0, use_floor ? 1 : 0, floor_});
}

int dtfill_outlier_removal(const float *x, int B, int H, int W, float *out, void *stream) {
    if (!x || !out) return DTFILL_ERR_NULL;
    if (B < 1 || H < 4 || W < 4 || (long long)B * H * W >= (1ll << 31) || B > 65535) return DTFILL_ERR_SHAPE;
    k_outlier<<<dim3((W + O_TW - 1) / O_TW, (H + O_TH - 1) / O_TH, B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        x, H, W, out);
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

int dtfill_generate_multi_channel(const float *data, const float *mask, int B, int H, int W, int table_size,
                                  int scale_num, float *out2, float *out3, float *out4, void *stream) {
    if (!data || !mask) return DTFILL_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || (long long)B * H * W >= (1ll << 31) || B > 65535) return DTFILL_ERR_SHAPE;
    if (table_size < 1 || (table_size & 1) == 0 || (table_size - 1) / 2 > GM_MAXHALF || scale_num < 1 || scale_num > 4)
        return DTFILL_ERR_SHAPE;
    float *outs[3] = {out2, out3, out4};
    for (int k = 0; k < scale_num - 1; ++k)
        if (!outs[k]) return DTFILL_ERR_NULL;
    const int half = (table_size - 1) / 2;
    const size_t lds = (size_t)2 * (GM_TH + 2 * half) * (GM_TW + 2 * half) * sizeof(float);
    const dim3 grid((W + GM_TW - 1) / GM_TW, (H + GM_TH - 1) / GM_TH, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const float *src = data, *msk = mask;
    for (int k = 0; k < scale_num - 1; ++k) {
        // (table size 7: a step also writes the later steps' values of the pixels that will pass their masks)
        float *n1 = k + 1 < scale_num - 1 ? outs[k + 1] : nullptr, *n2 = k + 2 < scale_num - 1 ? outs[k + 2] : nullptr;
        if (table_size == 7 && msk)
            k_gmc7<false><<<grid, 256, 0, st>>>(src, msk, H, W, outs[k], n1, n2);
        else if (table_size == 7)
            k_gmc7<true><<<grid, 256, 0, st>>>(src, nullptr, H, W, outs[k], n1, n2);
        else
            k_gmc<<<grid, 256, lds, st>>>(src, msk, H, W, table_size, outs[k]);
        src = outs[k];
        msk = nullptr;  // the next step's mask is (previous output > 0.001)
    }
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

int dtfill_num_kernels(int metric) {
    return metric == DTFILL_METRIC_L1_CV ? NK_L1 : metric == DTFILL_METRIC_L2 ? NK_L2 : 0;
}

const char *dtfill_kernel_name(int metric, int k) {
    if (metric == DTFILL_METRIC_L1_CV && k >= 0 && k < NK_L1) return kNamesL1[k];
    if (metric == DTFILL_METRIC_L2 && k >= 0 && k < NK_L2) return kNamesL2[k];
    return "";
}

int dtfill_batch_timed(const float *x, int B, int H, int W, float src_thr, float val_thr, int metric,
                       float *out_depth, float *out_dt, int32_t *out_index, int32_t *frame_status,
                       void *workspace, size_t ws_bytes, void *stream, unsigned flags, float *kernel_ms) {
    int rc = check_args(x, B, H, W, metric, out_depth, out_dt, out_index, workspace, ws_bytes, flags);
    if (rc != DTFILL_OK) return rc;
    if (!kernel_ms) return DTFILL_ERR_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nk = dtfill_num_kernels(metric);
    static_assert(NK_L2 <= NK_L1, "event array");
    hipEvent_t ev[NK_L1 + 1];
    for (int k = 0; k <= nk; ++k)
        if (hipEventCreate(&ev[k]) != hipSuccess) {
            while (k-- > 0) (void)hipEventDestroy(ev[k]);  // nothing created so far is left behind
            return DTFILL_ERR_NO_DEVICE;
        }
    rc = metric == DTFILL_METRIC_L2
             ? run_l2(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, flags, st, ev)
             : run_l1(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, flags, st,
                      ev);
    (void)hipEventSynchronize(ev[nk]);
    for (int k = 0; k < nk; ++k) (void)hipEventElapsedTime(&kernel_ms[k], ev[k], ev[k + 1]);
    for (int k = 0; k <= nk; ++k) (void)hipEventDestroy(ev[k]);
    return rc;
}

int dtfill_pass_stats(const void *workspace, size_t ws_bytes, int B, int H, int W, int metric, long long *out_px, void *stream) {
    if (!workspace || !out_px) return DTFILL_ERR_NULL;
    if (!shape_ok(B, H, W)) return DTFILL_ERR_SHAPE;
    if (metric != DTFILL_METRIC_L1_CV && metric != DTFILL_METRIC_L2) return DTFILL_ERR_METRIC;
    if (ws_bytes < dtfill_workspace_bytes(B, H, W, metric) || ((uintptr_t)workspace & 255)) return DTFILL_ERR_WORKSPACE;
    const Carve c = carve(const_cast<void *>(workspace), B, H, W);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(out_px, 0, DTFILL_STATS_N * sizeof(long long), st) != hipSuccess) return DTFILL_ERR_LAUNCH;
    k_stats<<<B, 256, 0, st>>>(c.route, c.fflag2, c.rowfar, c.finfo, H, W, metric == DTFILL_METRIC_L2 ? 1 : 0, out_px);
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

int dtfill_crop_floor(const float *x, int B, int H, int W, int r0, int r1, int c0, int c1, int use_floor, float floor_,
                      float *out, void *stream) {
    if (!x || !out) return DTFILL_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || (long long)B * H * W >= (1ll << 31) || B > 65535 || H > 65535) return DTFILL_ERR_SHAPE;
    if (r0 < 0 || r1 > H || r0 >= r1 || c0 < 0 || c1 > W || c0 >= c1) return DTFILL_ERR_SHAPE;
    const int OH = r1 - r0, OW = c1 - c0;
    k_crop_floor<<<dim3(min((OW + 255) / 256, 8), OH, B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        x, H, W, r0, c0, OH, OW, use_floor, floor_, out);
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

int dtfill_png16(const float *x, int B, int H, int W, int pad_top, int use_floor, float floor_, float lo, float hi,
                 float scale, uint16_t *out, void *stream) {
    if (!x || !out) return DTFILL_ERR_NULL;
    if (B < 1 || H < 1 || W < 1 || pad_top < 0 || (long long)B * ((long long)H + pad_top) * W >= (1ll << 31) || B > 65535 ||
        H + pad_top > 65535)
        return DTFILL_ERR_SHAPE;
    k_png16<<<dim3(min((W + 255) / 256, 8), H + pad_top, B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        x, H, W, pad_top, use_floor, floor_, lo, hi, scale, out);
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

size_t dtfill_metrics_workspace_bytes(int B) {
    if (B < 1 || B > 65535) return 0;
    return (size_t)B * M_NB * M_NS * sizeof(double);
}

int dtfill_metrics(const float *output, const float *target, int B, long long n, int kind, double *out, void *workspace,
                   size_t ws_bytes, void *stream) {
    if (!output || !target || !out || !workspace) return DTFILL_ERR_NULL;
    if (B < 1 || B > 65535 || n < 1 || n >= (1ll << 40)) return DTFILL_ERR_SHAPE;
    if (kind != DTFILL_METRICS_KITTI && kind != DTFILL_METRICS_NYU) return DTFILL_ERR_METRIC;
    if (ws_bytes < dtfill_metrics_workspace_bytes(B)) return DTFILL_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double *part = static_cast<double *>(workspace);
    if (kind == DTFILL_METRICS_KITTI) {
        k_metrics_part<0><<<dim3(M_NB, B), 256, 0, st>>>(output, target, n, part);
        k_metrics_final<0><<<B, 64, 0, st>>>(part, M_NB, out);
    } else {
        k_metrics_part<1><<<dim3(M_NB, B), 256, 0, st>>>(output, target, n, part);
        k_metrics_final<1><<<B, 64, 0, st>>>(part, M_NB, out);
    }
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

}  // extern "C"

#ifdef PTS_PROF
extern "C" int dtfill_pts_prof(unsigned long long *out8, int reset) {
    unsigned long long z[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_pts_prof), sizeof(z)) != hipSuccess) return -1;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_pts_prof), z, sizeof(z)) != hipSuccess) return -1;
    return 0;
}
#endif
