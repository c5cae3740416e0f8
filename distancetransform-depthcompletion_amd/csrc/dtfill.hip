// dtfill.hip -- MI355X (gfx950) distance-transform + nearest-valid-depth fill.
//
// Replaces, for B frames at once, the reference's per-frame
//   nearest_point()      solution_DeepNet/tools.py:7-10   (cv2.distanceTransformWithLabels, L1, 5x5, LABEL_PIXEL)
//   DT_complete_batch()  solution_DeepNet/tools.py:13-35  (value-list compaction + depth_list[lbl-1] gather)
//   Distance_Transform() solution_DeepNet/eval_NYU.py:120-133
//
// The reference's arithmetic is OpenCV's two-pass 5x5 chamfer: two raster sweeps that are sequential
// in both image axes.  This file does NOT sweep.  It uses the following identity (derived in
// DESIGN.md "Why no raster sweep", checked bit-for-bit against the sequential restatement in oracle/):
//
//   d(q)      = exact L1 distance to the nearest source               (weights {1,2,3} == L1 norms of the taps)
//   live(q)   = "the forward sweep already reached the final value at q"
//             = some nearest source s of q lies in q's forward cone:  s above-or-left of q, or above-right
//               with (s.col - q.col) <= 2 (q.row - s.row)
//             = (dA(q) == d(q)) or (dB(q) == d(q)),  where with gu(i,k) = distance to the nearest source
//               at-or-above row i in column k:
//                 dA(i,j) = min_{k<=j} gu(i,k) + (j-k)                          (row prefix scan)
//                 dB(i,j) = 3 + D(i-1,j+2),  D(i,j) = min(E(i,j), 3 + D(i-1,j+2)) (scan along knight lines)
//                 E(i,j)  = min(gu(i,j), gu(i,j-1) - 1)
//   parent(q) = live(q) ? first forward tap r (cv2 order) with live(r) and d(r)+w == d(q)
//                       : first backward tap r (cv2 order) with d(r)+w == d(q)
//   label(q)  = label(root of the parent chain) = 1 + raster rank of that source.
//
// Every quantity is a 1-D scan along image columns, rows or knight-move lines, or a 5x5 local rule,
// so all pixels of all frames are processed in parallel; only the final chain walk is data dependent
// (chain length <= d(q) hops).
//
// Locality: everything that decides label(q) lies inside the L1 ball of radius d(q) around q.  So a
// tile plus a halo of FR pixels, held on chip, gives the exact result for every tile pixel with
// d <= FR, and detects (d > FR) the ones it cannot decide.
//
// Kernels of one l1_cv pass:
//   k_mask     source / value bit words (ballots) + per-row prefix popcounts       reads x once
//   k_frame    per-frame row-count scan -> compaction ranks; frame facts; value list when the source
//              and value masks of a frame differ (else depth_list[lbl-1] == x at the source pixel)
//   k_fused<16>  one workgroup per (tile + halo 16) window, bit-sliced: the level-synchronous form of
//              the identity above on bit planes held in registers (one lane per window row, taps as word
//              shifts), byte codes un-sliced into LDS, lock-step chain walk, rank lookup, depth gather
//              and the three output stores.  Flags the frame if any tile pixel has d > 16.
//   k_fused<32>  the same with halo 32, only for frames the first stage flagged.
//   general path (only frames k_fused<32> flagged too; blocks of other frames exit at once):
//   k_colscan, k_skew, k_rowscan  full-frame column / knight-line / row scans through HBM (uint16)
//   k_exit     5x5 parent rule + in-tile chain resolution by pointer doubling in LDS -> exit pointers
//   k_final    follows the few tile-to-tile hops, rank -> label, gather, store.      Any distance.
// l2 pass (exact Euclidean, canonical tie-break): k_mask, k_frame, k_colscan<true>, k_l2row.
//
// No MFMA anywhere: this path is compare/min/index work (DESIGN.md "Roofline").

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>
#include "../../include/dtfill.h"

typedef uint16_t u16;
typedef uint8_t u8;
typedef unsigned long long u64;
typedef uint32_t u32;

namespace {

constexpr int BIG = 1 << 20;       // in-register "infinite" distance
constexpr int INF16 = 0xFFFF;      // stored "infinite" distance in the uint16 scan arrays
constexpr int DL_DMASK = 0x3FFF;   // dl: low 14 bits = d
constexpr int DL_NONE = 0x3FFF;    // dl: no source in the frame
constexpr int DL_LIVE = 0x8000;    // dl: live flag
constexpr int PAR_SRC = 0xFF;      // parent code: pixel is a source
constexpr int PAR_NONE = 0xFE;     // parent code: unreachable / undecided
constexpr int MAX_HW_SUM = 8192;   // cv2's Q16 INIT_DIST0 = INT_MAX>>2 caps distances at 8191

// frame facts written by k_frame: int32[FI_STRIDE] per frame
constexpr int FI_NSRC = 0, FI_NVAL = 1, FI_MISALIGNED = 2, FI_DLB = 3, FI_STRIDE = 4;  // DLB: lower bound of max d

// cv2 tap order (OpenCV 3.4 distanceTransformEx_5x5), forward taps 0..7; backward tap t is the
// NEGATED forward tap t with the same weight.  Parent code = t | (backward ? 8 : 0).
#define TAP_DI(t) ((t) < 2 ? -2 : (t) < 7 ? -1 : 0)
#define TAP_DJ(t) ((t) == 0 ? -1 : (t) == 1 ? 1 : (t) == 2 ? -2 : (t) == 3 ? -1 : (t) == 4 ? 0 : (t) == 5 ? 1 : (t) == 6 ? 2 : -1)
#define TAP_W(t) ((t) < 3 ? 3 : (t) == 3 ? 2 : (t) == 4 ? 1 : (t) == 5 ? 2 : (t) == 6 ? 3 : 1)
constexpr u32 TAP_DI_NIB = 0x21111100u;  // nibble t = di(t) + 2
constexpr u32 TAP_DJ_NIB = 0x14321031u;  // nibble t = dj(t) + 2

__device__ __forceinline__ void tap_decode(int code, int &di, int &dj) {
    const int sh = (code & 7) * 4;
    di = (int)((TAP_DI_NIB >> sh) & 15u) - 2;
    dj = (int)((TAP_DJ_NIB >> sh) & 15u) - 2;
    if (code & 8) {
        di = -di;
        dj = -dj;
    }
}

__device__ __forceinline__ int ld16(const u16 *p) {
    int v = *p;
    return v == INF16 ? BIG : v;
}
__device__ __forceinline__ u16 st16(int v) { return (u16)(v >= INF16 ? INF16 : v); }

// ------------------------------------------------------------------------------------------------
// k_mask: one wave per M_RPW image rows.  Source predicate exactly as tools.py:8, mask = (1.0 - x) > thr
// (1 = fill, 0 = source); value predicate as tools.py:22, x > thr.  Per 64-pixel word: the two bit
// words and the row-local exclusive popcount; per row: totals (+ "masks differ" in bit 31).
// ------------------------------------------------------------------------------------------------
constexpr int M_RPW = 1;  // image rows per wave in k_mask
constexpr int M_KU = 8;   // 64-pixel steps whose loads are issued together (M_KU * M_RPW loads in flight per lane)

__device__ __forceinline__ u32 wave_incl_sum(u32 v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 t = __shfl_up(v, off);
        if (lane >= off) v += t;
    }
    return v;
}

__global__ __launch_bounds__(256) void k_mask(const float *__restrict__ x, int H, int W, int Wd,
                                              float src_thr, float val_thr, u64 *__restrict__ srcbits,
                                              u64 *__restrict__ valbits, u16 *__restrict__ wpre_s,
                                              u16 *__restrict__ wpre_v, u32 *__restrict__ rowcnt_s,
                                              u32 *__restrict__ rowcnt_v) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 4 + wave) * M_RPW, b = blockIdx.y;
    if (i0 >= H) return;
    u32 run_s[M_RPW], run_v[M_RPW], mis[M_RPW];
#pragma unroll
    for (int q = 0; q < M_RPW; ++q) run_s[q] = run_v[q] = mis[q] = 0;
    // 64 words (4096 pixels) of every row per chunk: lane k ends up holding word k0 + k of each row, so the
    // words and their prefix counts leave as ONE coalesced store per row and array
    for (int k0 = 0; k0 < Wd; k0 += 64) {
        const int nk = min(64, Wd - k0);
        u64 ws[M_RPW], wv[M_RPW];
#pragma unroll
        for (int q = 0; q < M_RPW; ++q) ws[q] = wv[q] = 0;
        for (int kb = 0; kb < nk; kb += M_KU) {  // M_KU word steps x M_RPW rows: all loads first, then the ballots
            float v[M_KU][M_RPW];
#pragma unroll
            for (int u = 0; u < M_KU; ++u) {
                const int j = (k0 + kb + u) * 64 + lane;
#pragma unroll
                for (int q = 0; q < M_RPW; ++q) {
                    const int i = min(i0 + q, H - 1);
                    v[u][q] = (kb + u < nk && j < W) ? x[((size_t)b * H + i) * W + j] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < M_KU; ++u) {
                const int k = kb + u;
                const bool in = k < nk && (k0 + k) * 64 + lane < W;
#pragma unroll
                for (int q = 0; q < M_RPW; ++q) {
                    const u64 sb = __ballot(in && !((1.0f - v[u][q]) > src_thr));
                    const u64 vb = __ballot(in && (v[u][q] > val_thr));
                    ws[q] = lane == k ? sb : ws[q];
                    wv[q] = lane == k ? vb : wv[q];
                }
            }
        }
#pragma unroll
        for (int q = 0; q < M_RPW; ++q) {
            const u32 cs = __popcll(ws[q]), cv = __popcll(wv[q]);
            const u32 is = wave_incl_sum(cs, lane), iv = wave_incl_sum(cv, lane);
            mis[q] |= __any(ws[q] != wv[q]) ? 1u : 0u;
            if (lane < nk && i0 + q < H) {
                const size_t wi = ((size_t)b * H + i0 + q) * Wd + k0 + lane;
                srcbits[wi] = ws[q];
                valbits[wi] = wv[q];
                wpre_s[wi] = (u16)(run_s[q] + is - cs);
                wpre_v[wi] = (u16)(run_v[q] + iv - cv);
            }
            run_s[q] += __shfl(is, 63);
            run_v[q] += __shfl(iv, 63);
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < M_RPW; ++q)
            if (i0 + q < H) {
                rowcnt_s[(size_t)b * H + i0 + q] = run_s[q];
                rowcnt_v[(size_t)b * H + i0 + q] = run_v[q] | (mis[q] ? 0x80000000u : 0u);
            }
    }
}

// ------------------------------------------------------------------------------------------------
// k_frame: one workgroup per frame.  Exclusive scan of the row counts = raster rank of the first
// source / value pixel of every row: cv2's label init (k=1; every zero pixel gets k++) and numpy's
// boolean compaction x[with_value] (tools.py:24).  The value list is only materialised when the two
// masks differ somewhere in the frame.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame(const float *__restrict__ x, const u64 *__restrict__ valbits,
                                               const u16 *__restrict__ wpre_v,
                                               const u32 *__restrict__ rowcnt_s,
                                               const u32 *__restrict__ rowcnt_v, int H, int W, int Wd,
                                               u32 *__restrict__ rowbase_s, u32 *__restrict__ rowbase_v,
                                               int *__restrict__ finfo, float *__restrict__ vlist,
                                               int *__restrict__ fflag, int *__restrict__ fflag2,
                                               int *__restrict__ frame_status, int force_general) {
    __shared__ u32 s_ws[4], s_wv[4];
    __shared__ int s_mis, s_dlb;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 *cs_ = rowcnt_s + (size_t)b * H, *cv_ = rowcnt_v + (size_t)b * H;
    if (tid == 0) s_dlb = 0;
    u32 *bs_ = rowbase_s + (size_t)b * H, *bv_ = rowbase_v + (size_t)b * H;
    if (tid == 0) s_mis = 0;
    __syncthreads();
    u32 run_s = 0, run_v = 0;
    int mis = 0;
    for (int base = 0; base < H; base += 256) {
        const int i = base + tid;
        u32 cs = 0, cv = 0;
        if (i < H) {
            cs = cs_[i];
            cv = cv_[i];
            mis |= (int)(cv >> 31);
            cv &= 0x7FFFFFFFu;
        }
        u32 is = cs, iv = cv;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            u32 ts = __shfl_up(is, off), tv = __shfl_up(iv, off);
            if (lane >= off) {
                is += ts;
                iv += tv;
            }
        }
        if (lane == 63) {
            s_ws[wave] = is;
            s_wv[wave] = iv;
        }
        __syncthreads();
        u32 ps = 0, pv = 0, ts = 0, tv = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (k < wave) {
                ps += s_ws[k];
                pv += s_wv[k];
            }
            ts += s_ws[k];
            tv += s_wv[k];
        }
        if (i < H) {
            bs_[i] = run_s + ps + is - cs;
            bv_[i] = run_v + pv + iv - cv;
        }
        run_s += ts;
        run_v += tv;
        __syncthreads();
    }
    if (mis) atomicOr(&s_mis, 1);
    // Lower bound of the largest distance in the frame from runs of rows without any source: a run of k
    // empty rows forces d >= ceil(k/2) in its middle row, d >= k at the border row if it touches the top or
    // bottom edge.  Each thread looks at the run ENDING at its rows (cheap: the counts are in L2).
    {
        // "row has no source" as bits in LDS (H <= 8191 -> 256 words), so walking a run costs LDS reads only
        __shared__ u32 s_empty[256];
        s_empty[tid] = 0;
        __syncthreads();
        for (int i = tid; i < H; i += 256)
            if (cs_[i] == 0) atomicOr(&s_empty[i >> 5], 1u << (i & 31));
        __syncthreads();
        auto empty = [&](int i) { return (s_empty[i >> 5] >> (i & 31)) & 1u; };
        int dlb = 0;
        for (int i = tid; i < H; i += 256) {
            if (!empty(i) || (i + 1 < H && empty(i + 1))) continue;  // not the last row of a run
            int k = 1;
            while (i - k >= 0 && empty(i - k)) ++k;
            const bool edge = (i - k < 0) || (i + 1 >= H);
            dlb = max(dlb, (i - k < 0 && i + 1 >= H) ? BIG : edge ? k : (k + 1) / 2);
        }
        if (dlb) atomicMax(&s_dlb, dlb);
    }
    __syncthreads();
    const int misaligned = s_mis;
    if (tid == 0) {
        finfo[b * FI_STRIDE + FI_NSRC] = (int)run_s;
        finfo[b * FI_STRIDE + FI_NVAL] = (int)run_v;
        finfo[b * FI_STRIDE + FI_MISALIGNED] = misaligned;
        finfo[b * FI_STRIDE + FI_DLB] = s_dlb;
        fflag[b] = 0;                       // set by k_fused<16>: the frame needs a wider halo
        fflag2[b] = force_general ? 1 : 0;  // set by k_fused<32>: the frame needs the general path
        frame_status[b] = force_general ? DTFILL_FRAME_GENERAL_PATH : DTFILL_FRAME_OK;
    }
    if (misaligned) {
        // rare path: scatter x at value pixels into the compacted value list
        const float *xf = x + (size_t)b * H * W;
        float *vl = vlist + (size_t)b * H * W;
        const int nwords = H * Wd;
        for (int w = tid; w < nwords; w += 256) {
            u64 vb = valbits[(size_t)b * nwords + w];
            const int i = w / Wd, j0 = (w - i * Wd) * 64;
            u32 k = bv_[i] + wpre_v[(size_t)b * nwords + w];
            while (vb) {
                const int bit = __ffsll((long long)vb) - 1;
                vb &= vb - 1;
                vl[k++] = xf[(size_t)i * W + j0 + bit];
            }
        }
    }
}

// label of the source at (i, j): 1 + number of sources before it in raster order
__device__ __forceinline__ int source_rank(u32 base, u64 word, int j) {
    return (int)base + __popcll(word & ((1ull << (j & 63)) - 1ull)) + 1;
}

// gather depth_list[label-1] with numpy's index semantics (tools.py:26)
__device__ __forceinline__ float gather_depth(const float *__restrict__ xf, const float *__restrict__ vlf,
                                              int label, int src_pixel, int nval, int misaligned,
                                              int *frame_status_b) {
    int idx = label - 1;
    if (idx < 0) idx += nval;  // numpy: index -1 wraps to the last element
    if (idx < 0 || idx >= nval) {
        atomicOr(frame_status_b, DTFILL_FRAME_INDEX_ERROR);
        return nanf("");
    }
    if (misaligned) return vlf[idx];
    return xf[src_pixel];  // masks agree: the label-th value IS the source pixel's own depth
}

// ------------------------------------------------------------------------------------------------
// k_fused: one workgroup (2 waves) per window = tile (<= 96 x 160) + halo FR, bit-sliced.
//
// Lane r owns window row r as six 32-bit words per bit plane (bit c%32 of word c/32 = window column c).
// Level-synchronous form of the identity in the file header (checked in
// tests/parallel_model.py): with E_t = {d == t} and L_t = live pixels of E_t (E_0 = L_0 = sources),
//   E_t = dilate4(D_{t-1}) & ~D_{t-1} & in-image
//   forward tap T = (di,dj,w) offers   shift(L_{t-w}, di, dj)   to the pixels of E_t; L_t = those offered any
//   backward tap (negated offset)      shift(E_{t-w}, -di, -dj) to E_t \ L_t
//   the FIRST tap in cv2 order wins (taken-mask chain); the winning step is recorded in six "code planes"
//   holding the bits of enc = (di+2)<<3 | (dj+2)  (sources: enc 18 = step (0,0)).
// A horizontal shift of a row is one v_alignbit per word; rows r-2..r+2 of the previous three levels
// come from a 4-slot LDS ring (one barrier per level).  Levels stop at FR or when a level is empty.
// Then every lane un-slices its row, 4 pixels per step, into the byte array s_par (0x80 | enc; 0x80
// itself = undecided), which reuses the ring's memory, and the tile pixels walk to their sources in
// lock-step; d is |drow| + |dcol| to the root.
// LDS: 25.3 KB ring/s_par + 6 KB bit words and ranks.
// ------------------------------------------------------------------------------------------------
constexpr int F_WHM = 128;  // window rows
constexpr int F_WWM = 192;  // window columns = 6 words
constexpr int F_NT = 256;   // lane = (row, half): waves 0-1 own words 0..2 of rows 0..127, waves 2-3 words 3..5
constexpr int F_P = 196;               // s_par row pitch: 49 dwords (odd) -> lane-per-row dword stores are conflict-free
constexpr int F_NWD = 6;               // 32-bit words per window row
constexpr int F_HW = 3;                // words per lane
constexpr int F_RS = 7;                // ring row stride in words: 6 + one zero pad (odd: conflict-free; the pad is
                                       // also the zero "word -1" of the next row and "word 6" of this one)
constexpr int F_RROWS = F_WHM + 4;     // ring rows: 2 zero rows above and below the window
constexpr int F_RPLANE = F_RROWS * F_RS;
constexpr int F_RING = 1 + 4 * 2 * F_RPLANE;  // one leading zero word, then [slot][plane E/L][row][7]
constexpr int F_EB = 8;                // tile pixels per lane walked in lock-step
// byte code of a source: 0x80 | 18 = step (0,0)
constexpr int F_NONE = 0xC0 | 18;      // byte code of an undecided pixel: also step (0,0), plus bit 6
static_assert(F_WWM == 32 * F_NWD, "window width must be six words");
static_assert(F_RING * 4 >= F_WHM * F_P, "s_par must fit in the ring's memory");

#define ENC_F(t) (((TAP_DI(t) + 2) << 3) | (TAP_DJ(t) + 2))

// a[0..4] = the lane's three words a[1..3] with their left / right neighbour words; word i (0..2) of the
// row shifted so that result[c] = row[c + DJ]
template <int DJ>
__device__ __forceinline__ u32 hshift(const u32 (&a)[5], int i) {
    if (DJ == 0) return a[i + 1];
    if (DJ > 0) return __builtin_amdgcn_alignbit(a[i + 2], a[i + 1], DJ);
    return __builtin_amdgcn_alignbit(a[i + 1], a[i], 32 + DJ);
}

// one tap of the first-match chain: cand = shift(src, DJ); winners get the bits of ENC in the code planes
template <int DJ, int ENC>
__device__ __forceinline__ void tap_step(const u32 (&src)[5], u32 (&taken)[F_HW], u32 (&C)[6][F_HW]) {
#pragma unroll
    for (int i = 0; i < F_HW; ++i) {
        const u32 cand = hshift<DJ>(src, i);
        const u32 sel = cand & ~taken[i];
        taken[i] |= cand;
#pragma unroll
        for (int j = 0; j < 6; ++j)
            if (ENC & (1 << j)) C[j][i] |= sel;
    }
}

// the lane's three words of a ring row plus one neighbour word on each side (pads / other half)
__device__ __forceinline__ void ring_load5(const u32 *__restrict__ ring, int slot, int plane, int row, int wb,
                                           u32 (&a)[5]) {
    const u32 *p = ring + 1 + (slot * 2 + plane) * F_RPLANE + row * F_RS + wb - 1;
#pragma unroll
    for (int i = 0; i < 5; ++i) a[i] = p[i];
}
__device__ __forceinline__ void ring_store3(u32 *__restrict__ ring, int slot, int plane, int row, int wb,
                                            const u32 (&w)[F_HW]) {
    u32 *p = ring + 1 + (slot * 2 + plane) * F_RPLANE + row * F_RS + wb;
#pragma unroll
    for (int i = 0; i < F_HW; ++i) p[i] = w[i];
}

// FR = halo = largest distance the window can decide.  gate (nullable): only frames with gate[b] != 0
// are processed (the second, FR = 32 stage only redoes the frames the FR = 16 stage flagged).
template <int FR>
__global__ __launch_bounds__(F_NT) void k_fused(
    const float *__restrict__ x, const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s,
    const u32 *__restrict__ rowbase_s, const int *__restrict__ finfo, const float *__restrict__ vlist,
    int H, int W, int Wd, int TH, int TW, int tiles_x, float *__restrict__ out_depth,
    float *__restrict__ out_dt, int32_t *__restrict__ out_index, const int *__restrict__ gate,
    int *__restrict__ fflag, int *__restrict__ frame_status, int stop_after) {
    if (gate && !gate[blockIdx.y]) return;
    // Speed heuristic only (never correctness): with source density p the chance that a pixel has no source
    // within L1 distance FR is about (1-p)^(2 FR^2 + 2 FR + 1); if the frame is expected to hold such a pixel
    // anyway (N (1-p)^ball > ~1, i.e. p * ball < ln N ~ 14), this stage would only flag the frame after doing
    // all the work -- hand it on right away.  Likewise when k_frame found a run of source-free rows that forces
    // some distance above FR (real LiDAR frames: the empty sky rows).  (The two finfo loads are issued together
    // with the window loads below; the branch comes after those are in flight.)
    const long long h_nsrc = finfo[blockIdx.y * FI_STRIDE + FI_NSRC];
    const int h_dlb = finfo[blockIdx.y * FI_STRIDE + FI_DLB];
    __shared__ __attribute__((aligned(16))) u32 s_ring[F_RING];  // later: s_par bytes
    __shared__ u32 s_sb[F_WHM * 8];  // source bits of the window rows, image-aligned 64-pixel words (as u32 pairs)
    __shared__ u32 s_rk[F_WHM * 4];  // sources before each of those 64-pixel words (frame raster order)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWAVE = F_NT / 64;
    const int b = blockIdx.y;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * TH, c0 = tx * TW;
    const int th = min(TH, H - r0), tw = min(TW, W - c0);
    const int wr0 = r0 - FR, wc0 = c0 - FR;  // image coords of window cell (0,0)
    const int WH = th + 2 * FR, WW = tw + 2 * FR;
    const int ca = max(0, -wc0), cb = min(WW, W - wc0);  // in-image window columns [ca, cb)
    const int ra = max(0, -wr0), rb = min(WH, H - wr0);  // in-image window rows
    const int w0 = wc0 >> 6;                             // first image word column the window touches (-1 if wc0 < 0)
    const int sh = wc0 - 64 * w0;                        // window column 0 is bit sh of image word w0

    // ---- P0: this lane's half row: image-aligned words -> LDS (for the ranks), window-aligned planes -> registers
    const int r = tid & (F_WHM - 1);  // window row of this lane
    const int hf = tid >> 7;          // which half of the row (wave-uniform)
    const int wb = F_HW * hf;         // first of the lane's three words
    u32 M[F_HW], D[F_HW];
    {
        const int gi = wr0 + r;
        const bool rowin = r < WH && gi >= 0 && gi < H;
        // the lane's 96 window columns start at bit sh + 96 hf of the row's image-aligned bit string;
        // three image words (192 bits) starting at word (sh + 96 hf) / 64 cover them
        const int bit0 = sh + 96 * hf;
        const int kw = bit0 >> 6;  // wave-uniform
        u32 g[7];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int kk = kw + k;  // 0..3 relative to w0
            const int w = w0 + kk;
            u64 sb = 0;
            u32 rk = 0;
            if (rowin && kk < 4 && w >= 0 && w < Wd) {
                const size_t wi = ((size_t)b * H + gi) * Wd + w;
                sb = srcbits[wi];
                rk = rowbase_s[(size_t)b * H + gi] + wpre_s[wi];
            }
            g[2 * k] = (u32)sb;
            g[2 * k + 1] = (u32)(sb >> 32);
            // each of the four image words of a row is stored once: half 0 stores its words kk = 0, 1 (and 2
            // if half 1 starts later), half 1 the rest
            const bool mine = hf == 0 ? kk < 2 : (kk >= 2 && kk < 4);
            if (mine) {
                s_sb[r * 8 + 2 * kk] = g[2 * k];
                s_sb[r * 8 + 2 * kk + 1] = g[2 * k + 1];
                s_rk[r * 4 + kk] = rk;
            }
        }
        g[6] = 0;
        const int s6 = bit0 & 63;
        const bool hi = s6 & 32;  // wave-uniform
        const int s5 = s6 & 31;
#pragma unroll
        for (int i = 0; i < F_HW; ++i) {
            const u32 lo_w = hi ? g[i + 1] : g[i], hi_w = hi ? g[i + 2] : g[i + 1];
            const u32 word = s5 ? __builtin_amdgcn_alignbit(hi_w, lo_w, s5) : lo_w;
            // in-image columns of window word wb + i: [max(ca, 32 (wb+i)), min(cb, 32 (wb+i) + 32))
            const int lo = max(ca - 32 * (wb + i), 0), up = min(cb - 32 * (wb + i), 32);
            u32 m = 0;
            if (rowin && up > lo) m = (up >= 32 ? 0xFFFFFFFFu : ((1u << up) - 1u)) & ~((1u << lo) - 1u);
            M[i] = m;
            D[i] = word & m;
        }
    }
    if (h_nsrc * (2 * FR * FR + 2 * FR + 1) < 14ll * H * W || h_dlb > FR) {  // block-uniform
        if (threadIdx.x == 0 && blockIdx.x == 0) {
            fflag[blockIdx.y] = 1;
            if (FR == 32) atomicOr(frame_status + blockIdx.y, DTFILL_FRAME_GENERAL_PATH);
        }
        return;
    }
    // level 0: E_0 = L_0 = sources; zero the rest of the ring (levels "-1,-2,-3", the guard rows, the pads)
    for (int k = tid; k < F_RING; k += F_NT) s_ring[k] = 0;
    __syncthreads();
    ring_store3(s_ring, 0, 0, r + 2, wb, D);
    ring_store3(s_ring, 0, 1, r + 2, wb, D);
    u32 C[6][F_HW];
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int i = 0; i < F_HW; ++i) C[j][i] = ((18 >> j) & 1) ? D[i] : 0u;  // sources: enc 18
    u32 Dup[F_HW], Ddn[F_HW];
    u32 Dl = 0, Dr = 0;  // D's neighbour words left / right of the lane's three (other half or nothing)
#pragma unroll
    for (int i = 0; i < F_HW; ++i) Dup[i] = Ddn[i] = 0;
    __syncthreads();
    if (stop_after == 0) return;  // timing-only builds of bench (DTFILL_FUSED_STOP)

    // ---- P1: levels
    for (int t = 1; t <= FR; ++t) {
        const int s1 = (t - 1) & 3, s2 = (t - 2) & 3, s3 = (t - 3) & 3, sw = t & 3;
        u32 nb[5], e1[5], l1[5], taken[F_HW], Et[F_HW], Lt[F_HW];
        // dilation of D_{t-1}: left/right in registers (+ the neighbour words), up/down through E_{t-1}
        ring_load5(s_ring, s1, 0, r + 2, wb, e1);  // E_{t-1}, this row (also the last tap's source)
        Dl |= e1[0];
        Dr |= e1[4];
        ring_load5(s_ring, s1, 0, r + 1, wb, nb);  // E_{t-1}, row r-1
#pragma unroll
        for (int i = 0; i < F_HW; ++i) Dup[i] |= nb[i + 1];
        u32 e1d[5];
        ring_load5(s_ring, s1, 0, r + 3, wb, e1d);  // E_{t-1}, row r+1
        bool nonempty = false;
        {
            const u32 dd[5] = {Dl, D[0], D[1], D[2], Dr};
#pragma unroll
            for (int i = 0; i < F_HW; ++i) {
                Ddn[i] |= e1d[i + 1];
                const u32 dil = hshift<1>(dd, i) | hshift<-1>(dd, i) | Dup[i] | Ddn[i];
                Et[i] = dil & ~D[i] & M[i];
                taken[i] = ~Et[i];
                nonempty |= Et[i] != 0;
            }
        }
        // forward taps in cv2 order; the candidates are live pixels of levels t-3, t-2, t-1
        ring_load5(s_ring, s3, 1, r + 0, wb, nb);  // L_{t-3}, row r-2
        tap_step<-1, ENC_F(0)>(nb, taken, C);
        tap_step<1, ENC_F(1)>(nb, taken, C);
        {
            u32 l3[5], l2[5];
            ring_load5(s_ring, s3, 1, r + 1, wb, l3);  // L_{t-3}, row r-1
            ring_load5(s_ring, s2, 1, r + 1, wb, l2);  // L_{t-2}, row r-1
            ring_load5(s_ring, s1, 1, r + 1, wb, nb);  // L_{t-1}, row r-1
            tap_step<-2, ENC_F(2)>(l3, taken, C);
            tap_step<-1, ENC_F(3)>(l2, taken, C);
            tap_step<0, ENC_F(4)>(nb, taken, C);
            tap_step<1, ENC_F(5)>(l2, taken, C);
            tap_step<2, ENC_F(6)>(l3, taken, C);
        }
        ring_load5(s_ring, s1, 1, r + 2, wb, l1);  // L_{t-1}, this row
        tap_step<-1, ENC_F(7)>(l1, taken, C);
#pragma unroll
        for (int i = 0; i < F_HW; ++i) {
            Lt[i] = taken[i] & Et[i];
            taken[i] = ~(Et[i] & ~Lt[i]);  // backward chain only for the non-live pixels of E_t
        }
        // backward taps (negated offsets, same order); candidates are ALL pixels of levels t-3, t-2, t-1
        ring_load5(s_ring, s3, 0, r + 4, wb, nb);  // E_{t-3}, row r+2
        tap_step<1, 36 - ENC_F(0)>(nb, taken, C);
        tap_step<-1, 36 - ENC_F(1)>(nb, taken, C);
        {
            u32 e3[5], e2[5];
            ring_load5(s_ring, s3, 0, r + 3, wb, e3);  // E_{t-3}, row r+1
            ring_load5(s_ring, s2, 0, r + 3, wb, e2);  // E_{t-2}, row r+1
            tap_step<2, 36 - ENC_F(2)>(e3, taken, C);
            tap_step<1, 36 - ENC_F(3)>(e2, taken, C);
            tap_step<0, 36 - ENC_F(4)>(e1d, taken, C);
            tap_step<-1, 36 - ENC_F(5)>(e2, taken, C);
            tap_step<-2, 36 - ENC_F(6)>(e3, taken, C);
        }
        tap_step<1, 36 - ENC_F(7)>(e1, taken, C);
#pragma unroll
        for (int i = 0; i < F_HW; ++i) D[i] |= Et[i];
        ring_store3(s_ring, sw, 0, r + 2, wb, Et);
        ring_store3(s_ring, sw, 1, r + 2, wb, Lt);
        if (!__syncthreads_or(nonempty)) break;  // nothing at distance t anywhere: nothing farther either
    }
    if (stop_after == 1) return;  // timing-only builds of bench (DTFILL_FUSED_STOP)

    // ---- P2: un-slice the code planes of this half row into bytes, 4 pixels per step.
    // ((nibble * 0x00204081) & 0x01010101) spreads bits 0..3 of the nibble to the low bits of 4 bytes.
    u8 *s_par = reinterpret_cast<u8 *>(s_ring);
    __syncthreads();  // the ring is dead for everybody before its memory becomes s_par
    {
        u32 *prow = reinterpret_cast<u32 *>(s_par + r * F_P) + wb * 8;
#pragma unroll
        for (int i = 0; i < F_HW; ++i) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                u32 v = 0x80808080u;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const u32 nib = (C[j][i] >> (4 * q)) & 0xFu;
                    v |= ((nib * 0x00204081u) & 0x01010101u) << j;
                }
                // undecided pixels (not in D): no plane bit is set; give them F_NONE = 0x80 | 0x52
                const u32 und = ((~D[i] >> (4 * q)) & 0xFu) * 0x00204081u & 0x01010101u;
                prow[i * 8 + q] = v | und * 0x52u;
            }
        }
    }
    __syncthreads();
    if (stop_after == 2) return;  // timing-only builds of bench (DTFILL_FUSED_STOP)

    // ---- P3: tile pixels: walk to the source, d, rank -> label, gather, store.  Each lane walks F_EB
    // pixels in lock-step (their LDS reads are independent, so the hop latencies overlap) and then has
    // F_EB global gathers in flight together.
    const size_t fo = (size_t)b * H * W;
    const int nval = finfo[b * FI_STRIDE + FI_NVAL];
    const int misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    bool overflow = false;
    for (int tc = lane; tc < tw; tc += 64) {
        const int cc = FR + tc;
        for (int trb = wave; trb < th; trb += NWAVE * F_EB) {
            int pos[F_EB], code[F_EB];  // pos = row * F_P + col: byte index of the walker in s_par
            bool ok[F_EB];
#pragma unroll
            for (int e = 0; e < F_EB; ++e) {
                const int tr = trb + e * NWAVE;
                pos[e] = (FR + min(tr, th - 1)) * F_P + cc;
                code[e] = s_par[pos[e]];
                ok[e] = tr < th && code[e] != F_NONE;
                overflow |= tr < th && code[e] == F_NONE;  // undecidable here: the frame takes the general path
            }
            // Unconditional hops: sources and undecided cells carry the step (0,0), so a walker that has
            // arrived just stays.  No selects, no divergent control flow -- the reads of the F_EB walkers
            // overlap.  Two hops between "everybody arrived?" checks.
            for (int hop = 0; hop < FR; hop += 2) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                    for (int e = 0; e < F_EB; ++e) {
                        const int c = code[e];
                        pos[e] += (int)((c >> 3) & 7) * F_P + (c & 7) - (2 * F_P + 2);
                    }
#pragma unroll
                    for (int e = 0; e < F_EB; ++e) code[e] = s_par[pos[e]];
                }
                int notdone = 0;
#pragma unroll
                for (int e = 0; e < F_EB; ++e) notdone |= (code[e] | 0x40) ^ F_NONE;  // 0 iff step (0,0)
                if (!__any(notdone != 0)) break;
            }
            if (stop_after == 3) {  // timing only: keep the walk alive, skip the rest
#pragma unroll
                for (int e = 0; e < F_EB; ++e) asm volatile("" ::"v"(pos[e]));
                continue;
            }
            int lab[F_EB], goff[F_EB], dd[F_EB];
            float val[F_EB];
            bool bad = false;
            const float *gbase = misaligned ? vlist + fo : x + fo;  // block-uniform
#pragma unroll
            for (int e = 0; e < F_EB; ++e) {
                // a decided chain ends on a source inside the in-image window; the clamps only make sure
                // that a logic error could never become a wild global access
                const int pr_ = pos[e] / F_P, pc_ = pos[e] - pr_ * F_P;
                const int r_ = min(max(pr_, ra), rb - 1), c_ = min(max(pc_, ca), cb - 1);
                const int tr = min(trb + e * NWAVE, th - 1);
                dd[e] = abs(r_ - (FR + tr)) + abs(c_ - cc);  // L1 distance to the nearest source IS d
                const int gj = wc0 + c_;
                const int k = r_ * 4 + (gj >> 6) - w0;
                const u32 lo = s_sb[2 * k], hi = s_sb[2 * k + 1];
                const u32 below = (1u << (gj & 31)) - 1u;
                lab[e] = (int)s_rk[k] + ((gj & 32) ? __popc(lo) + __popc(hi & below) : __popc(lo & below)) + 1;
                // depth_list[label-1] (tools.py:26).  A decided pixel has label >= 1, so the numpy wrap of
                // index -1 cannot occur here; an index past the value list is numpy's IndexError.
                const int idx = lab[e] - 1;
                const bool oob = idx >= nval;
                bad |= ok[e] && oob;
                goff[e] = oob ? 0 : (misaligned ? idx : (wr0 + r_) * W + gj);  // masks agree: the label-th value is x at the source
            }
#pragma unroll
            for (int e = 0; e < F_EB; ++e) val[e] = gbase[goff[e]];
            if (stop_after == 4) {
#pragma unroll
                for (int e = 0; e < F_EB; ++e) asm volatile("" ::"v"(val[e]), "v"(lab[e]), "v"(dd[e]));
                continue;
            }
            if (bad && out_depth) atomicOr(frame_status + b, DTFILL_FRAME_INDEX_ERROR);
#pragma unroll
            for (int e = 0; e < F_EB; ++e) {
                if (!ok[e]) continue;
                const size_t o = fo + (size_t)(r0 + trb + e * NWAVE) * W + (c0 + tc);
                if (out_index) out_index[o] = lab[e];
                if (out_dt) out_dt[o] = (float)dd[e];
                if (out_depth) out_depth[o] = val[e];
            }
        }
    }
    if (overflow) {
        fflag[b] = 1;  // same-value race
        if (FR == 32) atomicOr(frame_status + b, DTFILL_FRAME_GENERAL_PATH);  // last fused stage: general path next
    }
}

// ================================================================================================
// General path (any distance).  Every kernel returns at once for frames k_fused did not flag.
// ================================================================================================

constexpr int G_NCH = 16;  // row chunks (= waves per block) of the chunked column / knight-line scans

// k_colscan: 64 adjacent image columns per block, one wave per chunk of rows.
//   pass A: last / first source row of every column inside the chunk -> LDS
//   pass B: carry in the nearest source row above / below the chunk, then
//           down sweep: gu(i,j) = rows to the nearest source at or above (i,j);  up sweep: g = min(gu, gd).
// L2 = true (the `l2` metric): g carries in bit 15 whether that nearest source is BELOW the pixel (strictly
// nearer than the one above: on a vertical tie the upper source has the smaller raster index).
template <bool L2>
__global__ __launch_bounds__(64 * G_NCH) void k_colscan(const u64 *__restrict__ srcbits,
                                                        const int *__restrict__ fflag, int H, int W, int Wd,
                                                        int CR, u16 *__restrict__ gu, u16 *__restrict__ g) {
    __shared__ int s_last[G_NCH][64], s_first[G_NCH][64];
    const int b = blockIdx.y, wd = blockIdx.x, lane = threadIdx.x & 63, ch = threadIdx.x >> 6;
    if (fflag && !fflag[b]) return;
    const int j = wd * 64 + lane;
    const bool inb = j < W;
    const size_t fo = (size_t)b * H * W;
    u16 *guf = gu + fo, *gf = g + fo;
    const u64 *sbf = srcbits + (size_t)b * H * Wd + wd;
    const int i0 = min(ch * CR, H), i1 = min(i0 + CR, H);

    int last = -BIG, first = BIG;
    for (int i = i0; i < i1; ++i) {
        const bool s = (sbf[(size_t)i * Wd] >> lane) & 1ull;
        last = s ? i : last;
        first = s ? min(first, i) : first;
    }
    s_last[ch][lane] = last;
    s_first[ch][lane] = first;
    __syncthreads();
    int above = -BIG, below = BIG;
#pragma unroll
    for (int c = 0; c < G_NCH; ++c) {
        above = c < ch ? max(above, s_last[c][lane]) : above;
        below = c > ch ? min(below, s_first[c][lane]) : below;
    }
    int up = min(i0 - 1 - above, BIG);  // value "at row i0-1"
    int dn = min(below - i1, BIG);      // value "at row i1"
    if (CR <= 32) {
        // fast path (H <= 512): the chunk's source bits sit in one register, the from-below distances of its
        // rows in (statically indexed) registers; one store pass, nothing is re-read
        u32 bits = 0;
        for (int i = i0; i < i1; ++i) bits |= (u32)((sbf[(size_t)i * Wd] >> lane) & 1ull) << (i - i0);
        const int n = i1 - i0;
        int dnv[32];
#pragma unroll
        for (int k = 31; k >= 0; --k) {
            if (k < n) dn = (bits >> k) & 1u ? 0 : min(dn + 1, BIG);  // rows past the chunk end leave dn at "row i1"
            dnv[k] = dn;
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (k < n) {
                up = (bits >> k) & 1u ? 0 : min(up + 1, BIG);
                if (inb) {
                    const size_t o = (size_t)(i0 + k) * W + j;
                    guf[o] = st16(up);
                    if (L2) {
                        const int m = min(up, dnv[k]);
                        gf[o] = m >= 0x7FFF ? (u16)INF16 : (u16)(m | (dnv[k] < up ? 0x8000 : 0));
                    } else {
                        gf[o] = st16(min(up, dnv[k]));
                    }
                }
            }
        }
        return;
    }
#pragma unroll 4
    for (int i = i0; i < i1; ++i) {
        const bool s = (sbf[(size_t)i * Wd] >> lane) & 1ull;
        up = s ? 0 : min(up + 1, BIG);
        if (inb) guf[(size_t)i * W + j] = st16(up);
    }
#pragma unroll 4
    for (int i = i1 - 1; i >= i0; --i) {
        const bool s = (sbf[(size_t)i * Wd] >> lane) & 1ull;
        dn = s ? 0 : min(dn + 1, BIG);
        if (inb) {
            const int u = ld16(guf + (size_t)i * W + j);
            if (L2) {
                const int m = min(u, dn);
                gf[(size_t)i * W + j] = m >= 0x7FFF ? (u16)INF16 : (u16)(m | (dn < u ? 0x8000 : 0));
            } else {
                gf[(size_t)i * W + j] = st16(min(u, dn));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// l2 metric: k_l2row.  One lane per pixel.  With g(i,k) the vertical distance to the nearest source of
// column k, the exact squared Euclidean distance is min_k g(i,k)^2 + (j-k)^2; the columns are visited
// outward from j (r = |j-k| = 0,1,2,...) and the search stops once r^2 exceeds the best value, so the
// work per pixel is ~2 sqrt(d^2) candidates.  Ties go to the smallest raster index of the SOURCE
// (smaller row, then smaller column) -- the order brute force gives.  Then rank -> label, gather, store.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_l2row(const float *__restrict__ x, const u16 *__restrict__ g,
                                               const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s,
                                               const u32 *__restrict__ rowbase_s, const int *__restrict__ finfo,
                                               const float *__restrict__ vlist, int H, int W, int Wd,
                                               float *__restrict__ out_depth, float *__restrict__ out_dt,
                                               int32_t *__restrict__ out_index, int *__restrict__ frame_status) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= H * W) return;
    const size_t fo = (size_t)b * H * W;
    const int i = p / W, j = p - i * W;
    const u16 *grow = g + fo + (size_t)i * W;
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

// k_skew: knight lines u = j + 2 i over the extended column range j in [0, W] (column W is virtual:
// E(i,W) = gu(i,W-1) - 1).  64 adjacent lines per block, one wave per chunk of the rows those lines
// cross; all lanes of a wave sit in the same image row at each step, so the gu reads and dB writes
// of a step are contiguous.  D(i) = min(E(i), 3 + D(i-1)) is scanned per chunk from "infinity"
// (pass A), the true value at each chunk start follows from the chunk ends (LDS), and pass B rescans
// from it and stores dB = 3 + D(previous row).
// rows [c0, c1) of the lane's knight line.  8 rows at a time: their 16 loads are unconditional (clamped
// addresses, the predicates are applied afterwards with selects), so they are all in flight together.
// Returns D after row c1-1; if dBf, stores dB = 3 + D(previous row).
__device__ __forceinline__ int skew_run(const u16 *__restrict__ guf, u16 *__restrict__ dBf, int W, int nU, int u,
                                        int c0, int c1, int D) {
    const bool lane_on = u < nU;
    for (int ib = c0; ib < c1; ib += 8) {
        int ga[8], gb[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = min(ib + t, c1 - 1);
            const int j = u - 2 * i;
            const u16 *row = guf + (size_t)i * W;
            ga[t] = row[min(max(j, 0), W - 1)];
            gb[t] = row[min(max(j - 1, 0), W - 1)];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int i = ib + t;
            const int j = u - 2 * i;
            const bool on = lane_on && i < c1 && j >= 0 && j <= W;
            const int dbv = min(D + 3, BIG);  // 3 + D(i-1, j+2)
            const int a = ga[t] == INF16 ? BIG : ga[t], bq = gb[t] == INF16 ? BIG : gb[t] - 1;
            int e = (on && j < W) ? a : BIG;
            e = (on && j >= 1) ? min(e, bq) : e;
            if (dBf && on && j < W) dBf[(size_t)i * W + j] = st16(dbv);
            D = i < c1 ? (on ? min(e, dbv) : BIG) : D;
        }
    }
    return D;
}

__global__ __launch_bounds__(64 * G_NCH) void k_skew(const u16 *__restrict__ gu, const int *__restrict__ fflag,
                                                     int H, int W, u16 *__restrict__ dB) {
    __shared__ int s_end[G_NCH][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, ch = threadIdx.x >> 6;
    if (!fflag[b]) return;
    const int nU = W + 2 * (H - 1) + 1;
    const int u0 = blockIdx.x * 64;
    const int u = u0 + lane;
    const int u1 = min(u0 + 63, nU - 1);
    const size_t fo = (size_t)b * H * W;
    const u16 *guf = gu + fo;
    u16 *dBf = dB + fo;

    const int i_lo = max(0, (u0 - W + 1) / 2);  // first row any lane of this block is inside [0, W]
    const int i_hi = min(H - 1, u1 / 2);
    const int CR = (i_hi - i_lo + 1 + G_NCH - 1) / G_NCH;
    const int c0 = min(i_lo + ch * CR, i_hi + 1), c1 = min(c0 + CR, i_hi + 1);
    int D = skew_run(guf, nullptr, W, nU, u, c0, c1, BIG);
    s_end[ch][lane] = D;
    __syncthreads();
    // D just before row c0: chain the chunk ends (a line is "on" for one contiguous row range, and an
    // "off" row resets D to BIG exactly as in the local scans)
    int K = BIG;
    for (int c = 0; c < ch; ++c) {
        const int len = min(i_lo + (c + 1) * CR, i_hi + 1) - min(i_lo + c * CR, i_hi + 1);
        K = min(s_end[c][lane], min(K + 3 * len, BIG));
        // a line that was off at the end of chunk c has s_end == BIG and, being contiguous, was never on
        // before: K + 3 len stays >= BIG only if K was BIG -- which holds, because any earlier on-rows
        // would make the line on at the end of chunk c as well (it leaves the image only at its last row)
    }
    skew_run(guf, dBf, W, nU, u, c0, c1, K);
}

// k_rowscan: one wave per image row, 8 consecutive pixels per lane (one 16-byte load per array), 512
// pixels per segment.  With a(j) = min_{k<=j} g(k) + (j-k) and dA likewise from gu (left-to-right), then
// d(j) = min_{k>=j} a(k) + (k-j) over a itself (right-to-left; a <= g and a(k)+(k-j) is a real path length,
// so this equals the two-sided minimum over g):  live = (dA == d) or (dB == d).
// Inside a lane the scans are sequential (8 steps); across lanes ONE wave scan of the lane totals per
// segment and quantity; across segments a wave-uniform carry.  The left-to-right results wait in a
// per-wave LDS row buffer.
__device__ __forceinline__ int wave_excl_prefix_min(int v, int lane) {  // min over lanes < lane (BIG for lane 0)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off);
        if (lane >= off) v = min(v, t);
    }
    const int e = __shfl_up(v, 1);
    return lane == 0 ? BIG : e;
}
__device__ __forceinline__ int wave_excl_suffix_min(int v, int lane) {  // min over lanes > lane (BIG for lane 63)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_down(v, off);
        if (lane + off < 64) v = min(v, t);
    }
    const int e = __shfl_down(v, 1);
    return lane == 63 ? BIG : e;
}

// 8 consecutive uint16 of a row starting at element idx0 (multiple of 8); vectorised when the row is 16-byte aligned
__device__ __forceinline__ void load8(const u16 *__restrict__ row, int idx0, int W, bool vec, int (&v)[8]) {
    if (vec && idx0 + 8 <= W) {
        const uint4 q = *reinterpret_cast<const uint4 *>(row + idx0);
        v[0] = q.x & 0xFFFF; v[1] = q.x >> 16; v[2] = q.y & 0xFFFF; v[3] = q.y >> 16;
        v[4] = q.z & 0xFFFF; v[5] = q.z >> 16; v[6] = q.w & 0xFFFF; v[7] = q.w >> 16;
    } else {
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = idx0 + q < W ? (int)row[idx0 + q] : INF16;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = v[q] == INF16 ? BIG : v[q];
}

__global__ __launch_bounds__(256) void k_rowscan(const u16 *__restrict__ g, const u16 *__restrict__ gu,
                                                 const u16 *__restrict__ dB, const int *__restrict__ fflag,
                                                 int H, int W, int nseg, u16 *__restrict__ dl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * (blockDim.x >> 6) + wave, b = blockIdx.y;
    if (!fflag[b] || i >= H) return;  // wave-uniform; no block-level barrier below
    int *s_a = reinterpret_cast<int *>(smem) + (size_t)wave * nseg * 512;  // a | (dA == a) << 24, lane-private slots
    const size_t ro = ((size_t)b * H + i) * W;
    const u16 *grow = g + ro, *gurow = gu + ro, *dBrow = dB + ro;
    const bool vec = (W & 7) == 0;  // rows start 16-byte aligned (the arrays are 256-byte aligned)

    int carry_a = BIG, carry_dA = BIG;  // min of (value - index) over everything left of the segment
    for (int sg = 0; sg < nseg; ++sg) {
        const int idx0 = sg * 512 + lane * 8;
        int gv[8], uv[8];
        load8(grow, idx0, W, vec, gv);
        load8(gurow, idx0, W, vec, uv);
        int ma = BIG, md = BIG, la[8], ld[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            ma = min(ma, gv[q] - (idx0 + q));
            md = min(md, uv[q] - (idx0 + q));
            la[q] = ma;
            ld[q] = md;
        }
        const int ea = min(wave_excl_prefix_min(ma, lane), carry_a);
        const int ed = min(wave_excl_prefix_min(md, lane), carry_dA);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int a = min(min(la[q], ea) + idx0 + q, BIG);
            const int dA = min(min(ld[q], ed) + idx0 + q, BIG);
            s_a[sg * 512 + q * 64 + lane] = a | (dA == a ? 1 << 24 : 0);  // [q][lane]: conflict-free, lane-private
        }
        carry_a = __shfl(min(ma, ea), 63);
        carry_dA = __shfl(min(md, ed), 63);
    }
    int carry_b = BIG;  // min of (a + index) over everything right of the segment
    for (int sg = nseg - 1; sg >= 0; --sg) {
        const int idx0 = sg * 512 + lane * 8;
        int av[8], fl[8], ms = BIG, ls[8];
#pragma unroll
        for (int q = 7; q >= 0; --q) {
            const int v = s_a[sg * 512 + q * 64 + lane];
            av[q] = v & 0xFFFFFF;
            fl[q] = v >> 24;
            ms = min(ms, av[q] + idx0 + q);
            ls[q] = ms;
        }
        const int es = min(wave_excl_suffix_min(ms, lane), carry_b);
        int dbv[8];
        load8(dBrow, idx0, W, vec, dbv);
        u32 out[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int d = min(ls[q], es) - (idx0 + q);
            const bool live = (fl[q] && av[q] == d) || dbv[q] == d;
            out[q] = d >= DL_NONE ? (u32)DL_NONE : (u32)(d | (live ? DL_LIVE : 0));  // DL_NONE: no source in the frame
        }
        carry_b = __shfl(min(ms, es), 0);
        if (vec && idx0 + 8 <= W) {
            uint4 o;
            o.x = out[0] | out[1] << 16; o.y = out[2] | out[3] << 16; o.z = out[4] | out[5] << 16; o.w = out[6] | out[7] << 16;
            *reinterpret_cast<uint4 *>(dl + ro + idx0) = o;
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (idx0 + q < W) dl[ro + idx0 + q] = (u16)out[q];
        }
    }
}

constexpr int G_PPT = 16;  // pixels per thread in k_final (keeps its no-op grid small)

// k_exit: one block per 128 x 128 tile.  Loads the tile of dl (d | live<<15) with a 2-cell halo into LDS,
// applies the 5x5 parent rule there, and resolves the chains inside the tile by pointer doubling in LDS
// (every cell does the same work each round: no divergent walks, and the number of rounds is log2 of the
// longest in-tile chain, whatever the distances are).  A cell is terminal if it is a source, has no
// parent, or its parent lies outside the tile.  Result per pixel: an exit pointer
//   bit 31 set : the chain's root source, pixel index in the low bits
//   0x7FFFFFFF : no source in the frame
//   otherwise  : pixel index (another tile) where the chain continues
// Also stores the float distance map (the last consumer of d).
constexpr int X_T = 128;             // tile edge
constexpr int X_NT = 1024;           // threads per block
constexpr int X_P = X_T + 4;         // dl tile pitch (2-cell halo each side)
constexpr u32 X_ROOT = 0x80000000u;  // exit pointer: resolved to a root
constexpr u32 X_NONE = 0x7FFFFFFFu;  // exit pointer: frame without sources
constexpr int X_BORDER = 0x3FF0;     // dl value outside the image: (v & mask) + w never equals a d | live<<15

__global__ __launch_bounds__(X_NT) void k_exit(const u16 *__restrict__ dl, const int *__restrict__ fflag, int H,
                                              int W, int tiles_x, u32 *__restrict__ exitp,
                                              float *__restrict__ out_dt, int stop_after) {
    __shared__ __attribute__((aligned(16))) u16 s_big[X_P * X_P];  // dl tile + halo; later the pointers (X_T*X_T)
    __shared__ u8 s_code[X_T * X_T];
    const int b = blockIdx.y;
    if (!fflag[b]) return;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
    const int r0 = ty * X_T, c0 = tx * X_T;
    const size_t fo = (size_t)b * H * W;
    const u16 *dlf = dl + fo;
    const int tid = threadIdx.x;

    {   // tile + halo: one wave per row, lanes along the row (coalesced, no divisions); the loads of 8 rows
        // (24 per lane) are issued before the first LDS store, so the memory round trips overlap
        const int lane = tid & 63, wave = tid >> 6;
        for (int rb = wave; rb < X_P; rb += (X_NT / 64) * 8) {
            u16 v[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + (X_NT / 64) * u;
                const int gi = r0 + r - 2;
                const bool rin = r < X_P && gi >= 0 && gi < H;
                const u16 *src = dlf + (size_t)(rin ? gi : 0) * W;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = lane + 64 * q, gj = c0 + c - 2;
                    v[u][q] = (rin && c < X_P && gj >= 0 && gj < W) ? src[gj] : (u16)X_BORDER;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rb + (X_NT / 64) * u;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = lane + 64 * q;
                    if (r < X_P && c < X_P) s_big[r * X_P + c] = v[u][q];
                }
            }
        }
    }
    __syncthreads();
    if (stop_after == 0) return;  // timing-only (DTFILL_EXIT_STOP)
    // parent rule, straight-line: tap t forward for live cells, the negated tap for the others; keep the
    // code format of tap_decode (t | backward << 3)
    for (int k = tid; k < X_T * X_T; k += X_NT) {
        const int r = k >> 7, c = k & (X_T - 1);
        const u16 *p = s_big + (r + 2) * X_P + c + 2;
        const int v = *p;
        const int d = v & DL_DMASK;
        const int live = v >> 15;
        const int sgn = live ? 1 : -1;
        const int msk = live ? 0xFFFF : DL_DMASK;
        int t_sel = -1;
#pragma unroll
        for (int t = 7; t >= 0; --t) {  // descending: the FIRST matching tap is kept
            const int nv = p[sgn * (TAP_DI(t) * X_P + TAP_DJ(t))];
            t_sel = ((nv & msk) + TAP_W(t) == v) ? t : t_sel;
        }
        int code = t_sel < 0 ? PAR_NONE : (t_sel | (live ? 0 : 8));
        code = d == DL_NONE ? PAR_NONE : code;
        code = d == 0 ? PAR_SRC : code;
        s_code[k] = (u8)code;
        const int gi = r0 + r, gj = c0 + c;
        if (out_dt && gi < H && gj < W)
            out_dt[fo + (size_t)gi * W + gj] = d == DL_NONE ? 8192.0f : (float)d;  // float(INIT_DIST0 * 2^-16) == 8192.0f
    }
    __syncthreads();
    if (stop_after == 1) return;
    u16 *s_ptr = s_big;  // the dl tile is dead
    for (int k = tid; k < X_T * X_T; k += X_NT) {
        const int r = k >> 7, c = k & (X_T - 1);
        const int code = s_code[k];
        int di, dj;
        tap_decode(code, di, dj);
        const u32 nr = (u32)(r + di), nc = (u32)(c + dj);
        const bool inside = code < 16 && nr < (u32)X_T && nc < (u32)X_T;
        s_ptr[k] = inside ? (u16)(nr * X_T + nc) : (u16)(k | 0x8000);
    }
    __syncthreads();
    if (stop_after == 2) return;
    // pointer doubling.  Each thread owns cells tid + 256 j and keeps the still-open ones as bits, so late
    // rounds only touch what is left; two jumps per round.  Any pointer value read here is an ancestor of the
    // cell (other threads only ever replace a pointer by a farther ancestor): races just speed things up.
    {
        u64 open = 0;
#pragma unroll 8
        for (int j = 0; j < X_T * X_T / X_NT; ++j) open |= (u64)(!(s_ptr[tid + X_NT * j] & 0x8000)) << j;
        for (int round = 0; round < 16; ++round) {  // 4^16 > any in-tile chain
            u64 m = open;
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const int k = tid + X_NT * j;
                int q = s_ptr[s_ptr[k]];  // s_ptr[k] has no flag: k is open
                if (!(q & 0x8000)) q = s_ptr[q];
                s_ptr[k] = (u16)q;
                if (q & 0x8000) open &= ~(1ull << j);
            }
            if (!__syncthreads_or(open != 0)) break;
        }
    }
    if (stop_after == 3) return;
    for (int k = tid; k < X_T * X_T; k += X_NT) {
        const int r = k >> 7, c = k & (X_T - 1);
        const int gi = r0 + r, gj = c0 + c;
        if (gi >= H || gj >= W) continue;
        const int t = s_ptr[k] & 0x3FFF;  // terminal cell of k's in-tile chain
        const int code = s_code[t];
        const int tr = r0 + (t >> 7), tc = c0 + (t & (X_T - 1));
        u32 e;
        if (code == PAR_SRC) {
            e = X_ROOT | (u32)(tr * W + tc);
        } else if (code >= 16) {
            e = X_NONE;
        } else {
            int di, dj;
            tap_decode(code, di, dj);
            e = (u32)min(max((tr + di) * W + tc + dj, 0), H * W - 1);  // clamp: a logic error must not become a wild access
        }
        exitp[fo + (size_t)gi * W + gj] = e;
    }
}

// k_final: follow the exit pointers from tile to tile (a chain crosses few tiles), then label, gather, store.
__global__ __launch_bounds__(256) void k_final(
    const float *__restrict__ x, const u32 *__restrict__ exitp,
    const u64 *__restrict__ srcbits, const u16 *__restrict__ wpre_s, const u32 *__restrict__ rowbase_s,
    const int *__restrict__ finfo, const float *__restrict__ vlist, const int *__restrict__ fflag, int H,
    int W, int Wd, float *__restrict__ out_depth, int32_t *__restrict__ out_index,
    int *__restrict__ frame_status) {
    const int b = blockIdx.y;
    if (!fflag[b]) return;
    const size_t fo = (size_t)b * H * W;
    const u32 *ef = exitp + fo;
    const int N1 = H * W;
    const int nval = finfo[b * FI_STRIDE + FI_NVAL], misaligned = finfo[b * FI_STRIDE + FI_MISALIGNED];
    constexpr int FB = 8;  // pixels per lane whose loads are in flight together
    for (int pb = blockIdx.x * (256 * G_PPT) + threadIdx.x; pb < min(N1, (int)(blockIdx.x + 1) * 256 * G_PPT);
         pb += 256 * FB) {
        u32 e[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) e[u] = ef[min(pb + 256 * u, N1 - 1)];
        for (int hop = 0; hop < MAX_HW_SUM; ++hop) {  // tile-to-tile hops, all FB chains in lock-step
            bool open = false;
#pragma unroll
            for (int u = 0; u < FB; ++u) {
                const bool mv = !(e[u] & X_ROOT) && e[u] != X_NONE;
                const u32 nx = ef[mv ? e[u] : 0u];
                e[u] = mv ? nx : e[u];
                open |= mv;
            }
            if (!__any(open)) break;
        }
        int label[FB], q[FB];
        u32 base[FB];
        u64 word[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const bool root = e[u] & X_ROOT;
            q[u] = root ? (int)(e[u] & ~X_ROOT) : 0;
            const int i = q[u] / W;
            const size_t w = ((size_t)b * H + i) * Wd + ((q[u] - i * W) >> 6);
            base[u] = rowbase_s[(size_t)b * H + i] + wpre_s[w];
            word[u] = srcbits[w];
        }
        float val[FB];
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int i = q[u] / W;
            label[u] = (e[u] & X_ROOT) ? source_rank(base[u], word[u], q[u] - i * W) : 0;
            const int p = pb + 256 * u;
            val[u] = (out_depth && p < N1) ? gather_depth(x + fo, vlist + fo, label[u], q[u], nval, misaligned, frame_status + b)
                                          : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < FB; ++u) {
            const int p = pb + 256 * u;
            if (p >= N1) continue;
            if (out_index) out_index[fo + p] = label[u];
            if (out_depth) out_depth[fo + p] = val[u];
        }
    }
}

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

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
inline size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

struct Carve {
    u16 *gu, *g, *dB, *dl;
    u32 *exitp;
    u64 *srcbits, *valbits;
    u16 *wpre_s, *wpre_v;
    u32 *rowcnt_s, *rowcnt_v, *rowbase_s, *rowbase_v;
    int *finfo, *fflag, *fflag2, *status;
    float *vlist;
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
    c.fflag = (int *)take((size_t)B * 4);
    c.fflag2 = (int *)take((size_t)B * 4);
    c.status = (int *)take((size_t)B * 4);
    // general-path arrays (touched only for frames the fused kernel flags)
    c.gu = (u16 *)take(N * 2);
    c.g = (u16 *)take(N * 2);
    c.dB = (u16 *)take(N * 2);
    c.dl = (u16 *)take(N * 2);
    c.exitp = (u32 *)take(N * 4);
    c.vlist = (float *)take(N * 4);
    c.total = off;
    return c;
}

bool shape_ok(int B, int H, int W) {
    return B >= 1 && H >= 1 && W >= 1 && (long long)H + W - 2 < MAX_HW_SUM &&
           (long long)B * H * W < (1ll << 31);
}

constexpr int NK_L1 = 8;
const char *const kNamesL1[NK_L1] = {"k_mask", "k_frame",   "k_fused", "k_colscan",
                                     "k_skew", "k_rowscan", "k_exit",  "k_final"};

int run_l1(const float *x, int B, int H, int W, float src_thr, float val_thr, float *out_depth,
           float *out_dt, int32_t *out_index, int32_t *frame_status, void *workspace, unsigned flags,
           hipStream_t st, hipEvent_t *ev) {
    const Carve c = carve(workspace, B, H, W);
    const int Wd = (W + 63) / 64;
    const int N1 = H * W;
    int *status = frame_status ? frame_status : c.status;
    const bool general_only = flags & DTFILL_FLAG_GENERAL_ONLY;
    const bool fused_only = flags & DTFILL_FLAG_FUSED_ONLY;
    // debug: DTFILL_FUSED_STOP=n makes k_fused return after phase n (timing only, outputs undefined)
    const char *stop_env = ev ? getenv("DTFILL_FUSED_STOP") : nullptr;
    const int fused_stop = stop_env ? atoi(stop_env) : -1;
    const char *xstop_env = ev ? getenv("DTFILL_EXIT_STOP") : nullptr;
    const int exit_stop = xstop_env ? atoi(xstop_env) : -1;
    int k = 0;
    auto mark = [&]() {
        if (ev) (void)hipEventRecord(ev[k++], st);
    };
    mark();
    k_mask<<<dim3((H + 4 * M_RPW - 1) / (4 * M_RPW), B), 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits,
                                                c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v);
    mark();
    k_frame<<<B, 256, 0, st>>>(x, c.valbits, c.wpre_v, c.rowcnt_s, c.rowcnt_v, H, W, Wd, c.rowbase_s,
                               c.rowbase_v, c.finfo, c.vlist, c.fflag, c.fflag2, status, general_only ? 1 : 0);
    mark();
    if (!general_only) {
        // stage 1: halo 16 (tiles up to 96 x 160); stage 2, only for frames stage 1 flagged: halo 32
        {
            constexpr int R = 16, THM = F_WHM - 2 * R, TWM = F_WWM - 2 * R;
            const int nty = (H + THM - 1) / THM, ntx = (W + TWM - 1) / TWM;
            const int TH = (H + nty - 1) / nty, TW = (W + ntx - 1) / ntx;  // even split
            k_fused<R><<<dim3(ntx * nty, B), F_NT, 0, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist,
                                                           H, W, Wd, TH, TW, ntx, out_depth, out_dt, out_index,
                                                           nullptr, c.fflag, status, fused_stop);
        }
        {
            constexpr int R = 32, THM = F_WHM - 2 * R, TWM = F_WWM - 2 * R;
            const int nty = (H + THM - 1) / THM, ntx = (W + TWM - 1) / TWM;
            const int TH = (H + nty - 1) / nty, TW = (W + ntx - 1) / ntx;
            k_fused<R><<<dim3(ntx * nty, B), F_NT, 0, st>>>(x, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist,
                                                           H, W, Wd, TH, TW, ntx, out_depth, out_dt, out_index,
                                                           c.fflag, c.fflag2, status, -1);
        }
    }
    mark();
    if (!fused_only) {
        k_colscan<false><<<dim3(Wd, B), 64 * G_NCH, 0, st>>>(c.srcbits, c.fflag2, H, W, Wd, (H + G_NCH - 1) / G_NCH, c.gu, c.g);
        mark();
        const int nU = W + 2 * (H - 1) + 1;
        k_skew<<<dim3((nU + 63) / 64, B), 64 * G_NCH, 0, st>>>(c.gu, c.fflag2, H, W, c.dB);
        mark();
        const int nseg = (W + 511) / 512;
        const size_t per_wave = (size_t)nseg * 512 * sizeof(int);  // <= 32 KiB at W = 8191
        const int wpb = (int)max((size_t)1, min((size_t)4, (size_t)65536 / per_wave));
        k_rowscan<<<dim3((H + wpb - 1) / wpb, B), 64 * wpb, wpb * per_wave, st>>>(c.g, c.gu, c.dB, c.fflag2, H, W, nseg,
                                                                              c.dl);
        mark();
        {
            const int etx = (W + X_T - 1) / X_T, ety = (H + X_T - 1) / X_T;
            k_exit<<<dim3(etx * ety, B), X_NT, 0, st>>>(c.dl, c.fflag2, H, W, etx, c.exitp, out_dt, exit_stop);
        }
        mark();
        k_final<<<dim3((N1 + 256 * G_PPT - 1) / (256 * G_PPT), B), 256, 0, st>>>(
            x, c.exitp, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, c.fflag2, H, W, Wd, out_depth, out_index,
            status);
        mark();
    } else {
        for (int t = 0; t < 5; ++t) mark();
    }
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

constexpr int NK_L2 = 4;
const char *const kNamesL2[NK_L2] = {"k_mask", "k_frame", "k_colscan", "k_l2row"};

int run_l2(const float *x, int B, int H, int W, float src_thr, float val_thr, float *out_depth, float *out_dt,
           int32_t *out_index, int32_t *frame_status, void *workspace, hipStream_t st, hipEvent_t *ev) {
    const Carve c = carve(workspace, B, H, W);
    const int Wd = (W + 63) / 64;
    int *status = frame_status ? frame_status : c.status;
    int k = 0;
    auto mark = [&]() {
        if (ev) (void)hipEventRecord(ev[k++], st);
    };
    mark();
    k_mask<<<dim3((H + 4 * M_RPW - 1) / (4 * M_RPW), B), 256, 0, st>>>(x, H, W, Wd, src_thr, val_thr, c.srcbits, c.valbits,
                                                                      c.wpre_s, c.wpre_v, c.rowcnt_s, c.rowcnt_v);
    mark();
    k_frame<<<B, 256, 0, st>>>(x, c.valbits, c.wpre_v, c.rowcnt_s, c.rowcnt_v, H, W, Wd, c.rowbase_s, c.rowbase_v,
                               c.finfo, c.vlist, c.fflag, c.fflag2, status, 0);
    mark();
    k_colscan<true><<<dim3(Wd, B), 64 * G_NCH, 0, st>>>(c.srcbits, nullptr, H, W, Wd, (H + G_NCH - 1) / G_NCH, c.gu, c.g);
    mark();
    k_l2row<<<dim3((H * W + 255) / 256, B), 256, 0, st>>>(x, c.g, c.srcbits, c.wpre_s, c.rowbase_s, c.finfo, c.vlist, H, W,
                                                         Wd, out_depth, out_dt, out_index, status);
    mark();
    return hipGetLastError() == hipSuccess ? DTFILL_OK : DTFILL_ERR_LAUNCH;
}

int check_args(const float *x, int B, int H, int W, int metric, float *out_depth, float *out_dt,
               int32_t *out_index, void *workspace, size_t ws_bytes) {
    if (!x || !workspace || (!out_depth && !out_dt && !out_index)) return DTFILL_ERR_NULL;
    if (!shape_ok(B, H, W)) return DTFILL_ERR_SHAPE;
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
        case DTFILL_ERR_SHAPE: return "bad shape: need B,H,W >= 1, H+W-2 < 8192 and B*H*W < 2^31";
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
    int rc = check_args(x, B, H, W, metric, out_depth, out_dt, out_index, workspace, ws_bytes);
    if (rc != DTFILL_OK) return rc;
    if (metric == DTFILL_METRIC_L2)
        return run_l2(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace,
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

int dtfill_outlier_removal(const float *x, int B, int H, int W, float *out, void *stream) {
    if (!x || !out) return DTFILL_ERR_NULL;
    if (B < 1 || H < 4 || W < 4 || (long long)B * H * W >= (1ll << 31) || B > 65535) return DTFILL_ERR_SHAPE;
    k_outlier<<<dim3((W + O_TW - 1) / O_TW, (H + O_TH - 1) / O_TH, B), 256, 0, static_cast<hipStream_t>(stream)>>>(
        x, H, W, out);
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
    int rc = check_args(x, B, H, W, metric, out_depth, out_dt, out_index, workspace, ws_bytes);
    if (rc != DTFILL_OK) return rc;
    if (!kernel_ms) return DTFILL_ERR_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nk = dtfill_num_kernels(metric);
    hipEvent_t ev[NK_L1 + 1];
    for (int k = 0; k <= nk; ++k)
        if (hipEventCreate(&ev[k]) != hipSuccess) return DTFILL_ERR_NO_DEVICE;
    rc = metric == DTFILL_METRIC_L2
             ? run_l2(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, st, ev)
             : run_l1(x, B, H, W, src_thr, val_thr, out_depth, out_dt, out_index, frame_status, workspace, flags, st,
                      ev);
    (void)hipEventSynchronize(ev[nk]);
    for (int k = 0; k < nk; ++k) (void)hipEventElapsedTime(&kernel_ms[k], ev[k], ev[k + 1]);
    for (int k = 0; k <= nk; ++k) (void)hipEventDestroy(ev[k]);
    return rc;
}

}  // extern "C"
