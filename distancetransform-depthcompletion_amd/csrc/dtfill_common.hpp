// dtfill_common.hpp -- constants, tap tables and small device helpers shared by every kernel
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

constexpr int BIG = 1 << 20;       // in-register "infinite" distance
constexpr int INF16 = 0xFFFF;      // stored "infinite" distance in the uint16 scan arrays
constexpr int MAX_HW_SUM = 8192;   // cv2's Q16 INIT_DIST0 = INT_MAX>>2 caps distances at 8191

// frame facts written by k_frame: int32[FI_STRIDE] per frame
constexpr int FI_NSRC = 0, FI_NVAL = 1, FI_MISALIGNED = 2, FI_DLB = 3;  // DLB: lower bound of the largest distance (empty rows)
constexpr int FI_NUNRES = 4;  // tie pixels k_fin handed to k_tiesx (zeroed by k_frame)
constexpr int FI_SKY = 5;     // l1_cv: rows [0, FI_SKY) hold no source and lie above every source: k_sky's (0: none, or called off)
constexpr int FI_SKY0 = 6;    // ... as k_frame set it.  k_fused calls the sky off (FI_SKY = 0) when it has to hand on one of the two rows
                              // k_sky would start from (FI_SKY0, FI_SKY0 + 1): the sky's rows then count as flagged 1
constexpr int FI_TR0 = 7;     // l1_cv, a window kernel's frame: the first row of its tiling (the rows above are the sky's: the tile rows split
                              // the rest evenly, so that no tile row is spent on rows that are not the window's)
constexpr int FI_STRIDE = 8;
constexpr int ROUTE_POINTS = -1;  // route[b]: l2, at most L2_PTS_MAX sources in the frame (k_l2pts)
constexpr int L2_PTS_MAX = 512;
constexpr int W2_R16 = 10, W2_R32 = 15;  // l2: window radius of k_l2win for the frames k_frame routes 16 / 32
constexpr u32 L2_ROW_GONE = 0x40000000u;  // l2 row flag: handed to the row search up front (above any far-pixel count's threshold)
struct PtsSrc {  // one source of such a frame's list (k_frame writes it in raster order: index = label - 1; k_pts reads it)
    u32 rc;      // row << 16 | column
    float v;     // its depth
};
constexpr int PTS_BAND_MAX = 96;  // l1_cv: ... and at most this many in any band of 32 rows
// route[b] | ROUTE_PREMARK (l1_cv): k_frame marked rows of the frame for the any-distance kernels up front (rows farther than
// PM16 / PM32 from every row that holds a source: the window kernel with halo 16 / 32 could not decide them, or only by running
// its level loop to the end for a handful of their pixels)
constexpr int ROUTE_PREMARK = 0x100;
constexpr int PM16 = 8, PM32 = 16;
constexpr int SKY_MAX = 320;  // ... and up to this many (k_sky's blocks stage r0 + 260 columns in LDS, in k_colT's launch)
constexpr int SKY_MIN = 9;  // rows above the first source row are k_sky's from this many on (fewer: within every window's reach)

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


// What the reference's drivers do to the filled depth right after the fill, folded into the depth stores:
// rows [row0, H) only (demo.py:292-293: lidar_batch[:, 96:]) and / or the depth floor relu(d - floor) + floor in
// float32 with both roundings (eval_NYU.py:205, test.py:133).  out_depth is then [B, H - row0, W].
struct DepthEpilogue {
    int row0;
    int use_floor;
    float floor_;
};
__device__ __forceinline__ float depth_epilogue(float d, const DepthEpilogue &ep) {
    return ep.use_floor ? __fadd_rn(fmaxf(__fsub_rn(d, ep.floor_), 0.0f), ep.floor_) : d;
}

// The per-row flags (l1_cv: "k_fused left a pixel of this row undecided"; l2: far pixels counted by k_l2win) live in the
// workspace right behind the per-frame flags (carve(): fflag2, then rowfar), so that k_fused needs no pointer of its own
// for something it touches once in a blue moon.  B = frames of the batch (gridDim.y of every kernel here).
__host__ __device__ inline size_t rowflag_offset_bytes(int B) { return ((size_t)B * 4 + 255) & ~(size_t)255; }
__device__ __forceinline__ u32 *rowflag_of(int *fflag, int B) {
    return reinterpret_cast<u32 *>(reinterpret_cast<char *>(fflag) + rowflag_offset_bytes(B));
}

// row flag f of a frame whose sky is / is not k_sky's: is the row one of the any-distance kernels'?
__device__ __forceinline__ bool row_is_anydist(u32 f, int sky_live) { return f == 1u || (f == 2u && !sky_live); }

// Rank records (k_colT -> k_fin, k_l2env): per 64-pixel word of a row {low 32 source bits, sources before them in the frame,
// high 32 bits, sources before those}; records of one word column are consecutive in the row index.  The label of the source
// at (si, sj) = 1 + its raster rank among the frame's sources (cv2's label init) = one 8-byte read + a popcount.
__device__ __forceinline__ int label_from_rec(const uint2 *__restrict__ rec_f /* the frame's records */, int H, int si, int sj) {
    const uint2 r = rec_f[(((size_t)(sj >> 6) * H + si) << 1) + ((sj >> 5) & 1)];
    return (int)r.y + __popc(r.x & ((1u << (sj & 31)) - 1u)) + 1;
}
