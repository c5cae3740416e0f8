// dtfill_prepass.hpp -- k_mask, k_frame: bit words, compaction ranks, frame facts (every pass starts with these)
// Part of libdtfill.so; included by dtfill.hip inside its anonymous namespace (one translation unit).
#pragma once

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
// k_mask4: the same outputs for rows whose pixels can be read 16 bytes at a time (W % 4 == 0, 16-byte aligned
// frames): one wave per row, a lane reads 4 consecutive pixels, 256 pixels = four 64-pixel words per load
// instruction, and all loads of a row (up to M4_NC at a time) are in flight together.  A lane's four predicate
// bits form a nibble; the 16 lanes of a DPP row hold one word, combined with four DPP OR steps.
// ------------------------------------------------------------------------------------------------
constexpr int M4_NC = 8;  // 256-pixel chunks per batch (2048 pixels)

template <int CTRL>
__device__ __forceinline__ u32 dpp_or(u32 v) {  // v | v of the lane CTRL pairs it with (inside a row of 16)
    return v | (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
// the 64-pixel word of this lane's 16-lane row from the lanes' nibbles (lane i of the row owns bits 4i..4i+3)
__device__ __forceinline__ u64 row_word(u32 nib, int lane) {
    u32 part = nib << (4 * (lane & 7));  // lanes 0-7 of the row build the low half, lanes 8-15 the high half
    part = dpp_or<0xB1>(part);           // quad_perm [1,0,3,2]
    part = dpp_or<0x4E>(part);           // quad_perm [2,3,0,1]
    part = dpp_or<0x141>(part);          // row_half_mirror: now every lane has its half-row's 32 bits
    const u32 other = (u32)__builtin_amdgcn_update_dpp(0, (int)part, 0x140, 0xF, 0xF, false);  // row_mirror
    return (lane & 8) ? ((u64)part << 32 | other) : ((u64)other << 32 | part);
}

// OM != 0: outlier_removal() (data_read.py:103-128) in front of the predicates, see dtfill_outlier.hpp: the candidates of the
// 2048 pixels in registers (v > 1.0; every pixel when OM == 2) are listed in LDS, one lane each gathers a candidate's 25 taps,
// and the pixels it removes read as 0.0f below.  OM == 1 raises negflag[b] on a negative value, the OM == 2 launch redoes
// exactly those frames.
__device__ __forceinline__ bool outlier_at(const float *__restrict__ xf, int H, int W, int i, int j, float v);
template <int OM, int R>  // R image rows per wave, the loads of all of them in flight before the first is worked on (R = 2 measured no
                          // faster than 1 on the KITTI batch: the kernel is not short of bytes in flight; only R = 1 is launched)
__global__ __launch_bounds__(256) void k_mask4(const float *__restrict__ x, int H, int W, int Wd, float src_thr,
                                               float val_thr, u64 *__restrict__ srcbits, u64 *__restrict__ valbits,
                                               u16 *__restrict__ wpre_s, u16 *__restrict__ wpre_v,
                                               u32 *__restrict__ rowcnt_s, u32 *__restrict__ rowcnt_v, int *__restrict__ negflag) {
    __shared__ u16 s_list[OM ? 4 : 1][OM ? M4_NC * 256 : 1];
    __shared__ u32 s_drop[OM ? 4 : 1][OM ? M4_NC * 8 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // (frames along grid x: with a batch of a multiple of eight frames a frame's rows are one XCD's, where the window kernel's blocks of
    // that frame will look for its bit words)
    const int i0 = (blockIdx.y * 4 + wave) * R, b = blockIdx.x;
    if (i0 >= H) return;
    if (OM == 2 && !negflag[b]) return;
    bool neg = false;
    const int w_in_chunk = lane >> 4;  // which of the chunk's four words this lane's row builds
    const int nchunk = (W + 255) >> 8;
    // rows of at most M4_NC chunks (2048 pixels) with R > 1: the first batch of every row is loaded up front
    float4 pre[R][M4_NC];
    if (R > 1) {
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const float4 *rowp = reinterpret_cast<const float4 *>(x + ((size_t)b * H + min(i0 + rr, H - 1)) * W);
#pragma unroll
            for (int u = 0; u < M4_NC; ++u) {
                const int px = (u << 8) + 4 * lane;
                pre[rr][u] = (u < nchunk && px < W) ? rowp[px >> 2] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            }
        }
    }
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int i = i0 + rr;
    if (i >= H) break;  // wave-uniform
    const float4 *row = reinterpret_cast<const float4 *>(x + ((size_t)b * H + i) * W);
    const size_t wrow = ((size_t)b * H + i) * Wd;
    u32 run_s = 0, run_v = 0, mis = 0;  // wave-uniform
    for (int c0 = 0; c0 < nchunk; c0 += M4_NC) {
        float4 v[M4_NC];
#pragma unroll
        for (int u = 0; u < M4_NC; ++u) {
            const int px = ((c0 + u) << 8) + 4 * lane;
            if (R > 1 && c0 == 0)
                v[u] = pre[rr][u];
            else
                v[u] = (c0 + u < nchunk && px < W) ? row[px >> 2] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        if (OM) {
            if (lane < M4_NC * 8) s_drop[wave][lane] = 0u;
            int n = 0;  // wave-uniform
#pragma unroll
            for (int u = 0; u < M4_NC; ++u) {
                if (c0 + u >= nchunk) break;
                const int px = ((c0 + u) << 8) + 4 * lane;
                const float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    neg |= px < W && f[q] < 0.0f;
                    const bool cand = px < W && (OM == 2 || f[q] > 1.0f);
                    const u64 bal = __ballot(cand);
                    if (cand) s_list[wave][n + (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, 0u))] = (u16)((u << 8) + 4 * lane + q);
                    n += __popcll(bal);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const float *xf = x + (size_t)b * H * W;
            for (int t = lane; t < n; t += 64) {
                const int jr = s_list[wave][t], j = (c0 << 8) + jr;
                if (outlier_at(xf, H, W, i, j, xf[(size_t)i * W + j])) atomicOr(&s_drop[wave][jr >> 5], 1u << (jr & 31));
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int u = 0; u < M4_NC; ++u) {
            if (c0 + u >= nchunk) break;  // wave-uniform
            const int px = ((c0 + u) << 8) + 4 * lane;
            const bool in = px < W;  // W % 4 == 0: a lane's four pixels are inside or outside together
            float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            if (OM) {
                const u32 d = s_drop[wave][(u << 3) + (lane >> 3)] >> ((4 * lane) & 31);
#pragma unroll
                for (int q = 0; q < 4; ++q) f[q] = ((d >> q) & 1u) ? 0.0f : f[q];
            }
            u32 ns = 0, nv = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ns |= (in && !((1.0f - f[q]) > src_thr)) ? (1u << q) : 0u;
                nv |= (in && (f[q] > val_thr)) ? (1u << q) : 0u;
            }
            const u64 ws = row_word(ns, lane), wv = row_word(nv, lane);
            const u32 cs = (u32)__popcll(ws), cv = (u32)__popcll(wv);
            // counts of the chunk's four words (every lane of a row holds its word's count)
            const u32 s0 = __builtin_amdgcn_readlane(cs, 0), s1 = __builtin_amdgcn_readlane(cs, 16),
                      s2 = __builtin_amdgcn_readlane(cs, 32), s3 = __builtin_amdgcn_readlane(cs, 48);
            const u32 v0 = __builtin_amdgcn_readlane(cv, 0), v1 = __builtin_amdgcn_readlane(cv, 16),
                      v2 = __builtin_amdgcn_readlane(cv, 32), v3 = __builtin_amdgcn_readlane(cv, 48);
            const u32 pre_s = run_s + (w_in_chunk > 0 ? s0 : 0u) + (w_in_chunk > 1 ? s1 : 0u) + (w_in_chunk > 2 ? s2 : 0u);
            const u32 pre_v = run_v + (w_in_chunk > 0 ? v0 : 0u) + (w_in_chunk > 1 ? v1 : 0u) + (w_in_chunk > 2 ? v2 : 0u);
            mis |= __any(ws != wv) ? 1u : 0u;
            const int k = ((c0 + u) << 2) + w_in_chunk;  // word index in the row
            if ((lane & 15) == 0 && k < Wd) {             // one lane per word: four adjacent words per store
                srcbits[wrow + k] = ws;
                valbits[wrow + k] = wv;
                wpre_s[wrow + k] = (u16)pre_s;
                wpre_v[wrow + k] = (u16)pre_v;
            }
            run_s += s0 + s1 + s2 + s3;
            run_v += v0 + v1 + v2 + v3;
        }
    }
    if (lane == 0) {
        rowcnt_s[(size_t)b * H + i] = run_s;
        rowcnt_v[(size_t)b * H + i] = run_v | (mis ? 0x80000000u : 0u);
    }
  }
    if (OM == 1 && __any(neg) && lane == 0) negflag[b] = 1;  // this frame is redone by the exhaustive launch
}

// ------------------------------------------------------------------------------------------------
// k_frame: one workgroup per frame.  Exclusive scan of the row counts = raster rank of the first
// source / value pixel of every row: cv2's label init (k=1; every zero pixel gets k++) and numpy's
// boolean compaction x[with_value] (tools.py:24).  The value list is only materialised when the two
// masks differ somewhere in the frame.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_frame(const float *__restrict__ x, const u64 *__restrict__ valbits,
                                               const u16 *__restrict__ wpre_v, const u64 *__restrict__ srcbits,
                                               const u16 *__restrict__ wpre_s, PtsSrc *__restrict__ ptslist,
                                               const u32 *__restrict__ rowcnt_s,
                                               const u32 *__restrict__ rowcnt_v, int H, int W, int Wd,
                                               u32 *__restrict__ rowbase_s, u32 *__restrict__ rowbase_v,
                                               int *__restrict__ finfo, float *__restrict__ vlist,
                                               int *__restrict__ fflag2,
                                               int *__restrict__ route, int *__restrict__ frame_status, int mode, int *__restrict__ negflag,
                                               u32 *__restrict__ rowfar, int nty16, int nty32) {
    const bool force_general = mode & 1;  // every frame takes the any-distance kernels (tests)
    const bool l2 = mode & 2;             // l2: the window kernel's cost does not grow with the distances it meets
    const bool premark = mode & 4;        // l1_cv without a depth epilogue: rows too far from every source row are handed on up front
    const bool pts_ok = mode & 8;         // l1_cv: a frame with a handful of sources may go to k_pts
    __shared__ u32 s_ws[4], s_wv[4];
    __shared__ int s_mis, s_dlb;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const u32 *cs_ = rowcnt_s + (size_t)b * H, *cv_ = rowcnt_v + (size_t)b * H;
    u32 *bs_ = rowbase_s + (size_t)b * H, *bv_ = rowbase_v + (size_t)b * H;
    __shared__ u32 s_empty[256];   // bit i: row i holds no source (the rows past H count as empty)
    __shared__ u32 s_far[2][256];  // bit i: row i >= r0 and vd(i) > PM16 / > PM32
    __shared__ int s_nfar[2], s_r0, s_route, s_bandmax, s_nrow;
    __shared__ u16 s_ptrow[L2_PTS_MAX];  // rows that hold a source (for the source list of a k_pts frame)
    const int Hp = (H + 63) & ~63;
    if (tid == 0) {
        s_mis = 0;
        s_dlb = 0;
        s_r0 = H;
        s_bandmax = 0;
        s_nrow = 0;
    }
    if (tid < 2) s_nfar[tid] = 0;
    for (int w = tid + (Hp >> 5); w < 256; w += 256) s_empty[w] = 0xFFFFFFFFu;
    __syncthreads();
    u32 run_s = 0, run_v = 0;
    int mis = 0;
    for (int base = 0; base < Hp; base += 256) {
        const int i = base + tid;
        u32 cs = 0, cv = 0;
        if (i < H) {
            cs = cs_[i];
            cv = cv_[i];
            mis |= (int)(cv >> 31);
            cv &= 0x7FFFFFFFu;
        }
        {
            // the same pass over the row counts: "row has no source" as bits (a wave holds 64 consecutive rows: two whole words per
            // ballot, no atomics), the first row with a source, the most sources in a band of 32 rows (k_pts's tiles are that high)
            const u64 has = __ballot(cs != 0u), bal = ~has;
            if (lane == 0 && i < Hp) {
                s_empty[i >> 5] = (u32)bal;
                s_empty[(i >> 5) + 1] = (u32)(bal >> 32);
                if (has) atomicMin(&s_r0, i + __ffsll((long long)has) - 1);
            }
            u32 c = cs;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) c += (u32)__shfl_xor((int)c, o);
            if ((lane & 31) == 0 && c) atomicMax(&s_bandmax, (int)c);
            // ... and the rows that hold a source, listed (in any order; at most L2_PTS_MAX matter: a frame with more of them
            // is no k_pts frame)
            int at = 0;
            if (lane == 0 && has) at = atomicAdd(&s_nrow, __popcll(has));
            at = __builtin_amdgcn_readfirstlane(at) + (int)__builtin_amdgcn_mbcnt_hi((u32)(has >> 32), __builtin_amdgcn_mbcnt_lo((u32)has, 0u));
            if (cs != 0u && at < L2_PTS_MAX) s_ptrow[at] = (u16)i;
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
    // Row structure of the frame (l1_cv uses all of it, l2 only FI_DLB):
    //   r0     the first row that holds a source.  The rows above it -- the empty sky of a LiDAR frame -- need no search at all:
    //          every source is below them, so cv2's backward sweep alone decides them, row by row from the two rows beneath
    //          (k_sky).  Flag 2.
    //   vd(i)  vertical distance of row i to the nearest row with a source: every pixel of row i is at least that far from
    //          every source.  max vd = lower bound of the largest distance in the frame (FI_DLB).  A row at or below r0 with
    //          vd above a window kernel's reach (PM16 / PM32) cannot be decided by it: it is handed to the any-distance kernels
    //          up front (flag 1), and the window kernel takes the rest of the frame instead of nothing.
    // "Row has no source" as bits in LDS (H <= 8191 -> 256 words); the rows past H count as empty.
    // (a frame in which every row holds a source -- the common dense one -- has vd = 0 everywhere: nothing to search)
    bool someempty = false;
    for (int w = tid; w <= (H - 1) >> 5; w += 256) someempty |= (s_empty[w] & (w == (H - 1) >> 5 ? (2u << ((H - 1) & 31)) - 1u : 0xFFFFFFFFu)) != 0u;
    someempty = __syncthreads_or(someempty);
    if (!someempty) {
        s_far[0][tid] = s_far[1][tid] = 0u;
    } else {
        int dlb = 0;
        const int lastw = (H - 1) >> 5, r0 = s_r0;
        for (int base = 0; base < Hp; base += 256) {
            const int i = base + tid;
            int vd = 0;
            if (i < H) {
                int up = BIG, dn = BIG;  // distance to the nearest row with a source at or above / at or below row i
                {
                    int w = i >> 5;
                    u32 m = ~s_empty[w] & ((2u << (i & 31)) - 1u);
                    while (!m && w > 0) m = ~s_empty[--w];
                    if (m) up = i - (w * 32 + 31 - __clz((int)m));
                }
                {
                    int w = i >> 5;
                    u32 m = ~s_empty[w] & ~((1u << (i & 31)) - 1u);
                    while (!m && w < lastw) m = ~s_empty[++w];
                    if (m) dn = w * 32 + __ffs((int)m) - 1 - i;
                }
                vd = min(up, dn);
                dlb = max(dlb, vd);
            }
            // l2: the rows farther from every row with a source than the window kernel's radius (no pixel of them has a source in its window)
            const u64 f16 = __ballot(i < H && (l2 ? vd > W2_R16 : (i >= r0 && vd > PM16))), f32 = __ballot(i < H && (l2 ? vd > W2_R32 : (i >= r0 && vd > PM32)));
            if (lane == 0 && i < Hp) {
                s_far[0][i >> 5] = (u32)f16;
                s_far[0][(i >> 5) + 1] = (u32)(f16 >> 32);
                s_far[1][i >> 5] = (u32)f32;
                s_far[1][(i >> 5) + 1] = (u32)(f32 >> 32);
                if (f16) atomicAdd(&s_nfar[0], __popcll(f16));
                if (f32) atomicAdd(&s_nfar[1], __popcll(f32));
            }
        }
#pragma unroll
        for (int o = 32; o; o >>= 1) dlb = max(dlb, __shfl_xor(dlb, o));
        if (lane == 0 && dlb) atomicMax(&s_dlb, dlb);
    }
    __syncthreads();
    const int misaligned = s_mis;
    const int r0 = s_r0;
    const bool sky_ok = premark && r0 >= SKY_MIN && r0 <= SKY_MAX && r0 < H;  // rows [0, r0) can be k_sky's
    if (tid == 0) {
        // Which kernel family takes the frame -- a speed heuristic, never a correctness condition (the window kernels hand on
        // every row in which they meet a pixel they cannot decide).  With source density p the chance that a pixel has no
        // source within L1 distance R is about (1-p)^(2 R^2 + 2 R + 1); if the frame is expected to hold such pixels all
        // over (N (1-p)^ball > ~1, i.e. p * ball < ln N ~ 14), a window kernel with halo R would do its work for nothing.
        // Runs of source-free rows do not say no (l1_cv): the density is judged on the rows that are left to the window.
        // Without the row flags (a depth epilogue: a handed-on row costs the whole frame there, k_fused's kept depths being
        // cropped / floored already; the forced paths of the tests) the old rule stays: a run of source-free rows that forces a
        // distance above R sends the frame to the any-distance kernels.  The l2 window kernel counts its far pixels itself.
        // route: halo 16 if it fits, else halo 32 (three times the work per pixel, still cheaper than the any-distance
        // kernels at a few percent density), else the any-distance kernels.
        auto fits = [&](int R, int nfar) {
            const long long ball = 2 * R * R + 2 * R + 1;
            if (l2) return (long long)run_s * ball >= 14ll * H * W;
            if (!premark) return (long long)run_s * ball >= 14ll * H * W && s_dlb <= R;
            const int rest = H - (sky_ok ? r0 : 0) - nfar;
            return rest > 0 && (long long)run_s * ball >= 14ll * rest * W;
        };
        // l2, a handful of sources (the NYU sampling patterns): ROUTE_POINTS -- every 32 x 32 tile looks at the sources that can
        // own one of its pixels, found by distance from the tile's centre (l2pts_tile); the blocks of k_l2win's launch
        // write the list of sources
        const bool points = l2 && !force_general && run_s > 0 && run_s <= (u32)L2_PTS_MAX;
        // l1_cv, a handful of sources and too thin for a window: k_pts (per tile, the sources whose cells reach it).  Not when they
        // crowd into one band of rows: its tiles would look at all of them for every pixel.
        const bool points1 = !l2 && pts_ok && run_s > 0 && run_s <= (u32)L2_PTS_MAX && s_bandmax <= PTS_BAND_MAX;
        const int window = fits(16, s_nfar[0]) ? 16 : fits(32, s_nfar[1]) ? 32 : 0;
        s_route = force_general ? 0 : points ? ROUTE_POINTS : window ? window : points1 ? ROUTE_POINTS : 0;
    }
    __syncthreads();
    const int r = s_route;
    if (!l2 && r == ROUTE_POINTS) {
        // the frame's sources in raster order (index = label - 1), for k_pts
        PtsSrc *list = ptslist + (size_t)b * L2_PTS_MAX;
        const float *xf = x + (size_t)b * H * W;
        // an item = one 64-pixel word of a row that holds a source (the rows were listed above); a word's sources go to their
        // raster ranks.  Eight items per step, and everything an item needs before its depths is loaded at once (this block alone
        // works on the frame: the chain of dependent loads is what the step costs)
        constexpr int NQ = 8;
        const int nitems = min(s_nrow, L2_PTS_MAX) * Wd;
        for (int it0 = tid; it0 < nitems; it0 += 256 * NQ) {
            u64 sb[NQ];
            u32 k[NQ];
            int row[NQ], w[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int it = min(it0 + 256 * q, nitems - 1);
                row[q] = s_ptrow[it / Wd];
                w[q] = it % Wd;
                const size_t at = ((size_t)b * H + row[q]) * Wd + w[q];
                sb[q] = it0 + 256 * q < nitems ? srcbits[at] : 0ull;
                k[q] = bs_[row[q]] + wpre_s[at];  // (this block wrote bs_ above, before a barrier)
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                u64 m = sb[q];
                while (m) {
                    const int j = w[q] * 64 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    list[k[q]++] = PtsSrc{(u32)row[q] << 16 | (u32)j, xf[(size_t)row[q] * W + j]};  // ... with its depth
                }
            }
        }
    }
    const bool flags = !l2 && premark && r >= 0;  // this frame's rows carry flags
    // k_sky starts from rows r0 and r0 + 1 as the window kernel leaves them (it runs beside the first any-distance kernel): a
    // frame that is not a window kernel's, or whose row r0 / r0 + 1 is handed on up front, keeps its sky with the other rows
    bool sky = sky_ok && r > 0;
    u32 *far = s_far[r == 32 ? 1 : 0];
    if (flags && r > 0) {
        // a tile row of the window kernel that is left with only a few rows is not worth its windows: all of it goes to the
        // any-distance kernels.  One thread per tile row, whole words at a time.
        // the window kernel's tile rows: nty of them over the rows from the first source row on when the rows above are the
        // sky's, else over the whole frame (fused_body computes the same)
        const int nty = r == 16 ? nty16 : nty32, tbase = sky_ok ? r0 : 0;
        const int TH = (H - tbase + nty - 1) / nty;
        auto bits = [&](int a, int e, int w) {  // the rows [a, e) as bits of word w
            const int lo = max(a - 32 * w, 0), hi = min(e - 32 * w, 32);
            return hi <= lo ? 0u : (hi >= 32 ? 0xFFFFFFFFu : ((1u << hi) - 1u)) & ~((1u << lo) - 1u);
        };
        for (int t = tid; tbase + t * TH < H; t += 256) {
            const int a = tbase + t * TH, e = min(H, tbase + (t + 1) * TH);  // its rows (all below the sky)
            int keep = 0;
            for (int w = a >> 5; w <= (e - 1) >> 5 && a < e; ++w) keep += __popc(~far[w] & bits(a, e, w));
            if (keep && 4 * keep <= e - a)
                for (int w = a >> 5; w <= (e - 1) >> 5; ++w) atomicOr(&far[w], bits(a, e, w));
        }
    }
    __syncthreads();
    if (sky && (((far[r0 >> 5] >> (r0 & 31)) & 1u) || (r0 + 1 < H && ((far[(r0 + 1) >> 5] >> ((r0 + 1) & 31)) & 1u)))) sky = false;
    // per row: l1_cv: 0 = the window kernel's, 1 = the any-distance kernels' (pre-marked here, or set by k_fused when it meets
    // a pixel farther than its halo), 2 = k_sky's; l2: far pixels k_l2win counted
    bool one = false;
    for (int i = tid; i < H; i += 256) {
        u32 f = 0u;
        if (flags && r > 0) f = i < r0 && sky_ok ? (sky ? 2u : 1u) : (far[i >> 5] >> (i & 31)) & 1u;
        if (l2 && r > 0 && ((far[i >> 5] >> (i & 31)) & 1u)) f = L2_ROW_GONE;  // (a frame without sources is not routed to a window)
        one |= f == 1u;
        rowfar[(size_t)b * H + i] = f;
    }
    const bool any1 = __syncthreads_or(one);
    const bool anygone = l2 && r > 0 && s_nfar[r == 32 ? 1 : 0] > 0;  // l2: rows for k_colT + k_l2env from the start
    if (tid == 0) {
        finfo[b * FI_STRIDE + FI_NSRC] = (int)run_s;
        finfo[b * FI_STRIDE + FI_NVAL] = (int)run_v;
        finfo[b * FI_STRIDE + FI_MISALIGNED] = misaligned;
        finfo[b * FI_STRIDE + FI_DLB] = s_dlb;
        finfo[b * FI_STRIDE + FI_NUNRES] = 0;
        finfo[b * FI_STRIDE + FI_SKY] = finfo[b * FI_STRIDE + FI_SKY0] = (flags && sky) ? r0 : 0;
        finfo[b * FI_STRIDE + FI_TR0] = (flags && r > 0 && sky_ok) ? r0 : 0;
        const bool marked = flags && r > 0 && (sky || any1);
        route[b] = marked ? (r | ROUTE_PREMARK) : r;
        negflag[b] = 0;  // k_mask_o's "this frame holds a negative value": consumed before this kernel, reset for the next pass
        const bool general = r == 0;
        // 2: the any-distance kernels take the whole frame; 1: the rows flagged 1 (pre-marked here, or by k_fused, or (l2) by
        // k_l2win when it hands a row of far pixels on); 0: nothing for them
        // 3 (l1_cv, ROUTE_POINTS): k_pts takes the frame, of the any-distance kernels only k_tiesx has something to do
        fflag2[b] = (!l2 && r == ROUTE_POINTS) ? 3 : general ? 2 : ((flags && any1) || anygone) ? 1 : 0;
        frame_status[b] = (general || r == ROUTE_POINTS || marked) ? DTFILL_FRAME_GENERAL_PATH : DTFILL_FRAME_OK;
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
