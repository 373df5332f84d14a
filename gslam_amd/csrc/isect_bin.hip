// isect_bin.hip — K3..K7 v2: tile-binned depth sort ("per-tile radix depth sort" of the north star, done as
// bin-by-tile + per-tile LDS sort).  Same integer-exact contract as isect.hip (gsplat isect_tiles +
// isect_offset_encode, gslam/rasterization.py:259-274): within a (camera, tile) the order is ascending
// (float_bits(depth), flatten id), which is exactly what a stable sort of gsplat's 64-bit keys gives.
//
// Why not a device-wide radix sort: the key's high bits are the tile id, which is known when an intersection is
// emitted.  Binning by tile costs ONE 8-byte scattered write per intersection, the per-tile offsets fall out of
// the binning for free (no offset-encode pass), the residual sort is tile-local and runs out of LDS, and nothing
// needs the host: every size lives in device memory, so the whole render is sync-free (the M read-back of the
// reference's isect_tiles is gone).  HBM traffic per intersection: 8 B written + 8 B read + 4 B written (+8 B if
// isect_ids are materialised) against >= 24 B x 6 digit passes for the 44-47 live key bits of the global sort.
//
//   1. count_matrix  : per workgroup (1024+ Gaussians): per-tile counts in LDS -> one row of a [workgroups][tiles]
//                      matrix (rectangles larger than 16 tiles are walked cooperatively by the whole wavefront)
//   2. column_scan   : one thread per tile: exclusive scan down the workgroup dimension (each workgroup's base inside
//                      the tile) and the tile's total; tile_scan: exclusive scan of the totals -> offsets[T+1], M
//   3. place         : LDS cursors start at offsets[tile] + base[workgroup][tile]; 8-byte entries
//                      (depth_bits<<32 | flatten_id) go straight to their tile's range.  No global atomic anywhere
//                      (the first version reserved ranges with one contended atomic per ~3 entries and needed a
//                      difference-grid pass for the totals).  What bounds this step now is the 8-byte granularity of
//                      the scattered stores.
//   4. tile_sort     : one workgroup per tile.  Tiles of up to 2048 keys: counting sort on a monotone depth -> bucket
//                      map plus an exact in-bucket rank (two LDS atomics per key, no search).  Larger tiles (and
//                      degenerate depth distributions): merge sort by ranks in an LDS window of up to 8192 keys,
//                      continued through global memory beyond that.  Writes flatten_ids / isect_ids
#include <stdlib.h>
#include <algorithm>

#include "gsx_common.h"
#include "tile_balance.h"
#include "tile_rect.h"

namespace {

using namespace gsx_rect;

#ifndef GSX_FAST_CULL
#define GSX_FAST_CULL 1
#endif
constexpr int BIN_THREADS = 256;
constexpr int BIN_ITEMS = 4;          // Gaussians per thread of the coarse pre-sort passes
constexpr int SORT_THREADS = 512;
constexpr int COOP_AREA = 16;         // rectangles with more tiles than this are walked by the whole wavefront

// rects (nullable): the rectangle of every (camera, Gaussian) packed by the projection (gsx_project_fwd_rects, tile_rect.h) - read
// INSTEAD of means2d + radius (4 bytes for 12), and tight if the projection was asked for that (gsx_isect_bin_sort_rects)
__device__ __forceinline__ Rect load_rect(const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                                          int64_t idx, int tile_w, int tile_h, bool in_range,
                                          const uint32_t *__restrict__ rects = nullptr) {
    Rect r = {0, 0, 0, 0};
    if (in_range) {
        if (rects) return unpack_rect(rects[idx]);
        const int32_t rad = radii[idx];
        if (rad > 0) r = tile_rect(means2d[2 * idx], means2d[2 * idx + 1], rad, tile_w, tile_h);
    }
    return r;
}

// ---- 3. binning ---------------------------------------------------------------------------------------------------
// walks every tile of every rectangle held by the wavefront's lanes and calls op(tile_local_index, lo, hi) with the
// owning lane's 64-bit payload; small rectangles are walked by their own lane, large ones by all 64 lanes together
// (payload broadcast once per large rectangle).  Must be called by all lanes of the wavefront.
template <typename Op>
__device__ __forceinline__ void walk_rects(const Rect &r, int tile_w, unsigned int lo, unsigned int hi, Op op,
                                           int base = 0 /* added to every tile index (camera offset) */) {
    const int w = r.x1 - r.x0, h = r.y1 - r.y0;
    const int area = (w > 0 && h > 0) ? w * h : 0;
    const int lane = threadIdx.x & 63;
    if (area > 0 && area <= COOP_AREA) {
        for (int y = r.y0; y < r.y1; ++y)
            for (int x = r.x0; x < r.x1; ++x) op(base + y * tile_w + x, lo, hi);
    }
    unsigned long long big = __ballot(area > COOP_AREA);
    while (big != 0ull) {
        const int l = __ffsll((long long)big) - 1;
        big &= big - 1ull;
        const int bx0 = __builtin_amdgcn_readlane(r.x0, l), by0 = __builtin_amdgcn_readlane(r.y0, l);
        const int bw = __builtin_amdgcn_readlane(w, l), ba = __builtin_amdgcn_readlane(area, l);
        const int bbase = __builtin_amdgcn_readlane(base, l);
        const unsigned int blo = (unsigned int)__builtin_amdgcn_readlane((int)lo, l);
        const unsigned int bhi = (unsigned int)__builtin_amdgcn_readlane((int)hi, l);
        for (int k = lane; k < ba; k += 64) {
            const int yy = k / bw, xx = k - yy * bw;
            op(bbase + (by0 + yy) * tile_w + bx0 + xx, blo, bhi);
        }
    }
}

// walk_rects for the placement passes: appends the lane's key to every tile of its rectangle through the LDS cursors.
// Small rectangles take two tiles per trip - both returning LDS adds are issued before the first result is waited
// for - because the chain add -> wait -> store -> next tile ran at the LDS round-trip latency per tile.
template <bool CUT = false>
__device__ __forceinline__ void place_rects(const Rect &r, int tile_w, unsigned int lo, unsigned int hi, int *s_cur,
                                            int64_t M_cap, unsigned long long *__restrict__ entries, int base = 0,
                                            const unsigned int *s_cut = nullptr /* CUT: only keys with hi <= s_cut[tile] */) {
    const int w = r.x1 - r.x0, h = r.y1 - r.y0;
    const int area = (w > 0 && h > 0) ? w * h : 0;
    const unsigned long long key = ((unsigned long long)hi << 32) | lo;
    if (area > 0 && area <= COOP_AREA) {
        int x = r.x0, y = r.y0;
        for (int k = 0; k < area; k += 2) {
            const int t0 = base + y * tile_w + x;
            if (++x == r.x1) { x = r.x0; ++y; }
            bool two = k + 1 < area;
            const int t1 = base + y * tile_w + x;
            if (++x == r.x1) { x = r.x0; ++y; }
            bool one = true;
            if constexpr (CUT) { one = hi <= s_cut[t0]; two = two && hi <= s_cut[t1]; }
            const int p0 = one ? atomicAdd(&s_cur[t0], 1) : -1;
            const int p1 = two ? atomicAdd(&s_cur[t1], 1) : -1;
            if (one && (uint64_t)(uint32_t)p0 < (uint64_t)M_cap) entries[p0] = key;
            if (two && (uint64_t)(uint32_t)p1 < (uint64_t)M_cap) entries[p1] = key;
        }
    }
    const Rect none = {0, 0, 0, 0};
    walk_rects(area > COOP_AREA ? r : none, tile_w, lo, hi, [&](int tile, unsigned int l, unsigned int hh) {
        if constexpr (CUT) { if (hh > s_cut[tile]) return; }
        const int pos = atomicAdd(&s_cur[tile], 1);
        if ((uint64_t)(uint32_t)pos < (uint64_t)M_cap) entries[pos] = ((unsigned long long)hh << 32) | l;
    }, base);
}

// ---- 3b. binning without global atomics --------------------------------------------------------------------------
// The first version reserved a range per (workgroup, touched tile) with a global atomic; a workgroup of 1024 random
// Gaussians touches most tiles with ~3 entries each, so that is one contended atomic per ~3 entries (3.5 M atomics on
// 9600 addresses at 500 k x 8 cameras: 188 us, the largest kernel of the sort).  Here the per-(workgroup, tile) counts
// go to a matrix instead, a column scan turns them into each workgroup's base inside every tile and into the per-tile
// totals (which makes the difference-grid pass unnecessary), and the placement pass starts its LDS cursors from
// offsets[tile] + base[workgroup][tile]: plain coalesced stores and loads, no global atomic anywhere.
constexpr int GB_MAX = 640;           // workgroups per camera (the matrix in the workspace is sized for this)

__global__ __launch_bounds__(BIN_THREADS) void count_matrix_kernel(const float *__restrict__ means2d,
                                                                   const int32_t *__restrict__ radii, int64_t N,
                                                                   int tile_w, int tile_h, int items,
                                                                   int32_t *__restrict__ cnt /*[C][gblocks][n_tiles]*/,
    const uint32_t *__restrict__ trec /* nullable: packed rectangles of the projection */) {
    extern __shared__ int s_cnt[];  // [n_tiles]
    const int c = blockIdx.y;
    const int n_tiles = tile_w * tile_h;
    for (int i = threadIdx.x; i < n_tiles; i += BIN_THREADS) s_cnt[i] = 0;
    __syncthreads();
    for (int it = 0; it < items; ++it) {
        const int64_t g = ((int64_t)blockIdx.x * items + it) * BIN_THREADS + threadIdx.x;
        const Rect r = load_rect(means2d, radii, (int64_t)c * N + g, tile_w, tile_h, g < N, trec);
        walk_rects(r, tile_w, 0u, 0u, [&](int tile, unsigned int, unsigned int) { atomicAdd(&s_cnt[tile], 1); });
    }
    __syncthreads();
    int32_t *row = cnt + ((int64_t)c * gridDim.x + blockIdx.x) * n_tiles;
    for (int i = threadIdx.x; i < n_tiles; i += BIN_THREADS) row[i] = s_cnt[i];
}

// exclusive scan down the workgroup dimension of the count matrix (in place), column totals into counts[].
// A workgroup takes 64 adjacent tile columns of one camera (one 256-byte row segment per wavefront load) and splits the
// rows over its 16 wavefronts: each lane holds its <= GB_MAX / 16 rows in registers (all loads in flight at once), the
// wavefronts exchange their column sums through LDS, and the rows go back as exclusive prefixes.  One thread per
// column walking all rows serially was a 23 us latency chain at 500 k Gaussians (489 rows, 19 wavefronts in flight).
constexpr int CS_GROUPS = 16;
constexpr int CS_ROWS = GB_MAX / CS_GROUPS;                 // rows per wavefront, at most

// NEAR (near placement, gsx_front_fwd_near): a word of the matrix holds the pairs in front of the tile's cut-off in its low half and
// the pairs behind it in its high half.  Only the near pairs are placed: the rows get the exclusive prefix of the NEAR counts, the
// per-tile totals count BOTH (the tile's segment keeps room for the far keys, which its rasteriser workgroup appends itself if a
// pixel outlives the near ones), and near_out[] gets the near totals = the keys the placement writes.
template <bool NEAR>
__global__ __launch_bounds__(64 * CS_GROUPS) void column_scan_kernel(int32_t *__restrict__ cnt, int gblocks,
                                                                     int n_tiles, int32_t *__restrict__ counts /*[T]*/,
                                                                     int32_t *__restrict__ near_out /*[T]*/) {
    __shared__ int s_tot[CS_GROUPS][64];
    __shared__ int s_far[NEAR ? CS_GROUPS : 1][64];
    const int c = blockIdx.y;
    const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int tl = blockIdx.x * 64 + lane;
    const bool in = tl < n_tiles;
    const int rpg = (gblocks + CS_GROUPS - 1) / CS_GROUPS;
    const int r0 = rg * rpg, r1 = min(gblocks, r0 + rpg);
    int32_t *col = cnt + (int64_t)c * gblocks * n_tiles + tl;
    int v[CS_ROWS];
    int sum = 0, far = 0;
#pragma unroll
    for (int u = 0; u < CS_ROWS; ++u) {
        v[u] = (in && r0 + u < r1) ? col[(int64_t)(r0 + u) * n_tiles] : 0;
        if constexpr (NEAR) { far += (int)((unsigned int)v[u] >> 16); v[u] &= 0xffff; }
        sum += v[u];
    }
    s_tot[rg][lane] = sum;
    if constexpr (NEAR) s_far[rg][lane] = far;
    __syncthreads();
    int run = 0, total = 0, far_total = 0;
#pragma unroll
    for (int w = 0; w < CS_GROUPS; ++w) {
        const int tw = s_tot[w][lane];
        run += (w < rg) ? tw : 0;
        total += tw;
        if constexpr (NEAR) far_total += s_far[w][lane];
    }
#pragma unroll
    for (int u = 0; u < CS_ROWS; ++u) {
        if (in && r0 + u < r1) col[(int64_t)(r0 + u) * n_tiles] = run;
        run += v[u];
    }
    if (in && rg == 0) {
        counts[(int64_t)c * n_tiles + tl] = total + far_total;
        if constexpr (NEAR) near_out[(int64_t)c * n_tiles + tl] = total;
    }
}

// exclusive scan of the T per-tile counts held in offsets[] -> offsets[T+1], M, overflow status (one workgroup)
// `order` (nullable, [T]): the tiles grouped by descending list length (256 buckets relative to the longest list):
// the launch order of the rasteriser's workgroups.  Heaviest first spreads the long lists over the CUs instead of
// leaving them where the image puts them (-5..7 % rasteriser time at 100 k Gaussians, tools/ab_raster.py).
__global__ __launch_bounds__(1024) void tile_scan_kernel(int T, int64_t M_cap, int32_t *__restrict__ offsets,
                                                         int64_t *__restrict__ M_dev, int32_t *__restrict__ status,
                                                         int32_t *__restrict__ order) {
    __shared__ long long s_scan[1024];
    __shared__ int s_bucket[257];
    __shared__ int s_wmax[16];
    const int t = threadIdx.x;
    const int per = (T + 1023) / 1024;
    const int lo = min(T, t * per), hi = min(T, lo + per);
    long long sum = 0;
    int cmax = 0;
    for (int i = lo; i < hi; ++i) { const int c = max(offsets[i], 0); sum += c; cmax = max(cmax, c); }
    s_scan[t] = sum;
    if (order) {
        if (t < 257) s_bucket[t] = 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) cmax = max(cmax, __shfl_xor(cmax, off, 64));
        if ((t & 63) == 0) s_wmax[t >> 6] = cmax;            // (1024 atomicMax on one LDS word took as long as the scan)
        __syncthreads();
        int mx = 1;
#pragma unroll
        for (int w = 0; w < 16; ++w) mx = max(mx, s_wmax[w]);
        for (int i = lo; i < hi; ++i) {
            const int b = 255 - (int)(((long long)max(offsets[i], 0) * 255) / mx);     // bucket 0 = longest lists
            atomicAdd(&s_bucket[b + 1], 1);
        }
        __syncthreads();
        if (t < 64) {                                        // one wavefront: counts at [b + 1] -> bucket starts at [b]
            const int c0 = s_bucket[4 * t + 1], c1 = s_bucket[4 * t + 2], c2 = s_bucket[4 * t + 3], c3 = s_bucket[4 * t + 4];
            const int v = c0 + c1 + c2 + c3;
            int incl = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int u = __shfl_up(incl, off, 64);
                if (t >= off) incl += u;
            }
            const int base = incl - v;
            s_bucket[4 * t] = base; s_bucket[4 * t + 1] = base + c0; s_bucket[4 * t + 2] = base + c0 + c1;
            s_bucket[4 * t + 3] = base + c0 + c1 + c2;
        }
        __syncthreads();
        for (int i = lo; i < hi; ++i) {
            const int b = 255 - (int)(((long long)max(offsets[i], 0) * 255) / mx);
            order[atomicAdd(&s_bucket[b], 1)] = i;
        }
    }
    for (int off = 1; off < 1024; off <<= 1) {
        __syncthreads();
        const long long add = (t >= off) ? s_scan[t - off] : 0;
        __syncthreads();
        s_scan[t] += add;
    }
    __syncthreads();
    long long run = s_scan[t] - sum;
    const long long total = s_scan[1023];
    for (int i = lo; i < hi; ++i) {
        const int cnt = max(offsets[i], 0);
        offsets[i] = (int32_t)min(run, (long long)0x7fffffff);
        run += cnt;
    }
    if (t == 0) {
        offsets[T] = (int32_t)min(total, (long long)0x7fffffff);
        M_dev[0] = total;
        if (total > M_cap || total > 0x7fffffffLL) atomicOr(status, 1);
    }
}

__global__ __launch_bounds__(BIN_THREADS) void place_kernel(const float *__restrict__ means2d,
                                                            const int32_t *__restrict__ radii,
                                                            const float *__restrict__ depths, int64_t N, int tile_w,
                                                            int tile_h, int items, int64_t M_cap,
                                                            const int32_t *__restrict__ offsets,
                                                            const int32_t *__restrict__ cnt,
                                                            unsigned long long *__restrict__ entries,
    const uint32_t *__restrict__ trec /* nullable: packed rectangles of the projection */) {
    extern __shared__ int s_cur[];  // [n_tiles]: this workgroup's absolute write cursor per tile
    const int c = blockIdx.y;
    const int n_tiles = tile_w * tile_h;
    const int32_t *row = cnt + ((int64_t)c * gridDim.x + blockIdx.x) * n_tiles;
    for (int i = threadIdx.x; i < n_tiles; i += BIN_THREADS) s_cur[i] = offsets[(int64_t)c * n_tiles + i] + row[i];
    __syncthreads();
    for (int it = 0; it < items; ++it) {
        const int64_t g = ((int64_t)blockIdx.x * items + it) * BIN_THREADS + threadIdx.x;
        const int64_t idx = (int64_t)c * N + g;
        const Rect r = load_rect(means2d, radii, idx, tile_w, tile_h, g < N, trec);
        const bool has = (r.x1 > r.x0) && (r.y1 > r.y0);
        const unsigned int klo = (unsigned int)idx, khi = has ? __float_as_uint(depths[idx]) : 0u;
        place_rects(r, tile_w, klo, khi, s_cur, M_cap, entries);
    }
}

// ---- 3c. binning for large maps: spatial pre-sort of the Gaussians --------------------------------------------------
// place_kernel's workgroups hold arbitrary Gaussians, so each of them appends to every tile a few entries at a time: with
// hundreds of resident workgroups the partially written 128-byte lines of the output are (workgroups x tiles) many -
// far beyond the L2 - and go out to HBM half empty (5M Gaussians at 1080p: 1.48 ms for 606 MB, the largest kernel of
// the render; 2M x 8 cameras: 0.95 ms).  Here the visible (camera, Gaussian) instances are first binned by the 4x4-tile
// block that holds the top-left tile of their rectangle (16-byte records: packed rectangle, depth bits, flatten id;
// same count matrix / column scan / placement scheme, one level up).  The tile-level count and placement passes then
// read those records in order: a workgroup's instances share a neighbourhood, touch ~100 tiles instead of all of
// them, and fill each line of the output within one trip of its loop, so the L2 merges the 8-byte stores into full lines.
// The records take 16 B x C x N of extra workspace (gsx_isect_bin_workspace_bytes_n); a caller that sized its
// workspace without them (gsx_isect_bin_workspace_bytes) gets the direct placement.
constexpr int SUPER = 4;              // tiles per side of a pre-sort block

struct PreRec {
    uint32_t xs, ys_c, depth, id;     // x0 | x1 << 16 ; y0 | y1 << 12 | camera << 24 ; depth bits ; flatten id
};

__global__ __launch_bounds__(BIN_THREADS) void coarse_count_kernel(const float *__restrict__ means2d,
                                                                   const int32_t *__restrict__ radii, int64_t N,
                                                                   int tile_w, int tile_h, int items, int sw, int S,
                                                                   int32_t *__restrict__ cnt /*[C][gblocks][S]*/,
    const uint32_t *__restrict__ trec /* nullable: packed rectangles of the projection */) {
    extern __shared__ int s_cnt[];  // [S]
    const int c = blockIdx.y;
    for (int i = threadIdx.x; i < S; i += BIN_THREADS) s_cnt[i] = 0;
    __syncthreads();
    for (int it = 0; it < items; ++it) {
        const int64_t g = ((int64_t)blockIdx.x * items + it) * BIN_THREADS + threadIdx.x;
        const Rect r = load_rect(means2d, radii, (int64_t)c * N + g, tile_w, tile_h, g < N, trec);
        if (r.x1 > r.x0 && r.y1 > r.y0) atomicAdd(&s_cnt[(r.y0 / SUPER) * sw + r.x0 / SUPER], 1);
    }
    __syncthreads();
    int32_t *row = cnt + ((int64_t)c * gridDim.x + blockIdx.x) * S;
    for (int i = threadIdx.x; i < S; i += BIN_THREADS) row[i] = s_cnt[i];
}

__global__ __launch_bounds__(BIN_THREADS) void coarse_place_kernel(const float *__restrict__ means2d,
                                                                   const int32_t *__restrict__ radii,
                                                                   const float *__restrict__ depths, int64_t N,
                                                                   int tile_w, int tile_h, int items, int sw, int S,
                                                                   int64_t rec_cap, const int32_t *__restrict__ coff,
                                                                   const int32_t *__restrict__ cnt,
                                                                   PreRec *__restrict__ recs,
    const uint32_t *__restrict__ trec /* nullable: packed rectangles of the projection */) {
    extern __shared__ int s_cur[];  // [S]
    const int c = blockIdx.y;
    const int32_t *row = cnt + ((int64_t)c * gridDim.x + blockIdx.x) * S;
    for (int i = threadIdx.x; i < S; i += BIN_THREADS) s_cur[i] = coff[(int64_t)c * S + i] + row[i];
    __syncthreads();
    for (int it = 0; it < items; ++it) {
        const int64_t g = ((int64_t)blockIdx.x * items + it) * BIN_THREADS + threadIdx.x;
        const int64_t idx = (int64_t)c * N + g;
        const Rect r = load_rect(means2d, radii, idx, tile_w, tile_h, g < N, trec);
        if (r.x1 > r.x0 && r.y1 > r.y0) {
            const int pos = atomicAdd(&s_cur[(r.y0 / SUPER) * sw + r.x0 / SUPER], 1);
            if ((uint64_t)(uint32_t)pos < (uint64_t)rec_cap) {
                PreRec o;
                o.xs = (uint32_t)r.x0 | ((uint32_t)r.x1 << 16);
                o.ys_c = (uint32_t)r.y0 | ((uint32_t)r.y1 << 12) | ((uint32_t)c << 24);
                o.depth = __float_as_uint(depths[idx]);
                o.id = (uint32_t)idx;
                reinterpret_cast<uint4 *>(recs)[pos] = make_uint4(o.xs, o.ys_c, o.depth, o.id);
            }
        }
    }
}

__device__ __forceinline__ Rect unpack_rec(const uint4 &v, int n_tiles, int &base) {
    Rect r;
    r.x0 = (int)(v.x & 0xffffu); r.x1 = (int)(v.x >> 16);
    r.y0 = (int)(v.y & 0xfffu);  r.y1 = (int)((v.y >> 12) & 0xfffu);
    base = (int)(v.y >> 24) * n_tiles;
    return r;
}

// (Tried and dropped: a tile-major walk - the wavefront loops over the tiles of its 64 records' common window, a ballot
// gives each tile's count and ranks, one LDS add per tile - to get rid of the same-address LDS atomics that neighbouring
// records cause; the serial ballot -> scalar -> writelane chain per tile made both passes 2-2.4x slower.)
// Per-workgroup tile counts of a chunk of pre-sorted records.  Neighbouring records cover the same few tiles, so one LDS
// atomic per (record, tile) piles up on a handful of addresses (384 us at 5M / 1080p).  Here every rectangle adds +-1 at
// its four corners of a per-camera (tile_h + 1) x (tile_w + 1) difference grid in LDS - 4 atomics per record instead of
// ~15 - and the grid is integrated once per workgroup (rows, then columns).
constexpr int FINE_THREADS = 512;     // the tile-level passes of the pre-sorted path: few, fat workgroups (<= GB_MAX)

__global__ __launch_bounds__(FINE_THREADS) void fine_count_kernel(const PreRec *__restrict__ recs,
                                                                 const int64_t *__restrict__ n_inst, int chunk,
                                                                 int tile_w, int tile_h, int C,
                                                                 int32_t *__restrict__ cnt /*[gblocks][C * tiles]*/) {
    extern __shared__ int s_grid[];  // [C][tile_h + 1][tile_w + 1]
    const int gw = tile_w + 1, gh = tile_h + 1, G = gw * gh;
    const int n_tiles = tile_w * tile_h;
    for (int i = threadIdx.x; i < C * G; i += FINE_THREADS) s_grid[i] = 0;
    __syncthreads();
    const int64_t n = n_inst[0];
    // equal shares of the records that exist (n is only known on the device; the grid is sized for the capacity C x N,
    // and with fixed-size chunks the workgroups beyond the visible instances - a third of them - had nothing to do)
    const int64_t lo = n * blockIdx.x / gridDim.x, hi = n * (blockIdx.x + 1) / gridDim.x;
    (void)chunk;
    for (int64_t i = lo + threadIdx.x; i < hi; i += FINE_THREADS) {
        int base = 0;
        const Rect r = unpack_rec(reinterpret_cast<const uint4 *>(recs)[i], n_tiles, base);
        int *g = s_grid + (base / n_tiles) * G;
        atomicAdd(&g[r.y0 * gw + r.x0], 1);
        atomicAdd(&g[r.y0 * gw + r.x1], -1);
        atomicAdd(&g[r.y1 * gw + r.x0], -1);
        atomicAdd(&g[r.y1 * gw + r.x1], 1);
    }
    __syncthreads();
    for (int row = threadIdx.x; row < C * gh; row += FINE_THREADS) {          // prefix along x
        int *p = s_grid + (row / gh) * G + (row % gh) * gw;
        int run = 0;
        for (int x = 0; x < gw; ++x) { run += p[x]; p[x] = run; }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < C * gw; col += FINE_THREADS) {          // prefix along y
        int *p = s_grid + (col / gw) * G + (col % gw);
        int run = 0;
        for (int y = 0; y < gh; ++y) { run += p[y * gw]; p[y * gw] = run; }
    }
    __syncthreads();
    int32_t *row_out = cnt + (int64_t)blockIdx.x * C * n_tiles;
    for (int i = threadIdx.x; i < C * n_tiles; i += FINE_THREADS) {
        const int c = i / n_tiles, tl = i - c * n_tiles;
        const int y = tl / tile_w, x = tl - y * tile_w;
        row_out[i] = s_grid[c * G + y * gw + x];
    }
}

__global__ __launch_bounds__(FINE_THREADS) void fine_place_kernel(const PreRec *__restrict__ recs,
                                                                 const int64_t *__restrict__ n_inst, int chunk,
                                                                 int tile_w, int n_tiles, int T, int64_t M_cap,
                                                                 const int32_t *__restrict__ offsets,
                                                                 const int32_t *__restrict__ cnt,
                                                                 unsigned long long *__restrict__ entries) {
    extern __shared__ int s_cur[];  // [T]
    const int32_t *row = cnt + (int64_t)blockIdx.x * T;
    for (int i = threadIdx.x; i < T; i += FINE_THREADS) s_cur[i] = offsets[i] + row[i];
    __syncthreads();
    const int64_t n = n_inst[0];
    // equal shares of the records that exist (n is only known on the device; the grid is sized for the capacity C x N,
    // and with fixed-size chunks the workgroups beyond the visible instances - a third of them - had nothing to do)
    const int64_t lo = n * blockIdx.x / gridDim.x, hi = n * (blockIdx.x + 1) / gridDim.x;
    (void)chunk;
    uint4 v_nxt = make_uint4(0u, 0u, 0u, 0u);               // the record of the next trip is in flight during this one
    if (lo + threadIdx.x < hi) v_nxt = reinterpret_cast<const uint4 *>(recs)[lo + threadIdx.x];
    for (int64_t i0 = lo; i0 < hi; i0 += FINE_THREADS) {
        const int64_t i = i0 + threadIdx.x;
        const uint4 v = v_nxt;
        if (i + FINE_THREADS < hi) v_nxt = reinterpret_cast<const uint4 *>(recs)[i + FINE_THREADS];
        Rect r = {0, 0, 0, 0};
        int base = 0;
        unsigned int klo = 0u, khi = 0u;
        if (i < hi) {
            r = unpack_rec(v, n_tiles, base);
            klo = v.w; khi = v.z;
        }
        place_rects(r, tile_w, klo, khi, s_cur, M_cap, entries, base);
    }
}

// ---- 4. per-tile sort -----------------------------------------------------------------------------------------------
// One workgroup per tile.  Up to `cap` keys (LDS window chosen by the host from the capacity) are merge-sorted by
// RANKS, ping-ponging between two LDS buffers:
//   1. runs of 64 keys are sorted by counting ranks (position = number of smaller keys in the run; all lanes stream
//      the run as 16-byte broadcast reads, (a - k) >> 63 is the comparison: keys are < 2^63)
//   2. log2(n/64) merge levels: a key's position in the merged pair of runs = its position in its own run + the
//      number of partner keys before it (lower bound from the left run, upper bound from the right run: a stable
//      merge), found by a binary search.  One barrier per level, every key moves exactly once per level.
// ~ (32 + sum of log2(run)) short steps per key instead of log^2(n)/2 compare-exchanges with a barrier each.
// Tiles larger than the window sort window-sized chunks in LDS and continue the same merge levels in global memory
// (L2), ping-ponging between the entry buffer and a scratch copy.
constexpr int RUN0 = 64;

// one rank-merge level: runs of length `run` in src[0..n) -> runs of 2*run in dst
template <typename Ptr, int TH = SORT_THREADS>
__device__ __forceinline__ void merge_level(Ptr src, Ptr dst, int n, int run) {
    for (int i = threadIdx.x; i < n; i += TH) {
        const unsigned long long k = src[i];
        const int r = i / run;
        const int own = r * run, pair = (r & ~1) * run;
        const int pb = (r ^ 1) * run;                          // partner run
        const int plen = max(0, min(run, n - pb));
        const bool right = (r & 1) != 0;
        int lo = 0, hi = plen;                                  // left run: #partner < k ; right run: #partner <= k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const unsigned long long v = src[pb + mid];
            const bool before = right ? (v <= k) : (v < k);
            if (before) lo = mid + 1; else hi = mid;
        }
        dst[pair + (i - own) + lo] = k;
    }
}

// sorts n <= cap keys read from `in` (global); returns the LDS buffer that holds the sorted keys
template <int TH = SORT_THREADS>
__device__ __forceinline__ unsigned long long *lds_sort(const unsigned long long *__restrict__ in, int n,
                                                        unsigned long long *bufA, unsigned long long *bufB) {
    const unsigned long long INF = ~0ull >> 1;            // larger than any key, still < 2^63
    const int n_pad = (n + RUN0 - 1) / RUN0 * RUN0;       // <= cap (cap is a multiple of 64)
    unsigned long long *src = bufA, *dst = bufB;
    __syncthreads();                                       // callers may still be reading the buffers
    for (int i = threadIdx.x; i < n_pad; i += TH) src[i] = (i < n) ? in[i] : INF;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += TH) {
        const unsigned long long k = src[i];
        const int cb = i & ~(RUN0 - 1);
        const ulonglong2 *s2 = reinterpret_cast<const ulonglong2 *>(src + cb);
        unsigned int rank = 0;
#pragma unroll 8
        for (int m = 0; m < RUN0 / 2; ++m) {
            const ulonglong2 a = s2[m];
            rank += (unsigned int)((a.x - k) >> 63) + (unsigned int)((a.y - k) >> 63);
        }
        dst[cb + rank] = k;
    }
    __syncthreads();
    { unsigned long long *t = src; src = dst; dst = t; }
    for (int run = RUN0; run < n; run <<= 1) {
        merge_level<unsigned long long *, TH>(src, dst, n, run);
        __syncthreads();
        unsigned long long *t = src; src = dst; dst = t;
    }
    return src;
}

// ---- 4b. per-tile sort by counting ------------------------------------------------------------------------------------
// The merge sort above is a chain of ~40 dependent LDS reads per key (binary searches); depth keys inside one tile are
// spread over a narrow range, so a counting sort gets there with two LDS atomics per key and no search:
//   1. min / max depth of the tile (block reduction) -> monotone map depth -> bucket in [0, NB)
//   2. histogram (ds_add), exclusive scan over the NB buckets, scatter (returning ds_add) -> keys grouped by bucket
//   3. exact rank inside the bucket by counting smaller (depth, id) keys among the bucket's members (1-2 on average)
// The map is monotone in the float depth (subtract, multiply and float->int conversion are all monotone), so bucket
// order never contradicts key order and step 3 makes the result the same total order the merge sort produces.
//
// One launch, one workgroup per tile, two regimes chosen by the tile's size (known only on the device):
//   * up to 2048 keys: keys staged in LDS, 1024 buckets;
//   * more: keys streamed from global memory (L2), 4096 buckets, scatter into the scratch copy grouped by bucket, then
//     the grouped keys come back through a 2048-key LDS window cut at bucket boundaries for step 3.  Every pass is a
//     coalesced stream, the tile size is unbounded.  (The merge sort it replaced walked log2(n / 8192) levels of
//     dependent binary searches through L2: 6.1 ms of a 13 ms render at 5M Gaussians / 1080p.)
// A tile whose keys pile up in one bucket (same depth everywhere) falls back to the merge sort inside the same LDS.
// After the scatter a bucket's cursor is its end, which is the next bucket's start: no separate start array, and the
// key buffers stay at 36 KiB (+ 16 KiB of parked output, below: 52 KiB, three workgroups of 8 wavefronts per CU; at 36 KiB
// and four per CU the launch measured 14.8 against 15.0 us - the longest tiles set its duration, not the occupancy).
constexpr int CNT_NB = 1024;          // buckets, keys staged in LDS
constexpr int CNT_MAXN = 2048;        // largest tile of the LDS regime = LDS window of the streaming regime
constexpr int CNT_MAX_BUCKET = 96;    // largest bucket the quadratic step 3 accepts (LDS regime)
constexpr int BIG_NB = 4096;          // buckets, streaming regime
constexpr int BIG_MAX_BUCKET = 256;

// exclusive scan of s_cur[0..NB) in place (NB = PER * SORT_THREADS); returns true if some count exceeds `limit`
template <int NB, int TH = SORT_THREADS>
__device__ __forceinline__ bool bucket_scan(int *s_cur, int *s_wtot, int limit) {
    constexpr int PER = NB / TH;
    const int t = threadIdx.x;
    int cnt[PER];
    int v = 0, big = 0;
#pragma unroll
    for (int j = 0; j < PER; ++j) { cnt[j] = s_cur[PER * t + j]; v += cnt[j]; big = max(big, cnt[j]); }
    int incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int u = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += u;
    }
    if ((t & 63) == 63) s_wtot[t >> 6] = incl;
    const int any_big = __syncthreads_or(big > limit);
    int run = incl - v;
    for (int w = 0; w < (t >> 6); ++w) run += s_wtot[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) { s_cur[PER * t + j] = run; run += cnt[j]; }
    __syncthreads();
    return any_big != 0;
}

// merge sort of one tile with a `cap`-key LDS window (2 * cap keys of LDS at s_mem): chunks sorted in LDS and written
// back in place, remaining levels through global memory.  Returns where the sorted keys are (LDS or global).
template <int TH = SORT_THREADS>
__device__ __forceinline__ const unsigned long long *merge_sort_tile(unsigned long long *seg, unsigned long long *scr,
                                                                     int n, unsigned long long *s_mem, int cap) {
    if (n <= cap) return lds_sort<TH>(seg, n, s_mem, s_mem + cap);
    for (int cb = 0; cb < n; cb += cap) {
        const int len = min(cap, n - cb);
        const unsigned long long *res = lds_sort<TH>(seg + cb, len, s_mem, s_mem + cap);
        for (int i = threadIdx.x; i < len; i += TH) seg[cb + i] = res[i];
    }
    __syncthreads();
    unsigned long long *src = seg, *dst = scr;
    for (int run = cap; run < n; run <<= 1) {
        merge_level<unsigned long long *, TH>(src, dst, n, run);
        __syncthreads();
        unsigned long long *t = src; src = dst; dst = t;
    }
    return src;
}

__global__ __launch_bounds__(SORT_THREADS) void tile_sort_count_kernel(unsigned long long *__restrict__ entries,
                                                                       unsigned long long *__restrict__ scratch,
                                                                       const int32_t *__restrict__ offsets, int n_tiles,
                                                                       int tile_n_bits, int64_t M_cap, uint32_t id_max,
                                                                       int64_t *__restrict__ isect_ids,
                                                                       int32_t *__restrict__ flatten_ids) {
    // 36 KiB: [a | b | 1024 cursors] in the LDS regime, [window | 4096 cursors] in the streaming one
    __shared__ __attribute__((aligned(16))) unsigned long long s_keys[2 * CNT_MAXN + CNT_NB / 2];
    // Ranked keys are parked here and leave as coalesced runs: storing every key straight to its final position is a
    // 4-byte write to a random slot of the window, and each such write travelled to memory as its own 32-byte request
    // (WRITE_SIZE 2.6 GB for 0.9 GB of output at 5M Gaussians / 1080p).  The LDS regime reuses `a` for this.
    __shared__ __attribute__((aligned(16))) unsigned long long s_out[CNT_MAXN];
    __shared__ unsigned int s_red[2 * (SORT_THREADS / 64)];
    __shared__ int s_wtot[SORT_THREADS / 64];
    const int tile = blockIdx.x;
    const int t = threadIdx.x;
    const int64_t start = max((int64_t)0, min((int64_t)offsets[tile], M_cap));
    const int64_t end = max((int64_t)0, min((int64_t)offsets[tile + 1], M_cap));
    const int n = (int)(end - start);
    if (n <= 0) return;
    const int c = tile / n_tiles, tl = tile - c * n_tiles;
    const long long hi_part = ((long long)c << (32 + tile_n_bits)) | ((long long)tl << 32);
    unsigned long long *seg = entries + start;
    unsigned long long *grp = scratch + start;
    auto emit = [&](int64_t o, unsigned long long k) {
        flatten_ids[o] = (int32_t)min((uint32_t)k, id_max);  // never hand an out-of-range gather index on
        if (isect_ids) isect_ids[o] = hi_part | (long long)(k >> 32);
    };
    const bool small = n <= CNT_MAXN;
    unsigned long long *s_a = s_keys, *s_b = s_keys + CNT_MAXN;
    int *s_cur = reinterpret_cast<int *>(small ? s_keys + 2 * CNT_MAXN : s_keys + CNT_MAXN);
    static_assert(BIG_NB * 4 <= (CNT_MAXN + CNT_NB / 2) * 8, "streaming-regime cursors must fit behind the window");
    const int nb = small ? CNT_NB : BIG_NB;
    // 1. min / max of the depth bits (positive floats: same order as the values); the LDS regime stages the keys
    // The streaming regime samples one 512-key block in four for the range (a quarter of this pass's traffic): keys
    // outside the sampled range clamp to the first / last bucket, which keeps the map monotone (only the balance of the
    // two end buckets can suffer, and the degenerate-bucket check covers that).
    unsigned int dmin = 0xffffffffu, dmax = 0u;
    if (small) {
        unsigned long long kk[CNT_MAXN / SORT_THREADS];      // <= 4 keys per thread, all loads in flight together
#pragma unroll
        for (int q = 0; q < CNT_MAXN / SORT_THREADS; ++q) {
            const int i = t + q * SORT_THREADS;
            kk[q] = (i < n) ? seg[i] : 0ull;
        }
#pragma unroll
        for (int q = 0; q < CNT_MAXN / SORT_THREADS; ++q) {
            const int i = t + q * SORT_THREADS;
            if (i < n) {
                s_a[i] = kk[q];
                const unsigned int d = (unsigned int)(kk[q] >> 32);
                dmin = min(dmin, d); dmax = max(dmax, d);
            }
        }
    } else {
#pragma unroll 4
        for (int i = t; i < n; i += 4 * SORT_THREADS) {
            const unsigned int d = (unsigned int)(seg[i] >> 32);
            dmin = min(dmin, d); dmax = max(dmax, d);
        }
    }
    for (int i = t; i < nb; i += SORT_THREADS) s_cur[i] = 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dmin = min(dmin, (unsigned int)__shfl_xor((int)dmin, off, 64));
        dmax = max(dmax, (unsigned int)__shfl_xor((int)dmax, off, 64));
    }
    if ((t & 63) == 0) { s_red[t >> 6] = dmin; s_red[SORT_THREADS / 64 + (t >> 6)] = dmax; }
    __syncthreads();
    dmin = 0xffffffffu; dmax = 0u;
#pragma unroll
    for (int w = 0; w < SORT_THREADS / 64; ++w) { dmin = min(dmin, s_red[w]); dmax = max(dmax, s_red[SORT_THREADS / 64 + w]); }
    const float fmin_ = __uint_as_float(dmin), fmax_ = __uint_as_float(dmax);
    const float range = fmax_ - fmin_;
    const float scale = (range > 0.0f) ? (float)(nb - 1) / range : 0.0f;
    auto bucket_of = [&](unsigned long long k) -> int {
        const float d = __uint_as_float((unsigned int)(k >> 32));
        const int b = (int)((d - fmin_) * scale);
        return min(max(b, 0), nb - 1);
    };
    // 2. histogram, scan
    if (small) for (int i = t; i < n; i += SORT_THREADS) atomicAdd(&s_cur[bucket_of(s_a[i])], 1);
    else {
        // four keys in flight per thread: with one dependent load per trip this pass ran at the latency of L2, not at
        // its bandwidth (the tile sort spent 74 % of its wave cycles waiting, SQ_WAIT_ANY)
        int i = t;
        for (; i + 3 * SORT_THREADS < n; i += 4 * SORT_THREADS) {
            const unsigned long long k0 = seg[i], k1 = seg[i + SORT_THREADS], k2 = seg[i + 2 * SORT_THREADS],
                                     k3 = seg[i + 3 * SORT_THREADS];
            atomicAdd(&s_cur[bucket_of(k0)], 1); atomicAdd(&s_cur[bucket_of(k1)], 1);
            atomicAdd(&s_cur[bucket_of(k2)], 1); atomicAdd(&s_cur[bucket_of(k3)], 1);
        }
        for (; i < n; i += SORT_THREADS) atomicAdd(&s_cur[bucket_of(seg[i])], 1);
    }
    __syncthreads();
    const bool degenerate = small ? bucket_scan<CNT_NB>(s_cur, s_wtot, CNT_MAX_BUCKET)
                                  : bucket_scan<BIG_NB>(s_cur, s_wtot, BIG_MAX_BUCKET);
    if (degenerate) {                                        // same depth everywhere: merge sort in the same LDS
        const unsigned long long *sorted = merge_sort_tile(seg, grp, n, s_keys, CNT_MAXN);
        for (int i = t; i < n; i += SORT_THREADS) emit(start + i, sorted[i]);
        return;
    }
    auto bucket_start = [&](int b) -> int { return b ? s_cur[b - 1] : 0; };    // valid once the scatter is done
    if (small) {
        for (int i = t; i < n; i += SORT_THREADS) {
            const unsigned long long k = s_a[i];
            s_b[atomicAdd(&s_cur[bucket_of(k)], 1)] = k;
        }
        __syncthreads();
        // 3. exact position inside the bucket (into `a`, no longer needed), and out in order
        for (int i = t; i < n; i += SORT_THREADS) {
            const unsigned long long k = s_b[i];
            const int b = bucket_of(k);
            const int bs = bucket_start(b), be = s_cur[b];
            int rank = 0;
            for (int j = bs; j < be; ++j) rank += (s_b[j] < k) ? 1 : 0;
            s_a[bs + rank] = k;
        }
        __syncthreads();
        for (int i = t; i < n; i += SORT_THREADS) emit(start + i, s_a[i]);
        return;
    }
    {
        int i = t;
        for (; i + 3 * SORT_THREADS < n; i += 4 * SORT_THREADS) {
            const unsigned long long k0 = seg[i], k1 = seg[i + SORT_THREADS], k2 = seg[i + 2 * SORT_THREADS],
                                     k3 = seg[i + 3 * SORT_THREADS];
            const int p0 = atomicAdd(&s_cur[bucket_of(k0)], 1), p1 = atomicAdd(&s_cur[bucket_of(k1)], 1);
            const int p2 = atomicAdd(&s_cur[bucket_of(k2)], 1), p3 = atomicAdd(&s_cur[bucket_of(k3)], 1);
            grp[p0] = k0; grp[p1] = k1; grp[p2] = k2; grp[p3] = k3;
        }
        for (; i < n; i += SORT_THREADS) {
            const unsigned long long k = seg[i];
            grp[atomicAdd(&s_cur[bucket_of(k)], 1)] = k;
        }
    }
    __syncthreads();                                         // the grouped keys are read back by other lanes below
    unsigned long long *s_win = s_keys;
    int b0 = 0;
    while (b0 < BIG_NB) {
        const int ws = bucket_start(b0);
        if (ws >= n) break;                                  // only empty buckets are left
        int lo = b0 + 1, hi = BIG_NB;                        // largest b1 with start(b1) - ws <= window (>= b0 + 1)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (s_cur[mid - 1] - ws <= CNT_MAXN) lo = mid; else hi = mid - 1;
        }
        const int b1 = lo;
        const int len = s_cur[b1 - 1] - ws;
        {                                                    // <= 4 keys per thread, all loads in flight together
            unsigned long long kk[CNT_MAXN / SORT_THREADS];
#pragma unroll
            for (int q = 0; q < CNT_MAXN / SORT_THREADS; ++q) {
                const int i = t + q * SORT_THREADS;
                kk[q] = (i < len) ? grp[ws + i] : 0ull;
            }
#pragma unroll
            for (int q = 0; q < CNT_MAXN / SORT_THREADS; ++q) {
                const int i = t + q * SORT_THREADS;
                if (i < len) s_win[i] = kk[q];
            }
        }
        __syncthreads();
        for (int i = t; i < len; i += SORT_THREADS) {
            const unsigned long long k = s_win[i];
            const int b = bucket_of(k);
            const int bs = bucket_start(b) - ws, be = s_cur[b] - ws;
            int rank = 0;
            for (int j = bs; j < be; ++j) rank += (s_win[j] < k) ? 1 : 0;
            s_out[bs + rank] = k;
        }
        __syncthreads();
        for (int i = t; i < len; i += SORT_THREADS) emit(start + ws + i, s_out[i]);
        b0 = b1;
    }
}

// ---- 4c. per-tile sort of DEEP lists -----------------------------------------------------------------------------------
// tile_sort_count_kernel keeps 2048 keys in LDS; a longer list goes through memory three more times (histogram pass, scatter
// into the scratch copy grouped by bucket - 8-byte stores to 4096 moving cursors, partial lines - and the windows that read it
// back): 52 bytes per key counted at 5 M Gaussians / 1080p (8000-10400 keys per tile), 0.79 of the render's 3.2 ms.  A
// workgroup may use all 160 KiB of a CU's LDS.  Here the keys are staged ONCE (8 bytes each) and what moves afterwards are
// 16-bit indices into the staged keys: grouped by depth bucket, then ranked inside the bucket, then read out in order - 12
// bytes of LDS per key, so a list of up to 12160 keys is read from memory once and leaves as ids (12 bytes of traffic per key).
// A longer list is histogrammed once and then cut into depth windows of at most CAP keys; each window re-reads the segment and
// stages its own keys (8 bytes per key and window; no scatter through memory, nothing written but the ids).  Same order as
// tile_sort_count_kernel: monotone depth -> bucket map, exact (depth, id) rank inside a bucket; piles of equal depths fall back
// to the merge sort (in an LDS window of CAP / 2 keys).  1024 threads: every phase is a chain of LDS round trips, and one
// workgroup is all a CU holds.  Measured at 5 M / 1080p (tools/dbg/sort_trace.py, us per tile): load 6.2, histogram 1.1, scan
// 1.5, grouping 2.2, rank 7.0, read-out 1.8; 789 -> 520 us for the launch.  Tried on top and not kept: the tile's keys held in
// registers through histogram and grouping (same time), keys instead of indices in the grouped array with the staging buffer
// dropped and the rank / read-out steps in lock step over four keys (620 us: 128 registers per thread, spills), persistent
// workgroups that request the next tile's keys while they sort (load 6.2 -> 1.1 us per tile, launch 590-630 us: the
// hardware's own dispatch of 8160 workgroups overlaps their phases better), two workgroups of 6400 keys per CU (765-900 us:
// two windows per tile).
constexpr int DEEP_NB = 4096;
constexpr int DEEP_MAX_BUCKET = 256;
constexpr int DEEP_THREADS = 1024;
constexpr int DEEP_LOADS = 12;                                                // keys a thread has in flight while it stages / filters
// Thresholds on the list CAPACITY per tile (plans size it at 1.5 x the probed lists): the 8-keyframe BA window of the headline
// has 1500 keys per tile and a capacity of 2230 - tile_sort_count_kernel (three 512-thread workgroups per CU, most lists in its
// 2048-key window) sorts it in 71 us, the 5376-key kernel in 123
#ifndef GSX_DEEP_LO_FROM
#define GSX_DEEP_LO_FROM 3500
#endif
#ifndef GSX_DEEP_HI_FROM
#define GSX_DEEP_HI_FROM 5500
#endif
constexpr int DEEP_CAP_LO = 5376, DEEP_CAP_HI = 12160;                        // LDS: 79 KiB (two workgroups per CU) / 159 KiB
constexpr int64_t DEEP_CAP_LO_FROM = GSX_DEEP_LO_FROM, DEEP_CAP_HI_FROM = GSX_DEEP_HI_FROM;   // list capacity per tile from which each is used

// DIAGNOSTIC build only (-DGSX_WG_TRACE, tools/dbg/sort_trace.py): thread 0 of the first 4096 workgroups stamps s_memrealtime
// (100 MHz) at the phase boundaries.  Nothing of this is compiled into the product library.
#ifdef GSX_WG_TRACE
__device__ unsigned long long *g_sort_trace = nullptr;         // [4096 workgroups][8 stamps]
#define GSX_ST(k)                                                                                                          \
    if (g_sort_trace && threadIdx.x == 0 && blockIdx.x < 4096) g_sort_trace[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
}  // namespace
extern "C" int gsx_debug_sort_trace(void *buffer) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_sort_trace), &buffer, sizeof(buffer)) == hipSuccess ? 0 : 1;
}
namespace {
#else
#define GSX_ST(k)
#endif

template <int CAP>
__global__ __launch_bounds__(DEEP_THREADS) void tile_sort_deep_kernel(unsigned long long *__restrict__ entries,
                                                                      unsigned long long *__restrict__ scratch,
                                                                      const int32_t *__restrict__ offsets, int n_tiles,
                                                                      int tile_n_bits, int64_t M_cap, uint32_t id_max,
                                                                      int64_t *__restrict__ isect_ids,
                                                                      int32_t *__restrict__ flatten_ids) {
    static_assert(CAP % 128 == 0 && CAP < 65536 && 12 * CAP + 4 * DEEP_NB + 256 <= 163840, "keys, two index arrays, cursors: one CU's LDS");
    __shared__ __attribute__((aligned(16))) unsigned long long s_a[CAP];     // staged keys
    __shared__ unsigned short s_g[CAP];                                       // indices into s_a, grouped by bucket
    __shared__ unsigned short s_o[CAP];                                       // indices into s_a, in sorted order
    __shared__ int s_cur[DEEP_NB];
    __shared__ unsigned int s_red[2 * (DEEP_THREADS / 64)];
    __shared__ int s_wtot[DEEP_THREADS / 64];
    __shared__ int s_n;
    const int tile = blockIdx.x;
    const int t = threadIdx.x;
    const int64_t start = max((int64_t)0, min((int64_t)offsets[tile], M_cap));
    const int64_t end = max((int64_t)0, min((int64_t)offsets[tile + 1], M_cap));
    const int n = (int)(end - start);
    if (n <= 0) return;
    const int c = tile / n_tiles, tl = tile - c * n_tiles;
    const long long hi_part = ((long long)c << (32 + tile_n_bits)) | ((long long)tl << 32);
    unsigned long long *seg = entries + start;
    auto emit = [&](int64_t o, unsigned long long k) {
        flatten_ids[o] = (int32_t)min((uint32_t)k, id_max);  // never hand an out-of-range gather index on
        if (isect_ids) isect_ids[o] = hi_part | (long long)(k >> 32);
    };
    const bool fits = n <= CAP;
    GSX_ST(0)
    // 1. range of the depth bits; a list that fits is staged on the way (the only time its keys are read).  DEEP_LOADS keys
    // per thread are requested before the first is used: one workgroup per CU has nobody to hide a memory round trip behind
    unsigned int dmin = 0xffffffffu, dmax = 0u;
    if (fits) {
        for (int i0 = 0; i0 < n; i0 += DEEP_LOADS * DEEP_THREADS) {
            unsigned long long kk[DEEP_LOADS];
#pragma unroll
            for (int q = 0; q < DEEP_LOADS; ++q) {
                const int i = i0 + q * DEEP_THREADS + t;
                kk[q] = (i < n) ? seg[i] : 0ull;
            }
#pragma unroll
            for (int q = 0; q < DEEP_LOADS; ++q) {
                const int i = i0 + q * DEEP_THREADS + t;
                if (i < n) {
                    s_a[i] = kk[q];
                    const unsigned int d = (unsigned int)(kk[q] >> 32);
                    dmin = min(dmin, d); dmax = max(dmax, d);
                }
            }
        }
    } else {
        // one block in four is sampled for the range: keys outside it clamp to the end buckets (the map stays monotone)
#pragma unroll 4
        for (int i = t; i < n; i += 4 * DEEP_THREADS) {
            const unsigned int d = (unsigned int)(seg[i] >> 32);
            dmin = min(dmin, d); dmax = max(dmax, d);
        }
    }
    for (int i = t; i < DEEP_NB; i += DEEP_THREADS) s_cur[i] = 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dmin = min(dmin, (unsigned int)__shfl_xor((int)dmin, off, 64));
        dmax = max(dmax, (unsigned int)__shfl_xor((int)dmax, off, 64));
    }
    if ((t & 63) == 0) { s_red[t >> 6] = dmin; s_red[DEEP_THREADS / 64 + (t >> 6)] = dmax; }
    __syncthreads();
    GSX_ST(1)
    dmin = 0xffffffffu; dmax = 0u;
#pragma unroll
    for (int w = 0; w < DEEP_THREADS / 64; ++w) { dmin = min(dmin, s_red[w]); dmax = max(dmax, s_red[DEEP_THREADS / 64 + w]); }
    const float fmin_ = __uint_as_float(dmin), fmax_ = __uint_as_float(dmax);
    const float range = fmax_ - fmin_;
    const float scale = (range > 0.0f) ? (float)(DEEP_NB - 1) / range : 0.0f;
    auto bucket_of = [&](unsigned long long k) -> int {
        const float d = __uint_as_float((unsigned int)(k >> 32));
        const int b = (int)((d - fmin_) * scale);
        return min(max(b, 0), DEEP_NB - 1);
    };
    // 2. histogram over the whole list, exclusive scan: s_cur[b] = keys in the buckets before b
    if (fits) {
        for (int i = t; i < n; i += DEEP_THREADS) atomicAdd(&s_cur[bucket_of(s_a[i])], 1);
    } else {
        for (int i0 = 0; i0 < n; i0 += DEEP_LOADS * DEEP_THREADS) {
            unsigned long long kk[DEEP_LOADS];
#pragma unroll
            for (int q = 0; q < DEEP_LOADS; ++q) {
                const int i = i0 + q * DEEP_THREADS + t;
                kk[q] = (i < n) ? seg[i] : ~0ull;
            }
#pragma unroll
            for (int q = 0; q < DEEP_LOADS; ++q)
                if (i0 + q * DEEP_THREADS + t < n) atomicAdd(&s_cur[bucket_of(kk[q])], 1);
        }
    }
    __syncthreads();
    GSX_ST(2)
    const bool piles = bucket_scan<DEEP_NB, DEEP_THREADS>(s_cur, s_wtot, DEEP_MAX_BUCKET);
    GSX_ST(3)
    if (piles) {                                                 // piles of equal depths: merge sort, its two buffers in s_a
        const unsigned long long *sorted = merge_sort_tile<DEEP_THREADS>(seg, scratch + start, n, s_a, CAP / 2);
        for (int i = t; i < n; i += DEEP_THREADS) emit(start + i, sorted[i]);
        return;
    }
    // 3. windows of whole buckets [b0, b1) holding at most CAP keys.  The returning cursor of a bucket starts at the number of
    // keys before the bucket and ends at the next bucket's start; positions inside the window = cursor value - ws
    int b0 = 0;
    while (b0 < DEEP_NB) {
        const int ws = s_cur[b0];                                    // keys before the window (bucket b0 is untouched so far)
        if (ws >= n) break;
        int b1;
        if (fits) {
            b1 = DEEP_NB;
        } else {
            int lo = b0 + 1, hi = DEEP_NB;                           // largest b1 with (keys before b1) - ws <= CAP
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                const int upto = (mid < DEEP_NB) ? s_cur[mid] : n;
                if (upto - ws <= CAP) lo = mid; else hi = mid - 1;
            }
            b1 = lo;
        }
        const int len = ((b1 < DEEP_NB) ? s_cur[b1] : n) - ws;
        if (t == 0) s_n = 0;
        __syncthreads();                                             // every thread has read the window's bounds
        if (fits) {
            for (int i = t; i < n; i += DEEP_THREADS)
                s_g[atomicAdd(&s_cur[bucket_of(s_a[i])], 1)] = (unsigned short)i;
        } else {
            for (int i0 = 0; i0 < n; i0 += DEEP_LOADS * DEEP_THREADS) {
                unsigned long long kk[DEEP_LOADS];
#pragma unroll
                for (int q = 0; q < DEEP_LOADS; ++q) {
                    const int i = i0 + q * DEEP_THREADS + t;
                    kk[q] = (i < n) ? seg[i] : ~0ull;
                }
#pragma unroll
                for (int q = 0; q < DEEP_LOADS; ++q) {
                    const int b = bucket_of(kk[q]);
                    if (i0 + q * DEEP_THREADS + t < n && b >= b0 && b < b1) {
                        const int slot = atomicAdd(&s_n, 1);         // where the key is staged (any order)
                        s_a[slot] = kk[q];
                        s_g[atomicAdd(&s_cur[b], 1) - ws] = (unsigned short)slot;
                    }
                }
            }
        }
        __syncthreads();
        GSX_ST(4)
        for (int p = t; p < len; p += DEEP_THREADS) {
            const int i = s_g[p];
            const unsigned long long k = s_a[i];
            const int b = bucket_of(k);
            const int bs = (b > b0 ? s_cur[b - 1] : ws) - ws, be = s_cur[b] - ws;
            int rank = 0;
#pragma unroll 4
            for (int j = bs; j < be; ++j) rank += (s_a[s_g[j]] < k) ? 1 : 0;
            s_o[bs + rank] = (unsigned short)i;
        }
        __syncthreads();
        GSX_ST(5)
        for (int p = t; p < len; p += DEEP_THREADS) emit(start + ws + p, s_a[s_o[p]]);
        __syncthreads();
        GSX_ST(6)
        b0 = b1;
    }
}

int bit_length(uint32_t v) {
    int n = 0;
    while (v) { ++n; v >>= 1; }
    return n;
}

struct BinLayout {
    int64_t diff_off, cursor_off, entries_off, scratch_off, matrix_off, total;
};

BinLayout bin_layout(int64_t C, int tile_w, int tile_h, int64_t M_cap) {
    BinLayout L;
    const int64_t G = (int64_t)(tile_w + 1) * (tile_h + 1);
    const int64_t T = C * tile_w * tile_h;
    L.diff_off = 0;
    L.cursor_off = gsx_align256(C * G * 4);
    L.entries_off = L.cursor_off + gsx_align256((T + 1) * 4);
    L.scratch_off = L.entries_off + gsx_align256((M_cap > 0 ? M_cap : 1) * 8);
    L.matrix_off = L.scratch_off + gsx_align256((M_cap > 0 ? M_cap : 1) * 8);
    L.total = L.matrix_off + gsx_align256(T * (int64_t)GB_MAX * 4) + 256;     // [C][<= GB_MAX workgroups][tiles] counts
    L.total = gsx_align256(L.total);
    return L;
}

}  // namespace

extern "C" int64_t gsx_isect_bin_workspace_bytes(int64_t C, int tile_w, int tile_h, int64_t M_cap) {
    return bin_layout(C, tile_w, tile_h, M_cap).total;
}

extern "C" int64_t gsx_isect_bin_workspace_bytes_n(int64_t C, int64_t N, int tile_w, int tile_h, int64_t M_cap) {
    return bin_layout(C, tile_w, tile_h, M_cap).total + (C > 0 && N > 0 ? C * N * 16 : 0) + 256;
}

static int isect_bin_sort_impl(const float *means2d, const int32_t *radii, const float *depths, const uint32_t *trec, int64_t N,
                               int64_t C, int tile_w, int tile_h, int64_t M_cap, int32_t *offsets, int64_t *M_dev,
                               int32_t *status, int64_t *isect_ids, int32_t *flatten_ids, int32_t *tile_order,
                               void *workspace, int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(means2d && radii && depths && offsets && M_dev && status && N >= 0 && C >= 1);
    GSX_CHECK_ARG(tile_w > 0 && tile_h > 0 && M_cap >= 0 && M_cap < ((int64_t)1 << 31));
    GSX_CHECK_ARG(M_cap == 0 || flatten_ids);
    const int64_t n_tiles = (int64_t)tile_w * tile_h;
    const int64_t T = C * n_tiles;
    GSX_CHECK_ARG(T < ((int64_t)1 << 30) && C < 65536 && C * N < ((int64_t)1 << 31));
    const int64_t G = (int64_t)(tile_w + 1) * (tile_h + 1);
    GSX_CHECK_ARG(G * 4 + 8192 <= 65536);  // the difference grid must fit the default 64 KiB LDS window next to the scan buffer
    const BinLayout L = bin_layout(C, tile_w, tile_h, M_cap);
    if (!workspace || workspace_bytes < L.total) {
        gsx_set_error("gsx_isect_bin_sort: workspace too small (%lld < %lld)", (long long)workspace_bytes,
                      (long long)L.total);
        return GSX_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int *diff = (int *)(ws + L.diff_off);
    int32_t *cursor = (int32_t *)(ws + L.cursor_off);
    unsigned long long *entries = (unsigned long long *)(ws + L.entries_off);
    unsigned long long *scratch = (unsigned long long *)(ws + L.scratch_off);
    {
        // count matrix -> column scan -> tile scan -> placement; the Gaussians of a workgroup grow with N so that the
        // matrix never has more than GB_MAX rows per camera
        int32_t *cnt = (int32_t *)(ws + L.matrix_off);
        int64_t items = BIN_ITEMS;
        while ((N + BIN_THREADS * items - 1) / (BIN_THREADS * items) > GB_MAX) items *= 2;
        // large maps: spatial pre-sort of the visible instances first (see 3c)
        const int sw = (tile_w + SUPER - 1) / SUPER, S = sw * ((tile_h + SUPER - 1) / SUPER);
        const int64_t rec_cap = C * N;
        const bool presort_ok = N > 0 && M_cap > 0 && workspace_bytes >= L.total + rec_cap * 16 && T <= 16000 &&
                                C * G * 4 <= 65536 && C <= 255 && tile_w < 4096 && tile_h < 4096 &&
                                S * (int64_t)sizeof(int) <= 65536;
        // measured (MI355X, whole tile-list build, direct vs pre-sorted): 100 k x 8 cameras 138 vs 145 us, 250 k x 8: 209 vs
        // 178, 1M x 1: 143 vs 121, 500 k x 8: 330 vs 262, 2M x 8: 1774 vs 1123, 5M at 1080p: count + placement 1.70 ms vs
        // 0.6 - the pre-sort pays from about a million instances on
        const bool presort = presort_ok && C * N >= ((int64_t)1 << 20);
        if (presort) {
            const unsigned gblocks = (unsigned)((N + BIN_THREADS * items - 1) / (BIN_THREADS * items));
            int32_t *coff = cursor;                          // [C * S + 1] (S <= tiles per camera)
            int64_t *n_inst = (int64_t *)diff;
            PreRec *recs = (PreRec *)(ws + L.total);         // the tail behind the base layout (256-byte aligned)
            hipLaunchKernelGGL(coarse_count_kernel, dim3(gblocks, (unsigned)C), dim3(BIN_THREADS), (size_t)(S * 4), st,
                               means2d, radii, N, tile_w, tile_h, (int)items, sw, S, cnt, trec);
            GSX_CHECK_LAUNCH();
            hipLaunchKernelGGL(column_scan_kernel<false>, dim3((unsigned)((S + 63) / 64), (unsigned)C), dim3(64 * CS_GROUPS), 0,
                               st, cnt, (int)gblocks, S, coff, (int32_t *)nullptr);
            GSX_CHECK_LAUNCH();
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, (int)(C * S), C * N, coff, n_inst, status,
                               (int32_t *)nullptr);
            GSX_CHECK_LAUNCH();
            hipLaunchKernelGGL(coarse_place_kernel, dim3(gblocks, (unsigned)C), dim3(BIN_THREADS), (size_t)(S * 4), st,
                               means2d, radii, depths, N, tile_w, tile_h, (int)items, sw, S, rec_cap, coff, cnt, recs, trec);
            GSX_CHECK_LAUNCH();
            // tile level: chunks of instances, at most GB_MAX of them over the record capacity
            int64_t chunk = BIN_THREADS * 4;
            const int64_t inst_cap = rec_cap;
            while ((inst_cap + chunk - 1) / chunk > GB_MAX) chunk *= 2;
            const unsigned gb2 = (unsigned)((inst_cap + chunk - 1) / chunk);
            hipLaunchKernelGGL(fine_count_kernel, dim3(gb2), dim3(FINE_THREADS), (size_t)(C * G * 4), st, recs, n_inst,
                               (int)chunk, tile_w, tile_h, (int)C, cnt);
            GSX_CHECK_LAUNCH();
            hipLaunchKernelGGL(column_scan_kernel<false>, dim3((unsigned)((T + 63) / 64), 1u), dim3(64 * CS_GROUPS), 0, st, cnt,
                               (int)gb2, (int)T, offsets, (int32_t *)nullptr);
            GSX_CHECK_LAUNCH();
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, (int)T, M_cap, offsets, M_dev, status, tile_order);
            GSX_CHECK_LAUNCH();
            if (M_cap > 0) {
                hipLaunchKernelGGL(fine_place_kernel, dim3(gb2), dim3(FINE_THREADS), (size_t)(T * 4), st, recs, n_inst,
                                   (int)chunk, tile_w, (int)n_tiles, (int)T, M_cap, offsets, cnt, entries);
                GSX_CHECK_LAUNCH();
            }
        } else {
            // Gaussians per thread (measured, MI355X): the grid has to cover the chip (>= 512 workgroups over all
            // cameras: 100 k Gaussians at one camera want ONE Gaussian per thread, 36 vs 50 us for the whole tile-list
            // build), beyond that fatter workgroups write longer runs per tile, up to 8 per thread; the count matrix caps
            // the workgroups per camera at GB_MAX
            auto blocks_of = [&](int64_t it) { return (N + BIN_THREADS * it - 1) / (BIN_THREADS * it); };
            items = 1;
            while (items < 8 && C * blocks_of(items * 2) >= 512) items *= 2;
            while (blocks_of(items) > GB_MAX) items *= 2;
            const unsigned gblocks = (unsigned)((N + BIN_THREADS * items - 1) / (BIN_THREADS * items));
            if (N > 0) {
                hipLaunchKernelGGL(count_matrix_kernel, dim3(gblocks, (unsigned)C), dim3(BIN_THREADS),
                                   (size_t)(n_tiles * 4), st, means2d, radii, N, tile_w, tile_h, (int)items, cnt, trec);
                GSX_CHECK_LAUNCH();
                hipLaunchKernelGGL(column_scan_kernel<false>, dim3((unsigned)((n_tiles + 63) / 64), (unsigned)C),
                                   dim3(64 * CS_GROUPS), 0, st, cnt, (int)gblocks, (int)n_tiles, offsets, (int32_t *)nullptr);
                GSX_CHECK_LAUNCH();
            } else {
                if (!gsx_zero_async(offsets, T, st)) return GSX_E_LAUNCH;
            }
            hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, (int)T, M_cap, offsets, M_dev, status,
                               tile_order);
            GSX_CHECK_LAUNCH();
            if (N > 0 && M_cap > 0) {
                hipLaunchKernelGGL(place_kernel, dim3(gblocks, (unsigned)C), dim3(BIN_THREADS), (size_t)(n_tiles * 4),
                                   st, means2d, radii, depths, N, tile_w, tile_h, (int)items, M_cap, offsets, cnt,
                                   entries, trec);
                GSX_CHECK_LAUNCH();
            }
        }
    }
    if (N > 0 && M_cap > 0) {
        // Tile sizes are only known on the device (sync-free): one launch over all tiles, each workgroup picks its
        // regime from its tile's size (counting sort in LDS, streamed counting sort, merge sort for degenerate depths)
        const uint32_t id_max = (uint32_t)(C * N - 1);
        const int tnb = bit_length((uint32_t)n_tiles);
        // lists deeper than the 2048 keys tile_sort_count_kernel holds in LDS (judged by the capacity per tile - the sizes
        // themselves are only known on the device): the kernels with a 4032- / 9152-key window (two / one workgroup per CU)
        const int64_t per_tile = M_cap / (T > 0 ? T : 1);
        if (per_tile > DEEP_CAP_HI_FROM)
            hipLaunchKernelGGL(tile_sort_deep_kernel<DEEP_CAP_HI>, dim3((unsigned)T), dim3(DEEP_THREADS), 0, st, entries,
                               scratch, offsets, (int)n_tiles, tnb, M_cap, id_max, isect_ids, flatten_ids);
        else if (per_tile > DEEP_CAP_LO_FROM)
            hipLaunchKernelGGL(tile_sort_deep_kernel<DEEP_CAP_LO>, dim3((unsigned)T), dim3(DEEP_THREADS), 0, st, entries,
                               scratch, offsets, (int)n_tiles, tnb, M_cap, id_max, isect_ids, flatten_ids);
        else
            hipLaunchKernelGGL(tile_sort_count_kernel, dim3((unsigned)T), dim3(SORT_THREADS), 0, st, entries, scratch,
                               offsets, (int)n_tiles, tnb, M_cap, id_max, isect_ids, flatten_ids);
        GSX_CHECK_LAUNCH();
    }
    return GSX_OK;
}

extern "C" int gsx_isect_bin_sort(const float *means2d, const int32_t *radii, const float *depths, int64_t N, int64_t C,
                                  int tile_w, int tile_h, int64_t M_cap, int32_t *offsets, int64_t *M_dev,
                                  int32_t *status, int64_t *isect_ids, int32_t *flatten_ids, int32_t *tile_order,
                                  void *workspace, int64_t workspace_bytes, void *stream) {
    return isect_bin_sort_impl(means2d, radii, depths, nullptr, N, C, tile_w, tile_h, M_cap, offsets, M_dev, status, isect_ids,
                               flatten_ids, tile_order, workspace, workspace_bytes, stream);
}

// The same over rectangles the PROJECTION prepared: rects = uint32 [C * N] packed by gsx_project_fwd_rects (tile_rect.h) - the
// reference's squares, or tight ones (an instance listed only in the tiles of its 3-sigma square that hold a pixel centre inside
// the box of its alpha >= 1/255 ellipse) if the projection ran with GSX_PROJ_TILE_EXACT.  means2d / radii are not read.
extern "C" int gsx_isect_bin_sort_rects(const uint32_t *rects, const float *depths, int64_t N, int64_t C, int tile_w, int tile_h,
                                        int64_t M_cap, int32_t *offsets, int64_t *M_dev, int32_t *status, int64_t *isect_ids,
                                        int32_t *flatten_ids, int32_t *tile_order, void *workspace, int64_t workspace_bytes,
                                        void *stream) {
    GSX_CHECK_ARG(rects != nullptr && depths != nullptr && tile_w < 256 && tile_h < 256);
    // (means2d / radii only pass the argument check of the shared body)
    return isect_bin_sort_impl((const float *)rects, (const int32_t *)rects, depths, rects, N, C, tile_w, tile_h, M_cap, offsets,
                               M_dev, status, isect_ids, flatten_ids, tile_order, workspace, workspace_bytes, stream);
}

// =====================================================================================================================
// Fused front of a render for the launch plans (gslam_amd/plan.py): K1 + K3..K7 in FOUR launches instead of seven.
//
//   1. front_project_kernel  : the projection of project.hip (same arithmetic: project_core.h) for a chunk of 1024 x items
//                              Gaussians per workgroup, all cameras; while the rectangle of a visible instance is in
//                              registers it is counted into the workgroup's per-tile histogram (LDS) - the workgroup's row
//                              of the count matrix - and appended to the workgroup's segment of 16-byte instance records
//                              (packed rectangle, depth bits, flatten id).  Culled rows write radii = 0 / tiles = 0 only
//                              when the caller says nobody reads them (GSX_PROJ_SKIP_CULLED: two thirds of a 500 k map
//                              are outside a 640x480 frustum; 128 B of zero stores each)
//   2. column_scan_kernel    : as before (each row's base inside every tile, per-tile totals)
//   3. front_place_kernel    : every workgroup scans the T per-tile totals in LDS itself (the one-workgroup tile_scan launch
//                              is gone; workgroup 0 publishes offsets / M / status for the kernels behind), then places the
//                              instance records of ITS row that touch ITS stripe of tile rows through LDS cursors.  The
//                              stripe index is blockIdx % 8: workgroups that share an XCD (round-robin dispatch, a speed
//                              assumption only) write the same contiguous eighth of the entry buffer, so the 8-byte
//                              scattered stores of one 128-byte line meet in one L2 instead of going out as partial
//                              lines from eight of them
//   4. tile_sort_count_kernel: as before
// Results are those of gsx_project_fwd + gsx_isect_bin_sort (the tile sort fixes the order inside a tile).
// =====================================================================================================================
#include "project_core.h"
#include "pose_chain.h"
#include "pose_math.h"
#include "track_opt.h"
#include "track_tail.h"

// DIAGNOSTIC build only (-DGSX_WG_TRACE, tools/dbg/front_trace.sh): thread 0 of every workgroup of the two front kernels stamps
// s_memrealtime (100 MHz) at its phase boundaries.  Nothing of this is compiled into the product library.
#ifdef GSX_WG_TRACE
__device__ unsigned long long *g_front_trace = nullptr;        // [2 kernels][4096 workgroups][8 stamps]
#define GSX_FT(kern, k)                                                                                                    \
    if (g_front_trace && threadIdx.x == 0 && blockIdx.x < 4096)                                                            \
        g_front_trace[((size_t)(kern) * 4096 + blockIdx.x) * 8 + (k)] = __builtin_amdgcn_s_memrealtime();
extern "C" int gsx_debug_front_trace(void *buffer) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_front_trace), &buffer, sizeof(buffer)) == hipSuccess ? 0 : 1;
}
#else
#define GSX_FT(kern, k)
#endif

namespace {

using namespace gsx_proj;

constexpr int FRONT_THREADS = 1024;
#define GSX_ROW_CURSORS 64                      // = gsx_tsort::ROW_CURSORS (tile_sort_lds.h)
constexpr int FPLACE_THREADS = 256;
#ifndef GSX_FPLACE_PRE
#define GSX_FPLACE_PRE 4
#endif
constexpr int FPLACE_PRE = GSX_FPLACE_PRE;      // instance records a placement thread requests up front
constexpr int FPLACE_ROUND = 2;                 // trips of 256 records per round of the placement walk
constexpr int FPLACE_LIST = FPLACE_ROUND * FPLACE_THREADS;   // LDS list of a round's records that touch the stripe (16 B each)
static_assert(FPLACE_PRE % FPLACE_ROUND == 0, "the prefetched records are whole rounds");
#ifndef GSX_FRONT_STRIPES
#define GSX_FRONT_STRIPES 8
#endif
constexpr int FRONT_STRIPES = GSX_FRONT_STRIPES;

struct FrontArgs {
    const float *means, *quats, *scales, *viewmats, *Ks, *logit_opac, *logit_colors, *log_unc;
    int64_t N;
    int C, W, H, flags, tile_w, tile_h, items, R;
    float eps2d, near_p, far_p;
    int32_t *radii, *tiles, *vis_count;
    float *means2d, *depths, *conics, *rec, *v_rec;
    int32_t *cnt;        // [C][R][n_tiles]
    int32_t *n_inst;     // [C][R]
    PreRec *recs;        // [C][R][1024 * items]
    int compact;         // rec / v_rec rows are indexed by instance slot ((c * R + row) * seg_cap + position), not flatten id
    gsx_bal::Args bal;   // bal.order != nullptr: workgroup R of the launch computes the CU-balanced launch order instead
    // per-frame candidate set (gsx_front_candidates; pose-only plans): what a closure's projection reads instead of culling
    // all N Gaussians again for a pose that moved by a fraction of a pixel
    const struct CandRec *cand;   // [R][seg_cap] pose-independent records of the row's candidates, in Gaussian order
    const int32_t *cand_n;        // [R]
    float *cand_hdr;              // [CAND_HDR]: reference [R | t] of every camera, margins, mode word (see cand_valid)
    // GSX_PROJ_MAP_RECORDS: the candidate area holds a record for EVERY Gaussian (slot = Gaussian index) and this array what
    // the conservative cull reads of each, packed: (mean, largest scale squared).  Closures keep their own cull (one coalesced
    // 16-byte load per Gaussian instead of two 12-byte-stride arrays) and read ONE 64-byte line per survivor instead of
    // gathering 14 floats from five arrays and rebuilding the covariance (15 of the projection's 27 us at 500 k, traced)
    const float4 *cull4;          // [R * seg_cap]
    // near placement (gsx_front_fwd_near): depth bits of every tile's cut-off.  An (instance, tile) pair behind its tile's cut-off is
    // counted in the HIGH half of the count-matrix word (kept out of the placement, room left for it in the tile's segment)
    const uint32_t *tile_cut;     // [C * n_tiles], nullable
    // row keys (gsx_front_fwd_rows): the workgroup leaves its row's keys in its own segment of `row_keys`, grouped by tile, and in
    // its row of the count matrix the word (offset inside the segment << 13 | count) per tile - no column scan, no placement
    // launch: the rasteriser's tile workgroups collect their keys themselves (tile_sort_lds.h gather_tile_keys)
    unsigned long long *row_keys; // [R][row_cap], nullable
    int row_cap;
    int32_t *status;
    unsigned long long *cursor;   // the render's GSX_ROW_CURSORS key counters (M_dev [64]): zeroed here, advanced by the tile workgroups
};

// What the projection needs of one Gaussian that does not depend on the pose: built once per frame for the Gaussians that
// can be visible from ANY pose within the margins of the frame's first pose (gsx_front_candidates), read by every closure
// of the frame.  The covariance is the same float expression (covar_from_rot_scale) the per-closure path evaluates, so a
// closure over candidate records and a closure over the raw map arrays produce the same bits.
struct CandRec {
    float mean[3];
    float S[6];            // world covariance 00 01 02 11 12 22
    float opac, col[3], beta;
    int32_t g;             // Gaussian index
    int32_t pad;
};
static_assert(sizeof(CandRec) == 64, "candidate records are one 64-byte line each");
constexpr int CAND_MAX_CAMS = 16;
constexpr int CAND_HDR = CAND_MAX_CAMS * 12 + 8;   // [c][12] reference [R | t] rows, then rot_max, trans_max, -, -, mode, ...

// Is every camera's current pose within the margins of its reference pose?  p = dR p0 + dt with dR = R R0^T,
// dt = t - dR t0, so |p - p0| <= |dR - I|_F |p0| + |dt|: the candidate cull allowed exactly that displacement.
// Wave-uniform (scalar loads, the same few hundred flops in every wavefront).
__device__ __forceinline__ bool cand_valid(const float *__restrict__ viewmats, const float *__restrict__ hdr, int C) {
    const float rot_max = hdr[CAND_MAX_CAMS * 12], trans_max = hdr[CAND_MAX_CAMS * 12 + 1];
    bool ok = true;
    for (int c = 0; c < C; ++c) {
        const float *V = viewmats + 16 * c, *V0 = hdr + 12 * c;
        float dR[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                dR[i * 3 + j] = V[i * 4 + 0] * V0[j * 4 + 0] + V[i * 4 + 1] * V0[j * 4 + 1] + V[i * 4 + 2] * V0[j * 4 + 2];
        float rot2 = 0.f, dt2 = 0.f;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float e = dR[i * 3 + j] - (i == j ? 1.0f : 0.0f);
                rot2 += e * e;
            }
            const float d = V[i * 4 + 3] - (dR[i * 3 + 0] * V0[3] + dR[i * 3 + 1] * V0[7] + dR[i * 3 + 2] * V0[11]);
            dt2 += d * d;
        }
        // (NaN poses compare false: not valid -> the full path, which culls everything)
        ok = ok && (rot2 * 1.001f <= rot_max * rot_max) && (dt2 * 1.001f <= trans_max * trans_max);
    }
    return ok;
}

// Position of every flagged thread of the workgroup among the flagged ones, IN THREAD ORDER (a deterministic, monotone
// compaction: ballot prefix inside the wavefront, wavefront bases from a scan of the per-wavefront counts in LDS), and
// their number.  Two barriers; must be reached by every thread.  Monotone on purpose: slots follow the flatten ids, so a
// sort key (depth, slot) orders ties exactly like (depth, flatten id).
__device__ __forceinline__ int ordered_position(bool flag, int *s_wcnt, int &total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long m = __ballot(flag);
    if (lane == 0) s_wcnt[wave] = __popcll(m);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FRONT_THREADS / 64; ++w) {
        const int cw = s_wcnt[w];
        base += (w < wave) ? cw : 0;
        tot += cw;
    }
    __syncthreads();
    total = tot;
    return base + __popcll(m & ((1ull << lane) - 1ull));
}

// Upper bound of |R|_2^2 = lambda_max(R R^T) by its largest absolute row sum (Gershgorin): 1 for a rotation - a third of
// |R|_F^2, which the cull used until round 4 and which kept 260 k of a 500 k map for the exact projection where 170 k are
// visible - and still an upper bound for whatever else the caller hands in as a view matrix.  0.1 % on top for the rounding
// of the six products.
__device__ __forceinline__ float rot_norm2_bound(const float *R) {
    const float g00 = (R[0] * R[0] + R[1] * R[1]) + R[2] * R[2], g11 = (R[3] * R[3] + R[4] * R[4]) + R[5] * R[5];
    const float g22 = (R[6] * R[6] + R[7] * R[7]) + R[8] * R[8];
    const float g01 = fabsf((R[0] * R[3] + R[1] * R[4]) + R[2] * R[5]), g02 = fabsf((R[0] * R[6] + R[1] * R[7]) + R[2] * R[8]);
    const float g12 = fabsf((R[3] * R[6] + R[4] * R[7]) + R[5] * R[8]);
    return 1.001f * fmaxf(fmaxf((g00 + g01) + g02, (g01 + g11) + g12), (g02 + g12) + g22);
}

// Upper bound of z^2 |J|_2^2 = z^2 lambda_max(J J^T) over the frustum (J: the 2x3 perspective Jacobian with x / z, y / z clamped to
// lx, ly): z^2 J J^T = [[fx^2 (1 + a^2), fx fy a b], [fx fy a b, fy^2 (1 + b^2)]], largest absolute row sum at a = lx, b = ly.
// (Until round 4: the trace, |J|_F^2 - 1.5 times this at 640x480 / 525.)
__device__ __forceinline__ float jac_norm2_bound(float fx, float fy, float lx, float ly) {
    const float off = fabsf(fx * fy) * (lx * ly);
    return 1.001f * fmaxf(fx * fx * (1.0f + lx * lx) + off, fy * fy * (1.0f + ly * ly) + off);
}

// Conservative screen-space cull of one Gaussian for one camera from its mean and its largest scale alone (no covariance
// algebra): true only if the full projection is CERTAIN to cull it - same near / far comparison on the same expression, and
// a bounding box test with an upper bound of the radius: lambda_max(J Sc J^T) <= |J|_2^2 |R|_2^2 s_max^2 (KJ: jac_norm2_bound,
// RF: rot_norm2_bound; lx, ly: the frustum clamp of tx / z, ty / z), v1 = lambda_max of the blurred covariance <= that + eps,
// radius <= 3 sqrt(v1) + 1; one per cent and one pixel of slack on top cover the float rounding of this estimate.
__device__ __forceinline__ bool surely_culled(const float mean[3], float smax2, const Cam &cam, float RF, float KJ,
                                              int W, int H, float eps2d, float near_p, float far_p) {
    const float *R = cam.R;
    const float x = ((R[0] * mean[0] + R[1] * mean[1]) + R[2] * mean[2]) + cam.t[0];
    const float y = ((R[3] * mean[0] + R[4] * mean[1]) + R[5] * mean[2]) + cam.t[1];
    const float z = ((R[6] * mean[0] + R[7] * mean[1]) + R[8] * mean[2]) + cam.t[2];
    if (z < near_p || z > far_p) return true;            // exactly project_core's test
#if GSX_FAST_CULL
    // hardware reciprocal / square root (1 ulp) in place of the IEEE sequences (~10 instructions each, for every Gaussian of the
    // map in every closure): the estimate below carries one per cent and two pixels of slack
    const float rz = __builtin_amdgcn_rcpf(z);
#else
    const float rz = 1.0f / z;
#endif
    const float pmx = (cam.fx * x) * rz + cam.cx, pmy = (cam.fy * y) * rz + cam.cy;
    const float v1b = 1.01f * (rz * rz) * KJ * RF * smax2 + 2.0f * eps2d + 0.2f;
#if GSX_FAST_CULL
    const float rb = 3.0f * __builtin_amdgcn_sqrtf(v1b) + 2.0f;
#else
    const float rb = 3.0f * sqrtf(v1b) + 2.0f;
#endif
    return (pmx + rb <= 0.0f) || (pmx - rb >= (float)W) || (pmy + rb <= 0.0f) || (pmy - rb >= (float)H);
}

// surely_culled for EVERY pose within the margins of the reference pose `cam`: the camera-space position of the Gaussian may
// move by delta = rot_max |p0| + trans_max in any direction.  True only if the exact projection is certain to cull the
// Gaussian for all of them (z range against near / far; the bounding box test with the largest radius bound and the extreme
// screen positions over the box [p0 - delta, p0 + delta]).
__device__ __forceinline__ bool surely_culled_margin(const float mean[3], float smax2, const Cam &cam, float RF, float KJ,
                                                     int W, int H, float eps2d, float near_p, float far_p, float rot_max,
                                                     float trans_max) {
    const float *R = cam.R;
    const float x = ((R[0] * mean[0] + R[1] * mean[1]) + R[2] * mean[2]) + cam.t[0];
    const float y = ((R[3] * mean[0] + R[4] * mean[1]) + R[5] * mean[2]) + cam.t[1];
    const float z = ((R[6] * mean[0] + R[7] * mean[1]) + R[8] * mean[2]) + cam.t[2];
    const float delta = 1.001f * (rot_max * sqrtf((x * x + y * y) + z * z) + trans_max) + 1e-6f;
    const float zmax = z + delta, zmin = z - delta;
    if (!(zmax >= near_p) || !(zmin <= far_p)) return !(zmax != zmax);      // out of range for every pose (NaN: keep)
    if (!(zmin > fmaxf(near_p, 1e-3f))) return false;                        // may come arbitrarily close: no bound, keep
    const float rzmax = 1.0f / zmin, rzmin = 1.0f / zmax;
    const float v1b = 1.01f * (rzmax * rzmax) * KJ * RF * smax2 + 2.0f * eps2d + 0.2f;
    const float rb = 3.0f * sqrtf(v1b) + 2.0f;
    const float xh = x + delta, xl = x - delta, yh = y + delta, yl = y - delta;
    const float pxh = cam.fx * (xh * (xh >= 0.f ? rzmax : rzmin)) + cam.cx, pxl = cam.fx * (xl * (xl >= 0.f ? rzmin : rzmax)) + cam.cx;
    const float pyh = cam.fy * (yh * (yh >= 0.f ? rzmax : rzmin)) + cam.cy, pyl = cam.fy * (yl * (yl >= 0.f ? rzmin : rzmax)) + cam.cy;
    const float slack = 1.0f + 1e-3f * (fabsf(pxh) + fabsf(pxl) + fabsf(pyh) + fabsf(pyl));   // float rounding of the estimate
    return (pxh + rb + slack <= 0.0f) || (pxl - rb - slack >= (float)W) || (pyh + rb + slack <= 0.0f) ||
           (pyl - rb - slack >= (float)H);
}

// Per-frame candidate set: workgroup b culls ITS chunk of the map (the rows of front_project_kernel) against every camera's
// reference pose with the margins, compacts the survivors IN ORDER and leaves their pose-independent records and their
// count; workgroup 0 also publishes the reference poses and the margins.
template <int ITEMS>
__global__ __launch_bounds__(FRONT_THREADS) void front_candidates_kernel(FrontArgs a, CandRec *__restrict__ cand_out,
                                                                         int32_t *__restrict__ cand_n_out, float rot_max,
                                                                         float trans_max) {
    __shared__ int s_wcnt[FRONT_THREADS / 64];
    const int C = a.C;
    const int seg_cap = FRONT_THREADS * ITEMS;
    const int64_t g0 = (int64_t)blockIdx.x * seg_cap;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < CAND_HDR; i += FRONT_THREADS) {
            float v = 0.f;
            if (i < 12 * C) v = a.viewmats[16 * (i / 12) + (i % 12)];
            else if (i == CAND_MAX_CAMS * 12) v = (a.flags & GSX_PROJ_MAP_RECORDS) ? __builtin_inff() : rot_max;
            else if (i == CAND_MAX_CAMS * 12 + 1) v = (a.flags & GSX_PROJ_MAP_RECORDS) ? __builtin_inff() : trans_max;
            a.cand_hdr[i] = v;
        }
    }
    int n_surv = 0;
    const bool map_recs = (a.flags & GSX_PROJ_MAP_RECORDS) != 0;
    float4 *cull_out = const_cast<float4 *>(a.cull4);
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t g = g0 + it * FRONT_THREADS + threadIdx.x;
        const bool active = g < a.N;
        bool survive = false;
        float mean[3] = {0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
        if (active) {
            mean[0] = a.means[3 * g]; mean[1] = a.means[3 * g + 1]; mean[2] = a.means[3 * g + 2];
            s[0] = a.scales[3 * g]; s[1] = a.scales[3 * g + 1]; s[2] = a.scales[3 * g + 2];
            if (a.flags & GSX_PROJ_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
            const float sm = fmaxf(s[0], fmaxf(s[1], s[2]));
            const float smax2 = sm * sm;
            if (map_recs) {
                survive = true;
                cull_out[g] = make_float4(mean[0], mean[1], mean[2], smax2);
            }
            for (int c = 0; c < C && !map_recs; ++c) {
                Cam cam;
                load_cam(a.viewmats, a.Ks, c, cam);
                const float *R = cam.R;
                const float RF = rot_norm2_bound(R);
                const float tanx = 0.5f * (float)a.W / cam.fx, tany = 0.5f * (float)a.H / cam.fy;
                const float lx = fmaxf(((float)a.W - cam.cx) / cam.fx, cam.cx / cam.fx) + GSX_FOV_SLACK * tanx;
                const float ly = fmaxf(((float)a.H - cam.cy) / cam.fy, cam.cy / cam.fy) + GSX_FOV_SLACK * tany;
                const float KJ = jac_norm2_bound(cam.fx, cam.fy, lx, ly);
                if (!surely_culled_margin(mean, smax2, cam, 1.01f * RF, KJ, a.W, a.H, a.eps2d, a.near_p,
                                          a.far_p, rot_max, trans_max))
                    survive = true;
            }
        }
        int tot;
        __syncthreads();
        const unsigned long long m = __ballot(survive);
        if ((threadIdx.x & 63) == 0) s_wcnt[threadIdx.x >> 6] = __popcll(m);
        __syncthreads();
        int base = 0;
        tot = 0;
#pragma unroll
        for (int w = 0; w < FRONT_THREADS / 64; ++w) {
            const int cw = s_wcnt[w];
            base += (w < (int)(threadIdx.x >> 6)) ? cw : 0;
            tot += cw;
        }
        // (map records: every Gaussian survives, so the ordered position IS the local index: slot = Gaussian index)
        const int pos = n_surv + base + __popcll(m & ((1ull << (threadIdx.x & 63)) - 1ull));
        if (survive) {
            CandRec rec;
            float q[4] = {a.quats[4 * g], a.quats[4 * g + 1], a.quats[4 * g + 2], a.quats[4 * g + 3]};
            QuatRot qr;
            quat_to_rotmat(q, qr);
            float M[9];
            Sym3 S;
            covar_from_rot_scale(qr.R, s, M, S);
            rec.mean[0] = mean[0]; rec.mean[1] = mean[1]; rec.mean[2] = mean[2];
            rec.S[0] = S.a00; rec.S[1] = S.a01; rec.S[2] = S.a02; rec.S[3] = S.a11; rec.S[4] = S.a12; rec.S[5] = S.a22;
            rec.opac = gsx_sigmoid(a.logit_opac[g]);
            rec.col[0] = gsx_sigmoid(a.logit_colors[3 * g]);
            rec.col[1] = gsx_sigmoid(a.logit_colors[3 * g + 1]);
            rec.col[2] = gsx_sigmoid(a.logit_colors[3 * g + 2]);
            rec.beta = (a.flags & GSX_PROJ_BETAS) ? fmaxf(expf(a.log_unc[g]), GSX_BETA_MIN) : 0.f;
            rec.g = (int32_t)g;
            rec.pad = 0;
            float4 *o = reinterpret_cast<float4 *>(cand_out + (int64_t)blockIdx.x * seg_cap + pos);
            const float4 *src = reinterpret_cast<const float4 *>(&rec);
            o[0] = src[0]; o[1] = src[1]; o[2] = src[2]; o[3] = src[3];
        }
        n_surv += tot;
    }
    if (threadIdx.x == 0) cand_n_out[blockIdx.x] = n_surv;
}

template <int ITEMS>
__global__ __launch_bounds__(FRONT_THREADS) void front_project_kernel(FrontArgs a) {
    extern __shared__ __attribute__((aligned(16))) int s_front[];   // [C * n_tiles] counts, [C] instance counters, survivors
    if (blockIdx.x == (unsigned)a.R) {
        // one extra workgroup: the launch order of this render's rasteriser kernels from the tile work of the previous one
        // (tile_balance.h) - ~20 us of one CU hidden behind the projection instead of a launch of its own on the chain
        gsx_bal::run(a.bal, reinterpret_cast<unsigned char *>(s_front));
        return;
    }
    const int n_tiles = a.tile_w * a.tile_h;
    const int C = a.C;
    int *s_cnt = s_front, *s_ninst = s_front + C * n_tiles;
    int *s_wcnt = s_ninst + C;                                                     // [FRONT_THREADS / 64] scratch
    unsigned short *s_list = reinterpret_cast<unsigned short *>(s_wcnt + FRONT_THREADS / 64);   // [1024 * items] local indices
    // [C * n_tiles] cut-offs of the near placement, behind the survivor list (only with a.tile_cut)
    unsigned int *s_cutv = reinterpret_cast<unsigned int *>(s_list + FRONT_THREADS * ITEMS);
    int n_surv = 0;
    GSX_FT(0, 0)
    const int seg_cap = FRONT_THREADS * ITEMS;
    const bool skip_culled = (a.flags & GSX_PROJ_SKIP_CULLED) != 0;
    const int64_t g0 = (int64_t)blockIdx.x * seg_cap;
    // records of the whole map: the cull rows are requested before anything else (LDS clear, pose check) - used only if the
    // pose check below agrees (it does unless a pose is NaN)
    float4 cull_row[ITEMS];
    if ((a.flags & GSX_PROJ_MAP_RECORDS) && a.cull4 != nullptr) {
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int64_t g = g0 + it * FRONT_THREADS + threadIdx.x;
            cull_row[it] = g < a.N ? a.cull4[g] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    for (int i = threadIdx.x; i < C * n_tiles + C; i += FRONT_THREADS) s_front[i] = 0;
    if (a.tile_cut)
        for (int i = threadIdx.x; i < C * n_tiles; i += FRONT_THREADS) s_cutv[i] = a.tile_cut[i];
    __syncthreads();
    GSX_FT(0, 1)
    // per-frame candidate records (pose-only plans): valid while every camera stays within the margins they were built for;
    // otherwise this closure takes the full path below - same results either way, the candidates only save the cull
    const bool use_cand = a.cand != nullptr && cand_valid(a.viewmats, a.cand_hdr, C);
    if (a.cand != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
        a.cand_hdr[CAND_MAX_CAMS * 12 + 4] = use_cand ? 1.0f : 0.0f;            // mode of this closure (front_pose_bwd reads it)
        if (!use_cand) a.cand_hdr[CAND_MAX_CAMS * 12 + 5] += 1.0f;              // closures that fell back (diagnostics)
    }
    // records of the whole map: this closure culls as ever (from the packed cull rows) and reads the survivors' records
    const bool map_recs = use_cand && (a.flags & GSX_PROJ_MAP_RECORDS) != 0 && a.cull4 != nullptr;
    if (use_cand && !map_recs) n_surv = min(max(a.cand_n[blockIdx.x], 0), seg_cap);
    // ---- phase 1: cheap conservative cull; the survivors' local indices are compacted into LDS.  The loads of all ITEMS
    // Gaussians of a thread are issued before any of them is used (one memory round trip for the phase, not ITEMS) ---------
    float pm[ITEMS][3], psm[ITEMS];
    if (!use_cand || map_recs) {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int64_t g = g0 + it * FRONT_THREADS + threadIdx.x;
        const bool active = g < a.N;
        if (map_recs) {
            const float4 cr = cull_row[it];
            pm[it][0] = cr.x; pm[it][1] = cr.y; pm[it][2] = cr.z; psm[it] = cr.w;
        } else {
            pm[it][0] = active ? a.means[3 * g] : 0.f;
            pm[it][1] = active ? a.means[3 * g + 1] : 0.f;
            pm[it][2] = active ? a.means[3 * g + 2] : 0.f;
            psm[it] = active ? fmaxf(a.scales[3 * g], fmaxf(a.scales[3 * g + 1], a.scales[3 * g + 2])) : 0.f;
        }
    }
    // camera constants once per camera (not per item): the loop over the thread's items is the inner one
    bool survive[ITEMS];
    float smax2[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        survive[it] = false;
        float sm = psm[it];
        if (map_recs) { smax2[it] = sm; continue; }           // the cull row holds the square of the (exponentiated) scale
#if GSX_FAST_CULL
        if (a.flags & GSX_PROJ_LOG_SCALES) sm = __builtin_amdgcn_exp2f(sm * 1.4426950408889634f) * 1.00001f;
#else
        if (a.flags & GSX_PROJ_LOG_SCALES) sm = expf(sm);
#endif
        smax2[it] = sm * sm;
    }
    for (int c = 0; c < C; ++c) {
        Cam cam;
        load_cam(a.viewmats, a.Ks, c, cam);
        const float *R = cam.R;
        const float RF = rot_norm2_bound(R);
        const float tanx = 0.5f * (float)a.W / cam.fx, tany = 0.5f * (float)a.H / cam.fy;
        const float lx = fmaxf(((float)a.W - cam.cx) / cam.fx, cam.cx / cam.fx) + GSX_FOV_SLACK * tanx;
        const float ly = fmaxf(((float)a.H - cam.cy) / cam.fy, cam.cy / cam.fy) + GSX_FOV_SLACK * tany;
        const float KJ = jac_norm2_bound(cam.fx, cam.fy, lx, ly);
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int64_t g = g0 + it * FRONT_THREADS + threadIdx.x;
            if (g < a.N) {
                const float mean[3] = {pm[it][0], pm[it][1], pm[it][2]};
                const bool out = surely_culled(mean, smax2[it], cam, RF, KJ, a.W, a.H, a.eps2d, a.near_p, a.far_p);
                if (!out) survive[it] = true;
                else if (!skip_culled) survive[it] = true;   // rows of culled instances are wanted as zeros: full path writes them
                else {
                    const int64_t idx = (int64_t)c * a.N + g;
                    if (a.radii) a.radii[idx] = 0;
                    if (a.tiles) a.tiles[idx] = 0;
                }
            }
        }
    }
    if constexpr (ITEMS <= 4) {
        // ONE ordered compaction for all items of the thread (two barriers, not two per item): the per-wavefront counts of the
        // items travel in the bytes of one word; the list is ordered by local index = item * 1024 + thread, as before
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned long long m[ITEMS];
        unsigned int packed = 0u;
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            const int64_t g = g0 + it * FRONT_THREADS + threadIdx.x;
            if (g < a.N && !survive[it] && a.vis_count) a.vis_count[g] = 0;
            m[it] = __ballot(survive[it]);
            packed |= (unsigned int)__popcll(m[it]) << (8 * it);
        }
        if (lane == 0) s_wcnt[wave] = (int)packed;
        __syncthreads();
        int base[ITEMS], tot[ITEMS];
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) base[it] = tot[it] = 0;
#pragma unroll
        for (int w = 0; w < FRONT_THREADS / 64; ++w) {
            const unsigned int cw = (unsigned int)s_wcnt[w];
#pragma unroll
            for (int it = 0; it < ITEMS; ++it) {
                const int cnt = (int)((cw >> (8 * it)) & 0xffu);
                base[it] += (w < wave) ? cnt : 0;
                tot[it] += cnt;
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITEMS; ++it) {
            if (survive[it])
                s_list[n_surv + base[it] + __popcll(m[it] & ((1ull << lane) - 1ull))] = (unsigned short)(it * FRONT_THREADS + threadIdx.x);
            n_surv += tot[it];
        }
    } else {
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int loc = it * FRONT_THREADS + threadIdx.x;
        const int64_t g = g0 + loc;
        if (g < a.N && !survive[it] && a.vis_count) a.vis_count[g] = 0;
        int tot;
        const int pos = ordered_position(survive[it], s_wcnt, tot);
        if (survive[it]) s_list[n_surv + pos] = (unsigned short)loc;
        n_surv += tot;
    }
    }
    }   // !use_cand || map_recs
    __syncthreads();
    GSX_FT(0, 7)
#ifdef GSX_WG_TRACE
    if (g_front_trace && threadIdx.x == 0) g_front_trace[((size_t)4096 + blockIdx.x) * 8 + 4] = (unsigned long long)n_surv;
#endif
    // ---- phase 2: the projection proper, dense over the survivors (whole wavefronts of real work) ---------------------------
    for (int s0 = 0; s0 < n_surv; s0 += FRONT_THREADS) {
        const int si = s0 + threadIdx.x;
        const bool active = si < n_surv;
        int64_t g = g0;
        float mean[3] = {0.f, 0.f, 0.f};
        float opac = 0.f, col[3] = {0.f, 0.f, 0.f}, beta = 0.f;
        Sym3 S = {1.f, 0.f, 0.f, 1.f, 0.f, 1.f};
        // candidate record of this thread (candidate mode): the si-th of the row, or - records of the whole map - the one of
        // the si-th survivor of this closure's cull
        const int64_t cslot = (int64_t)blockIdx.x * seg_cap + (map_recs ? (active ? (int)s_list[si] : 0) : si);
        if (use_cand) {
            if (active) {
                const float4 *src = reinterpret_cast<const float4 *>(a.cand + cslot);
                const float4 r0 = src[0], r1 = src[1], r2 = src[2], r3 = src[3];
                mean[0] = r0.x; mean[1] = r0.y; mean[2] = r0.z;
                S.a00 = r0.w; S.a01 = r1.x; S.a02 = r1.y; S.a11 = r1.z; S.a12 = r1.w; S.a22 = r2.x;
                opac = r2.y; col[0] = r2.z; col[1] = r2.w; col[2] = r3.x; beta = r3.y;
                g = (int64_t)__float_as_int(r3.z);
            }
        } else {
            float q[4] = {1.f, 0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
            g = g0 + (active ? (int)s_list[si] : 0);
            if (active) {
                mean[0] = a.means[3 * g]; mean[1] = a.means[3 * g + 1]; mean[2] = a.means[3 * g + 2];
                q[0] = a.quats[4 * g]; q[1] = a.quats[4 * g + 1]; q[2] = a.quats[4 * g + 2]; q[3] = a.quats[4 * g + 3];
                s[0] = a.scales[3 * g]; s[1] = a.scales[3 * g + 1]; s[2] = a.scales[3 * g + 2];
                if (a.flags & GSX_PROJ_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
                opac = gsx_sigmoid(a.logit_opac[g]);
                col[0] = gsx_sigmoid(a.logit_colors[3 * g]);
                col[1] = gsx_sigmoid(a.logit_colors[3 * g + 1]);
                col[2] = gsx_sigmoid(a.logit_colors[3 * g + 2]);
                if (a.flags & GSX_PROJ_BETAS) beta = fmaxf(expf(a.log_unc[g]), GSX_BETA_MIN);
            }
            QuatRot qr;
            quat_to_rotmat(q, qr);
            float M[9];
            covar_from_rot_scale(qr.R, s, M, S);
        }
        if (__float_as_uint(S.a00) != 0x12345u && s0 == 0) { GSX_FT(0, 6) }
        int n_vis = 0;
        for (int c = 0; c < C; ++c) {
            Cam cam;
            load_cam(a.viewmats, a.Ks, c, cam);
            const int64_t idx = (int64_t)c * a.N + g;
            Proj p;
            int32_t radius_i = 0;
            float mx = 0.f, my = 0.f, depth = 0.f, con0 = 0.f, con1 = 0.f, con2 = 0.f;
            if (active && project_core(mean, S, cam, a.W, a.H, a.eps2d, a.near_p, a.far_p, p)) {
                const float pmx = (cam.fx * p.pc[0]) * p.rz + cam.cx, pmy = (cam.fy * p.pc[1]) * p.rz + cam.cy;
                const float b00 = p.c00 + a.eps2d, b11 = p.c11 + a.eps2d;
                const float b = 0.5f * (b00 + b11);
                const float v1 = b + sqrtf(fmaxf(GSX_RADIUS_FLOOR, b * b - p.det));
                const float radius = ceilf(GSX_RADIUS_SIGMA * sqrtf(v1));
                const bool keep = !(radius <= 0.0f) &&
                                  !(pmx + radius <= 0.0f || pmx - radius >= (float)a.W || pmy + radius <= 0.0f ||
                                    pmy - radius >= (float)a.H);
                if (keep) {
                    radius_i = (int32_t)radius;
                    mx = pmx; my = pmy; depth = p.pc[2];
                    con0 = p.conic[0]; con1 = p.conic[1]; con2 = p.conic[2];
                }
            }
            const bool vis = radius_i > 0;
            Rect r = {0, 0, 0, 0};
            if (vis) r = tile_rect(mx, my, radius_i, a.tile_w, a.tile_h);
            const int ref_area = (r.x1 > r.x0 && r.y1 > r.y0) ? (r.y1 - r.y0) * (r.x1 - r.x0) : 0;   // what the reference lists
            if (vis && (a.flags & GSX_PROJ_TILE_EXACT)) r = tighten_rect(r, mx, my, con0, con1, con2, opac);
            const bool has = (r.x1 > r.x0) && (r.y1 > r.y0);
            // slot of this instance in the workgroup's segment of camera c: monotone in the flatten id (ordered compaction)
            const int run = s_ninst[c];
            int tot;
            const int pos = run + ordered_position(has, s_wcnt, tot);
            if (s0 == 0) { GSX_FT(0, 2) }
            if (threadIdx.x == 0) s_ninst[c] = run + tot;      // read again only behind the barrier that ends the trip
            const int64_t slot = ((int64_t)c * a.R + blockIdx.x) * seg_cap + pos;
            if (active) {
                if (a.radii) a.radii[idx] = radius_i;
                if (a.tiles) a.tiles[idx] = ref_area;
                n_vis += vis ? 1 : 0;
                const bool write_row = a.compact ? has : (vis || !skip_culled);
                if (write_row) {
                    const int64_t row = a.compact ? slot : idx;
                    if (a.means2d) { a.means2d[2 * idx] = mx; a.means2d[2 * idx + 1] = my; }
                    if (a.depths) a.depths[idx] = depth;
                    if (a.conics) { a.conics[3 * idx] = con0; a.conics[3 * idx + 1] = con1; a.conics[3 * idx + 2] = con2; }
                    float ch[6] = {col[0], col[1], col[2], 0.f, 0.f, 0.f};
                    int n = 3;
                    if (a.flags & GSX_PROJ_RENDER_DEPTH) ch[n++] = depth;
                    if (a.flags & GSX_PROJ_BETAS) ch[n++] = beta;
                    float4 *o = reinterpret_cast<float4 *>(a.rec + row * 12);
                    o[0] = make_float4(mx, my, con0, con1);
                    o[1] = make_float4(con2, opac, vis ? ch[0] : 0.f, vis ? ch[1] : 0.f);
                    o[2] = make_float4(vis ? ch[2] : 0.f, vis ? ch[3] : 0.f, vis ? ch[4] : 0.f, 0.f);
                    if (a.v_rec) {
                        float4 *z = reinterpret_cast<float4 *>(a.v_rec + row * 12);
                        z[0] = z[1] = z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                if (has) {
                    PreRec pr;
                    pr.xs = (uint32_t)r.x0 | ((uint32_t)r.x1 << 16);
                    pr.ys_c = (uint32_t)r.y0 | ((uint32_t)r.y1 << 12) | ((uint32_t)c << 24);
                    pr.depth = __float_as_uint(depth);
                    // candidate mode (compact plans: the sort key's low word is the slot, not this field): the instance's
                    // candidate record, so that the pose backward finds mean and covariance without recomputing them
                    pr.id = use_cand ? (uint32_t)cslot : (uint32_t)idx;
                    a.recs[slot] = pr;
                }
            }
            if (s0 == 0) { GSX_FT(0, 3) }
            if (a.tile_cut)     // near placement: pairs behind their tile's cut-off count in the high half of the word
                walk_rects(r, a.tile_w, 0u, __float_as_uint(depth), [&](int tile, unsigned int, unsigned int dbits) {
                    atomicAdd(&s_cnt[tile], dbits <= s_cutv[tile] ? 1 : 0x10000); }, c * n_tiles);
            else
            walk_rects(r, a.tile_w, 0u, 0u, [&](int tile, unsigned int, unsigned int) { atomicAdd(&s_cnt[tile], 1); },
                       c * n_tiles);
            if (s0 == 0) { GSX_FT(0, 4) }
        }
        if (active && a.vis_count) a.vis_count[g] = n_vis;
        __syncthreads();                                     // s_ninst is read again at the top of the next trip
    }
    __syncthreads();
    GSX_FT(0, 5)
    if (a.row_keys != nullptr) {
        // ---- row keys: exclusive scan of the row's per-tile counts (its segment is grouped by tile), the words out, then the row's
        // instance records are walked a second time - they were written a few microseconds ago by this workgroup - and every key
        // goes to its tile's stretch of the row's segment through the LDS cursors.  Nothing waits for another workgroup.
        GSX_FT(1, 0)
        const int TT = C * n_tiles;
        const int per = (TT + FRONT_THREADS - 1) / FRONT_THREADS;
        const int lo = min(TT, (int)threadIdx.x * per), hi = min(TT, lo + per);
        int sum = 0;
        for (int i = lo; i < hi; ++i) sum += s_cnt[i];
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if ((int)(threadIdx.x & 63) >= o) incl += v;
        }
        if ((threadIdx.x & 63) == 63) s_wcnt[threadIdx.x >> 6] = incl;
        __syncthreads();
        int run = incl - sum, total = 0;
#pragma unroll
        for (int w = 0; w < FRONT_THREADS / 64; ++w) {
            const int ws = s_wcnt[w];
            run += (w < (int)(threadIdx.x >> 6)) ? ws : 0;
            total += ws;
        }
        for (int i = lo; i < hi; ++i) {
            const int cnt = s_cnt[i];
            const int c = i / n_tiles, tl = i - c * n_tiles;
            // (tile-major: a tile's workgroup reads the words of all rows as ONE contiguous stretch - row-major, its column was
            // R cache lines fetched for 4 bytes each)
            a.cnt[((int64_t)c * n_tiles + tl) * a.R + blockIdx.x] =
                (int32_t)(((uint32_t)min(run, (1 << 19) - 1) << 13) | (uint32_t)min(cnt, 8191));
            s_cnt[i] = run;                                 // the tile's write cursor inside the row's segment
            run += cnt;
        }
        if (threadIdx.x == 0) {
            if (total > a.row_cap || total >= (1 << 19)) atomicOr(a.status, 1);     // the row outgrew its segment: the plan grows, redoes
        }
        if (blockIdx.x == 0 && threadIdx.x < GSX_ROW_CURSORS) a.cursor[threadIdx.x] = 0ull;
        __syncthreads();
        GSX_FT(1, 1)
        unsigned long long *seg_keys = a.row_keys + (int64_t)blockIdx.x * a.row_cap;
        for (int c = 0; c < C; ++c) {
            const int n = min(s_ninst[c], seg_cap);
            const PreRec *seg = a.recs + ((int64_t)c * a.R + blockIdx.x) * seg_cap;
            for (int i0 = 0; i0 < n; i0 += FRONT_THREADS) {
                const int i = i0 + (int)threadIdx.x;
                Rect r = {0, 0, 0, 0};
                unsigned int klo = 0u, khi = 0u;
                if (i < n) {
                    const PreRec pr = seg[i];
                    r.x0 = (int)(pr.xs & 0xffffu); r.x1 = (int)(pr.xs >> 16);
                    r.y0 = (int)(pr.ys_c & 0xfffu); r.y1 = (int)((pr.ys_c >> 12) & 0xfffu);
                    klo = a.compact ? (unsigned int)(((int64_t)c * a.R + blockIdx.x) * seg_cap + i) : pr.id;
                    khi = pr.depth;
                }
                place_rects(r, a.tile_w, klo, khi, s_cnt, (int64_t)a.row_cap, seg_keys, c * n_tiles);
            }
        }
        GSX_FT(1, 2)
    } else
    for (int i = threadIdx.x; i < C * n_tiles; i += FRONT_THREADS) {
        const int c = i / n_tiles, tl = i - c * n_tiles;
        a.cnt[((int64_t)c * a.R + blockIdx.x) * n_tiles + tl] = s_cnt[i];
    }
    if (threadIdx.x < C) a.n_inst[threadIdx.x * a.R + blockIdx.x] = s_ninst[threadIdx.x];
}

// workgroup b: row = b / stripes (a chunk of the projection), stripe = b % stripes (tile rows [stripe * rps, +rps) of every
// camera).  counts: per-tile totals (prescanned = 0) or the finished offsets (prescanned = 1, tile_scan_kernel ran).
__global__ __launch_bounds__(FPLACE_THREADS) void front_place_kernel(
    const PreRec *__restrict__ recs, const int32_t *__restrict__ n_inst, int R, int seg_cap, int C, int tile_w,
    int tile_h, int stripes, int rps, int64_t M_cap, const int32_t *__restrict__ counts, int prescanned,
    const int32_t *__restrict__ cnt, int32_t *__restrict__ offsets_out, int64_t *__restrict__ M_dev,
    int32_t *__restrict__ status, unsigned long long *__restrict__ entries, int compact,
    const uint32_t *__restrict__ tile_cut /* nullable: near placement - only keys with depth bits <= tile_cut[tile] are written */) {
    extern __shared__ __attribute__((aligned(16))) int s_cur[];   // [T]: exclusive offsets, then this stripe's write cursors; [T] cut-offs; list
    __shared__ long long s_wsum[FPLACE_THREADS / 64];
    __shared__ int s_nlist;
    const int n_tiles = tile_w * tile_h, T = C * n_tiles;
    const int stripe = blockIdx.x % stripes, row = blockIdx.x / stripes;
    const int t = threadIdx.x;
    const int ys0 = stripe * rps, ys1 = min(tile_h, ys0 + rps);
    const int span = max(0, (ys1 - ys0) * tile_w);
    GSX_FT(1, 0)
    // Everything this workgroup needs from global memory that does not depend on the scan is requested first - the first
    // trip of camera 0's instance records, its row's bases for the stripe, the instance count - so that the kernel is one
    // memory round trip plus the scan deep, not four (a workgroup places only a few hundred entries: latency is all there is).
    // (ALL the records a thread will place for camera 0 - up to FPLACE_PRE trips, which covers a 1024-instance row - not only
    // the first: the walk below was a chain of record load -> LDS add -> store per trip, 9 of the workgroup's 13 us at 500 k)
    const PreRec *seg0 = recs + (int64_t)row * seg_cap;
    PreRec pre[FPLACE_PRE];
#pragma unroll
    for (int k = 0; k < FPLACE_PRE; ++k) {
        pre[k] = PreRec{0u, 0u, 0u, 0u};
        if (t + k * FPLACE_THREADS < seg_cap) pre[k] = seg0[t + k * FPLACE_THREADS];
    }
    const int n0 = n_inst[row];
    int base0 = 0;
    if (t < span) base0 = cnt[(int64_t)row * n_tiles + ys0 * tile_w + t];
    unsigned int *s_cut = reinterpret_cast<unsigned int *>(s_cur + T);
    if (tile_cut)                                           // the cut-offs of this stripe's tiles, every camera
        for (int c = 0; c < C; ++c)
            for (int i = t; i < span; i += FPLACE_THREADS) {
                const int tl = c * n_tiles + ys0 * tile_w + i;
                s_cut[tl] = tile_cut[tl];
            }
    if (prescanned) {
        for (int i = t; i < T; i += FPLACE_THREADS) s_cur[i] = counts[i];
    } else {
        // exclusive scan of the T totals: contiguous chunk per thread, wavefront scan of the chunk sums, 4 wavefront sums.
        // Up to 8 totals per thread (one camera at 640x480: 5) are fetched with independent loads into registers and reused
        // for the second pass - two loops of dependent loads were 5 us of a 20 us kernel.
        const int per = (T + FPLACE_THREADS - 1) / FPLACE_THREADS;
        const int lo = min(T, t * per), hi = min(T, lo + per);
        constexpr int PER_REG = 8;
        int cv[PER_REG];
        long long sum = 0;
        if (per <= PER_REG) {
#pragma unroll
            for (int k = 0; k < PER_REG; ++k) cv[k] = (lo + k < hi) ? counts[lo + k] : 0;
#pragma unroll
            for (int k = 0; k < PER_REG; ++k) sum += max(cv[k], 0);
        } else {
            for (int i = lo; i < hi; ++i) sum += max(counts[i], 0);
        }
        long long incl = sum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long u = __shfl_up(incl, off, 64);
            if ((t & 63) >= off) incl += u;
        }
        if ((t & 63) == 63) s_wsum[t >> 6] = incl;
        __syncthreads();
        long long run = incl - sum, total = 0;
        for (int w = 0; w < FPLACE_THREADS / 64; ++w) {
            const long long ws = s_wsum[w];
            if (w < (t >> 6)) run += ws;
            total += ws;
        }
        const bool pub = blockIdx.x == 0;
        if (per <= PER_REG) {
#pragma unroll
            for (int k = 0; k < PER_REG; ++k) {
                const int i = lo + k;
                if (i < hi) {
                    if (cv[k] < 0 && pub) atomicOr(status, 2);
                    const int32_t o = (int32_t)min(run, (long long)0x7fffffff);
                    s_cur[i] = o;
                    if (pub) offsets_out[i] = o;
                    run += max(cv[k], 0);
                }
            }
        } else {
            for (int i = lo; i < hi; ++i) {
                const int cval = counts[i];
                if (cval < 0 && pub) atomicOr(status, 2);
                const int32_t o = (int32_t)min(run, (long long)0x7fffffff);
                s_cur[i] = o;
                if (pub) offsets_out[i] = o;
                run += max(cval, 0);
            }
        }
        if (pub && t == 0) {
            offsets_out[T] = (int32_t)min(total, (long long)0x7fffffff);
            M_dev[0] = total;
            if (total > M_cap || total > 0x7fffffffLL) atomicOr(status, 1);
        }
    }
    __syncthreads();
    GSX_FT(1, 1)
    if (span <= 0) return;
    for (int c = 0; c < C; ++c) {
        const int32_t *brow = cnt + ((int64_t)c * R + row) * n_tiles;
        for (int i = t; i < span; i += FPLACE_THREADS) {
            const int tl = ys0 * tile_w + i;
            s_cur[c * n_tiles + tl] += (c == 0 && i == t) ? base0 : brow[tl];
        }
    }
    __syncthreads();
    GSX_FT(1, 2)
    // Seven of eight records do not touch this stripe: walked where they were loaded, every trip of the loop ran at the pace of
    // its longest rectangle with an eighth of the lanes at work (9 of the workgroup's 13 us at 500 k, traced).  The records that do
    // touch the stripe are compacted into an LDS list first (any order: the segment is sorted later), then walked one per lane.
    uint4 *s_list = reinterpret_cast<uint4 *>(s_cur + (((tile_cut ? 2 * T : T) + 3) & ~3));      // [FPLACE_LIST], 16-byte aligned
    for (int c = 0; c < C; ++c) {
        const int n = min(max(c == 0 ? n0 : n_inst[c * R + row], 0), seg_cap);
        const PreRec *seg = recs + ((int64_t)c * R + row) * seg_cap;
        auto place_one = [&](const Rect &r, unsigned int klo, unsigned int khi) {
            if (tile_cut) place_rects<true>(r, tile_w, klo, khi, s_cur, M_cap, entries, c * n_tiles, s_cut);
            else place_rects(r, tile_w, klo, khi, s_cur, M_cap, entries, c * n_tiles);
        };
        // rounds of FPLACE_ROUND (= 2) trips = at most FPLACE_LIST records: the list cannot overflow
        auto do_round = [&](int rbase, const PreRec &pa, const PreRec &pb) {
            if (t == 0) s_nlist = 0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < FPLACE_ROUND; ++k) {
                const PreRec &p = k == 0 ? pa : pb;
                const int i = rbase + k * FPLACE_THREADS + t;
                Rect r = {0, 0, 0, 0};
                if (i < n) {
                    r.x0 = (int)(p.xs & 0xffffu); r.x1 = (int)(p.xs >> 16);
                    r.y0 = max((int)(p.ys_c & 0xfffu), ys0); r.y1 = min((int)((p.ys_c >> 12) & 0xfffu), ys1);
                }
                const bool has = (r.y1 > r.y0) && (r.x1 > r.x0);
                // the low key word: flatten id, or the instance's slot (same order - slots are monotone in the flatten id)
                const unsigned int klo = compact ? (unsigned int)(((int64_t)c * R + row) * seg_cap + i) : p.id;
                const unsigned long long m = __ballot(has);
                int wbase = 0;
                if ((t & 63) == 0 && m != 0ull) wbase = atomicAdd(&s_nlist, __popcll(m));
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                const int pos = wbase + __popcll(m & ((1ull << (t & 63)) - 1ull));
                if (has && pos < FPLACE_LIST)
                    s_list[pos] = make_uint4(p.xs, (unsigned int)r.y0 | ((unsigned int)r.y1 << 16), p.depth, klo);
            }
            __syncthreads();
            if (rbase == 0) { GSX_FT(1, 4) }
            const int total = min(s_nlist, FPLACE_LIST);
            const int trips = (total + FPLACE_THREADS - 1) / FPLACE_THREADS;
            for (int tr = 0; tr < trips; ++tr) {        // (whole wavefronts: large rectangles are walked cooperatively)
                const int j = tr * FPLACE_THREADS + t;
                Rect r = {0, 0, 0, 0};
                unsigned int klo = 0u, khi = 0u;
                if (j < total) {
                    const uint4 e = s_list[j];
                    r.x0 = (int)(e.x & 0xffffu); r.x1 = (int)(e.x >> 16);
                    r.y0 = (int)(e.y & 0xffffu); r.y1 = (int)(e.y >> 16);
                    khi = e.z; klo = e.w;
                }
                place_one(r, klo, khi);
            }
            __syncthreads();                            // the list is rewritten by the next round
        };
        int rbase = 0;
        if (c == 0) {                                   // the two rounds whose records were requested at the top
            static_assert(FPLACE_PRE == 4 && FPLACE_ROUND == 2, "prefetched rounds are written out");
            if (n > 0) do_round(0, pre[0], pre[1]);
            if (n > FPLACE_LIST) do_round(FPLACE_LIST, pre[2], pre[3]);
            rbase = 2 * FPLACE_LIST;
        }
        for (; rbase < n; rbase += FPLACE_LIST) {
            const int ia = rbase + t, ib = rbase + FPLACE_THREADS + t;
            const PreRec none = {0u, 0u, 0u, 0u};
            const PreRec pa = (ia < n) ? seg[ia] : none, pb = (ib < n) ? seg[ib] : none;
            do_round(rbase, pa, pb);
        }
    }
    GSX_FT(1, 3)
}

// Pose gradient of a pose-only closure over the VISIBLE instances the front left behind (instead of a pass over all N
// Gaussians that skips the culled two thirds): one thread per instance record recomputes the projection of its Gaussian
// (project_core) and pushes the record-shaped gradient row [v_xy, v_conic, ..., v_depth] through conic -> cov2d -> J, Sc ->
// (R, t) exactly as project_bwd_kernel<POSE_ONLY> does; the 12 entries of d loss / d [R | t] are summed per workgroup and
// left as one partial row per (front row, camera) for the consumer that finishes the pose backward (gsx_track_opt_tail /
// gsx_pose_zhou_bwd_partials).
constexpr int FPB_THREADS = 256;
#ifndef GSX_FPB_SPLIT
#define GSX_FPB_SPLIT 2
#endif
// a (row, camera) segment of instance records is shared by FPB_SPLIT workgroups: with one workgroup per row the launch had one
// wavefront per SIMD and nothing to hide its gathers behind (11.1 us; 2: 9.9, 4: 10.1 and the tail's sum of the partial rows
// +0.4, 8: 10.2 / +0.8)
constexpr int FPB_SPLIT = GSX_FPB_SPLIT;

// TAIL (round 5; gsx_front_pose_bwd_tail, one camera): the closure's tail - the sum of the partial rows, PoseZhou backward, one
// step of the tracking optimiser, the next view matrix (track_tail.h) - is run by the LAST workgroup of this launch to finish
// instead of by a launch of its own (~5 us of dependent-launch latency per closure).  Who is last: a two-level ticket - workgroup
// w arrives at counter w % 32, the last arrival of each counter at the top counter (one counter for all would serve ~500 atomics on
// one address one after the other).  The rows travel by agent-scope stores, each followed by a fence, before the workgroup's ticket;
// the last workgroup reads them with agent-scope loads.  The counters end the launch at zero.
constexpr int FPB_TICKETS = 32;
template <bool TAIL>
__global__ __launch_bounds__(FPB_THREADS) void front_pose_bwd_kernel(
    const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks, int64_t N, int C, int W, int H, float eps2d,
    float near_p, float far_p, int flags, const float *__restrict__ v_rec, const PreRec *__restrict__ recs,
    const int32_t *__restrict__ n_inst, int R, int seg_cap, float *__restrict__ partials /*[R][C][12]*/, int compact,
    const CandRec *__restrict__ cand, const float *__restrict__ cand_hdr, TrackOptState *state, ToTailArgs tail,
    unsigned int *__restrict__ tickets /*[FPB_TICKETS + 1], zero between launches*/) {
    __shared__ float s_part[FPB_THREADS / 64][12];
    // candidate mode of THIS closure (left by its projection): the instance records name candidate records, not flatten ids
    const bool use_cand = cand != nullptr && cand_hdr[CAND_MAX_CAMS * 12 + 4] > 0.5f;
    const int row = blockIdx.x, c = blockIdx.y;
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int n = min(max(n_inst[c * R + row], 0), seg_cap);
    const PreRec *seg = recs + ((int64_t)c * R + row) * seg_cap;
    Cam cam;
    load_cam(viewmats, Ks, c, cam);
    float acc[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = t + (int)blockIdx.z * FPB_THREADS; i < n; i += FPB_THREADS * FPB_SPLIT) {
        const int64_t idx = (int64_t)seg[i].id;
        float mean[3];
        Sym3 S;
        if (use_cand) {
            if (idx < 0 || idx >= (int64_t)R * seg_cap) continue;      // never from a sane front; keeps the gather in range
            const float4 *src = reinterpret_cast<const float4 *>(cand + idx);
            const float4 r0 = src[0], r1 = src[1], r2 = src[2];
            mean[0] = r0.x; mean[1] = r0.y; mean[2] = r0.z;
            S.a00 = r0.w; S.a01 = r1.x; S.a02 = r1.y; S.a11 = r1.z; S.a12 = r1.w; S.a22 = r2.x;
        } else {
            const int64_t g = idx - (int64_t)c * N;
            if (g < 0 || g >= N) continue;                       // never from a sane front; keeps the gathers in range
            mean[0] = means[3 * g]; mean[1] = means[3 * g + 1]; mean[2] = means[3 * g + 2];
            const float q[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
            float s[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
            if (flags & GSX_PROJ_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
            QuatRot qr;
            quat_to_rotmat(q, qr);
            float M[9];
            covar_from_rot_scale(qr.R, s, M, S);
        }
        const int64_t vrow = compact ? (((int64_t)c * R + row) * seg_cap + i) : idx;
        const float4 *r4 = reinterpret_cast<const float4 *>(v_rec + vrow * 12);
        const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2];
        if (flags & GSX_PROJ_RESET_V_REC) {                  // consumed: the row goes back to zero for the next backward
            float4 *w4 = const_cast<float4 *>(r4);
            w4[0] = w4[1] = w4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        Proj p;
        if (!project_core(mean, S, cam, W, H, eps2d, near_p, far_p, p)) continue;
        const float vdepth = (flags & GSX_PROJ_RENDER_DEPTH) ? q2.y : 0.f;          // record column 9
        pose_chain_row(p, mean, S, cam, q0.x, q0.y, q0.z, q0.w, q1.x, vdepth, acc);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float tot = gsx_wave_sum(acc[k]);
        if (lane == 0) s_part[wave][k] = tot;
    }
    __syncthreads();
    if (t < 12) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < FPB_THREADS / 64; ++w) sum += s_part[w][t];
        float *dst = partials + (((int64_t)row * FPB_SPLIT + blockIdx.z) * C + c) * 12 + t;
        if constexpr (TAIL) {
            // write-through store (agent scope) and a wait for its acknowledgement: the row is out before this workgroup's ticket is
            // drawn.  NOT a release fence: that writes back and invalidates the whole L2 of the XCD (buffer_wbl2 / buffer_inv sc1) -
            // once per workgroup, 490 times per launch, under the workgroups still at work (measured: 34 us for this launch)
            __hip_atomic_store(dst, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            *dst = sum;
        }
    }
    if constexpr (TAIL) {
        __shared__ int s_last;
        __shared__ __attribute__((aligned(16))) ToTailLds<FPB_THREADS> s_tail;
        __syncthreads();
        if (t == 0) {
            const unsigned n_wgs = gridDim.x * gridDim.y * gridDim.z;
            const unsigned wg = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
            const unsigned grp = wg % FPB_TICKETS, n_grps = n_wgs < FPB_TICKETS ? n_wgs : (unsigned)FPB_TICKETS;
            const unsigned grp_n = (n_wgs - grp + FPB_TICKETS - 1) / FPB_TICKETS;          // workgroups that arrive at this counter
            int last = 0;
            if (__hip_atomic_fetch_add(&tickets[grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == grp_n - 1) {
                __hip_atomic_store(&tickets[grp], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(&tickets[FPB_TICKETS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_grps - 1) {
                    __hip_atomic_store(&tickets[FPB_TICKETS], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    last = 1;
                }
            }
            s_last = last;
        }
        __syncthreads();
        if (s_last) {                                         // (workgroup-uniform; the rows are read with agent-scope loads)
            to_tail_body<FPB_THREADS, true>(state, tail, s_tail);
        }
    }
}

struct FrontLayout {
    int64_t counts_off, ninst_off, entries_off, scratch_off, rowkeys_off, matrix_off, recs_off, total;
    int64_t cand_hdr_off, cand_n_off, cand_off, cull_off, total_cand;   // candidate area, appended behind `total` (optional)
    int items, R;
};

FrontLayout front_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap) {
    FrontLayout L;
    const int64_t T = C * tile_w * tile_h;
    int items = 1;
#ifndef GSX_FRONT_ROWS
#define GSX_FRONT_ROWS 256            // projection workgroups the front aims for at most (one per CU); A/B: tools/dbg/ab_build.sh
#endif
    while ((N + (int64_t)FRONT_THREADS * items - 1) / ((int64_t)FRONT_THREADS * items) > GSX_FRONT_ROWS && items < 8) items *= 2;
    while ((N + (int64_t)FRONT_THREADS * items - 1) / ((int64_t)FRONT_THREADS * items) > GB_MAX) items *= 2;
    L.items = items;
    L.R = (int)((N + (int64_t)FRONT_THREADS * items - 1) / ((int64_t)FRONT_THREADS * items));
    if (L.R < 1) L.R = 1;
    L.counts_off = 0;
    L.ninst_off = gsx_align256((T + 1) * 4);
    L.entries_off = L.ninst_off + gsx_align256(C * (int64_t)GB_MAX * 4);
    L.scratch_off = L.entries_off + gsx_align256((M_cap > 0 ? M_cap : 1) * 8);
    L.rowkeys_off = L.scratch_off + gsx_align256((M_cap > 0 ? M_cap : 1) * 8);         // [R][M_cap / R] row segments (row keys)
    L.matrix_off = L.rowkeys_off + gsx_align256((M_cap > 0 ? M_cap : 1) * 8);
    L.recs_off = L.matrix_off + gsx_align256(T * (int64_t)GB_MAX * 4);
    L.total = gsx_align256(L.recs_off + C * (int64_t)L.R * FRONT_THREADS * items * 16 + 256);
    L.cand_hdr_off = L.total;
    L.cand_n_off = L.cand_hdr_off + gsx_align256(CAND_HDR * 4);
    L.cand_off = L.cand_n_off + gsx_align256((int64_t)GB_MAX * 4);
    L.cull_off = gsx_align256(L.cand_off + (int64_t)L.R * FRONT_THREADS * items * (int64_t)sizeof(CandRec) + 256);
    L.total_cand = gsx_align256(L.cull_off + (int64_t)L.R * FRONT_THREADS * items * 16 + 256);
    return L;
}

}  // namespace

extern "C" int64_t gsx_front_workspace_bytes(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap) {
    return front_layout(N, C, tile_w, tile_h, M_cap).total;
}

extern "C" int64_t gsx_front_workspace_bytes_cand(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap) {
    return front_layout(N, C, tile_w, tile_h, M_cap).total_cand;
}

extern "C" int gsx_front_candidates(const float *means, const float *quats, const float *scales, const float *viewmats_ref,
                                    const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                                    float far_plane, int flags, const float *logit_opacities, const float *logit_colors,
                                    const float *log_uncertainties, float rot_max, float trans_max, int64_t M_cap,
                                    void *workspace, int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(N >= 1 && C >= 1 && C <= CAND_MAX_CAMS && W > 0 && H > 0 && M_cap >= 1);
    GSX_CHECK_ARG(means && quats && scales && viewmats_ref && Ks && logit_opacities && logit_colors);
    GSX_CHECK_ARG(!(flags & GSX_PROJ_BETAS) || log_uncertainties);
    GSX_CHECK_ARG(rot_max >= 0.f && trans_max >= 0.f && rot_max < 1.0f);
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    if (!workspace || workspace_bytes < L.total_cand) {
        gsx_set_error("gsx_front_candidates: workspace without the candidate area (%lld < %lld: size it with "
                      "gsx_front_workspace_bytes_cand)", (long long)workspace_bytes, (long long)L.total_cand);
        return GSX_E_WORKSPACE;
    }
    char *ws = (char *)workspace;
    FrontArgs a = {};
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats_ref; a.Ks = Ks;
    a.logit_opac = logit_opacities; a.logit_colors = logit_colors; a.log_unc = log_uncertainties;
    a.N = N; a.C = (int)C; a.W = W; a.H = H; a.flags = flags; a.tile_w = tile_w; a.tile_h = tile_h;
    a.items = L.items; a.R = L.R; a.eps2d = eps2d; a.near_p = near_plane; a.far_p = far_plane;
    a.cand_hdr = (float *)(ws + L.cand_hdr_off);
    a.cull4 = (const float4 *)(ws + L.cull_off);
    CandRec *cand = (CandRec *)(ws + L.cand_off);
    int32_t *cand_n = (int32_t *)(ws + L.cand_n_off);
    hipStream_t st = (hipStream_t)stream;
    switch (L.items) {
        case 1: hipLaunchKernelGGL(front_candidates_kernel<1>, dim3((unsigned)L.R), dim3(FRONT_THREADS), 0, st, a, cand, cand_n, rot_max, trans_max); break;
        case 2: hipLaunchKernelGGL(front_candidates_kernel<2>, dim3((unsigned)L.R), dim3(FRONT_THREADS), 0, st, a, cand, cand_n, rot_max, trans_max); break;
        case 4: hipLaunchKernelGGL(front_candidates_kernel<4>, dim3((unsigned)L.R), dim3(FRONT_THREADS), 0, st, a, cand, cand_n, rot_max, trans_max); break;
        case 8: hipLaunchKernelGGL(front_candidates_kernel<8>, dim3((unsigned)L.R), dim3(FRONT_THREADS), 0, st, a, cand, cand_n, rot_max, trans_max); break;
        default: gsx_set_error("gsx_front_candidates: %d Gaussians per thread unsupported", L.items); return GSX_E_UNSUPPORTED;
    }
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

static int front_fwd_impl(const float *means, const float *quats, const float *scales, const float *viewmats,
                          const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                          float far_plane, int flags, const float *logit_opacities, const float *logit_colors,
                          const float *log_uncertainties, int32_t *radii, float *means2d, float *depths, float *conics,
                          int32_t *tiles_per_gauss, float *rec, float *v_rec_clear, int32_t *vis_count, int64_t M_cap,
                          int32_t *offsets, int64_t *M_dev, int32_t *status, int32_t *flatten_ids, int32_t *tile_order,
                          const int32_t *tile_work, int32_t *balanced_order, float chunk_cost, float light_rate,
                          int n_cus, void *workspace, int64_t workspace_bytes, const uint32_t *tile_cut,
                          int32_t *tile_placed, void *stream) {
    // near placement: keys behind their tile's cut-off are counted (offsets, M: as ever) but not written
    GSX_CHECK_ARG(!tile_cut || (tile_placed && !tile_order && (flags & GSX_PROJ_DEFER_SORT)));
    GSX_CHECK_ARG(N >= 1 && C >= 1 && C <= 255 && W > 0 && H > 0 && C * N < ((int64_t)1 << 31));
    GSX_CHECK_ARG(!balanced_order || (tile_work && n_cus >= 1 && n_cus <= gsx_bal::MAX_BINS && chunk_cost >= 0.f &&
                                      chunk_cost < 1e6f && light_rate > 0.f && light_rate <= 1.f));
    const int compact = (flags & GSX_PROJ_COMPACT) ? 1 : 0;
    GSX_CHECK_ARG(means && quats && scales && viewmats && Ks && logit_opacities && logit_colors && rec);
    GSX_CHECK_ARG(compact || (radii && tiles_per_gauss));
    GSX_CHECK_ARG(!(flags & GSX_PROJ_BETAS) || log_uncertainties);
    GSX_CHECK_ARG(offsets && M_dev && status && (flatten_ids || (flags & GSX_PROJ_DEFER_SORT)) && M_cap >= 1 &&
                  M_cap < ((int64_t)1 << 31));
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const int64_t n_tiles = (int64_t)tile_w * tile_h, T = C * n_tiles;
    GSX_CHECK_ARG(tile_w < 65536 && tile_h < 4096);
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    if (!workspace || workspace_bytes < L.total) {
        gsx_set_error("gsx_front_fwd: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)L.total);
        return GSX_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int32_t *counts = (int32_t *)(ws + L.counts_off);
    FrontArgs a;
    a.means = means; a.quats = quats; a.scales = scales; a.viewmats = viewmats; a.Ks = Ks;
    a.logit_opac = logit_opacities; a.logit_colors = logit_colors; a.log_unc = log_uncertainties;
    a.N = N; a.C = (int)C; a.W = W; a.H = H; a.flags = flags; a.tile_w = tile_w; a.tile_h = tile_h;
    a.items = L.items; a.R = L.R; a.eps2d = eps2d; a.near_p = near_plane; a.far_p = far_plane;
    a.radii = radii; a.tiles = tiles_per_gauss; a.vis_count = vis_count; a.means2d = means2d; a.depths = depths;
    a.conics = conics; a.rec = rec; a.v_rec = v_rec_clear;
    a.cnt = (int32_t *)(ws + L.matrix_off); a.n_inst = (int32_t *)(ws + L.ninst_off); a.recs = (PreRec *)(ws + L.recs_off);
    a.compact = compact;
    a.cand = nullptr; a.cand_n = nullptr; a.cand_hdr = nullptr; a.cull4 = nullptr;
    a.tile_cut = tile_cut;
    a.row_keys = nullptr; a.row_cap = 0; a.status = status; a.cursor = (unsigned long long *)M_dev;
    if (flags & GSX_PROJ_ROW_KEYS) {
        // the front ends with the projection launch: keys per row, collected by the consumer's tile workgroups
        GSX_CHECK_ARG(compact && (flags & GSX_PROJ_DEFER_SORT) && !tile_order && !tile_cut && L.R <= 768);
        a.row_keys = (unsigned long long *)(ws + L.rowkeys_off);
        a.row_cap = (int)std::min<int64_t>(M_cap / L.R, (1 << 19) - 1);
        GSX_CHECK_ARG(a.row_cap >= 1);
    }
    GSX_CHECK_ARG(!(flags & GSX_PROJ_MAP_RECORDS) || (flags & GSX_PROJ_CANDIDATES));
    if (flags & GSX_PROJ_CANDIDATES) {
        // the candidate set gsx_front_candidates left in this workspace; a closure whose poses left its margins takes the
        // full path by itself (same results), so the flag is a promise about the workspace, not about the poses
        GSX_CHECK_ARG(compact && C <= CAND_MAX_CAMS);
        if (workspace_bytes < L.total_cand) {
            gsx_set_error("gsx_front_fwd: GSX_PROJ_CANDIDATES needs the candidate area (%lld < %lld)",
                          (long long)workspace_bytes, (long long)L.total_cand);
            return GSX_E_WORKSPACE;
        }
        a.cand = (const CandRec *)(ws + L.cand_off);
        a.cand_n = (const int32_t *)(ws + L.cand_n_off);
        a.cand_hdr = (float *)(ws + L.cand_hdr_off);
        a.cull4 = (const float4 *)(ws + L.cull_off);
    }
    size_t front_lds = (size_t)((T + C + FRONT_THREADS / 64) * 4 + 2 * FRONT_THREADS * L.items);   // histogram + survivor list
    if (tile_cut) front_lds += (size_t)T * 4;                                                      // + the tiles' cut-offs
    a.bal.order = nullptr;
    if (balanced_order) {
        GSX_CHECK_ARG(T <= gsx_bal::MAX_TILES);
        a.bal.work = tile_work; a.bal.order = balanced_order; a.bal.T = (int)T; a.bal.G = n_cus;
        a.bal.chunk_cost = chunk_cost; a.bal.light_rate = light_rate;
        front_lds = front_lds > (size_t)gsx_bal::LDS_BYTES ? front_lds : (size_t)gsx_bal::LDS_BYTES;
    }
    const unsigned front_grid = (unsigned)L.R + (balanced_order ? 1u : 0u);
    if (front_lds > 65536 || L.items > 8) {
        gsx_set_error("gsx_front_fwd: %lld tiles over all cameras / %d Gaussians per thread do not fit the LDS plan",
                      (long long)T, L.items);
        return GSX_E_UNSUPPORTED;
    }
    switch (L.items) {
        case 1: hipLaunchKernelGGL(front_project_kernel<1>, dim3(front_grid), dim3(FRONT_THREADS), front_lds, st, a); break;
        case 2: hipLaunchKernelGGL(front_project_kernel<2>, dim3(front_grid), dim3(FRONT_THREADS), front_lds, st, a); break;
        case 4: hipLaunchKernelGGL(front_project_kernel<4>, dim3(front_grid), dim3(FRONT_THREADS), front_lds, st, a); break;
        case 8: hipLaunchKernelGGL(front_project_kernel<8>, dim3(front_grid), dim3(FRONT_THREADS), front_lds, st, a); break;
        default: gsx_set_error("gsx_front_fwd: %d Gaussians per thread unsupported (N too large for the fused front)", L.items);
                 return GSX_E_UNSUPPORTED;
    }
    GSX_CHECK_LAUNCH();
    if (flags & GSX_PROJ_ROW_KEYS) return GSX_OK;
    // with a launch order wanted the totals are scanned (and bucketed) by the one-workgroup kernel into `offsets` itself;
    // otherwise every placement workgroup scans them on its own and workgroup 0 publishes the offsets
    int32_t *col_out = tile_order ? offsets : counts;
    if (tile_cut)
        hipLaunchKernelGGL(column_scan_kernel<true>, dim3((unsigned)((n_tiles + 63) / 64), (unsigned)C), dim3(64 * CS_GROUPS), 0, st,
                           a.cnt, L.R, (int)n_tiles, col_out, tile_placed);
    else
        hipLaunchKernelGGL(column_scan_kernel<false>, dim3((unsigned)((n_tiles + 63) / 64), (unsigned)C), dim3(64 * CS_GROUPS), 0, st,
                           a.cnt, L.R, (int)n_tiles, col_out, (int32_t *)nullptr);
    GSX_CHECK_LAUNCH();
    if (tile_order) {
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, st, (int)T, M_cap, offsets, M_dev, status, tile_order);
        GSX_CHECK_LAUNCH();
    }
    const int stripes = tile_h >= FRONT_STRIPES ? FRONT_STRIPES : 1;
    const int rps = (tile_h + stripes - 1) / stripes;
    unsigned long long *entries = (unsigned long long *)(ws + L.entries_off);
    unsigned long long *scratch = (unsigned long long *)(ws + L.scratch_off);
    hipLaunchKernelGGL(front_place_kernel, dim3((unsigned)(L.R * stripes)), dim3(FPLACE_THREADS),
                       (size_t)(((T * (tile_cut ? 8 : 4) + 15) & ~(int64_t)15) + FPLACE_LIST * 16), st,
                       a.recs, a.n_inst, L.R, FRONT_THREADS * L.items, (int)C, tile_w, tile_h, stripes, rps, M_cap,
                       col_out, tile_order ? 1 : 0, a.cnt, offsets, M_dev, status, entries, compact, tile_cut);
    GSX_CHECK_LAUNCH();
    if (flags & GSX_PROJ_DEFER_SORT) return GSX_OK;     // the consumer sorts each tile's keys itself (gsx_raster_track_fused_sorting)
    const uint32_t id_max = compact ? (uint32_t)(C * (int64_t)L.R * FRONT_THREADS * L.items - 1) : (uint32_t)(C * N - 1);
    hipLaunchKernelGGL(tile_sort_count_kernel, dim3((unsigned)T), dim3(SORT_THREADS), 0, st, entries, scratch, offsets,
                       (int)n_tiles, bit_length((uint32_t)n_tiles), M_cap, id_max, (int64_t *)nullptr, flatten_ids);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_front_fwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                             const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                             float far_plane, int flags, const float *logit_opacities, const float *logit_colors,
                             const float *log_uncertainties, int32_t *radii, float *means2d, float *depths, float *conics,
                             int32_t *tiles_per_gauss, float *rec, float *v_rec_clear, int32_t *vis_count, int64_t M_cap,
                             int32_t *offsets, int64_t *M_dev, int32_t *status, int32_t *flatten_ids, int32_t *tile_order,
                             const int32_t *tile_work, int32_t *balanced_order, float chunk_cost, float light_rate,
                             int n_cus, void *workspace, int64_t workspace_bytes, void *stream) {
    return front_fwd_impl(means, quats, scales, viewmats, Ks, N, C, W, H, eps2d, near_plane, far_plane, flags, logit_opacities,
                          logit_colors, log_uncertainties, radii, means2d, depths, conics, tiles_per_gauss, rec, v_rec_clear,
                          vis_count, M_cap, offsets, M_dev, status, flatten_ids, tile_order, tile_work, balanced_order,
                          chunk_cost, light_rate, n_cus, workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" int gsx_front_fwd_near(const float *means, const float *quats, const float *scales, const float *viewmats,
                                  const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                                  float far_plane, int flags, const float *logit_opacities, const float *logit_colors,
                                  const float *log_uncertainties, float *rec, float *v_rec_clear, int64_t M_cap,
                                  int32_t *offsets, int64_t *M_dev, int32_t *status, const int32_t *tile_work,
                                  int32_t *balanced_order, float chunk_cost, float light_rate, int n_cus, void *workspace,
                                  int64_t workspace_bytes, const uint32_t *tile_cut, int32_t *tile_placed, void *stream) {
    GSX_CHECK_ARG(tile_cut && tile_placed && (flags & GSX_PROJ_COMPACT) && (flags & GSX_PROJ_DEFER_SORT));
    return front_fwd_impl(means, quats, scales, viewmats, Ks, N, C, W, H, eps2d, near_plane, far_plane, flags, logit_opacities,
                          logit_colors, log_uncertainties, nullptr, nullptr, nullptr, nullptr, nullptr, rec, v_rec_clear,
                          nullptr, M_cap, offsets, M_dev, status, nullptr, nullptr,
                          tile_work, balanced_order, chunk_cost, light_rate, n_cus, workspace, workspace_bytes, tile_cut,
                          tile_placed, stream);
}

extern "C" int gsx_front_keys(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int flags, int64_t *out3) {
    GSX_CHECK_ARG(out3 && N >= 1 && C >= 1 && M_cap >= 1);
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    out3[0] = L.entries_off; out3[1] = L.scratch_off;
    out3[2] = (flags & GSX_PROJ_COMPACT) ? (C * (int64_t)L.R * FRONT_THREADS * L.items - 1) : (C * N - 1);   // largest valid id
    return GSX_OK;
}

// out4 = { byte offset of the row segments of keys ([R][row_cap] x 8 bytes), byte offset of the row words (int32 [C][tiles][R]:
// offset inside the row's segment << 13 | count), row_cap, R } of a front run with GSX_PROJ_ROW_KEYS
extern "C" int gsx_front_rows_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4) {
    GSX_CHECK_ARG(out4 && N >= 1 && C >= 1 && M_cap >= 1);
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    out4[0] = L.rowkeys_off; out4[1] = L.matrix_off; out4[2] = std::min<int64_t>(M_cap / L.R, (1 << 19) - 1); out4[3] = L.R;
    return GSX_OK;
}

extern "C" int64_t gsx_front_rows(int64_t N, int64_t C, int tile_w, int tile_h) {
    return front_layout(N, C, tile_w, tile_h, 1).R * FPB_SPLIT;       // partial rows gsx_front_pose_bwd leaves
}

static int front_pose_bwd_impl(const float *means, const float *quats, const float *scales, const float *viewmats,
                               const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                               float far_plane, int flags, const float *v_rec, int64_t M_cap, const void *workspace,
                               int64_t workspace_bytes, float *partials, void *tail_state, const ToTailArgs *tail_args,
                               unsigned int *tickets, void *stream) {
    GSX_CHECK_ARG(N >= 1 && C >= 1 && C <= 255 && W > 0 && H > 0 && M_cap >= 1);
    GSX_CHECK_ARG(means && quats && scales && viewmats && Ks && v_rec && partials);
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    if (!workspace || workspace_bytes < L.total) {
        gsx_set_error("gsx_front_pose_bwd: not the workspace of gsx_front_fwd (%lld < %lld)", (long long)workspace_bytes,
                      (long long)L.total);
        return GSX_E_WORKSPACE;
    }
    const char *ws = (const char *)workspace;
    const bool cand_on = (flags & GSX_PROJ_CANDIDATES) != 0;
    if (cand_on && workspace_bytes < L.total_cand) {
        gsx_set_error("gsx_front_pose_bwd: GSX_PROJ_CANDIDATES needs the candidate area");
        return GSX_E_WORKSPACE;
    }
    ToTailArgs none = {};
    if (tail_state == nullptr) {
        hipLaunchKernelGGL(front_pose_bwd_kernel<false>, dim3((unsigned)L.R, (unsigned)C, FPB_SPLIT), dim3(FPB_THREADS), 0,
                           (hipStream_t)stream, means, quats, scales, viewmats, Ks, N, (int)C, W, H, eps2d, near_plane, far_plane,
                           flags, v_rec, (const PreRec *)(ws + L.recs_off), (const int32_t *)(ws + L.ninst_off), L.R,
                           FRONT_THREADS * L.items, partials, (flags & GSX_PROJ_COMPACT) ? 1 : 0,
                           cand_on ? (const CandRec *)(ws + L.cand_off) : (const CandRec *)nullptr,
                           cand_on ? (const float *)(ws + L.cand_hdr_off) : (const float *)nullptr, (TrackOptState *)nullptr, none,
                           (unsigned int *)nullptr);
    } else {
        ToTailArgs ta = *tail_args;
        ta.partials = partials;
        ta.n_blocks = L.R * FPB_SPLIT;
        hipLaunchKernelGGL(front_pose_bwd_kernel<true>, dim3((unsigned)L.R, (unsigned)C, FPB_SPLIT), dim3(FPB_THREADS), 0,
                           (hipStream_t)stream, means, quats, scales, viewmats, Ks, N, (int)C, W, H, eps2d, near_plane, far_plane,
                           flags, v_rec, (const PreRec *)(ws + L.recs_off), (const int32_t *)(ws + L.ninst_off), L.R,
                           FRONT_THREADS * L.items, partials, (flags & GSX_PROJ_COMPACT) ? 1 : 0,
                           cand_on ? (const CandRec *)(ws + L.cand_off) : (const CandRec *)nullptr,
                           cand_on ? (const float *)(ws + L.cand_hdr_off) : (const float *)nullptr, (TrackOptState *)tail_state, ta,
                           tickets);
    }
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_front_pose_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                                  const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                                  float far_plane, int flags, const float *v_rec, int64_t M_cap, const void *workspace,
                                  int64_t workspace_bytes, float *partials, void *stream) {
    return front_pose_bwd_impl(means, quats, scales, viewmats, Ks, N, C, W, H, eps2d, near_plane, far_plane, flags, v_rec, M_cap,
                               workspace, workspace_bytes, partials, nullptr, nullptr, nullptr, stream);
}

extern "C" int64_t gsx_front_pose_bwd_tail_words(void) { return FPB_TICKETS + 1; }

// gsx_front_pose_bwd followed by gsx_track_opt_tail (one camera, loss rows given) as ONE launch: see front_pose_bwd_kernel<true>
extern "C" int gsx_front_pose_bwd_tail(const float *means, const float *quats, const float *scales, const float *viewmats,
                                       const float *Ks, int64_t N, int W, int H, float eps2d, float near_plane, float far_plane,
                                       int flags, const float *v_rec, int64_t M_cap, const void *workspace, int64_t workspace_bytes,
                                       float *partials, void *state, const float *Rt, float *dt, float *dR, float *exposure,
                                       float *viewmat, const void *loss_rows, int64_t n_loss_rows, float loss_coef,
                                       uint32_t *tickets, void *stream) {
    GSX_CHECK_ARG(state && Rt && dt && dR && exposure && viewmat && loss_rows && tickets);
    GSX_CHECK_ARG(n_loss_rows >= 1 && n_loss_rows < ((int64_t)1 << 31));
    ToTailArgs ta = {};
    ta.Rt = Rt; ta.dt = dt; ta.dR = dR; ta.exposure = exposure; ta.v_exposure = nullptr; ta.loss = nullptr; ta.viewmat = viewmat;
    ta.loss_rows = (const float *)loss_rows; ta.n_loss_rows = (int)n_loss_rows; ta.loss_coef = loss_coef;
    return front_pose_bwd_impl(means, quats, scales, viewmats, Ks, N, 1, W, H, eps2d, near_plane, far_plane, flags, v_rec, M_cap,
                               workspace, workspace_bytes, partials, state, &ta, tickets, stream);
}

// out4 = { rows R, slots per (camera, row) segment, byte offset of the instance records in the workspace, byte offset of the
// per-segment instance counts }: instance slot s = (c * R + row) * out4[1] + position; record s = 16 bytes
// {x0 | x1 << 16, y0 | y1 << 12 | c << 24, depth bits, flatten id} at workspace + out4[2] + 16 s
// out4 = byte offsets of { candidate header (floats: [16][12] reference poses, rot_max, trans_max, -, -, mode of the last
// closure, closures that fell back), candidate counts int32[R], candidate records (64 bytes each, [R][slots]) } and the
// number of header floats
extern "C" int gsx_front_cand_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4) {
    GSX_CHECK_ARG(out4 && N >= 1 && C >= 1);
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    out4[0] = L.cand_hdr_off; out4[1] = L.cand_n_off; out4[2] = L.cand_off; out4[3] = CAND_HDR;
    return GSX_OK;
}

extern "C" int gsx_front_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4) {
    GSX_CHECK_ARG(out4 && N >= 1 && C >= 1);
    const FrontLayout L = front_layout(N, C, tile_w, tile_h, M_cap);
    out4[0] = L.R; out4[1] = (int64_t)FRONT_THREADS * L.items; out4[2] = L.recs_off; out4[3] = L.ninst_off;
    return GSX_OK;
}
