// tile_sort_lds.h - depth sort of ONE tile's intersection keys by the 256-thread workgroup that rasterises the tile
// (device code for raster.hip; the stand-alone launch of the same sort is tile_sort_count_kernel in isect_bin.hip).
//
// Why it lives inside the rasteriser (round 4).  In a pose-only closure only the front of a tile's list is
// ever composited - ~240 of ~1100 entries at 500 k Gaussians before every pixel of the tile has saturated - yet the sort
// launch between the placement and the rasteriser sorted all of them and cost 15 us of a chain on which every launch is
// latency-bound.  Here the tile's workgroup sorts only the keys up to a depth CUT-OFF - the depth of the deepest entry the
// previous closure of the same tile composited, with a margin - in LDS, right before it composites them; the keys behind the
// cut-off stay as the placement left them.  Depth order is a total order on (depth bits, id) keys, and "depth <= cut" is a
// prefix of it: the sorted near keys ARE the first n_near entries of the fully sorted list, bit for bit.  If a pixel of the
// tile is still live when the near list ends (the cut-off was too tight: the pose moved, a surface left the tile), the
// workgroup sorts the next slab of depths behind it the same way and continues, slab after slab - each slab is the next stretch
// of the list the stand-alone sort would have produced.  Nothing is ever sorted through memory on the way: a tile that needed
// a second slab costs one more LDS sort, not a merge sort of its whole segment (the first version did that: the 3 % of the
// tiles whose cut-off had been too tight became the critical path of the whole launch, 104 us against 71).
//
// Keys: float_bits(depth) << 32 | id with depth > 0, so unsigned order of the high word = depth order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsx_tsort {
constexpr int THREADS = 256;
constexpr int CAPK = 1152;                 // keys the LDS counting sort takes (18 runs of 64)
constexpr int NB = 512;                    // depth buckets of the counting sort
constexpr int MAX_BUCKET = 96;             // largest bucket the quadratic in-bucket ranking accepts
constexpr int MERGE_CAP = 1024;            // LDS window of the generic path (chunks sorted in LDS, merged through memory)
constexpr int POOL_BYTES = 2 * CAPK * 8 + NB * 4;    // 20480: two key buffers + the bucket cursors
constexpr int CTL_WORDS = 16;              // small control block in LDS
constexpr uint32_t CUT_NONE = 0x7f800000u; // +inf: no cut-off, sort everything

typedef unsigned long long u64;

// (The workgroup reads back through the vector L1 what it has just written to memory - ids, sorted keys, appended keys.  All
// wavefronts of a workgroup share the CU's L1, which keeps its own stores and loads in order: the workgroup barrier is enough.  An
// agent-scope acquire here - tried in round 5 while chasing what turned out to be a vote taken under `if (lane == 0)` - invalidates
// the L1 and the XCD's L2 for every tile resident on the CU: the fused launch went from 80 to 110 us.)

// one rank-merge level: runs of length `run` in src[0..n) -> runs of 2 * run in dst (stable; LDS or global pointers)
template <typename Ptr>
__device__ __forceinline__ void merge_level(Ptr src, Ptr dst, int n, int run) {
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const u64 k = src[i];
        const int r = i / run;
        const int own = r * run, pair = (r & ~1) * run;
        const int pb = (r ^ 1) * run;                          // partner run
        const int plen = max(0, min(run, n - pb));
        const bool right = (r & 1) != 0;
        int lo = 0, hi = plen;                                  // left run: #partner < k ; right run: #partner <= k
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const u64 v = src[pb + mid];
            const bool before = right ? (v <= k) : (v < k);
            if (before) lo = mid + 1; else hi = mid;
        }
        dst[pair + (i - own) + lo] = k;
    }
}

// rank-merge sort of n <= cap keys already in bufA (padding to a multiple of 64 written here); returns the buffer that holds
// the sorted keys.  Runs of 64 by counting ranks, then log2(n / 64) merge levels (one barrier each).
__device__ __forceinline__ u64 *lds_rank_sort(u64 *bufA, u64 *bufB, int n) {
    const u64 INF = ~0ull >> 1;
    const int n_pad = (n + 63) / 64 * 64;
    for (int i = n + threadIdx.x; i < n_pad; i += THREADS) bufA[i] = INF;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += THREADS) {
        const u64 k = bufA[i];
        const int cb = i & ~63;
        unsigned int rank = 0;
#pragma unroll 8
        for (int m = 0; m < 64; ++m) rank += (unsigned int)((bufA[cb + m] - k) >> 63);
        bufB[cb + rank] = k;
    }
    __syncthreads();
    u64 *src = bufB, *dst = bufA;
    for (int run = 64; run < n; run <<= 1) {
        merge_level(src, dst, n, run);
        __syncthreads();
        u64 *t = src; src = dst; dst = t;
    }
    return src;
}

// exclusive scan of s_cur[0..NB) in place; returns true if some bucket holds more than `limit` keys.  s_w: 8 ints of LDS.
__device__ __forceinline__ bool bucket_scan(int *s_cur, int *s_w, int limit) {
    constexpr int PER = NB / THREADS;
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
    if ((t & 63) == 63) s_w[t >> 6] = incl;
    const int any_big = __syncthreads_or(big > limit);
    int run = incl - v;
    for (int w = 0; w < (t >> 6); ++w) run += s_w[w];
#pragma unroll
    for (int j = 0; j < PER; ++j) { s_cur[PER * t + j] = run; run += cnt[j]; }
    __syncthreads();
    return any_big != 0;
}

// Sorts the keys of keys[0..n) whose depth bits lie in (lo_bits, hi_bits] in LDS - the next SLAB of the tile's list behind the
// entries sorted so far - and writes their ids to flat[0..m) and the sorted keys to sorted[0..m) (the caller passes the
// pointers advanced to where the slab goes).  If more than CAPK keys fall into the window, the window is cut down to its
// nearest buckets of a monotone depth histogram that hold at most CAPK keys (two more passes over the segment, which sits
// in L2): the slab is then (lo_bits, hi'] with hi' = the largest depth it took - returned in s_ctl[3], the lower bound of
// the NEXT slab.  Returns m >= 0, or -1 if even one bucket of the histogram holds more than CAPK keys (a pile of equal
// depths: nothing written, take sort_all).  Every thread of the workgroup must call it; `pool` = POOL_BYTES of LDS nobody
// else uses meanwhile, s_ctl = CTL_WORDS ints.
// first stage of sort_window: the pool is taken over, the control block and the bucket counters start from zero
__device__ __forceinline__ void sort_window_begin(uint32_t hi_bits, void *pool, int *s_ctl) {
    u64 *s_a = reinterpret_cast<u64 *>(pool), *s_b = s_a + CAPK;
    int *s_cur = reinterpret_cast<int *>(s_b + CAPK);
    const int t = threadIdx.x;
    __syncthreads();                                           // the pool may still be in use as something else
    if (t == 0) { s_ctl[0] = 0; s_ctl[1] = 0x7fffffff; s_ctl[2] = 0; s_ctl[3] = (int)hi_bits; }
    for (int i = t; i < NB; i += THREADS) s_cur[i] = 0;
    __syncthreads();
}

// a key of the window into the LDS list (any order), the range of the window's depths in the caller's registers
struct WindowAcc {
    unsigned int dmin = 0x7fffffffu, dmax = 0u;
    __device__ __forceinline__ void add(u64 key, unsigned int d, void *pool, int *s_ctl) {
        const int p = atomicAdd(&s_ctl[0], 1);
        if (p < CAPK) reinterpret_cast<u64 *>(pool)[p] = key;
        dmin = min(dmin, d); dmax = max(dmax, d);
    }
    __device__ __forceinline__ void finish(int *s_ctl) {     // every thread of the workgroup
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            dmin = min(dmin, (unsigned int)__shfl_xor((int)dmin, off, 64));
            dmax = max(dmax, (unsigned int)__shfl_xor((int)dmax, off, 64));
        }
        if ((threadIdx.x & 63) == 0) { atomicMin(&s_ctl[1], (int)dmin); atomicMax(&s_ctl[2], (int)dmax); }
    }
};

__device__ __forceinline__ int sort_window_rest(const u64 *__restrict__ keys, u64 *__restrict__ sorted, int32_t *__restrict__ flat,
                                                int n, uint32_t lo_bits, uint32_t hi_bits, uint32_t id_max, void *pool,
                                                int *s_ctl);

__device__ __forceinline__ int sort_window(const u64 *__restrict__ keys, u64 *__restrict__ sorted, int32_t *__restrict__ flat,
                                           int n, uint32_t lo_bits, uint32_t hi_bits, uint32_t id_max, void *pool,
                                           int *s_ctl) {
    const int t = threadIdx.x;
    sort_window_begin(hi_bits, pool, s_ctl);
    // a. the keys of the window, compacted into LDS in any order; range of their depths
    WindowAcc acc;
    for (int i0 = 0; i0 < n; i0 += 4 * THREADS) {              // four loads in flight per thread
        u64 kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * THREADS + t;
            kk[u] = (i < n) ? keys[i] : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int d = (unsigned int)(kk[u] >> 32);
            if (d > lo_bits && d <= hi_bits) acc.add(kk[u], d, pool, s_ctl);   // (the padding's high word is 0xffffffff: never passes)
        }
    }
    acc.finish(s_ctl);
    return sort_window_rest(keys, sorted, flat, n, lo_bits, hi_bits, id_max, pool, s_ctl);
}

// Everything behind the first filter pass of a window (the window's keys - at most CAPK of them stored - sit in the LDS list, their
// count in s_ctl[0], their depth range in s_ctl[1..2]); `keys[0..n)` is scanned again only if the window holds more than CAPK keys.
__device__ __forceinline__ int sort_window_rest(const u64 *__restrict__ keys, u64 *__restrict__ sorted, int32_t *__restrict__ flat,
                                                int n, uint32_t lo_bits, uint32_t hi_bits, uint32_t id_max, void *pool,
                                                int *s_ctl) {
    u64 *s_a = reinterpret_cast<u64 *>(pool), *s_b = s_a + CAPK;
    int *s_cur = reinterpret_cast<int *>(s_b + CAPK);
    const int t = threadIdx.x;
    __syncthreads();
    int m = s_ctl[0];
    if (m == 0) return 0;
    float fmin_ = __uint_as_float((unsigned int)s_ctl[1]), fmax_ = __uint_as_float((unsigned int)s_ctl[2]);
    float range = fmax_ - fmin_;
    float scale = (range > 0.0f) ? (float)(NB - 1) / range : 0.0f;
    auto bucket_of_d = [&](unsigned int dbits) -> int {
        const int b = (int)((__uint_as_float(dbits) - fmin_) * scale);
        return min(max(b, 0), NB - 1);
    };
    if (m > CAPK) {
        // a'. too many: histogram of the window's keys straight from memory, the nearest buckets that fit, and their largest depth
        for (int i0 = 0; i0 < n; i0 += 4 * THREADS) {
            u64 kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS + t;
                kk[u] = (i < n) ? keys[i] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int d = (unsigned int)(kk[u] >> 32);
                if (d > lo_bits && d <= hi_bits) atomicAdd(&s_cur[bucket_of_d(d)], 1);
            }
        }
        __syncthreads();
        bucket_scan(s_cur, s_ctl + 4, CAPK);                   // exclusive: s_cur[b] = keys in the buckets before b
        if (t == 0) { s_ctl[8] = -1; }
        __syncthreads();
        // the last bucket b with (keys before b) + (keys in b) <= CAPK: exclusive prefix of b + 1 (or m for the last bucket)
        for (int b = t; b < NB; b += THREADS) {
            const int upto = (b + 1 < NB) ? s_cur[b + 1] : m;
            if (upto <= CAPK) atomicMax(&s_ctl[8], b);
        }
        __syncthreads();
        const int bstar = s_ctl[8];
        if (bstar < 0) return -1;                              // the nearest bucket alone does not fit: equal depths piled up
        __syncthreads();
        if (t == 0) { s_ctl[0] = 0; s_ctl[9] = 0; }
        for (int i = t; i < NB; i += THREADS) s_cur[i] = 0;
        __syncthreads();
        // the slab = the window's keys in buckets <= bstar; its largest depth is the exact threshold that describes it
        unsigned int hmax = 0u;
        for (int i0 = 0; i0 < n; i0 += 4 * THREADS) {
            u64 kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * THREADS + t;
                kk[u] = (i < n) ? keys[i] : ~0ull;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned int d = (unsigned int)(kk[u] >> 32);
                if (d > lo_bits && d <= hi_bits && bucket_of_d(d) <= bstar) {
                    const int p = atomicAdd(&s_ctl[0], 1);
                    if (p < CAPK) s_a[p] = kk[u];
                    hmax = max(hmax, d);
                }
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) hmax = max(hmax, (unsigned int)__shfl_xor((int)hmax, off, 64));
        if ((t & 63) == 0) atomicMax(&s_ctl[9], (int)hmax);
        __syncthreads();
        m = s_ctl[0];                                          // <= CAPK by construction
        if (m == 0) return -1;
        if (t == 0) { s_ctl[2] = s_ctl[9]; s_ctl[3] = s_ctl[9]; }
        __syncthreads();
        fmax_ = __uint_as_float((unsigned int)s_ctl[2]);
        range = fmax_ - fmin_;
        scale = (range > 0.0f) ? (float)(NB - 1) / range : 0.0f;
    }
    auto bucket_of = [&](u64 k) -> int { return bucket_of_d((unsigned int)(k >> 32)); };
    // b. histogram, scan, scatter by bucket, exact rank inside the bucket
    for (int i = t; i < m; i += THREADS) atomicAdd(&s_cur[bucket_of(s_a[i])], 1);
    __syncthreads();
    const bool degenerate = bucket_scan(s_cur, s_ctl + 4, MAX_BUCKET);
    const u64 *res;
    if (degenerate) {
        res = lds_rank_sort(s_a, s_b, m);
    } else {
        for (int i = t; i < m; i += THREADS) {
            const u64 k = s_a[i];
            s_b[atomicAdd(&s_cur[bucket_of(k)], 1)] = k;
        }
        __syncthreads();
        for (int i = t; i < m; i += THREADS) {
            const u64 k = s_b[i];
            const int b = bucket_of(k);
            const int bs = b ? s_cur[b - 1] : 0, be = s_cur[b];   // after the scatter a bucket's cursor is its end
            int rank = 0;
            for (int j = bs; j < be; ++j) rank += (s_b[j] < k) ? 1 : 0;
            s_a[bs + rank] = k;
        }
        __syncthreads();
        res = s_a;
    }
    // c. out: ids for the rasteriser, the sorted keys for the next cut-off
    for (int i = t; i < m; i += THREADS) {
        const u64 k = res[i];
        flat[i] = (int32_t)min((uint32_t)k, id_max);            // never hand an out-of-range gather index on
        sorted[i] = k;
    }
    __syncthreads();
    return m;
}

// Row keys (gsx_front_fwd_rows, round 5): the front does not build tile segments at all - no count matrix scan, no placement launch.
// Every projection workgroup leaves the keys of ITS row of Gaussians in the row's own segment, grouped by tile, and one word per
// (camera, tile, row): offset inside the row's segment << ROW_SHIFT | count.  The tile's workgroup collects its keys itself:
//   * reads its R words (one per thread, contiguous), scans the counts: the tile's total n and every row's place in the tile;
//   * reserves [base, base + n) of the contiguous key buffer with ONE atomic on one of ROW_CURSORS cursors, each over its own
//     1 / ROW_CURSORS of the buffer (segments are handed out in the order the tiles ask - nobody needs tile-major offsets in a
//     pose-only closure; the cursors add up to the render's M.  ONE cursor was 1200 same-address atomics per launch, served one
//     after the other: the tiles waited up to 20 us for their turn);
//   * feeds the keys of the first window (lo_bits, hi_bits] straight from the rows' stretches into the LDS list of sort_window_rest,
//     which the caller runs next; the contiguous copy of ALL the tile's keys is written only if something scans them again
//     (copy_tile_keys: a further slab, a window of more than CAPK keys, the through-memory sort).
// -> n (0 if the buffer is full: status bit 1, the overflow protocol of the launch plans), base in `base_out`.
// Every thread of the workgroup must call it.  R <= 3 * THREADS rows.
constexpr int ROW_SHIFT = 13;                  // count in the low 13 bits of a row word (a row holds at most 8192 instances)
constexpr int ROW_CURSORS = 64;                // key counters of a render (tile t draws on counter t % 64)
constexpr int ROW_RPT = 3;                     // rows per thread of the collecting workgroup (R <= 3 * THREADS)

// the thread's rows of the tile: counts and offsets inside the rows' segments, the thread's place in the tile (exclusive scan of the
// counts over the workgroup, s_ctl[12..15]) and the tile's total.  Two barriers.
struct RowStretches {
    int cnt[ROW_RPT], off[ROW_RPT], dst, total;
};
__device__ __forceinline__ RowStretches scan_row_words(const uint32_t (&wd)[ROW_RPT], int row_cap, int *s_ctl) {
    const int t = threadIdx.x;
    RowStretches rs;
    int sum = 0;
#pragma unroll
    for (int u = 0; u < ROW_RPT; ++u) {
        rs.cnt[u] = (int)(wd[u] & ((1u << ROW_SHIFT) - 1u));
        rs.off[u] = (int)(wd[u] >> ROW_SHIFT);
        if (rs.off[u] + rs.cnt[u] > row_cap) rs.cnt[u] = max(0, row_cap - rs.off[u]);   // (a row that overflowed its segment: flagged by the front)
        sum += rs.cnt[u];
    }
    int incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if ((t & 63) >= o) incl += v;
    }
    __syncthreads();                                           // (s_ctl[12..15] may still be read from an earlier scan)
    if ((t & 63) == 63) s_ctl[12 + (t >> 6)] = incl;
    __syncthreads();
    rs.dst = incl - sum;
    rs.total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) {
        const int ws = s_ctl[12 + w];
        rs.dst += (w < (t >> 6)) ? ws : 0;
        rs.total += ws;
    }
    return rs;
}

__device__ __forceinline__ int gather_tile_keys(const uint32_t *__restrict__ row_words, const u64 *__restrict__ row_keys, int R,
                                                int row_cap, int64_t col /* index of the tile's first word: (camera * tiles + tile-in-camera) * R */,
                                                int tile, unsigned long long *__restrict__ cursors, int64_t M_cap,
                                                int32_t *__restrict__ status, uint32_t lo_bits, uint32_t hi_bits, void *pool,
                                                int *s_ctl, int &base_out) {
    const int t = threadIdx.x;
    // (the tile's words are requested before the pool is taken over: nothing below depends on the two barriers of the hand-over)
    uint32_t wd[ROW_RPT];
#pragma unroll
    for (int u = 0; u < ROW_RPT; ++u) {
        const int r = t + u * THREADS;
        wd[u] = (r < R) ? row_words[col + r] : 0u;
    }
    sort_window_begin(hi_bits, pool, s_ctl);
    // the first four keys of the thread's first row: on their way while the counts are scanned
    const int c0 = (int)(wd[0] & ((1u << ROW_SHIFT) - 1u)), o0 = (int)(wd[0] >> ROW_SHIFT);
    const u64 *src0 = row_keys + (int64_t)t * row_cap + o0;
    u64 k0[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k0[j] = (j < c0 && o0 + j < row_cap) ? src0[j] : ~0ull;
    const RowStretches rs = scan_row_words(wd, row_cap, s_ctl);
    // the tile's segment: reserved now, needed only when the sorted slab is written out - the atomic's round trip (device scope:
    // ~2 us) runs beside the gather below
    long long base = -1;
    if (t == 0 && rs.total > 0) {
        const long long sub = (long long)M_cap / ROW_CURSORS;
        const int j = tile % ROW_CURSORS;
        const long long local = (long long)atomicAdd(cursors + j, (unsigned long long)rs.total);
        base = (long long)j * sub + local;
        if (local + rs.total > sub) base = -2;
    }
    WindowAcc acc;
#pragma unroll
    for (int u = 0; u < ROW_RPT; ++u) {
        const int r = t + u * THREADS;
        const u64 *src = row_keys + (int64_t)r * row_cap + rs.off[u];
        for (int i0 = 0; i0 < rs.cnt[u]; i0 += 4) {            // four loads in flight
            u64 kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kk[j] = (u == 0 && i0 == 0) ? k0[j] : ((i0 + j < rs.cnt[u]) ? src[i0 + j] : ~0ull);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const unsigned int d = (unsigned int)(kk[j] >> 32);
                if (i0 + j < rs.cnt[u] && d > lo_bits && d <= hi_bits) acc.add(kk[j], d, pool, s_ctl);
            }
        }
    }
    acc.finish(s_ctl);
    if (t == 0) {
        if (base == -2) atomicOr(status, 1);                   // this counter's share of the key buffer is full
        s_ctl[11] = (int)max(base, -1LL);
    }
    __syncthreads();
    const int b = s_ctl[11];
    base_out = max(b, 0);
    return b < 0 ? 0 : rs.total;
}

// the tile's own contiguous copy of its keys, out[0..n): what later slabs, windows of more than CAPK keys and the through-memory sort
// scan.  Written on demand only - most tiles composite ONE slab that fits the LDS list and never read their keys a second time.
__device__ __attribute__((noinline)) void copy_tile_keys(const uint32_t *__restrict__ row_words, const u64 *__restrict__ row_keys, int R,
                                               int row_cap, int64_t col, u64 *__restrict__ out, int *s_ctl) {
    const int t = threadIdx.x;
    uint32_t wd[ROW_RPT];
#pragma unroll
    for (int u = 0; u < ROW_RPT; ++u) {
        const int r = t + u * THREADS;
        wd[u] = (r < R) ? row_words[col + r] : 0u;
    }
    const RowStretches rs = scan_row_words(wd, row_cap, s_ctl);
    int dst = rs.dst;
#pragma unroll
    for (int u = 0; u < ROW_RPT; ++u) {
        const int r = t + u * THREADS;
        const u64 *src = row_keys + (int64_t)r * row_cap + rs.off[u];
        for (int i0 = 0; i0 < rs.cnt[u]; i0 += 4) {
            u64 kk[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) kk[j] = (i0 + j < rs.cnt[u]) ? src[i0 + j] : ~0ull;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (i0 + j < rs.cnt[u]) out[dst + i0 + j] = kk[j];
        }
        dst += rs.cnt[u];
    }
    __syncthreads();                                           // the workgroup reads the copy next
}

// Near placement (gsx_front_fwd_near): the front placed only the keys in front of the tile's depth cut-off; the tile's segment keeps
// room for the others.  When a pixel of the tile outlives the placed keys, the tile's workgroup appends the keys BEHIND the cut-off
// itself: it walks the instance records of camera c the front left behind ([R][seg_cap] 16-byte records {x0 | x1 << 16,
// y0 | y1 << 12 | c << 24, depth bits, flatten id}, n_inst[c * R + row] valid ones per row) and writes the key of every instance
// whose tile rectangle holds (tx, ty) and whose depth bits are > cut_bits - the complement of the placement's test, on the same
// words - to out[0..cap) in any order (the slab sort behind it fixes the order).  The rare, slow path: one workgroup reads every
// record of the camera (170 k x 16 B at 500 k Gaussians), 64-record stretches of the rows, eight in flight per wavefront; the
// stretches that hold records are found by a ballot over a row-length table in LDS.  Returns the number of keys appended
// (workgroup-uniform; more than cap = inconsistent counts, clamped by the caller).  Every thread must call it.
__device__ __forceinline__ int complete_tile(const uint4 *__restrict__ inst, const int32_t *__restrict__ n_inst, int R,
                                             int seg_cap, int c, int tx, int ty, uint32_t cut_bits, int compact,
                                             u64 *__restrict__ out, int cap, void *pool, int *s_ctl) {
    int *s_n = reinterpret_cast<int *>(pool);                  // [R] valid records per row
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __syncthreads();                                           // the pool may still be in use as something else
    if (t == 0) s_ctl[10] = 0;
    for (int i = t; i < R; i += THREADS) s_n[i] = min(max(n_inst[c * R + i], 0), seg_cap);
    __syncthreads();
    const int sh = 31 - __clz(max(seg_cap >> 6, 1));           // seg_cap = 1024 * 2^k: 64-record stretches per row = 2^sh
    const int n_str = R << sh;
    const uint4 *base = inst + (int64_t)c * R * seg_cap;
    const int64_t slot0 = (int64_t)c * R * seg_cap;
    for (int k0 = wave * 64; k0 < n_str; k0 += THREADS) {
        const int k = k0 + lane;
        const int nr = (k < n_str) ? s_n[k >> sh] : 0;
        unsigned long long mask = __ballot(((k & ((1 << sh) - 1)) << 6) < nr);
        while (mask != 0ull) {
            uint4 rr[8];
            int idx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                rr[u] = make_uint4(0u, 0u, 0u, 0u);
                idx[u] = -1;
                if (mask != 0ull) {
                    const int l = __ffsll((long long)mask) - 1;
                    mask &= mask - 1ull;
                    const int kk = k0 + l;
                    const int row = kk >> sh, i = ((kk & ((1 << sh) - 1)) << 6) + lane;
                    const int n_row = __builtin_amdgcn_readlane(nr, l);
                    if (i < n_row) {
                        idx[u] = row * seg_cap + i;
                        rr[u] = base[idx[u]];
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int x0 = (int)(rr[u].x & 0xffffu), x1 = (int)(rr[u].x >> 16);
                const int y0 = (int)(rr[u].y & 0xfffu), y1 = (int)((rr[u].y >> 12) & 0xfffu);
                if (idx[u] >= 0 && tx >= x0 && tx < x1 && ty >= y0 && ty < y1 && rr[u].z > cut_bits) {
                    const int p = atomicAdd(&s_ctl[10], 1);
                    const uint32_t klo = compact ? (uint32_t)(slot0 + idx[u]) : rr[u].w;
                    if (p < cap) out[p] = ((u64)rr[u].z << 32) | klo;
                }
            }
        }
    }
    __syncthreads();
    return s_ctl[10];
}

// Sorts ALL n keys of the segment, any n: chunks of MERGE_CAP keys rank-sorted in LDS and written back in place, the remaining
// merge levels through memory (L2), ping-ponging between the segment and `sorted`.  Ids to flat[0..n), keys to sorted[0..n).
// The slow, rare path (a near list that does not fit LDS, or a tile whose cut-off failed with more than CAPK keys).
__device__ __forceinline__ void sort_all(u64 *__restrict__ keys, u64 *__restrict__ sorted, int32_t *__restrict__ flat, int n,
                                         uint32_t id_max, void *pool) {
    u64 *s_a = reinterpret_cast<u64 *>(pool), *s_b = s_a + CAPK;
    const int t = threadIdx.x;
    __syncthreads();
    for (int cb = 0; cb < n; cb += MERGE_CAP) {
        const int len = min(MERGE_CAP, n - cb);
        for (int i = t; i < len; i += THREADS) s_a[i] = keys[cb + i];
        __syncthreads();
        const u64 *res = lds_rank_sort(s_a, s_b, len);
        for (int i = t; i < len; i += THREADS) keys[cb + i] = res[i];
        __syncthreads();
    }
    u64 *src = keys, *dst = sorted;
    for (int run = MERGE_CAP; run < n; run <<= 1) {
        merge_level(src, dst, n, run);
        __syncthreads();                                       // (global writes of the workgroup are visible to it behind the barrier)
        u64 *x = src; src = dst; dst = x;
    }
    for (int i = t; i < n; i += THREADS) {
        const u64 k = src[i];
        flat[i] = (int32_t)min((uint32_t)k, id_max);
        if (src != sorted) sorted[i] = k;
    }
    __syncthreads();
}
}  // namespace gsx_tsort
