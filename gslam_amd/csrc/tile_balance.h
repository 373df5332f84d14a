// tile_balance.h - launch order that balances the CUs (device code shared by raster.hip and isect_bin.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- launch order that balances the CUs -------------------------------------------------------------------------------------
// A 640x480 render is 1200 workgroups on 256 CUs: all resident at once, 4 or 5 per CU, so there is no dynamic balancing and
// a kernel lasts as long as its most loaded CU (traced with tools/dbg/wg_trace.sh: CUs finish at 0.73-0.79 of the kernel's
// span on average).  On an idle chip workgroup i lands on the same CU as workgroups i + G, i + 2G, ... (G = number of CUs;
// same trace: true for all 256 CUs, forward and backward launch), so the launch ORDER decides which tiles share a CU.
// This kernel deals the tiles into G bins of near-equal work: tiles sorted by weight (heaviest first); round r hands the
// next G tiles to the bins in the order of their current sums, heaviest tile to the lightest bin.  The bins that get a tile in
// the last, partial round start with a handicap d, so that the others collect heavier tiles before: with v = mean weight of
// the last round's tiles and `light_rate` = throughput of a CU that holds one workgroup less relative to a full one (0.92 in
// the trace: fewer wavefronts per SIMD hide less latency), the final sums S (full bins) and light_rate * S (others) follow
// from the total weight, and d = v - (1 - light_rate) * S.  order[r * G + bin] = tile.  Any permutation is a valid order: the weights (measured by the previous closure) only matter
// for speed.
namespace gsx_bal {
constexpr int THREADS = 1024, MAX_TILES = 2048, MAX_BINS = 1024;
constexpr int LDS_BYTES = MAX_TILES * 8 + MAX_BINS * 8 + 2 * 1024 * 4 + 4 * (THREADS / 64) * 4;

struct Args {
    const int32_t *work;   // [T][2] (chunks, trips) per tile
    int32_t *order;        // [T] out; nullptr = nothing to do
    int T, G;              // tiles, CUs
    float chunk_cost, light_rate;
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One workgroup of THREADS threads; `lds` = LDS_BYTES bytes, 8-byte aligned.  Tiles sorted by weight with a counting sort
// on 1024 weight levels (the order inside a level is arbitrary: any order is a valid one), bins ranked per round by counting
// (G <= 1024 keys, S threads per bin): a dozen barriers in all.
__device__ inline void run(const Args &a, unsigned char *lds) {
    unsigned long long *s_tile = reinterpret_cast<unsigned long long *>(lds);   // (weight bits, tile), heaviest first
    unsigned long long *s_bin = s_tile + MAX_TILES;                             // (sum bits, bin)
    int *s_hist = reinterpret_cast<int *>(s_bin + MAX_BINS), *s_base = s_hist + 1024;
    float *s_red = reinterpret_cast<float *>(s_base + 1024);                    // [3][THREADS / 64]
    int *s_wtot = reinterpret_cast<int *>(s_red + 3 * (THREADS / 64));
    const int t = threadIdx.x, T = a.T, G = a.G;
    constexpr int PER = MAX_TILES / THREADS;
    float wgt[PER];
    float lo = 3.0e38f, hi = 0.f, whole = 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = t + k * THREADS;
        wgt[k] = 0.f;
        if (i < T) {
            const int2 cw = reinterpret_cast<const int2 *>(a.work)[i];
            const float w = 1.0f + fmaxf(0.f, (float)cw.y + a.chunk_cost * (float)cw.x);
            wgt[k] = w;
            lo = fminf(lo, w); hi = fmaxf(hi, w); whole += w;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, off, 64));
        hi = fmaxf(hi, __shfl_xor(hi, off, 64));
    }
    whole = wave_sum(whole);
    if ((t & 63) == 0) { s_red[t >> 6] = lo; s_red[THREADS / 64 + (t >> 6)] = hi; s_red[2 * (THREADS / 64) + (t >> 6)] = whole; }
    s_hist[t] = 0;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) {
        lo = fminf(lo, s_red[i]); hi = fmaxf(hi, s_red[THREADS / 64 + i]); total += s_red[2 * (THREADS / 64) + i];
    }
    const float to_level = hi > lo ? 1023.0f / (hi - lo) : 0.f;
    int lev[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        lev[k] = 1023 - min(1023, max(0, (int)((wgt[k] - lo) * to_level)));    // heaviest -> level 0
        if (t + k * THREADS < T) atomicAdd(&s_hist[lev[k]], 1);
    }
    __syncthreads();
    {   // exclusive scan of the 1024 level counts: wavefront scan + the wavefront totals
        const int c = s_hist[t];
        int inc = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(inc, off, 64);
            if ((t & 63) >= off) inc += up;
        }
        if ((t & 63) == 63) s_wtot[t >> 6] = inc;
        __syncthreads();
        int before = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) before += (w < (t >> 6)) ? s_wtot[w] : 0;
        s_base[t] = before + inc - c;
        s_hist[t] = 0;                                           // becomes the cursor of the level
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = t + k * THREADS;
        if (i < T) {
            const int pos = s_base[lev[k]] + atomicAdd(&s_hist[lev[k]], 1);
            s_tile[pos] = ((unsigned long long)__float_as_uint(wgt[k]) << 32) | (unsigned)i;
        }
    }
    __syncthreads();
    const int rounds = (T + G - 1) / G, last = T - (rounds - 1) * G;
    // handicap of the bins that get a tile in the last round
    float part = 0.f;
    for (int i = T - last + t; i < T; i += THREADS) part += __uint_as_float((unsigned)(s_tile[i] >> 32));
    part = wave_sum(part);
    if ((t & 63) == 0) s_red[t >> 6] = part;                     // (every thread is past its reads of s_red: barriers above)
    __syncthreads();
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < THREADS / 64; ++i) v += s_red[i];
    v /= (float)last;
    const float S_full = total / ((float)last + (float)(G - last) * a.light_rate);
    const float handicap = last < G ? fmaxf(0.f, v - (1.f - a.light_rate) * S_full) : 0.f;
    int Gp = 1;
    while (Gp < G) Gp <<= 1;
    const int S = THREADS / Gp;                                  // threads per bin (power of two, adjacent lanes)
    const int b = t / S, sub = t - b * S;
    float my_sum = b < last ? handicap : 0.f;                    // running sum of bin b (kept by its S threads)
    for (int r = 0; r < rounds; ++r) {
        const int nb = r == rounds - 1 ? last : G;
        if (b < nb && sub == 0) s_bin[b] = ((unsigned long long)__float_as_uint(my_sum) << 32) | (unsigned)b;
        __syncthreads();
        int rank = 0;
        if (b < nb) {
            const unsigned long long mine = s_bin[b];
#pragma unroll 8
            for (int i = sub; i < nb; i += S) rank += s_bin[i] < mine ? 1 : 0;
        }
        for (int off = 1; off < S; off <<= 1) rank += __shfl_xor(rank, off, 64);
        if (b < nb) {                                            // the rank-th lightest bin takes the rank-th heaviest tile left
            const unsigned long long tk = s_tile[r * G + rank];
            if (sub == 0) a.order[r * G + b] = (int)(unsigned)tk;
            my_sum += __uint_as_float((unsigned)(tk >> 32));
        }
        __syncthreads();
    }
}
}  // namespace gsx_bal
