// isect.hip — K3..K7: tile intersection, depth sort and per-tile offsets.  Replaces gsplat `isect_tiles` and
// `isect_offset_encode` (gslam/rasterization.py:259-274).  Integer-exact contract, SURVEY.md §9.2:
//   key = cam << (32 + tile_n_bits) | tile << 32 | float_bits(depth),  value = flatten id (c*N+g),
//   stable order (ties keep emission order = ascending flatten id, tiles row-major inside the bbox).
//
// This file holds the gsplat-shaped pieces that are NOT the sort: the per-Gaussian tile count (K3), the int64 inclusive
// scan of those counts and the unsorted emission of (key, flatten id) pairs in gsplat's order (isect_tiles(sort=False)),
// and the offset encode of sorted keys (K7).  The sort itself is the tile-binned one of isect_bin.hip; the library links
// no sort / scan library (the device-wide radix sort the first version borrowed from rocPRIM is gone - the parity tests
// use torch.sort on the unsorted emission as their independent yardstick).
#include "gsx_common.h"

#include <cstring>

namespace {

__device__ __forceinline__ uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

struct Rect {
    uint32_t x0, y0, x1, y1;
};

__device__ __forceinline__ Rect tile_rect(float mx, float my, int32_t radius, int tile_w, int tile_h) {
    const float ts = (float)GSX_TILE;
    const float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    Rect r;
    r.x0 = min(sat_u32(floorf(tx - tr)), (uint32_t)tile_w);
    r.y0 = min(sat_u32(floorf(ty - tr)), (uint32_t)tile_h);
    r.x1 = min(sat_u32(ceilf(tx + tr)), (uint32_t)tile_w);
    r.y1 = min(sat_u32(ceilf(ty + tr)), (uint32_t)tile_h);
    return r;
}

__global__ void isect_count_kernel(const float *__restrict__ means2d, const int32_t *__restrict__ radii, int64_t CN,
                                   int tile_w, int tile_h, int32_t *__restrict__ tiles_per_gauss) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CN) return;
    const int32_t r = radii[i];
    int32_t n = 0;
    if (r > 0) {
        const Rect t = tile_rect(means2d[2 * i], means2d[2 * i + 1], r, tile_w, tile_h);
        n = (int32_t)((t.y1 - t.y0) * (t.x1 - t.x0));
    }
    tiles_per_gauss[i] = n;
}

__global__ void isect_emit_kernel(const float *__restrict__ means2d, const int32_t *__restrict__ radii,
                                  const float *__restrict__ depths, const int64_t *__restrict__ cum_tiles, int64_t N,
                                  int64_t CN, int tile_w, int tile_h, int tile_n_bits, int64_t *__restrict__ isect_ids,
                                  int32_t *__restrict__ flatten_ids) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CN) return;
    const int32_t r = radii[i];
    if (r <= 0) return;
    const Rect t = tile_rect(means2d[2 * i], means2d[2 * i + 1], r, tile_w, tile_h);
    const int64_t cid = i / N;
    const int64_t cam_part = cid << (32 + tile_n_bits);
    const int64_t dbits = (int64_t)__float_as_uint(depths[i]);
    int64_t k = (i == 0) ? 0 : cum_tiles[i - 1];
    for (uint32_t y = t.y0; y < t.y1; ++y)
        for (uint32_t x = t.x0; x < t.x1; ++x) {
            const int64_t tile_id = (int64_t)y * tile_w + x;
            isect_ids[k] = cam_part | (tile_id << 32) | dbits;
            flatten_ids[k] = (int32_t)i;
            ++k;
        }
}

// offsets[t] = lower bound of (cam,tile) key t in the sorted ids; tiles after the last entry get M.
__global__ void isect_offsets_kernel(const int64_t *__restrict__ isect_ids, int64_t M, int64_t T, int n_tiles,
                                     int tile_n_bits, int32_t *__restrict__ offsets) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= M) return;
    auto lin = [&](int64_t key) -> int64_t {
        const int64_t hi = key >> 32;
        const int64_t cid = hi >> tile_n_bits, tid = hi & (((int64_t)1 << tile_n_bits) - 1);
        return cid * n_tiles + tid;
    };
    const int64_t cur = lin(isect_ids[k]);
    if (k == 0) {
        for (int64_t t = 0; t <= cur; ++t) offsets[t] = 0;
    } else {
        const int64_t prev = lin(isect_ids[k - 1]);
        for (int64_t t = prev + 1; t <= cur; ++t) offsets[t] = (int32_t)k;
    }
    if (k == M - 1)
        for (int64_t t = cur + 1; t < T; ++t) offsets[t] = (int32_t)M;
}

int bit_length(uint32_t v) {
    int n = 0;
    while (v) { ++n; v >>= 1; }
    return n;
}

// ---- int32 -> int64 inclusive scan in three launches: 2048-element blocks (local scan + block total), one workgroup over the
// block totals, the block bases added back ------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

__global__ __launch_bounds__(SCAN_THREADS) void scan_blocks_kernel(const int32_t *__restrict__ in, int64_t n,
                                                                   int64_t *__restrict__ out,
                                                                   int64_t *__restrict__ block_sums) {
    __shared__ long long s_w[SCAN_THREADS / 64];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)t * SCAN_ITEMS;
    long long v[SCAN_ITEMS];
    long long sum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? (long long)in[base + k] : 0;
        sum += v[k];
    }
    long long incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const long long u = __shfl_up(incl, off, 64);
        if ((t & 63) >= off) incl += u;
    }
    if ((t & 63) == 63) s_w[t >> 6] = incl;
    __syncthreads();
    long long run = incl - sum, total = 0;
    for (int w = 0; w < SCAN_THREADS / 64; ++w) {
        if (w < (t >> 6)) run += s_w[w];
        total += s_w[w];
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        run += v[k];
        if (base + k < n) out[base + k] = run;
    }
    if (t == 0) block_sums[blockIdx.x] = total;
}

__global__ __launch_bounds__(1024) void scan_totals_kernel(int64_t *__restrict__ block_sums, int64_t nb) {
    __shared__ long long s_scan[1024];
    const int t = threadIdx.x;
    const int64_t per = (nb + 1023) / 1024;
    const int64_t lo = min(nb, (int64_t)t * per), hi = min(nb, lo + per);
    long long sum = 0;
    for (int64_t i = lo; i < hi; ++i) sum += block_sums[i];
    s_scan[t] = sum;
    for (int off = 1; off < 1024; off <<= 1) {
        __syncthreads();
        const long long add = (t >= off) ? s_scan[t - off] : 0;
        __syncthreads();
        s_scan[t] += add;
    }
    __syncthreads();
    long long run = s_scan[t] - sum;                       // exclusive base of this thread's chunk
    for (int64_t i = lo; i < hi; ++i) {
        const long long c = block_sums[i];
        block_sums[i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(int64_t *__restrict__ out, int64_t n,
                                                                const int64_t *__restrict__ block_bases) {
    const long long add = block_bases[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const int64_t i = base + (int64_t)k * SCAN_THREADS;
        if (i < n) out[i] += add;
    }
}

}  // namespace

extern "C" int gsx_isect_count(const float *means2d, const int32_t *radii, int64_t CN, int tile_w, int tile_h,
                               int32_t *tiles_per_gauss, void *stream) {
    GSX_CHECK_ARG(means2d && radii && tiles_per_gauss && CN >= 0 && tile_w > 0 && tile_h > 0);
    if (CN == 0) return GSX_OK;
    hipLaunchKernelGGL(isect_count_kernel, dim3((unsigned)((CN + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       means2d, radii, CN, tile_w, tile_h, tiles_per_gauss);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int64_t gsx_scan_workspace_bytes(int64_t CN) {
    const int64_t nb = (CN > 0 ? CN : 1 + SCAN_BLOCK - 1) / SCAN_BLOCK + 1;
    return gsx_align256(nb * 8) + 256;
}

extern "C" int gsx_isect_scan(const int32_t *tiles_per_gauss, int64_t CN, int64_t *cum_tiles, void *workspace,
                              int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(tiles_per_gauss && cum_tiles && CN >= 0);
    if (CN == 0) return GSX_OK;
    const int64_t nb = (CN + SCAN_BLOCK - 1) / SCAN_BLOCK;
    if (!workspace || workspace_bytes < nb * 8) {
        gsx_set_error("gsx_isect_scan: workspace too small (%lld < %lld)", (long long)workspace_bytes, (long long)(nb * 8));
        return GSX_E_WORKSPACE;
    }
    GSX_CHECK_ARG(nb < ((int64_t)1 << 31));
    hipStream_t st = (hipStream_t)stream;
    int64_t *sums = (int64_t *)workspace;
    hipLaunchKernelGGL(scan_blocks_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, tiles_per_gauss, CN, cum_tiles,
                       sums);
    GSX_CHECK_LAUNCH();
    if (nb > 1) {
        hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(1024), 0, st, sums, nb);
        GSX_CHECK_LAUNCH();
        hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, st, cum_tiles, CN, sums);
        GSX_CHECK_LAUNCH();
    }
    return GSX_OK;
}

extern "C" int gsx_isect_emit(const float *means2d, const int32_t *radii, const float *depths, const int64_t *cum_tiles,
                              int64_t N, int64_t C, int tile_w, int tile_h, int64_t M, int64_t *isect_ids,
                              int32_t *flatten_ids, void *stream) {
    GSX_CHECK_ARG(means2d && radii && depths && cum_tiles && N >= 0 && C >= 1 && tile_w > 0 && tile_h > 0 && M >= 0);
    if (M == 0 || N == 0) return GSX_OK;
    GSX_CHECK_ARG(isect_ids && flatten_ids);
    const int64_t CN = C * N;
    const int tile_n_bits = bit_length((uint32_t)(tile_w * tile_h));
    hipLaunchKernelGGL(isect_emit_kernel, dim3((unsigned)((CN + 255) / 256)), dim3(256), 0, (hipStream_t)stream, means2d,
                       radii, depths, cum_tiles, N, CN, tile_w, tile_h, tile_n_bits, isect_ids, flatten_ids);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_isect_offset_encode(const int64_t *isect_ids, int64_t M, int64_t C, int tile_w, int tile_h,
                                       int32_t *offsets, void *stream) {
    GSX_CHECK_ARG(offsets && C >= 1 && tile_w > 0 && tile_h > 0 && M >= 0);
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_tiles = (int64_t)tile_w * tile_h;
    const int64_t T = C * n_tiles;
    if (M == 0) {
        if (!gsx_zero_async(offsets, T, st)) return GSX_E_LAUNCH;
        return GSX_OK;
    }
    GSX_CHECK_ARG(isect_ids);
    const int tile_n_bits = bit_length((uint32_t)n_tiles);
    hipLaunchKernelGGL(isect_offsets_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, isect_ids, M, T,
                       (int)n_tiles, tile_n_bits, offsets);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
