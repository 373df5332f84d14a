// isect.hip — K3..K7: tile intersection, depth sort and per-tile offsets.  Replaces gsplat `isect_tiles` and
// `isect_offset_encode` (gslam/rasterization.py:259-274).  Integer-exact contract, SURVEY.md §9.2:
//   key = cam << (32 + tile_n_bits) | tile << 32 | float_bits(depth),  value = flatten id (c*N+g),
//   stable order (ties keep emission order = ascending flatten id, tiles row-major inside the bbox).
//
// v1 of the sort: device-wide LSD radix sort over the live key bits via rocPRIM (temporary yardstick);
// the tile-binned LDS sort that replaces it is tracked in DESIGN.md.
#include "gsx_common.h"

#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/transform_iterator.hpp>

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

struct I32ToI64 {
    __device__ __host__ int64_t operator()(int32_t v) const { return (int64_t)v; }
};

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
    size_t bytes = 0;
    auto in = rocprim::make_transform_iterator((const int32_t *)nullptr, I32ToI64());
    if (rocprim::inclusive_scan(nullptr, bytes, in, (int64_t *)nullptr, (size_t)(CN > 0 ? CN : 1),
                                rocprim::plus<int64_t>()) != hipSuccess)
        return -1;
    return gsx_align256((int64_t)bytes) + 256;
}

extern "C" int gsx_isect_scan(const int32_t *tiles_per_gauss, int64_t CN, int64_t *cum_tiles, void *workspace,
                              int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(tiles_per_gauss && cum_tiles && CN >= 0);
    if (CN == 0) return GSX_OK;
    size_t bytes = (size_t)workspace_bytes;
    auto in = rocprim::make_transform_iterator(tiles_per_gauss, I32ToI64());
    size_t need = 0;
    (void)rocprim::inclusive_scan(nullptr, need, in, cum_tiles, (size_t)CN, rocprim::plus<int64_t>());
    if (!workspace || bytes < need) {
        gsx_set_error("gsx_isect_scan: workspace too small (%zu < %zu)", bytes, need);
        return GSX_E_WORKSPACE;
    }
    hipError_t e = rocprim::inclusive_scan(workspace, bytes, in, cum_tiles, (size_t)CN, rocprim::plus<int64_t>(),
                                           (hipStream_t)stream);
    if (e != hipSuccess) {
        gsx_set_error("gsx_isect_scan: %s", hipGetErrorString(e));
        return GSX_E_LAUNCH;
    }
    return GSX_OK;
}

extern "C" int64_t gsx_isect_sort_workspace_bytes(int64_t M) {
    if (M <= 0) return 256;
    size_t bytes = 0;
    if (rocprim::radix_sort_pairs(nullptr, bytes, (const uint64_t *)nullptr, (uint64_t *)nullptr,
                                  (const int32_t *)nullptr, (int32_t *)nullptr, (size_t)M, 0, 64) != hipSuccess)
        return -1;
    // unsorted keys + values live in the workspace next to rocPRIM's temporary storage
    return gsx_align256((int64_t)bytes) + gsx_align256(M * 8) + gsx_align256(M * 4) + 256;
}

extern "C" int gsx_isect_emit_sort(const float *means2d, const int32_t *radii, const float *depths,
                                   const int64_t *cum_tiles, int64_t N, int64_t C, int tile_w, int tile_h, int64_t M,
                                   int sort, int64_t *isect_ids, int32_t *flatten_ids, void *workspace,
                                   int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(means2d && radii && depths && cum_tiles && N >= 0 && C >= 1 && tile_w > 0 && tile_h > 0 && M >= 0);
    if (M == 0 || N == 0) return GSX_OK;
    GSX_CHECK_ARG(isect_ids && flatten_ids);
    hipStream_t st = (hipStream_t)stream;
    const int64_t CN = C * N;
    const int tile_n_bits = bit_length((uint32_t)(tile_w * tile_h));
    const int cam_n_bits = bit_length((uint32_t)C);
    const unsigned blocks = (unsigned)((CN + 255) / 256);
    if (!sort) {
        hipLaunchKernelGGL(isect_emit_kernel, dim3(blocks), dim3(256), 0, st, means2d, radii, depths, cum_tiles, N,
                           CN, tile_w, tile_h, tile_n_bits, isect_ids, flatten_ids);
        GSX_CHECK_LAUNCH();
        return GSX_OK;
    }
    if (!workspace || workspace_bytes < gsx_isect_sort_workspace_bytes(M)) {
        gsx_set_error("gsx_isect_emit_sort: workspace too small");
        return GSX_E_WORKSPACE;
    }
    char *ws = (char *)workspace;
    int64_t *keys_in = (int64_t *)ws;
    int32_t *vals_in = (int32_t *)(ws + gsx_align256(M * 8));
    void *tmp = ws + gsx_align256(M * 8) + gsx_align256(M * 4);
    size_t tmp_bytes = (size_t)(workspace_bytes - gsx_align256(M * 8) - gsx_align256(M * 4));
    hipLaunchKernelGGL(isect_emit_kernel, dim3(blocks), dim3(256), 0, st, means2d, radii, depths, cum_tiles, N, CN,
                       tile_w, tile_h, tile_n_bits, keys_in, vals_in);
    GSX_CHECK_LAUNCH();
    const unsigned end_bit = (unsigned)(32 + tile_n_bits + cam_n_bits);
    hipError_t e = rocprim::radix_sort_pairs(tmp, tmp_bytes, (const uint64_t *)keys_in, (uint64_t *)isect_ids,
                                             (const int32_t *)vals_in, flatten_ids, (size_t)M, 0u, end_bit, st);
    if (e != hipSuccess) {
        gsx_set_error("gsx_isect_emit_sort: radix sort: %s", hipGetErrorString(e));
        return GSX_E_LAUNCH;
    }
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
