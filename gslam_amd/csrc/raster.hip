// raster.hip — K8/K9: tiled alpha-blended rasterisation of splat records, forward and backward.
// Replaces the gsplat fork's `rasterize_to_pixels` (gslam/rasterization.py:325-339; 3-tuple return with n_touched).
// Maths: SURVEY.md §9.3 / §9.4.
//
// Mapping (MI355X, wave64): one 256-thread workgroup per 16x16 tile = 4 wavefronts, each wavefront owns a
// 16x4-pixel strip.  The tile's depth-sorted list is staged through LDS 256 records at a time (one coalesced
// id load + one 48-byte record gather per thread); the compositing loop reads each record as an LDS broadcast
// (3 x ds_read_b128, same address in all lanes -> conflict free).  Termination is voted per wavefront
// (64-bit ballot) inside a batch and per workgroup between batches.  n_touched is counted with a wave ballot +
// popcount into an LDS counter and flushed with one global atomic per staged Gaussian.  The backward replays the
// list back to front, reduces each Gaussian's gradient over the 64 lanes with DPP row/bcast adds, merges the four
// wavefronts in LDS and issues one record-shaped (48-byte contiguous) atomic add per (tile, Gaussian).
#include "gsx_common.h"

namespace {

constexpr int BLOCK = 256;

template <int RS>
__device__ __forceinline__ void stage_record(float *s_rec, int t, const float *__restrict__ rec, int g) {
    const float4 *src = reinterpret_cast<const float4 *>(rec + (int64_t)g * RS);
    float4 *dst = reinterpret_cast<float4 *>(s_rec + t * RS);
#pragma unroll
    for (int k = 0; k < RS / 4; ++k) dst[k] = src[k];
}

template <int CH, int RS>
__global__ __launch_bounds__(BLOCK) void raster_fwd_kernel(const float *__restrict__ rec, const float *__restrict__ bg,
                                                           const int32_t *__restrict__ offsets,
                                                           const int32_t *__restrict__ flatten_ids, int64_t M, int W,
                                                           int H, int tile_w, int tile_h, float vis_min_T,
                                                           float *__restrict__ render, float *__restrict__ alphas,
                                                           int32_t *__restrict__ last_ids,
                                                           int32_t *__restrict__ n_touched) {
    __shared__ __attribute__((aligned(16))) float s_rec[BLOCK * RS];
    __shared__ int s_id[BLOCK];
    __shared__ int s_cnt[BLOCK];

    const int tiles_per_cam = tile_w * tile_h;
    const int tile = blockIdx.x;
    const int n_tiles_total = gridDim.x;
    const int c = tile / tiles_per_cam;
    const int tl = tile - c * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    const int t = threadIdx.x;
    const int px = tx * GSX_TILE + (t & 15), py = ty * GSX_TILE + (t >> 4);
    const bool inside = (px < W) && (py < H);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;

    const int start = offsets[tile];
    const int end = (tile + 1 < n_tiles_total) ? offsets[tile + 1] : (int)M;
    const int n_batches = (end - start + BLOCK - 1) / BLOCK;

    float T = 1.0f;
    float pix[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) pix[k] = 0.f;
    int last = -1;
    bool done = !inside;

    for (int b = 0; b < n_batches; ++b) {
        // all four wavefronts finished -> stop staging (also protects the LDS buffers of the previous batch)
        if (__syncthreads_and(done)) break;
        const int batch_start = start + b * BLOCK;
        const int bsize = min(BLOCK, end - batch_start);
        if (t < bsize) {
            const int g = flatten_ids[batch_start + t];
            s_id[t] = g;
            stage_record<RS>(s_rec, t, rec, g);
        }
        s_cnt[t] = 0;
        __syncthreads();
        for (int j = 0; j < bsize; ++j) {
            if (__all(done)) break;  // wave-uniform
            const float4 r0 = reinterpret_cast<const float4 *>(s_rec + j * RS)[0];
            const float4 r1 = reinterpret_cast<const float4 *>(s_rec + j * RS)[1];
            const float dx = r0.x - fx, dy = r0.y - fy;
            const float sigma = 0.5f * (r0.z * dx * dx + r1.x * dy * dy) + r0.w * dx * dy;
            const float alpha = fminf(GSX_ALPHA_MAX, r1.y * __expf(-sigma));
            bool valid = !done && (sigma >= 0.0f) && (alpha >= GSX_ALPHA_MIN);
            const float nT = T * (1.0f - alpha);
            if (valid && nT <= GSX_T_MIN) { done = true; valid = false; }
            bool touched = false;
            if (valid) {
                const float vis = alpha * T;
                float col[6];
                col[0] = r1.z; col[1] = r1.w;
                if (RS > 8) {
                    const float4 r2 = reinterpret_cast<const float4 *>(s_rec + j * RS)[RS > 8 ? 2 : 0];
                    col[2] = r2.x; col[3] = r2.y; col[4] = r2.z; col[5] = r2.w;
                }
#pragma unroll
                for (int k = 0; k < CH; ++k) pix[k] += col[k] * vis;
                touched = nT > vis_min_T;
                last = batch_start + j;
                T = nT;
            }
            const unsigned long long m = __ballot(touched);
            if (m != 0ull && (t & 63) == 0) atomicAdd(&s_cnt[j], __popcll(m));
        }
        __syncthreads();
        if (t < bsize) {
            const int cnt = s_cnt[t];
            if (cnt > 0) atomicAdd(&n_touched[s_id[t]], cnt);
        }
    }
    if (inside) {
        const int64_t p = ((int64_t)c * H + py) * W + px;
#pragma unroll
        for (int k = 0; k < CH; ++k) render[p * CH + k] = pix[k] + (bg ? T * bg[c * CH + k] : 0.f);
        alphas[p] = 1.0f - T;
        last_ids[p] = last;
    }
}

template <int CH, int RS, bool ABS>
__global__ __launch_bounds__(BLOCK) void raster_bwd_kernel(
    const float *__restrict__ rec, const float *__restrict__ bg, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int64_t M, int W, int H, int tile_w, int tile_h,
    const float *__restrict__ alphas, const int32_t *__restrict__ last_ids, const float *__restrict__ v_render,
    const float *__restrict__ v_alphas, float *__restrict__ v_rec, float *__restrict__ v_abs) {
    __shared__ __attribute__((aligned(16))) float s_rec[BLOCK * RS];
    __shared__ __attribute__((aligned(16))) float s_grad[BLOCK * RS];
    __shared__ float s_abs[ABS ? BLOCK * 2 : 2];
    __shared__ int s_id[BLOCK];
    __shared__ int s_wmax[BLOCK / GSX_WAVE];
    constexpr int NG = 6 + CH;  // gradient entries per record: xy(2) conic(3) opacity(1) colors(CH)

    const int tiles_per_cam = tile_w * tile_h;
    const int tile = blockIdx.x;
    const int c = tile / tiles_per_cam;
    const int tl = tile - c * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int px = tx * GSX_TILE + (t & 15), py = ty * GSX_TILE + (t >> 4);
    const bool inside = (px < W) && (py < H);
    const float fx = (float)px + 0.5f, fy = (float)py + 0.5f;
    const int64_t p = ((int64_t)c * H + min(py, H - 1)) * W + min(px, W - 1);

    const int start = offsets[tile];
    const int last = inside ? last_ids[p] : -1;
    // workgroup max of `last`
    int wmax = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, 64));
    if (lane == 0) s_wmax[wave] = wmax;
    __syncthreads();
    int bmax = s_wmax[0];
#pragma unroll
    for (int w = 1; w < BLOCK / GSX_WAVE; ++w) bmax = max(bmax, s_wmax[w]);
    if (bmax < start) return;  // nothing composited in this tile (uniform)

    const float T_final = inside ? 1.0f - alphas[p] : 1.0f;
    float T = T_final;
    float vo[CH], buf[CH];
    float bg_dot = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        vo[k] = inside ? v_render[p * CH + k] : 0.f;
        buf[k] = 0.f;
        if (bg) bg_dot += bg[c * CH + k] * vo[k];
    }
    const float va_out = inside ? v_alphas[p] : 0.f;

    const int n = bmax - start + 1;
    const int n_batches = (n + BLOCK - 1) / BLOCK;
    for (int b = n_batches - 1; b >= 0; --b) {
        const int batch_start = start + b * BLOCK;
        const int bsize = min(BLOCK, start + n - batch_start);
        __syncthreads();
        if (t < bsize) {
            const int g = flatten_ids[batch_start + t];
            s_id[t] = g;
            stage_record<RS>(s_rec, t, rec, g);
        }
#pragma unroll
        for (int k = 0; k < RS; ++k) s_grad[k * BLOCK + t] = 0.f;  // zero the whole [BLOCK*RS] buffer, conflict-free
        if (ABS) { s_abs[t] = 0.f; s_abs[BLOCK + t] = 0.f; }
        __syncthreads();
        for (int j = bsize - 1; j >= 0; --j) {
            const int e = batch_start + j;
            const float4 r0 = reinterpret_cast<const float4 *>(s_rec + j * RS)[0];
            const float4 r1 = reinterpret_cast<const float4 *>(s_rec + j * RS)[1];
            const float dx = r0.x - fx, dy = r0.y - fy;
            const float a = r0.z, bq = r0.w, cq = r1.x, opac = r1.y;
            const float sigma = 0.5f * (a * dx * dx + cq * dy * dy) + bq * dx * dy;
            const float vis = __expf(-sigma);
            const float alpha = fminf(GSX_ALPHA_MAX, opac * vis);
            const bool valid = (e <= last) && (sigma >= 0.0f) && (alpha >= GSX_ALPHA_MIN);
            if (!__any(valid)) continue;  // wave-uniform skip
            float gr[NG];
#pragma unroll
            for (int k = 0; k < NG; ++k) gr[k] = 0.f;
            float gax = 0.f, gay = 0.f;
            if (valid) {
                float col[6];
                col[0] = r1.z; col[1] = r1.w;
                if (RS > 8) {
                    const float4 r2 = reinterpret_cast<const float4 *>(s_rec + j * RS)[RS > 8 ? 2 : 0];
                    col[2] = r2.x; col[3] = r2.y; col[4] = r2.z; col[5] = r2.w;
                }
                const float ra = 1.0f / (1.0f - alpha);
                T *= ra;
                const float fac = alpha * T;
                float v_alpha = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    gr[6 + k] = fac * vo[k];
                    v_alpha += (col[k] * T - buf[k] * ra) * vo[k];
                    buf[k] += col[k] * fac;
                }
                v_alpha += T_final * ra * va_out;
                v_alpha -= T_final * ra * bg_dot;
                if (opac * vis <= GSX_ALPHA_MAX) {
                    const float v_sigma = -opac * vis * v_alpha;
                    gr[2] = 0.5f * v_sigma * dx * dx;
                    gr[3] = v_sigma * dx * dy;
                    gr[4] = 0.5f * v_sigma * dy * dy;
                    gr[0] = v_sigma * (a * dx + bq * dy);
                    gr[1] = v_sigma * (bq * dx + cq * dy);
                    gr[5] = vis * v_alpha;
                    if (ABS) { gax = fabsf(gr[0]); gay = fabsf(gr[1]); }
                }
            }
#pragma unroll
            for (int k = 0; k < NG; ++k) {
                const float tot = gsx_wave_sum(gr[k]);
                if (lane == 0) atomicAdd(&s_grad[j * RS + k], tot);
            }
            if (ABS) {
                const float tx_ = gsx_wave_sum(gax), ty_ = gsx_wave_sum(gay);
                if (lane == 0) { atomicAdd(&s_abs[2 * j], tx_); atomicAdd(&s_abs[2 * j + 1], ty_); }
            }
        }
        __syncthreads();
        // record-shaped flush: consecutive lanes -> consecutive floats of one 48-byte record row
        for (int i = t; i < bsize * RS; i += BLOCK) {
            const int j = i / RS, k = i - j * RS;
            if (k < NG) {
                const float v = s_grad[i];
                if (v != 0.f) atomicAdd(&v_rec[(int64_t)s_id[j] * RS + k], v);
            }
        }
        if (ABS) {
            for (int i = t; i < bsize * 2; i += BLOCK) {
                const float v = s_abs[i];
                if (v != 0.f) atomicAdd(&v_abs[(int64_t)s_id[i >> 1] * 2 + (i & 1)], v);
            }
        }
    }
}

}  // namespace

extern "C" int gsx_raster_fwd(const float *rec, int CH, const float *backgrounds, const int32_t *offsets,
                              const int32_t *flatten_ids, int64_t M, int64_t C, int W, int H, int tile_w, int tile_h,
                              float visibility_min_T, float *render, float *alphas, int32_t *last_ids,
                              int32_t *n_touched, void *stream) {
    GSX_CHECK_ARG(offsets && render && alphas && last_ids && n_touched && C >= 1 && W > 0 && H > 0);
    GSX_CHECK_ARG(tile_w == (W + GSX_TILE - 1) / GSX_TILE && tile_h == (H + GSX_TILE - 1) / GSX_TILE);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31));
    GSX_CHECK_ARG(M == 0 || (rec && flatten_ids));
    const int64_t T = C * tile_w * tile_h;
    GSX_CHECK_ARG(T < ((int64_t)1 << 31));
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(ch, rs)                                                                                             \
    hipLaunchKernelGGL((raster_fwd_kernel<ch, rs>), dim3((unsigned)T), dim3(BLOCK), 0, st, rec, backgrounds,      \
                       offsets, flatten_ids, M, W, H, tile_w, tile_h, visibility_min_T, render, alphas, last_ids, \
                       n_touched)
    switch (CH) {
        case 1: LAUNCH(1, 8); break;
        case 2: LAUNCH(2, 8); break;
        case 3: LAUNCH(3, 12); break;
        case 4: LAUNCH(4, 12); break;
        case 5: LAUNCH(5, 12); break;
        default: gsx_set_error("gsx_raster_fwd: CH=%d unsupported (1..5)", CH); return GSX_E_UNSUPPORTED;
    }
#undef LAUNCH
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_raster_bwd(const float *rec, int CH, const float *backgrounds, const int32_t *offsets,
                              const int32_t *flatten_ids, int64_t M, int64_t C, int W, int H, int tile_w, int tile_h,
                              const float *alphas, const int32_t *last_ids, const float *v_render,
                              const float *v_alphas, float *v_rec, float *v_abs, void *stream) {
    GSX_CHECK_ARG(offsets && alphas && last_ids && v_render && v_alphas && C >= 1 && W > 0 && H > 0);
    GSX_CHECK_ARG(tile_w == (W + GSX_TILE - 1) / GSX_TILE && tile_h == (H + GSX_TILE - 1) / GSX_TILE);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31));
    if (M == 0) return GSX_OK;
    GSX_CHECK_ARG(rec && flatten_ids && v_rec);
    const int64_t T = C * tile_w * tile_h;
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(ch, rs)                                                                                              \
    do {                                                                                                            \
        if (v_abs)                                                                                                  \
            hipLaunchKernelGGL((raster_bwd_kernel<ch, rs, true>), dim3((unsigned)T), dim3(BLOCK), 0, st, rec,      \
                               backgrounds, offsets, flatten_ids, M, W, H, tile_w, tile_h, alphas, last_ids,       \
                               v_render, v_alphas, v_rec, v_abs);                                                   \
        else                                                                                                        \
            hipLaunchKernelGGL((raster_bwd_kernel<ch, rs, false>), dim3((unsigned)T), dim3(BLOCK), 0, st, rec,     \
                               backgrounds, offsets, flatten_ids, M, W, H, tile_w, tile_h, alphas, last_ids,       \
                               v_render, v_alphas, v_rec, v_abs);                                                   \
    } while (0)
    switch (CH) {
        case 1: LAUNCH(1, 8); break;
        case 2: LAUNCH(2, 8); break;
        case 3: LAUNCH(3, 12); break;
        case 4: LAUNCH(4, 12); break;
        case 5: LAUNCH(5, 12); break;
        default: gsx_set_error("gsx_raster_bwd: CH=%d unsupported (1..5)", CH); return GSX_E_UNSUPPORTED;
    }
#undef LAUNCH
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
