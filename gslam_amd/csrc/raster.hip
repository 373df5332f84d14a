// raster.hip — K8/K9: tiled alpha-blended rasterisation of splat records, forward and backward.
// Replaces the gsplat fork's `rasterize_to_pixels` (gslam/rasterization.py:325-339; 3-tuple return with n_touched).
// Maths: SURVEY.md §9.3 / §9.4.
//
// wave64-native mapping.  A workgroup is one 16x16 tile; each of its wavefronts owns an 8x8-pixel quadrant (or a 16x8
// half tile with two pixels per lane when the chip is full) and walks the tile's depth-sorted list in chunks of 64 entries
// with the LANES MAPPED TO ENTRIES first:
//   1. lane i loads entry i of the chunk (coalesced id load + one 48-byte record gather),
//   2. lane i decides - exactly and conservatively - whether its Gaussian can reach alpha >= 1/255 anywhere in
//      the wavefront's pixel block (minimum of the conic quadratic over the block's rectangle),
//   3. a 64-bit ballot compacts the survivors into a per-wavefront LDS list (raster_v4.inc),
//   4. now LANES = PIXELS: a counted loop over the list with broadcast LDS reads and branch-free compositing.
// The kernels that ship: raster_fwd_kernel4q (forward), raster_bwd_kernel4q (backward, one camera / geometry-only),
// raster_bwd_kernel4 (backward, full chip), raster_bwd_kernel2 (absgrad side channel only).  The earlier generations
// (block-per-tile LDS staging, v_readlane broadcast loop, LDS-atomic accumulation modes) are gone from the product; the
// parity tests check these kernels against the CPU oracle.
#include <stdlib.h>

#include <type_traits>

#include "gsx_common.h"
#include "tile_sort_lds.h"


namespace {

constexpr int QUAD = 8;  // a wavefront renders an 8x8 pixel quadrant of the 16x16 tile

template <int RS>
struct Rec {
    float v[RS];
};

template <int RS>
__device__ __forceinline__ Rec<RS> load_record(const float *__restrict__ rec, int g) {
    Rec<RS> r;
    const float4 *src = reinterpret_cast<const float4 *>(rec + (int64_t)g * RS);
#pragma unroll
    for (int k = 0; k < RS / 4; ++k) {
        const float4 q = src[k];
        r.v[4 * k] = q.x; r.v[4 * k + 1] = q.y; r.v[4 * k + 2] = q.z; r.v[4 * k + 3] = q.w;
    }
    return r;
}

// a splat record held in SGPRs (fetched with s_load_dwordx4 from a wave-uniform address)
template <int RS>
struct SRec {
    gsx_f4 a, b, c;  // a = (mx, my, conic a, conic b)  b = (conic c, opacity, col0, col1)  c = (col2, col3, col4, pad)
    __device__ __forceinline__ float color(int k) const {
        switch (k) {
            case 0: return b.z;
            case 1: return b.w;
            case 2: return c.x;
            case 3: return c.y;
            default: return c.z;
        }
    }
};

template <int RS>
__device__ __forceinline__ SRec<RS> bcast_record(const float (&v)[RS], int j);

template <int RS>
__device__ __forceinline__ SRec<RS> sload_record(const float *__restrict__ rec, int g_uniform) {
    const gsx_cf4p p = gsx_scalar_ptr(rec + (int64_t)g_uniform * RS);
    SRec<RS> r;
    r.a = p[0];
    r.b = p[1];
    if (RS > 8) r.c = p[2]; else r.c = gsx_f4{0.f, 0.f, 0.f, 0.f};
    return r;
}

__device__ __forceinline__ float bcast(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// the same record broadcast out of lane j's VGPRs with v_readlane (no memory latency to hide, 1 VALU slot per float)
template <int RS>
__device__ __forceinline__ SRec<RS> bcast_record(const float (&v)[RS], int j) {
    SRec<RS> r;
    r.a = gsx_f4{bcast(v[0], j), bcast(v[1], j), bcast(v[2], j), bcast(v[3], j)};
    r.b = gsx_f4{bcast(v[4], j), bcast(v[5], j), bcast(v[6], j), bcast(v[7], j)};
    if constexpr (RS > 8) r.c = gsx_f4{bcast(v[8], j), bcast(v[9], j), bcast(v[10], j), 0.f};
    else r.c = gsx_f4{0.f, 0.f, 0.f, 0.f};
    return r;
}

// Can this Gaussian reach alpha >= 1/255 at ANY point of the rectangle [x0,x1]x[y0,y1] (pixel-centre coords)?
// alpha >= 1/255  <=>  sigma <= log(255*opac).  sigma is a convex quadratic (PD conic): its minimum over the
// rectangle is 0 if the mean is inside, otherwise it lies on one of the four edges, where it is a clamped 1-D
// parabola.  A small slack keeps the test conservative under fp32 rounding; anything uncertain is kept (the exact
// per-pixel predicate runs afterwards), so results are identical to visiting every entry.
__device__ __forceinline__ bool may_touch(float mx, float my, float a, float b, float c, float opac, float x0,
                                          float y0, float x1, float y1) {
    if (!(opac * 255.0f >= 1.0f)) return false;           // alpha < 1/255 everywhere (also NaN-safe)
    if (!(a > 0.0f && c > 0.0f)) return true;               // not a PD conic: keep, let the exact test decide
    // hardware log2 / reciprocal (1 ulp): the IEEE divides and the denormal-safe logarithm were 30 of this test's ~110
    // instructions; the argument is >= 1 and the 0.1 % + 1e-3 slack below is five orders above their error
    const float tau = __builtin_amdgcn_logf(opac * 255.0f) * 0.69314718f;
    const float u0 = x0 - mx, u1 = x1 - mx, v0 = y0 - my, v1 = y1 - my;
    if (u0 <= 0.0f && u1 >= 0.0f && v0 <= 0.0f && v1 >= 0.0f) return true;
    const float inv_a = __builtin_amdgcn_rcpf(a), inv_c = __builtin_amdgcn_rcpf(c);
    float smin;
    {   // edges u = u0 / u = u1, v free in [v0,v1]
        float v = fminf(fmaxf(-b * u0 * inv_c, v0), v1);
        smin = 0.5f * (a * u0 * u0 + c * v * v) + b * u0 * v;
        v = fminf(fmaxf(-b * u1 * inv_c, v0), v1);
        smin = fminf(smin, 0.5f * (a * u1 * u1 + c * v * v) + b * u1 * v);
    }
    {   // edges v = v0 / v = v1, u free in [u0,u1]
        float u = fminf(fmaxf(-b * v0 * inv_a, u0), u1);
        smin = fminf(smin, 0.5f * (a * u * u + c * v0 * v0) + b * u * v0);
        u = fminf(fmaxf(-b * v1 * inv_a, u0), u1);
        smin = fminf(smin, 0.5f * (a * u * u + c * v1 * v1) + b * u * v1);
    }
    return !(smin > tau * 1.001f + 1e-3f);
}

// Launch order: workgroup b renders tile b (or tile_order[b]).  An XCD-band order (the workgroups that share an XCD - and
// its L2 - walk one contiguous band of tile rows, as ssim.hip does) was measured and dropped: the per-tile work is not
// uniform, a band through the image centre is heavier than one along its border, and the imbalance between XCDs cost 6 %
// in the headline loop (73.7 -> 78.1 us backward, 33.5 -> 35.6 forward) for a traffic the kernel is not bound by.
struct Quad {
    int tile, c, px, py, wave, lane;
    bool inside;
    float fx, fy, x0, y0, x1, y1;
};

__device__ __forceinline__ Quad make_quad(int tile_w, int tile_h, int W, int H, int tile = -1) {
    Quad q;
    const int tiles_per_cam = tile_w * tile_h;
    q.tile = tile >= 0 ? min(tile, (int)gridDim.x - 1) : (int)blockIdx.x;      // (a launch order is a permutation; garbage stays in range)
    q.c = q.tile / tiles_per_cam;
    const int tl = q.tile - q.c * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    q.wave = threadIdx.x >> 6;
    q.lane = threadIdx.x & 63;
    const int bx = tx * GSX_TILE + (q.wave & 1) * QUAD, by = ty * GSX_TILE + (q.wave >> 1) * QUAD;
    q.px = bx + (q.lane & 7);
    q.py = by + (q.lane >> 3);
    q.inside = (q.px < W) && (q.py < H);
    q.fx = (float)q.px + 0.5f;
    q.fy = (float)q.py + 0.5f;
    q.x0 = (float)bx + 0.5f; q.y0 = (float)by + 0.5f;
    q.x1 = (float)(bx + QUAD - 1) + 0.5f; q.y1 = (float)(by + QUAD - 1) + 0.5f;
    return q;
}

// =====================================================================================================================
// v3: two pixels per lane.  A wavefront owns a 16x8 half tile: lane l composites the pixels (x, y) and (x, y + 4) with
// x = l & 15, y = l >> 4.  Both pixels share dx, so half of the conic quadratic is computed once; the per-entry fixed
// cost (record broadcast, loop control, ballots, and in the backward the DPP row sums and LDS adds) is paid once per
// 128 pixels instead of once per 64.  Workgroup = 2 wavefronts = 128 threads per 16x16 tile.
// =====================================================================================================================
struct Half {
    int tile, c, px, py0, py1, wave, lane;
    bool in0, in1;
    float fx, fy0, x0, y0, x1, y1;
};

__device__ __forceinline__ Half make_half(int tile_w, int tile_h, int W, int H, int tile = -1) {
    Half q;
    const int tiles_per_cam = tile_w * tile_h;
    q.tile = tile >= 0 ? min(tile, (int)gridDim.x - 1) : (int)blockIdx.x;      // (a launch order is a permutation; garbage stays in range)
    q.c = q.tile / tiles_per_cam;
    const int tl = q.tile - q.c * tiles_per_cam;
    const int ty = tl / tile_w, tx = tl - ty * tile_w;
    q.wave = threadIdx.x >> 6;
    q.lane = threadIdx.x & 63;
    const int bx = tx * GSX_TILE, by = ty * GSX_TILE + q.wave * 8;
    q.px = bx + (q.lane & 15);
    q.py0 = by + (q.lane >> 4);
    q.py1 = q.py0 + 4;
    q.in0 = (q.px < W) && (q.py0 < H);
    q.in1 = (q.px < W) && (q.py1 < H);
    q.fx = (float)q.px + 0.5f;
    q.fy0 = (float)q.py0 + 0.5f;
    q.x0 = (float)bx + 0.5f; q.y0 = (float)by + 0.5f;
    q.x1 = (float)(bx + 15) + 0.5f; q.y1 = (float)(by + 7) + 0.5f;
    return q;
}

template <int CH, int RS, bool ABS>
__global__ __launch_bounds__(128) void raster_bwd_kernel2(
    const float *__restrict__ rec, const float *__restrict__ bg, const int32_t *__restrict__ offsets,
    const int32_t *__restrict__ flatten_ids, int64_t M, int has_end, int W, int H, int tile_w, int tile_h,
    const float *__restrict__ alphas, const int32_t *__restrict__ last_ids, const float *__restrict__ v_render,
    const float *__restrict__ v_alphas, float *__restrict__ v_rec, float *__restrict__ v_abs) {
    constexpr int NG = 6 + CH;
    constexpr int BATCH = 256;
    __shared__ __attribute__((aligned(16))) float s_grad[BATCH * RS];
    __shared__ float s_abs[ABS ? 2 * BATCH : 2];
    __shared__ int s_id[BATCH];
    __shared__ int s_wmax[2];

    const Half q = make_half(tile_w, tile_h, W, H);
    const int t = threadIdx.x;
    const int64_t p0 = ((int64_t)q.c * H + min(q.py0, H - 1)) * W + min(q.px, W - 1);
    const int64_t p1 = ((int64_t)q.c * H + min(q.py1, H - 1)) * W + min(q.px, W - 1);
    const int start = max(0, min(offsets[q.tile], (int)M));
    const int last0 = q.in0 ? last_ids[p0] : -1, last1 = q.in1 ? last_ids[p1] : -1;
    int wmax = max(last0, last1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wmax = max(wmax, __shfl_xor(wmax, off, 64));
    if (q.lane == 0) s_wmax[q.wave] = wmax;
    __syncthreads();
    const int bmax = max(s_wmax[0], s_wmax[1]);
    if (bmax < start) return;

    const float Tf0 = q.in0 ? 1.0f - alphas[p0] : 1.0f, Tf1 = q.in1 ? 1.0f - alphas[p1] : 1.0f;
    float T0 = Tf0, T1 = Tf1;
    float vo0[CH], vo1[CH], buf0[CH], buf1[CH];
    float bgd0 = 0.f, bgd1 = 0.f;
#pragma unroll
    for (int k = 0; k < CH; ++k) {
        vo0[k] = q.in0 ? v_render[p0 * CH + k] : 0.f;
        vo1[k] = q.in1 ? v_render[p1 * CH + k] : 0.f;
        buf0[k] = 0.f; buf1[k] = 0.f;
        if (bg) { bgd0 += bg[q.c * CH + k] * vo0[k]; bgd1 += bg[q.c * CH + k] * vo1[k]; }
    }
    const float ka0 = ((q.in0 && v_alphas) ? v_alphas[p0] : 0.f) - bgd0;   // (v_alpha_out - bg . v_out)
    const float ka1 = ((q.in1 && v_alphas) ? v_alphas[p1] : 0.f) - bgd1;

    const int n = bmax - start + 1;
    const int n_batches = (n + BATCH - 1) / BATCH;
    for (int b = n_batches - 1; b >= 0; --b) {
        const int batch_start = start + b * BATCH;
        const int bsize = min(BATCH, start + n - batch_start);
        __syncthreads();
        for (int i = t; i < bsize; i += 128) s_id[i] = flatten_ids[batch_start + i];
        for (int i = t; i < BATCH * RS; i += 128) s_grad[i] = 0.f;
        if (ABS) for (int i = t; i < 2 * BATCH; i += 128) s_abs[i] = 0.f;
        __syncthreads();
        for (int sub = 3; sub >= 0; --sub) {
            const int cbase = batch_start + sub * 64;
            if (cbase >= batch_start + bsize || cbase > wmax) continue;  // wave-uniform
            const int e = cbase + q.lane;
            const bool have = (e < batch_start + bsize) && (e <= wmax);
            const int g = have ? s_id[sub * 64 + q.lane] : 0;
            Rec<RS> r = load_record<RS>(rec, g);
            const bool maybe = have && may_touch(r.v[0], r.v[1], r.v[2], r.v[3], r.v[4], r.v[5], q.x0, q.y0, q.x1, q.y1);
            unsigned long long mask = __ballot(maybe);
            while (mask != 0ull) {
                const int j = 63 - __clzll((long long)mask);  // back to front
                mask &= ~(1ull << j);
                const SRec<RS> cur = bcast_record<RS>(r.v, j);
                const float a = cur.a.z, bq = cur.a.w, cq = cur.b.x, opac = cur.b.y;
                const float dx = cur.a.x - q.fx, dy0 = cur.a.y - q.fy0, dy1 = dy0 - 4.0f;
                const float h = 0.5f * a * dx * dx, bdx = bq * dx, hc = 0.5f * cq;
                const float sig0 = h + dy0 * (hc * dy0 + bdx), sig1 = h + dy1 * (hc * dy1 + bdx);
                const float vis0 = __expf(-sig0), vis1 = __expf(-sig1);
                const float al0 = fminf(GSX_ALPHA_MAX, opac * vis0), al1 = fminf(GSX_ALPHA_MAX, opac * vis1);
                const float ae0 = ((cbase + j <= last0) && (sig0 >= 0.0f) && (al0 >= GSX_ALPHA_MIN)) ? al0 : 0.0f;
                const float ae1 = ((cbase + j <= last1) && (sig1 >= 0.0f) && (al1 >= GSX_ALPHA_MIN)) ? al1 : 0.0f;
                if (!__any(fmaxf(ae0, ae1) > 0.0f)) continue;
                const float ra0 = __builtin_amdgcn_rcpf(1.0f - ae0), ra1 = __builtin_amdgcn_rcpf(1.0f - ae1);
                T0 *= ra0; T1 *= ra1;
                const float fac0 = ae0 * T0, fac1 = ae1 * T1;
                float gr[NG];
                float va0 = 0.f, va1 = 0.f;
#pragma unroll
                for (int k = 0; k < CH; ++k) {
                    const float ck = cur.color(k);
                    gr[6 + k] = fac0 * vo0[k] + fac1 * vo1[k];
                    va0 += (ck * T0 - buf0[k] * ra0) * vo0[k];
                    va1 += (ck * T1 - buf1[k] * ra1) * vo1[k];
                    buf0[k] += ck * fac0;
                    buf1[k] += ck * fac1;
                }
                va0 += Tf0 * ra0 * ka0;
                va1 += Tf1 * ra1 * ka1;
                const float vw0 = ((ae0 > 0.0f) && (opac * vis0 <= GSX_ALPHA_MAX)) ? vis0 : 0.0f;
                const float vw1 = ((ae1 > 0.0f) && (opac * vis1 <= GSX_ALPHA_MAX)) ? vis1 : 0.0f;
                const float vs0 = -opac * vw0 * va0, vs1 = -opac * vw1 * va1;
                const float vss = vs0 + vs1;
                const float sdy = vs0 * dy0 + vs1 * dy1;                 // sum v_sigma * dy
                gr[2] = 0.5f * vss * dx * dx;
                gr[3] = sdy * dx;
                gr[4] = 0.5f * (vs0 * dy0 * dy0 + vs1 * dy1 * dy1);
                gr[0] = a * dx * vss + bq * sdy;
                gr[1] = bq * dx * vss + cq * sdy;
                gr[5] = vw0 * va0 + vw1 * va1;
                float ab[2] = {0.f, 0.f};
                if (ABS) {
                    ab[0] = fabsf(vs0 * (a * dx + bq * dy0)) + fabsf(vs1 * (a * dx + bq * dy1));
                    ab[1] = fabsf(vs0 * (bq * dx + cq * dy0)) + fabsf(vs1 * (bq * dx + cq * dy1));
                }
                const int slot = sub * 64 + j;
                gsx_row16_sum<NG>(gr);
                if (ABS) gsx_row16_sum<2>(ab);
                if ((q.lane & 15) == 15) {
                    gsx_lds_fadd_row<NG>((unsigned)(uintptr_t)&s_grad[slot * RS], gr);
                    if (ABS) { gsx_lds_fadd(&s_abs[2 * slot], ab[0]); gsx_lds_fadd(&s_abs[2 * slot + 1], ab[1]); }
                }
            }
        }
        __syncthreads();
        for (int i = t; i < bsize * RS; i += 128) {
            const int j = i / RS, k = i - j * RS;
            if (k < NG) {
                const float v = s_grad[i];
                if (v != 0.f) atomicAdd(&v_rec[(int64_t)s_id[j] * RS + k], v);
            }
        }
        if (ABS) {
            for (int i = t; i < bsize * 2; i += 128) {
                const float v = s_abs[i];
                if (v != 0.f) atomicAdd(&v_abs[(int64_t)s_id[i >> 1] * 2 + (i & 1)], v);
            }
        }
    }
}

// DIAGNOSTIC build only (-DGSX_WG_TRACE, tools/dbg/wg_trace.sh): every wavefront of the quadrant kernels stamps its start /
// end (s_memrealtime, 100 MHz) and where it ran (HW_ID, XCC_ID) into a buffer of its own.  Nothing of this is compiled into
// the product library.
#ifdef GSX_WG_TRACE
__device__ unsigned long long *g_wg_trace[2] = {nullptr, nullptr};   // [0] forward kernels, [1] backward kernels
struct WgTrace {
    unsigned long long t0;
    int which;
    __device__ WgTrace(int w) : t0(__builtin_amdgcn_s_memrealtime()), which(w) {}
    __device__ ~WgTrace() {
        unsigned long long *b = g_wg_trace[which];
        if (b && (threadIdx.x & 63) == 0) {
            const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
            const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
            const size_t i = ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 4;
            b[i] = t0, b[i + 1] = t1, b[i + 2] = hw, b[i + 3] = xcc;
        }
    }
};
#define GSX_WG_TRACE_SCOPE(w) WgTrace _wg_trace(w);
// Phase attribution inside the fused tracking kernel (tools/dbg/phase_trace.py): GSX_PT(k) books the shader cycles since the
// previous stamp of this wavefront under category k; 16 counters per wavefront.
__device__ unsigned long long *g_phase_trace = nullptr;
struct PhaseTrace {
    unsigned long long last, acc[16];
    __device__ PhaseTrace() : last(__builtin_amdgcn_s_memtime()) {
        for (int k = 0; k < 16; ++k) acc[k] = 0;
    }
    __device__ __forceinline__ void stamp(int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[k] += now - last;
        last = now;
    }
    __device__ ~PhaseTrace() {
        if (g_phase_trace && (threadIdx.x & 63) == 0) {
            const size_t i = ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64) * 16;
            for (int k = 0; k < 16; ++k) g_phase_trace[i + k] = acc[k];
        }
    }
};
#define GSX_PT_SCOPE PhaseTrace _pt;
#define GSX_PT(k) _pt.stamp(k);
#define GSX_PT_COUNT(k, n) _pt.acc[k] += (unsigned long long)(n);
#else
#define GSX_WG_TRACE_SCOPE(w)
#define GSX_PT_SCOPE
#define GSX_PT(k)
#define GSX_PT_COUNT(k, n)
#endif
#include "raster_v4.inc"

}  // namespace

extern "C" int gsx_raster_fwd(const float *rec, int CH, const float *backgrounds, const int32_t *offsets,
                              const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                              int tile_w, int tile_h, float visibility_min_T, float *render, float *alphas,
                              int32_t *last_ids, int32_t *n_touched, const int32_t *tile_order, void *stream) {
    GSX_CHECK_ARG(offsets && render && alphas && last_ids && C >= 1 && W > 0 && H > 0);   // n_touched: NULL = not wanted
    GSX_CHECK_ARG(tile_w == (W + GSX_TILE - 1) / GSX_TILE && tile_h == (H + GSX_TILE - 1) / GSX_TILE);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31));
    GSX_CHECK_ARG(M == 0 || (rec && flatten_ids));
    const int64_t T = C * tile_w * tile_h;
    GSX_CHECK_ARG(T < ((int64_t)1 << 31));
    hipStream_t st = (hipStream_t)stream;
    // one pixel per lane, four quadrant wavefronts per tile: fastest forward at every size tried on MI355X (1 and 8 cameras)
#define LAUNCH(ch, rs)                                                                                               \
    do {                                                                                                             \
        if (n_touched)                                                                                               \
            hipLaunchKernelGGL((raster_fwd_kernel4q<ch, rs, true>), dim3((unsigned)T), dim3(256), 0, st, rec,        \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h,          \
                               visibility_min_T, render, alphas, last_ids, n_touched, tile_order);                   \
        else                                                                                                         \
            hipLaunchKernelGGL((raster_fwd_kernel4q<ch, rs, false>), dim3((unsigned)T), dim3(256), 0, st, rec,       \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h,          \
                               visibility_min_T, render, alphas, last_ids, n_touched, tile_order);                   \
    } while (0)
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

extern "C" int gsx_raster_fwd_track_loss(const float *rec, const float *backgrounds, const int32_t *offsets,
                                         const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W,
                                         int H, const float *gt, const float *exposure, float w_photo, float *render,
                                         float *alphas, int32_t *last_ids, float *v_render, float *loss_rows,
                                         const int32_t *tile_order, int32_t *tile_work, void *stream) {
    GSX_CHECK_ARG(offsets && alphas && last_ids && gt && exposure && v_render && loss_rows && C >= 1 && W > 0 && H > 0);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31) && (M == 0 || (rec && flatten_ids)));
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const int64_t T = C * tile_w * tile_h;
    GSX_CHECK_ARG(T < ((int64_t)1 << 31));
    TrackLossArgs la;
    la.gt = gt; la.exposure = exposure; la.w_photo = w_photo; la.v_render = v_render; la.rows = loss_rows;
    la.tile_work = tile_work; la.refiner_loss = 0;
    hipLaunchKernelGGL((raster_fwd_kernel4q<4, 12, false, true>), dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, rec,
                       backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, 0.5f, render, alphas,
                       last_ids, (int32_t *)nullptr, tile_order, la);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_raster_track_fused(const float *rec, const float *backgrounds, const int32_t *offsets,
                                      const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                                      const float *gt, const float *exposure, float w_photo, float *alphas,
                                      int32_t *last_ids, float *v_render, float *loss_rows, float *v_rec,
                                      const int32_t *tile_order, int32_t *tile_work, void *stream) {
    GSX_CHECK_ARG(offsets && gt && exposure && loss_rows && v_rec && C >= 1 && W > 0 && H > 0);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31) && (M == 0 || (rec && flatten_ids)));
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const int64_t T = C * tile_w * tile_h;
    GSX_CHECK_ARG(T < ((int64_t)1 << 31));
    TrackLossArgs la;
    la.gt = gt; la.exposure = exposure; la.w_photo = w_photo; la.v_render = v_render; la.rows = loss_rows;
    la.tile_work = tile_work; la.refiner_loss = 0;
    hipLaunchKernelGGL((raster_track_fused_kernel<12, false>), dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, rec,
                       backgrounds, offsets, const_cast<int32_t *>(flatten_ids), M, offsets_has_end, W, H, tile_w, tile_h, alphas,
                       last_ids, v_rec, tile_order, la, TileSortArgs{});
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

static int raster_track_fused_sorting_impl(const float *rec, const float *backgrounds, const int32_t *offsets,
                                           int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                                           const float *gt, const float *exposure, float w_photo, float *alphas,
                                           int32_t *last_ids, float *v_render, float *loss_rows, float *v_rec,
                                           const int32_t *tile_order, int32_t *tile_work, uint64_t *keys,
                                           uint64_t *keys_sorted, uint32_t id_max, uint32_t *tile_cut, float cut_margin,
                                           int32_t *tile_near, int32_t *sort_stats, const int32_t *tile_placed,
                                           const void *inst_recs, const int32_t *n_inst, int64_t R, int64_t seg_cap,
                                           int compact, void *stream) {
    GSX_CHECK_ARG(offsets && gt && exposure && loss_rows && v_rec && C >= 1 && W > 0 && H > 0);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31) && (M == 0 || (rec && flatten_ids && keys && keys_sorted)));
    GSX_CHECK_ARG(tile_cut && cut_margin >= 0.f && cut_margin < 16.f && offsets_has_end == 1);
    // near placement: the instance records the tile workgroups complete their segments from ([C][R][seg_cap], seg_cap = 1024 * 2^k)
    GSX_CHECK_ARG(!tile_placed || (inst_recs && n_inst && R >= 1 && R <= 640 && seg_cap >= 1024 && seg_cap <= 8192 &&
                                   (seg_cap & (seg_cap - 1)) == 0 && C * R * seg_cap < ((int64_t)1 << 31)));
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const int64_t T = C * tile_w * tile_h;
    GSX_CHECK_ARG(T < ((int64_t)1 << 31));
    TrackLossArgs la;
    la.gt = gt; la.exposure = exposure; la.w_photo = w_photo; la.v_render = v_render; la.rows = loss_rows;
    la.tile_work = tile_work; la.refiner_loss = 0;
    TileSortArgs ts;
    ts.keys = (unsigned long long *)keys; ts.sorted = (unsigned long long *)keys_sorted; ts.tile_cut = tile_cut;
    ts.tile_near = tile_near; ts.stats = sort_stats; ts.id_max = id_max; ts.margin = cut_margin;
    ts.tile_placed = tile_placed; ts.inst = (const uint4 *)inst_recs; ts.n_inst = n_inst; ts.R = (int)R;
    ts.seg_cap = (int)seg_cap; ts.compact = compact;
    ts.row_words = nullptr; ts.row_keys = nullptr; ts.row_cap = 0; ts.cursor = nullptr; ts.status = nullptr; ts.tile_span = nullptr;
    hipLaunchKernelGGL((raster_track_fused_kernel<12, true>), dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, rec,
                       backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, alphas, last_ids, v_rec,
                       tile_order, la, ts);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_raster_track_fused_sorting(const float *rec, const float *backgrounds, const int32_t *offsets,
                                              int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                                              const float *gt, const float *exposure, float w_photo, float *alphas,
                                              int32_t *last_ids, float *v_render, float *loss_rows, float *v_rec,
                                              const int32_t *tile_order, int32_t *tile_work, uint64_t *keys,
                                              uint64_t *keys_sorted, uint32_t id_max, uint32_t *tile_cut, float cut_margin,
                                              int32_t *tile_near, int32_t *sort_stats, void *stream) {
    return raster_track_fused_sorting_impl(rec, backgrounds, offsets, flatten_ids, M, offsets_has_end, C, W, H, gt, exposure,
                                           w_photo, alphas, last_ids, v_render, loss_rows, v_rec, tile_order, tile_work, keys,
                                           keys_sorted, id_max, tile_cut, cut_margin, tile_near, sort_stats, nullptr, nullptr,
                                           nullptr, 0, 0, 0, stream);
}

extern "C" int gsx_raster_track_fused_near(const float *rec, const float *backgrounds, const int32_t *offsets,
                                           int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                                           const float *gt, const float *exposure, float w_photo, float *alphas,
                                           int32_t *last_ids, float *v_render, float *loss_rows, float *v_rec,
                                           const int32_t *tile_order, int32_t *tile_work, uint64_t *keys,
                                           uint64_t *keys_sorted, uint32_t id_max, uint32_t *tile_cut, float cut_margin,
                                           int32_t *tile_near, int32_t *sort_stats, const int32_t *tile_placed,
                                           const void *inst_recs, const int32_t *n_inst, int64_t R, int64_t seg_cap,
                                           int compact, void *stream) {
    GSX_CHECK_ARG(tile_placed != nullptr);
    return raster_track_fused_sorting_impl(rec, backgrounds, offsets, flatten_ids, M, offsets_has_end, C, W, H, gt, exposure,
                                           w_photo, alphas, last_ids, v_render, loss_rows, v_rec, tile_order, tile_work, keys,
                                           keys_sorted, id_max, tile_cut, cut_margin, tile_near, sort_stats, tile_placed,
                                           inst_recs, n_inst, R, seg_cap, compact, stream);
}

// static LDS of the fused tracking rasteriser's workgroup, read off the code object: what the start-up probe of the CU-balanced
// launch order allocates per workgroup so that its workgroups are placed like the launches it vouches for (plan.placement_ok)
extern "C" int64_t gsx_raster_track_fused_lds_bytes(void) {
    hipFuncAttributes attr;
    if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&raster_track_fused_kernel<12, true>)) != hipSuccess) {
        (void)hipGetLastError();
        gsx_set_error("gsx_raster_track_fused_lds_bytes: hipFuncGetAttributes failed");
        return -1;
    }
    return (int64_t)attr.sharedSizeBytes;
}

// layouts of the front's workspace (isect_bin.hip)
extern "C" int gsx_front_rows_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4);
extern "C" int gsx_front_keys(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int flags, int64_t *out3);
extern "C" int64_t gsx_front_workspace_bytes(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap);

extern "C" int gsx_raster_track_fused_rows(const float *rec, const float *backgrounds, int32_t *flatten_ids, int64_t M_cap,
                                           int64_t N, int64_t C, int W, int H, const float *gt, const float *exposure,
                                           float w_photo, int loss_kind, float *alphas, int32_t *last_ids, float *v_render,
                                           float *loss_rows, float *v_rec, const int32_t *tile_order, int32_t *tile_work,
                                           uint32_t *tile_cut, float cut_margin, int32_t *tile_near, int32_t *sort_stats,
                                           int32_t *tile_span, int64_t *M_dev, int32_t *status, void *front_workspace,
                                           int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(rec && flatten_ids && gt && exposure && loss_rows && v_rec && M_dev && status && front_workspace);
    GSX_CHECK_ARG(loss_kind == 0 || loss_kind == 1);
    GSX_CHECK_ARG(C >= 1 && C <= 255 && W > 0 && H > 0 && N >= 1 && M_cap >= 1 && M_cap < ((int64_t)1 << 31));
    GSX_CHECK_ARG(tile_cut && cut_margin >= 0.f && cut_margin < 16.f);
    const int tile_w = (W + GSX_TILE - 1) / GSX_TILE, tile_h = (H + GSX_TILE - 1) / GSX_TILE;
    const int64_t T = C * tile_w * tile_h;
    GSX_CHECK_ARG(T < ((int64_t)1 << 31));
    if (workspace_bytes < gsx_front_workspace_bytes(N, C, tile_w, tile_h, M_cap)) {
        gsx_set_error("gsx_raster_track_fused_rows: not the workspace of the front (%lld bytes)", (long long)workspace_bytes);
        return GSX_E_WORKSPACE;
    }
    int64_t rl[4], kl[3];
    if (gsx_front_rows_layout(N, C, tile_w, tile_h, M_cap, rl) != GSX_OK ||
        gsx_front_keys(N, C, tile_w, tile_h, M_cap, GSX_PROJ_COMPACT, kl) != GSX_OK)
        return GSX_E_INVALID;
    GSX_CHECK_ARG(rl[3] <= 768);
    char *ws = (char *)front_workspace;
    TrackLossArgs la;
    la.gt = gt; la.exposure = exposure; la.w_photo = w_photo; la.v_render = v_render; la.rows = loss_rows;
    la.tile_work = tile_work; la.refiner_loss = loss_kind;
    TileSortArgs ts;
    ts.keys = (unsigned long long *)(ws + kl[0]); ts.sorted = (unsigned long long *)(ws + kl[1]); ts.tile_cut = tile_cut;
    ts.tile_near = tile_near; ts.stats = sort_stats; ts.id_max = (uint32_t)kl[2]; ts.margin = cut_margin;
    ts.tile_placed = nullptr; ts.inst = nullptr; ts.n_inst = nullptr; ts.R = (int)rl[3]; ts.seg_cap = 0; ts.compact = 1;
    ts.row_words = (const uint32_t *)(ws + rl[1]); ts.row_keys = (const unsigned long long *)(ws + rl[0]);
    ts.row_cap = (int)rl[2]; ts.cursor = (unsigned long long *)M_dev; ts.status = status; ts.tile_span = tile_span;
    hipLaunchKernelGGL((raster_track_fused_kernel<12, true>), dim3((unsigned)T), dim3(256), 0, (hipStream_t)stream, rec,
                       backgrounds, (const int32_t *)nullptr, flatten_ids, M_cap, 1, W, H, tile_w, tile_h, alphas, last_ids,
                       v_rec, tile_order, la, ts);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_raster_bwd(const float *rec, int CH, const float *backgrounds, const int32_t *offsets,
                              const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                              int tile_w, int tile_h, const float *alphas, const int32_t *last_ids,
                              const float *v_render, const float *v_alphas, float *v_rec, float *v_abs,
                              const int32_t *tile_order, int geometry_only, void *stream) {
    GSX_CHECK_ARG(offsets && alphas && last_ids && v_render && C >= 1 && W > 0 && H > 0);  // v_alphas: NULL = 0
    GSX_CHECK_ARG(tile_w == (W + GSX_TILE - 1) / GSX_TILE && tile_h == (H + GSX_TILE - 1) / GSX_TILE);
    GSX_CHECK_ARG(M >= 0 && M < ((int64_t)1 << 31));
    if (M == 0) return GSX_OK;
    GSX_CHECK_ARG(rec && flatten_ids && v_rec);
    const int64_t T = C * tile_w * tile_h;
    hipStream_t st = (hipStream_t)stream;
    // Gradient accumulation without LDS atomics everywhere (reduce-scatter + cross-row sums in registers, plain stores into
    // per-wavefront accumulator copies).  Selection by launch shape (measured on MI355X):
    //   absgrad wanted                 : the two-pixels-per-lane kernel with LDS row sums (the only one with the |v_xy| channel)
    //   geometry-only (frozen map)     : quadrant kernel, batches of 64 - all 1200 workgroups of a camera resident at once
    //                                    (25 KiB of LDS, 73 VGPRs; with batches of 128 15 % of the tiles wait for a second
    //                                    round: 86.9 -> 80.9 us at 500 k)
    //   full chip (>= 4096 tiles)      : two pixels per lane, two wavefronts per tile, batches of 128
    //   one camera                     : quadrant kernel; batches of 128 with deep tile lists, 64 otherwise
#ifndef GSX_BWD_FULLCHIP_TILES
#define GSX_BWD_FULLCHIP_TILES 4096
#endif
    const bool geom_only = geometry_only != 0 && !v_abs;
    const int64_t per_tile = M / (T > 0 ? T : 1);
#define LAUNCH(ch, rs)                                                                                               \
    do {                                                                                                             \
        if (v_abs)                                                                                                   \
            hipLaunchKernelGGL((raster_bwd_kernel2<ch, rs, true>), dim3((unsigned)T), dim3(128), 0, st, rec,         \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, alphas,  \
                               last_ids, v_render, v_alphas, v_rec, v_abs);                                          \
        else if (geom_only)                                                                                          \
            hipLaunchKernelGGL((raster_bwd_geom_kernel<ch, rs>), dim3((unsigned)T), dim3(256), 0, st, rec,           \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, alphas,  \
                               last_ids, v_render, v_alphas, v_rec, tile_order);                                     \
        else if (T >= GSX_BWD_FULLCHIP_TILES)                                                                        \
            hipLaunchKernelGGL((raster_bwd_kernel4<ch, rs, 128>), dim3((unsigned)T), dim3(128), 0, st, rec,          \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, alphas,  \
                               last_ids, v_render, v_alphas, v_rec, tile_order);                                     \
        else if (per_tile > 1000)                                                                                    \
            hipLaunchKernelGGL((raster_bwd_kernel4q<ch, rs, 128, false>), dim3((unsigned)T), dim3(256), 0, st, rec,  \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, alphas,  \
                               last_ids, v_render, v_alphas, v_rec, tile_order);                                     \
        else                                                                                                         \
            hipLaunchKernelGGL((raster_bwd_kernel4q<ch, rs, 64, false>), dim3((unsigned)T), dim3(256), 0, st, rec,   \
                               backgrounds, offsets, flatten_ids, M, offsets_has_end, W, H, tile_w, tile_h, alphas,  \
                               last_ids, v_render, v_alphas, v_rec, tile_order);                                     \
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

#include "tile_balance.h"
namespace {
__global__ __launch_bounds__(gsx_bal::THREADS) void tile_balance_kernel(gsx_bal::Args a) {
    __shared__ __attribute__((aligned(16))) unsigned char s_bal[gsx_bal::LDS_BYTES];
    gsx_bal::run(a, s_bal);
}
}  // namespace

extern "C" int gsx_tile_balance(const int32_t *tile_work, int64_t T, float chunk_cost, float light_rate, int n_cus,
                                int32_t *tile_order, void *stream) {
    GSX_CHECK_ARG(tile_work && tile_order && T >= 1 && T <= gsx_bal::MAX_TILES && n_cus >= 1 && n_cus <= gsx_bal::MAX_BINS);
    GSX_CHECK_ARG(chunk_cost >= 0.f && chunk_cost < 1e6f && light_rate > 0.f && light_rate <= 1.f);
    gsx_bal::Args a;
    a.work = tile_work; a.order = tile_order; a.T = (int)T; a.G = n_cus; a.chunk_cost = chunk_cost; a.light_rate = light_rate;
    hipLaunchKernelGGL(tile_balance_kernel, dim3(1), dim3(gsx_bal::THREADS), 0, (hipStream_t)stream, a);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

#ifdef GSX_WG_TRACE
extern "C" int gsx_debug_phase_trace(void *buffer) {           // diagnostic build only; not part of include/gsx.h
    return hipMemcpyToSymbol(HIP_SYMBOL(g_phase_trace), &buffer, sizeof(buffer)) == hipSuccess ? 0 : 1;
}
extern "C" int gsx_debug_wg_trace(int which, void *buffer) {   // diagnostic build only; not part of include/gsx.h
    return hipMemcpyToSymbol(HIP_SYMBOL(g_wg_trace), &buffer, sizeof(buffer), (size_t)(which & 1) * sizeof(buffer)) ==
                   hipSuccess ? 0 : 1;
}
#endif
