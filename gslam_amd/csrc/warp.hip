// warp.hip — depth-based SE(3) image warp with pose Jacobian.  Replaces the ~20 torch kernels of
// gslam/warp.py:35-82 (backproject -> T -> project -> grid_sample -> in-bounds mask) with one streaming kernel per
// direction.  T = f1_pose @ inv(f2_pose) is formed by the caller (4x4, tiny) so both pose gradients flow through
// torch autograd from the 3x4 v_T this file produces.
//
// Quirks preserved on purpose: the pixel grid is [u, v, 1] with integer u,v (no +0.5), the
// back-projected point gets +1e-10 per component, normalisation is u*2/W-1 sampled with align_corners=False (half
// pixel shift), and the mask uses strict inequalities.
//
// One thread per pixel; 37 B/px forward (depth 4 + 4x12 gathered colour, mostly L2 hits; 12+8+1 written).  The
// backward reduces the 12 entries of v_T wave64 (DPP) -> LDS -> one partial row per workgroup -> finishing kernel.
#include "gsx_common.h"

namespace {

struct WarpConsts {
    float T[12];
    float K[9];
    float Kinv[9];
};

__device__ __forceinline__ void load_consts(const float *__restrict__ T, const float *__restrict__ K,
                                            const float *__restrict__ Kinv, WarpConsts &c) {
#pragma unroll
    for (int i = 0; i < 12; ++i) c.T[i] = T[i];
#pragma unroll
    for (int i = 0; i < 9; ++i) { c.K[i] = K[i]; c.Kinv[i] = Kinv[i]; }
}

struct WarpPt {
    float X[3], p[3], nw0, nw1, ix, iy;
};

__device__ __forceinline__ void warp_point(const WarpConsts &c, int W, int H, int u, int v, float d, WarpPt &o) {
    const float fu = (float)u, fv = (float)v;
#pragma unroll
    for (int i = 0; i < 3; ++i) o.X[i] = d * (c.Kinv[i * 3 + 0] * fu + c.Kinv[i * 3 + 1] * fv + c.Kinv[i * 3 + 2]) + 1e-10f;
    float Xn[3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        Xn[i] = (c.T[i * 4 + 0] * o.X[0] + c.T[i * 4 + 1] * o.X[1] + c.T[i * 4 + 2] * o.X[2]) + c.T[i * 4 + 3];
#pragma unroll
    for (int i = 0; i < 3; ++i) o.p[i] = Xn[0] * c.K[i * 3 + 0] + Xn[1] * c.K[i * 3 + 1] + Xn[2] * c.K[i * 3 + 2];
    o.nw0 = (o.p[0] / o.p[2]) * (2.0f / (float)W) - 1.0f;
    o.nw1 = (o.p[1] / o.p[2]) * (2.0f / (float)H) - 1.0f;
    o.ix = ((o.nw0 + 1.0f) * (float)W - 1.0f) * 0.5f;
    o.iy = ((o.nw1 + 1.0f) * (float)H - 1.0f) * 0.5f;
}

__global__ __launch_bounds__(256) void warp_fwd_kernel(const float *__restrict__ T, const float *__restrict__ K,
                                                       const float *__restrict__ Kinv, const float *__restrict__ c1,
                                                       const float *__restrict__ d1, int H, int W,
                                                       float *__restrict__ result, float *__restrict__ nwarps,
                                                       uint8_t *__restrict__ keep) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= (int64_t)H * W) return;
    WarpConsts c;
    load_consts(T, K, Kinv, c);
    const int v = (int)(o / W), u = (int)(o - (int64_t)v * W);
    WarpPt w;
    warp_point(c, W, H, u, v, d1[o], w);
    nwarps[2 * o] = w.nw0;
    nwarps[2 * o + 1] = w.nw1;
    keep[o] = (w.nw0 < 1.0f) && (w.nw1 < 1.0f) && (w.nw0 > -1.0f) && (w.nw1 > -1.0f);
    const float x0f = floorf(w.ix), y0f = floorf(w.iy);
    const float wx1 = w.ix - x0f, wy1 = w.iy - y0f, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
    float acc[3] = {0.f, 0.f, 0.f};
    if (x0f > -2.0f && x0f < (float)(W + 1) && y0f > -2.0f && y0f < (float)(H + 1)) {
        const int x0 = (int)x0f, y0 = (int)y0f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const int xx = x0 + dx, yy = y0 + dy;
                if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                const float wgt = (dx ? wx1 : wx0) * (dy ? wy1 : wy0);
                const float *px = c1 + ((int64_t)yy * W + xx) * 3;
                acc[0] += wgt * px[0]; acc[1] += wgt * px[1]; acc[2] += wgt * px[2];
            }
    }
    result[3 * o] = acc[0]; result[3 * o + 1] = acc[1]; result[3 * o + 2] = acc[2];
}

__global__ __launch_bounds__(256) void warp_bwd_kernel(const float *__restrict__ T, const float *__restrict__ K,
                                                       const float *__restrict__ Kinv, const float *__restrict__ c1,
                                                       const float *__restrict__ d1, int H, int W,
                                                       const float *__restrict__ v_result,
                                                       const float *__restrict__ v_nwarps,
                                                       float *__restrict__ partials) {
    __shared__ float s_part[4][12];
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool active = o < (int64_t)H * W;
    float g[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) g[i] = 0.f;
    if (active) {
        WarpConsts c;
        load_consts(T, K, Kinv, c);
        const int v = (int)(o / W), u = (int)(o - (int64_t)v * W);
        WarpPt w;
        warp_point(c, W, H, u, v, d1[o], w);
        const float x0f = floorf(w.ix), y0f = floorf(w.iy);
        const float wx1 = w.ix - x0f, wy1 = w.iy - y0f, wx0 = 1.0f - wx1, wy0 = 1.0f - wy1;
        float g_ix = 0.f, g_iy = 0.f;
        if (x0f > -2.0f && x0f < (float)(W + 1) && y0f > -2.0f && y0f < (float)(H + 1)) {
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float vr0 = v_result[3 * o], vr1 = v_result[3 * o + 1], vr2 = v_result[3 * o + 2];
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const int xx = x0 + dx, yy = y0 + dy;
                    if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                    const float *px = c1 + ((int64_t)yy * W + xx) * 3;
                    const float dotv = px[0] * vr0 + px[1] * vr1 + px[2] * vr2;
                    g_ix += (dx ? 1.0f : -1.0f) * (dy ? wy1 : wy0) * dotv;
                    g_iy += (dy ? 1.0f : -1.0f) * (dx ? wx1 : wx0) * dotv;
                }
        }
        float g_nw0 = g_ix * 0.5f * (float)W, g_nw1 = g_iy * 0.5f * (float)H;
        if (v_nwarps) { g_nw0 += v_nwarps[2 * o]; g_nw1 += v_nwarps[2 * o + 1]; }
        const float s0 = 2.0f / (float)W, s1 = 2.0f / (float)H;
        float gp[3];
        gp[0] = g_nw0 * s0 / w.p[2];
        gp[1] = g_nw1 * s1 / w.p[2];
        gp[2] = -(g_nw0 * s0 * w.p[0] + g_nw1 * s1 * w.p[1]) / (w.p[2] * w.p[2]);
        float gX[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) gX[j] = gp[0] * c.K[j] + gp[1] * c.K[3 + j] + gp[2] * c.K[6 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) g[i * 4 + j] = gX[i] * w.X[j];
            g[i * 4 + 3] = gX[i];
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const float tot = gsx_wave_sum(g[k]);
        if (lane == 0) s_part[wave][k] = tot;
    }
    __syncthreads();
    if (threadIdx.x < 12)
        partials[(int64_t)blockIdx.x * 12 + threadIdx.x] =
            (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void warp_bwd_finish_kernel(const float *__restrict__ partials, int n_blocks,
                                                              float *__restrict__ v_T) {
    // 12 columns x up to 16 row-strided accumulators, then a serial fold by thread k
    __shared__ float s_acc[16][12];
    const int k = threadIdx.x % 12, r = threadIdx.x / 12;
    if (r < 16) {
        float acc = 0.f;
        for (int b = r; b < n_blocks; b += 16) acc += partials[(int64_t)b * 12 + k];
        s_acc[r][k] = acc;
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float acc = 0.f;
        if (threadIdx.x < 12)
            for (int rr = 0; rr < 16; ++rr) acc += s_acc[rr][threadIdx.x];
        v_T[threadIdx.x] = acc;  // row 3 of the 4x4 stays zero
    }
}

}  // namespace

extern "C" int gsx_warp_fwd(const float *T, const float *K, const float *Kinv, const float *c1, const float *d1,
                            int H, int W, float *result, float *nwarps, uint8_t *keep_mask, void *stream) {
    GSX_CHECK_ARG(T && K && Kinv && c1 && d1 && result && nwarps && keep_mask && H > 0 && W > 0);
    const int64_t n = (int64_t)H * W;
    hipLaunchKernelGGL(warp_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, T, K,
                       Kinv, c1, d1, H, W, result, nwarps, keep_mask);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int64_t gsx_warp_bwd_workspace_bytes(int H, int W) {
    const int64_t blocks = ((int64_t)H * W + 255) / 256;
    return gsx_align256(blocks * 12 * (int64_t)sizeof(float)) + 256;
}

extern "C" int gsx_warp_bwd(const float *T, const float *K, const float *Kinv, const float *c1, const float *d1,
                            int H, int W, const float *v_result, const float *v_nwarps, float *v_T, void *workspace,
                            int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(T && K && Kinv && c1 && d1 && v_result && v_T && H > 0 && W > 0);
    if (!workspace || workspace_bytes < gsx_warp_bwd_workspace_bytes(H, W)) {
        gsx_set_error("gsx_warp_bwd: workspace too small");
        return GSX_E_WORKSPACE;
    }
    const int64_t n = (int64_t)H * W;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(warp_bwd_kernel, dim3(blocks), dim3(256), 0, st, T, K, Kinv, c1, d1, H, W, v_result, v_nwarps,
                       (float *)workspace);
    GSX_CHECK_LAUNCH();
    hipLaunchKernelGGL(warp_bwd_finish_kernel, dim3(1), dim3(256), 0, st, (const float *)workspace, (int)blocks, v_T);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
