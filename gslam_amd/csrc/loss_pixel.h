// loss_pixel.h - one pixel of the mapping / tracking loss block (gslam/backend.py:273-318, gslam/frontend.py:113-138) with its
// analytic gradient: shared by map_loss_kernel (loss.hip) and by the SSIM backward that finishes the loss block in the same pass
// (ssim.hip: ssim_bwd_loss_kernel), so that both produce the same bits for d loss / d render.
#pragma once
#include "gsx_common.h"

namespace gsx_loss {

constexpr int LB = 256;
constexpr int NPART = 6;  // per-workgroup partials: S-term, log-beta term, tv, v_a, v_b, (spare)

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

struct LossArgs {
    const float *render;   // [C,H,W,CH]
    const float *alphas;   // [C,H,W]
    const float *gt;       // [C,H,W,3]
    const float *exposure; // [C,2] (a, b): rendered = rgb * exp(a) + b
    const float *ssim_grad; // [C,3,H,W] planar, already scaled; nullable
    float *v_render;       // [C,H,W,CH]
    float *partials;       // [C][blocks_per_cam][NPART]
    int H, W, CH, depth_index, beta_index;
    int mode;              // 0: active-gs mapping (backend.py:277-283), 1: plain mse on un-exposed rgb (backend.py:285),
                           // 2: active-nerf tracking (frontend.py:127: err^2 * beta^-2, no log term)
    float w_photo;         // weight / (C*H*W) (mode 1: / (C*H*W*3))
    float w_tv;            // weight of the edge-aware depth TV SUM (0 disables)
    float mask_thresh;     // alphas > thresh (backend.py:301: 0.4)
};

// pixel (x, y) of camera c: r* = the render's colours, g* = the target's; ssim_g (has_ssim) = the SSIM gradient of the three colours,
// already scaled, added to the colour gradient last.  Writes v_render[p][0..CH) and adds the pixel's terms to part[NPART].
__device__ __forceinline__ void map_loss_pixel(const LossArgs &A, int c, int x, int y, float r0, float r1, float r2, float g0,
                                               float g1, float g2, bool has_ssim, const float *ssim_g, float *part) {
    const int HW = A.H * A.W;
    const int i = y * A.W + x;
    const int64_t p = (int64_t)c * HW + i;
    const int CH = A.CH;
    const float *rp = A.render + p * CH;
    const float ea = __expf(A.exposure[2 * c]), eb = A.exposure[2 * c + 1];
    float v0 = 0.f, v1 = 0.f, v2 = 0.f, vdepth = 0.f, vbeta = 0.f;
    if (A.mode == 1) {
        const float d0 = r0 - g0, d1 = r1 - g1, d2 = r2 - g2;
        part[0] = d0 * d0 + d1 * d1 + d2 * d2;
        v0 = 2.f * A.w_photo * d0; v1 = 2.f * A.w_photo * d1; v2 = 2.f * A.w_photo * d2;
    } else {
        const float d0 = r0 * ea + eb - g0, d1 = r1 * ea + eb - g1, d2 = r2 * ea + eb - g2;
        const float S = d0 * d0 + d1 * d1 + d2 * d2;
        const float beta = rp[A.beta_index];
        const float ib = 1.0f / beta, ib2 = ib * ib;
        if (A.mode == 0) {
            const float lb = __logf(beta);
            part[0] = 0.5f * S * ib2;
            part[1] = 0.5f * lb * lb;
            const float w = A.w_photo;
            v0 = w * d0 * ea * ib2; v1 = w * d1 * ea * ib2; v2 = w * d2 * ea * ib2;
            vbeta = w * (-S * ib2 * ib + lb * ib);
            part[3] = w * (d0 * r0 + d1 * r1 + d2 * r2) * ea * ib2;
            part[4] = w * (d0 + d1 + d2) * ib2;
        } else {
            part[0] = S * ib2;
            const float w = 2.f * A.w_photo;
            v0 = w * d0 * ea * ib2; v1 = w * d1 * ea * ib2; v2 = w * d2 * ea * ib2;
            vbeta = -w * S * ib2 * ib;
            part[3] = w * (d0 * r0 + d1 * r1 + d2 * r2) * ea * ib2;
            part[4] = w * (d0 + d1 + d2) * ib2;
        }
    }
    if (A.w_tv != 0.f && A.depth_index >= 0) {
        // edge_aware_tv: pairs (p,right) and (p,down) are owned by p and masked by p's alpha; gather the four
        // incident pairs so that every gradient entry is written exactly once.
        const float dp = rp[A.depth_index];
        const bool mp = A.alphas[p] > A.mask_thresh;
        float tv = 0.f;
        auto pair = [&](int64_t q_, bool owner_is_p, bool mask_owner) {
            // returns contribution to (v_depth[p], v_rgb[p]) from the pair {p, q_}; owner = left/upper pixel
            const float *rq = A.render + q_ * CH;
            const float dq = rq[A.depth_index];
            const float a0 = r0 - rq[0], a1 = r1 - rq[1], a2 = r2 - rq[2];   // p - q
            const float gi = (fabsf(a0) + fabsf(a1) + fabsf(a2)) * (1.0f / 3.0f);
            const float e = __expf(-gi);
            const float gd = fabsf(dp - dq);
            if (!mask_owner) return;
            if (owner_is_p) tv += gd * e;
            // d/d depth_p |dp - dq| = sgn(dp - dq) regardless of who owns the pair
            vdepth += A.w_tv * sgn(dp - dq) * e;
            const float k = -A.w_tv * gd * e * (1.0f / 3.0f);
            v0 += k * sgn(a0); v1 += k * sgn(a1); v2 += k * sgn(a2);
        };
        if (x + 1 < A.W) pair(p + 1, true, mp);
        if (x > 0) pair(p - 1, false, A.alphas[p - 1] > A.mask_thresh);
        if (y + 1 < A.H) pair(p + A.W, true, mp);
        if (y > 0) pair(p - A.W, false, A.alphas[p - A.W] > A.mask_thresh);
        part[2] = tv;
    }
    if (has_ssim) { v0 += ssim_g[0]; v1 += ssim_g[1]; v2 += ssim_g[2]; }
    float *vp = A.v_render + p * CH;
    vp[0] = v0; vp[1] = v1; vp[2] = v2;
    for (int k = 3; k < CH; ++k) vp[k] = (k == A.depth_index) ? vdepth : ((k == A.beta_index) ? vbeta : 0.f);
}

}  // namespace gsx_loss
