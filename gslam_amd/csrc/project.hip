// project.hip — K1/K2: EWA projection of 3D Gaussians, forward and backward, plus the fused gslam front-end
// (activations + splat-record packing) and K10.  Replaces gsplat `fully_fused_projection`
// (gslam/rasterization.py:153-170,390-407) and `quat_scale_to_covar_preci` (gslam/insertion.py:88-91).
//
// Compiled with -ffp-contract=off: radii / tile rectangles are integer outputs that must be reproducible
// bit-for-bit against the CPU oracle, so every product-sum below is written in a fixed order and never fused.
//
// Layout / mapping (MI355X): one thread per Gaussian g, looping over the C cameras of the window.  The world
// covariance is built once per Gaussian and reused by every camera; camera constants are wave-uniform (scalar
// loads).  Per-Gaussian attribute reads are contiguous across the wave (12/16-byte rows), every output row
// [c, g] is written exactly once.  The kernel is a pure HBM stream: 40 B read per Gaussian + 28 B (+ record)
// written per (camera, Gaussian).
#include "gsx_common.h"
#include "project_core.h"
#include "tile_rect.h"

namespace {

using namespace gsx_proj;

__device__ __forceinline__ uint32_t sat_u32(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}

__device__ __forceinline__ int32_t tile_count(float mx, float my, int32_t radius, int tile_w, int tile_h) {
    const float ts = (float)GSX_TILE;
    const float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    const uint32_t x0 = min(sat_u32(floorf(tx - tr)), (uint32_t)tile_w);
    const uint32_t y0 = min(sat_u32(floorf(ty - tr)), (uint32_t)tile_h);
    const uint32_t x1 = min(sat_u32(ceilf(tx + tr)), (uint32_t)tile_w);
    const uint32_t y1 = min(sat_u32(ceilf(ty + tr)), (uint32_t)tile_h);
    return (int32_t)((y1 - y0) * (x1 - x0));
}

// ------------------------------------------------------------------------------------------------------------------
// K1 forward
// ------------------------------------------------------------------------------------------------------------------
template <int RS>  // record stride in floats (0 = no record)
__global__ __launch_bounds__(256) void project_fwd_kernel(
    const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks, int64_t N, int C, int W, int H, float eps2d,
    float near_p, float far_p, float radius_clip, int flags, int32_t *__restrict__ radii,
    float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics, float *__restrict__ comps,
    int32_t *__restrict__ tiles_per_gauss, int tile_w, int tile_h, const float *__restrict__ logit_opacities,
    const float *__restrict__ logit_colors, const float *__restrict__ log_unc, float *__restrict__ rec,
    int32_t *__restrict__ vis_count, float *__restrict__ v_rec_clear,
    uint32_t *__restrict__ rects /* nullable (RS > 0): packed tile rectangle per (camera, Gaussian) for gsx_isect_bin_sort_rects */) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= N) return;
    int n_vis = 0;
    const float mean[3] = {means[3 * g], means[3 * g + 1], means[3 * g + 2]};
    const float q[4] = {quats[4 * g], quats[4 * g + 1], quats[4 * g + 2], quats[4 * g + 3]};
    float s[3] = {scales[3 * g], scales[3 * g + 1], scales[3 * g + 2]};
    if (flags & GSX_PROJ_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
    QuatRot qr;
    quat_to_rotmat(q, qr);
    float M[9];
    Sym3 S;
    covar_from_rot_scale(qr.R, s, M, S);

    float opac = 0.f, col[3] = {0.f, 0.f, 0.f}, beta = 0.f;
    if (RS > 0) {
        opac = gsx_sigmoid(logit_opacities[g]);
        col[0] = gsx_sigmoid(logit_colors[3 * g]);
        col[1] = gsx_sigmoid(logit_colors[3 * g + 1]);
        col[2] = gsx_sigmoid(logit_colors[3 * g + 2]);
        if (flags & GSX_PROJ_BETAS) beta = fmaxf(expf(log_unc[g]), GSX_BETA_MIN);
    }

    for (int c = 0; c < C; ++c) {
        Cam cam;
        load_cam(viewmats, Ks, c, cam);
        const int64_t idx = (int64_t)c * N + g;
        Proj p;
        int32_t radius_i = 0;
        float mx = 0.f, my = 0.f, depth = 0.f, con0 = 0.f, con1 = 0.f, con2 = 0.f, comp = 0.f;
        if (project_core(mean, S, cam, W, H, eps2d, near_p, far_p, p)) {
            const float pmx = (cam.fx * p.pc[0]) * p.rz + cam.cx, pmy = (cam.fy * p.pc[1]) * p.rz + cam.cy;
            const float b00 = p.c00 + eps2d, b11 = p.c11 + eps2d;
            const float b = 0.5f * (b00 + b11);
            const float v1 = b + sqrtf(fmaxf(GSX_RADIUS_FLOOR, b * b - p.det));
            const float radius = ceilf(GSX_RADIUS_SIGMA * sqrtf(v1));
            const bool keep = !(radius <= radius_clip) &&
                              !(pmx + radius <= 0.0f || pmx - radius >= (float)W || pmy + radius <= 0.0f ||
                                pmy - radius >= (float)H);
            if (keep) {
                radius_i = (int32_t)radius;
                mx = pmx; my = pmy; depth = p.pc[2];
                con0 = p.conic[0]; con1 = p.conic[1]; con2 = p.conic[2];
                comp = sqrtf(fmaxf(0.0f, p.det_orig / p.det));
            }
        }
        radii[idx] = radius_i;
        n_vis += radius_i > 0 ? 1 : 0;
        if (RS > 0 && rects && radius_i == 0 && (flags & GSX_PROJ_SKIP_CULLED)) {
            // nobody reads the other columns of a culled row (the binning reads the packed rectangle, the projection backward the
            // radius): 60 bytes of zeros per culled (camera, Gaussian) not written - three quarters of the rows of a 2 M x 8 window
            rects[idx] = 0u;
            if (tiles_per_gauss) tiles_per_gauss[idx] = 0;
            if (v_rec_clear) {
                float4 *z = reinterpret_cast<float4 *>(v_rec_clear + idx * RS);
                z[0] = z[1] = z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        means2d[2 * idx] = mx; means2d[2 * idx + 1] = my;
        depths[idx] = depth;
        if (conics) { conics[3 * idx] = con0; conics[3 * idx + 1] = con1; conics[3 * idx + 2] = con2; }
        if (comps) comps[idx] = comp;
        if (tiles_per_gauss && !(RS > 0 && rects))
            tiles_per_gauss[idx] = radius_i > 0 ? tile_count(mx, my, radius_i, tile_w, tile_h) : 0;
        if (RS > 0) {
            float *r = rec + idx * RS;
            float ch[6] = {col[0], col[1], col[2], 0.f, 0.f, 0.f};
            int n = 3;
            if (flags & GSX_PROJ_RENDER_DEPTH) ch[n++] = depth;
            if (flags & GSX_PROJ_BETAS) ch[n++] = beta;
            const bool vis = radius_i > 0;
            // culled rows carry zeros except the opacity column, which mirrors `opacities.repeat(C,1)`
            // (rasterization.py:187); they are never gathered: they own no tile intersections
            float4 a = make_float4(mx, my, con0, con1);
            float4 b4 = make_float4(con2, opac, vis ? ch[0] : 0.f, vis ? ch[1] : 0.f);
            float4 c4 = make_float4(vis ? ch[2] : 0.f, vis ? ch[3] : 0.f, vis ? ch[4] : 0.f, 0.f);
            reinterpret_cast<float4 *>(r)[0] = a;
            reinterpret_cast<float4 *>(r)[1] = b4;
            reinterpret_cast<float4 *>(r)[2] = c4;
            if (v_rec_clear) {                               // the backward's accumulation rows, cleared while we are here
                float4 *z = reinterpret_cast<float4 *>(v_rec_clear + idx * RS);
                z[0] = z[1] = z[2] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (rects) {
                gsx_rect::Rect tr = {0, 0, 0, 0};
                if (vis) {
                    tr = gsx_rect::tile_rect(mx, my, radius_i, tile_w, tile_h);
                    if (flags & GSX_PROJ_TILE_EXACT) tr = gsx_rect::tighten_rect(tr, mx, my, con0, con1, con2, opac);
                }
                rects[idx] = gsx_rect::pack_rect(tr);
                // (with packed rectangles the count is that of the rectangle the binning will walk: what a capacity probe sums)
                if (tiles_per_gauss)
                    tiles_per_gauss[idx] = (tr.x1 > tr.x0 && tr.y1 > tr.y0) ? (tr.y1 - tr.y0) * (tr.x1 - tr.x0) : 0;
            }
        }
    }
    if (vis_count) vis_count[g] = n_vis;
}

// ------------------------------------------------------------------------------------------------------------------
// K2 backward.  One thread per Gaussian; the camera loop accumulates v_mean and v_Sigma(world) in registers, the
// covariance -> (quat, scale) chain runs once.  Pose gradients: wave64 DPP reduce -> LDS -> one partial row per
// block in the workspace, then a finishing kernel (deterministic, no atomics).
// ------------------------------------------------------------------------------------------------------------------
#define PBWD_THREADS 256
#define PBWD_MAX_C 16

// POSE_ONLY: only the pose-gradient partials are wanted (tracking: the map is frozen, gslam/frontend.py:613-658), so the
// world-covariance accumulation, the quaternion / scale chain and 60 B of gradient stores per Gaussian are skipped.
template <bool POSE_ONLY>
__global__ __launch_bounds__(PBWD_THREADS) void project_bwd_kernel(
    const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks, int64_t N, int C, int W, int H, float eps2d,
    float near_p, float far_p, int flags, const int32_t *__restrict__ radii, const float *__restrict__ v_means2d,
    int64_t m2d_stride, const float *__restrict__ v_depths, const float *__restrict__ v_conics, int64_t con_stride,
    const float *__restrict__ v_comps, const float *__restrict__ logit_opacities,
    const float *__restrict__ logit_colors, const float *__restrict__ log_unc, const float *__restrict__ v_rec,
    int RS, float *__restrict__ v_means, float *__restrict__ v_quats, float *__restrict__ v_scales,
    float *__restrict__ view_partials /*[blocks of the whole map][C][12] or null*/, float *__restrict__ v_logit_opac,
    float *__restrict__ v_logit_colors, float *__restrict__ v_log_unc, int block0 /*first block of this launch's range*/) {
    __shared__ float s_part[PBWD_THREADS / GSX_WAVE][12];
    const int block = block0 + (int)blockIdx.x;
    const int64_t g = (int64_t)block * blockDim.x + threadIdx.x;
    const bool active = g < N;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    float mean[3] = {0.f, 0.f, 0.f}, q[4] = {1.f, 0.f, 0.f, 0.f}, s[3] = {1.f, 1.f, 1.f};
    if (active) {
        mean[0] = means[3 * g]; mean[1] = means[3 * g + 1]; mean[2] = means[3 * g + 2];
        q[0] = quats[4 * g]; q[1] = quats[4 * g + 1]; q[2] = quats[4 * g + 2]; q[3] = quats[4 * g + 3];
        s[0] = scales[3 * g]; s[1] = scales[3 * g + 1]; s[2] = scales[3 * g + 2];
        if (flags & GSX_PROJ_LOG_SCALES) { s[0] = expf(s[0]); s[1] = expf(s[1]); s[2] = expf(s[2]); }
    }
    QuatRot qr;
    quat_to_rotmat(q, qr);
    float M[9];
    Sym3 S;
    covar_from_rot_scale(qr.R, s, M, S);

    float vmean[3] = {0.f, 0.f, 0.f};
    float vS[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // matrix-form gradient of the world covariance
    float v_opac_sum = 0.f, v_col_sum[3] = {0.f, 0.f, 0.f}, v_beta_sum = 0.f;

    for (int c = 0; c < C; ++c) {
        Cam cam;
        load_cam(viewmats, Ks, c, cam);
        const int64_t idx = (int64_t)c * N + g;
        float vR[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float vpc[3] = {0.f, 0.f, 0.f};
        Proj p;
        if (active && radii[idx] > 0 && project_core(mean, S, cam, W, H, eps2d, near_p, far_p, p)) {
            const float fx = cam.fx, fy = cam.fy;
            float vdepth = v_depths ? v_depths[idx] : 0.f;
            // the usual case: v_means2d / v_conics are the xy / conic columns of the same gradient record: the 48-byte
            // row comes in as three 16-byte loads instead of twelve scalar ones
            float row[12];
            const bool fused_row = v_rec && RS == 12 && v_means2d == v_rec && m2d_stride == 12 && v_conics == v_rec + 2 &&
                                   con_stride == 12;
            if (fused_row) {
                const float4 *r4 = reinterpret_cast<const float4 *>(v_rec + idx * 12);
                const float4 q0 = r4[0], q1 = r4[1], q2 = r4[2];
                if (flags & GSX_PROJ_RESET_V_REC) {             // consumed: the row goes back to zero for the next backward
                    // (a visible pair the rasteriser never reached - every pixel under it saturated before - is zero already)
                    const unsigned any = (__float_as_uint(q0.x) | __float_as_uint(q0.y) | __float_as_uint(q0.z) | __float_as_uint(q0.w)) |
                                         (__float_as_uint(q1.x) | __float_as_uint(q1.y) | __float_as_uint(q1.z) | __float_as_uint(q1.w)) |
                                         (__float_as_uint(q2.x) | __float_as_uint(q2.y) | __float_as_uint(q2.z) | __float_as_uint(q2.w));
                    if (any << 1) {
                        float4 *w4 = const_cast<float4 *>(r4);
                        w4[0] = w4[1] = w4[2] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
                row[0] = q0.x; row[1] = q0.y; row[2] = q0.z; row[3] = q0.w; row[4] = q1.x; row[5] = q1.y; row[6] = q1.z;
                row[7] = q1.w; row[8] = q2.x; row[9] = q2.y; row[10] = q2.z; row[11] = q2.w;
            } else {
                row[0] = v_means2d[idx * m2d_stride]; row[1] = v_means2d[idx * m2d_stride + 1];
                row[2] = v_conics[idx * con_stride]; row[3] = v_conics[idx * con_stride + 1];
                row[4] = v_conics[idx * con_stride + 2];
#pragma unroll
                for (int k = 5; k < 12; ++k) row[k] = (v_rec && k < RS) ? v_rec[idx * RS + k] : 0.f;
            }
            if (v_rec) {
                if (!POSE_ONLY) {
                    v_opac_sum += row[5];
                    v_col_sum[0] += row[6]; v_col_sum[1] += row[7]; v_col_sum[2] += row[8];
                }
                int n = 9;
                if (flags & GSX_PROJ_RENDER_DEPTH) vdepth += row[n++];
                if (!POSE_ONLY && (flags & GSX_PROJ_BETAS)) v_beta_sum += row[n++];
            }
            // 1. conic = inverse(blurred cov2d): GX = -Y G Y
            const float a = p.conic[0], b = p.conic[1], cc = p.conic[2];
            const float va = row[2], vb = 0.5f * row[3], vc = row[4];
            const float P00 = va * a + vb * b, P01 = va * b + vb * cc;
            const float P10 = vb * a + vc * b, P11 = vb * b + vc * cc;
            float G00 = -(a * P00 + b * P10), G01 = -(a * P01 + b * P11), G11 = -(b * P01 + cc * P11);
            // 2. compensation
            if (v_comps) {
                const float comp = sqrtf(fmaxf(0.0f, p.det_orig / p.det));
                if (comp > 0.0f) {
                    const float vr_ = v_comps[idx] * 0.5f / comp;
                    const float b00 = p.c00 + eps2d, b11 = p.c11 + eps2d;
                    const float inv_d2 = 1.0f / (p.det * p.det);
                    G00 += vr_ * (p.c11 * p.det - p.det_orig * b11) * inv_d2;
                    G11 += vr_ * (p.c00 * p.det - p.det_orig * b00) * inv_d2;
                    G01 += vr_ * 0.5f * (-2.0f * p.c01 * p.det + p.det_orig * 2.0f * p.c01) * inv_d2;
                }
            }
            // 3. cov2d = J Sc J^T
            const float Jm[6] = {p.J00, 0.f, p.J02, 0.f, p.J11, p.J12};
            const float Gm[4] = {G00, G01, G01, G11};
            float GJ[6];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) GJ[i * 3 + j] = Gm[i * 2 + 0] * Jm[j] + Gm[i * 2 + 1] * Jm[3 + j];
            float vSc[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) vSc[i * 3 + j] = Jm[i] * GJ[j] + Jm[3 + i] * GJ[3 + j];
            float vJ[6];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    vJ[i * 3 + j] = 2.0f * (GJ[i * 3 + 0] * symget(p.Sc, 0, j) + GJ[i * 3 + 1] * symget(p.Sc, 1, j) +
                                            GJ[i * 3 + 2] * symget(p.Sc, 2, j));
            const float x = p.pc[0], y = p.pc[1], rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
            const float vmx = row[0], vmy = row[1];
            vpc[0] = fx * rz * vmx;
            vpc[1] = fy * rz * vmy;
            vpc[2] = -(fx * x * vmx + fy * y * vmy) * rz2 + vdepth;
            const float vJ00 = vJ[0], vJ02 = vJ[2], vJ11 = vJ[4], vJ12 = vJ[5];
            if (p.x_in) vpc[0] += -fx * rz2 * vJ02; else vpc[2] += -fx * rz3 * vJ02 * p.tx;
            if (p.y_in) vpc[1] += -fy * rz2 * vJ12; else vpc[2] += -fy * rz3 * vJ12 * p.ty;
            vpc[2] += -fx * rz2 * vJ00 - fy * rz2 * vJ11 + 2.0f * fx * p.tx * rz3 * vJ02 + 2.0f * fy * p.ty * rz3 * vJ12;
            // 5. Sc = R S R^T ; pc = R mu + t
            const float *R = cam.R;
            float A[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    A[i * 3 + j] = vSc[i * 3 + 0] * R[j] + vSc[i * 3 + 1] * R[3 + j] + vSc[i * 3 + 2] * R[6 + j];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (!POSE_ONLY) vS[i * 3 + j] += R[i] * A[j] + R[3 + i] * A[3 + j] + R[6 + i] * A[6 + j];
                    vR[i * 3 + j] = 2.0f * (A[i * 3 + 0] * symget(S, 0, j) + A[i * 3 + 1] * symget(S, 1, j) +
                                            A[i * 3 + 2] * symget(S, 2, j)) +
                                    vpc[i] * mean[j];
                }
            if (!POSE_ONLY) {
#pragma unroll
                for (int j = 0; j < 3; ++j) vmean[j] += R[j] * vpc[0] + R[3 + j] * vpc[1] + R[6 + j] * vpc[2];
            }
        }
        if (view_partials) {
            // block reduction of the 12 pose-gradient entries of camera c
            float vals[12] = {vR[0], vR[1], vR[2], vpc[0], vR[3], vR[4], vR[5], vpc[1], vR[6], vR[7], vR[8], vpc[2]};
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const float tot = gsx_wave_sum(vals[k]);
                if (lane == 0) s_part[wave][k] = tot;
            }
            __syncthreads();
            if (threadIdx.x < 12) {
                float acc = 0.f;
#pragma unroll
                for (int w = 0; w < PBWD_THREADS / GSX_WAVE; ++w) acc += s_part[w][threadIdx.x];
                view_partials[((int64_t)block * C + c) * 12 + threadIdx.x] = acc;
            }
            __syncthreads();
        }
    }
    if (!active || POSE_ONLY) return;
    // 6. S = M M^T : vM = 2 vS M ; M = Rq diag(s)
    float vM[9], vRq[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            vM[i * 3 + j] = 2.0f * (vS[i * 3 + 0] * M[j] + vS[i * 3 + 1] * M[3 + j] + vS[i * 3 + 2] * M[6 + j]);
    float vs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        vs[j] = qr.R[j] * vM[j] + qr.R[3 + j] * vM[3 + j] + qr.R[6 + j] * vM[6 + j];
#pragma unroll
        for (int i = 0; i < 3; ++i) vRq[i * 3 + j] = vM[i * 3 + j] * s[j];
    }
    const float w = qr.qn[0], qx = qr.qn[1], qy = qr.qn[2], qz = qr.qn[3];
    float vq[4];
    vq[0] = 2.0f * (qx * (vRq[7] - vRq[5]) + qy * (vRq[2] - vRq[6]) + qz * (vRq[3] - vRq[1]));
    vq[1] = 2.0f * (-2.0f * qx * (vRq[4] + vRq[8]) + qy * (vRq[1] + vRq[3]) + qz * (vRq[2] + vRq[6]) + w * (vRq[7] - vRq[5]));
    vq[2] = 2.0f * (qx * (vRq[1] + vRq[3]) - 2.0f * qy * (vRq[0] + vRq[8]) + qz * (vRq[5] + vRq[7]) + w * (vRq[2] - vRq[6]));
    vq[3] = 2.0f * (qx * (vRq[2] + vRq[6]) + qy * (vRq[5] + vRq[7]) - 2.0f * qz * (vRq[0] + vRq[4]) + w * (vRq[3] - vRq[1]));
    const float dotp = vq[0] * qr.qn[0] + vq[1] * qr.qn[1] + vq[2] * qr.qn[2] + vq[3] * qr.qn[3];
    v_means[3 * g] = vmean[0]; v_means[3 * g + 1] = vmean[1]; v_means[3 * g + 2] = vmean[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) v_quats[4 * g + k] = (vq[k] - dotp * qr.qn[k]) * qr.inv_norm;
    if (flags & GSX_PROJ_LOG_SCALES) { vs[0] *= s[0]; vs[1] *= s[1]; vs[2] *= s[2]; }
    v_scales[3 * g] = vs[0]; v_scales[3 * g + 1] = vs[1]; v_scales[3 * g + 2] = vs[2];
    if (v_rec) {
        const float o = gsx_sigmoid(logit_opacities[g]);
        v_logit_opac[g] = v_opac_sum * o * (1.0f - o);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float cl = gsx_sigmoid(logit_colors[3 * g + k]);
            v_logit_colors[3 * g + k] = v_col_sum[k] * cl * (1.0f - cl);
        }
        if (v_log_unc) {
            float gl = 0.f;
            if (flags & GSX_PROJ_BETAS) {
                const float e = expf(log_unc[g]);
                gl = (e >= GSX_BETA_MIN) ? v_beta_sum * e : 0.f;  // clamp(min) passes the gradient where e >= min
            }
            v_log_unc[g] = gl;
        }
    }
}

__global__ __launch_bounds__(256) void project_bwd_finish_kernel(const float *__restrict__ partials, int n_blocks,
                                                                 int C, float *__restrict__ v_viewmats) {
    // one workgroup per camera: 21 row-strided accumulators x 12 columns, then a 21-term fold per column
    __shared__ float s_acc[21][12];
    const int c = blockIdx.x;
    const int k = threadIdx.x % 12, r = threadIdx.x / 12;
    if (r < 21) {
        // eight independent accumulators: the loads of a trip are all in flight together (a 500 k map has ~2000
        // block partials per camera; one dependent load per trip made this tiny kernel take 23 us)
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        int b = r;
        for (; b + 7 * 21 < n_blocks; b += 8 * 21) {
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[u] += partials[((int64_t)(b + u * 21) * C + c) * 12 + k];
        }
        for (; b < n_blocks; b += 21) acc[0] += partials[((int64_t)b * C + c) * 12 + k];
        s_acc[r][k] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float acc = 0.f;
        if (threadIdx.x < 12)
            for (int rr = 0; rr < 21; ++rr) acc += s_acc[rr][threadIdx.x];
        v_viewmats[c * 16 + threadIdx.x] = acc;  // row 3 (entries 12..15) stays zero
    }
}

__global__ void qs2cp_kernel(const float *__restrict__ quats, const float *__restrict__ scales, int64_t n,
                             float *__restrict__ covars, float *__restrict__ precis) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float q[4] = {quats[4 * i], quats[4 * i + 1], quats[4 * i + 2], quats[4 * i + 3]};
    const float s[3] = {scales[3 * i], scales[3 * i + 1], scales[3 * i + 2]};
    QuatRot qr;
    quat_to_rotmat(q, qr);
    float M[9];
    Sym3 S;
    covar_from_rot_scale(qr.R, s, M, S);
    float *Cc = covars + 9 * i;
    Cc[0] = S.a00; Cc[1] = S.a01; Cc[2] = S.a02; Cc[3] = S.a01; Cc[4] = S.a11; Cc[5] = S.a12;
    Cc[6] = S.a02; Cc[7] = S.a12; Cc[8] = S.a22;
    if (precis) {
        const float is[3] = {1.0f / s[0], 1.0f / s[1], 1.0f / s[2]};
        covar_from_rot_scale(qr.R, is, M, S);
        float *P = precis + 9 * i;
        P[0] = S.a00; P[1] = S.a01; P[2] = S.a02; P[3] = S.a01; P[4] = S.a11; P[5] = S.a12;
        P[6] = S.a02; P[7] = S.a12; P[8] = S.a22;
    }
}

template <int CH, int RS>
__global__ void pack_records_kernel(const float *__restrict__ means2d, const float *__restrict__ conics,
                                    const float *__restrict__ opac, const float *__restrict__ colors, int64_t CN,
                                    float *__restrict__ rec) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CN) return;
    float r[RS];
#pragma unroll
    for (int k = 0; k < RS; ++k) r[k] = 0.f;
    r[0] = means2d[2 * i]; r[1] = means2d[2 * i + 1];
    r[2] = conics[3 * i]; r[3] = conics[3 * i + 1]; r[4] = conics[3 * i + 2];
    r[5] = opac[i];
#pragma unroll
    for (int k = 0; k < CH; ++k) r[6 + k] = colors[i * CH + k];
    float4 *o = reinterpret_cast<float4 *>(rec + i * RS);
#pragma unroll
    for (int k = 0; k < RS / 4; ++k) o[k] = make_float4(r[4 * k], r[4 * k + 1], r[4 * k + 2], r[4 * k + 3]);
}

}  // namespace

extern "C" int gsx_record_stride(int CH) {
    if (CH >= 1 && CH <= 2) return 8;
    if (CH >= 3 && CH <= 5) return 12;
    return GSX_E_UNSUPPORTED;
}

static int project_fwd_impl(const float *means, const float *quats, const float *scales, const float *viewmats,
                            const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                            float far_plane, float radius_clip, int flags, int32_t *radii, float *means2d,
                            float *depths, float *conics, float *comps, int32_t *tiles_per_gauss, int tile_w,
                            int tile_h, const float *logit_opacities, const float *logit_colors,
                            const float *log_uncertainties, float *rec, int32_t *vis_count, float *v_rec_clear,
                            uint32_t *rects, void *stream) {
    GSX_CHECK_ARG(N >= 0 && C >= 1 && W > 0 && H > 0);
    if (rects) GSX_CHECK_ARG(rec && tile_w > 0 && tile_h > 0 && tile_w < 256 && tile_h < 256);
    GSX_CHECK_ARG(!v_rec_clear || rec);
    GSX_CHECK_ARG(means && quats && scales && viewmats && Ks && radii && means2d && depths);
    GSX_CHECK_ARG(conics || rec);                             // the conic is in the record as well (columns 2..4)
    if (rec) {
        GSX_CHECK_ARG(logit_opacities && logit_colors);
        GSX_CHECK_ARG(!(flags & GSX_PROJ_BETAS) || log_uncertainties);
    }
    if (tiles_per_gauss) GSX_CHECK_ARG(tile_w > 0 && tile_h > 0);
    if (N == 0) return GSX_OK;
    const int threads = 256;
    const unsigned blocks = (unsigned)((N + threads - 1) / threads);
    hipStream_t st = (hipStream_t)stream;
    if (rec)
        hipLaunchKernelGGL((project_fwd_kernel<12>), dim3(blocks), dim3(threads), 0, st, means, quats, scales,
                           viewmats, Ks, N, (int)C, W, H, eps2d, near_plane, far_plane, radius_clip, flags, radii,
                           means2d, depths, conics, comps, tiles_per_gauss, tile_w, tile_h, logit_opacities,
                           logit_colors, log_uncertainties, rec, vis_count, v_rec_clear, rects);
    else
        hipLaunchKernelGGL((project_fwd_kernel<0>), dim3(blocks), dim3(threads), 0, st, means, quats, scales,
                           viewmats, Ks, N, (int)C, W, H, eps2d, near_plane, far_plane, radius_clip, flags, radii,
                           means2d, depths, conics, comps, tiles_per_gauss, tile_w, tile_h, logit_opacities,
                           logit_colors, log_uncertainties, rec, vis_count, v_rec_clear, rects);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_project_fwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                               const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                               float far_plane, float radius_clip, int flags, int32_t *radii, float *means2d,
                               float *depths, float *conics, float *comps, int32_t *tiles_per_gauss, int tile_w,
                               int tile_h, const float *logit_opacities, const float *logit_colors,
                               const float *log_uncertainties, float *rec, int32_t *vis_count, float *v_rec_clear,
                               void *stream) {
    return project_fwd_impl(means, quats, scales, viewmats, Ks, N, C, W, H, eps2d, near_plane, far_plane, radius_clip, flags, radii,
                            means2d, depths, conics, comps, tiles_per_gauss, tile_w, tile_h, logit_opacities, logit_colors,
                            log_uncertainties, rec, vis_count, v_rec_clear, nullptr, stream);
}

// gsx_project_fwd that also packs, per (camera, Gaussian), the rectangle of tiles the instance is listed in (tile_rect.h: one
// uint32; the reference's 3-sigma square, or - flags & GSX_PROJ_TILE_EXACT - the tight one) for gsx_isect_bin_sort_rects
extern "C" int gsx_project_fwd_rects(const float *means, const float *quats, const float *scales, const float *viewmats,
                                     const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                                     float far_plane, float radius_clip, int flags, int32_t *radii, float *means2d,
                                     float *depths, float *conics, float *comps, int32_t *tiles_per_gauss, int tile_w,
                                     int tile_h, const float *logit_opacities, const float *logit_colors,
                                     const float *log_uncertainties, float *rec, int32_t *vis_count, float *v_rec_clear,
                                     uint32_t *rects, void *stream) {
    GSX_CHECK_ARG(rects != nullptr);
    return project_fwd_impl(means, quats, scales, viewmats, Ks, N, C, W, H, eps2d, near_plane, far_plane, radius_clip, flags, radii,
                            means2d, depths, conics, comps, tiles_per_gauss, tile_w, tile_h, logit_opacities, logit_colors,
                            log_uncertainties, rec, vis_count, v_rec_clear, rects, stream);
}

extern "C" int64_t gsx_project_bwd_blocks(int64_t N) { return (N + PBWD_THREADS - 1) / PBWD_THREADS; }

extern "C" int64_t gsx_project_bwd_workspace_bytes(int64_t N, int64_t C) {
    const int64_t blocks = (N + PBWD_THREADS - 1) / PBWD_THREADS;
    return gsx_align256(blocks * C * 12 * (int64_t)sizeof(float)) + 256;
}

// g_begin .. g_end: the rows this launch covers (g_begin on a workgroup boundary); the pose partials keep the row numbering of
// the whole map, so any set of ranges that tiles [0, N) leaves the workspace as ONE launch over [0, N) would.
static int project_bwd_impl(const float *means, const float *quats, const float *scales, const float *viewmats,
                            const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                            float far_plane, int flags, const int32_t *radii, const float *v_means2d,
                            int64_t v_means2d_stride, const float *v_depths, const float *v_conics,
                            int64_t v_conics_stride, const float *v_comps, const float *logit_opacities,
                            const float *logit_colors, const float *log_uncertainties, const float *v_rec,
                            float *v_means, float *v_quats, float *v_scales, float *v_viewmats,
                            float *v_logit_opacities, float *v_logit_colors, float *v_log_unc, void *workspace,
                            int64_t workspace_bytes, int64_t g_begin, int64_t g_end, void *stream) {
    GSX_CHECK_ARG(N >= 0 && C >= 1 && W > 0 && H > 0);
    GSX_CHECK_ARG(g_begin >= 0 && g_begin <= g_end && g_end <= N && g_begin % PBWD_THREADS == 0);
    GSX_CHECK_ARG(g_end == N || g_end % PBWD_THREADS == 0);     // a workgroup belongs to one range
    const bool whole = g_begin == 0 && g_end == N;
    GSX_CHECK_ARG(whole || (!v_viewmats && (flags & GSX_PROJ_VIEW_PARTIALS)));   // a range leaves partials, never the folded sum
    GSX_CHECK_ARG(means && quats && scales && viewmats && Ks && radii && v_means2d && v_conics);
    const bool pose_only = !v_means && !v_quats && !v_scales;     // Gaussian gradients not wanted (tracking)
    const bool partials_only = (flags & GSX_PROJ_VIEW_PARTIALS) != 0;   // leave the pose partials for a fused consumer
    GSX_CHECK_ARG(pose_only ? (v_viewmats != nullptr || partials_only) : (v_means && v_quats && v_scales));
    GSX_CHECK_ARG(v_means2d_stride >= 2 && v_conics_stride >= 3);
    GSX_CHECK_ARG(!(flags & GSX_PROJ_RESET_V_REC) || (v_rec && v_means2d == v_rec && v_means2d_stride == 12 &&
                                                      v_conics == v_rec + 2 && v_conics_stride == 12));
    if (v_rec && !pose_only) GSX_CHECK_ARG(logit_opacities && logit_colors && v_logit_opacities && v_logit_colors);
    if (v_rec && !pose_only && (flags & GSX_PROJ_BETAS)) GSX_CHECK_ARG(log_uncertainties && v_log_unc);
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        if (v_viewmats && !gsx_zero_async(v_viewmats, 16 * C, st)) return GSX_E_LAUNCH;
        return GSX_OK;
    }
    const unsigned all_blocks = (unsigned)((N + PBWD_THREADS - 1) / PBWD_THREADS);
    const int block0 = (int)(g_begin / PBWD_THREADS);
    const unsigned blocks = (unsigned)((g_end + PBWD_THREADS - 1) / PBWD_THREADS) - (unsigned)block0;
    if (blocks == 0) return GSX_OK;
    float *partials = nullptr;
    if (v_viewmats || partials_only) {
        if (workspace_bytes < gsx_project_bwd_workspace_bytes(N, C) || !workspace) {
            gsx_set_error("gsx_project_bwd: workspace too small");
            return GSX_E_WORKSPACE;
        }
        partials = (float *)workspace;
    }
#define GSX_PBWD_ARGS                                                                                                  \
    means, quats, scales, viewmats, Ks, N, (int)C, W, H, eps2d, near_plane, far_plane, flags, radii, v_means2d,        \
        v_means2d_stride, v_depths, v_conics, v_conics_stride, v_comps, logit_opacities, logit_colors,                 \
        log_uncertainties, v_rec, 12, v_means, v_quats, v_scales, partials, v_logit_opacities, v_logit_colors, v_log_unc, \
        block0
    if (pose_only)
        hipLaunchKernelGGL(project_bwd_kernel<true>, dim3(blocks), dim3(PBWD_THREADS), 0, st, GSX_PBWD_ARGS);
    else
        hipLaunchKernelGGL(project_bwd_kernel<false>, dim3(blocks), dim3(PBWD_THREADS), 0, st, GSX_PBWD_ARGS);
#undef GSX_PBWD_ARGS
    GSX_CHECK_LAUNCH();
    if (v_viewmats && !partials_only) {
        hipLaunchKernelGGL(project_bwd_finish_kernel, dim3((unsigned)C), dim3(256), 0, st, partials, (int)all_blocks,
                           (int)C, v_viewmats);
        GSX_CHECK_LAUNCH();
    }
    return GSX_OK;
}

#define GSX_PBWD_FORWARD                                                                                                \
    means, quats, scales, viewmats, Ks, N, C, W, H, eps2d, near_plane, far_plane, flags, radii, v_means2d,               \
        v_means2d_stride, v_depths, v_conics, v_conics_stride, v_comps, logit_opacities, logit_colors,                   \
        log_uncertainties, v_rec, v_means, v_quats, v_scales, v_viewmats, v_logit_opacities, v_logit_colors, v_log_unc,  \
        workspace, workspace_bytes

extern "C" int gsx_project_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                               const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                               float far_plane, int flags, const int32_t *radii, const float *v_means2d,
                               int64_t v_means2d_stride, const float *v_depths, const float *v_conics,
                               int64_t v_conics_stride, const float *v_comps, const float *logit_opacities,
                               const float *logit_colors, const float *log_uncertainties, const float *v_rec,
                               float *v_means, float *v_quats, float *v_scales, float *v_viewmats,
                               float *v_logit_opacities, float *v_logit_colors, float *v_log_unc, void *workspace,
                               int64_t workspace_bytes, void *stream) {
    return project_bwd_impl(GSX_PBWD_FORWARD, 0, N > 0 ? N : 0, stream);
}

extern "C" int gsx_project_bwd_range(const float *means, const float *quats, const float *scales, const float *viewmats,
                                     const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                                     float far_plane, int flags, const int32_t *radii, const float *v_means2d,
                                     int64_t v_means2d_stride, const float *v_depths, const float *v_conics,
                                     int64_t v_conics_stride, const float *v_comps, const float *logit_opacities,
                                     const float *logit_colors, const float *log_uncertainties, const float *v_rec,
                                     float *v_means, float *v_quats, float *v_scales, float *v_viewmats,
                                     float *v_logit_opacities, float *v_logit_colors, float *v_log_unc, void *workspace,
                                     int64_t workspace_bytes, int64_t g_begin, int64_t g_end, void *stream) {
    return project_bwd_impl(GSX_PBWD_FORWARD, g_begin, g_end, stream);
}
#undef GSX_PBWD_FORWARD

extern "C" int gsx_quat_scale_to_covar_preci(const float *quats, const float *scales, int64_t n, float *covars,
                                             float *precis, void *stream) {
    GSX_CHECK_ARG(n >= 0 && quats && scales && covars);
    if (n == 0) return GSX_OK;
    hipLaunchKernelGGL(qs2cp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, quats,
                       scales, n, covars, precis);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_pack_records(const float *means2d, const float *conics, const float *opacities,
                                const float *colors, int64_t N, int64_t C, int CH, float *rec, void *stream) {
    GSX_CHECK_ARG(means2d && conics && opacities && colors && rec && N >= 0 && C >= 1);
    const int64_t CN = N * C;
    if (CN == 0) return GSX_OK;
    const unsigned blocks = (unsigned)((CN + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
#define LAUNCH(ch, rs)                                                                                              \
    hipLaunchKernelGGL((pack_records_kernel<ch, rs>), dim3(blocks), dim3(256), 0, st, means2d, conics, opacities, \
                       colors, CN, rec)
    switch (CH) {
        case 1: LAUNCH(1, 8); break;
        case 2: LAUNCH(2, 8); break;
        case 3: LAUNCH(3, 12); break;
        case 4: LAUNCH(4, 12); break;
        case 5: LAUNCH(5, 12); break;
        default: gsx_set_error("gsx_pack_records: CH=%d unsupported (1..5)", CH); return GSX_E_UNSUPPORTED;
    }
#undef LAUNCH
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
