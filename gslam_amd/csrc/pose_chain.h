// pose_chain.h - d loss / d [R | t] of ONE visible (camera, Gaussian) instance from its record-shaped gradient row: the pose part of
// K2 (project_bwd_kernel<POSE_ONLY>, gslam/rasterization.py:151-181 through autograd).  Shared by front_pose_bwd_kernel
// (isect_bin.hip: one thread per instance record, after the rasteriser's backward) and by the fused tracking rasteriser
// (raster_v4.inc, round 5: one thread per accumulator row of a tile, inside its flush - the launch of its own is gone).
#pragma once
#include "project_core.h"

namespace gsx_proj {

// p = project_core(mean, S, cam) of the instance (must have returned true); (vmx, vmy) = d loss / d mean2d, (va, vb_raw, vc) = the
// record's conic gradient columns (column 3 carries the OFF-diagonal term once: vb = vb_raw / 2 per symmetric entry), vdepth =
// d loss / d depth.  Adds the 12 entries of d loss / d [R | t] (row-major 3x4) into acc.
__device__ __forceinline__ void pose_chain_row(const Proj &p, const float mean[3], const Sym3 &S, const Cam &cam, float vmx,
                                               float vmy, float va_raw, float vb_raw, float vc_raw, float vdepth,
                                               float acc[12]) {
        // 1. conic = inverse(blurred cov2d): GX = -Y G Y
        const float a = p.conic[0], b = p.conic[1], cc = p.conic[2];
        const float va = va_raw, vb = 0.5f * vb_raw, vc = vc_raw;
        const float P00 = va * a + vb * b, P01 = va * b + vb * cc;
        const float P10 = vb * a + vc * b, P11 = vb * b + vc * cc;
        const float G00 = -(a * P00 + b * P10), G01 = -(a * P01 + b * P11), G11 = -(b * P01 + cc * P11);
        // 3. cov2d = J Sc J^T
        const float Jm[6] = {p.J00, 0.f, p.J02, 0.f, p.J11, p.J12};
        const float Gm[4] = {G00, G01, G01, G11};
        float GJ[6];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 3; ++j) GJ[i2 * 3 + j] = Gm[i2 * 2 + 0] * Jm[j] + Gm[i2 * 2 + 1] * Jm[3 + j];
        float vSc[9];
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
            for (int j = 0; j < 3; ++j) vSc[i2 * 3 + j] = Jm[i2] * GJ[j] + Jm[3 + i2] * GJ[3 + j];
        float vJ[6];
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                vJ[i2 * 3 + j] = 2.0f * (GJ[i2 * 3 + 0] * symget(p.Sc, 0, j) + GJ[i2 * 3 + 1] * symget(p.Sc, 1, j) +
                                         GJ[i2 * 3 + 2] * symget(p.Sc, 2, j));
        const float fx = cam.fx, fy = cam.fy;
        const float x = p.pc[0], y = p.pc[1], rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
        float vpc[3];
        vpc[0] = fx * rz * vmx;
        vpc[1] = fy * rz * vmy;
        vpc[2] = -(fx * x * vmx + fy * y * vmy) * rz2 + vdepth;
        const float vJ00 = vJ[0], vJ02 = vJ[2], vJ11 = vJ[4], vJ12 = vJ[5];
        if (p.x_in) vpc[0] += -fx * rz2 * vJ02; else vpc[2] += -fx * rz3 * vJ02 * p.tx;
        if (p.y_in) vpc[1] += -fy * rz2 * vJ12; else vpc[2] += -fy * rz3 * vJ12 * p.ty;
        vpc[2] += -fx * rz2 * vJ00 - fy * rz2 * vJ11 + 2.0f * fx * p.tx * rz3 * vJ02 + 2.0f * fy * p.ty * rz3 * vJ12;
        // 5. Sc = R S R^T ; pc = R mu + t
        const float *Rm = cam.R;
        float A[9];
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                A[i2 * 3 + j] = vSc[i2 * 3 + 0] * Rm[j] + vSc[i2 * 3 + 1] * Rm[3 + j] + vSc[i2 * 3 + 2] * Rm[6 + j];
#pragma unroll
        for (int i2 = 0; i2 < 3; ++i2) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[i2 * 4 + j] += 2.0f * (A[i2 * 3 + 0] * symget(S, 0, j) + A[i2 * 3 + 1] * symget(S, 1, j) +
                                           A[i2 * 3 + 2] * symget(S, 2, j)) +
                                   vpc[i2] * mean[j];
            acc[i2 * 4 + 3] += vpc[i2];
        }
}

}  // namespace gsx_proj
