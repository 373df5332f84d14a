// project_core.h — the EWA projection arithmetic shared by K1 / K2 (project.hip) and the fused projection + binning front
// of the launch plans (isect_bin.hip).  Both translation units are compiled with -ffp-contract=off: radii / tile
// rectangles are integer outputs that must be reproducible bit for bit against the CPU oracle, so every product-sum below
// is written in a fixed order and never fused (DESIGN.md "numeric contract").
#pragma once
#include "gsx_common.h"

namespace gsx_proj {

struct Sym3 {  // symmetric 3x3: 00 01 02 11 12 22
    float a00, a01, a02, a11, a12, a22;
};

struct QuatRot {
    float R[9];
    float qn[4];
    float inv_norm;
};

__device__ __forceinline__ void quat_to_rotmat(const float q[4], QuatRot &o) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    const float n2 = w * w + x * x + y * y + z * z;
    const float inv = 1.0f / sqrtf(n2);
    w *= inv; x *= inv; y *= inv; z *= inv;
    o.qn[0] = w; o.qn[1] = x; o.qn[2] = y; o.qn[3] = z;
    o.inv_norm = inv;
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    const float xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    o.R[0] = 1.0f - 2.0f * (y2 + z2); o.R[1] = 2.0f * (xy - wz); o.R[2] = 2.0f * (xz + wy);
    o.R[3] = 2.0f * (xy + wz); o.R[4] = 1.0f - 2.0f * (x2 + z2); o.R[5] = 2.0f * (yz - wx);
    o.R[6] = 2.0f * (xz - wy); o.R[7] = 2.0f * (yz + wx); o.R[8] = 1.0f - 2.0f * (x2 + y2);
}

// M = Rq diag(s);  S = M M^T
__device__ __forceinline__ void covar_from_rot_scale(const float Rq[9], const float s[3], float M[9], Sym3 &S) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * s[j];
    S.a00 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2];
    S.a01 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5];
    S.a02 = M[0] * M[6] + M[1] * M[7] + M[2] * M[8];
    S.a11 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    S.a12 = M[3] * M[6] + M[4] * M[7] + M[5] * M[8];
    S.a22 = M[6] * M[6] + M[7] * M[7] + M[8] * M[8];
}

__device__ __forceinline__ float symget(const Sym3 &S, int i, int j) {
    // i, j compile-time after unrolling
    const int k = (i <= j) ? (i * 3 + j) : (j * 3 + i);
    switch (k) {
        case 0: return S.a00;
        case 1: return S.a01;
        case 2: return S.a02;
        case 4: return S.a11;
        case 5: return S.a12;
        default: return S.a22;
    }
}

struct Cam {
    float R[9];
    float t[3];
    float fx, fy, cx, cy;
};

__device__ __forceinline__ void load_cam(const float *__restrict__ viewmats, const float *__restrict__ Ks, int64_t c,
                                         Cam &cam) {
    const float *V = viewmats + 16 * c;
    const float *K = Ks + 9 * c;
    cam.R[0] = V[0]; cam.R[1] = V[1]; cam.R[2] = V[2]; cam.t[0] = V[3];
    cam.R[3] = V[4]; cam.R[4] = V[5]; cam.R[5] = V[6]; cam.t[1] = V[7];
    cam.R[6] = V[8]; cam.R[7] = V[9]; cam.R[8] = V[10]; cam.t[2] = V[11];
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
}

struct Proj {
    float pc[3];
    Sym3 Sc;
    float J00, J11, J02, J12, tx, ty, rz;
    bool x_in, y_in;
    float c00, c01, c11, det_orig, det;
    float conic[3];
};

// Returns false when culled by the near/far planes or det <= 0.  Operation order is the numeric contract shared
// with the oracle (DESIGN.md "numeric contract"): left-to-right sums, no FMA contraction.
__device__ __forceinline__ bool project_core(const float mean[3], const Sym3 &S, const Cam &cam, int W, int H,
                                             float eps2d, float near_p, float far_p, Proj &p) {
    const float *R = cam.R;
    p.pc[0] = ((R[0] * mean[0] + R[1] * mean[1]) + R[2] * mean[2]) + cam.t[0];
    p.pc[1] = ((R[3] * mean[0] + R[4] * mean[1]) + R[5] * mean[2]) + cam.t[1];
    p.pc[2] = ((R[6] * mean[0] + R[7] * mean[1]) + R[8] * mean[2]) + cam.t[2];
    if (p.pc[2] < near_p || p.pc[2] > far_p) return false;
    // Sc = R S R^T : Wm = R S, Sc = Wm R^T
    float Wm[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Wm[i * 3 + j] = (R[i * 3 + 0] * symget(S, 0, j) + R[i * 3 + 1] * symget(S, 1, j)) + R[i * 3 + 2] * symget(S, 2, j);
    p.Sc.a00 = (Wm[0] * R[0] + Wm[1] * R[1]) + Wm[2] * R[2];
    p.Sc.a01 = (Wm[0] * R[3] + Wm[1] * R[4]) + Wm[2] * R[5];
    p.Sc.a02 = (Wm[0] * R[6] + Wm[1] * R[7]) + Wm[2] * R[8];
    p.Sc.a11 = (Wm[3] * R[3] + Wm[4] * R[4]) + Wm[5] * R[5];
    p.Sc.a12 = (Wm[3] * R[6] + Wm[4] * R[7]) + Wm[5] * R[8];
    p.Sc.a22 = (Wm[6] * R[6] + Wm[7] * R[7]) + Wm[8] * R[8];

    const float fx = cam.fx, fy = cam.fy, cx = cam.cx, cy = cam.cy;
    const float tanx = 0.5f * (float)W / fx, tany = 0.5f * (float)H / fy;
    const float lim_xp = ((float)W - cx) / fx + GSX_FOV_SLACK * tanx;
    const float lim_xn = cx / fx + GSX_FOV_SLACK * tanx;
    const float lim_yp = ((float)H - cy) / fy + GSX_FOV_SLACK * tany;
    const float lim_yn = cy / fy + GSX_FOV_SLACK * tany;
    const float x = p.pc[0], y = p.pc[1], z = p.pc[2];
    const float rz = 1.0f / z, rz2 = rz * rz;
    const float xr = x * rz, yr = y * rz;
    p.x_in = (xr <= lim_xp) && (xr >= -lim_xn);
    p.y_in = (yr <= lim_yp) && (yr >= -lim_yn);
    p.tx = z * fminf(lim_xp, fmaxf(-lim_xn, xr));
    p.ty = z * fminf(lim_yp, fmaxf(-lim_yn, yr));
    p.rz = rz;
    p.J00 = fx * rz; p.J11 = fy * rz;
    p.J02 = -(fx * p.tx) * rz2; p.J12 = -(fy * p.ty) * rz2;
    float T0[3], T1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T0[j] = p.J00 * symget(p.Sc, 0, j) + p.J02 * symget(p.Sc, 2, j);
        T1[j] = p.J11 * symget(p.Sc, 1, j) + p.J12 * symget(p.Sc, 2, j);
    }
    p.c00 = T0[0] * p.J00 + T0[2] * p.J02;
    p.c01 = T0[1] * p.J11 + T0[2] * p.J12;
    p.c11 = T1[1] * p.J11 + T1[2] * p.J12;
    p.det_orig = p.c00 * p.c11 - p.c01 * p.c01;
    const float b00 = p.c00 + eps2d, b11 = p.c11 + eps2d;
    p.det = b00 * b11 - p.c01 * p.c01;
    if (p.det <= 0.0f) return false;
    const float inv_det = 1.0f / p.det;
    p.conic[0] = b11 * inv_det; p.conic[1] = -p.c01 * inv_det; p.conic[2] = b00 * inv_det;
    return true;
}

}  // namespace gsx_proj
