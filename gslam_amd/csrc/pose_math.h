// pose_math.h - PoseZhou for ONE pose as device functions (gslam/primitives.py:15-36,82-92): used by the window
// kernels of pose.hip and by the tracker's fused closure tail (track_opt_impl.inc).
#pragma once

namespace gsx_pose {

constexpr float NORM_EPS = 1e-12f;

struct Frame3 {
    float a1[3], a2[3], b1[3], b2[3], b3[3], b2u[3];
    float n1, n2, d12;
};

__device__ __forceinline__ float dot3(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const float *a, const float *b, float *o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ void gram_schmidt(const float *dR, Frame3 &f) {
    f.a1[0] = dR[0] + 1.0f; f.a1[1] = dR[1]; f.a1[2] = dR[2];
    f.a2[0] = dR[3]; f.a2[1] = dR[4] + 1.0f; f.a2[2] = dR[5];
    f.n1 = fmaxf(sqrtf(dot3(f.a1, f.a1)), NORM_EPS);
    for (int i = 0; i < 3; ++i) f.b1[i] = f.a1[i] / f.n1;
    f.d12 = dot3(f.b1, f.a2);
    for (int i = 0; i < 3; ++i) f.b2u[i] = f.a2[i] - f.d12 * f.b1[i];
    f.n2 = fmaxf(sqrtf(dot3(f.b2u, f.b2u)), NORM_EPS);
    for (int i = 0; i < 3; ++i) f.b2[i] = f.b2u[i] / f.n2;
    cross3(f.b1, f.b2, f.b3);
}

// viewmat [16] = Rt * [[GramSchmidt(dR + id6), dt], [0 0 0 1]]
__device__ __forceinline__ void pose_fwd_one(const float *Rt, const float *dR, const float *dt, float *V) {
    Frame3 f;
    gram_schmidt(dR, f);
    const float D[16] = {f.b1[0], f.b1[1], f.b1[2], dt[0], f.b2[0], f.b2[1], f.b2[2], dt[1],
                         f.b3[0], f.b3[1], f.b3[2], dt[2], 0.f, 0.f, 0.f, 1.f};
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 4; ++k) acc += Rt[i * 4 + k] * D[k * 4 + j];
            V[i * 4 + j] = acc;
        }
}

// gradient of the above: vV [16] -> v_dR [6], v_dt [3]
__device__ __forceinline__ void pose_bwd_one(const float *Rt, const float *dR, const float *vV, float *vdR, float *vdt) {
    float vD[16];  // Rt^T vV
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = 0.f;
            for (int k = 0; k < 4; ++k) acc += Rt[k * 4 + i] * vV[k * 4 + j];
            vD[i * 4 + j] = acc;
        }
    Frame3 f;
    gram_schmidt(dR, f);
    float vb1[3] = {vD[0], vD[1], vD[2]}, vb2[3] = {vD[4], vD[5], vD[6]}, vb3[3] = {vD[8], vD[9], vD[10]};
    float t[3];
    cross3(f.b2, vb3, t);                       // b3 = b1 x b2
    for (int i = 0; i < 3; ++i) vb1[i] += t[i];
    cross3(vb3, f.b1, t);
    for (int i = 0; i < 3; ++i) vb2[i] += t[i];
    float vb2u[3];                              // b2 = b2u / n2
    const float s2 = dot3(vb2, f.b2);
    const bool live2 = sqrtf(dot3(f.b2u, f.b2u)) > NORM_EPS;
    for (int i = 0; i < 3; ++i) vb2u[i] = live2 ? (vb2[i] - s2 * f.b2[i]) / f.n2 : vb2[i] / f.n2;
    float va2[3];                               // b2u = a2 - (b1.a2) b1
    const float s3 = dot3(f.b1, vb2u);
    for (int i = 0; i < 3; ++i) {
        va2[i] = vb2u[i] - s3 * f.b1[i];
        vb1[i] += -f.d12 * vb2u[i] - s3 * f.a2[i];
    }
    float va1[3];                               // b1 = a1 / n1
    const float s1 = dot3(vb1, f.b1);
    const bool live1 = sqrtf(dot3(f.a1, f.a1)) > NORM_EPS;
    for (int i = 0; i < 3; ++i) va1[i] = live1 ? (vb1[i] - s1 * f.b1[i]) / f.n1 : vb1[i] / f.n1;
    vdR[0] = va1[0]; vdR[1] = va1[1]; vdR[2] = va1[2]; vdR[3] = va2[0]; vdR[4] = va2[1]; vdR[5] = va2[2];
    vdt[0] = vD[3]; vdt[1] = vD[7]; vdt[2] = vD[11];
}

}  // namespace gsx_pose
