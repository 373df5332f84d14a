// pose.hip — PoseZhou for a whole window in one launch: viewmat_c = Rt_c * [[GramSchmidt(dR_c + id6), dt_c], [0 0 0 1]]
// (gslam/primitives.py:15-36,82-92; pytorch3d's rotation_6d_to_matrix with F.normalize's eps = 1e-12), forward and
// backward.  Replaces ~15 + ~30 torch micro-kernels per pose per render; the backward consumes the v_viewmats that K2
// produces.  One thread per pose; parameters of the (separate) pose modules are reached through pointer tables passed
// by value.
#include "gsx_common.h"
#include "pose_math.h"

namespace {

constexpr int POSE_MAX = 16;

struct PoseArgs {
    const float *Rt[POSE_MAX];
    const float *dR[POSE_MAX];
    const float *dt[POSE_MAX];
    float *v_dR[POSE_MAX];
    float *v_dt[POSE_MAX];
    int learnable[POSE_MAX];
    int count;
};

__global__ void pose_fwd_kernel(PoseArgs a, float *__restrict__ viewmats) {
    const int c = threadIdx.x;
    if (c >= a.count) return;
    const float *Rt = a.Rt[c];
    float *V = viewmats + 16 * c;
    if (!a.learnable[c]) {
        for (int i = 0; i < 16; ++i) V[i] = Rt[i];
        return;
    }
    gsx_pose::pose_fwd_one(Rt, a.dR[c], a.dt[c], V);
}

__global__ void pose_bwd_kernel(PoseArgs a, const float *__restrict__ v_viewmats) {
    const int c = threadIdx.x;
    if (c >= a.count || !a.learnable[c]) return;
    gsx_pose::pose_bwd_one(a.Rt[c], a.dR[c], v_viewmats + 16 * c, a.v_dR[c], a.v_dt[c]);
}

// The projection backward's per-workgroup pose partials [n_blocks][C][12] summed straight into the PoseZhou backward:
// one workgroup per pose (the summation order of project_bwd_finish_kernel, eight loads in flight per lane), lane 0
// then runs the 4x4 algebra.  Replaces project_bwd_finish_kernel + pose_bwd_kernel (one launch less on the BA step's
// critical path).  v_extra: gradient of the view matrices that reached them by other routes, or null.
__global__ __launch_bounds__(256) void pose_bwd_partials_kernel(PoseArgs a, const float *__restrict__ partials,
                                                                int n_blocks, const float *__restrict__ v_extra) {
    __shared__ float s_acc[21][12];
    __shared__ float s_v[16];
    const int c = blockIdx.x, C = a.count;
    if (!a.learnable[c]) return;                              // workgroup-uniform
    const int k = threadIdx.x % 12, r = threadIdx.x / 12;
    if (r < 21) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int b = r; b < n_blocks; b += 8 * 21) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int bb = b + u * 21;
                acc[u] += bb < n_blocks ? partials[((int64_t)bb * C + c) * 12 + k] : 0.f;
            }
        }
        s_acc[r][k] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float acc = 0.f;
        if (threadIdx.x < 12) {
            for (int rr = 0; rr < 21; ++rr) acc += s_acc[rr][threadIdx.x];
            if (v_extra) acc += v_extra[c * 16 + threadIdx.x];
        }
        s_v[threadIdx.x] = acc;                               // row 3 of the view-matrix gradient does not reach the pose
    }
    __syncthreads();
    if (threadIdx.x == 0) gsx_pose::pose_bwd_one(a.Rt[c], a.dR[c], s_v, a.v_dR[c], a.v_dt[c]);
}

int fill(PoseArgs &a, int C, const float *const *Rt, const float *const *dR, const float *const *dt, const int *learnable,
         float *const *v_dR, float *const *v_dt) {
    a.count = C;
    for (int c = 0; c < POSE_MAX; ++c) {
        const bool in = c < C;
        a.Rt[c] = in ? Rt[c] : nullptr;
        a.learnable[c] = in ? learnable[c] : 0;
        a.dR[c] = (in && learnable[c]) ? dR[c] : nullptr;
        a.dt[c] = (in && learnable[c]) ? dt[c] : nullptr;
        a.v_dR[c] = (in && learnable[c] && v_dR) ? v_dR[c] : nullptr;
        a.v_dt[c] = (in && learnable[c] && v_dt) ? v_dt[c] : nullptr;
        if (in && !Rt[c]) return -1;
        if (in && learnable[c] && (!dR[c] || !dt[c])) return -1;
    }
    return 0;
}

}  // namespace

extern "C" int gsx_pose_zhou_fwd(int C, const float *const *Rt, const float *const *dR, const float *const *dt,
                                 const int *learnable, float *viewmats, void *stream) {
    GSX_CHECK_ARG(C >= 1 && C <= POSE_MAX && Rt && dR && dt && learnable && viewmats);
    PoseArgs a;
    GSX_CHECK_ARG(fill(a, C, Rt, dR, dt, learnable, nullptr, nullptr) == 0);
    hipLaunchKernelGGL(pose_fwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, viewmats);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_pose_zhou_bwd(int C, const float *const *Rt, const float *const *dR, const float *const *dt,
                                 const int *learnable, const float *v_viewmats, float *const *v_dR, float *const *v_dt,
                                 void *stream) {
    GSX_CHECK_ARG(C >= 1 && C <= POSE_MAX && Rt && dR && dt && learnable && v_viewmats && v_dR && v_dt);
    PoseArgs a;
    GSX_CHECK_ARG(fill(a, C, Rt, dR, dt, learnable, v_dR, v_dt) == 0);
    for (int c = 0; c < C; ++c) GSX_CHECK_ARG(!learnable[c] || (v_dR[c] && v_dt[c]));
    hipLaunchKernelGGL(pose_bwd_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, v_viewmats);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_pose_zhou_bwd_partials(int C, const float *const *Rt, const float *const *dR, const float *const *dt,
                                          const int *learnable, const float *partials, int64_t n_blocks,
                                          const float *v_viewmats_extra, float *const *v_dR, float *const *v_dt,
                                          void *stream) {
    GSX_CHECK_ARG(C >= 1 && C <= POSE_MAX && Rt && dR && dt && learnable && partials && v_dR && v_dt);
    GSX_CHECK_ARG(n_blocks >= 0 && n_blocks < ((int64_t)1 << 31));
    PoseArgs a;
    GSX_CHECK_ARG(fill(a, C, Rt, dR, dt, learnable, v_dR, v_dt) == 0);
    for (int c = 0; c < C; ++c) GSX_CHECK_ARG(!learnable[c] || (v_dR[c] && v_dt[c]));
    hipLaunchKernelGGL(pose_bwd_partials_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, a, partials,
                       (int)n_blocks, v_viewmats_extra);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
