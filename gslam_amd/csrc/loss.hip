// loss.hip — the mapping / tracking loss block fused into single passes with analytic gradients.
// Replaces the dozens of torch elementwise + reduction kernels of gslam/backend.py:273-318 (exposure affine,
// active-NeRF photometric + 0.5 log^2 beta, edge_aware_tv of gslam/utils.py:136-161, isotropic regulariser) and of
// gslam/frontend.py:113-138,632-646 (tracking loss), reading the render and the target once (~80 B/pixel) and
// writing the gradient of the render once (20 B/pixel).  The loss is always followed by its backward in the
// reference loops, so value and gradient are produced together; the autograd wrapper only scales by the upstream
// gradient.  All reductions are wave64 DPP -> LDS -> one partial row per workgroup -> finishing kernel
// (deterministic, no atomics).
#include "gsx_common.h"
#include "loss_pixel.h"

namespace {

using namespace gsx_loss;

__global__ __launch_bounds__(LB) void map_loss_kernel(LossArgs A) {
    __shared__ float s_part[LB / GSX_WAVE][NPART];
    const int c = blockIdx.y;
    const int HW = A.H * A.W;
    // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs, each with its own L2; the workgroups of one
    // XCD take one contiguous eighth of the image's 256-pixel blocks, so that the rows above and below a block (the
    // edge-aware TV reads all four neighbours) are in THIS XCD's L2 instead of being fetched again through the fabric:
    // FETCH_SIZE showed 2.0x the algorithmic bytes with the natural order (tools/ubench/fetch_calib.hip `read_loss_shape`
    // reproduces it: the x2 correction of the counter holds, the re-fetch is real).  A speed assumption only.
    int lb;
    {
        const int total = (int)gridDim.x, b = (int)blockIdx.x, xcd = b % 8, k = b / 8;
        const int qn = total / 8, rn = total % 8;
        lb = (xcd < rn ? xcd * (qn + 1) : rn * (qn + 1) + (xcd - rn) * qn) + k;
    }
    const int i = lb * LB + threadIdx.x;
    float part[NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (i < HW) {
        const int y = i / A.W, x = i - y * A.W;
        const int64_t p = (int64_t)c * HW + i;
        const float *rp = A.render + p * A.CH;
        float sg[3] = {0.f, 0.f, 0.f};
        if (A.ssim_grad) {
            const int64_t o = (int64_t)c * 3 * HW + i;
            sg[0] = A.ssim_grad[o]; sg[1] = A.ssim_grad[o + HW]; sg[2] = A.ssim_grad[o + 2 * HW];
        }
        map_loss_pixel(A, c, x, y, rp[0], rp[1], rp[2], A.gt[p * 3], A.gt[p * 3 + 1], A.gt[p * 3 + 2], A.ssim_grad != nullptr,
                       sg, part);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < NPART; ++k) {
        const float tot = gsx_wave_sum(part[k]);
        if (lane == 0) s_part[wave][k] = tot;
    }
    __syncthreads();
    if (threadIdx.x < NPART) {
        float acc = 0.f;
#pragma unroll
        for (int w = 0; w < LB / GSX_WAVE; ++w) acc += s_part[w][threadIdx.x];
        A.partials[((int64_t)c * gridDim.x + blockIdx.x) * NPART + threadIdx.x] = acc;
    }
}

// one workgroup: sums[0..2] = total S-term, log-beta term, tv over all cameras; v_exposure[c] = (v_a, v_b)
__global__ __launch_bounds__(LB) void map_loss_finish_kernel(const float *__restrict__ partials, int C, int blocks_per_cam,
                                                             float *__restrict__ sums, float *__restrict__ v_exposure) {
    __shared__ float s_red[LB / GSX_WAVE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_sum = [&](float v) -> float {
        const float tot = gsx_wave_sum(v);
        __syncthreads();
        if (lane == 0) s_red[wave] = tot;
        __syncthreads();
        return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
    };
    for (int k = 0; k < 3; ++k) {
        float acc = 0.f;
        for (int i = threadIdx.x; i < C * blocks_per_cam; i += LB) acc += partials[(int64_t)i * NPART + k];
        const float tot = block_sum(acc);
        if (threadIdx.x == 0) sums[k] = tot;
    }
    for (int c = 0; c < C; ++c)
        for (int k = 3; k < 5; ++k) {
            float acc = 0.f;
            for (int i = threadIdx.x; i < blocks_per_cam; i += LB)
                acc += partials[((int64_t)c * blocks_per_cam + i) * NPART + k];
            const float tot = block_sum(acc);
            if (threadIdx.x == 0 && v_exposure) v_exposure[2 * c + (k - 3)] = tot;
        }
}

// isotropic regulariser (backend.py:287-296): sum over visible Gaussians of sum_j |exp(s_j) - exp(mean(s))|, mean detached
template <bool ACC>
__global__ __launch_bounds__(LB) void isotropic_kernel(const float *__restrict__ log_scales,
                                                       const int32_t *__restrict__ vis_count, int64_t N, float weight,
                                                       float *__restrict__ partials, float *__restrict__ v_log_scales) {
    __shared__ float s_red[LB / GSX_WAVE];
    const int64_t g = (int64_t)blockIdx.x * LB + threadIdx.x;
    float term = 0.f;
    if (g < N) {
        const float s0 = log_scales[3 * g], s1 = log_scales[3 * g + 1], s2 = log_scales[3 * g + 2];
        float v0 = 0.f, v1 = 0.f, v2 = 0.f;
        if (vis_count[g] > 0) {
            const float m = expf((s0 + s1 + s2) * (1.0f / 3.0f));
            const float e0 = expf(s0), e1 = expf(s1), e2 = expf(s2);
            term = fabsf(e0 - m) + fabsf(e1 - m) + fabsf(e2 - m);
            v0 = weight * sgn(e0 - m) * e0; v1 = weight * sgn(e1 - m) * e1; v2 = weight * sgn(e2 - m) * e2;
        }
        if (ACC) {
            if (v0 != 0.f || v1 != 0.f || v2 != 0.f) {
                v_log_scales[3 * g] += v0; v_log_scales[3 * g + 1] += v1; v_log_scales[3 * g + 2] += v2;
            }
        } else {
            v_log_scales[3 * g] = v0; v_log_scales[3 * g + 1] = v1; v_log_scales[3 * g + 2] = v2;
        }
    }
    const float tot = gsx_wave_sum(term);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ __launch_bounds__(LB) void sum_partials_kernel(const float *__restrict__ partials, int64_t n,
                                                          float *__restrict__ out) {
    __shared__ float s_red[LB / GSX_WAVE];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += LB) acc += partials[i];
    const float tot = gsx_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

// out[0] = sum_k coef[k] * (*terms[k]) + bias ; out[1] = second linear form (the reference also reports the
// photometric term on its own: backend.py:340-354)
struct CombineArgs {
    const float *t[6];
    float c0[6], c1[6];
    float bias0, bias1;
    int n;
};
__global__ void combine_kernel(CombineArgs a, float *out) {
    if (threadIdx.x != 0) return;
    float o0 = a.bias0, o1 = a.bias1;
    for (int k = 0; k < a.n; ++k) {
        const float v = a.t[k][0];
        o0 += a.c0[k] * v;
        o1 += a.c1[k] * v;
    }
    out[0] = o0;
    out[1] = o1;
}

// Finishing kernel of a whole loss block: every producer above leaves one partial row per workgroup, and the separate
// one-workgroup reductions (map_loss_finish, the SSIM and isotropic partial sums, combine) each cost a ~4.5 us launch
// of their own in the replayed step.  Here the 16 wavefronts of one workgroup take the reductions round-robin
// (3 photometric sums, 2 exposure gradients per camera, SSIM, isotropic), then lane 0 forms the two linear forms.
struct FinishArgs {
    const float *map_part;   // [C * bpc][NPART]
    const float *ssim_part;  // [n_ssim] or null
    const float *iso_part;   // [n_iso] or null
    float *sums5;            // raw sums: photometric, log-beta, tv, ssim, iso (nullable)
    float *v_exposure;       // [C, 2] (nullable)
    float *out2;
    int C, bpc;
    int64_t n_ssim, n_iso;
    float c0[5], c1[5], bias0, bias1;
};
constexpr int FIN_WAVES = 16;

__global__ __launch_bounds__(64 * FIN_WAVES) void loss_finish_kernel(FinishArgs a) {
    // Work items = (camera, segment of its partial rows): S segments per camera so that a window of few cameras still
    // uses all 16 wavefronts; an item sums all five columns of its rows (one 24-byte row per load group, four rows in
    // flight per lane).  Wide windows (C > 16): one item per camera, the wavefronts take the cameras round-robin.
    __shared__ float s_glob[FIN_WAVES][3];                   // per-wavefront sums of the three window-wide terms
    __shared__ float s_exp[2 * FIN_WAVES][2];                // exposure-gradient pieces of (camera, segment), C * S <= 32
    __shared__ float s_aux[FIN_WAVES][2];                    // SSIM / isotropic pieces
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int S = a.C >= FIN_WAVES ? 1 : (FIN_WAVES + a.C - 1) / a.C;
    const int n_items = a.C * S;
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    for (int it = wave; it < n_items; it += FIN_WAVES) {
        const int c = it / S, seg = it - c * S;
        const int r0 = (int)((int64_t)a.bpc * seg / S), r1 = (int)((int64_t)a.bpc * (seg + 1) / S);
        const float *base = a.map_part + (int64_t)c * a.bpc * NPART;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int i = r0 + lane;
        for (; i + 192 < r1; i += 256) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float *row = base + (int64_t)(i + 64 * u) * NPART;
#pragma unroll
                for (int k = 0; k < 5; ++k) acc[k] += row[k];
            }
        }
        for (; i < r1; i += 64) {
            const float *row = base + (int64_t)i * NPART;
#pragma unroll
            for (int k = 0; k < 5; ++k) acc[k] += row[k];
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] = gsx_wave_sum(acc[k]);
        g0 += acc[0]; g1 += acc[1]; g2 += acc[2];
        if (lane == 0) {
            if (S == 1) { if (a.v_exposure) { a.v_exposure[2 * c] = acc[3]; a.v_exposure[2 * c + 1] = acc[4]; } }
            else { s_exp[it][0] = acc[3]; s_exp[it][1] = acc[4]; }
        }
    }
    // SSIM and isotropic partials: every wavefront takes a slice
    float sa = 0.f, ia = 0.f;
    if (a.ssim_part) for (int64_t i = threadIdx.x; i < a.n_ssim; i += 64 * FIN_WAVES) sa += a.ssim_part[i];
    if (a.iso_part) for (int64_t i = threadIdx.x; i < a.n_iso; i += 64 * FIN_WAVES) ia += a.iso_part[i];
    sa = gsx_wave_sum(sa); ia = gsx_wave_sum(ia);
    if (lane == 0) { s_glob[wave][0] = g0; s_glob[wave][1] = g1; s_glob[wave][2] = g2; s_aux[wave][0] = sa; s_aux[wave][1] = ia; }
    __syncthreads();
    if (S > 1 && a.v_exposure && threadIdx.x < 2 * a.C) {   // (camera, component): add the camera's segments
        const int c = threadIdx.x >> 1, k = threadIdx.x & 1;
        float acc = 0.f;
        for (int seg = 0; seg < S; ++seg) acc += s_exp[c * S + seg][k];
        a.v_exposure[2 * c + k] = acc;
    }
    if (threadIdx.x == 0) {
        float sum5[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int w = 0; w < FIN_WAVES; ++w) {
            sum5[0] += s_glob[w][0]; sum5[1] += s_glob[w][1]; sum5[2] += s_glob[w][2];
            sum5[3] += s_aux[w][0]; sum5[4] += s_aux[w][1];
        }
        float o0 = a.bias0, o1 = a.bias1;
#pragma unroll
        for (int k = 0; k < 5; ++k) { o0 += a.c0[k] * sum5[k]; o1 += a.c1[k] * sum5[k]; }
        a.out2[0] = o0; a.out2[1] = o1;
        if (a.sums5) for (int k = 0; k < 5; ++k) a.sums5[k] = sum5[k];
    }
}

__global__ __launch_bounds__(LB) void opacity_decay_kernel(float *__restrict__ logit_opac,
                                                           const int32_t *__restrict__ vis_count, int64_t N,
                                                           int min_count, float decay) {
    const int64_t g = (int64_t)blockIdx.x * LB + threadIdx.x;
    if (g < N && vis_count[g] > min_count) logit_opac[g] *= decay;
}

}  // namespace

extern "C" int64_t gsx_map_loss_workspace_bytes(int64_t C, int H, int W) {
    const int64_t blocks = ((int64_t)H * W + LB - 1) / LB;
    return gsx_align256(C * blocks * NPART * (int64_t)sizeof(float)) + 256;
}

extern "C" int gsx_map_loss(const float *render, const float *alphas, const float *gt, const float *exposure, int64_t C,
                            int H, int W, int CH, int depth_index, int beta_index, int mode, float w_photo, float w_tv,
                            float mask_thresh, const float *ssim_grad, float *sums, float *v_render, float *v_exposure,
                            void *workspace, int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(render && gt && exposure && v_render && C >= 1 && H > 0 && W > 0 && CH >= 3);
    GSX_CHECK_ARG(mode >= 0 && mode <= 2);
    GSX_CHECK_ARG(mode == 1 || (beta_index >= 3 && beta_index < CH));
    GSX_CHECK_ARG(w_tv == 0.f || (alphas && depth_index >= 3 && depth_index < CH));
    GSX_CHECK_ARG(C < 65536);
    if (!workspace || workspace_bytes < gsx_map_loss_workspace_bytes(C, H, W)) {
        gsx_set_error("gsx_map_loss: workspace too small");
        return GSX_E_WORKSPACE;
    }
    LossArgs A;
    A.render = render; A.alphas = alphas; A.gt = gt; A.exposure = exposure; A.ssim_grad = ssim_grad;
    A.v_render = v_render; A.partials = (float *)workspace;
    A.H = H; A.W = W; A.CH = CH; A.depth_index = depth_index; A.beta_index = beta_index; A.mode = mode;
    A.w_photo = w_photo; A.w_tv = w_tv; A.mask_thresh = mask_thresh;
    const unsigned blocks = (unsigned)(((int64_t)H * W + LB - 1) / LB);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(map_loss_kernel, dim3(blocks, (unsigned)C), dim3(LB), 0, st, A);
    GSX_CHECK_LAUNCH();
    if (!sums) return GSX_OK;                                // deferred: gsx_loss_finish reads the partial rows
    hipLaunchKernelGGL(map_loss_finish_kernel, dim3(1), dim3(LB), 0, st, (const float *)workspace, (int)C, (int)blocks,
                       sums, v_exposure);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int64_t gsx_isotropic_workspace_bytes(int64_t N) {
    return gsx_align256(((N + LB - 1) / LB) * (int64_t)sizeof(float)) + 256;
}

static int isotropic_launch(const float *log_scales, const int32_t *vis_count, int64_t N, float weight, float *sum_out,
                            float *v_log_scales, bool accumulate, void *workspace, int64_t workspace_bytes,
                            void *stream) {
    GSX_CHECK_ARG(log_scales && vis_count && v_log_scales && N >= 0);
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        if (sum_out && !gsx_zero_async(sum_out, 1, st)) return GSX_E_LAUNCH;
        return GSX_OK;
    }
    if (!workspace || workspace_bytes < gsx_isotropic_workspace_bytes(N)) {
        gsx_set_error("gsx_isotropic_loss: workspace too small");
        return GSX_E_WORKSPACE;
    }
    const unsigned blocks = (unsigned)((N + LB - 1) / LB);
    if (accumulate)
        hipLaunchKernelGGL(isotropic_kernel<true>, dim3(blocks), dim3(LB), 0, st, log_scales, vis_count, N, weight,
                           (float *)workspace, v_log_scales);
    else
        hipLaunchKernelGGL(isotropic_kernel<false>, dim3(blocks), dim3(LB), 0, st, log_scales, vis_count, N, weight,
                           (float *)workspace, v_log_scales);
    GSX_CHECK_LAUNCH();
    if (!sum_out) return GSX_OK;                             // deferred: gsx_loss_finish reads the partials
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(LB), 0, st, (const float *)workspace, (int64_t)blocks, sum_out);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_isotropic_loss(const float *log_scales, const int32_t *vis_count, int64_t N, float weight,
                                  float *sum_out, float *v_log_scales, void *workspace, int64_t workspace_bytes,
                                  void *stream) {
    return isotropic_launch(log_scales, vis_count, N, weight, sum_out, v_log_scales, false, workspace, workspace_bytes,
                            stream);
}

extern "C" int gsx_isotropic_loss_acc(const float *log_scales, const int32_t *vis_count, int64_t N, float weight,
                                      float *sum_out, float *v_log_scales, void *workspace, int64_t workspace_bytes,
                                      void *stream) {
    return isotropic_launch(log_scales, vis_count, N, weight, sum_out, v_log_scales, true, workspace, workspace_bytes,
                            stream);
}

extern "C" int gsx_loss_finish(const void *map_loss_ws, int64_t C, int H, int W, const void *ssim_ws,
                               int64_t ssim_partials, const void *iso_ws, int64_t N, const float *coef0,
                               const float *coef1, float bias0, float bias1, float *sums5, float *v_exposure,
                               float *out2, void *stream) {
    GSX_CHECK_ARG(map_loss_ws && C >= 1 && C < 65536 && H > 0 && W > 0 && coef0 && coef1 && out2);
    GSX_CHECK_ARG(ssim_partials >= 0 && N >= 0);
    FinishArgs a;
    a.map_part = (const float *)map_loss_ws;
    a.ssim_part = ssim_partials > 0 ? (const float *)ssim_ws : nullptr;
    a.iso_part = N > 0 ? (const float *)iso_ws : nullptr;
    GSX_CHECK_ARG(ssim_partials == 0 || ssim_ws);
    a.sums5 = sums5; a.v_exposure = v_exposure; a.out2 = out2;
    a.C = (int)C; a.bpc = (int)(((int64_t)H * W + LB - 1) / LB);
    a.n_ssim = ssim_partials; a.n_iso = iso_ws ? (N + LB - 1) / LB : 0;
    if (!iso_ws) a.iso_part = nullptr;
    for (int k = 0; k < 5; ++k) { a.c0[k] = coef0[k]; a.c1[k] = coef1[k]; }
    a.bias0 = bias0; a.bias1 = bias1;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64 * FIN_WAVES), 0, (hipStream_t)stream, a);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_combine_terms(int n, const float *const *terms, const float *coef0, const float *coef1, float bias0,
                                 float bias1, float *out2, void *stream) {
    GSX_CHECK_ARG(n >= 1 && n <= 6 && terms && coef0 && coef1 && out2);
    CombineArgs a;
    a.n = n; a.bias0 = bias0; a.bias1 = bias1;
    for (int k = 0; k < 6; ++k) {
        a.t[k] = k < n ? terms[k] : nullptr;
        a.c0[k] = k < n ? coef0[k] : 0.f;
        a.c1[k] = k < n ? coef1[k] : 0.f;
        if (k < n) GSX_CHECK_ARG(terms[k] != nullptr);
    }
    hipLaunchKernelGGL(combine_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a, out2);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_opacity_decay(float *logit_opacities, const int32_t *vis_count, int64_t N, int min_count,
                                 float decay, void *stream) {
    GSX_CHECK_ARG(logit_opacities && vis_count && N >= 0);
    if (N == 0) return GSX_OK;
    hipLaunchKernelGGL(opacity_decay_kernel, dim3((unsigned)((N + LB - 1) / LB)), dim3(LB), 0, (hipStream_t)stream,
                       logit_opacities, vis_count, N, min_count, decay);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
