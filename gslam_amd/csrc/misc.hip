// misc.hip — error plumbing, K13 spherical harmonics, fused multi-tensor Adam, device self-tests.
#include <stdarg.h>

#include "gsx_common.h"

// ---- error plumbing -------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

void gsx_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gsx_last_error(void) { return g_err; }
extern "C" int gsx_version(void) { return 100; }

extern "C" int gsx_read_i64(const int64_t *dev_ptr, int64_t *host_out, void *stream) {
    GSX_CHECK_ARG(dev_ptr && host_out);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemcpyAsync(host_out, dev_ptr, sizeof(int64_t), hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess) {
        gsx_set_error("gsx_read_i64: %s", hipGetErrorString(hipGetLastError()));
        return GSX_E_LAUNCH;
    }
    return GSX_OK;
}

namespace {

// ---- K13: real spherical harmonics up to degree 3 ----------------------------------------------
constexpr float SH_C0 = 0.2820947917738781f;
constexpr float SH_C1 = 0.48860251190292f;
constexpr float SH_C2_0 = 1.0925484305920792f, SH_C2_1 = -1.0925484305920792f, SH_C2_2 = 0.31539156525252005f,
                SH_C2_3 = -1.0925484305920792f, SH_C2_4 = 0.5462742152960396f;
constexpr float SH_C3_0 = -0.5900435899266435f, SH_C3_1 = 2.890611442640554f, SH_C3_2 = -0.4570457994644658f,
                SH_C3_3 = 0.3731763325901154f, SH_C3_4 = -0.4570457994644658f, SH_C3_5 = 1.445305721320277f,
                SH_C3_6 = -0.5900435899266435f;

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float *B) {
    B[0] = SH_C0;
    if (deg < 1) return;
    B[1] = -SH_C1 * y; B[2] = SH_C1 * z; B[3] = -SH_C1 * x;
    if (deg < 2) return;
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = SH_C2_0 * xy; B[5] = SH_C2_1 * yz; B[6] = SH_C2_2 * (2.0f * zz - xx - yy);
    B[7] = SH_C2_3 * xz; B[8] = SH_C2_4 * (xx - yy);
    if (deg < 3) return;
    B[9] = SH_C3_0 * y * (3.0f * xx - yy);
    B[10] = SH_C3_1 * xy * z;
    B[11] = SH_C3_2 * y * (4.0f * zz - xx - yy);
    B[12] = SH_C3_3 * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
    B[13] = SH_C3_4 * x * (4.0f * zz - xx - yy);
    B[14] = SH_C3_5 * z * (xx - yy);
    B[15] = SH_C3_6 * x * (xx - 3.0f * yy);
}

__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float *dx, float *dy, float *dz) {
#pragma unroll
    for (int i = 0; i < 16; ++i) dx[i] = dy[i] = dz[i] = 0.f;
    if (deg < 1) return;
    dy[1] = -SH_C1; dz[2] = SH_C1; dx[3] = -SH_C1;
    if (deg < 2) return;
    dx[4] = SH_C2_0 * y; dy[4] = SH_C2_0 * x;
    dy[5] = SH_C2_1 * z; dz[5] = SH_C2_1 * y;
    dx[6] = SH_C2_2 * -2.0f * x; dy[6] = SH_C2_2 * -2.0f * y; dz[6] = SH_C2_2 * 4.0f * z;
    dx[7] = SH_C2_3 * z; dz[7] = SH_C2_3 * x;
    dx[8] = SH_C2_4 * 2.0f * x; dy[8] = SH_C2_4 * -2.0f * y;
    if (deg < 3) return;
    const float xx = x * x, yy = y * y, zz = z * z;
    dx[9] = SH_C3_0 * 6.0f * x * y; dy[9] = SH_C3_0 * (3.0f * xx - 3.0f * yy);
    dx[10] = SH_C3_1 * y * z; dy[10] = SH_C3_1 * x * z; dz[10] = SH_C3_1 * x * y;
    dx[11] = SH_C3_2 * -2.0f * x * y; dy[11] = SH_C3_2 * (4.0f * zz - xx - 3.0f * yy); dz[11] = SH_C3_2 * 8.0f * y * z;
    dx[12] = SH_C3_3 * -6.0f * x * z; dy[12] = SH_C3_3 * -6.0f * y * z; dz[12] = SH_C3_3 * (6.0f * zz - 3.0f * xx - 3.0f * yy);
    dx[13] = SH_C3_4 * (4.0f * zz - 3.0f * xx - yy); dy[13] = SH_C3_4 * -2.0f * x * y; dz[13] = SH_C3_4 * 8.0f * x * z;
    dx[14] = SH_C3_5 * 2.0f * x * z; dy[14] = SH_C3_5 * -2.0f * y * z; dz[14] = SH_C3_5 * (xx - yy);
    dx[15] = SH_C3_6 * (3.0f * xx - 3.0f * yy); dy[15] = SH_C3_6 * -6.0f * x * y;
}

// One thread per Gaussian, looping cameras, so the 192-byte coefficient row (degree 3) is read once per Gaussian.
// The rows of a workgroup's 256 Gaussians are one contiguous block of coeffs: it is copied global -> LDS with
// consecutive lanes on consecutive floats (full 128-byte lines), each lane then picks its own row out of LDS (row
// stride padded to an odd word count: no bank conflicts), and the backward's coefficient gradients leave the same
// way.  DEG is a template parameter so that basis, coefficients and accumulators are registers, not scratch.
constexpr int SH_BLOCK = 256;

// Staging of the coefficient rows of a workgroup's 256 Gaussians, ONLY of those some camera sees (radii > 0): nobody reads the
// others - two thirds of a 5 M map are outside a 1080p frustum, 640 of the 960 MB of coefficient rows.  Every wavefront stages
// the 64 rows its own lanes consume: the ballot of the lanes' visibility is the list of rows, one 192-byte load (48 lanes) per
// visible row, all of them independent.  (A per-element test inside the streaming copy - a division and a flag read per float -
// made both SH kernels twice as slow as reading everything.)
template <int DEG>
__device__ __forceinline__ void sh_stage_in(const float *__restrict__ coeffs, const int32_t *__restrict__ radii, int64_t g0,
                                            int64_t N, int C, int Kc, float *s_rows) {
    constexpr int NBC = (DEG + 1) * (DEG + 1) * 3, PITCH = NBC | 1;
    const int lane = threadIdx.x & 63, w0 = threadIdx.x & ~63;
    const int64_t g = g0 + threadIdx.x;
    bool vis = g < N;
    if (vis && radii) {
        vis = false;
        for (int c = 0; c < C; ++c) vis = vis || radii[(int64_t)c * N + g] > 0;
    }
    unsigned long long m = __ballot(vis);
    const float *src = coeffs + (g0 + w0) * (int64_t)Kc * 3;
    constexpr int U = 32;                                    // rows in flight per wavefront (one load each)
    while (m != 0ull) {
        int j[U];
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            j[u] = (m != 0ull) ? __ffsll((long long)m) - 1 : -1;
            m &= m - 1ull;                                   // (0 & anything = 0: stays empty once it is)
        }
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = (j[u] >= 0 && lane < NBC) ? src[(int64_t)j[u] * Kc * 3 + lane] : 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (j[u] >= 0 && lane < NBC) s_rows[(w0 + j[u]) * PITCH + lane] = v[u];
    }
}

// campos != nullptr: `dirs` holds the MEANS [N,3] and the view direction of (camera c, Gaussian g) is means[g] - campos[c],
// formed in registers - no [C,N,3] direction array is written and read back (120 MB at 5 M Gaussians)
template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_fwd_kernel(const float *__restrict__ dirs,
                                                          const float *__restrict__ coeffs,
                                                          const int32_t *__restrict__ radii, int64_t N, int C, int Kc,
                                                          float *__restrict__ colors,
                                                          const float *__restrict__ campos = nullptr) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NBC = NB * 3, PITCH = NBC | 1;
    __shared__ float s_rows[SH_BLOCK * PITCH];
    const int64_t g0 = (int64_t)blockIdx.x * SH_BLOCK;
    sh_stage_in<DEG>(coeffs, radii, g0, N, C, Kc, s_rows);
    __syncthreads();
    const int64_t g = g0 + threadIdx.x;
    if (g >= N) return;
    float co[NBC];
#pragma unroll
    for (int k = 0; k < NBC; ++k) co[k] = s_rows[threadIdx.x * PITCH + k];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f;
    if (campos) { m0 = dirs[3 * g]; m1 = dirs[3 * g + 1]; m2 = dirs[3 * g + 2]; }
    for (int c = 0; c < C; ++c) {
        const int64_t idx = (int64_t)c * N + g;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (!radii || radii[idx] > 0) {
            float x, y, z;
            if (campos) { x = m0 - campos[3 * c]; y = m1 - campos[3 * c + 1]; z = m2 - campos[3 * c + 2]; }
            else { x = dirs[3 * idx]; y = dirs[3 * idx + 1]; z = dirs[3 * idx + 2]; }
            const float n = sqrtf(x * x + y * y + z * z);
            const float inv = n > 0.f ? 1.0f / n : 0.f;
            x *= inv; y *= inv; z *= inv;
            float B[16];
            sh_basis(DEG, x, y, z, B);
#pragma unroll
            for (int k = 0; k < NB; ++k) { o0 += B[k] * co[3 * k]; o1 += B[k] * co[3 * k + 1]; o2 += B[k] * co[3 * k + 2]; }
            o0 = fmaxf(0.f, o0 + 0.5f); o1 = fmaxf(0.f, o1 + 0.5f); o2 = fmaxf(0.f, o2 + 0.5f);
        }
        colors[3 * idx] = o0; colors[3 * idx + 1] = o1; colors[3 * idx + 2] = o2;
    }
}

template <int DEG>
__global__ __launch_bounds__(SH_BLOCK) void sh_bwd_kernel(const float *__restrict__ dirs,
                                                          const float *__restrict__ coeffs,
                                                          const int32_t *__restrict__ radii,
                                                          const float *__restrict__ v_colors, int64_t N, int C, int Kc,
                                                          float *__restrict__ v_coeffs, float *__restrict__ v_dirs,
                                                          const float *__restrict__ campos = nullptr,
                                                          float *__restrict__ v_means = nullptr,
                                                          float *__restrict__ v_campos = nullptr) {
    constexpr int NB = (DEG + 1) * (DEG + 1), NBC = NB * 3, PITCH = NBC | 1;
    __shared__ float s_rows[SH_BLOCK * PITCH];
    __shared__ float s_cam[SH_BLOCK / 64][3];
    const int64_t g0 = (int64_t)blockIdx.x * SH_BLOCK;
    const int rows = (int)min((int64_t)SH_BLOCK, N - g0);
    sh_stage_in<DEG>(coeffs, radii, g0, N, C, Kc, s_rows);
    __syncthreads();
    const int64_t g = g0 + threadIdx.x;
    float co[NBC], vco[NBC];
#pragma unroll
    for (int k = 0; k < NBC; ++k) { co[k] = s_rows[threadIdx.x * PITCH + k]; vco[k] = 0.f; }
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, vm0 = 0.f, vm1 = 0.f, vm2 = 0.f;
    if (campos && g < N) { m0 = dirs[3 * g]; m1 = dirs[3 * g + 1]; m2 = dirs[3 * g + 2]; }
    for (int c = 0; c < C; ++c) {
        const int64_t idx = (int64_t)c * N + g;
        float vdx = 0.f, vdy = 0.f, vdz = 0.f;
        if (g < N && (!radii || radii[idx] > 0)) {
            float dx_, dy_, dz_;
            if (campos) { dx_ = m0 - campos[3 * c]; dy_ = m1 - campos[3 * c + 1]; dz_ = m2 - campos[3 * c + 2]; }
            else { dx_ = dirs[3 * idx]; dy_ = dirs[3 * idx + 1]; dz_ = dirs[3 * idx + 2]; }
            const float n = sqrtf(dx_ * dx_ + dy_ * dy_ + dz_ * dz_);
            const float inv = n > 0.f ? 1.0f / n : 0.f;
            const float x = dx_ * inv, y = dy_ * inv, z = dz_ * inv;
            float B[16], bx[16], by[16], bz[16];
            sh_basis(DEG, x, y, z, B);
            sh_basis_grad(DEG, x, y, z, bx, by, bz);
            float vx = 0.f, vy = 0.f, vz = 0.f;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < NB; ++k) acc += B[k] * co[3 * k + ch];
                const float vc = (acc + 0.5f > 0.f) ? v_colors[3 * idx + ch] : 0.f;      // clamped channels pass nothing
#pragma unroll
                for (int k = 0; k < NB; ++k) {
                    vco[3 * k + ch] += B[k] * vc;
                    const float cv = co[3 * k + ch] * vc;
                    vx += bx[k] * cv; vy += by[k] * cv; vz += bz[k] * cv;
                }
            }
            const float dotp = vx * x + vy * y + vz * z;
            vdx = (vx - dotp * x) * inv; vdy = (vy - dotp * y) * inv; vdz = (vz - dotp * z) * inv;
        }
        if (v_dirs && g < N) { v_dirs[3 * idx] = vdx; v_dirs[3 * idx + 1] = vdy; v_dirs[3 * idx + 2] = vdz; }
        if (campos) {
            // d dir / d mean = I, d dir / d campos = -I: the mean's gradient is the sum over cameras (registers), the camera
            // centre's the negative sum over Gaussians (wave sum, workgroup sum, one atomic per component and workgroup)
            vm0 += vdx; vm1 += vdy; vm2 += vdz;
            if (v_campos) {
                const float t0 = gsx_wave_sum(vdx), t1 = gsx_wave_sum(vdy), t2 = gsx_wave_sum(vdz);
                __syncthreads();
                if ((threadIdx.x & 63) == 0) { s_cam[threadIdx.x >> 6][0] = t0; s_cam[threadIdx.x >> 6][1] = t1; s_cam[threadIdx.x >> 6][2] = t2; }
                __syncthreads();
                if (threadIdx.x < 3) {
                    float tot = 0.f;
                    for (int w = 0; w < SH_BLOCK / 64; ++w) tot += s_cam[w][threadIdx.x];
                    if (tot != 0.f) atomicAdd(&v_campos[3 * c + threadIdx.x], -tot);
                }
            }
        }
    }
    if (campos && v_means && g < N) { v_means[3 * g] = vm0; v_means[3 * g + 1] = vm1; v_means[3 * g + 2] = vm2; }
    __syncthreads();                                         // every lane has taken its row: reuse the block for output
#pragma unroll
    for (int k = 0; k < NBC; ++k) s_rows[threadIdx.x * PITCH + k] = vco[k];
    __syncthreads();
    float *dst = v_coeffs + g0 * Kc * 3;
    const int total = rows * Kc * 3;
    if (Kc * 3 == NBC) {
        for (int i = threadIdx.x; i < total; i += SH_BLOCK) dst[i] = s_rows[(i / NBC) * PITCH + (i % NBC)];
    } else {
        const int rl = Kc * 3;
        for (int i = threadIdx.x; i < total; i += SH_BLOCK) {
            const int r = i / rl, col = i - r * rl;
            dst[i] = (col < NBC) ? s_rows[r * PITCH + col] : 0.f;       // bands above the active degree get no gradient
        }
    }
}

// ---- fused multi-tensor Adam --------------------------------------------------------------------------------------
constexpr int ADAM_MAX = 32;
struct AdamArgs {
    float *p[ADAM_MAX];
    const float *g[ADAM_MAX];
    float *m[ADAM_MAX];
    float *v[ADAM_MAX];
    int64_t start[ADAM_MAX + 1];  // exclusive prefix of the tensors' sizes in 4-float units (each rounded up)
    int64_t numel[ADAM_MAX];
    float step_size[ADAM_MAX];    // lr / bias_correction1 (host-step mode)
    float lr[ADAM_MAX];
    const int64_t *step_dev;      // non-null: the step lives on the device (graph-capturable)
    const int64_t *step_of[ADAM_MAX];   // per-tensor mode: each tensor's own device step counter (1-based, already
                                        // incremented for this update; tensors join the optimiser at different times)
    int per_tensor;
    int count;
    float beta1, beta2, eps, bc2_sqrt;
    // optional post-update decay of one tensor: p *= decay where decay_mask[j] > decay_min (the opacity decay of
    // gslam/backend.py:356-359 folded into the update that precedes it: one launch and one pass over the tensor less)
    int decay_k;
    int decay_min;
    float decay;
    const int32_t *decay_mask;
    // optional gate: the launch does nothing when skip_if_positive[0] > 0 (a render of this iteration overflowed its tile
    // lists on some rank: no update from truncated gradients; the host grows the lists and redoes the iteration)
    const float *skip_if_positive;
};

// One element of the update (torch.optim.Adam, lerp form of the first moment)
__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float beta1, float beta2, float eps,
                                         float step_size, float bc2_sqrt) {
    m = m + (g - m) * (1.0f - beta1);
    v = beta2 * v + (1.0f - beta2) * g * g;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - (step_size * m) / denom;
}

// A workgroup owns ADAM_UNITS consecutive 4-float units of the concatenated tensors (every tensor padded to whole
// units) and walks the tensors that overlap its range: the tensor lookup, its pointers, learning rate and bias
// corrections are workgroup-uniform (scalar registers), and a lane moves 16 bytes per access.  (The first version
// looked every element's tensor up with a per-lane binary search over the argument block and moved 4 bytes per lane:
// 21 us for the 1.5 M parameters of a 100 k map, 2 TB/s.)
// (Tried and rejected: ADAM_UNITS = 256 (slower), and issuing both of a lane's 4 x 16-byte loads ahead of the step-counter
// read and the bias corrections - 14.1 us against 12.4 us for the 1.5 M parameters.)
#ifndef GSX_ADAM_ILP
#define GSX_ADAM_ILP 1      /* 2 and 4 units in flight per thread measured slower (76 / 88 us against 71 at 7.5 M parameters) */
#endif
#ifndef GSX_ADAM_UNITS
#define GSX_ADAM_UNITS 512
#endif
typedef float gsx_f4v __attribute__((ext_vector_type(4)));
constexpr int ADAM_UNITS = GSX_ADAM_UNITS, ADAM_ILP = GSX_ADAM_ILP;

__global__ __launch_bounds__(256) void adam_multi_kernel(AdamArgs a) {
    if (a.skip_if_positive && a.skip_if_positive[0] > 0.f) return;         // wave-uniform (scalar load)
    int64_t u0 = (int64_t)blockIdx.x * ADAM_UNITS;
    const int64_t u_end = min(a.start[a.count], u0 + ADAM_UNITS);       // start[] is in units here
    if (u0 >= u_end) return;
    int k = 0;
    while (k + 1 < a.count && a.start[k + 1] <= u0) ++k;                   // uniform: first tensor of the range
    for (; u0 < u_end; ++k) {
        const int64_t seg_end = min(u_end, a.start[k + 1]);
        float bc1 = 1.0f, bc2_sqrt = a.bc2_sqrt;
        if (a.step_dev) {
            const float t = (float)(a.per_tensor ? a.step_of[k][0] : a.step_dev[0]);
            bc1 = 1.0f - powf(a.beta1, t);
            bc2_sqrt = sqrtf(1.0f - powf(a.beta2, t));
        }
        const float step_size = a.step_dev ? a.lr[k] / bc1 : a.step_size[k];
        float *p = a.p[k], *m = a.m[k], *v = a.v[k];
        const float *g = a.g[k];
        const int64_t numel = a.numel[k];
        const bool decays = (k == a.decay_k);
        const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0) &&
                         (!decays || (((uintptr_t)a.decay_mask & 15) == 0));
        // ADAM_ILP units per thread and trip: all their loads are issued before the first result is needed
        for (int64_t ub = u0 + threadIdx.x; ub < seg_end; ub += 256 * ADAM_ILP) {
            float4 pp[ADAM_ILP], mm[ADAM_ILP], vv[ADAM_ILP], gg[ADAM_ILP];
            int4 mk[ADAM_ILP];
            bool fast[ADAM_ILP];
#pragma unroll
            for (int q = 0; q < ADAM_ILP; ++q) {
                const int64_t u = ub + 256 * q;
                const int64_t j = (u - a.start[k]) * 4;
                fast[q] = u < seg_end && vec && j + 4 <= numel;
                if (fast[q]) {
                    // the moments and the gradient are touched by this kernel alone, once per iteration: streamed past the
                    // caches, so that they do not evict the map and the records the next iteration reads
                    pp[q] = *reinterpret_cast<float4 *>(p + j);
                    const gsx_f4v mn = __builtin_nontemporal_load(reinterpret_cast<gsx_f4v *>(m + j));
                    const gsx_f4v vn = __builtin_nontemporal_load(reinterpret_cast<gsx_f4v *>(v + j));
                    const gsx_f4v gn = __builtin_nontemporal_load(reinterpret_cast<const gsx_f4v *>(g + j));
                    mm[q] = make_float4(mn.x, mn.y, mn.z, mn.w); vv[q] = make_float4(vn.x, vn.y, vn.z, vn.w);
                    gg[q] = make_float4(gn.x, gn.y, gn.z, gn.w);
                    if (decays) mk[q] = *reinterpret_cast<const int4 *>(a.decay_mask + j);
                }
            }
#pragma unroll
            for (int q = 0; q < ADAM_ILP; ++q) {
                const int64_t u = ub + 256 * q;
                const int64_t j = (u - a.start[k]) * 4;
                // A unit whose gradient and moments are all zero - a Gaussian no camera of any window has seen yet, most of
                // a large map - comes out of the update exactly as it went in (p - 0 / eps, m = 0, v = 0): its 12 bytes
                // per parameter of stores are skipped.  Bit-identical by construction.
                bool idle = false;
                if (fast[q]) {
                    const float any = (fabsf(gg[q].x) + fabsf(gg[q].y)) + (fabsf(gg[q].z) + fabsf(gg[q].w)) +
                                      (fabsf(mm[q].x) + fabsf(mm[q].y)) + (fabsf(mm[q].z) + fabsf(mm[q].w)) +
                                      (vv[q].x + vv[q].y) + (vv[q].z + vv[q].w);
                    idle = any == 0.0f;
                    if (idle && decays)
                        idle = !(mk[q].x > a.decay_min || mk[q].y > a.decay_min || mk[q].z > a.decay_min || mk[q].w > a.decay_min);
                }
                if (idle) continue;
                if (fast[q]) {
                    adam_one(pp[q].x, gg[q].x, mm[q].x, vv[q].x, a.beta1, a.beta2, a.eps, step_size, bc2_sqrt);
                    adam_one(pp[q].y, gg[q].y, mm[q].y, vv[q].y, a.beta1, a.beta2, a.eps, step_size, bc2_sqrt);
                    adam_one(pp[q].z, gg[q].z, mm[q].z, vv[q].z, a.beta1, a.beta2, a.eps, step_size, bc2_sqrt);
                    adam_one(pp[q].w, gg[q].w, mm[q].w, vv[q].w, a.beta1, a.beta2, a.eps, step_size, bc2_sqrt);
                    if (decays) {
                        if (mk[q].x > a.decay_min) pp[q].x *= a.decay;
                        if (mk[q].y > a.decay_min) pp[q].y *= a.decay;
                        if (mk[q].z > a.decay_min) pp[q].z *= a.decay;
                        if (mk[q].w > a.decay_min) pp[q].w *= a.decay;
                    }
                    *reinterpret_cast<float4 *>(p + j) = pp[q];
                    __builtin_nontemporal_store(gsx_f4v{mm[q].x, mm[q].y, mm[q].z, mm[q].w}, reinterpret_cast<gsx_f4v *>(m + j));
                    __builtin_nontemporal_store(gsx_f4v{vv[q].x, vv[q].y, vv[q].z, vv[q].w}, reinterpret_cast<gsx_f4v *>(v + j));
                } else if (u < seg_end) {
                    for (int64_t e = j; e < min(numel, j + 4); ++e) {
                        float pe = p[e], me = m[e], ve = v[e];
                        adam_one(pe, g[e], me, ve, a.beta1, a.beta2, a.eps, step_size, bc2_sqrt);
                        if (decays && a.decay_mask[e] > a.decay_min) pe *= a.decay;
                        p[e] = pe; m[e] = me; v[e] = ve;
                    }
                }
            }
        }
        u0 = seg_end;
    }
}

struct CounterArgs {
    int64_t *p[16];
    int n;
    int64_t delta;
    const float *skip_if_positive;
};

__global__ void counters_add_kernel(CounterArgs a) {
    if (a.skip_if_positive && a.skip_if_positive[0] > 0.f) return;
    if (threadIdx.x < a.n) a.p[threadIdx.x][0] += a.delta;
}

// flag[0] = 1 if any of the status words has a bit of `mask` set, else 0 (a float: it travels in the step bucket and is
// summed over ranks by the iteration's one all-reduce)
__global__ void status_flag_kernel(const int32_t *__restrict__ status, int n, int mask, float *__restrict__ flag) {
    int any = 0;
    for (int i = threadIdx.x; i < n; i += 64) any |= status[i] & mask;
    // bit 2 (clamped, corrupt tile counts) weighs 1024: after the sum over <= 1023 ranks the host still tells "some rank's
    // lists overflowed" (0 < flag < 1024: grow and redo) from "some rank's counts are corrupt" (flag >= 1024: every rank raises)
    const unsigned long long over = __ballot((any & ~2) != 0), bad = __ballot((any & 2) != 0);
    if (threadIdx.x == 0) flag[0] = (over ? 1.0f : 0.0f) + (bad ? 1024.0f : 0.0f);
}

// ---- self test ------------------------------------------------------------------------------------------------------
__global__ void selftest_kernel(float *out) {
    const int lane = threadIdx.x & 63;
    const float v = (float)((lane * 37 + 11) % 101) - 50.0f + 0.25f * (float)(threadIdx.x >> 6);
    const float a = gsx_wave_sum_dpp(v);
    const float b = gsx_wave_sum_shfl(v);
    out[threadIdx.x] = a;
    out[256 + threadIdx.x] = b;
}

}  // namespace

extern "C" int gsx_sh_fwd(int degree, const float *dirs, const float *coeffs, const int32_t *radii, int64_t N,
                          int64_t C, int Kc, float *colors, void *stream) {
    GSX_CHECK_ARG(degree >= 0 && degree <= 3 && dirs && coeffs && colors && N >= 0 && C >= 1);
    GSX_CHECK_ARG((degree + 1) * (degree + 1) <= Kc);
    if (N == 0) return GSX_OK;
    const dim3 grid((unsigned)((N + SH_BLOCK - 1) / SH_BLOCK)), block(SH_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    switch (degree) {
    case 0: hipLaunchKernelGGL(sh_fwd_kernel<0>, grid, block, 0, st, dirs, coeffs, radii, N, (int)C, Kc, colors); break;
    case 1: hipLaunchKernelGGL(sh_fwd_kernel<1>, grid, block, 0, st, dirs, coeffs, radii, N, (int)C, Kc, colors); break;
    case 2: hipLaunchKernelGGL(sh_fwd_kernel<2>, grid, block, 0, st, dirs, coeffs, radii, N, (int)C, Kc, colors); break;
    default: hipLaunchKernelGGL(sh_fwd_kernel<3>, grid, block, 0, st, dirs, coeffs, radii, N, (int)C, Kc, colors); break;
    }
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_sh_bwd(int degree, const float *dirs, const float *coeffs, const int32_t *radii,
                          const float *v_colors, int64_t N, int64_t C, int Kc, float *v_coeffs, float *v_dirs,
                          void *stream) {
    GSX_CHECK_ARG(degree >= 0 && degree <= 3 && dirs && coeffs && v_colors && v_coeffs && N >= 0 && C >= 1);
    GSX_CHECK_ARG((degree + 1) * (degree + 1) <= Kc);
    if (N == 0) return GSX_OK;
    const dim3 grid((unsigned)((N + SH_BLOCK - 1) / SH_BLOCK)), block(SH_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define GSX_SH_BWD(D)                                                                                                  \
    hipLaunchKernelGGL(sh_bwd_kernel<D>, grid, block, 0, st, dirs, coeffs, radii, v_colors, N, (int)C, Kc, v_coeffs,   \
                       v_dirs)
    switch (degree) {
    case 0: GSX_SH_BWD(0); break;
    case 1: GSX_SH_BWD(1); break;
    case 2: GSX_SH_BWD(2); break;
    default: GSX_SH_BWD(3); break;
    }
#undef GSX_SH_BWD
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_sh_fwd_means(int degree, const float *means, const float *campos, const float *coeffs,
                                const int32_t *radii, int64_t N, int64_t C, int Kc, float *colors, void *stream) {
    GSX_CHECK_ARG(degree >= 0 && degree <= 3 && means && campos && coeffs && colors && N >= 0 && C >= 1);
    GSX_CHECK_ARG((degree + 1) * (degree + 1) <= Kc);
    if (N == 0) return GSX_OK;
    const dim3 grid((unsigned)((N + SH_BLOCK - 1) / SH_BLOCK)), block(SH_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    switch (degree) {
    case 0: hipLaunchKernelGGL(sh_fwd_kernel<0>, grid, block, 0, st, means, coeffs, radii, N, (int)C, Kc, colors, campos); break;
    case 1: hipLaunchKernelGGL(sh_fwd_kernel<1>, grid, block, 0, st, means, coeffs, radii, N, (int)C, Kc, colors, campos); break;
    case 2: hipLaunchKernelGGL(sh_fwd_kernel<2>, grid, block, 0, st, means, coeffs, radii, N, (int)C, Kc, colors, campos); break;
    default: hipLaunchKernelGGL(sh_fwd_kernel<3>, grid, block, 0, st, means, coeffs, radii, N, (int)C, Kc, colors, campos); break;
    }
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_sh_bwd_means(int degree, const float *means, const float *campos, const float *coeffs,
                                const int32_t *radii, const float *v_colors, int64_t N, int64_t C, int Kc,
                                float *v_coeffs, float *v_means, float *v_campos, void *stream) {
    GSX_CHECK_ARG(degree >= 0 && degree <= 3 && means && campos && coeffs && v_colors && v_coeffs && N >= 0 && C >= 1);
    GSX_CHECK_ARG((degree + 1) * (degree + 1) <= Kc);
    if (N == 0) return GSX_OK;
    const dim3 grid((unsigned)((N + SH_BLOCK - 1) / SH_BLOCK)), block(SH_BLOCK);
    hipStream_t st = (hipStream_t)stream;
#define GSX_SH_BWDM(D)                                                                                                 \
    hipLaunchKernelGGL(sh_bwd_kernel<D>, grid, block, 0, st, means, coeffs, radii, v_colors, N, (int)C, Kc, v_coeffs,  \
                       (float *)nullptr, campos, v_means, v_campos)
    switch (degree) {
    case 0: GSX_SH_BWDM(0); break;
    case 1: GSX_SH_BWDM(1); break;
    case 2: GSX_SH_BWDM(2); break;
    default: GSX_SH_BWDM(3); break;
    }
#undef GSX_SH_BWDM
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

static int adam_launch(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                       float *const *exp_avg_sq, const int64_t *numels, const float *lrs, float beta1, float beta2,
                       float eps, int64_t step_host, const int64_t *step_dev, const int64_t *const *steps, void *stream,
                       int decay_tensor = -1, const int32_t *decay_mask = nullptr, int decay_min = 0,
                       float decay = 1.0f, const float *skip_if_positive = nullptr) {
    GSX_CHECK_ARG(n_tensors >= 1 && n_tensors <= ADAM_MAX && params && grads && exp_avg && exp_avg_sq && numels && lrs);
    GSX_CHECK_ARG(decay_tensor < n_tensors && (decay_tensor < 0 || decay_mask));
    GSX_CHECK_ARG(step_host >= 1 || step_dev || steps);
    AdamArgs a;
    a.count = n_tensors;
    a.start[0] = 0;
    a.step_dev = steps ? steps[0] : step_dev;
    a.per_tensor = steps ? 1 : 0;
    const double sh = (double)(step_host >= 1 ? step_host : 1);
    const double bc1 = 1.0 - pow((double)beta1, sh), bc2 = 1.0 - pow((double)beta2, sh);
    for (int k = 0; k < ADAM_MAX; ++k) {
        const bool in = k < n_tensors;
        a.step_of[k] = (steps && in) ? steps[k] : nullptr;
        if (steps && in) GSX_CHECK_ARG(steps[k]);
        a.p[k] = in ? params[k] : nullptr; a.g[k] = in ? grads[k] : nullptr;
        a.m[k] = in ? exp_avg[k] : nullptr; a.v[k] = in ? exp_avg_sq[k] : nullptr;
        // every tensor starts on a workgroup boundary: a workgroup then serves ONE tensor.  With the tensors packed back to
        // back the workgroup at the end of the list walked the (dt, dR) pairs of all window poses - 16 tensors of 3 and 6
        // floats - one after the other, each with its own dependent chain step counter -> powf -> loads -> stores, and the
        // launch lasted as long as that one workgroup: 69 us for 7.5 M parameters that stream in ~35 (the same kernel runs
        // at 6.8 TB/s on one large tensor: tools/ubench/adam_variants.hip)
        const int64_t units = in ? (numels[k] + 3) / 4 : 0;
        a.start[k + 1] = a.start[k] + (units + ADAM_UNITS - 1) / ADAM_UNITS * ADAM_UNITS;
        a.numel[k] = in ? numels[k] : 0;
        a.step_size[k] = in ? (float)((double)lrs[k] / bc1) : 0.f;
        a.lr[k] = in ? lrs[k] : 0.f;
        if (in) GSX_CHECK_ARG(numels[k] >= 1 && params[k] && grads[k] && exp_avg[k] && exp_avg_sq[k]);
    }
    a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.bc2_sqrt = (float)sqrt(bc2);
    a.decay_k = decay_tensor < 0 ? -1 : decay_tensor; a.decay_mask = decay_mask; a.decay_min = decay_min; a.decay = decay;
    a.skip_if_positive = skip_if_positive;
    const int64_t total = a.start[n_tensors];
    const int64_t blocks = (total + ADAM_UNITS - 1) / ADAM_UNITS;
    GSX_CHECK_ARG(blocks < ((int64_t)1 << 31));
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_adam_multi(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                              float *const *exp_avg_sq, const int64_t *numels, const float *lrs, float beta1,
                              float beta2, float eps, int64_t step_host, const int64_t *step_dev, void *stream) {
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numels, lrs, beta1, beta2, eps, step_host,
                       step_dev, nullptr, stream);
}

extern "C" int gsx_adam_multi_steps(int n_tensors, float *const *params, const float *const *grads,
                                    float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numels,
                                    const float *lrs, float beta1, float beta2, float eps,
                                    const int64_t *const *steps, void *stream) {
    GSX_CHECK_ARG(steps);
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numels, lrs, beta1, beta2, eps, 0, nullptr, steps,
                       stream);
}

extern "C" int gsx_adam_multi_steps_decay(int n_tensors, float *const *params, const float *const *grads,
                                          float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numels,
                                          const float *lrs, float beta1, float beta2, float eps,
                                          const int64_t *const *steps, int decay_tensor, const int32_t *decay_mask,
                                          int decay_min_count, float decay, void *stream) {
    GSX_CHECK_ARG(steps);
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numels, lrs, beta1, beta2, eps, 0, nullptr, steps,
                       stream, decay_tensor, decay_mask, decay_min_count, decay);
}

extern "C" int gsx_adam_multi_steps_gated(int n_tensors, float *const *params, const float *const *grads,
                                          float *const *exp_avg, float *const *exp_avg_sq, const int64_t *numels,
                                          const float *lrs, float beta1, float beta2, float eps,
                                          const int64_t *const *steps, int decay_tensor, const int32_t *decay_mask,
                                          int decay_min_count, float decay, const float *skip_if_positive,
                                          void *stream) {
    GSX_CHECK_ARG(steps);
    return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, numels, lrs, beta1, beta2, eps, 0, nullptr, steps,
                       stream, decay_tensor, decay_mask, decay_min_count, decay, skip_if_positive);
}

// ---- staging copies of the ranged gradient / parameter exchange (gslam_amd.dist.StepBucket) ---------------------
// The map's arrays are tensor-major in one flat buffer; the exchange of ONE Gaussian range wants, per rank, one contiguous
// block [ its part of array 0 | its part of array 1 | ... ].  slice t = `parts` consecutive parts of part_len[t] floats at
// slices[t]; the staging buffer holds `parts` blocks of L = sum part_len floats.  to_flat = 0: slices -> staging (before a
// reduce-scatter, or with parts = 1 the owner's block before an all-gather); 1: staging -> slices (after the all-gather).
namespace {
constexpr int RANGE_COPY_MAX = 8;
struct RangeCopyArgs {
    float *slice[RANGE_COPY_MAX];
    int64_t part_quads[RANGE_COPY_MAX];     // part_len / 4
    int64_t off_quads[RANGE_COPY_MAX + 1];  // prefix sums of part_quads
    int n;
    int64_t total_quads;                    // parts * off_quads[n]
};

template <bool TO_FLAT>
__global__ __launch_bounds__(256) void range_copy_kernel(RangeCopyArgs a, float4 *__restrict__ staging) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.total_quads) return;
    const int64_t lq = a.off_quads[a.n];
    const int64_t r = i / lq, j = i - r * lq;
    int t = 0;
#pragma unroll
    for (int k = 1; k < RANGE_COPY_MAX; ++k) t += (k < a.n && j >= a.off_quads[k]) ? 1 : 0;
    float4 *f = reinterpret_cast<float4 *>(a.slice[t]) + r * a.part_quads[t] + (j - a.off_quads[t]);
    if (TO_FLAT) *f = staging[i];
    else staging[i] = *f;
}
}  // namespace

extern "C" int gsx_range_copy(int n_slices, float *const *slices, const int64_t *part_len, int parts, float *staging,
                              int to_flat, void *stream) {
    GSX_CHECK_ARG(n_slices >= 1 && n_slices <= RANGE_COPY_MAX && slices && part_len && parts >= 1 && staging);
    GSX_CHECK_ARG((((uintptr_t)staging) & 15) == 0);
    RangeCopyArgs a;
    a.n = n_slices;
    a.off_quads[0] = 0;
    for (int t = 0; t < RANGE_COPY_MAX; ++t) {
        const bool in = t < n_slices;
        if (in) GSX_CHECK_ARG(slices[t] && part_len[t] >= 0 && part_len[t] % 4 == 0 && (((uintptr_t)slices[t]) & 15) == 0);
        a.slice[t] = in ? slices[t] : nullptr;
        a.part_quads[t] = in ? part_len[t] / 4 : 0;
        a.off_quads[t + 1] = a.off_quads[t] + a.part_quads[t];
    }
    a.total_quads = (int64_t)parts * a.off_quads[n_slices];
    if (a.total_quads == 0) return GSX_OK;
    GSX_CHECK_ARG(a.total_quads < ((int64_t)1 << 31) * 256);
    const unsigned blocks = (unsigned)((a.total_quads + 255) / 256);
    if (to_flat)
        hipLaunchKernelGGL(range_copy_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, (float4 *)staging);
    else
        hipLaunchKernelGGL(range_copy_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, (float4 *)staging);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_status_flag(const int32_t *status, int n, int mask, float *flag, void *stream) {
    GSX_CHECK_ARG(status && flag && n >= 1 && n <= 4096);
    hipLaunchKernelGGL(status_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, status, n, mask, flag);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

static int counters_launch(int n, int64_t *const *counters, int64_t delta, const float *skip_if_positive, void *stream);

extern "C" int gsx_counters_add_gated(int n, int64_t *const *counters, int64_t delta, const float *skip_if_positive,
                                      void *stream) {
    return counters_launch(n, counters, delta, skip_if_positive, stream);
}

extern "C" int gsx_counters_add(int n, int64_t *const *counters, int64_t delta, void *stream) {
    return counters_launch(n, counters, delta, nullptr, stream);
}

static int counters_launch(int n, int64_t *const *counters, int64_t delta, const float *skip_if_positive, void *stream) {
    GSX_CHECK_ARG(n >= 1 && n <= 16 && counters);
    CounterArgs a;
    a.n = n;
    a.delta = delta;
    a.skip_if_positive = skip_if_positive;
    for (int k = 0; k < 16; ++k) {
        a.p[k] = k < n ? counters[k] : nullptr;
        if (k < n) GSX_CHECK_ARG(counters[k]);
    }
    hipLaunchKernelGGL(counters_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_selftest(void *scratch, int64_t scratch_bytes, void *stream) {
    GSX_CHECK_ARG(scratch && scratch_bytes >= 65536);
    hipStream_t st = (hipStream_t)stream;
    float *d = (float *)scratch;
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(256), 0, st, d);
    GSX_CHECK_LAUNCH();
    float h[512];
    if (hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        gsx_set_error("gsx_selftest: copy back failed");
        return GSX_E_LAUNCH;
    }
    for (int w = 0; w < 4; ++w) {
        double ref = 0.0;
        for (int l = 0; l < 64; ++l) ref += (double)((l * 37 + 11) % 101) - 50.0 + 0.25 * w;
        for (int l = 0; l < 64; ++l) {
            if (fabs(h[w * 64 + l] - ref) > 1e-3 || fabs(h[256 + w * 64 + l] - ref) > 1e-3) {
                gsx_set_error("gsx_selftest: wave_sum mismatch wave %d lane %d: dpp=%f shfl=%f ref=%f", w, l,
                              h[w * 64 + l], h[256 + w * 64 + l], ref);
                return GSX_E_LAUNCH;
            }
        }
    }
    return GSX_OK;
}
