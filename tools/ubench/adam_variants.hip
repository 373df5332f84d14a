// Which shape of the fused Adam launch streams best on MI355X?  One tensor of n floats (p, g, m, v: 28 B per element moved),
// n large enough that the four arrays do not fit the Infinity Cache (the BA iteration touches ~2 GB between two updates).
//   v0  the shipped shape: a workgroup owns 512 consecutive 16-byte units, 2 units per thread, bias corrections from
//       powf / sqrtf evaluated by every thread, non-temporal loads of g / m / v and stores of m / v
//   v1  v0 with the bias corrections passed in (no transcendental work in the kernel)
//   v2  v1 with plain (default policy) loads and stores
//   v3  grid-stride: 2048 workgroups, each thread walks the units with a stride of the grid, one unit in flight
//   v4  v3 with two units in flight per thread
//   v5  v1 with 1024 units per workgroup (4 per thread, all loads issued first)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/adam_variants tools/ubench/adam_variants.hip ; run it on the GPU box
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, float b1, float b2, float eps, float ss, float bc2s) {
    m = m + (g - m) * (1.0f - b1);
    v = b2 * v + (1.0f - b2) * g * g;
    const float denom = sqrtf(v) / bc2s + eps;
    p = p - (ss * m) / denom;
}

template <bool NT>
__device__ __forceinline__ f4 ld(const float *p) {
    if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
    return *reinterpret_cast<const f4 *>(p);
}
template <bool NT>
__device__ __forceinline__ void st(float *p, f4 v) {
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(p));
    else *reinterpret_cast<f4 *>(p) = v;
}

__device__ __forceinline__ void update(f4 &p, f4 g, f4 &m, f4 &v, float ss, float bc2s) {
    float pe[4] = {p.x, p.y, p.z, p.w}, me[4] = {m.x, m.y, m.z, m.w}, ve[4] = {v.x, v.y, v.z, v.w};
    const float ge[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) adam_one(pe[k], ge[k], me[k], ve[k], 0.9f, 0.999f, 1e-8f, ss, bc2s);
    p = f4{pe[0], pe[1], pe[2], pe[3]}; m = f4{me[0], me[1], me[2], me[3]}; v = f4{ve[0], ve[1], ve[2], ve[3]};
}

template <int UNITS, bool POW, bool NT>
__global__ __launch_bounds__(256) void adam_block(float *p, const float *g, float *m, float *v, int64_t n_units,
                                                  const int64_t *step, float lr, float ss_in, float bc2s_in) {
    const int64_t u0 = (int64_t)blockIdx.x * UNITS;
    float ss = ss_in, bc2s = bc2s_in;
    if (POW) {
        const float t = (float)step[0];
        ss = lr / (1.0f - powf(0.9f, t));
        bc2s = sqrtf(1.0f - powf(0.999f, t));
    }
    constexpr int PER = UNITS / 256;
    f4 pp[PER], gg[PER], mm[PER], vv[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int64_t u = u0 + threadIdx.x + 256 * q;
        if (u < n_units) {
            pp[q] = *reinterpret_cast<const f4 *>(p + 4 * u);
            mm[q] = ld<NT>(m + 4 * u); vv[q] = ld<NT>(v + 4 * u); gg[q] = ld<NT>(g + 4 * u);
        }
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int64_t u = u0 + threadIdx.x + 256 * q;
        if (u < n_units) {
            update(pp[q], gg[q], mm[q], vv[q], ss, bc2s);
            *reinterpret_cast<f4 *>(p + 4 * u) = pp[q];
            st<NT>(m + 4 * u, mm[q]); st<NT>(v + 4 * u, vv[q]);
        }
    }
}

template <int ILP, bool NT>
__global__ __launch_bounds__(256) void adam_stride(float *p, const float *g, float *m, float *v, int64_t n_units, float ss,
                                                   float bc2s) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n_units; u += stride * ILP) {
        f4 pp[ILP], gg[ILP], mm[ILP], vv[ILP];
#pragma unroll
        for (int q = 0; q < ILP; ++q) {
            const int64_t w = u + q * stride;
            if (w < n_units) {
                pp[q] = *reinterpret_cast<const f4 *>(p + 4 * w);
                mm[q] = ld<NT>(m + 4 * w); vv[q] = ld<NT>(v + 4 * w); gg[q] = ld<NT>(g + 4 * w);
            }
        }
#pragma unroll
        for (int q = 0; q < ILP; ++q) {
            const int64_t w = u + q * stride;
            if (w < n_units) {
                update(pp[q], gg[q], mm[q], vv[q], ss, bc2s);
                *reinterpret_cast<f4 *>(p + 4 * w) = pp[q];
                st<NT>(m + 4 * w, mm[q]); st<NT>(v + 4 * w, vv[q]);
            }
        }
    }
}

__global__ void fill(float *p, int64_t n, float s) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        p[i] = s * (float)((i * 2654435761ull) % 1000) * 1e-3f;
}

template <typename F>
static double time_us(F launch, int reps) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / reps;
}

int main() {
    const int64_t n = 40ll * 1000 * 1000;                  // 160 MB per array, 640 MB in all: beyond the Infinity Cache
    const int64_t nu = n / 4;
    float *p, *g, *m, *v;
    int64_t *step;
    CHECK(hipMalloc(&p, n * 4)); CHECK(hipMalloc(&g, n * 4)); CHECK(hipMalloc(&m, n * 4)); CHECK(hipMalloc(&v, n * 4));
    CHECK(hipMalloc(&step, 8));
    const int64_t one = 7;
    CHECK(hipMemcpy(step, &one, 8, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, p, n, 1.0f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, g, n, 0.01f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, m, n, 0.001f);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, v, n, 0.0001f);
    CHECK(hipDeviceSynchronize());
    const float ss = 1e-3f / (1.0f - powf(0.9f, 7.f)), bc2s = sqrtf(1.0f - powf(0.999f, 7.f));
    const double bytes = 28.0 * n;
    auto report = [&](const char *name, double us) { printf("%-44s %8.1f us  %6.2f TB/s\n", name, us, bytes / us * 1e-6); };
    const int reps = 20;
    const unsigned b512 = (unsigned)((nu + 511) / 512), b1024 = (unsigned)((nu + 1023) / 1024), b256 = (unsigned)((nu + 255) / 256);
    report("v0 block 512 units, powf in kernel, nt", time_us([&] { hipLaunchKernelGGL((adam_block<512, true, true>), dim3(b512), dim3(256), 0, 0, p, g, m, v, nu, step, 1e-3f, ss, bc2s); }, reps));
    report("v1 block 512 units, bias passed in, nt", time_us([&] { hipLaunchKernelGGL((adam_block<512, false, true>), dim3(b512), dim3(256), 0, 0, p, g, m, v, nu, step, 1e-3f, ss, bc2s); }, reps));
    report("v2 block 512 units, bias passed in, plain", time_us([&] { hipLaunchKernelGGL((adam_block<512, false, false>), dim3(b512), dim3(256), 0, 0, p, g, m, v, nu, step, 1e-3f, ss, bc2s); }, reps));
    report("v2b block 256 units (1 per thread), nt", time_us([&] { hipLaunchKernelGGL((adam_block<256, false, true>), dim3(b256), dim3(256), 0, 0, p, g, m, v, nu, step, 1e-3f, ss, bc2s); }, reps));
    report("v5 block 1024 units (4 per thread), nt", time_us([&] { hipLaunchKernelGGL((adam_block<1024, false, true>), dim3(b1024), dim3(256), 0, 0, p, g, m, v, nu, step, 1e-3f, ss, bc2s); }, reps));
    for (unsigned grid : {1024u, 2048u, 4096u, 8192u}) {
        char nm[96];
        snprintf(nm, sizeof nm, "v3 grid-stride %u WGs, 1 in flight, nt", grid);
        report(nm, time_us([&] { hipLaunchKernelGGL((adam_stride<1, true>), dim3(grid), dim3(256), 0, 0, p, g, m, v, nu, ss, bc2s); }, reps));
        snprintf(nm, sizeof nm, "v4 grid-stride %u WGs, 2 in flight, nt", grid);
        report(nm, time_us([&] { hipLaunchKernelGGL((adam_stride<2, true>), dim3(grid), dim3(256), 0, 0, p, g, m, v, nu, ss, bc2s); }, reps));
        snprintf(nm, sizeof nm, "v4p grid-stride %u WGs, 2 in flight, plain", grid);
        report(nm, time_us([&] { hipLaunchKernelGGL((adam_stride<2, false>), dim3(grid), dim3(256), 0, 0, p, g, m, v, nu, ss, bc2s); }, reps));
    }
    return 0;
}
