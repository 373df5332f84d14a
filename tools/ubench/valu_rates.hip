// Micro-benchmark: issue cost of the VALU instruction kinds the raster kernels use, at 1 / 2 / 4 / 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates.hip -o /tmp/valu_rates ; run on the GPU box
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define REP 64
#define ITERS 2000

template <int KIND>
__global__ void kern(float *out, unsigned long long *cyc, float seed) {
    float v[8];
    for (int k = 0; k < 8; ++k) v[k] = seed + threadIdx.x * 0.001f + k;
    float a = seed * 0.5f, b = seed * 0.25f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
                if (KIND == 1) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(v[k]));
                if (KIND == 2) asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[k]));
                if (KIND == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(v[k]));
                if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a));
                if (KIND == 5) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[k]), "v"(a) : "vcc");
                if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
                if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&v[k & 6]) : "v"(*(double *)&a));
                if (KIND == 8) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[k]) : "s20");
                if (KIND == 9) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[k]));
                if (KIND == 10) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[k]) : "v"(v[(k + 1) & 7]));
                if (KIND == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[k]) : "v"(a) : "s20", "s21");
                if (KIND == 12) asm volatile("v_cndmask_b32_e64 %0, %1, %2, s[20:21]" : "=v"(v[k]) : "v"(a), "v"(b) : "s20", "s21");
                if (KIND == 13) asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[k]) : "v"(a), "v"(b) : "s20", "s21");
                if (KIND == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
                if (KIND == 15) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(v[k]), "v"(a) : "s20", "s21");
                if (KIND == 16) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[k]));
                if (KIND == 17) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(v[k]) : "v"(a), "v"(b));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += v[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 8192 * 256 * sizeof(float));
    hipMalloc(&cyc, 8192 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-28s", name);
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;              // 256-thread blocks: wps blocks per CU = wps waves per SIMD
        hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(kern<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[256];
        hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
        const double per_wave = avg / (double)(ITERS * REP);
        // wall-clock: wave-instructions per SIMD per second -> cycles per wave-instruction per SIMD at 2.4 GHz
        const double winstr = (double)blocks * 4.0 * ITERS * REP * 20.0;      // 4 waves per block
        const double per_simd_s = winstr / 1024.0 / (ms * 1e-3);
        printf("  wps=%d: %5.2f cyc/instr/wave, wall %5.2f cyc/instr/SIMD@2.4GHz", wps, per_wave, 2.4e9 / per_simd_s);
    }
    printf("\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fma_f32");
    run<6>("v_mul_f32");
    run<1>("v_add_f32_dpp row_shr:1");
    run<2>("v_add_f32_dpp row_bcast:15");
    run<9>("v_add_f32_dpp quad_perm");
    run<10>("v_mov_b32_dpp row_shr:1");
    run<3>("v_exp_f32");
    run<4>("v_cndmask_b32 (vcc)");
    run<5>("v_cmp_lt_f32 (vcc)");
    run<7>("v_pk_fma_f32");
    run<8>("v_readlane_b32");
    run<11>("v_cndmask e64 sgpr (dep)");
    run<12>("v_cndmask e64 sgpr (indep)");
    run<17>("v_cndmask vcc (indep)");
    run<13>("v_cmp_e64 + v_cndmask_e64");
    run<15>("v_cmp_lt_f32_e64 sgpr");
    run<14>("v_max_f32");
    run<16>("v_rcp_f32");
    return 0;
}
