// Micro-benchmark (round 4): issue cost of EVERY instruction kind the rasteriser loops use or could use, per SIMD, at the
// occupancies the kernels run at (4 / 5 / 8 wavefronts per SIMD), plus the mixes that decide the round-4 rewrite:
// exec-masked updates instead of compare + select, f32 MFMA beside VALU, LDS stores / broadcast reads beside VALU.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_rates2.hip -o tools/ubench/valu_rates2 ; run on the GPU box
// output: one line per kind: cycles per instruction per SIMD at 2.4 GHz (wall clock), cycles per instruction of one wave
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

#define ITERS 1500
typedef float f4 __attribute__((ext_vector_type(4)));

// BODY is one "unit" written against v[0..7] (floats), a, b (floats), iv[0..7] via bit casts; UNITS = instructions per body
#define KERNEL(NAME, UNITS, ...)                                                                                  \
    __global__ __launch_bounds__(256) void k_##NAME(float *out, unsigned long long *cyc, float seed) {            \
        __shared__ float lds[4096];                                                                               \
        float v[8];                                                                                               \
        for (int k = 0; k < 8; ++k) v[k] = seed + threadIdx.x * 0.001f + k;                                       \
        float a = seed * 0.5f, b = seed * 0.25f;                                                                  \
        f4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};                                               \
        const unsigned la = (threadIdx.x & 63) * 4u + (threadIdx.x >> 6) * 1024u;                                 \
        lds[threadIdx.x] = seed; (void)la; (void)acc2;                                                            \
        __syncthreads();                                                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                               \
        for (int it = 0; it < ITERS; ++it) {                                                                      \
            _Pragma("unroll") for (int r = 0; r < 8; ++r) { __VA_ARGS__ }                                         \
        }                                                                                                         \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                               \
        float s = acc.x + acc.y + acc.z + acc.w + acc2.x;                                                         \
        for (int k = 0; k < 8; ++k) s += v[k];                                                                    \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + a + b + lds[(threadIdx.x * 7) & 4095];                   \
        if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;                                                          \
    }                                                                                                             \
    static const int units_##NAME = (UNITS) * 8;

#define A1(INS) asm volatile(INS : "+v"(v[r]) : "v"(a), "v"(b));
#define A1K(INS, K) asm volatile(INS : "+v"(v[(r + K) & 7]) : "v"(a), "v"(b));

KERNEL(fma, 1, A1("v_fma_f32 %0, %0, %1, %2"))
KERNEL(fmac, 1, A1("v_fmac_f32 %0, %1, %2"))
KERNEL(mul, 1, A1("v_mul_f32 %0, %0, %1"))
KERNEL(add, 1, A1("v_add_f32 %0, %0, %1"))
KERNEL(sub, 1, A1("v_sub_f32 %0, %0, %1"))
KERNEL(fma_neg, 1, A1("v_fma_f32 %0, -%0, %1, %2"))
KERNEL(fma_lit, 1, A1("v_fma_f32 %0, %0, %1, 1.0"))
KERNEL(mul_sgpr, 1, asm volatile("v_mul_f32 %0, s20, %0" : "+v"(v[r]) : : "s20");)
KERNEL(mov, 1, asm volatile("v_mov_b32 %0, %1" : "=v"(v[r]) : "v"(a));)
KERNEL(mov_lit, 1, asm volatile("v_mov_b32 %0, 0" : "=v"(v[r]));)
KERNEL(min, 1, A1("v_min_f32 %0, %0, %1"))
KERNEL(max, 1, A1("v_max_f32 %0, %0, %1"))
KERNEL(med3, 1, A1("v_med3_f32 %0, %0, %1, %2"))
KERNEL(max3, 1, A1("v_max3_f32 %0, %0, %1, %2"))
KERNEL(bfi, 1, A1("v_bfi_b32 %0, %1, %0, %2"))
KERNEL(and_b32, 1, A1("v_and_b32 %0, %0, %1"))
KERNEL(or_b32, 1, A1("v_or_b32 %0, %0, %1"))
KERNEL(lshl, 1, A1("v_lshlrev_b32 %0, 1, %0"))
KERNEL(add_u32, 1, A1("v_add_u32 %0, %0, %1"))
KERNEL(sub_u32, 1, A1("v_sub_u32 %0, %0, %1"))
KERNEL(mad_u24, 1, A1("v_mad_u32_u24 %0, %0, 48, %1"))
KERNEL(mul_lo, 1, A1("v_mul_lo_u32 %0, %0, %1"))
KERNEL(cvt_f2i, 1, A1("v_cvt_i32_f32 %0, %0"))
KERNEL(cvt_i2f, 1, A1("v_cvt_f32_i32 %0, %0"))
KERNEL(exp, 1, A1("v_exp_f32 %0, %0"))
KERNEL(exp_neg, 1, A1("v_exp_f32 %0, -%0"))
KERNEL(rcp, 1, A1("v_rcp_f32 %0, %0"))
KERNEL(log, 1, A1("v_log_f32 %0, %0"))
KERNEL(rsq, 1, A1("v_rsq_f32 %0, %0"))
KERNEL(cmp_vcc, 1, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[r]), "v"(a) : "vcc");)
KERNEL(cmp_e64, 1, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(v[r]), "v"(a) : "s20", "s21");)
KERNEL(cmp_i32, 1, asm volatile("v_cmp_le_i32 vcc, %0, %1" : : "v"(v[r]), "v"(a) : "vcc");)
KERNEL(cmp_lit, 1, asm volatile("v_cmp_lt_f32 vcc, 0x3b808081, %0" : : "v"(v[r]) : "vcc");)
KERNEL(cmpx, 1, asm volatile("v_cmpx_gt_f32 exec, %0, %1" : : "v"(v[r]), "v"(b) : "exec");)
KERNEL(cndmask_e64, 1, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[r]) : "v"(a) : "s20", "s21");)
// the pair as the compiler emits it: compare into vcc, the hazard pad, select on vcc
KERNEL(cmp_nop_cnd, 2, asm volatile("v_cmp_lt_f32 vcc, %1, %2\n\ts_nop 1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[r]) : "v"(a), "v"(b) : "vcc");)
KERNEL(cmp_cnd_e64, 2, asm volatile("v_cmp_lt_f32_e64 s[20:21], %1, %2\n\ts_nop 1\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(v[r]) : "v"(a), "v"(b) : "s20", "s21");)
// exec-masked update in place of the select: compare, s_and_saveexec, two VALU ops under the mask, restore
KERNEL(saveexec2, 3, asm volatile("v_cmp_gt_f32 vcc, %2, %3\n\ts_and_saveexec_b64 s[20:21], vcc\n\tv_fma_f32 %0, %0, %2, %3\n\tv_mul_f32 %1, %1, %2\n\ts_mov_b64 exec, s[20:21]" : "+v"(v[r]), "+v"(v[(r + 1) & 7]) : "v"(a), "v"(b) : "vcc", "scc", "s20", "s21");)
KERNEL(pk_fma, 1, asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double *)&v[r & 6]) : "v"(*(double *)&a));)
KERNEL(pk_mul, 1, asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&v[r & 6]) : "v"(*(double *)&a));)
KERNEL(pk_add, 1, asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&v[r & 6]) : "v"(*(double *)&a));)
KERNEL(dpp_add_row_shr, 1, A1("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"))
KERNEL(dpp_mul_row_ror, 1, A1("v_mul_f32_dpp %0, %0, %1 row_ror:1 row_mask:0xf bank_mask:0xf"))
KERNEL(dpp_mov_quad, 1, asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));)
KERNEL(dpp_mov_wave_ror, 1, asm volatile("v_mov_b32_dpp %0, %1 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(v[r]) : "v"(v[(r + 1) & 7]));)
KERNEL(permlane32_swap, 1, asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 1) & 7]));)
KERNEL(permlane16_swap, 1, asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(v[r]), "+v"(v[(r + 1) & 7]));)
KERNEL(readlane, 1, asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(v[r]) : "s20");)
KERNEL(readfirstlane, 1, asm volatile("v_readfirstlane_b32 s20, %0" : : "v"(v[r]) : "s20");)
KERNEL(s_nop0, 1, asm volatile("s_nop 0");)
KERNEL(s_nop1, 1, asm volatile("s_nop 1");)
// matrix core beside the vector ALU: what does an f32 MFMA cost the VALU stream of the same / other waves?
KERNEL(mfma16_indep, 1, asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(r & 1 ? acc2 : acc) : "v"(a), "v"(b));)
KERNEL(mfma16_dep, 1, asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));)
KERNEL(mfma16_fma4, 5, asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); A1K("v_fma_f32 %0, %0, %1, %2", 0) A1K("v_fma_f32 %0, %0, %1, %2", 1) A1K("v_fma_f32 %0, %0, %1, %2", 2) A1K("v_fma_f32 %0, %0, %1, %2", 3))
KERNEL(mfma16_fma12, 13, asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); A1K("v_fma_f32 %0, %0, %1, %2", 0) A1K("v_fma_f32 %0, %0, %1, %2", 1) A1K("v_fma_f32 %0, %0, %1, %2", 2) A1K("v_fma_f32 %0, %0, %1, %2", 3) A1K("v_fma_f32 %0, %0, %1, %2", 4) A1K("v_fma_f32 %0, %0, %1, %2", 5) A1K("v_fma_f32 %0, %0, %1, %2", 6) A1K("v_fma_f32 %0, %0, %1, %2", 7) A1K("v_fma_f32 %0, %0, %1, %2", 0) A1K("v_fma_f32 %0, %0, %1, %2", 1) A1K("v_fma_f32 %0, %0, %1, %2", 2) A1K("v_fma_f32 %0, %0, %1, %2", 3))
KERNEL(mfma16_fma28, 29, asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b)); for (int q = 0; q < 28; ++q) { A1K("v_fma_f32 %0, %0, %1, %2", q) })
// LDS instructions beside VALU: do they take VALU issue time?
KERNEL(fma4_dswrite, 5, asm volatile("ds_write_b32 %0, %1" : : "v"(la), "v"(v[r]) : "memory"); A1K("v_fma_f32 %0, %0, %1, %2", 0) A1K("v_fma_f32 %0, %0, %1, %2", 1) A1K("v_fma_f32 %0, %0, %1, %2", 2) A1K("v_fma_f32 %0, %0, %1, %2", 3))
KERNEL(fma12_dswrite, 13, asm volatile("ds_write_b32 %0, %1" : : "v"(la), "v"(v[r]) : "memory"); for (int q = 0; q < 12; ++q) { A1K("v_fma_f32 %0, %0, %1, %2", q) })
KERNEL(fma12_dsread128, 13, { f4 t; asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(0u) : "memory"); acc += t; } for (int q = 0; q < 11; ++q) { A1K("v_fma_f32 %0, %0, %1, %2", q) })
KERNEL(fma12_only, 12, for (int q = 0; q < 12; ++q) { A1K("v_fma_f32 %0, %0, %1, %2", q) })
// mixes: are class costs additive?
KERNEL(mix_fma_cmp, 2, A1("v_fma_f32 %0, %0, %1, %2") asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(v[r]), "v"(a) : "vcc");)
KERNEL(mix_fma_exp, 2, A1("v_fma_f32 %0, %0, %1, %2") A1K("v_exp_f32 %0, %0", 1))
KERNEL(mix_fma3_dpp, 4, A1("v_fma_f32 %0, %0, %1, %2") A1K("v_fma_f32 %0, %0, %1, %2", 1) A1K("v_fma_f32 %0, %0, %1, %2", 2) A1K("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", 3))

struct Entry { const char *name; void (*fn)(float *, unsigned long long *, float); int units; };
#define E(NAME) {#NAME, k_##NAME, units_##NAME}
static const Entry table[] = {
    E(fma), E(fmac), E(mul), E(add), E(sub), E(fma_neg), E(fma_lit), E(mul_sgpr), E(mov), E(mov_lit), E(min), E(max), E(med3), E(max3),
    E(bfi), E(and_b32), E(or_b32), E(lshl), E(add_u32), E(sub_u32), E(mad_u24), E(mul_lo), E(cvt_f2i), E(cvt_i2f), E(exp), E(exp_neg),
    E(rcp), E(log), E(rsq), E(cmp_vcc), E(cmp_e64), E(cmp_i32), E(cmp_lit), E(cmpx), E(cndmask_e64), E(cmp_nop_cnd), E(cmp_cnd_e64),
    E(saveexec2), E(pk_fma), E(pk_mul), E(pk_add), E(dpp_add_row_shr), E(dpp_mul_row_ror), E(dpp_mov_quad), E(dpp_mov_wave_ror),
    E(permlane32_swap), E(permlane16_swap), E(readlane), E(readfirstlane), E(s_nop0), E(s_nop1), E(mfma16_indep), E(mfma16_dep),
    E(mfma16_fma4), E(mfma16_fma12), E(mfma16_fma28), E(fma4_dswrite), E(fma12_dswrite), E(fma12_dsread128), E(fma12_only),
    E(mix_fma_cmp), E(mix_fma_exp), E(mix_fma3_dpp),
};

int main(int argc, char **argv) {
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 8192 * 256 * sizeof(float));
    hipMalloc(&cyc, 8192 * sizeof(unsigned long long));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("# cycles per INSTRUCTION per SIMD at 2.4 GHz (wall clock, all 256 CUs), and cycles per instruction of one wave (s_memtime)\n");
    printf("%-20s %5s", "kind", "n/it");
    const int wpss[] = {1, 2, 4, 5, 8};
    for (int w : wpss) printf("   wps=%d simd | wave", w);
    printf("\n");
    for (const Entry &en : table) {
        if (argc > 1 && !strstr(en.name, argv[1])) continue;
        printf("%-20s %5d", en.name, en.units);
        for (int wps : wpss) {
            const int blocks = 256 * wps;
            hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(en.fn, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[256];
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i]; avg /= 256;
            const double n_inst = (double)ITERS * en.units;
            const double winstr = (double)blocks * 4.0 * n_inst * 10.0;
            const double per_simd_s = winstr / 1024.0 / (ms * 1e-3);
            printf("   %10.2f | %5.2f", 2.4e9 / per_simd_s, avg / n_inst);
        }
        printf("\n");
        fflush(stdout);
    }
    hipFree(out); hipFree(cyc);
    return 0;
}
