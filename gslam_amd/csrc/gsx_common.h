// gsx_common.h — shared host/device helpers for the gfx950 kernels (wave64, LDS, error plumbing).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gsx.h"

#define GSX_WAVE 64

// ---- error plumbing (thread-local message, negative return codes) ------------------------------------------------
void gsx_set_error(const char *fmt, ...);

#define GSX_CHECK_ARG(cond)                                                       \
    do {                                                                          \
        if (!(cond)) {                                                            \
            gsx_set_error("%s:%d: invalid argument: %s", __FILE__, __LINE__, #cond); \
            return GSX_E_INVALID;                                                 \
        }                                                                         \
    } while (0)

#define GSX_CHECK_LAUNCH()                                                              \
    do {                                                                                \
        hipError_t e__ = hipGetLastError();                                             \
        if (e__ != hipSuccess) {                                                        \
            gsx_set_error("%s:%d: launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return GSX_E_LAUNCH;                                                        \
        }                                                                               \
    } while (0)

static inline int64_t gsx_align256(int64_t x) { return (x + 255) & ~(int64_t)255; }

// ---- zero fill -----------------------------------------------------------------------------------------------------
// All zero fills go through a kernel, never hipMemsetAsync: a memset node captured into a HIP graph was observed
// (ROCm 7.2, gfx950) to write a non-zero pattern from the second replay of the graph on.
#ifdef __HIPCC__
static __global__ __launch_bounds__(256) void gsx_zero_u32_kernel(uint32_t *__restrict__ p, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0u;
}
// n_words 4-byte words at p (4-byte aligned); returns false if the launch failed
static inline bool gsx_zero_async(void *p, int64_t n_words, hipStream_t st) {
    if (n_words <= 0) return true;
    hipLaunchKernelGGL(gsx_zero_u32_kernel, dim3((unsigned)((n_words + 255) / 256)), dim3(256), 0, st, (uint32_t *)p,
                       n_words);
    return hipGetLastError() == hipSuccess;
}
#endif

// ---- wave64 primitives --------------------------------------------------------------------------------------------
#ifdef __HIPCC__
// DPP row/bcast reduction: 4 row_shr steps inside each 16-lane row, then row_bcast:15 / row_bcast:31 (gfx9).
// Result (sum over the 64 lanes) is returned in every lane via readlane(63).
__device__ __forceinline__ float gsx_wave_sum_dpp(float v) {
    int x;
#define GSX_DPP_ADD(ctrl, rmask, bmask)                                                              \
    x = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rmask, bmask, true);        \
    v += __builtin_bit_cast(float, x);
    GSX_DPP_ADD(0x111, 0xf, 0xf)  // row_shr:1
    GSX_DPP_ADD(0x112, 0xf, 0xf)  // row_shr:2
    GSX_DPP_ADD(0x114, 0xf, 0xf)  // row_shr:4
    GSX_DPP_ADD(0x118, 0xf, 0xf)  // row_shr:8   -> lane 15 of each row holds the row sum
    GSX_DPP_ADD(0x142, 0xa, 0xf)  // row_bcast:15 into rows 1,3
    GSX_DPP_ADD(0x143, 0xc, 0xf)  // row_bcast:31 into rows 2,3 -> lane 63 holds the total
#undef GSX_DPP_ADD
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Inline-asm DPP sequences get no automatic hazard padding (a VALU write needs 2 wait states before a DPP read of the
// same register), and `asm volatile` does not stop the scheduler from sinking the ordinary VALU instruction that
// produces an operand right in front of the statement that reads it (seen in the disassembly: wrong sums).  A
// scheduling barrier that lets everything but VALU / transcendental instructions through pins the producers in front of
// the leading s_nop and keeps the DPP chain as written.
#define GSX_DPP_FENCE() __builtin_amdgcn_sched_barrier(0)

// Interleaved partial reduction of N independent values: after the call, lanes 15/31/47/63 hold the sums of their
// 16-lane DPP rows.  The N chains are advanced step by step so that consecutive DPP instructions are independent
// (no s_nop wait states between a VALU write and the DPP read of the same register).
// Written as inline asm (one v_add_f32_dpp per value and step): hipcc otherwise lowers about half of the steps to
// v_mov_b32_dpp + v_add_f32.  Inline asm gets no automatic hazard padding, so the VALU-write -> DPP-read distance of
// 2 wait states is kept by construction: N >= 3 interleaved chains, or an explicit s_nop between the steps.
template <int N>
__device__ __forceinline__ void gsx_row16_sum(float (&v)[N]) {
#define GSX_ROW_STEP(SHR)                                                                                      \
    if (N < 3) asm volatile("s_nop 1");                                                                        \
    _Pragma("unroll") for (int k = 0; k < N; ++k) {                                                            \
        asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:" #SHR " row_mask:0xf bank_mask:0xf bound_ctrl:1"       \
                     : "+v"(v[k]));                                                                            \
        GSX_DPP_FENCE();                                                                                       \
    }
    GSX_DPP_FENCE();
    asm volatile("s_nop 1");  // the values were just produced by ordinary VALU instructions
    GSX_DPP_FENCE();
    GSX_ROW_STEP(1)
    GSX_ROW_STEP(2)
    GSX_ROW_STEP(4)
    GSX_ROW_STEP(8)
    if (N < 3) asm volatile("s_nop 1");
#undef GSX_ROW_STEP
}

// Same, continued over the four rows: afterwards lane 63 holds the sum over all 64 lanes (row_bcast:15 into rows 1 and
// 3, then row_bcast:31 into rows 2 and 3).
template <int N>
__device__ __forceinline__ void gsx_wave63_sum(float (&v)[N]) {
    gsx_row16_sum<N>(v);
    if (N < 3) asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < N; ++k) {
        asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(v[k]));
        GSX_DPP_FENCE();
    }
    if (N < 3) asm volatile("s_nop 1");
#pragma unroll
    for (int k = 0; k < N; ++k) {
        asm volatile("v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(v[k]));
        GSX_DPP_FENCE();
    }
}

// Reduce-scatter of N <= 12 per-lane values over the wavefront's four 16-lane rows (measured on MI355X: every DPP
// instruction costs 4.2 SIMD cycles against 2.4 for a plain v_fma, tools/ubench/valu_rates.hip, so the 6 N DPP adds of a
// full reduction of each value dominate the rasteriser backward).  Steps:
//   1. scatter over the banks (4-lane groups) with row_shl / row_shr 4 and bank_mask, which selects the
//      writing lanes for free: register 2j = values (2j | 2j+1) in (even | odd) banks, summed over the bank pair  N ops
//   2. the same over bank pairs with row_shl / row_shr 8: register 4m = value 4m + bank, summed over the
//      four lanes of the row that sit at the same position of their quads                                      N/2 ops
//   3. quad butterflies (quad_perm xor 1, xor 2) on the (N+3)/4 registers that are left                          N/2 ops
// Afterwards, in every row, a lane of bank b holds in v[m] (m < (N+3)/4) that ROW's sum of value 4m + b; the caller
// adds the four rows up in memory (one ds_add_f32 per m with the 16 lanes `lane % 4 == 0`).  2 N DPP ops instead of
// 6 N and (N+3)/4 LDS instructions instead of N.  Slots 4m + b >= N hold garbage.  (The first version ran the quad
// butterflies first, on all N registers: 3.5 N ops.)
template <int N>
__device__ __forceinline__ void gsx_reduce_scatter(float (&v)[N]) {
    static_assert(N >= 1 && N <= 12, "gsx_reduce_scatter: N <= 12");
    GSX_DPP_FENCE();
    asm volatile("s_nop 1");
    GSX_DPP_FENCE();
    constexpr int N1 = (N + 1) / 2;
#pragma unroll
    for (int j = 0; j < N1; ++j) {          // banks 0,2 <- value 2j summed over the bank pair; banks 1,3 <- value 2j+1
        asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5" : "+v"(v[2 * j]));
        GSX_DPP_FENCE();
        if (2 * j + 1 < N) {
            asm volatile("v_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa" : "+v"(v[2 * j]) : "v"(v[2 * j + 1]));
            GSX_DPP_FENCE();
        }
    }
    asm volatile("s_nop 1");
    GSX_DPP_FENCE();
    constexpr int N2 = (N1 + 1) / 2;
#pragma unroll
    for (int m = 0; m < N2; ++m) {          // banks 0,1 <- register 2m summed over the row; banks 2,3 <- register 2m+1
        asm volatile("v_add_f32_dpp %0, %0, %0 row_shl:8 row_mask:0xf bank_mask:0x3" : "+v"(v[4 * m]));
        GSX_DPP_FENCE();
        if (2 * m + 1 < N1) {
            asm volatile("v_add_f32_dpp %0, %1, %1 row_shr:8 row_mask:0xf bank_mask:0xc" : "+v"(v[4 * m]) : "v"(v[4 * m + 2]));
            GSX_DPP_FENCE();
        }
    }
    asm volatile("s_nop 1");
    GSX_DPP_FENCE();
#pragma unroll
    for (int m = 0; m < N2; ++m) {
        asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[4 * m]));
        GSX_DPP_FENCE();
    }
    asm volatile("s_nop 1");
    GSX_DPP_FENCE();
#pragma unroll
    for (int m = 0; m < N2; ++m) {
        asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(v[4 * m]));
        GSX_DPP_FENCE();
    }
    // compact: result register m lives in v[4m]
#pragma unroll
    for (int m = 1; m < N2; ++m) v[m] = v[4 * m];
}

// lane i + lane i^16 + lane i^32 + lane i^48 in every lane (gfx950 v_permlane32_swap / v_permlane16_swap: no LDS, no DPP
// row limit): the cross-row step that the DPP network lacks for lane-wise data.
__device__ __forceinline__ float gsx_xrow_sum(float v) {
    unsigned a = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    a = __float_as_uint(v);
    const auto s = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    return __uint_as_float(s[0]) + __uint_as_float(s[1]);
}

// Cross-row step of the reduce-scatter for NR = 2 or 3 registers (value sets): instead of summing every register over
// the four rows (3 instructions per register and level), the swaps hand each row ONE value set to total:
//   v_permlane32_swap(a, b): a' = [a.lo, b.lo], b' = [a.hi, b.hi]          -> a' + b' = [a.lo + a.hi | b.lo + b.hi]
//   v_permlane16_swap(a, b): a' = [a.r0, b.r0, a.r2, b.r2], b' = [a.r1, b.r1, a.r3, b.r3]
// (probed on gfx950: tools/ubench/permlane_test.hip).  Result: row 0 = total of set 0, row 2 = total of set 1, row 1
// (and 3) = total of set 2 (NR = 3) or copies of rows 0 / 2 (NR = 2).  7 (5) instructions instead of 18 (12).
__device__ __forceinline__ float gsx_add_swapped(unsigned a, unsigned b, bool rows16) {
    if (rows16) {
        const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int NR>
__device__ __forceinline__ float gsx_xrow_scatter(const float *v) {
    static_assert(NR == 2 || NR == 3, "gsx_xrow_scatter: 2 or 3 value sets");
    const float t = gsx_add_swapped(__float_as_uint(v[0]), __float_as_uint(v[1]), false);
    const float u = (NR == 3) ? gsx_add_swapped(__float_as_uint(v[2]), __float_as_uint(v[2]), false) : t;
    return gsx_add_swapped(__float_as_uint(t), __float_as_uint(u), true);
}
// column (value index) a lane of row `row`, bank `bank` holds after gsx_reduce_scatter + gsx_xrow_scatter, or -1
template <int NR>
__device__ __forceinline__ int gsx_xrow_column(int lane) {
    const int row = lane >> 4, bank = (lane >> 2) & 3;
    if ((lane & 3) != 0) return -1;
    if (row == 0) return bank;
    if (row == 2) return 4 + bank;
    return (NR == 3 && row == 1) ? 8 + bank : -1;
}

// 48-byte splat record fetched through the scalar data cache into SGPRs (uniform address): the broadcast of a
// Gaussian to all 64 pixel lanes costs no VALU instruction and no LDS traffic.
typedef float gsx_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) gsx_f4 *gsx_cf4p;
__device__ __forceinline__ gsx_cf4p gsx_scalar_ptr(const float *p) { return (gsx_cf4p)(uintptr_t)p; }

__device__ __forceinline__ float gsx_wave_sum_shfl(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

#ifndef GSX_USE_SHFL_REDUCE
#define gsx_wave_sum gsx_wave_sum_dpp
#else
#define gsx_wave_sum gsx_wave_sum_shfl
#endif

// LDS float add without return value (ds_add_f32).  Written as inline asm so that the AMDGPU atomic optimizer does not
// wrap it in its own wave-wide DPP reduction: the callers have already reduced to a handful of active lanes.
__device__ __forceinline__ void gsx_lds_fadd(float *lds_ptr, float v) {
    const unsigned addr = (unsigned)(uintptr_t)lds_ptr;  // LDS pointers are 32-bit offsets in the generic->local cast
    asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

template <int BYTE_OFF>
__device__ __forceinline__ void gsx_lds_fadd_off(unsigned lds_addr, float v) {
    asm volatile("ds_add_f32 %0, %1 offset:%2" : : "v"(lds_addr), "v"(v), "n"(BYTE_OFF) : "memory");
}

template <int N, int K = 0>
__device__ __forceinline__ void gsx_lds_fadd_row(unsigned lds_addr, const float (&v)[N]) {
    if constexpr (K < N) {
        gsx_lds_fadd_off<4 * K>(lds_addr, v[K]);
        gsx_lds_fadd_row<N, K + 1>(lds_addr, v);
    }
}

__device__ __forceinline__ int gsx_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float gsx_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif  // __HIPCC__
