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

__device__ __forceinline__ int gsx_lane() { return threadIdx.x & 63; }

__device__ __forceinline__ float gsx_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
#endif  // __HIPCC__
