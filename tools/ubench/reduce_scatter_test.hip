// checks gsx_reduce_scatter<N> against host sums with lane-distinct inputs: hipcc --offload-arch=gfx950 -O3 -I gslam_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "gsx_common.h"

template <int N>
__global__ void k(float *out) {
    float v[N];
    const int lane = threadIdx.x;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = (float)((lane * 7 + i * 13) % 29) + 0.25f * i;
    gsx_reduce_scatter<N>(v);
    constexpr int R = (N + 3) / 4;
#pragma unroll
    for (int m = 0; m < R; ++m) out[m * 64 + lane] = v[m];
}

template <int N>
int run() {
    float *d;
    hipMalloc(&d, 3 * 64 * 4);
    hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d);
    float h[192];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int m = 0; m < (N + 3) / 4; ++m)
        for (int lane = 0; lane < 64; ++lane) {
            const int row = lane / 16, bank = (lane / 4) % 4, val = 4 * m + bank;
            if (val >= N) continue;
            float ref = 0.f;
            for (int l = row * 16; l < row * 16 + 16; ++l) ref += (float)((l * 7 + val * 13) % 29) + 0.25f * val;
            if (fabsf(h[m * 64 + lane] - ref) > 1e-3f) {
                if (bad < 8) printf("N=%d m=%d lane=%d (row %d bank %d value %d): got %f want %f\n", N, m, lane, row, bank, val, h[m * 64 + lane], ref);
                ++bad;
            }
        }
    printf("N=%d: %d mismatches\n", N, bad);
    hipFree(d);
    return bad;
}

int main() { return run<11>() + run<12>() + run<9>() + run<8>() + run<7>() + run<5>() + run<3>() + run<1>() ? 1 : 0; }
