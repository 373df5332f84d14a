// semantics of v_permlane32_swap / v_permlane16_swap on gfx950, probed with lane-distinct values
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *o) {
    const unsigned lane = threadIdx.x;
    unsigned a = lane, b = 100 + lane;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    o[lane] = r[0]; o[64 + lane] = r[1];
    a = lane; b = 100 + lane;
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[128 + lane] = s[0]; o[192 + lane] = s[1];
}
int main() {
    unsigned *d; hipMalloc(&d, 1024); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    const char *names[4] = {"swap32 first ", "swap32 second", "swap16 first ", "swap16 second"};
    for (int q = 0; q < 4; ++q) { printf("%s:", names[q]); for (int i = 0; i < 64; i += 8) printf(" [%d]=%u", i, h[q * 64 + i]); printf("\n"); }
    return 0;
}
