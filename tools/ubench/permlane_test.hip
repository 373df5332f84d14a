#include <hip/hip_runtime.h>
__global__ void k(float *o) {
    float v = (float)threadIdx.x;
    // lane i + lane i^32, then + lane i^16
    {
        unsigned a = __float_as_uint(v), b = a;
        auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    {
        unsigned a = __float_as_uint(v), b = a;
        auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
        v = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    }
    o[threadIdx.x] = v;
}
int main() {
    float *d; hipMalloc(&d, 256); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[64]; hipMemcpy(h, d, 256, hipMemcpyDeviceToHost);
    for (int i = 0; i < 64; ++i) printf("%d:%g ", i, h[i]);
    printf("\n"); return 0;
}
