// Calibration of rocprofv3's FETCH_SIZE for the access shapes of this code base (MI355X_MICROARCH.md, HBM: "other access
// widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel reads the SAME
// 512 MiB buffer exactly once (larger than the 256 MiB Infinity Cache) and folds it into one word per thread so that
// nothing is elided:
//   read_b128       16 B per lane, contiguous (the calibrated reference: FETCH_SIZE reports half the bytes)
//   read_b32         4 B per lane, contiguous                     (SSIM derivative planes, planar images)
//   read_nhwc3of5    3 adjacent dwords out of every 20-byte pixel  (SSIM / loss reading the colours of a CH = 5 render)
//   read_b96_aos    12 B per lane from 12-byte rows                (means / scales in the projection)
//   gather_b128x3   48-byte records gathered by a random index      (rasteriser record gather)
//   read_b160_aos   20 B per lane from 20-byte rows                (the CH = 5 render / its gradient in the loss kernel)
//   read_loss_shape the mapping loss kernel's reads of a 640-wide image: the pixel's 20-byte row, its right and lower
//                   neighbours' rows (edge-aware TV), a 4-byte alpha, a 12-byte gt row and 3 planar ssim-gradient values;
//                   bytes it touches = 20 + 4 + 12 + 12 per pixel (the neighbours are re-reads of other lanes' rows)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
// run  : rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o c -- tools/ubench/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void read_b128(const float4 *p, int64_t n4, float *out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = p[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read_b32(const float *p, int64_t n, float *out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) acc += p[i];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read_nhwc3of5(const float *p, int64_t npix, float *out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x)
        acc += p[5 * i] + p[5 * i + 1] + p[5 * i + 2];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void read_b96_aos(const float *p, int64_t nrow, float *out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += (int64_t)gridDim.x * blockDim.x)
        acc += p[3 * i] + p[3 * i + 1] + p[3 * i + 2];
    if (acc == 123.456f) out[0] = acc;
}
__global__ void gather_b128x3(const float4 *p, int64_t nrec, float *out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = (int64_t)((uint64_t)i * 2654435761ull % (uint64_t)nrec);     // a permutation-like scatter of the rows
        const float4 a = p[3 * j], b = p[3 * j + 1], c = p[3 * j + 2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) out[0] = acc;
}

__global__ void read_b160_aos(const float *p, int64_t nrow, float *out) {
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrow; i += (int64_t)gridDim.x * blockDim.x)
        acc += p[5 * i] + p[5 * i + 1] + p[5 * i + 2] + p[5 * i + 3] + p[5 * i + 4];
    if (acc == 123.456f) out[0] = acc;
}
// one block of 256 consecutive pixels per trip, like map_loss_kernel; W = 640
__global__ void read_loss_shape(const float *render, const float *alphas, const float *gt, const float *sg, int64_t npix,
                                float *out) {
    const int W = 640;
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix - W - 1; i += (int64_t)gridDim.x * blockDim.x) {
        const float *r = render + 5 * i, *rx = render + 5 * (i + 1), *ry = render + 5 * (i + W);
        acc += r[0] + r[1] + r[2] + r[3] + r[4] + rx[0] + rx[1] + rx[2] + rx[3] + ry[0] + ry[1] + ry[2] + ry[3];
        acc += alphas[i] + gt[3 * i] + gt[3 * i + 1] + gt[3 * i + 2] + sg[i] + sg[npix + i] + sg[2 * npix + i];
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const int64_t bytes = 512ll << 20;
    float *buf, *out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(buf, 0, bytes));
    const int64_t n = bytes / 4;
    const dim3 grid(256 * 16), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(read_b128, grid, block, 0, 0, (const float4 *)buf, n / 4, out);
        hipLaunchKernelGGL(read_b32, grid, block, 0, 0, buf, n, out);
        hipLaunchKernelGGL(read_nhwc3of5, grid, block, 0, 0, buf, n / 5, out);
        hipLaunchKernelGGL(read_b96_aos, grid, block, 0, 0, buf, n / 3, out);
        hipLaunchKernelGGL(gather_b128x3, grid, block, 0, 0, (const float4 *)buf, n / 12, out);
        hipLaunchKernelGGL(read_b160_aos, grid, block, 0, 0, buf, n / 5, out);
        // the loss shape over one 512 MiB buffer cut into render (20 B/px) | alphas (4) | gt (12) | ssim gradient (12) = 48 B/px
        const int64_t npix = bytes / 48;
        hipLaunchKernelGGL(read_loss_shape, grid, block, 0, 0, buf, buf + 5 * npix, buf + 6 * npix, buf + 9 * npix, npix, out);
    }
    CHECK(hipDeviceSynchronize());
    printf("buffer_bytes %lld (every kernel touches all of its lines once)\n", (long long)bytes);
    return 0;
}
