// ssim.hip — K11/K12: fused SSIM loss, forward and backward.  Replaces rahul-goel/fused-ssim@30fb258c
// `fused_ssim(img1, img2, padding, train)` (gslam/backend.py:13,303-307).  Maths: SURVEY.md §9.5.
//
// Mapping: one 256-thread workgroup per 32x16 output tile of one (batch, channel) plane (forward) or of the three colour
// planes of an image (backward); workgroups take their tiles in an XCD-aware order (xcd_tile) so that the halo lines
// neighbouring tiles share are fetched into one L2 once: PMC traffic 0.92x / 1.08x the algorithmic 72 B per pixel (it was
// 1.7x / 2.9x with the natural order; FETCH_SIZE calibrated for these access shapes with tools/ubench/fetch_calib.hip).  The 42x26 input halo of
// both images is staged in LDS once (zero padded), the 11-tap separable Gaussian runs horizontally into LDS
// (5 moments) and vertically into registers, both passes register-blocked.  Inputs are read through explicit (B,C,H,W) strides so the NHWC
// renders of the rasteriser are consumed without a permute copy.  The map mean is reduced wave64 -> LDS -> one
// partial per workgroup, then a single-block finishing kernel (deterministic, no atomics).
#include "gsx_common.h"
#include "loss_pixel.h"

namespace {

constexpr int TSX = 32;          // output tile: 32 x 16 pixels per 256-thread workgroup (2 outputs per thread in the
constexpr int TSY = 16;          // vertical pass, 4 per active thread in the horizontal one)
constexpr int HALO = 5;          // 11-tap window
constexpr int INX = TSX + 2 * HALO;  // 42
constexpr int INY = TSY + 2 * HALO;  // 26
constexpr int STAGE_TRIPS = (INY * INX + 255) / 256;   // trips of a 256-thread workgroup over the 42 x 26 halo: 5

__device__ __constant__ float c_win[11] = {1.0283800845e-03f, 7.5987581352e-03f, 3.6000772128e-02f, 1.0936068951e-01f,
                                           2.1300553771e-01f, 2.6601172486e-01f, 2.1300553771e-01f, 1.0936068951e-01f,
                                           3.6000772128e-02f, 7.5987581352e-03f, 1.0283800845e-03f};

typedef float f2 __attribute__((ext_vector_type(2)));

struct Strides {
    int64_t b, c, h, w;
};

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (observed, a speed assumption only), each with
// its own L2: with the natural order horizontally adjacent tiles - which share 10 of their 42 halo columns and, at 128-byte
// line granularity, most of their edge lines - sit on different XCDs and every L2 fetches its own copy (PMC: 2.9x the
// algorithmic bytes in the backward).  Here workgroup b takes tile  chunk(b % 8) + b / 8  of the linearised (plane, row,
// column) space, so the workgroups that share an XCD walk one contiguous band of tile rows.  Bijective for any grid size.
// ch_fast > 1 (one channel per workgroup): the ch_fast channel planes of an image are the FASTEST index, so the workgroups
// that read the same interleaved pixel rows for different channels run back to back on one XCD and share the lines in L2.
struct TileId { int bx, by, bz; };
__device__ __forceinline__ TileId xcd_tile(int gx, int gy, int gz, int ch_fast = 1) {
    const int total = gx * gy * gz;
    const int b = blockIdx.x, xcd = b % 8, k = b / 8;
    const int q = total / 8, r = total % 8;
    int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    const int ch = id % ch_fast;
    id /= ch_fast;
    TileId t;
    t.bx = id % gx;
    t.by = (id / gx) % gy;
    t.bz = (id / (gx * gy)) * ch_fast + ch;
    return t;
}

// Register-blocked separable filter: a thread of the horizontal pass produces 4 adjacent outputs of one row from 14
// inputs (3.5 LDS reads per output and moment pair instead of 11), a thread of the vertical pass 2 adjacent rows from
// 12 (6 reads per output and moment instead of 11); the halo overhead of the 32 x 16 tile is 2.13x (16 x 16: 2.64x).
// NC = channels per workgroup.  The renders arrive channel-interleaved (NHWC, 20 bytes per pixel with depth and beta behind
// the colours): one workgroup per (tile, channel) read every input line once PER CHANNEL - 3x the bytes (PMC: 37.9 MB per
// forward and 64.4 MB per backward against 22.1 MB algorithmic at 640x480).  With NC = 3 the halo tile of all three colour
// channels is staged once, from contiguous pixel rows, and the channels are filtered one after the other out of LDS.
template <int NC>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float *__restrict__ img1, const float *__restrict__ img2,
                                                       int CH, int H, int W, Strides s1, Strides s2, int crop,
                                                       float *__restrict__ partials, float *__restrict__ dm_dmu1,
                                                       float *__restrict__ dm_ds1, float *__restrict__ dm_ds12, int gx_n,
                                                       int gy_n, int gz_n) {
    __shared__ float sx[NC][INY][INX + 1];
    __shared__ float sy[NC][INY][INX + 1];
    __shared__ float hz[5][INY][TSX + 1];
    __shared__ float s_red[4];
    const TileId tid = xcd_tile(gx_n, gy_n, gz_n, NC == 1 ? CH : 1);
    const int groups = CH / NC;                               // channel groups per image
    const int b = tid.bz / groups, ch0 = (tid.bz - b * groups) * NC;
    const int x0 = tid.bx * TSX, y0 = tid.by * TSY;
    const int t = threadIdx.x;
    const float *p1 = img1 + b * s1.b + ch0 * s1.c;
    const float *p2 = img2 + b * s2.b + ch0 * s2.c;
    // every load of the halo is issued before the first value is stored: a workgroup is a chain load -> LDS -> filter -> store whose
    // length sets the kernel's time (56 workgroups per CU, six at a time), and the rolled loop waited for each of its five trips'
    // loads in turn (forward 83 -> 5x us, backward 100 -> 6x us at 8 x 640 x 480: DESIGN.md 6)
    {
        float va[STAGE_TRIPS][NC], vc[STAGE_TRIPS][NC];
#pragma unroll
        for (int u = 0; u < STAGE_TRIPS; ++u) {
            const int i = t + u * 256;
            const int ly = i / INX, lx = i - ly * INX;
            const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
            const bool in = i < INY * INX && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
            for (int k = 0; k < NC; ++k) {
                va[u][k] = in ? p1[gy * s1.h + gx * s1.w + k * s1.c] : 0.f;
                vc[u][k] = in ? p2[gy * s2.h + gx * s2.w + k * s2.c] : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < STAGE_TRIPS; ++u) {
            const int i = t + u * 256;
            if (i < INY * INX) {
                const int ly = i / INX, lx = i - ly * INX;
#pragma unroll
                for (int k = 0; k < NC; ++k) { sx[k][ly][lx] = va[u][k]; sy[k][ly][lx] = vc[u][k]; }
            }
        }
    }
    __syncthreads();
    for (int kc = 0; kc < NC; ++kc) {
        const int plane = b * CH + ch0 + kc;
        if (t < INY * (TSX / 4)) {                 // 26 rows x 8 groups of 4 outputs = 208 threads
            const int ly = t / (TSX / 4), lx = (t - ly * (TSX / 4)) * 4;
            float a[14], c[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) { a[k] = sx[kc][ly][lx + k]; c[k] = sy[kc][ly][lx + k]; }
            // packed fp32 (v_pk_mul / v_pk_add / v_pk_fma: two lanes of arithmetic per instruction) on the pairs (image 1, image 2):
            // 4 instructions per tap instead of 7, the same IEEE operations on every component - bit-identical sums
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f2 m = {0.f, 0.f}, e = {0.f, 0.f};
                float e12 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float w = c_win[k];
                    const f2 v = {a[o + k], c[o + k]}, ww = {w, w};
                    const f2 wv = ww * v;
                    m += wv;
                    e = __builtin_elementwise_fma(wv, v, e);
                    e12 = __builtin_fmaf(wv.x, v.y, e12);
                }
                hz[0][ly][lx + o] = m.x; hz[1][ly][lx + o] = m.y; hz[2][ly][lx + o] = e.x; hz[3][ly][lx + o] = e.y;
                hz[4][ly][lx + o] = e12;
            }
        }
        __syncthreads();
        const int lx = t & 31, ly = (t >> 5) * 2;  // two vertically adjacent outputs per thread
        const int gx = x0 + lx;
        // the two outputs of a thread as the two halves of packed accumulators: row r feeds output 0 with tap r and output 1 with
        // tap r - 1 (a zero weight at the ends adds + 0 to a sum that is never - 0: the same bits as leaving the term out)
        f2 pacc[5];
#pragma unroll
        for (int m = 0; m < 5; ++m) pacc[m] = f2{0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const f2 w = {r < 11 ? c_win[r] : 0.f, r >= 1 ? c_win[r - 1] : 0.f};
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const float v = hz[m][ly + r][lx];
                pacc[m] = __builtin_elementwise_fma(w, f2{v, v}, pacc[m]);
            }
        }
        float acc[2][5];
#pragma unroll
        for (int m = 0; m < 5; ++m) { acc[0][m] = pacc[m].x; acc[1][m] = pacc[m].y; }
        float val = 0.f;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int gy = y0 + ly + o;
            if (gx < W && gy < H) {
                const float m1 = acc[o][0], m2 = acc[o][1], e11 = acc[o][2], e22 = acc[o][3], e12 = acc[o][4];
                const float s1q = e11 - m1 * m1, s2q = e22 - m2 * m2, s12 = e12 - m1 * m2;
                const float A = 2.0f * m1 * m2 + GSX_SSIM_C1, B = 2.0f * s12 + GSX_SSIM_C2;
                const float Cq = m1 * m1 + m2 * m2 + GSX_SSIM_C1, D = s1q + s2q + GSX_SSIM_C2;
                // One division: with icd = 1/(Cq D), 1/D = icd Cq and 1/Cq = icd D, so the four quotients of d(map)/d(mu1)
                // collapse to 2 icd (m2 (B - A) + m1 m (Cq - D)) (the IEEE divides were a quarter of the kernel's VALU work).
                const float icd = 1.0f / (Cq * D);
                const float m = (A * B) * icd;
                const bool in_crop = gx >= crop && gx < W - crop && gy >= crop && gy < H - crop;
                if (in_crop) val += m;
                if (dm_dmu1) {
                    const int64_t o_ = ((int64_t)plane * H + gy) * W + gx;
                    dm_dmu1[o_] = 2.0f * icd * (m2 * (B - A) + m1 * m * (Cq - D));
                    dm_ds1[o_] = -m * (icd * Cq);
                    dm_ds12[o_] = 2.0f * A * icd;
                }
            }
        }
        const float tot = gsx_wave_sum(val);
        if ((t & 63) == 0) s_red[t >> 6] = tot;
        __syncthreads();                                     // also: hz is free for the next channel
        if (t == 0) {
            const int bid = (plane * gy_n + tid.by) * gx_n + tid.bx;                       // one partial per (plane, tile)
            partials[bid] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        }
    }
}

__global__ __launch_bounds__(256) void reduce_partials_kernel(const float *__restrict__ partials, int64_t n,
                                                              float *__restrict__ out) {
    __shared__ float s_red[4];
    float acc = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc += partials[i];
    const float tot = gsx_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = tot;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

template <int NC>
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float *__restrict__ img1, const float *__restrict__ img2,
                                                       int CH, int H, int W, Strides s1, Strides s2, int crop,
                                                       const float *__restrict__ dm_dmu1,
                                                       const float *__restrict__ dm_ds1,
                                                       const float *__restrict__ dm_ds12,
                                                       const float *__restrict__ scale, float scale_mul,
                                                       float *__restrict__ dL_dimg1, int gx_n, int gy_n, int gz_n) {
    __shared__ float sm[3][INY][INX + 1];
    __shared__ float hz[3][INY][TSX + 1];
    __shared__ float s_xy[2][NC][TSY][TSX + 1];               // the output pixels of both images, all NC channels
    const TileId tid = xcd_tile(gx_n, gy_n, gz_n);
    const int groups = CH / NC;
    const int b = tid.bz / groups, ch0 = (tid.bz - b * groups) * NC;
    const int x0 = tid.bx * TSX, y0 = tid.by * TSY;
    const int t = threadIdx.x;
    // the image pixels of the tile: read once for all NC channels (contiguous pixel rows of the interleaved render)
    for (int i = t; i < TSY * TSX; i += 256) {
        const int ly = i / TSX, lx = i - ly * TSX;
        const int gy = y0 + ly, gx = x0 + lx;
        const bool in = gy < H && gx < W;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            s_xy[0][k][ly][lx] = in ? img1[b * s1.b + (ch0 + k) * s1.c + gy * s1.h + gx * s1.w] : 0.f;
            s_xy[1][k][ly][lx] = in ? img2[b * s2.b + (ch0 + k) * s2.c + gy * s2.h + gx * s2.w] : 0.f;
        }
    }
    const float sc = scale[0] * scale_mul;
    for (int kc = 0; kc < NC; ++kc) {
        const int plane = b * CH + ch0 + kc;
        __syncthreads();                                     // sm / hz of the previous channel are no longer read
        {
            float va[STAGE_TRIPS], vc[STAGE_TRIPS], vd[STAGE_TRIPS];   // (all loads in flight before the first LDS store: see the forward)
#pragma unroll
            for (int u = 0; u < STAGE_TRIPS; ++u) {
                const int i = t + u * 256;
                const int ly = i / INX, lx = i - ly * INX;
                const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
                const bool in = i < INY * INX && gx >= crop && gx < W - crop && gy >= crop && gy < H - crop;   // dL/dmap is zero outside the crop
                const int64_t o = ((int64_t)plane * H + (in ? gy : 0)) * W + (in ? gx : 0);
                va[u] = in ? dm_dmu1[o] : 0.f; vc[u] = in ? dm_ds1[o] : 0.f; vd[u] = in ? dm_ds12[o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < STAGE_TRIPS; ++u) {
                const int i = t + u * 256;
                if (i < INY * INX) {
                    const int ly = i / INX, lx = i - ly * INX;
                    sm[0][ly][lx] = va[u]; sm[1][ly][lx] = vc[u]; sm[2][ly][lx] = vd[u];
                }
            }
        }
        __syncthreads();
        if (t < INY * (TSX / 4)) {
            const int ly = t / (TSX / 4), lx = (t - ly * (TSX / 4)) * 4;
            // maps 0 and 1 as the halves of a packed accumulator (v_pk_fma_f32), map 2 on its own: the same fused multiply-adds
            float in0[14], in1[14], in2[14];
#pragma unroll
            for (int k = 0; k < 14; ++k) { in0[k] = sm[0][ly][lx + k]; in1[k] = sm[1][ly][lx + k]; in2[k] = sm[2][ly][lx + k]; }
#pragma unroll
            for (int o = 0; o < 4; ++o) {
                f2 acc01 = {0.f, 0.f};
                float acc2 = 0.f;
#pragma unroll
                for (int k = 0; k < 11; ++k) {
                    const float w = c_win[k];
                    acc01 = __builtin_elementwise_fma(f2{w, w}, f2{in0[o + k], in1[o + k]}, acc01);
                    acc2 = __builtin_fmaf(w, in2[o + k], acc2);
                }
                hz[0][ly][lx + o] = acc01.x; hz[1][ly][lx + o] = acc01.y; hz[2][ly][lx + o] = acc2;
            }
        }
        __syncthreads();
        const int lx = t & 31, ly = (t >> 5) * 2;
        const int gx = x0 + lx;
        // the thread's two outputs as the halves of packed accumulators (tap r for output 0, tap r - 1 for output 1; see the forward)
        f2 pa[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const f2 w = {r < 11 ? c_win[r] : 0.f, r >= 1 ? c_win[r - 1] : 0.f};
            const float v0 = hz[0][ly + r][lx], v1 = hz[1][ly + r][lx], v2 = hz[2][ly + r][lx];
            pa[0] = __builtin_elementwise_fma(w, f2{v0, v0}, pa[0]);
            pa[1] = __builtin_elementwise_fma(w, f2{v1, v1}, pa[1]);
            pa[2] = __builtin_elementwise_fma(w, f2{v2, v2}, pa[2]);
        }
        const float acc[2][3] = {{pa[0].x, pa[1].x, pa[2].x}, {pa[0].y, pa[1].y, pa[2].y}};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int gy = y0 + ly + o;
            if (gx < W && gy < H) {
                const float x = s_xy[0][kc][ly + o][lx], y = s_xy[1][kc][ly + o][lx];
                dL_dimg1[((int64_t)plane * H + gy) * W + gx] = sc * (acc[o][0] + 2.0f * x * acc[o][1] + y * acc[o][2]);
            }
        }
    }
}

// ---- SSIM backward + the mapping loss block in one pass (round 5) ------------------------------------------------
// ssim_bwd_kernel<3> holds, for its 32 x 16 output pixels, the colours of render and target in LDS and the SSIM gradient of the
// three colours in registers - everything gsx_map_loss reads per pixel except depth / beta / alpha (same cache lines as the
// colours).  Finishing the pixel here (loss_pixel.h: exposure affine, active-nerf photometric + log^2 beta, edge-aware depth TV,
// SSIM gradient added last - the same expressions in the same order as map_loss_kernel, so d loss / d render comes out bit for bit)
// takes the planar SSIM gradient's round trip through HBM (write 12 B, read 12 B per pixel), one more pass over render + target
// and one launch out of the BA iteration.  One partial row of the loss sums per tile (gsx_ssim_bwd_map_loss_rows).
__global__ __launch_bounds__(256, 5) void ssim_bwd_loss_kernel(gsx_loss::LossArgs A, int crop, const float *__restrict__ dm_dmu1,
                                                            const float *__restrict__ dm_ds1,
                                                            const float *__restrict__ dm_ds12,
                                                            const float *__restrict__ scale, float scale_mul, int gx_n,
                                                            int gy_n, int gz_n) {
    constexpr int NC = 3;
    __shared__ float sm[3][INY][INX + 1];
    __shared__ float hz[3][INY][TSX + 1];
    __shared__ float s_part[4][gsx_loss::NPART];
    const TileId tid = xcd_tile(gx_n, gy_n, gz_n);
    const int b = tid.bz;                                     // camera
    const int H = A.H, W = A.W, CH = A.CH;
    const int x0 = tid.bx * TSX, y0 = tid.by * TSY;
    const int t = threadIdx.x;
    const float sc = scale[0] * scale_mul;
    const int lx = t & 31, ly = (t >> 5) * 2;
    const int gx = x0 + lx;
    // the colours of the thread's two output pixels in both images: registers, requested before anything else (they were a 12.7 KB
    // LDS tile: without it five workgroups fit a CU instead of four - the kernel's time is its workgroups' chain length times the
    // rounds it takes to run them all)
    float px[2][NC], py[2][NC];
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int gy = y0 + ly + o;
        const bool in = gy < H && gx < W;
        const int64_t p = ((int64_t)b * H + (in ? gy : 0)) * W + (in ? gx : 0);
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const float r = A.render[p * CH + k], q = A.gt[p * 3 + k];
            px[o][k] = in ? r : 0.f;
            py[o][k] = in ? q : 0.f;
        }
    }
    float g[2][3];
#pragma nounroll
    for (int kc = 0; kc < NC; ++kc) {
        const int plane = b * 3 + kc;
        __syncthreads();
        {
            float va[STAGE_TRIPS], vc[STAGE_TRIPS], vd[STAGE_TRIPS];   // (all loads in flight before the first LDS store: see the forward)
#pragma unroll
            for (int u = 0; u < STAGE_TRIPS; ++u) {
                const int i = t + u * 256;
                const int yy = i / INX, xx = i - yy * INX;
                const int gy = y0 + yy - HALO, gxx = x0 + xx - HALO;
                const bool in = i < INY * INX && gxx >= crop && gxx < W - crop && gy >= crop && gy < H - crop;
                const int64_t o = ((int64_t)plane * H + (in ? gy : 0)) * W + (in ? gxx : 0);
                va[u] = in ? dm_dmu1[o] : 0.f; vc[u] = in ? dm_ds1[o] : 0.f; vd[u] = in ? dm_ds12[o] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < STAGE_TRIPS; ++u) {
                const int i = t + u * 256;
                if (i < INY * INX) {
                    const int yy = i / INX, xx = i - yy * INX;
                    sm[0][yy][xx] = va[u]; sm[1][yy][xx] = vc[u]; sm[2][yy][xx] = vd[u];
                }
            }
        }
        __syncthreads();
        if (t < INY * (TSX / 4)) {
            const int yy = t / (TSX / 4), xx = (t - yy * (TSX / 4)) * 4;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                float in[14];
#pragma unroll
                for (int k = 0; k < 14; ++k) in[k] = sm[m][yy][xx + k];
#pragma unroll
                for (int o = 0; o < 4; ++o) {
                    float acc = 0.f;
#pragma unroll
                    for (int k = 0; k < 11; ++k) acc = __builtin_fmaf(c_win[k], in[o + k], acc);
                    hz[m][yy][xx + o] = acc;
                }
            }
        }
        __syncthreads();
        // the thread's two outputs as the halves of packed accumulators (tap r for output 0, tap r - 1 for output 1; see the forward)
        f2 pa[3] = {f2{0.f, 0.f}, f2{0.f, 0.f}, f2{0.f, 0.f}};
#pragma unroll
        for (int r = 0; r < 12; ++r) {
            const f2 w = {r < 11 ? c_win[r] : 0.f, r >= 1 ? c_win[r - 1] : 0.f};
            const float v0 = hz[0][ly + r][lx], v1 = hz[1][ly + r][lx], v2 = hz[2][ly + r][lx];
            pa[0] = __builtin_elementwise_fma(w, f2{v0, v0}, pa[0]);
            pa[1] = __builtin_elementwise_fma(w, f2{v1, v1}, pa[1]);
            pa[2] = __builtin_elementwise_fma(w, f2{v2, v2}, pa[2]);
        }
        const float acc[2][3] = {{pa[0].x, pa[1].x, pa[2].x}, {pa[0].y, pa[1].y, pa[2].y}};
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const float x = kc == 0 ? px[o][0] : (kc == 1 ? px[o][1] : px[o][2]);
            const float y = kc == 0 ? py[o][0] : (kc == 1 ? py[o][1] : py[o][2]);
            const float v = sc * (acc[o][0] + 2.0f * x * acc[o][1] + y * acc[o][2]);
            if (kc == 0) g[o][0] = v; else if (kc == 1) g[o][1] = v; else g[o][2] = v;
        }
    }
    float part[gsx_loss::NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int o = 0; o < 2; ++o) {
        const int gy = y0 + ly + o;
        if (gx < W && gy < H) {
            float pp[gsx_loss::NPART] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            gsx_loss::map_loss_pixel(A, b, gx, gy, px[o][0], px[o][1], px[o][2], py[o][0], py[o][1], py[o][2], true, g[o], pp);
#pragma unroll
            for (int k = 0; k < gsx_loss::NPART; ++k) part[k] += pp[k];
        }
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int k = 0; k < gsx_loss::NPART; ++k) {
        const float tot = gsx_wave_sum(part[k]);
        if (lane == 0) s_part[wave][k] = tot;
    }
    __syncthreads();
    if (t < gsx_loss::NPART)
        A.partials[(((int64_t)b * gy_n + tid.by) * gx_n + tid.bx) * gsx_loss::NPART + t] =
            (s_part[0][t] + s_part[1][t]) + (s_part[2][t] + s_part[3][t]);
}

}  // namespace

extern "C" int64_t gsx_ssim_bwd_map_loss_rows(int64_t C, int H, int W) {
    return C * ((H + TSY - 1) / TSY) * ((W + TSX - 1) / TSX);
}

extern "C" int gsx_ssim_bwd_map_loss(const float *render, const float *alphas, const float *gt, const float *exposure, int64_t C,
                                     int H, int W, int CH, int depth_index, int beta_index, int mode, float w_photo, float w_tv,
                                     float mask_thresh, int crop, const float *dm_dmu1, const float *dm_dsigma1_sq,
                                     const float *dm_dsigma12, const float *scale, float scale_mul, float *v_render,
                                     void *workspace, int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(render && gt && exposure && v_render && C >= 1 && C < 65536 && H > 0 && W > 0 && CH >= 3);
    GSX_CHECK_ARG(mode >= 0 && mode <= 2 && (mode == 1 || (beta_index >= 3 && beta_index < CH)));
    GSX_CHECK_ARG(w_tv == 0.f || (alphas && depth_index >= 3 && depth_index < CH));
    GSX_CHECK_ARG(dm_dmu1 && dm_dsigma1_sq && dm_dsigma12 && scale && crop >= 0);
    const int64_t rows = gsx_ssim_bwd_map_loss_rows(C, H, W);
    if (!workspace || workspace_bytes < rows * gsx_loss::NPART * (int64_t)sizeof(float)) {
        gsx_set_error("gsx_ssim_bwd_map_loss: workspace too small");
        return GSX_E_WORKSPACE;
    }
    gsx_loss::LossArgs A;
    A.render = render; A.alphas = alphas; A.gt = gt; A.exposure = exposure; A.ssim_grad = nullptr;
    A.v_render = v_render; A.partials = (float *)workspace;
    A.H = H; A.W = W; A.CH = CH; A.depth_index = depth_index; A.beta_index = beta_index; A.mode = mode;
    A.w_photo = w_photo; A.w_tv = w_tv; A.mask_thresh = mask_thresh;
    const int gx = (W + TSX - 1) / TSX, gy = (H + TSY - 1) / TSY, gz = (int)C;
    GSX_CHECK_ARG((int64_t)gx * gy * gz < ((int64_t)1 << 31));
    hipLaunchKernelGGL(ssim_bwd_loss_kernel, dim3((unsigned)(gx * gy * gz)), dim3(256), 0, (hipStream_t)stream, A, crop, dm_dmu1,
                       dm_dsigma1_sq, dm_dsigma12, scale, scale_mul, gx, gy, gz);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int64_t gsx_ssim_workspace_bytes(int64_t B, int CH, int H, int W) {
    const int64_t blocks = B * CH * ((H + TSY - 1) / TSY) * ((W + TSX - 1) / TSX);
    return gsx_align256(blocks * (int64_t)sizeof(float)) + 256;
}

extern "C" int64_t gsx_ssim_partials(int64_t B, int CH, int H, int W) {
    return B * CH * ((H + TSY - 1) / TSY) * ((W + TSX - 1) / TSX);
}

extern "C" int gsx_ssim_fwd(const float *img1, const float *img2, int64_t B, int CH, int H, int W,
                            const int64_t *strides1, const int64_t *strides2, int crop, float *out_sum,
                            float *dm_dmu1, float *dm_dsigma1_sq, float *dm_dsigma12, void *workspace,
                            int64_t workspace_bytes, void *stream) {
    GSX_CHECK_ARG(img1 && img2 && strides1 && strides2 && B >= 1 && CH >= 1 && H > 0 && W > 0);
    GSX_CHECK_ARG(crop >= 0 && H > 2 * crop && W > 2 * crop);
    GSX_CHECK_ARG((dm_dmu1 == nullptr) == (dm_dsigma1_sq == nullptr) && (dm_dmu1 == nullptr) == (dm_dsigma12 == nullptr));
    if (!workspace || workspace_bytes < gsx_ssim_workspace_bytes(B, CH, H, W)) {
        gsx_set_error("gsx_ssim_fwd: workspace too small");
        return GSX_E_WORKSPACE;
    }
    hipStream_t st = (hipStream_t)stream;
    // forward: one channel per workgroup (26 KiB of LDS, six workgroups per CU) in channel-fastest XCD order - same HBM bytes
    // as three channels per workgroup (PMC), a little faster; the backward stages the image pixels of all three channels
    const bool three = false;
    const int gx = (W + TSX - 1) / TSX, gy = (H + TSY - 1) / TSY, gz = (int)(three ? B * CH / 3 : B * CH);
    GSX_CHECK_ARG((int64_t)gx * gy * gz < ((int64_t)1 << 31));
    const dim3 grid((unsigned)(gx * gy * gz));
    const Strides s1{strides1[0], strides1[1], strides1[2], strides1[3]};
    const Strides s2{strides2[0], strides2[1], strides2[2], strides2[3]};
    float *partials = (float *)workspace;
    if (three)
        hipLaunchKernelGGL(ssim_fwd_kernel<3>, grid, dim3(256), 0, st, img1, img2, CH, H, W, s1, s2, crop, partials,
                           dm_dmu1, dm_dsigma1_sq, dm_dsigma12, gx, gy, gz);
    else
        hipLaunchKernelGGL(ssim_fwd_kernel<1>, grid, dim3(256), 0, st, img1, img2, CH, H, W, s1, s2, crop, partials,
                           dm_dmu1, dm_dsigma1_sq, dm_dsigma12, gx, gy, gz);
    GSX_CHECK_LAUNCH();
    if (!out_sum) return GSX_OK;                             // deferred: gsx_loss_finish sums the gsx_ssim_partials() floats
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, partials,
                       gsx_ssim_partials(B, CH, H, W), out_sum);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_ssim_bwd(const float *img1, const float *img2, int64_t B, int CH, int H, int W,
                            const int64_t *strides1, const int64_t *strides2, int crop, const float *dm_dmu1,
                            const float *dm_dsigma1_sq, const float *dm_dsigma12, const float *scale, float scale_mul,
                            float *dL_dimg1, void *stream) {
    GSX_CHECK_ARG(img1 && img2 && strides1 && strides2 && dm_dmu1 && dm_dsigma1_sq && dm_dsigma12 && scale && dL_dimg1);
    GSX_CHECK_ARG(B >= 1 && CH >= 1 && H > 0 && W > 0 && crop >= 0 && B * CH < 65536);
    const bool three = (CH % 3) == 0;
    const int gx = (W + TSX - 1) / TSX, gy = (H + TSY - 1) / TSY, gz = (int)(three ? B * CH / 3 : B * CH);
    GSX_CHECK_ARG((int64_t)gx * gy * gz < ((int64_t)1 << 31));
    const dim3 grid((unsigned)(gx * gy * gz));
    const Strides s1{strides1[0], strides1[1], strides1[2], strides1[3]};
    const Strides s2{strides2[0], strides2[1], strides2[2], strides2[3]};
    if (three)
        hipLaunchKernelGGL(ssim_bwd_kernel<3>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, CH, H, W, s1, s2, crop,
                           dm_dmu1, dm_dsigma1_sq, dm_dsigma12, scale, scale_mul, dL_dimg1, gx, gy, gz);
    else
        hipLaunchKernelGGL(ssim_bwd_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, img1, img2, CH, H, W, s1, s2, crop,
                           dm_dmu1, dm_dsigma1_sq, dm_dsigma12, scale, scale_mul, dL_dimg1, gx, gy, gz);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
