// maintain.hip — map maintenance between BA iterations: the row-wise re-packing of every
// per-Gaussian array that gslam/pruning.py:10-55 (prune_using_mask) and gslam/insertion.py:27-100 (_add_new_splats,
// _duplicate, _split) do with one torch indexing / cat kernel per tensor - 7 parameters + 2 Adam moments for each of
// the 6 optimised ones = 19 arrays, i.e. 19-38 launches per call - as ONE launch over all arrays.
#include "gsx_common.h"

namespace {

constexpr int ROWS_MAX = 32;

struct RowsArgs {
    const uint32_t *src[ROWS_MAX];
    const uint32_t *src_b[ROWS_MAX];   // concat: second source
    uint32_t *dst[ROWS_MAX];
    int start[ROWS_MAX + 1];           // exclusive prefix of the row widths (4-byte words)
    int n;
};

__device__ __forceinline__ int find_tensor(const RowsArgs &a, int w) {
    int lo = 0, hi = a.n - 1;
#pragma unroll
    for (int it = 0; it < 5; ++it) {
        const int mid = (lo + hi + 1) >> 1;
        if (a.start[mid] <= w) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// one thread per (output row, word of the concatenated row); index[r] = source row of output row r
__global__ __launch_bounds__(256) void gather_rows_kernel(RowsArgs a, const int64_t *__restrict__ index, int64_t n_out,
                                                          int64_t n_src) {
    const int total = a.start[a.n];
    const int64_t n_items = n_out * total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / total;
        const int w = (int)(i - r * total);
        const int k = find_tensor(a, w);
        const int width = a.start[k + 1] - a.start[k];
        int64_t s = index[r];
        s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);      // never read out of bounds on a bad index
        a.dst[k][r * width + (w - a.start[k])] = a.src[k][s * width + (w - a.start[k])];
    }
}

// dst = [a rows (n_a) ; b rows (n_b)]; a NULL b source means zero rows (fresh Adam moments, insertion.py:52-56)
__global__ __launch_bounds__(256) void concat_rows_kernel(RowsArgs a, int64_t n_a, int64_t n_b) {
    const int total = a.start[a.n];
    const int64_t n_items = (n_a + n_b) * total;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_items; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / total;
        const int w = (int)(i - r * total);
        const int k = find_tensor(a, w);
        const int width = a.start[k + 1] - a.start[k];
        const int c = w - a.start[k];
        uint32_t v;
        if (r < n_a) v = a.src[k][r * width + c];
        else v = a.src_b[k] ? a.src_b[k][(r - n_a) * width + c] : 0u;
        a.dst[k][r * width + c] = v;
    }
}

int fill_args(RowsArgs &a, int n_tensors, const void *const *src, const void *const *src_b, void *const *dst,
              const int *row_words) {
    a.n = n_tensors;
    a.start[0] = 0;
    for (int k = 0; k < ROWS_MAX; ++k) {
        const bool in = k < n_tensors;
        a.src[k] = in ? (const uint32_t *)src[k] : nullptr;
        a.src_b[k] = (in && src_b) ? (const uint32_t *)src_b[k] : nullptr;
        a.dst[k] = in ? (uint32_t *)dst[k] : nullptr;
        a.start[k + 1] = a.start[k] + (in ? row_words[k] : 0);
        if (in && (row_words[k] < 1 || !dst[k])) return -1;
    }
    return 0;
}

}  // namespace

extern "C" int gsx_gather_rows(int n_tensors, const void *const *src, void *const *dst, const int *row_words,
                               const int64_t *index, int64_t n_out, int64_t n_src, void *stream) {
    GSX_CHECK_ARG(n_tensors >= 1 && n_tensors <= ROWS_MAX && src && dst && row_words && n_out >= 0 && n_src >= 0);
    if (n_out == 0) return GSX_OK;
    GSX_CHECK_ARG(index && n_src >= 1);
    RowsArgs a;
    GSX_CHECK_ARG(fill_args(a, n_tensors, src, nullptr, dst, row_words) == 0);
    for (int k = 0; k < n_tensors; ++k) GSX_CHECK_ARG(src[k]);
    const int64_t items = n_out * a.start[n_tensors];
    int64_t blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, index, n_out,
                       n_src);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_concat_rows(int n_tensors, const void *const *a_rows, int64_t n_a, const void *const *b_rows,
                               int64_t n_b, void *const *dst, const int *row_words, void *stream) {
    GSX_CHECK_ARG(n_tensors >= 1 && n_tensors <= ROWS_MAX && a_rows && b_rows && dst && row_words && n_a >= 0 && n_b >= 0);
    if (n_a + n_b == 0) return GSX_OK;
    RowsArgs a;
    GSX_CHECK_ARG(fill_args(a, n_tensors, a_rows, b_rows, dst, row_words) == 0);
    for (int k = 0; k < n_tensors; ++k) GSX_CHECK_ARG(n_a == 0 || a_rows[k]);
    const int64_t items = (n_a + n_b) * a.start[n_tensors];
    int64_t blocks = (items + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(concat_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, n_a, n_b);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
