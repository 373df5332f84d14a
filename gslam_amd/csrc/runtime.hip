// runtime.hip — the part of the C ABI that is about HIP streams and HIP graphs rather than kernels.
//
// The reference drives every closure of its optimisers from Python, one launch and (for the loss) one read-back at a time
// (gslam/frontend.py:621-649, gslam/backend.py:260-359,465-504).  Here a closure is a fixed chain of libgsx launches over
// caller-owned, persistent device buffers (gslam_amd/plan.py); that chain is recorded ONCE into a hipGraph by capturing the
// stream it is issued on and replayed with one call.  Capture, instantiation and launch go through these entry points -
// plain HIP, no framework graph object, no allocator hooks: nothing is allocated or freed while a stream captures.
#include "gsx_common.h"

#include <string.h>

#define GSX_HIP(call)                                                                             \
    do {                                                                                          \
        hipError_t e__ = (call);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            gsx_set_error("%s:%d: %s: %s", __FILE__, __LINE__, #call, hipGetErrorString(e__));    \
            (void)hipGetLastError();                                                              \
            return GSX_E_LAUNCH;                                                                  \
        }                                                                                         \
    } while (0)

extern "C" int gsx_stream_create(void **stream_out) {
    GSX_CHECK_ARG(stream_out != nullptr);
    hipStream_t s = nullptr;
    GSX_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream_out = (void *)s;
    return GSX_OK;
}

// A stream whose kernels may only use the compute units whose bit is set (hipExtStreamCreateWithCUMask; mask words are 32 CUs each).
// For a stream of long, chip-filling launches that runs beside a chain of short dependent ones (the mapping stream beside the
// tracker): the CUs it leaves alone are where the chain's next launch starts at once instead of waiting for a workgroup to drain.
extern "C" int gsx_stream_create_masked(void **stream_out, const uint32_t *cu_mask, int mask_words) {
    GSX_CHECK_ARG(stream_out != nullptr && cu_mask != nullptr && mask_words >= 1 && mask_words <= 32);
    hipStream_t s = nullptr;
    GSX_HIP(hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask));
    *stream_out = (void *)s;
    return GSX_OK;
}

extern "C" int gsx_stream_destroy(void *stream) {
    if (stream == nullptr) return GSX_OK;
    GSX_HIP(hipStreamDestroy((hipStream_t)stream));
    return GSX_OK;
}

extern "C" int gsx_stream_synchronize(void *stream) {
    GSX_HIP(hipStreamSynchronize((hipStream_t)stream));
    return GSX_OK;
}

// `stream` waits (on the device, no host sync) for everything issued so far on `other`
extern "C" int gsx_stream_wait_stream(void *stream, void *other) {
    hipEvent_t ev = nullptr;
    GSX_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)other);
    if (e == hipSuccess) e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
    (void)hipEventDestroy(ev);      // released once the recorded work has completed
    if (e != hipSuccess) {
        gsx_set_error("gsx_stream_wait_stream: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return GSX_E_LAUNCH;
    }
    return GSX_OK;
}

// mode: 0 = global, 1 = thread-local (other threads of the process may allocate / synchronise meanwhile), 2 = relaxed
extern "C" int gsx_graph_begin(void *stream, int mode) {
    GSX_CHECK_ARG(stream != nullptr);      // the legacy default stream cannot capture
    GSX_CHECK_ARG(mode >= 0 && mode <= 2);
    const hipStreamCaptureMode m = mode == 0 ? hipStreamCaptureModeGlobal
                                 : mode == 1 ? hipStreamCaptureModeThreadLocal : hipStreamCaptureModeRelaxed;
    GSX_HIP(hipStreamBeginCapture((hipStream_t)stream, m));
    return GSX_OK;
}

// ends the capture and instantiates the recorded chain; *n_nodes_out (nullable) = number of graph nodes.  On any error
// the capture is over and nothing is left allocated.
extern "C" int gsx_graph_end(void *stream, void **exec_out, int64_t *n_nodes_out) {
    GSX_CHECK_ARG(stream != nullptr && exec_out != nullptr);
    *exec_out = nullptr;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess || g == nullptr) {
        gsx_set_error("gsx_graph_end: hipStreamEndCapture: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        return GSX_E_LAUNCH;
    }
    if (n_nodes_out) {
        size_t n = 0;
        if (hipGraphGetNodes(g, nullptr, &n) == hipSuccess) *n_nodes_out = (int64_t)n;
        else { *n_nodes_out = -1; (void)hipGetLastError(); }
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        gsx_set_error("gsx_graph_end: hipGraphInstantiate: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return GSX_E_LAUNCH;
    }
    *exec_out = (void *)ex;
    return GSX_OK;
}

// abandons a capture after a failed launch inside it (the stream leaves capture mode, nothing is instantiated)
extern "C" int gsx_graph_abort(void *stream) {
    hipGraph_t g = nullptr;
    (void)hipStreamEndCapture((hipStream_t)stream, &g);
    if (g) (void)hipGraphDestroy(g);
    (void)hipGetLastError();
    return GSX_OK;
}

extern "C" int gsx_graph_launch(void *exec, void *stream) {
    GSX_CHECK_ARG(exec != nullptr);
    GSX_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
    return GSX_OK;
}

// `count` launches back to back (one tracked frame = n_adam + max_eval + 1 replays of one closure graph)
extern "C" int gsx_graph_launch_n(void *exec, int count, void *stream) {
    GSX_CHECK_ARG(exec != nullptr && count >= 0);
    for (int i = 0; i < count; ++i) GSX_HIP(hipGraphLaunch((hipGraphExec_t)exec, (hipStream_t)stream));
    return GSX_OK;
}

extern "C" int gsx_graph_destroy(void *exec) {
    if (exec == nullptr) return GSX_OK;
    GSX_HIP(hipGraphExecDestroy((hipGraphExec_t)exec));
    return GSX_OK;
}

// ---- pinned host words the device can write and the host can poll without a stream sync (closure status / M) ------
extern "C" int gsx_host_alloc(void **host_out, void **dev_out, int64_t bytes) {
    GSX_CHECK_ARG(host_out != nullptr && dev_out != nullptr && bytes > 0);
    void *h = nullptr, *d = nullptr;
    GSX_HIP(hipHostMalloc(&h, (size_t)bytes, hipHostMallocMapped));
    hipError_t e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) {
        (void)hipHostFree(h);
        gsx_set_error("gsx_host_alloc: hipHostGetDevicePointer: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return GSX_E_LAUNCH;
    }
    memset(h, 0, (size_t)bytes);
    *host_out = h;
    *dev_out = d;
    return GSX_OK;
}

extern "C" int gsx_host_free(void *host) {
    if (host == nullptr) return GSX_OK;
    GSX_HIP(hipHostFree(host));
    return GSX_OK;
}

// zero fill by a kernel (see gsx_common.h: memset nodes misbehave in replayed graphs on this runtime)
extern "C" int gsx_zero_words(void *ptr, int64_t n_words, void *stream) {
    GSX_CHECK_ARG(n_words >= 0 && (ptr != nullptr || n_words == 0) && (((uintptr_t)ptr) & 3) == 0);
    if (!gsx_zero_async(ptr, n_words, (hipStream_t)stream)) {
        gsx_set_error("gsx_zero_words: launch failed");
        return GSX_E_LAUNCH;
    }
    return GSX_OK;
}


// ---- where does the dispatcher put workgroup i?  (self-check of the CU-balanced launch order, tile_balance.h) ------------
// The balanced order relies on a property the hardware does not document: on a drained chip whose launch fits at once,
// workgroups i, i + G, i + 2 G, ... (G = compute units) land on the same compute unit.  This probe launches `n_wgs` workgroups
// of the rasteriser's shape (256 threads, `lds_bytes` of LDS) that stay resident for `spin_us` and record where they run:
// key = XCC_ID << 8 | SE_ID << 5 | SH_ID << 4 | CU_ID (HW_ID register).  The host checks the pattern and falls back to the
// identity order, with the reason logged, when it does not hold (another firmware, partition mode or a busy chip).
namespace {
__global__ __launch_bounds__(256) void wg_placement_probe_kernel(int32_t *__restrict__ keys, int spin_ticks) {
    extern __shared__ int s_probe[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
        const unsigned cu = (hw >> 8) & 15u, sh = (hw >> 12) & 1u, se = (hw >> 13) & 7u;
        keys[blockIdx.x] = (int32_t)(((xcc & 15u) << 8) | (se << 5) | (sh << 4) | cu);
        s_probe[0] = (int)hw;                                  // (keeps the dynamic LDS allocation alive)
    }
    while ((long long)(__builtin_amdgcn_s_memrealtime() - t0) < (long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
}
}  // namespace

extern "C" int gsx_probe_wg_placement(int n_wgs, int lds_bytes, int spin_us, int32_t *keys, void *stream) {
    GSX_CHECK_ARG(n_wgs >= 1 && n_wgs <= 65536 && lds_bytes >= 4 && lds_bytes <= 65536 && spin_us >= 0 && spin_us <= 1000 && keys);
    hipLaunchKernelGGL(wg_placement_probe_kernel, dim3((unsigned)n_wgs), dim3(256), (size_t)lds_bytes, (hipStream_t)stream,
                       keys, spin_us * 100);                      // s_memrealtime counts at 100 MHz
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
