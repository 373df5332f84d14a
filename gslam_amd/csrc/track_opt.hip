// track_opt.hip — device-resident tracking optimiser (track_opt.h) behind the C ABI: one single-lane launch per
// closure evaluation replaces the host-side torch.optim.Adam / torch.optim.LBFGS logic of
// gslam/frontend.py:604-662 and its per-closure `loss.item()` (frontend.py:648).
#include "gsx_common.h"
#include "track_opt.h"

namespace {

constexpr int TO_MAX_TENSORS = 4;

struct AdvanceArgs {
    float *params[TO_MAX_TENSORS];
    const float *grads[TO_MAX_TENSORS];
    int numels[TO_MAX_TENSORS];
    int n_tensors;
};

constexpr int STATE_WORDS = (int)((sizeof(TrackOptState) + 3) / 4);

__global__ __launch_bounds__(64) void track_opt_init_kernel(TrackOptState *state, int n, int n_adam, float lr_adam,
                                                            double lr, int history, int max_iter, int max_eval,
                                                            double tol_grad, double tol_change) {
    if (threadIdx.x == 0) to_init(state, n, n_adam, lr_adam, lr, history, max_iter, max_eval, tol_grad, tol_change);
}

// The state (a few KB) is staged through LDS by the whole wavefront; lane 0 runs the state machine on the LDS copy.
__global__ __launch_bounds__(64) void track_opt_advance_kernel(TrackOptState *state, AdvanceArgs a,
                                                               const float *__restrict__ loss) {
    __shared__ __attribute__((aligned(16))) uint32_t s_words[STATE_WORDS];
    __shared__ float s_p[TO_MAXN], s_g[TO_MAXN];
    uint32_t *gw = reinterpret_cast<uint32_t *>(state);
    for (int i = threadIdx.x; i < STATE_WORDS; i += 64) s_words[i] = gw[i];
    if (threadIdx.x < TO_MAXN) {
        int k = threadIdx.x, t = 0;
        float p = 0.f, g = 0.f;
        for (; t < a.n_tensors; ++t) {
            if (k < a.numels[t]) { p = a.params[t][k]; g = a.grads[t] ? a.grads[t][k] : 0.f; break; }
            k -= a.numels[t];
        }
        s_p[threadIdx.x] = p;
        s_g[threadIdx.x] = g;
    }
    __syncthreads();
    if (threadIdx.x == 0) to_advance(reinterpret_cast<TrackOptState *>(s_words), s_p, s_g, (double)loss[0]);
    __syncthreads();
    for (int i = threadIdx.x; i < STATE_WORDS; i += 64) gw[i] = s_words[i];
    if (threadIdx.x < TO_MAXN) {
        int k = threadIdx.x;
        for (int t = 0; t < a.n_tensors; ++t) {
            if (k < a.numels[t]) { a.params[t][k] = s_p[threadIdx.x]; break; }
            k -= a.numels[t];
        }
    }
}

__global__ __launch_bounds__(64) void track_opt_report_kernel(const TrackOptState *state, float *out) {
    if (threadIdx.x != 0) return;
    out[0] = (float)state->phase;
    out[1] = (float)state->total_evals;
    out[2] = (float)state->n_iter;
    out[3] = (float)state->stop_reason;
    out[4] = (float)state->last_eval_loss;
    out[5] = (float)state->loss;
    out[6] = (float)state->current_evals;
    out[7] = (float)state->adam_step;
}

}  // namespace

extern "C" int64_t gsx_track_opt_state_bytes(void) { return (int64_t)gsx_align256((int64_t)sizeof(TrackOptState)); }

extern "C" int gsx_track_opt_init(void *state, int n_params, int n_adam, float lr_adam, double lr_lbfgs, int history,
                                  int max_iter, int max_eval, double tol_grad, double tol_change, void *stream) {
    GSX_CHECK_ARG(state && n_params >= 1 && n_params <= TO_MAXN && n_adam >= 0 && history >= 1 && history <= TO_MAXH);
    GSX_CHECK_ARG(max_iter >= 1 && max_eval >= 2);
    hipLaunchKernelGGL(track_opt_init_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (TrackOptState *)state,
                       n_params, n_adam, lr_adam, lr_lbfgs, history, max_iter, max_eval, tol_grad, tol_change);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_track_opt_advance(void *state, int n_tensors, float *const *params, const float *const *grads,
                                     const int *numels, const float *loss, void *stream) {
    GSX_CHECK_ARG(state && params && grads && numels && loss && n_tensors >= 1 && n_tensors <= TO_MAX_TENSORS);
    AdvanceArgs a;
    int total = 0;
    for (int t = 0; t < TO_MAX_TENSORS; ++t) {
        a.params[t] = t < n_tensors ? params[t] : nullptr;
        a.grads[t] = t < n_tensors ? grads[t] : nullptr;
        a.numels[t] = t < n_tensors ? numels[t] : 0;
        if (t < n_tensors) {
            GSX_CHECK_ARG(params[t] && numels[t] >= 1);
            total += numels[t];
        }
    }
    GSX_CHECK_ARG(total <= TO_MAXN);
    a.n_tensors = n_tensors;
    hipLaunchKernelGGL(track_opt_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (TrackOptState *)state, a,
                       loss);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}

extern "C" int gsx_track_opt_report(const void *state, float *out8, void *stream) {
    GSX_CHECK_ARG(state && out8);
    hipLaunchKernelGGL(track_opt_report_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                       (const TrackOptState *)state, out8);
    GSX_CHECK_LAUNCH();
    return GSX_OK;
}
