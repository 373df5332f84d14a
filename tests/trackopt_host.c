/* Host build of the tracking-optimiser state machine (gslam_amd/csrc/track_opt.h), for the CPU tests only: the
 * same C code the HIP kernel runs, compiled by gcc, driven by tests/test_track_opt_cpu.py against
 * torch.optim.Adam / torch.optim.LBFGS. */
#include "../gslam_amd/csrc/track_opt.h"

long trackopt_state_bytes(void) { return (long)sizeof(TrackOptState); }

void trackopt_init(void *s, int n, int n_adam, float lr_adam, double lr, int history, int max_iter, int max_eval,
                   double tol_grad, double tol_change) {
    to_init((TrackOptState *)s, n, n_adam, lr_adam, lr, history, max_iter, max_eval, tol_grad, tol_change);
}

void trackopt_advance(void *s, float *params, const float *grad, double loss) {
    to_advance((TrackOptState *)s, params, grad, loss);
}

int trackopt_phase(const void *s) { return ((const TrackOptState *)s)->phase; }
int trackopt_evals(const void *s) { return ((const TrackOptState *)s)->total_evals; }
int trackopt_iters(const void *s) { return ((const TrackOptState *)s)->n_iter; }
int trackopt_stop_reason(const void *s) { return ((const TrackOptState *)s)->stop_reason; }
double trackopt_loss(const void *s) { return ((const TrackOptState *)s)->loss; }
