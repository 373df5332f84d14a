// track_opt.h — device-resident optimiser of the tracking closure: the 10-step Adam warm-up followed by ONE
// torch.optim.LBFGS(line_search_fn='strong_wolfe').step() of gslam/frontend.py:604-662, restated as a state machine
// that is advanced once per closure evaluation.  The reference runs this logic on the host and pays one
// `loss.item()` synchronisation per closure (frontend.py:648); here the whole logic lives in one small device struct
// and one single-wavefront kernel per evaluation, so a tracked frame is a fixed sequence of graph launches with no
// read-back.
//
// The algorithm is torch's (torch/optim/lbfgs.py: `LBFGS.step`, `_strong_wolfe`, `_cubic_interpolate`; Adam with
// betas (0.9, 0.999), eps 1e-8), for n <= TO_MAXN parameters.  Scalars are double (torch keeps python floats),
// vectors float.  Plain C so that the same code compiles for the host in the tests (tests/trackopt_host.c).
#pragma once
#include <math.h>
#include <stdint.h>

#ifdef __HIPCC__
#define TO_FN __host__ __device__ static inline
#else
#define TO_FN static inline
#endif

/* Vector work.  On the device to_advance() is entered by one whole wavefront in lockstep: every lane carries the same
 * scalars and takes the same branches, lane l owns elements l, l + 64, ... of every vector (it is the only lane that
 * reads or writes them), and the reductions combine the per-lane partial sums with a butterfly, so that all lanes
 * hold the bit-identical result.  A single lane needed 50-300 us per L-BFGS direction update at 72 parameters.  On the
 * host there is one lane.  TO_EACH(i, n) walks the elements the calling lane owns. */
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define TO_LANE ((int)(threadIdx.x & 63u))
#define TO_LANES 64
/* Lane exchange inside a row of 16 by DPP on the two halves of the double (quad swaps, then the two mirrors: after the
 * four steps all 16 lanes of a row hold the row's result), rows combined through v_readlane in a fixed order.  An
 * L-BFGS iteration makes ~16 such reductions; through ds_bpermute they were most of the closure tail's time. */
template <int CTRL>
static __device__ __forceinline__ double to_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
static __device__ __forceinline__ double to_row_value(double v, int row) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), row * 16),
                            __builtin_amdgcn_readlane(__double2loint(v), row * 16));
}
TO_FN double to_lanes_sum(double v) {
    v += to_dpp<0xB1>(v);                      /* quad_perm [1,0,3,2] */
    v += to_dpp<0x4E>(v);                      /* quad_perm [2,3,0,1] */
    v += to_dpp<0x141>(v);                     /* row_half_mirror */
    v += to_dpp<0x140>(v);                     /* row_mirror */
    return (to_row_value(v, 0) + to_row_value(v, 1)) + (to_row_value(v, 2) + to_row_value(v, 3));
}
TO_FN double to_lanes_max(double v) {
    double w;
    w = to_dpp<0xB1>(v); v = w > v ? w : v;
    w = to_dpp<0x4E>(v); v = w > v ? w : v;
    w = to_dpp<0x141>(v); v = w > v ? w : v;
    w = to_dpp<0x140>(v); v = w > v ? w : v;
    const double a = to_row_value(v, 0), b = to_row_value(v, 1), c = to_row_value(v, 2), d = to_row_value(v, 3);
    const double ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}
#else
#define TO_LANE 0
#define TO_LANES 1
TO_FN double to_lanes_sum(double v) { return v; }
TO_FN double to_lanes_max(double v) { return v; }
#endif
#define TO_EACH(i, n) for (int i = TO_LANE; i < (n); i += TO_LANES)

/* Capacity of one instantiation; a translation unit may set both before including this header (window_opt.hip:
 * 8 poses x 9 parameters, history 10 for gslam/backend.py:447-506). */
#ifndef TO_MAXN
#define TO_MAXN 16   /* parameters: 3 (dt) + 6 (dR) + 2 (exposure) = 11 in the tracker */
#endif
#ifndef TO_MAXH
#define TO_MAXH 8    /* L-BFGS history (the tracker uses 5) */
#endif

enum { TO_PHASE_ADAM = 0, TO_PHASE_LBFGS_INIT = 1, TO_PHASE_LS_BRACKET = 2, TO_PHASE_LS_ZOOM = 3, TO_PHASE_DONE = 4 };

typedef struct {
    /* configuration */
    int32_t n, n_adam, history, max_iter, max_eval;
    float lr_adam, beta1, beta2, eps;
    double lr, tol_grad, tol_change, c1, c2;
    /* progress */
    int32_t phase, adam_step, n_iter, current_evals, n_hist, total_evals;
    int32_t ls_iter, ls_evals, max_ls, ls_first, ls_done, bracket_n, low_pos, high_pos, insuf_progress, stop_reason;
    /* Adam */
    float m[TO_MAXN], v[TO_MAXN];
    /* L-BFGS */
    float x[TO_MAXN];                 /* line-search base point x_init */
    float d[TO_MAXN];                 /* search direction */
    float g[TO_MAXN];                 /* flat_grad at the base point (the `g` of _strong_wolfe) */
    float prev_g[TO_MAXN];            /* prev_flat_grad */
    float old_dirs[TO_MAXH][TO_MAXN]; /* y */
    float old_stps[TO_MAXH][TO_MAXN]; /* s */
    double ro[TO_MAXH];
    double H_diag, t, loss, prev_loss, gtd, d_norm;
    /* line search */
    double t_prev, f_prev, gtd_prev;
    float g_prev[TO_MAXN];
    double bracket[2], bracket_f[2], bracket_gtd[2];
    float bracket_g[2][TO_MAXN];
    /* reporting */
    double last_eval_loss;            /* loss of the most recent closure (what the reference returns as last_loss) */
    /* temporaries of to_begin_iteration.  They live in the state, not on the stack: the device copy of the state sits
     * in LDS, while dynamically indexed stack arrays go to private (scratch) memory - the direction update then took
     * 30-50 us instead of ~10 on one lane. */
    double tmp_al[TO_MAXH];
    float tmp_y[TO_MAXN], tmp_s[TO_MAXN], tmp_q[TO_MAXN];
} TrackOptState;

TO_FN void to_init(TrackOptState *s, int n, int n_adam, float lr_adam, double lr, int history, int max_iter,
                   int max_eval, double tol_grad, double tol_change) {
    unsigned char *p = (unsigned char *)s;
    for (unsigned i = 0; i < sizeof(TrackOptState); ++i) p[i] = 0;
    s->n = n; s->n_adam = n_adam; s->history = history < TO_MAXH ? history : TO_MAXH;
    s->max_iter = max_iter; s->max_eval = max_eval;
    s->lr_adam = lr_adam; s->beta1 = 0.9f; s->beta2 = 0.999f; s->eps = 1e-8f;
    s->lr = lr; s->tol_grad = tol_grad; s->tol_change = tol_change; s->c1 = 1e-4; s->c2 = 0.9;
    s->phase = n_adam > 0 ? TO_PHASE_ADAM : TO_PHASE_LBFGS_INIT;
    s->H_diag = 1.0;
}

TO_FN double to_dot(const float *a, const float *b, int n) {
    double acc = 0.0;
    TO_EACH(i, n) acc += (double)a[i] * (double)b[i];
    return to_lanes_sum(acc);
}

TO_FN double to_absmax(const float *a, int n) {
    double m = 0.0;
    TO_EACH(i, n) { const double v = fabs((double)a[i]); if (v > m) m = v; }
    return to_lanes_max(m);
}

/* torch/optim/lbfgs.py:_cubic_interpolate */
TO_FN double to_cubic(double x1, double f1, double g1, double x2, double f2, double g2, int has_bounds, double lo,
                      double hi) {
    double xmin = lo, xmax = hi;
    if (!has_bounds) { if (x1 <= x2) { xmin = x1; xmax = x2; } else { xmin = x2; xmax = x1; } }
    /* torch evaluates this with g1, g2 as float32 tensors (dot products) and f, x as python floats: the python
     * scalar 3 (f1 - f2) / (x1 - x2) is rounded to float32 and the rest runs in float32.  The sign of d2_square is
     * decided by cancellation when the objective is nearly linear between the two points (the usual case at the
     * tracker's lr = 2e-3), so the same precision is kept here to take the same branch. */
    const float g1f = (float)g1, g2f = (float)g2;
    const float d1 = (g1f + g2f) - (float)(3.0 * (f1 - f2) / (x1 - x2));
    const float d2s = d1 * d1 - g1f * g2f;
    if (d2s >= 0.0f) {
        const float d2 = sqrtf(d2s);
        double mp;
        if (x1 <= x2) mp = x2 - (x2 - x1) * (double)((g2f + d2 - d1) / (g2f - g1f + 2.0f * d2));
        else mp = x1 - (x1 - x2) * (double)((g1f + d2 - d1) / (g1f - g2f + 2.0f * d2));
        /* min(max(mp, xmin), xmax) with python semantics for NaN: max(nan, a) = nan, min(nan, b) = nan */
        double r = (xmin > mp) ? xmin : mp;
        r = (xmax < r) ? xmax : r;
        return r;
    }
    return (xmin + xmax) / 2.0;
}

/* params <- x + t * d (LBFGS._directional_evaluate / _add_grad from x_init) */
TO_FN void to_place(const TrackOptState *s, float *params, double t) {
    TO_EACH(i, s->n) params[i] = (float)((double)s->x[i] + t * (double)s->d[i]);
}

TO_FN void to_finish(TrackOptState *s, int reason) { s->phase = TO_PHASE_DONE; s->stop_reason = reason; }

/* top of the `while n_iter < max_iter` body: needs s->loss / s->g at the current parameters `params`.  Leaves the
 * machine either DONE or waiting for the first evaluation of a line search (params already moved to x + t d). */
TO_FN void to_begin_iteration(TrackOptState *s, float *params) {
    const int n = s->n;
    s->n_iter += 1;
    if (s->n_iter == 1) {
        TO_EACH(i, n) s->d[i] = -s->g[i];
        s->n_hist = 0;
        s->H_diag = 1.0;
    } else {
        float *y = s->tmp_y, *sv = s->tmp_s;
        TO_EACH(i, n) { y[i] = s->g[i] - s->prev_g[i]; sv[i] = (float)((double)s->d[i] * s->t); }
        const double ys = to_dot(y, sv, n);
        if (ys > 1e-10) {
            if (s->n_hist == s->history) {
                for (int h = 1; h < s->n_hist; ++h) {
                    TO_EACH(i, n) { s->old_dirs[h - 1][i] = s->old_dirs[h][i]; s->old_stps[h - 1][i] = s->old_stps[h][i]; }
                    s->ro[h - 1] = s->ro[h];
                }
                s->n_hist -= 1;
            }
            TO_EACH(i, n) { s->old_dirs[s->n_hist][i] = y[i]; s->old_stps[s->n_hist][i] = sv[i]; }
            s->ro[s->n_hist] = 1.0 / ys;
            s->n_hist += 1;
            s->H_diag = ys / to_dot(y, y, n);
        }
        double *al = s->tmp_al;
        float *q = s->tmp_q;
        TO_EACH(i, n) q[i] = -s->g[i];
        for (int h = s->n_hist - 1; h >= 0; --h) {
            al[h] = to_dot(s->old_stps[h], q, n) * s->ro[h];
            TO_EACH(i, n) q[i] = (float)((double)q[i] - al[h] * (double)s->old_dirs[h][i]);
        }
        TO_EACH(i, n) q[i] = (float)((double)q[i] * s->H_diag);
        for (int h = 0; h < s->n_hist; ++h) {
            const double be = to_dot(s->old_dirs[h], q, n) * s->ro[h];
            TO_EACH(i, n) q[i] = (float)((double)q[i] + (al[h] - be) * (double)s->old_stps[h][i]);
        }
        TO_EACH(i, n) s->d[i] = q[i];
    }
    TO_EACH(i, n) s->prev_g[i] = s->g[i];
    s->prev_loss = s->loss;
    if (s->n_iter == 1) {
        double l1 = 0.0;
        TO_EACH(i, n) l1 += fabs((double)s->g[i]);
        l1 = to_lanes_sum(l1);
        const double inv = 1.0 / l1;
        s->t = (inv < 1.0 ? inv : 1.0) * s->lr;
    } else {
        s->t = s->lr;
    }
    s->gtd = to_dot(s->g, s->d, n);
    if (s->gtd > -s->tol_change) { to_finish(s, 1); return; }
    /* _strong_wolfe(obj_func, x_init, t, d, loss, flat_grad, gtd, max_ls = max_eval - current_evals) */
    TO_EACH(i, n) s->x[i] = params[i];
    s->d_norm = to_absmax(s->d, n);
    s->max_ls = s->max_eval - s->current_evals;
    s->ls_iter = 0; s->ls_evals = 0; s->ls_first = 1; s->ls_done = 0; s->insuf_progress = 0; s->bracket_n = 0;
    s->t_prev = 0.0; s->f_prev = s->loss; s->gtd_prev = s->gtd;
    TO_EACH(i, n) s->g_prev[i] = s->g[i];
    to_place(s, params, s->t);
    s->phase = TO_PHASE_LS_BRACKET;
}

/* after the line search returned (t, f_new, g_new) = the low end of the bracket */
TO_FN void to_after_line_search(TrackOptState *s, float *params) {
    const int n = s->n;
    const int lp = s->low_pos;
    s->t = s->bracket[lp];
    s->loss = s->bracket_f[lp];
    TO_EACH(i, n) s->g[i] = s->bracket_g[lp][i];
    to_place(s, params, s->t);                     /* self._add_grad(t, d) from x_init */
    const int opt_cond = to_absmax(s->g, n) <= s->tol_grad;
    s->current_evals += s->ls_evals;
    if (s->n_iter == s->max_iter) { to_finish(s, 2); return; }
    if (s->current_evals >= s->max_eval) { to_finish(s, 3); return; }
    if (opt_cond) { to_finish(s, 4); return; }
    if (to_absmax(s->d, n) * fabs(s->t) <= s->tol_change) { to_finish(s, 5); return; }
    if (fabs(s->loss - s->prev_loss) < s->tol_change) { to_finish(s, 6); return; }
    to_begin_iteration(s, params);
}

/* top of the zoom loop: either asks for one more evaluation (phase LS_ZOOM) or ends the line search */
TO_FN void to_zoom_next(TrackOptState *s, float *params) {
    if (!s->ls_done && s->ls_iter < s->max_ls && s->bracket_n == 2) {
        const double b0 = s->bracket[0], b1 = s->bracket[1];
        if (fabs(b1 - b0) * s->d_norm >= s->tol_change) {
            double t = to_cubic(b0, s->bracket_f[0], s->bracket_gtd[0], b1, s->bracket_f[1], s->bracket_gtd[1], 0, 0.0,
                                0.0);
            const double bmax = b0 > b1 ? b0 : b1, bmin = b0 > b1 ? b1 : b0;
            const double eps = 0.1 * (bmax - bmin);
            const double a1 = bmax - t, a2 = t - bmin;
            if ((a1 < a2 ? a1 : a2) < eps) {
                if (s->insuf_progress || t >= bmax || t <= bmin) {
                    if (fabs(t - bmax) < fabs(t - bmin)) t = bmax - eps; else t = bmin + eps;
                    s->insuf_progress = 0;
                } else {
                    s->insuf_progress = 1;
                }
            } else {
                s->insuf_progress = 0;
            }
            s->t = t;
            to_place(s, params, t);
            s->phase = TO_PHASE_LS_ZOOM;
            return;
        }
    }
    to_after_line_search(s, params);
}

TO_FN void to_set_bracket2(TrackOptState *s, const float *g_new, double f_new, double gtd_new) {
    const int n = s->n;
    s->bracket_n = 2;
    s->bracket[0] = s->t_prev; s->bracket[1] = s->t;
    s->bracket_f[0] = s->f_prev; s->bracket_f[1] = f_new;
    s->bracket_gtd[0] = s->gtd_prev; s->bracket_gtd[1] = gtd_new;
    TO_EACH(i, n) { s->bracket_g[0][i] = s->g_prev[i]; s->bracket_g[1][i] = g_new[i]; }
}

/* One evaluation of the closure at `params` gave (`loss`, `grad`): advance.  On return `params` hold the point of
 * the next evaluation, or - once s->phase == TO_PHASE_DONE - the result.  Calls after DONE change nothing. */
TO_FN void to_advance(TrackOptState *s, float *params, const float *grad, double loss) {
    const int n = s->n;
    if (s->phase == TO_PHASE_DONE) return;
    s->last_eval_loss = loss;
    s->total_evals += 1;
    if (s->phase == TO_PHASE_ADAM) {
        s->adam_step += 1;
        const double bc1 = 1.0 - pow((double)s->beta1, (double)s->adam_step);
        const double bc2 = 1.0 - pow((double)s->beta2, (double)s->adam_step);
        TO_EACH(i, n) {
            const float gi = grad[i];
            s->m[i] = s->beta1 * s->m[i] + (1.0f - s->beta1) * gi;
            s->v[i] = s->beta2 * s->v[i] + (1.0f - s->beta2) * gi * gi;
            const double denom = sqrt((double)s->v[i]) / sqrt(bc2) + (double)s->eps;
            params[i] = (float)((double)params[i] - ((double)s->lr_adam / bc1) * ((double)s->m[i] / denom));
        }
        if (s->adam_step >= s->n_adam) s->phase = TO_PHASE_LBFGS_INIT;
        return;
    }
    if (s->phase == TO_PHASE_LBFGS_INIT) {
        s->loss = loss;
        s->current_evals = 1;
        TO_EACH(i, n) s->g[i] = grad[i];
        if (to_absmax(s->g, n) <= s->tol_grad) { to_finish(s, 7); return; }
        s->n_iter = 0;
        to_begin_iteration(s, params);
        return;
    }
    const double f_new = loss;
    const double gtd_new = to_dot(grad, s->d, n);
    s->ls_evals += 1;
    if (s->phase == TO_PHASE_LS_BRACKET) {
        if (s->ls_first) s->ls_first = 0; else s->ls_iter += 1;
        if (s->ls_iter < s->max_ls) {
            if (f_new > (s->loss + s->c1 * s->t * s->gtd) || (s->ls_iter > 1 && f_new >= s->f_prev)) {
                to_set_bracket2(s, grad, f_new, gtd_new);
            } else if (fabs(gtd_new) <= -s->c2 * s->gtd) {
                s->bracket_n = 1;
                s->bracket[0] = s->t; s->bracket_f[0] = f_new;
                TO_EACH(i, n) s->bracket_g[0][i] = grad[i];
                s->ls_done = 1;
            } else if (gtd_new >= 0.0) {
                to_set_bracket2(s, grad, f_new, gtd_new);
            } else {
                const double min_step = s->t + 0.01 * (s->t - s->t_prev);
                const double max_step = s->t * 10.0;
                const double tmp = s->t;
                s->t = to_cubic(s->t_prev, s->f_prev, s->gtd_prev, s->t, f_new, gtd_new, 1, min_step, max_step);
                s->t_prev = tmp; s->f_prev = f_new; s->gtd_prev = gtd_new;
                TO_EACH(i, n) s->g_prev[i] = grad[i];
                to_place(s, params, s->t);
                return;                                  /* next bracket evaluation */
            }
        } else {                                         /* ls_iter == max_ls: bracket = [0, t] */
            s->bracket_n = 2;
            s->bracket[0] = 0.0; s->bracket[1] = s->t;
            s->bracket_f[0] = s->loss; s->bracket_f[1] = f_new;
            s->bracket_gtd[0] = s->gtd; s->bracket_gtd[1] = gtd_new;
            TO_EACH(i, n) { s->bracket_g[0][i] = s->g[i]; s->bracket_g[1][i] = grad[i]; }
        }
        if (s->bracket_f[0] <= s->bracket_f[s->bracket_n - 1]) { s->low_pos = 0; s->high_pos = 1; }
        else { s->low_pos = 1; s->high_pos = 0; }
        s->insuf_progress = 0;
        to_zoom_next(s, params);
        return;
    }
    /* TO_PHASE_LS_ZOOM */
    s->ls_iter += 1;
    {
        const double t = s->t;
        int lp = s->low_pos, hp = s->high_pos;
        if (f_new > (s->loss + s->c1 * t * s->gtd) || f_new >= s->bracket_f[lp]) {
            s->bracket[hp] = t; s->bracket_f[hp] = f_new; s->bracket_gtd[hp] = gtd_new;
            TO_EACH(i, n) s->bracket_g[hp][i] = grad[i];
            if (s->bracket_f[0] <= s->bracket_f[1]) { s->low_pos = 0; s->high_pos = 1; }
            else { s->low_pos = 1; s->high_pos = 0; }
        } else {
            if (fabs(gtd_new) <= -s->c2 * s->gtd) {
                s->ls_done = 1;
            } else if (gtd_new * (s->bracket[hp] - s->bracket[lp]) >= 0.0) {
                s->bracket[hp] = s->bracket[lp]; s->bracket_f[hp] = s->bracket_f[lp];
                s->bracket_gtd[hp] = s->bracket_gtd[lp];
                TO_EACH(i, n) s->bracket_g[hp][i] = s->bracket_g[lp][i];
            }
            s->bracket[lp] = t; s->bracket_f[lp] = f_new; s->bracket_gtd[lp] = gtd_new;
            TO_EACH(i, n) s->bracket_g[lp][i] = grad[i];
        }
    }
    to_zoom_next(s, params);
}
