/*
 * gsx_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the algorithms on the gslam hot path
 * (SURVEY.md §8).  It is the checker the parity tests compare the HIP kernels
 * against.  It is NOT part of the product: nothing under gslam_amd/ may import,
 * link or call it.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load the library built from this file.
 *
 * PARITY UNPINNED: the arithmetic of the reference's hot path lives in two
 * third-party CUDA packages that are absent from /root/reference
 *   - gsplat fork github.com/abhigyan7/gsplat, branch "gslam", unpinned
 *     (base ~ upstream v1.4.x, commit 7951619...), called at
 *     gslam/rasterization.py:9-14,153-170,261-274,325-339
 *   - fused-ssim github.com/rahul-goel/fused-ssim @ 30fb258c, called at
 *     gslam/backend.py:13,303-307
 * and the reference holds no tests / golden vectors for this path.
 * The kernel maths below restates the published algorithm of those packages
 *  and is anchored on the reference's call sites.  The parts of
 * the path whose source IS in /root/reference (gslam/warp.py, gslam/utils.py,
 * the host logic of gslam/rasterization.py) are pinned by golden vectors made
 * from the importable reference (oracle/gen_golden.py -> tests/golden/).
 *
 * Build: see oracle/Makefile.  Compiled twice: REAL=float (the oracle proper,
 * -ffp-contract=off so integer outputs are reproducible) and REAL=double
 * (finite-difference checks of the hand-derived VJPs).
 *
 * Third build, for bench.py's cpu_baseline leg ONLY: -fopenmp (libgsx_oracle_f32_omp.so): the loops over Gaussians,
 * image rows and tiles below carry OpenMP pragmas (all host cores).  Without -fopenmp the pragmas are
 * no-ops and the two oracle builds are the serial code they always were; with it, per-Gaussian gradient sums use
 * `omp atomic` (their summation order - not their value beyond float round-off - then depends on the schedule, which is
 * why the tests never load that build).
 *
 * All pointers are host pointers.  Every function returns 0 on success.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
int gsxo_omp_threads(void) { return omp_get_max_threads(); }
/* bench.py's cpu_baseline: a 10 k-Gaussian problem does not feed 128 threads (its time moved 20x from run to run with all of
 * them: fork / join and cache-line ping-pong, not work) - the caller sets how many the next calls use and says so */
void gsxo_omp_set_threads(int n) { if (n >= 1) omp_set_num_threads(n); }
#else
int gsxo_omp_threads(void) { return 1; }
void gsxo_omp_set_threads(int n) { (void)n; }
#endif

#ifdef GSXO_DOUBLE
typedef double real;
#define R_SQRT sqrt
#define R_EXP exp
#define R_LOG log
#define R_CEIL ceil
#define R_FLOOR floor
#define R_FABS fabs
#define R_FMIN fmin
#define R_FMAX fmax
#else
typedef float real;
#define R_SQRT sqrtf
#define R_EXP expf
#define R_LOG logf
#define R_CEIL ceilf
#define R_FLOOR floorf
#define R_FABS fabsf
#define R_FMIN fminf
#define R_FMAX fmaxf
#endif

#define RC(x) ((real)(x))

/* ---- named constants of the external kernels (INFERRED) ---- */
#define GSXO_FOV_SLACK RC(0.3)      /* frustum clamp slack, x tan(fov/2)        */
#define GSXO_RADIUS_FLOOR RC(0.01)  /* max(0.01, b^2-det) in the radius formula */
#define GSXO_RADIUS_SIGMA RC(3.0)   /* 3-sigma extent                           */
#define GSXO_ALPHA_MAX RC(0.999)
#define GSXO_ALPHA_MIN (RC(1.0) / RC(255.0))
#define GSXO_T_MIN RC(1e-4)
#define GSXO_SSIM_C1 RC(0.0001)
#define GSXO_SSIM_C2 RC(0.0009)

int gsxo_sizeof_real(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------------ */
/* K10  quat_scale_to_covar_preci   (gslam/insertion.py:88-91)              */
/* ------------------------------------------------------------------------ */
static void quat_to_rotmat(const real *q, real *R /*9 row-major*/, real *qn_out, real *inv_norm_out) {
    real w = q[0], x = q[1], y = q[2], z = q[3];
    real n2 = w * w + x * x + y * y + z * z;
    real inv = RC(1.0) / R_SQRT(n2);
    w *= inv; x *= inv; y *= inv; z *= inv;
    if (qn_out) { qn_out[0] = w; qn_out[1] = x; qn_out[2] = y; qn_out[3] = z; }
    if (inv_norm_out) *inv_norm_out = inv;
    real x2 = x * x, y2 = y * y, z2 = z * z;
    real xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    R[0] = RC(1.0) - RC(2.0) * (y2 + z2); R[1] = RC(2.0) * (xy - wz); R[2] = RC(2.0) * (xz + wy);
    R[3] = RC(2.0) * (xy + wz); R[4] = RC(1.0) - RC(2.0) * (x2 + z2); R[5] = RC(2.0) * (yz - wx);
    R[6] = RC(2.0) * (xz - wy); R[7] = RC(2.0) * (yz + wx); R[8] = RC(1.0) - RC(2.0) * (x2 + y2);
}

/* Sigma = (Rq diag(s)) (Rq diag(s))^T, 6 unique entries: 00 01 02 11 12 22 */
static void quat_scale_to_covar(const real *q, const real *s, real *cov6, real *M /*9*/, real *Rq /*9*/) {
    real Rl[9], Ml[9];
    if (!Rq) Rq = Rl;
    if (!M) M = Ml;
    quat_to_rotmat(q, Rq, NULL, NULL);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = Rq[i * 3 + j] * s[j];
    int k = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j)
            cov6[k++] = M[i * 3 + 0] * M[j * 3 + 0] + M[i * 3 + 1] * M[j * 3 + 1] + M[i * 3 + 2] * M[j * 3 + 2];
}

int gsxo_quat_scale_to_covar_preci(int64_t n, const real *quats, const real *scales, real *covars /*[n,3,3]*/,
                                   real *precis /*[n,3,3] or NULL*/) {
    for (int64_t i = 0; i < n; ++i) {
        real c6[6];
        quat_scale_to_covar(quats + 4 * i, scales + 3 * i, c6, NULL, NULL);
        real *C = covars + 9 * i;
        C[0] = c6[0]; C[1] = c6[1]; C[2] = c6[2];
        C[3] = c6[1]; C[4] = c6[3]; C[5] = c6[4];
        C[6] = c6[2]; C[7] = c6[4]; C[8] = c6[5];
        if (precis) {
            /* precision = (Rq diag(1/s)) (Rq diag(1/s))^T */
            real inv_s[3] = {RC(1.0) / scales[3 * i], RC(1.0) / scales[3 * i + 1], RC(1.0) / scales[3 * i + 2]};
            real p6[6];
            quat_scale_to_covar(quats + 4 * i, inv_s, p6, NULL, NULL);
            real *P = precis + 9 * i;
            P[0] = p6[0]; P[1] = p6[1]; P[2] = p6[2];
            P[3] = p6[1]; P[4] = p6[3]; P[5] = p6[4];
            P[6] = p6[2]; P[7] = p6[4]; P[8] = p6[5];
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K1  fully_fused_projection forward  (gslam/rasterization.py:153-170;     */
/*     maths: SURVEY.md §9.1)                                               */
/* ------------------------------------------------------------------------ */
typedef struct {
    real pc[3];           /* camera-space mean */
    real Rq[9], M[9];     /* quat rotation, Rq*diag(s) */
    real S[6];            /* world covariance (00 01 02 11 12 22) */
    real Sc[6];           /* camera covariance */
    real J00, J11, J02, J12, tx, ty, rz;
    int x_in, y_in;       /* frustum clamp inactive? */
    real c00, c01, c11;   /* 2D covariance BEFORE blur */
    real det_orig, det;   /* det before / after blur */
    real conic[3];
    real comp;
} proj_ctx;

static inline real sym6(const real *S, int i, int j) {
    static const int idx[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    return S[idx[i][j]];
}

/* returns 1 if the Gaussian survives culling up to (and including) det<=0 */
static int project_core(const real *mean, const real *quat, const real *scale, const real *V /*4x4 row-major*/,
                        const real *K /*3x3*/, int W, int H, real eps2d, real near_p, real far_p, proj_ctx *c) {
    const real r00 = V[0], r01 = V[1], r02 = V[2], t0 = V[3];
    const real r10 = V[4], r11 = V[5], r12 = V[6], t1 = V[7];
    const real r20 = V[8], r21 = V[9], r22 = V[10], t2 = V[11];
    c->pc[0] = ((r00 * mean[0] + r01 * mean[1]) + r02 * mean[2]) + t0;
    c->pc[1] = ((r10 * mean[0] + r11 * mean[1]) + r12 * mean[2]) + t1;
    c->pc[2] = ((r20 * mean[0] + r21 * mean[1]) + r22 * mean[2]) + t2;
    if (c->pc[2] < near_p || c->pc[2] > far_p) return 0;

    quat_scale_to_covar(quat, scale, c->S, c->M, c->Rq);

    /* Sc = R S R^T  : first Wm = R S (3x3), then Sc = Wm R^T (upper part) */
    const real Rv[9] = {r00, r01, r02, r10, r11, r12, r20, r21, r22};
    real Wm[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            Wm[i * 3 + j] = (Rv[i * 3 + 0] * sym6(c->S, 0, j) + Rv[i * 3 + 1] * sym6(c->S, 1, j)) +
                            Rv[i * 3 + 2] * sym6(c->S, 2, j);
    int k = 0;
    for (int i = 0; i < 3; ++i)
        for (int j = i; j < 3; ++j)
            c->Sc[k++] = (Wm[i * 3 + 0] * Rv[j * 3 + 0] + Wm[i * 3 + 1] * Rv[j * 3 + 1]) + Wm[i * 3 + 2] * Rv[j * 3 + 2];

    const real fx = K[0], fy = K[4], cx = K[2], cy = K[5];
    const real tanx = RC(0.5) * (real)W / fx, tany = RC(0.5) * (real)H / fy;
    const real lim_xp = ((real)W - cx) / fx + GSXO_FOV_SLACK * tanx;
    const real lim_xn = cx / fx + GSXO_FOV_SLACK * tanx;
    const real lim_yp = ((real)H - cy) / fy + GSXO_FOV_SLACK * tany;
    const real lim_yn = cy / fy + GSXO_FOV_SLACK * tany;
    const real x = c->pc[0], y = c->pc[1], z = c->pc[2];
    const real rz = RC(1.0) / z, rz2 = rz * rz;
    const real xr = x * rz, yr = y * rz;
    c->x_in = (xr <= lim_xp) && (xr >= -lim_xn);
    c->y_in = (yr <= lim_yp) && (yr >= -lim_yn);
    c->tx = z * R_FMIN(lim_xp, R_FMAX(-lim_xn, xr));
    c->ty = z * R_FMIN(lim_yp, R_FMAX(-lim_yn, yr));
    c->rz = rz;
    c->J00 = fx * rz; c->J11 = fy * rz;
    c->J02 = -(fx * c->tx) * rz2; c->J12 = -(fy * c->ty) * rz2;
    /* T = J Sc (2x3) */
    real T0[3], T1[3];
    for (int j = 0; j < 3; ++j) {
        T0[j] = c->J00 * sym6(c->Sc, 0, j) + c->J02 * sym6(c->Sc, 2, j);
        T1[j] = c->J11 * sym6(c->Sc, 1, j) + c->J12 * sym6(c->Sc, 2, j);
    }
    c->c00 = T0[0] * c->J00 + T0[2] * c->J02;
    c->c01 = T0[1] * c->J11 + T0[2] * c->J12;
    c->c11 = T1[1] * c->J11 + T1[2] * c->J12;
    c->det_orig = c->c00 * c->c11 - c->c01 * c->c01;
    const real b00 = c->c00 + eps2d, b11 = c->c11 + eps2d;
    c->det = b00 * b11 - c->c01 * c->c01;
    if (c->det <= RC(0.0)) return 0;
    const real inv_det = RC(1.0) / c->det;
    c->conic[0] = b11 * inv_det; c->conic[1] = -c->c01 * inv_det; c->conic[2] = b00 * inv_det;
    c->comp = R_SQRT(R_FMAX(RC(0.0), c->det_orig / c->det));
    return 1;
}

int gsxo_project_fwd(int64_t N, int64_t C, const real *means, const real *quats, const real *scales,
                     const real *viewmats, const real *Ks, int W, int H, real eps2d, real near_p, real far_p,
                     real radius_clip, int32_t *radii, real *means2d, real *depths, real *conics,
                     real *comps /*nullable*/) {
    for (int64_t c = 0; c < C; ++c) {
        const real *V = viewmats + 16 * c, *K = Ks + 9 * c;
#pragma omp parallel for schedule(static)
        for (int64_t g = 0; g < N; ++g) {
            int64_t idx = c * N + g;
            radii[idx] = 0;
            means2d[2 * idx] = means2d[2 * idx + 1] = RC(0.0);
            depths[idx] = RC(0.0);
            conics[3 * idx] = conics[3 * idx + 1] = conics[3 * idx + 2] = RC(0.0);
            if (comps) comps[idx] = RC(0.0);
            proj_ctx p;
            if (!project_core(means + 3 * g, quats + 4 * g, scales + 3 * g, V, K, W, H, eps2d, near_p, far_p, &p))
                continue;
            const real fx = K[0], fy = K[4], cx = K[2], cy = K[5];
            const real mx = (fx * p.pc[0]) * p.rz + cx, my = (fy * p.pc[1]) * p.rz + cy;
            const real b00 = p.c00 + eps2d, b11 = p.c11 + eps2d;
            const real b = RC(0.5) * (b00 + b11);
            const real v1 = b + R_SQRT(R_FMAX(GSXO_RADIUS_FLOOR, b * b - p.det));
            const real radius = R_CEIL(GSXO_RADIUS_SIGMA * R_SQRT(v1));
            if (radius <= radius_clip) continue;
            if (mx + radius <= RC(0.0) || mx - radius >= (real)W || my + radius <= RC(0.0) || my - radius >= (real)H)
                continue;
            radii[idx] = (int32_t)radius;
            means2d[2 * idx] = mx; means2d[2 * idx + 1] = my;
            depths[idx] = p.pc[2];
            conics[3 * idx] = p.conic[0]; conics[3 * idx + 1] = p.conic[1]; conics[3 * idx + 2] = p.conic[2];
            if (comps) comps[idx] = p.comp;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K2  fully_fused_projection backward (VJP derived from the forward above; */
/*     pinned by fp64 finite differences, tests/test_oracle_grads.py)       */
/* ------------------------------------------------------------------------ */
int gsxo_project_bwd(int64_t N, int64_t C, const real *means, const real *quats, const real *scales,
                     const real *viewmats, const real *Ks, int W, int H, real eps2d, real near_p, real far_p,
                     const int32_t *radii, const real *v_means2d, const real *v_depths, const real *v_conics,
                     const real *v_comps /*nullable*/, real *v_means /*[N,3] +=*/, real *v_quats /*[N,4] +=*/,
                     real *v_scales /*[N,3] +=*/, real *v_viewmats /*[C,4,4] += , nullable*/) {
    for (int64_t c = 0; c < C; ++c) {
        const real *V = viewmats + 16 * c, *K = Ks + 9 * c;
        const real Rv[9] = {V[0], V[1], V[2], V[4], V[5], V[6], V[8], V[9], V[10]};
        real vVc[16] = {0};      /* this camera's view-matrix sums (left to right over g in the serial builds) */
        if (v_viewmats) memcpy(vVc, v_viewmats + 16 * c, sizeof(vVc));
#pragma omp parallel for schedule(static) reduction(+ : vVc[:16])
        for (int64_t g = 0; g < N; ++g) {
            int64_t idx = c * N + g;
            if (radii[idx] <= 0) continue;
            proj_ctx p;
            const real *mean = means + 3 * g, *quat = quats + 4 * g, *scale = scales + 3 * g;
            if (!project_core(mean, quat, scale, V, K, W, H, eps2d, near_p, far_p, &p)) continue;
            const real fx = K[0], fy = K[4];
            /* 1. conic = inverse(blurred cov2d): GX = -Y G Y, G = [[va, vb/2],[vb/2, vc]] */
            const real a = p.conic[0], b = p.conic[1], cc = p.conic[2];
            const real va = v_conics[3 * idx], vb = RC(0.5) * v_conics[3 * idx + 1], vc = v_conics[3 * idx + 2];
            /* P = G Y */
            const real P00 = va * a + vb * b, P01 = va * b + vb * cc;
            const real P10 = vb * a + vc * b, P11 = vb * b + vc * cc;
            real G00 = -(a * P00 + b * P10), G01 = -(a * P01 + b * P11), G11 = -(b * P01 + cc * P11);
            /* 2. compensation = sqrt(max(0, det_orig/det)) */
            if (v_comps && p.comp > RC(0.0)) {
                const real vr = v_comps[idx] * RC(0.5) / p.comp; /* d comp / d ratio */
                const real b00 = p.c00 + eps2d, b11 = p.c11 + eps2d;
                const real inv_d2 = RC(1.0) / (p.det * p.det);
                G00 += vr * (p.c11 * p.det - p.det_orig * b11) * inv_d2;
                G11 += vr * (p.c00 * p.det - p.det_orig * b00) * inv_d2;
                /* single-variable derivative wrt c01, halved for the symmetric matrix form */
                G01 += vr * RC(0.5) * (RC(-2.0) * p.c01 * p.det + p.det_orig * RC(2.0) * p.c01) * inv_d2;
            }
            /* 3. cov2d = J Sc J^T ; v_Sc = J^T G J ; v_J = 2 G J Sc */
            const real Jm[6] = {p.J00, RC(0.0), p.J02, RC(0.0), p.J11, p.J12};
            const real Gm[4] = {G00, G01, G01, G11};
            real GJ[6]; /* 2x3 */
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j) GJ[i * 3 + j] = Gm[i * 2 + 0] * Jm[0 * 3 + j] + Gm[i * 2 + 1] * Jm[1 * 3 + j];
            real vSc[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) vSc[i * 3 + j] = Jm[0 * 3 + i] * GJ[0 * 3 + j] + Jm[1 * 3 + i] * GJ[1 * 3 + j];
            real vJ[6];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 3; ++j)
                    vJ[i * 3 + j] = RC(2.0) * (GJ[i * 3 + 0] * sym6(p.Sc, 0, j) + GJ[i * 3 + 1] * sym6(p.Sc, 1, j) +
                                               GJ[i * 3 + 2] * sym6(p.Sc, 2, j));
            /* mean2d, depth, J -> v_pc */
            const real x = p.pc[0], y = p.pc[1], rz = p.rz, rz2 = rz * rz, rz3 = rz2 * rz;
            const real vmx = v_means2d[2 * idx], vmy = v_means2d[2 * idx + 1];
            real vpc[3];
            vpc[0] = fx * rz * vmx;
            vpc[1] = fy * rz * vmy;
            vpc[2] = -(fx * x * vmx + fy * y * vmy) * rz2;
            if (v_depths) vpc[2] += v_depths[idx];
            const real vJ00 = vJ[0], vJ02 = vJ[2], vJ11 = vJ[4], vJ12 = vJ[5];
            if (p.x_in) vpc[0] += -fx * rz2 * vJ02; else vpc[2] += -fx * rz3 * vJ02 * p.tx;
            if (p.y_in) vpc[1] += -fy * rz2 * vJ12; else vpc[2] += -fy * rz3 * vJ12 * p.ty;
            vpc[2] += -fx * rz2 * vJ00 - fy * rz2 * vJ11 + RC(2.0) * fx * p.tx * rz3 * vJ02 +
                      RC(2.0) * fy * p.ty * rz3 * vJ12;
            /* 5. Sc = R S R^T ; pc = R mu + t */
            real vR[9], vS[9];
            /* A = vSc R (3x3) */
            real A[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    A[i * 3 + j] = vSc[i * 3 + 0] * Rv[0 * 3 + j] + vSc[i * 3 + 1] * Rv[1 * 3 + j] + vSc[i * 3 + 2] * Rv[2 * 3 + j];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    /* vS = R^T vSc R */
                    vS[i * 3 + j] = Rv[0 * 3 + i] * A[0 * 3 + j] + Rv[1 * 3 + i] * A[1 * 3 + j] + Rv[2 * 3 + i] * A[2 * 3 + j];
                    /* vR = 2 vSc R S  (vSc, S symmetric) + vpc mu^T */
                    vR[i * 3 + j] = RC(2.0) * (A[i * 3 + 0] * sym6(p.S, 0, j) + A[i * 3 + 1] * sym6(p.S, 1, j) +
                                               A[i * 3 + 2] * sym6(p.S, 2, j)) +
                                    vpc[i] * mean[j];
                }
            for (int j = 0; j < 3; ++j)
                v_means[3 * g + j] += Rv[0 * 3 + j] * vpc[0] + Rv[1 * 3 + j] * vpc[1] + Rv[2 * 3 + j] * vpc[2];
            if (v_viewmats) {
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) vVc[i * 4 + j] += vR[i * 3 + j];
                    vVc[i * 4 + 3] += vpc[i];
                }
            }
            /* 6. S = M M^T : vM = 2 vS M ; M = Rq diag(s) */
            real vM[9], vRq[9];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    vM[i * 3 + j] = RC(2.0) * (vS[i * 3 + 0] * p.M[0 * 3 + j] + vS[i * 3 + 1] * p.M[1 * 3 + j] +
                                               vS[i * 3 + 2] * p.M[2 * 3 + j]);
            for (int j = 0; j < 3; ++j) {
                v_scales[3 * g + j] += p.Rq[0 * 3 + j] * vM[0 * 3 + j] + p.Rq[1 * 3 + j] * vM[1 * 3 + j] + p.Rq[2 * 3 + j] * vM[2 * 3 + j];
                for (int i = 0; i < 3; ++i) vRq[i * 3 + j] = vM[i * 3 + j] * scale[j];
            }
            /* 7. Rq(qn), qn = q/|q| */
            real Rtmp[9], qn[4], inv;
            quat_to_rotmat(quat, Rtmp, qn, &inv);
            const real w = qn[0], qx = qn[1], qy = qn[2], qz = qn[3];
            real vq[4];
            vq[0] = RC(2.0) * (qx * (vRq[7] - vRq[5]) + qy * (vRq[2] - vRq[6]) + qz * (vRq[3] - vRq[1]));
            vq[1] = RC(2.0) * (RC(-2.0) * qx * (vRq[4] + vRq[8]) + qy * (vRq[1] + vRq[3]) + qz * (vRq[2] + vRq[6]) + w * (vRq[7] - vRq[5]));
            vq[2] = RC(2.0) * (qx * (vRq[1] + vRq[3]) - RC(2.0) * qy * (vRq[0] + vRq[8]) + qz * (vRq[5] + vRq[7]) + w * (vRq[2] - vRq[6]));
            vq[3] = RC(2.0) * (qx * (vRq[2] + vRq[6]) + qy * (vRq[5] + vRq[7]) - RC(2.0) * qz * (vRq[0] + vRq[4]) + w * (vRq[3] - vRq[1]));
            const real dotp = vq[0] * qn[0] + vq[1] * qn[1] + vq[2] * qn[2] + vq[3] * qn[3];
            for (int k = 0; k < 4; ++k) v_quats[4 * g + k] += (vq[k] - dotp * qn[k]) * inv;
        }
        if (v_viewmats) memcpy(v_viewmats + 16 * c, vVc, sizeof(vVc));
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K3-K7  tile intersection (gslam/rasterization.py:259-274).  */
/* Integer-exact contract.  Depth keys are always IEEE float32 bit patterns. */
/* ------------------------------------------------------------------------ */
static inline uint32_t sat_u32(float f) { /* float -> uint32, negatives saturate to 0 */
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)f;
}
static inline uint32_t umin32(uint32_t a, uint32_t b) { return a < b ? a : b; }

static void tile_rect(float mx, float my, int32_t radius, int tile_size, int tile_w, int tile_h, uint32_t *x0,
                      uint32_t *y0, uint32_t *x1, uint32_t *y1) {
    const float ts = (float)tile_size;
    const float tr = (float)radius / ts, tx = mx / ts, ty = my / ts;
    *x0 = umin32(sat_u32(floorf(tx - tr)), (uint32_t)tile_w);
    *y0 = umin32(sat_u32(floorf(ty - tr)), (uint32_t)tile_h);
    *x1 = umin32(sat_u32(ceilf(tx + tr)), (uint32_t)tile_w);
    *y1 = umin32(sat_u32(ceilf(ty + tr)), (uint32_t)tile_h);
}

static int bit_length_u32(uint32_t v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

int gsxo_isect_count(int64_t C, int64_t N, const float *means2d, const int32_t *radii, int tile_size, int tile_w,
                     int tile_h, int32_t *tiles_per_gauss, int64_t *cum_tiles /*[C*N] inclusive*/) {
    int64_t run = 0;
    for (int64_t idx = 0; idx < C * N; ++idx) {
        int32_t n = 0;
        if (radii[idx] > 0) {
            uint32_t x0, y0, x1, y1;
            tile_rect(means2d[2 * idx], means2d[2 * idx + 1], radii[idx], tile_size, tile_w, tile_h, &x0, &y0, &x1, &y1);
            n = (int32_t)((y1 - y0) * (x1 - x0));
        }
        tiles_per_gauss[idx] = n;
        run += n;
        cum_tiles[idx] = run;
    }
    return 0;
}

typedef struct { int64_t key; int32_t val; int64_t seq; } kv_t;
static int kv_cmp(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq ? 1 : 0); /* stable */
}

/* emit + stable sort on the low (32 + tile_n_bits + cam_n_bits) bits (all other bits are zero by construction) */
int gsxo_isect_emit_sort(int64_t C, int64_t N, const float *means2d, const int32_t *radii, const float *depths,
                         const int64_t *cum_tiles, int tile_size, int tile_w, int tile_h, int do_sort,
                         int64_t *isect_ids /*[M]*/, int32_t *flatten_ids /*[M]*/) {
    const int64_t M = (C * N > 0) ? cum_tiles[C * N - 1] : 0;
    if (M == 0) return 0;
    const int tile_n_bits = bit_length_u32((uint32_t)(tile_w * tile_h));
    kv_t *kv = (kv_t *)malloc(sizeof(kv_t) * (size_t)M);
    if (!kv) return -1;
    for (int64_t idx = 0; idx < C * N; ++idx) {
        if (radii[idx] <= 0) continue;
        uint32_t x0, y0, x1, y1;
        tile_rect(means2d[2 * idx], means2d[2 * idx + 1], radii[idx], tile_size, tile_w, tile_h, &x0, &y0, &x1, &y1);
        const int64_t cid = idx / N;
        const int64_t cam_part = cid << (32 + tile_n_bits);
        uint32_t dbits;
        memcpy(&dbits, depths + idx, 4);
        int64_t k = (idx == 0) ? 0 : cum_tiles[idx - 1];
        for (uint32_t i = y0; i < y1; ++i)
            for (uint32_t j = x0; j < x1; ++j) {
                const int64_t tile_id = (int64_t)i * tile_w + j;
                kv[k].key = cam_part | (tile_id << 32) | (int64_t)dbits;
                kv[k].val = (int32_t)idx;
                kv[k].seq = k;
                ++k;
            }
    }
#ifdef _OPENMP
    if (do_sort) {
        /* same total order (key, emission sequence): a stable counting partition by (camera, tile) = key >> 32, then the
           segments sorted independently on all cores */
        const int64_t n_seg = (C << tile_n_bits);
        int64_t *seg = (int64_t *)calloc((size_t)n_seg + 1, sizeof(int64_t));
        kv_t *kv2 = (kv_t *)malloc(sizeof(kv_t) * (size_t)M);
        if (!seg || !kv2) { free(seg); free(kv2); free(kv); return -1; }
        for (int64_t k = 0; k < M; ++k) seg[(kv[k].key >> 32) + 1] += 1;
        for (int64_t t = 0; t < n_seg; ++t) seg[t + 1] += seg[t];
        int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_seg);
        memcpy(cur, seg, sizeof(int64_t) * (size_t)n_seg);
        for (int64_t k = 0; k < M; ++k) kv2[cur[kv[k].key >> 32]++] = kv[k];
#pragma omp parallel for schedule(dynamic, 16)
        for (int64_t t = 0; t < n_seg; ++t)
            if (seg[t + 1] - seg[t] > 1) qsort(kv2 + seg[t], (size_t)(seg[t + 1] - seg[t]), sizeof(kv_t), kv_cmp);
        free(cur); free(seg); free(kv);
        kv = kv2;
    }
#else
    if (do_sort) qsort(kv, (size_t)M, sizeof(kv_t), kv_cmp);
#endif
    for (int64_t k = 0; k < M; ++k) { isect_ids[k] = kv[k].key; flatten_ids[k] = kv[k].val; }
    free(kv);
    return 0;
}

int gsxo_isect_offset_encode(int64_t M, const int64_t *isect_ids, int64_t C, int tile_w, int tile_h,
                             int32_t *offsets /*[C,tile_h,tile_w]*/) {
    const int64_t n_tiles = (int64_t)tile_w * tile_h;
    const int tile_n_bits = bit_length_u32((uint32_t)n_tiles);
    const int64_t T = C * n_tiles;
    if (M == 0) { for (int64_t t = 0; t < T; ++t) offsets[t] = 0; return 0; }
    /* offsets[t] = first k whose (cam,tile) >= t ; tiles after the last non-empty one get M */
    int64_t k = 0;
    for (int64_t t = 0; t < T; ++t) {
        const int64_t cid = t / n_tiles, tid = t % n_tiles;
        const int64_t want = (cid << tile_n_bits) | tid;
        while (k < M && (isect_ids[k] >> 32) < want) ++k;
        offsets[t] = (int32_t)k;
    }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K8  rasterize_to_pixels forward (fork: + n_touched)                       */
/*     gslam/rasterization.py:325-339 ; SURVEY §9.3                          */
/* ------------------------------------------------------------------------ */
int gsxo_raster_fwd(int64_t C, int64_t N, int CH, const real *means2d, const real *conics, const real *colors,
                    const real *opacities, const real *backgrounds /*[C,CH] nullable*/, int W, int H, int tile_size,
                    int tile_w, int tile_h, const int32_t *offsets, const int32_t *flatten_ids, int64_t M,
                    real vis_min_T, real *render /*[C,H,W,CH]*/, real *alphas /*[C,H,W]*/,
                    int32_t *last_ids /*[C,H,W]*/, int32_t *n_touched /*[C*N] zeroed here*/) {
    if (CH < 1 || CH > 64) return -2;
    for (int64_t i = 0; i < C * N; ++i) n_touched[i] = 0;
    const int64_t n_tiles = (int64_t)tile_w * tile_h;
    for (int64_t c = 0; c < C; ++c)
#pragma omp parallel for schedule(dynamic, 4)
        for (int py = 0; py < H; ++py) {
            real pix[64];                                           /* CH <= 64 (checked by the caller) */
            for (int px = 0; px < W; ++px) {
                const int64_t tile = c * n_tiles + (int64_t)(py / tile_size) * tile_w + (px / tile_size);
                const int64_t start = offsets[tile];
                const int64_t end = (tile + 1 < C * n_tiles) ? offsets[tile + 1] : M;
                const real fx = (real)px + RC(0.5), fy = (real)py + RC(0.5);
                real T = RC(1.0);
                int32_t last = -1;
                for (int k = 0; k < CH; ++k) pix[k] = RC(0.0);
                for (int64_t e = start; e < end; ++e) {
                    const int32_t g = flatten_ids[e];
                    const real dx = means2d[2 * g] - fx, dy = means2d[2 * g + 1] - fy;
                    const real a = conics[3 * g], b = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const real sigma = RC(0.5) * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    const real alpha = R_FMIN(GSXO_ALPHA_MAX, opacities[g] * R_EXP(-sigma));
                    if (sigma < RC(0.0) || alpha < GSXO_ALPHA_MIN) continue;
                    const real nT = T * (RC(1.0) - alpha);
                    if (nT <= GSXO_T_MIN) break;
                    const real vis = alpha * T;
                    for (int k = 0; k < CH; ++k) pix[k] += colors[(int64_t)g * CH + k] * vis;
                    if (nT > vis_min_T) {
#pragma omp atomic
                        n_touched[g] += 1;
                    }
                    last = (int32_t)e;
                    T = nT;
                }
                const int64_t p = (c * H + py) * W + px;
                for (int k = 0; k < CH; ++k)
                    render[p * CH + k] = pix[k] + (backgrounds ? T * backgrounds[c * CH + k] : RC(0.0));
                alphas[p] = RC(1.0) - T;
                last_ids[p] = last;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K9  rasterize_to_pixels backward                             */
/* ------------------------------------------------------------------------ */
int gsxo_raster_bwd(int64_t C, int64_t N, int CH, const real *means2d, const real *conics, const real *colors,
                    const real *opacities, const real *backgrounds, int W, int H, int tile_size, int tile_w,
                    int tile_h, const int32_t *offsets, const int32_t *flatten_ids, int64_t M, const real *alphas,
                    const int32_t *last_ids, const real *v_render /*[C,H,W,CH]*/, const real *v_alphas /*[C,H,W]*/,
                    real *v_means2d /*+=*/, real *v_conics /*+=*/, real *v_colors /*+=*/, real *v_opacities /*+=*/,
                    real *v_means2d_abs /*nullable, +=*/) {
    (void)N; (void)M;
    if (CH < 1 || CH > 64) return -2;
    const int64_t n_tiles = (int64_t)tile_w * tile_h;
    for (int64_t c = 0; c < C; ++c)
#pragma omp parallel for schedule(dynamic, 4)
        for (int py = 0; py < H; ++py) {
            real buf[64];                                           /* CH <= 64 (checked by the caller) */
            for (int px = 0; px < W; ++px) {
                const int64_t p = (c * H + py) * W + px;
                const int32_t last = last_ids[p];
                if (last < 0) continue;
                const int64_t tile = c * n_tiles + (int64_t)(py / tile_size) * tile_w + (px / tile_size);
                const int64_t start = offsets[tile];
                const real fx = (real)px + RC(0.5), fy = (real)py + RC(0.5);
                const real T_final = RC(1.0) - alphas[p];
                real T = T_final;
                const real *vo = v_render + p * CH;
                const real va_out = v_alphas[p];
                real bg_dot = RC(0.0);
                if (backgrounds) for (int k = 0; k < CH; ++k) bg_dot += backgrounds[c * CH + k] * vo[k];
                for (int k = 0; k < CH; ++k) buf[k] = RC(0.0);
                for (int64_t e = last; e >= start; --e) {
                    const int32_t g = flatten_ids[e];
                    const real dx = means2d[2 * g] - fx, dy = means2d[2 * g + 1] - fy;
                    const real a = conics[3 * g], b = conics[3 * g + 1], cc = conics[3 * g + 2];
                    const real sigma = RC(0.5) * (a * dx * dx + cc * dy * dy) + b * dx * dy;
                    const real vis = R_EXP(-sigma);
                    const real alpha = R_FMIN(GSXO_ALPHA_MAX, opacities[g] * vis);
                    if (sigma < RC(0.0) || alpha < GSXO_ALPHA_MIN) continue;
                    const real ra = RC(1.0) / (RC(1.0) - alpha);
                    T *= ra;
                    const real fac = alpha * T;
                    real v_alpha = RC(0.0);
                    for (int k = 0; k < CH; ++k) {
                        const real ck = colors[(int64_t)g * CH + k];
#pragma omp atomic
                        v_colors[(int64_t)g * CH + k] += fac * vo[k];
                        v_alpha += (ck * T - buf[k] * ra) * vo[k];
                        buf[k] += ck * fac;
                    }
                    v_alpha += T_final * ra * va_out;
                    v_alpha += -T_final * ra * bg_dot;
                    if (opacities[g] * vis <= GSXO_ALPHA_MAX) {
                        const real v_sigma = -opacities[g] * vis * v_alpha;
#pragma omp atomic
                        v_conics[3 * g] += RC(0.5) * v_sigma * dx * dx;
#pragma omp atomic
                        v_conics[3 * g + 1] += v_sigma * dx * dy;
#pragma omp atomic
                        v_conics[3 * g + 2] += RC(0.5) * v_sigma * dy * dy;
                        const real gx = v_sigma * (a * dx + b * dy), gy = v_sigma * (b * dx + cc * dy);
#pragma omp atomic
                        v_means2d[2 * g] += gx;
#pragma omp atomic
                        v_means2d[2 * g + 1] += gy;
                        if (v_means2d_abs) {
#pragma omp atomic
                            v_means2d_abs[2 * g] += R_FABS(gx);
#pragma omp atomic
                            v_means2d_abs[2 * g + 1] += R_FABS(gy);
                        }
#pragma omp atomic
                        v_opacities[g] += vis * v_alpha;
                    }
                }
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K13  spherical harmonics ; coeffs [N,Kc,3], dirs [C,N,3]    */
/* ------------------------------------------------------------------------ */
static const double SH_C0 = 0.2820947917738781;
static const double SH_C1 = 0.48860251190292;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792,
                                0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
                                -0.4570457994644658, 1.445305721320277, -0.5900435899266435};

static void sh_basis(int deg, real x, real y, real z, real *B /*16*/) {
    B[0] = RC(SH_C0);
    if (deg < 1) return;
    B[1] = -RC(SH_C1) * y; B[2] = RC(SH_C1) * z; B[3] = -RC(SH_C1) * x;
    if (deg < 2) return;
    const real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    B[4] = RC(SH_C2[0]) * xy; B[5] = RC(SH_C2[1]) * yz; B[6] = RC(SH_C2[2]) * (RC(2.0) * zz - xx - yy);
    B[7] = RC(SH_C2[3]) * xz; B[8] = RC(SH_C2[4]) * (xx - yy);
    if (deg < 3) return;
    B[9] = RC(SH_C3[0]) * y * (RC(3.0) * xx - yy);
    B[10] = RC(SH_C3[1]) * xy * z;
    B[11] = RC(SH_C3[2]) * y * (RC(4.0) * zz - xx - yy);
    B[12] = RC(SH_C3[3]) * z * (RC(2.0) * zz - RC(3.0) * xx - RC(3.0) * yy);
    B[13] = RC(SH_C3[4]) * x * (RC(4.0) * zz - xx - yy);
    B[14] = RC(SH_C3[5]) * z * (xx - yy);
    B[15] = RC(SH_C3[6]) * x * (xx - RC(3.0) * yy);
}
/* dB/d(x,y,z) for unit-vector arguments treated as free variables */
static void sh_basis_grad(int deg, real x, real y, real z, real *dBx, real *dBy, real *dBz) {
    for (int i = 0; i < 16; ++i) dBx[i] = dBy[i] = dBz[i] = RC(0.0);
    if (deg < 1) return;
    dBy[1] = -RC(SH_C1); dBz[2] = RC(SH_C1); dBx[3] = -RC(SH_C1);
    if (deg < 2) return;
    dBx[4] = RC(SH_C2[0]) * y; dBy[4] = RC(SH_C2[0]) * x;
    dBy[5] = RC(SH_C2[1]) * z; dBz[5] = RC(SH_C2[1]) * y;
    dBx[6] = RC(SH_C2[2]) * RC(-2.0) * x; dBy[6] = RC(SH_C2[2]) * RC(-2.0) * y; dBz[6] = RC(SH_C2[2]) * RC(4.0) * z;
    dBx[7] = RC(SH_C2[3]) * z; dBz[7] = RC(SH_C2[3]) * x;
    dBx[8] = RC(SH_C2[4]) * RC(2.0) * x; dBy[8] = RC(SH_C2[4]) * RC(-2.0) * y;
    if (deg < 3) return;
    const real xx = x * x, yy = y * y, zz = z * z;
    dBx[9] = RC(SH_C3[0]) * RC(6.0) * x * y; dBy[9] = RC(SH_C3[0]) * (RC(3.0) * xx - RC(3.0) * yy);
    dBx[10] = RC(SH_C3[1]) * y * z; dBy[10] = RC(SH_C3[1]) * x * z; dBz[10] = RC(SH_C3[1]) * x * y;
    dBx[11] = RC(SH_C3[2]) * RC(-2.0) * x * y; dBy[11] = RC(SH_C3[2]) * (RC(4.0) * zz - xx - RC(3.0) * yy);
    dBz[11] = RC(SH_C3[2]) * RC(8.0) * y * z;
    dBx[12] = RC(SH_C3[3]) * RC(-6.0) * x * z; dBy[12] = RC(SH_C3[3]) * RC(-6.0) * y * z;
    dBz[12] = RC(SH_C3[3]) * (RC(6.0) * zz - RC(3.0) * xx - RC(3.0) * yy);
    dBx[13] = RC(SH_C3[4]) * (RC(4.0) * zz - RC(3.0) * xx - yy); dBy[13] = RC(SH_C3[4]) * RC(-2.0) * x * y;
    dBz[13] = RC(SH_C3[4]) * RC(8.0) * x * z;
    dBx[14] = RC(SH_C3[5]) * RC(2.0) * x * z; dBy[14] = RC(SH_C3[5]) * RC(-2.0) * y * z; dBz[14] = RC(SH_C3[5]) * (xx - yy);
    dBx[15] = RC(SH_C3[6]) * (RC(3.0) * xx - RC(3.0) * yy); dBy[15] = RC(SH_C3[6]) * RC(-6.0) * x * y;
}

/* colors[c,g,:] = max(0, sum_k B_k(dir) coeffs[g,k,:] + 0.5) where radii>0 (else 0) */
int gsxo_sh_fwd(int64_t C, int64_t N, int Kc, int deg, const real *dirs /*[C,N,3]*/, const real *coeffs /*[N,Kc,3]*/,
                const int32_t *radii /*nullable*/, real *colors /*[C,N,3]*/) {
    const int nb = (deg + 1) * (deg + 1);
    if (nb > Kc) return -2;
    for (int64_t c = 0; c < C; ++c)
        for (int64_t g = 0; g < N; ++g) {
            const int64_t idx = c * N + g;
            real *o = colors + 3 * idx;
            o[0] = o[1] = o[2] = RC(0.0);
            if (radii && radii[idx] <= 0) continue;
            real x = dirs[3 * idx], y = dirs[3 * idx + 1], z = dirs[3 * idx + 2];
            const real n = R_SQRT(x * x + y * y + z * z);
            const real inv = n > RC(0.0) ? RC(1.0) / n : RC(0.0);
            x *= inv; y *= inv; z *= inv;
            real B[16];
            sh_basis(deg, x, y, z, B);
            for (int ch = 0; ch < 3; ++ch) {
                real acc = RC(0.0);
                for (int k = 0; k < nb; ++k) acc += B[k] * coeffs[((int64_t)g * Kc + k) * 3 + ch];
                o[ch] = R_FMAX(RC(0.0), acc + RC(0.5));
            }
        }
    return 0;
}

int gsxo_sh_bwd(int64_t C, int64_t N, int Kc, int deg, const real *dirs, const real *coeffs, const int32_t *radii,
                const real *v_colors /*[C,N,3]*/, real *v_coeffs /*[N,Kc,3] +=*/, real *v_dirs /*[C,N,3] = , nullable*/) {
    const int nb = (deg + 1) * (deg + 1);
    if (nb > Kc) return -2;
    for (int64_t c = 0; c < C; ++c)
        for (int64_t g = 0; g < N; ++g) {
            const int64_t idx = c * N + g;
            if (v_dirs) v_dirs[3 * idx] = v_dirs[3 * idx + 1] = v_dirs[3 * idx + 2] = RC(0.0);
            if (radii && radii[idx] <= 0) continue;
            const real dx = dirs[3 * idx], dy = dirs[3 * idx + 1], dz = dirs[3 * idx + 2];
            const real n = R_SQRT(dx * dx + dy * dy + dz * dz);
            const real inv = n > RC(0.0) ? RC(1.0) / n : RC(0.0);
            const real x = dx * inv, y = dy * inv, z = dz * inv;
            real B[16], dBx[16], dBy[16], dBz[16];
            sh_basis(deg, x, y, z, B);
            sh_basis_grad(deg, x, y, z, dBx, dBy, dBz);
            real vx = RC(0.0), vy = RC(0.0), vz = RC(0.0);
            for (int ch = 0; ch < 3; ++ch) {
                real acc = RC(0.0);
                for (int k = 0; k < nb; ++k) acc += B[k] * coeffs[((int64_t)g * Kc + k) * 3 + ch];
                if (!(acc + RC(0.5) > RC(0.0))) continue; /* clamp_min(0) gate */
                const real vcol = v_colors[3 * idx + ch];
                for (int k = 0; k < nb; ++k) {
                    const real co = coeffs[((int64_t)g * Kc + k) * 3 + ch];
                    v_coeffs[((int64_t)g * Kc + k) * 3 + ch] += B[k] * vcol;
                    vx += dBx[k] * co * vcol; vy += dBy[k] * co * vcol; vz += dBz[k] * co * vcol;
                }
            }
            if (v_dirs) {
                /* through the normalisation d = dir/|dir| */
                const real dotp = vx * x + vy * y + vz * z;
                v_dirs[3 * idx] = (vx - dotp * x) * inv;
                v_dirs[3 * idx + 1] = (vy - dotp * y) * inv;
                v_dirs[3 * idx + 2] = (vz - dotp * z) * inv;
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* K11/K12  fused SSIM  (gslam/backend.py:303-307 )             */
/* img layout [B,CH,H,W] contiguous.  Returns the map (same shape) and the   */
/* three partial-derivative maps used by the backward.                       */
/* ------------------------------------------------------------------------ */
static void ssim_window(real *w /*11*/) {
    double s = 0.0, g[11];
    for (int i = 0; i < 11; ++i) { double d = (double)(i - 5); g[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); s += g[i]; }
    for (int i = 0; i < 11; ++i) w[i] = (real)(g[i] / s);
}

static void conv11_same(const real *src, real *dst, real *tmp, int H, int W, const real *w) {
    /* zero-padded separable 11x11: horizontal then vertical */
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            real acc = RC(0.0);
            for (int k = 0; k < 11; ++k) { int xx = x + k - 5; if (xx >= 0 && xx < W) acc += w[k] * src[(int64_t)y * W + xx]; }
            tmp[(int64_t)y * W + x] = acc;
        }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            real acc = RC(0.0);
            for (int k = 0; k < 11; ++k) { int yy = y + k - 5; if (yy >= 0 && yy < H) acc += w[k] * tmp[(int64_t)yy * W + x]; }
            dst[(int64_t)y * W + x] = acc;
        }
}

int gsxo_ssim_fwd(int64_t B, int CH, int H, int W, const real *img1, const real *img2, real *ssim_map,
                  real *dm_dmu1 /*nullable*/, real *dm_dsigma1_sq, real *dm_dsigma12) {
    real w[11];
    ssim_window(w);
    const int64_t HW = (int64_t)H * W;
    real *t = (real *)malloc(sizeof(real) * (size_t)HW * 7);
    real *tmp = t, *mu1 = t + HW, *mu2 = t + 2 * HW, *e11 = t + 3 * HW, *e22 = t + 4 * HW, *e12 = t + 5 * HW, *prod = t + 6 * HW;
    for (int64_t pl = 0; pl < B * CH; ++pl) {
        const real *x = img1 + pl * HW, *y = img2 + pl * HW;
        conv11_same(x, mu1, tmp, H, W, w);
        conv11_same(y, mu2, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) prod[i] = x[i] * x[i];
        conv11_same(prod, e11, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) prod[i] = y[i] * y[i];
        conv11_same(prod, e22, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) prod[i] = x[i] * y[i];
        conv11_same(prod, e12, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) {
            const real m1 = mu1[i], m2 = mu2[i];
            const real s1 = e11[i] - m1 * m1, s2 = e22[i] - m2 * m2, s12 = e12[i] - m1 * m2;
            const real A = RC(2.0) * m1 * m2 + GSXO_SSIM_C1, Bq = RC(2.0) * s12 + GSXO_SSIM_C2;
            const real Cq = m1 * m1 + m2 * m2 + GSXO_SSIM_C1, D = s1 + s2 + GSXO_SSIM_C2;
            ssim_map[pl * HW + i] = (A * Bq) / (Cq * D);
            if (dm_dmu1) {
                dm_dmu1[pl * HW + i] = (m2 * RC(2.0) * Bq) / (Cq * D) - (m2 * RC(2.0) * A) / (Cq * D) -
                                       (m1 * RC(2.0) * A * Bq) / (Cq * Cq * D) + (m1 * RC(2.0) * A * Bq) / (Cq * D * D);
                dm_dsigma1_sq[pl * HW + i] = (-A * Bq) / (Cq * D * D);
                dm_dsigma12[pl * HW + i] = (RC(2.0) * A) / (Cq * D);
            }
        }
    }
    free(t);
    return 0;
}

/* dL/dimg1 given dL/dmap (already zero outside the 'valid' crop if padding=='valid') */
int gsxo_ssim_bwd(int64_t B, int CH, int H, int W, const real *img1, const real *img2, const real *dL_dmap,
                  const real *dm_dmu1, const real *dm_dsigma1_sq, const real *dm_dsigma12, real *dL_dimg1) {
    real w[11];
    ssim_window(w);
    const int64_t HW = (int64_t)H * W;
    real *t = (real *)malloc(sizeof(real) * (size_t)HW * 5);
    real *tmp = t, *src = t + HW, *c1 = t + 2 * HW, *c2 = t + 3 * HW, *c3 = t + 4 * HW;
    for (int64_t pl = 0; pl < B * CH; ++pl) {
        const real *x = img1 + pl * HW, *y = img2 + pl * HW;
        for (int64_t i = 0; i < HW; ++i) src[i] = dL_dmap[pl * HW + i] * dm_dmu1[pl * HW + i];
        conv11_same(src, c1, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) src[i] = dL_dmap[pl * HW + i] * dm_dsigma1_sq[pl * HW + i];
        conv11_same(src, c2, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) src[i] = dL_dmap[pl * HW + i] * dm_dsigma12[pl * HW + i];
        conv11_same(src, c3, tmp, H, W, w);
        for (int64_t i = 0; i < HW; ++i) dL_dimg1[pl * HW + i] = c1[i] + RC(2.0) * x[i] * c2[i] + y[i] * c3[i];
    }
    free(t);
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Warp (gslam/warp.py:7-82).  T = f1_pose @ inv(f2_pose) is formed by the   */
/* caller (4x4 row-major).  c1 [H,W,3], d1 [H,W].                            */
/* out: result [H,W,3], nwarps [H,W,2] (the [1,H,W,2] grid), mask [H,W] u8   */
/* ------------------------------------------------------------------------ */
static void warp_point(const real *T, const real *K, const real *Kinv, int W, int H, int u, int v, real d,
                       real *Xn /*3*/, real *p /*3*/, real *nw /*2*/) {
    /* unprojected = Kinv @ [u,v,1] (warp.py:14-30) */
    const real fu = (real)u, fv = (real)v;
    real X[3];
    for (int i = 0; i < 3; ++i) {
        const real un = Kinv[i * 3 + 0] * fu + Kinv[i * 3 + 1] * fv + Kinv[i * 3 + 2];
        X[i] = d * un + RC(1e-10); /* warp.py:46-47 */
    }
    for (int i = 0; i < 3; ++i) Xn[i] = (T[i * 4 + 0] * X[0] + T[i * 4 + 1] * X[1] + T[i * 4 + 2] * X[2]) + T[i * 4 + 3];
    for (int i = 0; i < 3; ++i) p[i] = Xn[0] * K[i * 3 + 0] + Xn[1] * K[i * 3 + 1] + Xn[2] * K[i * 3 + 2]; /* Xn @ K^T */
    nw[0] = (p[0] / p[2]) * (RC(2.0) / (real)W) - RC(1.0);
    nw[1] = (p[1] / p[2]) * (RC(2.0) / (real)H) - RC(1.0);
}

int gsxo_warp_fwd(int H, int W, const real *T, const real *K, const real *Kinv, const real *c1, const real *d1,
                  real *result, real *nwarps, uint8_t *mask) {
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            real Xn[3], p[3], nw[2];
            warp_point(T, K, Kinv, W, H, u, v, d1[(int64_t)v * W + u], Xn, p, nw);
            const int64_t o = (int64_t)v * W + u;
            nwarps[2 * o] = nw[0]; nwarps[2 * o + 1] = nw[1];
            mask[o] = (nw[0] < RC(1.0)) && (nw[1] < RC(1.0)) && (nw[0] > RC(-1.0)) && (nw[1] > RC(-1.0));
            /* grid_sample bilinear, zeros padding, align_corners=False */
            const real ix = ((nw[0] + RC(1.0)) * (real)W - RC(1.0)) * RC(0.5);
            const real iy = ((nw[1] + RC(1.0)) * (real)H - RC(1.0)) * RC(0.5);
            const real x0f = R_FLOOR(ix), y0f = R_FLOOR(iy);
            const real wx1 = ix - x0f, wy1 = iy - y0f, wx0 = RC(1.0) - wx1, wy0 = RC(1.0) - wy1;
            real acc[3] = {RC(0.0), RC(0.0), RC(0.0)};
            if (x0f > RC(-2.0) && x0f < (real)(W + 1) && y0f > RC(-2.0) && y0f < (real)(H + 1)) {
                const int x0 = (int)x0f, y0 = (int)y0f;
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx) {
                        const int xx = x0 + dx, yy = y0 + dy;
                        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                        const real wgt = (dx ? wx1 : wx0) * (dy ? wy1 : wy0);
                        for (int k = 0; k < 3; ++k) acc[k] += wgt * c1[((int64_t)yy * W + xx) * 3 + k];
                    }
            }
            for (int k = 0; k < 3; ++k) result[3 * o + k] = acc[k];
        }
    return 0;
}

/* VJP wrt T (rows 0..2 of the 4x4; row 3 gets 0), given v_result [H,W,3] and v_nwarps [H,W,2] (nullable) */
int gsxo_warp_bwd(int H, int W, const real *T, const real *K, const real *Kinv, const real *c1, const real *d1,
                  const real *v_result, const real *v_nwarps, real *v_T /*16, = */) {
    for (int i = 0; i < 16; ++i) v_T[i] = RC(0.0);
    for (int v = 0; v < H; ++v)
        for (int u = 0; u < W; ++u) {
            const int64_t o = (int64_t)v * W + u;
            const real d = d1[o];
            const real fu = (real)u, fv = (real)v;
            real X[3];
            for (int i = 0; i < 3; ++i)
                X[i] = d * (Kinv[i * 3 + 0] * fu + Kinv[i * 3 + 1] * fv + Kinv[i * 3 + 2]) + RC(1e-10);
            real Xn[3], p[3], nw[2];
            warp_point(T, K, Kinv, W, H, u, v, d, Xn, p, nw);
            const real ix = ((nw[0] + RC(1.0)) * (real)W - RC(1.0)) * RC(0.5);
            const real iy = ((nw[1] + RC(1.0)) * (real)H - RC(1.0)) * RC(0.5);
            const real x0f = R_FLOOR(ix), y0f = R_FLOOR(iy);
            const real wx1 = ix - x0f, wy1 = iy - y0f, wx0 = RC(1.0) - wx1, wy0 = RC(1.0) - wy1;
            real g_ix = RC(0.0), g_iy = RC(0.0);
            if (x0f > RC(-2.0) && x0f < (real)(W + 1) && y0f > RC(-2.0) && y0f < (real)(H + 1)) {
                const int x0 = (int)x0f, y0 = (int)y0f;
                for (int dy = 0; dy < 2; ++dy)
                    for (int dx = 0; dx < 2; ++dx) {
                        const int xx = x0 + dx, yy = y0 + dy;
                        if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
                        real dotv = RC(0.0);
                        for (int k = 0; k < 3; ++k) dotv += c1[((int64_t)yy * W + xx) * 3 + k] * v_result[3 * o + k];
                        g_ix += (dx ? RC(1.0) : RC(-1.0)) * (dy ? wy1 : wy0) * dotv;
                        g_iy += (dy ? RC(1.0) : RC(-1.0)) * (dx ? wx1 : wx0) * dotv;
                    }
            }
            real g_nw0 = g_ix * RC(0.5) * (real)W, g_nw1 = g_iy * RC(0.5) * (real)H;
            if (v_nwarps) { g_nw0 += v_nwarps[2 * o]; g_nw1 += v_nwarps[2 * o + 1]; }
            /* nw0 = p0/p2 * 2/W - 1 */
            const real s0 = RC(2.0) / (real)W, s1 = RC(2.0) / (real)H;
            real gp[3];
            gp[0] = g_nw0 * s0 / p[2];
            gp[1] = g_nw1 * s1 / p[2];
            gp[2] = -(g_nw0 * s0 * p[0] + g_nw1 * s1 * p[1]) / (p[2] * p[2]);
            /* p = K Xn */
            real gX[3];
            for (int j = 0; j < 3; ++j) gX[j] = gp[0] * K[0 * 3 + j] + gp[1] * K[1 * 3 + j] + gp[2] * K[2 * 3 + j];
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) v_T[i * 4 + j] += gX[i] * X[j];
                v_T[i * 4 + 3] += gX[i];
            }
        }
    return 0;
}

/* ------------------------------------------------------------------------ */
/* Adam step with torch.optim.Adam(fused=True) semantics (defaults):         */
/* gslam/backend.py:565-602                                                  */
/* ------------------------------------------------------------------------ */
int gsxo_adam(int64_t n, real *p, const real *g, real *m, real *v, real lr, real beta1, real beta2, real eps,
              int64_t step /*1-based*/) {
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    const real step_size = (real)((double)lr / bc1);
    const real bc2_sqrt = (real)sqrt(bc2);
    for (int64_t i = 0; i < n; ++i) {
        m[i] = m[i] + (g[i] - m[i]) * (RC(1.0) - beta1); /* lerp form used by torch */
        v[i] = beta2 * v[i] + (RC(1.0) - beta2) * g[i] * g[i];
        const real denom = R_SQRT(v[i]) / bc2_sqrt + eps;
        p[i] -= (step_size * m[i]) / denom;
    }
    return 0;
}
