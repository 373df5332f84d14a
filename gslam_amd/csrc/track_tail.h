// track_tail.h - the tail of a pose optimiser's closure (C = 1) as a device function over a caller-provided block of LDS: sum of the
// per-workgroup pose-gradient partial rows -> PoseZhou backward -> one step of the optimiser state machine (track_opt.h) on
// (dt, dR, exposure) -> PoseZhou forward of the new parameters into the view matrix the next closure renders with; optionally the
// loss is finished here as well from per-workgroup rows.  Two callers: the one-workgroup launch gsx_track_opt_tail
// (track_opt_impl.inc, THREADS = 1024) and - round 5 - the LAST workgroup to finish of the fused tracking rasteriser launch
// (raster_v4.inc, THREADS = 256), which takes a launch off the closure's chain (gslam/frontend.py:621-658).
// Included after track_opt.h (TO_MAXN / TO_MAXH decide the state's layout) and pose_math.h.
#pragma once

struct ToTailArgs {
    const float *partials;   // [n_blocks][1][12]
    int n_blocks;
    const float *Rt;         // [4,4]
    float *dt, *dR, *exposure;
    const float *v_exposure; // [2]   (ignored when loss_rows is given)
    const float *loss;       //       (ignored when loss_rows is given)
    float *viewmat;          // [4,4] out
    // optional: finish the tracking loss here as well - the per-workgroup rows gsx_map_loss(sums = NULL) left in its
    // workspace ([n_loss_rows][6]: S term, log-beta term, tv, v_a, v_b, spare) -> loss = loss_coef * (col 0 + col 1),
    // exposure gradient = (col 3, col 4)
    const float *loss_rows;
    int n_loss_rows;
    float loss_coef;
};

constexpr int TO_STATE_WORDS = (int)((sizeof(TrackOptState) + 3) / 4);

template <int THREADS>
struct ToTailLds {
    static constexpr int ROWS = THREADS / 12;                // row-strided accumulators x 12 columns
    uint32_t words[(TO_STATE_WORDS + 3) / 4 * 4];
    float acc[ROWS][12];
    float v[16];
    float p[TO_MAXN], g[TO_MAXN];
    float Rt[16], dR[8];
    float loss[THREADS / 64][4];                             // per-wavefront sums of the loss columns 0, 1, 3, 4
};

// COHERENT: the rows were written by OTHER workgroups of the same launch (agent-scope stores): read them with agent-scope loads.
// One workgroup of THREADS threads, all of which must call it.  Fixed summation order, no atomics.
template <int THREADS, bool COHERENT>
__device__ __forceinline__ void to_tail_body(TrackOptState *state, const ToTailArgs &a, ToTailLds<THREADS> &L) {
    constexpr int ROWS = ToTailLds<THREADS>::ROWS;
    auto ld = [](const float *p) -> float {
        if constexpr (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else return *p;
    };
    const int t = (int)threadIdx.x;
    uint32_t *gw = reinterpret_cast<uint32_t *>(state);
    for (int i = t; i < TO_STATE_WORDS; i += THREADS) L.words[i] = gw[i];
    // read by lane 0 after the reductions: fetched now, by others
    if (t >= 64 && t < 80) L.Rt[t - 64] = a.Rt[t - 64];
    if (t >= 80 && t < 86) L.dR[t - 80] = a.dR[t - 80];
    if (t >= 86 && t < 89) L.p[t - 86] = a.dt[t - 86];
    if (t >= 89 && t < 95) L.p[3 + t - 89] = a.dR[t - 89];
    if (t >= 95 && t < 97) L.p[9 + t - 95] = a.exposure[t - 95];
    float c0 = 0.f, c1 = 0.f, c3 = 0.f, c4 = 0.f;
    if (a.loss_rows) {                                       // issued first: independent of the partial rows below
        // (written by an EARLIER launch in both callers - the rasteriser's tiles - so plain loads; eight rows per thread in
        // flight at once: a 256-thread caller walks 1200 rows in one trip instead of five dependent ones)
        for (int i0 = t; i0 < a.n_loss_rows; i0 += 8 * THREADS) {
            float v[8][4];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * THREADS;
                const bool in = i < a.n_loss_rows;
                const float *row = a.loss_rows + (int64_t)(in ? i : 0) * 6;
                v[u][0] = in ? row[0] : 0.f; v[u][1] = in ? row[1] : 0.f; v[u][2] = in ? row[3] : 0.f; v[u][3] = in ? row[4] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { c0 += v[u][0]; c1 += v[u][1]; c3 += v[u][2]; c4 += v[u][3]; }
        }
    }
    {
        const int k = t % 12, r = t / 12;
        if (r < ROWS) {
            // loads in flight per thread: 8 (cached rows of an earlier launch) or 24 (COHERENT: every load goes to memory, ~2 us a
            // round trip - 490 rows over 21 row-strided accumulators are ONE trip of 24 instead of three of 8)
            constexpr int FLY = COHERENT ? 24 : 8;
            float acc[FLY];
#pragma unroll
            for (int u = 0; u < FLY; ++u) acc[u] = 0.f;
            for (int b = r; b < a.n_blocks; b += FLY * ROWS) {           // always FLY loads in flight, the ragged end too
#pragma unroll
                for (int u = 0; u < FLY; ++u) {
                    const int bb = b + u * ROWS;
                    acc[u] += bb < a.n_blocks ? ld(a.partials + (int64_t)bb * 12 + k) : 0.f;
                }
            }
            float tot = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
#pragma unroll
            for (int u = 8; u < FLY; u += 8)
                tot += ((acc[u] + acc[u + 1]) + (acc[u + 2] + acc[u + 3])) + ((acc[u + 4] + acc[u + 5]) + (acc[u + 6] + acc[u + 7]));
            L.acc[r][k] = tot;
        }
    }
    if (a.loss_rows) {
        c0 = gsx_wave_sum(c0); c1 = gsx_wave_sum(c1); c3 = gsx_wave_sum(c3); c4 = gsx_wave_sum(c4);
        if ((t & 63) == 0) {
            float *o = L.loss[t >> 6];
            o[0] = c0; o[1] = c1; o[2] = c3; o[3] = c4;
        }
    }
    __syncthreads();
    if (t < 16) {
        float acc4[4] = {0.f, 0.f, 0.f, 0.f};
        if (t < 12) {
            int rr = 0;
            for (; rr + 3 < ROWS; rr += 4)
#pragma unroll
                for (int u = 0; u < 4; ++u) acc4[u] += L.acc[rr + u][t];
            for (; rr < ROWS; ++rr) acc4[0] += L.acc[rr][t];
        }
        L.v[t] = (acc4[0] + acc4[1]) + (acc4[2] + acc4[3]);   // row 3 of the view-matrix gradient stays zero
    }
    __syncthreads();
    if (t < 64) {                                            // wavefront 0: lane 0 does the pose algebra, all 64 the vectors
        float v_exp[2], loss_v;
        if (a.loss_rows) {
            float t4[4];
            for (int k = 0; k < 4; ++k) {
                float q[4] = {0.f, 0.f, 0.f, 0.f};
                for (int w = 0; w < THREADS / 64; w += 4)
                    for (int u = 0; u < 4; ++u) q[u] += L.loss[w + u][k];
                t4[k] = (q[0] + q[1]) + (q[2] + q[3]);
            }
            loss_v = a.loss_coef * (t4[0] + t4[1]);
            v_exp[0] = t4[2]; v_exp[1] = t4[3];
        } else {
            loss_v = a.loss[0]; v_exp[0] = a.v_exposure[0]; v_exp[1] = a.v_exposure[1];
        }
        if (t == 0) {
            float vdR[6], vdt[3];
            gsx_pose::pose_bwd_one(L.Rt, L.dR, L.v, vdR, vdt);
            for (int i = 0; i < 3; ++i) L.g[i] = vdt[i];
            for (int i = 0; i < 6; ++i) L.g[3 + i] = vdR[i];
            for (int i = 0; i < 2; ++i) L.g[9 + i] = v_exp[i];
        }
        // (LDS traffic of one wavefront is ordered: the other lanes see lane 0's p / g, and lane 0 theirs below)
        to_advance(reinterpret_cast<TrackOptState *>(L.words), L.p, L.g, (double)loss_v);
        if (t == 0) {
            for (int i = 0; i < 3; ++i) a.dt[i] = L.p[i];
            for (int i = 0; i < 6; ++i) a.dR[i] = L.p[3 + i];
            for (int i = 0; i < 2; ++i) a.exposure[i] = L.p[9 + i];
            gsx_pose::pose_fwd_one(L.Rt, L.p + 3, L.p, a.viewmat);
        }
    }
    __syncthreads();
    for (int i = t; i < TO_STATE_WORDS; i += THREADS) gw[i] = L.words[i];
}
