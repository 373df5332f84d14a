/*
 * gsx.h — C ABI of libgsx.so, the MI355X (gfx950) native kernels behind the gslam render / loss / warp hot path.
 *
 * This is the drop-in boundary.  The reference binds its CUDA kernels through two Python
 * extension modules that are NOT in /root/reference (gsplat fork, fused-ssim); the entry points below are what a
 * binding for THIS path needs, one per kernel stage, and each cites the reference call site it replaces.
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no torch / C++ types.
 *  - All data pointers are DEVICE pointers (HBM) unless marked host.  fp32 everywhere, ids int32, isect keys int64.
 *  - The caller owns every buffer, including workspaces (query *_workspace_bytes first).  The library never
 *    allocates or frees device memory and never synchronises the stream, except gsx_read_i64 (explicit read-back).
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous.
 *  - Return value: 0 = ok, negative = GSX_E_*; text via gsx_last_error() (thread-local).  No exceptions cross.
 *  - Re-entrant; no global mutable state; safe from several processes / threads on distinct streams.
 *  - [C,N,...] arrays are row-major; "flatten id" = c*N + g.
 */
#ifndef GSX_H
#define GSX_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSX_OK 0
#define GSX_E_INVALID -1   /* bad argument / shape            */
#define GSX_E_LAUNCH -2    /* hip launch or runtime error      */
#define GSX_E_UNSUPPORTED -3
#define GSX_E_WORKSPACE -4 /* workspace too small              */

/* numeric constants of the external kernels; one place, one line to change */
#define GSX_FOV_SLACK 0.3f
#define GSX_RADIUS_FLOOR 0.01f
#define GSX_RADIUS_SIGMA 3.0f
#define GSX_ALPHA_MAX 0.999f
#define GSX_ALPHA_MIN (1.0f / 255.0f)
#define GSX_T_MIN 1e-4f
#define GSX_SSIM_C1 0.0001f
#define GSX_SSIM_C2 0.0009f
#define GSX_BETA_MIN 0.01f        /* gslam/rasterization.py:149 */
#define GSX_TILE 16               /* gslam/rasterization.py:59 (only 16 is supported, as upstream tests) */
#define GSX_REC_STRIDE_MAX 12     /* splat record: xy(2) conic(3) opac(1) colors(<=5), padded to 8 or 12 floats */

/* projection flags */
#define GSX_PROJ_LOG_SCALES 1     /* `scales` holds log-scales; exp() is fused (gslam/rasterization.py:147) */
#define GSX_PROJ_RENDER_DEPTH 2   /* gslam record: append camera depth channel (rasterization.py:234-240)   */
#define GSX_PROJ_VIEW_PARTIALS 8  /* gsx_project_bwd only: leave the per-workgroup pose-gradient partials
                                     ([gsx_project_bwd_blocks(N)][C][12] floats at the start of the workspace) for a fused
                                     consumer (gsx_track_opt_tail) and skip the finishing launch; v_viewmats is ignored */
#define GSX_PROJ_SKIP_CULLED 16   /* gsx_front_fwd only: rows of culled (camera, Gaussian) pairs get radii = 0 and
                                     tiles_per_gauss = 0 and nothing else (nobody reads them in a pose-only closure) */
#define GSX_PROJ_RESET_V_REC 64   /* gsx_project_bwd / gsx_front_pose_bwd: every gradient-record row is zeroed right after it has
                                     been read, so the buffer is all zeros again when the launch ends and the next forward need
                                     not clear it (a third of what the projection writes); needs the 12-float record rows */
#define GSX_PROJ_COMPACT 32       /* gsx_front_fwd / gsx_front_pose_bwd: rec and v_rec hold one row per visible INSTANCE, indexed
                                     by its slot ((c * R + row) * slots_per_segment + position, gsx_front_layout) instead of one
                                     row per flatten id, and flatten_ids carries slots.  Slots are assigned in flatten-id order,
                                     so (depth, slot) sorts exactly like (depth, flatten id); record s of the front's workspace
                                     holds the flatten id of slot s.  radii / tiles_per_gauss become nullable. */
#define GSX_PROJ_BETAS 4          /* gslam record: append beta=clamp(exp(log_unc),0.01) (rasterization.py:149,249-256) */
#define GSX_PROJ_DEFER_SORT 256   /* gsx_front_fwd only: stop after the placement - every tile's segment of the key buffer holds its
                                   * (depth bits << 32 | id) keys UNSORTED, flatten_ids is not written; the consumer
                                   * (gsx_raster_track_fused_sorting) sorts what it composites.  gsx_front_keys tells where the keys are */
#define GSX_PROJ_CANDIDATES 128   /* gsx_front_fwd / gsx_front_pose_bwd (with GSX_PROJ_COMPACT): the workspace carries the per-frame
                                     candidate set of gsx_front_candidates; closures whose poses stay within its margins project
                                     the candidates' pose-independent records instead of culling all N Gaussians again - same
                                     results bit for bit, and a closure outside the margins takes the full path by itself */
#define GSX_PROJ_ROW_KEYS 1024    /* gsx_front_fwd only, with GSX_PROJ_COMPACT | GSX_PROJ_DEFER_SORT (round 5): the front ENDS with the
                                     projection launch - no count-matrix scan, no placement launch.  Every projection workgroup leaves
                                     the keys of its row of Gaussians in the row's own segment of the workspace, grouped by tile, and
                                     one word per (camera, tile, row): offset inside the segment << 13 | count (gsx_front_rows_layout).
                                     offsets / flatten_ids are not written; M_dev is int64 [64] here: 64 key counters, zeroed by this
                                     launch, whose SUM is the render's M once the consumer's tile workgroups have collected their
                                     keys (gsx_raster_track_fused_rows: tile t reserves its segment on counter t % 64, each counter
                                     over its own 64th of the key buffer); status bit 1 also reports a row that outgrew its segment
                                     (M_cap / rows keys) or a counter that outgrew its 64th. */
#define GSX_PROJ_TILE_EXACT 2048   /* gsx_front_fwd (pose-only closures) and gsx_project_fwd_rects: an instance is listed only in the tiles of its 3-sigma square
                                    * that hold a pixel centre inside the axis-aligned box of the ellipse alpha >= 1/255 (from the conic
                                    * and the opacity, with margins).  The reference lists the whole square (gslam/rasterization.py:259-272)
                                    * and its rasteriser skips the rest pixel by pixel: render, loss and gradients are unchanged, the tile
                                    * lists a quarter shorter; tiles_per_gauss (if asked for) stays the reference's count. */
#define GSX_PROJ_MAP_RECORDS 512   /* with GSX_PROJ_CANDIDATES, to gsx_front_candidates / gsx_front_fwd / gsx_front_pose_bwd: the
                                     candidate area holds the pose-independent record of EVERY Gaussian (slot = Gaussian
                                     index; margins ignored) and a packed cull row (mean, largest scale squared) each.  Valid
                                     for any pose until the map's arrays change.  Closures keep their own conservative cull
                                     (one coalesced 16-byte load per Gaussian) and read one 64-byte record per survivor instead
                                     of gathering 14 floats from five arrays and rebuilding the covariance - same results
                                     bit for bit (the record's covariance is the same float expression) */

int gsx_version(void);
const char *gsx_last_error(void);
/* number of floats per splat record for CH colour channels (8 for CH<=2, 12 for CH<=5); <0 if unsupported */
int gsx_record_stride(int CH);
/* explicit device->host read-back of one int64 (synchronises `stream`) */
int gsx_read_i64(const int64_t *dev_ptr, int64_t *host_out, void *stream);

/* ---- K1: gsplat fully_fused_projection fwd  (gslam/rasterization.py:153-170, :390-407) -------------------------
 * outputs [C,N]: radii i32, means2d [.,2], depths, conics [.,3] (nullable when `rec` is given: columns 2..4 of the record
 * hold the same values), comps (nullable; 'antialiased' compensation).
 * Culled rows are written as zeros.  tiles_per_gauss (nullable) = K3 count for tile 16 (rasterization.py:259-272).
 * Optional fused gslam front-end (rasterization.py:145-149,183-256): when `rec` != NULL a splat record
 * [xy, conic, opacity, colors...] of gsx_record_stride(CH) floats is written per (c,g) with
 *   opacity = sigmoid(logit_opacities[g]); colors = sigmoid(logit_colors[g,0:3]) (+ depth) (+ beta), CH = 3 + flags.
 */
int gsx_project_fwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                    const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                    float far_plane, float radius_clip, int flags, int32_t *radii, float *means2d, float *depths,
                    float *conics, float *comps, int32_t *tiles_per_gauss, int tile_w, int tile_h,
                    const float *logit_opacities, const float *logit_colors, const float *log_uncertainties,
                    float *rec, int32_t *vis_count /*[N] nullable: number of cameras with radii>0 (backend.py:287,357)*/,
                    float *v_rec_clear /*[C,N,12] nullable (needs rec): zero-filled on the way, for gsx_raster_bwd's
                                         accumulation - saves the separate clear in front of every backward */,
                    void *stream);
/* gsx_project_fwd (rec given) that also packs the tile rectangle of every (camera, Gaussian) into rects uint32 [C,N] for
 * gsx_isect_bin_sort_rects: the reference's 3-sigma square, or the tight one under flags | GSX_PROJ_TILE_EXACT;
 * tiles_per_gauss (if asked for) counts the tiles of THAT rectangle.  flags | GSX_PROJ_SKIP_CULLED: a culled row gets radii = 0 and
 * rects = 0 and nothing else (its means2d / depths / record keep whatever they held) */
int gsx_project_fwd_rects(const float *means, const float *quats, const float *scales, const float *viewmats,
                          const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                          float far_plane, float radius_clip, int flags, int32_t *radii, float *means2d, float *depths,
                          float *conics, float *comps, int32_t *tiles_per_gauss, int tile_w, int tile_h,
                          const float *logit_opacities, const float *logit_colors, const float *log_uncertainties,
                          float *rec, int32_t *vis_count, float *v_rec_clear, uint32_t *rects, void *stream);

/* ---- K2: projection bwd (autograd of K1; pose gradient used at gslam/frontend.py:627-646, backend.py:665-670) --
 * v_means2d / v_conics are read with a row stride in floats (2 / 3 when dense, the record stride when they alias
 * a v_rec buffer).  v_depths, v_comps, v_rec nullable.  Outputs are OVERWRITTEN (sum over cameras done inside):
 * v_means [N,3], v_quats [N,4], v_scales [N,3] (w.r.t. log-scales when GSX_PROJ_LOG_SCALES), v_viewmats [C,4,4]
 * (nullable).  With v_rec: also v_logit_opacities [N], v_logit_colors [N,3], v_log_unc [N] (gslam front-end bwd).
 * Pose-only call (tracking against a frozen map): v_means = v_quats = v_scales = NULL (and the v_logit_* / v_log_unc
 * outputs ignored) with v_viewmats != NULL skips the per-Gaussian chain and its stores.
 * workspace: gsx_project_bwd_workspace_bytes(N, C).
 */
int64_t gsx_project_bwd_workspace_bytes(int64_t N, int64_t C);
int64_t gsx_project_bwd_blocks(int64_t N);
int gsx_project_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                    const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                    float far_plane, int flags, const int32_t *radii, const float *v_means2d,
                    int64_t v_means2d_stride, const float *v_depths, const float *v_conics, int64_t v_conics_stride,
                    const float *v_comps, const float *logit_opacities, const float *logit_colors,
                    const float *log_uncertainties, const float *v_rec, float *v_means, float *v_quats,
                    float *v_scales, float *v_viewmats, float *v_logit_opacities, float *v_logit_colors,
                    float *v_log_unc, void *workspace, int64_t workspace_bytes, void *stream);
/* The same over the rows g_begin .. g_end of the map only (g_begin a multiple of 256, g_end a multiple of 256 or N; needs
 * GSX_PROJ_VIEW_PARTIALS and v_viewmats = NULL): launches over ranges that tile [0, N) leave the gradients and the pose partials
 * exactly as ONE gsx_project_bwd does.  New design (the reference is single-GPU): the multi-GPU BA step hands each
 * finished range to the gradient exchange while the next one is still being computed (gslam_amd.dist.StepBucket). */
int gsx_project_bwd_range(const float *means, const float *quats, const float *scales, const float *viewmats,
                          const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                          float far_plane, int flags, const int32_t *radii, const float *v_means2d,
                          int64_t v_means2d_stride, const float *v_depths, const float *v_conics, int64_t v_conics_stride,
                          const float *v_comps, const float *logit_opacities, const float *logit_colors,
                          const float *log_uncertainties, const float *v_rec, float *v_means, float *v_quats,
                          float *v_scales, float *v_viewmats, float *v_logit_opacities, float *v_logit_colors,
                          float *v_log_unc, void *workspace, int64_t workspace_bytes, int64_t g_begin, int64_t g_end,
                          void *stream);

/* ---- K10: gsplat.quat_scale_to_covar_preci (gslam/insertion.py:88-91) ------------------------------------------ */
int gsx_quat_scale_to_covar_preci(const float *quats, const float *scales, int64_t n, float *covars /*[n,3,3]*/,
                                  float *precis /*[n,3,3], nullable*/, void *stream);

/* ---- pack separate [C,N,*] arrays into splat records (front of gsplat rasterize_to_pixels,
 *      gslam/rasterization.py:325-339).  opac_stride_c / color_stride_c = 0 broadcasts an [N]-shaped input over C. */
int gsx_pack_records(const float *means2d, const float *conics, const float *opacities, const float *colors,
                     int64_t N, int64_t C, int CH, float *rec, void *stream);

/* ---- K3..K7: gsplat isect_tiles + isect_offset_encode (gslam/rasterization.py:259-274) ----------------------------
 * Integer-exact contract: key = cam<<(32+tile_n_bits) | tile<<32 | float_bits(depth); stable order.
 */
int gsx_isect_count(const float *means2d, const int32_t *radii, int64_t CN, int tile_w, int tile_h,
                    int32_t *tiles_per_gauss, void *stream);
/* inclusive scan of tiles_per_gauss -> cum_tiles int64 [CN]; M = cum_tiles[CN-1] (own three-launch scan, no library) */
int64_t gsx_scan_workspace_bytes(int64_t CN);
int gsx_isect_scan(const int32_t *tiles_per_gauss, int64_t CN, int64_t *cum_tiles, void *workspace,
                   int64_t workspace_bytes, void *stream);
/* unsorted emission in gsplat's order (isect_tiles(sort=False)): for every visible (camera, Gaussian) in flatten order, its
 * tiles row-major, isect_ids[k] = key, flatten_ids[k] = flatten id, k from cum_tiles.  The sort is gsx_isect_bin_sort. */
int gsx_isect_emit(const float *means2d, const int32_t *radii, const float *depths, const int64_t *cum_tiles, int64_t N,
                   int64_t C, int tile_w, int tile_h, int64_t M, int64_t *isect_ids, int32_t *flatten_ids, void *stream);
int gsx_isect_offset_encode(const int64_t *isect_ids, int64_t M, int64_t C, int tile_w, int tile_h,
                            int32_t *offsets /*[C,tile_h,tile_w]*/, void *stream);

/* ---- K3..K7 v2: tile-binned depth sort (csrc/isect_bin.hip).  Same results as count/scan/emit_sort/offset_encode for
 * sort=1, but sync-free: every size stays on the device.  offsets: int32 [T+1] (offsets[T] = min(M, INT_MAX));
 * M_dev: int64 [1] = true number of intersections; status: int32 [1], bit 0 is OR-ed in when M > M_cap (entries
 * beyond M_cap are dropped - the caller must re-run with a larger capacity); flatten_ids [M_cap]; isect_ids
 * [M_cap] nullable; tile_order [T] nullable: the tiles grouped by descending list length, to be handed to
 * gsx_raster_fwd / gsx_raster_bwd as their workgroup launch order (a scheduling hint: results do not depend on it). */
int64_t gsx_isect_bin_workspace_bytes(int64_t C, int tile_w, int tile_h, int64_t M_cap);
/* the same plus 16 B x C x N for the spatial pre-sort of large maps (C * N >= 2^21): with the smaller workspace the call
 * still works and places the intersections directly (slower beyond ~2M instances) */
int64_t gsx_isect_bin_workspace_bytes_n(int64_t C, int64_t N, int tile_w, int tile_h, int64_t M_cap);
int gsx_isect_bin_sort(const float *means2d, const int32_t *radii, const float *depths, int64_t N, int64_t C,
                       int tile_w, int tile_h, int64_t M_cap, int32_t *offsets, int64_t *M_dev, int32_t *status,
                       int64_t *isect_ids, int32_t *flatten_ids, int32_t *tile_order, void *workspace,
                       int64_t workspace_bytes, void *stream);
/* gsx_isect_bin_sort over rectangles the PROJECTION prepared (round 5): rects = uint32 [C*N] packed by gsx_project_fwd_rects
 * (x0 | x1 << 8 | y0 << 16 | y1 << 24 in tiles, 0 = lists nowhere; tile grids below 256 x 256) - read instead of means2d + radii.
 * With GSX_PROJ_TILE_EXACT on the projection the rectangles are TIGHT: an instance is listed only in the tiles of its 3-sigma square
 * that hold a pixel centre inside the axis-aligned box of its alpha >= 1/255 ellipse.  The reference lists the whole square
 * (gslam/rasterization.py:259-272) and its rasteriser skips the rest pixel by pixel: every output of the rasteriser and of its
 * backward is unchanged, the lists are shorter - for launch plans whose tile lists only the rasteriser reads (offsets /
 * flatten_ids are then NOT those of gsplat.isect_tiles). */
int gsx_isect_bin_sort_rects(const uint32_t *rects, const float *depths, int64_t N, int64_t C, int tile_w, int tile_h,
                             int64_t M_cap, int32_t *offsets, int64_t *M_dev, int32_t *status, int64_t *isect_ids,
                             int32_t *flatten_ids, int32_t *tile_order, void *workspace, int64_t workspace_bytes, void *stream);

/* ---- K1 + K3..K7 fused for the launch plans (csrc/isect_bin.hip, "fused front"): the gslam front-end projection of
 * gsx_project_fwd (flags: GSX_PROJ_LOG_SCALES implied by the caller's data, RENDER_DEPTH, BETAS, SKIP_CULLED; radius_clip 0)
 * and the sync-free tile lists of gsx_isect_bin_sort, in four launches: the projection counts every visible instance into
 * its workgroup's row of the count matrix and leaves a 16-byte instance record, the placement scans the per-tile totals
 * itself.  Same outputs, bit for bit: radii, tiles_per_gauss, rec [C,N,12], vis_count, offsets [T+1], M_dev, status,
 * flatten_ids [M_cap], tile_order (nullable).  means2d / depths / conics are nullable (the rasteriser reads the record).
 * v_rec_clear nullable.  Limits: C <= 255, C * tile_w * tile_h * 4 + 64 bytes of LDS <= 64 KiB. */
int64_t gsx_front_workspace_bytes(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap);
/* out4 = {rows R, slots per (camera, row) segment, byte offset of the 16-byte instance records in the workspace, byte offset
 * of the int32 instance counts [C][R]}; a record = {x0 | x1 << 16, y0 | y1 << 12 | c << 24, depth bits, flatten id} */
int gsx_front_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4);
int gsx_front_fwd(const float *means, const float *quats, const float *scales, const float *viewmats, const float *Ks,
                  int64_t N, int64_t C, int W, int H, float eps2d, float near_plane, float far_plane, int flags,
                  const float *logit_opacities, const float *logit_colors, const float *log_uncertainties,
                  int32_t *radii, float *means2d, float *depths, float *conics, int32_t *tiles_per_gauss, float *rec,
                  float *v_rec_clear, int32_t *vis_count, int64_t M_cap, int32_t *offsets, int64_t *M_dev,
                  int32_t *status, int32_t *flatten_ids, int32_t *tile_order,
                  /* balanced_order nullable.  If given (C * tile_w * tile_h <= 2048): one extra workgroup of the projection
                   * launch does the work of gsx_tile_balance(tile_work, ...) -> balanced_order, hidden behind the projection */
                  const int32_t *tile_work, int32_t *balanced_order, float chunk_cost, float light_rate, int n_cus,
                  void *workspace, int64_t workspace_bytes, void *stream);

/* Per-frame candidate set for the closures of a pose optimiser (tracking: gslam/frontend.py:604-662 evaluates 36 renders per
 * frame whose poses differ by fractions of a pixel; window refinement: gslam/backend.py:447-506).  For every camera's
 * REFERENCE pose (viewmats_ref) the Gaussians that the exact projection could keep for ANY pose within the margins - relative
 * rotation |R R0^T - I|_F <= rot_max, relative translation |t - R R0^T t0| <= trans_max - are compacted in Gaussian order,
 * per projection workgroup, together with what the projection needs of them that does not depend on the pose (mean, world
 * covariance, activated opacity / colour / beta: 64 bytes each).  Lives in the candidate area behind the front's workspace
 * (gsx_front_workspace_bytes_cand); gsx_front_fwd / gsx_front_pose_bwd use it under GSX_PROJ_CANDIDATES.  Must be rebuilt
 * whenever the map's arrays change. */
int64_t gsx_front_workspace_bytes_cand(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap);
int gsx_front_candidates(const float *means, const float *quats, const float *scales, const float *viewmats_ref,
                         const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane, float far_plane,
                         int flags, const float *logit_opacities, const float *logit_colors, const float *log_uncertainties,
                         float rot_max, float trans_max, int64_t M_cap, void *workspace, int64_t workspace_bytes,
                         void *stream);
/* out4 = byte offsets in the workspace of {candidate header, candidate counts int32 [rows], candidate records [rows][slots]
 * x 64 bytes} and the number of header floats: [16][12] reference [R | t] rows, rot_max, trans_max, 0, 0, mode of the last
 * closure (1 = candidates used), number of closures that fell back to the full path since the set was built */
int gsx_front_cand_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4);

/* Pose gradient of a pose-only closure from the instance records gsx_front_fwd left in its workspace (same N, C, W, H,
 * M_cap): the pose part of gsx_project_bwd(v_means = NULL, flags | GSX_PROJ_VIEW_PARTIALS) over the visible instances
 * only.  v_rec [C,N,12]: the gradient records of gsx_raster_bwd (xy, conic and - with RENDER_DEPTH - depth columns read).
 * partials [gsx_front_rows(...)][C][12]: partial rows of d loss / d [R | t] (a few per front row and camera), to be summed by
 * gsx_track_opt_tail / gsx_pose_zhou_bwd_partials (n_blocks = gsx_front_rows). */
int64_t gsx_front_rows(int64_t N, int64_t C, int tile_w, int tile_h);
int gsx_front_pose_bwd(const float *means, const float *quats, const float *scales, const float *viewmats,
                       const float *Ks, int64_t N, int64_t C, int W, int H, float eps2d, float near_plane,
                       float far_plane, int flags, const float *v_rec, int64_t M_cap, const void *workspace,
                       int64_t workspace_bytes, float *partials, void *stream);
/* gsx_front_pose_bwd followed by gsx_track_opt_tail(loss_rows given) as ONE launch (round 5; one camera): the closure's tail - sum
 * of the partial rows, PoseZhou backward, one step of the tracking optimiser on (dt, dR, exposure), the next view matrix - is run
 * by the LAST workgroup of the pose backward to finish (two-level ticket in `tickets`: gsx_front_pose_bwd_tail_words() uint32,
 * zero before the first launch, left at zero by every launch).  Same arguments as the two calls; `partials` still receives the
 * rows.  gslam/frontend.py:621-658. */
int64_t gsx_front_pose_bwd_tail_words(void);
int gsx_front_pose_bwd_tail(const float *means, const float *quats, const float *scales, const float *viewmats, const float *Ks,
                            int64_t N, int W, int H, float eps2d, float near_plane, float far_plane, int flags,
                            const float *v_rec, int64_t M_cap, const void *workspace, int64_t workspace_bytes, float *partials,
                            void *state, const float *Rt, float *dt, float *dR, float *exposure, float *viewmat,
                            const void *loss_rows, int64_t n_loss_rows, float loss_coef, uint32_t *tickets, void *stream);

/* ---- K8: gsplat(fork) rasterize_to_pixels fwd (gslam/rasterization.py:325-339; fork: + n_touched) ----------------
 * rec: splat records [C*N, gsx_record_stride(CH)].  render [C,H,W,CH], alphas [C,H,W], last_ids [C,H,W] (global
 * sorted index of the last contributing entry, -1 if none), n_touched [C*N] int32 (must be zeroed by the caller).
 * offsets_has_end = 0: offsets has T entries and M is the exact number of intersections (gsplat layout);
 * offsets_has_end = 1: offsets has T+1 entries (gsx_isect_bin_sort) and M is the CAPACITY of flatten_ids - tile
 * ranges are clamped to it, nothing is read back to the host.
 */
int gsx_raster_fwd(const float *rec, int CH, const float *backgrounds /*[C,CH] nullable*/, const int32_t *offsets,
                   const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H, int tile_w,
                   int tile_h,
                   float visibility_min_T, float *render, float *alphas, int32_t *last_ids, int32_t *n_touched,
                   const int32_t *tile_order /*[T] nullable: workgroup i renders tile tile_order[i] (gsx_isect_bin_sort)*/,
                   void *stream);
/* K8 for the tracking closure (gslam/frontend.py:621-649): RGB + beta records (CH = 4), and in the same launch the
 * active-nerf tracking loss with the exposure affine (frontend.py:113-138,632-636; mode 2 of gsx_map_loss with
 * w_photo = weight / (C*H*W)): v_render [C,H,W,4] = d loss / d render, loss_rows [T][6] = one row of partial sums per tile
 * in the layout gsx_track_opt_tail / gsx_loss_finish read ([0] = sum of S / beta^2, [3], [4] = exposure gradient).
 * render is nullable (the closure never reads it); alphas and last_ids are what gsx_raster_bwd needs. */
int gsx_raster_fwd_track_loss(const float *rec, const float *backgrounds, const int32_t *offsets,
                              const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H,
                              const float *gt, const float *exposure, float w_photo, float *render, float *alphas,
                              int32_t *last_ids, float *v_render, float *loss_rows, const int32_t *tile_order,
                              int32_t *tile_work /*[T][2] nullable: (chunks, trips) per tile, the weights of gsx_tile_balance*/,
                              void *stream);
/* K8 + tracking loss + geometry-only K9 of the tracking closure in ONE launch: gsx_raster_fwd_track_loss followed by
 * gsx_raster_bwd(CH = 4, geometry_only = 1) with the same arguments, tile by tile inside one workgroup - colour, transmittance,
 * last entry and d loss / d render stay in registers.  v_rec [rows,12] accumulates (cleared by the caller / the projection);
 * alphas, last_ids and v_render are optional outputs (NULL: not written). */
int gsx_raster_track_fused(const float *rec, const float *backgrounds, const int32_t *offsets, const int32_t *flatten_ids,
                           int64_t M, int offsets_has_end, int64_t C, int W, int H, const float *gt, const float *exposure,
                           float w_photo, float *alphas, int32_t *last_ids, float *v_render, float *loss_rows, float *v_rec,
                           const int32_t *tile_order, int32_t *tile_work, void *stream);
/* The same launch for a front that stopped after the placement (GSX_PROJ_DEFER_SORT): every tile's workgroup first sorts the keys
 * of its segment up to the tile's depth cut-off in LDS - tile_cut [T], depth bits, in / out: 0x7f800000 = no cut-off (initial
 * value); the launch leaves depth of the deepest composited entry * (1 + cut_margin) for the NEXT closure of the same view - writes
 * their ids to flatten_ids and composites them; a tile whose pixels outlive its near list sorts its whole segment and goes on
 * (slower, same result).  "depth <= cut" is a prefix of the (depth, id) order, so what is composited is the reference's list, entry
 * for entry.  keys / keys_sorted: the key buffer of gsx_front_fwd's workspace and its scratch copy (gsx_front_keys: byte offsets
 * [0], [1] and the largest valid id [2]); tile_near [T] (nullable): how many entries of each tile ended up sorted (= valid in
 * flatten_ids); sort_stats [4] (nullable, accumulating): [0] slabs sorted behind a tile's first one, [1] segments that took the
 * through-memory merge sort (piles of equal depths; keys whose depth bits no window takes - NaN, sign bit), [2], [3] see
 * gsx_raster_track_fused_near.  offsets must carry its end ([T + 1] entries). */
int gsx_raster_track_fused_sorting(const float *rec, const float *backgrounds, const int32_t *offsets, int32_t *flatten_ids,
                                   int64_t M, int offsets_has_end, int64_t C, int W, int H, const float *gt,
                                   const float *exposure, float w_photo, float *alphas, int32_t *last_ids, float *v_render,
                                   float *loss_rows, float *v_rec, const int32_t *tile_order, int32_t *tile_work,
                                   uint64_t *keys, uint64_t *keys_sorted, uint32_t id_max, uint32_t *tile_cut,
                                   float cut_margin, int32_t *tile_near, int32_t *sort_stats, void *stream);
int gsx_front_keys(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int flags, int64_t *out3);
/* out4 = { byte offset of the row segments of keys ([rows][row_cap] x 8 bytes), byte offset of the row words (int32 [C][tiles][rows]),
 * row_cap, rows } of a front run with GSX_PROJ_ROW_KEYS in a workspace of gsx_front_workspace_bytes(.., M_cap) */
int gsx_front_rows_layout(int64_t N, int64_t C, int tile_w, int tile_h, int64_t M_cap, int64_t *out4);
/* gsx_raster_track_fused_sorting behind a front that ran with GSX_PROJ_ROW_KEYS: every tile's workgroup first COLLECTS its keys -
 * reads its column of row words, reserves a segment of the contiguous key buffer with one atomic on one of the 64 counters M_dev [64]
 * (segments are handed out in the order the tiles ask; tile_span [T][2] = {start, count} of every tile, out), copies the rows'
 * stretches there and feeds the
 * keys of the first depth window straight into the LDS sort - and then runs exactly the launch above on that segment.  Two launches
 * (column scan, placement) and the 8-fold re-read of the instance records are gone from the closure's chain; what a tile composites
 * is the reference's list, entry for entry (gslam/rasterization.py:259-274).  status: the plan's sticky status word (bit 1: the
 * key buffer is full). */
int gsx_raster_track_fused_rows(const float *rec, const float *backgrounds, int32_t *flatten_ids, int64_t M_cap, int64_t N,
                                int64_t C, int W, int H, const float *gt, const float *exposure, float w_photo,
                                int loss_kind /* 0: the tracker's loss sum_k err_k^2 / beta^2 (gslam/frontend.py:113-138); 1: the
                                                 window refiner's sum_k err_k^2 / (2 beta^2) + log(beta)^2 / 2 (gslam/backend.py:
                                                 484-492), loss_rows column 1 = the log term */,
                                float *alphas,
                                int32_t *last_ids, float *v_render, float *loss_rows, float *v_rec, const int32_t *tile_order,
                                int32_t *tile_work, uint32_t *tile_cut, float cut_margin, int32_t *tile_near,
                                int32_t *sort_stats, int32_t *tile_span, int64_t *M_dev, int32_t *status,
                                void *front_workspace, int64_t workspace_bytes, void *stream);
/* static LDS bytes of a workgroup of gsx_raster_track_fused_sorting's kernel, read off the loaded code object (-1 on error) */
int64_t gsx_raster_track_fused_lds_bytes(void);

/* ---- Near placement (round 5): the depth cut-off of a pose-only closure applied where the tile lists are BUILT, not only where they
 * are sorted.  A tracking closure composites the nearest 17-27 % of a tile's keys before every pixel of the tile has saturated
 * (gslam/frontend.py:604-662 evaluates 36 renders per frame from nearly the same pose), so the front of the next closure counts an
 * (instance, tile) pair behind the tile's cut-off of the previous closure (tile_cut [T], as left by the rasteriser launch below)
 * WITHOUT placing its key: offsets [T + 1], M_dev and status describe the FULL lists exactly as gsx_front_fwd leaves them - every
 * tile's segment keeps room for all of its keys - but only the keys with depth bits <= tile_cut[t] are written, at the front of
 * the segment, unsorted; tile_placed [T] tells how many.  gsx_front_fwd_near = gsx_front_fwd(flags | GSX_PROJ_COMPACT |
 * GSX_PROJ_DEFER_SORT) restricted to what such a closure reads (no radii / means2d / depths / conics / tiles_per_gauss /
 * vis_count / flatten_ids / tile_order outputs).
 * gsx_raster_track_fused_near = gsx_raster_track_fused_sorting over such segments: a tile whose pixels outlive its placed keys
 * appends the keys behind the cut-off to its segment ITSELF, from the front's instance records (inst_recs / n_inst / R / seg_cap:
 * gsx_front_layout out4[2], [3], [0], [1] of the same workspace; compact = GSX_PROJ_COMPACT was set), and goes on slab by slab -
 * slower (one workgroup reads every instance record of its camera), rare, and the same list: what is composited is a prefix of
 * the reference's sorted list, entry for entry (gslam/rasterization.py:259-274).  A tile whose pixels outlive its WHOLE list
 * leaves no cut-off (0x7f800000).  sort_stats [4] (accumulating): [0] slabs sorted behind a tile's first one, [1] segments sorted
 * through memory, [2] tiles that appended their far keys, [3] tiles whose appended count disagreed with offsets (must stay 0). */
int gsx_front_fwd_near(const float *means, const float *quats, const float *scales, const float *viewmats, const float *Ks,
                       int64_t N, int64_t C, int W, int H, float eps2d, float near_plane, float far_plane, int flags,
                       const float *logit_opacities, const float *logit_colors, const float *log_uncertainties, float *rec,
                       float *v_rec_clear, int64_t M_cap, int32_t *offsets, int64_t *M_dev, int32_t *status,
                       const int32_t *tile_work, int32_t *balanced_order, float chunk_cost, float light_rate, int n_cus,
                       void *workspace, int64_t workspace_bytes, const uint32_t *tile_cut, int32_t *tile_placed, void *stream);
int gsx_raster_track_fused_near(const float *rec, const float *backgrounds, const int32_t *offsets, int32_t *flatten_ids,
                                int64_t M, int offsets_has_end, int64_t C, int W, int H, const float *gt,
                                const float *exposure, float w_photo, float *alphas, int32_t *last_ids, float *v_render,
                                float *loss_rows, float *v_rec, const int32_t *tile_order, int32_t *tile_work,
                                uint64_t *keys, uint64_t *keys_sorted, uint32_t id_max, uint32_t *tile_cut,
                                float cut_margin, int32_t *tile_near, int32_t *sort_stats, const int32_t *tile_placed,
                                const void *inst_recs, const int32_t *n_inst, int64_t R, int64_t seg_cap, int compact,
                                void *stream);
/* Launch order for the rasteriser kernels of a render whose T workgroups are all resident at once (T <= 2048): deals the
 * tiles into n_cus groups of near-equal weight (weight = trips + chunk_cost * chunks of tile_work, as measured by an earlier
 * gsx_raster_fwd_track_loss of a nearby pose) and writes tile_order [T] so that the workgroups i, i + n_cus, i + 2 n_cus, ...
 * - which an idle MI355X places on one CU - hold one group.  Always a permutation: the weights only matter for speed. */
int gsx_tile_balance(const int32_t *tile_work, int64_t T, float chunk_cost,
                     float light_rate /* (0,1]: throughput of a CU holding one workgroup less than the fullest, relative */,
                     int n_cus, int32_t *tile_order, void *stream);

/* ---- K9: rasterize_to_pixels bwd.  v_rec [C*N, stride] must be zeroed by the caller; gradients are accumulated
 * in record layout: v_xy(2) v_conic(3) v_opacity(1) v_colors(CH).  v_abs (nullable, [C*N,2], zeroed): absgrad.
 * v_alphas nullable (= zero gradient).                                                                              */
int gsx_raster_bwd(const float *rec, int CH, const float *backgrounds, const int32_t *offsets,
                   const int32_t *flatten_ids, int64_t M, int offsets_has_end, int64_t C, int W, int H, int tile_w,
                   int tile_h,
                   const float *alphas, const int32_t *last_ids, const float *v_render, const float *v_alphas,
                   float *v_rec, float *v_abs, const int32_t *tile_order /*[T] nullable, as in gsx_raster_fwd*/,
                   int geometry_only /* != 0: only the xy / conic columns of v_rec are wanted (tracking against a frozen map,
                                        no depth channel); the opacity / colour columns are then left as they are */,
                   void *stream);

/* ---- K13: spherical harmonics (gsplat.rendering.rasterization(sh_degree=)) -------------------------- */
int gsx_sh_fwd(int degree, const float *dirs /*[C,N,3]*/, const float *coeffs /*[N,Kc,3]*/, const int32_t *radii,
               int64_t N, int64_t C, int Kc, float *colors /*[C,N,3]*/, void *stream);
int gsx_sh_bwd(int degree, const float *dirs, const float *coeffs, const int32_t *radii, const float *v_colors,
               int64_t N, int64_t C, int Kc, float *v_coeffs /*[N,Kc,3] overwritten*/, float *v_dirs /*[C,N,3] nullable*/,
               void *stream);
/* The same with the view directions formed in registers: dir(c, g) = means[g] - campos[c] (campos [C,3] = inverse view
 * matrices' translation column), so no [C,N,3] direction array is written and read back.  Backward: v_coeffs as above,
 * v_means [N,3] = sum over cameras of d loss / d dir (written, nullable), v_campos [C,3] -= sum over Gaussians (ADDED:
 * zero it first; nullable). */
int gsx_sh_fwd_means(int degree, const float *means, const float *campos, const float *coeffs, const int32_t *radii,
                     int64_t N, int64_t C, int Kc, float *colors, void *stream);
int gsx_sh_bwd_means(int degree, const float *means, const float *campos, const float *coeffs, const int32_t *radii,
                     const float *v_colors, int64_t N, int64_t C, int Kc, float *v_coeffs, float *v_means,
                     float *v_campos, void *stream);

/* ---- K11/K12: fused_ssim (gslam/backend.py:303-307).  strides: HOST int64[4] in floats (sB,sC,sH,sW) ------------------------
 * fwd: out_sum[0] = sum of the SSIM map over the crop (crop = 5 for 'valid', 0 for 'same'); the mean is
 * out_sum/(B*CH*(H-2crop)*(W-2crop)).  dm_* (nullable when train=0): three [B,CH,H,W] planar maps for the bwd.    */
int64_t gsx_ssim_workspace_bytes(int64_t B, int CH, int H, int W);
/* number of per-workgroup partial sums gsx_ssim_fwd leaves at the start of its workspace when out_sum == NULL */
int64_t gsx_ssim_partials(int64_t B, int CH, int H, int W);
int gsx_ssim_fwd(const float *img1, const float *img2, int64_t B, int CH, int H, int W, const int64_t *strides1,
                 const int64_t *strides2, int crop, float *out_sum, float *dm_dmu1, float *dm_dsigma1_sq,
                 float *dm_dsigma12, void *workspace, int64_t workspace_bytes, void *stream);
/* bwd: dL_dimg1 [B,CH,H,W] planar = scale[0] * (conv(w; m*dm_dmu1) + 2 img1 conv(m*dm_ds1) + img2 conv(m*dm_ds12)),
 * m = crop mask; `scale` is a DEVICE scalar (upstream grad / numel folded by the caller into scale_mul).           */
int gsx_ssim_bwd(const float *img1, const float *img2, int64_t B, int CH, int H, int W, const int64_t *strides1,
                 const int64_t *strides2, int crop, const float *dm_dmu1, const float *dm_dsigma1_sq,
                 const float *dm_dsigma12, const float *scale, float scale_mul, float *dL_dimg1, void *stream);

/* ---- fused loss block (gslam/backend.py:273-318 mapping; gslam/frontend.py:113-138 tracking) -----------------------
 * One pass over the render: value partial sums AND the analytic gradient w.r.t. the render / exposure.
 *  mode 0: active-gs mapping loss   sum_px [ S/(2 b^2) + 0.5 log^2 b ],  S = sum_k (rgb_k e^a + b_c - gt_k)^2
 *  mode 1: plain mse on the un-exposed rgb (backend.py:285)
 *  mode 2: active-nerf tracking     sum_px S / b^2                        (frontend.py:127)
 * w_photo multiplies the per-pixel photometric gradient (caller folds weight / (C*H*W) into it); w_tv multiplies the
 * edge-aware depth TV SUM of gslam/utils.py:136-161 (mask = alphas > mask_thresh, owned by the left / upper pixel).
 * ssim_grad (nullable, planar [C,3,H,W], already scaled) is added to the rgb gradient.
 * out: sums[0..2] = raw sums of (photometric term, 0.5 log^2 beta term, tv); v_render [C,H,W,CH]; v_exposure [C,2]. */
int64_t gsx_map_loss_workspace_bytes(int64_t C, int H, int W);
int gsx_map_loss(const float *render, const float *alphas, const float *gt, const float *exposure, int64_t C, int H,
                 int W, int CH, int depth_index, int beta_index, int mode, float w_photo, float w_tv,
                 float mask_thresh, const float *ssim_grad, float *sums, float *v_render, float *v_exposure,
                 void *workspace, int64_t workspace_bytes, void *stream);
/* gsx_ssim_bwd(scale_mul) + gsx_map_loss(ssim_grad = its output) in ONE pass (round 5): render [C,H,W,CH] (colours first), gt
 * [C,H,W,3], the three derivative maps gsx_ssim_fwd left ([C,3,H,W] planar), crop as given to it; the other arguments and
 * v_render as gsx_map_loss.  d loss / d render comes out bit for bit as from the two launches (same expressions, same order;
 * csrc/loss_pixel.h); the planar SSIM gradient never exists.  Leaves gsx_ssim_bwd_map_loss_rows(C, H, W) rows of 6 partial sums
 * in `workspace` (>= rows * 24 bytes): finish with gsx_loss_finish(workspace, C, rows / C, 256, ...) - it reads one row per 256
 * "pixels" of an H x W = (rows / C) x 256 image. */
int64_t gsx_ssim_bwd_map_loss_rows(int64_t C, int H, int W);
int gsx_ssim_bwd_map_loss(const float *render, const float *alphas, const float *gt, const float *exposure, int64_t C, int H,
                          int W, int CH, int depth_index, int beta_index, int mode, float w_photo, float w_tv,
                          float mask_thresh, int crop, const float *dm_dmu1, const float *dm_dsigma1_sq,
                          const float *dm_dsigma12, const float *scale, float scale_mul, float *v_render, void *workspace,
                          int64_t workspace_bytes, void *stream);
/* isotropic regulariser (backend.py:287-296): sum_out[0] = sum over visible g of sum_j |exp(s_j) - exp(mean s)|;
 * v_log_scales [N,3] = weight * d/ds (mean detached), zero rows for invisible Gaussians. */
int64_t gsx_isotropic_workspace_bytes(int64_t N);
int gsx_isotropic_loss(const float *log_scales, const int32_t *vis_count, int64_t N, float weight, float *sum_out,
                       float *v_log_scales, void *workspace, int64_t workspace_bytes, void *stream);
/* same, but v_log_scales += (the regulariser lands directly in the parameter's gradient; untouched rows are skipped) */
int gsx_isotropic_loss_acc(const float *log_scales, const int32_t *vis_count, int64_t N, float weight, float *sum_out,
                           float *v_log_scales, void *workspace, int64_t workspace_bytes, void *stream);
/* One finishing launch for a whole loss block.  gsx_map_loss(sums = NULL), gsx_ssim_fwd(out_sum = NULL) and
 * gsx_isotropic_loss[_acc](sum_out = NULL) skip their own one-workgroup reductions and leave their per-workgroup partial
 * rows in their workspaces; this call sums them: sums5 (nullable) = raw (photometric, 0.5 log^2 beta, tv, ssim,
 * isotropic) sums, v_exposure [C,2] (nullable), out2[i] = bias_i + sum_k coef_i[k] * sums5[k] (coef: HOST float[5]).
 * ssim_ws / iso_ws may be NULL (ssim_partials = 0 / term absent); C, H, W, N as given to the producers. */
int gsx_loss_finish(const void *map_loss_ws, int64_t C, int H, int W, const void *ssim_ws, int64_t ssim_partials,
                    const void *iso_ws, int64_t N, const float *coef0, const float *coef1, float bias0, float bias1,
                    float *sums5, float *v_exposure, float *out2, void *stream);
/* out2[i] = bias_i + sum_k coef_i[k] * terms[k][0]  (device scalars in, device scalars out; host arrays of pointers) */
int gsx_combine_terms(int n, const float *const *terms, const float *coef0, const float *coef1, float bias0,
                      float bias1, float *out2, void *stream);
/* logit_opacities[g] *= decay where vis_count[g] > min_count (backend.py:356-359) */
int gsx_opacity_decay(float *logit_opacities, const int32_t *vis_count, int64_t N, int min_count, float decay,
                      void *stream);

/* ---- Warp (gslam/warp.py:35-82).  T = f1_pose @ inv(f2_pose) [4,4], K, Kinv [3,3] device pointers ---------------- */
int gsx_warp_fwd(const float *T, const float *K, const float *Kinv, const float *c1 /*[H,W,3]*/,
                 const float *d1 /*[H,W]*/, int H, int W, float *result /*[H,W,3]*/, float *nwarps /*[H,W,2]*/,
                 uint8_t *keep_mask /*[H,W]*/, void *stream);
int64_t gsx_warp_bwd_workspace_bytes(int H, int W);
int gsx_warp_bwd(const float *T, const float *K, const float *Kinv, const float *c1, const float *d1, int H, int W,
                 const float *v_result, const float *v_nwarps /*nullable*/, float *v_T /*[4,4] overwritten*/,
                 void *workspace, int64_t workspace_bytes, void *stream);

/* ---- PoseZhou for a window of <= 16 poses in one launch (gslam/primitives.py:15-36,82-92).  Rt/dR/dt (and v_dR/v_dt)
 * are HOST arrays of device pointers, one per pose ([4,4], [6], [3]); learnable: host int[C] (0 -> viewmat = Rt). */
int gsx_pose_zhou_fwd(int C, const float *const *Rt, const float *const *dR, const float *const *dt,
                      const int *learnable, float *viewmats /*[C,4,4]*/, void *stream);
int gsx_pose_zhou_bwd(int C, const float *const *Rt, const float *const *dR, const float *const *dt,
                      const int *learnable, const float *v_viewmats, float *const *v_dR, float *const *v_dt,
                      void *stream);
/* The same backward fed by the pose-gradient partials gsx_project_bwd(flags | GSX_PROJ_VIEW_PARTIALS) left in its
 * workspace ([n_blocks][C][12], n_blocks = gsx_project_bwd_blocks(N)): sums them (the order of the projection's own
 * finishing pass), adds v_viewmats_extra [C,4,4] if not NULL, and runs the PoseZhou backward - one launch instead of
 * the finishing pass plus gsx_pose_zhou_bwd. */
int gsx_pose_zhou_bwd_partials(int C, const float *const *Rt, const float *const *dR, const float *const *dt,
                               const int *learnable, const float *partials, int64_t n_blocks,
                               const float *v_viewmats_extra /*nullable*/, float *const *v_dR, float *const *v_dt,
                               void *stream);

/* ---- fused Adam (torch.optim.Adam(fused=True) defaults; gslam/backend.py:565-602).  Up to 32 tensors per launch
 * (every numel >= 1).
 * step_dev: device int64 holding the 1-based step AFTER increment is step_dev[0]+1; kernel does not modify it
 * when step_host > 0 (then step_host is used).                                                                      */
int gsx_adam_multi(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                   float *const *exp_avg_sq, const int64_t *numels, const float *lrs, float beta1, float beta2,
                   float eps, int64_t step_host, const int64_t *step_dev /*nullable device int64: overrides step_host*/,
                   void *stream);

/* gsx_adam_multi_steps + a masked post-update decay of ONE of the tensors: params[decay_tensor][j] *= decay where
 * decay_mask[j] > decay_min_count (the opacity decay of gslam/backend.py:356-359 riding in the update before it;
 * decay_tensor = -1: none).  decay_mask has one int32 per element of that tensor. */
int gsx_adam_multi_steps_decay(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                               float *const *exp_avg_sq, const int64_t *numels, const float *lrs, float beta1,
                               float beta2, float eps, const int64_t *const *steps, int decay_tensor,
                               const int32_t *decay_mask, int decay_min_count, float decay, void *stream);
/* Same update with one device step counter PER TENSOR (steps: HOST array of n_tensors DEVICE int64 pointers; the
 * counters hold the 1-based step of this update), so tensors that joined the optimiser at different times (poses of
 * new keyframes, backend.py:665-670) update in the same launch.  gsx_counters_add bumps up to 16 such counters in one
 * launch (what torch's capturable Adam does with one `step += 1` kernel per parameter). */
int gsx_adam_multi_steps(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                         float *const *exp_avg_sq, const int64_t *numels, const float *lrs, float beta1, float beta2,
                         float eps, const int64_t *const *steps, void *stream);
int gsx_counters_add(int n, int64_t *const *counters, int64_t delta, void *stream);
/* The update behind a DEVICE-SIDE GATE: gsx_adam_multi_steps_decay / gsx_counters_add that do nothing when
 * skip_if_positive[0] > 0 (nullable: no gate).  The mapping plans set that float from the sticky overflow status of the
 * iteration's render (gsx_status_flag: flag = (any(status[i] & mask & ~2) ? 1 : 0) + (any(status[i] & mask & 2) ? 1024 : 0):
 * status bit 2 - clamped, corrupt tile counts - weighs 1024, so the sum over <= 1023 ranks still tells it from a plain
 * overflow) - summed over ranks by the iteration's one
 * all-reduce - so an iteration whose tile lists were truncated on ANY rank applies no update on EVERY rank (no roll-back
 * needed, the replicas stay identical); the host reads the flag with the loss value it reads anyway (the reference's
 * loss.item(), gslam/backend.py:351), grows the lists and redoes the iteration. */
int gsx_adam_multi_steps_gated(int n_tensors, float *const *params, const float *const *grads, float *const *exp_avg,
                               float *const *exp_avg_sq, const int64_t *numels, const float *lrs, float beta1,
                               float beta2, float eps, const int64_t *const *steps, int decay_tensor /* -1: none */,
                               const int32_t *decay_mask, int decay_min_count, float decay,
                               const float *skip_if_positive, void *stream);
int gsx_counters_add_gated(int n, int64_t *const *counters, int64_t delta, const float *skip_if_positive, void *stream);
int gsx_status_flag(const int32_t *status, int n, int mask, float *flag, void *stream);
/* Staging copy of the ranged multi-GPU exchange (new design; gslam_amd.dist.StepBucket): slice t = `parts`
 * consecutive parts of part_len[t] floats (a multiple of 4, 16-byte aligned) starting at slices[t]; staging = `parts` blocks of
 * sum(part_len) floats, block r = [part r of slice 0 | part r of slice 1 | ...].  to_flat = 0: slices -> staging; 1: back.
 * n_slices <= 8; host arrays of pointers / lengths. */
int gsx_range_copy(int n_slices, float *const *slices, const int64_t *part_len, int parts, float *staging, int to_flat,
                   void *stream);

/* ---- device-resident tracking optimiser: the host logic of gslam/frontend.py:604-662 (10 torch.optim.Adam steps, then
 * ONE torch.optim.LBFGS(line_search_fn='strong_wolfe').step()) as a state machine advanced once per closure
 * evaluation, so that a tracked frame needs no loss.item() read-back (frontend.py:648).  `state`: device buffer of
 * gsx_track_opt_state_bytes().  advance: params/grads/numels are HOST arrays (<= 4 tensors, <= 16 scalars in all) of
 * device pointers; `loss` is a device float.  After advance the parameters hold the next evaluation point, or the
 * result once the machine is done (further calls are no-ops).  report: out8 (device) = phase (4 = done),
 * evaluations, L-BFGS iterations, stop reason, last evaluated loss, loss at the result, L-BFGS evaluations, Adam steps. */
int64_t gsx_track_opt_state_bytes(void);
int gsx_track_opt_init(void *state, int n_params, int n_adam, float lr_adam, double lr_lbfgs, int history, int max_iter,
                       int max_eval, double tol_grad, double tol_change, void *stream);
int gsx_track_opt_advance(void *state, int n_tensors, float *const *params, const float *const *grads,
                          const int *numels, const float *loss, void *stream);
int gsx_track_opt_report(const void *state, float *out8, void *stream);
/* The tail of the tracker's closure in one launch (one camera): reduces the pose-gradient partials that
 * gsx_project_bwd(flags | GSX_PROJ_VIEW_PARTIALS) left in its workspace, runs the PoseZhou backward (gslam/primitives.py:
 * 82-92), advances the state machine on the 11 parameters (dt[3], dR[6], exposure[2], in this order; v_exposure from the
 * loss block), writes them back and writes the PoseZhou forward of the new parameters to viewmat [4,4].
 * loss_rows != NULL: the tracking loss is finished here too - the n_loss_rows per-workgroup rows that
 * gsx_map_loss(sums = NULL, C = 1) left in its workspace give loss = loss_coef * (col 0 + col 1) and the exposure
 * gradient (col 3, col 4); v_exposure / loss are then ignored and gsx_loss_finish is not needed. */
int gsx_track_opt_tail(void *state, const void *pose_partials, int64_t n_blocks, const float *Rt, float *dt, float *dR,
                       float *exposure, const float *v_exposure, const float *loss, float *viewmat,
                       const void *loss_rows, int64_t n_loss_rows, float loss_coef, void *stream);
/* The same state machine sized for the backend's window pose refinement (gslam/backend.py:447-506): up to 80 parameters
 * in up to 16 tensors, L-BFGS history up to 10; n_adam = 0 skips the Adam phase.  Arguments as above. */
int64_t gsx_window_opt_state_bytes(void);
int gsx_window_opt_init(void *state, int n_params, int n_adam, float lr_adam, double lr_lbfgs, int history, int max_iter,
                        int max_eval, double tol_grad, double tol_change, void *stream);
int gsx_window_opt_advance(void *state, int n_tensors, float *const *params, const float *const *grads,
                           const int *numels, const float *loss, void *stream);
int gsx_window_opt_report(const void *state, float *out8, void *stream);
int gsx_window_opt_tail(void *state, const void *pose_partials, int64_t n_blocks, const float *Rt, float *dt, float *dR,
                        float *exposure, const float *v_exposure, const float *loss, float *viewmat,
                        const void *loss_rows, int64_t n_loss_rows, float loss_coef, void *stream);

/* ---- map maintenance: every per-Gaussian array re-packed in ONE launch.
 * gsx_gather_rows: dst[k][r] = src[k][index[r]] for r < n_out, k < n_tensors (<= 32); the masked re-allocation of
 *   gslam/pruning.py:24-47 (`splat_param[keep_mask]`, `value[keep_mask]` per parameter and Adam moment) and the
 *   `_duplicate` / `_split` selections of gslam/insertion.py:66-96.  row_words: width of one row of tensor k in 4-byte
 *   words (int64 ages = 2).  index: device int64 [n_out] (e.g. nonzero(keep_mask)); values are clamped to [0, n_src).
 * gsx_concat_rows: dst[k] = [a[k] (n_a rows); b[k] (n_b rows)], b[k] == NULL meaning zero rows; the torch.cat pairs of
 *   gslam/insertion.py:38-58 (`_add_new_splats`: parameters + zero Adam moments).
 * src / dst / a / b / row_words are HOST arrays; the pointers in them are DEVICE pointers. */
int gsx_gather_rows(int n_tensors, const void *const *src, void *const *dst, const int *row_words, const int64_t *index,
                    int64_t n_out, int64_t n_src, void *stream);
int gsx_concat_rows(int n_tensors, const void *const *a, int64_t n_a, const void *const *b, int64_t n_b,
                    void *const *dst, const int *row_words, void *stream);

/* ---- streams and HIP graphs (csrc/runtime.hip).  A closure of the tracking / mapping optimisers is a fixed chain of the
 * launches above over caller-owned persistent buffers; it is recorded once by capturing the stream it is issued on and
 * replayed with one call - what replaces the reference's per-iteration Python loops (gslam/frontend.py:621-658,
 * gslam/backend.py:260-359,465-504).  Nothing may be allocated, freed or synchronised on the capturing thread between
 * gsx_graph_begin and gsx_graph_end.  capture mode: 0 global, 1 thread-local, 2 relaxed (hipStreamCaptureMode). */
int gsx_stream_create(void **stream_out);                    /* non-blocking stream */
int gsx_stream_create_masked(void **stream_out, const uint32_t *cu_mask, int mask_words);   /* kernels of this stream run on the
                                                                 CUs whose bit is set only (32 CUs per word) */
int gsx_stream_destroy(void *stream);
int gsx_stream_synchronize(void *stream);
int gsx_stream_wait_stream(void *stream, void *other);       /* device-side: `stream` waits for what `other` holds now */
int gsx_graph_begin(void *stream, int mode);
int gsx_graph_end(void *stream, void **exec_out, int64_t *n_nodes_out /*nullable*/);
int gsx_graph_abort(void *stream);                           /* leave capture mode after a failed launch; no graph */
int gsx_graph_launch(void *exec, void *stream);
int gsx_graph_launch_n(void *exec, int count, void *stream); /* count launches back to back */
int gsx_graph_destroy(void *exec);
/* pinned, device-mapped host memory (zeroed): kernels write status words the host polls without a stream sync */
int gsx_host_alloc(void **host_out, void **dev_out, int64_t bytes);
int gsx_host_free(void *host);
/* n_words 4-byte words at ptr := 0, by a kernel (a captured hipMemsetAsync node misbehaved on replay, ROCm 7.2) */
int gsx_zero_words(void *ptr, int64_t n_words, void *stream);
/* Self-check of the CU-balanced launch order (gsx_tile_balance): n_wgs workgroups of 256 threads and lds_bytes of LDS that
 * stay resident for spin_us microseconds write where they ran: keys[i] = XCC_ID << 8 | SE_ID << 5 | SH_ID << 4 | CU_ID of
 * workgroup i.  The balanced order assumes keys[i] == keys[i mod G] with G distinct values on a drained chip; the host
 * verifies that once per device and shape and falls back to the identity order otherwise. */
int gsx_probe_wg_placement(int n_wgs, int lds_bytes, int spin_us, int32_t *keys, void *stream);

/* ---- self tests of device primitives (wave64 reductions); returns 0 if all pass.  scratch: >= 64 KiB device ----- */
int gsx_selftest(void *scratch, int64_t scratch_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H */
