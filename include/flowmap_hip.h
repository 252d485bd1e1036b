/* flowmap_hip.h — C ABI of libflowmap_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary for FlowMap's per-iteration reprojection / flow-consistency
 * inner loop.  The reference (dcharatan/flowmap) has NO native layer: the path is a
 * Python call surface (flowmap/model/projection.py, flowmap/model/procrustes.py,
 * flowmap/loss/) executed as chains of ATen ops.  Each entry point below replaces
 * one such chain; the citation names the reference code it stands in for.  The
 * Python host mirror (flowmap_amd/) binds these through ctypes and re-exposes the
 * reference's function / class names; INTEGRATION.md shows the binding a maintainer
 * of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to contiguous row-major fp32 data unless
 *     stated otherwise (double = fp64 workspace, int64_t = index tensors);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call
 *     is asynchronous on that stream, never synchronises, never allocates;
 *   - return value: 0 ok, 1 invalid argument, 2 launch/runtime failure.  Nothing
 *     throws across the boundary;
 *   - B batch, F frames, H×W pixels, N = H·W, P Procrustes points, pair i = frames
 *     (i, i+1); poses are 4×4 row-major, intrinsics 3×3 row-major, normalised image
 *     coordinates (x right, y down, both in (0,1)).
 */
#ifndef FLOWMAP_HIP_H
#define FLOWMAP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FM_MAPPING_HUBER 0 /* flowmap/loss/mapping/mapping_huber.py:18-34 */
#define FM_MAPPING_L1 1    /* flowmap/loss/mapping/mapping_l1.py:15-20 */
#define FM_MAPPING_L2 2    /* flowmap/loss/mapping/mapping_l2.py:15-21 */

#define FM_FLOW_ACC_STRIDE 20  /* doubles per (frame, direction) in `acc` */
/* Version of this interface: bumped whenever an entry point changes its arguments or its contract (round 3: the workspaces of
 * fm_flow_loss_fused / fm_procrustes_fit_chain are self-cleaning — zero on entry, left zero — instead of being cleared by the
 * call; fm_scale_if_needed reports a non-unit scalar; round 4, version 4: the tap exchange entry
 * points fm_flow_loss_fused_taps / fm_track_loss_fused_fwd_taps / fm_tap_grad_apply; round 5, version 5: fm_tap_grad_apply reports a
 * non-zero correction through a device flag; fm_track_presample and the `presampled` argument of fm_track_loss_fused_fwd_taps; round 6,
 * version 6: that entry point and that argument are gone again — measured, not adopted: docs/history/patches/r05_track_presample.patch).  A binding checks fm_abi_version() == FM_ABI_VERSION when it loads the library. */
#define FM_ABI_VERSION 6
int fm_abi_version(void);

#define FM_STAT_STRIDE 16      /* doubles per pair in `stats` */
#define FM_AUX_STRIDE 40       /* doubles per pair in `aux` */
#define FM_PAIR_GRAD_STRIDE 20 /* doubles per pair in `pair_grad` */

/* ---------------------------------------------------------------------------------
 * Fused flow loss (the roofline kernel).
 * Replaces LossFlow.compute_unweighted_loss (flowmap/loss/loss_flow.py:31-70) with
 * everything under it — unproject (projection.py:76-90), compute_forward_flow /
 * compute_backward_flow (projection.py:143-184), reproject_points (:116-134),
 * project_camera_space (:49-58), Mapping.forward (loss/mapping/mapping.py:35-43) —
 * AND the autograd backward of that chain, in one pass over HBM.
 *
 *   depth (B,F,H,W); k, kinv (B,F,3,3); t_fwd (B,F-1,4,4) camera i -> camera i+1;
 *   t_bwd (B,F-1,4,4) camera i+1 -> camera i; flow_* (B,F-1,H,W,2); mask_* (B,F-1,H,W).
 *   scale: device scalar multiplied into every residual's gradient (weight/valid_sum
 *          from fm_flow_valid_norm); NULL = loss only, no gradients.
 *   grad_depth (B,F,H,W) out: dL/ddepth with poses held fixed (may be NULL).
 *   acc (B*F, 2, FM_FLOW_ACC_STRIDE) fp64 in/out: per (source frame, direction) sums are ADDED into it: it must
 *          be zero on entry.  fm_flow_loss_finalize consumes it and leaves it zero again, so a workspace kept
 *          across steps is zeroed once, when it is allocated.
 *   packed: NULL, or flows + masks re-laid-out by fm_flow_pack_inputs — then flow_* / mask_*
 *          are not read (may be NULL).  Same bytes, one stream instead of six (needs W % 4 == 0).
 *   items_per_thread: tuning knob (<=0 -> default).
 * Hard limits (return 1, "invalid argument", when exceeded — the host layer names the limit in its message): batch·frames <= 65 535
 * (one grid row per source frame), H·W < 2^30 (32-bit pixel indices inside a frame; base pointers are 64-bit, so frames·H·W is
 * unlimited: configs[4] whole, 2.5e9 elements, runs), W % 4 == 0 and 16-byte aligned buffers for the 16-byte path (else the scalar
 * path runs).  Determinism: dL/ddepth is written once per element (bit-reproducible); the 13 per-(frame, direction) sums meet in
 * `acc` through fp64 ATOMICS across workgroups, so the loss and the pose / intrinsics gradients derived from them can differ in their
 * last bits from run to run (tests: 1e-6 relative); the tracking loss's sums (fm_track_loss_*) are reduced in a fixed order and are
 * bit-reproducible.
 */
int fm_flow_loss_fused(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                       const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                       const float* packed, const float* scale, int batch, int frames, int height, int width,
                       int mapping_kind, float delta, float aspect_x, float aspect_y, float* grad_depth, double* acc,
                       int items_per_thread, void* stream);

/* fm_flow_loss_fused with the Adam update of the depth parameter (model_wrapper_overfit.py:104-105: torch.optim.Adam,
 * no weight decay) applied BY THE SAME PASS — SURVEY.md §8f-2's end state: no dL/ddepth round trip through HBM and no
 * separate optimiser pass over depth (48 B per pixel and frame instead of 32 + 28).  `depth`, `exp_avg`, `exp_avg_sq`
 * (B,F,H,W) are rewritten in place for every pixel whose bit in `touched` is clear; `touched` (B·F·H·W/4 bytes: bit e of
 * byte q = pixel 4q+e) marks the pixels another operator of the step still reads or adds gradient to (the Procrustes
 * samples and their taps, the track taps): they keep their values and get dL/ddepth written to grad_depth (only those
 * entries of grad_depth are written); fm_adam_step_elements updates them once their gradient is complete.  The gradient
 * is final as computed (`scale` must already hold everything upstream).  Needs W % 4 == 0; step >= 1 is the step number
 * of the bias corrections. */
int fm_flow_loss_fused_adam(float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd, const float* flow_fwd,
                            const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* packed, const float* scale,
                            int batch, int frames, int height, int width, int mapping_kind, float delta, float aspect_x, float aspect_y,
                            float* grad_depth, double* acc, int items_per_thread, float* exp_avg, float* exp_avg_sq, const uint8_t* touched,
                            long step, double lr, double beta1, double beta2, double eps, void* stream);

/* The tap exchange between the fused flow loss and the fused tracking loss (round 4).  Track positions are constants of an
 * optimisation (flowmap/tracking/__init__.py:49-70 computes them once per video), so the set of depth pixels the tracking loss
 * bilinearly samples (projection.py:266-272) — its "taps" — is STATIC: M distinct pixels, ranked in (frame, row, column) order
 * (fm_track_scatter_plan's sorted `pixels`).  fm_flow_taps describes that set to the pass that streams every depth pixel anyway:
 *   chunk_base (B·F·ceil(H·W/256) + 1) int32: rank of the first tap at or after quad 64c of frame bf (chunks of 64 quads per frame; the
 *              last entry is M) — a workgroup of the pass owns a run of consecutive quads of a frame, hence the taps [base(first), base(end));
 *   pixel      (M) int32: each tap's pixel index inside its frame (row·W + col);
 *   grad       (M) or NULL: the tracking loss's UNSCALED dL/ddepth at the taps (fm_track_loss_fused_fwd_taps' tap_grad) — the pass
 *              adds scale[0]·grad[rank] into the dL/ddepth it writes (and into the gradient its in-pass Adam update uses);
 *   scale      device scalar: the tracking loss's weight / max(count, 1) (its `scale` output), assumed to reach backward() unscaled
 *              (fm_tap_grad_apply corrects the dense gradient otherwise);
 *   depth      (M) out or NULL: the depth value the pass leaves at each tap (after an in-pass Adam update: the updated one) — the
 *              tracking loss's next evaluation samples from these M floats instead of 4 cold lines per track point;
 *   stale      one int32 on the device or NULL (without the Adam update only): set to 1 when a tap's depth differs from the value `depth`
 *              held on entry, i.e. when the tracking loss of this step sampled an image that no longer matched the depth tensor.
 * fm_flow_loss_fused_taps = fm_flow_loss_fused (exp_avg NULL) or fm_flow_loss_fused_adam (exp_avg etc. given) with that exchange;
 * dense depth, W % 4 == 0, gradients on. */
typedef struct fm_flow_taps {
  const int32_t* chunk_base;
  const int32_t* pixel;
  const float* grad;
  const float* scale;
  float* depth;
  int32_t* stale;
} fm_flow_taps;
int fm_flow_loss_fused_taps(float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd, const float* flow_fwd,
                            const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, const float* packed, const float* scale,
                            int batch, int frames, int height, int width, int mapping_kind, float delta, float aspect_x, float aspect_y,
                            float* grad_depth, double* acc, int items_per_thread, const fm_flow_taps* taps, float* exp_avg, float* exp_avg_sq,
                            const uint8_t* touched, long step, double lr, double beta1, double beta2, double eps, void* stream);

/* Flows and masks are constants of an optimisation (computed once by
 * FlowPredictor.compute_bidirectional_flow, flowmap/flow/flow_predictor.py:82-102).  Copies them
 * once into the layout the fused kernel streams best: per source frame f and quad q (4
 * consecutive pixels), six float4 — forward flow of pair f (2), its mask (1), backward flow of
 * pair f-1 (2), its mask (1) — stored [b·F+f][q/64][vector 0..5][q%64]; absent pairs and
 * padding lanes are zero.  packed holds B·F·ceil(H·W/256)·6·64·4 floats (24 B per pixel and
 * frame, i.e. the size of the originals).  Requires W % 4 == 0 and 16-byte aligned inputs. */
int fm_flow_pack_inputs(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                        int frames, int height, int width, float* packed, void* stream);

/* Turns `acc` into (and then clears it): loss[0] = norm[0]·Σρm (loss.py:47 weight and loss_flow.py:70
 * normalisation folded into norm[0]); g_t_fwd / g_t_bwd (B,F-1,4,4) = dL/dT (bottom
 * rows 0); g_k (B,F,3,3) = dL/dK through both the projection (rows 0,1) and K⁻¹. */
int fm_flow_loss_finalize(double* acc, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                          const float* norm, int batch, int frames, float aspect_x, float aspect_y, float* loss, float* g_t_fwd,
                          float* g_t_bwd, float* g_k, void* stream);

/* valid_sum of loss_flow.py:56,66,70: vsum[0] = Σmask_fwd + Σmask_bwd (fp64);
 * norm[0] = weight / (vsum or 1), norm[1] = (vsum or 1).  `count` = elements per mask. */
int fm_flow_valid_norm(const float* mask_fwd, const float* mask_bwd, long count, float weight, double* vsum, float* norm,
                       void* stream);

/* x[i] *= scalar[0] and y[i] *= scalar[0] unless scalar[0] == 1 (autograd's grad_output at the
 * root); either buffer may be NULL with a count of 0.  not_one (optional, one int32 on the device): set to 1 when
 * scalar[0] != 1 and left alone otherwise — the in-pass Adam update (fm_flow_loss_fused_adam) has used the UNSCALED gradient
 * by then, so its caller watches this flag (round 3; FM_ABI_VERSION 3). */
int fm_scale_if_needed(float* x, long count, float* y, long count_y, const float* scalar, int* not_one, void* stream);

/* ---------------------------------------------------------------------------------
 * Procrustes pose fit.  Replaces align_surfaces (projection.py:213-252) +
 * align_rigid (flowmap/model/procrustes.py:7-51) and their backward.
 * Exactly one source must be given: (depth, kinv) — xyz recomputed on the fly — or
 * `surfaces` (B,F,H,W,3).  indices (P) int64 flat pixel indices, NULL = arange(N).
 * stats (B*(F-1), FM_STAT_STRIDE) fp64 out: Σw, Σw·p, Σw·q, covariance.
 * weight_sensitivity: 0 = `weights` are the correspondence weights; s != 0 = `weights`
 * are LOGITS and w = sigmoid(s·logit) is applied at the gathered points only
 * (BackboneExplicitDepth, backbone_explicit_depth.py:38-41), grad_weights then being
 * the gradient w.r.t. the logits.
 * batch_repeat R >= 1: kinv and the fitted poses have `batch` entries while depth /
 * surfaces / bwd_flow / weights (and their gradients) have batch/R entries, each shared by
 * R consecutive pose-batch entries — the candidate sweep of IntrinsicsSoftmin
 * (intrinsics_softmin.py:92-103 repeats the images 60x; here they are read in place).
 */
int fm_procrustes_stats(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                        const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch,
                        int batch_repeat, int frames, int height, int width, double* stats, void* stream);

/* fm_procrustes_stats followed by fm_pose_solve (below) for the same pairs, the per-pair
 * normalisation and the solve sharing one launch: what align_surfaces does up to the pose chain
 * (projection.py:213-249).  `stats` is the workspace / by-product. */
int fm_procrustes_fit(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                      float weight_sensitivity, const int64_t* indices, long points, int batch, int batch_repeat, int frames,
                      int height, int width, double* stats, float* t_bwd, float* t_fwd, double* aux, void* stream);

/* fm_procrustes_fit (batch_repeat 1) AND fm_pose_chain_fwd in ONE launch (projection.py:187-252 entire): up to 4096
 * points one workgroup of 1024 threads per pair keeps the pair's sums in LDS, solves the pose, and the last workgroup to
 * finish chains the poses into ext (B,F,4,4); larger index sets use four workgroups per pair and fp64 atomics, the last of a
 * pair solving.  ext may be NULL (poses only).
 * work: persistent workspace of B·(F-1)·FM_STAT_STRIDE doubles followed by B·(F-1)+1 ints, ZERO on entry and left
 * zero (self-cleaning: zero it once, when it is allocated; one launch at a time per workspace).
 * corr_out (optional, needs tap_records; (B·(F-1)·P, 8) floats, 16-byte aligned, P <= 4096): the record of every correspondence —
 * q (3), p (3), w, the bits of its pixel index — for fm_procrustes_bwd_planned, which then re-gathers nothing.
 * tap_records (optional, (B·(F-1)·P, 8), 16-byte aligned; depth source, indices given, P <= 4096): with constant flows and indices the
 * taps of every correspondence are static — per correspondence the four taps' pixel offsets row·W + col in the earlier frame
 * (int32 bits, -1 = outside the image) followed by their four bilinear weights, i.e. slots 0..3 of fm_procrustes_scatter_plan's
 * keys (minus the frame's base) and weights.  The kernel then does not read bwd_flow and issues all scattered reads of a
 * correspondence in one dependent round. */
int fm_procrustes_fit_chain(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                            float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                            double* work, float* t_bwd, float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records,
                            void* stream);

/* procrustes.py:35-51: R = U·diag(1,1,±1)·Vᵀ by in-register 3×3 SVD, t = q̄ − R·p̄.
 * t_bwd (pairs,4,4) = [R|t] ("inverse relative transformation", later -> earlier
 * camera); t_fwd (pairs,4,4) = its inverse (may be NULL); aux (pairs, FM_AUX_STRIDE). */
int fm_pose_solve(const double* stats, int pairs, float* t_bwd, float* t_fwd, double* aux, void* stream);

/* Backward of fm_pose_solve (replaces linalg_svd_backward et al.).  g_t_bwd / g_t_fwd
 * (pairs,4,4) may be NULL.  pair_grad (pairs, FM_PAIR_GRAD_STRIDE) fp64 out.  `clear` (optional,
 * clear_count doubles) is zeroed by the same launch: the kinv_acc that fm_procrustes_scatter,
 * next on the stream, accumulates into. */
int fm_pose_solve_bwd(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux, int pairs,
                      double* pair_grad, double* clear, long clear_count, void* stream);
/* fm_pose_solve_bwd for the B·(F-1) pairs of a video batch AND dL/dK⁻¹ of every frame: it is linear in the statistics
 * fm_pose_solve left in `aux` (Σw, p̄, q̄, M), so kinv_acc (B·F,9) fp64 is WRITTEN here, one thread per frame (earlier role of its
 * pair + later role of the previous one), and the per-point passes (fm_procrustes_scatter / _scatter_dense with
 * kinv_acc = NULL) carry no sums for it.  kinv (B,F,3,3): the inverse intrinsics the fit was given. */
int fm_pose_solve_bwd_kinv(const float* g_t_bwd, const float* g_t_fwd, const float* t_bwd, const double* aux, const float* kinv, int batch,
                           int frames, double* pair_grad, double* kinv_acc, void* stream);

/* Per-point backward (replaces grid_sampler_2d_backward + index_put of
 * projection.py:226-249).  ATOMICALLY ADDS into grad_depth (B,F,H,W) or grad_surfaces
 * (B,F,H,W,3) and grad_weights (B,F-1,H,W): callers zero them (or pass a buffer that
 * already holds another gradient to fuse the accumulation).  kinv_acc (B*F,9) fp64:
 * dL/dK⁻¹ accumulators (caller zeroes), depth source only.  Any output may be NULL.
 * The dense depth-sourced case (surfaces == NULL, indices == NULL, points == H·W, batch_repeat == 1)
 * has its own atomic-free entry point, fm_procrustes_scatter_dense below; through this one it runs on
 * the generic kernels. */
int fm_procrustes_scatter(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                          const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch,
                          int batch_repeat, int frames, int height, int width, const double* aux, const double* pair_grad,
                          float* grad_depth, float* grad_surfaces, float* grad_weights, double* kinv_acc, float* point_grads,
                          float* point_weight_grads, void* stream);

/* Dense Procrustes backward (`num_points: null`, config/experiment/ablation_explicit_depth.yaml:11-12:
 * every pixel of every pair is a correspondence; replaces grid_sampler_2d_backward + index_put of
 * projection.py:226-249 over all H·W pixels).  Planned form, WITHOUT atomics on the big tensors: the pattern is static
 * (the flows are constants of the optimisation), so it is planned once per flow tensor:
 *   fm_procrustes_dense_tiles(H, W, &tiles)                      tiles of the earlier frame per pair (host only)
 *   pass 1: fm_procrustes_dense_plan(bwd_flow, B, F, H, W, counts, NULL, NULL)   counts (B·(F-1)·tiles) int32,
 *           zeroed by the caller, += 1 per (later pixel, tile its bilinear taps land in)
 *   first  = exclusive prefix sum of counts, (B·(F-1)·tiles + 1) int64            [caller]
 *   pass 2: fm_procrustes_dense_plan(bwd_flow, B, F, H, W, cursor, first, list)   cursor: zeroed int32 of the
 *           same size; list (first[last]) uint32 = row << 16 | col of the later pixels, grouped by tile and, inside a
 *           tile, in ascending order (sorted by the same call: a wave of the per-step kernel then reads nearly
 *           consecutive pixels).
 * Per step fm_procrustes_scatter_dense: grad_weights (B,F-1,H,W) is STORED (every element exactly once:
 * need not be zeroed, must not hold another gradient); grad_depth (B,F,H,W) is ADDED to with plain
 * read-modify-writes (each pixel has one writer per launch).  dL/dK⁻¹ is not formed here: it is linear in the
 * statistics of the forward pass and comes from fm_pose_solve_bwd_kinv.
 * grad_depth / grad_weights may be NULL.
 * first / list: both given = the planned pair of kernels described above (no atomics; dL/ddepth bit-reproducible; 2.0-2.2 ms per
 * 150 x 720x1280 step on ANY flow).  Both NULL (round 3) = ONE fused pass over the later pixels that needs no plan: the tap gradients
 * are summed in an LDS image of the earlier-frame window the block stages for its depth samples (64-bit integer sums in a per-pair
 * fixed point) and leave with float atomics, the later pixel's own gradient is an atomic add too — 1.2 ms when the flow varies by a
 * few pixels inside a 32x64 tile (camera motion), but every tap that leaves the window (+-4 rows / +-8 columns around the tile
 * displaced by the flow at its centre) is a scattered atomic: slower than the planned kernels on rough flows.  Same gradients to
 * 1e-7 (norm-wise).  flowmap_amd's host layer picks per flow tensor (_ops._dense_flow_is_rough).
 * consts: workspace of B·(F-1)·FM_DENSE_CONST_STRIDE doubles (per-pair constants, written by the call). */
#define FM_DENSE_CONST_STRIDE 40
int fm_procrustes_dense_tiles(int height, int width, int* tiles);
int fm_procrustes_dense_plan(const float* bwd_flow, int batch, int frames, int height, int width, int* counts, const int64_t* first,
                             uint32_t* list, void* stream);
int fm_procrustes_scatter_dense(const float* depth, const float* kinv, const float* bwd_flow, const float* weights,
                                float weight_sensitivity, int batch, int frames, int height, int width, const double* aux,
                                const double* pair_grad, float* grad_depth, float* grad_weights, const int64_t* first, const uint32_t* list,
                                double* consts, void* stream);

/* Planned form of the sparse depth-sourced scatter.  With DISTINCT `indices` and constant flows the
 * pixels a step's Procrustes gradient touches never change: fm_procrustes_scatter_plan lists them
 * once — keys (B·(F-1)·P, 5) int64 = frame·H·W + pixel for the four taps in the earlier frame and
 * (slot 4) the correspondence's own pixel in the later frame, -1 for a tap outside; weights the
 * bilinear weights (1 for slot 4).  Per step fm_procrustes_scatter is then called with
 * point_grads (B·(F-1)·P, 2, 3) != NULL (depth source, indices given, batch_repeat 1): it writes
 * dL/dq, dL/dp per correspondence there, STORES grad_weights at the sampled pixels and leaves
 * grad_depth alone; fm_depth_gather (vectors = point_grads, entries = 2·(key index / 5) + (slot == 4))
 * adds the depth gradient without atomics.  With point_weight_grads (B·(F-1)·P) != NULL as well, dL/dweight
 * per correspondence goes there instead of into grad_weights, and fm_sparse_store places it later —
 * so the dense buffer can still be being zeroed (fm_fill_zero on another stream) while this runs. */
int fm_procrustes_scatter_plan(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                               int64_t* keys, float* weights, void* stream);

/* out[g·stride + indices[j]] = values[g·points + j] for g < groups, j < points (see above). */
int fm_sparse_store(const float* values, const int64_t* indices, long points, int groups, long stride, float* out, void* stream);

/* get_extrinsics (projection.py:187-210): ext (B,steps+1,4,4), ext[0]=I,
 * ext[k] = ext[k-1]·rel[k-1]; and its backward (replaces the Python loop of matmuls). */
int fm_pose_chain_fwd(const float* rel, int batch, int steps, float* ext, void* stream);
int fm_pose_chain_bwd(const float* rel, const float* ext, const float* g_ext, int batch, int steps, float* g_rel, void* stream);

/* later(E).inverse() @ earlier(E) and earlier(E).inverse() @ later(E)
 * (projection.py:154,176) with a general 4×4 inverse, and the backward (g_ext is
 * overwritten; g_fwd / g_bwd may be NULL). */
int fm_relative_pose_fwd(const float* ext, int batch, int frames, float* fwd, float* bwd, void* stream);
int fm_relative_pose_bwd(const float* ext, const float* g_fwd, const float* g_bwd, int batch, int frames, float* g_ext,
                         void* stream);

/* extrinsics_target.inverse() @ extrinsics_source for every (source, target) pair of a
 * track segment (projection.py:288): ext (B,f,4,4) -> rel (B,f,f,4,4) indexed
 * [b, source, target]; backward g_rel -> g_ext (overwritten). */
int fm_allpairs_pose_fwd(const float* ext, int batch, int frames, float* rel, void* stream);
int fm_allpairs_pose_bwd(const float* ext, const float* g_rel, int batch, int frames, float* g_ext, void* stream);

/* focal_lengths_to_intrinsics (flowmap/model/intrinsics/common.py:6-20) as IntrinsicsRegressed.forward
 * uses it (flowmap/model/intrinsics/intrinsics_regressed.py:34-41): focal (count) -> k (count*repeat, 3, 3),
 * each focal length repeated over `repeat` consecutive frames, fx = f*sqrt(hw)/w, fy = f*sqrt(hw)/h,
 * cx = cy = 0.5; kinv (optional, same shape) receives K^-1 as fm_intrinsics_inverse computes it. */
int fm_focal_intrinsics_fwd(const float* focal, long count, long repeat, int height, int width, float* k, float* kinv, void* stream);
/* Its backward: grad_focal[i] = sqrt(hw) * sum over the i-th `repeat` frames of (grad_k[0][0]/w + grad_k[1][1]/h). */
int fm_focal_intrinsics_bwd(const float* grad_k, long count, long repeat, int height, int width, float* grad_focal, void* stream);

/* intrinsics.inverse() of unproject (projection.py:86) for `count` 3×3 matrices, and
 * g_k (+)= −K⁻ᵀ·kinv_acc·K⁻ᵀ. */
int fm_intrinsics_inverse(const float* k, int count, float* kinv, void* stream);
int fm_intrinsics_inverse_bwd(const double* kinv_acc, const float* kinv, int count, float* g_k, int accumulate, void* stream);

/* ---------------------------------------------------------------------------------
 * Function-level building blocks on explicit point sets (the standalone call surface
 * used by the visualiser, IntrinsicsSoftmin and the exporters).  G = number of groups
 * (one small matrix per group), `points` = points per group.
 */

/* unproject (projection.py:76-90): out (G,points,3) = (kinv_g·[x,y,1])·z.
 * xy is (G,points,2) with xy_group_stride = points*2, or one shared (points,2) grid
 * with xy_group_stride = 0.  bwd: g_z (G,points), kinv_acc (G,9) fp64 (zeroed here). */
int fm_unproject_fwd(const float* xy, long xy_group_stride, const float* z, const float* kinv, int groups, long points,
                     float* out, void* stream);
int fm_unproject_bwd(const float* xy, long xy_group_stride, const float* z, const float* kinv, const float* g_out, int groups,
                     long points, float* g_z, double* kinv_acc, void* stream);

/* reproject_points (projection.py:116-134) with project_camera_space (:49-58):
 * xyz (G,points,3), t (G,4,4), k (G,3,3) -> xy (G,points,2).  bwd: g_xyz (nullable),
 * g_t (G,4,4) bottom row 0, g_k (G,3,3) bottom row 0; acc (G,18) fp64 workspace. */
int fm_reproject_fwd(const float* xyz, const float* t, const float* k, int groups, long points, float* xy, void* stream);
int fm_reproject_bwd(const float* xyz, const float* t, const float* k, const float* g_xy, int groups, long points, float* g_xyz,
                     float* g_t, float* g_k, double* acc, void* stream);

/* F.grid_sample(mode="bilinear", padding_mode="border", align_corners=False) as used
 * at projection.py:235-241,266-272: img (G,H,W,C) channels-last, xy (G,points,2) in
 * (0,1) -> out (G,points,C).  bwd ATOMICALLY ADDS into g_img (caller zeroes). */
int fm_bilinear_sample_fwd(const float* img, const float* xy, int groups, int height, int width, int channels, long points,
                           float* out, void* stream);
int fm_bilinear_sample_bwd(const float* g_out, const float* xy, int groups, int height, int width, int channels, long points,
                           float* g_img, void* stream);

/* Mapping.forward (loss/mapping/mapping.py:35-43) with huber / l1 / l2:
 * a, b (count,2) -> out (count);  bwd: g_a, g_b (count,2), either nullable. */
int fm_mapping_fwd(const float* a, const float* b, long count, int kind, float delta, float aspect_x, float aspect_y, float* out,
                   void* stream);
int fm_mapping_bwd(const float* a, const float* b, const float* g_out, long count, int kind, float delta, float aspect_x,
                   float aspect_y, float* g_a, float* g_b, void* stream);

/* align_rigid (procrustes.py:7-51) on explicit p, q (G,points,3), w (G,points):
 * statistics for fm_pose_solve, and the per-point backward given fm_pose_solve_bwd's
 * pair_grad. */
int fm_align_rigid_stats(const float* p, const float* q, const float* w, int groups, long points, double* stats, void* stream);
int fm_align_rigid_bwd(const float* p, const float* q, const float* w, int groups, long points, const double* aux,
                       const double* pair_grad, float* g_p, float* g_q, float* g_w, void* stream);

/* ---------------------------------------------------------------------------------
 * Fused point-tracking loss.  Replaces LossTracking.compute_unweighted_loss
 * (flowmap/loss/loss_tracking.py:28-61) + compute_track_flow (projection.py:255-298)
 * for ALL segments at once, batch 1 (the reference asserts b == 1,
 * tracking/__init__.py:89-90).  Tracks are packed once by the host:
 *   xy (total,2) fp32, vis (total) uint8, total = Σ_s f_s·P_s, point (s, frame, p) at
 *   seg[s].offset + frame·P_s + p;  seg (S,4) int32 = {start_frame, f, P, offset};
 *   blocks (nblocks,2) int32 = every (segment, local frame);  pmax = max P, fmax = max f.
 * depth (F,H,W); k, kinv (F,3,3); ext, ext_inv (F,4,4).
 */
int fm_extrinsics_inverse(const float* ext, int count, float* inv, void* stream); /* general 4x4 inverse, projection.py:288 */

/* Per (frame-in-segment, point): ws (total·9 floats: an opaque work space between these calls) = [xyz camera-space point sampled from the
 * source frame's surface (projection.py:266-274) | X_w = E·xyz | h = Σ_taps w·z·[u,v,1]];
 * flag (total) = visibility ∧ source-in-frame (projection.py:290-294).  Per frame: tgt
 * (frames,12) = rows 0,1 of K·inv(E)[:3,:] and row 2 of inv(E) (the target-role projection
 * u = q·(au·[X_w;1]), q = 1/(c·[X_w;1]+eps), projection.py:288,49-58).
 * Frame sharding: kinv / ext / ext_inv / k cover all `frames` of the video, `depth` only the
 * frames from depth_frame0 on (0 = unsharded); `blocks` then lists only the (segment, frame)
 * entries this rank owns as SOURCES and the caller zeroes `flag` first (an unlisted entry is an
 * invisible source). */
int fm_track_points(const float* depth, int depth_frame0, const float* kinv, const float* ext, const float* ext_inv, const float* k,
                    int frames, const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks, int pmax,
                    int height, int width, float* ws, uint8_t* flag, float* tgt, void* stream);

/* All (source, target, point) residuals of all segments, each evaluated once.
 *   tiles (ntiles,2) int32: (segment, first local source frame) for source tiles of
 *   FM_TRACK_TILE frames; fmax = longest segment;
 *   partial: fp32 workspace of ntiles·ceil(pmax/64)·FM_TRACK_PARTIAL(fmax) floats (per-wave sums,
 *   reduced per frame without atomics: results are bit-reproducible run to run);
 *   acc (frames*20) fp64 out.
 *   loss[0] = weight·Σρ·vis/max(Σvis,1); scale[0] = weight/max(Σvis,1), scale[1] = Σvis;
 *   totals (2 doubles, may be NULL) = [Σρ·vis, Σvis] for callers that reduce them across
 *   shards and overwrite loss / scale before the backward calls read `scale`.
 *   gws (total,3) and acc2 (frames*24): both NULL = loss only; else UNSCALED dL/dxyz per
 *   track point and the per-frame source-role sums for fm_track_loss_bwd. */
#define FM_TRACK_TILE 6
#define FM_TRACK_PARTIAL(fmax) ((fmax) * 14 + FM_TRACK_TILE * 21)
int fm_track_loss_fwd(const float* ws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                      const int32_t* tiles, int ntiles, int pmax, int fmax, const float* ext, const float* tgt, int frames, int height,
                      int width, int mapping_kind, float delta, float aspect_x, float aspect_y, float weight, float* partial,
                      double* acc, float* loss, float* scale, double* totals, float* gws, double* acc2, void* stream);

/* fm_track_points + fm_track_loss_fwd in one call, the sampling done by the pair kernel itself: every wave samples
 * the points of its own tile's source frames in its prologue and writes ws / flag (outputs here) for its epilogue and
 * for the backward calls; tgt (frames,12) is an output too.  own_first / own_end: the frames [own_first, own_end)
 * whose points act as SOURCES here (frame sharding; 0 and `frames` when the whole video is local) — `tiles` lists
 * the tiles with at least one such frame, other frames of a listed tile get flag = 0. */
int fm_track_loss_fused_fwd(const float* depth, int depth_frame0, int own_first, int own_end, const float* kinv, const float* ext,
                            const float* ext_inv, const float* k, int frames, const float* xy, const uint8_t* vis, const int32_t* seg,
                            const int32_t* tiles, int ntiles, int pmax, int fmax, int height, int width, int mapping_kind, float delta,
                            float aspect_x, float aspect_y, float weight, float* ws, uint8_t* flag, float* tgt, float* partial, double* acc,
                            float* loss, float* scale, double* totals, float* gws, double* acc2, void* stream);

/* fm_track_loss_fused_fwd (whole video local) on the static tap set (fm_flow_taps above; loss_tracking.py:28-61 /
 * projection.py:266-272 unchanged in value):
 *   tap_slot (total,4) int32: tap k of track point i is the tap of rank (tap_slot[4i+k] & 0x1fffffff); bit 30: another track point shares
 *     the pixel; bit 29: the pixel is updated by another operator after the flow pass; -1: the tap contributes nothing;
 *   tap_depth (M + 1 readable) or NULL: with it, the tap depths are read from the compact image fm_flow_loss_fused_taps left behind —
 *     tap_depth[rank] — instead of the depth image (slots with bit 29 read `depth` after all);
 *   tap_grad (M) out or NULL, with the plan of fm_track_scatter_plan sorted as fm_depth_gather takes it (plan_count = M): the UNSCALED
 *     dL/ddepth of the tracking loss at each tap, for the `grad` member of fm_flow_taps (needs gws).  shared_ranks (shared_count) int32 or
 *     NULL: the ranks of the taps with more than one plan entry — with the list (and tap_slot) the pair kernel stores the gradient of
 *     every other tap itself and only these are summed from the plan; without it all M are. */
int fm_track_loss_fused_fwd_taps(const float* depth, const float* kinv, const float* ext, const float* ext_inv, const float* k, int frames,
                                 const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* tiles, int ntiles, int pmax, int fmax,
                                 int height, int width, int mapping_kind, float delta, float aspect_x, float aspect_y, float weight, float* ws,
                                 uint8_t* flag, float* tgt, float* partial, double* acc, float* loss, float* scale, double* totals, float* gws,
                                 double* acc2, const int32_t* tap_slot, const float* tap_depth, const int64_t* plan_pixels,
                                 const int32_t* plan_first, const int32_t* plan_entries, const float* plan_weights, long plan_count,
                                 const int32_t* shared_ranks, long shared_count, float* tap_grad, void* stream);
/* grad_depth[pixels[m]] += scale[0]·(plus − minus)·tap_grad[m] for the M taps (plus / minus: device scalars, NULL = 0); no memory is
 * touched when the factor is 0.  The correction of the tap exchange when the tracking loss's upstream gradient (plus) differs from
 * the factor the flow pass's copy of it was delivered with (minus), and the plain scatter (minus NULL) when nothing was delivered.
 * mismatch_flag (one int32 on the device, or NULL): set to 1 when the factor is not 0 — the caller passes it when something has ALREADY
 * consumed the absorbed gradient at factor 1 and cannot be corrected (the in-pass Adam update of the absorbing flow pass). */
int fm_tap_grad_apply(const float* tap_grad, const int64_t* pixels, long count, const float* scale, const float* upstream_plus,
                      const float* upstream_minus, float* grad_depth, int* mismatch_flag, void* stream);

/* g_ext (F,4,4), g_k (F,3,3) from acc / acc2, multiplied by scale[0]·upstream[0]
 * (upstream NULL = 1). */
int fm_track_loss_bwd(const double* acc, const double* acc2, const float* scale, const float* upstream, const float* ext_inv,
                      const float* k, const float* kinv, int frames, float* g_ext, float* g_k, void* stream);

/* Scatter scale[0]·upstream[0]·gws through the bilinear taps: ATOMICALLY ADDS into grad_depth,
 * which holds the frames from depth_frame0 on (as `depth` in fm_track_points). */
int fm_track_scatter(const float* gws, const uint8_t* flag, const float* xy, const uint8_t* vis, const int32_t* seg,
                     const int32_t* blocks, int nblocks, int pmax, const float* kinv, const float* scale, const float* upstream,
                     int height, int width, int depth_frame0, float* grad_depth, void* stream);

/* The pattern of fm_track_scatter is static — track positions and visibility are inputs of the
 * optimisation — so it can be planned once: keys (total,4) int64 = frame·H·W + row·W + col of tap k
 * of source point i (global frame index), -1 for a tap that contributes nothing (invisible /
 * outside / clipped); weights (total,4) its bilinear weight.  Only the (segment, frame) entries of
 * `blocks` are written: pre-fill keys with -1. */
int fm_track_scatter_plan(const float* xy, const uint8_t* vis, const int32_t* seg, const int32_t* blocks, int nblocks, int pmax, int height,
                          int width, int64_t* keys, float* weights, void* stream);

/* A planned scatter executed as a gather, without atomics and deterministic: pixels (count) int64 =
 * the distinct keys >= 0 of a plan in ascending order (key = frame·H·W + row·W + col); first
 * (count+1) int32 = where each pixel's entries begin; entries (E) int32 = index of the 3-vector
 * each entry contributes, weights (E) its weight, both sorted by key.
 *   grad_depth[key − frame0·H·W] += scale[0]·upstream[0]·Σ_e weights[e]·<vectors[entries[e]], K⁻¹(frame)·[u,v,1]>
 * (scale / upstream NULL = 1).  With fm_track_scatter_plan: vectors = gws, entries = plan index / 4.
 * With fm_procrustes_scatter_plan: vectors = point_grads of fm_procrustes_scatter. */
int fm_depth_gather(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries, const float* weights,
                    long count, const float* kinv, const float* scale, const float* upstream, int height, int width, long frame0,
                    float* grad_depth, void* stream);
/* fm_depth_gather (scale = upstream = NULL, frame0 = 0) with fm_intrinsics_inverse_bwd riding in the same launch (a few
 * extra workgroups): g_k (frames_k,3,3) = [accumulate ? g_k : 0] − K⁻ᵀ·kinv_acc·K⁻ᵀ. */
int fm_depth_gather_kgrad(const float* vectors, const int64_t* pixels, const int32_t* first, const int32_t* entries, const float* weights,
                          long count, const float* kinv, int height, int width, float* grad_depth, const double* kinv_acc, int frames_k,
                          float* g_k, int accumulate, void* stream);

/* The WHOLE backward of a planned sparse fit in one launch (fm_pose_solve_bwd_kinv + fm_procrustes_scatter with point_grads +
 * fm_depth_gather_kgrad, projection.py:226-249 / procrustes.py:7-51 backward): one workgroup per frame (b, f) reads the
 * records (corr: what fm_procrustes_fit_chain left in corr_out) of the two pairs the frame belongs to — coalesced, no gather
 * chain —, keeps their gradient vectors in LDS and adds its slice of the gather plan into grad_depth (plain
 * read-modify-writes, one writer per pixel), STORES grad_weights (B,F-1,H,W) at the sampled pixels of pair f and writes
 * g_k[b, f] = [accumulate_k ? g_k : 0] − K⁻ᵀ·dK⁻¹·K⁻ᵀ with dK⁻¹ in closed form from `aux`.
 * Depth source, batch_repeat 1, distinct indices, points <= FM_FIT_BWD_MAX_POINTS.  weight_sensitivity as given to the fit.
 * plan_* as fm_depth_gather takes them for fm_procrustes_scatter_plan; frame_first (B·F + 1) int32: index of the first plan
 * pixel with key >= frame·H·W (the plan is sorted by key).  g_t_bwd / g_t_fwd (B·(F-1),4,4) may be NULL; grad_depth /
 * grad_weights / g_k may be NULL. */
#define FM_FIT_BWD_MAX_POINTS 2048
int fm_procrustes_bwd_planned(const float* corr, const float* kinv, float weight_sensitivity, long points, int batch, int frames, int height, int width,
                              const double* aux, const float* t_bwd, const float* g_t_bwd, const float* g_t_fwd, const int64_t* plan_pixels,
                              const int32_t* plan_first, const int32_t* plan_vectors, const float* plan_weights, const int32_t* frame_first,
                              float* grad_depth, float* grad_weights, float* g_k, int accumulate_k, void* stream);

/* ---------------------------------------------------------------------------------
 * Frame windows read in place (SURVEY.md §8b "Ownership"; round 3).
 * The reference hands its functions views — `earlier(x)` / `later(x)` = x[:, :-1], x[:, 1:] (projection.py:139-140), the
 * segment windows `surfaces[:, s:s+f]` of the tracking loss (loss_tracking.py:44-52), slices of a pretraining batch.  Such a
 * window of a (B, F_full, ...) tensor is a pointer to its first frame plus two strides: fm_layout, in ELEMENTS of the tensor —
 * frame_stride between consecutive frames, batch_stride between batch entries ({0, 0} = dense: frames back to back, batch
 * entries back to back).  The `_views` entry points below are their dense namesakes with one fm_layout per image-stack
 * argument (a HOST array; NULL = all dense): the kernels apply the strides when they form each frame's base pointer, so a
 * window costs no copy.  Everything these functions WRITE (gradients, workspaces) stays dense.  For the 16-byte paths
 * (W % 4 == 0) the strides must be multiples of 4 elements, else the scalar path runs.
 *   fm_flow_loss_fused_views       layouts[5] = depth, flow_fwd, flow_bwd, mask_fwd, mask_bwd
 *   fm_flow_valid_norm_views       layouts[2] = mask_fwd, mask_bwd (each (batch, pairs, pixels))
 *   fm_flow_pack_inputs_views      layouts[4] = flow_fwd, flow_bwd, mask_fwd, mask_bwd
 *   fm_procrustes_fit_views / fm_procrustes_fit_chain_views / fm_procrustes_scatter_views (batch_repeat 1, sparse index sets)
 *                                  layouts[4] = depth, surfaces, bwd_flow, weights
 *   fm_procrustes_scatter_plan_views   one layout: bwd_flow
 * --------------------------------------------------------------------------------- */
typedef struct fm_layout {
  long frame_stride; /* elements between consecutive frames (0 with batch_stride 0: dense) */
  long batch_stride; /* elements between batch entries */
} fm_layout;

int fm_flow_loss_fused_views(const float* depth, const float* k, const float* kinv, const float* t_fwd, const float* t_bwd,
                             const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd,
                             const float* packed, const float* scale, int batch, int frames, int height, int width,
                             int mapping_kind, float delta, float aspect_x, float aspect_y, float* grad_depth, double* acc,
                             int items_per_thread, const fm_layout* layouts, void* stream);
int fm_flow_valid_norm_views(const float* mask_fwd, const float* mask_bwd, int batch, int pairs, long pixels, float weight, double* vsum,
                             float* norm, const fm_layout* layouts, void* stream);
int fm_flow_pack_inputs_views(const float* flow_fwd, const float* flow_bwd, const float* mask_fwd, const float* mask_bwd, int batch,
                              int frames, int height, int width, float* packed, const fm_layout* layouts, void* stream);
int fm_procrustes_fit_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                            float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                            double* stats, float* t_bwd, float* t_fwd, double* aux, const fm_layout* layouts, void* stream);
int fm_procrustes_fit_chain_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow, const float* weights,
                                  float weight_sensitivity, const int64_t* indices, long points, int batch, int frames, int height, int width,
                                  double* work, float* t_bwd, float* t_fwd, double* aux, float* ext, float* corr_out, const float* tap_records,
                                  const fm_layout* layouts, void* stream);
int fm_procrustes_scatter_views(const float* depth, const float* kinv, const float* surfaces, const float* bwd_flow,
                                const float* weights, float weight_sensitivity, const int64_t* indices, long points, int batch, int frames,
                                int height, int width, const double* aux, const double* pair_grad, float* grad_depth, float* grad_surfaces,
                                float* grad_weights, double* kinv_acc, float* point_grads, const fm_layout* layouts, void* stream);
int fm_procrustes_scatter_plan_views(const float* bwd_flow, const int64_t* indices, long points, int batch, int frames, int height, int width,
                                     int64_t* keys, float* weights, const fm_layout* flow_layout, void* stream);

/* ---------------------------------------------------------------------------------
 * Frame sharding: the local work of the halo exchange (flowmap_amd/sharding.py; no reference counterpart — the reference
 * replicates the video per rank, flowmap/overfit.py:94-108).  grad (frames, frame_elements): this rank's dL/ddepth; its first frame is
 * shared with rank−1 ("first"), its last with rank+1 ("last"); a NULL buffer switches that side off.
 *   fm_halo_copy     sent_* = the boundary frames as they are now (the early exchange sends these while backward still adds to grad)
 *   fm_halo_delta    out_*[i] = grad[boundary frame][pixels_*[i]] − sent_*[pixels_*[i]]   (what backward added since, at the touched pixels)
 *   fm_halo_add      boundary frame += dense_*                                            (the neighbour's dense part)
 *   fm_halo_scatter  boundary frame[pixels_*[i]] += values_*[i]                           (the neighbour's sparse part; pixels distinct)
 * pixels: int64 offsets inside one frame.  One launch each, both boundaries at once. */
int fm_halo_copy(const float* grad, long frame_elements, int frames, float* sent_first, float* sent_last, void* stream);
/* The ghost halo (round 4; FrameShard.enable_ghost_halo): the neighbour's DENSE part of a shared frame's dL/ddepth is not received but
 * evaluated — it is one direction of the flow loss of one pair with the shared frame as its source (the backward term of the pair
 * before a rank's first frame, the forward term of the pair after its last; loss_flow.py:46-68, projection.py:143-184 for one frame):
 *   grad_* (H,W) += norm[0]·upstream[0]·d/ddepth [ ρ(project(pose_*·unproject(depth_*)) − grid, flow_*)·mask_* ]
 * with pose_* (4,4) the source camera -> destination camera transform the neighbour's Procrustes fit produced this step (the only
 * per-step message: 64 bytes), flow_* (H,W,2) / mask_* (H,W) that pair's constant flow and mask in this direction (handed over once),
 * kinv (3,3) of the shared frame and k_dst (3,3) of the destination camera (intrinsics shared by all frames).  Same arithmetic per
 * pixel as fm_flow_loss_fused.  A NULL grad_* switches the side off; upstream NULL = 1. */
int fm_flow_ghost_terms(const float* depth_first, const float* pose_first, const float* flow_first, const float* mask_first, float* grad_first,
                        const float* depth_last, const float* pose_last, const float* flow_last, const float* mask_last, float* grad_last,
                        const float* kinv, const float* k_dst, const float* norm, const float* upstream, int height, int width,
                        int mapping_kind, float delta, float aspect_x, float aspect_y, void* stream);
int fm_halo_delta(const float* grad, long frame_elements, int frames, const float* sent_first, const int64_t* pixels_first, long count_first,
                  float* out_first, const float* sent_last, const int64_t* pixels_last, long count_last, float* out_last, void* stream);
/* The ghost halo's forward end and its sparse round against a COMPACT baseline (round 4): with the ghost halo no frame travels, so the
 * boundary frames need not be copied whole (fm_halo_copy) only to be subtracted at a few thousand pixels later.
 *   fm_halo_ghost_begin   base_*[i] = grad[boundary frame][pixels_*[i]], and — the same launch — pack (82 floats) =
 *                         [t_fwd[0] | t_bwd[pairs−1] | t_bwd[0] | t_fwd[pairs−1] | k (3,3) | kinv (3,3)]: the boundary pairs' poses (4,4 each; what
 *                         FrameShard sends to its neighbours / the one-GPU proxy evaluates its own ghost terms from), K and K⁻¹ of the first
 *                         frame, copied into storage that outlives the step (a step replayed as hipGraphs keeps its own tensors in the
 *                         graphs' pool).  t_fwd / t_bwd: (pairs,4,4).  pack NULL: the baseline only.
 *   fm_halo_delta_sparse  out_*[i] = grad[boundary frame][pixels_*[i]] − base_*[i]
 * pixels as in fm_halo_delta; a NULL base_* / out_* switches the side off. */
int fm_halo_ghost_begin(const float* grad, long frame_elements, int frames, const int64_t* pixels_first, long count_first, float* base_first,
                        const int64_t* pixels_last, long count_last, float* base_last, const float* t_fwd, const float* t_bwd, int pairs,
                        const float* k, const float* kinv, float* pack, void* stream);
int fm_halo_delta_sparse(const float* grad, long frame_elements, int frames, const float* base_first, const int64_t* pixels_first, long count_first,
                         float* out_first, const float* base_last, const int64_t* pixels_last, long count_last, float* out_last, void* stream);
int fm_halo_add(float* grad, long frame_elements, int frames, const float* dense_first, const float* dense_last, void* stream);
int fm_halo_scatter(float* grad, long frame_elements, int frames, const int64_t* pixels_first, const float* values_first, long count_first,
                    const int64_t* pixels_last, const float* values_last, long count_last, void* stream);

/* The tail of IntrinsicsSoftmin.forward (flowmap/model/intrinsics/intrinsics_softmin.py:123-141):
 * soft = softmin((err - min err) * 10) over the N candidates (fp32), K = sum_n soft[n] * candidate_k[n],
 * repeated over `frames`.  err (B,N) fp64 as fm_softmin_score_fwd leaves it; candidate_k (N,3,3);
 * out: soft (B,N), k (B,frames,3,3), kinv (optional, same shape: K^-1 as fm_intrinsics_inverse). */
int fm_softmin_blend_fwd(const double* err, const float* candidate_k, int batch, int candidates, int frames, float* soft, float* k,
                         float* kinv, void* stream);
/* Its backward: grad_k (B,frames,3,3) -> grad_err (B,N) fp32 (feeds fm_softmin_score_bwd). */
int fm_softmin_blend_bwd(const float* grad_k, const float* soft, const float* candidate_k, int batch, int candidates, int frames,
                         float* grad_err, void* stream);

/* IntrinsicsSoftmin's per-candidate score (intrinsics_softmin.py:105-121), straight from the
 * images: depth (B,2,H,W) frames 0/1 (the later frame 1 is un-projected), weights (B,H,W) of pair
 * 0 (logits when weight_sensitivity != 0), bwd_flow (B,H,W,2) of pair 0, indices (P) DISTINCT
 * sampled pixels, k / kinv (N,3,3) candidates, rel (B·N,4,4) fitted poses later -> earlier.
 *   err (B·N) fp64 out (zeroed here): Σ_j |w_j (flow_x − gt_x)| + |w_j (flow_y − gt_y)|. */
int fm_softmin_score_fwd(const float* depth, const float* weights, float weight_sensitivity, const float* bwd_flow,
                         const int64_t* indices, long points, const float* k, const float* kinv, const float* rel, int batch,
                         int candidates, int height, int width, double* err, void* stream);

/* Backward: g_err (B·N) fp32.  g_depth (B,2,H,W) and g_weights (B,H,W): ADDED at the sampled
 * pixels of frame 1 / pair 0, one add per group of candidates (caller zeroes the buffers; either may be NULL); g_rel (B·N,4,4)
 * out, bottom rows 0; g_rel_acc (B·N,12) fp64 workspace (zeroed here). */
int fm_softmin_score_bwd(const float* depth, const float* weights, float weight_sensitivity, const float* bwd_flow,
                         const int64_t* indices, long points, const float* k, const float* kinv, const float* rel, int batch,
                         int candidates, int height, int width, const float* g_err, float* g_depth, float* g_weights,
                         double* g_rel_acc, float* g_rel, void* stream);

/* `count` distinct pseudo-random indices of [0, n) in pseudo-random order — the role of
 * torch.randperm(n)[:count] in IntrinsicsSoftmin (intrinsics_softmin.py:90) without sorting n
 * keys: out[i] = π_seed(i) for a keyed Feistel permutation π of [0, n). */
int fm_random_subset(unsigned long long seed, long n, long count, int64_t* out, void* stream);

/* As fm_random_subset with the seed in device memory: uses state[0], then advances it
 * (splitmix64).  Capturable in a hipGraph: every replay draws a fresh subset. */
int fm_random_subset_stateful(unsigned long long* state, long n, long count, int64_t* out, void* stream);

/* ---- export ------------------------------------------------------
 * Point cloud of export_to_colmap (flowmap/export/colmap.py:86-101): depth (F,H,W), kinv
 * (F,3,3), ext (F,4,4) camera-to-world, colors (F,3,H,W) or NULL -> out_xyz (F·H·W,3) world
 * points, out_rgb (F·H·W,3). */
int fm_world_points(const float* depth, const float* kinv, const float* ext, const float* colors, int frames, int height, int width,
                    float* out_xyz, float* out_rgb, void* stream);

/* ---- flow post-processing ---------------------------------------
 * FlowPredictor.compute_consistency_mask (flowmap/flow/flow_predictor.py:60-80):
 * videos (B,F,3,H,W), flow (B,F-1,H,W,2) in normalised image units -> mask (B,F-1,H,W) =
 * (1 - max_c |frame_i - bilinear(frame_{i+1}, xy + flow)|)^8, zeros padding. */
int fm_consistency_mask(const float* videos, const float* flow, int batch, int frames, int height, int width, float* mask,
                        void* stream);

/* Everything FlowPredictor.compute_bidirectional_flow (:82-102) does after the flow network,
 * for one temporal direction: consistency mask at (height,width), then rescale_flow /
 * rescale_mask (:39-57, bilinear, align_corners=False) to (out_height,out_width).
 *   reverse = 0: flow[pair] maps frame pair -> pair+1; out_*[pair] is that pair.
 *   reverse = 1: flow was predicted on the time-flipped video (videos.flip(1)), so
 *                flow[j] maps frame F-1-j -> F-2-j; out_*[pair] = result for raw index
 *                F-2-pair, i.e. already flipped back (:99-100).  `videos` is NOT flipped.
 * out_flow (B,F-1,oh,ow,2), out_mask (B,F-1,oh,ow). */
int fm_flow_postprocess(const float* videos, const float* flow, int batch, int frames, int height, int width, int out_height,
                        int out_width, int reverse, float* out_flow, float* out_mask, void* stream);

/* resize_batch followed by center_crop_images (flowmap/misc/cropping.py:19-51, the body of
 * crop_and_resize_batch_for_model / _for_flow) in one pass over `planes` images (H,W):
 * out[p][y][x] = F.interpolate(in[p], (resized_height, resized_width), bilinear,
 * align_corners=False)[y + row0][x + col0], out (planes, out_height, out_width). */
int fm_resize_crop(const float* in, long planes, int height, int width, int resized_height, int resized_width, int row0, int col0,
                   int out_height, int out_width, float* out, void* stream);

/* x[0..count) = 0 with at most `blocks` workgroups (16-byte non-temporal stores, x 16-byte aligned):
 * the zero fill of a dense gradient buffer that only a few entries will be written into
 * (dL/dweights of the sparse Procrustes fit, projection.py:226-249), sized so that it can run beside
 * latency-bound kernels on another stream. */
int fm_fill_zero(float* x, long count, int blocks, void* stream);

/* ---- optimiser step ------------------------------------------
 * torch.optim.Adam as configured by ModelWrapperOverfit.configure_optimizers
 * (flowmap/model/model_wrapper_overfit.py:104-105), one tensor per call, in place:
 *   g += weight_decay·p; m = m + (1-β1)(g-m); v = β2·v + (1-β2)g²;
 *   p -= lr/(1-β1^step) · m / (sqrt(v)/sqrt(1-β2^step) + eps)
 * `step` is the 1-based step number AFTER the increment.  No amsgrad / maximize. */
int fm_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, long step, double lr, double beta1,
                 double beta2, double eps, double weight_decay, void* stream);

/* As fm_adam_step with the step number read from device memory (step[0], a float holding the
 * 1-based step AFTER the increment): capturable in a hipGraph, like torch.optim.Adam(capturable=True). */
int fm_adam_step_capturable(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long count, const float* step,
                            double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);
/* The same update on a LIST of elements (`elements` (count) int64 flat indices): the touched pixels of
 * fm_flow_loss_fused_adam. */
int fm_adam_step_elements(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* elements, long count, long step,
                          double lr, double beta1, double beta2, double eps, double weight_decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FLOWMAP_HIP_H */
