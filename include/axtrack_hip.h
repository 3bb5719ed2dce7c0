/*
 * axtrack_hip.h -- C ABI of libaxtrack_hip.so: AxTrack's detect + associate hot path on MI355X (gfx950).
 *
 * The reference (LoaloaF/axtrack) is pure Python; it has no FFI of its own. The seams this
 * library replaces are the Python call sites listed per function below (paths relative to the
 * reference repo). INTEGRATION.md shows the ctypes binding a maintainer of the reference adds.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes. Pointers prefixed d_ are DEVICE pointers (HIP),
 *     h_ are HOST pointers. `stream` is a hipStream_t passed as void* (NULL = default stream).
 *   - Every function returns 0 on success, a negative AXT_E* code on error and never throws;
 *     axt_last_error() returns a human-readable message for the calling thread.
 *   - No hidden global state: all device state lives in an axt_detector handle or in
 *     caller-provided buffers; calls are asynchronous on `stream` unless stated otherwise.
 *   - Detections of a frame are three parallel arrays (conf f32, x i32, y i32) of capacity
 *     `cap` per frame, sorted by descending confidence, plus a per-frame count.
 */
#ifndef AXTRACK_HIP_H
#define AXTRACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AXT_OK 0
#define AXT_EINVAL (-22)   /* bad argument (shape, null pointer, capacity)          */
#define AXT_ENOMEM (-12)   /* device or host allocation failed                       */
#define AXT_EHIP (-5)      /* a HIP runtime call failed, see axt_last_error()        */
#define AXT_ERUNTIME (-71)  /* an internal consistency check failed (a bug: please report) */
#define AXT_INFEASIBLE 1   /* axt_mcf_solve: fewer than min_flow unit flows exist    */

#define AXT_TILE 512       /* TILESIZE (deployed_model/params.txt:33)                */
#define AXT_S 12           /* SX = SY  (deployed_model/params.txt:34-35)             */
#define AXT_CELLS (AXT_S * AXT_S)
#define AXT_YOLO_FLOATS (AXT_CELLS * 3)
#define AXT_IN_CH 5        /* 1 * (2*TEMPORAL_CONTEXT+1), core_functionality.py:66   */
#define AXT_N_WEIGHT_TENSORS 54   /* 8 conv blocks x 6 + 3 linear x 2                */

const char *axt_last_error(void);
int axt_abi_version(void);

/* ------------------------------------------------------------------------------------------
 * Detector: replaces YOLO_AXTrack (axtrack/machinelearning/model.py:20-125) in eval mode.
 * ------------------------------------------------------------------------------------------ */
typedef struct axt_detector axt_detector;

/* Builds the device-resident detector from the tensors of the reference's state_dict
 * (checkpoint layout: utils.py:258-272; key set: model.py:85-117).
 * h_tensors[54], host f32, in this order:
 *   for blk in ConvBlock_{0,1,2,4,5,7,8,10}:
 *     conv.weight[Co,Ci,3,3], conv.bias[Co], batchnorm.weight[Co], batchnorm.bias[Co],
 *     batchnorm.running_mean[Co], batchnorm.running_var[Co]
 *   then fcs.1.weight[1024,40960], fcs.1.bias, fcs.3.weight[1024,1024], fcs.3.bias,
 *        fcs.5.weight[432,1024], fcs.5.bias
 * BatchNorm (eps 1e-5) is folded into the convolution in f64 and rounded once to f32.
 * max_batch = largest number of tile-forwards one call will be asked for (sizes the workspace).
 * Synchronous (returns after the upload has completed). */
int axt_detector_create(const float *const *h_tensors, int n_tensors, int max_batch,
                        axt_detector **out);
void axt_detector_destroy(axt_detector *det);

/* Arithmetic of the stride-1 conv blocks (blocks 2, 4, 5, 7, 8, 10 of model.py:85-103; 73 % of the forward pass's FLOPs;
 * mode 1 leaves block 10 on the direct kernel); parameters['CNN_ARITH'].
 * mode 2 (default after axt_detector_create, "f32" / "f32_winograd"): Winograd F(2x2,3x3) on the f32 matrix pipe -- every
 *   operation f32 (transforms are additions, G g G^T formed in f64 and rounded once), 16/36 of the direct multiplications;
 *   as close to an f64 convolution as the direct kernel, like the algorithms cuDNN / oneDNN choose for
 *   the reference's own Conv2d.
 * mode 0 ("f32_direct"): direct convolution, f32-in / f32-accumulate MFMA, a k-ordered chain of f32 FMAs per output.
 * mode 1 ("bf16x3", opt-in): every f32 operand split exactly into three bf16 terms, six partial products per multiply on
 *   the bf16 matrix pipe, f32 accumulation -- per-product error of the size of an f32 rounding.
 * The modes agree to ~1e-6 on the grids and are not bit-identical to each other. Packed weights of modes 1 and 2 are built
 * on the first switch to them. Synchronous. */
int axt_detector_set_arith(axt_detector *det, int mode);
/* The two stride-2 conv blocks (0 and 1) of the detector: fused = 1 (the default at create) runs them as ONE kernel that
 * keeps block 0's output in LDS (conv_s2_fused, cnn_front.hip) -- used whenever the frame width is a multiple of 4;
 * fused = 0 runs the two separate kernels (conv3x3_s2_k1), which other widths take in either setting. The fused kernel
 * sums block 1's products in another order: the grids of the two settings agree to f32 rounding (~1e-6), not bit for bit.
 * Takes effect from the next forward pass. (axt_detector_create starts with fused = 1 unless the environment holds
 * AXT_FUSE_S2=0 -- for A/B runs of unmodified callers.) */
int axt_detector_set_fused_front(axt_detector *det, int fused);
/* bytes of device memory held by the handle (packed weights + activation workspace) */
size_t axt_detector_device_bytes(const axt_detector *det);

/* detect_axons (model.py:119-125; call site AxonDetections.py:118):
 * d_x f32 [B,5,512,512] NCHW -> d_yolo f32 [B,12,12,3] (dim1 = x cell, dim2 = y cell,
 * last = conf, x_in_cell, y_in_cell). B <= max_batch. */
int axt_cnn_forward(axt_detector *det, const float *d_x, int B, float *d_yolo, void *stream);

/* The same forward pass fed straight from the timelapse, fusing
 * Timelapse.get_frametiles_stack (Timelapse.py:111-125,150-157) into the first convolution:
 * d_frames f32 [T_all,H,W] is channel 0 of the context-padded sequence; detection frame t of
 * tile k reads frames t..t+4 at tile origin (tile_yx[2k]*512, tile_yx[2k+1]*512), zero beyond
 * H/W (ZeroPad2d at Timelapse.py:529-533). Output d_yolo f32 [n_frames, n_tiles, 12,12,3] for
 * detection frames t0 .. t0+n_frames-1. n_frames*n_tiles may exceed max_batch (chunked). */
int axt_cnn_forward_frames(axt_detector *det, const float *d_frames, int T_all, int H, int W,
                           int t0, int n_frames, const int32_t *h_tile_yx, int n_tiles,
                           float *d_yolo, void *stream);

/* The same forward pass in two calls, for input that arrives in chunks (Timelapse.from_host_u16: the reference's inference()
 * starts from a host Timelapse and construct_tiles() moves it to the device, Timelapse.py:492-566): axt_cnn_front_frames runs
 * conv blocks 0-5 (through the second max-pool) for the (frame, tile) items of frames t0 .. t0+n_frames-1 and leaves their
 * features in the detector's batch buffer from item `item0` on; after the last chunk axt_cnn_back runs the remaining conv
 * blocks and the three linear layers ONCE for items 0 .. n_items-1 (the first linear layer streams its 168 MB of weights per
 * call, whatever the batch) and writes d_yolo [n_items,12,12,3]. item0 + items <= max_batch. Same results, bit for bit, as
 * axt_cnn_forward_frames over the same items. Asynchronous. */
int axt_cnn_front_frames(axt_detector *det, const float *d_frames, int T_all, int H, int W, int t0, int n_frames,
                         const int32_t *h_tile_yx, int n_tiles, int item0, void *stream);
int axt_cnn_back(axt_detector *det, int n_items, float *d_yolo, void *stream);

/* FLOPs of one tile-forward (algorithmic: 2*M*N*K of the unpadded layers). */
double axt_cnn_flops_per_tile(void);

/* Per-kernel timing for the roofline report (bench.py): on = 1, every kernel launch of the
 * forward pass is bracketed by HIP events on the launch stream; on = 2 + k, only the launches of
 * kernel id k are (two events per launch cost ~10 us of launch gap: bracketing all 19 launches of a
 * pass slows it by ~8 %, bracketing the dominant kernel alone by < 1 %); on = 0, none. Kernel ids: 0..7 the eight conv
 * blocks, 8/10/12 the linear-layer GEMMs, 9/11/13 their split-K reductions.
 * axt_detector_read_profile synchronises on the recorded events and returns, per kernel id, the
 * summed milliseconds, the number of launches and the number of tile-forwards since the last read. */
#define AXT_N_CNN_KERNELS 14
int axt_detector_set_profiling(axt_detector *det, int on);
int axt_detector_read_profile(axt_detector *det, double *ms, int64_t *launches, int64_t *items, int n);
double axt_cnn_kernel_flops_per_tile(int kernel);

/* ------------------------------------------------------------------------------------------
 * Preprocessing ("next" row of the scope table, in front of the hot path): the dense part of
 * Timelapse._read_tiff, _clip_image_values, _log_adjust_image, _standardize
 * (Timelapse.py:205-326) as one fused pass: x = u16/65535 -> mask -> max(x-offset,0) ->
 * (x<clip_lower ? 0 : x) -> log2(1+x) -> x/scale. d_raw u16 [T,H,W]; d_mask u8 [H,W] or NULL;
 * offset / clip_lower in [0,1] units (0 = skip); d_out f32 [T,H,W].
 * ------------------------------------------------------------------------------------------ */
int axt_preprocess_u16(const uint16_t *d_raw, const uint8_t *d_mask, int T, int H, int W, float offset,
                       float clip_lower, int log_correct, float scale, float *d_out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Tiling: which 512x512 tiles the reference keeps (Timelapse.py:551-558): a tile is kept if
 * any pixel of it is > 0 at any t. d_occ u8 [ceil(H/512)*ceil(W/512)] row-major, 1 = keep.
 * ------------------------------------------------------------------------------------------ */
int axt_tile_occupancy(const float *d_frames, int T_all, int H, int W, uint8_t *d_occ, void *stream);

/* ------------------------------------------------------------------------------------------
 * YOLO grid -> frame detections. Replaces, per frame,
 *   _yolo_Y2pandas_det  (AxonDetections.py:178-248)  decode, round-half-even, conf >= thr
 *   stitch_tiles        (Timelapse.py:166-197)       + tile origin
 *   _non_max_supression (AxonDetections.py:250-278)  greedy, dx^2+dy^2 < min_dist^2 dies
 * d_yolo f32 [n_frames, n_tiles, 12,12,3]; h_tile_yx i32 [n_tiles,2] (tile row, tile col).
 * Outputs, capacity cap >= n_tiles*144 per frame: d_conf f32, d_x i32, d_y i32 [n_frames,cap]
 * in descending confidence (ties: tile order, then cell order), d_count i32 [n_frames].
 * n_tiles <= 256; frames of up to 28 kept tiles keep their candidates in LDS, larger ones in a workspace in HBM
 * (allocated on the stream for the call): same order, same result.
 * ------------------------------------------------------------------------------------------ */
int axt_decode_stitch_nms(const float *d_yolo, int n_frames, int n_tiles, const int32_t *h_tile_yx,
                          float conf_thr, int min_dist, int cap,
                          float *d_conf, int32_t *d_x, int32_t *d_y, int32_t *d_count, void *stream);

/* ------------------------------------------------------------------------------------------
 * Association costs.
 * ------------------------------------------------------------------------------------------ */
/* observation_model (mincostflow_models.py:6-27) after the confidence capping of
 * AxonDetections.py:655-659. method 0 = 'scale_to_max' (conf / max over all frames),
 * 1 = 'ceil' (min(conf,1)). d_cost f64 [n_frames,cap] (entries >= count are untouched). */
int axt_obs_costs(const float *d_conf, const int32_t *d_count, int n_frames, int cap,
                  int method, double max_conf_cost, double *d_cost, void *stream);

/* A masked grid: the reference's weight image {1 on mask, 65536 off} (AxonDetections.py:587-598) in the forms the
 * kernels need -- byte mask, bit-packed rows and connected-component labels (4- or 8-connected, flood fill on the
 * host). Built once per timelapse from a HOST mask u8 [H,W] (1 = on mask). Synchronous. */
typedef struct axt_grid axt_grid;
int axt_grid_create(const uint8_t *h_mask, int H, int W, int conn8, axt_grid **out);
void axt_grid_destroy(axt_grid *grid);

/* _compute_detections_astar_paths + _get_astar_path_distances for ONE frame pair
 * (AxonDetections.py:526-629,717-752; A* call utils.py:379): D[i,j] = number of cells on the
 * minimum-cost 4-connected (conn8: 8-connected) path from detection i of the earlier frame
 * (rows) to detection j of the later frame on weights {1 on mask, 65536 off}; max_dist where
 * the euclidean distance is >= max_dist, an end point is outside the grid, or the path has
 * more than max_dist cells. grid = NULL for an all-ones mask (closed form); with a grid its
 * connectivity must equal conn8 (exact single-source search per detection of the earlier frame).
 * D i32 [na, nb] row-major. */
int axt_path_cost(const int32_t *d_xa, const int32_t *d_ya, int na,
                  const int32_t *d_xb, const int32_t *d_yb, int nb,
                  const axt_grid *grid, int H, int W, int max_dist, int conn8,
                  int32_t *d_D, void *stream);

/* The paths themselves (the coo matrices _compute_astar_path returns, utils.py:379-387; kept by
 * _compute_detections_astar_paths, AxonDetections.py:570-577, for the cache file and for drawing): D as above, and
 * d_cells i32 [na, nb, max_dist] receives, for every pair with D < max_dist, the D cells of one minimum-cost path
 * as y*W + x, source first; other entries are left untouched. Needs a grid (all-ones masks have closed-form
 * staircases); which of several equally cheap paths pyastar2d would return is not pinned. */
int axt_path_cells(const int32_t *d_xa, const int32_t *d_ya, int na,
                   const int32_t *d_xb, const int32_t *d_yb, int nb,
                   const axt_grid *grid, int H, int W, int max_dist, int conn8,
                   int32_t *d_D, int32_t *d_cells, void *stream);

/* All admissible transition arcs of a timelapse in one pass (what transition_model,
 * mincostflow_models.py:67-119, and the tracker's cost_threshold keep): for every detection
 * a of frame t and b of frame t+g (g = 1..max_gap) with D(a,b) <= h_dmax[g-1], one arc a->b.
 * Detections are numbered globally in frame order (frame offsets = prefix sum of counts).
 * CSR by tail: d_row_ptr i64 [n_det+1] (allocate n_frames*cap+1); d_col i32 [n_arcs] (global
 * index of b), d_len i16 [n_arcs] (D), d_gap u8 [n_arcs]. Rows are sorted by (gap, b):
 * deterministic. d_work i32 [n_frames*cap*max_gap + n_frames + 1 + max_gap + 4] (plus, with a masked grid,
 * n_frames*cap*max_gap*cap/2 for the i16 path-length table) is scratch that
 * must stay untouched between the two phases:
 *   phase 1: d_col == NULL -> counts, row_ptr, *n_arcs (synchronises the stream);
 *   phase 2: d_col/d_len/d_gap sized by *n_arcs -> filled (asynchronous).
 * Optional integer arc costs: d_cost_units i64 [max_gap, max_dist+1] holds round(cost*1e6) of
 * transition_model for every (gap, D); d_cost i64 [n_arcs] then receives
 * axt_arc_cost_int-compatible values (units << 16 | hash16(3, a, b)). Both NULL to skip.
 * grid = NULL: all-ones mask, closed-form path lengths. With a grid the minimum-cost paths are ordered by (off-mask
 * cells entered, moves); the grid holds, per connected component of the mask, the fewest off-mask cells to every cell.
 * A detection on the mask gets all its path lengths from one bit-parallel breadth-first search (depth h_dmax-1 <= 250
 * moves) over the steps that keep that count minimal; a detection off the mask from a label-correcting search on
 * (off-mask cells, moves) inside the window paths of <= dmax cells cannot leave, checked against the same counts
 * (masks with more than 64 components: on-mask search within the component + the exact search of axt_path_cost). */
int axt_build_arcs(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                   const axt_grid *grid, int H, int W, int max_dist, int conn8,
                   int max_gap, const int32_t *h_dmax,
                   int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len, uint8_t *d_gap,
                   const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs, void *stream);

/* axt_build_arcs with the path lengths supplied by the caller instead of computed: d_len_table i16
 * [n_frames, cap, max_gap, cap], entry [t][i][g-1][j] = number of cells of the path between detection i of
 * frame t and detection j of frame t+g, <= 0 = none. This is how path lengths cached by the reference
 * ('{name}_astar_dets_paths.pkl': coo matrices, length = getnnz(), AxonDetections.py:717-752) re-enter the
 * pipeline. d_x / d_y are not read for lengths but must be valid [n_frames, cap] arrays. */
int axt_build_arcs_from_lengths(const int16_t *d_len_table, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                                int n_frames, int cap, int max_dist, int max_gap, const int32_t *h_dmax,
                                int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len, uint8_t *d_gap,
                                const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs, void *stream);

/* ------------------------------------------------------------------------------------------
 * Appearance term of the transition cost, MCF_VIS_SIM_WEIGHT > 0.
 *
 * axt_box_histograms = feature_model (axtrack/mincostflow_models.py:30-65): for every detection the
 * 180-bin histogram on [0,1) of the box x box crop of frame (f + t_offset) of d_frames -- the centre frame
 * of detection frame f, which is what the tracker is shown (AxonDetections.py:682-685) -- min-max
 * normalised. The crop starts at (max(y - box/2, 0), max(x - box/2, 0)) and is clipped at the far edges.
 * d_hist f32 [n_frames, cap, 180], d_hist_sum f64 [n_frames, cap] (bin sums, used by the distance).
 *
 * axt_build_arcs_vis = axt_build_arcs with the cost of transition_model (:100-118) computed per pair:
 *   cost = -log((1-w) * (1 - D/max_dist) * miss_rate^(gap-1) + w * (1 - d_B(hist_a, hist_b)) + 1e-6)
 * d_B = cv2.compareHist(.., HISTCMP_BHATTACHARYYA); an arc is kept iff cost < edge_cost_thr, and
 * d_cost (may be NULL) receives round(cost * 1e6) << 16 | hash16(3, a, b). h_dmax[g-1] only bounds the
 * candidates: pass the largest D whose cost with similarity 1 is still below the threshold. Same two
 * phases and scratch as axt_build_arcs. cv2 is absent from the reference tree: parity unpinned.
 * ------------------------------------------------------------------------------------------ */
int axt_box_histograms(const float *d_frames, int T_all, int H, int W, int t_offset, const int32_t *d_x,
                       const int32_t *d_y, const int32_t *d_count, int n_frames, int cap, int box,
                       float *d_hist, double *d_hist_sum, void *stream);
int axt_build_arcs_vis(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                       const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                       const float *d_hist, const double *d_hist_sum, double vis_weight, double miss_rate,
                       double edge_cost_thr, int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len,
                       uint8_t *d_gap, int64_t *d_cost, int64_t *n_arcs, void *stream);

/* The same arc builder (_compute_detections_astar_paths + transition_model + the tracker's edge admission,
 * AxonDetections.py:526-629,663-690, mincostflow_models.py:67-119) for a frame-sharded run (one process per GPU after
 * the all-gather of the detections): rows are built only for the first d_src_count[t] detections of every frame t -- a rank passes
 * its own frames' counts and zeros elsewhere -- while targets, detection numbering and integer costs are those of the
 * whole timelapse (d_count), so that the ranks' arc lists, concatenated in rank order, are exactly the single-process
 * list. d_hist == NULL: costs from d_cost_units as in axt_build_arcs; otherwise the appearance term as in
 * axt_build_arcs_vis (d_cost_units ignored). row_ptr covers all detections; rows outside the source frames are empty.
 * This is the general entry point: d_src_count == NULL builds every row; d_len_table != NULL (i16
 * [n_frames, cap, max_gap, cap], as for axt_build_arcs_from_lengths) takes the path lengths from a cache instead of
 * computing them (grid ignored) -- the combination the reference's parameter search runs with the appearance term
 * (assign_ids(astar_paths_cache='from') with MCF_VIS_SIM_WEIGHT > 0, AxonDetections.py:882,911-912). */
int axt_build_arcs_rows(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, const int32_t *d_src_count,
                        int n_frames, int cap, const axt_grid *grid, int H, int W, int max_dist, int conn8,
                        int max_gap, const int32_t *h_dmax,
                        const float *d_hist, const double *d_hist_sum, double vis_weight, double miss_rate,
                        double edge_cost_thr,
                        int64_t *d_row_ptr, int32_t *d_work, int32_t *d_col, int16_t *d_len, uint8_t *d_gap,
                        const int64_t *d_cost_units, int64_t *d_cost, int64_t *n_arcs,
                        const int16_t *d_len_table, void *stream);

/* ------------------------------------------------------------------------------------------
 * Global data association. Replaces libmot's MinCostFlowTracker as driven by
 * AxonDetections.py:663-690 (process() per frame, then compute_trajectories()).
 * HOST function (exact successive-shortest-path min-cost flow on the unit-capacity tracking
 * network; the solve is serial by nature -- DESIGN.md).
 *   n_det detections numbered globally; per detection: integer observation cost;
 *   entry/exit arcs cost h_entry[k], h_exit[k]; transition arcs in CSR (row_ptr, col, cost).
 *   min_flow/max_flow: bounds on the number of trajectories (MCF_MIN_FLOW / MCF_MAX_FLOW).
 * Outputs: h_next i32 [n_det]: successor detection on the same trajectory or -1;
 *          h_track i32 [n_det]: trajectory id (numbered by first detection) or -1;
 *          *n_tracks, *total_cost.
 * Returns AXT_INFEASIBLE if fewer than min_flow trajectories can be routed (the reference's
 * falsy compute_trajectories(), AxonDetections.py:691-696).
 * ------------------------------------------------------------------------------------------ */
int axt_mcf_solve(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                  const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost,
                  int min_flow, int max_flow,
                  int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost);

/* axt_mcf_solve plus an optimality certificate (what a caller of the reference's tracker cannot get: libmot returns
 * trajectories only, AxonDetections.py:690). Node potentials of the optimum, pi(S) = 0: h_pot_u[k] = pi(u_k), h_pot_v[k] = pi(v_k)
 * (i64 [n_det] each), *pot_t = pi(T). With the reduced cost rc(x -> y) = cost + pi(x) - pi(y) of the arcs S -> u_k (entry),
 * u_k -> v_k (observation), v_k -> T (exit), v_a -> u_b (transition) and T -> S (cost 0, between min_flow and max_flow units):
 * rc >= 0 on every arc without flow, rc <= 0 on every arc that carries its unit, and for T -> S: pi(T) >= 0 unless the flow
 * count sits at max_flow, pi(T) <= 0 unless it sits at min_flow. Together with a feasible flow (node-disjoint paths) these are
 * the complementary-slackness conditions of the flow LP: checking them over all arcs -- O(arcs), no second solve -- proves that
 * the returned trajectories are a minimum (tests/helpers.py: check_flow_certificate). Not written when AXT_INFEASIBLE. */
int axt_mcf_solve_duals(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                        const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost,
                        int min_flow, int max_flow,
                        int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost,
                        int64_t *h_pot_u, int64_t *h_pot_v, int64_t *pot_t);

/* The same solve shared between frame-sharded ranks (one process per GPU; AxonDetections.py:663-690 has one process). Every
 * rank holds the whole network (the detections are all-gathered, the arcs rebuilt or gathered: DESIGN.md section 7) and calls
 *   axt_mcf_shard_begin   -- sets the solver up and solves THIS rank's run of time blocks (1/world of the leaves of the
 *                            time-block tree, on the rank's host threads); *state_bytes = size of the state to exchange;
 *   axt_mcf_shard_export  -- writes that state (duals and matching of the rank's rows and columns) to h_state;
 *   [one all-gather of the states between the ranks: torch.distributed, RCCL or gloo -- not this library's business]
 *   axt_mcf_shard_finish  -- h_states[r] / h_state_bytes[r] = rank r's state (the own entry is ignored): joins the runs
 *                            through the separator rows between them, runs the second phase, and returns exactly what
 *                            axt_mcf_solve returns (the optimum is unique: every rank ends with the same trajectories);
 *   axt_mcf_shard_free.
 * world must be a power of two. The network arrays must stay valid from begin to finish. Host only. */
typedef struct axt_mcf_shard axt_mcf_shard;
int axt_mcf_shard_begin(int n_det, const int64_t *h_obs, const int64_t *h_entry, const int64_t *h_exit,
                        const int64_t *h_row_ptr, const int32_t *h_col, const int64_t *h_cost, int rank, int world,
                        axt_mcf_shard **out, int64_t *state_bytes);
int axt_mcf_shard_export(const axt_mcf_shard *shard, void *h_state);
int axt_mcf_shard_finish(axt_mcf_shard *shard, const void *const *h_states, const int64_t *h_state_bytes, int min_flow,
                         int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost);
/* axt_mcf_shard_finish with the certificate of axt_mcf_solve_duals */
int axt_mcf_shard_finish_duals(axt_mcf_shard *shard, const void *const *h_states, const int64_t *h_state_bytes, int min_flow,
                               int max_flow, int32_t *h_next, int32_t *h_track, int *n_tracks, int64_t *total_cost,
                               int64_t *h_pot_u, int64_t *h_pot_v, int64_t *pot_t);
void axt_mcf_shard_free(axt_mcf_shard *shard);

/* ------------------------------------------------------------------------------------------
 * Frame-to-frame Hungarian association (BASELINE config 3; a build-side variant -- the reference
 * only runs the global tracker above, AxonDetections.py:663-690). Same cost model: linking a (frame t)
 * to b (frame t+g) costs transition_model(D(a,b), g) (mincostflow_models.py:67-119) and is admitted for D <= h_dmax[g-1]; leaving a detection
 * without successor costs thr_units (= round(MCF_EDGE_COST_THR * 1e6)). Every frame pair
 * (t,t+1) is solved exactly and independently (one wavefront each), then (t,t+2) among the
 * detections left unlinked, then chains are numbered by (first frame, index).
 * d_cost_units i64 [max_gap, max_dist+1]; d_work i32 [4*n_frames*cap + n_frames + 1] scratch;
 * d_track i32 [n_frames, cap]: trajectory id of every detection slot (-1 beyond count);
 * d_n_tracks i32 [1]. All-ones mask only (closed-form path lengths). Asynchronous.
 * ------------------------------------------------------------------------------------------ */
int axt_hungarian_assoc(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                        int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                        const int64_t *d_cost_units, int64_t thr_units, int32_t *d_work,
                        int32_t *d_track, int32_t *d_n_tracks, void *stream);

/* The two halves of axt_hungarian_assoc (same cost model, mincostflow_models.py:67-119; stands where the reference
 * runs its tracker, AxonDetections.py:663-690), for frame-sharded runs: every rank solves the frame pairs of its own
 * source frames [t_begin, t_end) into d_pred = pred1 | pred2 (i32 [2, n_frames*cap], predecessor index in frame
 * t-1 / t-2 or -1; entries of other ranks' frames stay -1), the ranks combine d_pred with one element-wise MAX
 * all-reduce, then each numbers the chains. d_work: i32 [2*n_frames*cap + n_frames + 1] for the pairs,
 * i32 [2*n_frames*cap] for the chains. */
int axt_hungarian_pairs(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                        int H, int W, int max_dist, int conn8, int max_gap, const int32_t *h_dmax,
                        const int64_t *d_cost_units, int64_t thr_units, int t_begin, int t_end,
                        int32_t *d_pred, int32_t *d_work, void *stream);
/* axt_hungarian_pairs on a masked grid (grid = NULL: identical to it): the path lengths of the pairs come from the
 * masked-grid searches of the arc builder (_compute_detections_astar_paths on weights {1 on mask, 65536 off},
 * AxonDetections.py:526-629) instead of the closed form. */
int axt_hungarian_pairs_grid(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                             const axt_grid *grid, int H, int W, int max_dist, int conn8, int max_gap,
                             const int32_t *h_dmax, const int64_t *d_cost_units, int64_t thr_units,
                             int t_begin, int t_end, int32_t *d_pred, int32_t *d_work, void *stream);
/* The same two passes with the link costs GIVEN instead of derived from the path lengths: d_ctab i64
 * [n_frames][cap][max_gap][cap], entry (t, i, g-1, j) = the integer cost (axt_arc_cost_int, kind 3, global detection
 * indices) of linking detection i of frame t to detection j of frame t+g, or 0x3fffffffffffffff where the link is not
 * admitted. For cost models the closed form does not cover -- the appearance term of transition_model
 * (mincostflow_models.py:107-113, MCF_VIS_SIM_WEIGHT > 0): the costs of axt_build_arcs_vis scattered into the table.
 * Build-side variant like axt_hungarian_pairs itself (the reference runs only the flow tracker). */
int axt_hungarian_pairs_costs(const int32_t *d_x, const int32_t *d_y, const int32_t *d_count, int n_frames, int cap,
                              int max_gap, const int64_t *d_ctab, int64_t thr_units, int t_begin, int t_end,
                              int32_t *d_pred, int32_t *d_work, void *stream);
int axt_chain_tracks(const int32_t *d_count, int n_frames, int cap, const int32_t *d_pred, int32_t *d_work,
                     int32_t *d_track, int32_t *d_n_tracks, void *stream);

/* ------------------------------------------------------------------------------------------
 * IDed_dets_all (AxonDetections._agg_all_IDed_dets, AxonDetections.py:825-842) as a dense f64
 * table [n_rows, 3*n_frames] on the device: columns 3*slot(f)+{0,1,2} = anchor_x, anchor_y, conf of
 * the detection of frame f that carries the row's identity, NaN elsewhere. slot(f) = f, or with
 * label_quirk != 0 the rank of f among the frames that have an IDed detection (the reference's
 * concat drops the others, :831, and labels the rest by position).
 * d_track i32 [n_frames, cap] (-1 = no identity), d_id_row: NULL (row = id, n_rows == n_ids) or
 * i32 [n_ids] id -> row (-1 = drop); d_work i32 [n_frames] scratch. Asynchronous.
 * ------------------------------------------------------------------------------------------ */
int axt_ided_table(const int32_t *d_track, const float *d_conf, const int32_t *d_x, const int32_t *d_y,
                   const int32_t *d_count, int n_frames, int cap, int n_ids, const int32_t *d_id_row, int n_rows,
                   int label_quirk, int32_t *d_work, double *d_table, void *stream);

/* ------------------------------------------------------------------------------------------
 * Detection metrics: compute_TP_FP_FN (AxonDetections.py:409-466) for every frame
 * and every confidence threshold. Detections as axt_decode_stitch_nms leaves them; labels d_gx, d_gy
 * i32 [n_frames, gcap], d_gcount i32 [n_frames]; d_thrs f64 [n_thr] on the device (the reference's
 * all_conf_thrs, :76); min_dist = NON_MAX_SUPRESSION_DIST. d_confusion i32 [n_frames, 3, n_thr]:
 * TP, FP, FN. Labels are matched in order; a label whose closest candidate is already claimed is a
 * false negative; an empty side is replaced by one phantom row at the origin with conf 0 (:434-437).
 * k_mask >= 0: additionally d_fp_mask u8 [n_frames, cap] / d_fn_mask u8 [n_frames, gcap] for that
 * threshold index (either may be NULL). Asynchronous.
 * ------------------------------------------------------------------------------------------ */
int axt_detection_confusion(const float *d_conf, const int32_t *d_x, const int32_t *d_y, const int32_t *d_count,
                            int n_frames, int cap, const int32_t *d_gx, const int32_t *d_gy, const int32_t *d_gcount,
                            int gcap, const double *d_thrs, int n_thr, int min_dist, int k_mask,
                            int32_t *d_confusion, uint8_t *d_fp_mask, uint8_t *d_fn_mask, void *stream);

/* Integer arc cost used by the flow network (the costs libmot hands its solver at AxonDetections.py:663-690, from
 * observation_model / transition_model, mincostflow_models.py:6-27,67-119): round(cost * 1e6) << 16 | hash16(kind, a, b).
 * kind 0 entry, 1 exit, 2 observation, 3 transition. The low 16 bits make the optimum unique
 * (DESIGN.md "Unpinned third-party semantics"). */
int64_t axt_arc_cost_int(double cost, int kind, int64_t a, int64_t b);

#ifdef __cplusplus
}
#endif
#endif
