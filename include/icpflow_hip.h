/*
 * icpflow_hip.h -- C ABI of libicpflow_hip.so: the MI355X (gfx950) drop-in for
 * ICP-Flow's cluster-pair registration hot path.
 *
 * Every pointer named d_* is a DEVICE pointer owned by the caller (e.g. the
 * data_ptr() of a torch-ROCm tensor); the library never allocates or frees
 * caller memory and keeps NO mutable process-global state: the only state outside
 * the arguments is a thread-local error string, a thread-local helper stream and
 * per-device caches of immutable device properties.  Tuning switches and the
 * launch-timing recorder travel with each call in an icpflow_options_t.
 * Scratch comes from a caller-provided workspace (icpflow_workspace_bytes()).
 * All work is enqueued asynchronously on `stream` (a hipStream_t passed as
 * void*; NULL = the default stream); no entry point synchronises the device.
 *
 * Return value: 0 = ok, negative = argument error (ICPFLOW_E_*), positive = the
 * hipError_t of a failed runtime call / kernel launch.  icpflow_last_error()
 * returns the message of the calling thread's last failure.  (The reference
 * only printf()s launch errors, hist_cuda_core.cuh:94-98; here they surface.)
 *
 * Data contract (reference: utils_helper.py:185-196 `pad_segment`):
 *   clouds are float32 [B, N, 4] contiguous, columns (x, y, z, flag);
 *   a point is valid iff flag > 0.  The NN / ICP / evaluation entry points
 *   additionally require the layout pad_segment produces -- valid rows first,
 *   then pads -- because, like pytorch3d's `lengths`, they scan the first
 *   n = count(flag > 0) rows.  icpflow_hist_vote accepts arbitrary flag patterns
 *   (the reference's own hist_cuda/test.py uses random flags).
 *
 * All citations `file:line` are into the reference repository
 * (yanconglin/ICP-Flow).
 */
#ifndef ICPFLOW_HIP_H
#define ICPFLOW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICPFLOW_VERSION 214 /* 0.2.14: ICPFLOW_OPT_NO_DIR_KEYS (sort keys of the sweeps: horizontal directions next to the axes); 0.2.13: ICPFLOW_OPT_TWO_LAUNCH (ICP of batches of a few rounds: the persistent grid drained for a second launch of whole-CU workgroups; off by default); 0.2.12: ICPFLOW_OPT_NO_SCORE_PREBOUND (scoring sweeps: a scan's whole sum bounded from below by the other cloud's occupancy grid before any target is evaluated); 0.2.11: ICPFLOW_OPT_NO_CHECK_REUSE (hist_icp: the roll-back check takes its sum under the initial pose from the scoring); 0.2.10: ICPFLOW_OPT_NO_VOTE_LIST (the vote's work list on ragged batches); 0.2.9: icpflow_register_stage_begin / _finish, icpflow_associate_frame_begun (stage 2's initial poses beside stage 1's ICP), ICPFLOW_E_HOSTMEM; 0.2.8: ICPFLOW_OPT_NO_SHARED_SCANS (teams: window scans shared by a member's waves); 0.2.7: icpflow_track_frame (one frame pair per call, host half in C++); team-launch chains per device instead of per host thread; 0.2.6: icpflow_register_stage, icpflow_associate_frame (a stage / the rest of match_pcds per call); 0.2.5: options.d_pair_active, icpflow_assoc_assign / _collect (device-side association of a frame pair), ICPFLOW_OPT_TEAMS_HALF_GPU; 0.2.3: icpflow_hist_icp_eval; 0.2.2: icpflow_hist_icp_many; 0.2.1: per-call options replace the process-global switches of 0.1;
                               icpflow_icp takes an initial transform and returns its per-iteration history */

#define ICPFLOW_OK 0
#define ICPFLOW_E_ARG (-1)       /* bad pointer / size / enum                      */
#define ICPFLOW_E_WORKSPACE (-2) /* workspace pointer NULL or too small            */
#define ICPFLOW_E_LIMIT (-3)     /* size beyond what the kernels index (see docs)  */
#define ICPFLOW_E_HOSTMEM (-4)   /* pinned host memory could not be allocated (icpflow_track_frame) */

/* ICP stopping rule */
#define ICPFLOW_STOP_REFERENCE 0 /* batch-global: stop when EVERY pair has rel<=thr
                                    (utils_icp_pytorch3d.py:209); bit-for-bit the
                                    reference's iteration count, no host sync      */
#define ICPFLOW_STOP_PER_PAIR 1  /* each pair stops on its own rel<=thr (or NaN)   */

typedef void *icpflow_stream_t; /* hipStream_t */

int icpflow_version(void);
const char *icpflow_last_error(void);
/* Hash (16 hex digits) of the sources, headers and compiler flags this library was built from
 * (icp_flow_amd/build.py); lets a deployment tell which tree a prebuilt .so belongs to. */
const char *icpflow_build_info(void);

/* ---------------------------------------------------------------------------
 * Per-call options of the fused entry points (last argument; NULL = defaults).
 * Nothing here changes results except icp_arith: every search mode and every
 * ICPFLOW_OPT_* switch selects between implementations that are bit-identical in
 * their outputs (the parity tests run all of them).
 *
 * icp_search -- correspondence search inside the ICP loop:
 *   ICPFLOW_SEARCH_AUTO (0)   sorted sweep when 64 <= N <= 16384, else the all-pairs scan
 *   ICPFLOW_SEARCH_SCAN (1)   all-pairs LDS-tiled scan of the fixed cloud every iteration
 *   ICPFLOW_SEARCH_GRID (2)   exact hashed uniform grid of the fixed cloud, built once per
 *                             registration: only the 27 cells within the gate radius are evaluated
 *   ICPFLOW_SEARCH_SWEEP (3)  both clouds sorted once along the fixed cloud's longest axis; each
 *                             wave scans (LDS broadcast) only the window its queries can gate
 * icp_arith -- arithmetic of the Kabsch step (utils_icp_pytorch3d.py:303-382):
 *   ICPFLOW_ARITH_FP64 (0)    one pass of 18 raw moments in fp64, closed-form rotation in fp64
 *                             (the default: at least as accurate as the reference)
 *   ICPFLOW_ARITH_FP32_REFERENCE (1)  the reference's own operation order in fp32: weighted
 *                             means (:314-315), centring (:318-325), 3x3 product and division
 *                             (:335-336), T = mu_y - mu_x R (:376), rmse from the moved points
 *                             (:191-192) -- reproduces the rounding NOISE LEVEL of the reference's
 *                             tensors (not its bits: the summation order of a GPU reduction differs
 *                             from any other backend's).  All-pairs search, slower; a study mode.
 * profile -- optional recorder of the dominant kernel's launches (see the end of this header).
 * d_vote_bins_u32 -- optional debug output of icpflow_estimate_init_pose / icpflow_hist_icp: the
 *   uint32 bins [B, Lx*Ly*Lz] of the fused (sorted) vote exactly as the peak search reads them.
 * d_icp_init_R [B,3,3], d_icp_init_T [B,3] -- icpflow_icp only, both or neither: `init_transform` of
 *   iterative_closest_point (utils_icp_pytorch3d.py:118-138, scale 1): the first correspondence search runs
 *   on X R0 + T0 instead of X (every iteration still solves for the absolute transform of X).
 * icp_allow_reflection -- icpflow_icp only: `allow_reflection` of corresponding_points_alignment (:354-362): R = U V^T
 *   whatever its determinant (the best ORTHOGONAL matrix; a reflection when det H < 0) instead of the best rotation.
 * icp_estimate_scale, d_icp_scale [B] -- icpflow_icp only: `estimate_scale` of iterative_closest_point (:364-374): the
 *   transform is a similarity, Xt = s X R + T with s = trace(E S) / Xcov; d_icp_scale receives s (may be NULL).
 *   d_icp_init_s [B] (with d_icp_init_R / d_icp_init_T): the scale of the initial transform, NULL = 1.
 * d_icp_history [max_iterations, B, 16] -- icpflow_icp only: `t_history` of the reference's ICPSolution
 *   (:187), i.e. (R row-major 9, T 3, rmse, scale, number of gated correspondences sum(w) of :161, 1 unused) after
 *   every iteration; rows of iterations the batch rule did not reach are unspecified.  Available in the single-launch reference stop mode (max_iterations <= 128,
 *   fp64 arithmetic); otherwise ICPFLOW_E_ARG.
 * ------------------------------------------------------------------------- */
#define ICPFLOW_SEARCH_AUTO 0
#define ICPFLOW_SEARCH_SCAN 1
#define ICPFLOW_SEARCH_GRID 2
#define ICPFLOW_SEARCH_SWEEP 3
#define ICPFLOW_ARITH_FP64 0
#define ICPFLOW_ARITH_FP32_REFERENCE 1
/* developer switches (flags): each turns one optimisation off; results are identical */
#define ICPFLOW_OPT_NO_SORTED_VOTE (1u << 0)  /* all-pairs vote instead of the z-sorted one        */
#define ICPFLOW_OPT_NO_SIDE_STREAM (1u << 1)  /* everything on the caller's stream                 */
#define ICPFLOW_OPT_NO_EVAL_SWEEP (1u << 2)   /* all-pairs match_eval scans                        */
#define ICPFLOW_OPT_NO_CHECK_SWEEP (1u << 3)  /* all-pairs roll-back check                         */
#define ICPFLOW_OPT_NO_SCORE_SWEEP (1u << 4)  /* all-pairs candidate scoring                       */
#define ICPFLOW_OPT_NO_SCORE_PRUNE (1u << 5)  /* every scoring scan runs to the end                */
#define ICPFLOW_OPT_NO_TEAMS (1u << 6)        /* always one workgroup per pair                     */
#define ICPFLOW_OPT_NO_SPECULATIVE (1u << 7)  /* batch-global stop: one launch per iteration       */
#define ICPFLOW_OPT_NO_ADAPTIVE_WINDOWS (1u << 8) /* ICP sweep: no neighbour certificates / probes: every query scans */
#define ICPFLOW_OPT_NO_PERSISTENT (1u << 9)   /* ICP: one workgroup per pair dealt by the hardware dispatcher, whatever B */
#define ICPFLOW_OPT_NO_HELPERS (1u << 10)     /* ICP, persistent grid: workgroups without a pair left do not take passes of others */
/* NOT a bit-identity switch: teams of workgroups (several per large pair) take at most HALF of the CUs, so that two team
 * launches may run side by side (frame pairs in flight; the chains are per device, whatever host threads launch).  The plan -- hence the order of a team's sums --
 * follows the number of workgroups: results equal those of the full-GPU plan to rounding, and are the same whatever else is
 * in flight.  Team launches WITH the flag are chained two deep (two lanes), launches without it wait for both lanes. */
#define ICPFLOW_OPT_TEAMS_HALF_GPU (1u << 11)
/* (a bit-identity switch again) ICP, teams: a wave scans its long windows alone instead of sharing them with the member's other waves */
#define ICPFLOW_OPT_NO_SHARED_SCANS (1u << 12)
/* icpflow_track_frame: stage 2's first half runs behind stage 1 on the caller's stream instead of beside stage 1's ICP on a second
 * one (identical results; the overlap shortens ONE frame pair by ~0.1 ms but registers the whole candidate superset of stage 2 --
 * with several frame pairs in flight on a busy GPU that extra work costs throughput: frame_pairs.register_in_flight sets the flag) */
#define ICPFLOW_OPT_NO_STAGE_OVERLAP (1u << 13)
/* (a bit-identity switch) vote on batches of few wide pairs: the grid covers the padded width of every pair instead of following
 * a work list of the workgroups that have rows, largest pairs first */
#define ICPFLOW_OPT_NO_VOTE_LIST (1u << 14)
/* (a bit-identity switch) icpflow_hist_icp and the entry points built on it: the roll-back check (utils_icp.py:27-35) scans under the
 * initial pose as well, instead of taking that sum from the candidate scoring, whose forward scan of the picked candidate
 * (utils_hist.py:86-101) is the same scan of the same points */
#define ICPFLOW_OPT_NO_CHECK_REUSE (1u << 15)
/* (a bit-identity switch) candidate scoring as pruned sweeps: no occupancy grids of the sorted clouds, i.e. a scan is only ever
 * ended by the lower bounds its waves accumulate while they evaluate targets */
#define ICPFLOW_OPT_NO_SCORE_PREBOUND (1u << 16)
/* (a bit-identity switch, OFF by default: it does not pay -- DESIGN.md 8) ICP of a batch of two to four rounds of half-CU workgroups
 * under the reference's batch-global stop (utils_icp_pytorch3d.py:153-213): the persistent grid is DRAINED once the unfinished pairs
 * fit one CU each (they leave behind their current iteration, still moving), and a second launch gives each of them a 1024-thread
 * workgroup and resumes it at its own iteration from its history rows (icp.hip: icp_split_kernel).  Same sums (added in the order of
 * the 64-query units in either kernel), same history: transforms and iteration counts are those of one launch, bit for bit. */
#define ICPFLOW_OPT_TWO_LAUNCH (1u << 17)
/* (NOT strictly a bit-identity switch: same neighbours, gate decisions, iteration counts and picks; the queries of a pair are visited in
 * another order, so its fp64 moment sums are added in another order -- measured: 1 transform in 8192 differs in one bit, 1.5e-13 m)
 * the sorted sweeps (ICP search, candidate scoring, roll-back check, match_eval): both clouds of every pair are
 * sorted along the fixed cloud's longest AXIS, as before round 6, instead of by the key -- an axis or one of six horizontal
 * directions -- that spreads the fixed cloud best (a vehicle heading along an axis shows a face across it: a third of its points on
 * one key).  The searches are exact under any such key: same neighbours, same sums (utils_icp_pytorch3d.py:153-168, utils_hist.py:86-101) */
#define ICPFLOW_OPT_NO_DIR_KEYS (1u << 18)

typedef struct icpflow_profile icpflow_profile_t; /* opaque */

typedef struct icpflow_options {
    size_t struct_size; /* sizeof(icpflow_options_t) of the caller's header */
    int icp_search;
    int icp_arith;
    unsigned flags;
    icpflow_profile_t *profile;
    uint32_t *d_vote_bins_u32;
    const float *d_icp_init_R;
    const float *d_icp_init_T;
    float *d_icp_history;
    int icp_allow_reflection;
    int icp_estimate_scale;
    float *d_icp_scale;
    const float *d_icp_init_s;
    /* icpflow_hist_icp / icpflow_hist_icp_eval: uint8 [B] or NULL.  Pairs flagged 0 are NOT IN THE BATCH: the batch-global
     * stop of the ICP (utils_icp_pytorch3d.py:209) is taken over the flagged pairs only, their outputs are unspecified.  For
     * callers that must size a batch before the device has decided which of its candidate pairs exist (stage 2 of match_pcds
     * enqueued before stage 1's results are known: icpflow_assoc_assign); the caller hands such pairs over as EMPTY clouds
     * (all flags 0), so that every other kernel of the path skips them.  Reference stop rule, max_iterations <= 128 only. */
    const uint8_t *d_pair_active;
} icpflow_options_t;

/* Bytes of device scratch the fused entry points below need for a batch of B
 * pairs padded to N points with a translation histogram of Lx*Ly*Lz bins.
 * (Pass Lx=Ly=Lz=0 for entry points that do not vote.) */
size_t icpflow_workspace_bytes(int B, int N, int Lx, int Ly, int Lz);

/* ---------------------------------------------------------------------------
 * a-1  translation-histogram vote.
 * Replaces: HIST.hist / hist_cuda() -- hist_cuda/hist.py:39-51, cpp/hist.cpp:4-27,
 * cpp/hist_cuda.cu:19-90, kernel cpp/hist_cuda_core.cuh:23-64.
 * For every valid i in X[b], valid j in Y[b]: v = X_i - Y_j; if min <= v < max on
 * all axes, p = floor((v-min)/(max-min)*float(len)) and bins[b,px,py,pz] += 1.
 * d_bins: float32 [B, len_x, len_y, len_z], fully overwritten (the reference
 * returns a fresh zero-initialised tensor, hist_cuda.cu:59).  Bit-exact with the
 * reference's integer-valued float counters; `mini_batch` of the reference only
 * chunks launches (hist_cuda.cu:61-85) and has no equivalent here.
 * ------------------------------------------------------------------------- */
int icpflow_hist_vote(const float *d_X, const float *d_Y, int B, int NX, int NY,
                      float min_x, float min_y, float min_z,
                      float max_x, float max_y, float max_z,
                      int len_x, int len_y, int len_z,
                      float *d_bins, icpflow_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-2  3-D non-maximum suppression + top-k peaks.
 * Replaces: topk_nms -- utils_hist.py:21-29 (max_pool3d(kernel, stride 1,
 * pad (kernel-1)/2) == x, then topk over the flattened surviving votes).
 * d_votes float32 [B,k], d_idx int64 [B,k] (flat index into [Lx,Ly,Lz]).
 * Order: vote descending, ties by ascending flat index (torch.topk's order among
 * equal votes is implementation-defined; this rule is deterministic).
 * d_ws: at least 2 * B*Lx*Ly*Lz * 4 bytes of scratch.
 * ------------------------------------------------------------------------- */
int icpflow_hist_peaks(const float *d_bins, int B, int len_x, int len_y, int len_z,
                       int k, int kernel_size, float *d_votes, int64_t *d_idx,
                       void *d_ws, size_t ws_bytes, icpflow_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-4  brute-force K=1 nearest neighbour, both clouds scanned in full.
 * Replaces: nearest_neighbor_batch -- utils_helper.py:20-30, i.e.
 * pytorch3d.ops.knn_points(src[:,:,0:3], dst[:,:,0:3], K=1) followed by sqrt.
 * d_Q float32 [B,NQ,q_stride], d_T float32 [B,NT,t_stride] (strides in floats,
 * >= 3; the first three columns are x,y,z).  d_len_q / d_len_t: optional int32
 * [B] valid prefixes (pytorch3d `lengths1/2`, utils_icp_pytorch3d.py:154-156);
 * NULL = all NQ / NT rows, pads included, exactly like the reference's
 * un-lengthed call.  Rows >= len_q get idx 0 / dist 0 (pytorch3d convention).
 * d_idx int64 [B,NQ]; d_dist float32 [B,NQ]: Euclidean distance if
 * sqrt_dist != 0, squared distance otherwise.  First minimum wins ties.
 * ------------------------------------------------------------------------- */
int icpflow_nn_batch(const float *d_Q, const float *d_T, int B, int NQ, int NT,
                     int q_stride, int t_stride, const int32_t *d_len_q,
                     const int32_t *d_len_t, int sqrt_dist, int64_t *d_idx,
                     float *d_dist, icpflow_stream_t stream);

/* a-10  transform_points_batch -- utils_helper.py:76-87:
 * out[b,i,0:3] = [x y z 1] * pose[b]^T, flag column copied.  In-place allowed. */
int icpflow_transform_points(const float *d_xyz, const float *d_pose, int B, int N,
                             float *d_out, icpflow_stream_t stream);

/* Number of valid points per pair: d_len[b] = count(flag > 0) (int32 [B]). */
int icpflow_count_valid(const float *d_pts, int B, int N, int32_t *d_len,
                        icpflow_stream_t stream);

/* ---------------------------------------------------------------------------
 * a-3  initial pose from the translation histogram.
 * Replaces: estimate_init_pose / estimate_init_pose_batch -- utils_hist.py:33-124
 * (vote with X=dst, Y=src; NMS + top-5; decode to left bin edges; append the zero
 * translation; score the 6 candidates by min(mean fwd NN dist, mean bwd NN dist)
 * over valid points; first arg-min).  `chunk_size` of the reference only bounds
 * memory and does not change results, so it has no equivalent.
 * d_edges_{x,y,z}: float32 left bin edges exactly as the reference builds them
 * (torch.arange, utils_hist.py:63-65); vote box = [edges[0], edges[L-1]) with L bins
 * (the reference passes bins.min()/bins.max()/len(bins), utils_hist.py:69-72).
 * decode_shift = thres_dist // 2 (utils_hist.py:78; 0.0 for thres_dist 0.1).
 * d_T_out float32 [B,4,4]: identity with the winning translation in column 3.
 * ------------------------------------------------------------------------- */
int icpflow_estimate_init_pose(const float *d_src, const float *d_dst, int B, int N,
                               const float *d_edges_x, int len_x,
                               const float *d_edges_y, int len_y,
                               const float *d_edges_z, int len_z,
                               float decode_shift, float *d_T_out,
                               void *d_ws, size_t ws_bytes, icpflow_stream_t stream,
                               const icpflow_options_t *opt);

/* ---------------------------------------------------------------------------
 * a-5..a-7  masked batched point-to-point ICP.
 * Replaces: iterative_closest_point -- utils_icp_pytorch3d.py:37-225 with
 * corresponding_points_alignment (:233-382) and _apply_similarity_transform
 * (:385-396), estimate_scale=False, allow_reflection=False.
 * Per iteration: NN of the current X in Y (valid prefixes), gate d^2 <= thres^2,
 * Kabsch on the gated pairs FROM THE ORIGINAL X (absolute, not incremental),
 * rmse, relative-rmse stop (see ICPFLOW_STOP_*).  Row-vector convention
 * y = x R + T as in the reference.
 * d_pre_pose: optional float32 [B,4,4]; when non-NULL the ICP input is
 * transform_points_batch(X, pre_pose) (utils_icp.py:21) computed on the fly.
 * Thresholds are passed as the reference's Python doubles and narrowed exactly where
 * torch narrows them: the gate compares the fp32 squared distance with
 * (float)(thres*thres) (= 0x3C23D70A for 0.1; squaring the fp32 value 0.1f instead would
 * give the next float up), the stop compares with (float)relative_rmse_thr.
 * Outputs (any may be NULL): d_R [B,3,3], d_T [B,3], d_rmse [B],
 * d_iters int32 [1] (loop bodies executed; per-pair mode: max over pairs),
 * d_converged int32 [1].
 * Failure inside the launch: when several workgroups share one large pair (N > 1024, batch smaller
 * than half the GPU) and a member is not scheduled for 2 s -- another process saturating the GPU --
 * the registration is abandoned instead of hanging the queue: every R of the batch is NaN and
 * *d_iters = -1 (also for icpflow_apply_icp / icpflow_hist_icp, whose transforms are then NaN).
 * Callers must treat iters < 0 as an error (the Python mirrors raise).
 * ------------------------------------------------------------------------- */
int icpflow_icp(const float *d_X, const float *d_Y, const float *d_pre_pose, int B, int N,
                double thres, int max_iterations, double relative_rmse_thr, int stop_mode,
                float *d_R, float *d_T, float *d_rmse, int32_t *d_iters,
                int32_t *d_converged, void *d_ws, size_t ws_bytes,
                icpflow_stream_t stream, const icpflow_options_t *opt);

/* ---------------------------------------------------------------------------
 * a-8, a-9  ICP from an initial pose with roll-back.
 * Replaces: apply_icp + pytorch3d_icp -- utils_icp.py:20-73 (pre-transform src,
 * ICP with max_iterations / relative_rmse_thr (reference: 100 / 1e-6), 4x4 =
 * [[R^T, T],[0,0,0,1]] * init, mean NN error of valid src before / after, and
 * `Rts[error_icp >= error_init] = init`).
 * d_T_out float32 [B,4,4]; may alias d_init.
 * ------------------------------------------------------------------------- */
int icpflow_apply_icp(const float *d_src, const float *d_dst, const float *d_init,
                      int B, int N, double thres_dist, int max_iterations,
                      double relative_rmse_thr, int stop_mode, float *d_T_out,
                      int32_t *d_iters, void *d_ws, size_t ws_bytes,
                      icpflow_stream_t stream, const icpflow_options_t *opt);

/* ---------------------------------------------------------------------------
 * a-11  one full registration per cluster pair.
 * Replaces: hist_icp -- utils_match.py:138-157 (register the cloud with fewer
 * valid points onto the other (strict >, ties not swapped), a-3 then a-9, and
 * invert the 4x4 of swapped pairs).  The swap is done by pointer selection, the
 * inputs are never copied.  The inverse is the analytic rigid inverse (the
 * reference's general torch.linalg.inv leaves ~1e-8 noise in the bottom row).
 * d_T_out float32 [B,4,4] maps the ORIGINAL src onto dst.
 * ------------------------------------------------------------------------- */
int icpflow_hist_icp(const float *d_src, const float *d_dst, int B, int N,
                     const float *d_edges_x, int len_x,
                     const float *d_edges_y, int len_y,
                     const float *d_edges_z, int len_z,
                     float decode_shift, double thres_dist, int max_iterations,
                     double relative_rmse_thr, int stop_mode, float *d_T_out,
                     int32_t *d_iters, void *d_ws, size_t ws_bytes,
                     icpflow_stream_t stream, const icpflow_options_t *opt);

/* ---------------------------------------------------------------------------
 * a-11, several batches in flight.  K independent calls of icpflow_hist_icp (same N, histogram and ICP parameters;
 * batch k: d_src[k], d_dst[k] of B[k] pairs, outputs d_T_out[k], d_iters[k], its OWN workspace d_ws[k] of
 * ws_bytes[k] >= icpflow_workspace_bytes(B[k], N, ...)), enqueued on internal worker streams that are forked from and
 * joined back into `stream`: to the caller it is one asynchronous call on `stream`.  Every batch keeps its own
 * batch-global ICP stop (utils_icp_pytorch3d.py:209 couples the pairs of ONE hist_icp call, utils_match.py:138-157) and
 * its results are bit-identical to a call of its own; what the overlap buys is the tail of one batch's ICP launch
 * -- few pairs still iterating, most CUs idle -- running under the vote and scoring of the others.  The reference
 * has no counterpart (it registers one batch after the other, main.py:184-215); frame pairs and association stages
 * of different frames are independent and can be registered this way.
 * ------------------------------------------------------------------------- */
int icpflow_hist_icp_many(int K, const float *const *d_src, const float *const *d_dst, const int *B, int N,
                          const float *d_edges_x, int len_x, const float *d_edges_y, int len_y,
                          const float *d_edges_z, int len_z, float decode_shift, double thres_dist,
                          int max_iterations, double relative_rmse_thr, int stop_mode, float *const *d_T_out,
                          int32_t *const *d_iters, void *const *d_ws, const size_t *ws_bytes,
                          icpflow_stream_t stream, const icpflow_options_t *opt);

/* ---------------------------------------------------------------------------
 * a-12  registration quality metrics.
 * Replaces: match_eval -- utils_match.py:159-213.  All outputs float32:
 * d_errors, d_inliers, d_ratios, d_ious [B,2] (src direction, dst direction);
 * d_translations [B,3]; d_rotations [B,3] (Euler ZYX, degrees).
 * ------------------------------------------------------------------------- */
int icpflow_match_eval(const float *d_pcd1, const float *d_pcd2, const float *d_T,
                       int B, int N, double thres_dist, float *d_errors,
                       float *d_inliers, float *d_ratios, float *d_ious,
                       float *d_translations, float *d_rotations, void *d_ws,
                       size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt);

/* a-11 + a-12 in one call: icpflow_hist_icp followed by icpflow_match_eval of the same clouds under the transforms it
 * found (utils_match.py:92-93: `hist_icp(...)` then `match_eval(...)`, every candidate pair of match_pairs).  Same
 * results as the two calls; the metrics reuse what the registration left in the workspace (valid-row counts, both clouds
 * sorted along one axis) instead of counting and sorting again.  Arguments as in the two entry points. */
int icpflow_hist_icp_eval(const float *d_src, const float *d_dst, int B, int N, const float *d_edges_x, int len_x,
                          const float *d_edges_y, int len_y, const float *d_edges_z, int len_z, float decode_shift,
                          double thres_dist, int max_iterations, double relative_rmse_thr, int stop_mode,
                          float *d_T_out, int32_t *d_iters, float *d_errors, float *d_inliers, float *d_ratios,
                          float *d_ious, float *d_translations, float *d_rotations, void *d_ws, size_t ws_bytes,
                          icpflow_stream_t stream, const icpflow_options_t *opt);

/* ---------------------------------------------------------------------------
 * a-15 / 8(f)  helpers of the host association around the path.
 *
 * icpflow_gather_pad builds the padded batch of match_pairs (utils_match.py:81-91 with
 * pad_segment, utils_helper.py:185-196): d_rows int32 [B,N] holds, per pair and slot, the row of
 * d_points ([M,3] float32) to copy (flag 1) or -1 for a pad row (1e8,1e8,1e8,0).  The caller
 * decides the rows (cluster order, random subsample of over-long clusters).
 *
 * icpflow_gather_segments builds the same batch straight from the label-sorted row table d_order
 * (int64 [M]): d_seg int64 [3,B] holds per pair the first row of its cluster in d_order, the number of
 * rows to take (<= N) and -1 or an offset into d_perm (int32) when the host subsampled an over-long
 * cluster (random_choice, utils_helper.py:198-201): row i = d_order[start + d_perm[off + i]].
 *
 * icpflow_cluster_stats reduces what sanity_check reads per cluster (utils_check.py:34-43,
 * get_bbox_tensor utils_helper.py:166-170): d_order int64 [M] = rows of d_points ([M,3]) sorted by
 * label, cluster c = d_order[d_start[c] .. d_start[c]+d_count[c]) (int64 [L] each); outputs the
 * centroid d_mean [L,3] and the ascending-sorted bbox extents d_extent [L,3] (float32).  d_labels
 * (float32 [L], optional): clusters with a negative label (ground, noise -- never candidates,
 * utils_check.py:32) are skipped and report zeros.
 *
 * icpflow_flow_rigid replaces flow_estimation_torch (utils_flow.py:57-69): every point whose
 * float label equals d_pair_labels[p] moves with T[p]*pose, every other point with pose alone;
 * flow = moved - point.  d_ws: not used since version 204 (may be NULL).
 * ------------------------------------------------------------------------- */
int icpflow_gather_pad(const float *d_points, const int32_t *d_rows, int B, int N, float *d_out,
                       icpflow_stream_t stream);
/* Device-side association of a frame pair (match_pairs' host half, utils_match.py:96-115, as kernels -- so that match_pcds can
 * enqueue both of its stages and the flow without reading a stage's results back).
 *
 * icpflow_assoc_assign: one association stage.  d_result = the results of icpflow_hist_icp_eval for the stage's K candidate
 * pairs, float32 array after array [T 16K | errors 2K | inliers 2K | ratios 2K | ious 2K | translations 3K | rotations 3K];
 * candidate k pairs source cluster d_si[k] (row of the source table, < S) with destination cluster d_di[k] (< D); d_active
 * (uint8 [K] or NULL): candidates flagged 0 do not exist.  A candidate survives check_transformation (utils_check.py:51-66)
 * iff |translation| <= translation_frame, min(iou) >= thres_iou and max(|rot_y|, |rot_x|) <= rot_limit_deg; every source
 * row takes the surviving candidate with the smallest min(error_src, error_dst), ties to the smallest destination row
 * (np.argmin's first minimum, utils_helper.py:108-110), if that error is < thres_error (utils_match.py:112): d_best int32 [S]
 * receives its index or -1.  With K2 > 0 the NEXT stage's candidates (d_si2, d_di2 [K2]) are switched on or off in the same
 * launch (utils_match.py:45-53: only clusters that found no partner go on): d_active2 uint8 [K2], and the lengths of the
 * inactive ones in d_seg2 (int64 [2,3,K2], the segment rows of icpflow_gather_segments for both clouds) are set to 0.
 *
 * icpflow_assoc_collect: the pair rows of both stages, stage 1 first, each in ascending source row: d_rows float32 [cap,10]
 * (source label, destination label, errors, inliers, ratios, ious: utils_match.py:120-131), d_T float32 [cap,16]; rows beyond
 * the matches carry the label -3e38 (no point has it) and the identity.  d_count int32: the number of matches, or -1 when a
 * stage reported an abandoned batch (iteration count < 0).  Labels: float64 tables with `label_stride` doubles per row (the
 * d_table of icpflow_cluster_table).  Stage 2 may be absent (K2 = 0). */
int icpflow_assoc_assign(const float *d_result, const int32_t *d_si, const int32_t *d_di, int K, const uint8_t *d_active,
                         int S, int D, float translation_frame, float thres_iou, float rot_limit_deg, float thres_error,
                         int32_t *d_best, int K2, const int32_t *d_si2, const int32_t *d_di2, int64_t *d_seg2,
                         uint8_t *d_active2, icpflow_stream_t stream);
int icpflow_assoc_collect(const int32_t *d_best1, const float *d_result1, const int32_t *d_si1, const int32_t *d_di1, int K1,
                          const int32_t *d_best2, const float *d_result2, const int32_t *d_si2, const int32_t *d_di2, int K2,
                          const double *d_src_table, const double *d_dst_table, int label_stride, int S, int cap,
                          float *d_rows, float *d_T, int32_t *d_count, icpflow_stream_t stream);
int icpflow_gather_segments(const float *d_points, const int64_t *d_order, const int64_t *d_seg,
                            const int32_t *d_perm, int B, int N, float *d_out, icpflow_stream_t stream);

/* The same, one call per stage instead of one per kernel (version 206) -- match_pairs (utils_match.py:69-136) and match_pcds
 * (utils_match.py:26-66) from the cluster tables on.  A frame pair in flight costs its host thread a fixed price per call
 * into the library; these two take what used to be ten.
 *
 * icpflow_tables_t: both clouds of the frame pair as icpflow_cluster_table leaves them (points float32 rows of 3, the
 * label-sorted order, the float64 table with `label_stride` doubles per cluster, label first; S and D clusters).
 * icpflow_stage_t: K candidate pairs of an association stage.  d_seg int64 [2,3,K] are the segment rows of
 * icpflow_gather_segments (start, length, offset of the subsample in d_perm or -1; source cloud first), d_si / d_di int32 [K]
 * the table rows of each candidate, N the width of the padded batch; d_clouds float32 [2,K,N,4] is scratch for the padded
 * batch, d_result float32 [30 K + 1] receives the results of icpflow_hist_icp_eval, array after array
 * [T 16K | errors 2K | inliers 2K | ratios 2K | ious 2K | translations 3K | rotations 3K | iteration count (int32)].
 * icpflow_registration_t: the arguments of icpflow_hist_icp_eval that do not depend on the batch.
 *
 * icpflow_register_stage: pad_segment of both clouds (utils_match.py:81-91) + hist_icp + match_eval of one stage.
 * icpflow_associate_frame: everything of match_pcds behind stage 1's registration (`stage1`, registered by
 *   icpflow_register_stage on the same stream): icpflow_assoc_assign of stage 1 (which switches stage 2's candidates on or
 *   off: d_active2 uint8 [stage2->K], scratch), icpflow_register_stage of stage 2 with options.d_pair_active = d_active2,
 *   icpflow_assoc_assign of stage 2, icpflow_assoc_collect (d_best int32 [2 S + 2]: stage 1's choice per source row, stage
 *   2's, the number of matches; d_rows [cap,10], d_T [cap,16]) and, with d_flow != NULL, icpflow_flow_rigid_rows of n_flow
 *   points on the padded pair rows.  stage2 may be NULL or have K = 0.  d_ws / ws_bytes: icpflow_workspace_bytes of the
 *   larger stage. */
typedef struct icpflow_tables {
    const float *d_points_src;
    const int64_t *d_order_src;
    const double *d_table_src;
    const float *d_points_dst;
    const int64_t *d_order_dst;
    const double *d_table_dst;
    int S, D, label_stride;
} icpflow_tables_t;

typedef struct icpflow_stage {
    const int64_t *d_seg;
    const int32_t *d_perm;
    const int32_t *d_si;
    const int32_t *d_di;
    float *d_clouds;
    float *d_result;
    int K, N;
} icpflow_stage_t;

typedef struct icpflow_registration {
    const float *d_edges_x;
    const float *d_edges_y;
    const float *d_edges_z;
    int len_x, len_y, len_z;
    float decode_shift;
    double thres_dist;
    double relative_rmse_thr;
    int max_iterations;
    int stop_mode;
} icpflow_registration_t;

int icpflow_register_stage(const icpflow_tables_t *tables, const icpflow_stage_t *stage, const icpflow_registration_t *reg,
                           void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt);
int icpflow_associate_frame(const icpflow_tables_t *tables, const icpflow_stage_t *stage1, const icpflow_stage_t *stage2,
                            uint8_t *d_active2, const icpflow_registration_t *reg, float translation_frame, float thres_iou,
                            float rot_limit_deg, float thres_error, int32_t *d_best, int cap, float *d_rows, float *d_T,
                            const float *d_flow_points, const float *d_flow_labels, int n_flow, const float *d_pose,
                            float *d_flow, void *d_ws, size_t ws_bytes, icpflow_stream_t stream,
                            const icpflow_options_t *opt);
/* (version 209) icpflow_register_stage in two halves, so that a frame pair's stage 2 can estimate its initial poses on another
 * stream BESIDE stage 1's ICP (a chain of dependent iterations on a few long pairs that leaves most of the GPU idle):
 * _begin gathers the clouds of the WHOLE candidate list and enqueues lengths, sorts, vote, peaks and scoring (utils_hist.py:82-124)
 * -- and, where one speculative launch runs the batch rule (reference stop, 2 <= max_iterations <= 128, fp64 arithmetic), the ICP of
 * ALL candidates (h_carry[2] = 1) -- on `stream`, leaving initial poses (and trajectories) in d_ws and a few host words in h_carry;
 * _finish enqueues the rest (the batch rule over the pairs that are in the batch after all, found from the trajectories; else the ICP
 * itself; roll-back check, metrics; utils_icp.py:20-35, utils_match.py:159-213) from that workspace -- options.d_pair_active of the second call says which
 * candidates are in the batch after all: the others keep their clouds and are passed over by every kernel (their result rows are
 * unspecified).  The pairs in the batch get bit for bit what icpflow_register_stage gives them with the others handed over as
 * empty clouds.  The caller orders the two calls (an event from _begin's stream to _finish's); d_ws must not be used in between.
 * icpflow_associate_frame_begun: icpflow_associate_frame for a stage 2 that has been begun (d_ws2 / h_carry2: its workspace). */
int icpflow_register_stage_begin(const icpflow_tables_t *tables, const icpflow_stage_t *stage, const icpflow_registration_t *reg,
                                 void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt,
                                 int32_t *h_carry /* [4] */);
int icpflow_register_stage_finish(const icpflow_tables_t *tables, const icpflow_stage_t *stage, const icpflow_registration_t *reg,
                                  void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt,
                                  const int32_t *h_carry /* [4] */);
int icpflow_associate_frame_begun(const icpflow_tables_t *tables, const icpflow_stage_t *stage1, const icpflow_stage_t *stage2,
                                  uint8_t *d_active2, const icpflow_registration_t *reg, float translation_frame, float thres_iou,
                                  float rot_limit_deg, float thres_error, int32_t *d_best, int cap, float *d_rows, float *d_T,
                                  const float *d_flow_points, const float *d_flow_labels, int n_flow, const float *d_pose,
                                  float *d_flow, void *d_ws2, size_t ws2_bytes, icpflow_stream_t stream,
                                  const icpflow_options_t *opt, const int32_t *h_carry2 /* [4] */);
/* ---------------------------------------------------------------------------
 * One frame pair per call (version 207): match_pcds (utils_match.py:24-66) + flow_estimation_torch (utils_flow.py:57-69) from two
 * labelled clouds on, the HOST half included -- cluster tables, candidate lists, sanity_check (utils_check.py:21-49), padded
 * batches with the reference's stream of random subsamples, both association stages (device-side association: icpflow_
 * register_stage + icpflow_associate_frame), the flow.  BLOCKING: returns when the results are complete on the device (two
 * waits inside: the cluster tables, the end).  A host thread per frame pair in flight, each on its own stream, keeps several
 * going: the time a Python host spends between the calls of the finer-grained entry points is what binds a stream of frame
 * pairs.
 *
 * d_points_* float32 [n,3], d_labels_* float32 [n].  par: the flags of the reference's argument parser that the path reads
 * (main.py / demo.sh) + `seed`: the over-long clusters of stage 1 are subsampled by torch.randperm's algorithm on MT19937
 * seeded with it (= torch.Generator().manual_seed(seed), the generator frame_pairs.py gives every frame pair).
 * Outputs: d_rows float32 [1024,10], d_T float32 [1024,16] (the first *h_pairs rows are the matches, the rest padding),
 * d_flow float32 [n_src,3] of d_flow_points (or NULL: no flow) under d_pose [4,4].  *h_pairs: the number of matched pairs;
 * ICPFLOW_FRAME_HOST_PATH when this call cannot serve the frame pair (no candidate pair at all, more than 512 clusters: the caller runs
 * the finer-grained path, nothing was consumed).  When a stage-2 candidate with a cluster too long for the superset turns out
 * to be needed, the call registers stage 2 once more the reference's way -- its exact candidates, read from stage 1's
 * assignment, subsamples drawn next from the same generator -- on top of stage 1's results (one more wait); ICPFLOW_FRAME_ABANDONED when a team's wait timed out (transforms NaN).  d_scratch / scratch_bytes: device
 * scratch; on ICPFLOW_E_WORKSPACE *scratch_needed says how much this frame pair needs (call again with that much).
 * ------------------------------------------------------------------------- */
#define ICPFLOW_FRAME_HOST_PATH (-2)
#define ICPFLOW_FRAME_HOST_ASSOCIATION (-3) /* (reserved: as HOST_PATH, and the device-side association cannot serve the frame pair either) */
#define ICPFLOW_FRAME_ABANDONED (-1)
/* The state of torch's CPU generator (at::mt19937): the 624 words and how many of them have been consumed (624 = the next
 * draw regenerates the block; a generator fresh from manual_seed).  icp_flow_amd/_lib.py converts to and from
 * torch.Generator.get_state(). */
typedef struct icpflow_mt19937 {
    uint32_t state[624];
    int32_t index;
} icpflow_mt19937_t;
typedef struct icpflow_frame_params {
    size_t struct_size;       /* sizeof(icpflow_frame_params_t) */
    uint64_t seed;            /* the draws start from torch.Generator().manual_seed(seed) ... */
    icpflow_mt19937_t *generator; /* ... or, when not NULL, from this state, which is advanced (only when the call serves the frame pair) */
    int max_points;           /* --max_points */
    int min_cluster_size;     /* --min_cluster_size */
    float translation_frame;  /* --translation_frame (or 2 * speed * gap, main.py:200) */
    float thres_box;          /* --thres_box */
    float thres_iou;          /* --thres_iou */
    float rot_limit_deg;      /* --thres_rot * 90 */
    float thres_error;        /* --thres_error */
    int tight_padding;        /* 1: batches as wide as their longest cluster (icp_flow_amd's default), 0: max_points wide */
    int superset_width;       /* clusters above it stay out of stage 2's superset (0 = 1024) */
} icpflow_frame_params_t;
int icpflow_track_frame(const float *d_points_src, const float *d_labels_src, int n_src, const float *d_points_dst,
                        const float *d_labels_dst, int n_dst, const icpflow_registration_t *reg,
                        const icpflow_frame_params_t *par, float *d_rows, float *d_T, int32_t *h_pairs,
                        const float *d_flow_points, const float *d_pose, float *d_flow, void *d_scratch,
                        size_t scratch_bytes, size_t *scratch_needed, icpflow_stream_t stream,
                        const icpflow_options_t *opt);
/* torch.randperm(n, generator)[0:take] as this library restates it (host only; the CPU tests pin it against torch) */
int icpflow_selftest_randperm(icpflow_mt19937_t *gen, int64_t n, int take, int32_t *h_out);

int icpflow_cluster_stats(const float *d_points, const int64_t *d_order, const int64_t *d_start,
                          const int64_t *d_count, const float *d_labels, int L, float *d_mean,
                          float *d_extent, icpflow_stream_t stream);
/* icpflow_cluster_table: everything match_pcds needs to know about the clusters of ONE labelled cloud, in one chain of
 * launches: d_order int64 [M] = the rows sorted by label, STABLE (rows of a cluster in their original order: the
 * reference's random subsample of over-long clusters, utils_helper.py:198-201, indexes into that order); d_table
 * float64 [Lmax, 9], row c = (label, number of rows, first position in d_order, centroid x y z, bounding-box extents in
 * ascending order) of the c-th distinct label in ascending order -- what torch.unique(labels) (utils_match.py:27-29),
 * the per-pair masks (:81-86) and sanity_check (utils_check.py:34-43, get_bbox_tensor utils_helper.py:166-170) compute
 * cluster by cluster; centroid and extents are zero for negative labels (ground, noise: never candidates,
 * utils_check.py:32).  d_num int32 [1]: the number of distinct labels, or its negative when it exceeds Lmax (<= 4096;
 * the table then holds nothing usable).  float64 carries the float32 statistics and the counts exactly. */
size_t icpflow_cluster_table_workspace_bytes(int M, int Lmax);
int icpflow_cluster_table(const float *d_points, const float *d_labels, int M, int64_t *d_order, double *d_table, int Lmax,
                          int32_t *d_num, void *d_ws, size_t ws_bytes, icpflow_stream_t stream);
/* The tables of BOTH clouds of a frame pair (what match_pcds needs before anything else, utils_match.py:27-29) in one
 * chain of launches -- half as many as two icpflow_cluster_table calls; every output exactly what those calls write. */
size_t icpflow_cluster_table_pair_workspace_bytes(int MA, int MB, int Lmax);
int icpflow_cluster_table_pair(const float *d_points_a, const float *d_labels_a, int MA, int64_t *d_order_a,
                               double *d_table_a, int32_t *d_num_a, const float *d_points_b, const float *d_labels_b,
                               int MB, int64_t *d_order_b, double *d_table_b, int32_t *d_num_b, int Lmax, void *d_ws,
                               size_t ws_bytes, icpflow_stream_t stream);
int icpflow_flow_rigid(const float *d_points, const float *d_labels, int N, const float *d_pair_labels,
                       const float *d_T, int P, const float *d_pose, float *d_flow, void *d_ws,
                       size_t ws_bytes, icpflow_stream_t stream);
/* The same with the matched source labels read where match_pcds leaves them: column 0 of its pair rows (d_pair_rows
 * float32 [P, pair_stride], pair_stride = 10 for the [P,10] rows of utils_match.py:118-127; 1 = a plain array). */
int icpflow_flow_rigid_rows(const float *d_points, const float *d_labels, int N, const float *d_pair_rows,
                            int pair_stride, const float *d_T, int P, const float *d_pose, float *d_flow,
                            icpflow_stream_t stream);

/* ---------------------------------------------------------------------------
 * 8(f) row 4  density clustering of a frame pair's points -- the `cluster_dbscan` branch of
 * cluster_pcd (utils_cluster.py:32-48, 50-63), i.e. open3d 0.17.0 PointCloud::cluster_dbscan(eps,
 * min_points) (environment.yml:227; third-party, not in the reference tree).
 *
 * d_points float32 rows of `stride` floats (x, y, z first), n rows.  d_mask (uint8 [n], optional):
 * rows with mask 0 are not clustered (the ground rows of cluster_pcd's idxs_nonground) and report -2.
 * A neighbour is a point whose squared distance, evaluated in fp64, is STRICTLY below eps^2; a point
 * with >= min_points neighbours (itself included) is a core point.
 * Outputs: d_labels int32 [n]: cluster id 0..C-1 in order of each cluster's smallest core-point row
 * (Open3D's numbering), -1 noise, -2 masked out;  d_counts int32 [n]: the first C entries are the
 * cluster sizes;  d_num_clusters int32 [1] = C.  Keeping only the num_clusters largest clusters
 * (utils_cluster.py:38-45) is host logic on d_counts (icp_flow_amd/utils_cluster.py).
 * ------------------------------------------------------------------------- */
size_t icpflow_dbscan_workspace_bytes(int n);
int icpflow_dbscan(const float *d_points, int stride, const uint8_t *d_mask, int n, double eps, int min_points,
                   int32_t *d_labels, int32_t *d_counts, int32_t *d_num_clusters, void *d_ws, size_t ws_bytes,
                   icpflow_stream_t stream);

/* ---------------------------------------------------------------------------
 * 8(f) row 4, HDBSCAN branch of cluster_pcd (utils_cluster.py:10-29: hdbscan.HDBSCAN(min_cluster_size,
 * min_samples=None, metric='euclidean', alpha=1.0), third-party, environment.yml:57): the part that is
 * O(n^2) on the CPU -- core distances and the minimum spanning tree of the mutual-reachability graph
 *     d(a, b) = max(core(a), core(b), |a - b|),   core(a) = distance to a's min_samples-th nearest
 *     neighbour, a ITSELF COUNTED (tree.query(X, k)[:, -1]: scikit-learn's convention).  The hdbscan package
 *     does not count the point (its boruvka_kdtree path queries k = min_samples + 1, _hdbscan_boruvka.pyx):
 *     pass its min_samples + 1 here -- icp_flow_amd/utils_cluster.py does, for the reference's
 *     HDBSCAN(min_cluster_size, min_samples=None) that is min_cluster_size + 1.
 * The dendrogram, condensed tree and cluster selection on the n - 1 edges are sequential host logic
 * (icp_flow_amd/utils_cluster.py).
 *
 * d_points / stride / d_mask as icpflow_dbscan.  cell: edge of the uniform sort grid in metres (speed
 * only; 0.25 suits LiDAR frames).  Outputs: d_edge_a / d_edge_b int32 [n], d_edge_w2 float64 [n]: the
 * n_live - 1 tree edges (caller rows, SQUARED weight, unordered); d_num_edges, d_num_live int32 [1];
 * d_core2 float64 [n] (optional): squared core distance per caller row, NaN for rows that took no part.
 * Exact: ties between equal weights are broken by (smaller row, larger row), so the tree is unique.
 * ------------------------------------------------------------------------- */
size_t icpflow_hdbscan_mst_workspace_bytes(int n);
int icpflow_hdbscan_mst(const float *d_points, int stride, const uint8_t *d_mask, int n, int min_samples, double cell,
                        double *d_core2, int32_t *d_edge_a, int32_t *d_edge_b, double *d_edge_w2,
                        int32_t *d_num_edges, int32_t *d_num_live, void *d_ws, size_t ws_bytes,
                        icpflow_stream_t stream);

/* HOST function (host pointers, no device work, callable without a GPU): HDBSCAN's sequential remainder on
 * the n_points - 1 spanning-tree edges -- what hdbscan.HDBSCAN / sklearn's HDBSCAN do after their spanning tree
 * (sklearn/cluster/_hdbscan/_linkage.pyx make_single_linkage, _tree.pyx tree_to_labels with "eom", no single
 * cluster, epsilon 0): edges (rows of the clustered subset, 0-based; weight = mutual-reachability DISTANCE,
 * i.e. the square root of icpflow_hdbscan_mst's output) -> h_labels int32 [n_points], -1 = noise.
 * Returns ICPFLOW_E_ARG when the edges do not span the points or min_cluster_size < 2. */
int icpflow_hdbscan_labels(const int32_t *h_edge_a, const int32_t *h_edge_b, const double *h_edge_w,
                           int n_points, int min_cluster_size, int32_t *h_labels);

/* ---------------------------------------------------------------------------
 * Diagnostics: the vote kernels evaluate (v - min) / (max - min) with the loop-invariant part of
 * the IEEE division hoisted (hist.hip, AxisQuot).  For numerators d_a [n] this returns that
 * quotient (d_fast) next to the compiler's correctly rounded a / (max - min) (d_ieee); the two
 * must be bit-identical (hist_cuda_core.cuh:52-54 is an IEEE division).
 * ------------------------------------------------------------------------- */
int icpflow_selftest_vote_quotient(const float *d_a, int n, float min_v, float max_v, float *d_fast,
                                   float *d_ieee, icpflow_stream_t stream);

/* ---------------------------------------------------------------------------
 * Measurement aid (no reference counterpart): per-launch timing of the dominant kernel, the ICP
 * iteration.  A recorder is an object the caller owns: every launch of that kernel made by a call
 * that carries it in its options (up to `capacity` launches) is bracketed by HIP events recorded on
 * the caller's stream; icpflow_profile_collect() waits for the recorded events, returns the summed
 * duration in milliseconds and the number of launches, and re-arms the recorder.  One recorder must
 * not be used by two host threads at once; distinct recorders are independent.
 * ------------------------------------------------------------------------- */
int icpflow_profile_create(int capacity, icpflow_profile_t **out);
int icpflow_profile_collect(icpflow_profile_t *profile, double *total_ms, int *launches);
int icpflow_profile_destroy(icpflow_profile_t *profile);

#ifdef __cplusplus
}
#endif
#endif /* ICPFLOW_HIP_H */
