// kernels.hpp -- internal launcher interface between the translation units of
// libicpflow_hip.so (each .hip file owns its kernels; api.hip sequences them).
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <stddef.h>
#include <stdint.h>

#include "../../include/icpflow_hip.h"

namespace icpflow {

constexpr int ICPFLOW_STOP_REFERENCE_ = ICPFLOW_STOP_REFERENCE;
constexpr int ICPFLOW_STOP_PER_PAIR_ = ICPFLOW_STOP_PER_PAIR;

constexpr int kPartial = 8;       // doubles per (job, query block) partial record
constexpr int kMaxIterCap = 1024; // upper bound on max_iterations
constexpr int kCand = 6;          // 5 peaks + the zero translation (utils_hist.py:83)
constexpr int kTopK = 5;          // utils_hist.py:21
constexpr int kNmsKernel = 11;    // utils_hist.py:21

struct IcpState {
    float R[9];
    float T[3];
    float rmse;
    int active;
    int iters;
    float s;   // scale (1 unless estimate_scale, utils_icp_pytorch3d.py:364-374)
};

struct IcpCtrl {
    int done;
    int iters;
    int error;     // a team member waited too long for its peers (results are poisoned with NaN)
    int ticket;    // persistent launches: pairs handed out beyond the first gridDim.x (icp_kernel)
    int finished;  // a launch that is drained for a second one (icp_split_kernel): pairs that are through
    int pad[3];
    int notconv[kMaxIterCap];
    // speculative single-launch mode: per iteration, pairs arrived (low 32 bits) and pairs not
    // converged (high 32 bits), updated by ONE 64-bit atomic per pair so both are read consistently
    unsigned long long tally[kMaxIterCap];
};

// Helpers (icp.hip, persistent launches): a workgroup that finds no pair left to start takes whole PASSES (BLOCK
// consecutive sorted queries) of a pair another workgroup is still iterating on.  One HelpPair per pair, one tag / owner
// word / outbox per workgroup of the grid; every word is read and written with relaxed agent-scope atomics.
// Encodings: E(n) = n + 1 for an iteration number n, so that 0 means "nothing yet"; -1 = the pair is finished.
struct HelpPair {          // 64 bytes, cleared with the control block at the start of every call
    int iter;              // E(iteration the owner is running or about to run); 0 = not started, -1 = finished
    int nclaim;            // helper slots handed out (helpers: atomic add)
    int epoch;             // E(n): state[(n + 1) & 1] holds (R, T) of iteration n; 0 = none published; -1 = finished
    int passes;            // passes of this pair (slot j takes pass `passes - 1 - j`; the owner always keeps pass 0)
    int from[4];           // helper j: (its workgroup << 8) | E(first iteration it takes part in); 0 = not announced
    int pad[8];
};
constexpr int kHelpSlots = 3;
constexpr int kHelpMaxEpoch = 255;   // the epoch E = iteration + 1 travels in the low 8 bits of HelpPair::from and IcpHelp::tag
constexpr int kHelpMaxWG = 1024;              // workgroups of a persistent grid, at most
constexpr int kHelpOutStride = 16 * 18;       // doubles per outbox: the moment sums of one pass, <= 16 waves x 18
struct IcpHelp {
    HelpPair *pair = nullptr;   // [B]            (zeroed per call)
    int *tag = nullptr;         // [kHelpMaxWG]   (zeroed per call) (pair << 8) | E(iteration) of the outbox's content
    int *owner = nullptr;       // [kHelpMaxWG]   (zeroed per call) pair + 1 the workgroup is iterating on as its owner
    float *state = nullptr;     // [B, 2, 16]     (R row-major 9, T 3) double-buffered by epoch parity
    double *out = nullptr;      // [kHelpMaxWG, kHelpOutStride]
};
// bytes of the control block + everything cleared with it (IcpCtrl, HelpPair[B], tag, owner)
inline size_t icp_ctrl_bytes(int B);

// Teams (icp.hip): several workgroups share one LARGE pair.  Every member owns a contiguous range
// of the sorted moving cloud, the 18 moments of an iteration are exchanged through `mom`, and each
// member solves for the same (R, T) redundantly -- one exchange per iteration, nobody broadcasts.
#ifndef ICPFLOW_MAX_TEAM
#define ICPFLOW_MAX_TEAM 16
#endif
constexpr int kMaxTeam = ICPFLOW_MAX_TEAM;      // workgroups per pair, at most
constexpr int kTeamStride = 20;   // doubles per (pair, parity, member): 18 moments, stop flag
constexpr int kShareParts = 8;    // parts of a shared window scan, at most
struct IcpTeam {
    int32_t *wgPair;        // [maxWG] pair served by workgroup w, -1 = none
    int32_t *wgRank;        // [maxWG] rank inside the team
    int32_t *teamSize;      // [B]
    int32_t *next;          // [B] single-pass pairs chained on one workgroup: the pair served after pair b, -1 = none
    unsigned int *arrived;  // [B] members that have published (monotone over the launch)
    double *mom;            // [B, 2, kMaxTeam, kTeamStride]
    int maxWG;
};

inline size_t icp_ctrl_bytes(int B)
{
    return sizeof(IcpCtrl) + (size_t)B * sizeof(HelpPair) + 2 * (size_t)kHelpMaxWG * sizeof(int);
}
inline IcpHelp icp_help_carve(IcpCtrl *ctrl, int B, float *state, double *out)
{
    IcpHelp h;
    h.pair = reinterpret_cast<HelpPair *>(ctrl + 1);
    h.tag = reinterpret_cast<int *>(h.pair + B);
    h.owner = h.tag + kHelpMaxWG;
    h.state = state;
    h.out = out;
    return h;
}

constexpr int kHistIters = 128;   // iterations of per-pair history kept in the workspace
constexpr int kHistStride = 16;   // floats per (iteration, pair): R (9), T (3), rmse, scale, gated correspondences

// hist.hip
void launch_count_valid(const float *pts, int B, int N, int32_t *len, hipStream_t s);
// (also zeroes up to two scratch buffers the later kernels of the call want cleared: saves their memsets)
void launch_count_pair(const float *A, const float *C, int B, int N, int32_t *lenA, int32_t *lenC, uint8_t *swap,
                       hipStream_t s, void *zero0 = nullptr, size_t bytes0 = 0, void *zero1 = nullptr, size_t bytes1 = 0,
                       float *boxes = nullptr);   // boxes: [B, 24] bounding boxes for the sorts of long clouds (votekey.hpp), or NULL
hipError_t launch_hist_vote(const float *X, const float *Y, int B, int NX, int NY,
                            const float mins[3], const float maxs[3], const int lens[3],
                            const float *ex, const float *ey, const float *ez,
                            const uint8_t *swap, uint32_t *bins_u32, hipStream_t s);
// count_pair folded into the vote's sort (zsort_kernel, N <= kChunkSortMinN): the sort counts the valid rows of both
// clouds itself, writes nX / nY / swapOut (swapOut[b] = nY > nX: Y is the src role) and clears the two scratch buffers.
struct PairCountFuse {
    uint8_t *swapOut;
    void *zero0; size_t bytes0;
    void *zero1; size_t bytes1;
};
hipError_t launch_hist_vote_sorted(const float *X, const float *Y, int32_t *nX, int32_t *nY,
                                   int B, int N, const int lens[3], const float *ex, const float *ey,
                                   const float *ez, const uint8_t *swap, float *sortX, float *sortY,
                                   uint32_t *bins_u32, float *ckey, int *cidx, float *keyRec, hipStream_t s,
                                   const PairCountFuse *fuse = nullptr, bool sideBusy = false, const float *boxes = nullptr,
                                   int32_t *work = nullptr, size_t workCap = 0, int32_t *orderOut = nullptr, bool *planned = nullptr);   // work: scratch of the vote's work list (ints), or NULL
size_t vote_work_capacity(int B, int N);   // ints the work list of a batch can take (0: such batches never use one)
// sort.hip: several workgroups per long cloud; same outputs as zsort_kernel / sort_clouds_kernel
constexpr int kChunkSortMinN = 4096;
constexpr int kPairBoxStride = 24;   // floats per pair of count_pair's boxes (votekey.hpp)
int chunk_sort_length(int N);
hipError_t launch_zsort_chunked(const float *P, const float *Q, const int32_t *nP, const int32_t *nQ, int B, int N,
                                float *outP, float *outQ, uint32_t *bins, int L, float *ckey, int *cidx,
                                const float *ez, int len_z, float *keyRec, hipStream_t s, const float *boxes = nullptr);
hipError_t launch_sort_clouds_chunked(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                                      const uint8_t *swap, const float *prePose, int B, int N, int32_t *axisOut,
                                      float *Xs, float *Ys, float *Ysoa, float *Xsoa, float *ckey, int *cidx,
                                      hipStream_t s, const float *boxes = nullptr, int dirKeys = 0);
hipError_t launch_u32_to_f32(const uint32_t *in, float *out, size_t n, hipStream_t s);
hipError_t launch_hist_peaks_f32(const float *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s);
// optional fused decode of the peaks into candidate translations [B, k + 1, 3] (zero translation last)
struct PeakDecode {
    const float *ex = nullptr, *ey = nullptr, *ez = nullptr;
    float shift = 0.f;
    float *cand = nullptr;
};
hipError_t launch_hist_peaks_u32(const uint32_t *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s, PeakDecode dec = PeakDecode{});

// nn.hip
struct GridScratch;
hipError_t launch_sweep_pair_table(const int32_t *lenA, const int32_t *lenC, const uint8_t *swap, const int32_t *pairOrder,
                                   int B, int N, int32_t *tab, hipStream_t s);
int scan_qblocks(int maxRows, int batch);
int sweep_qblocks(int maxRows);
hipError_t launch_sweep_eval(const GridScratch *grid, const int32_t *len1, const int32_t *len2, int B, int N,
                             const float *pose, float thres, float *srcT, double *partial, hipStream_t s, const uint8_t *swap = nullptr,
                             const uint8_t *active = nullptr);
struct PoseSource;   // posefuse.hpp
// poseFinal == NULL: the final pose of every pair is composed inside the kernel from `fused`
hipError_t launch_sweep_check(const GridScratch *grid, const float *X, const float *Y, const int32_t *lenA,
                              const int32_t *lenC, const uint8_t *swap, int B, int N, const float *poseInit,
                              const float *poseFinal, double *partial, hipStream_t s, const PoseSource *fused = nullptr,
                              const uint8_t *active = nullptr, const double *initSum = nullptr);
// after launch_sort_clouds_soa on the same stream: the occupancy grids of both sorted clouds (GridScratch.occHdr / occBits)
hipError_t launch_occupancy(const GridScratch *grid, int B, int N, hipStream_t s);
hipError_t launch_sweep_score(const GridScratch *grid, const int32_t *lenA, const int32_t *lenC, const uint8_t *swap,
                              int B, int N, const float *cand, double *partial, hipStream_t s);
hipError_t launch_sweep_score_pruned(const GridScratch *grid, const int32_t *lenA, const int32_t *lenC,
                                     const uint8_t *swap, int B, int N, const float *cand, double *partial,
                                     double *accum, hipStream_t s);
hipError_t launch_scan_score(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                             const uint8_t *swap, int B, int N, const float *cand, double *partial,
                             hipStream_t s);
int score_qblocks(int maxRows);
hipError_t launch_scan_score_pruned(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                                    const uint8_t *swap, int B, int N, const float *cand, double *partial,
                                    double *accum, hipStream_t s, bool accumCleared = false);
hipError_t launch_scan_check(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                             const uint8_t *swap, int B, int N, const float *poseInit,
                             const float *poseFinal, double *partial, hipStream_t s);
hipError_t launch_scan_eval(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                            int B, int N, const float *pose, float thres, double *partial, hipStream_t s);
hipError_t launch_scan_nn(const float *Qp, const float *Tp, int B, int NQ, int NT, int strideQ,
                          int strideT, const int32_t *lenQ, const int32_t *lenT, int sqrt_dist,
                          int64_t *idx, float *dist, hipStream_t s);

// icp.hip
constexpr int kOccWords = 1024;   // 32 768 cells per (pair, role) occupancy grid (nn.hip)
#ifndef ICPFLOW_OCC_RINGS
#define ICPFLOW_OCC_RINGS 3
#endif
constexpr int kOccRings = ICPFLOW_OCC_RINGS;   // ... in that many planes: the cloud dilated once, twice, ...
constexpr int kSweepShareSlots = 16;   // (query blocks x shares) of a job whose blocks split ALL targets between them (nn.hip)
struct GridScratch {   // scratch of the exact gated NN searches of the ICP loop (see icp.hip)
    int mode;          // 2 = hashed grid, 3 = sorted sweep
    int H;             // grid: buckets per pair, power of two >= 2N
    float *origin;     // grid: [B,4]
    int32_t *start;    // grid: [B,H+1]
    int32_t *cursor;   // grid: [B,H]
    float *pts;        // grid: fixed cloud sorted by bucket; sweep: fixed cloud sorted by axis [B,N,4]
    float *sortX;      // sweep: moving cloud sorted by axis, pre-pose applied [B,N,4]
    float *sortYsoa;   // sweep: fixed cloud sorted, x[] y[] z[] padded with +inf [B,3,NP16]
    float *sortXsoa;   // scoring sweep: moving cloud sorted (no pre-pose), same layout
    int32_t *axis;     // sweep: [B] the pair's sort key (sortdir.hpp): 0 .. 2 a coordinate, 3 .. a horizontal direction
    int dirKeys;       // the sorts of this call may choose direction keys (ICPFLOW_OPT_NO_DIR_KEYS: coordinates only, as before round 6)
    float *ckey;       // long clouds (N > kChunkSortMinN): chunk-sorted keys / rows of the multi-workgroup sort
    int *cidx;         //   [B,2,chunk_sort_length(N)] each (sort.hip), else NULL
    const float *pairBox;  // long clouds: [B, 24] boxes left by count_pair for THESE clouds, lengths and roles (NULL: the sorts look)
    const int32_t *pairTab;    // [B, 4] (nn.hip: sweep_pair_table_kernel) for THESE lengths, roles and pair order, or NULL
    const int32_t *pairOrder;  // [B] pairs by decreasing size (vote_plan_kernel ran for THIS batch), or NULL: the sweeps take the pairs as they come
    float *shareBest;  // sweeps (nn.hip): [B*12, kSweepShareSlots, 256] partial minima of small-against-long jobs shared by several blocks, or NULL
    int *shareCount;   //   [B*12, <= kSweepShareSlots] blocks delivered (zero between launches: the last block to deliver resets its counter)
    int shareCountClean;   //   host-side: this call has cleared shareCount already (api.hip: with scoreAccum) -- no memset per launch
    // scoring sweeps with branch and bound (nn.hip occ_build_kernel): per (pair, role) a DILATED occupancy grid of the sorted cloud --
    // bit set: some point of the cloud lies in the cell or in one of its 26 neighbours -- from which a scan gets a lower bound of its
    // whole sum before it has evaluated a single target.  occHdr [B,2,8]: origin x y z, 1 / h, 0.98 h, nx, ny, nz (ints as bits);
    // occBits [B,2,kOccRings,kOccWords]; occReady: host-side, built for THIS call's sort (launch_occupancy)
    float *occHdr;
    uint32_t *occBits;
    int occReady;
    int *sweepTicket;   // pruned scoring in two launches (nn.hip): the number of listed scans, cleared with the scoring scratch (api.hip), or NULL: one plain grid
    int *scoreList;     //   [B * 10] the scans that go on behind the deciding launch
    int presorted;     // sortX / pts / sortYsoa / axis already hold both clouds sorted WITHOUT the pre-pose
                       // (scoring sweep ran on this batch): the ICP applies the pre-pose when it loads
};
int grid_buckets(int N);
// per-call switches of the ICP launch (icpflow_options_t, include/icpflow_hip.h); nothing process-global
struct LaunchProfile;   // icp.hip: HIP-event recorder behind icpflow_profile_t
struct IcpOpts {
    int arith = 0;                 // ICPFLOW_ARITH_*
    bool teams = true;             // several workgroups per large pair when the batch leaves CUs idle
    bool speculative = true;       // batch-global stop in ONE launch (false: one launch per iteration)
    bool adaptiveWindows = true;   // sorted sweep: per-query windows from the previous iteration's neighbours
    bool persistent = true;        // batches larger than the GPU: a grid as large as the GPU, further pairs by ticket
    bool helpers = true;           // ... whose workgroups, once the tickets are gone, take passes of pairs still iterating
    IcpHelp help{};                // scratch of the helpers (NULL pointers: no helpers)
    const float *initR = nullptr;  // [B,3,3] / [B,3]: state before the first iteration (init_transform), or identity
    const float *initT = nullptr;
    bool allowReflection = false;  // R = U V^T whatever its determinant (utils_icp_pytorch3d.py:354-362)
    bool estimateScale = false;    // s = trace(E S) / Xcov (:364-374); Xt = s X R + T
    const float *initS = nullptr;  // [B] scale of the initial transform (with initR / initT), NULL = 1
    LaunchProfile *profile = nullptr;
    bool ctrlCleared = false;      // the caller's count_pair launch already zeroed *ctrl
    bool *historyPending = nullptr;   // non-NULL: do not launch the history epilogue; *historyPending = "the final
                                   // states are still in the per-iteration history" (the consumers resolve it)
    float *fp32Scratch = nullptr;  // ICPFLOW_ARITH_FP32_REFERENCE: [B,N,4] floats (neighbour and weight per point)
    bool teamPlanned = false;      // the caller has launched the team plan itself (launch_icp_team_plan, ordered before the ICP)
    const uint8_t *pairActive = nullptr;   // options.d_pair_active: pairs flagged 0 are not in the batch (speculative reference stop only)
    bool teamsHalfGpu = false;     // ICPFLOW_OPT_TEAMS_HALF_GPU: a team launch takes at most half of the CUs (two may run side by side)
    bool sharedScans = true;       // teams: the waves of a member share their long window scans (ICPFLOW_OPT_NO_SHARED_SCANS)
    bool twoLaunch = false;        // persistent grids with helpers: drained for a second launch of whole-CU workgroups (ICPFLOW_OPT_TWO_LAUNCH)
    int32_t *splitScratch = nullptr;   // [B + 64] ints: the second launch's pair list, its count and the floor of its rule search (NULL: one launch)
};
bool icp_teams_wanted(const IcpTeam *team, const IcpOpts &opts, const GridScratch *grid, int B, int N, int maxIter,
                      int stopMode, const float *history);
int icp_team_workgroups(const IcpOpts &opts);
void launch_icp_team_plan(const IcpTeam *team, const int32_t *lenX, const int32_t *lenY, const uint8_t *swap, int B, int N,
                          const IcpOpts &opts, hipStream_t s);
// icp_fp32.hip: the reference's fp32 operation order (study mode), batch-global stop via the history epilogue
hipError_t launch_icp_fp32ref(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                              const uint8_t *swap, const float *prePose, int B, int N, double thres, int maxIter,
                              double relThr, IcpState *state, IcpCtrl *ctrl, float *history, float *nnScratch,
                              hipStream_t s);
hipError_t launch_icp(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                      const uint8_t *swap, const float *prePose, int B, int N, double thres,
                      int maxIter, double relThr, int stopMode, IcpState *state, IcpCtrl *ctrl,
                      const GridScratch *grid, float *history, const IcpTeam *team, const IcpOpts &opts,
                      hipStream_t s);
hipError_t launch_sort_clouds_soa(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                                  const uint8_t *swap, int B, int N, const GridScratch *grid, hipStream_t s, int selfCount = 0);
LaunchProfile *profile_create(int capacity, hipError_t *err);
void profile_destroy(LaunchProfile *p);
hipError_t profile_collect(LaunchProfile *p, double *total_ms, int *launches);
// per-device caches (a process may drive several GPUs): CU count, and the opt-in of a kernel to more
// than the default dynamic LDS, which HIP keeps per device
int device_cus();
void ensure_dynamic_lds(const void *func, int bytes, std::atomic<unsigned long long> *doneMask);
hipError_t launch_icp_export(IcpState *state, IcpCtrl *ctrl, int B, int stopMode, float *R,
                             float *T, float *rmse, int32_t *iters, int32_t *converged, hipStream_t s,
                             float *scale = nullptr);
// the batch rule over the pairs flagged in `active`, from the history of a launch that iterated all of them: rewrites the tallies
hipError_t launch_icp_retally(IcpCtrl *ctrl, const float *history, const uint8_t *active, int B, int maxIter, double relThr,
                              hipStream_t s);
hipError_t launch_icp_resolve_history(IcpState *state, IcpCtrl *ctrl, const float *history, int B, int maxIter,
                                      hipStream_t s);

// pose.hip
hipError_t launch_score_pick(const double *partial, int qblocks, const int32_t *lenA,
                             const int32_t *lenC, const uint8_t *swap, const float *cand, int B,
                             float *Tinit, hipStream_t s, double *initSum = nullptr);
hipError_t launch_compose(const IcpState *state, const float *init, int B, float *M, hipStream_t s,
                          const IcpCtrl *ctrl = nullptr, int32_t *iters = nullptr);
// M == NULL: the composed pose comes from `fused` (and the iteration count is written to *iters)
hipError_t launch_select(const double *partial, int qblocks, const int32_t *lenA, const int32_t *lenC,
                         const uint8_t *swap, const float *init, const float *M, int B, int invertSwapped,
                         float *out, hipStream_t s, const PoseSource *fused = nullptr, int32_t *iters = nullptr,
                         const double *initSum = nullptr, const uint8_t *active = nullptr);
hipError_t launch_eval_epilogue(const double *partial, int qblocks, const int32_t *len1,
                                const int32_t *len2, const float *T, int B, float *errors,
                                float *inliers, float *ratios, float *ious, float *translations,
                                float *rotations, hipStream_t s);
hipError_t launch_vote_quotient_probe(const float *a, int n, float mn, float mx, float *fast, float *ieee,
                                      hipStream_t s);
hipError_t launch_gather_pad(const float *points, const int32_t *rows, int B, int N, float *out, hipStream_t s);
hipError_t launch_gather_segments(const float *points, const int64_t *order, const int64_t *seg, const int32_t *perm,
                                  int B, int N, float *out, hipStream_t s);
hipError_t launch_cluster_stats(const float *points, const int64_t *order, const int64_t *start,
                                const int64_t *count, const float *labels, int L, float *mean, float *extent,
                                hipStream_t s);
hipError_t launch_flow_rigid(const float *points, const float *labels, int N, const float *pairRows, int pairStride,
                             const float *T, int P, const float *pose, float *flow, hipStream_t s);
hipError_t launch_transform_points(const float *xyz, const float *pose, int B, int N, float *out,
                                   hipStream_t s);


// table.hip: the cluster table of a labelled cloud (rows sorted by label, distinct labels, per-cluster statistics)
hipError_t cluster_table_workspace_bytes(int M, int Lmax, size_t *bytes);
hipError_t cluster_table_pair_workspace_bytes(int MA, int MB, int Lmax, size_t *bytes);
hipError_t launch_cluster_table_pair(const float *pointsA, const float *labelsA, int MA, int64_t *orderA, double *tableA,
                                     int32_t *numA, const float *pointsB, const float *labelsB, int MB, int64_t *orderB,
                                     double *tableB, int32_t *numB, int Lmax, void *ws, size_t wsBytes, bool *wsTooSmall,
                                     hipStream_t s);
hipError_t launch_cluster_table(const float *points, const float *labels, int M, int64_t *order, double *table, int Lmax,
                                int32_t *num, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s);

// cluster.hip: DBSCAN of a frame pair's points (labels int32 [n]: cluster id, -1 noise, -2 masked out)
hipError_t dbscan_workspace_bytes(int n, size_t *bytes);
hipError_t launch_dbscan(const float *pts, int stride, const uint8_t *mask, int n, double eps, int minPoints,
                         int32_t *labels, int32_t *counts, int32_t *numClusters, void *ws, size_t wsBytes,
                         bool *wsTooSmall, hipStream_t s);

// hdbscan.hip: core distances + minimum spanning tree of the mutual-reachability graph
hipError_t hdbscan_workspace_bytes(int n, size_t *bytes);
hipError_t launch_hdbscan_mst(const float *pts, int stride, const uint8_t *mask, int n, int minSamples, double cell,
                              double *core2, int32_t *edgeA, int32_t *edgeB, double *edgeW2, int32_t *numEdges,
                              int32_t *numLive, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s);

// assoc.hip: the host half of an association stage as kernels (icpflow_assoc_assign / icpflow_assoc_collect)
hipError_t launch_assoc_assign(const float *r, const int32_t *si, const int32_t *di, int K, const uint8_t *active, int S, int D,
                               float tf, float iouMin, float rotMax, float errMax, int32_t *best, int K2, const int32_t *si2,
                               const int32_t *di2, int64_t *seg2, uint8_t *active2, hipStream_t s);
hipError_t launch_assoc_collect(const int32_t *best1, const float *r1, const int32_t *si1, const int32_t *di1, int K1,
                                const int32_t *best2, const float *r2, const int32_t *si2, const int32_t *di2, int K2,
                                const double *srcTable, const double *dstTable, int stride, int S, int cap, float *rows,
                                float *T, int32_t *count, hipStream_t s);
int assoc_max_rows();

}  // namespace icpflow
