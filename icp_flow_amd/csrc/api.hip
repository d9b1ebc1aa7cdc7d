// api.hip -- the C ABI of libicpflow_hip.so (include/icpflow_hip.h): argument checks,
// workspace carving and kernel sequencing.  No device synchronisation anywhere.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "kernels.hpp"
#include "posefuse.hpp"

using namespace icpflow;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hipfail(hipError_t e, const char *what)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return (int)e;
}

}  // namespace

namespace icpflow {
// icpflow_last_error for the entry points that live in other files (frame.hip)
int report_error(int code, const char *message)
{
    snprintf(g_err, sizeof(g_err), "%s", message);
    return code;
}
}  // namespace icpflow

namespace {

#define ICPFLOW_TRY(expr)                                  \
    do {                                                   \
        hipError_t e__ = (expr);                           \
        if (e__ != hipSuccess) return hipfail(e__, #expr); \
    } while (0)

// The options of one call (icpflow_options_t, validated): nothing here outlives the call.
struct Opts {
    int search = ICPFLOW_SEARCH_AUTO;
    int arith = ICPFLOW_ARITH_FP64;
    unsigned flags = 0u;
    LaunchProfile *profile = nullptr;
    uint32_t *voteBins = nullptr;
    const float *initR = nullptr, *initT = nullptr;
    float *history = nullptr;
    bool allowReflection = false;
    bool estimateScale = false;
    float *scaleOut = nullptr;
    const float *initS = nullptr;
    const uint8_t *pairActive = nullptr;
    bool on(unsigned offFlag) const { return (flags & offFlag) == 0u; }
    IcpOpts icp(float *scratch) const
    {
        IcpOpts o;
        o.fp32Scratch = scratch;
        o.ctrlCleared = true;   // every fused entry point clears the control block in its count_pair launch
        o.arith = arith;
        o.teams = on(ICPFLOW_OPT_NO_TEAMS);
        o.speculative = on(ICPFLOW_OPT_NO_SPECULATIVE);
        o.adaptiveWindows = on(ICPFLOW_OPT_NO_ADAPTIVE_WINDOWS);
        o.persistent = on(ICPFLOW_OPT_NO_PERSISTENT);
        o.helpers = on(ICPFLOW_OPT_NO_HELPERS);
        o.teamsHalfGpu = (flags & ICPFLOW_OPT_TEAMS_HALF_GPU) != 0u;
        o.sharedScans = on(ICPFLOW_OPT_NO_SHARED_SCANS);
        o.twoLaunch = (flags & ICPFLOW_OPT_TWO_LAUNCH) != 0u;
        o.pairActive = pairActive;
        o.profile = profile;
        return o;
    }
};

// the scoring sweep prunes by the largest NN distance inside a wave: it pays on large clusters (real
// data, hundreds of queries per metre along the sort axis), not on ~1000-point vehicles
#ifndef ICPFLOW_SCORE_SWEEP_MIN_N
#define ICPFLOW_SCORE_SWEEP_MIN_N 2048
#endif
constexpr int kScoreSweepMinN = ICPFLOW_SCORE_SWEEP_MIN_N;
constexpr int kMaxSortN = 16384;   // bitonic sort of (key, index) pairs in 128 KiB of LDS
// match_eval as sorted sweeps (its own sort included) against the all-pairs scans, measured (round 3, ms per batch, scan ->
// sweep): 1024 x 2048 0.998 -> 0.368, 1024 x 1500 0.713 -> 0.286, 256 x 1024 0.089 -> 0.067, ragged 600 x 2048 0.213 -> 0.170;
// the scans keep the small batches, where the sort is pure latency: 256 x 512 0.037 vs 0.045, ragged 128 x 1024 0.044 vs 0.056
inline bool eval_by_sweep(int B, int N, const Opts &o)
{
    if (N > kMaxSortN || !o.on(ICPFLOW_OPT_NO_EVAL_SWEEP)) return false;
    return N > kScoreSweepMinN || (N >= 1024 && (long long)B * N >= (1LL << 18));
}
// ... with the branch and bound of the all-pairs scoring and the clouds sorted anyway (hist_icp sorts them for the
// ICP on the side stream) the sweep wins from ~1000 points on (config 2: 171 -> 153 us, config 4's shard: 1.50 -> 0.94 ms)
#ifndef ICPFLOW_SCORE_SWEEP_MIN_N_SORTED
#define ICPFLOW_SCORE_SWEEP_MIN_N_SORTED 512
#endif
constexpr int kScoreSweepMinNSorted = ICPFLOW_SCORE_SWEEP_MIN_N_SORTED;
inline bool score_by_sweep(int N, bool sortedAnyway, const Opts &o)
{
    const int minN = (sortedAnyway && o.on(ICPFLOW_OPT_NO_SCORE_PRUNE)) ? kScoreSweepMinNSorted : kScoreSweepMinN;
    return N > minN && N <= kMaxSortN && o.on(ICPFLOW_OPT_NO_SCORE_SWEEP);
}
constexpr size_t kAlign = 256;
size_t up(size_t n) { return (n + kAlign - 1) / kAlign * kAlign; }

// One carve of the caller's workspace; the same layout backs workspace_bytes().
struct Workspace {
    int32_t *lenA = nullptr, *lenC = nullptr;
    uint8_t *swap = nullptr;
    uint32_t *bins = nullptr, *volA = nullptr, *volB = nullptr;
    float *peakVotes = nullptr;
    int64_t *peakIdx = nullptr;
    float *cand = nullptr;
    double *scoreAccum = nullptr;
    double *partial = nullptr;
    float *Tinit = nullptr, *M = nullptr;
    IcpState *state = nullptr;
    IcpCtrl *ctrl = nullptr;
    float *helpState = nullptr;
    double *helpOut = nullptr;
    GridScratch grid{};
    float *history = nullptr;
    float *zsortA = nullptr, *zsortC = nullptr;   // z-sorted copies of both clouds (vote)
    float *zckey = nullptr;
    int32_t *voteWork = nullptr;   // work list of the sorted vote on wide ragged batches (hist.hip: vote_plan_kernel)
    size_t voteWorkCap = 0;
    int32_t *pairTab = nullptr;    // sweeps: the pair table of workgroups without rows (grid.pairTab points here once it is written)
    int32_t *pairOrder = nullptr;  // ... and the pairs by decreasing size (grid.pairOrder points here once the plan has run)
    float *pairBox = nullptr;   // long clouds: boxes by count_pair (grid.pairBox points here once they are written)
    float *voteKey = nullptr;   // per-pair sort-key parameters of the vote (votekey.hpp)
    int *zcidx = nullptr;
    IcpTeam team{};
    int32_t *icpSplit = nullptr;   // [B + 64] the pair list (and its count) of the second of two ICP launches (icp.hip: icp_split_kernel)
    // host-side note of THIS call: score_pick_kernel has left the forward totals of the picked candidates in scoreAccum[0 .. B)
    // (the scoring ran as sweeps over the sort the check sweep will use): the roll-back check scans under the final pose only
    bool initSumValid = false;
    int *ticketScratch = nullptr;
    size_t accumBytes = 0;   // scoreAccum and, where the sweeps share jobs between blocks, grid.shareCount behind it: cleared together
    size_t bytes = 0;
    static bool shareScratch(int B, int N) { return N >= 2048 && (size_t)B * 12 * kSweepShareSlots * 256 * 4 <= ((size_t)64 << 20); }

    Workspace(void *base, int B, int N, size_t L)
    {
        size_t off = 0;
        auto take = [&](size_t n) {
            void *p = base ? (void *)((char *)base + off) : nullptr;
            off += up(n);
            return p;
        };
        const size_t b = (size_t)B;
        lenA = (int32_t *)take(b * 4);
        lenC = (int32_t *)take(b * 4);
        swap = (uint8_t *)take(b);
        bins = (uint32_t *)take(b * L * 4);
        volA = (uint32_t *)take(b * L * 4);
        volB = (uint32_t *)take(b * L * 4);
        peakVotes = (float *)take(b * kTopK * 4);
        peakIdx = (int64_t *)take(b * kTopK * 8);
        cand = (float *)take(b * kCand * 3 * 4);
        {
            size_t qb = (size_t)(scan_qblocks(N, B) > sweep_qblocks(N) ? scan_qblocks(N, B) : sweep_qblocks(N));
            if ((size_t)score_qblocks(N) > qb) qb = (size_t)score_qblocks(N);
            partial = (double *)take(b * 12 * qb * kPartial * 8);
            scoreAccum = (double *)take(b * 12 * 8);
            // (the sweeps' delivery counters right behind it: the clear at the start of a registration covers both -- accumBytes --,
            // and every sweep launch leaves its counters at zero again, so that none of them needs a memset of its own)
            ticketScratch = (int *)take(256);   // (the count of the scoring's listed scans: grid.sweepTicket once a call has cleared it)
            if (shareScratch(B, N)) grid.shareCount = (int *)take(b * 12 * kSweepShareSlots * 4);
            accumBytes = up(b * 12 * 8) + up(256) + (shareScratch(B, N) ? up(b * 12 * kSweepShareSlots * 4) : 0);
        }
        Tinit = (float *)take(b * 16 * 4);
        M = (float *)take(b * 16 * 4);
        state = (IcpState *)take(b * sizeof(IcpState));
        ctrl = (IcpCtrl *)take(icp_ctrl_bytes(B));   // + the helpers' per-pair words and tags, cleared with it
        helpState = (float *)take(b * 32 * 4);
        helpOut = (double *)take((size_t)kHelpMaxWG * kHelpOutStride * 8);
        grid.H = grid_buckets(N);
        grid.origin = (float *)take(b * 4 * 4);
        grid.start = (int32_t *)take(b * ((size_t)grid.H + 1) * 4);
        grid.cursor = (int32_t *)take(b * (size_t)grid.H * 4);
        grid.pts = (float *)take(b * (size_t)N * 16);
        grid.sortX = (float *)take(b * (size_t)N * 16);
        grid.sortYsoa = (float *)take(b * 3 * (size_t)((N + 15) / 16 * 16) * 4 + 256);  // + prefetch slack
        grid.sortXsoa = (float *)take(b * 3 * (size_t)((N + 15) / 16 * 16) * 4 + 256);
        zsortA = (float *)take(b * (size_t)N * 16);
        zsortC = (float *)take(b * (size_t)N * 16);
        voteKey = (float *)take(b * 8 * 4);
        voteWorkCap = vote_work_capacity(B, N);
        if (voteWorkCap != 0) { voteWork = (int32_t *)take(voteWorkCap * 4); pairOrder = (int32_t *)take(b * 4); }
        if (N > kChunkSortMinN) {   // scratch of the multi-workgroup sorts (one set per concurrent sort)
            const size_t cs = b * 2 * (size_t)chunk_sort_length(N) * 4;
            grid.ckey = (float *)take(cs);
            grid.cidx = (int *)take(cs);
            zckey = (float *)take(cs);
            zcidx = (int *)take(cs);
            pairBox = (float *)take(b * kPairBoxStride * 4);
        }
        grid.axis = (int32_t *)take(b * 4);
        grid.scoreList = (int *)take(b * 12 * 4);
        // (occupancy grids of both sorted clouds for the pre-bound of the scoring sweeps, nn.hip: 24 KiB per pair)
        if (N <= kMaxSortN) {
            grid.occHdr = (float *)take(b * 2 * 8 * 4);
            grid.occBits = (uint32_t *)take(b * 2 * (size_t)kOccRings * kOccWords * 4);
        }
        // (sweeps of a small cloud against a long one, shared by several blocks: nn.hip; only where the partial minima stay small)
        if (shareScratch(B, N)) {
            grid.shareBest = (float *)take(b * 12 * kSweepShareSlots * 256 * 4);
            if (B <= 32767 && N <= 255 * 256) pairTab = (int32_t *)take(b * 4 * 4);
        }
        history = (float *)take(b * (size_t)kHistIters * kHistStride * 4);
        team.maxWG = 1024;
        team.wgPair = (int32_t *)take((size_t)team.maxWG * 4);
        team.wgRank = (int32_t *)take((size_t)team.maxWG * 4);
        team.teamSize = (int32_t *)take(b * 4);
        team.next = (int32_t *)take(b * 4);
        team.arrived = (unsigned int *)take(b * 4);
        team.mom = (double *)take((b < 256 ? b : 256) * 2 * (size_t)kMaxTeam * kTeamStride * 8);
        icpSplit = (int32_t *)take((b + 64) * 4);
        bytes = off;
    }
};

const GridScratch *search_scratch(Workspace &w, int N, const Opts &o)
{
    int mode = o.search;
    if (o.arith == ICPFLOW_ARITH_FP32_REFERENCE) mode = 1;   // the study mode is written for the all-pairs search
    if (mode == 0) mode = (N >= 64 && N <= kMaxSortN) ? 3 : 1;
    if (mode == 3 && N > kMaxSortN) mode = 1;   // the bitonic sort holds (key, index) pairs in LDS
    if (mode == 1) return nullptr;
    w.grid.mode = mode;
    w.grid.dirKeys = o.on(ICPFLOW_OPT_NO_DIR_KEYS) ? 1 : 0;   // (the sorts of this call may pick a direction key: sortdir.hpp)
    return &w.grid;
}

// -> 0 and `o` filled, or an argument error
int parse_options(const char *fn, const icpflow_options_t *opt, Opts &o)
{
    if (opt == nullptr) return 0;
    if (opt->struct_size != sizeof(icpflow_options_t))
        return fail(ICPFLOW_E_ARG, "%s: options struct_size %zu, this library expects %zu", fn, opt->struct_size,
                    sizeof(icpflow_options_t));
    if (opt->icp_search < 0 || opt->icp_search > 3)
        return fail(ICPFLOW_E_ARG, "%s: options.icp_search must be 0..3 (got %d)", fn, opt->icp_search);
    if (opt->icp_arith != ICPFLOW_ARITH_FP64 && opt->icp_arith != ICPFLOW_ARITH_FP32_REFERENCE)
        return fail(ICPFLOW_E_ARG, "%s: options.icp_arith must be 0 or 1 (got %d)", fn, opt->icp_arith);
    if (opt->flags >> 19) return fail(ICPFLOW_E_ARG, "%s: unknown option flags 0x%x", fn, opt->flags);
    o.search = opt->icp_search;
    o.arith = opt->icp_arith;
    o.flags = opt->flags;
    o.profile = reinterpret_cast<LaunchProfile *>(opt->profile);
    o.voteBins = opt->d_vote_bins_u32;
    if ((opt->d_icp_init_R == nullptr) != (opt->d_icp_init_T == nullptr))
        return fail(ICPFLOW_E_ARG, "%s: options.d_icp_init_R and d_icp_init_T come together", fn);
    o.initR = opt->d_icp_init_R;
    o.initT = opt->d_icp_init_T;
    o.history = opt->d_icp_history;
    o.allowReflection = opt->icp_allow_reflection != 0;
    o.estimateScale = opt->icp_estimate_scale != 0;
    o.scaleOut = opt->d_icp_scale;
    o.initS = opt->d_icp_init_s;
    o.pairActive = opt->d_pair_active;
    if (o.initS != nullptr && o.initR == nullptr)
        return fail(ICPFLOW_E_ARG, "%s: options.d_icp_init_s comes with d_icp_init_R / d_icp_init_T", fn);
    return 0;
}

// options.d_pair_active: only the fused registrations honour it, and only where ONE speculative launch runs the batch rule
int check_pair_active(const char *fn, const Opts &o, bool fused, int maxIter, int stopMode)
{
    if (o.pairActive == nullptr) return 0;
    if (!fused) return fail(ICPFLOW_E_ARG, "%s: options.d_pair_active is honoured by icpflow_hist_icp / icpflow_hist_icp_eval only", fn);
    if (stopMode != ICPFLOW_STOP_REFERENCE || maxIter < 2 || maxIter > kHistIters || !o.on(ICPFLOW_OPT_NO_SPECULATIVE) ||
        o.arith != ICPFLOW_ARITH_FP64)
        return fail(ICPFLOW_E_ARG, "%s: options.d_pair_active needs the single-launch reference stop (2 <= max_iterations <= %d, "
                                   "fp64 arithmetic, speculation on)", fn, kHistIters);
    return 0;
}

int check_arith(const char *fn, const Opts &o, int maxIter, int stopMode)
{
    if (o.arith != ICPFLOW_ARITH_FP32_REFERENCE) return 0;
    if (stopMode != ICPFLOW_STOP_REFERENCE)
        return fail(ICPFLOW_E_ARG, "%s: ICPFLOW_ARITH_FP32_REFERENCE implements the reference's batch-global stop only", fn);
    if (maxIter > kHistIters)
        return fail(ICPFLOW_E_LIMIT, "%s: ICPFLOW_ARITH_FP32_REFERENCE keeps a per-iteration history: max_iterations <= %d",
                    fn, kHistIters);
    return 0;
}

int check_ws(void *ws, size_t have, size_t need)
{
    if (ws == nullptr || have < need)
        return fail(ICPFLOW_E_WORKSPACE, "workspace too small: have %zu bytes, need %zu", have, need);
    if (((uintptr_t)ws & 15) != 0) return fail(ICPFLOW_E_WORKSPACE, "workspace must be 16-byte aligned");
    return 0;
}

int check_batch(const char *fn, int B, int N)
{
    if (B <= 0 || N <= 0) return fail(ICPFLOW_E_ARG, "%s: B (%d) and N (%d) must be positive", fn, B, N);
    if ((long long)B * N > (1ll << 30)) return fail(ICPFLOW_E_LIMIT, "%s: B*N exceeds 2^30 rows", fn);
    return 0;
}

int check_hist_dims(const char *fn, int lx, int ly, int lz)
{
    if (lx <= 0 || ly <= 0 || lz <= 0)
        return fail(ICPFLOW_E_ARG, "%s: histogram lengths must be positive (%d,%d,%d)", fn, lx, ly, lz);
    if ((long long)lx * ly * lz > (1ll << 28)) return fail(ICPFLOW_E_LIMIT, "%s: histogram too large", fn);
    return 0;
}

// Fork / join on a private side stream: in hist_icp the axis sort of both clouds (input of the scoring
// sweep and of the ICP) depends only on the inputs, so it runs next to the vote / peaks chain instead of
// in front of it.  One side stream and event pair per host thread (entry points are re-entrant for
// distinct streams and workspaces); the join makes the side work part of the caller's stream order, so
// stream capture by the caller still sees one connected graph.
struct SideStream {
    hipStream_t stream = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    int device = -1;
    bool ok = false;
    void destroy()   // (on the device the objects were created on; errors are moot at this point)
    {
        if (stream) (void)hipStreamDestroy(stream);
        if (fork) (void)hipEventDestroy(fork);
        if (join) (void)hipEventDestroy(join);
        stream = nullptr; fork = join = nullptr; ok = false;
    }
    void create()
    {
        ok = hipGetDevice(&device) == hipSuccess &&
             hipStreamCreateWithFlags(&stream, hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&fork, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&join, hipEventDisableTiming) == hipSuccess;
    }
};

// One side stream per (host thread, caller stream): registrations issued on different streams from one thread -- the
// batches of icpflow_hist_icp_many -- must not queue their sorts behind each other.  A handful of slots, recycled
// round robin (a recycled slot keeps its stream: only the association changes).
SideStream &side_stream(hipStream_t caller)
{
    constexpr int kSlots = 6;
    struct Slot { SideStream s; hipStream_t caller = nullptr; bool used = false; };
    static thread_local Slot slots[kSlots];
    static thread_local int next = 0;
    static thread_local SideStream none;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { none.ok = false; return none; }
    for (int k = 0; k < kSlots; ++k)
        if (slots[k].used && slots[k].caller == caller && slots[k].s.device == dev) return slots[k].s;
    Slot &sl = slots[next];
    next = (next + 1) % kSlots;
    if (!sl.used || sl.s.device != dev) {   // first use of the slot on this thread, or the thread moved to another GPU
        sl.s.destroy();
        sl.s = SideStream{};
        sl.s.create();
    }
    sl.used = true;
    sl.caller = caller;
    return sl.s;
}

// Joins the side stream back into the caller's stream when a fused entry point leaves early through an
// error path: without it the side work would still be writing into the caller's workspace after the
// call has returned, and a capturing caller would be left with a dangling fork.
struct JoinGuard {
    hipStream_t s = nullptr;
    hipEvent_t join = nullptr;
    ~JoinGuard() { if (join != nullptr) (void)hipStreamWaitEvent(s, join, 0); }
    void joined() { join = nullptr; }
};

// shared tail of apply_icp / hist_icp: ICP from Tinit, compose, check, select
int run_icp_and_select(const float *src, const float *dst, Workspace &w, const uint8_t *swap,
                       const float *init, int B, int N, double thres, int maxIter, double relThr,
                       int stopMode, int invertSwapped, float *Tout, int32_t *iters, const Opts &o, hipStream_t s,
                       bool teamPlanned = false, int part = 0, int32_t *pending = nullptr, bool initSumValid = false)
{
    // part 0: everything.  part 1: the ICP launch only, with the whole batch iterating (no pair mask); *pending = 1 when the launch
    // left every pair's trajectory in the history (the single speculative launch with the fused finish), 0 when it could not be
    // split off (then part 2 runs everything).  part 2 with *pending: the batch rule over o.pairActive from that history
    // (launch_icp_retally), then roll-back check and select.
    const GridScratch *search = search_scratch(w, N, o);
    const bool sweepCheck = search != nullptr && search->mode == 3 && o.on(ICPFLOW_OPT_NO_CHECK_SWEEP);
    // fused finish: the roll-back check and the select kernel take the final pose of every pair straight from the
    // ICP's per-iteration history (or its state): no icp_resolve_history / compose launches in between
    bool historyPending = false;
    IcpOpts io = o.icp(w.grid.sortX);
    io.help = icp_help_carve(w.ctrl, B, w.helpState, w.helpOut);
    io.splitScratch = w.icpSplit;
    io.teamPlanned = teamPlanned;
    if (sweepCheck && o.arith == ICPFLOW_ARITH_FP64) io.historyPending = &historyPending;
    const bool splittable = sweepCheck && o.arith == ICPFLOW_ARITH_FP64 && stopMode == ICPFLOW_STOP_REFERENCE && w.history != nullptr &&
                            o.on(ICPFLOW_OPT_NO_SPECULATIVE) && maxIter > 1 && maxIter <= kHistIters;
    if (part == 1) {
        *pending = 0;
        if (!splittable) return 0;
        // Where the launch takes TEAMS the plan -- the order of a team's sums -- follows the pairs that are in the batch
        // (icp_team_plan_kernel counts a masked pair as empty), and the mask is not known yet: the ICP waits for part 2,
        // whose masked launch plans exactly as the serial path's does (the two modes could differ in rounding).
        if (icp_teams_wanted(&w.team, io, search, B, N, maxIter, stopMode, w.history)) return 0;
        io.pairActive = nullptr;
    }
    if (part == 2 && pending != nullptr && *pending != 0) {
        historyPending = true;
        if (o.pairActive != nullptr) ICPFLOW_TRY(launch_icp_retally(w.ctrl, w.history, o.pairActive, B, maxIter, relThr, s));
    } else {
        ICPFLOW_TRY(launch_icp(src, dst, w.lenA, w.lenC, swap, init, B, N, thres, maxIter, relThr, stopMode,
                               w.state, w.ctrl, search, w.history, &w.team, io, s));
        if (part == 1) {
            if (!historyPending) return fail(ICPFLOW_E_ARG, "icpflow_register_stage_begin: the ICP launch kept no history");
            *pending = 1;
            return 0;
        }
    }
    if (sweepCheck) {
        PoseSource ps{w.state, w.ctrl, historyPending ? w.history : nullptr, init, B, maxIter};
        // (hist_icp: the scan under the initial pose is the scoring's forward scan of the picked candidate -- score_pick_kernel,
        // sweep_scan_kernel -- as long as the clouds the check reads are the raw ones of that very sort)
        const double *initSum = (initSumValid && search->presorted && o.on(ICPFLOW_OPT_NO_CHECK_REUSE)) ? w.scoreAccum : nullptr;
        ICPFLOW_TRY(launch_sweep_check(search, src, dst, w.lenA, w.lenC, swap, B, N, init, nullptr, w.partial, s, &ps, o.pairActive,
                                       initSum));
        ICPFLOW_TRY(launch_select(w.partial, sweep_qblocks(N), w.lenA, w.lenC, swap, init, nullptr, B, invertSwapped,
                                  Tout, s, &ps, iters, initSum, o.pairActive));
        return 0;
    }
    ICPFLOW_TRY(launch_compose(w.state, init, B, w.M, s, w.ctrl, iters));   // also reports the iteration count
    {   // roll-back check: the all-pairs scan
        ICPFLOW_TRY(launch_scan_check(src, dst, w.lenA, w.lenC, swap, B, N, init, w.M, w.partial, s));
        ICPFLOW_TRY(launch_select(w.partial, scan_qblocks(N, B), w.lenA, w.lenC, swap, init, w.M, B,
                                  invertSwapped, Tout, s));
    }
    return 0;
}

// joinBefore (hist_icp): event after which the side stream has both clouds sorted (w.grid.presorted)
int run_init_pose(const float *src, const float *dst, Workspace &w, const uint8_t *swap, int B,
                  int N, const float *ex, int lx, const float *ey, int ly, const float *ez, int lz,
                  float shift, float *Tout, const Opts &o, hipStream_t s, hipEvent_t joinBefore = nullptr,
                  const PairCountFuse *countFuse = nullptr, bool sideBusy = false)
{
    const int lens[3] = {lx, ly, lz};
    // vote with X = dst role, Y = src role (utils_hist.py:69); z-sorted sweep while the sort fits LDS
    // (N <= 16384), all-pairs otherwise -- identical bins either way
    if (N <= kMaxSortN && o.on(ICPFLOW_OPT_NO_SORTED_VOTE)) {
        bool planned = false;
        ICPFLOW_TRY(launch_hist_vote_sorted(dst, src, w.lenC, w.lenA, B, N, lens, ex, ey, ez, swap, w.zsortC,
                                            w.zsortA, w.bins, w.zckey, w.zcidx, w.voteKey, s, countFuse, sideBusy,
                                            w.grid.pairBox, o.on(ICPFLOW_OPT_NO_VOTE_LIST) ? w.voteWork : nullptr, w.voteWorkCap,
                                            w.pairOrder, &planned));
        if (planned) w.grid.pairOrder = w.pairOrder;   // (the sweeps of this call take the pairs largest first)
    } else
        ICPFLOW_TRY(launch_hist_vote(dst, src, B, N, N, nullptr, nullptr, lens, ex, ey, ez, swap, w.bins, s));
    if (o.voteBins != nullptr)   // debug export of the bins the peak search is about to read
        ICPFLOW_TRY(hipMemcpyAsync(o.voteBins, w.bins, (size_t)B * lx * ly * lz * sizeof(uint32_t),
                                   hipMemcpyDeviceToDevice, s));
    // the sweeps' pair table (nn.hip): for the lengths, roles and pair order the vote has just fixed; batches whose sweeps take the
    // sharing instantiation (<= 700 pairs of width >= 2048)
    if (w.pairTab != nullptr && B <= 700 && N <= kMaxSortN && o.on(ICPFLOW_OPT_NO_SORTED_VOTE)) {
        ICPFLOW_TRY(launch_sweep_pair_table(w.lenA, w.lenC, swap, w.grid.pairOrder, B, N, w.pairTab, s));
        w.grid.pairTab = w.pairTab;
    }
    PeakDecode dec;
    dec.ex = ex; dec.ey = ey; dec.ez = ez; dec.shift = shift; dec.cand = w.cand;   // peaks -> 6 candidate translations
    ICPFLOW_TRY(launch_hist_peaks_u32(w.bins, B, lx, ly, lz, kTopK, kNmsKernel, w.volA, w.volB,
                                      w.peakVotes, w.peakIdx, s, dec));
    // candidate scoring: sorted sweep while the sort fits LDS, all-pairs scan otherwise (same sums)
    if (score_by_sweep(N, joinBefore != nullptr || w.grid.presorted, o)) {
        if (joinBefore != nullptr) {
            ICPFLOW_TRY(hipStreamWaitEvent(s, joinBefore, 0));
        } else if (!w.grid.presorted) {
            ICPFLOW_TRY(launch_sort_clouds_soa(src, dst, w.lenA, w.lenC, swap, B, N, &w.grid, s));
            w.grid.presorted = 1;
            if (o.on(ICPFLOW_OPT_NO_SCORE_PRUNE) && o.on(ICPFLOW_OPT_NO_SCORE_PREBOUND)) {
                ICPFLOW_TRY(launch_occupancy(&w.grid, B, N, s));
                w.grid.occReady = 1;
            }
        }
        if (o.on(ICPFLOW_OPT_NO_SCORE_PRUNE))
            ICPFLOW_TRY(launch_sweep_score_pruned(&w.grid, w.lenA, w.lenC, swap, B, N, w.cand, w.partial, w.scoreAccum, s));
        else
            ICPFLOW_TRY(launch_sweep_score(&w.grid, w.lenA, w.lenC, swap, B, N, w.cand, w.partial, s));
        ICPFLOW_TRY(launch_score_pick(w.partial, sweep_qblocks(N), w.lenA, w.lenC, swap, w.cand, B, Tout, s, w.scoreAccum));
        w.initSumValid = true;   // (scoreAccum is free once the scans are over; the next call clears it)
    } else if (o.on(ICPFLOW_OPT_NO_SCORE_PRUNE)) {
        ICPFLOW_TRY(launch_scan_score_pruned(src, dst, w.lenA, w.lenC, swap, B, N, w.cand, w.partial, w.scoreAccum, s, true));
        ICPFLOW_TRY(launch_score_pick(w.partial, score_qblocks(N), w.lenA, w.lenC, swap, w.cand, B, Tout, s));
    } else {
        ICPFLOW_TRY(launch_scan_score(src, dst, w.lenA, w.lenC, swap, B, N, w.cand, w.partial, s));
        ICPFLOW_TRY(launch_score_pick(w.partial, scan_qblocks(N, B), w.lenA, w.lenC, swap, w.cand, B, Tout, s));
    }
    return 0;
}

}  // namespace

extern "C" {

int icpflow_version(void) { return ICPFLOW_VERSION; }

#ifndef ICPFLOW_SOURCE_HASH
#define ICPFLOW_SOURCE_HASH "unknown"
#endif
// the marker in front of the hash lets icp_flow_amd/build.py find it in the file without loading the library
static const char g_build_info[] = "ICPFLOW_SOURCE_HASH=" ICPFLOW_SOURCE_HASH;
const char *icpflow_build_info(void) { return g_build_info + sizeof("ICPFLOW_SOURCE_HASH=") - 1; }

const char *icpflow_last_error(void) { return g_err; }

size_t icpflow_workspace_bytes(int B, int N, int Lx, int Ly, int Lz)
{
    if (B <= 0 || N <= 0) return 0;
    const size_t L = (size_t)(Lx > 0 ? Lx : 0) * (size_t)(Ly > 0 ? Ly : 0) * (size_t)(Lz > 0 ? Lz : 0);
    return Workspace(nullptr, B, N, L).bytes;
}

int icpflow_profile_create(int capacity, icpflow_profile_t **out)
{
    if (out == nullptr) return fail(ICPFLOW_E_ARG, "icpflow_profile_create: null pointer");
    if (capacity < 0 || capacity > (1 << 20)) return fail(ICPFLOW_E_ARG, "icpflow_profile_create: bad capacity %d", capacity);
    hipError_t e = hipSuccess;
    LaunchProfile *p = profile_create(capacity, &e);
    if (p == nullptr) return hipfail(e, "icpflow_profile_create");
    *out = reinterpret_cast<icpflow_profile_t *>(p);
    return 0;
}

int icpflow_profile_collect(icpflow_profile_t *profile, double *total_ms, int *launches)
{
    if (profile == nullptr) return fail(ICPFLOW_E_ARG, "icpflow_profile_collect: null recorder");
    ICPFLOW_TRY(profile_collect(reinterpret_cast<LaunchProfile *>(profile), total_ms, launches));
    return 0;
}

int icpflow_profile_destroy(icpflow_profile_t *profile)
{
    profile_destroy(reinterpret_cast<LaunchProfile *>(profile));
    return 0;
}

int icpflow_selftest_vote_quotient(const float *d_a, int n, float min_v, float max_v, float *d_fast,
                                   float *d_ieee, icpflow_stream_t stream)
{
    if (!d_a || !d_fast || !d_ieee) return fail(ICPFLOW_E_ARG, "icpflow_selftest_vote_quotient: null pointer");
    if (n <= 0) return fail(ICPFLOW_E_ARG, "icpflow_selftest_vote_quotient: n must be positive");
    ICPFLOW_TRY(launch_vote_quotient_probe(d_a, n, min_v, max_v, d_fast, d_ieee, (hipStream_t)stream));
    return 0;
}

int icpflow_hist_vote(const float *d_X, const float *d_Y, int B, int NX, int NY, float min_x,
                      float min_y, float min_z, float max_x, float max_y, float max_z, int len_x,
                      int len_y, int len_z, float *d_bins, icpflow_stream_t stream)
{
    if (!d_X || !d_Y || !d_bins) return fail(ICPFLOW_E_ARG, "icpflow_hist_vote: null pointer");
    if (int r = check_batch("icpflow_hist_vote", B, NX)) return r;
    if (int r = check_batch("icpflow_hist_vote", B, NY)) return r;
    if (int r = check_hist_dims("icpflow_hist_vote", len_x, len_y, len_z)) return r;
    hipStream_t s = (hipStream_t)stream;
    const float mins[3] = {min_x, min_y, min_z}, maxs[3] = {max_x, max_y, max_z};
    const int lens[3] = {len_x, len_y, len_z};
    // counters are accumulated as uint32 in the output buffer, then converted in place
    uint32_t *u = reinterpret_cast<uint32_t *>(d_bins);
    ICPFLOW_TRY(launch_hist_vote(d_X, d_Y, B, NX, NY, mins, maxs, lens, nullptr, nullptr, nullptr, nullptr, u, s));
    ICPFLOW_TRY(launch_u32_to_f32(u, d_bins, (size_t)B * len_x * len_y * len_z, s));
    return 0;
}

int icpflow_hist_peaks(const float *d_bins, int B, int len_x, int len_y, int len_z, int k,
                       int kernel_size, float *d_votes, int64_t *d_idx, void *d_ws, size_t ws_bytes,
                       icpflow_stream_t stream)
{
    if (!d_bins || !d_votes || !d_idx) return fail(ICPFLOW_E_ARG, "icpflow_hist_peaks: null pointer");
    if (B <= 0) return fail(ICPFLOW_E_ARG, "icpflow_hist_peaks: B must be positive");
    if (int r = check_hist_dims("icpflow_hist_peaks", len_x, len_y, len_z)) return r;
    if (k <= 0 || k > 8) return fail(ICPFLOW_E_ARG, "icpflow_hist_peaks: k must be in 1..8 (got %d)", k);
    if (kernel_size <= 0 || kernel_size % 2 == 0)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_peaks: kernel_size must be odd and positive");
    const size_t L = (size_t)len_x * len_y * len_z;
    if ((size_t)k > L) return fail(ICPFLOW_E_ARG, "icpflow_hist_peaks: k exceeds the number of bins");
    const size_t vol = up((size_t)B * L * 4);
    if (int r = check_ws(d_ws, ws_bytes, 2 * vol)) return r;
    uint32_t *A = (uint32_t *)d_ws;
    uint32_t *Bv = (uint32_t *)((char *)d_ws + vol);
    ICPFLOW_TRY(launch_hist_peaks_f32(d_bins, B, len_x, len_y, len_z, k, kernel_size, A, Bv, d_votes, d_idx,
                                      (hipStream_t)stream));
    return 0;
}

int icpflow_nn_batch(const float *d_Q, const float *d_T, int B, int NQ, int NT, int q_stride,
                     int t_stride, const int32_t *d_len_q, const int32_t *d_len_t, int sqrt_dist,
                     int64_t *d_idx, float *d_dist, icpflow_stream_t stream)
{
    if (!d_Q || !d_T || !d_idx || !d_dist) return fail(ICPFLOW_E_ARG, "icpflow_nn_batch: null pointer");
    if (int r = check_batch("icpflow_nn_batch", B, NQ)) return r;
    if (int r = check_batch("icpflow_nn_batch", B, NT)) return r;
    if (q_stride < 3 || t_stride < 3)
        return fail(ICPFLOW_E_ARG, "icpflow_nn_batch: row strides must be >= 3 floats (got %d, %d)", q_stride, t_stride);
    if ((q_stride == 4 && ((uintptr_t)d_Q & 15)) || (t_stride == 4 && ((uintptr_t)d_T & 15)))
        return fail(ICPFLOW_E_ARG, "icpflow_nn_batch: stride-4 clouds must be 16-byte aligned");
    ICPFLOW_TRY(launch_scan_nn(d_Q, d_T, B, NQ, NT, q_stride, t_stride, d_len_q, d_len_t, sqrt_dist, d_idx,
                               d_dist, (hipStream_t)stream));
    return 0;
}

int icpflow_transform_points(const float *d_xyz, const float *d_pose, int B, int N, float *d_out,
                             icpflow_stream_t stream)
{
    if (!d_xyz || !d_pose || !d_out) return fail(ICPFLOW_E_ARG, "icpflow_transform_points: null pointer");
    if (int r = check_batch("icpflow_transform_points", B, N)) return r;
    ICPFLOW_TRY(launch_transform_points(d_xyz, d_pose, B, N, d_out, (hipStream_t)stream));
    return 0;
}

int icpflow_gather_pad(const float *d_points, const int32_t *d_rows, int B, int N, float *d_out,
                       icpflow_stream_t stream)
{
    if (!d_points || !d_rows || !d_out) return fail(ICPFLOW_E_ARG, "icpflow_gather_pad: null pointer");
    if (int r = check_batch("icpflow_gather_pad", B, N)) return r;
    ICPFLOW_TRY(launch_gather_pad(d_points, d_rows, B, N, d_out, (hipStream_t)stream));
    return 0;
}

int icpflow_gather_segments(const float *d_points, const int64_t *d_order, const int64_t *d_seg,
                            const int32_t *d_perm, int B, int N, float *d_out, icpflow_stream_t stream)
{
    if (!d_points || !d_order || !d_seg || !d_out) return fail(ICPFLOW_E_ARG, "icpflow_gather_segments: null pointer");
    if (int r = check_batch("icpflow_gather_segments", B, N)) return r;
    ICPFLOW_TRY(launch_gather_segments(d_points, d_order, d_seg, d_perm, B, N, d_out, (hipStream_t)stream));
    return 0;
}

int icpflow_assoc_assign(const float *d_result, const int32_t *d_si, const int32_t *d_di, int K, const uint8_t *d_active,
                         int S, int D, float translation_frame, float thres_iou, float rot_limit_deg, float thres_error,
                         int32_t *d_best, int K2, const int32_t *d_si2, const int32_t *d_di2, int64_t *d_seg2,
                         uint8_t *d_active2, icpflow_stream_t stream)
{
    if (!d_result || !d_si || !d_di || !d_best) return fail(ICPFLOW_E_ARG, "icpflow_assoc_assign: null pointer");
    if (K <= 0 || S <= 0 || D <= 0 || S > assoc_max_rows() || D > assoc_max_rows())
        return fail(ICPFLOW_E_LIMIT, "icpflow_assoc_assign: K (%d) must be positive and S, D (%d, %d) in 1..%d", K, S, D, assoc_max_rows());
    if (K2 < 0 || (K2 > 0 && (!d_si2 || !d_di2 || !d_seg2 || !d_active2)))
        return fail(ICPFLOW_E_ARG, "icpflow_assoc_assign: the next stage's candidates come with d_si2, d_di2, d_seg2 and d_active2");
    ICPFLOW_TRY(launch_assoc_assign(d_result, d_si, d_di, K, d_active, S, D, translation_frame, thres_iou, rot_limit_deg,
                                    thres_error, d_best, K2, d_si2, d_di2, d_seg2, d_active2, (hipStream_t)stream));
    return 0;
}

int icpflow_assoc_collect(const int32_t *d_best1, const float *d_result1, const int32_t *d_si1, const int32_t *d_di1, int K1,
                          const int32_t *d_best2, const float *d_result2, const int32_t *d_si2, const int32_t *d_di2, int K2,
                          const double *d_src_table, const double *d_dst_table, int label_stride, int S, int cap,
                          float *d_rows, float *d_T, int32_t *d_count, icpflow_stream_t stream)
{
    if (!d_best1 || !d_result1 || !d_si1 || !d_di1 || !d_src_table || !d_dst_table || !d_rows || !d_T || !d_count)
        return fail(ICPFLOW_E_ARG, "icpflow_assoc_collect: null pointer");
    if (K1 <= 0 || S <= 0 || S > assoc_max_rows() || cap <= 0 || label_stride <= 0)
        return fail(ICPFLOW_E_LIMIT, "icpflow_assoc_collect: K1 (%d), cap (%d), label_stride (%d) must be positive and S (%d) in 1..%d", K1, cap,
                    label_stride, S, assoc_max_rows());
    if (K2 < 0 || (K2 > 0 && (!d_best2 || !d_result2 || !d_si2 || !d_di2)))
        return fail(ICPFLOW_E_ARG, "icpflow_assoc_collect: stage 2 comes with d_best2, d_result2, d_si2 and d_di2");
    ICPFLOW_TRY(launch_assoc_collect(d_best1, d_result1, d_si1, d_di1, K1, K2 > 0 ? d_best2 : nullptr, d_result2, d_si2, d_di2, K2,
                                     d_src_table, d_dst_table, label_stride, S, cap, d_rows, d_T, d_count, (hipStream_t)stream));
    return 0;
}

// One association stage per call: the padded batch of both clouds, the registration, its metrics (declared below)
int icpflow_hist_icp_eval(const float *d_src, const float *d_dst, int B, int N, const float *d_edges_x, int len_x,
                          const float *d_edges_y, int len_y, const float *d_edges_z, int len_z, float decode_shift,
                          double thres_dist, int max_iterations, double relative_rmse_thr, int stop_mode,
                          float *d_T_out, int32_t *d_iters, float *d_errors, float *d_inliers, float *d_ratios,
                          float *d_ious, float *d_translations, float *d_rotations, void *d_ws, size_t ws_bytes,
                          icpflow_stream_t stream, const icpflow_options_t *opt);
int icpflow_flow_rigid_rows(const float *d_points, const float *d_labels, int N, const float *d_pair_rows, int pair_stride,
                            const float *d_T, int P, const float *d_pose, float *d_flow, icpflow_stream_t stream);

int icpflow_register_stage(const icpflow_tables_t *t, const icpflow_stage_t *st, const icpflow_registration_t *reg,
                           void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt)
{
    if (!t || !st || !reg) return fail(ICPFLOW_E_ARG, "icpflow_register_stage: null argument");
    if (!t->d_points_src || !t->d_order_src || !t->d_points_dst || !t->d_order_dst || !st->d_seg || !st->d_clouds || !st->d_result)
        return fail(ICPFLOW_E_ARG, "icpflow_register_stage: null pointer");
    const int K = st->K, N = st->N;
    if (int r = check_batch("icpflow_register_stage", K, N)) return r;
    const size_t cloud = (size_t)K * N * 4;
    if (int r = icpflow_gather_segments(t->d_points_src, t->d_order_src, st->d_seg, st->d_perm, K, N, st->d_clouds, stream)) return r;
    if (int r = icpflow_gather_segments(t->d_points_dst, t->d_order_dst, st->d_seg + (size_t)3 * K, st->d_perm, K, N,
                                        st->d_clouds + cloud, stream))
        return r;
    float *R = st->d_result;
    const size_t k = (size_t)K;
    return icpflow_hist_icp_eval(st->d_clouds, st->d_clouds + cloud, K, N, reg->d_edges_x, reg->len_x, reg->d_edges_y, reg->len_y,
                                 reg->d_edges_z, reg->len_z, reg->decode_shift, reg->thres_dist, reg->max_iterations,
                                 reg->relative_rmse_thr, reg->stop_mode, R, reinterpret_cast<int32_t *>(R + 30 * k), R + 16 * k,
                                 R + 18 * k, R + 20 * k, R + 22 * k, R + 24 * k, R + 27 * k, d_ws, ws_bytes, stream, opt);
}

static int associate_frame_impl(const icpflow_tables_t *t, const icpflow_stage_t *s1, const icpflow_stage_t *s2, uint8_t *d_active2,
                            const icpflow_registration_t *reg, float translation_frame, float thres_iou, float rot_limit_deg,
                            float thres_error, int32_t *d_best, int cap, float *d_rows, float *d_T, const float *d_flow_points,
                            const float *d_flow_labels, int n_flow, const float *d_pose, float *d_flow, void *d_ws,
                            size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt, const int32_t *h_carry2)
{
    if (!t || !s1 || !reg || !d_best || !d_rows || !d_T) return fail(ICPFLOW_E_ARG, "icpflow_associate_frame: null argument");
    if (!t->d_table_src || !t->d_table_dst || !s1->d_result || !s1->d_si || !s1->d_di)
        return fail(ICPFLOW_E_ARG, "icpflow_associate_frame: null pointer");
    const int S = t->S, D = t->D, K1 = s1->K, K2 = s2 ? s2->K : 0;
    if (K2 > 0 && (!d_active2 || !s2->d_si || !s2->d_di || !s2->d_seg || !s2->d_result))
        return fail(ICPFLOW_E_ARG, "icpflow_associate_frame: stage 2 comes with d_active2, d_si, d_di, d_seg and d_result");
    int32_t *best1 = d_best, *best2 = d_best + S, *count = d_best + 2 * (size_t)S;
    // (stage 2's segment rows are written by the assignment: the lengths of the switched-off candidates become 0)
    if (int r = icpflow_assoc_assign(s1->d_result, s1->d_si, s1->d_di, K1, nullptr, S, D, translation_frame, thres_iou, rot_limit_deg,
                                     thres_error, best1, K2, K2 ? s2->d_si : nullptr, K2 ? s2->d_di : nullptr,
                                     K2 ? const_cast<int64_t *>(s2->d_seg) : nullptr, K2 ? d_active2 : nullptr, stream))
        return r;
    if (K2 > 0) {
        icpflow_options_t o2{};
        o2.struct_size = sizeof(icpflow_options_t);
        if (opt) o2 = *opt;
        o2.d_pair_active = d_active2;
        // (h_carry2: stage 2's first half has run already -- icpflow_register_stage_begin on the WHOLE superset, in d_ws --; the
        // switched-off candidates then carry their clouds, and every kernel of the second half passes them over by the mask)
        if (h_carry2 != nullptr) {
            if (int r = icpflow_register_stage_finish(t, s2, reg, d_ws, ws_bytes, stream, &o2, h_carry2)) return r;
        } else if (int r = icpflow_register_stage(t, s2, reg, d_ws, ws_bytes, stream, &o2)) return r;
        if (int r = icpflow_assoc_assign(s2->d_result, s2->d_si, s2->d_di, K2, d_active2, S, D, translation_frame, thres_iou,
                                         rot_limit_deg, thres_error, best2, 0, nullptr, nullptr, nullptr, nullptr, stream))
            return r;
    }
    if (int r = icpflow_assoc_collect(best1, s1->d_result, s1->d_si, s1->d_di, K1, K2 ? best2 : nullptr, K2 ? s2->d_result : nullptr,
                                      K2 ? s2->d_si : nullptr, K2 ? s2->d_di : nullptr, K2, t->d_table_src, t->d_table_dst,
                                      t->label_stride, S, cap, d_rows, d_T, count, stream))
        return r;
    if (d_flow != nullptr)
        return icpflow_flow_rigid_rows(d_flow_points, d_flow_labels, n_flow, d_rows, 10, d_T, cap, d_pose, d_flow, stream);
    return 0;
}

int icpflow_associate_frame(const icpflow_tables_t *t, const icpflow_stage_t *s1, const icpflow_stage_t *s2, uint8_t *d_active2,
                            const icpflow_registration_t *reg, float translation_frame, float thres_iou, float rot_limit_deg,
                            float thres_error, int32_t *d_best, int cap, float *d_rows, float *d_T, const float *d_flow_points,
                            const float *d_flow_labels, int n_flow, const float *d_pose, float *d_flow, void *d_ws,
                            size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt)
{
    return associate_frame_impl(t, s1, s2, d_active2, reg, translation_frame, thres_iou, rot_limit_deg, thres_error, d_best, cap, d_rows, d_T,
                                d_flow_points, d_flow_labels, n_flow, d_pose, d_flow, d_ws, ws_bytes, stream, opt, nullptr);
}

int icpflow_associate_frame_begun(const icpflow_tables_t *t, const icpflow_stage_t *s1, const icpflow_stage_t *s2, uint8_t *d_active2,
                                  const icpflow_registration_t *reg, float translation_frame, float thres_iou, float rot_limit_deg,
                                  float thres_error, int32_t *d_best, int cap, float *d_rows, float *d_T, const float *d_flow_points,
                                  const float *d_flow_labels, int n_flow, const float *d_pose, float *d_flow, void *d_ws2,
                                  size_t ws2_bytes, icpflow_stream_t stream, const icpflow_options_t *opt, const int32_t *h_carry2)
{
    if (!h_carry2) return fail(ICPFLOW_E_ARG, "icpflow_associate_frame_begun: null argument");
    return associate_frame_impl(t, s1, s2, d_active2, reg, translation_frame, thres_iou, rot_limit_deg, thres_error, d_best, cap, d_rows, d_T,
                                d_flow_points, d_flow_labels, n_flow, d_pose, d_flow, d_ws2, ws2_bytes, stream, opt, h_carry2);
}

size_t icpflow_dbscan_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    size_t bytes = 0;
    if (dbscan_workspace_bytes(n, &bytes) != hipSuccess) return 0;
    return bytes;
}

int icpflow_dbscan(const float *d_points, int stride, const uint8_t *d_mask, int n, double eps, int min_points,
                   int32_t *d_labels, int32_t *d_counts, int32_t *d_num_clusters, void *d_ws, size_t ws_bytes,
                   icpflow_stream_t stream)
{
    if (!d_points || !d_labels || !d_counts || !d_num_clusters)
        return fail(ICPFLOW_E_ARG, "icpflow_dbscan: null pointer");
    if (n <= 0) return fail(ICPFLOW_E_ARG, "icpflow_dbscan: n must be positive (got %d)", n);
    if (stride < 3) return fail(ICPFLOW_E_ARG, "icpflow_dbscan: stride must be >= 3 floats (got %d)", stride);
    if (!(eps > 0.0) || min_points < 1)
        return fail(ICPFLOW_E_ARG, "icpflow_dbscan: eps must be positive and min_points >= 1 (got %g, %d)", eps,
                    min_points);
    if (!d_ws) return fail(ICPFLOW_E_WORKSPACE, "icpflow_dbscan: workspace is NULL");
    bool tooSmall = false;
    ICPFLOW_TRY(launch_dbscan(d_points, stride, d_mask, n, eps, min_points, d_labels, d_counts, d_num_clusters, d_ws,
                              ws_bytes, &tooSmall, (hipStream_t)stream));
    if (tooSmall)
        return fail(ICPFLOW_E_WORKSPACE, "icpflow_dbscan: workspace too small (%zu bytes, need %zu)", ws_bytes,
                    icpflow_dbscan_workspace_bytes(n));
    return 0;
}

size_t icpflow_hdbscan_mst_workspace_bytes(int n)
{
    if (n <= 0) return 0;
    size_t bytes = 0;
    if (hdbscan_workspace_bytes(n, &bytes) != hipSuccess) return 0;
    return bytes;
}

int icpflow_hdbscan_mst(const float *d_points, int stride, const uint8_t *d_mask, int n, int min_samples, double cell,
                        double *d_core2, int32_t *d_edge_a, int32_t *d_edge_b, double *d_edge_w2,
                        int32_t *d_num_edges, int32_t *d_num_live, void *d_ws, size_t ws_bytes,
                        icpflow_stream_t stream)
{
    if (!d_points || !d_edge_a || !d_edge_b || !d_edge_w2 || !d_num_edges || !d_num_live)
        return fail(ICPFLOW_E_ARG, "icpflow_hdbscan_mst: null pointer");
    if (n <= 0) return fail(ICPFLOW_E_ARG, "icpflow_hdbscan_mst: n must be positive (got %d)", n);
    if (stride < 3) return fail(ICPFLOW_E_ARG, "icpflow_hdbscan_mst: stride must be >= 3 floats (got %d)", stride);
    if (min_samples < 1 || min_samples > 64)
        return fail(ICPFLOW_E_LIMIT, "icpflow_hdbscan_mst: min_samples must be in 1..64 (got %d)", min_samples);
    if (!(cell > 0.0)) return fail(ICPFLOW_E_ARG, "icpflow_hdbscan_mst: cell must be positive (got %g)", cell);
    if (!d_ws) return fail(ICPFLOW_E_WORKSPACE, "icpflow_hdbscan_mst: workspace is NULL");
    bool tooSmall = false;
    ICPFLOW_TRY(launch_hdbscan_mst(d_points, stride, d_mask, n, min_samples, cell, d_core2, d_edge_a, d_edge_b,
                                   d_edge_w2, d_num_edges, d_num_live, d_ws, ws_bytes, &tooSmall,
                                   (hipStream_t)stream));
    if (tooSmall)
        return fail(ICPFLOW_E_WORKSPACE, "icpflow_hdbscan_mst: workspace too small (%zu bytes, need %zu)", ws_bytes,
                    icpflow_hdbscan_mst_workspace_bytes(n));
    return 0;
}

int icpflow_cluster_stats(const float *d_points, const int64_t *d_order, const int64_t *d_start,
                          const int64_t *d_count, const float *d_labels, int L, float *d_mean, float *d_extent,
                          icpflow_stream_t stream)
{
    if (!d_points || !d_order || !d_start || !d_count || !d_mean || !d_extent)
        return fail(ICPFLOW_E_ARG, "icpflow_cluster_stats: null pointer");
    if (L <= 0) return fail(ICPFLOW_E_ARG, "icpflow_cluster_stats: L must be positive (got %d)", L);
    ICPFLOW_TRY(launch_cluster_stats(d_points, d_order, d_start, d_count, d_labels, L, d_mean, d_extent,
                                     (hipStream_t)stream));
    return 0;
}

size_t icpflow_cluster_table_workspace_bytes(int M, int Lmax)
{
    if (M <= 0 || Lmax <= 0) return 0;
    size_t bytes = 0;
    if (cluster_table_workspace_bytes(M, Lmax, &bytes) != hipSuccess) return 0;
    return bytes;
}

int icpflow_cluster_table(const float *d_points, const float *d_labels, int M, int64_t *d_order, double *d_table, int Lmax,
                          int32_t *d_num, void *d_ws, size_t ws_bytes, icpflow_stream_t stream)
{
    if (!d_points || !d_labels || !d_order || !d_table || !d_num)
        return fail(ICPFLOW_E_ARG, "icpflow_cluster_table: null pointer");
    if (M <= 0) return fail(ICPFLOW_E_ARG, "icpflow_cluster_table: M must be positive (got %d)", M);
    if (Lmax <= 0 || Lmax > 4096) return fail(ICPFLOW_E_LIMIT, "icpflow_cluster_table: 1 <= Lmax <= 4096 (got %d)", Lmax);
    if (!d_ws) return fail(ICPFLOW_E_WORKSPACE, "icpflow_cluster_table: workspace is NULL");
    bool tooSmall = false;
    ICPFLOW_TRY(launch_cluster_table(d_points, d_labels, M, d_order, d_table, Lmax, d_num, d_ws, ws_bytes, &tooSmall,
                                     (hipStream_t)stream));
    if (tooSmall)
        return fail(ICPFLOW_E_WORKSPACE, "icpflow_cluster_table: workspace too small (%zu bytes, need %zu)", ws_bytes,
                    icpflow_cluster_table_workspace_bytes(M, Lmax));
    return 0;
}

size_t icpflow_cluster_table_pair_workspace_bytes(int MA, int MB, int Lmax)
{
    if (MA <= 0 || MB <= 0 || Lmax <= 0 || (long long)MA + MB > 0x7fffffffll) return 0;
    size_t bytes = 0;
    if (cluster_table_pair_workspace_bytes(MA, MB, Lmax, &bytes) != hipSuccess) return 0;
    return bytes;
}

int icpflow_cluster_table_pair(const float *d_points_a, const float *d_labels_a, int MA, int64_t *d_order_a, double *d_table_a,
                               int32_t *d_num_a, const float *d_points_b, const float *d_labels_b, int MB,
                               int64_t *d_order_b, double *d_table_b, int32_t *d_num_b, int Lmax, void *d_ws, size_t ws_bytes,
                               icpflow_stream_t stream)
{
    if (!d_points_a || !d_labels_a || !d_order_a || !d_table_a || !d_num_a || !d_points_b || !d_labels_b || !d_order_b ||
        !d_table_b || !d_num_b)
        return fail(ICPFLOW_E_ARG, "icpflow_cluster_table_pair: null pointer");
    if (MA <= 0 || MB <= 0) return fail(ICPFLOW_E_ARG, "icpflow_cluster_table_pair: MA, MB must be positive (got %d, %d)", MA, MB);
    if ((long long)MA + MB > 0x7fffffffll)
        return fail(ICPFLOW_E_LIMIT, "icpflow_cluster_table_pair: MA + MB must stay below 2^31 (got %d + %d)", MA, MB);
    if (Lmax <= 0 || Lmax > 4096) return fail(ICPFLOW_E_LIMIT, "icpflow_cluster_table_pair: 1 <= Lmax <= 4096 (got %d)", Lmax);
    if (!d_ws) return fail(ICPFLOW_E_WORKSPACE, "icpflow_cluster_table_pair: workspace is NULL");
    bool tooSmall = false;
    ICPFLOW_TRY(launch_cluster_table_pair(d_points_a, d_labels_a, MA, d_order_a, d_table_a, d_num_a, d_points_b, d_labels_b, MB,
                                          d_order_b, d_table_b, d_num_b, Lmax, d_ws, ws_bytes, &tooSmall, (hipStream_t)stream));
    if (tooSmall)
        return fail(ICPFLOW_E_WORKSPACE, "icpflow_cluster_table_pair: workspace too small (%zu bytes, need %zu)", ws_bytes,
                    icpflow_cluster_table_pair_workspace_bytes(MA, MB, Lmax));
    return 0;
}

int icpflow_flow_rigid_rows(const float *d_points, const float *d_labels, int N, const float *d_pair_rows, int pair_stride,
                            const float *d_T, int P, const float *d_pose, float *d_flow, icpflow_stream_t stream)
{
    if (!d_points || !d_labels || !d_pose || !d_flow) return fail(ICPFLOW_E_ARG, "icpflow_flow_rigid: null pointer");
    if (N <= 0) return fail(ICPFLOW_E_ARG, "icpflow_flow_rigid: N must be positive");
    if (P < 0 || P > (1 << 24)) return fail(ICPFLOW_E_LIMIT, "icpflow_flow_rigid: 0 <= P <= 2^24 pairs (got %d)", P);
    if (P > 0 && (!d_pair_rows || !d_T)) return fail(ICPFLOW_E_ARG, "icpflow_flow_rigid: null pair arrays");
    if (pair_stride < 1 || pair_stride > 1024)
        return fail(ICPFLOW_E_ARG, "icpflow_flow_rigid: pair_stride must be in 1 ... 1024 floats (got %d)", pair_stride);
    ICPFLOW_TRY(launch_flow_rigid(d_points, d_labels, N, d_pair_rows, pair_stride, d_T, P, d_pose, d_flow, (hipStream_t)stream));
    return 0;
}

int icpflow_flow_rigid(const float *d_points, const float *d_labels, int N, const float *d_pair_labels,
                       const float *d_T, int P, const float *d_pose, float *d_flow, void *d_ws, size_t ws_bytes,
                       icpflow_stream_t stream)
{
    (void)d_ws;        // (the products T[p] * pose are formed per point since version 204: no scratch)
    (void)ws_bytes;
    return icpflow_flow_rigid_rows(d_points, d_labels, N, d_pair_labels, 1, d_T, P, d_pose, d_flow, stream);
}

int icpflow_count_valid(const float *d_pts, int B, int N, int32_t *d_len, icpflow_stream_t stream)
{
    if (!d_pts || !d_len) return fail(ICPFLOW_E_ARG, "icpflow_count_valid: null pointer");
    if (int r = check_batch("icpflow_count_valid", B, N)) return r;
    launch_count_valid(d_pts, B, N, d_len, (hipStream_t)stream);
    ICPFLOW_TRY(hipGetLastError());
    return 0;
}

int icpflow_estimate_init_pose(const float *d_src, const float *d_dst, int B, int N,
                               const float *d_edges_x, int len_x, const float *d_edges_y, int len_y,
                               const float *d_edges_z, int len_z, float decode_shift, float *d_T_out,
                               void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt)
{
    Opts o;
    if (int r = parse_options("icpflow_estimate_init_pose", opt, o)) return r;
    if (int r = check_pair_active("icpflow_estimate_init_pose", o, false, 0, 0)) return r;
    if (!d_src || !d_dst || !d_edges_x || !d_edges_y || !d_edges_z || !d_T_out)
        return fail(ICPFLOW_E_ARG, "icpflow_estimate_init_pose: null pointer");
    if (int r = check_batch("icpflow_estimate_init_pose", B, N)) return r;
    if (int r = check_hist_dims("icpflow_estimate_init_pose", len_x, len_y, len_z)) return r;
    const size_t L = (size_t)len_x * len_y * len_z;
    if (L < (size_t)kTopK) return fail(ICPFLOW_E_ARG, "icpflow_estimate_init_pose: fewer than 5 bins");
    Workspace w(d_ws, B, N, L);
    if (int r = check_ws(d_ws, ws_bytes, w.bytes)) return r;
    hipStream_t s = (hipStream_t)stream;
    launch_count_pair(d_src, d_dst, B, N, w.lenA, w.lenC, nullptr, s, w.scoreAccum, (size_t)B * 12 * sizeof(double));
    return run_init_pose(d_src, d_dst, w, nullptr, B, N, d_edges_x, len_x, d_edges_y, len_y, d_edges_z,
                         len_z, decode_shift, d_T_out, o, s);
}

int icpflow_icp(const float *d_X, const float *d_Y, const float *d_pre_pose, int B, int N, double thres,
                int max_iterations, double relative_rmse_thr, int stop_mode, float *d_R, float *d_T,
                float *d_rmse, int32_t *d_iters, int32_t *d_converged, void *d_ws, size_t ws_bytes,
                icpflow_stream_t stream, const icpflow_options_t *opt)
{
    Opts o;
    if (int r = parse_options("icpflow_icp", opt, o)) return r;
    if (int r = check_pair_active("icpflow_icp", o, false, 0, 0)) return r;
    if (int r = check_arith("icpflow_icp", o, max_iterations, stop_mode)) return r;
    if (!d_X || !d_Y) return fail(ICPFLOW_E_ARG, "icpflow_icp: null pointer");
    if (int r = check_batch("icpflow_icp", B, N)) return r;
    if (max_iterations <= 0 || max_iterations > kMaxIterCap)
        return fail(ICPFLOW_E_ARG, "icpflow_icp: max_iterations must be in 1..%d (got %d)", kMaxIterCap, max_iterations);
    if (stop_mode != ICPFLOW_STOP_REFERENCE && stop_mode != ICPFLOW_STOP_PER_PAIR)
        return fail(ICPFLOW_E_ARG, "icpflow_icp: unknown stop_mode %d", stop_mode);
    Workspace w(d_ws, B, N, 0);
    if (int r = check_ws(d_ws, ws_bytes, w.bytes)) return r;
    hipStream_t s = (hipStream_t)stream;
    IcpOpts io = o.icp(w.grid.sortX);
    io.help = icp_help_carve(w.ctrl, B, w.helpState, w.helpOut);
    io.splitScratch = w.icpSplit;
    io.initR = o.initR;
    io.initT = o.initT;
    io.allowReflection = o.allowReflection;
    io.estimateScale = o.estimateScale;
    io.initS = o.initS;
    // (every argument check sits in front of the first launch: a refused call leaves the stream untouched)
    if ((o.allowReflection || o.estimateScale) && o.arith != ICPFLOW_ARITH_FP64)
        return fail(ICPFLOW_E_ARG, "icpflow_icp: allow_reflection / estimate_scale are not built for ICPFLOW_ARITH_FP32_REFERENCE");
    if (o.history != nullptr &&
        !(stop_mode == ICPFLOW_STOP_REFERENCE && max_iterations > 1 && max_iterations <= kHistIters &&
          o.arith == ICPFLOW_ARITH_FP64 && io.speculative))
        return fail(ICPFLOW_E_ARG, "icpflow_icp: options.d_icp_history needs the single-launch reference stop mode "
                                   "(2 <= max_iterations <= %d, fp64 arithmetic)", kHistIters);
    if (o.initR != nullptr && o.arith != ICPFLOW_ARITH_FP64)
        return fail(ICPFLOW_E_ARG, "icpflow_icp: an initial transform is not built for ICPFLOW_ARITH_FP32_REFERENCE");
    // similarity transforms are built into the sorted-sweep kernels only: below 64 points, where the automatic choice is the
    // all-pairs scan, a request for a scale takes the sweep (the drop-in must not raise where the
    // reference works); beyond the sort's length (kMaxSortN) it is still refused
    Opts os = o;
    if ((o.estimateScale || o.initS != nullptr) && o.search == 0 && N < 64) os.search = 3;
    const GridScratch *search = search_scratch(w, N, os);
    if ((o.estimateScale || o.initS != nullptr) && (search == nullptr || search->mode != 3))
        return fail(ICPFLOW_E_ARG, "icpflow_icp: estimate_scale / an initial transform with a scale need the sorted-sweep "
                                   "search (the default up to N = %d)", kMaxSortN);
    launch_count_pair(d_X, d_Y, B, N, w.lenA, w.lenC, nullptr, s, w.ctrl, icp_ctrl_bytes(B));
    ICPFLOW_TRY(launch_icp(d_X, d_Y, w.lenA, w.lenC, nullptr, d_pre_pose, B, N, thres, max_iterations,
                           relative_rmse_thr, stop_mode, w.state, w.ctrl, search, w.history, &w.team,
                           io, s));
    if (o.history != nullptr)   // t_history: the per-iteration records of the speculative launch
        ICPFLOW_TRY(hipMemcpyAsync(o.history, w.history, (size_t)max_iterations * B * kHistStride * sizeof(float),
                                   hipMemcpyDeviceToDevice, s));
    ICPFLOW_TRY(launch_icp_export(w.state, w.ctrl, B, stop_mode, d_R, d_T, d_rmse, d_iters, d_converged, s, o.scaleOut));
    return 0;
}

int icpflow_apply_icp(const float *d_src, const float *d_dst, const float *d_init, int B, int N,
                      double thres_dist, int max_iterations, double relative_rmse_thr, int stop_mode,
                      float *d_T_out, int32_t *d_iters, void *d_ws, size_t ws_bytes,
                      icpflow_stream_t stream, const icpflow_options_t *opt)
{
    Opts o;
    if (int r = parse_options("icpflow_apply_icp", opt, o)) return r;
    if (int r = check_arith("icpflow_apply_icp", o, max_iterations, stop_mode)) return r;
    if (int r = check_pair_active("icpflow_apply_icp", o, false, max_iterations, stop_mode)) return r;
    if (!d_src || !d_dst || !d_init || !d_T_out) return fail(ICPFLOW_E_ARG, "icpflow_apply_icp: null pointer");
    if (int r = check_batch("icpflow_apply_icp", B, N)) return r;
    if (max_iterations <= 0 || max_iterations > kMaxIterCap)
        return fail(ICPFLOW_E_ARG, "icpflow_apply_icp: max_iterations must be in 1..%d", kMaxIterCap);
    if (stop_mode != ICPFLOW_STOP_REFERENCE && stop_mode != ICPFLOW_STOP_PER_PAIR)
        return fail(ICPFLOW_E_ARG, "icpflow_apply_icp: unknown stop_mode %d", stop_mode);
    Workspace w(d_ws, B, N, 0);
    if (int r = check_ws(d_ws, ws_bytes, w.bytes)) return r;
    hipStream_t s = (hipStream_t)stream;
    launch_count_pair(d_src, d_dst, B, N, w.lenA, w.lenC, nullptr, s, w.ctrl, icp_ctrl_bytes(B));
    // d_T_out may alias d_init: keep a private copy of the init poses
    ICPFLOW_TRY(hipMemcpyAsync(w.Tinit, d_init, (size_t)B * 16 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return run_icp_and_select(d_src, d_dst, w, nullptr, w.Tinit, B, N, thres_dist, max_iterations,
                              relative_rmse_thr, stop_mode, 0, d_T_out, d_iters, o, s);
}

// icpflow_hist_icp behind its argument checks: everything it enqueues, on a workspace already carved
static int hist_icp_core(const float *d_src, const float *d_dst, int B, int N, const float *d_edges_x, int len_x,
                         const float *d_edges_y, int len_y, const float *d_edges_z, int len_z, float decode_shift,
                         double thres_dist, int max_iterations, double relative_rmse_thr, int stop_mode,
                         float *d_T_out, int32_t *d_iters, Workspace &w, const Opts &o, hipStream_t s,
                         int phase = 0, int32_t *carry = nullptr)
{
    // phase 0: the whole registration.  phase 1: its first half -- lengths, sorts, vote, peaks, scoring: the initial poses --,
    // leaving in carry[0..1] what the second half has to know about the workspace (clouds sorted by role, team plan made);
    // phase 2: the second half (ICP, roll-back check, select) from a workspace phase 1 has filled.  (icpflow_register_stage_begin
    // / _finish: a frame pair's stage 2 estimates its initial poses beside stage 1's ICP, on another stream.)
    if (phase == 2) {
        w.grid.presorted = carry[0];
        w.grid.shareCountClean = 1;   // (phase 1 cleared the counters, and every launch since has left them at zero)
        int32_t pending = carry[2];
        return run_icp_and_select(d_src, d_dst, w, w.swap, w.Tinit, B, N, thres_dist, max_iterations,
                                  relative_rmse_thr, stop_mode, 1, d_T_out, d_iters, o, s, carry[1] != 0, 2, &pending, carry[3] != 0);
    }
    // lengths + swap (utils_match.py:139-146) + cleared scratch: by the vote's sort itself where one workgroup sorts a
    // cloud (PairCountFuse), by count_pair otherwise
    const bool countInSort = N <= kMaxSortN && N <= kChunkSortMinN && o.on(ICPFLOW_OPT_NO_SORTED_VOTE);
    if (!countInSort) {
        launch_count_pair(d_src, d_dst, B, N, w.lenA, w.lenC, w.swap, s, w.ctrl, icp_ctrl_bytes(B), w.scoreAccum,
                          w.accumBytes, w.pairBox);
        w.grid.pairBox = w.pairBox;   // (for the sorts of exactly these clouds, lengths and roles: this call's)
    }
    // the axis sort of both clouds (scoring sweep, ICP) runs on the side stream next to the vote
    hipEvent_t join = nullptr;
    bool teamPlanned = false;
    JoinGuard guard;   // every return below leaves the side stream joined into s
    const GridScratch *search = search_scratch(w, N, o);
    SideStream *side = nullptr;
    if (search != nullptr && search->mode == 3 && N >= 64 && o.on(ICPFLOW_OPT_NO_SIDE_STREAM)) {
        side = &side_stream(s);
        if (!side->ok) side = nullptr;
    }
    if (side != nullptr) {
        ICPFLOW_TRY(hipEventRecord(side->fork, s));
        ICPFLOW_TRY(hipStreamWaitEvent(side->stream, side->fork, 0));
        // from here on the side stream belongs to the caller's stream order (and capture): a failed
        // launch still records the join so that the fork never dangles.  With the counting folded into the sorts
        // nothing on the side stream waits for a kernel of this call: the axis sort counts for itself (selfCount).
        hipError_t se = launch_sort_clouds_soa(d_src, d_dst, w.lenA, w.lenC, w.swap, B, N, &w.grid, side->stream,
                                               countInSort ? 2 : 0);
        // ... and, behind it, the occupancy grids of the two sorted clouds for the scoring's pre-bound (nn.hip)
        if (se == hipSuccess && score_by_sweep(N, true, o) && o.on(ICPFLOW_OPT_NO_SCORE_PRUNE) && o.on(ICPFLOW_OPT_NO_SCORE_PREBOUND)) {
            se = launch_occupancy(&w.grid, B, N, side->stream);
            if (se == hipSuccess) w.grid.occReady = 1;
        }
        // the ICP's team plan reads the lengths and roles only: where count_pair has written them before the fork it runs
        // here, beside the vote, instead of in front of the ICP launch (22-31 us of a serial chain)
        // (not in a first half, phase 1: the plan follows the pair mask, which the second half brings)
        if (se == hipSuccess && !countInSort && phase != 1 &&
            icp_teams_wanted(&w.team, o.icp(w.grid.sortX), search, B, N, max_iterations, stop_mode, w.history)) {
            launch_icp_team_plan(&w.team, w.lenA, w.lenC, w.swap, B, N, o.icp(w.grid.sortX), side->stream);
            teamPlanned = true;
        }
        const hipError_t je = hipEventRecord(side->join, side->stream);
        if (je == hipSuccess) { guard.s = s; guard.join = side->join; }
        ICPFLOW_TRY(se);
        ICPFLOW_TRY(je);
        w.grid.presorted = 1;
        join = side->join;
    }
    PairCountFuse fuse{};
    if (countInSort) {
        fuse.swapOut = w.swap;
        fuse.zero0 = w.ctrl; fuse.bytes0 = icp_ctrl_bytes(B);
        fuse.zero1 = w.scoreAccum; fuse.bytes1 = w.accumBytes;
    }
    w.grid.shareCountClean = 1;   // (cleared with scoreAccum either way; the sweep launches of this call skip their memsets)
    w.grid.sweepTicket = w.ticketScratch;   // (cleared with it as well: the pruned scoring's count of listed scans)
    const bool sweepScore = score_by_sweep(N, join != nullptr, o);
    if (int r = run_init_pose(d_src, d_dst, w, w.swap, B, N, d_edges_x, len_x, d_edges_y, len_y, d_edges_z,
                              len_z, decode_shift, w.Tinit, o, s, sweepScore ? join : nullptr,
                              countInSort ? &fuse : nullptr, join != nullptr))
        return r;
    if (join != nullptr && !sweepScore) ICPFLOW_TRY(hipStreamWaitEvent(s, join, 0));
    guard.joined();
    if (phase == 1) {
        carry[0] = w.grid.presorted;
        carry[1] = teamPlanned ? 1 : 0;
        carry[3] = w.initSumValid ? 1 : 0;
        // ... and, where ONE speculative launch runs the batch rule, the ICP of the whole batch as well (carry[2] = 1): the second
        // half then only has to find where the rule over ITS pairs stops
        return run_icp_and_select(d_src, d_dst, w, w.swap, w.Tinit, B, N, thres_dist, max_iterations, relative_rmse_thr,
                                  stop_mode, 1, d_T_out, d_iters, o, s, teamPlanned, 1, &carry[2]);
    }
    return run_icp_and_select(d_src, d_dst, w, w.swap, w.Tinit, B, N, thres_dist, max_iterations,
                              relative_rmse_thr, stop_mode, 1, d_T_out, d_iters, o, s, teamPlanned, 0, nullptr, w.initSumValid);
}

int icpflow_hist_icp(const float *d_src, const float *d_dst, int B, int N, const float *d_edges_x,
                     int len_x, const float *d_edges_y, int len_y, const float *d_edges_z, int len_z,
                     float decode_shift, double thres_dist, int max_iterations, double relative_rmse_thr,
                     int stop_mode, float *d_T_out, int32_t *d_iters, void *d_ws, size_t ws_bytes,
                     icpflow_stream_t stream, const icpflow_options_t *opt)
{
    Opts o;
    if (int r = parse_options("icpflow_hist_icp", opt, o)) return r;
    if (int r = check_arith("icpflow_hist_icp", o, max_iterations, stop_mode)) return r;
    if (int r = check_pair_active("icpflow_hist_icp", o, true, max_iterations, stop_mode)) return r;
    if (!d_src || !d_dst || !d_edges_x || !d_edges_y || !d_edges_z || !d_T_out)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp: null pointer");
    if (int r = check_batch("icpflow_hist_icp", B, N)) return r;
    if (int r = check_hist_dims("icpflow_hist_icp", len_x, len_y, len_z)) return r;
    if (max_iterations <= 0 || max_iterations > kMaxIterCap)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp: max_iterations must be in 1..%d", kMaxIterCap);
    if (stop_mode != ICPFLOW_STOP_REFERENCE && stop_mode != ICPFLOW_STOP_PER_PAIR)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp: unknown stop_mode %d", stop_mode);
    const size_t L = (size_t)len_x * len_y * len_z;
    if (L < (size_t)kTopK) return fail(ICPFLOW_E_ARG, "icpflow_hist_icp: fewer than 5 bins");
    Workspace w(d_ws, B, N, L);
    if (int r = check_ws(d_ws, ws_bytes, w.bytes)) return r;
    return hist_icp_core(d_src, d_dst, B, N, d_edges_x, len_x, d_edges_y, len_y, d_edges_z, len_z, decode_shift,
                         thres_dist, max_iterations, relative_rmse_thr, stop_mode, d_T_out, d_iters, w, o,
                         (hipStream_t)stream);
}

// K independent batches in flight: batch k runs on worker stream k % 4 of the calling thread, forked from and joined
// back into the caller's stream (stream capture sees one connected graph).  Every batch is one icpflow_hist_icp with its
// own workspace, its own batch-global stop and bit-identical results; what overlaps is the tail of one batch's ICP
// launch (few pairs still iterating, most CUs idle) with the vote and scoring of the others.
int icpflow_hist_icp_many(int K, const float *const *d_src, const float *const *d_dst, const int *B, int N,
                          const float *d_edges_x, int len_x, const float *d_edges_y, int len_y,
                          const float *d_edges_z, int len_z, float decode_shift, double thres_dist,
                          int max_iterations, double relative_rmse_thr, int stop_mode, float *const *d_T_out,
                          int32_t *const *d_iters, void *const *d_ws, const size_t *ws_bytes,
                          icpflow_stream_t stream, const icpflow_options_t *opt)
{
    if (K <= 0 || K > 64) return fail(ICPFLOW_E_ARG, "icpflow_hist_icp_many: K must be in 1..64 (got %d)", K);
    if (!d_src || !d_dst || !B || !d_T_out || !d_iters || !d_ws || !ws_bytes)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp_many: null pointer");
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < k; ++j)
            if (d_ws[k] == d_ws[j]) return fail(ICPFLOW_E_WORKSPACE, "icpflow_hist_icp_many: batches %d and %d share a workspace", j, k);
    hipStream_t s = (hipStream_t)stream;
    constexpr int kWorkers = 4;
    struct Pool {
        hipStream_t w[kWorkers] = {nullptr, nullptr, nullptr, nullptr};
        hipEvent_t fork = nullptr, join[kWorkers] = {nullptr, nullptr, nullptr, nullptr};
        int device = -1;
        bool ok = false;
    };
    static thread_local Pool pool;
    int dev = -1;
    ICPFLOW_TRY(hipGetDevice(&dev));
    if (!pool.ok || pool.device != dev) {
        // (streams and events of another device, if any, are abandoned to that device's context)
        pool = Pool{};
        pool.device = dev;
        ICPFLOW_TRY(hipEventCreateWithFlags(&pool.fork, hipEventDisableTiming));
        for (int k = 0; k < kWorkers; ++k) {
            ICPFLOW_TRY(hipStreamCreateWithFlags(&pool.w[k], hipStreamNonBlocking));
            ICPFLOW_TRY(hipEventCreateWithFlags(&pool.join[k], hipEventDisableTiming));
        }
        pool.ok = true;
    }
    const int used = K < kWorkers ? K : kWorkers;
    ICPFLOW_TRY(hipEventRecord(pool.fork, s));
    for (int k = 0; k < used; ++k) ICPFLOW_TRY(hipStreamWaitEvent(pool.w[k], pool.fork, 0));
    int rc = 0;
    for (int k = 0; k < K && rc == 0; ++k)
        rc = icpflow_hist_icp(d_src[k], d_dst[k], B[k], N, d_edges_x, len_x, d_edges_y, len_y, d_edges_z, len_z, decode_shift,
                              thres_dist, max_iterations, relative_rmse_thr, stop_mode, d_T_out[k], d_iters[k], d_ws[k],
                              ws_bytes[k], pool.w[k % kWorkers], opt);
    // join whatever was enqueued, also on an error path (the workers must not outlive the call)
    for (int k = 0; k < used; ++k) {
        const hipError_t e1 = hipEventRecord(pool.join[k], pool.w[k]);
        const hipError_t e2 = e1 == hipSuccess ? hipStreamWaitEvent(s, pool.join[k], 0) : e1;
        if (rc == 0 && e2 != hipSuccess) rc = hipfail(e2, "icpflow_hist_icp_many: join");
    }
    return rc;
}

// match_eval behind its argument checks.  haveLens: w.lenA / w.lenC already hold the valid-row counts of pcd1 / pcd2;
// sortedByRole: w.grid holds both clouds sorted raw along one axis by hist_icp's sort (by ROLE: the pairs flagged in
// w.swap have pcd1 in the fixed cloud's arrays) -- no count, no sort, the sweeps read what the registration left.
static int match_eval_core(const float *d_pcd1, const float *d_pcd2, const float *d_T, int B, int N, double thres_dist,
                           float *d_errors, float *d_inliers, float *d_ratios, float *d_ious, float *d_translations,
                           float *d_rotations, Workspace &w, const Opts &o, hipStream_t s, bool haveLens,
                           bool sortedByRole)
{
    if (!haveLens) launch_count_pair(d_pcd1, d_pcd2, B, N, w.lenA, w.lenC, nullptr, s);
    const bool reuse = sortedByRole && N <= kMaxSortN && o.on(ICPFLOW_OPT_NO_EVAL_SWEEP);
    // long clouds and large batches: both directions as sorted sweeps (eval_by_sweep)
    if (reuse || eval_by_sweep(B, N, o)) {
        if (!reuse) ICPFLOW_TRY(launch_sort_clouds_soa(d_pcd1, d_pcd2, w.lenA, w.lenC, nullptr, B, N, &w.grid, s));
        ICPFLOW_TRY(launch_sweep_eval(&w.grid, w.lenA, w.lenC, B, N, d_T, (float)thres_dist, w.zsortA, w.partial, s,
                                      reuse ? w.swap : nullptr, o.pairActive));
        ICPFLOW_TRY(launch_eval_epilogue(w.partial, sweep_qblocks(N), w.lenA, w.lenC, d_T, B, d_errors, d_inliers,
                                         d_ratios, d_ious, d_translations, d_rotations, s));
        return 0;
    }
    ICPFLOW_TRY(launch_scan_eval(d_pcd1, d_pcd2, w.lenA, w.lenC, B, N, d_T, (float)thres_dist, w.partial, s));
    ICPFLOW_TRY(launch_eval_epilogue(w.partial, scan_qblocks(N, B), w.lenA, w.lenC, d_T, B, d_errors, d_inliers,
                                     d_ratios, d_ious, d_translations, d_rotations, s));
    return 0;
}

int icpflow_match_eval(const float *d_pcd1, const float *d_pcd2, const float *d_T, int B, int N,
                       double thres_dist, float *d_errors, float *d_inliers, float *d_ratios,
                       float *d_ious, float *d_translations, float *d_rotations, void *d_ws,
                       size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt)
{
    Opts o;
    if (int r = parse_options("icpflow_match_eval", opt, o)) return r;
    if (int r = check_pair_active("icpflow_match_eval", o, false, 0, 0)) return r;
    if (!d_pcd1 || !d_pcd2 || !d_T || !d_errors || !d_inliers || !d_ratios || !d_ious || !d_translations ||
        !d_rotations)
        return fail(ICPFLOW_E_ARG, "icpflow_match_eval: null pointer");
    if (int r = check_batch("icpflow_match_eval", B, N)) return r;
    Workspace w(d_ws, B, N, 0);
    if (int r = check_ws(d_ws, ws_bytes, w.bytes)) return r;
    return match_eval_core(d_pcd1, d_pcd2, d_T, B, N, thres_dist, d_errors, d_inliers, d_ratios, d_ious, d_translations,
                           d_rotations, w, o, (hipStream_t)stream, false, false);
}

// hist_icp + match_eval of the same clouds in one call (utils_match.py:92-93 calls them back to back): the metrics are
// taken on what the registration left in the workspace -- the valid-row counts and both clouds sorted along the fixed
// cloud's longest axis -- instead of counting and sorting again.
static int hist_icp_eval_phase(int phase, int32_t *carry, const float *d_src, const float *d_dst, int B, int N, const float *d_edges_x, int len_x,
                          const float *d_edges_y, int len_y, const float *d_edges_z, int len_z, float decode_shift,
                          double thres_dist, int max_iterations, double relative_rmse_thr, int stop_mode,
                          float *d_T_out, int32_t *d_iters, float *d_errors, float *d_inliers, float *d_ratios,
                          float *d_ious, float *d_translations, float *d_rotations, void *d_ws, size_t ws_bytes,
                          icpflow_stream_t stream, const icpflow_options_t *opt)
{
    Opts o;
    if (int r = parse_options("icpflow_hist_icp_eval", opt, o)) return r;
    if (int r = check_arith("icpflow_hist_icp_eval", o, max_iterations, stop_mode)) return r;
    if (int r = check_pair_active("icpflow_hist_icp_eval", o, true, max_iterations, stop_mode)) return r;
    if (!d_src || !d_dst || !d_edges_x || !d_edges_y || !d_edges_z || !d_T_out || !d_errors || !d_inliers || !d_ratios ||
        !d_ious || !d_translations || !d_rotations)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp_eval: null pointer");
    if (int r = check_batch("icpflow_hist_icp_eval", B, N)) return r;
    if (int r = check_hist_dims("icpflow_hist_icp_eval", len_x, len_y, len_z)) return r;
    if (max_iterations <= 0 || max_iterations > kMaxIterCap)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp_eval: max_iterations must be in 1..%d", kMaxIterCap);
    if (stop_mode != ICPFLOW_STOP_REFERENCE && stop_mode != ICPFLOW_STOP_PER_PAIR)
        return fail(ICPFLOW_E_ARG, "icpflow_hist_icp_eval: unknown stop_mode %d", stop_mode);
    const size_t L = (size_t)len_x * len_y * len_z;
    if (L < (size_t)kTopK) return fail(ICPFLOW_E_ARG, "icpflow_hist_icp_eval: fewer than 5 bins");
    Workspace w(d_ws, B, N, L);
    if (int r = check_ws(d_ws, ws_bytes, w.bytes)) return r;
    hipStream_t s = (hipStream_t)stream;
    if (int r = hist_icp_core(d_src, d_dst, B, N, d_edges_x, len_x, d_edges_y, len_y, d_edges_z, len_z, decode_shift,
                              thres_dist, max_iterations, relative_rmse_thr, stop_mode, d_T_out, d_iters, w, o, s, phase, carry))
        return r;
    if (phase == 1) return 0;
    return match_eval_core(d_src, d_dst, d_T_out, B, N, thres_dist, d_errors, d_inliers, d_ratios, d_ious, d_translations,
                           d_rotations, w, o, s, true, w.grid.presorted != 0);
}

int icpflow_hist_icp_eval(const float *d_src, const float *d_dst, int B, int N, const float *d_edges_x, int len_x,
                          const float *d_edges_y, int len_y, const float *d_edges_z, int len_z, float decode_shift,
                          double thres_dist, int max_iterations, double relative_rmse_thr, int stop_mode,
                          float *d_T_out, int32_t *d_iters, float *d_errors, float *d_inliers, float *d_ratios,
                          float *d_ious, float *d_translations, float *d_rotations, void *d_ws, size_t ws_bytes,
                          icpflow_stream_t stream, const icpflow_options_t *opt)
{
    return hist_icp_eval_phase(0, nullptr, d_src, d_dst, B, N, d_edges_x, len_x, d_edges_y, len_y, d_edges_z, len_z, decode_shift, thres_dist,
                               max_iterations, relative_rmse_thr, stop_mode, d_T_out, d_iters, d_errors, d_inliers, d_ratios, d_ious,
                               d_translations, d_rotations, d_ws, ws_bytes, stream, opt);
}

// icpflow_register_stage in two halves (phase 1: gather + initial poses; phase 2: ICP, roll-back check, metrics)
static int register_stage_phase(int phase, int32_t *carry, const icpflow_tables_t *t, const icpflow_stage_t *st,
                                const icpflow_registration_t *reg, void *d_ws, size_t ws_bytes, icpflow_stream_t stream,
                                const icpflow_options_t *opt)
{
    const char *fn = phase == 1 ? "icpflow_register_stage_begin" : "icpflow_register_stage_finish";
    if (!t || !st || !reg || !carry) return fail(ICPFLOW_E_ARG, "%s: null argument", fn);
    if (!t->d_points_src || !t->d_order_src || !t->d_points_dst || !t->d_order_dst || !st->d_seg || !st->d_clouds || !st->d_result)
        return fail(ICPFLOW_E_ARG, "%s: null pointer", fn);
    const int K = st->K, N = st->N;
    if (int r = check_batch(fn, K, N)) return r;
    const size_t cloud = (size_t)K * N * 4;
    if (phase == 1) {
        if (int r = icpflow_gather_segments(t->d_points_src, t->d_order_src, st->d_seg, st->d_perm, K, N, st->d_clouds, stream)) return r;
        if (int r = icpflow_gather_segments(t->d_points_dst, t->d_order_dst, st->d_seg + (size_t)3 * K, st->d_perm, K, N,
                                            st->d_clouds + cloud, stream))
            return r;
    }
    float *R = st->d_result;
    const size_t k = (size_t)K;
    return hist_icp_eval_phase(phase, carry, st->d_clouds, st->d_clouds + cloud, K, N, reg->d_edges_x, reg->len_x, reg->d_edges_y, reg->len_y,
                               reg->d_edges_z, reg->len_z, reg->decode_shift, reg->thres_dist, reg->max_iterations,
                               reg->relative_rmse_thr, reg->stop_mode, R, reinterpret_cast<int32_t *>(R + 30 * k), R + 16 * k,
                               R + 18 * k, R + 20 * k, R + 22 * k, R + 24 * k, R + 27 * k, d_ws, ws_bytes, stream, opt);
}

int icpflow_register_stage_begin(const icpflow_tables_t *t, const icpflow_stage_t *st, const icpflow_registration_t *reg,
                                 void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt, int32_t *h_carry)
{
    return register_stage_phase(1, h_carry, t, st, reg, d_ws, ws_bytes, stream, opt);
}

int icpflow_register_stage_finish(const icpflow_tables_t *t, const icpflow_stage_t *st, const icpflow_registration_t *reg,
                                  void *d_ws, size_t ws_bytes, icpflow_stream_t stream, const icpflow_options_t *opt, const int32_t *h_carry)
{
    int32_t carry[4] = {h_carry ? h_carry[0] : 0, h_carry ? h_carry[1] : 0, h_carry ? h_carry[2] : 0, h_carry ? h_carry[3] : 0};
    if (!h_carry) return fail(ICPFLOW_E_ARG, "icpflow_register_stage_finish: null argument");
    return register_stage_phase(2, carry, t, st, reg, d_ws, ws_bytes, stream, opt);
}

}  // extern "C"
