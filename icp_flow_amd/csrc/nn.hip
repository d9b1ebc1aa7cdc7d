// nn.hip -- batched nearest-neighbour scans with fused epilogues (gfx950).
//
// One kernel template, four uses:
//   MODE_SCORE : 6 candidate translations x 2 directions per pair (utils_hist.py:86-104)
//   MODE_CHECK : mean NN error of src under the init pose and under the ICP pose
//                (utils_icp.py:27-33)
//   MODE_EVAL  : match_eval's two directions with inlier counts and centroid sums
//                (utils_match.py:160-183)
//   MODE_NN    : plain nearest_neighbor_batch (utils_helper.py:20-30), idx + dist out
// Unlike the reference's un-lengthed knn over the padded N x N grid, the fused modes
// scan only the valid prefixes -- identical results for valid queries.
//
// Grid: 1-D, XCD-aware.  Consecutive workgroup ids land on consecutive XCDs (id % 8), so
// ids are laid out as [group][qblock][job-in-group(8)]: all query blocks of one job share
// an XCD and therefore the L2 copy of that job's target cloud.
#include "scan.hpp"
#include "sortdir.hpp"
#include "kernels.hpp"
#include "posefuse.hpp"

namespace icpflow {

enum ScanMode : int { MODE_SCORE = 0, MODE_CHECK = 1, MODE_EVAL = 2, MODE_NN = 3 };

struct ScanParams {
    // clouds [B,N,4]; `swap[b]` exchanges the roles of A and C for that pair
    const float *A;        // "src" role (queries in the forward direction)
    const float *C;        // "dst" role
    const int32_t *lenA;   // valid prefixes per pair (may be NULL in MODE_NN)
    const int32_t *lenC;
    const uint8_t *swap;   // optional
    int N;                 // rows per pair in A and C (MODE_NN: NQ)
    int NT;                // MODE_NN: rows per pair of the target cloud
    int strideQ, strideT;  // MODE_NN: floats per row
    int njobs;             // jobs in this launch
    int qblocks;           // query blocks per job
    // MODE_SCORE: candidate translations [B,6,3]
    const float *cand;
    // MODE_SCORE in two launches (launch_scan_score_pruned): this launch covers the scans subBegin ..
    // subBegin + subCount - 1 of every pair (scan = 2 * candidate + direction); with `prune` a workgroup
    // first looks at what the earlier workgroups of its scan have summed (`accum`, [B,12]) and leaves when
    // that already exceeds the score of candidate 0
    int subBegin, subCount, prune;
    double *accum;
    // MODE_SCORE, pruned launches: 2 = a workgroup takes 128 query rows and its two halves scan one half of
    // every target tile each (finer blocks: a scan can leave after an eighth of a 1024-point cloud)
    int split;
    // MODE_CHECK: poseA = init [B,4,4], poseB = final [B,4,4];  MODE_EVAL: poseA = T
    const float *poseA;
    const float *poseB;
    float thres;           // MODE_EVAL inlier threshold on the Euclidean distance
    // outputs
    double *partial;       // [njobs, qblocks, kPartial]
    int64_t *idx;          // MODE_NN
    float *dist;           // MODE_NN
    int sqrt_dist;         // MODE_NN
};

template <int Q, int MODE>
__global__ __launch_bounds__(kScanBlock) void nn_scan_kernel(ScanParams p)
{
    __shared__ ScanTile tileMem;
    ScanTile *tile = &tileMem;
    __shared__ double red[(kScanBlock / kWave) * kPartial];

    // ---- XCD-aware decode of the linear workgroup id --------------------------------
    const int lin = blockIdx.x;
    int job = (lin / (8 * p.qblocks)) * 8 + (lin & 7);
    int qb = (lin >> 3) % p.qblocks;
    if (MODE == MODE_SCORE && p.prune) {
        // query block major: the first blocks of ALL scans are dispatched before anybody's second block, so
        // later blocks find the sums of earlier ones (njobs is padded to 8: a scan stays on one XCD)
        const int padded = (p.njobs + 7) & ~7;
        qb = lin / padded;
        job = lin % padded;
    }
    if (job >= p.njobs) return;

    // ---- job -> (query cloud, target cloud, maps) -------------------------------------
    int b, sub;
    if (MODE == MODE_SCORE) {
        b = job / p.subCount;
        sub = p.subBegin + job % p.subCount;
        job = b * 12 + sub;   // the scan's place in the partial records
    }
    else if (MODE == MODE_NN) { b = job; sub = 0; }
    else { b = job >> 1; sub = job & 1; }

    CloudView qc, tc;
    PointXf qxf, txf;
    qxf.kind = XF_NONE; txf.kind = XF_NONE;
    qxf.a = affine_identity(); txf.a = qxf.a;

    if (MODE == MODE_NN) {
        qc.base = p.A + (size_t)b * p.N * p.strideQ; qc.stride = p.strideQ;
        qc.n = p.lenA ? p.lenA[b] : p.N;
        tc.base = p.C + (size_t)b * p.NT * p.strideT; tc.stride = p.strideT;
        tc.n = p.lenC ? p.lenC[b] : p.NT;
    } else {
        const bool sw = p.swap != nullptr && p.swap[b] != 0;
        CloudView a, c;
        a.base = (sw ? p.C : p.A) + (size_t)b * p.N * 4; a.stride = 4; a.n = (sw ? p.lenC : p.lenA)[b];
        c.base = (sw ? p.A : p.C) + (size_t)b * p.N * 4; c.stride = 4; c.n = (sw ? p.lenA : p.lenC)[b];
        PointXf axf;
        axf.a = affine_identity();
        if (MODE == MODE_SCORE) {
            const int k = sub >> 1;
            axf.kind = XF_TRANSLATE;
            const float *t = p.cand + ((size_t)b * 6 + k) * 3;
            axf.a.t[0] = t[0]; axf.a.t[1] = t[1]; axf.a.t[2] = t[2];
        } else if (MODE == MODE_CHECK) {
            axf.kind = XF_AFFINE;
            axf.a = affine_from_pose((sub == 0 ? p.poseA : p.poseB) + (size_t)b * 16);
        } else {  // MODE_EVAL
            axf.kind = XF_AFFINE;
            axf.a = affine_from_pose(p.poseA + (size_t)b * 16);
        }
        const bool backward = (MODE == MODE_SCORE) ? (sub & 1) : (MODE == MODE_EVAL ? (sub == 1) : false);
        if (!backward) { qc = a; qxf = axf; tc = c; }
        else           { qc = c; tc = a; txf = axf; }
    }

    if (MODE == MODE_SCORE && p.prune) {
        // score_k = min(mean forward, mean backward) and only the smallest score matters (utils_hist.py:103-105).
        // Nearest-neighbour distances are non-negative, so the part of this scan's sum that is already known is
        // a lower bound of its mean; once that exceeds candidate 0's forward mean (>= its score) this
        // direction cannot make candidate k the winner, and if the other direction still does, the score is
        // that direction's mean alone.  The scan reports +inf instead of its sum; the pick is unchanged.
        __shared__ int leave;
        // (the sums of the earlier launches: wave 0 reads them one query block per lane, all scans in flight at once, and
        // adds them in block order -- as a loop on one thread they were up to 88 dependent scalar loads; see
        // sweep_scan_kernel)
        __shared__ double scanSum[12];
        if (threadIdx.x < kWave && p.qblocks <= kWave) {
            const int ln = threadIdx.x, q = min(ln, p.qblocks - 1);
            double v[11];
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                const int sc = u == 0 ? 0 : u + 1;                                     // scans 0, 2 .. 11
                v[u] = (u == 0 || p.prune == 2) ? p.partial[((size_t)(b * 12 + sc) * p.qblocks + q) * kPartial] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                if (u > 0 && p.prune != 2) break;
                double t = 0.0;
                for (int qq = 0; qq < p.qblocks; ++qq) t += __shfl(v[u], qq, kWave);   // block order, as before
                if (ln == 0) scanSum[u == 0 ? 0 : u + 1] = t;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const bool sw = p.swap != nullptr && p.swap[b] != 0;
            const float na = (float)(sw ? p.lenC : p.lenA)[b], nc = (float)(sw ? p.lenA : p.lenC)[b];
            // the forward mean of candidate 0 (complete: earlier launch) is an upper bound of its score
            double f0 = 0.0;
            if (p.qblocks <= kWave) f0 = scanSum[0];
            else for (int q = 0; q < p.qblocks; ++q) f0 += p.partial[((size_t)(b * 12 + 0) * p.qblocks + q) * kPartial];
            const float bound = (float)f0 / na;
            if (p.prune == 2) {
                // third launch: the backward scan of candidate 0.  score_0 = min(forward, backward) <= bound, and
                // candidate 0 wins ties (first arg-min): when every other candidate's score -- known by now, exact
                // wherever it is not +inf -- exceeds the bound, candidate 0 is the pick whatever this scan finds
                bool othersOut = true;
                for (int k = 1; k < 6; ++k) {
                    double fk = 0.0, bk = 0.0;
                    if (p.qblocks <= kWave) { fk = scanSum[2 * k]; bk = scanSum[2 * k + 1]; }
                    else for (int q = 0; q < p.qblocks; ++q) {
                        fk += p.partial[((size_t)(b * 12 + 2 * k) * p.qblocks + q) * kPartial];
                        bk += p.partial[((size_t)(b * 12 + 2 * k + 1) * p.qblocks + q) * kPartial];
                    }
                    const float sk = fminf((float)fk / na, (float)bk / nc);
                    othersOut = othersOut && (sk > bound * 1.0001f);   // NaN keeps the candidate in
                }
                leave = othersOut ? 1 : 0;
            } else {
            const double seen = __hip_atomic_load(p.accum + job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float low = (float)(seen / (double)((sub & 1) ? nc : na));
            leave = low > bound * 1.0001f ? 1 : 0;   // NaN / inf bounds never prune
            }
            if (leave) {
                double *o = p.partial + ((size_t)job * p.qblocks + qb) * kPartial;
                o[0] = __builtin_huge_val();
                for (int k = 1; k < kPartial; ++k) o[k] = 0.0;
            }
        }
        __syncthreads();
        if (leave) return;
    }

    const int split = (MODE == MODE_SCORE && Q == 1 && p.split > 1) ? p.split : 1;   // 1, 2 or 4
    const int rowsPerBlock = kScanBlock * Q / split;
    const int q0 = qb * rowsPerBlock;
    const int nq_rows = (MODE == MODE_NN) ? p.N : qc.n;  // rows that get an output / a sum
    if (q0 >= nq_rows) {                                 // whole block beyond the cloud
        // its partial record is still summed by the epilogue kernels: publish zeros
        if (MODE != MODE_NN && threadIdx.x < kPartial)
            p.partial[((size_t)job * p.qblocks + qb) * kPartial + threadIdx.x] = 0.0;
        return;
    }

    // ---- load this lane's queries -----------------------------------------------------
    float qx[Q], qy[Q], qz[Q], ox[Q], oy[Q], oz[Q];
    bool live[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const int i = q0 + q * kScanBlock + (threadIdx.x & (rowsPerBlock - 1));
        live[q] = i < qc.n;
        qx[q] = qy[q] = qz[q] = 0.f;
        ox[q] = oy[q] = oz[q] = 0.f;
        if (live[q]) {
            cloud_load(qc, i, ox[q], oy[q], oz[q]);
            xf_apply(qxf, ox[q], oy[q], oz[q], qx[q], qy[q], qz[q]);
        }
    }

    ScanAcc<Q> acc;
    if (split > 1) {
        // the `split` parts of the workgroup hold the same queries; part h scans share h of every tile and the
        // minima meet in LDS (the minimum of the partial minima is the minimum over all targets)
        __shared__ float partMin[kScanBlock];
        const int part = threadIdx.x / rowsPerBlock;
        scan_cloud<Q>(tc, txf, tile, qx, qy, qz, acc, part, split);
        partMin[threadIdx.x] = acc.best[0];
        __syncthreads();
        if (part == 0) {
            for (int h = 1; h < split; ++h) acc.best[0] = fminf(acc.best[0], partMin[threadIdx.x + h * rowsPerBlock]);
        } else {
            live[0] = false;   // its rows are summed by part 0
        }
    } else {
        scan_cloud<Q>(tc, txf, tile, qx, qy, qz, acc);
    }

    if (MODE == MODE_NN) {
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int i = q0 + q * kScanBlock + threadIdx.x;
            if (i >= p.N) continue;
            int64_t j = 0;
            float d = 0.f;  // rows >= len keep idx 0 / dist 0 (pytorch3d)
            if (live[q] && tc.n > 0) {
                float nx, ny, nz;
                j = scan_resolve(tc, txf, qx[q], qy[q], qz[q], acc.best[q], acc.chunk[q], nx, ny, nz);
                d = p.sqrt_dist ? sqrtf(acc.best[q]) : acc.best[q];
            }
            p.idx[(size_t)b * p.N + i] = j;
            p.dist[(size_t)b * p.N + i] = d;
        }
        return;
    }

    // ---- fused epilogue: masked sums over this block's queries ---------------------------
    double v[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) v[k] = 0.0;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        if (!live[q]) continue;
        const float d = sqrtf(acc.best[q]);  // utils_helper.py:30
        v[0] += (double)d;
        if (MODE == MODE_EVAL) {
            v[1] += (d < p.thres) ? 1.0 : 0.0;  // utils_match.py:168 (strict <)
            if (sub == 0) {
                v[2] += (double)qx[q]; v[3] += (double)qy[q]; v[4] += (double)qz[q];
                v[5] += (double)ox[q]; v[6] += (double)oy[q]; v[7] += (double)oz[q];
            }
        }
    }
    block_sum<kPartial, double>(v, red);
    if (threadIdx.x == 0) {
        double *o = p.partial + ((size_t)job * p.qblocks + qb) * kPartial;
#pragma unroll
        for (int k = 0; k < kPartial; ++k) o[k] = v[k];
        if (MODE == MODE_SCORE && p.accum != nullptr) atomicAdd(p.accum + job, v[0]);
    }
}

// Queries per lane.  More queries per lane amortise the LDS broadcast reads (Q = 4: VALU bound), but a
// workgroup then owns 256 Q queries against the WHOLE target cloud, and on small (frame-level) batches
// of ragged clusters the few workgroups of the largest pair are the critical path: stay at Q = 2 there.
static int scan_queries_per_lane(int maxRows, int batch)
{
    if (maxRows <= 512) return 1;
    if (maxRows <= 1024 || batch <= 128) return 2;
    return 4;
}

template <int MODE>
static hipError_t launch_scan(ScanParams p, int maxRows, int batch, hipStream_t s)
{
    const int Q = scan_queries_per_lane(maxRows, batch);
    p.qblocks = (maxRows + kScanBlock * Q - 1) / (kScanBlock * Q);
    const int groups = (p.njobs + 7) / 8;
    const dim3 grid((unsigned)(groups * 8 * p.qblocks));
    if (Q == 1) hipLaunchKernelGGL((nn_scan_kernel<1, MODE>), grid, dim3(kScanBlock), 0, s, p);
    else if (Q == 2) hipLaunchKernelGGL((nn_scan_kernel<2, MODE>), grid, dim3(kScanBlock), 0, s, p);
    else hipLaunchKernelGGL((nn_scan_kernel<4, MODE>), grid, dim3(kScanBlock), 0, s, p);
    return hipGetLastError();
}

int scan_qblocks(int maxRows, int batch)
{
    const int Q = scan_queries_per_lane(maxRows, batch);
    return (maxRows + kScanBlock * Q - 1) / (kScanBlock * Q);
}

// ---------------------------------------------------------------------------------
// Candidate scoring as a sorted SWEEP (a-3): the same sums as MODE_SCORE above -- for each of the
// 6 candidate translations the sum of nearest-neighbour distances src + t -> dst and dst -> src + t
// -- but a wave of 64 consecutive sorted queries only visits the targets that can be anybody's
// nearest neighbour.  Both clouds are sorted along the dst role's longest axis u (a translation
// does not change the order).  Pass 1 scans the targets whose u lies within r0 of the wave's
// queries; that yields an upper bound R on every lane's NN distance (R = the largest minimum found,
// infinite if some lane saw no target at all); if R > r0, pass 2 scans the rest of the window of
// radius R.  A target outside that window differs from every query of the wave by more than R along
// u alone, so the minima -- evaluated with the same instruction sequence as the all-pairs scan --
// are bit-identical; only the order of the (fp64) sum over queries changes.
// One query per lane, targets streamed through scalar loads, no LDS, no barriers in the scan.
// ---------------------------------------------------------------------------------
struct SweepParams {
    const float *Asoa;      // [B,3,NP16] src role sorted along axis[b], +inf padded (SWEEP_SCORE)
    const float *Csoa;      // [B,3,NP16] dst role
    const int32_t *lenA, *lenC;
    const uint8_t *swap;
    const int32_t *axis;
    const float *cand;      // SWEEP_SCORE: [B,6,3]
    // SWEEP_CHECK: the sorted moving cloud of the ICP (float4: point, original row in w) and the clouds
    // themselves; rawSorted != 0: the sorted points are the raw cloud, else they carry the ICP's pre-pose
    // and the raw point is fetched through w
    const float4 *sortX;
    const float *X, *Y;     // [B,N,4] as passed to the registration (src, dst)
    const float *poseA, *poseB;   // [B,4,4] init pose and composed final pose
    PoseSource fused;             // poseB == NULL: the final pose is composed in the kernel (posefuse.hpp)
    int rawSorted;
    // SWEEP_EVAL (match_eval): Asoa = pcd1 sorted (raw), Csoa = pcd2 sorted, srcT = pcd1 * T in pcd1's sorted
    // order (transform_soa_kernel), poseA = T, thres = inlier threshold on the Euclidean distance
    const float *srcT;
    float thres;
    int N, NP16, njobs, qblocks;
    float r0;
    double *partial;        // [njobs, qblocks, kPartial]
    // SWEEP_SCORE with branch and bound (launch_sweep_score_pruned; same scheme as ScanParams above): this launch
    // covers the scans subBegin .. subBegin + subCount - 1 of every pair; prune 1: leave when the blocks before
    // this one have summed more than candidate 0's forward mean allows, prune 2: candidate 0's backward scan
    int subBegin, subCount, prune;
    double *accum;          // [B,12] running sums of the scans (cleared by the caller)
    float *shareBest;       // [jobs, kSweepFullQb, kSweepShares, kSweepBlock] partial minima of query blocks scanned by several blocks, or NULL
    int *shareCount;        // [jobs, kSweepFullQb] blocks that have delivered (zero before the launch; the last one resets it)
    int shareClean;         // 1: shareCount is known to be zero (GridScratch.shareCountClean): no memset in front of the launch
    int shareWindows;       // 1: blocks of at most 64 queries against a long cloud split every range over their four waves (sweep_scan_kernel)
    const int32_t *pairTab;     // optional [B, 4]: what a workgroup needs to know to find out that it has no rows (sweep_pair_table_kernel), or NULL
    const int32_t *pairOrder;   // optional [B]: the pair the k-th group of jobs works on (largest pairs first: vote_plan_kernel), or NULL
    const uint8_t *active;  // SWEEP_CHECK / SWEEP_EVAL: optional [B], 0 = the pair is not in the batch (options.d_pair_active): its records are zeros
    // SWEEP_CHECK in hist_icp: optional [B], the scoring's forward total of the candidate score_pick_kernel picked -- the very sum
    // the scan under the initial pose (sub 0) would form: where it is finite that scan is left out and select_kernel reads
    // the total instead (+inf: the scan was pruned, the check scans for itself)
    const double *initSum;
    // SWEEP_SCORE with prune == 1: optional occupancy grids of both sorted clouds (GridScratch.occHdr / occBits): a scan first bounds
    // its WHOLE sum from below by (queries in cells without a target in the 27-neighbourhood) x 0.98 h and leaves if that rules it out
    const float *occHdr;
    const uint32_t *occBits;
    // The pruned scoring launch in two (launch_sweep_score_pruned): listMode 1 = the DECIDING launch, one block per scan (query block
    // 0): prologue and occupancy pre-bound, then either the scan's records (+inf) or its number appended to `list`; listMode 2 = the
    // launch of the LISTED scans: job lin = (query block lin / count, listed scan lin % count), drawn by sweep_list_kernel.
    int listMode;
    int *list, *listCount;
    int boundBoth;          // SWEEP_SCORE, prune == 1: candidate 0's BACKWARD scan is complete as well (the first launch ran both): the bound is its score, min(forward, backward)
};

#ifdef ICPFLOW_OCC_STATS
__device__ unsigned long long g_occStats[12];   // tools/dbg/prebound_stats.py
__device__ unsigned int g_occPair[1024];
#endif
constexpr int kSweepBlock = 256;
#ifndef ICPFLOW_SWEEP_SHARE_MIN_TARGETS
#define ICPFLOW_SWEEP_SHARE_MIN_TARGETS 512
#endif
constexpr int kSweepShareMinTargets = ICPFLOW_SWEEP_SHARE_MIN_TARGETS;
#ifndef ICPFLOW_SWEEP_FULLSCAN_MIN_TARGETS
#define ICPFLOW_SWEEP_FULLSCAN_MIN_TARGETS 2048
#endif
constexpr int kSweepFullScanMinTargets = ICPFLOW_SWEEP_FULLSCAN_MIN_TARGETS;
constexpr int kSweepShares = 8;   // blocks that share ONE query block of a small cloud against a long one
#ifndef ICPFLOW_SWEEP_FULLSCAN_QBLOCKS
#define ICPFLOW_SWEEP_FULLSCAN_QBLOCKS 2
#endif
constexpr int kSweepFullQb = ICPFLOW_SWEEP_FULLSCAN_QBLOCKS;   // ... of clouds of up to that many query blocks
#ifndef ICPFLOW_SWEEP_FULLSCAN_MIN_NT
#define ICPFLOW_SWEEP_FULLSCAN_MIN_NT 512
#endif
constexpr int kSweepFullScanMinNt = ICPFLOW_SWEEP_FULLSCAN_MIN_NT;   // ... against at least that many targets
static_assert(kSweepFullQb * kSweepShares <= kSweepShareSlots, "GridScratch.shareBest holds kSweepShareSlots records of 256 minima per job");
constexpr int kSweepStage = 4096;   // sort keys staged in LDS up to this many targets
enum SweepMode : int { SWEEP_SCORE = 0, SWEEP_CHECK = 1, SWEEP_EVAL = 2 };

#ifdef ICPFLOW_SWEEP_CLOCK
__device__ long long g_sweep_clk[4096 * 8];   // per (mode 1 job, block 0): wall start, wall end, shader clocks: entry->loop, loop, rounds, nt, nq, chunks scanned
#endif
// SHARE: the instantiation whose blocks may share a window (one-wave blocks: over their four waves; one-block clouds against a long
// one: over several blocks) -- batches of a few hundred pairs; batches that fill the GPU many times over keep the plain loop
// One block's job (the kernel's body since round 5's end: sweep_scan_kernel runs it for lin = blockIdx.x, sweep_list_kernel for the
// jobs of the listed scans).  Every `return` below is taken by the whole block alike.
template <int MODE, bool SHARE>
__device__ __forceinline__ void sweep_job(const SweepParams &p, const int lin)
{
#ifdef ICPFLOW_SWEEP_CLOCK
    const long long dbgW0 = wall_clock64(), dbgC0 = clock64();
    long long dbgC1 = 0, dbgC2 = 0;
    int dbgRounds = 0, dbgChunks = 0;
#endif
    __shared__ double red[(kSweepBlock / kWave) * kPartial];
    extern __shared__ __attribute__((aligned(16))) float keyLds[];   // the targets' sort keys (window searches)
    int job = (lin / (8 * p.qblocks)) * 8 + (lin & 7);   // XCD-aware: the 8 XCDs take 8 jobs
    int qb = (lin >> 3) % p.qblocks;
    if (MODE == SWEEP_SCORE && p.prune) {   // query block major (see nn_scan_kernel): later blocks find earlier sums
        const int padded = (p.njobs + 7) & ~7;
        qb = lin / padded;
        job = lin % padded;
        if (p.listMode == 2) {   // (the caller has checked lin < count * qblocks)
            const int cnt = *p.listCount;
            qb = lin / cnt;
            job = p.list[lin % cnt];
        }
    }
    if (MODE == SWEEP_CHECK && p.initSum != nullptr) {
        // Two halves, each dealt like the whole (eight pairs to the eight XCDs): the scans under the final pose first, then
        // those under the initial pose, most of which leave at once (below).  With the two scans of a pair side by side in
        // the grid the working ones would all sit on the odd XCDs.
        const int pairs = p.njobs >> 1;
        const int half = ((pairs + 7) / 8) * 8 * p.qblocks;
        const int l = lin < half ? lin : lin - half;
        const int k = (l / (8 * p.qblocks)) * 8 + (l & 7);
        qb = (l >> 3) % p.qblocks;
        if (k >= pairs) return;
        job = k * 2 + (lin < half ? 1 : 0);
    }
    if (job >= p.njobs) return;
    const int listedAs = job;   // (the scan's number in this launch's own numbering: what the list holds)
    int b = (MODE == SWEEP_SCORE) ? job / p.subCount : job >> 1;
    const int sub = (MODE == SWEEP_SCORE) ? p.subBegin + job % p.subCount : (job & 1);
    // Nine workgroups in ten of a ragged batch are launched for rows their cloud does not have (the grid covers the padded
    // width) and used to find that out behind THREE dependent loads (pair order -> swap flag -> lengths): ~3 us each, 58 000 of
    // them per launch, in front of the working ones in dispatch order -- the shares of a job scanned by several workgroups
    // started 60 us into a 113 us launch (tools/dbg/sweep_clocks.py).  One load from the pair table tells such a workgroup the
    // pair and that it has nothing to do.
    if (SHARE && p.pairTab != nullptr && (MODE == SWEEP_SCORE || p.active == nullptr)) {
        const bool bw = (MODE == SWEEP_SCORE) ? (sub & 1) : (MODE == SWEEP_EVAL ? sub == 1 : false);
        const int e = p.pairTab[b * 4 + (MODE == SWEEP_EVAL ? 2 : 0) + (bw ? 1 : 0)];
        const int blocks = (p.shareCount != nullptr) ? ((e >> 8) & 255) : (e & 255);   // with / without jobs scanned by several workgroups
        if (qb >= blocks) {
            const int bb = e >> 16;
            double *rec = p.partial + ((size_t)((MODE == SWEEP_SCORE) ? bb * 12 + sub : bb * 2 + sub) * p.qblocks + qb) * kPartial;
            if (threadIdx.x < kPartial) rec[threadIdx.x] = 0.0;
            if (MODE == SWEEP_SCORE && p.listMode == 1)   // (deciding launch: all of the scan's records)
                for (int k = kPartial + threadIdx.x; k < p.qblocks * kPartial; k += kSweepBlock) rec[k] = 0.0;
            return;
        }
    }
    // (dispatch order is job order: with the pairs taken largest first the long jobs of a ragged batch start at once instead of
    // wherever the batch put them; records, counters and sums keep the pair's own place)
    if (p.pairOrder != nullptr) b = p.pairOrder[b];
    job = (MODE == SWEEP_SCORE) ? b * 12 + sub : b * 2 + sub;   // the scan's place in the partial records
    if (MODE == SWEEP_SCORE && p.listMode == 1) {
        // deciding launch: nobody else writes this scan's records unless it gets listed -- the later query blocks' first (zeros: a
        // scan that ends here reports through block 0's record alone)
        double *rec = p.partial + ((size_t)job * p.qblocks + 1) * kPartial;
        for (int k = threadIdx.x; k < (p.qblocks - 1) * kPartial; k += kSweepBlock) rec[k] = 0.0;
    }
    if (MODE == SWEEP_SCORE && p.prune == 1 && p.occBits != nullptr && qb > 0) {
        // a scan that its block 0 has ended by the occupancy pre-bound (below): +inf in its running sum -- one load and out
        const double seen = __hip_atomic_load(p.accum + job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen == __builtin_huge_val()) {
            double *rec = p.partial + ((size_t)job * p.qblocks + qb) * kPartial;
            if (threadIdx.x < kPartial) rec[threadIdx.x] = threadIdx.x == 0 ? __builtin_huge_val() : 0.0;
            return;
        }
    }
    if (MODE == SWEEP_CHECK && sub == 0 && p.initSum != nullptr) {   // (every block of the job decides alike; nobody reads its records)
        const double t = p.initSum[b];
        if (t - t == 0.0 && (p.active == nullptr || p.active[b] != 0)) return;
    }
    const bool backward = (MODE == SWEEP_SCORE) ? (sub & 1) : (MODE == SWEEP_EVAL ? sub == 1 : false);
    const bool sw = p.swap != nullptr && p.swap[b] != 0;
    const int na = (sw ? p.lenC : p.lenA)[b], nc = (sw ? p.lenA : p.lenC)[b];
    const float *as = p.Asoa + (size_t)b * 3 * p.NP16, *cs = p.Csoa + (size_t)b * 3 * p.NP16;
    int naE = na, ncE = nc;
    if (MODE == SWEEP_EVAL && sw) {   // clouds sorted by ROLE (hist_icp's sort): pcd1 sits in the dst role's array
        const float *t = as; as = cs; cs = t;
        naE = nc; ncE = na;
    }
    // queries come from qs; the scan reads ts; the window searches read the (exactly sorted) key row of ks
    const float *qs = backward ? cs : as, *ts = backward ? as : cs, *ks = ts;
    if (MODE == SWEEP_EVAL) {
        const float *st = p.srcT + (size_t)b * 3 * p.NP16;
        if (!backward) qs = st;            // src * T against dst
        else { ts = st; ks = as; }          // dst against src * T, windows in the raw frame of src
    }
    const int nq = backward ? ncE : naE, nt = backward ? naE : ncE;
    double *out = p.partial + ((size_t)job * p.qblocks + qb) * kPartial;
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
    __shared__ float boundSh;   // pruned scoring: candidate 0's forward mean
    __shared__ int prunedSh;    // ... a wave of this block has proven the scan out of the race
    const bool off = MODE != SWEEP_SCORE && p.active != nullptr && p.active[b] == 0;
    // A small cloud (ONE query block) against a long one: every wave would walk thousands of targets while the job's other
    // blocks -- launched for the padded width -- return at once (round 5: 0.12-0.2 ms for such a block, the length of the launch).
    // The first kSweepShares blocks of the job take the SAME queries and one share of ALL targets each (no window: a handful of
    // queries spans the other cloud anyway); partial minima go to global memory, the last block to deliver combines them and
    // writes block 0's record.  Minima are exact: the sums are bit for bit those of the windowed scan.
    // (Second half of round 5: also clouds of TWO query blocks -- a sparse query cloud spreads a wave's 64 sorted queries over
    // a sixth of the other cloud's length, and the window of such a wave is most of it anyway: the jobs with 300-400 queries
    // against 7-9 k targets paced the check sweep of the ragged batch at 60-70 us a block.  Block qb of the job takes query
    // block qb % nqb and share qb / nqb of the targets.)
    const int nqb = (nq + kSweepBlock - 1) / kSweepBlock;
    const bool fullScan = SHARE && !off && p.shareCount != nullptr && nq > 0 && nqb <= kSweepFullQb && nt >= kSweepFullScanMinNt && p.qblocks >= 2 * nqb;
    const int shares = fullScan ? min(p.qblocks / nqb, kSweepShares) : 1;
    const int qi = fullScan ? qb % nqb : qb, si = fullScan ? qb / nqb : 0;   // query block, share of the targets
    if (fullScan && qb >= nqb && qb < shares * nqb) {
        if (threadIdx.x < kPartial) out[threadIdx.x] = 0.0;    // (this block's own record: a block beyond the cloud)
        out = p.partial + ((size_t)job * p.qblocks + qi) * kPartial;
    } else
    if (off || qb * kSweepBlock >= nq) {   // block beyond the cloud (or a pair that is not in the batch): its record is still summed
        // (before the pruning prologue: on a batch padded far beyond its clusters -- a frame's candidate pairs at max_points
        // 10000 -- nine blocks in ten are such blocks, and the prologue reads and adds the sums of up to eleven scans.  A
        // scan that the prologue would have declared out of the race here reports its true sum instead of +inf: above the
        // bound either way, the pick is the same.)
        if (threadIdx.x < kPartial) out[threadIdx.x] = 0.0;
        return;
    }
    if (MODE == SWEEP_SCORE && threadIdx.x == 0) prunedSh = 0;
    if (MODE == SWEEP_SCORE && p.prune && !fullScan) {   // (a job shared by several blocks is scanned to the end: its blocks must agree)
        // branch and bound exactly as in nn_scan_kernel (the argument is written there): the sum of the blocks
        // before this one bounds the scan's mean from below; beyond candidate 0's forward mean it reports +inf
        __shared__ int leave;
        // The sums of earlier launches (candidate 0's forward scan; in the third launch every other scan too) are read by
        // wave 0, one query block per lane and all scans in flight at once, then added in block order.  (As a loop on one
        // thread these were 8, in the third launch 88, dependent scalar loads of a few hundred nanoseconds each: the
        // whole third launch, and the first microseconds of every block of the second.)
        __shared__ double scanSum[12];
        if (wave == 0 && p.qblocks <= kWave) {
            const int q = min(lane, p.qblocks - 1);
            double v[11];
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                const int sc = u == 0 ? 0 : u + 1;                                     // scans 0, 2 .. 11
                v[u] = (u == 0 || p.prune == 2) ? p.partial[((size_t)(b * 12 + sc) * p.qblocks + q) * kPartial] : 0.0;
            }
            if (p.boundBoth) {   // candidate 0's backward scan, complete since the first launch
                const double v1 = p.partial[((size_t)(b * 12 + 1) * p.qblocks + q) * kPartial];
                double t = 0.0;
                for (int qq = 0; qq < p.qblocks; ++qq) t += __shfl(v1, qq, kWave);
                if (lane == 0) scanSum[1] = t;
            }
#pragma unroll
            for (int u = 0; u < 11; ++u) {
                if (u > 0 && p.prune != 2) break;
                double t = 0.0;
                for (int qq = 0; qq < p.qblocks; ++qq) t += __shfl(v[u], qq, kWave);   // block order, as before
                if (lane == 0) scanSum[u == 0 ? 0 : u + 1] = t;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double f0 = 0.0;
            if (p.qblocks <= kWave) f0 = scanSum[0];
            else for (int q = 0; q < p.qblocks; ++q) f0 += p.partial[((size_t)(b * 12 + 0) * p.qblocks + q) * kPartial];
            float bound = (float)f0 / (float)na;
            if (p.boundBoth) {   // candidate 0's score as score_pick_kernel forms it: fminf(forward mean, backward mean)
                double b0 = 0.0;
                if (p.qblocks <= kWave) b0 = scanSum[1];
                else for (int q = 0; q < p.qblocks; ++q) b0 += p.partial[((size_t)(b * 12 + 1) * p.qblocks + q) * kPartial];
                bound = fminf(bound, (float)b0 / (float)nc);
            }
            boundSh = bound;
            if (p.prune == 2) {
                bool othersOut = true;
                for (int k = 1; k < 6; ++k) {
                    double fk = 0.0, bk = 0.0;
                    if (p.qblocks <= kWave) { fk = scanSum[2 * k]; bk = scanSum[2 * k + 1]; }
                    else for (int q = 0; q < p.qblocks; ++q) {
                        fk += p.partial[((size_t)(b * 12 + 2 * k) * p.qblocks + q) * kPartial];
                        bk += p.partial[((size_t)(b * 12 + 2 * k + 1) * p.qblocks + q) * kPartial];
                    }
                    const float sk = fminf((float)fk / (float)na, (float)bk / (float)nc);
                    othersOut = othersOut && (sk > bound * 1.0001f);   // NaN keeps the candidate in
                }
                leave = othersOut ? 1 : 0;
            } else {
                const double seen = __hip_atomic_load(p.accum + job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const float low = (float)(seen / (double)(float)(backward ? nc : na));
                leave = low > bound * 1.0001f ? 1 : 0;   // NaN / inf bounds never prune
            }
            if (leave) {
                out[0] = __builtin_huge_val();
                for (int k = 1; k < kPartial; ++k) out[k] = 0.0;
            }
        }
        __syncthreads();
        if (leave) return;
        // The occupancy pre-bound (round 5, end): candidates other than the vote's winner are other peaks of the histogram, six bins
        // (0.6 m) or more away, or the zero translation -- the moved cloud mostly lies where the other cloud has nothing.  88 % of the
        // scanning waves of this launch used to be pruned only after ~2 rounds and ~160 evaluated targets each: here every
        // block first counts, over ALL queries of its scan, those whose cell of the other cloud's dilated occupancy grid is empty: such a
        // query has no target within 0.98 h (cells differ by two or more along some axis; the cell arithmetic's rounding, < 0.5 % of a cell,
        // is inside the 2 % margin), so count x 0.98 h bounds the scan's sum from below before a single target has been evaluated.
        // A bound only ever ends a scan whose mean provably exceeds
        // candidate 0's forward mean: picks and sums are unchanged (ICPFLOW_OPT_NO_SCORE_PREBOUND; test_scoring_variants_change_nothing).
        // ONE block per scan does it -- query block 0, dispatched before the others (query-block-major grid) -- and, where the bound ends
        // the scan, leaves +inf in the scan's running sum: the prologue above then ends every later block of the scan at once.  (A block
        // that starts before block 0 has got that far simply scans as it always did.)
        if (p.prune == 1 && p.occBits != nullptr && sub != 1 && qb == 0 && p.listMode != 2) {
            __shared__ int occCount[kSweepBlock / kWave];
            __shared__ int occLeave;
            const float *t3 = p.cand + ((size_t)b * 6 + (sub >> 1)) * 3;
            const float ptx = t3[0], pty = t3[1], ptz = t3[2];
            const int role = backward ? 0 : 1;   // the targets' cloud: src role for a backward scan, dst role for a forward one
            const float *hdr = p.occHdr + ((size_t)b * 2 + role) * 8;
            const uint32_t *bits = p.occBits + ((size_t)b * 2 + role) * kOccRings * kOccWords;
            const float gox = hdr[0], goy = hdr[1], goz = hdr[2], ginv = hdr[3], hlb = hdr[4];
            const int gnx = __float_as_int(hdr[5]), gny = __float_as_int(hdr[6]), gnz = __float_as_int(hdr[7]);
            int cnt = 0;   // sum over the queries of their ring level (0 .. kOccRings): level L = no target within L cells
            if (gnx > 0) {
                for (int i = threadIdx.x; i < nq; i += kSweepBlock) {
                    float x = qs[i], y = qs[p.NP16 + i], z = qs[2 * p.NP16 + i];
                    if (!backward) { x += ptx; y += pty; z += ptz; }   // the moved source cloud against dst
                    else { x -= ptx; y -= pty; z -= ptz; }             // dst against the moved source cloud: |c - (a + t)| = |(c - t) - a| to rounding
                    const float ux = floorf((x - gox) * ginv), uy = floorf((y - goy) * ginv), uz = floorf((z - goz) * ginv);
                    // (NaN coordinates fail the range test below and count as "occupied": no bound from them)
                    const bool inside = ux >= 0.f && ux < (float)gnx && uy >= 0.f && uy < (float)gny && uz >= 0.f && uz < (float)gnz;
                    const bool finite = fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f;
                    int level = (finite && !inside) ? kOccRings : 0;   // beyond the grid: kOccRings + 1 cells or more from every target's cell
                    if (inside) {
                        const int c = ((int)ux * gny + (int)uy) * gnz + (int)uz;
#pragma unroll
                        for (int ring = 0; ring < kOccRings; ++ring)   // (the planes are nested: empty in plane k implies empty in the planes below)
                            level += ((bits[ring * kOccWords + (c >> 5)] >> (c & 31)) & 1u) == 0u ? 1 : 0;
                    }
                    cnt += level;
                }
            }
            cnt = wave_sum(cnt);
            if (lane == 0) occCount[wave] = cnt;
            __syncthreads();
            if (threadIdx.x == 0) {
                int tot = 0;
                for (int w = 0; w < kSweepBlock / kWave; ++w) tot += occCount[w];
                const float low = ((float)tot * hlb * 0.999f) / (float)(backward ? nc : na);
                occLeave = low > boundSh * 1.0001f ? 1 : 0;   // NaN / inf bounds never prune
#ifdef ICPFLOW_OCC_STATS
                atomicAdd(&g_occStats[0], 1ull); if (occLeave) atomicAdd(&g_occStats[1], 1ull);
                if (!occLeave && b < 1024) atomicAdd(&g_occPair[b], 1u);   // scans of the pair that go on (tools/dbg/order_predictor.py)
                atomicAdd(&g_occStats[2], (unsigned long long)tot); atomicAdd(&g_occStats[3], (unsigned long long)nq);
                const float ratio = low / boundSh;   // histogram of (pre-bound / bound) in steps of 0.25, [4 .. 11]
                atomicAdd(&g_occStats[4 + min(7, max(0, (int)(ratio * 4.0f)))], 1ull);
#endif
                if (occLeave) {
                    out[0] = __builtin_huge_val();
                    for (int k = 1; k < kPartial; ++k) out[k] = 0.0;
                    atomicAdd(p.accum + job, __builtin_huge_val());   // (the running LOWER bound of the scan's sum: +inf ends the scan)
                }
            }
            __syncthreads();
            if (occLeave) return;
        }
    }
    if (MODE == SWEEP_SCORE && p.listMode == 1) {   // deciding launch: the scan goes on -- in the launch of the listed scans
        if (threadIdx.x == 0) p.list[atomicAdd(p.listCount, 1)] = listedAs;
        return;
    }
    __shared__ float poseSh[16];
    if (MODE == SWEEP_CHECK) {   // the pose of this job: init (sub 0) or final (sub 1; composed here when fused)
        if (sub == 1 && p.poseB == nullptr) {
            if (wave == 0) {
                const int n = pose_stop_iteration_wave(p.fused, lane);
                if (lane == 0) {
                    float M[16];
                    final_pose(p.fused, b, n, M);
#pragma unroll
                    for (int k = 0; k < 16; ++k) poseSh[k] = M[k];
                }
            }
        } else if (threadIdx.x < 16) {
            poseSh[threadIdx.x] = ((sub == 0 ? p.poseA : p.poseB) + (size_t)b * 16)[threadIdx.x];
        }
        __syncthreads();
    }
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (MODE == SWEEP_SCORE) {
        const float *t3 = p.cand + ((size_t)b * 6 + (sub >> 1)) * 3;
        tx = t3[0]; ty = t3[1]; tz = t3[2];
    }
    const int axis = p.axis[b];   // the pair's sort key (sortdir.hpp): a coordinate or a horizontal direction
    float dirX = 0.f, dirY = 0.f;
    if (axis >= 3) sort_dir(axis, dirX, dirY);
    const float tu = sort_key_of(axis, dirX, dirY, tx, ty, tz);
    // A block whose queries fit ONE wave while the other cloud is long (a 40-point cluster against a 7000-point one: real
    // candidate pairs of very unequal size) used to leave three of its four waves without rows while the one wave walked
    // thousands of targets -- 0.2 ms for one such block, the length of the whole launch (round 5: tools/dbg/sweep_clocks.py).
    // There all four waves take the SAME 64 queries and a quarter of every range each; the minima meet in LDS after every
    // round.  A minimum does not depend on who found it: the sums are bit for bit those of one wave scanning alone.
    __shared__ float shareBest[kSweepBlock / kWave][kWave];
    __shared__ int shareLeave;
    const bool sharedWindow = SHARE && !fullScan && p.shareWindows != 0 && nq - qb * kSweepBlock <= kWave && nt >= kSweepShareMinTargets;   // (block-uniform)
    const int i = qi * kSweepBlock + (sharedWindow ? 0 : wave * kWave) + lane;
    const bool live = i < nq;
    float qx = 0.f, qy = 0.f, qz = 0.f, ox = 0.f, oy = 0.f, oz = 0.f;
    float r = p.r0, shrink = 1.0f;   // shrink: the proof radius along u relative to the search radius
    float cu = 0.f;                  // position of the query along u in the frame of the target keys
    if (MODE == SWEEP_SCORE) {
        if (live) {
            qx = qs[i]; qy = qs[p.NP16 + i]; qz = qs[2 * p.NP16 + i];
            if (!backward) { qx += tx; qy += ty; qz += tz; }     // the moved source cloud, as the reference forms it
        }
        cu = sort_key_of(axis, dirX, dirY, qx, qy, qz);
        if (backward) cu -= tu;
    } else if (MODE == SWEEP_CHECK) {
        if (live) {
            // src role point i of the ICP's sorted order, moved by the init (sub 0) or final (sub 1) pose
            // exactly as transform_points_batch does (utils_icp.py:21,27-33)
            const float4 s4 = p.sortX[(size_t)b * p.N + i];
            float rx = s4.x, ry = s4.y, rz = s4.z;
            if (!p.rawSorted) {
                const float4 r4 = reinterpret_cast<const float4 *>(sw ? p.Y : p.X)[(size_t)b * p.N + __float_as_int(s4.w)];
                rx = r4.x; ry = r4.y; rz = r4.z;
            }
            const Affine pose = affine_from_pose(poseSh);
            affine_apply(pose, rx, ry, rz, qx, qy, qz);
        }
        cu = sort_key_of(axis, dirX, dirY, qx, qy, qz);
    } else {   // SWEEP_EVAL
        if (live) {
            qx = qs[i]; qy = qs[p.NP16 + i]; qz = qs[2 * p.NP16 + i];
            if (!backward) { ox = as[i]; oy = as[p.NP16 + i]; oz = as[2 * p.NP16 + i]; }   // the untransformed src point
        }
        cu = sort_key_of(axis, dirX, dirY, qx, qy, qz);
        if (backward) {
            // the targets src * T are ordered by their RAW coordinate along u: look for the query where it
            // sits in that frame, c' = R^T (c - t) (T rigid: distances are the same in both frames up to the
            // rounding of R, covered by `shrink`); a non-rigid T makes the window the whole cloud
            const float *M = p.poseA + (size_t)b * 16;
            float dev = 0.f;
#pragma unroll
            for (int a0 = 0; a0 < 3; ++a0)
#pragma unroll
                for (int a1 = 0; a1 < 3; ++a1) {
                    const float g = M[0 * 4 + a0] * M[0 * 4 + a1] + M[1 * 4 + a0] * M[1 * 4 + a1] + M[2 * 4 + a0] * M[2 * 4 + a1];
                    dev = fmaxf(dev, fabsf(g - (a0 == a1 ? 1.f : 0.f)));
                }
            if (!(dev <= 1e-4f)) r = 3e37f;
            shrink = 0.9995f;
            {   // u . R^T (c - t): the raw frame's coordinates of the query, then the key
                const float dx = qx - M[3], dy = qy - M[7], dz = qz - M[11];
                const float rx = M[0] * dx + M[4] * dy + M[8] * dz, ry = M[1] * dx + M[5] * dy + M[9] * dz, rz = M[2] * dx + M[6] * dy + M[10] * dz;
                cu = axis >= 3 ? sort_key_dir(dirX, dirY, rx, ry) : (axis == 0 ? rx : (axis == 1 ? ry : rz));
            }
        }
    }
    const float lo = wave_min_uniform(live ? cu : kInf), hi = wave_max_uniform(live ? cu : -kInf);
    float best = kInf;
    const float *tkx = ts, *tky = ts + p.NP16, *tkz = ts + 2 * p.NP16;
    const int np16 = (nt + kChunk - 1) / kChunk * kChunk;
    // window searches: keys staged in LDS while that is cheap (<= 16 KiB per workgroup), read from
    // global memory (L2) on long clouds, where staging the whole key array would cost more than the
    // two or three dependent probes of a search
    const float *gkey = ks + (size_t)min(axis, 2) * p.NP16;   // (a coordinate key's row; a direction key is formed from the x and y rows)
    const bool stage = np16 <= kSweepStage;
    if (stage) {
        // (padded: sorted_refine<.., PAD>; the +inf rows behind the cloud keep +inf)
        for (int j = threadIdx.x; j < np16; j += kSweepBlock)
            keyLds[j + (j >> 5)] = (axis >= 3 && j < nt) ? sort_key_dir(dirX, dirY, ks[j], ks[p.NP16 + j]) : (axis >= 3 ? kInf : gkey[j]);
        __syncthreads();
    }
#ifdef ICPFLOW_SWEEP_CLOCK
    dbgC1 = clock64();
#endif
    if (fullScan) {
        __shared__ int lastSh;
        const int nchunks = np16 / kChunk, per = (nchunks + shares - 1) / shares * kChunk;
        const int c0 = min(si * per, np16), c1 = min(c0 + per, np16);
        if (lo <= hi) {   // (a wave with queries)
            if (MODE == SWEEP_SCORE && backward) scan_range_min_uniform<true>(tkx, tky, tkz, c0, c1, qx, qy, qz, tx, ty, tz, best);
            else scan_range_min_uniform<false>(tkx, tky, tkz, c0, c1, qx, qy, qz, 0.f, 0.f, 0.f, best);
        }
        const size_t slot0 = ((size_t)job * kSweepFullQb + qi) * kSweepShares;   // this query block's shares
        float *mine = p.shareBest + (slot0 + si) * kSweepBlock;
        __hip_atomic_store(&mine[threadIdx.x], best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // (write-through)
        __syncthreads();
        if (threadIdx.x == 0) {
            // (release / acquire at agent scope around the delivery counter -- the barrier in front carries the other
            // threads' stores into the release, the one behind carries the acquire to their loads: the hand-over between blocks on
            // different XCDs rests on the memory model, not on the write-through behaviour of the stores.  One thread per block.)
            int *counter = p.shareCount + (size_t)job * kSweepFullQb + qi;
            const int before = __hip_atomic_fetch_add(counter, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            lastSh = before == shares - 1 ? 1 : 0;
            if (lastSh) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (for the next launch)
        }
        __syncthreads();
        if (!lastSh) return;
        const float *all = p.shareBest + slot0 * kSweepBlock;
        for (int sh = 0; sh < shares; ++sh)
            best = fminf(best, __hip_atomic_load(&all[(size_t)sh * kSweepBlock + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    } else
    if (lo <= hi && nt > 0) {   // wave-uniform

        // slack: rounding of src + t / of the inverse map (ulps of the coordinates) and of the window arithmetic
        // (a direction key is computed: 2^-23 of |x| + |y| on either side, sortdir.hpp -- the wave's largest, and the translation's)
        const float dirSlack = axis >= 3 ? 4.8e-7f * (wave_max_uniform(live ? fabsf(qx) + fabsf(qy) : 0.f) + fabsf(tx) + fabsf(ty) + 4.0f * r) : 0.f;
        const float slack = (MODE == SWEEP_EVAL ? 2e-3f : 1e-4f) + (MODE == SWEEP_EVAL ? 2e-5f : 2e-6f) * (fabsf(lo) + fabsf(hi) + fabsf(tu)) + dirSlack;
        // Grow the window until it provably holds every lane's nearest neighbour: scan the targets within
        // r of the wave's queries along u (only the parts not scanned yet); if every lane's minimum is
        // within r, done.  Otherwise the largest minimum R bounds every NN distance -- one more round with
        // r = R settles it -- unless some lane has not seen any target yet (r quadruples: a query beyond the
        // end of the other cloud must not cost a scan of the whole cloud).
        int cb = 0, ce = 0;          // chunk range scanned so far
        float lbAdded = 0.f;         // pruned scoring: what this wave has added to the scan's running lower bound
        for (int round = 0; round < 24; ++round) {
            int j0, j1;
            // (staged keys are padded against the bank conflicts of the strided first level: at 2048 keys its 64 samples are 32 words
            // apart -- r05 counters: 51-61 % of these kernels' LDS cycles were conflicts; 1 % of their wave cycles, profiles/README)
            if (stage) sorted_window<true>(keyLds, nt, lo - r - slack, hi + r + slack, lane, j0, j1);
            else if (axis < 3) sorted_window<false>(gkey, nt, lo - r - slack, hi + r + slack, lane, j0, j1);
            else sorted_window_fn([&](int j) { return sort_key_dir(dirX, dirY, ks[j], ks[p.NP16 + j]); }, nt, lo - r - slack, hi + r + slack, lane, j0, j1);
            const int k0 = (j0 / kChunk) * kChunk, k1 = min((j1 + kChunk - 1) / kChunk * kChunk, np16);
            if (ce <= cb) { cb = k0; ce = k0; }   // nothing scanned yet
            // the two ranges not scanned yet; in a shared window this wave's quarter of each (whole chunks)
            int a0 = k0, a1 = min(cb, k1), b0 = max(ce, k0), b1 = k1;
            if (sharedWindow) {
                constexpr int W4 = kSweepBlock / kWave;
                const int la = max(a1 - a0, 0), lb = max(b1 - b0, 0);
                const int pa = (la / kChunk + W4 - 1) / W4 * kChunk, pb = (lb / kChunk + W4 - 1) / W4 * kChunk;
                const int ea = a1, eb = b1;
                a0 = a0 + wave * pa; a1 = min(a0 + pa, ea);
                b0 = b0 + wave * pb; b1 = min(b0 + pb, eb);
            }
            if (MODE == SWEEP_SCORE && backward) {
                scan_range_min_uniform<true>(tkx, tky, tkz, a0, a1, qx, qy, qz, tx, ty, tz, best);
                scan_range_min_uniform<true>(tkx, tky, tkz, b0, b1, qx, qy, qz, tx, ty, tz, best);
            } else {
                scan_range_min_uniform<false>(tkx, tky, tkz, a0, a1, qx, qy, qz, 0.f, 0.f, 0.f, best);
                scan_range_min_uniform<false>(tkx, tky, tkz, b0, b1, qx, qy, qz, 0.f, 0.f, 0.f, best);
            }
            if (sharedWindow) {   // every wave ends the round with the minimum over all four quarters
                shareBest[wave][lane] = best;
                __syncthreads();
#pragma unroll
                for (int w = 0; w < kSweepBlock / kWave; ++w) best = fminf(best, shareBest[w][lane]);
                __syncthreads();
            }
#ifdef ICPFLOW_SWEEP_CLOCK
            ++dbgRounds; dbgChunks = ce - cb;
#endif
            cb = min(cb, k0); ce = max(ce, k1);
            const float worst = wave_max_uniform(live ? best : 0.f);
            const float proven = r * shrink;
            if (MODE == SWEEP_SCORE && p.prune == 1) {
                // Every target not scanned yet is farther than `proven` from every query of the wave, so
                // min(sqrt(best), proven) bounds each lane's distance from below whether its neighbour has been
                // found or not.  The wave adds what its bound has GROWN by to the scan's running sum and leaves as
                // soon as that sum -- a lower bound of the scan's total at any time -- rules the candidate out:
                // a candidate half a metre off is gone after the first round of its waves, no neighbour found.
                const float sb = sqrtf(best), pr = proven * 0.99999f;
                const float lbLane = !live ? 0.f : (sb < pr ? sb : (sb != sb ? sb : pr));   // NaN stays NaN: never prunes
                const float L = wave_sum(lbLane) * 0.99999f;
                if (lane == 0 && L > lbAdded && (!sharedWindow || wave == 0)) atomicAdd(p.accum + job, (double)(L - lbAdded));
                if (L > lbAdded) lbAdded = L;
                const bool done = worst <= proven * proven || (cb == 0 && ce == np16);
                if (!done) {   // before another (wider) round: is the scan still in the race?
                    double seen = 0.0;
                    if (lane == 0) seen = __hip_atomic_load(p.accum + job, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const float low = __int_as_float(__builtin_amdgcn_readfirstlane(
                        __float_as_int((float)(seen / (double)(float)(backward ? nc : na)))));
                    bool leaveNow = low > boundSh * 1.0001f;
                    if (sharedWindow) {   // (the waves read the running sum at different times: wave 0's reading decides for all)
                        if (threadIdx.x == 0) shareLeave = leaveNow ? 1 : 0;
                        __syncthreads();
                        leaveNow = shareLeave != 0;
                        __syncthreads();
                    }
                    if (leaveNow) {
                        if (lane == 0) prunedSh = 1;
                        break;
                    }
                }
            }
            if (worst <= proven * proven || (cb == 0 && ce == np16)) break;   // proven, or everything scanned
            const float rPrev = r;
            r = (worst < kInf) ? sqrtf(worst) * (MODE == SWEEP_EVAL ? 1.001f : 1.000002f) : r * 4.0f;
            // a scan that may still be pruned widens its window by doublings: the lower bound grows with the proven
            // radius, and a wave whose worst lane is a metre from everything must not pay for a metre of targets
            // before the scan's running sum has had the chance to end it (config 2: -5 % per step)
#ifndef ICPFLOW_SWEEP_GROWTH
#define ICPFLOW_SWEEP_GROWTH 2.0f
#endif
            if (MODE == SWEEP_SCORE && p.prune == 1) r = fminf(r, rPrev * ICPFLOW_SWEEP_GROWTH);
        }
    }
#ifdef ICPFLOW_SWEEP_CLOCK
    dbgC2 = clock64();
#ifndef ICPFLOW_SWEEP_CLOCK_MODE
#define ICPFLOW_SWEEP_CLOCK_MODE 1   // SWEEP_CHECK; 0: the scoring sweeps (the LAST launch of a call overwrites: slot = b * 12 + scan)
#endif
    // (the block of the job that ENDS last; racy between blocks, good enough for a debug build)
    if (MODE == ICPFLOW_SWEEP_CLOCK_MODE && threadIdx.x == 0) {
        const int slot = (MODE == SWEEP_SCORE ? job : b * 2 + sub) & 4095;
        long long *o = g_sweep_clk + slot * 8;
        const long long now = wall_clock64();
        if (now > o[1]) { o[0] = dbgW0; o[1] = now; o[2] = dbgC1 - dbgC0; o[3] = dbgC2 - dbgC1; o[4] = dbgRounds; o[5] = nt; o[6] = nq; o[7] = qb; }
    }
#endif
    // masked sums over this block's queries: sum of Euclidean NN distances (utils_helper.py:30,
    // utils_hist.py:89-95); match_eval adds inlier counts and, forward, the centroids (utils_match.py:168-181)
    double v[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) v[k] = 0.0;
    if (live && nt > 0 && (!sharedWindow || wave == 0)) {
        const float d = sqrtf(best);
        v[0] = (double)d;
        if (MODE == SWEEP_EVAL) {
            v[1] = (d < p.thres) ? 1.0 : 0.0;  // utils_match.py:168 (strict <)
            if (!backward) { v[2] = qx; v[3] = qy; v[4] = qz; v[5] = ox; v[6] = oy; v[7] = oz; }
        }
    }
    constexpr int NV = (MODE == SWEEP_EVAL) ? kPartial : 1;
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0)
        for (int k = 0; k < NV; ++k) red[wave * kPartial + k] = v[k];
    __syncthreads();
    if (threadIdx.x < kPartial) {
        double ssum = 0.0;
        if (threadIdx.x < NV)
            for (int w = 0; w < kSweepBlock / kWave; ++w) ssum += red[w * kPartial + threadIdx.x];
        if (MODE == SWEEP_SCORE && prunedSh != 0 && threadIdx.x == 0) ssum = __builtin_huge_val();
        out[threadIdx.x] = ssum;
    }
}

template <int MODE, bool SHARE = false>
__global__ __launch_bounds__(kSweepBlock) void sweep_scan_kernel(SweepParams p)
{
    sweep_job<MODE, SHARE>(p, (int)blockIdx.x);
}

// The pruned scoring launch behind the occupancy pre-bound: 98 % of its (scan, query block) jobs ended at their first load, and a
// workgroup launched only to leave still costs the dispatcher ~2 ns -- 82 000 of them on config 4's shard: the whole 168 us of that
// launch.  (Drawing the jobs by tickets was worse: a global atomic per job, 4.6 ms per shard step.)  So the launch comes in two: a
// DECIDING launch of one block per scan (listMode 1: prologue, pre-bound, +inf records or an entry in the list) and this one, whose
// blocks walk the (query block, listed scan) jobs of the scans that go on -- a count the host never learns: the grid is a few
// workgroups per CU, striding.
template <int MODE, bool SHARE = false>
__global__ __launch_bounds__(kSweepBlock) void sweep_list_kernel(SweepParams p)
{
    const int entries = *p.listCount * p.qblocks;
    for (int e = (int)blockIdx.x; e < entries; e += (int)gridDim.x) {
        sweep_job<MODE, SHARE>(p, e);
        __syncthreads();   // (the job's LDS -- records, flags, staged keys -- is this block's again)
    }
}

// pcd1 * T in pcd1's sorted order, as transform_points_batch forms it (utils_match.py:162), +inf padded
__global__ void transform_soa_kernel(const float *__restrict__ soa, const int32_t *__restrict__ len,
                                     const float *__restrict__ pose, int NP16, float *__restrict__ out,
                                     const float *__restrict__ soaSwapped, const uint8_t *__restrict__ swap)
{
    const int b = blockIdx.y, k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= NP16) return;
    const float *in = ((swap != nullptr && swap[b] != 0) ? soaSwapped : soa) + (size_t)b * 3 * NP16;
    float *o = out + (size_t)b * 3 * NP16;
    float x = kInf, y = kInf, z = kInf;
    if (k < len[b]) {
        const Affine a = affine_from_pose(pose + (size_t)b * 16);
        affine_apply(a, in[k], in[NP16 + k], in[2 * NP16 + k], x, y, z);
    }
    o[k] = x; o[NP16 + k] = y; o[2 * NP16 + k] = z;
}

#ifdef ICPFLOW_SWEEP_CLOCK
extern "C" int icpflow_debug_sweep_clk(long long *out32768)
{
    return (int)hipMemcpyFromSymbol(out32768, HIP_SYMBOL(g_sweep_clk), sizeof(long long) * 32768);
}
#endif
int sweep_qblocks(int maxRows) { return (maxRows + kSweepBlock - 1) / kSweepBlock; }

// Per pair in DISPATCH order (pairOrder, or as they come) and per direction -- [0] / [1]: the roles of scoring and check (queries =
// src role / dst role), [2] / [3]: match_eval's (pcd1 / pcd2) --: pair << 16 | workgroups that have work when small-against-long
// jobs are scanned by several workgroups << 8 | query blocks with rows.  The conditions are sweep_scan_kernel's own.
__global__ void sweep_pair_table_kernel(const int32_t *__restrict__ lenA, const int32_t *__restrict__ lenC,
                                        const uint8_t *__restrict__ swap, const int32_t *__restrict__ pairOrder, int B,
                                        int qblocks, int32_t *__restrict__ tab)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int b = pairOrder != nullptr ? pairOrder[k] : k;
    const bool sw = swap != nullptr && swap[b] != 0;
    const int na = (sw ? lenC : lenA)[b], nc = (sw ? lenA : lenC)[b];
    const int nqs[4] = {na, nc, lenA[b], lenC[b]}, nts[4] = {nc, na, lenC[b], lenA[b]};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int nq = nqs[d], nt = nts[d];
        const int nqb = (nq + kSweepBlock - 1) / kSweepBlock;
        const bool fullScan = nq > 0 && nqb <= kSweepFullQb && nt >= kSweepFullScanMinNt && qblocks >= 2 * nqb;
        const int shares = fullScan ? min(qblocks / nqb, kSweepShares) : 1;
        tab[k * 4 + d] = (b << 16) | (min(shares * nqb, 255) << 8) | min(nqb, 255);
    }
}

hipError_t launch_sweep_pair_table(const int32_t *lenA, const int32_t *lenC, const uint8_t *swap, const int32_t *pairOrder,
                                   int B, int N, int32_t *tab, hipStream_t s)
{
    if (B > 32767 || sweep_qblocks(N) > 255) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sweep_pair_table_kernel, dim3((B + 255) / 256), dim3(256), 0, s, lenA, lenC, swap, pairOrder, B,
                       sweep_qblocks(N), tab);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_sweep(SweepParams p, hipStream_t s)
{
    p.NP16 = (p.N + kChunk - 1) / kChunk * kChunk;
    p.qblocks = sweep_qblocks(p.N);
    p.r0 = 0.15f;
#ifndef ICPFLOW_NO_SWEEP_SHARE
    p.shareWindows = 1;
#endif
#ifdef ICPFLOW_NO_SWEEP_SHARE
    p.shareCount = nullptr;
#endif
    if (p.N < kSweepFullScanMinTargets || p.shareBest == nullptr) p.shareCount = nullptr;
    if (p.shareCount != nullptr && !p.shareClean) {
        // (the jobs of this launch index the counters by their place in the partial records: b * 12 + scan, or b * 2 + direction)
        const hipError_t me = hipMemsetAsync(p.shareCount, 0, (size_t)(MODE == SWEEP_SCORE ? (p.njobs / p.subCount) * 12 : p.njobs) * kSweepFullQb * sizeof(int), s);
        if (me != hipSuccess) return me;
    }
    // (the check sweep beside the scoring's totals: two halves of whole groups of eight pairs, see the kernel)
    const int groups = (MODE == SWEEP_CHECK && p.initSum != nullptr) ? 2 * ((p.njobs / 2 + 7) / 8) : (p.njobs + 7) / 8;
    const size_t lds = (size_t)((p.NP16 < kSweepStage ? p.NP16 : kSweepStage) * 33 / 32 + 1) * sizeof(float);   // (one word of padding per 32 keys)
    // (njobs <= 12 * 700: the batches whose workspace holds the shared minima; larger ones fill the GPU with whole pairs)
    const int total = groups * 8 * p.qblocks;
    if (MODE == SWEEP_SCORE && p.listMode == 1) {   // the deciding launch: query block 0 of every scan
        const int first = groups * 8;
        if (p.shareWindows != 0 && p.N >= kSweepFullScanMinTargets && p.njobs <= 12 * 700)
            hipLaunchKernelGGL((sweep_scan_kernel<MODE, true>), dim3((unsigned)first), dim3(kSweepBlock), lds, s, p);
        else
            hipLaunchKernelGGL((sweep_scan_kernel<MODE, false>), dim3((unsigned)first), dim3(kSweepBlock), lds, s, p);
        return hipGetLastError();
    }
    if (MODE == SWEEP_SCORE && p.listMode == 2) {   // the listed scans: eight workgroups per CU at most, striding over what the list holds
        const int wgs = total < 8 * 256 ? total : 8 * 256;
        if (p.shareWindows != 0 && p.N >= kSweepFullScanMinTargets && p.njobs <= 12 * 700)
            hipLaunchKernelGGL((sweep_list_kernel<MODE, true>), dim3((unsigned)wgs), dim3(kSweepBlock), lds, s, p);
        else
            hipLaunchKernelGGL((sweep_list_kernel<MODE, false>), dim3((unsigned)wgs), dim3(kSweepBlock), lds, s, p);
        return hipGetLastError();
    }
    if (p.shareWindows != 0 && p.N >= kSweepFullScanMinTargets && p.njobs <= 12 * 700)
        hipLaunchKernelGGL((sweep_scan_kernel<MODE, true>), dim3((unsigned)total), dim3(kSweepBlock), lds, s, p);
    else
        hipLaunchKernelGGL((sweep_scan_kernel<MODE, false>), dim3((unsigned)total), dim3(kSweepBlock), lds, s, p);
    return hipGetLastError();
}

// Dilated occupancy grid of one sorted cloud (SoA image of launch_sort_clouds_soa: valid rows first, +inf behind them), grid (B, 2):
// y = 0 the src role's cloud, 1 the dst role's.  Cell edge h = 0.125 m, grown by a quarter at a time until the box of the cloud --
// plus one cell of margin all round -- fits kOccWords * 32 cells.  A point sets the bits of its cell and of the 26 around it.
constexpr int kOccBlock = 256;
// The whole bit array moved by `sh` bits towards higher (up) or lower cell numbers; bits moved in from outside are zero.
__device__ __forceinline__ uint32_t occ_shifted(const uint32_t *in, int w, int sh, bool up)
{
    const int q = sh >> 5, r = sh & 31;
    if (up) {
        const int a = w - q;
        const uint32_t lo = a >= 0 ? in[a] : 0u, lo1 = a - 1 >= 0 ? in[a - 1] : 0u;
        return r == 0 ? lo : ((lo << r) | (lo1 >> (32 - r)));
    }
    const int a = w + q;
    const uint32_t hi = a < kOccWords ? in[a] : 0u, hi1 = a + 1 < kOccWords ? in[a + 1] : 0u;
    return r == 0 ? hi : ((hi >> r) | (hi1 << (32 - r)));
}

__global__ __launch_bounds__(kOccBlock) void occ_build_kernel(const float *__restrict__ Xsoa, const float *__restrict__ Ysoa,
                                                            int NP16, float *__restrict__ hdrOut, uint32_t *__restrict__ bitsOut)
{
    __shared__ uint32_t bitsA[kOccWords], bitsB[kOccWords];
    __shared__ float red[6 * (kOccBlock / kWave)];
    __shared__ float hd[8];
    const int b = blockIdx.x, which = blockIdx.y, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    const float *px = (which == 0 ? Xsoa : Ysoa) + (size_t)b * 3 * NP16, *py = px + NP16, *pz = py + NP16;
    for (int k = tid; k < kOccWords; k += kOccBlock) bitsA[k] = 0u;
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    for (int i = tid; i < NP16; i += kOccBlock) {
        const float x = px[i], y = py[i], z = pz[i];
        if (fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f) {   // (pads are +inf; NaN rows fail too: they are no target anybody is near)
            mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
            mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            mn[a] = fminf(mn[a], __shfl_xor(mn[a], o, kWave));
            mx[a] = fmaxf(mx[a], __shfl_xor(mx[a], o, kWave));
        }
        if (lane == 0) { red[wave * 6 + a] = mn[a]; red[wave * 6 + 3 + a] = mx[a]; }
    }
    __syncthreads();
    if (tid == 0) {
        float lo[3], hi[3];
        for (int a = 0; a < 3; ++a) {
            lo[a] = kInf; hi[a] = -kInf;
            for (int w = 0; w < kOccBlock / kWave; ++w) { lo[a] = fminf(lo[a], red[w * 6 + a]); hi[a] = fmaxf(hi[a], red[w * 6 + 3 + a]); }
        }
        int n[3] = {0, 0, 0};
        float h = 0.125f;
        if (lo[0] <= hi[0]) {   // a cloud with points
            for (int grow = 0; grow < 64; ++grow) {
                bool fits = true;
                long long cells = 1;
                for (int a = 0; a < 3; ++a) {
                    const float e = (hi[a] - lo[a]) / h;
                    if (!(e < 30000.f)) { fits = false; break; }
                    n[a] = (int)e + 2 + 2 * kOccRings;   // kOccRings cells of margin below the lowest point, the points' cells, as many above, one of slack
                    cells *= n[a];
                }
                if (fits && cells <= (long long)kOccWords * 32) break;
                h *= 1.25f; n[0] = 0;
            }
        }
        if (n[0] <= 0) n[0] = n[1] = n[2] = 0;   // empty cloud (or one no grid holds): no bound
        hd[0] = lo[0] - kOccRings * h; hd[1] = lo[1] - kOccRings * h; hd[2] = lo[2] - kOccRings * h;   // the lowest point falls into cell kOccRings
        // rounding of the cell arithmetic: two ulps of the largest coordinate, in cells -- below 0.5 % of a cell or no grid at all
        // (coordinates of kilometres against a 12.5 cm cell); the bound itself keeps 2 % in hand (0.98 h)
        float big = 0.f;
        for (int a = 0; a < 3; ++a) big = fmaxf(big, fmaxf(fabsf(lo[a]), fabsf(hi[a])));
        if (!(big * 2.4e-7f / h < 0.005f)) n[0] = n[1] = n[2] = 0;
        hd[3] = 1.0f / h; hd[4] = 0.98f * h;
        hd[5] = __int_as_float(n[0]); hd[6] = __int_as_float(n[1]); hd[7] = __int_as_float(n[2]);
    }
    __syncthreads();
    const float gox = hd[0], goy = hd[1], goz = hd[2], ginv = hd[3];
    const int gnx = __float_as_int(hd[5]), gny = __float_as_int(hd[6]), gnz = __float_as_int(hd[7]);
    if (gnx > 0) {
        for (int i = tid; i < NP16; i += kOccBlock) {
            const float x = px[i], y = py[i], z = pz[i];
            if (!(fabsf(x) < 1e30f && fabsf(y) < 1e30f && fabsf(z) < 1e30f)) continue;
            // the cell, kept kOccRings cells inside the grid (rounding at the box's faces): every dilation stays inside, and a
            // shift of the bit array by one cell along any axis never carries a bit across a face
            const int cx = min(max((int)floorf((x - gox) * ginv), kOccRings), gnx - 1 - kOccRings);
            const int cy = min(max((int)floorf((y - goy) * ginv), kOccRings), gny - 1 - kOccRings);
            const int cz = min(max((int)floorf((z - goz) * ginv), kOccRings), gnz - 1 - kOccRings);
            const int c = (cx * gny + cy) * gnz + cz;
            atomicOr(&bitsA[c >> 5], 1u << (c & 31));
        }
    }
    __syncthreads();
    // dilation by one cell along z, y, x in turn (separable: the 27-neighbourhood), each a shift of the whole bit array by 1, nz,
    // ny * nz bits up and down
    // Plane k (k = 0 .. kOccRings - 1) is the cloud dilated k + 1 times: a cell that is empty there has no point within k + 1 cells.
    const int strides[3] = {1, gnz, gny * gnz};
    uint32_t *in = bitsA, *outb = bitsB;
    uint32_t *bo = bitsOut + ((size_t)b * 2 + which) * kOccRings * kOccWords;
    for (int ring = 0; ring < kOccRings; ++ring) {
        for (int a = 0; a < 3; ++a) {
            if (gnx > 0)
                for (int w = tid; w < kOccWords; w += kOccBlock) outb[w] = in[w] | occ_shifted(in, w, strides[a], true) | occ_shifted(in, w, strides[a], false);
            __syncthreads();
            uint32_t *t = in; in = outb; outb = t;
        }
        for (int k = tid; k < kOccWords; k += kOccBlock) bo[ring * kOccWords + k] = gnx > 0 ? in[k] : 0u;
    }
    if (tid < 8) hdrOut[((size_t)b * 2 + which) * 8 + tid] = hd[tid];
}

hipError_t launch_occupancy(const GridScratch *grid, int B, int N, hipStream_t s)
{
    if (grid->occHdr == nullptr || grid->occBits == nullptr || grid->sortXsoa == nullptr || grid->sortYsoa == nullptr) return hipSuccess;
    const int NP16 = (N + kChunk - 1) / kChunk * kChunk;
    hipLaunchKernelGGL(occ_build_kernel, dim3(B, 2), dim3(kOccBlock), 0, s, grid->sortXsoa, grid->sortYsoa, NP16, grid->occHdr, grid->occBits);
    return hipGetLastError();
}

hipError_t launch_sweep_score(const GridScratch *grid, const int32_t *lenA, const int32_t *lenC, const uint8_t *swap,
                              int B, int N, const float *cand, double *partial, hipStream_t s)
{
    SweepParams p{};
    p.Asoa = grid->sortXsoa; p.Csoa = grid->sortYsoa; p.lenA = lenA; p.lenC = lenC; p.swap = swap;
    p.axis = grid->axis; p.cand = cand; p.N = N; p.njobs = B * 12; p.partial = partial;
    p.subBegin = 0; p.subCount = 12;
    p.shareBest = grid->shareBest; p.shareCount = grid->shareCount; p.shareClean = grid->shareCountClean; p.pairOrder = grid->pairOrder; p.pairTab = grid->pairTab;
    return launch_sweep<SWEEP_SCORE>(p, s);
}

// The same sweeps with the branch and bound of launch_scan_score_pruned, in two launches: candidate 0 forward, then the
// other eleven scans pruned against it; accum: [B,12] zeros.  A pruned scan reports +inf, the pick is unchanged.
hipError_t launch_sweep_score_pruned(const GridScratch *grid, const int32_t *lenA, const int32_t *lenC,
                                     const uint8_t *swap, int B, int N, const float *cand, double *partial,
                                     double *accum, hipStream_t s)
{
    SweepParams p{};
    p.Asoa = grid->sortXsoa; p.Csoa = grid->sortYsoa; p.lenA = lenA; p.lenC = lenC; p.swap = swap;
    p.axis = grid->axis; p.cand = cand; p.N = N; p.partial = partial; p.accum = accum;
    if (grid->occReady) { p.occHdr = grid->occHdr; p.occBits = grid->occBits; }
    p.shareBest = grid->shareBest; p.shareCount = grid->shareCount; p.shareClean = grid->shareCountClean; p.pairOrder = grid->pairOrder; p.pairTab = grid->pairTab;
    if (grid->occReady) {
        // With the occupancy pre-bound (sweep_scan_kernel) nearly every scan of the other five candidates ends before it has evaluated a
        // target: what is left of the second launch is candidate 0's backward scan, which nothing ends before its last block.  It
        // runs to the end beside the forward scan instead (the first launch filled half of the GPU at config 2), and the others are
        // pruned against candidate 0's SCORE, min(forward, backward), instead of its forward mean alone.
        p.njobs = B * 2; p.subBegin = 0; p.subCount = 2; p.prune = 0;
        hipError_t e2 = launch_sweep<SWEEP_SCORE>(p, s);
        if (e2 != hipSuccess) return e2;
        p.njobs = B * 10; p.subBegin = 2; p.subCount = 10; p.prune = 1; p.boundBoth = 1;
        // In two launches (sweep_list_kernel) where the dead jobs' dispatch is what the launch lasts: batches several times the GPU
        // (config 4's shard: 81 920 jobs, 168 -> 66 + 43 us).  Smaller batches keep the one launch -- there the scans that go on
        // start at once instead of behind the last deciding block (ragged 600 x 1024: 0.818 against 0.825 ms per step).
#ifndef ICPFLOW_SCORE_LIST_MIN_JOBS
#define ICPFLOW_SCORE_LIST_MIN_JOBS 65536
#endif
        if (grid->sweepTicket != nullptr && grid->scoreList != nullptr &&   // (the call has cleared the counter)
            (long long)B * 10 * sweep_qblocks(N) >= ICPFLOW_SCORE_LIST_MIN_JOBS) {
            p.list = grid->scoreList; p.listCount = grid->sweepTicket;
            p.listMode = 1;
            const hipError_t e3 = launch_sweep<SWEEP_SCORE>(p, s);
            if (e3 != hipSuccess) return e3;
            p.listMode = 2;
        }
        return launch_sweep<SWEEP_SCORE>(p, s);
    }
    p.njobs = B; p.subBegin = 0; p.subCount = 1; p.prune = 0;
    hipError_t e = launch_sweep<SWEEP_SCORE>(p, s);
    if (e != hipSuccess) return e;
    // Second launch: the other ten scans AND candidate 0's backward scan, all under the bound of candidate 0's forward mean.
    // (Round 3; until then candidate 0's backward scan was a third launch, skipped where every other candidate was out.
    // Its score is min(forward, backward): a backward scan pruned for exceeding the forward mean leaves the score where
    // it was, so it can run beside the others under the same rule -- one launch less in front of every ICP: config 2's
    // step 0.711 -> 0.702 ms, config 4's shard 4.18 -> 4.10 ms, picks identical, tools/dbg/score_fuzz.py.)
    p.njobs = B * 11; p.subBegin = 1; p.subCount = 11; p.prune = 1;
    return launch_sweep<SWEEP_SCORE>(p, s);
}

// match_eval (utils_match.py:159-213) on clouds sorted by launch_sort_clouds_soa(pcd1, pcd2): without swap flags pcd1 is
// the moving cloud of that sort; with them (the sort hist_icp made, by role) pcd1 sits in the fixed cloud's arrays for
// the pairs flagged.  srcT: scratch of B * 3 * NP16 floats (+ 64 of slack)
hipError_t launch_sweep_eval(const GridScratch *grid, const int32_t *len1, const int32_t *len2, int B, int N,
                             const float *pose, float thres, float *srcT, double *partial, hipStream_t s,
                             const uint8_t *swap, const uint8_t *active)
{
    SweepParams p{};
    p.active = active;
    p.Asoa = grid->sortXsoa; p.Csoa = grid->sortYsoa; p.lenA = len1; p.lenC = len2; p.axis = grid->axis;
    p.swap = swap;
    p.poseA = pose; p.thres = thres; p.srcT = srcT; p.N = N; p.njobs = B * 2; p.partial = partial;
    p.shareBest = grid->shareBest; p.shareCount = grid->shareCount; p.shareClean = grid->shareCountClean; p.pairOrder = grid->pairOrder; p.pairTab = grid->pairTab;
    const int NP16 = (N + kChunk - 1) / kChunk * kChunk;
    hipLaunchKernelGGL(transform_soa_kernel, dim3((NP16 + 255) / 256, B), dim3(256), 0, s, grid->sortXsoa, len1, pose,
                       NP16, srcT, grid->sortYsoa, swap);
    return launch_sweep<SWEEP_EVAL>(p, s);
}

// roll-back check (utils_icp.py:27-33): mean NN distance of the src role under the init pose and under
// the composed final pose, as sweeps over the sorted clouds the ICP left in `grid`
hipError_t launch_sweep_check(const GridScratch *grid, const float *X, const float *Y, const int32_t *lenA,
                              const int32_t *lenC, const uint8_t *swap, int B, int N, const float *poseInit,
                              const float *poseFinal, double *partial, hipStream_t s, const PoseSource *fused,
                              const uint8_t *active, const double *initSum)
{
    SweepParams p{};
    p.active = active;
    p.initSum = initSum;
    if (poseFinal == nullptr) {
        if (fused == nullptr) return hipErrorInvalidValue;
        p.fused = *fused;
    }
    p.Asoa = grid->sortYsoa;   // unused in this mode (valid pointer for the address arithmetic)
    p.Csoa = grid->sortYsoa; p.lenA = lenA; p.lenC = lenC; p.swap = swap; p.axis = grid->axis;
    p.sortX = (const float4 *)grid->sortX; p.X = X; p.Y = Y; p.poseA = poseInit; p.poseB = poseFinal;
    p.rawSorted = grid->presorted; p.N = N; p.njobs = B * 2; p.partial = partial;
    p.shareBest = grid->shareBest; p.shareCount = grid->shareCount; p.shareClean = grid->shareCountClean; p.pairOrder = grid->pairOrder; p.pairTab = grid->pairTab;
    return launch_sweep<SWEEP_CHECK>(p, s);
}

hipError_t launch_scan_score(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                             const uint8_t *swap, int B, int N, const float *cand, double *partial,
                             hipStream_t s)
{
    ScanParams p{};
    p.A = A; p.C = C; p.lenA = lenA; p.lenC = lenC; p.swap = swap; p.N = N;
    p.njobs = B * 12; p.cand = cand; p.partial = partial;
    p.subBegin = 0; p.subCount = 12;
    return launch_scan<MODE_SCORE>(p, N, B, s);
}

// The same twelve scans per pair with branch and bound: the forward scan of candidate 0 (the highest peak of
// the vote) runs to the end first; the scans of the other five candidates run in query blocks of 128 rows, each block leaving at once
// when the blocks before it have already summed more than candidate 0's score allows (see nn_scan_kernel).
constexpr int kScoreSplit = 2;   // 128-row blocks (64-row blocks measured slower: every block stages the whole target cloud)
int score_qblocks(int maxRows) { return (maxRows + kScanBlock / kScoreSplit - 1) / (kScanBlock / kScoreSplit); }

hipError_t launch_scan_score_pruned(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                                    const uint8_t *swap, int B, int N, const float *cand, double *partial,
                                    double *accum, hipStream_t s, bool accumCleared)
{
    if (!accumCleared) {
        hipError_t e = hipMemsetAsync(accum, 0, (size_t)B * 12 * sizeof(double), s);
        if (e != hipSuccess) return e;
    }
    ScanParams p{};
    p.A = A; p.C = C; p.lenA = lenA; p.lenC = lenC; p.swap = swap; p.N = N;
    p.cand = cand; p.partial = partial; p.accum = accum;
    p.qblocks = score_qblocks(N);
    p.split = kScoreSplit;
    p.njobs = B; p.subBegin = 0; p.subCount = 1; p.prune = 0;
    hipLaunchKernelGGL((nn_scan_kernel<1, MODE_SCORE>), dim3((unsigned)(((p.njobs + 7) / 8) * 8 * p.qblocks)),
                       dim3(kScanBlock), 0, s, p);
    // the other eleven scans under the bound of candidate 0's forward mean -- candidate 0's backward scan among them (its
    // score is min(forward, backward): pruned for exceeding the forward mean, it leaves the score where it was; a third
    // launch of its own until round 3, see launch_sweep_score_pruned)
    p.njobs = B * 11; p.subBegin = 1; p.subCount = 11; p.prune = 1;
    hipLaunchKernelGGL((nn_scan_kernel<1, MODE_SCORE>), dim3((unsigned)(((p.njobs + 7) & ~7) * p.qblocks)),
                       dim3(kScanBlock), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_scan_check(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                             const uint8_t *swap, int B, int N, const float *poseInit,
                             const float *poseFinal, double *partial, hipStream_t s)
{
    ScanParams p{};
    p.A = A; p.C = C; p.lenA = lenA; p.lenC = lenC; p.swap = swap; p.N = N;
    p.njobs = B * 2; p.poseA = poseInit; p.poseB = poseFinal; p.partial = partial;
    return launch_scan<MODE_CHECK>(p, N, B, s);
}

hipError_t launch_scan_eval(const float *A, const float *C, const int32_t *lenA, const int32_t *lenC,
                            int B, int N, const float *pose, float thres, double *partial, hipStream_t s)
{
    ScanParams p{};
    p.A = A; p.C = C; p.lenA = lenA; p.lenC = lenC; p.swap = nullptr; p.N = N;
    p.njobs = B * 2; p.poseA = pose; p.thres = thres; p.partial = partial;
    return launch_scan<MODE_EVAL>(p, N, B, s);
}

hipError_t launch_scan_nn(const float *Qp, const float *Tp, int B, int NQ, int NT, int strideQ,
                          int strideT, const int32_t *lenQ, const int32_t *lenT, int sqrt_dist,
                          int64_t *idx, float *dist, hipStream_t s)
{
    ScanParams p{};
    p.A = Qp; p.C = Tp; p.lenA = lenQ; p.lenC = lenT; p.N = NQ; p.NT = NT;
    p.strideQ = strideQ; p.strideT = strideT; p.njobs = B; p.idx = idx; p.dist = dist;
    p.sqrt_dist = sqrt_dist;
    return launch_scan<MODE_NN>(p, NQ, 1 << 30, s);
}

}  // namespace icpflow

#ifdef ICPFLOW_OCC_STATS
extern "C" int icpflow_debug_occ_pairs(unsigned int *out1024, int reset)
{
    (void)hipDeviceSynchronize();
    const int rc = (int)hipMemcpyFromSymbol(out1024, HIP_SYMBOL(icpflow::g_occPair), sizeof(icpflow::g_occPair));
    if (reset) {
        static unsigned int z[1024];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(icpflow::g_occPair), z, sizeof(z));
    }
    return rc;
}
extern "C" int icpflow_debug_occ_stats(unsigned long long *out12, int reset)
{
    (void)hipDeviceSynchronize();
    const int rc = (int)hipMemcpyFromSymbol(out12, HIP_SYMBOL(icpflow::g_occStats), sizeof(icpflow::g_occStats));
    if (reset) {
        const unsigned long long z[12] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(icpflow::g_occStats), z, sizeof(z));
    }
    return rc;
}
#endif
