// icp.hip -- masked batched point-to-point ICP for gfx950 (a-5, a-6, a-7).
//
// Reference semantics: utils_icp_pytorch3d.py:100-225 (loop), :303-382 (Kabsch via SVD),
// :385-396 (apply).  Design: one workgroup per cluster pair runs a whole ICP iteration --
// NN scan of the moved source against the LDS-staged target (scan.hpp), inlier gate,
// weighted centroids, centred 3x3 covariance, closed-form rotation, rmse -- so the 15-odd
// torch kernels, the cuSOLVER call and the host sync of one reference iteration collapse
// into one launch with ONE block reduction of 18 raw moments.  All sums and the 3x3 solve are
// fp64 (fp32 inputs), i.e. at least as accurate as the reference's fp32 torch reductions.
//
// Stopping: ICPFLOW_STOP_REFERENCE reproduces the batch-global rule (stop when every pair
// has rel <= thr, :209) WITHOUT a host round trip: one launch per iteration is enqueued up
// front; each pair that is not converged bumps ctrl->notconv[it]; the launch of iteration
// it+1 returns immediately when notconv[it] == 0.  ICPFLOW_STOP_PER_PAIR loops inside one
// launch and lets every pair stop on its own.
#include <atomic>
#include <mutex>
#include <vector>

#include "scan.hpp"
#include "sortdir.hpp"
#include "kernels.hpp"

namespace icpflow {

#ifdef ICPFLOW_PHASE_TIMING
// debug builds only (tools/dbg/phase_timing.py): shader-clock stamps of workgroup 0
__device__ long long g_phase_stamps[16];
__device__ long long g_wave_stamps[16 * 16];   // [wave][k] of workgroup 0
__device__ int g_stamp_block;                  // the workgroup that stamps (icpflow_debug_set_stamp_block)
#define ICPFLOW_STAMP(k) do { if ((int)blockIdx.x == g_stamp_block && threadIdx.x == 0) g_phase_stamps[k] = clock64(); \
    if ((int)blockIdx.x == g_stamp_block && (threadIdx.x & 63) == 0) g_wave_stamps[(threadIdx.x >> 6) * 16 + (k)] = clock64(); } while (0)
#else
#define ICPFLOW_STAMP(k) do { } while (0)
#endif

#ifdef ICPFLOW_TAIL_CLOCK
// debug builds only (tools/dbg/tail_clock.py): per pair, shader clocks wave 0 spent between the block barrier and the
// publication of (R, T) (the serial tail), in the rest of the loop, and the iterations it executed; and the tail split
// at the phase stamps (accumulated in LDS by thread 0)
__device__ long long g_tail_clock[1024 * 3];
__device__ long long g_wg_wall[8192 * 4];
__device__ int g_unit_pair = -1;                       // pair whose per-(iteration, pass, wave) clocks are recorded (owner, no helpers)
__device__ long long g_unit_clk[64 * 8 * 16];
__device__ int g_unit_win[64 * 8 * 16 * 2];   // per (iteration, pass, wave): targets in the scanned window, lanes that searched
__device__ unsigned long long g_pair_help[1024];   // passes the pair's owner received from helpers
__device__ unsigned long long g_pair_hclk[1024 * 4];   // per pair: helper pass clocks, helper passes, helper waits (100 MHz), owner waits (100 MHz)
__device__ unsigned long long g_help_stats[8];   // helpers that joined a pair, passes the owners took from helpers, owner clocks spent waiting   // per pair: wall clock (100 MHz) at entry and exit of its workgroup, HW_ID, XCC_ID
__device__ long long g_tail_split[1024 * 16];
__shared__ long long g_tcSh[17];
#ifdef ICPFLOW_TAIL_SPLIT   // (each stamp costs ~200 clocks: the totals above are measured without)
#undef ICPFLOW_STAMP
#define ICPFLOW_STAMP(k) do { if (threadIdx.x == 0) { const long long t_ = clock64(); g_tcSh[k] += t_ - g_tcSh[16]; g_tcSh[16] = t_; } } while (0)
#endif
#endif
#ifdef ICPFLOW_DEBUG_SOLVE
// debug builds only (tools/dbg/onestep_case.py): the 18 moments, H, lambda and R of one pair's FIRST iteration
__device__ double g_dbg_solve[64];
__device__ float g_dbg_xt[4096 * 3];   // the moved points of that iteration by ORIGINAL row
__device__ int g_dbg_pair = 0;
#endif
#ifdef ICPFLOW_CERT_STATS
// debug builds only (tools/dbg/cert_stats.py): per iteration, over the whole batch: waves that ran, waves that searched,
// queries that searched, targets scanned (per wave)
__device__ unsigned long long g_cert_stats[128 * 4];
__device__ unsigned long long g_probe_stats[128 * 2];   // probes, conclusive probes
__device__ unsigned long long g_occ_cert[128 * 4];      // per iteration: queries without a certificate; of those, queries whose cell of the fixed cloud's grid is empty (plane 1: nothing within 0.98 h > gate); queries with certificate (B); outliers among ALL live queries by the grid
__device__ int g_stats_block = -1;                       // >= 0: only this workgroup counts
#endif

}  // namespace icpflow
#include "kabsch.hpp"   // after ICPFLOW_STAMP: the solver carries the phase stamps of debug builds
namespace icpflow {

// ---------------------------------------------------------------------------------
struct IcpParams {
#ifdef ICPFLOW_CERT_STATS
    const float *occHdr;   // (statistics only: the fixed cloud's occupancy grids of nn.hip, or NULL)
    const uint32_t *occBits;
#endif
    const float *X;        // [B,N,4] moving cloud
    const float *Y;        // [B,N,4] fixed cloud
    const int32_t *lenX;
    const int32_t *lenY;
    const uint8_t *swap;   // optional: exchange X and Y roles per pair
    const float *prePose;  // optional [B,4,4]: X0 = transform_points_batch(X, prePose)
    int N;
    float thr2;            // fp32(thres**2), :160
    float relThr;          // fp32(relative_rmse_thr), :209
    int stopMode;
    int maxIter;
    IcpState *state;       // [B]
    IcpCtrl *ctrl;
    // exact uniform-grid search (GRID kernels): built once per registration by grid_build_kernel
    const float4 *gridPts;   // [B,N]   fixed-cloud points sorted by bucket, w = original index bits
    const int32_t *gridStart;// [B,H+1] bucket -> first slot in gridPts
    const float *gridOrigin; // [B,4]   cell origin (the first fixed-cloud point)
    int gridH;               // buckets per pair (power of two)
    float gridInvH;          // 1 / cell edge
    // sorted sweep (SWEEP kernels): both clouds sorted once per registration along the fixed
    // cloud's longest axis by sort_clouds_kernel; w = original index bits
    const float4 *sortX;     // [B,N] moving cloud, pre-pose already applied
    const float4 *sortY;     // [B,N] fixed cloud
    const float *sortYsoa;   // [B,3,NP16] fixed cloud as x[], y[], z[] padded with +inf to 16
    const int32_t *sortAxis; // [B]
    float sweepMargin;       // window half-width beyond the wave's query span (1.01 * thres)
    int sortedRaw;           // sortX holds the moving cloud WITHOUT the pre-pose: apply it at load
    // speculative single-launch execution of the batch-global stop rule (see launch_icp)
    float *history;          // [kHistIters, B, kHistStride] or NULL
    int B;
    IcpTeam team;            // wgPair == NULL: one workgroup per pair (blockIdx.x = pair)
    const float *initR;      // [B,3,3] / [B,3]: the state before the first iteration (init_transform), NULL = identity
    const float *initT;
    int allowReflection;     // R = U V^T whatever its determinant (:354-362 with E = I)
    int estimateScale;       // s = trace(E S) / Xcov (:364-374), Xt = s X R + T
    const float *initS;      // [B] scale of the initial transform, NULL = 1
    int halfCu;              // launch policy: 512-thread workgroups, two per CU (see launch_icp_iters)
    int persistent;          // grid = the workgroups the GPU holds at once; further pairs by ticket (see icp_kernel)
    IcpHelp help;            // helpers of persistent launches (pair == NULL: none)
    int helpOn;
    int redPasses;           // sorted sweep in LDS, one workgroup per pair, N <= kRecMaxN: the moment sums of every (pass, wave)
                             // are kept apart in dynamic LDS (this many passes) and added in (pass, wave) order; 0: one
                             // running sum per lane over all passes (static LDS)
    int x0Cache;             // the records are followed by the queries' own points (12 B each): no L2 round trip per iteration
    int recCap;              // sorted sweep in LDS: room for this many per-query records behind the LDS image (neighbour
                             // certificates, see the search phase); a workgroup whose share of the queries fits uses them
    int shareScans;          // team kernel with the LDS image: the waves of a member share their long window scans (the launch left room for the accumulators)
    int teamLanes;           // host side only: 2 = this team launch takes at most half of the CUs (chained two deep)
    const uint8_t *pairActive;   // options.d_pair_active or NULL: pairs flagged 0 are not in the batch (speculative mode only)
    // two launches (see icp_split_kernel): the FIRST (persistent grid with helpers) is drained once at most `drainAt` pairs are
    // unfinished -- they leave at their next iteration, still moving --; the SECOND serves those pairs, pairList[0 .. pairMeta[0]),
    // one 1024-thread workgroup per CU, each resumed from IcpState at ITS iteration; pairMeta[1]: every iteration below it is
    // known to be complete and not to satisfy the batch rule
    int drainAt;                 // 0: the launch is not drained
    const int32_t *pairList;     // NULL: workgroup w serves pair w
    const int32_t *pairMeta;
    int twoLaunch;               // host side only: two launches wanted where they apply (launch_icp_variant)
    int32_t *splitScratch;       // host side only: [B + 64] ints (the list, then count and floor)
};


// ---------------------------------------------------------------------------------
// Exact nearest neighbour within the gate radius through a hashed uniform grid.
//
// The ICP loop consumes the NN search only through the gate d^2 <= thres^2 and the neighbour of
// gated points (utils_icp_pytorch3d.py:160-164), and the fixed cloud never changes during a
// registration.  So the fixed cloud is binned ONCE into cells of edge h = 1.01 * thres (hashed
// into H = 2^k >= 2N buckets, counting sort); a query then evaluates only the points of the 27
// cells around it.  Every point within the gate radius of the query lies in those cells
// (|coordinate difference| <= thres < h  =>  cell index difference <= 1; the cell index is a
// monotone function of the coordinate), distances are evaluated with the SAME fp32 instruction
// sequence as the brute-force scan and ties go to the lowest original index, so gate decisions
// and neighbours -- hence every transform -- are bit-identical to the all-pairs search, at
// ~30 instead of n distance evaluations per query.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ int grid_cell(float v, float o, float invh)
{
    return (int)floorf((v - o) * invh);
}

__device__ __forceinline__ unsigned grid_hash(int cx, int cy, int cz, unsigned mask)
{
    return ((unsigned)cx * 73856093u ^ (unsigned)cy * 19349663u ^ (unsigned)cz * 83492791u) & mask;
}

constexpr int kGridBlock = 256;

int grid_buckets(int N)
{
    int H = 64;
    while (H < 2 * N) H <<= 1;
    return H;
}

// One workgroup per pair.  counts/starts live in global scratch (L2 resident).
__global__ __launch_bounds__(kGridBlock) void grid_build_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const int32_t *__restrict__ lenX,
    const int32_t *__restrict__ lenY, const uint8_t *__restrict__ swap, int N, int H, float invh,
    float *__restrict__ origin, int32_t *__restrict__ start, int32_t *__restrict__ cursor,
    float4 *__restrict__ pts)
{
    __shared__ int part[kGridBlock];
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool sw = swap != nullptr && swap[b] != 0;
    const float4 *yb = reinterpret_cast<const float4 *>(sw ? X : Y) + (size_t)b * N;
    const int n = (sw ? lenX : lenY)[b];
    int32_t *st = start + (size_t)b * (H + 1);
    int32_t *cu = cursor + (size_t)b * H;
    float4 *out = pts + (size_t)b * N;
    const unsigned mask = (unsigned)H - 1u;
    float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n > 0) o4 = yb[0];
    if (tid == 0) { origin[b * 4 + 0] = o4.x; origin[b * 4 + 1] = o4.y; origin[b * 4 + 2] = o4.z; origin[b * 4 + 3] = 0.f; }
    for (int k = tid; k <= H; k += kGridBlock) st[k] = 0;
    __syncthreads();
    for (int j = tid; j < n; j += kGridBlock) {
        const float4 q = yb[j];
        const unsigned h = grid_hash(grid_cell(q.x, o4.x, invh), grid_cell(q.y, o4.y, invh),
                                     grid_cell(q.z, o4.z, invh), mask);
        atomicAdd(&st[h + 1], 1);
    }
    __syncthreads();
    // exclusive scan of st[1..H] in place: thread t owns a contiguous slice
    const int per = (H + kGridBlock - 1) / kGridBlock;
    const int lo = 1 + tid * per, hi = min(1 + (tid + 1) * per, H + 1);
    int sum = 0;
    for (int k = lo; k < hi; ++k) sum += st[k];
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int k = 0; k < kGridBlock; ++k) { const int v = part[k]; part[k] = run; run += v; }
    }
    __syncthreads();
    int run = part[tid];
    for (int k = lo; k < hi; ++k) { run += st[k]; st[k] = run; }   // st[k] = #points in buckets < k
    __syncthreads();
    for (int k = tid; k < H; k += kGridBlock) cu[k] = st[k];
    __syncthreads();
    for (int j = tid; j < n; j += kGridBlock) {
        const float4 q = yb[j];
        const unsigned h = grid_hash(grid_cell(q.x, o4.x, invh), grid_cell(q.y, o4.y, invh),
                                     grid_cell(q.z, o4.z, invh), mask);
        const int pos = atomicAdd(&cu[h], 1);
        out[pos] = make_float4(q.x, q.y, q.z, __int_as_float(j));
    }
}

// ---------------------------------------------------------------------------------
// Sorted sweep: exact gated nearest neighbour with BROADCAST target reads.
//
// Both clouds are sorted once per registration along the longest axis a of the fixed cloud.
// In every iteration a wave (64 consecutive sorted queries) computes the span [lo, hi] of its
// CURRENT query coordinates along a (exact, from the moved points) and scans only the fixed
// points with coordinate in [lo - m, hi + m], m = 1.01 * thres: a contiguous range of the sorted
// array, read through LDS at one address for the whole wave (the same broadcast scan core as the
// all-pairs search, just over ~1/10 of the targets).  A point outside that window is farther than
// the gate radius from every query of the wave, so gate decisions and gated neighbours are the
// ones of the all-pairs search; equal-distance ties are resolved to the lowest ORIGINAL index.
// ---------------------------------------------------------------------------------
constexpr int kSortBlock = 1024;
static std::atomic<unsigned long long> g_sortAttr{0ull};   // devices on which sort_clouds_kernel has its dynamic-LDS opt-in

// grid (B, 2): blockIdx.y == 0 sorts the fixed cloud, 1 the moving cloud (pre-pose applied)
__global__ __launch_bounds__(kSortBlock) void sort_clouds_kernel(
    const float *__restrict__ X, const float *__restrict__ Y, const int32_t *__restrict__ lenX,
    const int32_t *__restrict__ lenY, const uint8_t *__restrict__ swap, const float *__restrict__ prePose,
    int N, int NP2, int32_t *__restrict__ axisOut, float4 *__restrict__ Xs, float4 *__restrict__ Ys,
    float *__restrict__ Ysoa, float *__restrict__ Xsoa, int selfCount, int dirKeys)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
    unsigned long long *kv = reinterpret_cast<unsigned long long *>(dynLds);   // (sort key, row) pairs, NP2 of them
    __shared__ float bb[6 * (kSortBlock / kWave)];
    __shared__ int axisSh;
    const int b = blockIdx.x, tid = threadIdx.x;
    const bool moving = blockIdx.y == 1;
    int cX, cY;
    bool sw;
    if (selfCount) {
        // hist_icp on the side stream, forked before anything has counted: the lengths (rows with a positive flag) and
        // the smaller-cloud-first flag exactly as count_pair_kernel / zsort_kernel form them; lenX / lenY / swap unread
        __shared__ int cntScratch[2 * (kSortBlock / kWave)];
        const float4 *px = reinterpret_cast<const float4 *>(X) + (size_t)b * N;
        const float4 *py = reinterpret_cast<const float4 *>(Y) + (size_t)b * N;
        int c[2] = {0, 0};
        for (int i = tid; i < N; i += kSortBlock) {
            c[0] += (px[i].w > 0.0f) ? 1 : 0;
            c[1] += (py[i].w > 0.0f) ? 1 : 0;
        }
        block_sum<2, int>(c, cntScratch);
        cX = c[0]; cY = c[1];
        sw = selfCount == 2 && cX > cY;
    } else {
        cX = lenX[b]; cY = lenY[b];
        sw = swap != nullptr && swap[b] != 0;
    }
    const float4 *xb = reinterpret_cast<const float4 *>(sw ? Y : X) + (size_t)b * N;  // moving role
    const float4 *yb = reinterpret_cast<const float4 *>(sw ? X : Y) + (size_t)b * N;  // fixed role
    const int nx = sw ? cY : cX, ny = sw ? cX : cY;
    // axis of largest extent of the fixed cloud (both blocks compute it the same way)
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    for (int j = tid; j < ny; j += kSortBlock) {
        const float4 q = yb[j];
        mn[0] = fminf(mn[0], q.x); mn[1] = fminf(mn[1], q.y); mn[2] = fminf(mn[2], q.z);
        mx[0] = fmaxf(mx[0], q.x); mx[1] = fmaxf(mx[1], q.y); mx[2] = fmaxf(mx[2], q.z);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, kWave));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, kWave));
        }
    if ((tid & (kWave - 1)) == 0) {
#pragma unroll
        for (int k = 0; k < 3; ++k) { bb[(tid >> 6) * 6 + k] = mn[k]; bb[(tid >> 6) * 6 + 3 + k] = mx[k]; }
    }
    __syncthreads();
    __shared__ float boxSh[6];
    if (tid == 0) {
        float e[3];
        for (int k = 0; k < 3; ++k) {
            float lo = bb[k], hi = bb[3 + k];
            for (int w = 1; w < kSortBlock / kWave; ++w) { lo = fminf(lo, bb[w * 6 + k]); hi = fmaxf(hi, bb[w * 6 + 3 + k]); }
            e[k] = hi - lo;
            boxSh[k] = lo; boxSh[3 + k] = hi;
        }
        const int a = (e[0] >= e[1] && e[0] >= e[2]) ? 0 : (e[1] >= e[2] ? 1 : 2);
        axisSh = a;
    }
    __syncthreads();
    // The key that spreads the fixed cloud best (sortdir.hpp): among the three axes and kSortDirs horizontal directions, the one
    // with the smallest sum of squared populations of 0.1 m key bins -- the longest axis, as before, unless another key is at
    // least a tenth better (clouds of a thousand points and more: below that every window is short anyway).  Integer counts,
    // the same on both blocks of the pair.
    if (dirKeys && ny >= kSortDirMinN && nx >= kSortDirMinMoving) {
        __shared__ unsigned int scoreSh[kSortCodes];
        // (the sort has not started: its key array -- NP2 >= 2048 entries of 8 bytes here -- holds the counters)
        (void)choose_sort_code<kSortBlock>(yb, ny, boxSh, axisSh, reinterpret_cast<unsigned int *>(dynLds), scoreSh, &axisSh);
    }
    if (tid == 0 && !moving) axisOut[b] = axisSh;
    const int axis = axisSh;
    float dirX = 0.f, dirY = 0.f;
    if (axis >= 3) sort_dir(axis, dirX, dirY);
    const int n = moving ? nx : ny;
    const float4 *cloud = moving ? xb : yb;
    PointXf pre;
    pre.kind = (moving && prePose) ? XF_AFFINE : XF_NONE;
    pre.a = (moving && prePose) ? affine_from_pose(prePose + (size_t)b * 16) : affine_identity();
    // the sorting network only has to hold THIS cloud: next power of two >= n (ragged batches are
    // padded to the largest cluster, most clusters are far smaller)
    int np2 = kWave;
    while (np2 < n) np2 <<= 1;
    np2 = min(np2, NP2);
    for (int j = tid; j < np2; j += kSortBlock) {
        float k = kInf;
        if (j < n) {
            const float4 q = cloud[j];
            float px, py, pz;
            xf_apply(pre, q.x, q.y, q.z, px, py, pz);
            k = sort_key_of(axis, dirX, dirY, px, py, pz);
        }
        kv[j] = sort_pack(k, j);
    }
    __syncthreads();
    bitonic_sort_lds(kv, np2);
    float4 *out = (moving ? Xs : Ys) + (size_t)b * N;
    const int NP16 = (N + kChunk - 1) / kChunk * kChunk;
    // structure-of-arrays image (x[], y[], z[], padded with +inf to a multiple of 16): always for the
    // fixed cloud, for the moving cloud when the caller wants to sweep in both directions (Xsoa)
    float *soa = moving ? (Xsoa ? Xsoa + (size_t)b * 3 * NP16 : nullptr) : Ysoa + (size_t)b * 3 * NP16;
    for (int r = tid; r < (soa ? NP16 : n); r += kSortBlock) {
        float px = kInf, py = kInf, pz = kInf;
        if (r < n) {
            const int j = sort_index_of(kv[r]);
            const float4 q = cloud[j];
            xf_apply(pre, q.x, q.y, q.z, px, py, pz);
            out[r] = make_float4(px, py, pz, __int_as_float(j));
        }
        if (soa) { soa[r] = px; soa[NP16 + r] = py; soa[2 * NP16 + r] = pz; }
    }
}

// Moments accumulated per iteration (one block reduction, fp64).  With x' = x0 - o and
// y' = y_nn - o for a per-pair origin o (the first source point; keeps |x'|,|y'| at the cluster
// extent so that the raw-moment identities below lose nothing in fp64):
//   0      sum w                     1..3   sum w x'          4..6   sum w y'
//   7..15  sum w x'_i y'_j           16     sum w |x'|^2      17     sum w |y'|^2
// Centred covariance  H W = sum w x'y'^T - W mx' my'^T  (== :318-336 of the reference) and
// sum w |x R + T - y|^2 = (Sxx - W|mx'|^2) + (Syy - W|my'|^2) - 2 W sum_ij R_ij H_ij  (== :191,
// evaluated without rounding X R + T to fp32 first), so one pass over the points suffices.
constexpr int kMoments = 18;

// word k of a state (R row-major, T) as it enters the 32-bit hash of the state: rotated by a per-word amount, all
// twelve xor-ed (full-rate v_alignbit_b32 / v_xor3_b32; a filter only -- a hit is confirmed word by word)
__device__ __forceinline__ int state_hash_word(float w, int k)
{
    const unsigned u = (unsigned)__float_as_int(w);
    return (int)__builtin_amdgcn_alignbit(u, u, (unsigned)((5 * k + 3) & 31));
}
// neighbour certificates: LDS image (12 B / point) + per query the record (q0, L: 16 B) and the neighbour's slot (4 B)
// <= 128 KiB.  Results are identical with and without (ICPFLOW_OPT_NO_ADAPTIVE_WINDOWS).
constexpr int kRecMaxN = 4096;
#ifndef ICPFLOW_PROBE_MAX
#define ICPFLOW_PROBE_MAX 20
#endif
constexpr int kProbeMax = ICPFLOW_PROBE_MAX;   // uncertified queries a wave settles by probes; more take the window scan
#ifndef ICPFLOW_PROBE_STEPS
#define ICPFLOW_PROBE_STEPS 8
#endif
// blocks of 64 targets a probe may evaluate (measured at config 2: 2 / 3 / 4 / 6 / 8 / 12 / 16 blocks -> ICP launch
// 0.487 / 0.512 / 0.441 / 0.401 / 0.396 / 0.396 / 0.400 ms: an inconclusive probe sends its whole wave to the scan)
constexpr int kProbeSteps = ICPFLOW_PROBE_STEPS;
// ... and on clouds of several passes (more than one workgroup-full of queries: 2048 points and up).  A probe proves its
// answer by the distance, ALONG THE SORT AXIS, to the ends of the range it has evaluated; where hundreds of targets share
// one coordinate along that axis -- the face of a box that is perpendicular to it: a third of a 2048-point cloud in the
// slowest pairs of config 4 (profiles/r03_config4_unit_clocks.txt: windows of 650-800 targets) -- that distance stays zero
// until the range has swallowed the whole face: ten to thirteen blocks.  With eight, every probe there ended inconclusive
// and five or ten uncertified lanes sent their wave through a broadcast scan of the whole window, in every iteration
// (50-70 k clocks per 64 queries against 5 k).  Measured (steps, lanes) on config 4's shard, ICP launch: (8, 20) 2.19 ms,
// (12, 20) 1.91, (24, 20) 1.84, (24, 32) 1.84, (24, 48) 1.87, (32, 32) 1.83; all 8192 pairs 11.4 -> 10.2 ms; config 2
// (single pass) does not move with the steps and loses 6 % with 32 lanes.
#ifndef ICPFLOW_PROBE_MAX_LONG
#define ICPFLOW_PROBE_MAX_LONG 32
#endif
#ifndef ICPFLOW_PROBE_STEPS_LONG
#define ICPFLOW_PROBE_STEPS_LONG 24
#endif
constexpr int kProbeMaxLong = ICPFLOW_PROBE_MAX_LONG, kProbeStepsLong = ICPFLOW_PROBE_STEPS_LONG;
#ifndef ICPFLOW_WIDE_WINDOW
#define ICPFLOW_WIDE_WINDOW 512
#endif
constexpr int kWideWindow = ICPFLOW_WIDE_WINDOW;   // (targets in a wave's previous window from which every lane may probe)
#ifndef ICPFLOW_PROBE_CAP
#define ICPFLOW_PROBE_CAP 128
#endif
constexpr int kProbeCap = ICPFLOW_PROBE_CAP;   // entries of a pass's shared probe queue (teams; 28 B of LDS each)
// Shared window scans (teams, round 5): a wave whose window holds at least kShareMinW targets cuts it into up to kShareParts
// parts of at least kSharePartMin targets; the waves that are through with their own unit take parts (see the search phase).
#ifndef ICPFLOW_SHARE_MIN_W
#define ICPFLOW_SHARE_MIN_W 256
#endif
#ifndef ICPFLOW_SHARE_PART_MIN
#define ICPFLOW_SHARE_PART_MIN 128
#endif
#ifndef ICPFLOW_SHARE_MIN_N
#define ICPFLOW_SHARE_MIN_N 3000
#endif
constexpr int kShareMinW = ICPFLOW_SHARE_MIN_W, kSharePartMin = ICPFLOW_SHARE_PART_MIN, kShareMinN = ICPFLOW_SHARE_MIN_N;
constexpr int kRing = 8;   // states remembered for the detection of periodic trajectories (speculative mode)

// ---------------------------------------------------------------------------------
// Teams: several workgroups serve one pair (IcpTeam, kernels.hpp).  Per iteration every member
// publishes its 18 partial moments (+ member 0's stop flag) into the record of (pair, it & 1, rank),
// bumps the pair's arrival counter and waits until the whole team has arrived; two record sets
// suffice because nobody can be more than one exchange ahead of the slowest member.  All accesses
// to the records are device-scope atomics, ordered by the release / acquire pair around the counter.
// The wait is bounded: a member that is not being scheduled (workgroups of the team not co-resident)
// makes its peers give up after kTeamTimeoutTicks; the registration is then poisoned, not hung.
// ---------------------------------------------------------------------------------
constexpr long long kTeamTimeoutTicks = 200000000ll;   // 2 s of the 100 MHz wall clock

__device__ __forceinline__ void team_publish(const IcpTeam &t, int b, int it, int rank, double mine, double stop,
                                             int lane)
{
    double *slot = t.mom + (((size_t)b * 2 + (it & 1)) * kMaxTeam + rank) * kTeamStride;
    if (lane <= kMoments)
        __hip_atomic_store(&slot[lane], lane < kMoments ? mine : stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // The record is the ONLY data the peers read, and it was stored write-through (agent-scope atomic stores are `sc1`
    // stores: they do not linger in this XCD's L2).  Draining this wave's stores before the arrival is therefore all the
    // release the hand-off needs; an agent-scope release FENCE would also write back every other dirty line of the XCD's
    // L2 (~1.7 us and more, once per iteration, in the serial part of it).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(&t.arrived[b], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// -> true when the team has to stop (member 0 said so, or the wait timed out); otherwise `mine` of
// lane k < 18 is the team total of moment k, added in member order
__device__ __forceinline__ bool team_collect(const IcpTeam &t, IcpCtrl *ctrl, int b, int it, int itBegin, int G,
                                             int lane, double &mine)
{
    const unsigned want = (unsigned)G * (unsigned)(it - itBegin + 1);
    const long long t0 = wall_clock64();
    bool timeout = false;
    // (relaxed poll; the records are then read with agent-scope atomic loads -- `sc1` loads, which bypass this CU's L1 --
    // of data that was stored write-through: no acquire fence, i.e. no L1 invalidation, is needed for them)
    while (__hip_atomic_load(&t.arrived[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(1);
        if (wall_clock64() - t0 > kTeamTimeoutTicks) { timeout = true; break; }
    }
    asm volatile("" ::: "memory");
    if (timeout) {
        if (lane == 0) __hip_atomic_store(&ctrl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return true;
    }
    // The records are read with agent-scope atomic loads, which the compiler keeps in program order with a full
    // round trip each (~0.7 us): spread them over the lanes.  Lane 18 g + k (g < 3) adds moment k of the members
    // g, g + 3, g + 6, ...; the three partial sums are then added in the order g = 0, 1, 2 -- the same on every
    // member, so all members still get bit-identical totals.  Lane 63 fetches member 0's stop flag meanwhile.
    const double *base = t.mom + ((size_t)b * 2 + (it & 1)) * kMaxTeam * kTeamStride;
    constexpr int kLanesPerGroup = kMoments, kGroups = 3;
    const int g = lane / kLanesPerGroup, k = lane - g * kLanesPerGroup;
    // (all loads of a lane in flight together, then added in member order: as a loop of load-add pairs every record
    // cost its lane a full round trip -- six in a row in a team of sixteen, the larger part of the exchange)
    constexpr int kPerLane = (kMaxTeam + kGroups - 1) / kGroups;
    double rec[kPerLane];
#pragma unroll
    for (int u = 0; u < kPerLane; ++u) {
        // (unconditional, from a clamped address: a load under a branch is waited for at the branch's end)
        const int r = min(g + kGroups * u, G - 1), kk = g < kGroups ? k : 0;
        rec[u] = __hip_atomic_load(&base[r * kTeamStride + kk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    double part = 0.0;
#pragma unroll
    for (int u = 0; u < kPerLane; ++u)
        if (g < kGroups && g + kGroups * u < G) part += rec[u];
    double stop = 0.0;
    if (lane == 63) stop = __hip_atomic_load(&base[kMoments], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double p1 = __shfl(part, (lane + kLanesPerGroup) & 63, kWave), p2 = __shfl(part, (lane + 2 * kLanesPerGroup) & 63, kWave);
    mine = (part + p1) + p2;   // meaningful in lanes < 18
    return __shfl(stop, 63, kWave) != 0.0;
}

// Work decomposition inside the workgroup (one pair): the NWAVE waves form NWAVE/TS query
// groups x TS target shares.  The TS waves of a query group hold the SAME Q x 64 queries and
// each scans 1/TS of every target tile; their partial (distance, chunk) results meet in LDS
// and query slot q is finished (resolve, gate, moments) by the wave with share q % TS.
// This keeps 16 waves (4 per SIMD) busy on a 1024-point pair with Q = 4 queries per lane, i.e.
// 12 LDS cycles of broadcast reads per 144 VALU issue cycles per wave, where a plain
// one-query-set-per-wave split would leave either the LDS pipe (Q = 1) or the latency hiding
// (4 waves per CU at Q = 4) as the limiter.  Register budget: 128 VGPRs (4 waves per SIMD),
// hence the moments are reduced per query slot straight into LDS and the Jacobi state lives
// in LDS instead of being carried in registers across the scan.
// GRID: 0 = all-pairs LDS scan, 1 = exact grid read from global memory (L2), 2 = exact grid staged
// into LDS at kernel entry (dynamic shared memory: (H+1) ints + n float4), 3 = sorted sweep
// (targets streamed through scalar loads, no LDS image)
// (512-thread workgroups are compiled for four waves per SIMD, 128 VGPRs, so that two of them share a CU)
// SCALE: similarity transforms (estimate_scale, or an initial transform with a scale; sorted-sweep kernels only): the
// plain kernels do not carry the extra multiplications.
// One pair (one member of its team): every iteration of the launch.  Inlined into icp_kernel below, which decides WHICH
// pair(s) this workgroup serves.
// (P: IcpParams in whatever address space the caller reads it from)
// HELP (persistent sorted-sweep kernels): role 0 = the pair's owner, role j + 1 = helper in slot j of pair b: it runs pass
// `passes - 1 - j` of every iteration from the state the owner publishes and hands the pass's moment sums back (see
// HelpPair in kernels.hpp and icp_kernel).
template <int BLOCK, int Q, int TS, int GRID, bool TEAM, bool SCALE, bool HELP, bool LATE, bool SHAREK, typename P>
__device__ __forceinline__ void icp_pair(const P &p, const int b, const int rank, const int G, const int itBegin,
                                         const int itEnd, const int role = 0)
{
    ICPFLOW_STAMP(0);
    static_assert(GRID == 0 || TS == 1, "grid / sweep searches do not split targets over waves");
    static_assert(GRID == 0 || GRID >= 3 || Q == 1, "grid search: one query per thread");
    extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
    constexpr int NWAVE = BLOCK / kWave;
    constexpr int NQG = NWAVE / TS;          // query groups
    static_assert((NWAVE % TS == 0 && Q % TS == 0) || TS == 1, "Q slots are dealt round-robin to the TS waves");
    __shared__ __attribute__((aligned(16))) unsigned char tileMem[GRID ? 16 : sizeof(ScanTile)];
    ScanTile *tile = reinterpret_cast<ScanTile *>(tileMem);
    __shared__ double red[NWAVE * kMoments];  // per-wave moment sums of the current iteration
    __shared__ double tot[kMoments];          // block totals of the 18 moments
    __shared__ double Nsh[16];                // 4x4 scratch of the closed-form rotation
    __shared__ double ksh[17];                // centroids, second moments and H parked across the solve
    __shared__ float bcast[20];               // R (9), T (3), active flag, prev rmse, rmse, -, origin (3)
    __shared__ float preL[12];                // pre-pose (R row-major 9, t 3), read back where it is applied
    __shared__ __attribute__((aligned(16))) float ring[kRing * 16];        // the last kRing states (R, T) and their rmse (cycle detection)
    __shared__ float combD[TS > 1 ? NWAVE * Q * kWave : 1];   // [wave][q][lane]
    __shared__ int combC[TS > 1 ? NWAVE * Q * kWave : 1];

    static_assert(!TEAM || GRID >= 3, "teams: sorted sweep only");
    static_assert(!HELP || (GRID == 4 && !TEAM && Q == 1), "helpers: sorted sweep with the LDS image, one workgroup per pair");
    __shared__ int helpSh[8];                 // [0] passes done by helpers this iteration (bit g), [1..3] their workgroups,
                                              // [4] helper: E(next iteration), [5] scratch
    // teams: the probes of a pass, shared by the member's waves (see the search phase)
    __shared__ float probeQ[4][TEAM && GRID == 4 ? kProbeCap : 1];   // (query position, previous neighbour)
    __shared__ float probeR[3][TEAM && GRID == 4 ? kProbeCap : 1];   // (distance, bound on the others, neighbour)
    __shared__ int probeCnt[2], probeNext[2]; // entries posted / handed out, by pass parity
    // teams: window scans shared by the member's waves (see the search phase): per owner wave the posted window and its
    // ticket / completion counters, the waves with an open posting, the arrivals at the help barriers
    // (SHAREK: an instantiation of its own -- the team kernel sits at its register limit, and the code alone costs the pairs
    // that never share 3-6 % -- chosen by the launch: padded width from kShareLaunchMinN on, ICPFLOW_OPT_NO_SHARED_SCANS off)
#ifdef ICPFLOW_NO_SHARE
    constexpr bool SHARE = false;
#else
    constexpr bool SHARE = SHAREK && TEAM && GRID == 4 && Q == 1;
#endif
    __shared__ int shHdr[SHARE ? NWAVE : 1][4];   // cb, ce, part length, parts | pass << 8 | unit << 16
    __shared__ int shNext[SHARE ? NWAVE : 1], shDone[SHARE ? NWAVE : 1], shLock[SHARE ? NWAVE : 1];
    __shared__ unsigned int shAcc[SHARE ? NWAVE : 1][3][kWave];   // per owner wave and lane: what the helpers' parts add up to (minimum, runner-up, chunk | tie)
    __shared__ unsigned int shState;   // arrivals at the help phases so far (low 20 bits: they only grow) | open postings (bit 20 + wave)
    [[maybe_unused]] const bool helping = HELP && role != 0;
    [[maybe_unused]] HelpPair *hp = nullptr;
    if constexpr (HELP) hp = (p.helpOn && p.help.pair != nullptr) ? p.help.pair + b : nullptr;
    IcpTeam team{};   // (a copy in the generic address space: `p` may live in the kernel-argument segment)
    if constexpr (TEAM) {
        team.wgPair = p.team.wgPair; team.wgRank = p.team.wgRank; team.teamSize = p.team.teamSize;
        team.arrived = p.team.arrived; team.mom = p.team.mom; team.maxWG = p.team.maxWG;
    }
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wave = tid >> 6;
    const int ts = wave % TS;                // target share of this wave
    const int qg = wave / TS;                // query group of this wave
    IcpCtrl *ctrl = p.ctrl;
    if (p.pairActive != nullptr && p.history != nullptr && p.pairActive[b] == 0) {
        // Not in the batch (options.d_pair_active; the caller handed the pair over as two empty clouds): the batch rule is
        // taken over the other pairs -- this one reports "arrived, converged" for every iteration of the launch, like a pair
        // whose trajectory has become periodic does for its remaining iterations, and leaves.  Its outputs are unspecified.
        if (rank == 0 && wave == 0)
            for (int k = itBegin + lane; k < p.maxIter; k += kWave)   // (to the cap, whatever this launch's itEnd: the first of two launches)
                __hip_atomic_fetch_add(&ctrl->tally[k], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0 && rank == 0) {
            IcpState *st0 = p.state + b;
#pragma unroll
            for (int k = 0; k < 9; ++k) st0->R[k] = (k % 4 == 0) ? 1.f : 0.f;
            st0->T[0] = st0->T[1] = st0->T[2] = 0.f;
            st0->rmse = 0.f; st0->s = 1.f; st0->active = 0; st0->iters = 0;
            if constexpr (HELP) {
                if (p.drainAt > 0) __hip_atomic_fetch_add(&ctrl->finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    if (p.stopMode == ICPFLOW_STOP_REFERENCE_ && itBegin > 0 && p.history == nullptr) {
        // (one launch per iteration) previous iteration satisfied the batch-global rule (or an earlier one did)
        if (ctrl->done || ctrl->notconv[itBegin - 1] == 0) {
            if (tid == 0 && b == 0) ctrl->done = 1;
            return;
        }
    }
    const bool sw = p.swap != nullptr && p.swap[b] != 0;
    CloudView xc, yc;
    xc.base = (sw ? p.Y : p.X) + (size_t)b * p.N * 4; xc.stride = 4; xc.n = (sw ? p.lenY : p.lenX)[b];
    yc.base = (sw ? p.X : p.Y) + (size_t)b * p.N * 4; yc.stride = 4; yc.n = (sw ? p.lenX : p.lenY)[b];
    PointXf pre;
    pre.kind = p.prePose ? XF_AFFINE : XF_NONE;
    pre.a = p.prePose ? affine_from_pose(p.prePose + (size_t)b * 16) : affine_identity();
    PointXf none;
    none.kind = XF_NONE;
    none.a = affine_identity();

    IcpState *st = p.state + b;
    // The current (R, T), the origin of the moments and the pre-pose live in LDS and are read back by
    // every wave where they are needed: nothing of it stays in registers across the rotation solve
    // (the 1024-thread kernel has 128 VGPRs; spilled values came back one dependent scratch load
    // at a time in the serial tail of every iteration).
    int active = 1;
    // RESUME: the instantiation that serves the SECOND of two speculative launches (icp_split_kernel) -- a pair resumed at its own
    // iteration finds the states of the cycle detection and its own-convergence bits in its history rows.  (Only there: the code
    // costs the other instantiations registers they do not have.)
    constexpr bool RESUME = BLOCK == 1024 && GRID == 4 && !TEAM && !SCALE && !HELP && LATE && Q == 1;
    const bool resumed = RESUME && itBegin > 0 && p.history != nullptr;
    if (itBegin == 0 || (resumed && itBegin < kRing)) {
        // state 0: identity (:140) or the caller's init_transform (:118-138)
        float s0 = (tid < 9 && tid % 4 == 0) ? 1.f : 0.f;
        if (p.initR != nullptr && tid < 12) s0 = tid < 9 ? p.initR[(size_t)b * 9 + tid] : p.initT[(size_t)b * 3 + tid - 9];
        const float scale0 = p.initS != nullptr ? p.initS[b] : 1.f;
        if (itBegin == 0) {
            if (tid < 12) bcast[tid] = s0;
            if (tid == 0) { bcast[12] = 1.f; bcast[13] = 0.f; bcast[14] = 0.f; bcast[15] = scale0; }
        }
        if (tid < 16) ring[tid] = tid < 12 ? s0 : (tid == 14 ? scale0 : 0.f);
        if (tid < kWave) {   // and its hash (same formula as in the loop), every lane of wave 0 the same value
            int hash = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) hash ^= state_hash_word(__shfl(s0, k, kWave), k);
            hash ^= state_hash_word(scale0, 12);
            if (tid == 13) ring[13] = __int_as_float(hash);
        }
    }
    if (itBegin > 0) {
        if (tid < 9) bcast[tid] = st->R[tid];
        if (tid < 3) bcast[9 + tid] = st->T[tid];
        active = st->active;
        if (tid == 0) { bcast[12] = active ? 1.f : 0.f; bcast[13] = st->rmse; bcast[14] = st->rmse; bcast[15] = st->s; }
        if constexpr (RESUME) {
            if (resumed && tid < kWave) {
                // The remembered states of the cycle detection are the pair's last history rows (row n - 1 holds state n: R, T, rmse,
                // scale, count; hash as in the loop), so a period is found at the very iteration ONE launch would have found it.
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int n = itBegin - (r * 4 + (lane >> 4)), w = lane & 15;   // state n, word w of its ring row
                    const bool on = n >= 1;
                    const float *h = p.history + ((size_t)(on ? n - 1 : 0) * p.B + b) * kHistStride;
                    const float hv = h[w < 13 ? w : (w == 15 ? 14 : 13)];
                    int hw = w < 12 ? state_hash_word(hv, w) : (w == 13 ? state_hash_word(hv, 12) : 0);
#pragma unroll
                    for (int o = 1; o < 16; o <<= 1) hw ^= __shfl_xor(hw, o, kWave);
                    if (on) ring[(n % kRing) * 16 + w] = w == 13 ? __int_as_float(hw) : hv;
                }
            }
        }
    }
    int itersDone = (itBegin == 0) ? 0 : st->iters;
    if (tid == 0) { probeCnt[0] = 0; probeCnt[1] = 0; probeNext[0] = 0; probeNext[1] = 0; }
    if (SHARE && tid == 0) shState = 0u;
    [[maybe_unused]] int hbSeq = 0;      // help barriers passed so far (workgroup-uniform)
    [[maybe_unused]] int probePar = 0;   // parity of the next pass's queue counters (workgroup-uniform)

    // per-pair origin of the moment accumulation: the first (pre-posed) source point
    if (tid == 0) {
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (xc.n > 0) {
            float rx, ry, rz;
            cloud_load(xc, 0, rx, ry, rz);
            xf_apply(pre, rx, ry, rz, o0, o1, o2);
        }
        bcast[16] = o0; bcast[17] = o1; bcast[18] = o2;
#pragma unroll
        for (int k = 0; k < 9; ++k) preL[k] = pre.a.m[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) preL[9 + k] = pre.a.t[k];
    }
    __syncthreads();

    const int per = NQG * kWave * Q;         // queries per pass of the workgroup
    const int ngroups = (xc.n + per - 1) / per;

    int itFirst = itBegin;   // first iteration of THIS workgroup on the pair (a helper joins later)
    if constexpr (HELP) {
        if (tid == 0) helpSh[0] = 0;          // (no pass of the first iteration comes from a helper; none ever without helpers)
        if (hp == nullptr) __syncthreads();
        if (hp != nullptr && !helping) {
            // owner: how many passes the pair has, and that it is being iterated on by this workgroup
            if (tid == 0) {
                __hip_atomic_store(&hp->passes, (xc.n + BLOCK - 1) / BLOCK, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&hp->iter, itBegin + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&p.help.owner[blockIdx.x], b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
        if (helping) {
            // helper: announce "(this workgroup, from iteration cur + 1 on)", then wait for the owner's state of an
            // iteration >= that.  Two iterations of lead: the owner reads the announcement at the END of an iteration.
            if (tid == 0) {
                int e = 0;
                const int cur = __hip_atomic_load(&hp->iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur > 0) {
                    const int from = cur + 2;
                    __hip_atomic_store(&hp->from[role - 1], ((int)blockIdx.x << 8) | from, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const long long t0 = wall_clock64();
                    for (;;) {
                        e = __hip_atomic_load(&hp->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (e < 0 || e >= from) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (wall_clock64() - t0 > kTeamTimeoutTicks) { e = -1; break; }
                    }
                } else e = -1;
                helpSh[4] = e;
            }
            __syncthreads();
            const int e0 = __builtin_amdgcn_readfirstlane(helpSh[4]);   // (workgroup-uniform: keep the loop counter scalar)
#ifdef ICPFLOW_TAIL_CLOCK
            if (tid == 0) atomicAdd(&g_help_stats[e0 > 0 ? 0 : 3], 1ull);
#endif
            if (e0 <= 0) return;                       // the pair is finished (or was never started)
            itFirst = e0 - 1;
            if (tid < 12) bcast[tid] = __hip_atomic_load(&p.help.state[((size_t)b * 2 + (e0 & 1)) * 16 + tid], __ATOMIC_RELAXED,
                                                         __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        }
    }
    int specChk = 0;  // speculative mode: first iteration not yet known to be complete-and-unconverged
    if constexpr (RESUME) {
        if (resumed) specChk = p.pairList != nullptr ? p.pairMeta[1] : itBegin;   // (a second launch: icp_split_kernel has looked below that)
    }
    int winLo = -1, winHi = -1;   // sorted sweep: this wave's target window of the previous iteration
    int prevNN = -2;              // certificates, single pass: this lane's gated neighbour of the previous iteration
    [[maybe_unused]] int ownedPrev = 0;   // several passes: bit g = THIS workgroup computed pass g in the previous iteration (workgroup-uniform)
    int sweepAxis = 0;            // sorted sweep: the sort axis of this pair (read once: a load from L2 at the top of every
                                  // iteration is a round trip that every wave of the workgroup sits out together)
    if constexpr (GRID >= 3) sweepAxis = __builtin_amdgcn_readfirstlane(p.sortAxis[b]);
    // iterations at which this pair was converged: four 32-bit words in LDS, read and written by wave 0 only (as loop-carried
    // registers they were the first victims of every change to the loop: spilled, and reloaded in the serial tail)
    __shared__ unsigned int ownConvSh[4];
    __shared__ int drainSh;      // this pair leaves the launch still moving (the launch is being drained, see the bookkeeping)
    if (tid == 0) drainSh = 0;
    if (tid < 4) ownConvSh[tid] = 0u;
    if constexpr (RESUME) {
        if (resumed && tid < kWave) {
            // (its own-convergence bits of the iterations it has behind it, from the rmse column of its history rows with the loop's test)
            unsigned long long m[2] = {0ull, 0ull};
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int q = r * kWave + lane;
                bool conv = false;
                if (q < itBegin && q < 128) {
                    const float rm = p.history[((size_t)q * p.B + b) * kHistStride + 12];
                    const float prev = q > 0 ? p.history[((size_t)(q - 1) * p.B + b) * kHistStride + 12] : 0.f;
                    const float rel = q == 0 ? 1.0f : (prev - rm) / prev;
                    conv = rel <= p.relThr;
                }
                m[r] = __ballot(conv);
            }
            if (tid < 4) ownConvSh[tid] = (unsigned int)(m[tid >> 1] >> (32 * (tid & 1)));
        }
    }
#ifdef ICPFLOW_TAIL_CLOCK
    const long long tcWall0 = wall_clock64();
    long long tcTail = 0, tcSearch = 0, tcLoop0 = clock64();
    if (threadIdx.x < 17) g_tcSh[threadIdx.x] = threadIdx.x == 16 ? clock64() : 0;
    __syncthreads();
#endif
    // Round 4: the bookkeeping of an iteration (history record, tally, batch-rule check, cycle detection: ~1.5 k clocks of
    // one wave) runs BEHIND the barrier that publishes the new (R, T), under the next iteration's search phase, instead of
    // in front of it with the other waves waiting.  Whether the pair goes on is therefore known one search phase late: the
    // flag bcast[12] is looked at by every wave at the same point -- behind the barrier that ends a search phase -- and a
    // pair that has finished runs one search phase for nothing (no exchange, no solve, nothing recorded).  Kernels with
    // helpers keep the old order (the owner's hand-off words are written with the bookkeeping).
    // (Only where it pays: launches of one workgroup per CU, whose length is their slowest pair's chain of iterations.  With
    // two workgroups per CU or a persistent grid the bookkeeping of one pair already runs under another pair's search, and
    // the search phase run for nothing costs throughput: config 4's shard 1.83 -> 1.98 ms; in a team every wave meets wave 0
    // at the barriers of the shared probes, so nothing is hidden: p.lateBook, set by launch_icp_variant.)
    // (a property of the instantiation -- as a run-time flag the two orders side by side cost the 1024-thread kernel 17 spilled
    // registers: LATE is chosen by icp_kernel)
    constexpr bool kLateBook = LATE && !HELP;
    for (int it = itFirst; it < itEnd; ++it) {
        // (only the state the launch starts from: a pair retired by an earlier launch; workgroup-uniform)
        if (it == itFirst && !active && (p.stopMode == ICPFLOW_STOP_PER_PAIR_ || p.history != nullptr)) {
            if (TEAM && G > 1 && rank == 0 && p.stopMode == ICPFLOW_STOP_REFERENCE_ && wave == 0)
                team_publish(team, b, it, 0, 0.0, 1.0, lane);
            break;
        }
        // Speculative mode, wave 0: the batch rule cannot hold at an iteration at which THIS pair was not converged, so
        // those are skipped without looking at the tally (pairs that do converge leave through the periodic
        // fast-forward; the shared tally line is read only in the in-between cases).  The tally of the first candidate
        // is fetched HERE, a whole search phase before it is looked at: an agent-scope load crosses the fabric
        // (the XCDs' L2s are not coherent), which is not something to wait for in the serial tail.
        unsigned long long specTally = 0ull;
        bool specLoaded = false;
        if (wave == 0 && p.history != nullptr && rank == 0 && !helping) {
            // (wave-uniform values kept scalar: left to itself the compiler vectorises this search over the lanes)
            specChk = __builtin_amdgcn_readfirstlane(specChk);
            {
                const unsigned long long lo = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ownConvSh[1]) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)ownConvSh[0]);
                const unsigned long long hi = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)ownConvSh[3]) << 32) |
                                              (unsigned)__builtin_amdgcn_readfirstlane((int)ownConvSh[2]);
                // first own-converged iteration >= specChk: count trailing zeros of the remaining bits
                unsigned long long rest = specChk < 64 ? (lo >> specChk) : 0ull;
                if (rest != 0ull) specChk += __builtin_ctzll(rest);
                else {
                    const int from = specChk < 64 ? 64 : specChk;
                    rest = from < 128 ? (hi >> (from - 64)) : 0ull;
                    specChk = rest != 0ull ? from + __builtin_ctzll(rest) : 128;
                }
                specChk = min(specChk, it);
            }
            if (specChk < it) {
                specTally = __hip_atomic_load(&ctrl->tally[specChk], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                specLoaded = true;
            }
        }
        // a launch that is drained for a second one (p.drainAt, persistent grids with helpers): the pairs finished so far, fetched
        // here like the tally and looked at with the bookkeeping
        [[maybe_unused]] int finishedSeen = 0;
        if constexpr (HELP) {
            if (wave == 0 && p.drainAt > 0 && !helping)
                finishedSeen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&ctrl->finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        }
        // this iteration's (R, T) and the origin of the moments, from LDS (dead after the search phase)
        float Rf[9], Tf[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Rf[k] = bcast[k];
        Tf[0] = bcast[9]; Tf[1] = bcast[10]; Tf[2] = bcast[11];
        const float ox = bcast[16], oy = bcast[17], oz = bcast[18];
        const float sc = SCALE ? bcast[15] : 1.f;   // scale of this iteration's transform (SCALE: estimate_scale / a scaled init)
        double macc[kMoments];  // wave-uniform running totals of this wave's query slots
#pragma unroll
        for (int k = 0; k < kMoments; ++k) macc[k] = 0.0;
        // ------------- NN + gate + moments (one pass over the source points) ---------------
        // 18 moments of one query slot, each reduced over the wave on the VALU (non-inliers
        // contribute exact zeros) and accumulated wave-uniformly
#define ICPFLOW_ACC(k, expr) macc[k] += wave_sum_uniform(expr);
#define ICPFLOW_ACC_ALL()                                                                     \
        {                                                                                     \
            ICPFLOW_ACC(0, inl ? 1.0 : 0.0)                                                   \
            ICPFLOW_ACC(1, ax) ICPFLOW_ACC(2, ay) ICPFLOW_ACC(3, az)                          \
            ICPFLOW_ACC(4, bx) ICPFLOW_ACC(5, by) ICPFLOW_ACC(6, bz)                          \
            ICPFLOW_ACC(7, ax * bx) ICPFLOW_ACC(8, ax * by) ICPFLOW_ACC(9, ax * bz)           \
            ICPFLOW_ACC(10, ay * bx) ICPFLOW_ACC(11, ay * by) ICPFLOW_ACC(12, ay * bz)        \
            ICPFLOW_ACC(13, az * bx) ICPFLOW_ACC(14, az * by) ICPFLOW_ACC(15, az * bz)        \
            ICPFLOW_ACC(16, ax * ax + ay * ay + az * az)                                      \
            ICPFLOW_ACC(17, bx * bx + by * by + bz * bz)                                      \
        }
        if constexpr (GRID == 3 || GRID == 4) {
            const int NP16 = (p.N + kChunk - 1) / kChunk * kChunk;
            const int np16 = (yc.n + kChunk - 1) / kChunk * kChunk;
            const float *gx = p.sortYsoa + (size_t)b * 3 * NP16;
            const float *gy = gx + NP16;
            const float *gz = gy + NP16;
            const float4 *ys = p.sortY + (size_t)b * p.N;
            const float4 *xs = p.sortX + (size_t)b * p.N;
            const int axis = sweepAxis;   // the pair's sort key (sortdir.hpp): a coordinate (0 .. 2) or a horizontal direction (3 ..)
            float dirX = 0.f, dirY = 0.f;
            if (axis >= 3) sort_dir(axis, dirX, dirY);
            // GRID == 4: the sorted fixed cloud is staged into LDS once per launch and every
            // per-iteration access (window search, scan, resolve) stays on chip
            // dynamic LDS: [moment sums per (pass, wave): redPasses x NWAVE x 18 doubles][image][records][own points]
            unsigned char *dyn = dynLds + (size_t)p.redPasses * (NWAVE * kMoments * sizeof(double));
            float *lx = reinterpret_cast<float *>(dyn), *ly = lx + NP16, *lz = ly + NP16;
            const bool perPass = !TEAM && p.redPasses != 0;
            double *redDyn = reinterpret_cast<double *>(dynLds);
            // Neighbour certificates.  The search answers one question per query: which target is nearest, and is it
            // inside the gate.  After the first iterations the answer hardly ever changes, and that can be PROVEN
            // without searching: a search at position q0 that found the neighbour j1 also knows a lower bound L on the
            // distance of every OTHER target (the runner-up of the scan; targets outside the scanned window differ by
            // more than the window's half-width along the sort axis).  After the query has moved to q, every other
            // target is still at least L - |q - q0| away (triangle inequality on the fp32 points themselves), so
            //   (A)  |q - t_j1| < L - |q - q0|             => j1 is still the unique nearest neighbour: its distance is
            //        evaluated with the instruction sequence of the scan (same bits), gate decision as usual;
            //   (B)  |q - t_j1| > thres and L - |q - q0| > 1.01 thres   => nothing is inside the gate.
            // A query that holds neither certificate searches: its window is its own +-m, and a wave scans the union of
            // the windows of its uncertified lanes only -- nothing at all once every lane is certified.  Each query keeps
            // (q0, L; j1) in LDS behind the image.  Minimum, gate decision and neighbour are those of the full search in
            // every case: results are bit-identical (test_adaptive_windows_change_nothing), the iterations after the
            // first few cost a distance evaluation per query instead of a window scan.
            constexpr bool REC = (GRID == 4);
            // Team member `rank` takes the units (64 consecutive sorted queries: what a wave searches at a time) rank,
            // rank + G, rank + 2 G, ... of the pair.  (Round 3; a contiguous share per member before.  The cost of a unit
            // follows the density of the fixed cloud around it, so a contiguous share could hold all the dense units of a
            // pair: on the ragged real-shape batch with matched sizes member 0 of the slowest team searched for 25 k clocks per
            // iteration and waited 53 k for the others.)  Local slot li = the li-th query this member takes, in sorted order.
            const int unitsAll = (xc.n + kWave - 1) / kWave;
            const bool dealt = TEAM && G > 1;
            const int myCount = dealt ? (unitsAll > rank ? (unitsAll - rank + G - 1) / G : 0) * kWave : xc.n;
            // records of this workgroup's queries (indexed by local slot), if its share fits the room behind the image
            float4 *rec = reinterpret_cast<float4 *>(dyn + (size_t)NP16 * 12);
            int *recJ = reinterpret_cast<int *>(dyn + (size_t)NP16 * 12 + (size_t)p.recCap * 16);
            const bool recOn = REC && p.recCap > 0 && myCount <= p.recCap;
            // (and the query's own point, pre-pose applied, when there is room: read from L2 once, not once per iteration)
            float *x0c = reinterpret_cast<float *>(dyn + (size_t)NP16 * 12 + (size_t)p.recCap * 20);
            const bool x0On = recOn && p.x0Cache != 0;
            if (GRID == 4 && it == itFirst) {
                for (int k = tid; k < np16; k += BLOCK) { lx[k] = gx[k]; ly[k] = gy[k]; lz[k] = gz[k]; }
                __syncthreads();
            }
            const float *keyf = (GRID == 4) ? (axis == 0 ? lx : (axis == 1 ? ly : lz))
                                            : (axis == 0 ? gx : (axis == 1 ? gy : gz));
            // the key of sorted target j (j < yc.n) and of a point: the coordinate, or the direction's two instructions (sortdir.hpp)
            auto tkey = [&](int j) -> float {
                return axis >= 3 ? sort_key_dir(dirX, dirY, (GRID == 4 ? lx : gx)[j], (GRID == 4 ? ly : gy)[j]) : keyf[j];
            };
            auto pkey = [&](float x, float y, float z) -> float { return sort_key_of(axis, dirX, dirY, x, y, z); };
            constexpr int PER = BLOCK * Q;            // a wave owns 64 CONSECUTIVE sorted queries
            const int ngr = (myCount + PER - 1) / PER;
            double fold[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            bool reuseMoments = false;   // this wave's 18 sums are those of the previous iteration (still in `red`)
            [[maybe_unused]] int helpedPasses = 0;   // owner: passes of this iteration that helpers deliver (bit g)
            if constexpr (HELP) helpedPasses = helping ? 0 : __builtin_amdgcn_readfirstlane(helpSh[0]);
            // ---- shared window scans (teams, round 5) --------------------------------------------------------------------
            // Where a pair still slides along a face of its cluster, a few waves of a member scan windows of 700-1000 targets
            // (a face across the sort axis: every target of the face has the queries' key) while the others hold certificates
            // and wait at the next barrier: 60-77 k clocks for those units against 2-5 k (ragged real-shape batch, r04 profile),
            // and a wave that scans alone on its SIMD is bound by the latency of its own LDS reads, not by the VALU.
            // So a wave whose window holds kShareMinW targets or more POSTS it (window, pass, unit) in LDS, cut into parts of
            // >= kSharePartMin targets, and takes parts by ticket itself; every wave that reaches one of the search phase's
            // barriers first looks for open postings and takes parts too (it forms the owner's 64 queries itself: same
            // operations, same bits) and adds its (minimum, runner-up, chunk | tie) per lane to the owner's accumulator in
            // LDS under the posting's lock.  The owner merges that accumulator with the parts it scanned itself:
            // minimum with the first-chunk rule, tie flag and second-smallest chunk minimum combine exactly (the merge
            // gives what ONE scan over the whole window gives, in any order), so results are bit-identical.
            // A barrier of the search phase becomes: count the arrival, help while others have not arrived, then the barrier.
            // An owner arrives only after its posting is merged, so "everybody has arrived" means no part is open or in flight.
            [[maybe_unused]] bool shareOn = false;
            if constexpr (SHARE) shareOn = recOn && p.shareScans != 0 && yc.n >= kShareMinN;
            // the moved query of lane `lane` of unit `uw` of pass `gp` (what the unit's own wave computes below)
            [[maybe_unused]] auto unit_query = [&](int gp, int uw, float &ux, float &uy, float &uz) {
                const int li = gp * PER + uw * kWave + lane;
                const int i = dealt ? ((li >> 6) * G + rank) * kWave + lane : li;
                ux = uy = uz = 0.f;
                if (li < myCount && i < xc.n) {
                    float ax0, ay0, az0;
                    if (x0On && it > itFirst) {
                        ax0 = x0c[li]; ay0 = x0c[p.recCap + li]; az0 = x0c[2 * p.recCap + li];
                    } else {
                        const float4 s4 = xs[i];
                        ax0 = s4.x; ay0 = s4.y; az0 = s4.z;
                        if (p.sortedRaw) {
                            ax0 = fmaf(s4.z, preL[2], fmaf(s4.y, preL[1], s4.x * preL[0])) + preL[9];
                            ay0 = fmaf(s4.z, preL[5], fmaf(s4.y, preL[4], s4.x * preL[3])) + preL[10];
                            az0 = fmaf(s4.z, preL[8], fmaf(s4.y, preL[7], s4.x * preL[6])) + preL[11];
                        }
                    }
                    float rx = fmaf(az0, Rf[6], fmaf(ay0, Rf[3], ax0 * Rf[0]));
                    float ry = fmaf(az0, Rf[7], fmaf(ay0, Rf[4], ax0 * Rf[1]));
                    float rz = fmaf(az0, Rf[8], fmaf(ay0, Rf[5], ax0 * Rf[2]));
                    if constexpr (SCALE) { rx *= sc; ry *= sc; rz *= sc; }
                    ux = rx + Tf[0]; uy = ry + Tf[1]; uz = rz + Tf[2];
                }
            };
            [[maybe_unused]] auto help_phase = [&]() {
                if constexpr (SHARE) {
                    if (shareOn) {
                        ++hbSeq;
                        // (one word holds arrivals and open postings: the last wave to arrive sees "everybody here, nothing open" in the
                        // value its own arrival returns and goes straight on)
                        const unsigned int target = (unsigned)hbSeq * NWAVE;
                        unsigned int state = 0u;
                        if (lane == 0) state = atomicAdd(&shState, 1u) + 1u;
                        state = (unsigned)__builtin_amdgcn_readfirstlane((int)state);
                        unsigned int exhausted = 0u;
                        for (bool fresh = true;; fresh = false) {
                            if (!fresh) state = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(&shState, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                            const unsigned int open = (state >> 20) & ~exhausted;
                            if (open != 0u) {
                                const int w = __builtin_ctz(open);
                                const int hcb = __builtin_amdgcn_readfirstlane(shHdr[w][0]), hce = __builtin_amdgcn_readfirstlane(shHdr[w][1]);
                                const int hlen = __builtin_amdgcn_readfirstlane(shHdr[w][2]), hword = __builtin_amdgcn_readfirstlane(shHdr[w][3]);
                                int t = 0;
                                if (lane == 0) t = atomicAdd(&shNext[w], 1);
                                t = __builtin_amdgcn_readfirstlane(t);
                                if (t >= (hword & 0xff)) { exhausted |= 1u << w; continue; }
                                float hx[1], hy[1], hz[1];
                                unit_query((hword >> 8) & 0xff, (hword >> 16) & 0xff, hx[0], hy[0], hz[0]);
                                ScanAcc<1> ha;
                                bool ht[1] = {false};
                                float hs[1] = {kInf};
                                scan_init(ha);
                                const int c0 = hcb + t * hlen;
                                scan_range_tie<1, true>(reinterpret_cast<const float4 *>(lx), reinterpret_cast<const float4 *>(ly),
                                                        reinterpret_cast<const float4 *>(lz), c0, min(c0 + hlen, hce), hx, hy, hz, ha, ht, hs);
                                // add this part to the owner's accumulator (any order gives the same result), under the posting's lock
                                for (;;) {
                                    int got = 1;
                                    if (lane == 0) got = atomicCAS(&shLock[w], 0, 1);
                                    if (__builtin_amdgcn_readfirstlane(got) == 0) break;
                                    __builtin_amdgcn_s_sleep(0);
                                }
                                {
                                    const float b1 = __uint_as_float(shAcc[w][0][lane]), s1 = __uint_as_float(shAcc[w][1][lane]);
                                    const unsigned int w1 = shAcc[w][2][lane];
                                    const float b2 = ha.best[0];
                                    const float sec = min_nonneg(max_nonneg(b1, b2), min_nonneg(s1, hs[0]));
                                    unsigned int wn = w1;
                                    if (b2 < b1) wn = (unsigned)ha.chunk[0] | (ht[0] ? 0x40000000u : 0u);
                                    else if (b2 == b1) wn = min(w1 & 0x3fffffffu, (unsigned)ha.chunk[0]) | 0x40000000u;
                                    shAcc[w][0][lane] = __float_as_uint(fminf(b1, b2));
                                    shAcc[w][1][lane] = __float_as_uint(sec);
                                    shAcc[w][2][lane] = wn;
                                }
                                if (lane == 0) {
                                    __hip_atomic_store(&shLock[w], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (behind the stores: one wave's LDS operations execute in order)
                                    atomicAdd(&shDone[w], 1);
                                }
                                continue;
                            }
                            if ((state & 0xfffffu) >= target && (state >> 20) == 0u) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                }
            };
            for (int g = 0; g < ngr; ++g) {
                if constexpr (HELP) {
                    if (helping ? g != ngr - role : ((helpedPasses >> g) & 1) != 0) continue;   // (workgroup-uniform)
                }
                // Which 64 queries of the pass a wave takes: wave w takes unit w -- except in a team member's pass of fewer
                // units than waves (a member holds 4, 8 or 12 units, §3.5 of DESIGN.md), where the units go to the HIGH
                // waves: wave 0, which starts every search phase a bookkeeping late (kLateBook), then has no unit and reaches
                // the first barrier of the shared probes with the others.
                int unitWave = wave;
                if constexpr (TEAM) {
                    const int unitsHere = min(NWAVE, (myCount - g * PER + kWave - 1) / kWave);
                    unitWave = wave - (NWAVE - unitsHere);      // < 0: this wave has no unit in this pass
                }
                float x0x[Q], x0y[Q], x0z[Q], qx[Q], qy[Q], qz[Q];
                bool live[Q];
                float lo = kInf, hi = -kInf;
                int prevWord[Q];   // the neighbour word of this query's record as it was found (-2: none read: first iteration of the launch)
#pragma unroll
                for (int q = 0; q < Q; ++q) prevWord[q] = -2;
                float recM[Q];   // this query's half-window; < 0: certified, takes no part in the search
                int certJ[Q];    // certificate (A): the neighbour (slot of the sorted image)
                float certD[Q];  // certified squared distance (inf: outside the gate)
                float newL[Q];   // >= 0: this query's record is rewritten (bound on every target but the neighbour)
#pragma unroll
                for (int q = 0; q < Q; ++q) { recM[q] = p.sweepMargin; certJ[q] = -1; certD[q] = kInf; newL[q] = -1.f; }
                // (window of the queries that may be outside the gate: 30 % beyond the gate's own, so that a scan which finds
                // nothing closer certifies "outside" for the next three centimetres of drift.  With probes of up to eight
                // blocks, 1.04 / 1.1 / 1.2 / 1.3 / 1.4 measure within 2 % of each other, 1.3 in front)
#ifndef ICPFLOW_CERT_MARGIN
#define ICPFLOW_CERT_MARGIN 1.3f
#endif
                const float certMargin = ICPFLOW_CERT_MARGIN * p.sweepMargin;
                const float gateOut = p.sweepMargin;                  // > thres with 1 % to spare (1.01 thres)
                ICPFLOW_STAMP(1);
#ifdef ICPFLOW_TAIL_CLOCK
                const long long unit0 = clock64();
#endif
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int li = g * PER + (unitWave * Q + q) * kWave + lane;               // local slot
                    const int i = dealt ? ((li >> 6) * G + rank) * kWave + lane : li;        // sorted query
                    live[q] = unitWave >= 0 && li < myCount && i < xc.n;
                    x0x[q] = x0y[q] = x0z[q] = 0.f;
                    qx[q] = qy[q] = qz[q] = 0.f;
                    if (live[q]) {
                        if (x0On && it > itFirst) {
                            x0x[q] = x0c[li]; x0y[q] = x0c[p.recCap + li]; x0z[q] = x0c[2 * p.recCap + li];
                        } else {
                        const float4 s4 = xs[i];   // sorted; pre-pose (utils_icp.py:21) applied by the sort or here
                        x0x[q] = s4.x; x0y[q] = s4.y; x0z[q] = s4.z;
                        if (p.sortedRaw) {   // pre-pose (row-major rotation, translation) parked in LDS
                            x0x[q] = fmaf(s4.z, preL[2], fmaf(s4.y, preL[1], s4.x * preL[0])) + preL[9];
                            x0y[q] = fmaf(s4.z, preL[5], fmaf(s4.y, preL[4], s4.x * preL[3])) + preL[10];
                            x0z[q] = fmaf(s4.z, preL[8], fmaf(s4.y, preL[7], s4.x * preL[6])) + preL[11];
                        }
                        if (x0On) { x0c[li] = x0x[q]; x0c[p.recCap + li] = x0y[q]; x0c[2 * p.recCap + li] = x0z[q]; }
                        }
                        float rx = fmaf(x0z[q], Rf[6], fmaf(x0y[q], Rf[3], x0x[q] * Rf[0]));  // :177, :395
                        float ry = fmaf(x0z[q], Rf[7], fmaf(x0y[q], Rf[4], x0x[q] * Rf[1]));
                        float rz = fmaf(x0z[q], Rf[8], fmaf(x0y[q], Rf[5], x0x[q] * Rf[2]));
                        if constexpr (SCALE) { rx *= sc; ry *= sc; rz *= sc; }   // (similarity transforms only)
                        qx[q] = rx + Tf[0]; qy[q] = ry + Tf[1]; qz[q] = rz + Tf[2];
#ifdef ICPFLOW_DEBUG_SOLVE
                        if (b == g_dbg_pair && it == itBegin) {
                            const int orig = __float_as_int(xs[i].w);
                            if (orig >= 0 && orig < 4096) { g_dbg_xt[orig * 3] = qx[q]; g_dbg_xt[orig * 3 + 1] = qy[q]; g_dbg_xt[orig * 3 + 2] = qz[q]; }
                        }
#endif
                        float m = p.sweepMargin;
                        if (recOn) {
                            m = certMargin;   // first iteration: nothing known
                            if (it > itFirst) {
                                const float4 o = rec[li];
                                // (the record's neighbour word: slot | "was gated when the unit's moments were last formed" << 30, or -1)
                                const int word = recJ[li];
                                const int j1 = word < 0 ? word : (word & 0x3fffffff);
                                prevWord[q] = word;
                                const float ex = qx[q] - o.x, ey = qy[q] - o.y, ez = qz[q] - o.z;
                                // (raw v_sqrt_f32, 1 ulp: every bound below carries a relative margin of 1e-6, 8 ulp)
                                const float dq = __builtin_amdgcn_sqrtf(fmaf(ez, ez, fmaf(ey, ey, ex * ex)));
                                const float others = o.w * 0.999999f - dq * 1.000001f - 1e-6f;
                                float e2 = kInf;
                                if (j1 >= 0) e2 = sqdist(qx[q], qy[q], qz[q], lx[j1], ly[j1], lz[j1]);
                                const float e1 = __builtin_amdgcn_sqrtf(e2);
                                certJ[q] = j1;   // (also the hint of the probe below)
#ifdef ICPFLOW_CERT_STATS
                                if (false && it == 40 && !(e1 * 1.000001f + 1e-6f < others) && !(others > gateOut && !(e2 <= p.thr2)))
                                    printf("b %d i %d j1 %d e1 %.6f L %.6f dq %.3e others %.6f\n", b, i, j1, e1, o.w, dq, others);
#endif
                                if (e1 * 1.000001f + 1e-6f < others) { m = -1.f; certD[q] = e2; }   // (A)
                                else if (others > gateOut && !(e2 <= p.thr2)) m = -1.f;             // (B)
                                else if (e2 <= p.thr2) m = p.sweepMargin;   // certainly gated: the gate's own window
                            }
                        }
                        recM[q] = m;
#ifdef ICPFLOW_CERT_STATS
                        if (recOn && it > itFirst && it < 128 && p.occBits != nullptr && (g_stats_block < 0 || g_stats_block == (int)blockIdx.x)) {
                            const float *hdr = p.occHdr + ((size_t)b * 2 + 1) * 8;
                            const uint32_t *bits = p.occBits + ((size_t)b * 2 + 1) * kOccRings * kOccWords;
                            const int gnx = __float_as_int(hdr[5]), gny = __float_as_int(hdr[6]), gnz = __float_as_int(hdr[7]);
                            bool empty = false;
                            if (gnx > 0) {
                                const float ux = floorf((qx[q] - hdr[0]) * hdr[3]), uy = floorf((qy[q] - hdr[1]) * hdr[3]), uz = floorf((qz[q] - hdr[2]) * hdr[3]);
                                const bool inside = ux >= 0.f && ux < (float)gnx && uy >= 0.f && uy < (float)gny && uz >= 0.f && uz < (float)gnz;
                                empty = !inside;
                                if (inside) { const int c = ((int)ux * gny + (int)uy) * gnz + (int)uz; empty = ((bits[c >> 5] >> (c & 31)) & 1u) == 0u; }
                            }
                            if (m >= 0.f) { atomicAdd(&g_occ_cert[it * 4 + 0], 1ull); if (empty) atomicAdd(&g_occ_cert[it * 4 + 1], 1ull); }
                            if (empty) atomicAdd(&g_occ_cert[it * 4 + 3], 1ull);
                        }
#endif
                    }
                }
                // Probes.  The few queries of a wave that hold no certificate are settled eight at a time, each by a row
                // of 8 lanes that evaluates the 64 sorted targets around the query's previous neighbour (8 per lane, one
                // 16-byte LDS read per coordinate and four targets).  The targets NOT evaluated lie beyond the ends of the
                // evaluated range along the sort axis, at least rho away: the probe is conclusive when rho exceeds the
                // minimum found (then that is the nearest neighbour) or, for a minimum outside the gate, when rho exceeds
                // 1.01 thres; otherwise the range grows by 64 targets on its short side, up to kProbeSteps (kProbeStepsLong) times.
                // Inconclusive probes, equal minima (the first-index rule is the scan's business), queries without a
                // previous neighbour and waves with many uncertified lanes take the window scan.
                ScanAcc<Q> acc;
                bool tie[Q];
                float second[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) { tie[q] = false; second[q] = kInf; }
                scan_init(acc);
                int cb = 0, ce = 0;
                // the window scan of this wave: the part of the sort axis in which its searching queries can find their neighbours
                auto scan_window = [&](const bool mayShare) {
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const float qa = pkey(qx[q], qy[q], qz[q]);
                        const float qm = recM[q] + sort_key_slack(axis, qx[q], qy[q], recM[q]);   // (a computed key: the window gives its rounding away)
                        if (live[q] && recM[q] >= 0.f) { lo = fminf(lo, qa - qm); hi = fmaxf(hi, qa + qm); }
                    }
                    ICPFLOW_STAMP(11);
                    bool anyScan = false;
#pragma unroll
                    for (int q = 0; q < Q; ++q) anyScan = anyScan || __ballot(live[q] && recM[q] >= 0.f) != 0ull;
                    if (anyScan) {
                        lo = wave_min_uniform(lo);
                        hi = wave_max_uniform(hi);
                    }
                    if (lo <= hi) {  // wave has searching queries (wave-uniform)
                        // single pass: this wave's window of the previous iteration is the hint
                        int jlo = winLo, jhi = winHi;
                        if (ngr == 1 && winHi >= 0)
                            sorted_window_hint_fn(tkey, yc.n, lo, hi, lane, jlo, jhi);
                        else
                            sorted_window_fn(tkey, yc.n, lo, hi, lane, jlo, jhi);
                        winLo = jlo; winHi = jhi;
                        cb = (jlo / kChunk) * kChunk;
                        ce = min((jhi + kChunk - 1) / kChunk * kChunk, np16);
                        ICPFLOW_STAMP(12);
#ifdef ICPFLOW_PHASE_TIMING
                        if ((int)blockIdx.x == g_stamp_block && lane == 0) g_wave_stamps[wave * 16 + 15] = ce - cb;
#endif
                        if (GRID == 4) {
                            if (REC && recOn) {
                                // (ONE call site of the scan, in a loop that runs once unless the window is shared)
                                int c0 = cb, c1 = ce;
                                [[maybe_unused]] bool share = false;
                                [[maybe_unused]] int nParts = 1, partLen = 0, mine = 1;
                                if constexpr (SHARE) {
                                    share = mayShare && shareOn && it > itFirst && ce - cb >= kShareMinW;
                                    if (share) {
                                        const int W = ce - cb;
                                        nParts = min(kShareParts, W / kSharePartMin);
                                        partLen = ((W + nParts - 1) / nParts + kChunk - 1) / kChunk * kChunk;
                                        nParts = (W + partLen - 1) / partLen;
                                        shAcc[wave][0][lane] = 0x7f800000u; shAcc[wave][1][lane] = 0x7f800000u; shAcc[wave][2][lane] = 0x3fffffffu;
                                        if (lane == 0) {
                                            shHdr[wave][0] = cb; shHdr[wave][1] = ce; shHdr[wave][2] = partLen;
                                            shHdr[wave][3] = nParts | (g << 8) | (unitWave << 16);
                                            shNext[wave] = 1; shDone[wave] = 0; shLock[wave] = 0;   // (part 0 is the owner's)
                                            atomicOr(&shState, 1u << (20 + wave));     // (LDS operations of one wave execute in order)
                                        }
                                        c1 = min(cb + partLen, ce);
                                    }
                                }
                                for (;;) {
                                    scan_range_tie<Q, true>(reinterpret_cast<const float4 *>(lx), reinterpret_cast<const float4 *>(ly),
                                                            reinterpret_cast<const float4 *>(lz), c0, c1, qx, qy, qz, acc, tie, second);
                                    if constexpr (!SHARE) break;
                                    if (!share) break;
                                    int t = 0;
                                    if (lane == 0) t = atomicAdd(&shNext[wave], 1);
                                    t = __builtin_amdgcn_readfirstlane(t);
                                    if (t >= nParts) break;
                                    ++mine;
                                    c0 = cb + t * partLen;
                                    c1 = min(c0 + partLen, ce);
                                }
                                if constexpr (SHARE) {
                                    if (share) {
                                        if (nParts > mine) {
                                            while (__builtin_amdgcn_readfirstlane(__hip_atomic_load(&shDone[wave], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < nParts - mine)
                                                __builtin_amdgcn_s_sleep(1);
                                            asm volatile("" ::: "memory");
                                            const float b2 = __uint_as_float(shAcc[wave][0][lane]), s2 = __uint_as_float(shAcc[wave][1][lane]);
                                            const unsigned int w2 = shAcc[wave][2][lane];
                                            // second-smallest chunk minimum of the union: of {best, second} of either side
                                            second[0] = min_nonneg(max_nonneg(acc.best[0], b2), min_nonneg(second[0], s2));
                                            if (b2 < acc.best[0]) { acc.best[0] = b2; acc.chunk[0] = (int)(w2 & 0x3fffffffu); tie[0] = (w2 & 0x40000000u) != 0u; }
                                            else if (b2 == acc.best[0]) { tie[0] = true; acc.chunk[0] = min(acc.chunk[0], (int)(w2 & 0x3fffffffu)); }
                                        }
                                        if (lane == 0) atomicAnd(&shState, ~(1u << (20 + wave)));
                                    }
                                }
                            } else
                                scan_range_tie<Q>(reinterpret_cast<const float4 *>(lx), reinterpret_cast<const float4 *>(ly),
                                                  reinterpret_cast<const float4 *>(lz), cb, ce, qx, qy, qz, acc, tie);
                        } else
                            scan_range_tie_uniform<Q>(gx, gy, gz, cb, ce, qx, qy, qz, acc, tie);
                    }
                };
                bool scanNow = true;      // no shared probes in this pass: the wave scans right here (if it has searching queries)
                if constexpr (TEAM) {
                if (REC && recOn && it > itFirst) {
                    const bool longProbes = ngr > 1 || yc.n > 1024;   // (team members: one pass each, but of a long cloud)
                    // (and every lane of the wave where the alternative is a scan of a long window of a long cloud: this wave's window
                    // of the previous search held more than 512 targets -- a dense 10000-point cluster; on the ragged real-shape
                    // batch the ICP launch 1.33 -> 0.94 ms; the demo frame's wall, long but thin, keeps its short windows and 32)
                    const bool wideWindows = yc.n > 4096 && winHi - winLo > kWideWindow;
                    const int probeSteps = longProbes ? kProbeStepsLong : kProbeSteps;
                    const int probeMax = longProbes ? (wideWindows ? kWave : kProbeMaxLong) : kProbeMax;
                    // Round 4: the probes of a pass are SHARED by the workgroup.  The uncertified queries are few, but they
                    // cluster: where the pair still slides along a face, one wave holds twenty of them and its neighbours
                    // none, and the pass lasted as long as that wave's three rounds of probes while fifteen waves waited at
                    // the barrier behind the moments.  So every wave posts its uncertified queries (position, previous
                    // neighbour) to a queue in LDS, and after a barrier the waves serve the queue, eight entries at a time
                    // drawn from a ticket, each entry by a row of eight lanes exactly as before; the answers go back through
                    // LDS.  A wave that has to scan its window anyway -- more uncertified lanes than probes pay for, or a lane
                    // without a previous neighbour -- posts nothing and scans WHILE the others probe (then takes what tickets
                    // are left).  A probe's answer depends on its entry alone: which wave served it changes nothing.
                    static_assert(Q == 1, "teams: one query per lane");
                    const bool wants = live[0] && recM[0] >= 0.f && certJ[0] >= 0;
                    const unsigned long long need = __ballot(wants);
                    const int nNeed = __popcll(need);
                    const bool mustScan = nNeed > probeMax || __ballot(live[0] && recM[0] >= 0.f && certJ[0] < 0) != 0ull;
                    int myEntry = -1;
                    if (!mustScan && nNeed > 0) {   // (wave-uniform)
                        int base = 0;
                        if (lane == 0) base = atomicAdd(&probeCnt[probePar], nNeed);
                        base = __builtin_amdgcn_readfirstlane(base);
                        const int e = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
                        if (wants && e < kProbeCap) {        // (a full queue leaves the query to the window scan)
                            probeQ[0][e] = qx[0]; probeQ[1][e] = qy[0]; probeQ[2][e] = qz[0]; probeQ[3][e] = __int_as_float(certJ[0]);
                            myEntry = e;
                        }
                    }
                    barrier_lds_only();
                    const int posted = min(__builtin_amdgcn_readfirstlane(probeCnt[probePar]), kProbeCap);
                    if (tid == 0) { probeCnt[probePar ^ 1] = 0; probeNext[probePar ^ 1] = 0; }   // the next pass's (last read a whole pass ago)
                    if (mustScan) scan_window(false);   // (the others probe meanwhile and meet this wave at the barrier below: nobody to share with)
                    const int row = lane >> 3, li = lane & 7;
                    for (;;) {
                        int e0 = 0;
                        if (lane == 0) e0 = atomicAdd(&probeNext[probePar], 8);
                        e0 = __builtin_amdgcn_readfirstlane(e0);
                        if (e0 >= posted) break;
                        const int ent = e0 + row;
                        const bool rowOn = ent < posted;
                        const int entC = min(ent, kProbeCap - 1);
                        const float sx = probeQ[0][entC], sy = probeQ[1][entC], sz = probeQ[2][entC];
                        const int j1 = rowOn ? __float_as_int(probeQ[3][entC]) : 0;
                        const float qa = pkey(sx, sy, sz);
                        int a = min(max((j1 & ~7) - 32, 0), max(np16 - 64, 0));   // evaluated so far: [a, bEnd)
                        int bEnd = a + 64;
                        int bs = a, be = bEnd;                                    // this step's block
                        float lb = kInf, lsec = kInf;                             // this lane's minimum and runner-up
                        int lslot = 0;
                        bool done = !rowOn, ok = false, isNN = false;
                        float rowbest = kInf, rho = 0.f;
                        for (int step = 0; step < probeSteps; ++step) {
                            const int t0 = bs + 8 * li;
                            if (!done && t0 < be && t0 < np16) {
#pragma unroll
                                for (int h = 0; h < 2; ++h) {     // (four targets at a time: twelve registers of targets, not twenty-four)
                                    const float4 xa = *reinterpret_cast<const float4 *>(lx + t0 + 4 * h);
                                    const float4 ya = *reinterpret_cast<const float4 *>(ly + t0 + 4 * h);
                                    const float4 za = *reinterpret_cast<const float4 *>(lz + t0 + 4 * h);
                                    const float txs[4] = {xa.x, xa.y, xa.z, xa.w};
                                    const float tys[4] = {ya.x, ya.y, ya.z, ya.w};
                                    const float tzs[4] = {za.x, za.y, za.z, za.w};
#pragma unroll
                                    for (int e = 0; e < 4; ++e) {
                                        const float d = sqdist(sx, sy, sz, txs[e], tys[e], tzs[e]);
                                        const bool lt = d < lb;
                                        lsec = lt ? lb : min_nonneg(lsec, d);
                                        lslot = lt ? t0 + 4 * h + e : lslot;
                                        lb = lt ? d : lb;
                                    }
                                }
                            }
                            rowbest = row8_min_nonneg(lb);
                            const float kLo = a > 0 ? tkey(a) : -kInf;
                            const float kHi = (bEnd < np16 && bEnd <= yc.n) ? tkey(bEnd - 1) : kInf;   // (beyond the last target: nothing is left on that side)
                            const float rL = qa - kLo, rR = kHi - qa;
                            rho = fminf(rL, rR);
                            rho = (rho - sort_key_slack(axis, sx, sy, rho)) * 0.9999f - 1e-6f;
                            const bool nn = rho > 0.f && rho * rho > rowbest * 1.000003f;
                            const bool out = !(rowbest <= p.thr2) && rho > gateOut;
                            if (!done) {
                                if (nn || out) { done = true; ok = true; isNN = nn; }
                                else if (step + 1 == probeSteps) done = true;
                                else if (rL < rR) { be = a; bs = max(a - 64, 0); a = bs; }   // (a > 0: rL is finite)
                                else { bs = bEnd; be = bEnd + 64; bEnd = be; }               // (bEnd < np16: rR is finite)
                            }
                            if (__ballot(!done) == 0ull) break;
                        }
                        // the row's winner: exactly one lane may hold the minimum, and only once
                        const bool isW = rowOn && lb == rowbest;
                        const unsigned long long wm = __ballot(isW);
                        const int cnt = __popc((unsigned)(wm >> (lane & ~7)) & 0xffu);
                        const float secv = row8_min_nonneg(isW ? lsec : lb);
                        const int slot = row8_min(isW ? lslot : 0x7fffffff);
                        ok = ok && cnt == 1 && secv > rowbest;
#ifdef ICPFLOW_CERT_STATS
                        if (li == 0 && rowOn && it < 128 && (g_stats_block < 0 || g_stats_block == (int)blockIdx.x)) { atomicAdd(&g_probe_stats[it * 2], 1ull); atomicAdd(&g_probe_stats[it * 2 + 1], ok ? 1ull : 0ull); }
#endif
                        // the answer: (certified squared distance or inf, bound on every other target, the neighbour's slot or -1)
                        if (rowOn && li == 0) {
                            probeR[0][ent] = isNN ? rowbest : kInf;
                            probeR[1][ent] = fminf(__builtin_amdgcn_sqrtf(secv), rho);
                            probeR[2][ent] = __int_as_float(ok ? slot : -1);
                        }
                    }
                    probePar ^= 1;
                    barrier_lds_only();
                    if (myEntry >= 0) {
                        const int rSlot = __float_as_int(probeR[2][myEntry]);
                        if (rSlot >= 0) { recM[0] = -1.f; certD[0] = probeR[0][myEntry]; certJ[0] = rSlot; newL[0] = probeR[1][myEntry]; }
                    }
                    scanNow = !mustScan;   // (an inconclusive probe, or a full queue, leaves its query to the scan)
                }
                } else {
                if (REC && recOn && it > itFirst) {
                    const bool longProbes = ngr > 1 || yc.n > 1024;   // (team members: one pass each, but of a long cloud)
                    // (and every lane of the wave where the alternative is a scan of a long window of a long cloud: this wave's window
                    // of the previous search held more than 512 targets -- a dense 10000-point cluster; on the ragged real-shape
                    // batch the ICP launch 1.33 -> 0.94 ms; the demo frame's wall, long but thin, keeps its short windows and 32)
                    const bool wideWindows = yc.n > 4096 && winHi - winLo > kWideWindow;
                    const int probeSteps = longProbes ? kProbeStepsLong : kProbeSteps;
                    const int probeMax = longProbes ? (wideWindows ? kWave : kProbeMaxLong) : kProbeMax;
#pragma unroll
                    for (int q = 0; q < Q; ++q) {
                        const bool wants = live[q] && recM[q] >= 0.f && certJ[q] >= 0;
                        const unsigned long long need = __ballot(wants);
                        const int nNeed = __popcll(need);
                        if (nNeed > probeMax || nNeed == 0) continue;
                        // the uncertified lanes in lane order: lane k of the wave learns who the k-th of them is (one
                        // forward permute), row r of round t then serves the (8 t + r)-th
                        const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
                        const int kth = __builtin_amdgcn_ds_permute((wants ? rank : nNeed + lane - rank) << 2, lane);   // (a permutation)
                        const int row = lane >> 3, li = lane & 7;
                        for (int t = 0; t * 8 < nNeed; ++t) {
                            const int which = 8 * t + row;
                            const bool rowOn = which < nNeed;
                            const int kthOfRow = __shfl(kth, which & 63, kWave);   // (every lane takes part: the source lanes too)
                            const int srcl = rowOn ? kthOfRow : lane;
                            const float sx = __shfl(qx[q], srcl, kWave), sy = __shfl(qy[q], srcl, kWave),
                                        sz = __shfl(qz[q], srcl, kWave);
                            const int j1 = __shfl(certJ[q], srcl, kWave);
                            const float qa = pkey(sx, sy, sz);
                            int a = min(max((j1 & ~7) - 32, 0), max(np16 - 64, 0));   // evaluated so far: [a, bEnd)
                            int bEnd = a + 64;
                            int bs = a, be = bEnd;                                    // this step's block
                            float lb = kInf, lsec = kInf;                             // this lane's minimum and runner-up
                            int lslot = 0;
                            bool done = !rowOn, ok = false, isNN = false;
                            float rowbest = kInf, rho = 0.f;
                            for (int step = 0; step < probeSteps; ++step) {
                                const int t0 = bs + 8 * li;
                                if (!done && t0 < be && t0 < np16) {
                                    const float4 xa = *reinterpret_cast<const float4 *>(lx + t0), xb = *reinterpret_cast<const float4 *>(lx + t0 + 4);
                                    const float4 ya = *reinterpret_cast<const float4 *>(ly + t0), yb = *reinterpret_cast<const float4 *>(ly + t0 + 4);
                                    const float4 za = *reinterpret_cast<const float4 *>(lz + t0), zb = *reinterpret_cast<const float4 *>(lz + t0 + 4);
                                    const float txs[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
                                    const float tys[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
                                    const float tzs[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
                                    for (int e = 0; e < 8; ++e) {
                                        const float d = sqdist(sx, sy, sz, txs[e], tys[e], tzs[e]);
                                        const bool lt = d < lb;
                                        lsec = lt ? lb : min_nonneg(lsec, d);
                                        lslot = lt ? t0 + e : lslot;
                                        lb = lt ? d : lb;
                                    }
                                }
                                rowbest = row8_min_nonneg(lb);
                                const float kLo = a > 0 ? tkey(a) : -kInf;
                                const float kHi = (bEnd < np16 && bEnd <= yc.n) ? tkey(bEnd - 1) : kInf;   // (beyond the last target: nothing is left on that side)
                                const float rL = qa - kLo, rR = kHi - qa;
                                rho = fminf(rL, rR);
                                rho = (rho - sort_key_slack(axis, sx, sy, rho)) * 0.9999f - 1e-6f;
                                const bool nn = rho > 0.f && rho * rho > rowbest * 1.000003f;
                                const bool out = !(rowbest <= p.thr2) && rho > gateOut;
                                if (!done) {
                                    if (nn || out) { done = true; ok = true; isNN = nn; }
                                    else if (step + 1 == probeSteps) done = true;
                                    else if (rL < rR) { be = a; bs = max(a - 64, 0); a = bs; }   // (a > 0: rL is finite)
                                    else { bs = bEnd; be = bEnd + 64; bEnd = be; }               // (bEnd < np16: rR is finite)
                                }
                                if (__ballot(!done) == 0ull) break;
                            }
                            // the row's winner: exactly one lane may hold the minimum, and only once
                            const bool isW = rowOn && lb == rowbest;
                            const unsigned long long wm = __ballot(isW);
                            const int cnt = __popc((unsigned)(wm >> (lane & ~7)) & 0xffu);
                            const float secv = row8_min_nonneg(isW ? lsec : lb);
                            const int slot = row8_min(isW ? lslot : 0x7fffffff);
                            ok = ok && cnt == 1 && secv > rowbest;
#ifdef ICPFLOW_CERT_STATS
                            if (li == 0 && rowOn && it < 128 && (g_stats_block < 0 || g_stats_block == (int)blockIdx.x)) { atomicAdd(&g_probe_stats[it * 2], 1ull); atomicAdd(&g_probe_stats[it * 2 + 1], ok ? 1ull : 0ull); }
#endif
                            // hand the result to the lane that owns the query
                            const bool mine = wants && (rank >> 3) == t;
                            const int from = (rank & 7) * 8;
                            const float rD = __shfl(isNN ? rowbest : kInf, from, kWave);
                            const float rNewL = __shfl(fminf(__builtin_amdgcn_sqrtf(secv), rho), from, kWave);
                            const int rSlot = __shfl(ok ? slot : -1, from, kWave);
                            if (mine && rSlot >= 0) { recM[q] = -1.f; certD[q] = rD; certJ[q] = rSlot; newL[q] = rNewL; }
                        }
                    }
                }
                }
                if (scanNow) scan_window(true);
#ifdef ICPFLOW_TAIL_CLOCK
                if (b == g_unit_pair && it < 64 && g < 8 && wave < 16) {
                    int nsearch = 0;
#pragma unroll
                    for (int q = 0; q < Q; ++q) nsearch += __popcll(__ballot(live[q] && recM[q] >= 0.f));
                    if (lane == 0) { g_unit_win[((it * 8 + g) * 16 + wave) * 2] = ce - cb; g_unit_win[((it * 8 + g) * 16 + wave) * 2 + 1] = nsearch; }
                }
#endif
#ifdef ICPFLOW_CERT_STATS
                if (it < 128 && (g_stats_block < 0 || g_stats_block == (int)blockIdx.x)) {
                    int nsearch = 0;
#pragma unroll
                    for (int q = 0; q < Q; ++q) nsearch += __popcll(__ballot(live[q] && recM[q] >= 0.f));
                    if (lane == 0) {
                        atomicAdd(&g_cert_stats[it * 4 + 0], 1ull);
                        atomicAdd(&g_cert_stats[it * 4 + 1], lo <= hi ? 1ull : 0ull);
                        atomicAdd(&g_cert_stats[it * 4 + 2], (unsigned long long)nsearch);
                        atomicAdd(&g_cert_stats[it * 4 + 3], (unsigned long long)(ce - cb));
                    }
                }
#endif
                ICPFLOW_STAMP(2);
                // neighbour = the target at distance `best` (bit-equal re-evaluation of the winning
                // chunk).  If several targets tie -- in that chunk or, flagged by the scan, in another
                // one -- the lowest ORIGINAL index wins: only then are the original indices fetched
                // (w component of the sorted array).
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                double ax = 0.0, ay = 0.0, az = 0.0, bx = 0.0, by = 0.0, bz = 0.0, wq = 0.0;
                int nnSlot = -1;   // the gated neighbour (slot of the sorted image)
                const bool searched = live[q] && recM[q] >= 0.f;
                if (!searched) { acc.best[q] = certD[q]; tie[q] = false; }
                // with certificates every searched query with a target in its window is resolved (the record wants the
                // neighbour and the runner-up), otherwise only the gated ones
                if (live[q] && (acc.best[q] <= p.thr2 || (recOn && searched && acc.best[q] < kInf))) {  // :160-161
                    float ynx = 0.f, yny = 0.f, ynz = 0.f;
                    int slot = certJ[q];
                    if (!searched) {   // certified or probed
                        ynx = lx[slot]; yny = ly[slot]; ynz = lz[slot];
                    } else {
                    int matches = tie[q] ? 2 : 0;
                    float sec = second[q];
                    if (!tie[q]) {
                        const int c0 = acc.chunk[q];
#pragma unroll
                        for (int u = 0; u < kChunk / 4; ++u) {
                            const float4 tx = *reinterpret_cast<const float4 *>((GRID == 4 ? lx : gx) + c0 + 4 * u);
                            const float4 ty = *reinterpret_cast<const float4 *>((GRID == 4 ? ly : gy) + c0 + 4 * u);
                            const float4 tz = *reinterpret_cast<const float4 *>((GRID == 4 ? lz : gz) + c0 + 4 * u);
                            const float txs[4] = {tx.x, tx.y, tx.z, tx.w};
                            const float tys[4] = {ty.x, ty.y, ty.z, ty.w};
                            const float tzs[4] = {tz.x, tz.y, tz.z, tz.w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float d = sqdist(qx[q], qy[q], qz[q], txs[e], tys[e], tzs[e]);
                                if (d == acc.best[q]) { ++matches; ynx = txs[e]; yny = tys[e]; ynz = tzs[e]; slot = c0 + 4 * u + e; }
                                else if (REC) sec = min_nonneg(sec, d);
                            }
                        }
                    }
                    if (matches != 1) {
                        const int r0 = tie[q] ? cb : acc.chunk[q];
                        const int r1 = tie[q] ? ce : acc.chunk[q] + kChunk;
                        int bj = 0x7fffffff;
                        for (int k = r0; k < min(r1, yc.n); ++k) {
                            const float4 t = ys[k];
                            const float d = sqdist(qx[q], qy[q], qz[q], t.x, t.y, t.z);
                            const int j = __float_as_int(t.w);
                            if (d == acc.best[q] && j < bj) { bj = j; ynx = t.x; yny = t.y; ynz = t.z; slot = k; }
                        }
                        sec = acc.best[q];   // a second target at the same distance
                    }
                    second[q] = sec;
                    certJ[q] = slot;
                    }
                    if (acc.best[q] <= p.thr2) {
                        wq = 1.0;
                        nnSlot = slot;
                        ax = (double)(x0x[q] - ox); ay = (double)(x0y[q] - oy); az = (double)(x0z[q] - oz);
                        bx = (double)(ynx - ox); by = (double)(yny - oy); bz = (double)(ynz - oz);
                    }
                }
                if (recOn && searched) {
                    // every target other than the neighbour: at least the runner-up of the window away, and targets
                    // outside the window differ by more than this query's half-window along the sort axis (the window
                    // bounds qa -+ m are rounded: 4 ulp of head-room, and 0.1 % on the half-width)
                    const float qa = pkey(qx[q], qy[q], qz[q]);
                    const float cap = recM[q] * 0.999f - fabsf(qa) * 2.4e-7f - sort_key_slack(axis, qx[q], qy[q], recM[q]);
                    newL[q] = fmaxf(fminf(__builtin_amdgcn_sqrtf(second[q]), cap), 0.f);
                    if (!(acc.best[q] < kInf)) certJ[q] = -1;
                }
                if (recOn && live[q]) {
                    const int li = g * PER + (unitWave * Q + q) * kWave + lane;
                    if (newL[q] >= 0.f) rec[li] = make_float4(qx[q], qy[q], qz[q], newL[q]);
                    // the neighbour word follows the neighbour AND the gate decision of this iteration (see the unit's reuse below)
                    const int newWord = certJ[q] < 0 ? -1 : (certJ[q] | (nnSlot >= 0 ? 0x40000000 : 0));
                    if (newL[q] >= 0.f || newWord != prevWord[q]) recJ[li] = newWord;
                }
                ICPFLOW_STAMP(10);
#ifdef ICPFLOW_TAIL_CLOCK
                if (b == g_unit_pair && lane == 0 && it < 64 && g < 8 && wave < 16) g_unit_clk[(it * 8 + g) * 16 + wave] = clock64() - unit0;
#endif
                // The moments are sums over (x0, gated neighbour) only -- the pose enters through WHICH neighbour is
                // gated.  A wave none of whose queries changed its gated neighbour since the previous iteration would
                // add up the same numbers in the same order: its 18 sums are still in `red` (single-pass clouds).
                if constexpr (REC && Q == 1) {
                    if (recOn && ngr == 1) {
                        reuseMoments = it > itFirst && __ballot(nnSlot != prevNN) == 0ull;
                        prevNN = nnSlot;
                    }
#ifndef ICPFLOW_NO_UNIT_REUSE
                    else if (recOn && perPass) {
                        // (round 6) clouds of several passes whose sums are kept per (pass, wave) -- i.e. per UNIT, in dynamic LDS,
                        // from iteration to iteration: the same reuse, unit by unit.  What a query's gated neighbour was when
                        // the unit's sums were last formed rides in its record's neighbour word (slot | gated << 30; recJ above);
                        // a pass that a helper computed in the previous iteration holds the HELPER's sums: no reuse there.
                        const int nnPrev = (prevWord[0] >= 0 && (prevWord[0] & 0x40000000) != 0) ? (prevWord[0] & 0x3fffffff) : -1;
                        const bool differs = live[0] && (prevWord[0] == -2 || nnSlot != nnPrev);
                        reuseMoments = it > itFirst && ((ownedPrev >> g) & 1) != 0 && __ballot(differs) == 0ull;
                    }
#endif
                }
                // 18 moments -> 5 registers by two folding levels (see common.hpp): fold[j] holds, per
                // row of 16 lanes, partial sums of moments (4j, 4j+2, 4j+1, 4j+3); fold[4]: 16,16,17,17
                if (!reuseMoments) {
                    const double a0 = swap32_sum(wq, ax), a1 = swap32_sum(ay, az);
                    fold[0] += swap16_sum(a0, a1);
                    const double a2 = swap32_sum(bx, by), a3 = swap32_sum(bz, ax * bx);
                    fold[1] += swap16_sum(a2, a3);
                    const double a4 = swap32_sum(ax * by, ax * bz), a5 = swap32_sum(ay * bx, ay * by);
                    fold[2] += swap16_sum(a4, a5);
                    const double a6 = swap32_sum(ay * bz, az * bx), a7 = swap32_sum(az * by, az * bz);
                    fold[3] += swap16_sum(a6, a7);
                    const double a8 = swap32_sum(ax * ax + ay * ay + az * az, bx * bx + by * by + bz * bz);
                    fold[4] += swap16_sum(a8, a8);
                }
                            }
                // Sums per (pass, wave): the 18 moments of THIS pass go to their own slot and the lane sums start over.
                // The pair's totals are then added in (pass, wave) order whoever computed a pass (icp_kernel's helpers
                // take whole passes of a straggling pair): the order belongs to the pair, not to the launch.
                if (perPass && !reuseMoments) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const double r = row_sum_f64(fold[j]);
                        const int row = lane >> 4;
                        const int m = (j < 4) ? 4 * j + ((row & 1) * 2 + (row >> 1)) : 16 + (row >> 1);
                        if ((lane & 15) == 15) redDyn[(g * NWAVE + wave) * kMoments + m] = r;
                        fold[j] = 0.0;
                    }
                }
            }
            if constexpr (HELP) {
                // owner: wave 1 + j waits for helper j's pass of THIS iteration and copies its NWAVE x 18 sums into the
                // pass's slot (the waves are through with their own passes; wave 0's serial part starts behind the
                // barrier below either way)
                if (!helping && helpedPasses != 0 && wave >= 1 && wave <= kHelpSlots && wave < NWAVE) {
                    const int gj = ngr - wave;               // slot j = wave - 1 takes pass ngr - 1 - j
                    if (gj >= 1 && ((helpedPasses >> gj) & 1) != 0) {
                        const int wg = __builtin_amdgcn_readfirstlane(helpSh[wave]);
                        const int want = (b << 8) | (it + 1);
                        const long long t0 = wall_clock64();
                        bool ok = true;
                        while (__hip_atomic_load(&p.help.tag[wg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
                            __builtin_amdgcn_s_sleep(1);
                            if (wall_clock64() - t0 > kTeamTimeoutTicks) { ok = false; break; }
                        }
                        asm volatile("" ::: "memory");
#ifdef ICPFLOW_TAIL_CLOCK
                        if (lane == 0) { atomicAdd(&g_help_stats[1], 1ull); atomicAdd(&g_help_stats[2], (unsigned long long)(wall_clock64() - t0)); if (b < 1024) { atomicAdd(&g_pair_help[b], 1ull); atomicAdd(&g_pair_hclk[b * 4 + 3], (unsigned long long)(wall_clock64() - t0)); } }
#endif
                        if (!ok) {
                            if (lane == 0) __hip_atomic_store(&ctrl->error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            const double *out = p.help.out + (size_t)wg * kHelpOutStride;
                            // (all of a lane's loads in flight together: unconditional, from clamped addresses -- as a loop
                            // every record cost a round trip of its own, see team_collect)
                            constexpr int kFetch = (NWAVE * kMoments + kWave - 1) / kWave;
                            double got[kFetch];
#pragma unroll
                            for (int u = 0; u < kFetch; ++u)
                                got[u] = __hip_atomic_load(&out[min(lane + u * kWave, NWAVE * kMoments - 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                            for (int u = 0; u < kFetch; ++u)
                                if (lane + u * kWave < NWAVE * kMoments) redDyn[gj * NWAVE * kMoments + lane + u * kWave] = got[u];
                        }
                    }
                }
            }
            if constexpr (HELP) ownedPrev = helping ? (1 << (ngr - role)) : ~helpedPasses;
            else ownedPrev = -1;
            // rows of 16 lanes -> lane 15 of each row holds the wave total of "its" moment
            if (!perPass && !reuseMoments)
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const double r = row_sum_f64(fold[j]);
                const int row = lane >> 4;
                const int m = (j < 4) ? 4 * j + ((row & 1) * 2 + (row >> 1)) : 16 + (row >> 1);
                if ((lane & 15) == 15) red[wave * kMoments + m] = r;
            }
            help_phase();   // (teams: the windows posted behind the pass's last barrier; the block barrier follows below)
        } else if constexpr (GRID != 0) {
            const float4 *gpG = p.gridPts + (size_t)b * p.N;
            const int32_t *gsG = p.gridStart + (size_t)b * (p.gridH + 1);
            // LDS copies (GRID == 2): staged once per launch, before the first iteration
            int32_t *gsL = reinterpret_cast<int32_t *>(dynLds);
            float4 *gpL = reinterpret_cast<float4 *>(dynLds + (((size_t)p.gridH + 1) * 4 + 15) / 16 * 16);
            if (GRID == 2 && it == itBegin) {
                for (int k = tid; k <= p.gridH; k += BLOCK) gsL[k] = gsG[k];
                for (int k = tid; k < yc.n; k += BLOCK) gpL[k] = gpG[k];
                __syncthreads();
            }
            const float gox = p.gridOrigin[b * 4 + 0], goy = p.gridOrigin[b * 4 + 1], goz = p.gridOrigin[b * 4 + 2];
            const unsigned mask = (unsigned)p.gridH - 1u;
            const int ngr = (xc.n + BLOCK - 1) / BLOCK;
            for (int g = 0; g < ngr; ++g) {
                const int i = g * BLOCK + tid;
                float x0x = 0.f, x0y = 0.f, x0z = 0.f, qx = 0.f, qy = 0.f, qz = 0.f;
                float bd = kInf, ynx = 0.f, yny = 0.f, ynz = 0.f;
                int bj = 0x7fffffff;
                ICPFLOW_STAMP(1);
                if (i < xc.n && yc.n > 0) {
                    float rx, ry, rz;
                    cloud_load(xc, i, rx, ry, rz);
                    xf_apply(pre, rx, ry, rz, x0x, x0y, x0z);  // utils_icp.py:21
                    qx = fmaf(x0z, Rf[6], fmaf(x0y, Rf[3], x0x * Rf[0])) + Tf[0];  // :177, :395
                    qy = fmaf(x0z, Rf[7], fmaf(x0y, Rf[4], x0x * Rf[1])) + Tf[1];
                    qz = fmaf(x0z, Rf[8], fmaf(x0y, Rf[5], x0x * Rf[2])) + Tf[2];
                    const int cx = grid_cell(qx, gox, p.gridInvH), cy = grid_cell(qy, goy, p.gridInvH),
                              cz = grid_cell(qz, goz, p.gridInvH);
                    // One z-plane of 9 cells at a time: first all nine (start, count) look-ups, then
                    // rounds r = 0, 1, ... in which every non-exhausted cell contributes its r-th
                    // point.  All loads of a round are unconditional (an exhausted cell re-reads
                    // slot 0 and its distance is discarded), so they are in flight together instead
                    // of forming a chain of dependent LDS/L2 round trips.
                    for (int dz = -1; dz <= 1; ++dz) {
                        int s9[9], c9[9], maxc = 0;
#pragma unroll
                        for (int u = 0; u < 9; ++u) {
                            const unsigned h = grid_hash(cx + (u % 3) - 1, cy + (u / 3) - 1, cz + dz, mask);
                            const int sb = (GRID == 2) ? gsL[h] : gsG[h];
                            const int eb = (GRID == 2) ? gsL[h + 1] : gsG[h + 1];
                            s9[u] = sb;
                            c9[u] = eb - sb;
                            maxc = max(maxc, eb - sb);
                        }
                        for (int r = 0; r < maxc; ++r) {
                            float4 t9[9];
#pragma unroll
                            for (int u = 0; u < 9; ++u) {
                                const int k = (r < c9[u]) ? s9[u] + r : 0;
                                t9[u] = (GRID == 2) ? gpL[k] : gpG[k];
                            }
#pragma unroll
                            for (int u = 0; u < 9; ++u) {
                                const float d = (r < c9[u]) ? sqdist(qx, qy, qz, t9[u].x, t9[u].y, t9[u].z) : kInf;
                                const int j = __float_as_int(t9[u].w);
                                if (d < bd || (d == bd && j < bj && d < kInf)) {
                                    bd = d; bj = j; ynx = t9[u].x; yny = t9[u].y; ynz = t9[u].z;
                                }
                            }
                        }
                    }
                }
                ICPFLOW_STAMP(2);
                const bool inl = (i < xc.n) && (bd <= p.thr2);  // :160-161
                double ax = 0.0, ay = 0.0, az = 0.0, bx = 0.0, by = 0.0, bz = 0.0;
                if (inl) {
                    ax = (double)(x0x - ox); ay = (double)(x0y - oy); az = (double)(x0z - oz);
                    bx = (double)(ynx - ox); by = (double)(yny - oy); bz = (double)(ynz - oz);
                }
                ICPFLOW_STAMP(10);
                ICPFLOW_ACC_ALL()
            }
        } else {
        for (int g = 0; g < ngroups; ++g) {
            ScanAcc<Q> acc;
            float qx[Q], qy[Q], qz[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int i = g * per + (q * NQG + qg) * kWave + lane;
                qx[q] = qy[q] = qz[q] = 0.f;
                if (i < xc.n) {
                    float rx, ry, rz, x0x, x0y, x0z;
                    cloud_load(xc, i, rx, ry, rz);
                    xf_apply(pre, rx, ry, rz, x0x, x0y, x0z);  // utils_icp.py:21
                    // Xt = X0 R + T  (:177, :395), bmm order
                    qx[q] = fmaf(x0z, Rf[6], fmaf(x0y, Rf[3], x0x * Rf[0])) + Tf[0];
                    qy[q] = fmaf(x0z, Rf[7], fmaf(x0y, Rf[4], x0x * Rf[1])) + Tf[1];
                    qz[q] = fmaf(x0z, Rf[8], fmaf(x0y, Rf[5], x0x * Rf[2])) + Tf[2];
                }
            }
            ICPFLOW_STAMP(1);
            scan_cloud<Q>(yc, none, tile, qx, qy, qz, acc, ts, TS);  // :154-157
            ICPFLOW_STAMP(2);
            if (TS > 1) {
                __syncthreads();  // combD/combC of the previous pass fully consumed
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    combD[(wave * Q + q) * kWave + lane] = acc.best[q];
                    combC[(wave * Q + q) * kWave + lane] = acc.chunk[q];
                }
                __syncthreads();
            }
            ICPFLOW_STAMP(9);
            // finish the query slots owned by this wave (q % TS == ts)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (TS > 1 && (q % TS) != ts) continue;  // wave-uniform
                float bd = acc.best[q];
                int bc = acc.chunk[q];
                if (TS > 1) {
#pragma unroll
                    for (int t = 0; t < TS; ++t) {
                        const float d = combD[((qg * TS + t) * Q + q) * kWave + lane];
                        const int c = combC[((qg * TS + t) * Q + q) * kWave + lane];
                        if (t == 0 || scan_better(bd, bc, d, c)) { bd = d; bc = c; }
                    }
                }
                const int i = g * per + (q * NQG + qg) * kWave + lane;
                const bool inl = (i < xc.n) && (bd <= p.thr2);  // :160-161
                double ax = 0.0, ay = 0.0, az = 0.0, bx = 0.0, by = 0.0, bz = 0.0;
                if (inl) {
                    float ynx, yny, ynz, rx, ry, rz, x0x, x0y, x0z;
                    scan_resolve(yc, none, qx[q], qy[q], qz[q], bd, bc, ynx, yny, ynz);
                    cloud_load(xc, i, rx, ry, rz);
                    xf_apply(pre, rx, ry, rz, x0x, x0y, x0z);
                    ax = (double)(x0x - ox); ay = (double)(x0y - oy); az = (double)(x0z - oz);
                    bx = (double)(ynx - ox); by = (double)(yny - oy); bz = (double)(ynz - oz);
                }
                ICPFLOW_STAMP(10);
                ICPFLOW_ACC_ALL()
            }
        }
        }
#undef ICPFLOW_ACC_ALL
#undef ICPFLOW_ACC
        if constexpr (GRID < 3) {
            if (lane < kMoments) {  // lane k publishes moment k of this wave
                double mine = 0.0;
#pragma unroll
                for (int k = 0; k < kMoments; ++k) mine = (lane == k) ? macc[k] : mine;
                red[wave * kMoments + lane] = mine;
            }
        }
        ICPFLOW_STAMP(3);
        __syncthreads();  // every wave's row is complete
        ICPFLOW_STAMP(4);
#ifdef ICPFLOW_TAIL_CLOCK
        const long long tcTail0 = clock64();
        tcSearch += tcTail0 - tcLoop0;
#endif
        if constexpr (HELP) {
            if (helping) {
                // the pass's sums (one row of 18 per wave) go to this workgroup's outbox write-through, every wave drains
                // its own stores, then ONE lane tags the outbox with (pair, iteration): what the owner polls for
                {
                    const int passes = (xc.n + BLOCK - 1) / BLOCK, gj = passes - role;
                    double *out = p.help.out + (size_t)blockIdx.x * kHelpOutStride;
                    const double *src = reinterpret_cast<const double *>(dynLds) + (size_t)(gj * NWAVE + wave) * kMoments;
                    if (lane < kMoments)
                        __hip_atomic_store(&out[wave * kMoments + lane], src[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __syncthreads();
                if (tid == 0) {
#ifdef ICPFLOW_TAIL_CLOCK
                    atomicAdd(&g_help_stats[4], (unsigned long long)(clock64() - tcLoop0));   // the helper's pass, shader clocks
                    atomicAdd(&g_help_stats[5], 1ull);
                    if (b < 1024) { atomicAdd(&g_pair_hclk[b * 4], (unsigned long long)(clock64() - tcLoop0)); atomicAdd(&g_pair_hclk[b * 4 + 1], 1ull); }
                    const long long tw0 = wall_clock64();
#endif
                    __hip_atomic_store(&p.help.tag[blockIdx.x], (b << 8) | (it + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    // the state of a LATER iteration (the owner publishes one per iteration while a helper is signed up;
                    // it cannot get further than one iteration past a pass it is waiting for)
                    int e;
                    const long long t0 = wall_clock64();
                    for (;;) {
                        e = __hip_atomic_load(&hp->epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (e < 0 || e > it + 1) break;
                        __builtin_amdgcn_s_sleep(2);
                        if (wall_clock64() - t0 > kTeamTimeoutTicks) { e = -1; break; }
                    }
                    helpSh[4] = e;
#ifdef ICPFLOW_TAIL_CLOCK
                    atomicAdd(&g_help_stats[6], (unsigned long long)(wall_clock64() - tw0));       // waiting for the next state, 100 MHz ticks
                    if (b < 1024) atomicAdd(&g_pair_hclk[b * 4 + 2], (unsigned long long)(wall_clock64() - tw0));
#endif
                }
                __syncthreads();
                const int e = __builtin_amdgcn_readfirstlane(helpSh[4]);
                if (e <= 0 || e > itEnd) break;          // the pair is finished
                if (tid < 12) bcast[tid] = __hip_atomic_load(&p.help.state[((size_t)b * 2 + (e & 1)) * 16 + tid], __ATOMIC_RELAXED,
                                                             __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                // (had the owner run two epochs ahead while this was being read -- it only does when it is not waiting
                // for this helper -- the buffer may have been rewritten: the pass is then computed from garbage and
                // nobody reads it; the next epoch is read afresh)
                it = e - 2;                                // ++it -> iteration e - 1
#ifdef ICPFLOW_TAIL_CLOCK
                tcLoop0 = clock64();
#endif
                continue;
            }
        }
        // the pair's flag, written by wave 0 with the previous iteration's bookkeeping (before the barrier above)
        if (kLateBook && it > itFirst && bcast[12] == 0.f) {
            active = 0;
            // speculative mode: only member 0 watches the batch tally; it tells its team to stop
            // through the record of the iteration the others are about to exchange
            if (TEAM && G > 1 && rank == 0 && p.stopMode == ICPFLOW_STOP_REFERENCE_ && wave == 0)
                team_publish(team, b, it, 0, 0.0, 1.0, lane);
            break;
        }
        // ------------- wave 0 solves for (R, T, rmse) ---------------------------------------
        bool solved = false;      // (late bookkeeping) wave 0 has published a new state in this iteration
        [[maybe_unused]] int helpWord = 0;
        // The bookkeeping of an iteration: history record, tally, batch-rule check, cycle detection, the flag (and the
        // helpers' hand-off words).  Called by wave 0 in front of the barrier that publishes (R, T) with the values in registers, or
        // (kLateBook) behind it with the values read back from the published copy.
        auto bookkeeping = [&](const float (&Rn)[9], const float (&Tn)[3], const float sn, const float rmse, const float prev) {
            // relative rmse, :195-198 (fp32 like the reference's tensors)
            const float rel = (it == 0) ? 1.0f : (prev - rmse) / prev;
            const bool conv = rel <= p.relThr;  // NaN -> false, :209
            if (p.stopMode == ICPFLOW_STOP_REFERENCE_ && p.history != nullptr) {
                // speculative mode: record this iteration, publish (arrived, not converged) with one
                // atomic, and leave once SOME iteration s <= it is known to satisfy the batch rule
                // (every pair arrived at s, none unconverged).  Nobody ever waits.
                if (conv && it < 128 && lane == 0) ownConvSh[it >> 5] |= 1u << (it & 31);
                if (lane == 0 && rank == 0) {
                    float *h = p.history + ((size_t)it * p.B + b) * kHistStride;
#pragma unroll
                    for (int k = 0; k < 9; ++k) h[k] = Rn[k];
                    h[9] = Tn[0]; h[10] = Tn[1]; h[11] = Tn[2]; h[12] = rmse; h[13] = sn;
                    h[14] = (float)tot[0];   // gated correspondences of this iteration (sum w, :161; exact below 2^24)
                    __hip_atomic_fetch_add(&ctrl->tally[it], 1ull | (conv ? 0ull : (1ull << 32)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
                // specTally (fetched at the top of this iteration) describes iteration specChk <= it - 1
                if (specLoaded) {
                    if ((int)(specTally & 0xffffffffull) >= p.B) {   // everybody has been there
                        if ((specTally >> 32) == 0ull) active = 0;   // the batch stops at specChk
                        else ++specChk;
                    }
                }
                ICPFLOW_STAMP(9);
                // Periodic trajectory: the next state is a function of (R, T) alone, so once the new
                // state (number it + 1) equals, bit for bit, one of the last kRing states, everything
                // that follows repeats with that period (1 = the usual convergence by exact repetition,
                // 2 = a pair flipping between two inlier sets, which never satisfies the stop test and
                // would hold the whole batch at the iteration cap).  The pair then writes its history
                // and its tallies for ALL remaining iterations from the cycle and leaves: the outcome of
                // the batch rule is unchanged, the iterations are not executed.
                int period = 0;
                {
                    const int newest = it + 1;   // number of the new state
                    // oldest state that may be compared with: everything since the launch's first state, or (a resumed pair, RESUME
                    // above) the kRing states restored from the history
                    const int ringOldest = (RESUME && p.history != nullptr && itBegin > 0) ? (itBegin < kRing ? 0 : itBegin - (kRing - 1)) : itBegin;
                    // four candidate periods per round: quarter q of the wave compares the new state
                    // (replicated into every quarter) with state newest - (k0 + q)
                    // cheap first: a 32-bit hash of the state (xor of its twelve words) against the hashes of
                    // the remembered states, all eight at once; the word-by-word comparison only runs on a hit
                    int hash = 0;
#pragma unroll
                    for (int k = 0; k < 9; ++k) hash ^= state_hash_word(Rn[k], k);
#pragma unroll
                    for (int k = 0; k < 3; ++k) hash ^= state_hash_word(Tn[k], 9 + k);
                    hash ^= state_hash_word(sn, 12);
                    const int kk = lane + 1;   // lane l < kRing looks at state newest - (l + 1)
                    const bool cand = lane < kRing && kk <= newest - ringOldest &&
                                      __float_as_int(ring[((newest - kk) % kRing) * 16 + 13]) == hash;
                    const bool anyCand = __ballot(cand) != 0ull;
                    float cur16 = 0.f;   // word (lane & 15) of the new state: only a hash hit needs it
                    if (anyCand) {
                        const int wd = lane & 15;
#pragma unroll
                        for (int k = 0; k < 9; ++k) cur16 = (wd == k) ? Rn[k] : cur16;
                        cur16 = (wd == 9) ? Tn[0] : (wd == 10) ? Tn[1] : (wd == 11) ? Tn[2] : (wd == 14) ? sn : cur16;
                    }
                    for (int k0 = 1; anyCand && k0 <= kRing && period == 0; k0 += 4) {
                        const int k = k0 + (lane >> 4);
                        const bool valid = k <= kRing && k <= newest - ringOldest;
                        const float old = ring[(((newest - (valid ? k : 0)) % kRing + kRing) % kRing) * 16 + (lane & 15)];
                        const bool same = ((lane & 15) >= 12 && (lane & 15) != 14) || __float_as_int(old) == __float_as_int(cur16);
                        const unsigned long long m = __ballot(same);
#pragma unroll
                        for (int q = 3; q >= 0; --q) {
                            const bool okq = ((m >> (16 * q)) & 0xffffull) == 0xffffull;
                            const int kq = k0 + q;
                            if (okq && kq <= kRing && kq <= newest - ringOldest) period = kq;   // smallest period wins
                        }
                    }
                    if (lane == 0) {   // (the values are wave-uniform: one lane stores the row)
                        float *rw = ring + (newest % kRing) * 16;
#pragma unroll
                        for (int k = 0; k < 9; ++k) rw[k] = Rn[k];
                        rw[9] = Tn[0]; rw[10] = Tn[1]; rw[11] = Tn[2]; rw[12] = rmse; rw[13] = __int_as_float(hash); rw[14] = sn;
                        rw[15] = (float)tot[0];
                    }
                }
                if (period > 0 && active) {
                    if (rank == 0) {
                        // remaining iterations k = it+1 .. cap-1, one per lane and round; row k repeats
                        // row j(k) = it - period + 1 + ((k - it - 1) mod period)  (a row is stored in the
                        // ring under the number of the state it produced, row + 1)
                        // (to the iteration CAP, not to this launch's itEnd: a pair that turns periodic in the first of two launches is done)
                        for (int k0 = it + 1; k0 < p.maxIter; k0 += kWave) {
                            const int k = k0 + lane;
                            if (k < p.maxIter) {
                                const int j = it - period + 1 + ((k - it - 1) % period);
                                const int jp = (k - 1 == it) ? it : it - period + 1 + ((k - it - 2) % period);
                                const float *rj = ring + ((j + 1) % kRing) * 16;
                                const float rm = rj[12], rmPrev = ring[((jp + 1) % kRing) * 16 + 12];
                                float *h = p.history + ((size_t)k * p.B + b) * kHistStride;
                                for (int c = 0; c < 12; ++c) h[c] = rj[c];
                                h[12] = rm;
                                h[13] = rj[14];
                                h[14] = rj[15];
                                const float relk = (rmPrev - rm) / rmPrev;
                                const bool convk = relk <= p.relThr;
                                __hip_atomic_fetch_add(&ctrl->tally[k], 1ull | (convk ? 0ull : (1ull << 32)), __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
                            }
                        }
                    }
                    active = 0;
                }
            } else if (p.stopMode == ICPFLOW_STOP_REFERENCE_) {
                if (lane == 0 && !conv && rank == 0) atomicAdd(&ctrl->notconv[it], 1);
            } else {
                // per-pair rule: retire a pair once its rmse has stopped DEcreasing by more than
                // thr (0 <= rel <= thr).  A negative rel (rmse went up: the inlier set is still
                // changing) satisfies the reference's batch test but is not convergence of this
                // pair.  A constant (zero-inlier) pair has rel = NaN and is retired too.
                if (it > 0 && ((conv && rel >= 0.0f) || rel != rel)) active = 0;
            }
            // Draining (see icp_split_kernel): at most p.drainAt pairs of the batch are unfinished, a second launch with a whole CU
            // per pair will serve them faster than this one -- the pair leaves behind this iteration, STILL MOVING (IcpState.active
            // stays set, its rows and tallies up to here are in place), and its helpers are told that it is finished.
            bool drain = false;
            if constexpr (HELP) {
                drain = p.drainAt > 0 && active && it > itFirst && it + 1 < itEnd && p.B - finishedSeen <= p.drainAt;
                if (drain && lane == 0) drainSh = 1;
            }
            if (kLateBook) {
                if (lane == 0 && !active) bcast[12] = 0.f;   // seen by every wave behind the next search phase's barrier
            } else if (lane == 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) bcast[k] = Rn[k];
                bcast[9] = Tn[0]; bcast[10] = Tn[1]; bcast[11] = Tn[2];
                bcast[12] = (active && !drain) ? 1.f : 0.f;
                bcast[14] = rmse;  // :213 prev_rmse = rmse
                bcast[15] = sn;
            }
            if constexpr (HELP) {
                if (hp != nullptr) {
                    // progress for workgroups looking for a pair to help; with helpers signed up: the state of iteration
                    // it + 1 (double-buffered by parity, write-through, drained, THEN the epoch), and which of its passes
                    // the helpers that have announced themselves in time will deliver
                    const bool goesOn = active && !drain && it + 1 < itEnd;
                    int mask = 0;
                    const int nclaim = __builtin_amdgcn_readfirstlane(helpWord);
                    if (goesOn && nclaim > 0) {
                        const int e = it + 2;                 // E(it + 1)
                        if (lane < 12) {
                            float v = Tn[0];
#pragma unroll
                            for (int k = 0; k < 9; ++k) v = lane == k ? Rn[k] : v;
                            v = lane == 10 ? Tn[1] : (lane == 11 ? Tn[2] : v);
                            __hip_atomic_store(&p.help.state[((size_t)b * 2 + (e & 1)) * 16 + lane], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (lane == 0) __hip_atomic_store(&hp->epoch, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int passes = (xc.n + BLOCK - 1) / BLOCK;
                        const int f = __shfl(helpWord, (lane + 1) & 63, kWave);     // lane j < 3: slot j's announcement
                        const bool on = lane < kHelpSlots && f != 0 && (f & 0xff) <= e && passes - 1 - lane >= 1;
                        if (on) helpSh[1 + lane] = f >> 8;
                        const unsigned long long m = __ballot(on);
#pragma unroll
                        for (int j = 0; j < kHelpSlots; ++j)
                            if ((m >> j) & 1ull) mask |= 1 << (passes - 1 - j);
                    }
                    if (lane == 0) {
                        helpSh[0] = mask;
                        __hip_atomic_store(&hp->iter, goesOn ? it + 2 : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (!goesOn) __hip_atomic_store(&hp->epoch, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        };
        if (wave == 0) {
            // helpers: who has signed up (lane 0: the count, lanes 1..3: the announcements), fetched HERE, a whole solve
            // before it is looked at -- an agent-scope load is a round trip across the fabric
            if constexpr (HELP) {
                if (hp != nullptr && lane <= kHelpSlots)
                    helpWord = __hip_atomic_load(lane == 0 ? &hp->nclaim : &hp->from[lane - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // lane k < 18 sums moment k over the waves; totals are then wave-uniform via readlane
            double mine = 0.0;
            if (lane < kMoments) {
                // all NWAVE loads in flight at once (left to itself the compiler alternates load, wait, add: NWAVE / 2
                // dependent LDS round trips in the serial tail); the sum itself stays in wave order
                // (sums per (pass, wave) in dynamic LDS: pass after pass, the same way; a single pass is the static case)
                const bool perPassT = GRID == 4 && !TEAM && p.redPasses != 0;
                const int passes = perPassT ? (xc.n + BLOCK * Q - 1) / (BLOCK * Q) : 1;
                const double *src = perPassT ? reinterpret_cast<const double *>(dynLds) : red;
                for (int gp = 0; gp < passes; ++gp) {
                    double r[NWAVE];
#pragma unroll
                    for (int w = 0; w < NWAVE; ++w) r[w] = src[(gp * NWAVE + w) * kMoments + lane];
#pragma unroll
                    for (int w = 0; w < NWAVE; w += 4) {
                        if (w + 3 < NWAVE) asm volatile("" : "+v"(r[w]), "+v"(r[w + 1]), "+v"(r[w + 2]), "+v"(r[w + 3]));
                    }
                    mine = gp == 0 ? r[0] : mine + r[0];
#pragma unroll
                    for (int w = 1; w < NWAVE; ++w) mine += r[w];
                }
            }
            bool teamStop = false;
            if (TEAM && G > 1) {
                // every member publishes its 18 partial moments, waits for the whole team and adds
                // the records in member order: all members get bit-identical totals and solve for
                // the same (R, T) on their own
                team_publish(team, b, it, rank, mine, 0.0, lane);
                teamStop = team_collect(team, ctrl, b, it, itBegin, G, lane, mine);
            }
            if (teamStop) {   // leave (R, T) as they are; every wave of this member stops
                active = 0;
                if (lane == 0) bcast[12] = 0.f;
            } else {
            // totals go through LDS (lane k stores moment k, every lane reads what it needs): the
            // values stay out of the register file while the solve runs
            if (lane < kMoments) tot[lane] = mine;
            const double *mom = tot;
            const double W = mom[0] > 1e-9 ? mom[0] : 1e-9;  // clamp(eps), :314-315, :326
            double h9[9];
            {
                // (reciprocal refined to the last bit or two instead of an IEEE division; explicit fused multiply-adds
                // below: the serial tail is a chain of dependent fp64 operations, each one left out counts)
                double iW = __builtin_amdgcn_rcp(W);
                iW = fma(fma(-W, iW, 1.0), iW, iW);
                iW = fma(fma(-W, iW, 1.0), iW, iW);
                const double mx0 = mom[1] * iW, mx1 = mom[2] * iW, mx2 = mom[3] * iW;
                const double my0 = mom[4] * iW, my1 = mom[5] * iW, my2 = mom[6] * iW;
                const double mxv[3] = {mx0, mx1, mx2}, myv[3] = {my0, my1, my2};
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) h9[i * 3 + j] = fma(-mxv[i], myv[j], mom[7 + i * 3 + j] * iW);  // :318-336
                // park what is needed after the solve in LDS: the Jacobi sweeps want the registers
                // (every lane writes the same value to the same address and reads its own write)
                ksh[0] = mx0; ksh[1] = mx1; ksh[2] = mx2; ksh[3] = my0; ksh[4] = my1; ksh[5] = my2;
                ksh[6] = mom[16] * iW - fma(mx2, mx2, fma(mx1, mx1, mx0 * mx0));   // sum w |x_c|^2 / W
                ksh[7] = mom[17] * iW - fma(my2, my2, fma(my1, my1, my0 * my0));   // sum w |y_c|^2 / W
#pragma unroll
                for (int k = 0; k < 9; ++k) ksh[8 + k] = h9[k];
            }
            ICPFLOW_STAMP(5);
            double Rd[9];
            // allow_reflection (:354-362 with E = I): R = U V^T is the best ORTHOGONAL matrix; for det H < 0 that is the
            // reflection -R', R' the best proper rotation of -H (sum_ij (-R')_ij H_ij = sum_ij R'_ij (-H)_ij)
            bool mirror = false;
            if (p.allowReflection) {
                const double *h = ksh + 8;
                mirror = det3(h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8]) < 0.0;
                if (mirror) {
#pragma unroll
                    for (int k = 0; k < 9; ++k) ksh[8 + k] = -ksh[8 + k];   // (every lane writes the same values)
                }
            }
            double lam = 0.0;   // trace(E S): the scale's numerator (:364-366)
            if (!horn_rotation(ksh + 8, ksh[6] + ksh[7], Nsh, lane, Rd, &lam)) {
                rank1_rotation(ksh + 8, Rd);
                double f2 = 0.0;   // rank <= 1: the one singular value is the Frobenius norm
#pragma unroll
                for (int k = 0; k < 9; ++k) f2 = fma(ksh[8 + k], ksh[8 + k], f2);
                lam = sqrt(f2);
            }
            if (mirror) {
#pragma unroll
                for (int k = 0; k < 9; ++k) { Rd[k] = -Rd[k]; ksh[8 + k] = -ksh[8 + k]; }
            }
            ICPFLOW_STAMP(6);
#ifdef ICPFLOW_DEBUG_SOLVE
            if (b == g_dbg_pair && it == itBegin && lane == 0) {
                for (int k = 0; k < kMoments; ++k) g_dbg_solve[k] = tot[k];
                for (int k = 0; k < 17; ++k) g_dbg_solve[18 + k] = ksh[k];
                g_dbg_solve[35] = lam;
                for (int k = 0; k < 9; ++k) g_dbg_solve[36 + k] = Rd[k];
                g_dbg_solve[45] = (double)bcast[16]; g_dbg_solve[46] = (double)bcast[17]; g_dbg_solve[47] = (double)bcast[18];
            }
#endif
            // T = mu_y - mu_x R with mu = o + m', :376
            const double o0 = (double)bcast[16], o1 = (double)bcast[17], o2 = (double)bcast[18];
            const double mux[3] = {o0 + ksh[0], o1 + ksh[1], o2 + ksh[2]};
            const double muy[3] = {o0 + ksh[3], o1 + ksh[4], o2 + ksh[5]};
            // s = trace(E S) / clamp(Xcov, eps), :364-374 (1 otherwise: every product below is then exact)
            double sd = 1.0;
            double mR0 = fma(mux[2], Rd[6], fma(mux[1], Rd[3], mux[0] * Rd[0]));
            double mR1 = fma(mux[2], Rd[7], fma(mux[1], Rd[4], mux[0] * Rd[1]));
            double mR2 = fma(mux[2], Rd[8], fma(mux[1], Rd[5], mux[0] * Rd[2]));
            if constexpr (SCALE) {
                if (p.estimateScale) {
                    sd = lam / (ksh[6] > 1e-9 ? ksh[6] : 1e-9);
                    mR0 *= sd; mR1 *= sd; mR2 *= sd;
                }
            }
            const double Td0 = muy[0] - mR0, Td1 = muy[1] - mR1, Td2 = muy[2] - mR2;
            // rmse^2 = (Sxx_c + Syy_c)/W - 2 sum_ij R_ij H_ij, :191-192
            double rh = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) rh = fma(Rd[k], ksh[8 + k], rh);
            double ms = ksh[6] + ksh[7] - 2.0 * rh;                        // sum w |s x R + T - y|^2 / W
            if constexpr (SCALE) {
                if (p.estimateScale) ms = sd * sd * ksh[6] + ksh[7] - 2.0 * sd * rh;
            }
            float Rn[9], Tn[3];   // the new state
#pragma unroll
            for (int k = 0; k < 9; ++k) Rn[k] = (float)Rd[k];
            Tn[0] = (float)Td0; Tn[1] = (float)Td1; Tn[2] = (float)Td2;
            const float sn = SCALE ? (float)sd : 1.f;
            const float rmse = (float)sqrt(ms > 0.0 ? ms : 0.0);
            ICPFLOW_STAMP(15);
            const float prevRmse = bcast[14];
            if constexpr (!kLateBook) {
                bookkeeping(Rn, Tn, sn, rmse, prevRmse);
            } else {
                solved = true;
                if (lane == 0) {   // the new state, for every wave's next search phase (the flag stays as it is)
#pragma unroll
                    for (int k = 0; k < 9; ++k) bcast[k] = Rn[k];
                    bcast[9] = Tn[0]; bcast[10] = Tn[1]; bcast[11] = Tn[2];
                    bcast[13] = prevRmse;
                    bcast[14] = rmse;  // :213 prev_rmse = rmse
                    bcast[15] = sn;
                }
            }
            }  // !teamStop
        }
        if constexpr (kLateBook) {
            if (it + 1 < itEnd) {
                barrier_lds_only();                 // (R, T) are read back at the top of the loop
                if (bcast[12] == 0.f) { active = 0; itersDone = it + 1; break; }   // the team was told to stop (workgroup-uniform)
            }
            if (wave == 0 && solved) {   // (read back from where wave 0 has just published them: nothing is carried across the barrier)
                float R2[9], T2[3];
#pragma unroll
                for (int k = 0; k < 9; ++k) R2[k] = bcast[k];
                T2[0] = bcast[9]; T2[1] = bcast[10]; T2[2] = bcast[11];
                bookkeeping(R2, T2, bcast[15], bcast[14], bcast[13]);
            }
        }
        itersDone = it + 1;
        ICPFLOW_STAMP(7);
#ifdef ICPFLOW_TAIL_CLOCK
        tcLoop0 = clock64();
        tcTail += tcLoop0 - tcTail0;
#endif
        if (!kLateBook && it + 1 < itEnd) {  // more iterations inside this launch: publish (R, T) to the block
            barrier_lds_only();   // the history stores / tally atomic of wave 0 stay in flight
            if (bcast[12] == 0.f && (p.stopMode == ICPFLOW_STOP_PER_PAIR_ || p.history != nullptr)) {   // (workgroup-uniform)
                active = 0;
                // speculative mode: only member 0 watches the batch tally; it tells its team to stop
                // through the record of the iteration the others are about to exchange
                if (TEAM && G > 1 && rank == 0 && p.stopMode == ICPFLOW_STOP_REFERENCE_ && wave == 0)
                    team_publish(team, b, it + 1, 0, 0.0, 1.0, lane);
                break;
            }
        }
    }
    ICPFLOW_STAMP(8);
    __syncthreads();
#ifdef ICPFLOW_TAIL_CLOCK
    if (tid == 0 && b < 1024 && role == 0) { g_tail_clock[b * 3] = tcTail; g_tail_clock[b * 3 + 1] = tcSearch; g_tail_clock[b * 3 + 2] = itersDone; }
    if (tid < 16 && b < 1024) g_tail_split[b * 16 + tid] = g_tcSh[tid];
    if (tid == 0 && b < 8192 && role == 0) {   // (the pair's owner: helpers of the pair must not overwrite its record)
        g_wg_wall[b * 4] = tcWall0; g_wg_wall[b * 4 + 1] = wall_clock64();
        g_wg_wall[b * 4 + 2] = __builtin_amdgcn_s_getreg((4 /*HW_ID*/) | (0 << 6) | (31 << 11));
        g_wg_wall[b * 4 + 3] = __builtin_amdgcn_s_getreg((20 /*XCC_ID*/) | (0 << 6) | (31 << 11));
    }
#endif
    if constexpr (HELP) {
        if (helping) return;
        if (hp != nullptr && tid == 0) {
            __hip_atomic_store(&hp->iter, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&hp->epoch, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&p.help.owner[blockIdx.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (tid == 0 && rank == 0) {  // wave 0 (of member 0) holds the final state
#pragma unroll
        for (int k = 0; k < 9; ++k) st->R[k] = bcast[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) st->T[k] = bcast[9 + k];
        st->rmse = bcast[14];
        st->s = bcast[15];
        st->active = active | drainSh;
        st->iters = itersDone;
        if constexpr (HELP) {   // (a drained launch counts the pairs that are through; a pair that leaves still moving is not)
            if (p.drainAt > 0 && drainSh == 0) __hip_atomic_fetch_add(&ctrl->finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (p.stopMode == ICPFLOW_STOP_REFERENCE_) {
            if (b == 0) ctrl->iters = itersDone;
        } else {
            atomicMax(&ctrl->iters, itersDone);
            if (active) atomicAdd(&ctrl->notconv[0], 1);  // pairs still moving at maxIter
        }
    }
}

// Which pairs a workgroup serves.  Teams: the plan's (pair, rank).  Otherwise workgroup w starts with pair w; with
// `p.persistent` (a single launch of more pairs than the GPU holds workgroups of this kernel at once) the grid is only
// as large as the GPU and a workgroup that has finished a pair draws the next one from a ticket counter.  The hardware's
// own dispatcher deals workgroups to the eight XCDs round robin and IN ORDER: while one XCD has no free slot, the
// workgroups behind the one that waits for it do not start either, whatever is free elsewhere -- measured on 8192 pairs
// x 2048 points (per-workgroup wall clocks, tools/dbg/tail_clock_big.py): after the first third of the launch ~230 of
// the 512 slots are occupied.  A ticket has no such order.
// (PERSIST is a template parameter: the loop keeps more scalar state alive than the one-pair kernel, whose register
// allocation at 128 VGPRs must not move -- it is the kernel of batches that fit the GPU, config 2 among them.)
template <int BLOCK, int Q, int TS, int GRID, bool TEAM, bool SCALE = false, bool PERSIST = false, bool HELPK = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu((BLOCK == 512 && GRID == 4) ? 4 : 1)))
void icp_kernel(IcpParams p, int itBegin, int itEnd)
{
    int b = blockIdx.x, rank = 0, G = 1;     // pair, member and size of the team serving it
    if constexpr (TEAM) {
        b = p.team.wgPair[blockIdx.x];
        if (b < 0) return;                   // spare workgroup
        rank = p.team.wgRank[blockIdx.x];
        G = p.team.teamSize[b];
        // a chain of single-pass pairs (icp_team_plan_kernel): one after the other, like the tickets of a persistent grid
        for (;;) {
            icp_pair<BLOCK, Q, TS, GRID, TEAM, SCALE, false, true, HELPK>(p, b, rank, G, itBegin, itEnd);   // (teams: HELPK marks the instantiation with shared window scans)
            b = __builtin_amdgcn_readfirstlane(p.team.next[b]);
            if (b < 0) return;
            __syncthreads();                 // the pair's last reads of the static LDS state are done
            rank = 0; G = 1;
        }
    }
    if constexpr (!PERSIST) {
        // one workgroup per pair, at most one (1024 / 768 threads) per CU: the launch lasts as long as its slowest pair's chain
        // of iterations -- late bookkeeping.  512-thread workgroups share their CUs (two per CU: batches of two pairs per CU
        // and more), like the persistent grids below: one pair's bookkeeping already runs under another pair's search.
        int itB = itBegin;
        if constexpr (BLOCK == 1024 && GRID == 4 && !TEAM && !SCALE && Q == 1) {
            // the second of two launches (icp_split_kernel): workgroup w serves pair pairList[w] from the iteration IT had reached
            if (p.pairList != nullptr) {
                if ((int)blockIdx.x >= p.pairMeta[0]) return;
                b = __builtin_amdgcn_readfirstlane(p.pairList[blockIdx.x]);
                itB = __builtin_amdgcn_readfirstlane(p.state[b].iters);
            }
        }
        icp_pair<BLOCK, Q, TS, GRID, TEAM, SCALE, false, BLOCK != 512, false>(p, b, rank, G, itB, itEnd);
    } else {
        static_assert(!PERSIST || !TEAM, "teams are planned per launch");
        constexpr bool HELP = HELPK && GRID == 4 && !SCALE && Q == 1;
        __shared__ int nextPair, nextRole;
        // The parameters are read from the kernel-argument segment afresh for every pair (the pointer is laundered per
        // round): loads from that constant address space can be repeated where the register allocator would otherwise
        // keep ~60 scalars alive around the whole loop and spill them.
        typedef const __attribute__((address_space(4))) IcpParams KernargParams;
        KernargParams *pp = (KernargParams *)__builtin_amdgcn_kernarg_segment_ptr();
        int role = 0;
        for (;;) {
            asm volatile("" : "+s"(pp));
            icp_pair<BLOCK, Q, TS, GRID, TEAM, SCALE, HELP, false, false>(*pp, b, rank, G, itBegin, itEnd, role);
            __syncthreads();                     // the pair's last reads of the static LDS state are done
            if (threadIdx.x < kWave) {
                int nb = -1, nr = 0;
                if (role == 0) {                 // still drawing tickets
                    int t = 0;
                    if (threadIdx.x == 0) t = (int)gridDim.x + __hip_atomic_fetch_add(&p.ctrl->ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    nb = __shfl(t, 0, kWave);
                    if (nb >= p.B) nb = -1;
                }
                if constexpr (HELP) {
                    // No pair left to start: look, among the pairs other workgroups are iterating on, for the one that has
                    // come furthest and still has a pass to give away, and sign up for it (a few attempts: somebody else
                    // may get the slot first).  Nothing found: this workgroup is done.
                    if (nb < 0 && p.help.pair != nullptr && p.helpOn) {
                        const int lane = threadIdx.x;
                        // (Every workgroup ranks the pairs the same way, and they run out of tickets within microseconds of
                        // each other: with a strict ranking they all sign up for the same pair, three get in and the rest
                        // leave after a few attempts.  So the ranking is by classes (eight iterations of head-room each) and
                        // the choice inside a class by a hash of (pair, workgroup); a lost race for a slot just picks again.
                        // Measured on config 4's shard: 418 -> 557 helpers joining, 5.7 k -> 8.6 k passes taken over, the
                        // launch no shorter -- what paces a helped pair is its most expensive 64-query unit)
                        for (int attempt = 0; attempt < 64 && nb < 0; ++attempt) {
                            int best = -1, bestKey = 0;
                            for (int w = lane; w < (int)gridDim.x; w += kWave) {
                                const int ob = __hip_atomic_load(&p.help.owner[w], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
                                if (ob < 0) continue;
                                const HelpPair *h = p.help.pair + ob;
                                const int iter = __hip_atomic_load(&h->iter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const int nc = __hip_atomic_load(&h->nclaim, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const int ps = __hip_atomic_load(&h->passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                const int slots = min(kHelpSlots, ps - 1);
                                // The pair with the most iterations still ahead of it (nobody knows which pairs will
                                // stop early; the cap bounds what is left), among those past their first iterations --
                                // most pairs settle within a handful --, a pair that already has helpers first (the
                                // third helper shortens an iteration as much as the first).  A pair in its last
                                // iterations is not worth joining: two iterations pass before the first delivery.
                                if (iter > 0 && iter + 3 < itEnd && nc < slots) {
                                    const int cls = (iter > 6 ? 1 << 10 : 0) + ((itEnd - iter) >> 3) * 4 + nc + 1;
                                    const int key = cls * 64 + ((ob * 37 + (int)blockIdx.x * 11 + attempt * 5) & 63);
                                    if (key > bestKey) { bestKey = key; best = ob; }
                                }
                            }
#pragma unroll
                            for (int o = kWave / 2; o > 0; o >>= 1) {
                                const int ok = __shfl_xor(bestKey, o, kWave), ob2 = __shfl_xor(best, o, kWave);
                                if (ok > bestKey || (ok == bestKey && ob2 > best)) { bestKey = ok; best = ob2; }
                            }
                            if (best < 0) break;
                            int j = 0;
                            if (lane == 0) j = __hip_atomic_fetch_add(&p.help.pair[best].nclaim, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            j = __shfl(j, 0, kWave);
                            const int ps = __hip_atomic_load(&p.help.pair[best].passes, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (j < min(kHelpSlots, ps - 1)) { nb = best; nr = j + 1; }
                        }
                    }
                }
                if (threadIdx.x == 0) { nextPair = nb; nextRole = nr; }
            }
            __syncthreads();
            b = __builtin_amdgcn_readfirstlane(nextPair);   // (workgroup-uniform: keeps the pair's addresses scalar)
            role = __builtin_amdgcn_readfirstlane(nextRole);
            if (b < 0) break;
        }
    }
}

// speculative mode epilogue: the reference's stopping iteration is the first one at which every
// pair had arrived and none was unconverged; every pair's state is taken from its history there.
__global__ void icp_resolve_history_kernel(IcpState *__restrict__ st, IcpCtrl *__restrict__ ctrl,
                                           const float *__restrict__ history, int B, int maxIter)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = maxIter;
    for (int s = 0; s < maxIter; ++s) {
        const unsigned long long t = ctrl->tally[s];
        if ((int)(t & 0xffffffffull) == B && (t >> 32) == 0ull) { n = s + 1; break; }
    }
    const float *h = history + ((size_t)(n - 1) * B + b) * kHistStride;
    for (int k = 0; k < 9; ++k) st[b].R[k] = h[k];
    for (int k = 0; k < 3; ++k) st[b].T[k] = h[9 + k];
    st[b].rmse = h[12];
    st[b].s = h[13];
#ifdef ICPFLOW_DEBUG_EXECUTED
    st[b].rmse = (float)st[b].iters;   // developer builds: iterations this pair actually executed
#endif
    st[b].iters = n;
    if (ctrl->error) st[b].R[0] = __int_as_float(0x7fc00000);   // a team gave up waiting: poison
    if (b == 0) {
        ctrl->iters = n;
        // same convention as the per-iteration path: notconv[n-1] == 0 <=> converged
        ctrl->notconv[n - 1] = (int)(ctrl->tally[n - 1] >> 32);
    }
}

hipError_t launch_icp_resolve_history(IcpState *state, IcpCtrl *ctrl, const float *history, int B, int maxIter,
                                      hipStream_t s)
{
    hipLaunchKernelGGL(icp_resolve_history_kernel, dim3((B + 127) / 128), dim3(128), 0, s, state, ctrl, history, B,
                       maxIter);
    return hipGetLastError();
}

__global__ void icp_export_kernel(const IcpState *__restrict__ st, const IcpCtrl *__restrict__ ctrl,
                                  int B, int stopMode, float *__restrict__ R, float *__restrict__ T,
                                  float *__restrict__ rmse, int32_t *__restrict__ iters,
                                  int32_t *__restrict__ converged, float *__restrict__ scale)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        if (R) for (int k = 0; k < 9; ++k) R[(size_t)b * 9 + k] = ctrl->error ? __int_as_float(0x7fc00000) : st[b].R[k];
        if (T) for (int k = 0; k < 3; ++k) T[(size_t)b * 3 + k] = st[b].T[k];
        if (rmse) rmse[b] = st[b].rmse;
        if (scale) scale[b] = st[b].s;
    }
    if (b == 0) {
        const int n = ctrl->iters;
        if (iters) *iters = ctrl->error ? -1 : n;
        if (converged) {
            if (stopMode == ICPFLOW_STOP_REFERENCE_) *converged = (n > 0 && ctrl->notconv[n - 1] == 0) ? 1 : 0;
            else *converged = (ctrl->notconv[0] == 0) ? 1 : 0;
        }
    }
}

// Team plan for one launch (one block of 256 threads; B <= 256).  Round 4: sizes by PASS BOUNDARIES.
//
// A member's waves take one unit (64 consecutive sorted queries) per pass, so an iteration of a member lasts
// passes x (its slowest unit), passes = ceil(units per member / 12) for the 768-thread team kernel: a fifth workgroup on
// a pair of 73 units (19 -> 15 units per member) shortens nothing, the seventh (11 units: one pass) halves the iteration.
// So team sizes move from one level of "units per member" to the next -- ..., 36, 24, 12 (passes 3, 2, 1), then 8 and 4
// (fewer waves sharing a SIMD: the early iterations, where every unit scans its window, are VALU issue) -- and the spare
// workgroups go, one level at a time, to the pair whose estimated iteration is the longest (levels x a weight that grows
// with the length of the fixed cloud: what a unit costs follows the targets in its window).
//   * A pair first gets the team that lets its members keep the per-query RECORDS (neighbour certificates: a member's
//     share of the queries must fit `recCap`): a 2200-query pair served by ONE workgroup of a batch padded to 10000
//     searched every window in every iteration (90 k clocks per iteration against 25 k; ragged real-shape batch).
//   * Pairs of a single pass (<= 768 queries) are CHAINED: up to kTeamChain of them are served one after the other by one
//     workgroup (t.next), when the large pairs can use the workgroups that frees.  They finish within a few per cent
//     of the launch (nobody waits for anybody under the speculative batch rule; a chained pair only arrives later at
//     the tallies), and a workgroup that has finished its one small pair would idle for the rest of the launch.
// The sums of a team are added in member order: the plan decides the rounding of a registration's moment sums, so it
// depends on the batch's lengths alone (never on timing).
#ifndef ICPFLOW_TEAM_CHAIN
#define ICPFLOW_TEAM_CHAIN 4
#endif
constexpr int kTeamChain = ICPFLOW_TEAM_CHAIN;
#ifndef ICPFLOW_TEAM_MIN_SHARE
#define ICPFLOW_TEAM_MIN_SHARE 256
#endif
constexpr int kTeamMinShare = ICPFLOW_TEAM_MIN_SHARE;   // queries per member, at least
constexpr int kTeamWaves = 768 / kWave;   // units per pass of a member

// units per member at the level below `u`
__device__ __forceinline__ int team_next_level(int u)
{
    if (u > kTeamWaves) return (u - 1) / kTeamWaves * kTeamWaves;   // one pass fewer
    return u > 8 ? 8 : (u > 4 ? 4 : 0);
}
// relative length of an iteration at u units per member
__device__ __forceinline__ float team_level_cost(int u)
{
    if (u >= kTeamWaves) return (float)((u + kTeamWaves - 1) / kTeamWaves);
    return u > 8 ? 1.0f : (u > 4 ? 0.85f : 0.7f);
}

__global__ __launch_bounds__(256) void icp_team_plan_kernel(const int32_t *__restrict__ lenX,
                                                            const int32_t *__restrict__ lenY,
                                                            const uint8_t *__restrict__ swap, int B, IcpTeam t, int recCap,
                                                            const uint8_t *__restrict__ active)
{
    __shared__ int size[256];
    __shared__ int first[257];
    __shared__ int part[4];
    __shared__ int sh[8];        // [0] small pairs, [1] sum of minimum teams, [2] sum of wishes, [3] chain length, [4] slots for the large pairs
    __shared__ int teamList[256];
    __shared__ int xcdUsed[8];
    const int b = threadIdx.x, lane = b & (kWave - 1), wv = b >> 6;
    int n = 0, nf = 0;
    if (b < B) {
        const bool sw = swap != nullptr && swap[b] != 0;
        n = sw ? lenY[b] : lenX[b];
        nf = sw ? lenX[b] : lenY[b];
        // a pair that is not in the batch (options.d_pair_active) counts as the two EMPTY clouds a caller who knew the mask
        // beforehand hands over: the plan -- hence the order of every team's sums -- is the same whether the pair's clouds are
        // there or not (a frame pair's stage 2 on the whole superset, api.hip, against the serial path's empty clouds)
        if (active != nullptr && active[b] == 0) { n = 0; nf = 0; }
    }
    const int units = (n + kWave - 1) / kWave;
    const bool small = b < B && units <= kTeamWaves;
    const bool big = b < B && !small;
    // smallest team whose members keep their records; the team of one pass
    int gMin = 1;
    if (big && recCap > 0) {
        const int capUnits = max(recCap / kWave, 1);
        gMin = min(kMaxTeam, (units + capUnits - 1) / capUnits);
    }
    const int gWish = small ? 1 : min(kMaxTeam, (units + kTeamWaves - 1) / kTeamWaves);
    // (the square root: between no weight and the full ratio of the fixed clouds' lengths, measured in round 3)
    const float weight = big ? sqrtf((float)max(nf, 1024) / 1024.0f) : 0.f;
    for (int w = threadIdx.x; w < t.maxWG; w += blockDim.x) { t.wgPair[w] = -1; t.wgRank[w] = 0; }
    if (b < 8) sh[b] = 0;
    __syncthreads();
    // block-wide sum of one int per thread (all threads call it)
    auto block_sum_int = [&](int v) -> int {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
        __syncthreads();
        if (lane == 0) part[wv] = v;
        __syncthreads();
        return part[0] + part[1] + part[2] + part[3];
    };
    // block-wide exclusive prefix sum of one int per thread, in thread order (all threads call it)
    auto block_prefix_int = [&](int v) -> int {
        int inc = v;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int up = __shfl_up(inc, o, kWave);
            if (lane >= o) inc += up;
        }
        __syncthreads();
        if (lane == kWave - 1) part[wv] = inc;
        __syncthreads();
        int base = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) base += (k < wv) ? part[k] : 0;
        return base + inc - v;
    };
    const int nSmall = block_sum_int(small ? 1 : 0), nBig = B - nSmall;
    const int sumMin = block_sum_int(big ? gMin : 0);
    const int sumWish = block_sum_int(big ? max(gWish, gMin) : 0);
    // chain the single-pass pairs only as far as the large pairs can use the workgroups: chain length c frees
    // nSmall - ceil(nSmall / c) of them
    int chain = 1;
    while (chain < kTeamChain && nBig > 0 && t.maxWG - (nSmall + chain - 1) / chain < sumWish) ++chain;
    int slots = t.maxWG - (nSmall + chain - 1) / chain;     // workgroups for the large pairs
    const bool fits = slots >= sumMin;       // not even the minimum teams: every pair one workgroup, no chains (B <= maxWG)
    if (!fits) { chain = 1; slots = t.maxWG - nSmall; }
    // The levels this pair's team can take: (workgroups, estimated length of an iteration), from its minimum team down the
    // levels of units per member.  The spare workgroups go where they shorten the LONGEST estimated iteration: the
    // smallest bound tau such that every pair brought down to tau (or as far as it can go) still fits, by bisection
    // over the levels' costs (a block-wide sum per step), then what is left over to the pairs just above, in pair order.
    constexpr int kLevels = 8;
    int lvG[kLevels];          // (every loop over the levels is fully unrolled: the tables stay in registers)
    float lvC[kLevels];
    int gLast = 0;
    {
        int G = fits ? gMin : 1;
        int u = (units + G - 1) / G;
        bool open = big;
#pragma unroll
        for (int k = 0; k < kLevels; ++k) {
            lvG[k] = G; lvC[k] = open ? team_level_cost(u) * weight : 3.0e38f;
            if (open) gLast = G;
            const int uNext = team_next_level(u);
            const int gNext = uNext > 0 ? (units + uNext - 1) / uNext : 0;
            open = open && uNext > 0 && gNext <= kMaxTeam && gNext > G && n / max(gNext, 1) >= kTeamMinShare;
            if (open) { G = gNext; u = (units + G - 1) / G; }
        }
    }
    auto teams_at = [&](float tau) -> int {      // this pair's team under the bound tau: the first level that meets it, or its last
        int G = gLast;
#pragma unroll
        for (int k = kLevels - 1; k >= 0; --k)
            if (lvC[k] <= tau) G = lvG[k];
        return big ? G : 0;
    };
    int G = b < B ? 1 : 0;
    if (nBig > 0) {
        // bisection on tau between 0 (everybody at its last level) and the largest first-level cost
        float hiC = big ? lvC[0] : 0.f;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) hiC = fmaxf(hiC, __shfl_xor(hiC, o, kWave));
        __syncthreads();
        if (lane == 0) part[wv] = __float_as_int(hiC);
        __syncthreads();
        float hi = fmaxf(fmaxf(__int_as_float(part[0]), __int_as_float(part[1])), fmaxf(__int_as_float(part[2]), __int_as_float(part[3])));
        float lo = 0.f;
        // (hi always fits: every pair at its first level is sumMin <= slots, or one workgroup each)
        for (int step = 0; step < 14; ++step) {
            const float mid = 0.5f * (lo + hi);
            if (block_sum_int(teams_at(mid)) <= slots) hi = mid; else lo = mid;
        }
        G = big ? teams_at(hi) : G;
        // left-over workgroups: one more level for the pairs that can take one, in pair order
        int left = slots - block_sum_int(big ? G : 0);
        if (left > 0) {
            int want = 0;
#pragma unroll
            for (int k = 0; k + 1 < kLevels; ++k)
                if (big && lvG[k] == G && lvC[k + 1] < 3.0e38f && want == 0) want = lvG[k + 1] - G;
            const int before = block_prefix_int(want);
            if (want > 0 && before + want <= left) G += want;
        }
    }
    // a chain of single-pass pairs is one workgroup: its first pair carries the slot, the others hang on t.next
    {
        const int seq = block_prefix_int(small ? 1 : 0);   // this small pair's number among the small pairs (in pair order)
        if (small) first[seq] = b;      // (first[] is scratch here: small pair number -> pair)
        __syncthreads();
        const bool head = small && (seq % chain) == 0;
        if (b < B) t.next[b] = (small && seq + 1 < nSmall && (seq + 1) % chain != 0) ? first[seq + 1] : -1;
        __syncthreads();
        size[b] = (b < B) ? (small ? (head ? 1 : 0) : G) : 0;
    }
    __syncthreads();
    // Workgroup w is dispatched to XCD w % 8, so slot k = (w % 8) * per + w / 8 enumerates the
    // workgroups XCD by XCD (per = maxWG / 8 of them each).  A team takes consecutive slots of ONE
    // XCD (its exchange stays inside one L2); teams go to the XCD with the most free slots, which
    // spreads the launch over all eight L2s.  Only the teams of several members go through that (serial) loop; the
    // single workgroups then fill what is left, XCD by XCD.
    const int per = (t.maxWG % 8 == 0) ? t.maxWG / 8 : t.maxWG;
    const int nx = (per == t.maxWG) ? 1 : 8;
    // this pair's number among the teams / the single workgroups (in pair order; both counts in one word)
    const int numbers = block_prefix_int((b < B && size[b] > 1 ? 1 << 16 : 0) + (b < B && size[b] == 1 ? 1 : 0));
    const int teamNo = numbers >> 16, singleNo = numbers & 0xffff;
    if (b < B && size[b] > 1) teamList[teamNo] = b;
    const int nTeams = block_sum_int((b < B && size[b] > 1) ? 1 : 0);
    if (b == 0) {
        int used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool ok = true;
        for (int k = 0; k < nTeams && ok; ++k) {
            // (the least-used XCD, first one on ties; the counters stay in registers: no indexing by a variable)
            int x = 0, ux = used[0];
#pragma unroll
            for (int c = 1; c < 8; ++c)
                if (c < nx && used[c] < ux) { x = c; ux = used[c]; }
            const int sz = size[teamList[k]];
            if (ux + sz > per) ok = false;
            first[teamList[k]] = x * per + ux;
#pragma unroll
            for (int c = 0; c < 8; ++c) used[c] += (c == x) ? sz : 0;
        }
        if (!ok) {   // does not fit XCD by XCD: plain packing (teams may span two XCDs)
            int acc = 0;
            for (int k = 0; k < nTeams; ++k) { first[teamList[k]] = acc; acc += size[teamList[k]]; }
#pragma unroll
            for (int c = 0; c < 8; ++c) used[c] = 0;
            for (int c = 0; c < 8; ++c) { const int take = min(max(acc - c * per, 0), per); xcdUsed[c] = (c < nx) ? take : per; }
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c) xcdUsed[c] = (c < nx) ? used[c] : per;
        }
    }
    __syncthreads();
    if (b < B && size[b] == 1) {
        // the singleNo-th single workgroup: the singleNo-th free slot, XCD by XCD
        int skip = singleNo, slot = -1;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int freeC = per - xcdUsed[c];
            if (slot < 0 && c < nx) {
                if (skip < freeC) slot = c * per + xcdUsed[c] + skip;
                else skip -= freeC;
            }
        }
        first[b] = slot;    // (>= 0: the plan never hands out more workgroups than there are)
    }
    __syncthreads();
    if (b < B) {
        t.teamSize[b] = G;
        t.arrived[b] = 0u;
        for (int r = 0; r < size[b]; ++r) {
            const int k = first[b] + r;
            const int w = (per == t.maxWG) ? k : (k % per) * 8 + k / per;
            if (first[b] >= 0 && w < t.maxWG) { t.wgPair[w] = b; t.wgRank[w] = r; }
        }
    }
}

// Two launches for batches of a few rounds (round 6).  A persistent grid deals its pairs in index order, and which pairs
// are the long ones is not known beforehand (tools/dbg/order_predictor.py): config 4's shard (1024 pairs x 2048 points, two
// 512-thread workgroups per CU) keeps its 512 slots full for the first half of the launch and spends the second half on a
// thinning set of long pairs, each on HALF a CU (tools/dbg/help_timeline.py: 498 owners at 50 % of the span, 227 at 70 %, 55 at
// 85 %; 30-40 us per iteration while the CU is shared, ~20 us with helpers once it is not).  So the launch is DRAINED as soon as
// at most `drainAt` (the number of CUs) pairs are unfinished: every pair still iterating leaves behind its current iteration,
// still moving (IcpState: state, rmse, iterations; its history rows and tallies are in place).  This kernel, between the two
// launches, looks for the batch rule among the tallies (found: nobody goes on), finds the first iteration some pair has not
// reached yet (the floor of the second launch's search for the rule) and lists the pairs that left still moving; the SECOND launch
// gives each of them a whole CU -- one 1024-thread workgroup, two passes instead of four -- and resumes it at ITS iteration.
// What makes that bit-identical (ICPFLOW_OPT_TWO_LAUNCH against the default; tests/test_gpu_fullsize.py): the first launch keeps its moment sums
// per (pass, wave) and adds them in that order (redPasses, the helpers' bookkeeping) -- i.e. in the order of the UNITS of 64
// consecutive sorted queries, which is the same order whether 8 waves take 4 passes or 16 waves take 2; everything else of an
// iteration is a function of (R, T).  The neighbour certificates are rebuilt in a pair's first iteration of the second launch,
// the cycle detection and the own-convergence bits are restored from the pair's history rows (icp_pair).
#ifndef ICPFLOW_DRAIN_AT
#define ICPFLOW_DRAIN_AT 0   // 0: the number of CUs
#endif
__global__ __launch_bounds__(1024) void icp_split_kernel(const IcpCtrl *__restrict__ ctrl, const IcpState *__restrict__ st, int B, int maxIter,
                                                         int32_t *__restrict__ list, int32_t *__restrict__ meta)
{
    __shared__ int foundSh, floorSh, waveCnt[16];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
    if (tid == 0) { foundSh = 0; floorSh = maxIter; }
    __syncthreads();
    for (int s0 = tid; s0 < maxIter; s0 += 1024) {
        const unsigned long long t = ctrl->tally[s0];
        if ((int)(t & 0xffffffffull) >= B) { if ((t >> 32) == 0ull) foundSh = 1; }
        else atomicMin(&floorSh, s0);
    }
    __syncthreads();
    if (foundSh) {   // the batch rule holds at an iteration every pair has reached: nobody goes on
        if (tid == 0) { meta[0] = 0; meta[1] = 0; }
        return;
    }
    int total = 0;   // (workgroup-uniform)
    for (int b0 = 0; b0 < B; b0 += 1024) {
        const int b = b0 + tid;
        const bool on = b < B && st[b].active != 0 && st[b].iters < maxIter;
        const unsigned long long m = __ballot(on);
        if (lane == 0) waveCnt[wave] = __builtin_popcountll(m);
        __syncthreads();
        int before = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 16; ++w) { const int c = waveCnt[w]; before += w < wave ? c : 0; all += c; }
        if (on) list[total + before + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = b;
        total += all;
        __syncthreads();
    }
    if (tid == 0) { meta[0] = total; meta[1] = floorSh; }
}

template <int BLOCK, int Q, int TS, int GRID, bool TEAM = false, bool SCALE = false>
static void launch_icp_variant(const IcpParams &p, int B, int itBegin, int itEnd, hipStream_t s)
{
    if constexpr (GRID >= 3 && !SCALE) {   // similarity transforms: the sweep kernels' SCALE instantiation
        if (p.estimateScale != 0 || p.initS != nullptr) {
            launch_icp_variant<BLOCK, Q, TS, GRID, TEAM, true>(p, B, itBegin, itEnd, s);
            return;
        }
    }
    IcpParams q = p;
    q.persistent = 0;
    q.helpOn = 0;
    q.redPasses = 0;
    // workgroups of a kernel the GPU holds at once (per device and dynamic-LDS size: asked once per instantiation)
    auto capacity = [&](const void *kern, size_t dynBytes) -> long long {
        struct Slot { const void *k; int dev; size_t dyn; int perCu; };
        static std::atomic<int> nSlots{0};
        static Slot slots[32];
        int dev = 0;
        (void)hipGetDevice(&dev);
        int perCu = 0;
        const int have = nSlots.load(std::memory_order_acquire);
        for (int k = 0; k < have && k < 32; ++k)
            if (slots[k].k == kern && slots[k].dev == dev && slots[k].dyn == dynBytes) perCu = slots[k].perCu;
        if (perCu == 0) {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kern, BLOCK, dynBytes) != hipSuccess || perCu <= 0) perCu = 1;
            const int k = nSlots.load(std::memory_order_acquire);
            if (k < 32) { slots[k] = Slot{kern, dev, dynBytes, perCu}; nSlots.store(k + 1, std::memory_order_release); }   // (a lost race only asks again)
        }
        return (long long)perCu * device_cus();
    };
    auto dynBytes = [&](int redPasses) -> size_t {
        return (GRID == 2) ? (((size_t)p.gridH + 1) * 4 + 15) / 16 * 16 + (size_t)p.N * 16
               : (GRID == 4) ? (size_t)redPasses * (BLOCK / kWave) * kMoments * sizeof(double) +
                               (size_t)((p.N + kChunk - 1) / kChunk * kChunk) * 12 + (size_t)p.recCap * (p.x0Cache ? 32 : 20) : 0;
    };
    // Batches larger than the GPU (sorted-sweep kernels): a grid as large as the GPU, the other pairs by ticket
    // (icp_kernel<..., PERSIST>).  Batches of up to four rounds whose pairs take three passes or more: the launch ends
    // on a tail of stragglers, so the workgroups that run out of tickets HELP (icp_kernel<..., PERSIST, HELPK>), which
    // wants the moment sums per (pass, wave) -- a few per cent of an iteration, which larger batches (no tail to speak
    // of) and two-pass pairs (one pass to give away) do not get back.  The summation order is decided HERE, on the
    // shape of the batch alone: every variant of the launch (hardware dispatch, no helpers, one launch per iteration)
    // adds up the same way and gives the same bits.
    if constexpr (!TEAM && !SCALE && GRID == 4 && Q == 1) {
        constexpr auto kernH = &icp_kernel<BLOCK, Q, TS, GRID, false, false, true, true>;
        const int passes = (p.N + BLOCK - 1) / BLOCK;
        if (p.N <= kRecMaxN && passes >= 3 && dynBytes(passes) > 48 * 1024) {   // (before the occupancy is asked for)
            static std::atomic<unsigned long long> raisedH{0ull};
            ensure_dynamic_lds(reinterpret_cast<const void *>(kernH), 156 * 1024, &raisedH);
        }
        const long long capH = (p.N <= kRecMaxN && passes >= 3) ? capacity(reinterpret_cast<const void *>(kernH), dynBytes(passes)) : 0;
        const bool eligible = capH > 0 && (long long)B > capH && (long long)B <= 4 * capH && capH <= kHelpMaxWG;
        if (eligible) {
            q.redPasses = passes;
            if (p.persistent) {
                const size_t dyn = dynBytes(passes);
                q.persistent = 1;
                q.helpOn = (p.helpOn && p.help.pair != nullptr) ? 1 : 0;
                if constexpr (BLOCK == 512) {
                    // Two launches (icp_split_kernel): the grid of half-CU workgroups is drained once the unfinished pairs fit one
                    // whole CU each; the second launch is the 1024-thread kernel of batches that fit the GPU, its sums per (pass, wave)
                    // too.  Speculative single launch only (the pairs resume from their history rows).
                    const int cus = device_cus();
                    const int drainAt = ICPFLOW_DRAIN_AT > 0 ? min(ICPFLOW_DRAIN_AT, cus) : cus;
                    if (p.twoLaunch && q.history != nullptr && q.splitScratch != nullptr && itBegin == 0 && itEnd == q.maxIter && itEnd > 2 &&
                        q.stopMode == ICPFLOW_STOP_REFERENCE_ && q.helpOn && B > drainAt) {
                        constexpr auto kern2 = &icp_kernel<1024, 1, 1, 4, false, false>;
                        q.drainAt = drainAt;
                        hipLaunchKernelGGL(kernH, dim3((int)capH), dim3(BLOCK), dyn, s, q, itBegin, itEnd);
                        hipLaunchKernelGGL(icp_split_kernel, dim3(1), dim3(1024), 0, s, q.ctrl, q.state, B, q.maxIter, q.splitScratch, q.splitScratch + B);
                        IcpParams q2 = q;
                        q2.persistent = 0; q2.helpOn = 0; q2.drainAt = 0; q2.halfCu = 0;
                        q2.pairList = q.splitScratch;
                        q2.pairMeta = q.splitScratch + B;
                        q2.redPasses = (p.N + 1023) / 1024;
                        const size_t red2 = (size_t)q2.redPasses * 16 * kMoments * sizeof(double), img = (size_t)((p.N + kChunk - 1) / kChunk * kChunk) * 12;
                        q2.x0Cache = (p.recCap > 0 && red2 + img + (size_t)p.recCap * 32 <= (size_t)152 * 1024) ? 1 : 0;
                        const size_t dyn2 = red2 + img + (size_t)p.recCap * (q2.x0Cache ? 32 : 20);
                        if (dyn2 > 48 * 1024) {
                            static std::atomic<unsigned long long> raised2{0ull};
                            ensure_dynamic_lds(reinterpret_cast<const void *>(kern2), 156 * 1024, &raised2);
                        }
                        hipLaunchKernelGGL(kern2, dim3(drainAt), dim3(1024), dyn2, s, q2, 0, itEnd);
                        return;
                    }
                }
                hipLaunchKernelGGL(kernH, dim3((int)capH), dim3(BLOCK), dyn, s, q, itBegin, itEnd);
                return;
            }
        }
    }
    const size_t dyn = dynBytes(q.redPasses);
    if constexpr (!TEAM && !SCALE && GRID >= 3) {
        if (p.persistent && q.redPasses == 0) {
            constexpr auto kern = &icp_kernel<BLOCK, Q, TS, GRID, false, false, true, false>;
            if (dyn > 48 * 1024) {
                static std::atomic<unsigned long long> raisedP{0ull};
                ensure_dynamic_lds(reinterpret_cast<const void *>(kern), 156 * 1024, &raisedP);
            }
            const long long cap = capacity(reinterpret_cast<const void *>(kern), dyn);
            if ((long long)B > cap) {
                q.persistent = 1;
                hipLaunchKernelGGL(kern, dim3((int)cap), dim3(BLOCK), dyn, s, q, itBegin, itEnd);
                return;
            }
        }
    }
    if (dyn > 48 * 1024) {   // above the default dynamic-LDS limit: opt in once per instantiation and device
        static std::atomic<unsigned long long> raised{0ull};
        ensure_dynamic_lds(reinterpret_cast<const void *>(&icp_kernel<BLOCK, Q, TS, GRID, TEAM, SCALE>), 156 * 1024, &raised);
    }
    const int wgs = TEAM ? p.team.maxWG : B;
    if constexpr (TEAM && GRID == 4) {
        if (p.shareScans) {   // the instantiation with shared window scans (icp_pair: SHAREK)
            if (dyn > 48 * 1024) {
                static std::atomic<unsigned long long> raisedS{0ull};
                ensure_dynamic_lds(reinterpret_cast<const void *>(&icp_kernel<BLOCK, Q, TS, GRID, TEAM, SCALE, false, true>), 156 * 1024, &raisedS);
            }
            hipLaunchKernelGGL((icp_kernel<BLOCK, Q, TS, GRID, TEAM, SCALE, false, true>), dim3(wgs), dim3(BLOCK), dyn, s, q, itBegin, itEnd);
            return;
        }
    }
    hipLaunchKernelGGL((icp_kernel<BLOCK, Q, TS, GRID, TEAM, SCALE>), dim3(wgs), dim3(BLOCK), dyn, s, q, itBegin, itEnd);
}

// ---- optional per-launch timing of this (dominant) kernel with HIP events ---------------------
// bench.py needs the average launch duration of the dominant kernel measured on the stream it
// runs on; the events are recorded by the library because only it sees the individual launches.
// The recorder is an object the caller owns (icpflow_profile_t) and passes with the call's options.
struct LaunchProfile {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
};

#ifdef ICPFLOW_TAIL_CLOCK
extern "C" int icpflow_debug_tail_clock(long long *out3072)
{
    return (int)hipMemcpyFromSymbol(out3072, HIP_SYMBOL(g_tail_clock), sizeof(long long) * 3072);
}
extern "C" int icpflow_debug_pair_hclk(unsigned long long *out4096, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out4096, HIP_SYMBOL(g_pair_hclk), sizeof(unsigned long long) * 4096);
    if (reset) { static unsigned long long z[4096]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_hclk), z, sizeof(z)); }
    return rc;
}
extern "C" int icpflow_debug_unit_win(int *out16384)
{
    return (int)hipMemcpyFromSymbol(out16384, HIP_SYMBOL(g_unit_win), sizeof(int) * 16384);
}
extern "C" int icpflow_debug_unit_clk(long long *out8192, int pair)
{
    int rc = (int)hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_unit_clk), sizeof(long long) * 8192);
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_unit_pair), &pair, sizeof(int));
    return rc;
}
extern "C" int icpflow_debug_pair_help(unsigned long long *out1024, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out1024, HIP_SYMBOL(g_pair_help), sizeof(unsigned long long) * 1024);
    if (reset) { static unsigned long long z[1024]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_help), z, sizeof(z)); }
    return rc;
}
extern "C" int icpflow_debug_wg_wall(long long *out32768)
{
    return (int)hipMemcpyFromSymbol(out32768, HIP_SYMBOL(g_wg_wall), sizeof(long long) * 32768);
}
extern "C" int icpflow_debug_help_stats(unsigned long long *out8, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_help_stats), sizeof(unsigned long long) * 8);
    if (reset) { static unsigned long long z[8]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_help_stats), z, sizeof(z)); }
    return rc;
}
extern "C" int icpflow_debug_tail_split(long long *out16384)
{
    return (int)hipMemcpyFromSymbol(out16384, HIP_SYMBOL(g_tail_split), sizeof(long long) * 16384);
}
#endif
#ifdef ICPFLOW_DEBUG_SOLVE
extern "C" int icpflow_debug_solve(int pair, double *out64)
{
    if (out64 == nullptr) return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_dbg_pair), &pair, sizeof(int));
    return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_dbg_solve), sizeof(double) * 64);
}
extern "C" int icpflow_debug_xt(float *out12288)
{
    return (int)hipMemcpyFromSymbol(out12288, HIP_SYMBOL(g_dbg_xt), sizeof(float) * 12288);
}
#endif
#ifdef ICPFLOW_CERT_STATS
extern "C" int icpflow_debug_set_stats_block(int b)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stats_block), &b, sizeof(int));
}
extern "C" int icpflow_debug_cert_stats(unsigned long long *out512, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_cert_stats), sizeof(unsigned long long) * 512);
    rc |= (int)hipMemcpyFromSymbol(out512 + 512, HIP_SYMBOL(g_probe_stats), sizeof(unsigned long long) * 256);
    if (reset) {
        static unsigned long long zeros4[512];
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_occ_cert), zeros4, sizeof(zeros4));
        static unsigned long long zeros[512];
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_cert_stats), zeros, sizeof(zeros));
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_probe_stats), zeros, sizeof(unsigned long long) * 256);
    }
    return rc;
}
#endif
#ifdef ICPFLOW_CERT_STATS
extern "C" int icpflow_debug_occ_cert(unsigned long long *out512)
{
    return (int)hipMemcpyFromSymbol(out512, HIP_SYMBOL(g_occ_cert), sizeof(unsigned long long) * 512);
}
#endif
#ifdef ICPFLOW_PHASE_TIMING
extern "C" int icpflow_debug_set_stamp_block(int b)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_block), &b, sizeof(int));
}
extern "C" int icpflow_debug_phase_stamps(long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_stamps), sizeof(long long) * 16);
}
extern "C" int icpflow_debug_wave_stamps(long long *out256)
{
    return (int)hipMemcpyFromSymbol(out256, HIP_SYMBOL(g_wave_stamps), sizeof(long long) * 256);
}
#endif

LaunchProfile *profile_create(int capacity, hipError_t *err)
{
    LaunchProfile *p = new LaunchProfile;
    *err = hipSuccess;
    for (int i = 0; i < capacity; ++i) {
        hipEvent_t a, b;
        hipError_t e = hipEventCreate(&a);
        if (e == hipSuccess) {
            e = hipEventCreate(&b);
            if (e != hipSuccess) (void)hipEventDestroy(a);
        }
        if (e != hipSuccess) { *err = e; profile_destroy(p); return nullptr; }
        p->start.push_back(a); p->stop.push_back(b);
    }
    return p;
}

void profile_destroy(LaunchProfile *p)
{
    if (p == nullptr) return;
    for (hipEvent_t e : p->start) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->stop) (void)hipEventDestroy(e);
    delete p;
}

hipError_t profile_collect(LaunchProfile *p, double *total_ms, int *launches)
{
    double sum = 0.0;
    for (int i = 0; i < p->used; ++i) {
        hipError_t e = hipEventSynchronize(p->stop[i]);
        if (e != hipSuccess) return e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, p->start[i], p->stop[i]);
        if (e != hipSuccess) return e;
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = p->used;
    p->used = 0;
    return hipSuccess;
}

int device_cus()
{
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 1;
    int c = cache[dev].load(std::memory_order_relaxed);
    if (c == 0) {
        if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 1;
        cache[dev].store(c, std::memory_order_relaxed);
    }
    return c;
}

void ensure_dynamic_lds(const void *func, int bytes, std::atomic<unsigned long long> *mask)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    const unsigned long long bit = 1ull << (dev & 63);
    if (dev < 64 && (mask->load(std::memory_order_acquire) & bit)) return;
    // never more than the CU's 160 KiB less the kernel's static LDS (the request fails as a whole otherwise)
    hipFuncAttributes fa{};
    if (hipFuncGetAttributes(&fa, func) == hipSuccess) bytes = min(bytes, 160 * 1024 - (int)fa.sharedSizeBytes);
    (void)hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (dev < 64) mask->fetch_or(bit, std::memory_order_release);
}

static void launch_icp_iters(const IcpParams &p, int B, int itBegin, int itEnd, LaunchProfile *prof, hipStream_t s)
{
    const bool timed = prof != nullptr && prof->used < (int)prof->start.size();
    if (timed) (void)hipEventRecord(prof->start[prof->used], s);
    if (p.sortY != nullptr) {    // sorted sweep
        // one query per lane (Q = 1; two per lane on 8 waves measured 25 % slower at n = 1024): a wave's 64
        // consecutive sorted queries span the narrowest window;
        // clouds longer than the workgroup take several passes.  GRID 4 keeps the sorted fixed cloud
        // in LDS (12 B/point, up to 144 KiB of the CU's 160 KiB at N = 12288); beyond that GRID 3
        // streams it through scalar loads (no LDS image, any N the sort can handle).
        if (p.N <= 256) launch_icp_variant<256, 1, 1, 4>(p, B, itBegin, itEnd, s);
        else if (p.N <= 512) launch_icp_variant<512, 1, 1, 4>(p, B, itBegin, itEnd, s);
        else if (p.team.wgPair != nullptr) {   // several workgroups per large pair (N > 1024)
            // The members of a team wait for each other inside the launch.  Two team launches in flight at once (the
            // same host thread registering on several streams: hist_icp_many, frame pairs in flight) could each hold
            // part of the GPU with members that spin for teammates the other launch keeps from starting; so the team
            // launches on one device are chained by an event, whatever streams and host threads they come from.  (Not under stream
            // capture, where an event from outside the graph cannot be waited for: a captured registration stands alone.)
            // Round 4: launches that take at most half of the CUs (ICPFLOW_OPT_TEAMS_HALF_GPU, p.teamLanes == 2) are chained two
            // deep -- they alternate between two lanes, each lane a chain of its own, so that two of them (256 workgroups of
            // 152 KiB of LDS at most: all resident) run side by side; a full-GPU launch waits for both lanes and leaves its
            // event in both.
            // (three and four lanes on a third / a quarter of the CUs were measured: 1.42 / 1.53 ms per demo frame pair with four
            // in flight against 1.46 with two, and slower one at a time -- the teams get too small)
            // The chains belong to the DEVICE, not to the host thread (until version 206 they were thread-local: two host
            // threads registering on one GPU could have had four half-GPU team launches in flight): one table per process,
            // the wait / launch / record of a team launch under its lock (enqueues only; nothing here waits for the GPU).
            struct TeamLane { hipEvent_t ev[2] = {nullptr, nullptr}; int device = -1; bool recorded[2] = {false, false}; int next = 0; };
            static TeamLane lanes[64];
            static std::mutex lanesLock;
            int dev = -1;
            hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
            bool chain = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 && hipStreamIsCapturing(s, &cap) == hipSuccess &&
                         cap == hipStreamCaptureStatusNone;
            std::unique_lock<std::mutex> guard(lanesLock, std::defer_lock);
            if (chain) guard.lock();
            TeamLane &lane = lanes[chain ? dev : 0];
            if (chain && lane.ev[0] == nullptr) {
                lane = TeamLane{};
                lane.device = dev;
                if (hipEventCreateWithFlags(&lane.ev[0], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&lane.ev[1], hipEventDisableTiming) != hipSuccess) { lane.ev[0] = nullptr; chain = false; }
            }
            const bool half = p.teamLanes == 2;
            const int mine = lane.next & 1;
            if (chain) {
                for (int k = 0; k < 2; ++k)
                    if ((!half || k == mine) && lane.recorded[k]) (void)hipStreamWaitEvent(s, lane.ev[k], 0);
            }
            if (p.N <= 12288) launch_icp_variant<768, 1, 1, 4, true>(p, B, itBegin, itEnd, s);
            else launch_icp_variant<768, 1, 1, 3, true>(p, B, itBegin, itEnd, s);
            if (chain) {
                for (int k = 0; k < 2; ++k)
                    if (!half || k == mine) lane.recorded[k] = hipEventRecord(lane.ev[k], s) == hipSuccess;
                if (half) lane.next ^= 1;
            }
        }
        // 1024 threads (4 waves per SIMD, 128 VGPRs: the kernel fits but for three pointers spilled once
        // outside the loop) take a 1024-point cloud in one pass; up to 768 points 12 waves (170 VGPRs) do
        else if (p.N <= 768) launch_icp_variant<768, 1, 1, 4>(p, B, itBegin, itEnd, s);
        // batches larger than the GPU, clouds whose image and records fit half a CU's LDS: two 512-thread workgroups
        // per CU, so that one pair's serial tail (one wave) runs under the other pair's search phase
        else if (p.halfCu) launch_icp_variant<512, 1, 1, 4>(p, B, itBegin, itEnd, s);
        else if (p.N <= 12288) launch_icp_variant<1024, 1, 1, 4>(p, B, itBegin, itEnd, s);
        else launch_icp_variant<768, 1, 1, 3>(p, B, itBegin, itEnd, s);   // (the scalar-load sweep spills at 1024 threads)
    } else if (p.gridPts != nullptr) {  // exact grid search; grid in LDS while it fits 48 KiB (N <= 2048)
        if (p.N <= 256) launch_icp_variant<256, 1, 1, 2>(p, B, itBegin, itEnd, s);
        else if (p.N <= 2048) launch_icp_variant<1024, 1, 1, 2>(p, B, itBegin, itEnd, s);
        else launch_icp_variant<1024, 1, 1, 1>(p, B, itBegin, itEnd, s);
    } else {                     // all-pairs LDS scan; queries per pass = (BLOCK/64/TS) * 64 * Q
        if (p.N <= 256) launch_icp_variant<256, 2, 2, 0>(p, B, itBegin, itEnd, s);         // 256
        else if (p.N <= 512) launch_icp_variant<512, 2, 2, 0>(p, B, itBegin, itEnd, s);    // 512
        else if (p.N <= 1024) launch_icp_variant<1024, 4, 4, 0>(p, B, itBegin, itEnd, s);  // 1024
        else launch_icp_variant<1024, 4, 2, 0>(p, B, itBegin, itEnd, s);                   // 2048 per pass
    }
    if (timed) (void)hipEventRecord(prof->stop[prof->used++], s);
}

// Teams: with at most half of the CUs taken by one workgroup per pair, the spare CUs join the pairs
// whose moving cloud needs several passes (real clusters, N > 1024).  Needs every workgroup of the
// launch resident at once (members wait for each other): single-launch modes only.
bool icp_teams_wanted(const IcpTeam *team, const IcpOpts &opts, const GridScratch *grid, int B, int N, int maxIter,
                      int stopMode, const float *history)
{
    const bool speculative = stopMode == ICPFLOW_STOP_REFERENCE_ && history != nullptr && opts.speculative &&
                             maxIter > 1 && maxIter <= kHistIters;
    return opts.arith != ICPFLOW_ARITH_FP32_REFERENCE && team != nullptr && opts.teams && grid != nullptr && grid->mode == 3 &&
           N > 1024 && 2 * B <= device_cus() && B <= 256 && B <= min(icp_team_workgroups(opts), team->maxWG) &&
           (speculative || stopMode == ICPFLOW_STOP_PER_PAIR_);
    // (B <= the workgroups the plan really gets: with teamsHalfGpu on a part whose CUs / 2 is not a multiple of eight --
    // 104, 110, 120 CUs -- the plan has fewer slots than CUs / 2, and a pair without a slot would never be served)
}

// dynamic LDS of a team member with the LDS image (image + records): the CU's 160 KiB less the team kernel's static LDS
// (~16 KiB with the accumulators of the shared window scans: 143 KiB; ~6.7 KiB without: 152 KiB)
// workgroups of a team launch: one per CU, or one per CU of HALF the GPU (a multiple of the eight XCDs either way)
// Shared window scans (icp_pair, SHAREK): from this padded width on -- below, windows of 256 targets are rare.
#ifndef ICPFLOW_SHARE_LAUNCH_MIN_N
#define ICPFLOW_SHARE_LAUNCH_MIN_N 5000
#endif
bool icp_team_shares(const IcpOpts &opts, int N) { return opts.sharedScans && opts.adaptiveWindows && N >= ICPFLOW_SHARE_LAUNCH_MIN_N && N <= 12288; }
size_t icp_team_room(const IcpOpts &opts, int N) { return icp_team_shares(opts, N) ? (size_t)143 * 1024 : (size_t)152 * 1024; }
int icp_team_workgroups(const IcpOpts &opts)
{
    const int cus = device_cus();
    return opts.teamsHalfGpu ? max(8, cus / 2 / 8 * 8) : cus;
}

// the plan of a team launch (icp_team_plan_kernel): depends on the pairs' lengths and roles only
void launch_icp_team_plan(const IcpTeam *team, const int32_t *lenX, const int32_t *lenY, const uint8_t *swap, int B, int N,
                          const IcpOpts &opts, hipStream_t s)
{
    IcpTeam t = *team;
    t.maxWG = min(icp_team_workgroups(opts), team->maxWG);
    // (records behind the LDS image of the padded length: what a member's share of the queries has to fit, see launch_icp)
    const size_t imgT = (size_t)((N + kChunk - 1) / kChunk * kChunk) * 12;
    const size_t roomT = icp_team_room(opts, N);
    const int recCapT = (opts.adaptiveWindows && N <= 12288 && imgT + 64 * 20 <= roomT) ? (int)((roomT - imgT) / 20 / 64 * 64) : 0;
    hipLaunchKernelGGL(icp_team_plan_kernel, dim3(1), dim3(256), 0, s, lenX, lenY, swap, B, t, recCapT, opts.pairActive);
}

hipError_t launch_icp(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                      const uint8_t *swap, const float *prePose, int B, int N, double thres,
                      int maxIter, double relThr, int stopMode, IcpState *state, IcpCtrl *ctrl,
                      const GridScratch *grid, float *history, const IcpTeam *team, const IcpOpts &opts,
                      hipStream_t s)
{
    if (opts.arith == ICPFLOW_ARITH_FP32_REFERENCE) {   // study mode (icp_fp32.hip); api.hip validated the arguments
        if (history == nullptr || opts.fp32Scratch == nullptr || maxIter > kHistIters ||
            stopMode != ICPFLOW_STOP_REFERENCE_)
            return hipErrorInvalidValue;
        return launch_icp_fp32ref(X, Y, lenX, lenY, swap, prePose, B, N, thres, maxIter, relThr, state, ctrl, history,
                                  opts.fp32Scratch, s);
    }
    IcpParams p{};
    p.B = B;
    p.X = X; p.Y = Y; p.lenX = lenX; p.lenY = lenY; p.swap = swap; p.prePose = prePose; p.N = N;
    p.thr2 = (float)(thres * thres);
    p.relThr = (float)relThr;
    p.stopMode = stopMode; p.maxIter = maxIter; p.state = state; p.ctrl = ctrl;
    p.initR = opts.initR; p.initT = opts.initT; p.allowReflection = opts.allowReflection ? 1 : 0;
    p.initS = opts.initS; p.estimateScale = opts.estimateScale ? 1 : 0;
    hipError_t e = hipSuccess;
    bool recWanted = false;   // neighbour certificates (sorted sweep with the LDS image)
    if (opts.historyPending != nullptr) *opts.historyPending = false;
    if (!opts.ctrlCleared) {
        e = hipMemsetAsync(ctrl, 0, icp_ctrl_bytes(B), s);
        if (e != hipSuccess) return e;
    }
    if (grid != nullptr && grid->mode == 3 && grid->presorted) {
        // the scoring sweep of this batch left both clouds sorted along the fixed cloud's longest axis;
        // a translation-only pre-pose (hist_icp) keeps that order
        p.sortX = (const float4 *)grid->sortX; p.sortY = (const float4 *)grid->pts; p.sortAxis = grid->axis;
        p.sortYsoa = grid->sortYsoa;
        p.sweepMargin = (float)(1.01 * thres);
#ifdef ICPFLOW_CERT_STATS
        if (grid->occReady) { p.occHdr = grid->occHdr; p.occBits = grid->occBits; }
#endif
        p.sortedRaw = 1;
        recWanted = opts.adaptiveWindows;
    } else if (grid != nullptr && grid->mode == 3) {
        int NP2 = 64;
        while (NP2 < N) NP2 <<= 1;
        if ((size_t)NP2 * 8 > 64 * 1024)   // dynamic LDS above 64 KiB needs the attribute (N > 8192)
            ensure_dynamic_lds(reinterpret_cast<const void *>(&sort_clouds_kernel), 128 * 1024, &g_sortAttr);
        if (N > kChunkSortMinN && grid->ckey != nullptr) {   // long clouds: several workgroups per sort
            e = launch_sort_clouds_chunked(X, Y, lenX, lenY, swap, prePose, B, N, grid->axis, grid->sortX, grid->pts,
                                           grid->sortYsoa, nullptr, grid->ckey, grid->cidx, s, nullptr, grid->dirKeys);
            if (e != hipSuccess) return e;
        } else
        hipLaunchKernelGGL(sort_clouds_kernel, dim3(B, 2), dim3(kSortBlock), (size_t)NP2 * 8, s, X, Y, lenX, lenY,
                           swap, prePose, N, NP2, grid->axis, (float4 *)grid->sortX, (float4 *)grid->pts, grid->sortYsoa,
                           (float *)nullptr, 0, grid->dirKeys);
        p.sortX = (const float4 *)grid->sortX; p.sortY = (const float4 *)grid->pts; p.sortAxis = grid->axis;
        p.sortYsoa = grid->sortYsoa;
        p.sweepMargin = (float)(1.01 * thres);
        recWanted = opts.adaptiveWindows;
    } else if (grid != nullptr) {
        // bin the fixed cloud once: cell edge 1 % above the gate radius (rounding head-room)
        const float invh = (float)(1.0 / (1.01 * thres));
        hipLaunchKernelGGL(grid_build_kernel, dim3(B), dim3(kGridBlock), 0, s, X, Y, lenX, lenY, swap, N,
                           grid->H, invh, grid->origin, grid->start, grid->cursor, (float4 *)grid->pts);
        p.gridPts = (const float4 *)grid->pts; p.gridStart = grid->start; p.gridOrigin = grid->origin;
        p.gridH = grid->H; p.gridInvH = invh;
    }
    const int cus = device_cus();
    const bool speculative = stopMode == ICPFLOW_STOP_REFERENCE_ && history != nullptr && opts.speculative &&
                             maxIter > 1 && maxIter <= kHistIters;
    if (icp_teams_wanted(team, opts, grid, B, N, maxIter, stopMode, history)) {
        p.team = *team;
        p.team.maxWG = min(icp_team_workgroups(opts), team->maxWG);
        p.teamLanes = opts.teamsHalfGpu ? 2 : 1;
        // (the plan depends on the lengths alone: hist_icp launches it on its side stream, next to the vote)
        if (!opts.teamPlanned) launch_icp_team_plan(team, lenX, lenY, swap, B, N, opts, s);
    }
    if (p.sortY != nullptr) {
        // room for the per-query records behind the LDS image: every query of a pair when one workgroup serves it,
        // the member's own share of the queries in a team (the image of a 10^4-point cloud leaves room for ~1800)
        const size_t img = (size_t)((N + kChunk - 1) / kChunk * kChunk) * 12;
        const size_t room = p.team.wgPair != nullptr ? icp_team_room(opts, N) : (size_t)152 * 1024;   // dynamic LDS next to the kernel's static LDS (~3 KiB; teams: see kTeamRoom)
        int recCap = 0;
        if (p.team.wgPair != nullptr) {
            if (N <= 12288 && img + 64 * 20 <= room) recCap = (int)((room - img) / 20 / 64 * 64);
            p.shareScans = (recWanted && recCap > 0 && icp_team_shares(opts, N)) ? 1 : 0;
        } else if (N <= kRecMaxN) {
            recCap = N;
        }
        bool x0Cache = recCap > 0 && img + (size_t)recCap * 32 <= room;
        // Batches of at least two pairs per CU: 512-thread workgroups (compiled for 128 VGPRs), two per CU, so that one
        // pair's serial tail (one wave) runs under the other pair's search phase.  Measured +7 to +14 % at 512 ... 2048
        // pairs x 1024 points and 1024 x 1500 / 2048; -4 % at 300 x 1024 and -5 % at 600 x 2048, where half-size
        // workgroups run alone on their CUs or the last, partial round runs on a mostly empty GPU: hence two pairs per
        // CU at least, and the two ways of filling the GPU weighed.  (Decided on the shape alone, not on whether the
        // records are switched on: the moment sums follow the workgroup's shape.)
        const size_t half = 76 * 1024;   // (+ 2.2 KiB of static LDS each)
#ifndef ICPFLOW_HALF_CU_MIN_N
#define ICPFLOW_HALF_CU_MIN_N 768
#endif
        const double fillHalf = (double)B / ((double)((B + 2 * cus - 1) / (2 * cus)) * 2 * cus);
        const double fillFull = (double)B / ((double)((B + cus - 1) / cus) * cus);
        const size_t redHalf = N <= kRecMaxN ? (size_t)((N + 511) / 512) * 8 * kMoments * sizeof(double) : 0;   // (512 threads: 8 waves)
        const size_t redFull = N <= kRecMaxN ? (size_t)((N + 1023) / 1024) * 16 * kMoments * sizeof(double) : 0;
        if (p.team.wgPair == nullptr && B >= 2 * cus && N > ICPFLOW_HALF_CU_MIN_N && redHalf + img + (size_t)recCap * 20 <= half &&
            1.1 * fillHalf > fillFull) {
            p.halfCu = 1;
            x0Cache = redHalf + img + (size_t)recCap * 32 <= half;
        } else if (p.team.wgPair == nullptr) {
            x0Cache = x0Cache && redFull + img + (size_t)recCap * 32 <= room;
        }
        if (recWanted) { p.recCap = recCap; p.x0Cache = x0Cache ? 1 : 0; }
    }
    if (stopMode == ICPFLOW_STOP_REFERENCE_) {
        // Batch-global stop rule.  ONE launch runs every pair through all iterations speculatively,
        // keeping a per-iteration history; a pair leaves as soon as its trajectory is periodic (it then
        // publishes the rest of its history) or it observes that some iteration satisfied the batch
        // rule, and the epilogue picks every pair's state at exactly the reference's stopping iteration.
        // Nobody waits, so batches larger than the GPU are fine: late workgroups start as early pairs
        // leave (a pair that is neither periodic nor done by then iterates to the cap, <= kHistIters).
        // Beyond kHistIters iterations: one launch per iteration.
        if (speculative) {
            p.history = history;
            p.pairActive = opts.pairActive;   // (api.hip has refused the mask wherever this branch is not taken)
            p.persistent = opts.persistent ? 1 : 0;   // (one launch for all iterations: the ticket counter starts at zero)
            p.twoLaunch = (opts.twoLaunch && opts.splitScratch != nullptr) ? 1 : 0;   // (where it applies: launch_icp_variant)
            p.splitScratch = opts.splitScratch;
            p.help = opts.help;
            p.helpOn = (opts.helpers && maxIter <= kHelpMaxEpoch - 2) ? 1 : 0;   // (always: maxIter <= kHistIters here)
            launch_icp_iters(p, B, 0, maxIter, opts.profile, s);
            if (opts.historyPending != nullptr) {
                *opts.historyPending = true;   // the consumers read the history themselves (posefuse.hpp)
            } else {
                e = launch_icp_resolve_history(state, ctrl, history, B, maxIter, s);
                if (e != hipSuccess) return e;
            }
        } else {
            for (int it = 0; it < maxIter; ++it) launch_icp_iters(p, B, it, it + 1, opts.profile, s);
        }
    } else {
        p.persistent = opts.persistent ? 1 : 0;
        p.help = opts.help;
        // The hand-off words of the helper protocol carry the iteration epoch E = it + 1 in their low 8 bits
        // ((workgroup << 8) | E in HelpPair::from, (pair << 8) | E in IcpHelp::tag): a per-pair launch may run up to
        // kMaxIterCap iterations in one go, and an epoch past 255 would spill into the workgroup / pair bits.  The
        // batch-global rule never gets there (one launch covers <= kHistIters = 128 iterations).
        p.helpOn = (opts.helpers && maxIter <= kHelpMaxEpoch - 2) ? 1 : 0;
        launch_icp_iters(p, B, 0, maxIter, opts.profile, s);
    }
    return hipGetLastError();
}

// both clouds of every pair sorted along the fixed cloud's longest axis, without a pre-pose, with
// structure-of-arrays images of BOTH: input of the scoring sweep (nn.hip)
// selfCount (single-workgroup sorts only, N <= kChunkSortMinN): 1 = the kernel counts the valid rows itself, 2 = and
// forms the smaller-cloud-first flag itself (X = src); lenX / lenY / swap are then not read
hipError_t launch_sort_clouds_soa(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                                  const uint8_t *swap, int B, int N, const GridScratch *grid, hipStream_t s, int selfCount)
{
    if (selfCount != 0 && N > kChunkSortMinN) return hipErrorInvalidValue;
    int NP2 = 64;
    while (NP2 < N) NP2 <<= 1;
    if ((size_t)NP2 * 8 > 64 * 1024)
        ensure_dynamic_lds(reinterpret_cast<const void *>(&sort_clouds_kernel), 128 * 1024, &g_sortAttr);
    if (N > kChunkSortMinN && grid->ckey != nullptr)   // long clouds: several workgroups per sort (sort.hip)
        return launch_sort_clouds_chunked(X, Y, lenX, lenY, swap, nullptr, B, N, grid->axis, grid->sortX, grid->pts,
                                          grid->sortYsoa, grid->sortXsoa, grid->ckey, grid->cidx, s, grid->pairBox, grid->dirKeys);
    hipLaunchKernelGGL(sort_clouds_kernel, dim3(B, 2), dim3(kSortBlock), (size_t)NP2 * 8, s, X, Y, lenX, lenY, swap,
                       (const float *)nullptr, N, NP2, grid->axis, (float4 *)grid->sortX, (float4 *)grid->pts,
                       grid->sortYsoa, grid->sortXsoa, selfCount, grid->dirKeys);
    return hipGetLastError();
}

// The batch rule over a SUBSET of the pairs, after the fact (round 5: a frame pair's stage 2 iterates all the candidates of its
// superset beside stage 1, before it is known which of them are in the batch).  The speculative launch has left every pair's
// (R, T, rmse) of every iteration in the history and the tallies of the rule over ALL pairs; every pair has rows up to the first
// iteration s_all at which that rule held (a pair leaves only when it has seen such an iteration, when its trajectory is
// periodic -- it then writes all remaining rows -- or at the cap), and the rule over a subset holds no later.  This kernel
// recomputes, per iteration s <= s_all, "every ACTIVE pair converged" from the history's rmse values with the loop's own test
// (:195-198, :209: rel = (prev - rmse) / prev <= thr, false at iteration 0 and on a NaN) and REWRITES the tallies so that their
// readers (posefuse.hpp) find the subset's stopping iteration: exactly what a launch with options.d_pair_active would have left.
__global__ __launch_bounds__(1024) void icp_retally_kernel(IcpCtrl *__restrict__ ctrl, const float *__restrict__ history,
                                                           const uint8_t *__restrict__ active, int B, int maxIter, float relThr)
{
    __shared__ unsigned int bad[4];       // bit s: some active pair is not converged at iteration s
    __shared__ int limitSh;
    const int tid = threadIdx.x;
    if (tid < 4) bad[tid] = tid == 0 ? 1u : 0u;      // (iteration 0: rel = 1, nobody is converged)
    if (tid < kWave) {                               // the first iteration at which the rule over ALL pairs held (wave 0, 64 tallies a round)
        int lim = maxIter - 1;
        for (int s0 = 0; s0 < maxIter; s0 += kWave) {
            const int s = s0 + tid;
            bool hit = false;
            if (s < maxIter) {
                const unsigned long long t = ctrl->tally[s];
                hit = (int)(t & 0xffffffffull) == B && (t >> 32) == 0ull;
            }
            const unsigned long long m = __ballot(hit);
            if (m != 0ull) { lim = s0 + __builtin_ctzll(m); break; }
        }
        if (tid == 0) limitSh = lim;
    }
    __syncthreads();
    const int lim = limitSh;
    // one (pair, iteration) per thread and round, pairs fastest (neighbouring threads read neighbouring rows): every test reads
    // the two rmse values it compares -- no chain through the iterations
    for (int i = tid; i < B * lim; i += 1024) {
        const int b = i % B, s = i / B + 1;
        if (active[b] == 0) continue;
        const float prev = history[((size_t)(s - 1) * B + b) * kHistStride + 12];
        const float rm = history[((size_t)s * B + b) * kHistStride + 12];
        const float rel = (prev - rm) / prev;
        if (!(rel <= relThr)) atomicOr(&bad[s >> 5], 1u << (s & 31));
    }
    __syncthreads();
    for (int s = tid; s < maxIter; s += 1024) {
        unsigned long long t = 0ull;                              // beyond s_all: "not everybody has arrived"
        if (s <= lim) t = (unsigned long long)(unsigned)B | (((bad[s >> 5] >> (s & 31)) & 1u) ? (1ull << 32) : 0ull);
        ctrl->tally[s] = t;
    }
}

hipError_t launch_icp_retally(IcpCtrl *ctrl, const float *history, const uint8_t *active, int B, int maxIter, double relThr,
                              hipStream_t s)
{
    if (maxIter > kHistIters) return hipErrorInvalidValue;
    hipLaunchKernelGGL(icp_retally_kernel, dim3(1), dim3(1024), 0, s, ctrl, history, active, B, maxIter, (float)relThr);
    return hipGetLastError();
}

hipError_t launch_icp_export(IcpState *state, IcpCtrl *ctrl, int B, int stopMode, float *R,
                             float *T, float *rmse, int32_t *iters, int32_t *converged, hipStream_t s, float *scale)
{
    hipLaunchKernelGGL(icp_export_kernel, dim3((B + 127) / 128), dim3(128), 0, s, state, ctrl, B,
                       stopMode, R, T, rmse, iters, converged, scale);
    return hipGetLastError();
}

}  // namespace icpflow
