// hist.hip -- translation-histogram vote (a-1) and NMS + top-k peaks (a-2) for gfx950.
//
// Reference semantics: hist_cuda/cpp/hist_cuda_core.cuh:40-60 (vote) and
// utils_hist.py:21-29 (topk_nms).  Design (not a port): the reference launches one
// thread per (b,i,j) of the PADDED N x N grid and float-atomicAdds into global
// memory; here a workgroup owns a slice of X rows of one pair, keeps the Y tile in
// LDS (broadcast reads), votes into an LDS-private uint32 histogram when it fits
// and flushes the non-zero bins once.  Counters are uint32 (exact beyond 2^24).
#include "common.hpp"
#include "kernels.hpp"
#include "votekey.hpp"

namespace icpflow {

// ---------------------------------------------------------------------------------
// count_valid: len[b] = #(flag > 0)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_valid_kernel(const float4 *__restrict__ pts, int N,
                                                          int32_t *__restrict__ len)
{
    __shared__ int scratch[4];
    const int b = blockIdx.x;
    const float4 *p = pts + (size_t)b * N;
    int c[1] = {0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) c[0] += (p[i].w > 0.0f) ? 1 : 0;
    block_sum<1, int>(c, scratch);
    if (threadIdx.x == 0) len[b] = c[0];
}

// both clouds of a pair in one launch, plus the smaller-cloud-first flag of hist_icp
// (swap[b] = n_src > n_dst, strict: utils_match.py:139-146); swap may be NULL
__global__ __launch_bounds__(1024) void count_pair_kernel(const float4 *__restrict__ A, const float4 *__restrict__ C,
                                                         int N, int32_t *__restrict__ lenA,
                                                         int32_t *__restrict__ lenC, uint8_t *__restrict__ swap,
                                                         uint32_t *__restrict__ zero0, unsigned words0,
                                                         uint32_t *__restrict__ zero1, unsigned words1,
                                                         float *__restrict__ boxes)
{
    __shared__ int scratch[2 * 16];
    const int b = blockIdx.x;
    // scratch the later kernels of this call expect zeroed (ICP control block, scoring accumulators)
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < words0; k += gridDim.x * blockDim.x) zero0[k] = 0u;
    for (unsigned k = blockIdx.x * blockDim.x + threadIdx.x; k < words1; k += gridDim.x * blockDim.x) zero1[k] = 0u;
    const float4 *pa = A + (size_t)b * N, *pc = C + (size_t)b * N;
    int c[2] = {0, 0};
    // (1024 threads, four rows of each cloud in flight per thread: the kernel is one dependent chain of loads otherwise --
    // 17 us on a frame's 10000-point batch with 256 threads and one row at a time)
    for (int i0 = threadIdx.x; i0 < N; i0 += 4 * blockDim.x) {
        float wa[4], wc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = min(i0 + u * (int)blockDim.x, N - 1);
            wa[u] = pa[i].w; wc[u] = pc[i].w;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * (int)blockDim.x < N) { c[0] += (wa[u] > 0.0f) ? 1 : 0; c[1] += (wc[u] > 0.0f) ? 1 : 0; }
    }
    block_sum<2, int>(c, scratch);
    if (threadIdx.x == 0) {
        lenA[b] = c[0];
        lenC[b] = c[1];
        if (swap != nullptr) swap[b] = c[0] > c[1] ? 1 : 0;
    }
    if (boxes == nullptr) return;
    // Bounding boxes for the sorts of LONG clouds (sort.hip: every chunk's workgroup would read the whole pair for them -- 11-13 us
    // of each on a ragged batch of 128 pairs, measured with tools/dbg/sort_clocks.py): per cloud, the flagged rows and all rows below the count
    // (min / max are exact in any order: the same numbers the sorts would find).  The rows were fetched by the count just now.
    __shared__ float boxSh[16][kPairBoxStride];
    float bx[kPairBoxStride];
#pragma unroll
    for (int k = 0; k < kPairBoxStride; ++k) bx[k] = ((k % 6) < 3) ? kInf : -kInf;
#pragma unroll
    for (int side = 0; side < 2; ++side) {
        const float4 *pts = side == 0 ? pa : pc;
        const int n = c[side];
        for (int j0 = threadIdx.x; j0 < n; j0 += 4 * blockDim.x) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = pts[min(j0 + u * (int)blockDim.x, n - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u * (int)blockDim.x >= n) continue;
                const float v[3] = {q[u].x, q[u].y, q[u].z};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    bx[side * 12 + 6 + k] = fminf(bx[side * 12 + 6 + k], v[k]);
                    bx[side * 12 + 9 + k] = fmaxf(bx[side * 12 + 9 + k], v[k]);
                    if (q[u].w > 0.0f) {
                        bx[side * 12 + k] = fminf(bx[side * 12 + k], v[k]);
                        bx[side * 12 + 3 + k] = fmaxf(bx[side * 12 + 3 + k], v[k]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPairBoxStride; ++k)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            const float other = __shfl_xor(bx[k], o, kWave);
            bx[k] = ((k % 6) < 3) ? fminf(bx[k], other) : fmaxf(bx[k], other);
        }
    if ((threadIdx.x & (kWave - 1)) == 0)
#pragma unroll
        for (int k = 0; k < kPairBoxStride; ++k) boxSh[threadIdx.x >> 6][k] = bx[k];
    __syncthreads();
    if (threadIdx.x < kPairBoxStride) {
        const int k = threadIdx.x;
        float v = boxSh[0][k];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = ((k % 6) < 3) ? fminf(v, boxSh[w][k]) : fmaxf(v, boxSh[w][k]);
        boxes[(size_t)b * kPairBoxStride + k] = v;
    }
}

void launch_count_pair(const float *A, const float *C, int B, int N, int32_t *lenA, int32_t *lenC, uint8_t *swap,
                       hipStream_t s, void *zero0, size_t bytes0, void *zero1, size_t bytes1, float *boxes)
{
    hipLaunchKernelGGL(count_pair_kernel, dim3(B), dim3(1024), 0, s, (const float4 *)A, (const float4 *)C, N, lenA,
                       lenC, swap, (uint32_t *)zero0, (unsigned)(bytes0 / 4), (uint32_t *)zero1, (unsigned)(bytes1 / 4), boxes);
}

void launch_count_valid(const float *pts, int B, int N, int32_t *len, hipStream_t s)
{
    hipLaunchKernelGGL(count_valid_kernel, dim3(B), dim3(256), 0, s, (const float4 *)pts, N, len);
}

// ---------------------------------------------------------------------------------
// vote
// ---------------------------------------------------------------------------------
struct VoteBox {
    float min_x, min_y, min_z, max_x, max_y, max_z;
    int len_x, len_y, len_z;
};

// (v - min) / (max - min) with the loop-invariant denominator: the compiler expands an IEEE fp32
// division into  div_scale, rcp, two Newton steps on the reciprocal, a quotient, two residual
// corrections, div_fmas, div_fixup.  Scale and fixup are identities for operands in the normal
// range, and everything that depends only on the denominator is invariant: hoisting it leaves five
// dependent FMAs per quotient that reproduce the expansion's arithmetic bit for bit
// (tests: test_vote_quotient_is_the_ieee_quotient).  `fast` is false -- plain division -- when the
// box could put a numerator or the denominator near the ends of the exponent range.
struct AxisQuot {
    float r, y1;
    bool fast;
    bool pow2;   // r is a power of two (and `fast`): y1 = 1 / r exactly and the IEEE quotient a / r is the one product a * y1
};

__device__ __forceinline__ AxisQuot axis_quot_make(float mn, float mx)
{
    AxisQuot d;
    d.r = mx - mn;
    const float y0 = __builtin_amdgcn_rcpf(d.r);
    const float e0 = fmaf(-d.r, y0, 1.0f);
    d.y1 = fmaf(e0, y0, y0);
    // a nonzero numerator v - mn is at least half an ulp of mn: normal range guaranteed by |mn|
    d.fast = d.r > 1e-18f && d.r < 1e18f && fabsf(mn) > 1e-18f && fabsf(mn) < 1e18f;
    // A box whose extent is a power of two -- translation_frame 2.0: x and y run from -2 to 2, utils_hist.py:63-64 -- divides exactly:
    // y0 = rcp(2^k) = 2^-k, e0 = 0, y1 = y0; in axis_quot q0 = a * 2^-k is exact (normal range: `fast`), so e1 = e2 = 0 and the
    // five operations return q0.  The vote then skips the four that change nothing (vote_range<.., P2>).
    d.pow2 = d.fast && (__float_as_uint(d.r) & 0x007fffffu) == 0u;
    return d;
}

template <bool FAST = false>
__device__ __forceinline__ float axis_quot(float a, const AxisQuot &d)
{
    if (!FAST && !d.fast) return a / d.r;   // wave-uniform
    const float q0 = a * d.y1;
    const float e1 = fmaf(-d.r, q0, a);
    const float q1 = fmaf(e1, d.y1, q0);
    const float e2 = fmaf(-d.r, q1, a);
    return fmaf(e2, d.y1, q1);
}

// debug / test entry: out[i] = axis_quot(a[i]) next to the compiler's a[i] / r
__global__ void vote_quotient_probe_kernel(const float *__restrict__ a, int n, float mn, float mx,
                                           float *__restrict__ fast, float *__restrict__ ieee)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const AxisQuot d = axis_quot_make(mn, mx);
    fast[i] = d.pow2 ? a[i] * d.y1 : axis_quot(a[i], d);   // (what vote_range computes: the one product where the extent is a power of two)
    ieee[i] = a[i] / (mx - mn);
}

hipError_t launch_vote_quotient_probe(const float *a, int n, float mn, float mx, float *fast, float *ieee,
                                      hipStream_t s)
{
    hipLaunchKernelGGL(vote_quotient_probe_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, n, mn, mx, fast, ieee);
    return hipGetLastError();
}

constexpr int kVoteBlock = 256;  // threads; one X row per thread per slice
constexpr int kVoteTile = 1024;  // Y points staged in LDS per step (16 KiB)
constexpr int kVoteSpan = 2;     // sorted vote: Y tiles per workgroup (at most)


// The votes of one X row (per lane) against the staged targets [r0, r1) (wave-uniform bounds): the exact
// box test and bin arithmetic of hist_cuda_core.cuh:44-60.  FAST: all three quotients take the hoisted
// division and the bin index fits 24-bit multiplies (decided once per launch); LDS_HIST: counters in LDS.  Four targets per round, all four LDS
// reads issued before the first test: with one workgroup per CU (a frame-level batch) nothing else hides
// the LDS latency of a one-target loop.
// The quotient (v - min) / (max - min) of a difference one float below max can round to 1.0, i.e. p = len (the
// reference does not clamp, hist_cuda_core.cuh:52-58): with p_x = len_x the flat bin index runs past the pair's
// L bins into the next pair's (bins are one [B, L] allocation, hist_cuda.cu:59) -- or, for the last pair, past the
// allocation (undefined in the reference; dropped here).  LDS counters therefore hold vote_overflow_bins() extra slots
// behind the L bins (flushed into the next pair's bins), the global path drops what falls beyond `limit`.
__host__ __device__ inline int vote_overflow_bins(int len_y, int len_z) { return len_y * len_z + len_z + 1; }

#ifdef ICPFLOW_VOTE_STATS
// developer build (tools/dbg/vote_stats.py): [0] (row, target) evaluations the windows let through (valid rows x targets visited),
// [1] of those, evaluations inside the box (votes), [2] targets visited by waves (wave steps), [3] waves x windows with rows
__device__ unsigned long long g_vote_stats[4];
extern "C" int icpflow_debug_vote_stats(unsigned long long *out4, int reset)
{
    int rc = (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_vote_stats), sizeof(unsigned long long) * 4);
    if (reset) { static unsigned long long z[4]; rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_vote_stats), z, sizeof(z)); }
    return rc;
}
#endif

template <bool FAST, bool LDS_HIST, bool P2 = false>
__device__ __forceinline__ void vote_range(const float4 *__restrict__ tile, int r0, int r1, const float4 &xi,
                                           const VoteBox &box, const AxisQuot &dqx, const AxisQuot &dqy,
                                           const AxisQuot &dqz, uint32_t *__restrict__ counters, int limit)
{
    const float flx = (float)box.len_x, fly = (float)box.len_y, flz = (float)box.len_z;
    // largest floats below the (exclusive) upper ends of the box; an empty axis (max <= min) stays empty
    // because then pred(max) < min and the median is never v... unless v == pred(max) == min's neighbour:
    // guard it explicitly
    const bool empty = !(box.max_x > box.min_x && box.max_y > box.min_y && box.max_z > box.min_z);
    if (empty) return;
    const float hx = nextafterf(box.max_x, -INFINITY), hy = nextafterf(box.max_y, -INFINITY),
                hz = nextafterf(box.max_z, -INFINITY);
    [[maybe_unused]] const float p2x = dqx.y1 * flx, p2y = dqy.y1 * fly;   // (P2: exact products of a power of two and an integer below 2^23)
    for (int k = r0; k < r1; k += 4) {
        float4 t4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) t4[u] = tile[min(k + u, r1 - 1)];  // same address in every lane: broadcast
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (k + u >= r1) break;   // wave-uniform
            const float4 t = t4[u];
            const float vx = xi.x - t.x, vy = xi.y - t.y, vz = xi.z - t.z;
            // min <= v < max  <=>  the median of (v, min, pred(max)) is v: one v_med3 + one compare per axis
            // instead of two compares and a scalar AND (NaN fails both forms)
#ifdef ICPFLOW_VOTE_STATS
            {
                const bool in = __builtin_amdgcn_fmed3f(vx, box.min_x, hx) == vx && __builtin_amdgcn_fmed3f(vy, box.min_y, hy) == vy &&
                                __builtin_amdgcn_fmed3f(vz, box.min_z, hz) == vz;
                const unsigned long long m = __ballot(in);
                if ((threadIdx.x & 63) == __builtin_ctzll(__ballot(true))) atomicAdd(&g_vote_stats[1], (unsigned long long)__popcll(m));
            }
#endif
            if (__builtin_amdgcn_fmed3f(vx, box.min_x, hx) == vx && __builtin_amdgcn_fmed3f(vy, box.min_y, hy) == vy &&
                __builtin_amdgcn_fmed3f(vz, box.min_z, hz) == vz) {
                // (P2: the x and y extents of the box are powers of two -- the quotient is one exact product, see axis_quot_make)
                // ... and with it the quotient's product with float(len): (a 2^-k) len rounds once, like a (2^-k len) -- 2^-k len is exact
                // for any length the 24-bit fast path admits
                const int px = P2 ? (int)((vx - box.min_x) * p2x) : (int)(axis_quot<FAST>(vx - box.min_x, dqx) * flx);   // >= 0: truncation == floor
                const int py = P2 ? (int)((vy - box.min_y) * p2y) : (int)(axis_quot<FAST>(vy - box.min_y, dqy) * fly);
                const int pz = (int)(axis_quot<FAST>(vz - box.min_z, dqz) * flz);
                // FAST also promises len_x * len_y and len_z below 2^23: 24-bit multiplies (full rate) are exact
                int bin;
                if (FAST) {   // v_mad_i32_i24 is full rate (the compiler picks the quarter-rate v_mad_u64_u32 here)
                    int row;
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(row) : "v"(px), "s"(box.len_y), "v"(py));
                    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(bin) : "v"(row), "s"(box.len_z), "v"(pz));
                } else {
                    bin = (px * box.len_y + py) * box.len_z + pz;
                }
                if (!LDS_HIST && bin >= limit) continue;   // last pair: past the allocation
                atomicAdd(&counters[bin], 1u);
            }
        }
    }
}

// bins_u32: [B, L] zero-initialised.  swap (optional, per pair): vote with X and Y
// exchanged -- used by the fused registration path where "src" is the smaller cloud.
// edges (optional, device): box taken from edges (min = e[0], max = e[L-1]).
template <bool LDS_HIST>
__global__ __launch_bounds__(kVoteBlock) void hist_vote_kernel(
    const float4 *__restrict__ X, const float4 *__restrict__ Y, int NX, int NY, VoteBox box,
    const float *__restrict__ ex, const float *__restrict__ ey, const float *__restrict__ ez,
    const uint8_t *__restrict__ swap, uint32_t *__restrict__ bins_u32)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + sizeof(float4) * kVoteTile);

    const int b = blockIdx.y;
    const bool sw = swap != nullptr && swap[b] != 0;
    const float4 *xb = (sw ? Y : X) + (size_t)b * (sw ? NY : NX);
    const float4 *yb = (sw ? X : Y) + (size_t)b * (sw ? NX : NY);
    const int nx = sw ? NY : NX, ny = sw ? NX : NY;
    if (ex != nullptr) {
        box.min_x = ex[0]; box.max_x = ex[box.len_x - 1];
        box.min_y = ey[0]; box.max_y = ey[box.len_y - 1];
        box.min_z = ez[0]; box.max_z = ez[box.len_z - 1];
    }
    const int L = box.len_x * box.len_y * box.len_z;
    uint32_t *gb = bins_u32 + (size_t)b * L;
    const bool lastPair = b + 1 == (int)gridDim.y;
    const int Lx = L + vote_overflow_bins(box.len_y, box.len_z);   // LDS counters: the pair's bins + the overflow row
    const int limit = lastPair ? L : 0x7fffffff;

    const int i = blockIdx.x * kVoteBlock + threadIdx.x;
    float4 xi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nx) xi = xb[i];
    const bool xvalid = xi.w > 0.0f;
    // a slice without any valid X row has nothing to do (pads: utils_helper.py:191-192)
    if (!__syncthreads_or(xvalid ? 1 : 0)) return;

    if (LDS_HIST) {
        for (int k = threadIdx.x; k < Lx; k += kVoteBlock) lhist[k] = 0u;
    }
    // hist_cuda_core.cuh:52-54: (v-min)/(max-min) * float(len); the denominators
    // and float(len) are loop invariants
    const AxisQuot dqx = axis_quot_make(box.min_x, box.max_x), dqy = axis_quot_make(box.min_y, box.max_y),
                   dqz = axis_quot_make(box.min_z, box.max_z);
    const bool allFast = dqx.fast && dqy.fast && dqz.fast && (long long)box.len_x * box.len_y < (1 << 23) &&
                         box.len_z < (1 << 23);

    for (int j0 = 0; j0 < ny; j0 += kVoteTile) {
        const int tn = min(kVoteTile, ny - j0);
        __syncthreads();  // previous tile fully consumed (and lhist zeroed)
        int any = 0;
        for (int k = threadIdx.x; k < tn; k += kVoteBlock) {
            float4 t = yb[j0 + k];
            const bool valid = t.w > 0.0f;               // hist_cuda_core.cuh:40-43: only flagged points vote
            if (!valid) t.x = kInf;                      // an unflagged target fails the box test: v = x - inf
            tile[k] = t;
            any |= valid ? 1 : 0;
        }
        if (!__syncthreads_or(any)) continue;  // a tile of pads
        if (!xvalid) continue;
        if (allFast && dqx.pow2 && dqy.pow2) vote_range<true, LDS_HIST, true>(tile, 0, tn, xi, box, dqx, dqy, dqz, LDS_HIST ? lhist : gb, limit);
        else if (allFast) vote_range<true, LDS_HIST>(tile, 0, tn, xi, box, dqx, dqy, dqz, LDS_HIST ? lhist : gb, limit);
        else vote_range<false, LDS_HIST>(tile, 0, tn, xi, box, dqx, dqy, dqz, LDS_HIST ? lhist : gb, limit);
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int k = threadIdx.x; k < (lastPair ? L : Lx); k += kVoteBlock) {   // k >= L: the next pair's bin k - L
            const uint32_t v = lhist[k];
            if (v) atomicAdd(&gb[k], v);
        }
    }
}

// ---------------------------------------------------------------------------------
// z-sorted vote (fused registration path).  The vote box is thin in z (three 0.1 m bins,
// utils_hist.py:65), so of the n_x * n_y differences only those with |dz| < 0.1 can vote.
// Both clouds are sorted by z once (zsort_kernel); a wave of 64 consecutive sorted X rows then
// visits only the Y rows whose z lies in (z_lo - max_z, z_hi - min_z]: a contiguous range of the
// staged tile, read at one LDS address per step.  Every (i, j) that can pass the exact box test
// is still visited and the same test decides the vote: bins are bit-identical.
// ---------------------------------------------------------------------------------
constexpr int kZsortBlock = 1024;

struct ZsortCount {   // lenP != NULL: the kernel counts the valid rows itself (PairCountFuse, kernels.hpp)
    int32_t *lenP, *lenQ;
    uint8_t *swap;
    uint32_t *zero0; unsigned words0;
    uint32_t *zero1; unsigned words1;
};

// grid (B, 2): y = 0 sorts cloud P, y = 1 cloud Q; valid rows (flag > 0) first, ascending z
__global__ __launch_bounds__(kZsortBlock) void zsort_kernel(const float4 *__restrict__ P,
                                                           const float4 *__restrict__ Qc,
                                                           const int32_t *__restrict__ nP,
                                                           const int32_t *__restrict__ nQ, int N, int NP2full,
                                                           float4 *__restrict__ Ps, float4 *__restrict__ Qs,
                                                           uint32_t *__restrict__ bins, int L,
                                                           const float *__restrict__ ez, int len_z,
                                                           float *__restrict__ keyRec, ZsortCount cnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char dynLds[];
    unsigned long long *kv = reinterpret_cast<unsigned long long *>(dynLds);   // (sort key, row) pairs, NP2full of them
    __shared__ float bbScratch[6 * (kZsortBlock / kWave)];
    __shared__ float keyShared[kVoteKeyStride];
    const int b = blockIdx.x;
    int cntP, cntQ;
    if (cnt.lenP != nullptr) {
        // count_pair_kernel's work (lengths, swap flag, cleared scratch) done here: one launch less per registration
        const unsigned stride = gridDim.x * gridDim.y * kZsortBlock;
        const unsigned first = (blockIdx.y * gridDim.x + blockIdx.x) * kZsortBlock + threadIdx.x;
        for (unsigned k = first; k < cnt.words0; k += stride) cnt.zero0[k] = 0u;
        for (unsigned k = first; k < cnt.words1; k += stride) cnt.zero1[k] = 0u;
        __shared__ int cntScratch[2 * (kZsortBlock / kWave)];
        const float4 *pp = P + (size_t)b * N, *pq = Qc + (size_t)b * N;
        int c[2] = {0, 0};
        for (int i = threadIdx.x; i < N; i += kZsortBlock) {
            c[0] += (pp[i].w > 0.0f) ? 1 : 0;
            c[1] += (pq[i].w > 0.0f) ? 1 : 0;
        }
        block_sum<2, int>(c, cntScratch);
        cntP = c[0]; cntQ = c[1];
        if (blockIdx.y == 0 && threadIdx.x == 0) {
            cnt.lenP[b] = cntP;
            cnt.lenQ[b] = cntQ;
            if (cnt.swap != nullptr) cnt.swap[b] = cntQ > cntP ? 1 : 0;   // Q is the src role (utils_match.py:139-146)
        }
    } else {
        cntP = nP[b]; cntQ = nQ[b];
    }
    const float4 *in = (blockIdx.y == 0 ? P : Qc) + (size_t)b * N;
    float4 *out = (blockIdx.y == 0 ? Ps : Qs) + (size_t)b * N;
    // the vote that follows wants zeroed counters: each of the pair's two blocks clears one half
    // (hist_cuda.cu:59 at::zeros)
    {
        const int half = (L + 1) / 2;
        uint32_t *h = bins + (size_t)b * L + (size_t)blockIdx.y * half;
        const int cnt = blockIdx.y == 0 ? half : L - half;
        for (int k = threadIdx.x; k < cnt; k += kZsortBlock) h[k] = 0u;
    }
    // pad_segment layout (valid rows first): only the first n rows can be valid, and the network
    // only has to hold them -- next power of two >= n instead of >= N
    const int n = min(blockIdx.y == 0 ? cntP : cntQ, N);
    // sort key of this pair (votekey.hpp): z, or (z slab, long horizontal axis) for wide clusters
    const VoteKey vk = vote_key_params(P + (size_t)b * N, min(cntP, N), Qc + (size_t)b * N, min(cntQ, N),
                                       ez[len_z - 1] - ez[0], bbScratch, keyShared);
    if (blockIdx.y == 0 && threadIdx.x < kVoteKeyStride) keyRec[(size_t)b * kVoteKeyStride + threadIdx.x] = keyShared[threadIdx.x];
    int NP2 = kWave;
    while (NP2 < n) NP2 <<= 1;
    NP2 = min(NP2, NP2full);
    for (int j = threadIdx.x; j < NP2; j += kZsortBlock) {
        float k = kInf;
        if (j < n) {
            const float4 q = in[j];
            if (q.w > 0.0f) k = vote_key(vk, q.x, q.y, q.z);
        }
        kv[j] = sort_pack(k, j);
    }
    __syncthreads();
    bitonic_sort_lds(kv, NP2);
    for (int r = threadIdx.x; r < N; r += kZsortBlock) {
        // valid rows carry their key in w (the flag of a valid row is implied by its position below the
        // count); rows beyond the valid count have +inf keys
        float4 o = make_float4(0.f, 0.f, kInf, kInf);
        if (r < NP2) {
            const unsigned long long w = kv[r];
            const float k = sort_key_of(w);
            if (k < kInf) { o = in[sort_index_of(w)]; o.w = k; }
        }
        out[r] = o;
    }
}

// BLOCK: X rows (threads) per workgroup.  The counters in LDS (20 KiB for 41 x 41 x 3 bins) limit a CU to four
// workgroups whatever their size: batches that leave workgroups waiting take 512 rows -- eight waves share the counters,
// 32 waves per CU instead of 16 (config 4's shard: vote 1.66 -> 1.44 ms); batches that fit keep 256 (more, smaller
// workgroups spread better over the CUs).
// SPLIT: waves per 64 rows.  A wave walks the targets of its z window one after the other (~36 dependent-ish
// instructions per target, and a lone wave issues one instruction every ~2.4 ns): a pair's longest window paces its
// workgroup however idle the SIMD is.  With SPLIT = 2 a workgroup of BLOCK threads holds BLOCK / 2 rows and two waves
// share each 64 rows, half of the window each: twice the waves per CU (the LDS counters allow four workgroups per CU
// whatever their size), half the serial chain per wave.
template <int BLOCK, int SPLIT = 1>
__global__ __launch_bounds__(BLOCK) void hist_vote_sorted_kernel(
    const float4 *__restrict__ Xs, const float4 *__restrict__ Ys, const int32_t *__restrict__ nXv,
    const int32_t *__restrict__ nYv, int N, int len_x, int len_y, int len_z,
    const float *__restrict__ ex, const float *__restrict__ ey, const float *__restrict__ ez,
    const uint8_t *__restrict__ swap, int useLds, uint32_t *__restrict__ bins_u32,
    const float *__restrict__ keyRec, int span, const int32_t *__restrict__ work, int nPairs)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + sizeof(float4) * kVoteTile);
    // work list (vote_plan_kernel; a 1-D grid): this workgroup's (pair, row block, share of the Y rows), the workgroups
    // WITH rows first and the largest pairs first -- or nothing at all
    int item = 0;
    if (work != nullptr) {
        item = work[blockIdx.x];
        if (item < 0) return;
    }
    const int b = work != nullptr ? (item & 1023) : (int)blockIdx.y;
    const VoteKey vk = vote_key_load(keyRec + (size_t)b * kVoteKeyStride);
    const bool sw = swap != nullptr && swap[b] != 0;
    const float4 *xb = (sw ? Ys : Xs) + (size_t)b * N;
    const float4 *yb = (sw ? Xs : Ys) + (size_t)b * N;
    const int nx = (sw ? nYv : nXv)[b], ny = (sw ? nXv : nYv)[b];
    // blockIdx.x = row block * tsplit + share: on long clouds the Y tiles are dealt to `tsplit` workgroups
    // per row block (`span` sorted Y rows each), so that one huge pair does not pace the launch
    const int tsplit = (N + span - 1) / span;
    // The rows are sorted by z and a wave's work grows with the number of targets inside its z window: the slabs in the
    // middle of a cloud take twice as long as those at its ends.  So the waves of a pair are dealt to its workgroups
    // round robin -- wave w of row block rb takes the 64 rows of global wave w * rowBlocks + rb -- and every workgroup
    // gets slabs from everywhere (SQ counters before: 7.5 of a CU's 16 waves resident on average).
    constexpr int kRows = BLOCK / SPLIT;            // X rows per workgroup
    constexpr int kRowWaves = kRows / kWave;
    // (with a work list the waves are dealt over the row blocks the pair HAS, not over those of the padded width)
    const int rowBlocks = work != nullptr ? (nx + kRows - 1) / kRows : (N + kRows - 1) / kRows;
    const int rb = work != nullptr ? ((item >> 10) & 255) : (int)blockIdx.x / tsplit;
    const int jBegin = (work != nullptr ? (item >> 18) : (int)blockIdx.x % tsplit) * span;
    if (rb * kWave >= nx || jBegin >= ny) return;  // sorted: valid rows first (rb * 64: the first row of wave 0)
    const float min_x = ex[0], max_x = ex[len_x - 1];
    const float min_y = ey[0], max_y = ey[len_y - 1];
    const float min_z = ez[0], max_z = ez[len_z - 1];
    const int L = len_x * len_y * len_z;
    uint32_t *gb = bins_u32 + (size_t)b * L;
    const bool lastPair = b + 1 == nPairs;
    const int Lx = L + vote_overflow_bins(len_y, len_z);   // LDS counters: the pair's bins + the overflow row (vote_range)
    const int limit = lastPair ? L : 0x7fffffff;
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = (int)(threadIdx.x >> 6);
    const int share = SPLIT > 1 ? wv / kRowWaves : 0;           // which part of the window this wave takes
    const int i = ((SPLIT > 1 ? wv % kRowWaves : wv) * rowBlocks + rb) * kWave + lane;
    const bool xvalid = i < nx;
    float4 xi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (xvalid) xi = xb[i];
    if (useLds) {
        for (int k = threadIdx.x; k < Lx; k += BLOCK) lhist[k] = 0u;
    }
    const AxisQuot dqx = axis_quot_make(min_x, max_x), dqy = axis_quot_make(min_y, max_y), dqz = axis_quot_make(min_z, max_z);
    const bool allFast = dqx.fast && dqy.fast && dqz.fast && (long long)len_x * len_y < (1 << 23) && len_z < (1 << 23);
    const VoteBox box{min_x, min_y, min_z, max_x, max_y, max_z, len_x, len_y, len_z};
    // z window of this wave's rows: y.z in (zlo - max_z, zhi - min_z], widened by a rounding slack
    float zlo = xvalid ? xi.z : kInf, zhi = xvalid ? xi.z : -kInf;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        zlo = fminf(zlo, __shfl_xor(zlo, o, kWave));
        zhi = fmaxf(zhi, __shfl_xor(zhi, o, kWave));
    }
    const float slack = 1e-3f * (fabsf(max_z) + fabsf(min_z)) + 1e-5f * (fabsf(zlo) + fabsf(zhi)) + 1e-6f;
    const float wlo = zlo - max_z - slack, whi = zhi - min_z + slack;
    // wide pairs: the rows of this wave also share a neighbourhood along u; targets are visited slab by
    // slab, inside each slab only those whose u can fall into the box (u' = u - u0 is the key's minor part)
    float ulo = kInf, uhi = -kInf;
    int slabLo = 0, slabHi = 0;
    if (vk.wide) {
        const float xu = vk.uaxis == 0 ? xi.x : xi.y;
        ulo = xvalid ? xu : kInf; uhi = xvalid ? xu : -kInf;
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            ulo = fminf(ulo, __shfl_xor(ulo, o, kWave));
            uhi = fmaxf(uhi, __shfl_xor(uhi, o, kWave));
        }
        const float minu = vk.uaxis == 0 ? min_x : min_y, maxu = vk.uaxis == 0 ? max_x : max_y;
        const float su = 0.0625f + 1e-5f * (fabsf(ulo) + fabsf(uhi) + fabsf(vk.u0));   // two ulps of K, rounding
        ulo = ulo - maxu - vk.u0 - su;      // window of u' inside a slab
        uhi = uhi - minu - vk.u0 + su;
        slabLo = (int)floorf((wlo - vk.z0) / vk.h);
        slabHi = (int)floorf((whi - vk.z0) / vk.h);
    }
    const int jEnd = min(ny, jBegin + span);
    for (int j0 = jBegin; j0 < jEnd; j0 += kVoteTile) {
        const int tn = min(kVoteTile, jEnd - j0);
        __syncthreads();  // previous tile fully consumed (and lhist zeroed)
        for (int k = threadIdx.x; k < tn; k += BLOCK) tile[k] = yb[j0 + k];
        __syncthreads();
        if (!(zlo <= zhi)) continue;  // wave without valid rows (wave-uniform)
        const float *zkey = reinterpret_cast<const float *>(tile) + 3;   // the sort key rides in w
        const int nwin = vk.wide ? max(slabHi - slabLo + 1, 0) : 1;
        for (int win = 0; win < nwin; ++win) {
        int r0, r1;
        if (vk.wide) {
            const float sbase = (float)(slabLo + win) * kSlabStride;
            const float klo = sbase + fmaxf(ulo, 0.f), khi = sbase + fminf(uhi, kSlabStride - 0.5f);
            if (!(klo <= khi)) continue;
            r0 = sorted_count_below<false>(zkey, 4, tn, klo, lane);
            r1 = sorted_count_below<true>(zkey, 4, tn, khi, lane);
        } else {
            r0 = sorted_count_below<false>(zkey, 4, tn, wlo, lane);
            r1 = sorted_count_below<true>(zkey, 4, tn, whi, lane);
        }
        if (!xvalid) continue;
        // (r0, r1 are wave-uniform: scalar loop control)
        r0 = __builtin_amdgcn_readfirstlane(r0);
        r1 = __builtin_amdgcn_readfirstlane(r1);
        if (SPLIT > 1) {
            const int len = max(r1 - r0, 0);
            r1 = r0 + (len * (share + 1)) / SPLIT;
            r0 = r0 + (len * share) / SPLIT;
        }
#ifdef ICPFLOW_VOTE_STATS
        if (r1 > r0) {
            const unsigned long long rows = __ballot(true);      // (the valid rows: the others left at `continue` above)
            if (lane == __builtin_ctzll(rows)) {
                atomicAdd(&g_vote_stats[0], (unsigned long long)(r1 - r0) * __popcll(rows));
                atomicAdd(&g_vote_stats[2], (unsigned long long)(r1 - r0));
                atomicAdd(&g_vote_stats[3], 1ull);
            }
        }
#endif
        if (allFast && dqx.pow2 && dqy.pow2) {
            if (useLds) vote_range<true, true, true>(tile, r0, r1, xi, box, dqx, dqy, dqz, lhist, limit);
            else vote_range<true, false, true>(tile, r0, r1, xi, box, dqx, dqy, dqz, gb, limit);
        } else if (allFast) {
            if (useLds) vote_range<true, true>(tile, r0, r1, xi, box, dqx, dqy, dqz, lhist, limit);
            else vote_range<true, false>(tile, r0, r1, xi, box, dqx, dqy, dqz, gb, limit);
        } else {
            if (useLds) vote_range<false, true>(tile, r0, r1, xi, box, dqx, dqy, dqz, lhist, limit);
            else vote_range<false, false>(tile, r0, r1, xi, box, dqx, dqy, dqz, gb, limit);
        }
        }  // windows
    }
    if (useLds) {
        __syncthreads();
        for (int k = threadIdx.x; k < (lastPair ? L : Lx); k += BLOCK) {   // k >= L: the next pair's bin k - L
            const uint32_t v = lhist[k];
            if (v) atomicAdd(&gb[k], v);
        }
    }
}

// Work list of the sorted vote on batches of few, wide, RAGGED pairs (a frame's clusters padded to their longest, the ragged
// real-shape batch): the grid (row blocks of the padded width) x (shares of the padded Y rows) x pairs is 87 % workgroups
// without rows there -- 50 560 workgroups for 6 000 with work, each holding 36 KiB of LDS and eight waves while it finds that
// out, in front of the ones with work in dispatch order (a CU held 0.5 working workgroups on average: SQ_WAVE_CYCLES) -- and
// the largest pair starts wherever the batch put it.  Entry e of the list is the e-th workgroup WITH rows, the pairs taken
// by decreasing number of workgroups (ties by pair number), inside a pair row block by row block; entries behind the last
// one are -1 and their workgroups leave at once, after everybody else has been dispatched.  The counters are integers:
// neither the order nor the dealing of a pair's waves changes a bin.
// grid: ceil(U / 1024) blocks of 1024 threads, U = capacity of the list; nPairs <= 1024
__global__ __launch_bounds__(1024) void vote_plan_kernel(const int32_t *__restrict__ nXv, const int32_t *__restrict__ nYv,
                                                         const uint8_t *__restrict__ swap, int nPairs, int rows, int span,
                                                         int U, int32_t *__restrict__ work, int32_t *__restrict__ orderOut)
{
    __shared__ int wk[1024], by_[1024], order[1024], start[1025], part[16];
    const int t = threadIdx.x, lane = t & (kWave - 1), wv = t >> 6;
    int mine = 0, byMine = 0;
    if (t < nPairs) {
        const bool sw = swap != nullptr && swap[t] != 0;
        const int nx = (sw ? nYv : nXv)[t], ny = (sw ? nXv : nYv)[t];
        byMine = (ny + span - 1) / span;
        mine = ((nx + rows - 1) / rows) * byMine;
    }
    wk[t] = mine;
    __syncthreads();
    int rank = 0;
    if (t < nPairs)
        for (int u = 0; u < nPairs; ++u) { const int o = wk[u]; rank += (o > mine || (o == mine && u < t)) ? 1 : 0; }
    __syncthreads();
    if (t < nPairs) { order[rank] = t; by_[rank] = byMine; }
    if (blockIdx.x == 0 && t < nPairs && orderOut != nullptr) orderOut[rank] = t;   // (for the sweeps that follow: nn.hip)
    __syncthreads();
    // exclusive prefix over the sorted counts
    const int v = t < nPairs ? wk[order[t]] : 0;
    int inc = v;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) { const int up = __shfl_up(inc, o, kWave); if (lane >= o) inc += up; }
    if (lane == kWave - 1) part[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wv; ++k) base += part[k];
    start[t] = base + inc - v;
    if (t == 1023) start[1024] = base + inc;
    __syncthreads();
    const int total = start[1024];
    const int e = blockIdx.x * 1024 + t;
    if (e >= U) return;
    int item = -1;
    if (e < total) {
        int lo = 0, hi = nPairs;                 // last r with start[r] <= e
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (start[mid] <= e) lo = mid; else hi = mid; }
        const int q = e - start[lo], by = by_[lo];
        item = order[lo] | ((q / by) << 10) | ((q % by) << 18);
    }
    work[e] = item;
}

size_t vote_work_capacity(int B, int N)
{
    if (N <= 1023 || B > 1024) return 0;   // (the work list serves the four-waves-per-64-rows variant: ICPFLOW_VOTE_WIDE_N)
    const size_t full = (size_t)B * ((N + 127) / 128) * ((N + 511) / 512);   // (shares of 512 Y rows at the finest)
    return full > 32768 ? full : 32768;    // (small batches shorten the span: never above 16384 workgroups)
}

hipError_t launch_hist_vote_sorted(const float *X, const float *Y, int32_t *nX, int32_t *nY,
                                   int B, int N, const int lens[3], const float *ex, const float *ey,
                                   const float *ez, const uint8_t *swap, float *sortX, float *sortY,
                                   uint32_t *bins_u32, float *ckey, int *cidx, float *keyRec, hipStream_t s,
                                   const PairCountFuse *fuse, bool sideBusy, const float *boxes, int32_t *work, size_t workCap,
                                   int32_t *orderOut, bool *planned)
{
    if (planned != nullptr) *planned = false;
    if (fuse != nullptr && N > kChunkSortMinN) return hipErrorInvalidValue;   // only zsort_kernel counts
    const size_t L = (size_t)lens[0] * lens[1] * lens[2];
    int NP2 = 64;
    while (NP2 < N) NP2 <<= 1;
    if ((size_t)NP2 * 8 > 64 * 1024) {
        static std::atomic<unsigned long long> attr{0ull};   // per device
        ensure_dynamic_lds(reinterpret_cast<const void *>(&zsort_kernel), 128 * 1024, &attr);
    }
    if (N > kChunkSortMinN && ckey != nullptr) {   // long clouds: several workgroups per sort (sort.hip)
        hipError_t e = launch_zsort_chunked(X, Y, nX, nY, B, N, sortX, sortY, bins_u32, (int)L, ckey, cidx, ez, lens[2],
                                            keyRec, s, boxes);
        if (e != hipSuccess) return e;
    } else {
        ZsortCount zc{};
        if (fuse != nullptr) {
            zc.lenP = nX; zc.lenQ = nY; zc.swap = fuse->swapOut;
            zc.zero0 = (uint32_t *)fuse->zero0; zc.words0 = (unsigned)(fuse->bytes0 / 4);
            zc.zero1 = (uint32_t *)fuse->zero1; zc.words1 = (unsigned)(fuse->bytes1 / 4);
        }
        hipLaunchKernelGGL(zsort_kernel, dim3(B, 2), dim3(kZsortBlock), (size_t)NP2 * 8, s, (const float4 *)X,
                           (const float4 *)Y, nX, nY, N, NP2, (float4 *)sortX, (float4 *)sortY, bins_u32, (int)L, ez,
                           lens[2], keyRec, zc);
    }
    const size_t tile_bytes = sizeof(float4) * kVoteTile;
    const size_t lds_hist = tile_bytes + sizeof(uint32_t) * (L + (size_t)vote_overflow_bins(lens[1], lens[2]));
    const int useLds = lds_hist <= 64 * 1024;
    // Workgroup shape.  The counters in LDS limit a CU to four workgroups whatever their size: batches that leave
    // workgroups waiting take 512 rows per workgroup -- eight waves share the counters, 32 waves per CU instead of 16
    // (config 4's shard: vote 1.66 -> 1.44 ms; config 2, round 3: step 0.726 -> 0.715 ms); smaller batches keep 256.  Batches too
    // small to give every SIMD two waves (a frame's candidate pairs) deal the sorted Y rows of a pair to several
    // workgroups of 1024 rows, `span` Y rows each (64 x 1024: -5 % per registration; on a batch that fills the GPU the extra window
    // searches and counter flushes cost 13 %).
#ifndef ICPFLOW_VOTE_WIDE_N
#define ICPFLOW_VOTE_WIDE_N 1023
#endif
#ifndef ICPFLOW_VOTE_LIST_MIN_N
#define ICPFLOW_VOTE_LIST_MIN_N 4097
#endif
#ifndef ICPFLOW_VOTE_LIST_SPAN
#define ICPFLOW_VOTE_LIST_SPAN 2048
#endif
    constexpr int kVoteListSpan = ICPFLOW_VOTE_LIST_SPAN;
#ifndef ICPFLOW_VOTE_WIDE_B
#define ICPFLOW_VOTE_WIDE_B 2
#endif
    constexpr int kVoteWideB = ICPFLOW_VOTE_WIDE_B;   // ... on batches of at most that many pairs per CU
    constexpr int kVoteWideN = ICPFLOW_VOTE_WIDE_N;   // widths above it take the four-waves-per-64-rows variant on small batches
    const int cus = device_cus();
    const long long wgs256 = (long long)((N + kVoteBlock - 1) / kVoteBlock) * ((N + kVoteSpan * kVoteTile - 1) / (kVoteSpan * kVoteTile)) * B;
    int block = wgs256 >= 4LL * cus ? 512 : kVoteBlock;
    int span = kVoteSpan * kVoteTile;
    const long long waves = (long long)((N + kWave - 1) / kWave) * B;
    if (waves < 8LL * cus) {   // (1024 rows share one set of counters: the split's zeroing and flushing cost the least)
        block = 1024;
        while (span > 256 && waves * ((N + span - 1) / span) < 32LL * cus) span >>= 1;
    }
    const int tsplit = (N + span - 1) / span;
    const bool split = useLds && !sideBusy && block != 1024 && wgs256 >= 2LL * cus && wgs256 <= 4LL * cus;
    // Wide batches of few pairs (a frame's candidate clusters padded to max_points = 10000; the ragged real-shape batch): the
    // launch is paced by chains -- a wave walks its window target by target -- and by the LDS counters of the few CUs that
    // hold the large pairs' workgroups.  128 rows per workgroup, FOUR waves per 64 rows: a quarter of the chain per wave, four
    // times the CUs per pair.  Demo frame, stage 1 (93 pairs, width 10000; rows per workgroup x waves per 64 rows): 512 x 1
    // 186 us, 256 x 1 180, 128 x 1 186, 256 x 2 135, 128 x 2 121, 64 x 2 134, 128 x 4 105, 64 x 4 103; the pair's valid rows
    // dealt to as few 512-row workgroups as hold them: 188 (the padded width spreads a pair over more CUs: kept).
    // Round 4: from 1024 rows on instead of from 4097 (tools/dbg/define_sweep.sh, ICPFLOW_VOTE_WIDE_N = 4096 / 2047 / 1023 / 511 / 255):
    // the demo frame pair through icpflow_track_frame 1.68 / 1.60 / 1.56 / 1.55 / 1.57 ms at max_points 2048 (stage 1 is 97 pairs
    // 2048 wide: its vote 129 -> 60 us) and 1.95 / 1.90 / 1.88 / 1.86 / 1.88 at 10000 (stage 2's superset, 1024 wide); config 2
    // (256 x 1024) 0.712 / 0.709 / 0.700 / 0.700 / 0.703 ms per step; batches above 2 x #CUs pairs are not concerned.
    if (useLds && N > kVoteWideN && B <= kVoteWideB * cus) {
        dim3 grid(((N + 127) / 128) * tsplit, B);
        // the list's shares of a pair's Y rows (ICPFLOW_VOTE_LIST_SPAN each, at most).  Measured on the ragged real-shape batch
        // (vote, independent / matched sizes): 512 rows 236 / 421 us, 1024 184 / 352, 2048 171 / 305-313, 4096 186 / 285, one share
        // per pair 253 / 324 -- the launch is not paced by its last workgroups; a share costs a window search and a flush
        const int lspan = (work != nullptr && N >= ICPFLOW_VOTE_LIST_MIN_N) ? min(span, kVoteListSpan) : span;
        const int ltsplit = (N + lspan - 1) / lspan;
        const size_t U = (size_t)((N + 127) / 128) * ltsplit * B;
        // (from the width on at which count_pair, not the sort, counts the rows: a full batch like config 2's 256 x 1024 gains nothing
        // and pays the plan's launch, 0.705 -> 0.72 ms per step)
        const bool listed = work != nullptr && U <= workCap && B <= 1024 && (N + 127) / 128 <= 256 && ltsplit <= 128 &&
                            N >= ICPFLOW_VOTE_LIST_MIN_N;
        if (listed) {
            hipLaunchKernelGGL(vote_plan_kernel, dim3((unsigned)((U + 1023) / 1024)), dim3(1024), 0, s, nX, nY, swap, B, 128, lspan,
                               (int)U, work, orderOut);
            if (planned != nullptr) *planned = orderOut != nullptr;
            grid = dim3((unsigned)U, 1);
        }
        hipLaunchKernelGGL((hist_vote_sorted_kernel<512, 4>), grid, dim3(512), lds_hist, s,
                           (const float4 *)sortX, (const float4 *)sortY, nX, nY, N, lens[0], lens[1], lens[2], ex, ey,
                           ez, swap, useLds, bins_u32, keyRec, listed ? lspan : span, listed ? work : (const int32_t *)nullptr, B);
        return hipGetLastError();
    }
    if (block == 1024) {
        dim3 grid(((N + 1023) / 1024) * tsplit, B);
        hipLaunchKernelGGL(hist_vote_sorted_kernel<1024>, grid, dim3(1024), useLds ? lds_hist : tile_bytes, s,
                           (const float4 *)sortX, (const float4 *)sortY, nX, nY, N, lens[0], lens[1], lens[2], ex, ey,
                           ez, swap, useLds, bins_u32, keyRec, span, (const int32_t *)nullptr, B);
    } else if (block == 512 && !split) {
        dim3 grid(((N + 511) / 512) * tsplit, B);
        hipLaunchKernelGGL(hist_vote_sorted_kernel<512>, grid, dim3(512), useLds ? lds_hist : tile_bytes, s,
                           (const float4 *)sortX, (const float4 *)sortY, nX, nY, N, lens[0], lens[1], lens[2], ex, ey,
                           ez, swap, useLds, bins_u32, keyRec, span, (const int32_t *)nullptr, B);
    } else if (split) {
        // 256 rows per workgroup, two waves per 64 rows (SPLIT): 32 waves per CU at four workgroups per CU.  Not beside the
        // axis sort of hist_icp's side stream: its 1024-thread workgroups find no room on a CU that full and the sort,
        // instead of hiding in the vote's gaps, ends after it (config 2: vote 0.136 -> 0.113 ms alone, step unchanged).
        dim3 grid(((N + kVoteBlock - 1) / kVoteBlock) * tsplit, B);
        hipLaunchKernelGGL((hist_vote_sorted_kernel<2 * kVoteBlock, 2>), grid, dim3(2 * kVoteBlock), lds_hist, s,
                           (const float4 *)sortX, (const float4 *)sortY, nX, nY, N, lens[0], lens[1], lens[2], ex, ey,
                           ez, swap, useLds, bins_u32, keyRec, span, (const int32_t *)nullptr, B);
    } else {
        dim3 grid(((N + kVoteBlock - 1) / kVoteBlock) * tsplit, B);
        hipLaunchKernelGGL(hist_vote_sorted_kernel<kVoteBlock>, grid, dim3(kVoteBlock), useLds ? lds_hist : tile_bytes, s,
                           (const float4 *)sortX, (const float4 *)sortY, nX, nY, N, lens[0], lens[1], lens[2], ex, ey,
                           ez, swap, useLds, bins_u32, keyRec, span, (const int32_t *)nullptr, B);
    }
    return hipGetLastError();
}

__global__ void u32_to_f32_kernel(const uint32_t *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

hipError_t launch_hist_vote(const float *X, const float *Y, int B, int NX, int NY,
                            const float mins[3], const float maxs[3], const int lens[3],
                            const float *ex, const float *ey, const float *ez,
                            const uint8_t *swap, uint32_t *bins_u32, hipStream_t s)
{
    VoteBox box{mins ? mins[0] : 0.f, mins ? mins[1] : 0.f, mins ? mins[2] : 0.f,
                maxs ? maxs[0] : 0.f, maxs ? maxs[1] : 0.f, maxs ? maxs[2] : 0.f,
                lens[0], lens[1], lens[2]};
    const size_t L = (size_t)lens[0] * lens[1] * lens[2];
    hipError_t e = hipMemsetAsync(bins_u32, 0, sizeof(uint32_t) * L * (size_t)B, s);
    if (e != hipSuccess) return e;
    // rows of the X role per pair: NX, or either cloud when roles may be swapped
    const int rows = swap ? (NX > NY ? NX : NY) : NX;
    dim3 grid((rows + kVoteBlock - 1) / kVoteBlock, B);
    const size_t tile_bytes = sizeof(float4) * kVoteTile;
    const size_t lds_hist = tile_bytes + sizeof(uint32_t) * (L + (size_t)vote_overflow_bins(lens[1], lens[2]));
    if (lds_hist <= 64 * 1024) {
        hipLaunchKernelGGL(hist_vote_kernel<true>, grid, dim3(kVoteBlock), lds_hist, s,
                           (const float4 *)X, (const float4 *)Y, NX, NY, box, ex, ey, ez, swap,
                           bins_u32);
    } else {
        hipLaunchKernelGGL(hist_vote_kernel<false>, grid, dim3(kVoteBlock), tile_bytes, s,
                           (const float4 *)X, (const float4 *)Y, NX, NY, box, ex, ey, ez, swap,
                           bins_u32);
    }
    return hipGetLastError();
}

hipError_t launch_u32_to_f32(const uint32_t *in, float *out, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(u32_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// peaks: separable (z, y, x) running maximum through two scratch volumes, then k
// rounds of block-wide arg-max on the key (vote << 32 | ~index).
// One workgroup per pair; the volumes live in global scratch (L2 resident: a demo
// histogram is 20 KiB, the largest Waymo one 868 KiB) so every size takes the same path.
// ---------------------------------------------------------------------------------
constexpr int kPeakBlock = 1024;
constexpr int kPeakMaxK = 8;

// max of V over the window |q - c| <= R along one axis (extent n, `stride` words per step), around flat position f.  With the
// radius known at compile time the 2R + 1 reads are independent and in flight together -- positions beyond the volume are
// clamped onto its edge, which repeats a value and leaves a maximum alone; as a loop with run-time bounds every read waited
// for the one before it (the reference's kernel_size = 11: 11 + 11 + 3 dependent LDS reads per bin, 25 k of the kernel's
// 45 k clocks).
template <int R>
__device__ __forceinline__ uint32_t peak_window_max(const uint32_t *__restrict__ V, int f, int c, int n, int stride)
{
    uint32_t v[2 * R + 1];
#pragma unroll
    for (int dq = -R; dq <= R; ++dq) v[dq + R] = V[f + (min(max(c + dq, 0), n - 1) - c) * stride];
    uint32_t m = v[0];
#pragma unroll
    for (int u = 1; u < 2 * R + 1; ++u) m = max(m, v[u]);
    return m;
}

#ifdef ICPFLOW_PEAK_CLOCK
__device__ unsigned long long g_peakClock[1024][8];
#define PEAK_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_peakClock[blockIdx.x][k] = wall_clock64(); } while (0)
#else
#define PEAK_STAMP(k) do { } while (0)
#endif
// MEM = 0: scratch volumes in global memory (any size); MEM = 1: three volumes in dynamic LDS
// (3 * L * 4 bytes <= 150 KiB, i.e. up to 113 x 113 x 3 bins)
template <typename BinT, int MEM>
__global__ __launch_bounds__(kPeakBlock) void hist_peaks_kernel(
    const BinT *__restrict__ bins, int Lx, int Ly, int Lz, int k, int radius,
    uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB, float *__restrict__ votes,
    int64_t *__restrict__ idx_out, PeakDecode dec)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    __shared__ unsigned long long red[(kPeakBlock / kWave) * kPeakMaxK];   // the waves' k largest keys
    __shared__ unsigned long long chosen[kPeakMaxK];
    const int b = blockIdx.x;
    const int L = Lx * Ly * Lz;
    const BinT *h = bins + (size_t)b * L;
    uint32_t *H0 = reinterpret_cast<uint32_t *>(smem);           // MEM = 1: the votes themselves
    uint32_t *A = MEM ? H0 + L : wsA + (size_t)b * L;
    uint32_t *Bv = MEM ? H0 + 2 * (size_t)L : wsB + (size_t)b * L;
    const int tid = threadIdx.x;
    PEAK_STAMP(0);
    if (MEM) {
        for (int f = tid; f < L; f += kPeakBlock) H0[f] = (uint32_t)h[f];
        __syncthreads();
    }
    PEAK_STAMP(1);
    // (x, y, z) of this thread's bins f = tid, tid + kPeakBlock, ...: two divisions once, increments afterwards (the
    // three passes used to spend most of their instructions on % and / by run-time extents)
    const int LyLz = Ly * Lz;
    const int stepX = kPeakBlock / LyLz, stepY = (kPeakBlock % LyLz) / Lz, stepZ = kPeakBlock % Lz;
    const int x0 = tid / LyLz, y0 = (tid % LyLz) / Lz, z0 = tid % Lz;
#define ICPFLOW_PEAK_ADVANCE()                                   \
    do {                                                         \
        z += stepZ; y += stepY; x += stepX;                      \
        if (z >= Lz) { z -= Lz; ++y; }                           \
        if (y >= Ly) { y -= Ly; ++x; }                           \
    } while (0)
    // pass z: A = max over |dz| <= r of h   (-inf padding == ignore out of range)
    {
        int x = x0, y = y0, z = z0;
        for (int f = tid; f < L; f += kPeakBlock) {
            (void)x;
            uint32_t m = 0;
            if (MEM && radius >= 2 && Lz == 3) m = max(max(H0[f - z], H0[f - z + 1]), H0[f - z + 2]);   // (the whole z column)
            else {
                const int lo = max(0, z - radius), hi = min(Lz - 1, z + radius);
                for (int q = lo; q <= hi; ++q) m = max(m, MEM ? H0[f + (q - z)] : (uint32_t)h[f + (q - z)]);
            }
            A[f] = m;
            ICPFLOW_PEAK_ADVANCE();
        }
    }
    __syncthreads();
    PEAK_STAMP(2);
    // pass y: B = max over |dy| <= r of A
    {
        int x = x0, y = y0, z = z0;
        for (int f = tid; f < L; f += kPeakBlock) {
            (void)x;
            uint32_t m = 0;
            if (radius == 5) m = peak_window_max<5>(A, f, y, Ly, Lz);
            else {
                const int lo = max(0, y - radius), hi = min(Ly - 1, y + radius);
                for (int q = lo; q <= hi; ++q) m = max(m, A[f + (q - y) * Lz]);
            }
            Bv[f] = m;
            ICPFLOW_PEAK_ADVANCE();
        }
    }
    __syncthreads();
    PEAK_STAMP(3);
    // pass x: A = max over |dx| <= r of B  -> full 3-D window maximum
    {
        int x = x0, y = y0, z = z0;
        for (int f = tid; f < L; f += kPeakBlock) {
            uint32_t m = 0;
            if (radius == 5) m = peak_window_max<5>(Bv, f, x, Lx, LyLz);
            else {
                const int lo = max(0, x - radius), hi = min(Lx - 1, x + radius);
                for (int q = lo; q <= hi; ++q) m = max(m, Bv[f + (q - x) * LyLz]);
            }
            A[f] = m;
            ICPFLOW_PEAK_ADVANCE();
        }
    }
#undef ICPFLOW_PEAK_ADVANCE
    __syncthreads();
    PEAK_STAMP(4);
    // surviving vote = h where h == window max, else 0 (utils_hist.py:25-26); selection order (vote desc, flat index
    // asc) on the key (vote << 32 | ~index).  Every wave first finds ITS k largest keys with wave reductions only (the
    // global top k is a subset of the union of the waves' top k); one barrier; wave 0 then picks the k largest of those
    // 16 k keys -- one barrier instead of two per round.  A thread's keys are distinct, so when its best one was taken
    // its next one is the largest below it.
    const int lane = tid & (kWave - 1), wave = tid >> 6;
    unsigned long long *wtop = reinterpret_cast<unsigned long long *>(red);   // [waves][kPeakMaxK], see the declaration
    {
        unsigned long long below = ~0ull, best = 0ull;
        bool stale = true;
        for (int r = 0; r < k; ++r) {
            if (stale) {
                best = 0ull;
                for (int f = tid; f < L; f += kPeakBlock) {
                    const uint32_t v = MEM ? H0[f] : (uint32_t)h[f];
                    const uint32_t sv = (v == A[f]) ? v : 0u;
                    const unsigned long long key =
                        ((unsigned long long)sv << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)f);
                    if (key < below && key > best) best = key;
                }
                stale = false;   // (0 can only be "nothing left": the index bits of a real bin are never all zero)
            }
            const unsigned long long w = wave_max_u64_dpp(best);
            if (lane == 0) wtop[wave * kPeakMaxK + r] = w;
            if (w == best && w != 0ull) { below = best; stale = true; }   // mine was taken: look for my next one
        }
    }
    __syncthreads();
    PEAK_STAMP(5);
    if (wave == 0) {
        constexpr int kWaves = kPeakBlock / kWave;
        // candidate c = lane, lane + 64, ... of the kWaves * k keys (at most two per lane)
        unsigned long long c0 = 0ull, c1 = 0ull;
        {
            const int n = kWaves * k;
            if (lane < n) c0 = wtop[(lane / k) * kPeakMaxK + lane % k];
            if (lane + kWave < n) c1 = wtop[((lane + kWave) / k) * kPeakMaxK + (lane + kWave) % k];
        }
        for (int r = 0; r < k; ++r) {
            const unsigned long long m = wave_max_u64_dpp(c0 > c1 ? c0 : c1);
            if (lane == 0) chosen[r] = m;
            if (c0 == m) c0 = 0ull;
            if (c1 == m) c1 = 0ull;
        }
    }
    __syncthreads();
    PEAK_STAMP(6);
    // outputs: thread r writes peak r (and its candidate translation on the fused path)
    if (tid < k) {
        const int r = tid;
        const unsigned long long m = chosen[r];
        votes[(size_t)b * k + r] = (float)(uint32_t)(m >> 32);
        const uint32_t f = 0xFFFFFFFFu - (uint32_t)(m & 0xFFFFFFFFull);
        idx_out[(size_t)b * k + r] = (int64_t)f;
        if (dec.cand != nullptr) {
            // fused path: flat peak index -> candidate translation (left bin edges + shift,
            // utils_hist.py:78); the zero translation goes LAST (:83)
            float *o = dec.cand + ((size_t)b * (k + 1) + r) * 3;
            const int ix = (int)(f / Lz / Ly % Lx), iy = (int)(f / Lz % Ly), iz = (int)(f % Lz);
            o[0] = dec.ex[ix] + dec.shift; o[1] = dec.ey[iy] + dec.shift; o[2] = dec.ez[iz] + dec.shift;
            if (r == k - 1) { o[3] = 0.f; o[4] = 0.f; o[5] = 0.f; }
        }
    }
}

template <typename BinT>
static hipError_t launch_peaks_t(const BinT *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, PeakDecode dec, hipStream_t s)
{
    const size_t lds = 3 * sizeof(uint32_t) * (size_t)Lx * Ly * Lz;
    if (lds <= 150 * 1024) {
        static std::atomic<unsigned long long> attr{0ull};   // dynamic LDS above 64 KiB needs the attribute once per kernel and device
        ensure_dynamic_lds(reinterpret_cast<const void *>(&hist_peaks_kernel<BinT, 1>), 150 * 1024, &attr);
        hipLaunchKernelGGL((hist_peaks_kernel<BinT, 1>), dim3(B), dim3(kPeakBlock), lds, s, bins, Lx, Ly, Lz, k,
                           (kernel_size - 1) / 2, wsA, wsB, votes, idx, dec);
    } else {
        hipLaunchKernelGGL((hist_peaks_kernel<BinT, 0>), dim3(B), dim3(kPeakBlock), 0, s, bins, Lx, Ly, Lz, k,
                           (kernel_size - 1) / 2, wsA, wsB, votes, idx, dec);
    }
    return hipGetLastError();
}

hipError_t launch_hist_peaks_f32(const float *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s)
{
    return launch_peaks_t<float>(bins, B, Lx, Ly, Lz, k, kernel_size, wsA, wsB, votes, idx, PeakDecode{}, s);
}

hipError_t launch_hist_peaks_u32(const uint32_t *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s, PeakDecode dec)
{
    return launch_peaks_t<uint32_t>(bins, B, Lx, Ly, Lz, k, kernel_size, wsA, wsB, votes, idx, dec, s);
}

}  // namespace icpflow

#ifdef ICPFLOW_PEAK_CLOCK
extern "C" int icpflow_debug_peak_clock(unsigned long long *out)
{
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(icpflow::g_peakClock), sizeof(icpflow::g_peakClock));
}
#endif
