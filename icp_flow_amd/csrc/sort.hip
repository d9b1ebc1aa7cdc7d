// sort.hip -- sorting LONG clouds (N > 4096) with several workgroups per cloud.
//
// The per-pair sorts of the path (by z for the vote, along the fixed cloud's longest axis for the
// sweeps) run as one bitonic network in the LDS of one workgroup (hist.hip: zsort_kernel, icp.hip:
// sort_clouds_kernel).  That is the right shape for vehicle-sized clusters, but a 16 384-key network
// has 105 stages of 8 dependent LDS exchanges per thread (~240 us) and a frame with one wall-sized
// cluster waits for it twice.  Here a long cloud is cut into chunks of 2048 keys; every chunk is sorted
// by its own workgroup (66 stages, one exchange per thread), and a second kernel gives every element
// its final rank = rank inside its chunk + the number of smaller elements in every other chunk (binary
// searches; ties broken by the original index, so the order is the same total order the single-workgroup
// network produces) and writes it straight into the consumer's layout.  Outputs are identical to the
// single-workgroup kernels'.
#include <hip/hip_runtime.h>

#include "common.hpp"
#include "scan.hpp"
#include "sortdir.hpp"
#include "kernels.hpp"
#include "votekey.hpp"

namespace icpflow {

constexpr int kCsChunk = 2048;   // keys per chunk (one 1024-thread workgroup, 16 KiB of LDS)
#ifndef ICPFLOW_CS_BLOCK
#define ICPFLOW_CS_BLOCK 1024
#endif
constexpr int kCsBlock = ICPFLOW_CS_BLOCK;   // threads of a chunk's sort
constexpr int kCmBlock = 1024;               // threads of the merge (one element each)

struct ChunkSortParams {
    int mode;                 // 0: by z (vote), 1: along the fixed cloud's longest axis (sweeps)
    const float4 *P, *Q;      // mode 0: the two clouds;  mode 1: X (src), Y (dst) as passed to the registration
    const int32_t *nP, *nQ;   // valid counts of P / Q (mode 1: lenX, lenY)
    const uint8_t *swap;      // mode 1
    const float *prePose;     // mode 1, optional [B,4,4]
    int N, NPc;               // rows per cloud, chunked length (multiple of kCsChunk)
    float *ckey;              // [B, 2, NPc] sorted keys of every chunk
    int *cidx;                // [B, 2, NPc] original rows
    // outputs
    float4 *outP, *outQ;      // mode 0: z-sorted rows;  mode 1: Xs (moving) / Ys (fixed): posed point + row
    float *Ysoa, *Xsoa;       // mode 1 (Xsoa optional)
    int32_t *axisOut;         // mode 1
    uint32_t *bins;           // mode 0: counters to clear (L per pair)
    int L;
    const float *ez;          // mode 0: z edges of the vote box (slab thickness of the composite key)
    int len_z;
    float *keyRec;            // mode 0: [B, kVoteKeyStride] key parameters, written here, read by the vote
    const float *boxes;       // [B, kPairBoxStride] bounding boxes by count_pair_kernel, or NULL (every workgroup reads its pair)
    int codeChosen;           // mode 1: axisOut[b] already holds the pair's key code (sort_code_kernel, sortdir.hpp)
};

// cloud roles of mode 1 exactly as sort_clouds_kernel resolves them
struct Roles {
    const float4 *cloud;
    int n;
    bool moving;
};

__device__ __forceinline__ Roles roles_of(const ChunkSortParams &p, int b, int which)
{
    Roles r;
    if (p.mode == 0) {
        r.cloud = (which == 0 ? p.P : p.Q) + (size_t)b * p.N;
        r.n = min((which == 0 ? p.nP : p.nQ)[b], p.N);
        r.moving = false;
    } else {
        const bool sw = p.swap != nullptr && p.swap[b] != 0;
        r.moving = which == 1;
        const bool takeY = r.moving ? sw : !sw;          // moving role = sw ? Y : X;  fixed role = sw ? X : Y
        r.cloud = (takeY ? p.Q : p.P) + (size_t)b * p.N;
        r.n = (takeY ? p.nQ : p.nP)[b];
    }
    return r;
}

// longest axis of the FIXED cloud of pair b (all threads of the block take part)
__device__ int fixed_axis(const ChunkSortParams &p, int b, float *bb, int *axisSh)
{
    const Roles f = roles_of(p, b, 0);
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    bbox_rows(f.cloud, f.n, false, mn, mx);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, kWave));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, kWave));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0)
        for (int k = 0; k < 3; ++k) { bb[wave * 6 + k] = mn[k]; bb[wave * 6 + 3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float e[3];
        for (int k = 0; k < 3; ++k) {
            float lo = bb[k], hi = bb[3 + k];
            for (int w = 1; w < kCsBlock / kWave; ++w) { lo = fminf(lo, bb[w * 6 + k]); hi = fmaxf(hi, bb[w * 6 + 3 + k]); }
            e[k] = hi - lo;
        }
        *axisSh = (e[0] >= e[1] && e[0] >= e[2]) ? 0 : (e[1] >= e[2] ? 1 : 2);   // same rule as sort_clouds_kernel
    }
    __syncthreads();
    return *axisSh;
}

__device__ __forceinline__ float sort_key(const ChunkSortParams &p, const Roles &r, int b, int axis, const VoteKey &vk,
                                          int j, float &px, float &py, float &pz)
{
    const float4 q = r.cloud[j];
    if (p.mode == 0) {
        px = q.x; py = q.y; pz = q.z;
        return q.w > 0.0f ? vote_key(vk, q.x, q.y, q.z) : kInf;
    }
    PointXf pre;
    pre.kind = (r.moving && p.prePose) ? XF_AFFINE : XF_NONE;
    pre.a = (r.moving && p.prePose) ? affine_from_pose(p.prePose + (size_t)b * 16) : affine_identity();
    xf_apply(pre, q.x, q.y, q.z, px, py, pz);
    float ux = 0.f, uy = 0.f;
    if (axis >= 3) sort_dir(axis, ux, uy);
    return sort_key_of(axis, ux, uy, px, py, pz);
}

// mode 1 on long clouds: the pair's key code (sortdir.hpp) -- the longest axis of the fixed cloud or a key that spreads it at least
// a tenth better --, one workgroup per pair in front of the chunk sorts (every chunk of either cloud has to use the same key)
__global__ __launch_bounds__(kCsBlock) void sort_code_kernel(ChunkSortParams p)
{
    __shared__ float bb[6 * (kCsBlock / kWave)];
    __shared__ float boxSh[6];
    __shared__ int axisSh;
    __shared__ unsigned int hist[3 * kSortDirBins], scoreSh[kSortCodes];
    const int b = blockIdx.x;
    const Roles f = roles_of(p, b, 0);
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    bbox_rows(f.cloud, f.n, false, mn, mx);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, kWave));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, kWave));
        }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0)
        for (int k = 0; k < 3; ++k) { bb[wave * 6 + k] = mn[k]; bb[wave * 6 + 3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float e[3];
        for (int k = 0; k < 3; ++k) {
            float lo = bb[k], hi = bb[3 + k];
            for (int w = 1; w < kCsBlock / kWave; ++w) { lo = fminf(lo, bb[w * 6 + k]); hi = fmaxf(hi, bb[w * 6 + 3 + k]); }
            e[k] = hi - lo; boxSh[k] = lo; boxSh[3 + k] = hi;
        }
        axisSh = (e[0] >= e[1] && e[0] >= e[2]) ? 0 : (e[1] >= e[2] ? 1 : 2);   // same rule as sort_clouds_kernel
    }
    __syncthreads();
    int code = axisSh;
    if (f.n >= kSortDirMinN && roles_of(p, b, 1).n >= kSortDirMinMoving) code = choose_sort_code<kCsBlock>(f.cloud, f.n, boxSh, code, hist, scoreSh, &axisSh);
    if (threadIdx.x == 0) p.axisOut[b] = code;
}

#ifdef ICPFLOW_SORT_CLOCK
__device__ unsigned long long g_sortClock[2][4096][6];
#define SORT_STAMP(k) do { if (threadIdx.x == 0 && slotId < 4096) g_sortClock[p.mode][slotId][k] = wall_clock64(); } while (0)
#else
#define SORT_STAMP(k) do { } while (0)
#endif

// grid (B, 2, NPc / kCsChunk)
__global__ __launch_bounds__(kCsBlock) void chunk_sort_kernel(ChunkSortParams p)
{
    __shared__ unsigned long long kv[kCsChunk];   // (sort key, row) pairs
    __shared__ float bb[6 * (kCsBlock / kWave)];
    __shared__ int axisSh;
    __shared__ float keySh[kVoteKeyStride];
    const int b = blockIdx.x, which = blockIdx.y, c = blockIdx.z;
    const Roles r = roles_of(p, b, which);
    const int base = c * kCsChunk;
#ifdef ICPFLOW_SORT_CLOCK
    const int slotId = (c * 2 + which) * gridDim.x + b;
    if (threadIdx.x == 0 && slotId < 4096) { for (int k = 1; k < 5; ++k) g_sortClock[p.mode][slotId][k] = 0ull; g_sortClock[p.mode][slotId][5] = (unsigned long long)r.n; }
#endif
    SORT_STAMP(0);
    if (p.mode == 0 && c == 0) {   // clear this pair's counters (each of its two clouds takes one half)
        const int half = (p.L + 1) / 2;
        uint32_t *h = p.bins + (size_t)b * p.L + (size_t)which * half;
        const int cnt = which == 0 ? half : p.L - half;
        for (int k = threadIdx.x; k < cnt; k += kCsBlock) h[k] = 0u;
    }
    if (base >= r.n && c != 0) return;   // (chunk 0 still publishes the pair's axis / key parameters)
    int axis = 0;
    VoteKey vk{};
    if (p.mode == 1 && p.codeChosen) {
        axis = p.axisOut[b];
        if (base >= r.n) return;
    } else if (p.mode == 1 && p.boxes != nullptr) {
        // (fixed role = swap ? X : Y = cloud A / C of count_pair; its rows below the count, flagged or not)
        const float *bx = p.boxes + (size_t)b * kPairBoxStride + ((p.swap != nullptr && p.swap[b] != 0) ? 0 : 12) + 6;
        const float e0 = bx[3] - bx[0], e1 = bx[4] - bx[1], e2 = bx[5] - bx[2];
        axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2);
        if (c == 0 && which == 0 && threadIdx.x == 0) p.axisOut[b] = axis;
        if (base >= r.n) return;
    } else if (p.mode == 0 && p.boxes != nullptr) {
        vk = vote_key_params_boxed(p.boxes + (size_t)b * kPairBoxStride, min(p.nP[b], p.N), min(p.nQ[b], p.N),
                                   p.ez[p.len_z - 1] - p.ez[0], keySh);
        if (c == 0 && which == 0 && threadIdx.x < kVoteKeyStride) p.keyRec[(size_t)b * kVoteKeyStride + threadIdx.x] = keySh[threadIdx.x];
        if (base >= r.n) return;
    } else if (p.mode == 1) {
        axis = fixed_axis(p, b, bb, &axisSh);
        if (c == 0 && which == 0 && threadIdx.x == 0) p.axisOut[b] = axis;
        if (base >= r.n) return;
    } else {
        vk = vote_key_params(p.P + (size_t)b * p.N, min(p.nP[b], p.N), p.Q + (size_t)b * p.N, min(p.nQ[b], p.N),
                             p.ez[p.len_z - 1] - p.ez[0], bb, keySh);
        if (c == 0 && which == 0 && threadIdx.x < kVoteKeyStride) p.keyRec[(size_t)b * kVoteKeyStride + threadIdx.x] = keySh[threadIdx.x];
        if (base >= r.n) return;
    }
    SORT_STAMP(1);
    // the network only has to hold THIS chunk's rows: the next power of two >= their number (a batch padded to its
    // longest cloud is mostly short clouds; the merge reads the first r.n entries of a cloud and nothing behind them)
    int np2 = kWave;
    while (np2 < min(kCsChunk, r.n - base)) np2 <<= 1;
    for (int j = threadIdx.x; j < np2; j += kCsBlock) {
        float k = kInf, px, py, pz;
        if (base + j < r.n) k = sort_key(p, r, b, axis, vk, base + j, px, py, pz);
        kv[j] = sort_pack(k, base + j);
    }
    __syncthreads();
    SORT_STAMP(2);
    bitonic_sort_lds(kv, np2);
    SORT_STAMP(3);
    float *ck = p.ckey + ((size_t)b * 2 + which) * p.NPc + base;
    int *ci = p.cidx + ((size_t)b * 2 + which) * p.NPc + base;
    for (int j = threadIdx.x; j < np2; j += kCsBlock) { const unsigned long long w = kv[j]; ck[j] = sort_key_of(w); ci[j] = sort_index_of(w); }
    SORT_STAMP(4);
}

// number of elements (k, i) of a sorted chunk with (k, i) < (key, id), lexicographic
__device__ __forceinline__ int count_less(const float *__restrict__ ck, const int *__restrict__ ci, int len, float key,
                                          int id)
{
    int lo = 0, hi = len;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const float k = ck[mid];
        const bool less = k < key || (k == key && ci[mid] < id);
        if (less) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// grid (B, 2, ceil(NPc / kCmBlock)): one element per thread
__global__ __launch_bounds__(kCmBlock) void chunk_merge_kernel(ChunkSortParams p)
{
    const int b = blockIdx.x, which = blockIdx.y;
    const int e = blockIdx.z * kCmBlock + threadIdx.x;
    const Roles r = roles_of(p, b, which);
    const int NP16 = (p.N + kChunk - 1) / kChunk * kChunk;
    float4 *out = (p.mode == 0) ? ((which == 0 ? p.outP : p.outQ) + (size_t)b * p.N)
                                : ((r.moving ? p.outP : p.outQ) + (size_t)b * p.N);
    float *soa = nullptr;
    if (p.mode == 1) soa = r.moving ? (p.Xsoa ? p.Xsoa + (size_t)b * 3 * NP16 : nullptr) : p.Ysoa + (size_t)b * 3 * NP16;
    if (e >= r.n) {   // beyond the valid rows: padding of the consumer's layout
        // (mode 0: the vote reads the first n rows of a z-sorted cloud and nothing behind them -- rows x[i], i < nx, tiles y[j0 + k],
        // k < tn <= ny - j0: hist_vote_sorted_kernel -- so the 41 MB of invalid rows a ragged batch of 128 pairs padded to 10^4 used
        // to receive per sort stay unwritten; -DICPFLOW_POISON_ZSORT_PADDING fills them with NaN for the tests' proof of that)
#ifdef ICPFLOW_POISON_ZSORT_PADDING
        if (p.mode == 0) { if (e < p.N) out[e] = make_float4(__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")); }
#else
        if (p.mode == 0) { }
#endif
        else if (soa != nullptr && e < NP16) { soa[e] = kInf; soa[NP16 + e] = kInf; soa[2 * NP16 + e] = kInf; }
        return;
    }
    const float *ckAll = p.ckey + ((size_t)b * 2 + which) * p.NPc;
    const int *ciAll = p.cidx + ((size_t)b * 2 + which) * p.NPc;
    const float key = ckAll[e];
    const int id = ciAll[e];
    const int c = e / kCsChunk;
    int rank = e - c * kCsChunk;
    const int nchunks = (r.n + kCsChunk - 1) / kCsChunk;
    for (int o = 0; o < nchunks; ++o) {
        if (o == c) continue;
        const int len = min(kCsChunk, r.n - o * kCsChunk);
        rank += count_less(ckAll + (size_t)o * kCsChunk, ciAll + (size_t)o * kCsChunk, len, key, id);
    }
    if (p.mode == 0) {
        // rows whose flag is not set carry +inf keys and sort behind the valid ones: emitted as invalid rows
        float4 o = make_float4(0.f, 0.f, kInf, kInf);
        if (key < kInf) { o = r.cloud[id]; o.w = key; }   // the sort key rides in w (votekey.hpp)
        out[rank] = o;
        return;
    }
    const int axis = p.axisOut[b];
    float px, py, pz;
    (void)sort_key(p, r, b, axis, VoteKey{}, id, px, py, pz);
    out[rank] = make_float4(px, py, pz, __int_as_float(id));
    if (soa != nullptr) { soa[rank] = px; soa[NP16 + rank] = py; soa[2 * NP16 + rank] = pz; }
}

static hipError_t run_chunk_sort(ChunkSortParams p, int B, hipStream_t s)
{
    const int NP16 = (p.N + kChunk - 1) / kChunk * kChunk;
    const int span = NP16 > p.NPc ? NP16 : p.NPc;
    if (p.mode == 1 && p.codeChosen) hipLaunchKernelGGL(sort_code_kernel, dim3(B), dim3(kCsBlock), 0, s, p);
    hipLaunchKernelGGL(chunk_sort_kernel, dim3(B, 2, p.NPc / kCsChunk), dim3(kCsBlock), 0, s, p);
    hipLaunchKernelGGL(chunk_merge_kernel, dim3(B, 2, (span + kCmBlock - 1) / kCmBlock), dim3(kCmBlock), 0, s, p);
    return hipGetLastError();
}

int chunk_sort_length(int N) { return (N + kCsChunk - 1) / kCsChunk * kCsChunk; }

hipError_t launch_zsort_chunked(const float *P, const float *Q, const int32_t *nP, const int32_t *nQ, int B, int N,
                                float *outP, float *outQ, uint32_t *bins, int L, float *ckey, int *cidx,
                                const float *ez, int len_z, float *keyRec, hipStream_t s, const float *boxes)
{
    ChunkSortParams p{};
    p.mode = 0; p.P = (const float4 *)P; p.Q = (const float4 *)Q; p.nP = nP; p.nQ = nQ; p.N = N;
    p.NPc = chunk_sort_length(N); p.ckey = ckey; p.cidx = cidx; p.outP = (float4 *)outP; p.outQ = (float4 *)outQ;
    p.bins = bins; p.L = L; p.ez = ez; p.len_z = len_z; p.keyRec = keyRec; p.boxes = boxes;
    return run_chunk_sort(p, B, s);
}

hipError_t launch_sort_clouds_chunked(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                                      const uint8_t *swap, const float *prePose, int B, int N, int32_t *axisOut,
                                      float *Xs, float *Ys, float *Ysoa, float *Xsoa, float *ckey, int *cidx,
                                      hipStream_t s, const float *boxes, int dirKeys)
{
    ChunkSortParams p{};
    p.mode = 1; p.P = (const float4 *)X; p.Q = (const float4 *)Y; p.nP = lenX; p.nQ = lenY; p.swap = swap;
    p.prePose = prePose; p.N = N; p.NPc = chunk_sort_length(N); p.ckey = ckey; p.cidx = cidx;
    p.outP = (float4 *)Xs; p.outQ = (float4 *)Ys; p.Ysoa = Ysoa; p.Xsoa = Xsoa; p.axisOut = axisOut; p.boxes = boxes;
    p.codeChosen = dirKeys ? 1 : 0;
    return run_chunk_sort(p, B, s);
}

}  // namespace icpflow

#ifdef ICPFLOW_SORT_CLOCK
extern "C" int icpflow_debug_sort_clock(unsigned long long *out)
{
    (void)hipDeviceSynchronize();
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(icpflow::g_sortClock), sizeof(icpflow::g_sortClock));
}
#endif
