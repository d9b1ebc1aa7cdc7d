// table.hip -- the cluster table of a labelled cloud in one chain of launches (the host association
// around the registration path).
//
// Reference: match_pcds builds its candidate lists from torch.unique(labels) and, per candidate pair, boolean masks
// over ALL points (utils_match.py:24-66, 81-91; utils_check.py:21-49 reads centroid and bounding box of every cluster
// through device scalars).  Here: rows sorted by label (stable: the rows of a cluster keep their order, which the
// reference's random subsample of over-long clusters indexes into, utils_helper.py:198-201), the distinct labels with
// their row ranges, and per cluster the centroid and sorted bounding-box extents (utils_check.py:34-43,
// get_bbox_tensor utils_helper.py:166-170) -- a stable counting sort by label (dictionary, counts, scan, scatter) and the
// statistics kernel, five launches back to back on the caller's stream, where a chain of ~25 small ATen kernels (argsort,
// unique_consecutive, cumsum, casts, cat) ran before.
#include <cstring>

#include "common.hpp"
#include "kernels.hpp"

namespace icpflow {
namespace {

constexpr int kRowsBlock = 1024;
constexpr int kTableCols = 9;   // label, count, start, mean (3), sorted bbox extents (3)

// float -> uint32 whose unsigned order is the float order (-0.0 < +0.0 as bit patterns; labels are never -0.0 in
// practice and torch.argsort would keep them adjacent as equals -- they stay distinct clusters here only if the
// caller really passes both)
__device__ __forceinline__ uint32_t sortable(float f)
{
    const uint32_t u = (uint32_t)__float_as_int(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unsortable(uint32_t k)
{
    const uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __int_as_float((int)u);
}

// One chain serves one labelled cloud or the TWO clouds of a frame pair at once (the kernels take the cloud from the block
// index): half as many launches on the path of a frame pair, where every launch is a few microseconds of work behind a dispatch.
//
// Round 5: a stable COUNTING sort by label in four launches instead of a radix sort of (label, row) pairs in thirteen
// (rocprim::radix_sort_pairs: ~70 us of GPU time and ~40 us of host enqueue per frame pair, two thirds of the chain).  A
// frame has at most a few hundred distinct labels (Lmax <= 4096 rows in the table; more: the table reports overflow, as
// before), so:
//   1. table_dict_kernel    one workgroup per cloud: the distinct labels through a hash set in LDS, sorted -> dict, number
//   2. table_count_kernel   one workgroup per chunk of kChunkRows rows: row -> index of its label in dict (binary search in
//                           LDS), counts per (chunk, label)
//   3. table_scan_kernel    one workgroup per cloud: per label the running offsets over the chunks, the exclusive scan over
//                           the labels -> (label, count, start) rows of the table, first position of every (chunk, label)
//   4. table_scatter_kernel one WAVE per chunk: rows to their places in chunk order and, inside a chunk, in row order
//                           (ranks among the lanes of equal label by ballots: stable, deterministic)
// then the statistics kernel as before.
struct TableSides {
    const float *points[2];
    const float *labels[2];
    int M[2];              // rows of each cloud (M[1] = 0: one cloud)
    int64_t *order[2];
    double *table[2];
    int32_t *num[2];
    uint32_t *dict[2];     // [Lmax] distinct labels (sortable bits), ascending
    int *start[2];         // [Lmax] first position of every label
    uint16_t *rowOf;       // [M0 + M1] index of the row's label in dict
    int *counts;           // [chunks of both clouds][Lmax]
    int chunks0;           // chunks of cloud 0 (the chunks of cloud 1 follow)
};

constexpr int kChunkRows = 512;
constexpr int kDictSlots = 8192, kDictMax = 4096;
constexpr uint32_t kEmpty = 0xffffffffu;

__device__ __forceinline__ uint32_t label_key(float f)
{
    const uint32_t k = sortable(f);
    return k == kEmpty ? kEmpty - 1u : k;    // (one NaN payload shares the empty mark's pattern)
}

__global__ __launch_bounds__(1024) void table_dict_kernel(TableSides t, int Lmax)
{
    __shared__ uint32_t slot[kDictSlots];
    __shared__ uint32_t list[kDictMax];
    __shared__ int nFound, nListed;
    const int side = blockIdx.x, tid = threadIdx.x, M = t.M[side];
    const float *labels = t.labels[side];
    for (int k = tid; k < kDictSlots; k += 1024) slot[k] = kEmpty;
    if (tid == 0) { nFound = 0; nListed = 0; }
    __syncthreads();
    uint32_t last = kEmpty;
    constexpr int kPer = 8;                       // rows per thread and round: the loads of a round are in flight together
    for (int i0 = 0; i0 < M; i0 += 1024 * kPer) {
        float v[kPer];
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const int i = i0 + u * 1024 + tid;
            v[u] = labels[min(i, M - 1)];         // (clamped: the last row once more)
        }
#pragma unroll
        for (int u = 0; u < kPer; ++u) {
            const uint32_t k = label_key(v[u]);
            if (k == last) continue;              // (rows 1024 apart often share their label: ground, noise)
            last = k;
            uint32_t h = (k * 2654435761u) >> 19; // 13 bits
            for (int probe = 0; probe < kDictSlots; ++probe) {
                const uint32_t seen = slot[h];    // (a plain read first: most rows find their label already there)
                if (seen == k) break;
                if (seen == kEmpty) {
                    const uint32_t old = atomicCAS(&slot[h], kEmpty, k);
                    if (old == k) break;
                    if (old == kEmpty) { atomicAdd(&nFound, 1); break; }
                }
                h = (h + 1) & (kDictSlots - 1);
                if (nFound > kDictMax) break;     // (more labels than any table holds: the count is all that is reported)
            }
        }
    }
    __syncthreads();
    const int found = nFound;
    if (found > Lmax) {
        if (tid == 0) *t.num[side] = -found;      // < 0: more clusters than the table holds (a lower bound beyond kDictMax)
        return;
    }
    for (int k = tid; k < kDictSlots; k += 1024)
        if (slot[k] != kEmpty) list[atomicAdd(&nListed, 1)] = slot[k];
    __syncthreads();
    if (found <= 1024) {
        // few labels (a frame has a few hundred): every label counts the smaller ones -- its place in the sorted order -- with
        // all threads reading the same word per step (a broadcast), instead of ~40 barrier-separated stages of a sorting network
        const uint32_t mine = tid < found ? list[tid] : kEmpty;
        int rank = 0;
        for (int j = 0; j < found; ++j) rank += list[j] < mine ? 1 : 0;
        if (tid < found) t.dict[side][rank] = mine;
        if (tid == 0) *t.num[side] = found;
        return;
    }
    int P = 1;
    while (P < found) P <<= 1;
    for (int k = found + tid; k < P; k += 1024) list[k] = kEmpty;
    __syncthreads();
    for (int len = 2; len <= P; len <<= 1)
        for (int stride = len >> 1; stride > 0; stride >>= 1) {
            for (int u = tid; u < P / 2; u += 1024) {
                const int lo = (u / stride) * 2 * stride + (u % stride), hi = lo + stride;
                const bool up = ((lo & len) == 0);
                const uint32_t a = list[lo], b = list[hi];
                if ((a > b) == up) { list[lo] = b; list[hi] = a; }
            }
            __syncthreads();
        }
    for (int k = tid; k < found; k += 1024) t.dict[side][k] = list[k];
    if (tid == 0) *t.num[side] = found;
}

__global__ __launch_bounds__(256) void table_count_kernel(TableSides t, int Lmax)
{
    __shared__ uint32_t dict[kDictMax];
    __shared__ int cnt[kDictMax];
    const int side = (int)blockIdx.x >= t.chunks0 ? 1 : 0, chunk = blockIdx.x - (side ? t.chunks0 : 0);
    const int n = *t.num[side];
    if (n <= 0) return;
    const int tid = threadIdx.x, M = t.M[side], base = side ? t.M[0] : 0;
    for (int k = tid; k < n; k += 256) { dict[k] = t.dict[side][k]; cnt[k] = 0; }
    __syncthreads();
    uint32_t lastK = kEmpty;
    int lastR = 0;
    for (int j = tid; j < kChunkRows; j += 256) {
        const int i = chunk * kChunkRows + j;
        if (i >= M) break;
        const uint32_t k = label_key(t.labels[side][i]);
        int r = lastR;
        if (k != lastK) {
            int lo = 0, hi = n - 1;               // (the key is in the dictionary)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (dict[mid] < k) lo = mid + 1; else hi = mid;
            }
            r = lo; lastK = k; lastR = r;
        }
        atomicAdd(&cnt[r], 1);
        t.rowOf[base + i] = (uint16_t)r;
    }
    __syncthreads();
    int *out = t.counts + (size_t)blockIdx.x * Lmax;
    for (int k = tid; k < n; k += 256) out[k] = cnt[k];
}

__global__ __launch_bounds__(1024) void table_scan_kernel(TableSides t, int Lmax)
{
    __shared__ int part[1024 / kWave];
    __shared__ int carrySh;
    const int side = blockIdx.x, tid = threadIdx.x;
    const int n = *t.num[side];
    if (n <= 0) return;
    const int chunks = side ? ((t.M[1] + kChunkRows - 1) / kChunkRows) : t.chunks0;
    int *counts = t.counts + (size_t)(side ? t.chunks0 : 0) * Lmax;
    double *table = t.table[side];
    if (tid == 0) carrySh = 0;
    __syncthreads();
    for (int r0 = 0; r0 < n; r0 += 1024) {        // labels in rounds of 1024 (one round for the tables of a frame)
        const int r = r0 + tid;
        int total = 0;
        if (r < n)
            for (int c0 = 0; c0 < chunks; c0 += 16) {   // the label's rows before chunk c (sixteen loads in flight at a time)
                int v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = counts[(size_t)min(c0 + u, chunks - 1) * Lmax + r];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (c0 + u < chunks) {
                        counts[(size_t)(c0 + u) * Lmax + r] = total;
                        total += v[u];
                    }
            }
        // exclusive scan of the totals over the labels (ascending): where the label's rows start
        int incl = total;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
            const int up = __shfl_up(incl, o, kWave);
            if ((tid & (kWave - 1)) >= o) incl += up;
        }
        if ((tid & (kWave - 1)) == kWave - 1) part[tid >> 6] = incl;
        __syncthreads();
        int before = carrySh;
        for (int w = 0; w < (tid >> 6); ++w) before += part[w];
        const int start = before + incl - total;
        if (r < n) {
            t.start[side][r] = start;
            double *row = table + (size_t)r * kTableCols;
            row[0] = (double)unsortable(t.dict[side][r]);
            row[1] = (double)total;
            row[2] = (double)start;
        }
        __syncthreads();
        if (tid == 1023) carrySh = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(kWave) void table_scatter_kernel(TableSides t, int Lmax)
{
    __shared__ int cur[kDictMax];
    const int side = (int)blockIdx.x >= t.chunks0 ? 1 : 0, chunk = blockIdx.x - (side ? t.chunks0 : 0);
    const int n = *t.num[side];
    if (n <= 0) return;
    const int lane = threadIdx.x, M = t.M[side], base = side ? t.M[0] : 0;
    const int *mine = t.counts + (size_t)blockIdx.x * Lmax;
    for (int k = lane; k < n; k += kWave) cur[k] = mine[k] + t.start[side][k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    int64_t *order = t.order[side];
    const unsigned long long below = (1ull << lane) - 1ull;
    int rr[kChunkRows / kWave];
#pragma unroll
    for (int u = 0; u < kChunkRows / kWave; ++u) {
        const int i = chunk * kChunkRows + u * kWave + lane;
        rr[u] = i < M ? (int)t.rowOf[base + i] : -1;
    }
#pragma unroll
    for (int u = 0; u < kChunkRows / kWave; ++u) {
        const int i = chunk * kChunkRows + u * kWave + lane;
        const bool valid = i < M;
        const int r = rr[u];
        unsigned long long todo = __ballot(valid);
        while (todo != 0ull) {                    // one label of the round at a time, in lane order inside it
            const int leader = __builtin_ctzll(todo);
            const int rl = __builtin_amdgcn_readlane(r, leader);
            const unsigned long long m = __ballot(valid && r == rl);
            const int first = cur[rl];
            if (valid && r == rl) order[first + __popcll(m & below)] = (int64_t)i;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            if (lane == leader) cur[rl] = first + __popcll(m);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            todo &= ~m;
        }
    }
}

// one workgroup per cluster: centroid (fp64 sums) and sorted bounding-box extents; clusters with a negative label
// (ground, noise: never candidates, utils_check.py:32 -- and by far the largest "clusters" of a frame) report zeros
__global__ __launch_bounds__(kRowsBlock) void table_stats_kernel(TableSides t)
{
    __shared__ double ssum[kRowsBlock / kWave][3];
    __shared__ float smin[kRowsBlock / kWave][3], smax[kRowsBlock / kWave][3];
    const int c = blockIdx.x, side = blockIdx.y;
    const float *points = t.points[side];
    const int64_t *order = t.order[side];
    double *table = t.table[side];
    const int n_clusters = *t.num[side];
    if (c >= n_clusters) return;   // (also when the table overflowed: num < 0)
    double *row = table + (size_t)c * kTableCols;
    const int64_t n = (int64_t)row[1], s0 = (int64_t)row[2];
    if (row[0] < 0.0) {
        if (threadIdx.x < 6) row[3 + threadIdx.x] = 0.0;
        return;
    }
    double sum[3] = {0.0, 0.0, 0.0};
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    // (four rows of a thread at a time: their loads -- the position, then the point it names -- are in flight together; the sums
    // are added in the rows' order, as a loop over single rows adds them)
    for (int64_t i0 = threadIdx.x; i0 < n; i0 += 4 * kRowsBlock) {
        int64_t r[4];
        float v[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) r[u] = order[s0 + min(i0 + (int64_t)u * kRowsBlock, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[u][k] = points[r[u] * 3 + k];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + (int64_t)u * kRowsBlock < n)
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sum[k] += (double)v[u][k];
                    mn[k] = fminf(mn[k], v[u][k]);
                    mx[k] = fmaxf(mx[k], v[u][k]);
                }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        sum[k] = wave_sum(sum[k]);
        for (int o = 32; o > 0; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o));
        }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & (kWave - 1)) == 0)
        for (int k = 0; k < 3; ++k) { ssum[wave][k] = sum[k]; smin[wave][k] = mn[k]; smax[wave][k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float e[3];
        for (int k = 0; k < 3; ++k) {
            double t = ssum[0][k];
            float lo = smin[0][k], hi = smax[0][k];
            for (int w = 1; w < kRowsBlock / kWave; ++w) {
                t += ssum[w][k];
                lo = fminf(lo, smin[w][k]);
                hi = fmaxf(hi, smax[w][k]);
            }
            row[3 + k] = (double)(float)(t / (double)n);     // (the float32 value cluster_stats_kernel reports)
            e[k] = fabsf(hi - lo);
        }
        if (e[0] > e[1]) { const float t = e[0]; e[0] = e[1]; e[1] = t; }
        if (e[1] > e[2]) { const float t = e[1]; e[1] = e[2]; e[2] = t; }
        if (e[0] > e[1]) { const float t = e[0]; e[0] = e[1]; e[1] = t; }
        for (int k = 0; k < 3; ++k) row[6 + k] = (double)e[k];
    }
}

struct TableCarve {
    uint32_t *dict[2];
    int *start[2];
    uint16_t *rowOf;
    int *counts;
    size_t total;
};

inline int table_chunks(int M) { return (M + kChunkRows - 1) / kChunkRows; }

void table_carve(int MA, int MB, int Lmax, void *ws, TableCarve *c)
{
    char *p = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return q;
    };
    for (int k = 0; k < 2; ++k) {
        c->dict[k] = (uint32_t *)take((size_t)Lmax * 4);
        c->start[k] = (int *)take((size_t)Lmax * 4);
    }
    c->rowOf = (uint16_t *)take(((size_t)MA + MB) * 2);
    c->counts = (int *)take((size_t)(table_chunks(MA) + table_chunks(MB)) * Lmax * 4);
    c->total = off;
}

hipError_t table_chain(TableSides t, int Lmax, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s)
{
    const int sides = t.M[1] > 0 ? 2 : 1;
    if (Lmax > kDictMax) return hipErrorInvalidValue;      // (api.hip refuses it before: Lmax <= 4096)
    TableCarve c{};
    table_carve(t.M[0], t.M[1], Lmax, ws, &c);
    *wsTooSmall = wsBytes < c.total;
    if (*wsTooSmall) return hipSuccess;
    for (int k = 0; k < 2; ++k) { t.dict[k] = c.dict[k]; t.start[k] = c.start[k]; }
    t.rowOf = c.rowOf;
    t.counts = c.counts;
    t.chunks0 = table_chunks(t.M[0]);
    const int chunks = t.chunks0 + (sides == 2 ? table_chunks(t.M[1]) : 0);
    table_dict_kernel<<<sides, 1024, 0, s>>>(t, Lmax);
    table_count_kernel<<<chunks, 256, 0, s>>>(t, Lmax);
    table_scan_kernel<<<sides, 1024, 0, s>>>(t, Lmax);
    table_scatter_kernel<<<chunks, kWave, 0, s>>>(t, Lmax);
    table_stats_kernel<<<dim3(Lmax, sides), kRowsBlock, 0, s>>>(t);
    return hipGetLastError();
}

}  // namespace

hipError_t cluster_table_workspace_bytes(int M, int Lmax, size_t *bytes)
{
    TableCarve c{};
    table_carve(M, 0, Lmax, nullptr, &c);
    *bytes = c.total;
    return hipSuccess;
}

hipError_t cluster_table_pair_workspace_bytes(int MA, int MB, int Lmax, size_t *bytes)
{
    TableCarve c{};
    table_carve(MA, MB, Lmax, nullptr, &c);
    *bytes = c.total;
    return hipSuccess;
}

hipError_t launch_cluster_table(const float *points, const float *labels, int M, int64_t *order, double *table, int Lmax,
                                int32_t *num, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s)
{
    TableSides t{};
    t.points[0] = points; t.labels[0] = labels; t.M[0] = M; t.order[0] = order; t.table[0] = table; t.num[0] = num;
    return table_chain(t, Lmax, ws, wsBytes, wsTooSmall, s);
}

hipError_t launch_cluster_table_pair(const float *pointsA, const float *labelsA, int MA, int64_t *orderA, double *tableA,
                                     int32_t *numA, const float *pointsB, const float *labelsB, int MB, int64_t *orderB,
                                     double *tableB, int32_t *numB, int Lmax, void *ws, size_t wsBytes, bool *wsTooSmall,
                                     hipStream_t s)
{
    TableSides t{};
    t.points[0] = pointsA; t.labels[0] = labelsA; t.M[0] = MA; t.order[0] = orderA; t.table[0] = tableA; t.num[0] = numA;
    t.points[1] = pointsB; t.labels[1] = labelsB; t.M[1] = MB; t.order[1] = orderB; t.table[1] = tableB; t.num[1] = numB;
    return table_chain(t, Lmax, ws, wsBytes, wsTooSmall, s);
}

}  // namespace icpflow
