// table.hip -- the cluster table of a labelled cloud in one chain of launches (the host association
// around the registration path).
//
// Reference: match_pcds builds its candidate lists from torch.unique(labels) and, per candidate pair, boolean masks
// over ALL points (utils_match.py:24-66, 81-91; utils_check.py:21-49 reads centroid and bounding box of every cluster
// through device scalars).  Here: rows sorted by label (stable: the rows of a cluster keep their order, which the
// reference's random subsample of over-long clusters indexes into, utils_helper.py:198-201), the distinct labels with
// their row ranges, and per cluster the centroid and sorted bounding-box extents (utils_check.py:34-43,
// get_bbox_tensor utils_helper.py:166-170) -- key kernel, one radix sort, boundary kernel, row kernel, statistics
// kernel, back to back on the caller's stream, where a chain of ~25 small ATen kernels (argsort, unique_consecutive,
// cumsum, casts, cat) ran before.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "common.hpp"
#include "kernels.hpp"

namespace icpflow {
namespace {

constexpr int kTableBlock = 256;
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

// One chain serves one labelled cloud (32-bit keys) or the TWO clouds of a frame pair at once (64-bit keys: the cloud's
// number above the label's 32 bits, so that one sort leaves cloud 0's rows in front of cloud 1's): half as many launches
// on the path of a frame pair, where every launch of the chain is a few microseconds of work behind a dispatch.
struct TableSides {
    const float *points[2];
    const float *labels[2];
    int M[2];              // rows of each cloud (M[1] = 0: one cloud)
    int64_t *order[2];
    double *table[2];
    int32_t *num[2];
    int *bnd[2];
    int *counter;          // [2]
};

template <typename KeyT>
__global__ __launch_bounds__(kTableBlock) void table_key_kernel(TableSides t, KeyT *__restrict__ key,
                                                                uint32_t *__restrict__ val)
{
    const int i = blockIdx.x * kTableBlock + threadIdx.x;
    if (i < 2) t.counter[i] = 0;
    if (i >= t.M[0] + t.M[1]) return;
    const int side = i >= t.M[0] ? 1 : 0, local = i - (side ? t.M[0] : 0);
    KeyT k = (KeyT)sortable(t.labels[side][local]);
    if constexpr (sizeof(KeyT) == 8) k |= (KeyT)side << 32;
    key[i] = k;
    val[i] = (uint32_t)local;
}

template <typename KeyT>
__global__ __launch_bounds__(kTableBlock) void table_boundary_kernel(TableSides t, const KeyT *__restrict__ key,
                                                                     const uint32_t *__restrict__ val, int Lmax)
{
    const int i = blockIdx.x * kTableBlock + threadIdx.x;
    if (i >= t.M[0] + t.M[1]) return;
    const int side = i >= t.M[0] ? 1 : 0, local = i - (side ? t.M[0] : 0);
    t.order[side][local] = (int64_t)val[i];
    if (local == 0 || key[i] != key[i - 1]) {
        const int slot = atomicAdd(&t.counter[side], 1);
        if (slot < Lmax) t.bnd[side][slot] = local;
    }
}

// one workgroup per cloud: the boundaries in ascending order -> (label, count, start) of every cluster
template <typename KeyT>
__global__ __launch_bounds__(kRowsBlock) void table_rows_kernel(TableSides t, const KeyT *__restrict__ keyAll, int Lmax)
{
    extern __shared__ int sb[];
    const int side = blockIdx.x;
    const KeyT *key = keyAll + (side ? t.M[0] : 0);
    const int M = t.M[side];
    const int *bnd = t.bnd[side];
    double *table = t.table[side];
    const int found = t.counter[side];
    const int n = min(found, Lmax);
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = threadIdx.x; k < P; k += kRowsBlock) sb[k] = k < n ? bnd[k] : 0x7fffffff;
    __syncthreads();
    for (int len = 2; len <= P; len <<= 1)
        for (int stride = len >> 1; stride > 0; stride >>= 1) {
            for (int u = threadIdx.x; u < P / 2; u += kRowsBlock) {
                const int lo = (u / stride) * 2 * stride + (u % stride), hi = lo + stride;
                const bool up = ((lo & len) == 0);
                const int a = sb[lo], b = sb[hi];
                if ((a > b) == up) { sb[lo] = b; sb[hi] = a; }
            }
            __syncthreads();
        }
    for (int c = threadIdx.x; c < n; c += kRowsBlock) {
        const int start = sb[c], end = (c + 1 < n) ? sb[c + 1] : M;
        double *row = table + (size_t)c * kTableCols;
        row[0] = (double)unsortable((uint32_t)key[start]);
        row[1] = (double)(end - start);
        row[2] = (double)start;
    }
    if (threadIdx.x == 0) *t.num[side] = found <= Lmax ? found : -found;   // < 0: more clusters than the table holds
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
    for (int64_t i = threadIdx.x; i < n; i += kRowsBlock) {
        const int64_t r = order[s0 + i];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float v = points[r * 3 + k];
            sum[k] += (double)v;
            mn[k] = fminf(mn[k], v);
            mx[k] = fmaxf(mx[k], v);
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
    void *keyIn, *keyOut;
    uint32_t *valIn, *valOut;
    int *bnd[2], *counter;
    void *sortTmp;
    size_t sortTmpBytes, total;
};

template <typename KeyT>
hipError_t table_sort(void *tmp, size_t &tmpBytes, const TableCarve *c, size_t M, hipStream_t s)
{
    // (two clouds: 33 key bits -- the label's 32 and the cloud's number)
    return rocprim::radix_sort_pairs(tmp, tmpBytes, c ? (KeyT *)c->keyIn : (KeyT *)nullptr, c ? (KeyT *)c->keyOut : (KeyT *)nullptr,
                                     c ? c->valIn : (uint32_t *)nullptr, c ? c->valOut : (uint32_t *)nullptr, M, 0,
                                     sizeof(KeyT) == 8 ? 33 : 32, s);
}

template <typename KeyT>
hipError_t table_carve(int M, int Lmax, void *ws, TableCarve *c, hipStream_t s)
{
    size_t tmp = 0;
    hipError_t e = table_sort<KeyT>(nullptr, tmp, nullptr, (size_t)M, s);
    if (e != hipSuccess) return e;
    char *p = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return q;
    };
    c->keyIn = take((size_t)M * sizeof(KeyT));
    c->keyOut = take((size_t)M * sizeof(KeyT));
    c->valIn = (uint32_t *)take((size_t)M * 4);
    c->valOut = (uint32_t *)take((size_t)M * 4);
    c->bnd[0] = (int *)take((size_t)Lmax * 4);
    c->bnd[1] = (int *)take((size_t)Lmax * 4);
    c->counter = (int *)take(256);
    c->sortTmp = take(tmp);
    c->sortTmpBytes = tmp;
    c->total = off;
    return hipSuccess;
}

template <typename KeyT>
hipError_t table_chain(TableSides t, int Lmax, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s)
{
    const int M = t.M[0] + t.M[1], sides = t.M[1] > 0 ? 2 : 1;
    TableCarve c{};
    hipError_t e = table_carve<KeyT>(M, Lmax, ws, &c, s);
    if (e != hipSuccess) return e;
    *wsTooSmall = wsBytes < c.total;
    if (*wsTooSmall) return hipSuccess;
    t.bnd[0] = c.bnd[0];
    t.bnd[1] = c.bnd[1];
    t.counter = c.counter;
    const int blocks = (M + kTableBlock - 1) / kTableBlock;
    table_key_kernel<KeyT><<<blocks, kTableBlock, 0, s>>>(t, (KeyT *)c.keyIn, c.valIn);
    e = table_sort<KeyT>(c.sortTmp, c.sortTmpBytes, &c, (size_t)M, s);
    if (e != hipSuccess) return e;
    table_boundary_kernel<KeyT><<<blocks, kTableBlock, 0, s>>>(t, (const KeyT *)c.keyOut, c.valOut, Lmax);
    int P = 1;
    while (P < Lmax) P <<= 1;
    table_rows_kernel<KeyT><<<sides, kRowsBlock, (size_t)P * sizeof(int), s>>>(t, (const KeyT *)c.keyOut, Lmax);
    table_stats_kernel<<<dim3(Lmax, sides), kRowsBlock, 0, s>>>(t);
    return hipGetLastError();
}

}  // namespace

hipError_t cluster_table_workspace_bytes(int M, int Lmax, size_t *bytes)
{
    TableCarve c{};
    const hipError_t e = table_carve<uint32_t>(M, Lmax, nullptr, &c, nullptr);
    *bytes = c.total;
    return e;
}

hipError_t cluster_table_pair_workspace_bytes(int MA, int MB, int Lmax, size_t *bytes)
{
    TableCarve c{};
    const hipError_t e = table_carve<uint64_t>(MA + MB, Lmax, nullptr, &c, nullptr);
    *bytes = c.total;
    return e;
}

hipError_t launch_cluster_table(const float *points, const float *labels, int M, int64_t *order, double *table, int Lmax,
                                int32_t *num, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s)
{
    TableSides t{};
    t.points[0] = points; t.labels[0] = labels; t.M[0] = M; t.order[0] = order; t.table[0] = table; t.num[0] = num;
    return table_chain<uint32_t>(t, Lmax, ws, wsBytes, wsTooSmall, s);
}

hipError_t launch_cluster_table_pair(const float *pointsA, const float *labelsA, int MA, int64_t *orderA, double *tableA,
                                     int32_t *numA, const float *pointsB, const float *labelsB, int MB, int64_t *orderB,
                                     double *tableB, int32_t *numB, int Lmax, void *ws, size_t wsBytes, bool *wsTooSmall,
                                     hipStream_t s)
{
    TableSides t{};
    t.points[0] = pointsA; t.labels[0] = labelsA; t.M[0] = MA; t.order[0] = orderA; t.table[0] = tableA; t.num[0] = numA;
    t.points[1] = pointsB; t.labels[1] = labelsB; t.M[1] = MB; t.order[1] = orderB; t.table[1] = tableB; t.num[1] = numB;
    return table_chain<uint64_t>(t, Lmax, ws, wsBytes, wsTooSmall, s);
}

}  // namespace icpflow
