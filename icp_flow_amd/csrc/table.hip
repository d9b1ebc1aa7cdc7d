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

__global__ __launch_bounds__(kTableBlock) void table_key_kernel(const float *__restrict__ labels, int M,
                                                                uint32_t *__restrict__ key, uint32_t *__restrict__ val,
                                                                int *__restrict__ counter)
{
    const int i = blockIdx.x * kTableBlock + threadIdx.x;
    if (i == 0) *counter = 0;
    if (i < M) { key[i] = sortable(labels[i]); val[i] = (uint32_t)i; }
}

__global__ __launch_bounds__(kTableBlock) void table_boundary_kernel(const uint32_t *__restrict__ key,
                                                                     const uint32_t *__restrict__ val, int M,
                                                                     int64_t *__restrict__ order, int *__restrict__ bnd,
                                                                     int *__restrict__ counter, int Lmax)
{
    const int i = blockIdx.x * kTableBlock + threadIdx.x;
    if (i >= M) return;
    order[i] = (int64_t)val[i];
    if (i == 0 || key[i] != key[i - 1]) {
        const int slot = atomicAdd(counter, 1);
        if (slot < Lmax) bnd[slot] = i;
    }
}

// one workgroup: the boundaries in ascending order -> (label, count, start) of every cluster
__global__ __launch_bounds__(kRowsBlock) void table_rows_kernel(const uint32_t *__restrict__ key, int M,
                                                                const int *__restrict__ bnd, const int *__restrict__ counter,
                                                                int Lmax, double *__restrict__ table, int32_t *__restrict__ num)
{
    extern __shared__ int sb[];
    const int found = *counter;
    const int n = min(found, Lmax);
    int P = 1;
    while (P < n) P <<= 1;
    for (int k = threadIdx.x; k < P; k += kRowsBlock) sb[k] = k < n ? bnd[k] : 0x7fffffff;
    __syncthreads();
    for (int len = 2; len <= P; len <<= 1)
        for (int stride = len >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < P / 2; t += kRowsBlock) {
                const int lo = (t / stride) * 2 * stride + (t % stride), hi = lo + stride;
                const bool up = ((lo & len) == 0);
                const int a = sb[lo], b = sb[hi];
                if ((a > b) == up) { sb[lo] = b; sb[hi] = a; }
            }
            __syncthreads();
        }
    for (int c = threadIdx.x; c < n; c += kRowsBlock) {
        const int start = sb[c], end = (c + 1 < n) ? sb[c + 1] : M;
        double *row = table + (size_t)c * kTableCols;
        row[0] = (double)unsortable(key[start]);
        row[1] = (double)(end - start);
        row[2] = (double)start;
    }
    if (threadIdx.x == 0) *num = found <= Lmax ? found : -found;   // < 0: more clusters than the table holds
}

// one workgroup per cluster: centroid (fp64 sums) and sorted bounding-box extents; clusters with a negative label
// (ground, noise: never candidates, utils_check.py:32 -- and by far the largest "clusters" of a frame) report zeros
__global__ __launch_bounds__(kRowsBlock) void table_stats_kernel(const float *__restrict__ points,
                                                                 const int64_t *__restrict__ order,
                                                                 const int32_t *__restrict__ num, double *__restrict__ table)
{
    __shared__ double ssum[kRowsBlock / kWave][3];
    __shared__ float smin[kRowsBlock / kWave][3], smax[kRowsBlock / kWave][3];
    const int c = blockIdx.x;
    const int n_clusters = *num;
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
    uint32_t *keyIn, *keyOut, *valIn, *valOut;
    int *bnd, *counter;
    void *sortTmp;
    size_t sortTmpBytes, total;
};

hipError_t table_carve(int M, int Lmax, void *ws, TableCarve *c, hipStream_t s)
{
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (uint32_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr,
                                             (uint32_t *)nullptr, (size_t)M, 0, 32, s);
    if (e != hipSuccess) return e;
    char *p = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += (bytes + 255) / 256 * 256;
        return q;
    };
    c->keyIn = (uint32_t *)take((size_t)M * 4);
    c->keyOut = (uint32_t *)take((size_t)M * 4);
    c->valIn = (uint32_t *)take((size_t)M * 4);
    c->valOut = (uint32_t *)take((size_t)M * 4);
    c->bnd = (int *)take((size_t)Lmax * 4);
    c->counter = (int *)take(256);
    c->sortTmp = take(tmp);
    c->sortTmpBytes = tmp;
    c->total = off;
    return hipSuccess;
}

}  // namespace

hipError_t cluster_table_workspace_bytes(int M, int Lmax, size_t *bytes)
{
    TableCarve c{};
    const hipError_t e = table_carve(M, Lmax, nullptr, &c, nullptr);
    *bytes = c.total;
    return e;
}

hipError_t launch_cluster_table(const float *points, const float *labels, int M, int64_t *order, double *table, int Lmax,
                                int32_t *num, void *ws, size_t wsBytes, bool *wsTooSmall, hipStream_t s)
{
    TableCarve c{};
    hipError_t e = table_carve(M, Lmax, ws, &c, s);
    if (e != hipSuccess) return e;
    *wsTooSmall = wsBytes < c.total;
    if (*wsTooSmall) return hipSuccess;
    const int blocks = (M + kTableBlock - 1) / kTableBlock;
    table_key_kernel<<<blocks, kTableBlock, 0, s>>>(labels, M, c.keyIn, c.valIn, c.counter);
    e = rocprim::radix_sort_pairs(c.sortTmp, c.sortTmpBytes, c.keyIn, c.keyOut, c.valIn, c.valOut, (size_t)M, 0, 32, s);
    if (e != hipSuccess) return e;
    table_boundary_kernel<<<blocks, kTableBlock, 0, s>>>(c.keyOut, c.valOut, M, order, c.bnd, c.counter, Lmax);
    int P = 1;
    while (P < Lmax) P <<= 1;
    table_rows_kernel<<<1, kRowsBlock, (size_t)P * sizeof(int), s>>>(c.keyOut, M, c.bnd, c.counter, Lmax, table, num);
    table_stats_kernel<<<Lmax, kRowsBlock, 0, s>>>(points, order, num, table);
    return hipGetLastError();
}

}  // namespace icpflow
