// cluster.hip -- density clustering of one frame pair's non-ground points.
//
// Replaces the `cluster_dbscan` branch of the reference's clustering (utils_cluster.py:32-48 ->
// open3d 0.17.0 PointCloud::cluster_dbscan, environment.yml:227; cluster_pcd utils_cluster.py:50-63).
// Open3D's routine: radius neighbours of every point (nanoflann, squared distance STRICTLY below
// eps^2, the point itself included, coordinates widened to double), a point is a core point when it
// has >= min_points neighbours, then clusters are grown in index order; a point that is not a core
// point takes the label of the first cluster that reaches it.  That outcome does not depend on the
// visiting order:
//   * clusters = connected components of the core points under "closer than eps",
//   * cluster ids = rank of the components by their smallest member index,
//   * a non-core point with core neighbours joins the lowest-ranked of their clusters, else noise (-1),
// which is what the kernels below compute, one WAVE per point over a uniform grid of cell size eps:
//   key -> radix sort (rocPRIM) -> per point the 9 runs of sorted points covering its 27 neighbour
//   cells -> core flags -> union-find over core-core edges (hook to the first neighbour, flatten, lock-free
//   unions for the edges still crossing trees) -> first row of every component -> non-core points ->
//   rank of the first rows (one scan) -> labels + cluster sizes.
// Distances are evaluated in fp64 on the fp32 coordinates, like Open3D's Vector3dVector copy.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.hpp"
#include "cluster_util.hpp"

namespace icpflow {

namespace {

constexpr int kRuns = 9;     // (dx, dy) columns of neighbour cells
constexpr int kBlock = 256;

__global__ __launch_bounds__(kBlock) void dbscan_key_kernel(const float *__restrict__ pts, int stride,
                                                            const uint8_t *__restrict__ mask, int n, double invCell,
                                                            unsigned long long *__restrict__ key,
                                                            int *__restrict__ val, int *__restrict__ firstRow,
                                                            int *__restrict__ rootOf, int *__restrict__ counts)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float x = pts[(size_t)i * stride], y = pts[(size_t)i * stride + 1], z = pts[(size_t)i * stride + 2];
    const bool live = (!mask || mask[i]) && isfinite(x) && isfinite(y) && isfinite(z);
    key[i] = live ? pack_key(cell_coord(x, invCell), cell_coord(y, invCell), cell_coord(z, invCell)) : kMaskedKey;
    val[i] = i;
    firstRow[i] = 0x7fffffff;
    rootOf[i] = (mask && !mask[i]) ? -2 : -1;   // -2: not part of the clustered subset, -1: noise so far
    counts[i] = 0;
}

// sorted copy of the points: (x, y, z, original index)
__global__ __launch_bounds__(kBlock) void dbscan_gather_kernel(const float *__restrict__ pts, int stride,
                                                               const int *__restrict__ val, int n,
                                                               float4 *__restrict__ sorted)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const int i = val[j];
    sorted[j] = make_float4(pts[(size_t)i * stride], pts[(size_t)i * stride + 1], pts[(size_t)i * stride + 2],
                            __int_as_float(i));
}

__device__ inline int lower_bound_key(const unsigned long long *__restrict__ key, int n, unsigned long long k)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (key[mid] < k) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}

__device__ inline bool closer(const float4 a, const float4 b, double eps2)
{
    const double dx = (double)a.x - (double)b.x, dy = (double)a.y - (double)b.y, dz = (double)a.z - (double)b.z;
    // no contraction: the sum is rounded like the host libraries round it; nanoflann's radius result set
    // keeps dist < radius (strict)
    return __dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz)) < eps2;
}

constexpr int kWavesPerBlock = kBlock / 64;

__device__ inline int wave_point(int n)   // the sorted point this wave works on (wave-uniform), or -1
{
    const int j = __builtin_amdgcn_readfirstlane(blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6));
    return j < n ? j : -1;
}

// One WAVE per sorted point.  Cells (cx+dx, cy+dy, cz-1 .. cz+1) are consecutive keys, so the 27 neighbour
// cells are 9 runs of sorted points: lanes 0..17 find the run ends by binary search, then the wave
// walks the runs 64 candidates at a time (coalesced float4 rows) counting neighbours until min_points.
__global__ __launch_bounds__(kBlock) void dbscan_core_kernel(const float4 *__restrict__ sorted,
                                                             const unsigned long long *__restrict__ key, int n,
                                                             double eps2, int minPoints, int2 *__restrict__ runs,
                                                             uint8_t *__restrict__ core)
{
    const int j = wave_point(n);
    if (j < 0) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long k = key[j];
    if (k == kMaskedKey) {
        if (lane < kRuns) runs[(size_t)lane * n + j] = make_int2(0, 0);
        if (lane == 0) core[j] = 0;
        return;
    }
    const unsigned long long fieldMask = (1ull << kCellBits) - 1;
    const long long cx = (long long)(k >> (2 * kCellBits)), cy = (long long)((k >> kCellBits) & fieldMask),
                    cz = (long long)(k & fieldMask);
    int bound = 0;
    if (lane < 2 * kRuns) {
        const int r = lane >> 1;
        const long long nx = cx + r / 3 - 1, ny = cy + r % 3 - 1;
        bound = (lane & 1) ? lower_bound_key(key, n, pack_key(nx, ny, cz + 1) + 1)
                           : lower_bound_key(key, n, pack_key(nx, ny, cz - 1));
    }
    const int hiOfLane = __shfl(bound, 2 * (lane % kRuns) + 1), loOfLane = __shfl(bound, 2 * (lane % kRuns));
    if (lane < kRuns) runs[(size_t)lane * n + j] = make_int2(loOfLane, hiOfLane);
    const float4 p = sorted[j];
    int cnt = 0;
    for (int k = 0; k < kRuns && cnt < minPoints; ++k) {
        const int r = (int)((0x862075314ull >> (4 * k)) & 15);   // own column first, corners last: exits sooner
        const int lo = __shfl(bound, 2 * r), hi = __shfl(bound, 2 * r + 1);
        for (int q0 = lo; q0 < hi && cnt < minPoints; q0 += 64) {
            const int q = q0 + lane;
            const bool in = q < hi && closer(p, sorted[q], eps2);
            cnt += __popcll(__ballot(in));
        }
    }
    if (lane == 0) core[j] = cnt >= minPoints ? 1 : 0;   // the point itself is one of its neighbours (distance 0)
}

// The union-find runs over SORTED positions (neighbours of a point are contiguous rows, so the parent
// reads of a candidate run are coalesced); three steps keep the atomics rare:
//   hook    every core point points at its first core neighbour in sorted order (no atomics, a forest),
//   flatten every point points at the root of its tree,
//   union   only edges whose endpoints still have different roots take the lock-free path.
__global__ __launch_bounds__(kBlock) void dbscan_hook_kernel(const float4 *__restrict__ sorted,
                                                             const int2 *__restrict__ runs,
                                                             const uint8_t *__restrict__ core, int n, double eps2,
                                                             int *__restrict__ parent)
{
    const int j = wave_point(n);
    if (j < 0) return;
    const int lane = threadIdx.x & 63;
    int first = j;
    if (core[j]) {
        const float4 p = sorted[j];
        bool found = false;
        for (int r = 0; r < kRuns && !found; ++r) {   // runs are visited in increasing key order
            const int2 run = runs[(size_t)r * n + j];
            const int hi = min(run.y, j);
            for (int q0 = run.x; q0 < hi && !found; q0 += 64) {
                const int q = q0 + lane;
                const bool in = q < hi && core[q] && closer(p, sorted[q], eps2);
                const unsigned long long hit = __ballot(in);
                if (hit) {
                    first = q0 + __builtin_ctzll(hit);
                    found = true;
                }
            }
        }
    }
    if (lane == 0) parent[j] = first;
}

// every point points at the root of its tree; per chunk of 64 sorted rows: the root shared by all its core
// points (-1: none at all, -2: several) -- the union step skips such chunks without reading them
__global__ __launch_bounds__(kBlock) void dbscan_flatten_kernel(int n, const uint8_t *__restrict__ core,
                                                                int *__restrict__ parent,
                                                                int *__restrict__ chunkRoot)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int root = -1;
    bool isCore = false;
    if (j < n) {
        root = uf_find(parent, j);
        parent[j] = root;
        isCore = core[j] != 0;
    }
    const unsigned long long cores = __ballot(isCore);
    if (blockIdx.x * kBlock + (threadIdx.x & ~63) < n) {
        int shared = -1;
        if (cores) {
            const int first = __shfl(root, __builtin_ctzll(cores));
            shared = __ballot(isCore && root != first) == 0 ? first : -2;
        }
        if (lane == 0) chunkRoot[j >> 6] = shared;
    }
}

__global__ __launch_bounds__(kBlock) void dbscan_union_kernel(const float4 *__restrict__ sorted,
                                                              const int2 *__restrict__ runs,
                                                              const uint8_t *__restrict__ core, int n, double eps2,
                                                              const int *__restrict__ chunkRoot, int *parent)
{
    const int j = wave_point(n);
    if (j < 0 || !core[j]) return;
    const int lane = threadIdx.x & 63;
    const float4 p = sorted[j];
    int mine = parent[j];   // root after the flatten step; refreshed when this wave merges trees
    const int mine0 = mine;
    for (int r = 0; r < kRuns; ++r) {
        const int2 run = runs[(size_t)r * n + j];
        const int hi = min(run.y, j);   // every core-core edge once, from its later endpoint
        for (int q0 = run.x & ~63; q0 < hi; q0 += 64) {   // chunk-aligned steps
            // after hook + flatten nearly every chunk around a point holds only its own tree (or no core point)
            const int shared = chunkRoot[q0 >> 6];
            if (shared == -1 || shared == mine0) continue;
            const int q = q0 + lane;
            const bool cross = q >= run.x && q < hi && core[q] && parent[q] != mine && closer(p, sorted[q], eps2);
            if (__ballot(cross) == 0) continue;
            if (cross) uf_union(parent, q, j);
            mine = __shfl(uf_find(parent, j), 0);
        }
    }
}

// smallest caller row of every component (Open3D numbers the clusters by it), one atomic per (wave, root)
__global__ __launch_bounds__(kBlock) void dbscan_first_row_kernel(const float4 *__restrict__ sorted,
                                                                  const uint8_t *__restrict__ core, int n,
                                                                  int *__restrict__ parent,
                                                                  int *__restrict__ firstRow)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int root = -1, row = 0x7fffffff;
    if (j < n && core[j]) {
        root = uf_find(parent, j);
        parent[j] = root;
        row = __float_as_int(sorted[j].w);
    }
    bool pending = root >= 0;
    unsigned long long todo;
    while ((todo = __ballot(pending)) != 0) {
        const int lead = __builtin_ctzll(todo);
        const int rv = __shfl(root, lead);
        const bool same = pending && root == rv;
        int m = same ? row : 0x7fffffff;
        for (int d = 32; d > 0; d >>= 1) m = min(m, __shfl_xor(m, d));
        if (lane == lead) atomicMin(firstRow + rv, m);
        if (same) pending = false;
    }
}

// rootOf[row] = first row of the cluster the point belongs to (core: its component; non-core: the
// smallest among its core neighbours = the first cluster Open3D's growth reaches it from), -1 = noise
__global__ __launch_bounds__(kBlock) void dbscan_assign_kernel(const float4 *__restrict__ sorted,
                                                               const unsigned long long *__restrict__ key,
                                                               const int2 *__restrict__ runs,
                                                               const uint8_t *__restrict__ core, int n, double eps2,
                                                               const int *__restrict__ parent,
                                                               const int *__restrict__ firstRow,
                                                               int *__restrict__ rootOf)
{
    const int j = wave_point(n);
    if (j < 0 || key[j] == kMaskedKey) return;
    const int lane = threadIdx.x & 63;
    const float4 p = sorted[j];
    const int me = __float_as_int(p.w);
    if (core[j]) {
        if (lane == 0) rootOf[me] = firstRow[parent[j]];
        return;
    }
    int best = 0x7fffffff;
    for (int r = 0; r < kRuns; ++r) {
        const int2 run = runs[(size_t)r * n + j];
        for (int q0 = run.x; q0 < run.y; q0 += 64) {
            const int q = q0 + lane;
            if (q >= run.y || !core[q]) continue;
            if (closer(p, sorted[q], eps2)) best = min(best, firstRow[parent[q]]);
        }
    }
    for (int d = 32; d > 0; d >>= 1) best = min(best, __shfl_xor(best, d));
    if (lane == 0) rootOf[me] = best == 0x7fffffff ? -1 : best;
}

// rank[i] = number of component roots below index i.  One workgroup of 16 waves, each wave owns a
// contiguous slice and walks it 64 rows at a time (n is a frame pair's point count).
__global__ __launch_bounds__(1024) void dbscan_rank_kernel(const int *__restrict__ rootOf, int n,
                                                           int *__restrict__ rank, int *__restrict__ numClusters)
{
    __shared__ int part[16];
    constexpr int kU = 8;   // independent loads in flight per lane
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int per = ((n + 15) / 16 + 64 * kU - 1) / (64 * kU) * (64 * kU);
    const int lo = min(n, w * per), hi = min(n, lo + per);
    int sum = 0;
    for (int i0 = lo; i0 < hi; i0 += 64 * kU) {
        int v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * 64 + lane;
            v[u] = i < hi ? rootOf[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) sum += __popcll(__ballot(v[u] == i0 + u * 64 + lane));
    }
    if (lane == 0) part[w] = sum;
    __syncthreads();
    int base = 0;
    for (int v = 0; v < w; ++v) base += part[v];
    for (int i0 = lo; i0 < hi; i0 += 64 * kU) {
        int v[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * 64 + lane;
            v[u] = i < hi ? rootOf[i] : -1;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const int i = i0 + u * 64 + lane;
            const unsigned long long b = __ballot(v[u] == i);
            if (i < hi) rank[i] = base + __popcll(b & ((1ull << lane) - 1));
            base += __popcll(b);
        }
    }
    if (threadIdx.x == 1023) *numClusters = base;
}

// labels in the caller's row order; cluster sizes by one atomic per (wave, distinct label): rows are
// visited in sorted (spatial) order, where a wave sees very few distinct clusters
__global__ __launch_bounds__(kBlock) void dbscan_label_kernel(const float4 *__restrict__ sorted,
                                                              const int *__restrict__ rootOf,
                                                              const int *__restrict__ rank, int n,
                                                              int32_t *__restrict__ labels, int *__restrict__ counts)
{
    const int j = blockIdx.x * kBlock + threadIdx.x;
    const int lane = threadIdx.x & 63;
    int l = -3;
    if (j < n) {
        const int i = __float_as_int(sorted[j].w);
        const int r = rootOf[i];
        l = r >= 0 ? rank[r] : r;
        labels[i] = l;
    }
    bool pending = l >= 0;
    unsigned long long todo;
    while ((todo = __ballot(pending)) != 0) {
        const int first = __builtin_ctzll(todo);
        const int lv = __shfl(l, first);
        const unsigned long long same = __ballot(pending && l == lv);
        if (lane == first) atomicAdd(counts + lv, (int)__popcll(same));
        if (l == lv) pending = false;
    }
}

struct Carve {
    unsigned long long *keyIn, *keyOut;
    int *valIn, *valOut, *parent, *firstRow, *rootOf, *rank, *chunkRoot;
    float4 *sorted;
    int2 *runs;
    uint8_t *core;
    void *sortTmp;
    size_t sortTmpBytes, total;
};

hipError_t carve(int n, void *ws, Carve *c, hipStream_t s)
{
    size_t tmp = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, tmp, (unsigned long long *)nullptr,
                                             (unsigned long long *)nullptr, (int *)nullptr, (int *)nullptr,
                                             (size_t)n, 0, 63, s);
    if (e != hipSuccess) return e;
    char *p = (char *)ws;
    size_t off = 0;
    auto take = [&](size_t bytes) {
        char *q = p ? p + off : nullptr;
        off += up256(bytes);
        return q;
    };
    const size_t N = (size_t)n;
    c->keyIn = (unsigned long long *)take(N * 8);
    c->keyOut = (unsigned long long *)take(N * 8);
    c->valIn = (int *)take(N * 4);
    c->valOut = (int *)take(N * 4);
    c->parent = (int *)take(N * 4);
    c->firstRow = (int *)take(N * 4);
    c->chunkRoot = (int *)take(((N + 63) / 64) * 4);
    c->rootOf = (int *)take(N * 4);
    c->rank = (int *)take(N * 4);
    c->sorted = (float4 *)take(N * 16);
    c->runs = (int2 *)take(N * kRuns * 8);
    c->core = (uint8_t *)take(N);
    c->sortTmp = take(tmp);
    c->sortTmpBytes = tmp;
    c->total = off;
    return hipSuccess;
}

}  // namespace

hipError_t dbscan_workspace_bytes(int n, size_t *bytes)
{
    Carve c;
    hipError_t e = carve(n, nullptr, &c, nullptr);
    *bytes = e == hipSuccess ? c.total : 0;
    return e;
}

hipError_t launch_dbscan(const float *pts, int stride, const uint8_t *mask, int n, double eps, int minPoints,
                         int32_t *labels, int32_t *counts, int32_t *numClusters, void *ws, size_t wsBytes,
                         bool *wsTooSmall, hipStream_t s)
{
    Carve c;
    hipError_t e = carve(n, ws, &c, s);
    if (e != hipSuccess) return e;
    *wsTooSmall = c.total > wsBytes;
    if (*wsTooSmall) return hipSuccess;
    const int blocks = (n + kBlock - 1) / kBlock;
    const double cell = eps * (1.0 + 1.0 / (double)(1 << 20));   // strictly wider than eps, see cell_coord
    const double eps2 = eps * eps;
    dbscan_key_kernel<<<blocks, kBlock, 0, s>>>(pts, stride, mask, n, 1.0 / cell, c.keyIn, c.valIn, c.firstRow,
                                                c.rootOf, counts);
    e = rocprim::radix_sort_pairs(c.sortTmp, c.sortTmpBytes, c.keyIn, c.keyOut, c.valIn, c.valOut, (size_t)n, 0, 63,
                                  s);
    if (e != hipSuccess) return e;
    dbscan_gather_kernel<<<blocks, kBlock, 0, s>>>(pts, stride, c.valOut, n, c.sorted);
    const int waveBlocks = (n + kWavesPerBlock - 1) / kWavesPerBlock;
    dbscan_core_kernel<<<waveBlocks, kBlock, 0, s>>>(c.sorted, c.keyOut, n, eps2, minPoints, c.runs, c.core);
    dbscan_hook_kernel<<<waveBlocks, kBlock, 0, s>>>(c.sorted, c.runs, c.core, n, eps2, c.parent);
    dbscan_flatten_kernel<<<blocks, kBlock, 0, s>>>(n, c.core, c.parent, c.chunkRoot);
    dbscan_union_kernel<<<waveBlocks, kBlock, 0, s>>>(c.sorted, c.runs, c.core, n, eps2, c.chunkRoot, c.parent);
    dbscan_first_row_kernel<<<blocks, kBlock, 0, s>>>(c.sorted, c.core, n, c.parent, c.firstRow);
    dbscan_assign_kernel<<<waveBlocks, kBlock, 0, s>>>(c.sorted, c.keyOut, c.runs, c.core, n, eps2, c.parent,
                                                       c.firstRow, c.rootOf);
    dbscan_rank_kernel<<<1, 1024, 0, s>>>(c.rootOf, n, c.rank, numClusters);
    dbscan_label_kernel<<<blocks, kBlock, 0, s>>>(c.sorted, c.rootOf, c.rank, n, labels, counts);
    return hipGetLastError();
}

}  // namespace icpflow
