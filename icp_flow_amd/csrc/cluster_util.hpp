// cluster_util.hpp -- shared by cluster.hip (DBSCAN) and hdbscan.hip: uniform-grid cell keys and the
// lock-free union-find.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpflow {
namespace {

constexpr int kCellBits = 21;
constexpr unsigned long long kMaskedKey = 0x7fffffffffffffffull;   // sorts after every real cell

__device__ inline long long cell_coord(float v, double invCell)
{
    // monotone in v; two points closer than eps fall into the same or adjacent cells (cell > eps).
    // Clamped so that +-1 stays inside the 21-bit field (clamping keeps adjacency).
    double q = floor((double)v * invCell) + (double)(1 << (kCellBits - 1));
    q = fmin(fmax(q, 1.0), (double)((1 << kCellBits) - 2));
    return (long long)q;
}

__device__ inline unsigned long long pack_key(long long cx, long long cy, long long cz)
{
    return ((unsigned long long)cx << (2 * kCellBits)) | ((unsigned long long)cy << kCellBits) |
           (unsigned long long)cz;
}

__device__ inline int uf_load(const int *parent, int i)
{
    return __hip_atomic_load(parent + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ inline int uf_find(int *parent, int i)
{
    int p = uf_load(parent, i);
    while (p != i) {
        const int g = uf_load(parent, p);
        if (g != p) __hip_atomic_store(parent + i, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // path splitting
        i = p;
        p = g;
    }
    return i;
}

// roots only ever move to SMALLER indices, so the root of a finished component is its smallest member
__device__ inline void uf_union(int *parent, int a, int b)
{
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a < b) {
            const int t = a;
            a = b;
            b = t;
        }
        if (atomicCAS(parent + a, a, b) == a) return;
    }
}

inline size_t up256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace
}  // namespace icpflow
