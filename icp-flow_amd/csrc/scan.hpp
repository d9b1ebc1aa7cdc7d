// scan.hpp -- the all-pairs nearest-neighbour scan core shared by every kernel on the
// path (candidate scoring, ICP iterations, roll-back check, match_eval, plain NN).
//
// Mapping for gfx950: a 256-thread workgroup owns 256*Q query points of ONE cloud pair;
// every lane keeps Q queries in registers; the target cloud streams through a 12 KiB
// SoA LDS tile (coalesced 16-byte global loads, the optional rigid map applied once while
// staging) and all lanes read the SAME target element per step, i.e. LDS broadcast
// reads that are bank-conflict free and amortised over Q distance evaluations.
//
// Distance arithmetic is the pytorch3d order (direct differences, FMA chain), never the
// |a|^2+|b|^2-2ab expansion (pads sit at 1e8).  Arg-min with the first-minimum tie rule
// is recovered without per-evaluation compare/select: the hot loop keeps only a running
// minimum per 16-target chunk (v_min3_f32: one instruction per two evaluations), the
// first chunk that attains the global minimum is remembered, and that single chunk is
// re-evaluated at the end to find the first index whose distance equals the minimum
// bit-for-bit (same instruction sequence => same value).
#pragma once
#include "common.hpp"

namespace icpflow {

constexpr int kScanBlock = 256;  // threads per workgroup
constexpr int kScanTile = 1024;  // targets per LDS tile
constexpr int kChunk = 16;       // targets per running-minimum chunk

__device__ __forceinline__ float min3f(float a, float b, float c)
{
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// View of a cloud: rows of `stride` floats whose first three are x,y,z.
struct CloudView {
    const float *base;
    int stride;  // floats per row (4 for the [B,N,4] layout)
    int n;       // rows to use (valid prefix, or all rows)
};

__device__ __forceinline__ void cloud_load(const CloudView &c, int i, float &x, float &y, float &z)
{
    if (c.stride == 4) {
        const float4 p = reinterpret_cast<const float4 *>(c.base)[i];
        x = p.x; y = p.y; z = p.z;
    } else {
        const float *p = c.base + (size_t)i * c.stride;
        x = p[0]; y = p[1]; z = p[2];
    }
}

// LDS tile of targets, structure-of-arrays so that one ds_read_b128 delivers the same
// coordinate of FOUR targets (12 LDS cycles per 4 targets per wave; an array-of-float4
// tile compiles to ds_read_b96, 8 cycles per target).
struct ScanTile {
    float4 x[kScanTile / 4];
    float4 y[kScanTile / 4];
    float4 z[kScanTile / 4];
};

// Stage targets [j0, j0+tn) (mapped by xf) into the LDS tile, padded with +inf points
// up to a multiple of kChunk.  Caller brackets with __syncthreads().
__device__ __forceinline__ int stage_tile(const CloudView &tg, int j0, const PointXf &xf, ScanTile *tile)
{
    const int tn = min(kScanTile, tg.n - j0);
    const int tp = (tn + kChunk - 1) / kChunk * kChunk;
    float *tx = reinterpret_cast<float *>(tile->x);
    float *ty = reinterpret_cast<float *>(tile->y);
    float *tz = reinterpret_cast<float *>(tile->z);
    for (int k = threadIdx.x; k < tp; k += blockDim.x) {
        float ox = kInf, oy = kInf, oz = kInf;
        if (k < tn) {
            float x, y, z;
            cloud_load(tg, j0 + k, x, y, z);
            xf_apply(xf, x, y, z, ox, oy, oz);
        }
        tx[k] = ox; ty[k] = oy; tz[k] = oz;
    }
    return tp;
}

template <int Q>
struct ScanAcc {
    float best[Q];  // running minimum squared distance
    int chunk[Q];   // global index of the first target of the first chunk attaining it
};

template <int Q>
__device__ __forceinline__ void scan_init(ScanAcc<Q> &a)
{
#pragma unroll
    for (int q = 0; q < Q; ++q) { a.best[q] = kInf; a.chunk[q] = 0; }
}

template <int Q>
__device__ __forceinline__ void scan_tile(const ScanTile *__restrict__ tile, int tp, int j0,
                                          const float (&qx)[Q], const float (&qy)[Q],
                                          const float (&qz)[Q], ScanAcc<Q> &acc)
{
    for (int c = 0; c < tp; c += kChunk) {
        float m[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) m[q] = kInf;
#pragma unroll
        for (int u = 0; u < kChunk / 4; ++u) {
            const float4 tx = tile->x[(c >> 2) + u];  // same address in every lane: broadcast
            const float4 ty = tile->y[(c >> 2) + u];
            const float4 tz = tile->z[(c >> 2) + u];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const float d0 = sqdist(qx[q], qy[q], qz[q], tx.x, ty.x, tz.x);
                const float d1 = sqdist(qx[q], qy[q], qz[q], tx.y, ty.y, tz.y);
                const float d2 = sqdist(qx[q], qy[q], qz[q], tx.z, ty.z, tz.z);
                const float d3 = sqdist(qx[q], qy[q], qz[q], tx.w, ty.w, tz.w);
                m[q] = min3f(min3f(m[q], d0, d1), d2, d3);
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (m[q] < acc.best[q]) { acc.best[q] = m[q]; acc.chunk[q] = j0 + c; }
        }
    }
}

// First index in [chunk, chunk+kChunk) whose distance equals `best` (bit-exact
// re-evaluation from global memory).  Returns 0 when nothing was scanned.
__device__ __forceinline__ int scan_resolve(const CloudView &tg, const PointXf &xf, float qx,
                                            float qy, float qz, float best, int chunk,
                                            float &nx, float &ny, float &nz)
{
    int idx = -1;
    nx = ny = nz = 0.f;
    const int hi = min(chunk + kChunk, tg.n);
    for (int j = chunk; j < hi; ++j) {
        float x, y, z, tx, ty, tz;
        cloud_load(tg, j, x, y, z);
        xf_apply(xf, x, y, z, tx, ty, tz);
        const float d = sqdist(qx, qy, qz, tx, ty, tz);
        if (idx < 0 && d == best) { idx = j; nx = tx; ny = ty; nz = tz; }
    }
    return idx < 0 ? 0 : idx;
}

// Full scan of one target cloud for this lane's Q queries.  All threads of the block
// must call it (it contains barriers).
template <int Q>
__device__ __forceinline__ void scan_cloud(const CloudView &tg, const PointXf &txf, ScanTile *tile,
                                           const float (&qx)[Q], const float (&qy)[Q],
                                           const float (&qz)[Q], ScanAcc<Q> &acc)
{
    scan_init(acc);
    for (int j0 = 0; j0 < tg.n; j0 += kScanTile) {
        __syncthreads();
        const int tp = stage_tile(tg, j0, txf, tile);
        __syncthreads();
        scan_tile<Q>(tile, tp, j0, qx, qy, qz, acc);
    }
}

}  // namespace icpflow
