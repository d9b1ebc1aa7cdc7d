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
// minimum per 16-target chunk (v_min3_f32: one instruction per two evaluations; differences
// and squares as packed v_pk_add_f32 / v_pk_mul_f32, two targets per instruction), the
// first chunk that attains the global minimum is remembered, and that single chunk is
// re-evaluated at the end to find the first index whose distance equals the minimum
// bit-for-bit (same instruction sequence => same value).
#pragma once
#include "common.hpp"

#ifndef ICPFLOW_SCAN_LOADS
#define ICPFLOW_SCAN_LOADS 2
#endif

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

typedef float v2f __attribute__((ext_vector_type(2)));

// Two targets per packed instruction.  Measured on gfx950 (tools/microbench/valu_ops.hip, ns
// per wave-instruction per SIMD at full occupancy): v_sub/mul/fma_f32 1.1, v_pk_add/mul/fma_f32
// 1.95 (two results), v_min3_f32 1.76.  Per evaluation that is 6*1.95/2 + 1.76/2 = 6.7 against
// 6*1.1 + 1.76/2 = 7.5 for the scalar form: a ~10 % gain, and fewer instructions to fetch.
// Every operation is still an individually rounded IEEE op (identical results).
__device__ __forceinline__ v2f sqdist2(float qx, float qy, float qz, v2f tx, v2f ty, v2f tz)
{
    const v2f dx = qx - tx, dy = qy - ty, dz = qz - tz;
    v2f d = dx * dx;
    d = __builtin_elementwise_fma(dy, dy, d);
    d = __builtin_elementwise_fma(dz, dz, d);
    return d;
}

// Scan the chunks [cBegin, cEnd) (multiples of kChunk) of the staged tile.
template <int Q>
__device__ __forceinline__ void scan_tile_range(const ScanTile *__restrict__ tile, int cBegin, int cEnd,
                                                int j0, const float (&qx)[Q], const float (&qy)[Q],
                                                const float (&qz)[Q], ScanAcc<Q> &acc)
{
    for (int c = cBegin; c < cEnd; c += kChunk) {
        float m[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) m[q] = kInf;
#pragma unroll
        for (int u = 0; u < kChunk / 4; ++u) {
            const float4 tx = tile->x[(c >> 2) + u];  // same address in every lane: broadcast
            const float4 ty = tile->y[(c >> 2) + u];
            const float4 tz = tile->z[(c >> 2) + u];
            const v2f txa = {tx.x, tx.y}, txb = {tx.z, tx.w};
            const v2f tya = {ty.x, ty.y}, tyb = {ty.z, ty.w};
            const v2f tza = {tz.x, tz.y}, tzb = {tz.z, tz.w};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const v2f da = sqdist2(qx[q], qy[q], qz[q], txa, tya, tza);
                const v2f db = sqdist2(qx[q], qy[q], qz[q], txb, tyb, tzb);
                m[q] = min3f(min3f(m[q], da.x, da.y), db.x, db.y);
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (m[q] < acc.best[q]) { acc.best[q] = m[q]; acc.chunk[q] = j0 + c; }
        }
    }
}

template <int Q>
__device__ __forceinline__ void scan_tile(const ScanTile *__restrict__ tile, int tp, int j0,
                                          const float (&qx)[Q], const float (&qy)[Q],
                                          const float (&qz)[Q], ScanAcc<Q> &acc)
{
    scan_tile_range<Q>(tile, 0, tp, j0, qx, qy, qz, acc);
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

// Range scan over an arbitrary SoA image (three float4 arrays in LDS) that also reports
// whether the running minimum was attained (bit-equal) by more than one chunk: used by the
// sorted sweep of the ICP loop, where scan order is not index order and the first-index rule
// has to be restored explicitly in the (rare) tie case.
// SECOND: also keep the smallest chunk minimum among the chunks OTHER than the winning one (equal to the minimum
// itself when two chunks tie): with the runner-up inside the winning chunk, found when that chunk is re-evaluated,
// it bounds the distance of every target but the nearest (the neighbour certificates of the ICP loop, icp.hip).
template <int Q, bool SECOND = false>
__device__ __forceinline__ void scan_range_tie(const float4 *__restrict__ sx, const float4 *__restrict__ sy,
                                               const float4 *__restrict__ sz, int cBegin, int cEnd,
                                               const float (&qx)[Q], const float (&qy)[Q],
                                               const float (&qz)[Q], ScanAcc<Q> &acc, bool (&tie)[Q],
                                               float *second = nullptr)
{
    for (int c = cBegin; c < cEnd; c += kChunk) {
        float m[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) m[q] = kInf;
#pragma unroll
        for (int u = 0; u < kChunk / 4; ++u) {
            if (u > 0 && (u % ICPFLOW_SCAN_LOADS) == 0) __builtin_amdgcn_sched_barrier(0);   // bound the loads in flight
            const float4 tx = sx[(c >> 2) + u];  // same address in every lane: broadcast
            const float4 ty = sy[(c >> 2) + u];
            const float4 tz = sz[(c >> 2) + u];
            const v2f txa = {tx.x, tx.y}, txb = {tx.z, tx.w};
            const v2f tya = {ty.x, ty.y}, tyb = {ty.z, ty.w};
            const v2f tza = {tz.x, tz.y}, tzb = {tz.z, tz.w};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const v2f da = sqdist2(qx[q], qy[q], qz[q], txa, tya, tza);
                const v2f db = sqdist2(qx[q], qy[q], qz[q], txb, tyb, tzb);
                m[q] = min3f(min3f(m[q], da.x, da.y), db.x, db.y);
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (SECOND) second[q] = min_nonneg(second[q], max_nonneg(m[q], acc.best[q]));   // the larger of (old best, this chunk)
            if (m[q] < acc.best[q]) { acc.best[q] = m[q]; acc.chunk[q] = c; tie[q] = false; }
            else if (m[q] == acc.best[q]) tie[q] = true;
        }
    }
}

// The same range scan with the targets streamed from GLOBAL memory at wave-uniform addresses:
// the compiler turns these loads into scalar loads (s_load_dwordx4 into SGPRs), the VALU
// instructions take the coordinates as scalar operands, and neither LDS nor the vector memory
// pipe is touched by the hot loop.  gx/gy/gz: the sorted cloud as three float arrays, padded
// with +inf to a multiple of kChunk.  cBegin/cEnd must be wave-uniform.
typedef float v4f __attribute__((ext_vector_type(4)));
// constant address space: tells the compiler the data is invariant for the kernel, which is what
// lets it select scalar loads for wave-uniform addresses (it emits s_load_dwordx16 here)
typedef const __attribute__((address_space(4))) v4f *ConstV4;

template <int Q>
__device__ __forceinline__ void scan_range_tie_uniform(const float *__restrict__ gx, const float *__restrict__ gy,
                                                       const float *__restrict__ gz, int cBegin, int cEnd,
                                                       const float (&qx)[Q], const float (&qy)[Q],
                                                       const float (&qz)[Q], ScanAcc<Q> &acc, bool (&tie)[Q])
{
    cBegin = __builtin_amdgcn_readfirstlane(cBegin);
    cEnd = __builtin_amdgcn_readfirstlane(cEnd);
    if (cBegin >= cEnd) return;
    // software pipeline in half chunks (8 targets = 24 SGPRs per buffer): the scalar loads of the
    // next half are issued before the current half is evaluated, so their latency hides behind
    // ~30 VALU instructions per query.  Reading one half past cEnd stays inside the allocation
    // (+inf padding, then the next coordinate array).
    constexpr int HALF = kChunk / 2;
    v4f nx0 = *(ConstV4)(gx + cBegin), nx1 = *(ConstV4)(gx + cBegin + 4);
    v4f ny0 = *(ConstV4)(gy + cBegin), ny1 = *(ConstV4)(gy + cBegin + 4);
    v4f nz0 = *(ConstV4)(gz + cBegin), nz1 = *(ConstV4)(gz + cBegin + 4);
    for (int c = cBegin; c < cEnd; c += kChunk) {
        float m[Q];
#pragma unroll
        for (int q = 0; q < Q; ++q) m[q] = kInf;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const v4f tx0 = nx0, tx1 = nx1, ty0 = ny0, ty1 = ny1, tz0 = nz0, tz1 = nz1;
            const int cn = c + (hh + 1) * HALF;
            nx0 = *(ConstV4)(gx + cn); nx1 = *(ConstV4)(gx + cn + 4);
            ny0 = *(ConstV4)(gy + cn); ny1 = *(ConstV4)(gy + cn + 4);
            nz0 = *(ConstV4)(gz + cn); nz1 = *(ConstV4)(gz + cn + 4);
            const v2f xa = {tx0.x, tx0.y}, xb = {tx0.z, tx0.w}, xc = {tx1.x, tx1.y}, xd = {tx1.z, tx1.w};
            const v2f ya = {ty0.x, ty0.y}, yb = {ty0.z, ty0.w}, yc = {ty1.x, ty1.y}, yd = {ty1.z, ty1.w};
            const v2f za = {tz0.x, tz0.y}, zb = {tz0.z, tz0.w}, zc = {tz1.x, tz1.y}, zd = {tz1.z, tz1.w};
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const v2f da = sqdist2(qx[q], qy[q], qz[q], xa, ya, za);
                const v2f db = sqdist2(qx[q], qy[q], qz[q], xb, yb, zb);
                const v2f dc = sqdist2(qx[q], qy[q], qz[q], xc, yc, zc);
                const v2f dd = sqdist2(qx[q], qy[q], qz[q], xd, yd, zd);
                m[q] = min3f(min3f(m[q], da.x, da.y), db.x, db.y);
                m[q] = min3f(min3f(m[q], dc.x, dc.y), dd.x, dd.y);
            }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            if (m[q] < acc.best[q]) { acc.best[q] = m[q]; acc.chunk[q] = c; tie[q] = false; }
            else if (m[q] == acc.best[q]) tie[q] = true;
        }
    }
}

// Minimum squared distance only (no arg-min) over the chunks [cBegin, cEnd) of a sorted SoA image
// streamed through scalar loads, for the unbounded-NN sweeps of the scoring / check scans (nn.hip).
// SHIFT: the targets are the image translated by (sx, sy, sz); the sum is formed first, in fp32, like
// the reference's transformed cloud, so distances are bit-identical to scanning a translated copy.
template <bool SHIFT>
__device__ __forceinline__ void scan_range_min_uniform(const float *__restrict__ gx, const float *__restrict__ gy,
                                                       const float *__restrict__ gz, int cBegin, int cEnd,
                                                       float qx, float qy, float qz, float sx, float sy, float sz,
                                                       float &best)
{
    cBegin = __builtin_amdgcn_readfirstlane(cBegin);
    cEnd = __builtin_amdgcn_readfirstlane(cEnd);
    if (cBegin >= cEnd) return;
    constexpr int HALF = kChunk / 2;
    const v2f shx = {sx, sx}, shy = {sy, sy}, shz = {sz, sz};
    v4f nx0 = *(ConstV4)(gx + cBegin), nx1 = *(ConstV4)(gx + cBegin + 4);
    v4f ny0 = *(ConstV4)(gy + cBegin), ny1 = *(ConstV4)(gy + cBegin + 4);
    v4f nz0 = *(ConstV4)(gz + cBegin), nz1 = *(ConstV4)(gz + cBegin + 4);
    float m = best;
    for (int c = cBegin; c < cEnd; c += HALF) {
        const v4f tx0 = nx0, tx1 = nx1, ty0 = ny0, ty1 = ny1, tz0 = nz0, tz1 = nz1;
        const int cn = c + HALF;   // one half past cEnd stays inside the allocation (+inf padding)
        nx0 = *(ConstV4)(gx + cn); nx1 = *(ConstV4)(gx + cn + 4);
        ny0 = *(ConstV4)(gy + cn); ny1 = *(ConstV4)(gy + cn + 4);
        nz0 = *(ConstV4)(gz + cn); nz1 = *(ConstV4)(gz + cn + 4);
        v2f xa = {tx0.x, tx0.y}, xb = {tx0.z, tx0.w}, xc = {tx1.x, tx1.y}, xd = {tx1.z, tx1.w};
        v2f ya = {ty0.x, ty0.y}, yb = {ty0.z, ty0.w}, yc = {ty1.x, ty1.y}, yd = {ty1.z, ty1.w};
        v2f za = {tz0.x, tz0.y}, zb = {tz0.z, tz0.w}, zc = {tz1.x, tz1.y}, zd = {tz1.z, tz1.w};
        if (SHIFT) {
            xa += shx; xb += shx; xc += shx; xd += shx;
            ya += shy; yb += shy; yc += shy; yd += shy;
            za += shz; zb += shz; zc += shz; zd += shz;
        }
        const v2f da = sqdist2(qx, qy, qz, xa, ya, za);
        const v2f db = sqdist2(qx, qy, qz, xb, yb, zb);
        const v2f dc = sqdist2(qx, qy, qz, xc, yc, zc);
        const v2f dd = sqdist2(qx, qy, qz, xd, yd, zd);
        m = min3f(min3f(m, da.x, da.y), db.x, db.y);
        m = min3f(min3f(m, dc.x, dc.y), dd.x, dd.y);
    }
    best = m;
}

// Full scan of one target cloud for this lane's Q queries.  All threads of the block
// must call it (it contains barriers).  With TS > 1 the caller's wave scans only share `ts`
// of every tile (contiguous chunk ranges); the TS partial (best, chunk) results of a query
// are then merged by the caller with scan_better() -- order-independent, so the
// first-minimum rule survives the split.
template <int Q>
__device__ __forceinline__ void scan_cloud(const CloudView &tg, const PointXf &txf, ScanTile *tile,
                                           const float (&qx)[Q], const float (&qy)[Q],
                                           const float (&qz)[Q], ScanAcc<Q> &acc, int ts = 0, int TS = 1)
{
    scan_init(acc);
    for (int j0 = 0; j0 < tg.n; j0 += kScanTile) {
        __syncthreads();
        const int tp = stage_tile(tg, j0, txf, tile);
        __syncthreads();
        const int nchunk = tp / kChunk;
        const int share = (nchunk + TS - 1) / TS;
        const int cb = min(ts * share, nchunk) * kChunk;
        const int ce = min((ts + 1) * share, nchunk) * kChunk;
        scan_tile_range<Q>(tile, cb, ce, j0, qx, qy, qz, acc);
    }
}

// (d, chunk) candidate `b` beats `a`: smaller distance, or equal distance in an earlier chunk
__device__ __forceinline__ bool scan_better(float da, int ca, float db, int cb)
{
    return db < da || (db == da && cb < ca);
}

}  // namespace icpflow
