// votekey.hpp -- the sort key of the z-sorted vote (hist.hip, sort.hip).
//
// Vehicle-sized clusters are sorted by z: the vote box is thin in z (utils_hist.py:65), so a wave of
// consecutive sorted rows only has to visit the rows of the other cloud inside its z window.  A WIDE
// cluster (a wall, a building front: tens of metres along x or y, any height) defeats that -- its rows are
// spread along the long axis u, the box is +-translation_frame wide there, and although every target of the z
// window is visited only ~box/extent of the lanes vote.  Such pairs are sorted by a composite key
//      K = slab(z) * 1024 + (u - u0),     slab(z) = floor((z - z0) / h),  h = the z range of the box
// so that consecutive rows share a slab AND a neighbourhood along u; a wave then visits, slab by slab, only
// the targets whose u can fall into the box.  The visited set is still a superset of the pairs that can
// pass the exact box test, which decides every vote exactly as before: bins are bit-identical.
// K is exact enough by construction: < 256 slabs (K < 2^18, ulp 1/32 m) and u - u0 < 1000 m, else plain z.
#pragma once
#include "common.hpp"

namespace icpflow {

constexpr float kWideMinExtent = 8.0f;     // metres along u above which the composite key pays
constexpr float kSlabStride = 1024.0f;
constexpr int kWideMinPoints = 4096;       // both clouds together; below, the key is plain z
constexpr int kVoteKeyStride = 8;          // floats per pair in the parameter record

struct VoteKey {
    int wide;      // 0: K = z
    int uaxis;     // 0 = x, 1 = y
    float u0, z0, h;
};

__device__ __forceinline__ float vote_key(const VoteKey &k, float x, float y, float z)
{
    if (!k.wide) return z;
    const float s = floorf((z - k.z0) / k.h);
    const float u = fminf(fmaxf((k.uaxis == 0 ? x : y) - k.u0, 0.f), kSlabStride - 1.f);
    return fmaf(s, kSlabStride, u);
}

__device__ __forceinline__ VoteKey vote_key_load(const float *rec)
{
    VoteKey k;
    k.wide = rec[0] != 0.f; k.uaxis = (int)rec[1]; k.u0 = rec[2]; k.z0 = rec[3]; k.h = rec[4];
    return k;
}

// this thread's share of the bounding box of `n` rows (rows tid, tid + blockDim, ...; flagged rows only when `flagged`):
// four rows in flight per thread -- a row at a time, the loop is one dependent chain of loads (min / max are exact in any
// order)
__device__ __forceinline__ void bbox_rows(const float4 *__restrict__ pts, int n, bool flagged, float (&mn)[3], float (&mx)[3])
{
    const int stride = (int)blockDim.x;
    for (int j0 = (int)threadIdx.x; j0 < n; j0 += 4 * stride) {
        float4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = pts[min(j0 + u * stride, n - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (j0 + u * stride >= n || (flagged && !(q[u].w > 0.0f))) continue;
            mn[0] = fminf(mn[0], q[u].x); mn[1] = fminf(mn[1], q[u].y); mn[2] = fminf(mn[2], q[u].z);
            mx[0] = fmaxf(mx[0], q[u].x); mx[1] = fmaxf(mx[1], q[u].y); mx[2] = fmaxf(mx[2], q[u].z);
        }
    }
}

// key parameters from the bounding box of the valid rows of both clouds (one thread)
__device__ __forceinline__ void vote_key_from_box(const float (&lo)[3], const float (&hi)[3], float hBox, float *result)
{
    const int ua = (hi[0] - lo[0]) >= (hi[1] - lo[1]) ? 0 : 1;
    const float eu = hi[ua] - lo[ua], ez = hi[2] - lo[2];
    const float h = fmaxf(hBox, 1e-3f);
    const bool wide = eu > kWideMinExtent && eu < 1000.f && ez < 250.f * h;
    result[0] = wide ? 1.f : 0.f; result[1] = (float)ua; result[2] = lo[ua]; result[3] = lo[2]; result[4] = h;
}

// the same from the boxes count_pair_kernel leaves per pair (kPairBoxStride floats: cloud A / C x flagged rows / all rows
// below the count x (min xyz, max xyz)): the flagged boxes of both clouds, merged
__device__ inline VoteKey vote_key_params_boxed(const float *__restrict__ box, int nP, int nQ, float hBox, float *result)
{
    if (threadIdx.x == 0) {
        if (nP + nQ <= kWideMinPoints) {
            result[0] = 0.f; result[1] = 0.f; result[2] = 0.f; result[3] = 0.f; result[4] = fmaxf(hBox, 1e-3f);
        } else {
            float lo[3], hi[3];
            for (int k = 0; k < 3; ++k) { lo[k] = fminf(box[k], box[12 + k]); hi[k] = fmaxf(box[3 + k], box[15 + k]); }
            vote_key_from_box(lo, hi, hBox, result);
        }
    }
    __syncthreads();
    return vote_key_load(result);
}

// Block-cooperative: bounding box of the valid rows of BOTH clouds -> key parameters (identical in every
// block that calls it for the same pair).  scratch: 6 floats per wave of the block.
__device__ inline VoteKey vote_key_params(const float4 *__restrict__ P, int nP, const float4 *__restrict__ Q,
                                          int nQ, float hBox, float *scratch, float *result)
{
    if (nP + nQ <= kWideMinPoints) {   // small pair: the plain z window is already short; skip the bounding box
        if (threadIdx.x == 0) { result[0] = 0.f; result[1] = 0.f; result[2] = 0.f; result[3] = 0.f; result[4] = fmaxf(hBox, 1e-3f); }
        __syncthreads();
        return vote_key_load(result);
    }
    float mn[3] = {kInf, kInf, kInf}, mx[3] = {-kInf, -kInf, -kInf};
    bbox_rows(P, nP, true, mn, mx);
    bbox_rows(Q, nQ, true, mn, mx);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
            mn[k] = fminf(mn[k], __shfl_xor(mn[k], o, kWave));
            mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], o, kWave));
        }
    const int wave = threadIdx.x >> 6, nwave = (blockDim.x + kWave - 1) >> 6;
    __syncthreads();
    if ((threadIdx.x & (kWave - 1)) == 0)
        for (int k = 0; k < 3; ++k) { scratch[wave * 6 + k] = mn[k]; scratch[wave * 6 + 3 + k] = mx[k]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float lo[3], hi[3];
        for (int k = 0; k < 3; ++k) {
            lo[k] = scratch[k]; hi[k] = scratch[3 + k];
            for (int w = 1; w < nwave; ++w) { lo[k] = fminf(lo[k], scratch[w * 6 + k]); hi[k] = fmaxf(hi[k], scratch[w * 6 + 3 + k]); }
        }
        vote_key_from_box(lo, hi, hBox, result);
    }
    __syncthreads();
    return vote_key_load(result);
}

}  // namespace icpflow
