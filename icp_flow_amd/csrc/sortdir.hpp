// sortdir.hpp -- the sort key of the sweeps (icp.hip, nn.hip, sort.hip): a coordinate, or a horizontal DIRECTION.
//
// Both clouds of a pair are sorted once per registration by a key k(p); a wave of 64 consecutive sorted queries then only visits
// the targets whose key lies within the search radius of its queries' keys.  What makes that exact is |k(p) - k(t)| <= |p - t|,
// which holds for a coordinate and for u . p with ANY unit vector u.  Until round 6 the key was the coordinate along the fixed
// cloud's longest axis -- and a vehicle whose heading is along that axis shows a face ACROSS it: a third of a 2048-point shell
// shares one key, every window there holds 650-800 targets instead of ~270, a probe has to swallow the whole face before the
// distance along the key proves anything (profiles/r03_config4_unit_clocks.txt: 40-50 k clocks for such a unit against 2-5 k
// for a certified one -- the pairs that end every ICP launch).  So the sort kernels now pick, per pair, the key that spreads the
// fixed cloud best: the three axes and six horizontal directions (22.5 deg apart: a box has a direction at least 33 deg from both
// of its face normals) by the sum of squared populations of 0.1 m key bins -- the expected number of targets in a window is
// n + 2 m . integral(rho^2) for a key density rho.  The pair's code is what `GridScratch.axis` carries: 0, 1, 2 = x, y, z as
// before (exact keys, nothing else changes), 3 .. 8 = direction (cos phi, sin phi, 0).
//
// A direction key is COMPUTED, fl(fl(ux x) + uy y) by one multiply and one fused multiply-add -- the same two instructions
// wherever a key is formed, so the sorted order is exactly monotone in the keys the searches compare --, and differs from
// u . p by at most 2^-24 (|ux x| + |k|) <= 2^-23 (|x| + |y|).  Every deduction "the key differs by more than r, so the point is
// farther than r" therefore gives up sort_key_slack() first: for two points within r of each other the two errors add up to
// at most 2^-23 (2 (|x| + |y|) + 2 r) of the query's coordinates (the target's are within r of them).
#pragma once
#include <hip/hip_runtime.h>

namespace icpflow {

// Fixed clouds of at least this many points choose among the direction keys (sortdir.hpp): clouds of more than one pass of a
// 1024-thread workgroup.  At 1024 points (config 2, one pass, an iteration paced by the serial tail and its slowest wave) the
// direction keys measured +5 % on the ICP launch (0.390 -> 0.409 ms) against -5 ... -11 % at 1500-4000 points.
#ifndef ICPFLOW_SORT_DIR_MIN_N
#define ICPFLOW_SORT_DIR_MIN_N 1025
#endif
constexpr int kSortDirMinN = ICPFLOW_SORT_DIR_MIN_N;
// ... and a MOVING cloud of at least this many points: the gain is the moving cloud's units times what a window loses, and a
// small cluster against a long cloud has few units whose sparse queries span most of the other cloud under any key (ragged batch
// with independent sizes: 1.70 -> 1.74 ms with direction keys for every long fixed cloud)
#ifndef ICPFLOW_SORT_DIR_MIN_MOVING
#define ICPFLOW_SORT_DIR_MIN_MOVING 512
#endif
constexpr int kSortDirMinMoving = ICPFLOW_SORT_DIR_MIN_MOVING;
constexpr int kSortDirs = 6;          // codes 3 .. 3 + kSortDirs - 1
constexpr int kSortCodes = 3 + kSortDirs;

// components of direction `code` (>= 3), rounded TOWARDS zero so that |u| <= 1
__host__ __device__ __forceinline__ void sort_dir(int code, float &ux, float &uy)
{
    constexpr float c1 = 0.9238795f, s1 = 0.3826834f, c2 = 0.7071067f;   // cos / sin 22.5 deg, cos 45 deg
    switch (code) {
    case 3: ux = c1; uy = s1; break;     //  22.5 deg
    case 4: ux = c2; uy = c2; break;     //  45
    case 5: ux = s1; uy = c1; break;     //  67.5
    case 6: ux = -s1; uy = c1; break;    // 112.5
    case 7: ux = -c2; uy = c2; break;    // 135
    default: ux = -c1; uy = s1; break;   // 157.5
    }
}

// the key of a point; (ux, uy) = sort_dir(code) for code >= 3
__device__ __forceinline__ float sort_key_dir(float ux, float uy, float x, float y) { return fmaf(uy, y, ux * x); }
__device__ __forceinline__ float sort_key_of(int code, float ux, float uy, float x, float y, float z)
{
    return code >= 3 ? sort_key_dir(ux, uy, x, y) : (code == 0 ? x : (code == 1 ? y : z));
}
// what a deduction from a key difference gives up for a query at (x, y) and points within r of it; 0 for a coordinate key
__device__ __forceinline__ float sort_key_slack(int code, float x, float y, float r)
{
    return code >= 3 ? 2.4e-7f * (fabsf(x) + fabsf(y) + r) + 1e-30f : 0.f;   // (2^-22: twice the bound above)
}

// The key of a pair (all BLOCK threads of a workgroup call it; contains barriers): among the three axes and the kSortDirs
// directions the one with the smallest sum of squared populations of 0.1 m key bins over the FIXED cloud's rows -- `legacy`, the
// longest axis, unless another key is at least a tenth better.  Integer counts: every workgroup of a pair finds the same code.
// hist: 3 * kSortDirBins counters of LDS, scoreSh: kSortCodes, codeSh: one int; box = (min x y z, max x y z) of the rows.
constexpr int kSortDirBins = 512;
#ifndef ICPFLOW_SORT_DIR_BIN
#define ICPFLOW_SORT_DIR_BIN 0.1f
#endif
#ifndef ICPFLOW_SORT_DIR_KEEP
#define ICPFLOW_SORT_DIR_KEEP 9   // the longest axis stays unless another key's sum is at most this many tenths of its own
#endif
template <int BLOCK>
__device__ __forceinline__ int choose_sort_code(const float4 *__restrict__ rows, int n, const float *box, int legacy,
                                                unsigned int *hist, unsigned int *scoreSh, int *codeSh)
{
    constexpr int kPerRound = 3;
    const int tid = threadIdx.x;
    const float cx = 0.5f * (box[0] + box[3]), cy = 0.5f * (box[1] + box[4]), cz = 0.5f * (box[2] + box[5]);
    const float ex = box[3] - box[0], ey = box[4] - box[1], ez = box[5] - box[2];
    const float R = 0.5f * sqrtf(ex * ex + ey * ey + ez * ez) + 0.05f;
    const float inv = 1.0f / fmaxf(ICPFLOW_SORT_DIR_BIN, 2.0f * R / (float)kSortDirBins);
    if (tid < kSortCodes) scoreSh[tid] = 0u;
    for (int c0 = 0; c0 < kSortCodes; c0 += kPerRound) {
        for (int k = tid; k < kPerRound * kSortDirBins; k += BLOCK) hist[k] = 0u;
        __syncthreads();
        for (int j = tid; j < n; j += BLOCK) {
            const float4 q = rows[j];
#pragma unroll
            for (int u = 0; u < kPerRound; ++u) {
                const int code = c0 + u;
                float ux = 0.f, uy = 0.f;
                if (code >= 3) sort_dir(code, ux, uy);
                const float k = sort_key_of(code, ux, uy, q.x - cx, q.y - cy, q.z - cz);
                const int bin = min(max((int)((k + R) * inv), 0), kSortDirBins - 1);
                atomicAdd(&hist[u * kSortDirBins + bin], 1u);
            }
        }
        __syncthreads();
        for (int k = tid; k < kPerRound * kSortDirBins; k += BLOCK) {
            const unsigned int c = hist[k];
            if (c > 1u) atomicAdd(&scoreSh[c0 + k / kSortDirBins], c * c);
        }
        __syncthreads();
    }
    if (tid == 0) {
        int best = legacy;
        unsigned int sb = scoreSh[legacy];
        for (int c = 0; c < kSortCodes; ++c)
            if (scoreSh[c] < sb) { sb = scoreSh[c]; best = c; }
        // (a tenth better at least, in integers: 10 s_best <= 9 s_legacy)
        *codeSh = (best != legacy && (unsigned long long)sb * 10ull <= (unsigned long long)scoreSh[legacy] * (unsigned long long)ICPFLOW_SORT_DIR_KEEP) ? best : legacy;
    }
    __syncthreads();
    return *codeSh;
}

}  // namespace icpflow
