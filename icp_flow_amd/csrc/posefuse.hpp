// posefuse.hpp -- the final pose of a pair straight from what the ICP launch left behind, for the kernels that
// consume it (roll-back check, select): no icp_resolve_history / compose launches in between.
//
// After the speculative single launch (icp.hip) every pair has its (R, T, rmse) of every iteration in `history`
// and the per-iteration tallies say where the reference's batch-global rule stops (the first iteration at which
// every pair had arrived and none was unconverged, utils_icp_pytorch3d.py:209); otherwise `state` already holds
// the final (R, T).  final_pose() is utils_icp.py:60-65 + :24: [[R^T, T], [0 0 0 1]] * init, evaluated with the
// fmaf sequence of compose_kernel (pose.hip) -- bit-identical to the unfused path.
#pragma once
#include "common.hpp"
#include "kernels.hpp"

namespace icpflow {

struct PoseSource {
    const IcpState *state;   // [B]
    const IcpCtrl *ctrl;
    const float *history;    // [kHistIters, B, kHistStride] or NULL (state is final)
    const float *init;       // [B,4,4]
    int B, maxIter;
};

// -> number of iterations of the batch rule (history mode) or the recorded count; every thread may call it
// (all threads read the same addresses)
__device__ __forceinline__ int pose_stop_iteration(const PoseSource &ps)
{
    if (ps.history == nullptr) return ps.ctrl->iters;
    int n = ps.maxIter;
    for (int s = 0; s < ps.maxIter; ++s) {
        const unsigned long long t = ps.ctrl->tally[s];
        if ((int)(t & 0xffffffffull) == ps.B && (t >> 32) == 0ull) { n = s + 1; break; }
    }
    return n;
}

// the same, one wave cooperating (64 tallies per round); wave-uniform result
__device__ __forceinline__ int pose_stop_iteration_wave(const PoseSource &ps, int lane)
{
    if (ps.history == nullptr) return ps.ctrl->iters;
    for (int s0 = 0; s0 < ps.maxIter; s0 += kWave) {
        const int s = s0 + lane;
        bool hit = false;
        if (s < ps.maxIter) {
            const unsigned long long t = ps.ctrl->tally[s];
            hit = (int)(t & 0xffffffffull) == ps.B && (t >> 32) == 0ull;
        }
        const unsigned long long m = __ballot(hit);
        if (m != 0ull) return s0 + __builtin_ctzll(m) + 1;
    }
    return ps.maxIter;
}

__device__ __forceinline__ void final_pose(const PoseSource &ps, int b, int n, float (&M)[16])
{
    float R[9], T[3];
    if (ps.history != nullptr) {
        const float *h = ps.history + ((size_t)(n - 1) * ps.B + b) * kHistStride;
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = h[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) T[k] = h[9 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = ps.state[b].R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) T[k] = ps.state[b].T[k];
    }
    float A[16];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i * 4 + j] = R[j * 3 + i];
        A[i * 4 + 3] = T[i];
    }
    A[12] = A[13] = A[14] = 0.f; A[15] = 1.f;
    if (ps.ctrl->error) A[0] = __int_as_float(0x7fc00000);   // abandoned launch: poison every pose
    const float *I = ps.init + (size_t)b * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = A[i * 4 + 0] * I[0 * 4 + j];
            acc = fmaf(A[i * 4 + 1], I[1 * 4 + j], acc);
            acc = fmaf(A[i * 4 + 2], I[2 * 4 + j], acc);
            acc = fmaf(A[i * 4 + 3], I[3 * 4 + j], acc);
            M[i * 4 + j] = acc;
        }
}

}  // namespace icpflow
