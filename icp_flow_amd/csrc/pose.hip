// pose.hip -- the small per-pair kernels between the scans: candidate decoding and
// scoring (utils_hist.py:78-121), 4x4 assembly / roll-back / inverse
// (utils_icp.py:24-35,60-65; utils_match.py:139-156), match_eval's epilogue
// (utils_match.py:168-184) and transform_points_batch (utils_helper.py:76-87).
// One lane per pair: these are latency-trivial next to the O(n^2) scans.
#include "common.hpp"
#include "kernels.hpp"
#include "posefuse.hpp"

namespace icpflow {

// sum of a job's per-block records, in block order.  Eight loads in flight at a time (unconditional, from clamped
// addresses), then eight ordered adds: as a plain loop every record is a dependent round trip -- 40 of them on a
// 10000-point batch, where this was most of select_kernel's 19 us.
__device__ __forceinline__ double partial_total(const double *partial, int job, int qblocks, int k)
{
    double s = 0.0;
    const double *base = partial + (size_t)job * qblocks * kPartial + k;
    if (qblocks <= 4) {   // (config 2: four blocks -- no point in eight loads)
        double v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = base[(size_t)min(u, qblocks - 1) * kPartial];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (u < qblocks) s += v[u];
        return s;
    }
    for (int q0 = 0; q0 < qblocks; q0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)min(q0 + u, qblocks - 1) * kPartial];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q0 + u < qblocks) s += v[u];
    }
    return s;
}

// score_k = min(mean fwd, mean bwd); first arg-min; T = I with that translation
// (utils_hist.py:101-106, :121-122)
// one wave per pair: lane j < 12 totals job j (candidate j / 2, direction j % 2) over the query blocks
__global__ __launch_bounds__(kWave) void score_pick_kernel(const double *__restrict__ partial, int qblocks,
                                                           const int32_t *__restrict__ lenA,
                                                           const int32_t *__restrict__ lenC,
                                                           const uint8_t *__restrict__ swap,
                                                           const float *__restrict__ cand, int B,
                                                           float *__restrict__ Tinit, double *__restrict__ initSum)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    const bool sw = swap != nullptr && swap[b] != 0;
    const float na = (float)(sw ? lenC[b] : lenA[b]);
    const float nc = (float)(sw ? lenA[b] : lenC[b]);
    float mean = 0.f;
    double total = 0.0;
    if (lane < 2 * kCand) {
        total = partial_total(partial, b * 12 + lane, qblocks, 0);
        mean = (float)total / ((lane & 1) ? nc : na);
    }
    int pick = 0;
    float best = 0.f;
    for (int k = 0; k < kCand; ++k) {
        const float sc = fminf(__shfl(mean, 2 * k, kWave), __shfl(mean, 2 * k + 1, kWave));
        if (k == 0 || sc < best) { best = sc; pick = k; }
    }
    // The forward total of the picked candidate IS the roll-back check's sum under the initial pose (utils_icp.py:28-29 on
    // src + t, the points utils_hist.py:86-89 has just scored: same queries in the same sorted order, same targets, same
    // block records) -- kept for the check sweep of the same call, which then scans under the final pose only.  A scan
    // that was pruned reports +inf (the pick can still fall on it through its backward mean): the check scans for itself.
    if (initSum != nullptr && lane == 2 * pick) initSum[b] = total;
    if (lane < 16) {
        const float *t = cand + ((size_t)b * kCand + pick) * 3;
        float v = (lane % 5 == 0) ? 1.f : 0.f;
        if (lane == 3) v = t[0];
        if (lane == 7) v = t[1];
        if (lane == 11) v = t[2];
        Tinit[(size_t)b * 16 + lane] = v;
    }
}

hipError_t launch_score_pick(const double *partial, int qblocks, const int32_t *lenA,
                             const int32_t *lenC, const uint8_t *swap, const float *cand, int B,
                             float *Tinit, hipStream_t s, double *initSum)
{
    hipLaunchKernelGGL(score_pick_kernel, dim3(B), dim3(kWave), 0, s, partial, qblocks, lenA, lenC, swap, cand, B,
                       Tinit, initSum);
    return hipGetLastError();
}

// M = [[R^T, T],[0 0 0 1]] * init   (utils_icp.py:60-65, :24; fp32 bmm order)
__global__ void compose_kernel(const IcpState *__restrict__ st, const float *__restrict__ init, int B,
                               float *__restrict__ M, const IcpCtrl *__restrict__ ctrl, int32_t *__restrict__ iters)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    // iterations of the batch (what icp_export reports); -1 = a team gave up waiting, transforms are NaN
    if (b == 0 && iters != nullptr) *iters = ctrl->error ? -1 : ctrl->iters;
    if (b >= B) return;
    float A[16];
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) A[i * 4 + j] = st[b].R[j * 3 + i];
        A[i * 4 + 3] = st[b].T[i];
    }
    A[12] = A[13] = A[14] = 0.f; A[15] = 1.f;
    if (ctrl != nullptr && ctrl->error) A[0] = __int_as_float(0x7fc00000);   // abandoned launch: poison every pose
    const float *I = init + (size_t)b * 16;
    float *o = M + (size_t)b * 16;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float acc = A[i * 4 + 0] * I[0 * 4 + j];
            acc = fmaf(A[i * 4 + 1], I[1 * 4 + j], acc);
            acc = fmaf(A[i * 4 + 2], I[2 * 4 + j], acc);
            acc = fmaf(A[i * 4 + 3], I[3 * 4 + j], acc);
            o[i * 4 + j] = acc;
        }
}

hipError_t launch_compose(const IcpState *state, const float *init, int B, float *M, hipStream_t s,
                          const IcpCtrl *ctrl, int32_t *iters)
{
    hipLaunchKernelGGL(compose_kernel, dim3((B + 127) / 128), dim3(128), 0, s, state, init, B, M, ctrl, iters);
    return hipGetLastError();
}

#ifdef ICPFLOW_REUSE_STATS
__device__ unsigned long long g_reuseStats[4];
#endif
// roll back where the ICP pose did not lower the mean NN error (utils_icp.py:27-35), then
// invert the pose of swapped pairs (utils_match.py:152-154; exact affine inverse in fp64)
__global__ void select_kernel(const double *__restrict__ partial, int qblocks,
                              const int32_t *__restrict__ lenA, const int32_t *__restrict__ lenC,
                              const uint8_t *__restrict__ swap, const float *__restrict__ init,
                              const float *__restrict__ M, int B, int invertSwapped, float *__restrict__ out,
                              PoseSource fused, int32_t *__restrict__ iters, const double *__restrict__ initSum,
                              const uint8_t *__restrict__ active)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    // fused finish (M == NULL): the composed pose comes straight from the ICP's history / state
    int n = 0;
    if (M == nullptr) {
        n = pose_stop_iteration_wave(fused, threadIdx.x & (kWave - 1));   // 64 tallies per round, every wave for itself
        // iterations of the batch; -1 = a team gave up waiting, transforms are NaN
        if (b == 0 && iters != nullptr) *iters = fused.ctrl->error ? -1 : n;
    }
    if (b >= B) return;
    float Mf[16];
    if (M == nullptr) final_pose(fused, b, n, Mf);
    const bool sw = swap != nullptr && swap[b] != 0;
    const float na = (float)(sw ? lenC[b] : lenA[b]);
    // (initSum: the scoring's forward total of the picked candidate, where the check sweep has left the scan under the initial
    // pose to it -- the same condition as in sweep_scan_kernel: finite, and the pair in the batch)
    double s0 = 0.0;
    bool reuse = false;
    if (initSum != nullptr && (active == nullptr || active[b] != 0)) {
        s0 = initSum[b];
        reuse = s0 - s0 == 0.0;   // finite
    }
    if (!reuse) s0 = partial_total(partial, b * 2 + 0, qblocks, 0);
#ifdef ICPFLOW_REUSE_STATS
    // [0] pairs whose sum came from the scoring, [1] pairs the check scanned although a total was offered (pruned scan / masked
    // pair), [2] pairs of calls without an offer, [3] roll-backs (tools/dbg/check_reuse_stats.py)
    atomicAdd(&g_reuseStats[reuse ? 0 : (initSum != nullptr ? 1 : 2)], 1ull);
#endif
    const float e0 = (float)s0 / na;
    const float e1 = (float)partial_total(partial, b * 2 + 1, qblocks, 0) / na;
    const float *src = (e1 >= e0) ? init + (size_t)b * 16 : (M != nullptr ? M + (size_t)b * 16 : Mf);  // NaN keeps ICP
#ifdef ICPFLOW_REUSE_STATS
    if (e1 >= e0) atomicAdd(&g_reuseStats[3], 1ull);
#endif
    float P[16];
    for (int k = 0; k < 16; ++k) P[k] = src[k];
    if (sw && invertSwapped) {
        const double a = P[0], bb = P[1], c = P[2], d = P[4], e = P[5], f = P[6], g = P[8], h = P[9],
                     i = P[10];
        const double tx = P[3], ty = P[7], tz = P[11];
        const double c00 = e * i - f * h, c01 = c * h - bb * i, c02 = bb * f - c * e;
        const double c10 = f * g - d * i, c11 = a * i - c * g, c12 = c * d - a * f;
        const double c20 = d * h - e * g, c21 = bb * g - a * h, c22 = a * e - bb * d;
        const double det = a * c00 + bb * c10 + c * c20;
        const double r = 1.0 / det;
        const double I00 = c00 * r, I01 = c01 * r, I02 = c02 * r;
        const double I10 = c10 * r, I11 = c11 * r, I12 = c12 * r;
        const double I20 = c20 * r, I21 = c21 * r, I22 = c22 * r;
        P[0] = (float)I00; P[1] = (float)I01; P[2] = (float)I02;
        P[4] = (float)I10; P[5] = (float)I11; P[6] = (float)I12;
        P[8] = (float)I20; P[9] = (float)I21; P[10] = (float)I22;
        P[3] = (float)(-(I00 * tx + I01 * ty + I02 * tz));
        P[7] = (float)(-(I10 * tx + I11 * ty + I12 * tz));
        P[11] = (float)(-(I20 * tx + I21 * ty + I22 * tz));
        P[12] = P[13] = P[14] = 0.f; P[15] = 1.f;
    }
    float *o = out + (size_t)b * 16;
    for (int k = 0; k < 16; ++k) o[k] = P[k];
}

hipError_t launch_select(const double *partial, int qblocks, const int32_t *lenA, const int32_t *lenC,
                         const uint8_t *swap, const float *init, const float *M, int B, int invertSwapped,
                         float *out, hipStream_t s, const PoseSource *fused, int32_t *iters, const double *initSum,
                         const uint8_t *active)
{
    if (M == nullptr && fused == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(select_kernel, dim3((B + 127) / 128), dim3(128), 0, s, partial, qblocks, lenA, lenC,
                       swap, init, M, B, invertSwapped, out, fused ? *fused : PoseSource{}, iters, initSum, active);
    return hipGetLastError();
}

// match_eval epilogue (utils_match.py:168-184)
// one wave per pair: lane j < 16 totals column j % 8 of job j / 8 (forward, backward) over the query blocks
__global__ __launch_bounds__(kWave) void eval_epilogue_kernel(const double *__restrict__ partial, int qblocks,
                                                              const int32_t *__restrict__ len1,
                                                              const int32_t *__restrict__ len2,
                                                              const float *__restrict__ T, int B,
                                                              float *__restrict__ errors, float *__restrict__ inliers,
                                                              float *__restrict__ ratios, float *__restrict__ ious,
                                                              float *__restrict__ translations,
                                                              float *__restrict__ rotations)
{
    const int b = blockIdx.x, lane = threadIdx.x;
    double tot = 0.0;
    if (lane < 2 * kPartial) tot = partial_total(partial, b * 2 + lane / kPartial, qblocks, lane % kPartial);
    const float n1 = (float)len1[b], n2 = (float)len2[b];
    const float n12 = (float)(len1[b] + len2[b]);
    const float sum1 = (float)__shfl(tot, 0, kWave), sum2 = (float)__shfl(tot, kPartial, kWave);
    const float in1 = (float)__shfl(tot, 1, kWave), in2 = (float)__shfl(tot, kPartial + 1, kWave);
    float moved[3], orig[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        moved[k] = (float)__shfl(tot, 2 + k, kWave) / n1;   // :180
        orig[k] = (float)__shfl(tot, 5 + k, kWave) / n1;    // :181
    }
    if (lane != 0) return;
    errors[b * 2 + 0] = sum1 / n1;                                           // :177
    errors[b * 2 + 1] = sum2 / n2;                                           // :178
    inliers[b * 2 + 0] = in1;
    inliers[b * 2 + 1] = in2;
    ratios[b * 2 + 0] = in1 / n1;                                            // :171
    ratios[b * 2 + 1] = in2 / n2;                                            // :172
    ious[b * 2 + 0] = in1 / (n12 - in2);                                     // :174
    ious[b * 2 + 1] = in2 / (n12 - in1);                                     // :175
    for (int k = 0; k < 3; ++k) translations[b * 3 + k] = moved[k] - orig[k];  // :183
    const float *M = T + (size_t)b * 16;
    // pytorch3d matrix_to_euler_angles(M[0:3,0:3], 'ZYX'), then * 180. / np.pi  (:184)
    const float pi = 3.14159265358979323846f;
    rotations[b * 3 + 0] = (atan2f(M[4], M[0]) * 180.0f) / pi;
    rotations[b * 3 + 1] = (asinf(-M[8]) * 180.0f) / pi;
    rotations[b * 3 + 2] = (atan2f(M[9], M[10]) * 180.0f) / pi;
}

hipError_t launch_eval_epilogue(const double *partial, int qblocks, const int32_t *len1,
                                const int32_t *len2, const float *T, int B, float *errors,
                                float *inliers, float *ratios, float *ious, float *translations,
                                float *rotations, hipStream_t s)
{
    hipLaunchKernelGGL(eval_epilogue_kernel, dim3(B), dim3(kWave), 0, s, partial, qblocks, len1, len2, T, B, errors,
                       inliers, ratios, ious, translations, rotations);
    return hipGetLastError();
}

// transform_points_batch (utils_helper.py:76-87); in-place safe (one thread per row)
__global__ void transform_points_kernel(const float4 *xyz, const float *__restrict__ pose, int N, float4 *out)
{
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const Affine a = affine_from_pose(pose + (size_t)b * 16);
    const float4 p = xyz[(size_t)b * N + i];
    float4 o;
    affine_apply(a, p.x, p.y, p.z, o.x, o.y, o.z);
    o.w = p.w;
    out[(size_t)b * N + i] = o;
}

hipError_t launch_transform_points(const float *xyz, const float *pose, int B, int N, float *out,
                                   hipStream_t s)
{
    hipLaunchKernelGGL(transform_points_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s,
                       (const float4 *)xyz, pose, N, (float4 *)out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// host-association helpers: build the padded [B,N,4] batch of match_pairs
// (utils_match.py:81-91, pad_segment utils_helper.py:185-196) and the per-point flow
// (utils_flow.py:57-69) on the device.
// ---------------------------------------------------------------------------------
// rows[b,i] = row of `points` ([M,3]) that becomes point i of pair b, or -1 for a pad row
__global__ void gather_pad_kernel(const float *__restrict__ points, const int32_t *__restrict__ rows,
                                  size_t total, float4 *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int r = rows[t];
    float4 o = make_float4(1e8f, 1e8f, 1e8f, 0.f);                    // utils_helper.py:191-192
    if (r >= 0) o = make_float4(points[(size_t)r * 3 + 0], points[(size_t)r * 3 + 1], points[(size_t)r * 3 + 2], 1.f);
    out[t] = o;
}

hipError_t launch_gather_pad(const float *points, const int32_t *rows, int B, int N, float *out, hipStream_t s)
{
    const size_t total = (size_t)B * N;
    hipLaunchKernelGGL(gather_pad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, points, rows,
                       total, (float4 *)out);
    return hipGetLastError();
}

// pad_segment of whole clusters straight from the label-sorted row table (utils_match.py:81-91,
// utils_helper.py:185-201): pair b takes `count` rows of its cluster, order[start + i] -- or, for a
// cluster longer than N that the host subsampled (random_choice), order[start + perm[off + i]].
// seg: int64 [3,B] = start, count (already clipped to N), offset into perm or -1.
__global__ void gather_segments_kernel(const float *__restrict__ points, const int64_t *__restrict__ order,
                                       const int64_t *__restrict__ seg, const int32_t *__restrict__ perm, int B,
                                       int N, float4 *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * N) return;
    const int b = (int)(t / N), i = (int)(t % N);
    const int64_t start = seg[b], count = seg[B + b], off = seg[2 * B + b];
    float4 o = make_float4(1e8f, 1e8f, 1e8f, 0.f);                    // utils_helper.py:191-192
    if (i < count) {
        const int64_t r = order[start + (off >= 0 ? (int64_t)perm[off + i] : (int64_t)i)];
        o = make_float4(points[r * 3 + 0], points[r * 3 + 1], points[r * 3 + 2], 1.f);
    }
    out[t] = o;
}

hipError_t launch_gather_segments(const float *points, const int64_t *order, const int64_t *seg, const int32_t *perm,
                                  int B, int N, float *out, hipStream_t s)
{
    const size_t total = (size_t)B * N;
    hipLaunchKernelGGL(gather_segments_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, points, order,
                       seg, perm, B, N, (float4 *)out);
    return hipGetLastError();
}

// per-cluster statistics consumed by sanity_check (utils_check.py:34-43): centroid and the
// ascending-sorted axis-aligned bbox extents (get_bbox_tensor, utils_helper.py:166-170) of every
// cluster of a labelled cloud.  order = rows sorted by label; cluster c owns order[start[c] ..
// start[c]+count[c]).  One workgroup per cluster, sums in fp64 (order independent to fp32 rounding).
constexpr int kStatsBlock = 1024;
__global__ __launch_bounds__(kStatsBlock) void cluster_stats_kernel(
    const float *__restrict__ points, const int64_t *__restrict__ order, const int64_t *__restrict__ start,
    const int64_t *__restrict__ count, const float *__restrict__ labels, float *__restrict__ mean,
    float *__restrict__ extent)
{
    __shared__ double ssum[kStatsBlock / kWave][3];
    __shared__ float smin[kStatsBlock / kWave][3], smax[kStatsBlock / kWave][3];
    const int c = blockIdx.x;
    const int64_t s0 = start[c], n = count[c];
    // ground (-1e8) and noise (-1) are never candidates (utils_check.py:32) -- and by far the largest
    // "clusters" of a frame: skip them
    if (labels != nullptr && labels[c] < 0.0f) {
        if (threadIdx.x < 3) { mean[(size_t)c * 3 + threadIdx.x] = 0.f; extent[(size_t)c * 3 + threadIdx.x] = 0.f; }
        return;
    }
    double sum[3] = {0.0, 0.0, 0.0};
    float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int64_t i = threadIdx.x; i < n; i += kStatsBlock) {
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
            for (int w = 1; w < kStatsBlock / kWave; ++w) {
                t += ssum[w][k];
                lo = fminf(lo, smin[w][k]);
                hi = fmaxf(hi, smax[w][k]);
            }
            mean[(size_t)c * 3 + k] = (float)(t / (double)n);
            e[k] = fabsf(hi - lo);
        }
        if (e[0] > e[1]) { const float t = e[0]; e[0] = e[1]; e[1] = t; }
        if (e[1] > e[2]) { const float t = e[1]; e[1] = e[2]; e[2] = t; }
        if (e[0] > e[1]) { const float t = e[0]; e[0] = e[1]; e[1] = t; }
        for (int k = 0; k < 3; ++k) extent[(size_t)c * 3 + k] = e[k];
    }
}

hipError_t launch_cluster_stats(const float *points, const int64_t *order, const int64_t *start,
                                const int64_t *count, const float *labels, int L, float *mean, float *extent,
                                hipStream_t s)
{
    hipLaunchKernelGGL(cluster_stats_kernel, dim3(L), dim3(kStatsBlock), 0, s, points, order, start, count, labels,
                       mean, extent);
    return hipGetLastError();
}

constexpr int kFlowBlock = 256;
constexpr int kFlowTile = 2048;   // pair labels cached in LDS, one tile at a time (any number of pairs)

// Every point looks its label up among the matched source labels (pairRows[p * pairStride]: a column of the [P,10] pair
// rows, or a plain array) and moves with M = T[p] * pose, a point of an unmatched cluster with M = I * pose (T_per_point
// starts as the identity, utils_flow.py:62-65); the product is formed per point, in the fp32 order of the bmm -- one
// launch, no scratch
__global__ __launch_bounds__(kFlowBlock) void flow_rigid_kernel(
    const float *__restrict__ points, const float *__restrict__ labels, int N,
    const float *__restrict__ pairRows, int pairStride, int P, const float *__restrict__ T,
    const float *__restrict__ pose, float *__restrict__ flow)
{
    __shared__ float lab[kFlowTile];
    const int i = blockIdx.x * kFlowBlock + threadIdx.x;
    const float l = i < N ? labels[i] : 0.f;
    int p = P;                                   // not matched: pose only
    for (int k0 = 0; k0 < P; k0 += kFlowTile) {
        const int kn = min(kFlowTile, P - k0);
        if (k0 > 0) __syncthreads();
        for (int k = threadIdx.x; k < kn; k += kFlowBlock) lab[k] = pairRows[(size_t)(k0 + k) * pairStride];
        __syncthreads();
        for (int k = 0; k < kn; ++k)
            if (lab[k] == l) p = k0 + k;         // pairs[:,0] holds each source label at most once
    }
    if (i >= N) return;
    float m[12];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = (p < P) ? T[(size_t)p * 16 + r * 4 + k] : ((k == r) ? 1.f : 0.f);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float acc = a[0] * pose[0 * 4 + c];
            acc = fmaf(a[1], pose[1 * 4 + c], acc);
            acc = fmaf(a[2], pose[2 * 4 + c], acc);
            acc = fmaf(a[3], pose[3 * 4 + c], acc);
            m[r * 4 + c] = acc;
        }
    }
    const float x = points[(size_t)i * 3 + 0], y = points[(size_t)i * 3 + 1], z = points[(size_t)i * 3 + 2];
    // (T pose [x y z 1]^T)[0:3] - p, utils_flow.py:67-68
    flow[(size_t)i * 3 + 0] = fmaf(1.f, m[3], fmaf(z, m[2], fmaf(y, m[1], x * m[0]))) - x;
    flow[(size_t)i * 3 + 1] = fmaf(1.f, m[7], fmaf(z, m[6], fmaf(y, m[5], x * m[4]))) - y;
    flow[(size_t)i * 3 + 2] = fmaf(1.f, m[11], fmaf(z, m[10], fmaf(y, m[9], x * m[8]))) - z;
}

hipError_t launch_flow_rigid(const float *points, const float *labels, int N, const float *pairRows, int pairStride,
                             const float *T, int P, const float *pose, float *flow, hipStream_t s)
{
    hipLaunchKernelGGL(flow_rigid_kernel, dim3((N + kFlowBlock - 1) / kFlowBlock), dim3(kFlowBlock), 0, s, points,
                       labels, N, pairRows, pairStride, P, T, pose, flow);
    return hipGetLastError();
}

}  // namespace icpflow

#ifdef ICPFLOW_REUSE_STATS
extern "C" int icpflow_debug_reuse_stats(unsigned long long *out4, int reset)
{
    (void)hipDeviceSynchronize();
    const int rc = (int)hipMemcpyFromSymbol(out4, HIP_SYMBOL(icpflow::g_reuseStats), sizeof(icpflow::g_reuseStats));
    if (reset) {
        const unsigned long long z[4] = {0, 0, 0, 0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(icpflow::g_reuseStats), z, sizeof(z));
    }
    return rc;
}
#endif
