// assoc.hip -- the host half of an association stage as kernels (round 4).
//
// Reference: match_pairs, utils_match.py:96-115 (reject test check_transformation utils_check.py:51-66, S x D error matrices,
// row arg-min utils_helper.py:108-110, threshold :112) and match_pcds' step from stage 1 to stage 2 (utils_match.py:42-53: only
// clusters that found no partner go on).  The Python mirror does this with numpy on the results of a stage brought to the host
// (utils_match._finish_pairs); here ONE workgroup does it where the results are, so that match_pcds can enqueue stage 1, the
// step, stage 2, the pair rows and the flow without a device -> host hand-over in between.  The stage-2 candidates are a
// SUPERSET sized on the host (every pair the sanity grid lets through); the step switches each of them on or off
// (icpflow_options_t::d_pair_active keeps the switched-off ones out of the ICP's batch-global stop).
#include "common.hpp"
#include "kernels.hpp"

namespace icpflow {
namespace {

constexpr int kAssocBlock = 1024;
constexpr int kAssocRows = 1024;   // cluster rows per table the kernels hold in LDS

struct AssignParams {
    const float *r; const int32_t *si, *di; int K; const uint8_t *active;
    int S, D; float tf, iouMin, rotMax, errMax;
    int32_t *best;
    int K2; const int32_t *si2, *di2; int64_t *seg2; uint8_t *active2;
};

// check_transformation (utils_check.py:51-66) of candidate k in the fp32 arithmetic of the numpy mirror (utils_check.py of this
// package): ~(sqrt((t0*t0 + t1*t1) + t2*t2) > tf) & ~(min(iou) < thres_iou) & ~(max(|rot_1|, |rot_2|) > thres_rot * 90).  numpy's
// minimum / maximum PROPAGATE a NaN and every comparison with a NaN is false, i.e. a NaN passes the test it sits in; fminf /
// fmaxf drop a NaN, so the NaN cases are spelled out.  err = min(error_src, error_dst), NaN if either is.
__device__ __forceinline__ bool assoc_keep(const AssignParams &p, int k, float &err)
{
    const int K = p.K;
    const float *errors = p.r + 16 * (size_t)K, *ious = p.r + 22 * (size_t)K, *tr = p.r + 24 * (size_t)K, *rot = p.r + 27 * (size_t)K;
    const float t0 = tr[3 * k], t1 = tr[3 * k + 1], t2 = tr[3 * k + 2];
    const float nrm = sqrtf((t0 * t0 + t1 * t1) + t2 * t2);
    const float i0 = ious[2 * k], i1 = ious[2 * k + 1];
    const float r1 = rot[3 * k + 1], r2 = rot[3 * k + 2];
    const float e0 = errors[2 * k], e1 = errors[2 * k + 1];
    err = (e0 != e0 || e1 != e1) ? __int_as_float(0x7fc00000) : fminf(e0, e1);
    const bool farOff = nrm > p.tf;                                                   // (NaN: false)
    // (Python's builtin min(iou) on the two-element tensor, utils_match.py:99: iou[1] if iou[1] < iou[0] else iou[0] -- a NaN in
    // iou[1] leaves iou[0] to be tested, a NaN in iou[0] is the result and passes the comparison)
    const bool lowIou = ((i1 < i0) ? i1 : i0) < p.iouMin;
    const bool turned = (r1 == r1 && r2 == r2) && fmaxf(fabsf(r1), fabsf(r2)) > p.rotMax;
    return !farOff && !lowIou && !turned;
}

__global__ __launch_bounds__(kAssocBlock) void assoc_assign_kernel(AssignParams p)
{
    __shared__ unsigned long long key[kAssocRows];   // per source row: (error bits << 32) | destination row of the best kept candidate
    __shared__ unsigned char nanRow[kAssocRows], mS[kAssocRows], mD[kAssocRows];
    const int tid = threadIdx.x;
    for (int s = tid; s < kAssocRows; s += kAssocBlock) { key[s] = ~0ull; nanRow[s] = 0; mS[s] = 0; mD[s] = 0; }
    for (int s = tid; s < p.S; s += kAssocBlock) p.best[s] = -1;
    __syncthreads();
    for (int k = tid; k < p.K; k += kAssocBlock) {
        if (p.active != nullptr && p.active[k] == 0) continue;
        float err;
        if (!assoc_keep(p, k, err)) continue;
        const int s = p.si[k];
        if (err != err) { nanRow[s] = 1; continue; }   // np.argmin takes a NaN as the minimum, and NaN < thres_error is false: no match
        // (errors are mean distances: >= 0, so their bit patterns order like the values; -0.0 cannot occur)
        atomicMin(&key[s], ((unsigned long long)(unsigned)__float_as_int(err) << 32) | (unsigned)p.di[k]);
    }
    __syncthreads();
    for (int k = tid; k < p.K; k += kAssocBlock) {
        if (p.active != nullptr && p.active[k] == 0) continue;
        float err;
        if (!assoc_keep(p, k, err) || err != err) continue;
        const int s = p.si[k], d = p.di[k];
        if (nanRow[s]) continue;
        if (key[s] != (((unsigned long long)(unsigned)__float_as_int(err) << 32) | (unsigned)d)) continue;
        if (!(err < p.errMax)) continue;                // utils_match.py:112
        p.best[s] = k;                                  // (the (source, destination) pairs of a stage are distinct: one writer)
        mS[s] = 1; mD[d] = 1;
    }
    __syncthreads();
    // the next stage's candidates: both clusters still without a partner (utils_match.py:45-53)
    for (int k = tid; k < p.K2; k += kAssocBlock) {
        const bool on = !mS[p.si2[k]] && !mD[p.di2[k]];
        p.active2[k] = on ? 1 : 0;
        if (!on) { p.seg2[(0 * 3 + 1) * (size_t)p.K2 + k] = 0; p.seg2[(1 * 3 + 1) * (size_t)p.K2 + k] = 0; }
    }
}

struct CollectParams {
    const int32_t *best1; const float *r1; const int32_t *si1, *di1; int K1;
    const int32_t *best2; const float *r2; const int32_t *si2, *di2; int K2;
    const double *srcTable, *dstTable; int stride, S, cap;
    float *rows, *T; int32_t *count;
};

__global__ __launch_bounds__(kAssocBlock) void assoc_collect_kernel(CollectParams p)
{
    __shared__ int pos[2 * kAssocRows + 1];
    const int tid = threadIdx.x;
    // flags: stage 1 rows then stage 2 rows; exclusive prefix sum by one thread per 2S <= 2048 entries is too slow: blocked scan
    const int n = 2 * p.S;
    for (int i = tid; i < n; i += kAssocBlock) {
        const int s = i < p.S ? i : i - p.S;
        const int32_t *best = i < p.S ? p.best1 : p.best2;
        pos[i] = (best != nullptr && best[s] >= 0) ? 1 : 0;
    }
    __syncthreads();
    if (tid == 0) {   // (n <= 2048 additions once per frame pair; the block waits ~2 us)
        int run = 0;
        for (int i = 0; i < n; ++i) { const int v = pos[i]; pos[i] = run; run += v; }
        pos[n] = run;
    }
    __syncthreads();
    const int P = pos[n];
    for (int i = tid; i < n; i += kAssocBlock) {
        const int s = i < p.S ? i : i - p.S;
        const bool second = i >= p.S;
        const int32_t *best = second ? p.best2 : p.best1;
        if (best == nullptr || best[s] < 0) continue;
        const int k = best[s], K = second ? p.K2 : p.K1;
        const float *r = second ? p.r2 : p.r1;
        const int d = (second ? p.di2 : p.di1)[k];
        const int row = pos[i];
        if (row >= p.cap) continue;
        float *o = p.rows + (size_t)row * 10;
        o[0] = (float)p.srcTable[(size_t)s * p.stride];
        o[1] = (float)p.dstTable[(size_t)d * p.stride];
        const float *errors = r + 16 * (size_t)K, *inl = r + 18 * (size_t)K, *rat = r + 20 * (size_t)K, *iou = r + 22 * (size_t)K;
        o[2] = errors[2 * k]; o[3] = errors[2 * k + 1]; o[4] = inl[2 * k]; o[5] = inl[2 * k + 1];
        o[6] = rat[2 * k]; o[7] = rat[2 * k + 1]; o[8] = iou[2 * k]; o[9] = iou[2 * k + 1];
        for (int c = 0; c < 16; ++c) p.T[(size_t)row * 16 + c] = r[(size_t)k * 16 + c];
    }
    for (int row = P + tid; row < p.cap; row += kAssocBlock) {      // rows nobody matches: a label no point carries, the identity
        float *o = p.rows + (size_t)row * 10;
        o[0] = -3.0e38f; o[1] = -3.0e38f;
        for (int c = 2; c < 10; ++c) o[c] = 0.f;
        for (int c = 0; c < 16; ++c) p.T[(size_t)row * 16 + c] = (c % 5 == 0) ? 1.f : 0.f;
    }
    if (tid == 0) {
        const int it1 = p.K1 > 0 ? __float_as_int(p.r1[30 * (size_t)p.K1]) : 0;
        const int it2 = (p.r2 != nullptr && p.K2 > 0) ? __float_as_int(p.r2[30 * (size_t)p.K2]) : 0;
        *p.count = (it1 < 0 || it2 < 0) ? -1 : min(P, p.cap);
    }
}

}  // namespace

hipError_t launch_assoc_assign(const float *r, const int32_t *si, const int32_t *di, int K, const uint8_t *active, int S, int D,
                               float tf, float iouMin, float rotMax, float errMax, int32_t *best, int K2, const int32_t *si2,
                               const int32_t *di2, int64_t *seg2, uint8_t *active2, hipStream_t s)
{
    AssignParams p{r, si, di, K, active, S, D, tf, iouMin, rotMax, errMax, best, K2, si2, di2, seg2, active2};
    hipLaunchKernelGGL(assoc_assign_kernel, dim3(1), dim3(kAssocBlock), 0, s, p);
    return hipGetLastError();
}

hipError_t launch_assoc_collect(const int32_t *best1, const float *r1, const int32_t *si1, const int32_t *di1, int K1,
                                const int32_t *best2, const float *r2, const int32_t *si2, const int32_t *di2, int K2,
                                const double *srcTable, const double *dstTable, int stride, int S, int cap, float *rows,
                                float *T, int32_t *count, hipStream_t s)
{
    CollectParams p{best1, r1, si1, di1, K1, best2, r2, si2, di2, K2, srcTable, dstTable, stride, S, cap, rows, T, count};
    hipLaunchKernelGGL(assoc_collect_kernel, dim3(1), dim3(kAssocBlock), 0, s, p);
    return hipGetLastError();
}

int assoc_max_rows() { return kAssocRows; }

}  // namespace icpflow
