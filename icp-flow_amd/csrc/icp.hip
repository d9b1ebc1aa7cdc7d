// icp.hip -- masked batched point-to-point ICP for gfx950 (a-5, a-6, a-7).
//
// Reference semantics: utils_icp_pytorch3d.py:100-225 (loop), :303-382 (Kabsch via SVD),
// :385-396 (apply).  Design: one workgroup per cluster pair runs a whole ICP iteration --
// NN scan of the moved source against the LDS-staged target (scan.hpp), inlier gate,
// weighted centroids, centred 3x3 covariance, closed-form rotation, rmse -- so the 15-odd
// torch kernels, the cuSOLVER call and the host sync of one reference iteration collapse
// into one launch with three block reductions.  All sums and the 3x3 solve are fp64
// (fp32 inputs), i.e. at least as accurate as the reference's fp32 torch reductions.
//
// Stopping: ICPFLOW_STOP_REFERENCE reproduces the batch-global rule (stop when every pair
// has rel <= thr, :209) WITHOUT a host round trip: one launch per iteration is enqueued up
// front; each pair that is not converged bumps ctrl->notconv[it]; the launch of iteration
// it+1 returns immediately when notconv[it] == 0.  ICPFLOW_STOP_PER_PAIR loops inside one
// launch and lets every pair stop on its own.
#include <vector>

#include "scan.hpp"
#include "kernels.hpp"

namespace icpflow {

// ---------------------------------------------------------------------------------
// 3x3 Kabsch rotation, row-vector convention y = x R (R = U diag(1,1,det(U V^T)) V^T for
// H = U S V^T, utils_icp_pytorch3d.py:339-362).  One-sided Jacobi in fp64.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ void kabsch_rotation(const double (&Hin)[9], double (&R)[9])
{
    // columns of A (A = H V  ->  U S) and of V, stored column-major: a[c][r]
    double a[3][3], v[3][3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            a[c][r] = Hin[r * 3 + c];
            v[c][r] = (r == c) ? 1.0 : 0.0;
        }
    for (int sweep = 0; sweep < 16; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0;
            const int q = (pq == 0) ? 1 : 2;
            const double al = a[p][0] * a[p][0] + a[p][1] * a[p][1] + a[p][2] * a[p][2];
            const double be = a[q][0] * a[q][0] + a[q][1] * a[q][1] + a[q][2] * a[q][2];
            const double ga = a[p][0] * a[q][0] + a[p][1] * a[q][1] + a[p][2] * a[q][2];
            if (ga == 0.0 || fabs(ga) <= 1e-16 * sqrt(al * be)) continue;
            rotated = true;
            const double zeta = (be - al) / (2.0 * ga);
            const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
            const double cs = 1.0 / sqrt(1.0 + t * t);
            const double sn = cs * t;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double ap = a[p][r], aq = a[q][r];
                a[p][r] = cs * ap - sn * aq;
                a[q][r] = sn * ap + cs * aq;
                const double vp = v[p][r], vq = v[q][r];
                v[p][r] = cs * vp - sn * vq;
                v[q][r] = sn * vp + cs * vq;
            }
        }
        if (!rotated) break;
    }
    double s[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) s[c] = sqrt(a[c][0] * a[c][0] + a[c][1] * a[c][1] + a[c][2] * a[c][2]);
    // order singular values descending (the reflection fix acts on the SMALLEST one);
    // an odd permutation flips det(V)
    double detV = 1.0;
#define ICPFLOW_SWAP_COLS(i, j)                                                            \
    if (s[i] < s[j]) {                                                                      \
        const double ts = s[i]; s[i] = s[j]; s[j] = ts;                                     \
        for (int r = 0; r < 3; ++r) {                                                       \
            const double ta = a[i][r]; a[i][r] = a[j][r]; a[j][r] = ta;                     \
            const double tv = v[i][r]; v[i][r] = v[j][r]; v[j][r] = tv;                     \
        }                                                                                   \
        detV = -detV;                                                                       \
    }
    ICPFLOW_SWAP_COLS(0, 1)
    ICPFLOW_SWAP_COLS(0, 2)
    ICPFLOW_SWAP_COLS(1, 2)
#undef ICPFLOW_SWAP_COLS
    if (!(s[0] > 0.0)) {  // H == 0 (no inliers): torch.svd(0) gives U = V = I  ->  R = I
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        return;
    }
    double u0[3], u1[3], u2[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) u0[r] = a[0][r] / s[0];
    if (s[1] > 1e-300 && s[1] > 1e-14 * s[0]) {
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] = a[1][r] / s[1];
    } else {  // rank 1: rotation not unique (reference: backend dependent); any unit vector _|_ u0
        const int k = (fabs(u0[0]) <= fabs(u0[1]) && fabs(u0[0]) <= fabs(u0[2])) ? 0
                      : (fabs(u0[1]) <= fabs(u0[2]) ? 1 : 2);
        double e[3] = {0.0, 0.0, 0.0};
        e[k] = 1.0;
        cross3(u0, e, u1);
        const double n = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) u1[r] /= n;
    }
    // third left vector: +-(u0 x u1); the sign (= det U) follows the computed column when
    // it carries information, and cancels in R either way
    cross3(u0, u1, u2);
    double detU = 1.0;
    if (a[2][0] * u2[0] + a[2][1] * u2[1] + a[2][2] * u2[2] < 0.0) {
        detU = -1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) u2[r] = -u2[r];
    }
    const double d = detU * detV;  // det(U V^T), :358-359
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            R[i * 3 + j] = u0[i] * v[0][j] + u1[i] * v[1][j] + d * u2[i] * v[2][j];  // :362
}

// ---------------------------------------------------------------------------------
struct IcpParams {
    const float *X;        // [B,N,4] moving cloud
    const float *Y;        // [B,N,4] fixed cloud
    const int32_t *lenX;
    const int32_t *lenY;
    const uint8_t *swap;   // optional: exchange X and Y roles per pair
    const float *prePose;  // optional [B,4,4]: X0 = transform_points_batch(X, prePose)
    int N;
    float thr2;            // fp32(thres**2), :160
    float relThr;          // fp32(relative_rmse_thr), :209
    int stopMode;
    int maxIter;
    IcpState *state;       // [B]
    IcpCtrl *ctrl;
    int32_t *nnj;          // [B,N] scratch, used when a pair needs more than one query group
};

#ifdef ICPFLOW_PHASE_TIMING
// debug builds only (tools/dbg/phase_timing.py): shader-clock stamps of workgroup 0
__device__ long long g_phase_stamps[16];
#define ICPFLOW_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_phase_stamps[k] = clock64(); } while (0)
#else
#define ICPFLOW_STAMP(k) do { } while (0)
#endif

template <int BLOCK, int Q>
__global__ __launch_bounds__(BLOCK) void icp_kernel(IcpParams p, int itBegin, int itEnd)
{
    ICPFLOW_STAMP(0);
    __shared__ ScanTile tileMem;
    ScanTile *tile = &tileMem;
    __shared__ double red[(BLOCK / kWave) * 9];

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    IcpCtrl *ctrl = p.ctrl;
    if (p.stopMode == ICPFLOW_STOP_REFERENCE_ && itBegin > 0) {
        // previous iteration satisfied the batch-global rule (or an earlier one did)
        if (ctrl->done || ctrl->notconv[itBegin - 1] == 0) {
            if (tid == 0 && b == 0) ctrl->done = 1;
            return;
        }
    }
    const bool sw = p.swap != nullptr && p.swap[b] != 0;
    CloudView xc, yc;
    xc.base = (sw ? p.Y : p.X) + (size_t)b * p.N * 4; xc.stride = 4; xc.n = (sw ? p.lenY : p.lenX)[b];
    yc.base = (sw ? p.X : p.Y) + (size_t)b * p.N * 4; yc.stride = 4; yc.n = (sw ? p.lenX : p.lenY)[b];
    PointXf pre;
    pre.kind = p.prePose ? XF_AFFINE : XF_NONE;
    pre.a = p.prePose ? affine_from_pose(p.prePose + (size_t)b * 16) : affine_identity();
    PointXf none;
    none.kind = XF_NONE;
    none.a = affine_identity();

    IcpState *st = p.state + b;
    float Rf[9], Tf[3], prev;
    int active;
    if (itBegin == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rf[k] = (k % 4 == 0) ? 1.f : 0.f;  // :140
        Tf[0] = Tf[1] = Tf[2] = 0.f;
        prev = 0.f;
        active = 1;
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) Rf[k] = st->R[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) Tf[k] = st->T[k];
        prev = st->rmse;
        active = st->active;
    }
    float rmse = prev;
    int itersDone = (itBegin == 0) ? 0 : st->iters;

    const int per = BLOCK * Q;
    const int ngroups = (xc.n + per - 1) / per;
    int32_t *nnj = p.nnj + (size_t)b * p.N;

    for (int it = itBegin; it < itEnd; ++it) {
        if (p.stopMode == ICPFLOW_STOP_PER_PAIR_ && !active) break;
        // ---------------- pass 1: NN + gate + first moments ---------------------------
        double s7[7] = {0, 0, 0, 0, 0, 0, 0};
        float x0x[Q], x0y[Q], x0z[Q], ynx[Q], yny[Q], ynz[Q];
        bool w[Q];
        for (int g = 0; g < ngroups; ++g) {
            float qx[Q], qy[Q], qz[Q];
            bool live[Q];
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int i = g * per + q * BLOCK + tid;
                live[q] = i < xc.n;
                x0x[q] = x0y[q] = x0z[q] = 0.f;
                qx[q] = qy[q] = qz[q] = 0.f;
                if (live[q]) {
                    float rx, ry, rz;
                    cloud_load(xc, i, rx, ry, rz);
                    xf_apply(pre, rx, ry, rz, x0x[q], x0y[q], x0z[q]);  // utils_icp.py:21
                    // Xt = X0 R + T  (:177, :395), bmm order
                    qx[q] = fmaf(x0z[q], Rf[6], fmaf(x0y[q], Rf[3], x0x[q] * Rf[0])) + Tf[0];
                    qy[q] = fmaf(x0z[q], Rf[7], fmaf(x0y[q], Rf[4], x0x[q] * Rf[1])) + Tf[1];
                    qz[q] = fmaf(x0z[q], Rf[8], fmaf(x0y[q], Rf[5], x0x[q] * Rf[2])) + Tf[2];
                }
            }
            ScanAcc<Q> acc;
            ICPFLOW_STAMP(1);
            scan_cloud<Q>(yc, none, tile, qx, qy, qz, acc);  // :154-157
            ICPFLOW_STAMP(2);
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                w[q] = live[q] && (acc.best[q] <= p.thr2);  // :160-161
                ynx[q] = yny[q] = ynz[q] = 0.f;
                int j = -1;
                if (w[q]) {
                    j = scan_resolve(yc, none, qx[q], qy[q], qz[q], acc.best[q], acc.chunk[q],
                                     ynx[q], yny[q], ynz[q]);
                    s7[0] += 1.0;
                    s7[1] += (double)x0x[q]; s7[2] += (double)x0y[q]; s7[3] += (double)x0z[q];
                    s7[4] += (double)ynx[q]; s7[5] += (double)yny[q]; s7[6] += (double)ynz[q];
                }
                if (ngroups > 1) {
                    const int i = g * per + q * BLOCK + tid;
                    if (live[q]) nnj[i] = j;
                }
            }
        }
        ICPFLOW_STAMP(3);
        block_sum<7, double>(s7, red);
        ICPFLOW_STAMP(4);
        const double wsum = s7[0] > 1e-9 ? s7[0] : 1e-9;  // clamp(eps), :314-315, :326
        const double mux = s7[1] / wsum, muy = s7[2] / wsum, muz = s7[3] / wsum;
        const double nux = s7[4] / wsum, nuy = s7[5] / wsum, nuz = s7[6] / wsum;

        // ---------------- pass 2: centred covariance, :318-336 ---------------------------
        double h9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int g = 0; g < ngroups; ++g) {
            if (ngroups > 1) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int i = g * per + q * BLOCK + tid;
                    w[q] = false;
                    if (i < xc.n) {
                        const int j = nnj[i];
                        if (j >= 0) {
                            float rx, ry, rz;
                            cloud_load(xc, i, rx, ry, rz);
                            xf_apply(pre, rx, ry, rz, x0x[q], x0y[q], x0z[q]);
                            cloud_load(yc, j, ynx[q], yny[q], ynz[q]);
                            w[q] = true;
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (!w[q]) continue;
                const double cx = (double)x0x[q] - mux, cy = (double)x0y[q] - muy, cz = (double)x0z[q] - muz;
                const double ex = (double)ynx[q] - nux, ey = (double)yny[q] - nuy, ez = (double)ynz[q] - nuz;
                h9[0] += cx * ex; h9[1] += cx * ey; h9[2] += cx * ez;
                h9[3] += cy * ex; h9[4] += cy * ey; h9[5] += cy * ez;
                h9[6] += cz * ex; h9[7] += cz * ey; h9[8] += cz * ez;
            }
        }
        block_sum<9, double>(h9, red);
        ICPFLOW_STAMP(5);
#pragma unroll
        for (int k = 0; k < 9; ++k) h9[k] /= wsum;
        // every thread holds the same H: solve redundantly, no broadcast needed
        double Rd[9];
        kabsch_rotation(h9, Rd);
        ICPFLOW_STAMP(6);
        // T = mu_y - mu_x R, :376
        const double Td0 = nux - (mux * Rd[0] + muy * Rd[3] + muz * Rd[6]);
        const double Td1 = nuy - (mux * Rd[1] + muy * Rd[4] + muz * Rd[7]);
        const double Td2 = nuz - (mux * Rd[2] + muy * Rd[5] + muz * Rd[8]);
#pragma unroll
        for (int k = 0; k < 9; ++k) Rf[k] = (float)Rd[k];
        Tf[0] = (float)Td0; Tf[1] = (float)Td1; Tf[2] = (float)Td2;

        // ---------------- pass 3: rmse with the updated transform, :191-192 -----------------
        double e1[1] = {0.0};
        for (int g = 0; g < ngroups; ++g) {
            if (ngroups > 1) {
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    const int i = g * per + q * BLOCK + tid;
                    w[q] = false;
                    if (i < xc.n) {
                        const int j = nnj[i];
                        if (j >= 0) {
                            float rx, ry, rz;
                            cloud_load(xc, i, rx, ry, rz);
                            xf_apply(pre, rx, ry, rz, x0x[q], x0y[q], x0z[q]);
                            cloud_load(yc, j, ynx[q], yny[q], ynz[q]);
                            w[q] = true;
                        }
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                if (!w[q]) continue;
                const float tx = fmaf(x0z[q], Rf[6], fmaf(x0y[q], Rf[3], x0x[q] * Rf[0])) + Tf[0];
                const float ty = fmaf(x0z[q], Rf[7], fmaf(x0y[q], Rf[4], x0x[q] * Rf[1])) + Tf[1];
                const float tz = fmaf(x0z[q], Rf[8], fmaf(x0y[q], Rf[5], x0x[q] * Rf[2])) + Tf[2];
                const float dx = tx - ynx[q], dy = ty - yny[q], dz = tz - ynz[q];
                e1[0] += (double)(dx * dx + dy * dy + dz * dz);
            }
        }
        block_sum<1, double>(e1, red);
        ICPFLOW_STAMP(7);
        rmse = (float)sqrt(e1[0] / wsum);
        // relative rmse, :195-198 (fp32 like the reference's tensors)
        const float rel = (it == 0) ? 1.0f : (prev - rmse) / prev;
        const bool conv = rel <= p.relThr;  // NaN -> false, :209
        itersDone = it + 1;
        if (p.stopMode == ICPFLOW_STOP_REFERENCE_) {
            if (tid == 0 && !conv) atomicAdd(&ctrl->notconv[it], 1);
        } else {
            // per-pair rule: retire a pair once its rmse has stopped DEcreasing by more than
            // thr (0 <= rel <= thr).  A negative rel (rmse went up: the inlier set is still
            // changing) satisfies the reference's batch test but is not convergence of this
            // pair.  A constant (zero-inlier) pair has rel = NaN and is retired too.
            if (it > 0 && ((conv && rel >= 0.0f) || rel != rel)) active = 0;
        }
        prev = rmse;  // :213
    }
    ICPFLOW_STAMP(8);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) st->R[k] = Rf[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) st->T[k] = Tf[k];
        st->rmse = rmse;
        st->active = active;
        st->iters = itersDone;
        if (p.stopMode == ICPFLOW_STOP_REFERENCE_) {
            if (b == 0) ctrl->iters = itersDone;
        } else {
            atomicMax(&ctrl->iters, itersDone);
            if (active) atomicAdd(&ctrl->notconv[0], 1);  // pairs still moving at maxIter
        }
    }
}

__global__ void icp_export_kernel(const IcpState *__restrict__ st, const IcpCtrl *__restrict__ ctrl,
                                  int B, int stopMode, float *__restrict__ R, float *__restrict__ T,
                                  float *__restrict__ rmse, int32_t *__restrict__ iters,
                                  int32_t *__restrict__ converged)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        if (R) for (int k = 0; k < 9; ++k) R[(size_t)b * 9 + k] = st[b].R[k];
        if (T) for (int k = 0; k < 3; ++k) T[(size_t)b * 3 + k] = st[b].T[k];
        if (rmse) rmse[b] = st[b].rmse;
    }
    if (b == 0) {
        const int n = ctrl->iters;
        if (iters) *iters = n;
        if (converged) {
            if (stopMode == ICPFLOW_STOP_REFERENCE_) *converged = (n > 0 && ctrl->notconv[n - 1] == 0) ? 1 : 0;
            else *converged = (ctrl->notconv[0] == 0) ? 1 : 0;
        }
    }
}

template <int BLOCK, int Q>
static void launch_icp_variant(const IcpParams &p, int B, int itBegin, int itEnd, hipStream_t s)
{
    hipLaunchKernelGGL((icp_kernel<BLOCK, Q>), dim3(B), dim3(BLOCK), 0, s, p, itBegin, itEnd);
}

// ---- optional per-launch timing of this (dominant) kernel with HIP events ---------------------
// bench.py needs the average launch duration of the dominant kernel measured on the stream it
// runs on; the events are recorded by the library because only it sees the individual launches.
namespace {
struct LaunchProfile {
    std::vector<hipEvent_t> start, stop;
    int used = 0;
} g_prof;
}  // namespace

#ifdef ICPFLOW_PHASE_TIMING
extern "C" int icpflow_debug_phase_stamps(long long *out16)
{
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_stamps), sizeof(long long) * 16);
}
#endif

hipError_t profile_enable(int capacity)
{
    for (hipEvent_t e : g_prof.start) (void)hipEventDestroy(e);
    for (hipEvent_t e : g_prof.stop) (void)hipEventDestroy(e);
    g_prof.start.clear(); g_prof.stop.clear(); g_prof.used = 0;
    for (int i = 0; i < capacity; ++i) {
        hipEvent_t a, b;
        hipError_t e = hipEventCreate(&a);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&b);
        if (e != hipSuccess) return e;
        g_prof.start.push_back(a); g_prof.stop.push_back(b);
    }
    return hipSuccess;
}

hipError_t profile_collect(double *total_ms, int *launches)
{
    double sum = 0.0;
    for (int i = 0; i < g_prof.used; ++i) {
        hipError_t e = hipEventSynchronize(g_prof.stop[i]);
        if (e != hipSuccess) return e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]);
        if (e != hipSuccess) return e;
        sum += ms;
    }
    if (total_ms) *total_ms = sum;
    if (launches) *launches = g_prof.used;
    g_prof.used = 0;
    return hipSuccess;
}

static void launch_icp_iters(const IcpParams &p, int B, int itBegin, int itEnd, hipStream_t s)
{
    const bool timed = g_prof.used < (int)g_prof.start.size();
    if (timed) (void)hipEventRecord(g_prof.start[g_prof.used], s);
    if (p.N <= 256) launch_icp_variant<256, 1>(p, B, itBegin, itEnd, s);
    else if (p.N <= 512) launch_icp_variant<512, 1>(p, B, itBegin, itEnd, s);
    else if (p.N <= 1024) launch_icp_variant<512, 2>(p, B, itBegin, itEnd, s);
    else launch_icp_variant<512, 4>(p, B, itBegin, itEnd, s);
    if (timed) (void)hipEventRecord(g_prof.stop[g_prof.used++], s);
}

hipError_t launch_icp(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                      const uint8_t *swap, const float *prePose, int B, int N, double thres,
                      int maxIter, double relThr, int stopMode, IcpState *state, IcpCtrl *ctrl,
                      int32_t *nnj, hipStream_t s)
{
    IcpParams p{};
    p.X = X; p.Y = Y; p.lenX = lenX; p.lenY = lenY; p.swap = swap; p.prePose = prePose; p.N = N;
    p.thr2 = (float)(thres * thres);
    p.relThr = (float)relThr;
    p.stopMode = stopMode; p.maxIter = maxIter; p.state = state; p.ctrl = ctrl; p.nnj = nnj;
    hipError_t e = hipMemsetAsync(ctrl, 0, sizeof(IcpCtrl), s);
    if (e != hipSuccess) return e;
    if (stopMode == ICPFLOW_STOP_REFERENCE_) {
        for (int it = 0; it < maxIter; ++it) launch_icp_iters(p, B, it, it + 1, s);
    } else {
        launch_icp_iters(p, B, 0, maxIter, s);
    }
    return hipGetLastError();
}

hipError_t launch_icp_export(const IcpState *state, const IcpCtrl *ctrl, int B, int stopMode, float *R,
                             float *T, float *rmse, int32_t *iters, int32_t *converged, hipStream_t s)
{
    hipLaunchKernelGGL(icp_export_kernel, dim3((B + 127) / 128), dim3(128), 0, s, state, ctrl, B,
                       stopMode, R, T, rmse, iters, converged);
    return hipGetLastError();
}

}  // namespace icpflow
