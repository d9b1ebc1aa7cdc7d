// icp_fp32.hip -- ICPFLOW_ARITH_FP32_REFERENCE: the ICP loop with the Kabsch step evaluated in the
// reference's own operation order and precision (a study mode, not the fast path).
//
// The default kernel (icp.hip) accumulates 18 raw moments in fp64 and is therefore *more* accurate
// than the reference, whose tensors are fp32 throughout.  That matters for one observable: the
// batch-global stop (utils_icp_pytorch3d.py:209) fires when EVERY pair has a relative rmse change
// <= 1e-6, and in fp32 the weighted means of ~1000 coordinates of magnitude ~40 m carry ~1e-5 m of
// rounding noise, which moves T by as much, flips a near-gate correspondence now and then and keeps
// the rmse of "converged" pairs jittering above the threshold.  In exact arithmetic converged pairs
// repeat bit for bit (rel = 0) and the batch stops as soon as the last pair settles; the fp32
// reference typically runs into its iteration cap instead.  This kernel follows the reference line
// by line so that the noise level -- not the bits: a GPU tree reduction is no other backend's
// summation order -- is the reference's:
//   mu_x, mu_y   = sum(w x) / clamp(sum w, 1e-9)                     :314-315 (wmean)
//   Xc, Yc       = (x w - mu) w                                      :318-325
//   H            = Xc^T Yc / clamp(sum w, 1e-9)                      :326-336
//   R            = U diag(1, 1, det(U V^T)) V^T of H                 :339-362 (closed form, fp64 on the fp32 H)
//   T            = mu_y - mu_x R                                     :376
//   Xt           = X0 R + T                                          :177, :395
//   rmse         = sqrt(sum w |Xt - y|^2 / clamp(sum w, 1e-9))       :191-192
// everything fp32 except the 3x3 solve.  Search: the all-pairs LDS scan (same gate decisions and
// neighbours as every other search).  One workgroup per pair runs all iterations and records
// (R, T, rmse) per iteration; the epilogue of the speculative mode (icp_resolve_history_kernel)
// then applies the batch-global rule.  Trajectories of different pairs are independent, so this is
// exactly the reference's control flow.
#include "scan.hpp"
#include "kernels.hpp"
#include "kabsch.hpp"

namespace icpflow {

namespace {

constexpr int kBlock = 1024;
constexpr int kWaves = kBlock / kWave;

struct Fp32Params {
    const float *X, *Y;
    const int32_t *lenX, *lenY;
    const uint8_t *swap;
    const float *prePose;
    int N, B, maxIter;
    float thr2, relThr;
    IcpCtrl *ctrl;
    float *history;
    float4 *nn;   // [B,N]: masked neighbour (y w) and w of the current iteration
};

__device__ __forceinline__ float wave_sum_f32(float v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// K per-thread partial sums -> tot[0..K) (fp32; waves added in order).  Contains two barriers.
template <int K>
__device__ __forceinline__ void block_sum_f32(float (&v)[K], float *red, float *tot)
{
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float s = wave_sum_f32(v[k]);
        if (lane == 0) red[wave * K + k] = s;
    }
    __syncthreads();
    if (threadIdx.x < K) {
        float s = red[threadIdx.x];
        for (int w = 1; w < kWaves; ++w) s += red[w * K + threadIdx.x];
        tot[threadIdx.x] = s;
    }
    __syncthreads();
}

__global__ __launch_bounds__(kBlock) void icp_fp32ref_kernel(Fp32Params p)
{
    __shared__ ScanTile tile;
    __shared__ float red[kWaves * 9];
    __shared__ float tot[12];
    __shared__ float st[16];     // R (9), T (3), prev rmse
    __shared__ double Nsh[16];
    __shared__ double Hd[9];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid >> 6;
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
    float4 *nn = p.nn + (size_t)b * p.N;
    if (tid < 12) st[tid] = (tid < 9 && tid % 4 == 0) ? 1.f : 0.f;   // :140
    if (tid == 12) st[12] = 0.f;
    __syncthreads();
    const int ngroups = (xc.n + kBlock - 1) / kBlock;

    for (int it = 0; it < p.maxIter; ++it) {
        float R[9], T[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = st[k];
        T[0] = st[9]; T[1] = st[10]; T[2] = st[11];
        // ---- NN, gate (:154-161), masked clouds (:163-164), sums of the weighted means (:314-315)
        float s7[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < ngroups; ++g) {
            const int i = g * kBlock + tid;
            float qx[1] = {0.f}, qy[1] = {0.f}, qz[1] = {0.f};
            float x0 = 0.f, y0 = 0.f, z0 = 0.f;
            if (i < xc.n) {
                float rx, ry, rz;
                cloud_load(xc, i, rx, ry, rz);
                xf_apply(pre, rx, ry, rz, x0, y0, z0);   // utils_icp.py:21
                qx[0] = fmaf(z0, R[6], fmaf(y0, R[3], x0 * R[0])) + T[0];   // :177, :395
                qy[0] = fmaf(z0, R[7], fmaf(y0, R[4], x0 * R[1])) + T[1];
                qz[0] = fmaf(z0, R[8], fmaf(y0, R[5], x0 * R[2])) + T[2];
            }
            ScanAcc<1> acc;
            scan_cloud<1>(yc, none, &tile, qx, qy, qz, acc);
            if (i < xc.n) {
                const bool inl = acc.best[0] <= p.thr2;
                float nx = 0.f, ny = 0.f, nz = 0.f;
                if (inl) scan_resolve(yc, none, qx[0], qy[0], qz[0], acc.best[0], acc.chunk[0], nx, ny, nz);
                const float w = inl ? 1.f : 0.f;
                nn[i] = make_float4(nx, ny, nz, w);
                s7[0] += w;
                s7[1] += x0 * w; s7[2] += y0 * w; s7[3] += z0 * w;
                s7[4] += nx; s7[5] += ny; s7[6] += nz;
            }
        }
        block_sum_f32<7>(s7, red, tot);
        const float W = fmaxf(tot[0], 1e-9f);   // clamp(eps), :314-315, :326
        const float mux[3] = {tot[1] / W, tot[2] / W, tot[3] / W};
        const float muy[3] = {tot[4] / W, tot[5] / W, tot[6] / W};
        // ---- centred, re-masked clouds and their 3x3 product (:318-336)
        float s9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int g = 0; g < ngroups; ++g) {
            const int i = g * kBlock + tid;
            if (i < xc.n) {
                float rx, ry, rz, x0, y0, z0;
                cloud_load(xc, i, rx, ry, rz);
                xf_apply(pre, rx, ry, rz, x0, y0, z0);
                const float4 y = nn[i];
                const float w = y.w;
                const float xcv[3] = {(x0 * w - mux[0]) * w, (y0 * w - mux[1]) * w, (z0 * w - mux[2]) * w};
                const float ycv[3] = {(y.x - muy[0]) * w, (y.y - muy[1]) * w, (y.z - muy[2]) * w};
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) s9[a * 3 + c] = fmaf(xcv[a], ycv[c], s9[a * 3 + c]);
            }
        }
        block_sum_f32<9>(s9, red, tot);
        // ---- wave 0: rotation of the fp32 H (closed form in fp64), T = mu_y - mu_x R in fp32 (:376)
        if (wave == 0) {
            double frob2 = 0.0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const float h = tot[k] / W;
                Hd[k] = (double)h;
                frob2 += (double)h * (double)h;
            }
            double Rd[9];
            const double bound = 2.0 * 1.7320508075688774 * sqrt(frob2) * (1.0 + 1e-9);   // 2 (s1 + s2 + s3) at most
            if (!horn_rotation(Hd, bound, Nsh, lane, Rd)) rank1_rotation(Hd, Rd);
            if (lane == 0) {
                float Rn[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) Rn[k] = (float)Rd[k];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float m = fmaf(mux[2], Rn[6 + j], fmaf(mux[1], Rn[3 + j], mux[0] * Rn[j]));
                    st[9 + j] = muy[j] - m;
                }
#pragma unroll
                for (int k = 0; k < 9; ++k) st[k] = Rn[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 9; ++k) R[k] = st[k];
        T[0] = st[9]; T[1] = st[10]; T[2] = st[11];
        // ---- rmse of the moved cloud against this iteration's neighbours (:191-192)
        float s1[1] = {0.f};
        for (int g = 0; g < ngroups; ++g) {
            const int i = g * kBlock + tid;
            if (i < xc.n) {
                float rx, ry, rz, x0, y0, z0;
                cloud_load(xc, i, rx, ry, rz);
                xf_apply(pre, rx, ry, rz, x0, y0, z0);
                const float4 y = nn[i];
                const float dx = (fmaf(z0, R[6], fmaf(y0, R[3], x0 * R[0])) + T[0]) - y.x;
                const float dy = (fmaf(z0, R[7], fmaf(y0, R[4], x0 * R[1])) + T[1]) - y.y;
                const float dz = (fmaf(z0, R[8], fmaf(y0, R[5], x0 * R[2])) + T[2]) - y.z;
                s1[0] += ((dx * dx + dy * dy) + dz * dz) * y.w;
            }
        }
        block_sum_f32<1>(s1, red, tot);
        if (tid == 0) {
            const float rmse = sqrtf(tot[0] / W);
            const float prev = st[12];
            const float rel = (it == 0) ? 1.0f : (prev - rmse) / prev;   // :195-198
            const bool conv = rel <= p.relThr;                            // :209, NaN -> false
            st[12] = rmse;
            float *h = p.history + ((size_t)it * p.B + b) * kHistStride;
#pragma unroll
            for (int k = 0; k < 12; ++k) h[k] = st[k];
            h[12] = rmse;
            h[13] = 1.f;
            h[14] = W < 0.5f ? 0.f : W;   // gated correspondences (the clamp only acts on an empty gate)
            __hip_atomic_fetch_add(&p.ctrl->tally[it], 1ull | (conv ? 0ull : (1ull << 32)), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
}

}  // namespace

hipError_t launch_icp_fp32ref(const float *X, const float *Y, const int32_t *lenX, const int32_t *lenY,
                              const uint8_t *swap, const float *prePose, int B, int N, double thres, int maxIter,
                              double relThr, IcpState *state, IcpCtrl *ctrl, float *history, float *nnScratch,
                              hipStream_t s)
{
    Fp32Params p{};
    p.X = X; p.Y = Y; p.lenX = lenX; p.lenY = lenY; p.swap = swap; p.prePose = prePose;
    p.N = N; p.B = B; p.maxIter = maxIter;
    p.thr2 = (float)(thres * thres);
    p.relThr = (float)relThr;
    p.ctrl = ctrl; p.history = history; p.nn = reinterpret_cast<float4 *>(nnScratch);
    hipError_t e = hipMemsetAsync(ctrl, 0, sizeof(IcpCtrl), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(icp_fp32ref_kernel, dim3(B), dim3(kBlock), 0, s, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_icp_resolve_history(state, ctrl, history, B, maxIter, s);
}

}  // namespace icpflow
