// hist.hip -- translation-histogram vote (a-1) and NMS + top-k peaks (a-2) for gfx950.
//
// Reference semantics: hist_cuda/cpp/hist_cuda_core.cuh:40-60 (vote) and
// utils_hist.py:21-29 (topk_nms).  Design (not a port): the reference launches one
// thread per (b,i,j) of the PADDED N x N grid and float-atomicAdds into global
// memory; here a workgroup owns a slice of X rows of one pair, keeps the Y tile in
// LDS (broadcast reads), votes into an LDS-private uint32 histogram when it fits
// and flushes the non-zero bins once.  Counters are uint32 (exact beyond 2^24).
#include "common.hpp"
#include "kernels.hpp"

namespace icpflow {

// ---------------------------------------------------------------------------------
// count_valid: len[b] = #(flag > 0)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_valid_kernel(const float4 *__restrict__ pts, int N,
                                                          int32_t *__restrict__ len)
{
    __shared__ int scratch[4];
    const int b = blockIdx.x;
    const float4 *p = pts + (size_t)b * N;
    int c[1] = {0};
    for (int i = threadIdx.x; i < N; i += blockDim.x) c[0] += (p[i].w > 0.0f) ? 1 : 0;
    block_sum<1, int>(c, scratch);
    if (threadIdx.x == 0) len[b] = c[0];
}

void launch_count_valid(const float *pts, int B, int N, int32_t *len, hipStream_t s)
{
    hipLaunchKernelGGL(count_valid_kernel, dim3(B), dim3(256), 0, s, (const float4 *)pts, N, len);
}

// ---------------------------------------------------------------------------------
// vote
// ---------------------------------------------------------------------------------
struct VoteBox {
    float min_x, min_y, min_z, max_x, max_y, max_z;
    int len_x, len_y, len_z;
};

constexpr int kVoteBlock = 256;  // threads; one X row per thread per slice
constexpr int kVoteTile = 1024;  // Y points staged in LDS per step (16 KiB)

// bins_u32: [B, L] zero-initialised.  swap (optional, per pair): vote with X and Y
// exchanged -- used by the fused registration path where "src" is the smaller cloud.
// edges (optional, device): box taken from edges (min = e[0], max = e[L-1]).
template <bool LDS_HIST>
__global__ __launch_bounds__(kVoteBlock) void hist_vote_kernel(
    const float4 *__restrict__ X, const float4 *__restrict__ Y, int NX, int NY, VoteBox box,
    const float *__restrict__ ex, const float *__restrict__ ey, const float *__restrict__ ez,
    const uint8_t *__restrict__ swap, uint32_t *__restrict__ bins_u32)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *tile = reinterpret_cast<float4 *>(smem);
    uint32_t *lhist = reinterpret_cast<uint32_t *>(smem + sizeof(float4) * kVoteTile);

    const int b = blockIdx.y;
    const bool sw = swap != nullptr && swap[b] != 0;
    const float4 *xb = (sw ? Y : X) + (size_t)b * (sw ? NY : NX);
    const float4 *yb = (sw ? X : Y) + (size_t)b * (sw ? NX : NY);
    const int nx = sw ? NY : NX, ny = sw ? NX : NY;
    if (ex != nullptr) {
        box.min_x = ex[0]; box.max_x = ex[box.len_x - 1];
        box.min_y = ey[0]; box.max_y = ey[box.len_y - 1];
        box.min_z = ez[0]; box.max_z = ez[box.len_z - 1];
    }
    const int L = box.len_x * box.len_y * box.len_z;
    uint32_t *gb = bins_u32 + (size_t)b * L;

    const int i = blockIdx.x * kVoteBlock + threadIdx.x;
    float4 xi = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < nx) xi = xb[i];
    const bool xvalid = xi.w > 0.0f;
    // a slice without any valid X row has nothing to do (pads: utils_helper.py:191-192)
    if (!__syncthreads_or(xvalid ? 1 : 0)) return;

    if (LDS_HIST) {
        for (int k = threadIdx.x; k < L; k += kVoteBlock) lhist[k] = 0u;
    }
    // hist_cuda_core.cuh:52-54: (v-min)/(max-min) * float(len); the denominators
    // and float(len) are loop invariants
    const float rx = box.max_x - box.min_x, ry = box.max_y - box.min_y, rz = box.max_z - box.min_z;
    const float flx = (float)box.len_x, fly = (float)box.len_y, flz = (float)box.len_z;

    for (int j0 = 0; j0 < ny; j0 += kVoteTile) {
        const int tn = min(kVoteTile, ny - j0);
        __syncthreads();  // previous tile fully consumed (and lhist zeroed)
        int any = 0;
        for (int k = threadIdx.x; k < tn; k += kVoteBlock) {
            const float4 t = yb[j0 + k];
            tile[k] = t;
            any |= (t.w > 0.0f) ? 1 : 0;
        }
        if (!__syncthreads_or(any)) continue;  // a tile of pads
        if (!xvalid) continue;
        for (int k = 0; k < tn; ++k) {
            const float4 t = tile[k];  // same address in every lane: LDS broadcast
            if (!(t.w > 0.0f)) continue;  // wave-uniform
            const float vx = xi.x - t.x, vy = xi.y - t.y, vz = xi.z - t.z;
            if (vx >= box.min_x && vx < box.max_x && vy >= box.min_y && vy < box.max_y &&
                vz >= box.min_z && vz < box.max_z) {
                const int px = (int)floorf(((vx - box.min_x) / rx) * flx);
                const int py = (int)floorf(((vy - box.min_y) / ry) * fly);
                const int pz = (int)floorf(((vz - box.min_z) / rz) * flz);
                const int bin = (px * box.len_y + py) * box.len_z + pz;
                if (LDS_HIST) atomicAdd(&lhist[bin], 1u);
                else atomicAdd(&gb[bin], 1u);
            }
        }
    }
    if (LDS_HIST) {
        __syncthreads();
        for (int k = threadIdx.x; k < L; k += kVoteBlock) {
            const uint32_t v = lhist[k];
            if (v) atomicAdd(&gb[k], v);
        }
    }
}

__global__ void u32_to_f32_kernel(const uint32_t *__restrict__ in, float *__restrict__ out, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

hipError_t launch_hist_vote(const float *X, const float *Y, int B, int NX, int NY,
                            const float mins[3], const float maxs[3], const int lens[3],
                            const float *ex, const float *ey, const float *ez,
                            const uint8_t *swap, uint32_t *bins_u32, hipStream_t s)
{
    VoteBox box{mins ? mins[0] : 0.f, mins ? mins[1] : 0.f, mins ? mins[2] : 0.f,
                maxs ? maxs[0] : 0.f, maxs ? maxs[1] : 0.f, maxs ? maxs[2] : 0.f,
                lens[0], lens[1], lens[2]};
    const size_t L = (size_t)lens[0] * lens[1] * lens[2];
    hipError_t e = hipMemsetAsync(bins_u32, 0, sizeof(uint32_t) * L * (size_t)B, s);
    if (e != hipSuccess) return e;
    // rows of the X role per pair: NX, or either cloud when roles may be swapped
    const int rows = swap ? (NX > NY ? NX : NY) : NX;
    dim3 grid((rows + kVoteBlock - 1) / kVoteBlock, B);
    const size_t tile_bytes = sizeof(float4) * kVoteTile;
    const size_t lds_hist = tile_bytes + sizeof(uint32_t) * L;
    if (lds_hist <= 64 * 1024) {
        hipLaunchKernelGGL(hist_vote_kernel<true>, grid, dim3(kVoteBlock), lds_hist, s,
                           (const float4 *)X, (const float4 *)Y, NX, NY, box, ex, ey, ez, swap,
                           bins_u32);
    } else {
        hipLaunchKernelGGL(hist_vote_kernel<false>, grid, dim3(kVoteBlock), tile_bytes, s,
                           (const float4 *)X, (const float4 *)Y, NX, NY, box, ex, ey, ez, swap,
                           bins_u32);
    }
    return hipGetLastError();
}

hipError_t launch_u32_to_f32(const uint32_t *in, float *out, size_t n, hipStream_t s)
{
    hipLaunchKernelGGL(u32_to_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, n);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------
// peaks: separable (z, y, x) running maximum through two scratch volumes, then k
// rounds of block-wide arg-max on the key (vote << 32 | ~index).
// One workgroup per pair; the volumes live in global scratch (L2 resident: a demo
// histogram is 20 KiB, the largest Waymo one 868 KiB) so every size takes the same path.
// ---------------------------------------------------------------------------------
constexpr int kPeakBlock = 256;
constexpr int kPeakMaxK = 8;

template <typename BinT>
__global__ __launch_bounds__(kPeakBlock) void hist_peaks_kernel(
    const BinT *__restrict__ bins, int Lx, int Ly, int Lz, int k, int radius,
    uint32_t *__restrict__ wsA, uint32_t *__restrict__ wsB, float *__restrict__ votes,
    int64_t *__restrict__ idx_out)
{
    __shared__ unsigned long long red[kPeakBlock / kWave];
    __shared__ unsigned long long chosen[kPeakMaxK];
    const int b = blockIdx.x;
    const int L = Lx * Ly * Lz;
    const BinT *h = bins + (size_t)b * L;
    uint32_t *A = wsA + (size_t)b * L;
    uint32_t *Bv = wsB + (size_t)b * L;
    const int tid = threadIdx.x;

    // pass z: A = max over |dz| <= r of h   (-inf padding == ignore out of range)
    for (int f = tid; f < L; f += kPeakBlock) {
        const int z = f % Lz, base = f - z;
        const int lo = max(0, z - radius), hi = min(Lz - 1, z + radius);
        uint32_t m = 0;
        for (int q = lo; q <= hi; ++q) m = max(m, (uint32_t)h[base + q]);
        A[f] = m;
    }
    __syncthreads();
    // pass y: B = max over |dy| <= r of A
    for (int f = tid; f < L; f += kPeakBlock) {
        const int z = f % Lz, y = (f / Lz) % Ly, x = f / (Lz * Ly);
        const int lo = max(0, y - radius), hi = min(Ly - 1, y + radius);
        uint32_t m = 0;
        for (int q = lo; q <= hi; ++q) m = max(m, A[(x * Ly + q) * Lz + z]);
        Bv[f] = m;
    }
    __syncthreads();
    // pass x: A = max over |dx| <= r of B  -> full 3-D window maximum
    for (int f = tid; f < L; f += kPeakBlock) {
        const int z = f % Lz, y = (f / Lz) % Ly, x = f / (Lz * Ly);
        const int lo = max(0, x - radius), hi = min(Lx - 1, x + radius);
        uint32_t m = 0;
        for (int q = lo; q <= hi; ++q) m = max(m, Bv[(q * Ly + y) * Lz + z]);
        A[f] = m;
    }
    __syncthreads();
    // surviving vote = h where h == window max, else 0 (utils_hist.py:25-26);
    // k selection rounds, order (vote desc, flat index asc)
    for (int r = 0; r < k; ++r) {
        unsigned long long best = 0ull;
        bool have = false;
        for (int f = tid; f < L; f += kPeakBlock) {
            const uint32_t v = (uint32_t)h[f];
            const uint32_t s = (v == A[f]) ? v : 0u;
            const unsigned long long key =
                ((unsigned long long)s << 32) | (unsigned long long)(0xFFFFFFFFu - (uint32_t)f);
            bool taken = false;
            for (int q = 0; q < r; ++q) taken |= (chosen[q] == key);
            if (!taken && (!have || key > best)) { best = key; have = true; }
        }
        // keys are unique (index bits), 0 can only be "nothing found"
        unsigned long long w = wave_max_u64(have ? best : 0ull);
        if ((tid & (kWave - 1)) == 0) red[tid >> 6] = w;
        __syncthreads();
        if (tid == 0) {
            unsigned long long m = red[0];
            for (int q = 1; q < kPeakBlock / kWave; ++q) m = red[q] > m ? red[q] : m;
            chosen[r] = m;
            votes[(size_t)b * k + r] = (float)(uint32_t)(m >> 32);
            idx_out[(size_t)b * k + r] = (int64_t)(0xFFFFFFFFu - (uint32_t)(m & 0xFFFFFFFFull));
        }
        __syncthreads();
    }
}

template <typename BinT>
static hipError_t launch_peaks_t(const BinT *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s)
{
    hipLaunchKernelGGL(hist_peaks_kernel<BinT>, dim3(B), dim3(kPeakBlock), 0, s, bins, Lx, Ly, Lz, k,
                       (kernel_size - 1) / 2, wsA, wsB, votes, idx);
    return hipGetLastError();
}

hipError_t launch_hist_peaks_f32(const float *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s)
{
    return launch_peaks_t<float>(bins, B, Lx, Ly, Lz, k, kernel_size, wsA, wsB, votes, idx, s);
}

hipError_t launch_hist_peaks_u32(const uint32_t *bins, int B, int Lx, int Ly, int Lz, int k,
                                 int kernel_size, uint32_t *wsA, uint32_t *wsB, float *votes,
                                 int64_t *idx, hipStream_t s)
{
    return launch_peaks_t<uint32_t>(bins, B, Lx, Ly, Lz, k, kernel_size, wsA, wsB, votes, idx, s);
}

}  // namespace icpflow
