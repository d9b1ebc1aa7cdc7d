// scan_rate.hip -- clocks per target and wave of the ICP's window scan (scan.hpp: scan_range_tie<1, true> over an LDS image) when
// W waves of a 768-thread workgroup scan at once, one workgroup per CU: the production loop against variants (two queries per
// lane; two chunks per step with independent running minima; the next chunk's loads issued before the current chunk's arithmetic).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I icp_flow_amd/csrc tools/microbench/scan_rate.hip -o tools/microbench/scan_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "scan.hpp"
using namespace icpflow;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int NT = 8192;   // targets of the image (96 KiB of LDS)

// variant 2: two chunks per step, independent running minima (more independent work between dependent LDS reads)
template <bool SECOND>
__device__ __forceinline__ void scan_two_chunks(const float4 *sx, const float4 *sy, const float4 *sz, int cBegin, int cEnd, float qx, float qy, float qz,
                                                float &best, int &chunk, bool &tie, float &second)
{
    int c = cBegin;
    for (; c + 2 * kChunk <= cEnd; c += 2 * kChunk) {
        float m0 = kInf, m1 = kInf;
#pragma unroll
        for (int u = 0; u < kChunk / 4; ++u) {
            const float4 ax = sx[(c >> 2) + u], ay = sy[(c >> 2) + u], az = sz[(c >> 2) + u];
            const float4 bx = sx[((c + kChunk) >> 2) + u], by = sy[((c + kChunk) >> 2) + u], bz = sz[((c + kChunk) >> 2) + u];
            const v2f a0 = sqdist2(qx, qy, qz, v2f{ax.x, ax.y}, v2f{ay.x, ay.y}, v2f{az.x, az.y});
            const v2f a1 = sqdist2(qx, qy, qz, v2f{ax.z, ax.w}, v2f{ay.z, ay.w}, v2f{az.z, az.w});
            const v2f b0 = sqdist2(qx, qy, qz, v2f{bx.x, bx.y}, v2f{by.x, by.y}, v2f{bz.x, bz.y});
            const v2f b1 = sqdist2(qx, qy, qz, v2f{bx.z, bx.w}, v2f{by.z, by.w}, v2f{bz.z, bz.w});
            m0 = min3f(min3f(m0, a0.x, a0.y), a1.x, a1.y);
            m1 = min3f(min3f(m1, b0.x, b0.y), b1.x, b1.y);
        }
        if (SECOND) second = min_nonneg(second, max_nonneg(m0, best));
        if (m0 < best) { best = m0; chunk = c; tie = false; } else if (m0 == best) tie = true;
        if (SECOND) second = min_nonneg(second, max_nonneg(m1, best));
        if (m1 < best) { best = m1; chunk = c + kChunk; tie = false; } else if (m1 == best) tie = true;
    }
    for (; c < cEnd; c += kChunk) {
        float m0 = kInf;
#pragma unroll
        for (int u = 0; u < kChunk / 4; ++u) {
            const float4 ax = sx[(c >> 2) + u], ay = sy[(c >> 2) + u], az = sz[(c >> 2) + u];
            const v2f a0 = sqdist2(qx, qy, qz, v2f{ax.x, ax.y}, v2f{ay.x, ay.y}, v2f{az.x, az.y});
            const v2f a1 = sqdist2(qx, qy, qz, v2f{ax.z, ax.w}, v2f{ay.z, ay.w}, v2f{az.z, az.w});
            m0 = min3f(min3f(m0, a0.x, a0.y), a1.x, a1.y);
        }
        if (SECOND) second = min_nonneg(second, max_nonneg(m0, best));
        if (m0 < best) { best = m0; chunk = c; tie = false; } else if (m0 == best) tie = true;
    }
}

// MODE 0: production (Q = 1, SECOND); 1: production with Q = 2; 2: two chunks per step; 3: production without SECOND
template <int MODE>
__global__ __launch_bounds__(768) void k_scan(float *out, long long *clk, int window, int reps, int waves)
{
    extern __shared__ __attribute__((aligned(16))) float img[];
    float *lx = img, *ly = lx + NT, *lz = ly + NT;
    for (int k = threadIdx.x; k < NT; k += 768) { lx[k] = k * 0.013f; ly[k] = (k % 97) * 0.11f; lz[k] = (k % 13) * 0.07f; }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float acc = 0.f;
    long long t0 = 0, t1 = 0;
    if (wave < waves) {
        const int cb = ((wave * 611) % (NT - window)) / kChunk * kChunk, ce = cb + window;
        t0 = clock64();
        for (int r = 0; r < reps; ++r) {
            if (MODE == 1) {
                float qx[2] = {lx[cb] + lane * 0.01f + r, lx[cb] + lane * 0.02f + r}, qy[2] = {3.f, 4.f}, qz[2] = {0.3f, 0.2f};
                ScanAcc<2> a; bool tie[2] = {false, false}; float sec[2] = {kInf, kInf};
                scan_init(a);
                scan_range_tie<2, true>((const float4 *)lx, (const float4 *)ly, (const float4 *)lz, cb, ce, qx, qy, qz, a, tie, sec);
                acc += a.best[0] + a.best[1] + sec[0] + sec[1] + a.chunk[0] + a.chunk[1] + (tie[0] ? 1.f : 0.f) + (tie[1] ? 1.f : 0.f);
            } else if (MODE == 2) {
                float best = kInf, sec = kInf; int chunk = 0; bool tie = false;
                scan_two_chunks<true>((const float4 *)lx, (const float4 *)ly, (const float4 *)lz, cb, ce, lx[cb] + lane * 0.01f + r, 3.f, 0.3f, best, chunk, tie, sec);
                acc += best + sec + chunk + (tie ? 1.f : 0.f);
            } else {
                float qx[1] = {lx[cb] + lane * 0.01f + r}, qy[1] = {3.f}, qz[1] = {0.3f};
                ScanAcc<1> a; bool tie[1] = {false}; float sec[1] = {kInf};
                scan_init(a);
                if (MODE == 0) scan_range_tie<1, true>((const float4 *)lx, (const float4 *)ly, (const float4 *)lz, cb, ce, qx, qy, qz, a, tie, sec);
                else scan_range_tie<1, false>((const float4 *)lx, (const float4 *)ly, (const float4 *)lz, cb, ce, qx, qy, qz, a, tie);
                acc += a.best[0] + sec[0] + a.chunk[0] + (tie[0] ? 1.f : 0.f);
            }
        }
        t1 = clock64();
    }
    out[blockIdx.x * 768 + threadIdx.x] = acc;
    if (lane == 0 && blockIdx.x == 0) clk[wave] = t1 - t0;
}

template <int MODE>
int run(const char *name, int window, int waves)
{
    float *out; long long *clk;
    CHECK(hipMalloc(&out, 256 * 768 * 4)); CHECK(hipMalloc(&clk, 12 * 8));
    CHECK(hipFuncSetAttribute((const void *)k_scan<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, NT * 12));
    const int reps = 64;
    for (int it = 0; it < 2; ++it) hipLaunchKernelGGL(k_scan<MODE>, dim3(256), dim3(768), NT * 12, 0, out, clk, window, reps, waves);
    CHECK(hipDeviceSynchronize());
    long long h[12];
    CHECK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
    double mx = 0;
    for (int w = 0; w < waves; ++w) mx = h[w] > mx ? (double)h[w] : mx;
    const double evalsPerLane = (MODE == 1) ? 2.0 : 1.0;
    printf("%-34s window %5d, %2d waves scanning: %7.1f clocks per target and wave (%5.1f per query-target)\n", name, window, waves, mx / reps / window, mx / reps / window / evalsPerLane);
    CHECK(hipFree(out)); CHECK(hipFree(clk));
    return 0;
}

int main()
{
    for (int waves : {1, 3, 6, 12})
        for (int window : {256, 1024}) {
            if (run<0>("production (Q=1, runner-up)", window, waves)) return 1;
            if (run<3>("production without the runner-up", window, waves)) return 1;
            if (run<1>("two queries per lane (Q=2)", window, waves)) return 1;
            if (run<2>("two chunks per step", window, waves)) return 1;
        }
    return 0;
}
