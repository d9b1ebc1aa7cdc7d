// valu_rates.hip -- issue rates of the instruction mixes the NN scan can be built from
// (plain f32 VALU vs packed v_pk_*_f32, v_min3, broadcast LDS reads) on gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int ITERS = 4096;

// A: scalar-f32 distance mix: 3 sub, 1 mul, 2 fma per eval, min3 per 2 evals; 8 evals/iter
__global__ void k_plain(float *out, float a)
{
    float qx = threadIdx.x * 0.001f, qy = qx + 1.f, qz = qx + 2.f;
    float m = 1e30f;
    float tx = a, ty = a * 2.f, tz = a * 3.f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float dx0 = qx - tx, dy0 = qy - ty, dz0 = qz - tz;
            float d0 = __builtin_fmaf(dz0, dz0, __builtin_fmaf(dy0, dy0, dx0 * dx0));
            float dx1 = qx - ty, dy1 = qy - tz, dz1 = qz - tx;
            float d1 = __builtin_fmaf(dz1, dz1, __builtin_fmaf(dy1, dy1, dx1 * dx1));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(d0), "v"(d1));
            tx += 0.25f; ty += 0.5f; tz -= 0.125f;   // 3 extra adds per 2 evals (keeps values live)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = m;
}

// B: packed mix: 2 queries per lane in v2f; per target: 3 pk_add, 1 pk_mul, 2 pk_fma -> 2 evals
__global__ void k_packed(float *out, float a)
{
    v2f qx = {threadIdx.x * 0.001f, threadIdx.x * 0.002f}, qy = qx + 1.f, qz = qx + 2.f;
    float m0 = 1e30f, m1 = 1e30f;
    float tx = a, ty = a * 2.f, tz = a * 3.f;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v2f dx0 = qx - tx, dy0 = qy - ty, dz0 = qz - tz;
            v2f d0 = __builtin_elementwise_fma(dz0, dz0, __builtin_elementwise_fma(dy0, dy0, dx0 * dx0));
            v2f dx1 = qx - ty, dy1 = qy - tz, dz1 = qz - tx;
            v2f d1 = __builtin_elementwise_fma(dz1, dz1, __builtin_elementwise_fma(dy1, dy1, dx1 * dx1));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m0) : "v"(d0.x), "v"(d1.x));
            asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m1) : "v"(d0.y), "v"(d1.y));
            tx += 0.25f; ty += 0.5f; tz -= 0.125f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = m0 + m1;
}

// C: pure v_fma_f32 chain x8 independent
__global__ void k_fma(float *out, float a)
{
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = threadIdx.x * 0.01f + k;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = __builtin_fmaf(r[k], a, 0.5f);
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += r[k];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// D: pure v_pk_fma_f32 x8 independent
__global__ void k_pkfma(float *out, float a)
{
    v2f r[8];
    for (int k = 0; k < 8; ++k) r[k] = (v2f){threadIdx.x * 0.01f + k, threadIdx.x * 0.02f + k};
    v2f av = {a, a}, c = {0.5f, 0.5f};
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = __builtin_elementwise_fma(r[k], av, c);
    }
    float s = 0; for (int k = 0; k < 8; ++k) s += r[k].x + r[k].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// E: LDS broadcast ds_read_b128 stream + plain mix with Q queries per lane
template <int Q>
__global__ void k_lds(float *out, float a)
{
    __shared__ float4 tx4[256], ty4[256], tz4[256];
    for (int k = threadIdx.x; k < 256; k += blockDim.x) {
        tx4[k] = make_float4(a + k, a - k, a * k, a);
        ty4[k] = make_float4(a + 2 * k, a - 2 * k, a * k, a);
        tz4[k] = make_float4(a + 3 * k, a - 3 * k, a * k, a);
    }
    __syncthreads();
    float qx[Q], qy[Q], qz[Q], m[Q];
    for (int q = 0; q < Q; ++q) { qx[q] = threadIdx.x * 0.001f + q; qy[q] = qx[q] + 1.f; qz[q] = qx[q] + 2.f; m[q] = 1e30f; }
    for (int i = 0; i < ITERS / 64; ++i) {
        for (int c = 0; c < 256; c += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float4 X = tx4[c + u], Y = ty4[c + u], Z = tz4[c + u];
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    float dx, dy, dz, d0, d1, d2, d3;
                    dx = qx[q] - X.x; dy = qy[q] - Y.x; dz = qz[q] - Z.x; d0 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    dx = qx[q] - X.y; dy = qy[q] - Y.y; dz = qz[q] - Z.y; d1 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    dx = qx[q] - X.z; dy = qy[q] - Y.z; dz = qz[q] - Z.z; d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    dx = qx[q] - X.w; dy = qy[q] - Y.w; dz = qz[q] - Z.w; d3 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
                    asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[q]) : "v"(d0), "v"(d1));
                    asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(m[q]) : "v"(d2), "v"(d3));
                }
            }
        }
    }
    float s = 0; for (int q = 0; q < Q; ++q) s += m[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(); hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 256 * 8 * 1024 * sizeof(float) * 4));
    hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, 0));
    printf("device %s, %d CUs, clock %d kHz\n", pr.name, pr.multiProcessorCount, pr.clockRate);
    for (int wpc : {4, 8, 16, 32}) {           // waves per CU
        int blocks = 256 * wpc / 4; int threads = 256; // 4 waves per block
        double lanes = (double)blocks * threads;
        double ms;
        ms = time_ms([&] { hipLaunchKernelGGL(k_plain, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); });
        printf("waves/CU %2d  plain   : %7.3f ms  %7.2f Gevals/s  (%.1f T lane-instr/s incl 1.5 extra)\n", wpc, ms,
               lanes * ITERS * 8 / ms * 1e-6, lanes * ITERS * (8 * 6.5 + 12) / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_packed, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); });
        printf("waves/CU %2d  packed  : %7.3f ms  %7.2f Gevals/s\n", wpc, ms, lanes * ITERS * 16 / ms * 1e-6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f); });
        printf("waves/CU %2d  fma     : %7.3f ms  %7.2f TFLOP/s\n", wpc, ms, lanes * ITERS * 32 * 2 / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(threads), 0, 0, out, 1.0001f); });
        printf("waves/CU %2d  pk_fma  : %7.3f ms  %7.2f TFLOP/s\n", wpc, ms, lanes * ITERS * 32 * 4 / ms * 1e-9);
        ms = time_ms([&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); });
        printf("waves/CU %2d  lds Q=1 : %7.3f ms  %7.2f Gevals/s\n", wpc, ms, lanes * (ITERS / 64) * 1024.0 * 1 / ms * 1e-6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); });
        printf("waves/CU %2d  lds Q=2 : %7.3f ms  %7.2f Gevals/s\n", wpc, ms, lanes * (ITERS / 64) * 1024.0 * 2 / ms * 1e-6);
        ms = time_ms([&] { hipLaunchKernelGGL(k_lds<4>, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); });
        printf("waves/CU %2d  lds Q=4 : %7.3f ms  %7.2f Gevals/s\n", wpc, ms, lanes * (ITERS / 64) * 1024.0 * 4 / ms * 1e-6);
    }
    return 0;
}
