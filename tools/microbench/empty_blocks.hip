// empty_blocks.hip -- what a launch of mostly EMPTY 256-thread workgroups costs (the sweeps of a batch padded far beyond its
// clusters: nine blocks in ten read their pair's lengths and return).  Build: hipcc --offload-arch=gfx950 -O3 empty_blocks.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_empty(const int *len, double *out, int qblocks)
{
    const int job = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    if (qb * 256 >= len[job]) { if (threadIdx.x < 8) out[(size_t)blockIdx.x * 8 + threadIdx.x] = 0.0; return; }
    out[(size_t)blockIdx.x * 8] = 1.0;
}
int main()
{
    const int jobs = 128 * 11, qblocks = 40;
    int *len; double *out;
    CHECK(hipMalloc(&len, jobs * 4)); CHECK(hipMemset(len, 0, jobs * 4)); CHECK(hipMalloc(&out, (size_t)jobs * qblocks * 64));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int blocks : {1408, 5120, 14080, 56320}) {
        for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, len, out, qblocks);
        CHECK(hipEventRecord(a, 0));
        for (int it = 0; it < 20; ++it) hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, 0, len, out, qblocks);
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        printf("%6d empty workgroups of 256 threads: %.1f us per launch\n", blocks, ms / 20 * 1e3);
    }
    return 0;
}
