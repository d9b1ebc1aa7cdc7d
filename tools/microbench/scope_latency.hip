// Microbenchmark: round-trip latency of the memory operations a team exchange is made of, by scope (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o scope_latency scope_latency.hip && ./scope_latency
// One wave, dependent operations on one cache line: agent-scope atomic load (sc1: past the XCD's L2), workgroup-level
// L1-bypassing load (sc0: served by the XCD's L2), plain load (L1), agent-scope atomic add, L2-local atomic add,
// store + s_waitcnt (agent-scope / plain).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *buf, long long *out, int reps)
{
    unsigned *p = buf + 64 * blockIdx.x;
    unsigned v = 0;
    long long t0 = clock64();
    for (int i = 0; i < reps; ++i) v += __hip_atomic_load(p + (v & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t1 = clock64();
    for (int i = 0; i < reps; ++i) {
        unsigned r;
        const unsigned *q = p + (v & 1);
        asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(q) : "memory");
        v += r;
    }
    long long t2 = clock64();
    for (int i = 0; i < reps; ++i) v += *(volatile unsigned *)(p + (v & 1));
    long long t3 = clock64();
    for (int i = 0; i < reps; ++i) v += __hip_atomic_fetch_add(p + 2 + (v & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long long t4 = clock64();
    for (int i = 0; i < reps; ++i) v += __hip_atomic_fetch_add(p + 2 + (v & 1), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    long long t5 = clock64();
    for (int i = 0; i < reps; ++i) {
        __hip_atomic_store(p + 4, v + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t6 = clock64();
    for (int i = 0; i < reps; ++i) {
        *(volatile unsigned *)(p + 5) = v + i;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t7 = clock64();
    if (threadIdx.x == 0) {
        long long *o = out + 8 * blockIdx.x;
        o[0] = t1 - t0; o[1] = t2 - t1; o[2] = t3 - t2; o[3] = t4 - t3; o[4] = t5 - t4; o[5] = t6 - t5; o[6] = t7 - t6; o[7] = v;
    }
}
int main()
{
    unsigned *buf; long long *out;
    hipMalloc(&buf, 64 * 4 * 8); hipMemset(buf, 0, 64 * 4 * 8);
    hipMalloc(&out, 8 * 8 * 8);
    const int reps = 2000;
    for (int blocks : {1, 8}) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, buf, out, reps);
        long long h[64];
        hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
        const char *names[] = {"agent-scope atomic load (sc1)", "sc0 load (L2 of the XCD)", "plain load (L1)", "agent-scope atomic add (returning)",
                               "workgroup-scope atomic add (returning, L2)", "agent-scope store + vmcnt(0)", "plain store + vmcnt(0)"};
        printf("%d workgroup(s):\n", blocks);
        for (int j = 0; j < 7; ++j) printf("  %-44s %7.0f shader clocks per operation\n", names[j], (double)h[j] / reps);
    }
    return 0;
}
