// valu_ops.hip -- issue cost of individual gfx950 VALU instructions (wave64), measured as
// cycles per wave-instruction per SIMD with N independent register chains per lane.
// Build: hipcc --offload-arch=gfx950 -O3 valu_ops.hip -o valu_ops
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITERS = 2048;
constexpr int CH = 16;   // independent chains

#define KERNEL1(NAME, ASM)                                                              \
    __global__ void NAME(float *out, float a, float b)                                  \
    {                                                                                   \
        float r[CH];                                                                    \
        for (int k = 0; k < CH; ++k) r[k] = threadIdx.x * 0.001f + k;                   \
        for (int i = 0; i < ITERS; ++i) {                                               \
            _Pragma("unroll") for (int k = 0; k < CH; ++k)                              \
                asm volatile(ASM : "+v"(r[k]) : "v"(a), "v"(b));                        \
        }                                                                               \
        float s = 0; for (int k = 0; k < CH; ++k) s += r[k];                            \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                 \
    }

#define KERNEL2(NAME, ASM)                                                              \
    __global__ void NAME(float *out, float a, float b)                                  \
    {                                                                                   \
        v2f r[CH]; v2f av = {a, a + 1.f}, bv = {b, b + 1.f};                            \
        for (int k = 0; k < CH; ++k) r[k] = (v2f){threadIdx.x * 0.001f + k, 1.f * k};   \
        for (int i = 0; i < ITERS; ++i) {                                               \
            _Pragma("unroll") for (int k = 0; k < CH; ++k)                              \
                asm volatile(ASM : "+v"(r[k]) : "v"(av), "v"(bv));                      \
        }                                                                               \
        float s = 0; for (int k = 0; k < CH; ++k) s += r[k].x + r[k].y;                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                 \
    }

KERNEL1(k_add, "v_add_f32 %0, %0, %1")
KERNEL1(k_sub, "v_sub_f32 %0, %1, %0")
KERNEL1(k_mul, "v_mul_f32 %0, %0, %1")
KERNEL1(k_fma, "v_fma_f32 %0, %0, %1, %2")
KERNEL1(k_fmac, "v_fmac_f32 %0, %1, %2")
KERNEL1(k_min, "v_min_f32 %0, %0, %1")
KERNEL1(k_min3, "v_min3_f32 %0, %0, %1, %2")
KERNEL1(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL1(k_cndmask64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
KERNEL1(k_cmp, "v_cmp_lt_f32 vcc, %0, %1")
KERNEL1(k_cmp_sgpr, "v_cmp_lt_f32_e64 s[22:23], %0, %1")
KERNEL1(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL1(k_max, "v_max_f32 %0, %0, %1")
KERNEL1(k_med3, "v_med3_f32 %0, %0, %1, %2")
KERNEL1(k_sub_dpp, "v_sub_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf")
KERNEL1(k_mov_dpp, "v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf")
KERNEL1(k_readlane, "v_readlane_b32 s20, %1, 3\n\tv_add_f32 %0, s20, %0")
KERNEL2(k_pk_add, "v_pk_add_f32 %0, %0, %1")
KERNEL2(k_pk_add_neg, "v_pk_add_f32 %0, %1, %0 neg_lo:[0,1] neg_hi:[0,1]")
KERNEL2(k_pk_mul, "v_pk_mul_f32 %0, %0, %1")
KERNEL2(k_pk_fma, "v_pk_fma_f32 %0, %0, %1, %2")

template <typename F>
double time_ms(F launch)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    float *out; (void)hipMalloc(&out, 256 * 32 * 64 * sizeof(float));
    typedef void (*K)(float *, float, float);
    struct { const char *n; K k; } ks[] = {
        {"v_add_f32", k_add}, {"v_sub_f32", k_sub}, {"v_mul_f32", k_mul}, {"v_fma_f32", k_fma}, {"v_fmac_f32", k_fmac},
        {"v_sub_f32_dpp newbcast", k_sub_dpp}, {"v_mov_b32_dpp newbcast", k_mov_dpp}, {"v_readlane+v_add", k_readlane}, {"v_min_f32", k_min}, {"v_min3_f32", k_min3}, {"v_cndmask_b32 vcc", k_cndmask}, {"v_cndmask_b32 sgpr", k_cndmask64}, {"v_cmp_lt_f32 vcc", k_cmp}, {"v_cmp_lt_f32 sgpr", k_cmp_sgpr}, {"v_cmp+v_cndmask", k_cmp_cnd}, {"v_max_f32", k_max}, {"v_med3_f32", k_med3},
        {"v_pk_add_f32", k_pk_add}, {"v_pk_add_f32(neg)", k_pk_add_neg}, {"v_pk_mul_f32", k_pk_mul}, {"v_pk_fma_f32", k_pk_fma}};
    // effective clock: calibrate with v_fma at 32 waves/CU assuming 2 cycles/instr
    for (int wpc : {16}) {
        printf("waves/CU %d\n", wpc);
        for (auto &e : ks) {
            const int blocks = 256 * wpc / 4;
            double ms = time_ms([&] { hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f); });
            const double winst = (double)blocks * 4 * ITERS * CH;           // wave-instructions
            const double per_simd = winst / 1024.0;                          // per SIMD
            printf("  %-20s %8.3f ms   %6.2f ns per wave-instr per SIMD  (= %5.2f cycles @2.4GHz)\n", e.n, ms,
                   ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
        }
    }
    return 0;
}
