/*
 * oracle_core.c -- CPU restatement of the two device primitives on the ICP-Flow
 * cluster-pair registration hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (icp_flow_amd/) never does.
 *
 * What is restated here and from where:
 *
 *   oracle_hist_vote   <- /root/reference/hist_cuda/cpp/hist_cuda_core.cuh:40-60
 *                         (the CUDA vote kernel; CPU-refused by hist.cpp:13-22,
 *                         so it cannot be compiled here: restated operation by
 *                         operation in the kernel's own fp32 order)
 *                         launcher zero-fill: hist_cuda.cu:59
 *
 *   oracle_knn1        <- pytorch3d 0.7.4 `knn_points(K=1)` (environment.yml:141;
 *                         NOT vendored under /root/reference).  Published
 *                         algorithm: brute force over p1[:len1] x p2[:len2];
 *                         squared L2 distance accumulated from direct coordinate
 *                         differences `dist += diff*diff` (which nvcc contracts
 *                         to an FMA chain on the reference's CUDA build); first
 *                         minimum wins ties; rows >= len1 keep dist 0 / idx 0.
 *                         Call sites that anchor the semantics:
 *                         utils_helper.py:27 (no lengths) and
 *                         utils_icp_pytorch3d.py:154-156 (lengths, return_nn).
 *                         "parity unpinned" at this third-party boundary: the
 *                         reference holds no test vectors for it.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <stddef.h>

#if defined(_OPENMP)
#include <omp.h>
#endif

/* Threads of THIS library's loops only (a num_threads clause on its own parallel regions).  The OpenMP runtime is
 * shared with torch-CPU in the same process: omp_set_num_threads() here would also change the thread count of
 * every torch op of the oracle (measured on a 256-core host: 256 threads for the oracle's many small torch ops
 * make it 80x slower). */
static int g_threads = 0;   /* 0 = the runtime's default */

int oracle_num_threads(void)
{
#if defined(_OPENMP)
    return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_num_threads(int n)
{
    if (n > 0) g_threads = n;
}

/* ------------------------------------------------------------------------- */
/* translation vote, hist_cuda_core.cuh:40-60                                */
/* X,Y: [B,N?,4] contiguous fp32 (x,y,z,flag); bins: [B,Lx,Ly,Lz] fp32       */
/* ------------------------------------------------------------------------- */
void oracle_hist_vote(const float *X, const float *Y, int B, int NX, int NY,
                      float min_x, float min_y, float min_z,
                      float max_x, float max_y, float max_z,
                      int len_x, int len_y, int len_z, float *bins)
{
    const size_t per = (size_t)len_x * len_y * len_z;
    memset(bins, 0, sizeof(float) * per * (size_t)B);        /* hist_cuda.cu:59 */
    /* __int2float_rd(len): exact for every len < 2^24 */
    const float flx = (float)len_x, fly = (float)len_y, flz = (float)len_z;
    const float rx = max_x - min_x, ry = max_y - min_y, rz = max_z - min_z;
    /* p_x <= len_x, p_y <= len_y, p_z <= len_z: a flat index exceeds the pair's bins by at most this much */
    const size_t nspill = (size_t)len_y * len_z + (size_t)len_z + 1;
    float *spill = (float *)calloc((size_t)B * nspill, sizeof(float));

#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_num_threads())
    for (int b = 0; b < B; ++b) {
        const float *xb = X + (size_t)b * NX * 4;
        const float *yb = Y + (size_t)b * NY * 4;
        float *hb = bins + (size_t)b * per;
        for (int i = 0; i < NX; ++i) {
            if (!(xb[i * 4 + 3] > 0.0f)) continue;            /* :42,44 */
            const float xi = xb[i * 4 + 0], yi = xb[i * 4 + 1], zi = xb[i * 4 + 2];
            for (int j = 0; j < NY; ++j) {
                if (!(yb[j * 4 + 3] > 0.0f)) continue;        /* :43,44 */
                const float vx = xi - yb[j * 4 + 0];          /* :46 */
                const float vy = yi - yb[j * 4 + 1];          /* :47 */
                const float vz = zi - yb[j * 4 + 2];          /* :48 */
                if (vx >= min_x && vx < max_x && vy >= min_y && vy < max_y &&
                    vz >= min_z && vz < max_z) {              /* :49 */
                    /* :52-54  floor( (v-min)/(max-min) * float(len) ), fp32, IEEE div */
                    const float qx = (vx - min_x) / rx;
                    const float qy = (vy - min_y) / ry;
                    const float qz = (vz - min_z) / rz;
                    const int px = (int)floorf(qx * flx);
                    const int py = (int)floorf(qy * fly);
                    const int pz = (int)floorf(qz * flz);
                    /* :57-58.  The quotient of a difference one float below max can round to 1.0, i.e. p = len:
                     * the reference does not clamp, its flat index then runs into the NEXT pair's bins (one
                     * [B, L] allocation, hist_cuda.cu:59) -- or, for the last pair, past the allocation
                     * (undefined there; dropped here).  Such votes are counted per pair in `spill` and added
                     * to the next pair's bins after the parallel loop. */
                    const size_t flat = ((size_t)px * len_y + py) * len_z + pz;
                    if (flat < per) hb[flat] += 1.0f;
                    else if (spill && flat - per < nspill) spill[(size_t)b * nspill + (flat - per)] += 1.0f;
                }
            }
        }
    }
    if (spill) {
        for (int b = 0; b + 1 < B; ++b)
            for (size_t e = 0; e < nspill && e < per; ++e)
                bins[(size_t)(b + 1) * per + e] += spill[(size_t)b * nspill + e];
        free(spill);
    }
}

/* ------------------------------------------------------------------------- */
/* K=1 brute-force nearest neighbour (pytorch3d knn_points semantics)        */
/* P1: [B,N1,s1] fp32, P2: [B,N2,s2] fp32 (first three columns are x,y,z)    */
/* len1/len2: per-batch valid prefixes or NULL (= N1 / N2)                   */
/* idx: int64 [B,N1], d2: fp32 [B,N1] (squared); nn: fp32 [B,N1,3] or NULL   */
/* ------------------------------------------------------------------------- */
__attribute__((target_clones("avx2,fma", "default")))
static void knn1_one(const float *p1, const float *p2, int n1, int n2, int s1,
                     int s2, int l1, int l2, int64_t *idx, float *d2, float *nn)
{
    for (int i = 0; i < n1; ++i) {
        int64_t bj = 0;
        float bd = 0.0f;
        if (i < l1 && l2 > 0) {
            const float qx = p1[(size_t)i * s1 + 0];
            const float qy = p1[(size_t)i * s1 + 1];
            const float qz = p1[(size_t)i * s1 + 2];
            bd = INFINITY;
            for (int j = 0; j < l2; ++j) {
                const float dx = qx - p2[(size_t)j * s2 + 0];
                const float dy = qy - p2[(size_t)j * s2 + 1];
                const float dz = qz - p2[(size_t)j * s2 + 2];
                float d = dx * dx;
                d = fmaf(dy, dy, d);
                d = fmaf(dz, dz, d);
                if (d < bd) { bd = d; bj = j; }               /* first minimum */
            }
        }
        idx[i] = bj;
        d2[i] = bd;
        if (nn) {
            nn[(size_t)i * 3 + 0] = p2[(size_t)bj * s2 + 0];
            nn[(size_t)i * 3 + 1] = p2[(size_t)bj * s2 + 1];
            nn[(size_t)i * 3 + 2] = p2[(size_t)bj * s2 + 2];
        }
    }
    (void)n2;
}

void oracle_knn1(const float *P1, const float *P2, int B, int N1, int N2,
                 int s1, int s2, const int64_t *len1, const int64_t *len2,
                 int64_t *idx, float *d2, float *nn)
{
#pragma omp parallel for schedule(dynamic, 1) num_threads(oracle_num_threads())
    for (int b = 0; b < B; ++b) {
        int l1 = len1 ? (int)len1[b] : N1;
        int l2 = len2 ? (int)len2[b] : N2;
        if (l1 > N1) l1 = N1;
        if (l2 > N2) l2 = N2;
        knn1_one(P1 + (size_t)b * N1 * s1, P2 + (size_t)b * N2 * s2, N1, N2, s1,
                 s2, l1, l2, idx + (size_t)b * N1, d2 + (size_t)b * N1,
                 nn ? nn + (size_t)b * N1 * 3 : NULL);
    }
}
