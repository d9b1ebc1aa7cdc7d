// common.hpp -- shared device helpers for the gfx950 kernels of libicpflow_hip.so.
// wave = 64 lanes everywhere; compile with -ffp-contract=off: every FMA in this
// library is an explicit fmaf()/fma() so that the arithmetic is the one written.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icpflow {

constexpr int kWave = 64;
constexpr float kInf = __builtin_huge_valf();

// ---- 3x4 affine map applied to points ------------------------------------------
// p' = M p + t, evaluated as the reference's bmm([x y z 1], pose^T) does it:
// a left-to-right fp32 dot product per output coordinate (utils_helper.py:85).
struct Affine {
    float m[9];
    float t[3];
};

__device__ __forceinline__ Affine affine_identity()
{
    Affine a;
    a.m[0] = 1.f; a.m[1] = 0.f; a.m[2] = 0.f;
    a.m[3] = 0.f; a.m[4] = 1.f; a.m[5] = 0.f;
    a.m[6] = 0.f; a.m[7] = 0.f; a.m[8] = 1.f;
    a.t[0] = a.t[1] = a.t[2] = 0.f;
    return a;
}

// rows 0..2 of a row-major 4x4
__device__ __forceinline__ Affine affine_from_pose(const float *__restrict__ pose)
{
    Affine a;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        a.m[r * 3 + 0] = pose[r * 4 + 0];
        a.m[r * 3 + 1] = pose[r * 4 + 1];
        a.m[r * 3 + 2] = pose[r * 4 + 2];
        a.t[r] = pose[r * 4 + 3];
    }
    return a;
}

__device__ __forceinline__ void affine_apply(const Affine &a, float x, float y, float z,
                                             float &ox, float &oy, float &oz)
{
    ox = fmaf(z, a.m[2], fmaf(y, a.m[1], x * a.m[0])) + a.t[0];
    oy = fmaf(z, a.m[5], fmaf(y, a.m[4], x * a.m[3])) + a.t[1];
    oz = fmaf(z, a.m[8], fmaf(y, a.m[7], x * a.m[6])) + a.t[2];
}

// ---- how a cloud is presented to a scan ------------------------------------------
enum XfKind : int { XF_NONE = 0, XF_TRANSLATE = 1, XF_AFFINE = 2 };

struct PointXf {
    int kind;
    Affine a;  // XF_TRANSLATE uses a.t only: p' = p + t (utils_hist.py:86)
};

__device__ __forceinline__ void xf_apply(const PointXf &x, float px, float py, float pz,
                                         float &ox, float &oy, float &oz)
{
    if (x.kind == XF_NONE) {
        ox = px; oy = py; oz = pz;
    } else if (x.kind == XF_TRANSLATE) {
        ox = px + x.a.t[0]; oy = py + x.a.t[1]; oz = pz + x.a.t[2];
    } else {
        affine_apply(x.a, px, py, pz, ox, oy, oz);
    }
}

// ---- squared distance, the pytorch3d CUDA order: diff; dist = fma chain ------------
__device__ __forceinline__ float sqdist(float qx, float qy, float qz, float tx, float ty, float tz)
{
    const float dx = qx - tx, dy = qy - ty, dz = qz - tz;
    return fmaf(dz, dz, fmaf(dy, dy, dx * dx));
}

// ---- wave / block reductions (deterministic order) ---------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

// fp64 wave sum on the VALU only (DPP row shifts + row broadcasts, no LDS traffic): after
// the six steps lane 63 holds the total, which is returned wave-uniform via readlane.
// dpp_ctrl: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143 (GFX9 encodings).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_f64(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int l2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int h2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(h2, l2);
}

__device__ __forceinline__ double wave_sum_uniform(double v)
{
    v = dpp_add_f64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_f64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_f64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_f64<0x118, 0xf>(v);  // row_shr:8  -> inclusive scan inside each row of 16
    v = dpp_add_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
    return __hiloint2double(hi, lo);
}

// Folding reductions for MANY values per lane (gfx950 v_permlane32_swap / v_permlane16_swap): one
// swap pair + one add reduce TWO values by one level each -- after swap32_sum lanes 0..31 hold
// a(l) + a(l+32) and lanes 32..63 hold b(l-32) + b(l); after swap16_sum the four rows of 16 lanes
// hold  a: rows 0+1,  b: rows 0+1,  a: rows 2+3,  b: rows 2+3.
__device__ __forceinline__ double swap32_sum(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

__device__ __forceinline__ double swap16_sum(double a, double b)
{
    const auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(a), __double2loint(b), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(a), __double2hiint(b), false, false);
    return __hiloint2double(hi[0], lo[0]) + __hiloint2double(hi[1], lo[1]);
}

// inclusive scan inside each row of 16 lanes: lane 15 of every row ends with the row total
__device__ __forceinline__ double row_sum_f64(double v)
{
    v = dpp_add_f64<0x111, 0xf>(v);
    v = dpp_add_f64<0x112, 0xf>(v);
    v = dpp_add_f64<0x114, 0xf>(v);
    v = dpp_add_f64<0x118, 0xf>(v);
    return v;
}

// fp32 wave min / max on the VALU (same DPP pattern), wave-uniform results
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float v, float identity)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(identity), __float_as_int(v), CTRL, ROW_MASK,
                                                      0xf, false));
}

__device__ __forceinline__ float wave_min_uniform(float v)
{
    v = fminf(v, dpp_f32<0x111, 0xf>(v, kInf));
    v = fminf(v, dpp_f32<0x112, 0xf>(v, kInf));
    v = fminf(v, dpp_f32<0x114, 0xf>(v, kInf));
    v = fminf(v, dpp_f32<0x118, 0xf>(v, kInf));
    v = fminf(v, dpp_f32<0x142, 0xa>(v, kInf));
    v = fminf(v, dpp_f32<0x143, 0xc>(v, kInf));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_max_uniform(float v)
{
    v = fmaxf(v, dpp_f32<0x111, 0xf>(v, -kInf));
    v = fmaxf(v, dpp_f32<0x112, 0xf>(v, -kInf));
    v = fmaxf(v, dpp_f32<0x114, 0xf>(v, -kInf));
    v = fmaxf(v, dpp_f32<0x118, 0xf>(v, -kInf));
    v = fmaxf(v, dpp_f32<0x142, 0xa>(v, -kInf));
    v = fmaxf(v, dpp_f32<0x143, 0xc>(v, -kInf));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// minimum over each group of 8 consecutive lanes, in every lane of the group (quad swaps + half-row mirror):
// quad_perm:[1,0,3,2] = 0xB1, quad_perm:[2,3,0,1] = 0x4E, row_half_mirror = 0x141

__device__ __forceinline__ int row8_min(int v)
{
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, false));
    v = min(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, false));
    return v;
}

// Non-negative floats (squared distances, +inf) order like their bit patterns: integer minima need no NaN quieting
// and fuse with the DPP operand (one instruction per step instead of three); a NaN sorts above +inf.
__device__ __forceinline__ float min_nonneg(float a, float b)
{
    return __int_as_float(min(__float_as_int(a), __float_as_int(b)));
}

__device__ __forceinline__ float max_nonneg(float a, float b)
{
    return __int_as_float(max(__float_as_int(a), __float_as_int(b)));
}

__device__ __forceinline__ float row8_min_nonneg(float v)
{
    return __int_as_float(row8_min(__float_as_int(v)));
}

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for
// outstanding global stores / atomics (vmcnt), which would put an L2 round trip on the path.
__device__ __forceinline__ void barrier_lds_only()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// the same maximum on the VALU (DPP row shifts / broadcasts instead of a dozen ds_bpermute round trips); wave-uniform
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v)
{
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o > v ? o : v;
}

__device__ __forceinline__ unsigned long long wave_max_u64_dpp(unsigned long long v)
{
    v = dpp_max_u64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_max_u64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_max_u64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_max_u64<0x118, 0xf>(v);  // row_shr:8  -> lane 15 of each row holds the row's maximum
    v = dpp_max_u64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_max_u64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
        const unsigned long long w = __shfl_xor(v, o, kWave);
        v = w > v ? w : v;
    }
    return v;
}

// Bitonic sort of NP2 (power of two) (key, index) pairs held in LDS, ascending by key, ties by index (deterministic).
// A pair is ONE 64-bit word: the key as an order-preserving unsigned integer in the high half (sign flip; -0.0 is stored
// as +0.0, which compares equal to it), the non-negative index in the low half -- the lexicographic order of (key, index)
// is the unsigned order of the words, and a compare-exchange is two 8-byte LDS reads, one 64-bit compare and two 8-byte
// writes instead of four reads, three compares and four writes on two arrays (round 3: all 8192 pairs of config 4
// 27.8 -> 27.1 ms per step).  All threads of the block must call bitonic_sort_lds.
__device__ __forceinline__ unsigned long long sort_pack(float key, int index)
{
    unsigned u = __float_as_uint(key + 0.0f);                   // (-0.0 + 0.0 = +0.0)
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned)index;
}
__device__ __forceinline__ float sort_key_of(unsigned long long w)
{
    const unsigned u = (unsigned)(w >> 32);
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ int sort_index_of(unsigned long long w) { return (int)(unsigned)w; }

__device__ __forceinline__ void bitonic_exchange(unsigned long long *kv, int t, int j, int k)
{
    const int i = (t << 1) - (t & (j - 1));   // = (t / j) * 2j + t % j for the power of two j, without the division
    const int l = i + j;
    const unsigned long long a = kv[i], b = kv[l];
    const bool up = (i & k) == 0;
    if ((a > b) == up) { kv[i] = b; kv[l] = a; }
}

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long *kv, int NP2)
{
    const int half = NP2 >> 1;  // one compare-exchange per thread and step
    for (int k = 2; k <= NP2; k <<= 1) {
        // steps with partner distance >= 128 cross the 128-element blocks owned by single waves: barrier each
        int j = k >> 1;
        for (; j > kWave; j >>= 1) {
            for (int t = threadIdx.x; t < half; t += blockDim.x) bitonic_exchange(kv, t, j, k);
            __syncthreads();
        }
        // distance <= 64: the 64 exchanges of a wave stay inside one 128-element block for all remaining
        // steps of the phase, so each wave runs them back to back (the LDS operations of a wave complete in
        // order; the fences only keep the compiler from reordering them): 7 barriers per phase become 1
        for (int t = threadIdx.x; t < half; t += blockDim.x) {
            for (int jj = j; jj > 0; jj >>= 1) {
                bitonic_exchange(kv, t, jj, k);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        __syncthreads();
    }
}

// Number of sorted keys (ascending, n <= 4096 of them, `stride` floats apart, in LDS) that are
// < v (or <= v): two 64-way ballot steps instead of a dependent binary search.  Wave-uniform.
template <bool INCLUSIVE>
__device__ __forceinline__ int sorted_count_below(const float *__restrict__ key, int stride, int n, float v,
                                                  int lane)
{
    const int step = (n + kWave - 1) / kWave;
    if (step == 0) return 0;
    const int s0 = __mul24(lane, step);   // (24-bit multiply: full rate)
    const float k0 = s0 < n ? key[(size_t)s0 * stride] : kInf;
    const unsigned long long m0 = __ballot(INCLUSIVE ? (k0 <= v) : (k0 < v));
    const int cnt = __popcll(m0);
    if (cnt == 0) return 0;
    const int base = __mul24(cnt - 1, step);
    const int s1 = base + lane;
    const float k1 = (lane < step && s1 < n) ? key[(size_t)s1 * stride] : kInf;
    const unsigned long long m1 = __ballot(INCLUSIVE ? (k1 <= v) : (k1 < v));
    return base + __popcll(m1);
}

// Window [count(keys < vlo), count(keys <= vhi)) over n ascending keys by 64-way ballot steps
// (one step per factor of 64 in n: two for n <= 4096, three up to 262144).  The first level shares
// its sample load between both bounds.  Wave-uniform results.
// PAD: the keys sit in LDS with one word of padding behind every 32 (key i at i + (i >> 5)): the first level reads 64 samples
// `step` apart, and a step of 32 (2048 keys) puts all of them into one bank.
// the keys as an array in memory (PAD: see sorted_refine) -- or any callable int -> float (a key computed from coordinates)
template <bool PAD>
struct KeyArray {
    const float *p;
    __device__ __forceinline__ float operator()(int s) const { return p[PAD ? s + (s >> 5) : s]; }
};

template <bool INCLUSIVE, typename KEY>
__device__ __forceinline__ int sorted_refine_fn(const KEY &key, int n, float v, int lane, int base, int span)
{
    while (span > 0) {
        const int step = (span + kWave - 1) / kWave;
        const int ls = __mul24(lane, step);
        const int s = base + ls;
        const float k = (ls < span && s < n) ? key(s) : kInf;
        const int cnt = __popcll(__ballot(INCLUSIVE ? (k <= v) : (k < v)));
        if (cnt == 0) return base;
        if (step == 1) return base + cnt;
        base += __mul24(cnt - 1, step);
        span = min(step, n - base);
        base += 1; span -= 1;
        if (span <= 0) return base;
    }
    return base;
}
template <bool INCLUSIVE, typename KEY>
__device__ __forceinline__ int sorted_refine_hint_fn(const KEY &key, int n, float v, int lane, int hint)
{
    const int base = max(0, min(hint, n) - kWave / 2);
    const int s = base + lane;
    const float k = s < n ? key(s) : kInf;
    const int cnt = __popcll(__ballot(INCLUSIVE ? (k <= v) : (k < v)));
    if ((cnt > 0 || base == 0) && cnt < kWave) return base + cnt;
    return sorted_refine_fn<INCLUSIVE>(key, n, v, lane, 0, n);
}
template <typename KEY>
__device__ __forceinline__ void sorted_window_fn(const KEY &key, int n, float vlo, float vhi, int lane, int &jlo, int &jhi)
{
    jlo = sorted_refine_fn<false>(key, n, vlo, lane, 0, n);
    jhi = sorted_refine_fn<true>(key, n, vhi, lane, 0, n);
}
template <typename KEY>
__device__ __forceinline__ void sorted_window_hint_fn(const KEY &key, int n, float vlo, float vhi, int lane, int &jlo, int &jhi)
{
    jlo = sorted_refine_hint_fn<false>(key, n, vlo, lane, jlo);
    jhi = sorted_refine_hint_fn<true>(key, n, vhi, lane, jhi);
}

template <bool INCLUSIVE, bool PAD = false>
__device__ __forceinline__ int sorted_refine(const float *__restrict__ key, int n, float v, int lane, int base,
                                             int span)
{
    // invariant: the answer lies in [base, base + span]; all keys before `base` compare true
    while (span > 0) {
        const int step = (span + kWave - 1) / kWave;
        const int ls = __mul24(lane, step);   // (24-bit multiply: full rate)
        const int s = base + ls;
        const float k = (ls < span && s < n) ? key[PAD ? s + (s >> 5) : s] : kInf;
        const int cnt = __popcll(__ballot(INCLUSIVE ? (k <= v) : (k < v)));
        if (cnt == 0) return base;
        if (step == 1) return base + cnt;
        base += __mul24(cnt - 1, step);   // key[base] compares true, key[base + step] (if any) does not
        span = min(step, n - base);
        // the sample at `base` itself is known true: search strictly after it
        base += 1; span -= 1;
        if (span <= 0) return base;
    }
    return base;
}

// The same count with a hint (the answer of a nearby earlier query, e.g. the previous ICP iteration):
// ONE probe of the 64 keys around the hint settles it whenever the answer moved by less than 32
// positions; otherwise the full search runs.
template <bool INCLUSIVE>
__device__ __forceinline__ int sorted_refine_hint(const float *__restrict__ key, int n, float v, int lane, int hint)
{
    const int base = max(0, min(hint, n) - kWave / 2);
    const int s = base + lane;
    const float k = s < n ? key[s] : kInf;
    const int cnt = __popcll(__ballot(INCLUSIVE ? (k <= v) : (k < v)));
    // cnt in 1..63: key[base] compares true (so does everything before it), key[base + 63] does not
    if ((cnt > 0 || base == 0) && cnt < kWave) return base + cnt;
    return sorted_refine<INCLUSIVE>(key, n, v, lane, 0, n);
}

__device__ __forceinline__ void sorted_window_hint(const float *__restrict__ key, int n, float vlo, float vhi, int lane,
                                                   int &jlo, int &jhi)
{
    jlo = sorted_refine_hint<false>(key, n, vlo, lane, jlo);
    jhi = sorted_refine_hint<true>(key, n, vhi, lane, jhi);
}

template <bool PAD = false>
__device__ __forceinline__ void sorted_window(const float *__restrict__ key, int n, float vlo, float vhi, int lane,
                                              int &jlo, int &jhi)
{
    jlo = sorted_refine<false, PAD>(key, n, vlo, lane, 0, n);
    jhi = sorted_refine<true, PAD>(key, n, vhi, lane, 0, n);
}

// Sum K values per thread over the whole block.  `scratch` holds at least
// (blockDim.x/64)*K elements of T.  On return every thread has the totals in v[].
// Contains two __syncthreads(); all threads of the block must call it.
template <int K, typename T>
__device__ __forceinline__ void block_sum(T (&v)[K], T *scratch)
{
    const int lane = threadIdx.x & (kWave - 1);
    const int wave = threadIdx.x >> 6;
    const int nwave = (blockDim.x + kWave - 1) >> 6;
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    if (nwave == 1) return;
    __syncthreads();  // scratch may still be read by a previous reduction
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) scratch[wave * K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; ++k) {
        T s = scratch[k];
        for (int w = 1; w < nwave; ++w) s += scratch[w * K + k];
        v[k] = s;
    }
}

}  // namespace icpflow
