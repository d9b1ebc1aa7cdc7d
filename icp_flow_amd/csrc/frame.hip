// frame.hip -- one frame pair per call: the host half of match_pcds (utils_match.py:24-66) around the registration path, in
// C++ (HOST code; the only device work here is what the entry points it calls enqueue).
//
// A stream of frame pairs is bound by the host thread that feeds the GPU (wall time per frame pair = host
// busy time + 0.08 ms), and two thirds of that time is the Python between the calls: cluster-table bookkeeping, candidate
// lists, sanity tests, segment rows, the superset of stage 2.  icpflow_track_frame does all of it on a few hundred numbers of
// the cluster tables in microseconds, in ONE blocking call -- the interpreter lock is released for its whole duration, so
// several host threads (one frame pair each, on their own streams) keep as many frame pairs in flight.
//
// What it follows, line by line, is icp_flow_amd/utils_match.py (match_pcds_steps + _match_pcds_device: the association on
// the device) and utils_check.py (_sanity_mask, sanity_grid), themselves the drop-ins of the reference's
// utils_match.py:24-66 / utils_check.py:21-49: same candidate order, same float32 comparisons (numpy's NaN rules), same
// stream of random draws -- torch.randperm on a torch.Generator seeded with `seed` is a Fisher-Yates shuffle on MT19937
// (ATen randperm_cpu: z = engine() % (n - i), swap(i, i + z)), restated here and pinned against torch in
// tests/test_gpu_parity.py (test_native_frame_pair_equals_the_python_host, case "draws").  The result is bit for bit what utils_match.match_pcds returns with the device-side association.
#include <hip/hip_runtime.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/icpflow_hip.h"

namespace icpflow {
int report_error(int code, const char *message);   // api.hip: what icpflow_last_error returns
}
using icpflow::report_error;

namespace {

constexpr int kTableRows = 512;                    // utils_check.TABLE_ROWS
constexpr int kTableDoubles = 1 + kTableRows * 9;  // [0]: int32 number of clusters, then [L, 9] rows

// host time stamps of the last icpflow_track_frame call of this thread (icpflow_debug_frame_stamps): entry, tables enqueued,
// generator blocks ready, tables on the host, stage 1's candidates, segments + subsamples, stage 1 enqueued, all enqueued, matches read
thread_local double g_frameStamp[16];
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Mt19937 {   // at::mt19937 (the engine of torch's CPU generator)
    uint32_t s[624];
    int idx;
    explicit Mt19937(uint32_t seed)
    {
        s[0] = seed;
        for (int j = 1; j < 624; ++j) s[j] = 1812433253u * (s[j - 1] ^ (s[j - 1] >> 30)) + (uint32_t)j;
        idx = 624;
    }
    explicit Mt19937(const icpflow_mt19937_t &st)
    {
        std::memcpy(s, st.state, sizeof(s));
        idx = std::min(std::max(st.index, 0), 624);
    }
    void save(icpflow_mt19937_t *st) const
    {
        std::memcpy(st->state, s, sizeof(s));
        st->index = idx;
    }
    // the block of 624 words regenerated in place: three loops without index arithmetic (the first reads words the loop has not
    // written yet, the second words written 227 steps earlier: both vectorise)
#if defined(__x86_64__)
    // the same three loops eight words at a time (round 5: the draws nobody reads -- the tail of the 60 000-point wall's
    // permutation -- are ~100 regenerated blocks per frame pair, the larger part of the host's time between the cluster
    // tables and stage 1).  Loop 1 reads p[k+1 .. k+8] and p[k+397 ..] before it writes p[k .. k+7]; loop 2 reads words
    // written 227 steps earlier: the vector steps see exactly what the scalar steps see.
    __attribute__((target("avx2"))) void twist_avx2()
    {
        const __m256i U = _mm256_set1_epi32((int)0x80000000u), L = _mm256_set1_epi32(0x7fffffff), M = _mm256_set1_epi32((int)0x9908b0dfu);
        const __m256i one = _mm256_set1_epi32(1), zero = _mm256_setzero_si256();
        uint32_t *p = s;
        auto step = [&](int k, int src) __attribute__((target("avx2"))) {
            const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + k));
            const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + k + 1));
            const __m256i y = _mm256_or_si256(_mm256_and_si256(a, U), _mm256_and_si256(b, L));
            const __m256i mag = _mm256_and_si256(_mm256_sub_epi32(zero, _mm256_and_si256(y, one)), M);
            const __m256i v = _mm256_xor_si256(_mm256_xor_si256(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(p + src)), _mm256_srli_epi32(y, 1)), mag);
            _mm256_storeu_si256(reinterpret_cast<__m256i *>(p + k), v);
        };
        constexpr uint32_t Us = 0x80000000u, Ls = 0x7fffffffu, Ms = 0x9908b0dfu;
        int k = 0;
        for (; k + 8 <= 227; k += 8) step(k, k + 397);
        for (; k < 227; ++k) { const uint32_t y = (p[k] & Us) | (p[k + 1] & Ls); p[k] = p[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & Ms); }
        for (; k + 8 <= 623; k += 8) step(k, k - 227);
        for (; k < 623; ++k) { const uint32_t y = (p[k] & Us) | (p[k + 1] & Ls); p[k] = p[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & Ms); }
        const uint32_t y = (p[623] & Us) | (p[0] & Ls);
        p[623] = p[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & Ms);
        idx = 0;
    }
#endif
    void twist()
    {
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) { twist_avx2(); return; }
#endif
        constexpr uint32_t U = 0x80000000u, L = 0x7fffffffu, M = 0x9908b0dfu;
        uint32_t *p = s;
        for (int k = 0; k < 227; ++k) {
            const uint32_t y = (p[k] & U) | (p[k + 1] & L);
            p[k] = p[k + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & M);
        }
        for (int k = 227; k < 623; ++k) {
            const uint32_t y = (p[k] & U) | (p[k + 1] & L);
            p[k] = p[k - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & M);
        }
        const uint32_t y = (p[623] & U) | (p[0] & L);
        p[623] = p[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & M);
        idx = 0;
    }
    uint32_t next()
    {
        if (idx >= 624) twist();
        uint32_t y = s[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    // n draws whose values nobody reads: only the state moves
    void skip(int64_t n)
    {
        while (n > 0) {
            if (idx >= 624) twist();
            const int64_t step = std::min<int64_t>(n, 624 - idx);
            idx += (int)step;
            n -= step;
        }
    }
};

// The generator's output as a STREAM that can be generated ahead of its use: the state blocks (624 words each, untempered) from the
// generator's current one on, kept one after the other.  Draw number p of the stream is word (idx0 + p) % 624 of block
// (idx0 + p) / 624, tempered on demand; consuming draws is moving a position.  icpflow_track_frame regenerates the blocks a
// frame pair may need while it waits for the cluster tables (the GPU is busy with them for ~0.1 ms, the host is not): what
// is left between the tables and stage 1 is the few thousand draws that are actually read.
struct MtStream {
    std::vector<uint32_t> &blocks;  // [nBlocks][624] (the caller's buffer: it keeps its capacity from frame pair to frame pair)
    Mt19937 tail;                   // the generator at the last block generated
    int idx0;                       // position inside block 0 at which the stream starts (624: block 0 is used up)
    int64_t pos = 0;                // draws consumed
    int64_t firstWord = 0;          // words of the stream (block 0 = words 0 .. 623) in front of blocks[0]: blocks behind the position are let go
    MtStream(const Mt19937 &g, std::vector<uint32_t> &buffer) : blocks(buffer), tail(g), idx0(g.idx)
    {
        blocks.clear();
        if (blocks.capacity() < ((size_t)1 << 18) + 3 * 624) blocks.reserve(((size_t)1 << 18) + 3 * 624);
        blocks.insert(blocks.end(), g.s, g.s + 624);
    }
    int64_t covered() const { return firstWord + (int64_t)(blocks.size() / 624) * 624 - idx0; }   // draws the blocks reach
    void ensure(int64_t draws)      // blocks for the draws pos .. `draws` - 1 of the stream (and the one state() reads)
    {
        //  Draws are consumed in order, and randperm_head passes over most of a long permutation's draws unread (n - 1
        // per draw of a 60 000-point cluster): the blocks that end in front of the position are never read again.  They are
        // dropped -- and those not generated yet are passed over without being stored --, so the buffer holds what lies between the
        // position and the furthest draw asked for (plus the block in front of the position's, which state() may hand back),
        // not every block of the frame pair.
        const int64_t keep = std::max<int64_t>(((idx0 + pos) / 624 - 1) * 624, 0);   // first word that may still be read
        if (keep >= firstWord + (int64_t)blocks.size()) {             // everything held is behind the position
            firstWord += (int64_t)blocks.size();
            blocks.clear();
            while (firstWord + 624 <= keep) { tail.twist(); firstWord += 624; }   // (generated, not stored)
        } else if (keep - firstWord >= 64 * 624) {                    // a long dead prefix: let it go
            blocks.erase(blocks.begin(), blocks.begin() + (size_t)(keep - firstWord));
            firstWord = keep;
        }
        while (covered() < draws || blocks.empty()) {
            tail.twist();
            blocks.insert(blocks.end(), tail.s, tail.s + 624);
        }
    }
    static uint32_t temper(uint32_t y)
    {
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
    uint32_t draw(int64_t p) const { return temper(blocks[(size_t)(idx0 + p - firstWord)]); }   // (pos <= p < covered(), after ensure)
    // the generator after `pos` draws, in the form the draw-by-draw engine leaves it (a block used up stays, with idx = 624)
    Mt19937 state() const
    {
        const int64_t q = idx0 + pos;
        int64_t b = q / 624;
        int i = (int)(q % 624);
        if (i == 0 && b > 0) { --b; i = 624; }
        Mt19937 g(0u);
        std::memcpy(g.s, blocks.data() + (size_t)(b * 624 - firstWord), sizeof(g.s));
        g.idx = i;
        return g;
    }
};

// torch.randperm(n, generator)[0:take] as int32 (n < 2^32 / 20: the 32-bit branch of randperm_cpu)
// `tmp` is kept as the IDENTITY between calls (entries are set when it grows; the places a call writes are put back at its
// end): writing 60 000 indices per draw cost as much as the draw itself.
#if defined(__x86_64__)
// z[k] = temper(raw[k]) % (n - (i0 + k)) for k < count (count a multiple of 8), eight at a time: the tempering on 32-bit lanes, the
// remainder through a double division -- exact: the quotient r / m of two integers below 2^32 is at least 1 / m below the next
// integer and q m <= r < 2^32, so the correctly rounded quotient (53 bits) truncates to floor(r / m); r - q m is exact in double.
__attribute__((target("avx2"))) void draws_avx2(const uint32_t *raw, uint32_t n32, uint32_t i0, int count, uint32_t *z)
{
    const __m256i c7 = _mm256_set1_epi32((int)0x9d2c5680u), c15 = _mm256_set1_epi32((int)0xefc60000u), sign = _mm256_set1_epi32((int)0x80000000u);
    const __m256d two31 = _mm256_set1_pd(2147483648.0);
    const __m128i ramp = _mm_setr_epi32(0, 1, 2, 3);
    for (int k = 0; k < count; k += 8) {
        __m256i y = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(raw + k));
        y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 11));
        y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 7), c7));
        y = _mm256_xor_si256(y, _mm256_and_si256(_mm256_slli_epi32(y, 15), c15));
        y = _mm256_xor_si256(y, _mm256_srli_epi32(y, 18));
        const __m256i ys = _mm256_xor_si256(y, sign);                       // (unsigned -> signed + 2^31)
        for (int h = 0; h < 2; ++h) {
            const __m128i part = h ? _mm256_extracti128_si256(ys, 1) : _mm256_castsi256_si128(ys);
            const __m256d r = _mm256_add_pd(_mm256_cvtepi32_pd(part), two31);
            const uint32_t m0 = n32 - (i0 + (uint32_t)k + 4u * (uint32_t)h);
            const __m256d m = _mm256_sub_pd(_mm256_set1_pd((double)m0), _mm256_cvtepi32_pd(ramp));
            const __m256d q = _mm256_round_pd(_mm256_div_pd(r, m), _MM_FROUND_TO_ZERO | _MM_FROUND_NO_EXC);
            const __m256d rem = _mm256_sub_pd(r, _mm256_mul_pd(q, m));
            _mm_storeu_si128(reinterpret_cast<__m128i *>(z + k + 4 * h), _mm256_cvttpd_epi32(rem));
        }
    }
}
#endif

void randperm_head(MtStream &g, int64_t n, int take, int32_t *out, std::vector<int32_t> &tmp)
{
    static thread_local std::vector<int32_t> places;
    if ((int64_t)tmp.size() < n) {
        const size_t old = tmp.size();
        tmp.resize((size_t)n);
        for (size_t i = old; i < (size_t)n; ++i) tmp[i] = (int32_t)i;
    }
    // step i of the shuffle swaps places i and i + z and fixes place i for good: the first `take` steps give the head; the
    // rest of the shuffle only consumes its draws (the next randperm continues on the same stream).  Place i is never read
    // again, so only the other half of the swap is carried out.
    const int64_t steps = std::min<int64_t>(take, n - 1);
    if ((int64_t)places.size() < steps) places.resize((size_t)steps);
    g.ensure(g.pos + steps);
    const uint32_t n32 = (uint32_t)n;
    int32_t *t = tmp.data(), *pl = places.data();
    const int64_t p0 = g.pos;
    int64_t i = 0;
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2 && n32 < (1u << 28)) {
        // the remainders of the steps eight at a time (tempering and division vectorise; the swaps stay a chain through `t`)
        static thread_local std::vector<uint32_t> zs;
        const int64_t vec = steps / 8 * 8;
        if ((int64_t)zs.size() < vec) zs.resize((size_t)vec);
        draws_avx2(g.blocks.data() + (size_t)(g.idx0 + p0 - g.firstWord), n32, 0u, (int)vec, zs.data());
        for (; i < vec; ++i) {
            const int32_t place = (int32_t)((uint32_t)i + zs[(size_t)i]);
            out[i] = t[place];
            t[place] = t[i];
            pl[i] = place;
        }
    }
#endif
    for (; i < steps; ++i) {
        // (on 32-bit operands: the draw is a 32-bit word and n - i < 2^32 -- the value of the reference's 64-bit remainder)
        const uint32_t z = g.draw(p0 + i) % (n32 - (uint32_t)i);
        const int32_t place = (int32_t)((uint32_t)i + z);
        out[i] = t[place];
        t[place] = t[i];
        pl[i] = place;
    }
    if (steps < take) out[steps] = t[steps];   // (take == n: the last element stays where the shuffle left it)
    g.pos += n - 1;                            // (the draws of the steps nobody reads are passed over)
    for (int64_t k = 0; k < steps; ++k) t[pl[k]] = pl[k];   // back to the identity
}

// numpy's minimum / maximum: a NaN on either side gives NaN
inline float np_min(float a, float b) { return (std::isnan(a) || std::isnan(b)) ? NAN : (a < b ? a : b); }
inline float np_max(float a, float b) { return (std::isnan(a) || std::isnan(b)) ? NAN : (a > b ? a : b); }

struct Table {   // host copy of one cluster table (utils_check.ClusterTable._set_host)
    int L = 0;
    std::vector<float> label, mean, extent;   // [L], [L,3], [L,3]
    std::vector<int64_t> count, start;
    void set(const double *packed)
    {
        int32_t n;
        std::memcpy(&n, packed, 4);
        L = n;
        if (L < 0) return;
        label.resize(L); count.resize(L); start.resize(L); mean.resize(3 * (size_t)L); extent.resize(3 * (size_t)L);
        const double *rows = packed + 1;
        for (int r = 0; r < L; ++r) {
            label[r] = (float)rows[r * 9 + 0];
            count[r] = (int64_t)rows[r * 9 + 1];
            start[r] = (int64_t)rows[r * 9 + 2];
            for (int k = 0; k < 3; ++k) {
                mean[3 * r + k] = (float)rows[r * 9 + 3 + k];
                extent[3 * r + k] = (float)rows[r * 9 + 6 + k];
            }
        }
    }
    int find(float wanted) const   // ClusterTable.find_host: rows are in ascending label order
    {
        if (L <= 0) return -1;
        const int pos = std::min((int)(std::lower_bound(label.begin(), label.end(), wanted) - label.begin()), L - 1);
        return label[pos] == wanted ? pos : -1;
    }
};

// the pairwise part of sanity_check (utils_check.py:36, 41-43) on table rows s, d
inline bool pair_passes(const Table &st, const Table &dt, int s, int d, float translationFrame, float thresBox)
{
    const float dx = dt.mean[3 * d] - st.mean[3 * s], dy = dt.mean[3 * d + 1] - st.mean[3 * s + 1];
    const float xx = dx * dx, yy = dy * dy;
    const float sum = xx + yy;
    // sqrt(sum) > translation_frame, decided without the root away from the threshold (the root is correctly rounded and
    // monotone: it can only disagree with the comparison of the squares within a few ulps of equality)
    const float t2 = translationFrame * translationFrame;
    if (!(translationFrame >= 0.f)) {          // (a negative threshold rejects every pair, a NaN none: the comparison as written)
        if (std::sqrt(sum) > translationFrame) return false;
    } else {
        if (sum > t2 * 1.000001f) return false;
        if (!(sum < t2 * 0.999999f) && std::sqrt(sum) > translationFrame) return false;
    }
    for (int k = 0; k < 3; ++k) {
        const float es = st.extent[3 * s + k], ed = dt.extent[3 * d + k];
        const float rhs = thresBox * np_max(es, ed);
        if (np_min(es, ed) < rhs) return false;
    }
    return true;
}

// the destination rows that pass the pairwise sanity test with source row s (and are usable themselves: dOk), in ascending
// order -- what the loop "for b: if (dOk[b] && pair_passes(s, b))" visits.  A frame has ~150 x 150 cluster pairs of which a
// few hundred pass: the distance test runs first over contiguous arrays (a loop the compiler vectorises), the exact test
// (the root near the threshold, the extents) only on its survivors.
struct PassRows {
    std::vector<float> mx, my;      // destination means
    std::vector<uint8_t> near;
    void init(const Table &dt)
    {
        mx.resize(dt.L); my.resize(dt.L); near.resize(dt.L);
        for (int d = 0; d < dt.L; ++d) { mx[d] = dt.mean[3 * d]; my[d] = dt.mean[3 * d + 1]; }
    }
    template <typename F>
    void for_each(const Table &st, const Table &dt, const std::vector<uint8_t> &dOk, int s, float tf, float tb, F &&visit)
    {
        const int D = dt.L;
        const float sx = st.mean[3 * s], sy = st.mean[3 * s + 1];
        const float lim = tf * tf * 1.000001f;
        const bool filter = tf >= 0.f;      // (a negative or NaN threshold: the comparison as written decides, see pair_passes)
        const float *px = mx.data(), *py = my.data();
        const uint8_t *ok = dOk.data();
        uint8_t *nr = near.data();
        for (int d = 0; d < D; ++d) {
            const float dx = px[d] - sx, dy = py[d] - sy;
            const float sum = dx * dx + dy * dy;
            nr[d] = (uint8_t)(ok[d] & (uint8_t)(!filter | !(sum > lim)));   // (a NaN is not rejected here, like in pair_passes)
        }
        for (int d = 0; d < D; ++d)
            if (nr[d] && pair_passes(st, dt, s, d, tf, tb)) visit(d);
    }
};

struct Pinned {   // a pinned host buffer that grows (read in place by the kernels / target of the read-backs)
    void *ptr = nullptr;
    size_t bytes = 0;
    int device = -1;
    Pinned() = default;
    Pinned(const Pinned &) = delete;
    Pinned &operator=(const Pinned &) = delete;
    ~Pinned() { if (ptr != nullptr) (void)hipHostFree(ptr); }   // (thread_local: freed when the host thread ends)
    char *need(size_t want)
    {
        int dev = -1;
        (void)hipGetDevice(&dev);
        if (ptr == nullptr || bytes < want || device != dev) {
            if (ptr != nullptr) (void)hipHostFree(ptr);
            ptr = nullptr;
            bytes = std::max(want, (size_t)1 << 20);
            if (hipHostMalloc(&ptr, bytes, hipHostMallocDefault) != hipSuccess) { ptr = nullptr; bytes = 0; }
            device = dev;
        }
        return static_cast<char *>(ptr);
    }
};

struct Host {   // per host thread
    Pinned main, second;   // (second: the exact stage 2 of a frame pair whose superset fell short -- the first is still in use then)
    std::vector<int32_t> perm;
    std::vector<uint32_t> draws;   // the generator's blocks of the frame pair at hand (MtStream)
    char *need(size_t bytes) { return main.need(bytes); }
};

// Wait for the stream by POLLING it (then, should that last milliseconds, by blocking in the runtime): a frame pair has two
// waits of a fraction of a millisecond, and waking a blocked host thread costs more than that when several threads keep frame
// pairs in flight (tools/dbg/define_sweep.sh, blocking -> polling: the demo frame pair 1.60 -> 1.52 / 1.91 -> 1.85 ms on its own,
// 0.86 -> 0.72 / 1.01 -> 0.91 ms per frame pair with four in flight).
inline hipError_t wait_stream(hipStream_t s)
{
    // (bounded by TIME: 20 ms of polling, then the runtime blocks.  A frame pair's own waits are fractions of a millisecond, with
    // several frame pairs in flight a few milliseconds: a bound of 2 ms sent those waits into the blocking call and cost a stream
    // of frame pairs 0.75 -> 1.5 ms per frame pair -- round 5)
    const auto t0 = std::chrono::steady_clock::now();
    for (int spin = 0;; ++spin) {
        const hipError_t e = hipStreamQuery(s);
        if (e != hipErrorNotReady) return e;
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#endif
        if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
    }
    return hipStreamSynchronize(s);
}

// the stream stage 2's first half runs on (one per host thread and device; never destroyed: the thread's frame pairs reuse it)
struct SecondStream {
    hipStream_t stream = nullptr;
    hipEvent_t join = nullptr;
    int device = -1;
    bool ok = false;
};
inline SecondStream &second_stream()
{
    static thread_local SecondStream st[8];
    static thread_local SecondStream none;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) return none;
    SecondStream &e = st[dev & 7];
    if (e.device != dev) {
        e = SecondStream{};
        e.device = dev;
        // (a plain second stream.  HIP multiplexes a process's streams onto a few hardware queues (GPU_MAX_HW_QUEUES, four by
        // default): where the second stream lands in the caller's queue it runs behind stage 1, not beside it -- the overlap then
        // buys nothing and costs nothing (measured: -0.09 ms per demo frame pair from the default stream of a fresh process or
        // with GPU_MAX_HW_QUEUES=16, +-0 from a stream that shares its queue).  A stream of another PRIORITY always gets a queue
        // of its own -- and takes it from the pool every other stream of the process shares: frame pairs in flight 0.60 -> 0.82 ms,
        // four batches in one call 500 -> 466 k registrations/s when tried (ICPFLOW_SECOND_STREAM_PRIORITY 0 / 1: lowest / highest))
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
#ifndef ICPFLOW_SECOND_STREAM_PRIORITY
#define ICPFLOW_SECOND_STREAM_PRIORITY 2
#endif
        const int prio = ICPFLOW_SECOND_STREAM_PRIORITY ? greatest : least;
        e.ok = (ICPFLOW_SECOND_STREAM_PRIORITY == 2 ? hipStreamCreateWithFlags(&e.stream, hipStreamNonBlocking)
                                                    : hipStreamCreateWithPriority(&e.stream, hipStreamNonBlocking, prio)) == hipSuccess &&
               hipEventCreateWithFlags(&e.join, hipEventDisableTiming) == hipSuccess;
    }
    return e;
}

inline size_t up(size_t x) { return (x + 255) & ~(size_t)255; }
inline int round64(int64_t x) { return (int)((x + 63) / 64 * 64); }

}  // namespace

extern "C" int icpflow_track_frame(const float *d_points_src, const float *d_labels_src, int n_src, const float *d_points_dst,
                                   const float *d_labels_dst, int n_dst, const icpflow_registration_t *reg,
                                   const icpflow_frame_params_t *par, float *d_rows, float *d_T, int32_t *h_pairs,
                                   const float *d_flow_points, const float *d_pose, float *d_flow, void *d_scratch,
                                   size_t scratch_bytes, size_t *scratch_needed, icpflow_stream_t stream,
                                   const icpflow_options_t *opt)
{
    if (!d_points_src || !d_labels_src || !d_points_dst || !d_labels_dst || !reg || !par || !d_rows || !d_T || !h_pairs ||
        !scratch_needed)
        return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: null pointer");
    if (par->struct_size != sizeof(icpflow_frame_params_t) || n_src <= 0 || n_dst <= 0 || par->max_points <= 0)
        return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: params.struct_size, n_src, n_dst, max_points must be valid / positive");
    if ((d_flow != nullptr) != (d_flow_points != nullptr) || (d_flow != nullptr && d_pose == nullptr))
        return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: d_flow comes with d_flow_points and d_pose");
    *h_pairs = ICPFLOW_FRAME_HOST_PATH;
    hipStream_t s = (hipStream_t)stream;
    static thread_local Host H;
    g_frameStamp[0] = now_us();

    // ---- device scratch, first part: the label-sorted orders, both tables, the tables' workspace
    char *base = static_cast<char *>(d_scratch);
    size_t off = 0;
    const size_t oOrderS = off; off += up(sizeof(int64_t) * (size_t)n_src);
    const size_t oOrderD = off; off += up(sizeof(int64_t) * (size_t)n_dst);
    const size_t oTables = off; off += up(sizeof(double) * 2 * kTableDoubles);
    const size_t tableWs = icpflow_cluster_table_pair_workspace_bytes(n_src, n_dst, kTableRows);
    const size_t oTableWs = off; off += up(tableWs);
    const size_t fixedBytes = off;
    *scratch_needed = fixedBytes;
    if (d_scratch == nullptr || scratch_bytes < fixedBytes) return report_error(ICPFLOW_E_WORKSPACE, "icpflow_track_frame: scratch too small (see *scratch_needed)");
    int64_t *orderS = reinterpret_cast<int64_t *>(base + oOrderS), *orderD = reinterpret_cast<int64_t *>(base + oOrderD);
    double *tabS = reinterpret_cast<double *>(base + oTables), *tabD = tabS + kTableDoubles;

    // ---- both cluster tables, one read-back (ClusterTable.pair)
    if (int r = icpflow_cluster_table_pair(d_points_src, d_labels_src, n_src, orderS, tabS + 1, reinterpret_cast<int32_t *>(tabS),
                                           d_points_dst, d_labels_dst, n_dst, orderD, tabD + 1, reinterpret_cast<int32_t *>(tabD),
                                           kTableRows, base + oTableWs, tableWs, stream))
        return r;
    const size_t tableBytes = sizeof(double) * 2 * kTableDoubles;
    char *pin = H.need(tableBytes);
    if (pin == nullptr) return report_error(ICPFLOW_E_HOSTMEM, "icpflow_track_frame: no pinned host memory");
    if (hipMemcpyAsync(pin, tabS, tableBytes, hipMemcpyDeviceToHost, s) != hipSuccess)
        return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: the read-back of the cluster tables failed");
    // the frame pair's stream of draws: the caller's generator (advanced only if this call serves the frame pair) or a seed.
    // While the GPU forms the tables: the blocks of the generator that the subsamples of over-long clusters can consume -- a
    // randperm of every point of both clouds at most (frames without such a cluster: ~60 us of a host thread that waits anyway)
    g_frameStamp[1] = now_us();   // tables enqueued
    MtStream gen(par->generator != nullptr ? Mt19937(*par->generator) : Mt19937((uint32_t)par->seed), H.draws);
    gen.ensure(std::min<int64_t>((int64_t)n_src + n_dst, (int64_t)1 << 18));
    g_frameStamp[2] = now_us();   // blocks of the generator ready
    if (wait_stream(s) != hipSuccess)
        return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: the read-back of the cluster tables failed");
    g_frameStamp[3] = now_us();   // tables on the host
    Table st, dt;
    st.set(reinterpret_cast<const double *>(pin));
    dt.set(reinterpret_cast<const double *>(pin) + kTableDoubles);
    if (st.L < 0 || dt.L < 0) return 0;   // more distinct labels than the device table holds: the host path (torch ops)
    const int S = st.L, D = dt.L;
    if (S == 0 || D == 0) return 0;

    const float tf = par->translation_frame, tb = par->thres_box;
    const int minSize = par->min_cluster_size, maxPoints = par->max_points;

    // ---- stage 1: the labels both clouds carry (utils_match.py:29-35), through sanity_check
    // labels_unq = unique(int64(labels of either cloud)), >= 0; looked up as float32 in both tables
    std::vector<int64_t> unq;
    unq.reserve((size_t)S + D);
    for (int r = 0; r < S; ++r) unq.push_back((int64_t)st.label[r]);
    for (int r = 0; r < D; ++r) unq.push_back((int64_t)dt.label[r]);
    std::sort(unq.begin(), unq.end());
    unq.erase(std::unique(unq.begin(), unq.end()), unq.end());
    std::vector<int32_t> si1, di1;
    for (const int64_t v : unq) {
        if (v < 0) continue;
        const float f = (float)v;
        const int a = st.find(f), b = dt.find(f);
        if (a < 0 || b < 0) continue;
        if (std::min(st.count[a], dt.count[b]) < minSize) continue;
        if (!(f >= 0.f)) continue;
        if (!pair_passes(st, dt, a, b, tf, tb)) continue;
        si1.push_back(a);
        di1.push_back(b);
    }
    std::vector<uint8_t> sOk(S), dOk(D);   // (the per-cluster half of sanity_check, utils_check.py:31-32)
    for (int r = 0; r < S; ++r) sOk[r] = st.count[r] >= minSize && st.label[r] >= 0.f;
    for (int r = 0; r < D; ++r) dOk[r] = dt.count[r] >= minSize && dt.label[r] >= 0.f;
    // No cluster keeps its label and passes: the reference goes on with stage 2 alone, every source against every destination
    // through the sanity check (utils_match.py:42-53 with nothing matched) -- registered here as the only stage, in its order.
    const bool stage2Only = si1.empty();
    PassRows rowsOf;
    rowsOf.init(dt);
    if (stage2Only)
        for (int a = 0; a < S; ++a) {
            if (!sOk[a]) continue;
            rowsOf.for_each(st, dt, dOk, a, tf, tb, [&](int b) { si1.push_back(a); di1.push_back(b); });
        }
    const int K1 = (int)si1.size();
    g_frameStamp[4] = now_us();   // stage 1's candidates
    if (K1 == 0) return 0;   // (nothing to register at all: left to the caller)

    // ---- stage 1's segment rows and subsamples (utils_match._stage_rows), in pinned memory the kernels read in place
    int64_t longest = 0;
    int nPerm = 0;
    for (int k = 0; k < K1; ++k) {
        longest = std::max(longest, std::max(st.count[si1[k]], dt.count[di1[k]]));
        nPerm += (st.count[si1[k]] > maxPoints) + (dt.count[di1[k]] > maxPoints);
    }
    const int N1 = par->tight_padding ? std::min(maxPoints, std::max(64, round64(longest))) : maxPoints;

    // ---- stage 2's superset (utils_match._match_pcds_device): every pair of the sanity grid, less the over-long clusters
    const int capPts = std::min(maxPoints, par->superset_width > 0 ? par->superset_width : 1024);
    std::vector<int32_t> si2, di2, leftS, leftD;
    int64_t longest2 = 0;
    for (int a = 0; a < S && !stage2Only; ++a) {
        if (!sOk[a]) continue;
        rowsOf.for_each(st, dt, dOk, a, tf, tb, [&](int b) {
            if (st.count[a] > capPts || dt.count[b] > capPts) {
                leftS.push_back(a);
                leftD.push_back(b);
            } else {
                si2.push_back(a);
                di2.push_back(b);
                longest2 = std::max(longest2, std::max(st.count[a], dt.count[b]));
            }
        });
    }
    const int K2 = (int)si2.size();
    int N2 = K2 ? std::min(capPts, std::max(64, round64(longest2))) : 64;
    if (!par->tight_padding) N2 = maxPoints;

    g_frameStamp[9] = now_us();    // (sub-stamps of "segments + draws": the superset)
    // ---- device scratch, second part
    const size_t ws1 = icpflow_workspace_bytes(K1, N1, reg->len_x, reg->len_y, reg->len_z);
    const size_t ws2 = K2 ? icpflow_workspace_bytes(K2, N2, reg->len_x, reg->len_y, reg->len_z) : 0;
    const size_t oClouds1 = off; off += up(sizeof(float) * 8 * (size_t)K1 * N1);
    const size_t oRes1 = off; off += up(sizeof(float) * (30 * (size_t)K1 + 1));
    const size_t oClouds2 = off; off += up(sizeof(float) * 8 * (size_t)K2 * N2);
    const size_t oRes2 = off; off += up(sizeof(float) * (30 * (size_t)K2 + 1));
    const size_t oActive = off; off += up((size_t)K2 + 1);
    const size_t oBest = off; off += up(sizeof(int32_t) * (2 * (size_t)S + 2));
    const size_t oWs = off; off += up(ws1);
    const size_t oWs2 = off; off += up(ws2);     // (a workspace of its own: stage 2 estimates its initial poses beside stage 1's ICP)
    *scratch_needed = off;
    if (scratch_bytes < off) return report_error(ICPFLOW_E_WORKSPACE, "icpflow_track_frame: scratch too small (see *scratch_needed)");

    g_frameStamp[10] = now_us();   // (workspace sizes)
    // pinned staging: [tables (done with) | seg1 int64 [2,3,K1] | perm int32 [nPerm, maxPoints] | seg2 int64 [2,3,K2] |
    //                  si1, di1, si2, di2 int32 | best int32 [2S+2]]
    const size_t pSeg1 = 0, pPerm = pSeg1 + 48 * (size_t)K1, pSeg2 = up(pPerm + 4 * (size_t)nPerm * maxPoints);
    const size_t pIdx = pSeg2 + 48 * (size_t)K2, pBest = up(pIdx + 8 * ((size_t)K1 + K2));
    const size_t pinBytes = pBest + 4 * (2 * (size_t)S + 2);
    pin = H.need(pinBytes);   // (the tables have been parsed: the buffer may move)
    if (pin == nullptr) return report_error(ICPFLOW_E_HOSTMEM, "icpflow_track_frame: no pinned host memory");
    int64_t *seg1 = reinterpret_cast<int64_t *>(pin + pSeg1);
    int32_t *perm = reinterpret_cast<int32_t *>(pin + pPerm);
    int64_t *seg2 = reinterpret_cast<int64_t *>(pin + pSeg2);
    int32_t *idx = reinterpret_cast<int32_t *>(pin + pIdx);
    int32_t *hBest = reinterpret_cast<int32_t *>(pin + pBest);
    // the frame pair's stream of draws: the caller's generator (advanced only if this call serves the frame pair) or a seed
    {
        int drawn = 0;
        for (int k = 0; k < K1; ++k) {
            const int64_t cs = st.count[si1[k]], cd = dt.count[di1[k]];
            seg1[0 * K1 + k] = st.start[si1[k]]; seg1[1 * K1 + k] = std::min<int64_t>(cs, maxPoints); seg1[2 * K1 + k] = -1;
            seg1[3 * K1 + k] = dt.start[di1[k]]; seg1[4 * K1 + k] = std::min<int64_t>(cd, maxPoints); seg1[5 * K1 + k] = -1;
            if (cs > maxPoints) {   // random_choice, utils_helper.py:198-201: src then dst, pair by pair
                seg1[2 * K1 + k] = (int64_t)drawn * maxPoints;
                randperm_head(gen, cs, maxPoints, perm + (size_t)drawn * maxPoints, H.perm);
                ++drawn;
            }
            if (cd > maxPoints) {
                seg1[5 * K1 + k] = (int64_t)drawn * maxPoints;
                randperm_head(gen, cd, maxPoints, perm + (size_t)drawn * maxPoints, H.perm);
                ++drawn;
            }
        }
    }
    g_frameStamp[11] = now_us();   // (stage 1's segments and draws)
    for (int k = 0; k < K2; ++k) {
        seg2[0 * K2 + k] = st.start[si2[k]]; seg2[1 * K2 + k] = st.count[si2[k]]; seg2[2 * K2 + k] = -1;
        seg2[3 * K2 + k] = dt.start[di2[k]]; seg2[4 * K2 + k] = dt.count[di2[k]]; seg2[5 * K2 + k] = -1;
    }
    std::memcpy(idx, si1.data(), 4 * (size_t)K1);
    std::memcpy(idx + K1, di1.data(), 4 * (size_t)K1);
    if (K2) {
        std::memcpy(idx + 2 * K1, si2.data(), 4 * (size_t)K2);
        std::memcpy(idx + 2 * K1 + K2, di2.data(), 4 * (size_t)K2);
    }

    g_frameStamp[5] = now_us();   // segments, subsamples, stage 2's superset
    // ---- everything else is enqueued: stage 1, then the assignment / stage 2 / assignment / pair rows / flow
    icpflow_tables_t tables{d_points_src, orderS, tabS + 1, d_points_dst, orderD, tabD + 1, S, D, 9};
    icpflow_stage_t stage1{seg1, nPerm ? perm : nullptr, idx, idx + K1, reinterpret_cast<float *>(base + oClouds1),
                           reinterpret_cast<float *>(base + oRes1), K1, N1};
    icpflow_stage_t stage2{seg2, nullptr, idx + 2 * K1, idx + 2 * K1 + K2, reinterpret_cast<float *>(base + oClouds2),
                           reinterpret_cast<float *>(base + oRes2), K2, N2};
    if (int r = icpflow_register_stage(&tables, &stage1, reg, base + oWs, ws1, stream, opt)) return r;
    // Stage 2's first half -- clouds of the whole superset, vote, peaks, scoring: ~0.12 ms of kernels -- on a second stream, where
    // it runs beside stage 1's ICP (100 dependent iterations on a few long pairs: most of the GPU is idle meanwhile) instead of
    // behind it.  Everything it reads is complete (the tables were waited for; the segment rows are host memory written above).
    // The caller's stream waits for it BEFORE the assignment, which rewrites stage 2's segment rows.
    int32_t carry2[4] = {0, 0, 0, 0};
    bool begun = false;
    if (K2 > 0 && !(opt != nullptr && (opt->flags & ICPFLOW_OPT_NO_STAGE_OVERLAP) != 0u)) {
        SecondStream &b2 = second_stream();
        if (b2.ok) {
            // (whatever goes wrong from here on, the second stream is drained before the call returns: its kernels write the scratch)
            if (int r = icpflow_register_stage_begin(&tables, &stage2, reg, base + oWs2, ws2, (icpflow_stream_t)b2.stream, opt, carry2)) {
                (void)hipStreamSynchronize(b2.stream);
                return r;
            }
            if (hipEventRecord(b2.join, b2.stream) != hipSuccess || hipStreamWaitEvent(s, b2.join, 0) != hipSuccess) {
                (void)hipStreamSynchronize(b2.stream);
                return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: joining the second stream failed");
            }
            begun = true;
        }
    }
    g_frameStamp[6] = now_us();   // stage 1 enqueued
    int32_t *dBest = reinterpret_cast<int32_t *>(base + oBest);
    const int cap = 2 * S;
    if (begun) {
        if (int r = icpflow_associate_frame_begun(&tables, &stage1, &stage2, reinterpret_cast<uint8_t *>(base + oActive), reg,
                                                  par->translation_frame, par->thres_iou, par->rot_limit_deg, par->thres_error, dBest, cap,
                                                  d_rows, d_T, d_flow_points, d_flow != nullptr ? d_labels_src : nullptr, n_src, d_pose,
                                                  d_flow, base + oWs2, ws2, stream, opt, carry2))
            return r;
    } else
    if (int r = icpflow_associate_frame(&tables, &stage1, K2 ? &stage2 : nullptr, reinterpret_cast<uint8_t *>(base + oActive), reg,
                                        par->translation_frame, par->thres_iou, par->rot_limit_deg, par->thres_error, dBest, cap,
                                        d_rows, d_T, d_flow_points, d_flow != nullptr ? d_labels_src : nullptr, n_src, d_pose,
                                        d_flow, base + oWs2, ws2, stream, opt))
        return r;
    g_frameStamp[7] = now_us();   // everything enqueued
    if (hipMemcpyAsync(hBest, dBest, 4 * (2 * (size_t)S + 2), hipMemcpyDeviceToHost, s) != hipSuccess ||
        wait_stream(s) != hipSuccess)
        return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: the read-back of the matches failed");
    g_frameStamp[8] = now_us();   // matches on the host
    const int P0 = hBest[2 * S];
    if (P0 < 0) {
        *h_pairs = ICPFLOW_FRAME_ABANDONED;   // a team's wait timed out: the transforms are NaN (include/icpflow_hip.h, a-5)
        return 0;
    }
    // A pair that had to stay out of the superset and whose clusters both found no partner in stage 1 is a candidate of the
    // reference's stage 2: then stage 2 is registered again the reference's way -- its exact candidates (every unmatched source
    // against every unmatched destination through the sanity grid, utils_match.py:42-53), over-long clusters subsampled with the
    // next draws of the same generator -- on top of stage 1's results, which stand.  Same bits as the host-side association.
    bool exact = false;
    if (!leftS.empty()) {
        std::vector<uint8_t> mD(D, 0);
        for (int a = 0; a < S; ++a)
            if (hBest[a] >= 0) mD[di1[hBest[a]]] = 1;
        for (size_t k = 0; k < leftS.size() && !exact; ++k) exact = hBest[leftS[k]] < 0 && !mD[leftD[k]];
    }
    int P = P0;
    if (exact) {
        std::vector<uint8_t> mS(S, 0), mD(D, 0);
        size_t matched = 0;
        for (int a = 0; a < S; ++a)
            if (hBest[a] >= 0) { mS[a] = 1; mD[di1[hBest[a]]] = 1; ++matched; }
        std::vector<int32_t> si3, di3;
        int64_t longest3 = 0;
        int nPerm3 = 0;
        if (matched < unq.size()) {                                                                    // :42
            for (int a = 0; a < S; ++a) {
                if (mS[a] || !sOk[a]) continue;
                rowsOf.for_each(st, dt, dOk, a, tf, tb, [&](int b) {
                    if (mD[b]) return;
                    si3.push_back(a);
                    di3.push_back(b);
                    longest3 = std::max(longest3, std::max(st.count[a], dt.count[b]));
                    nPerm3 += (st.count[a] > maxPoints) + (dt.count[b] > maxPoints);
                });
            }
        }
        const int K3 = (int)si3.size();
        const int N3 = par->tight_padding ? std::min(maxPoints, std::max(64, round64(longest3))) : maxPoints;
        const size_t ws3 = K3 ? icpflow_workspace_bytes(K3, N3, reg->len_x, reg->len_y, reg->len_z) : 0;
        const size_t oClouds3 = off; off += up(sizeof(float) * 8 * (size_t)K3 * N3);
        const size_t oRes3 = off; off += up(sizeof(float) * (30 * (size_t)K3 + 1));
        const size_t oWs3 = off; off += up(ws3);
        *scratch_needed = off;
        if (scratch_bytes < off) return report_error(ICPFLOW_E_WORKSPACE, "icpflow_track_frame: scratch too small (see *scratch_needed)");
        const size_t qPerm = 48 * (size_t)K3, qIdx = up(qPerm + 4 * (size_t)nPerm3 * maxPoints), qBest = up(qIdx + 8 * (size_t)K3);
        char *pin3 = H.second.need(qBest + 4 * (2 * (size_t)S + 2));
        if (pin3 == nullptr) return report_error(ICPFLOW_E_HOSTMEM, "icpflow_track_frame: no pinned host memory");
        int64_t *seg3 = reinterpret_cast<int64_t *>(pin3);
        int32_t *perm3 = reinterpret_cast<int32_t *>(pin3 + qPerm), *idx3 = reinterpret_cast<int32_t *>(pin3 + qIdx);
        int32_t *hBest3 = reinterpret_cast<int32_t *>(pin3 + qBest);
        int drawn = 0;
        for (int k = 0; k < K3; ++k) {
            const int64_t cs = st.count[si3[k]], cd = dt.count[di3[k]];
            seg3[0 * K3 + k] = st.start[si3[k]]; seg3[1 * K3 + k] = std::min<int64_t>(cs, maxPoints); seg3[2 * K3 + k] = -1;
            seg3[3 * K3 + k] = dt.start[di3[k]]; seg3[4 * K3 + k] = std::min<int64_t>(cd, maxPoints); seg3[5 * K3 + k] = -1;
            if (cs > maxPoints) {
                seg3[2 * K3 + k] = (int64_t)drawn * maxPoints;
                randperm_head(gen, cs, maxPoints, perm3 + (size_t)drawn * maxPoints, H.perm);
                ++drawn;
            }
            if (cd > maxPoints) {
                seg3[5 * K3 + k] = (int64_t)drawn * maxPoints;
                randperm_head(gen, cd, maxPoints, perm3 + (size_t)drawn * maxPoints, H.perm);
                ++drawn;
            }
        }
        if (K3) {
            std::memcpy(idx3, si3.data(), 4 * (size_t)K3);
            std::memcpy(idx3 + K3, di3.data(), 4 * (size_t)K3);
        }
        icpflow_stage_t stage3{seg3, nPerm3 ? perm3 : nullptr, idx3, idx3 + K3, reinterpret_cast<float *>(base + oClouds3),
                               reinterpret_cast<float *>(base + oRes3), K3, N3};
        if (K3) {
            if (int r = icpflow_register_stage(&tables, &stage3, reg, base + oWs3, ws3, stream, opt)) return r;
            if (int r = icpflow_assoc_assign(stage3.d_result, stage3.d_si, stage3.d_di, K3, nullptr, S, D, par->translation_frame,
                                             par->thres_iou, par->rot_limit_deg, par->thres_error, dBest + S, 0, nullptr, nullptr,
                                             nullptr, nullptr, stream))
                return r;
        }
        if (int r = icpflow_assoc_collect(dBest, stage1.d_result, stage1.d_si, stage1.d_di, K1, K3 ? dBest + S : nullptr,
                                          K3 ? stage3.d_result : nullptr, K3 ? stage3.d_si : nullptr, K3 ? stage3.d_di : nullptr, K3,
                                          tables.d_table_src, tables.d_table_dst, tables.label_stride, S, cap, d_rows, d_T,
                                          dBest + 2 * (size_t)S, stream))
            return r;
        if (d_flow != nullptr)
            if (int r = icpflow_flow_rigid_rows(d_flow_points, d_labels_src, n_src, d_rows, 10, d_T, cap, d_pose, d_flow, stream)) return r;
        if (hipMemcpyAsync(hBest3, dBest, 4 * (2 * (size_t)S + 2), hipMemcpyDeviceToHost, s) != hipSuccess ||
            wait_stream(s) != hipSuccess)
            return report_error(ICPFLOW_E_ARG, "icpflow_track_frame: the read-back of the matches failed");
        P = hBest3[2 * S];
        if (P < 0) {
            *h_pairs = ICPFLOW_FRAME_ABANDONED;
            return 0;
        }
    }
    if (par->generator != nullptr) { gen.ensure(gen.pos); gen.state().save(par->generator); }
    *h_pairs = P;
    return 0;
}

// torch.randperm(n, generator)[0:take] on a generator in the state `gen` (advanced): the restatement above, for the CPU tests
extern "C" int icpflow_selftest_randperm(icpflow_mt19937_t *gen, int64_t n, int take, int32_t *h_out)
{
    if (!gen || !h_out || n <= 0 || take < 0 || take > n) return report_error(ICPFLOW_E_ARG, "icpflow_selftest_randperm: bad argument");
    static thread_local std::vector<uint32_t> buffer;
    MtStream g(Mt19937(*gen), buffer);
    static thread_local std::vector<int32_t> tmp;
    randperm_head(g, n, take, h_out, tmp);
    g.ensure(g.pos);
    g.state().save(gen);
    return 0;
}

// developer tool: the host's time stamps (microseconds, steady clock) inside the calling thread's last icpflow_track_frame
extern "C" int icpflow_debug_frame_stamps(double *out9)
{
    if (!out9) return ICPFLOW_E_ARG;
    for (int k = 0; k < 9; ++k) out9[k] = g_frameStamp[k];
    return 0;
}
// ... and the stamps inside "segments + draws": [0] its start, [1] superset done, [2] workspace sizes done, [3] stage 1's segments and draws done, [4] its end
extern "C" int icpflow_debug_frame_substamps(double *out5)
{
    if (!out5) return ICPFLOW_E_ARG;
    out5[0] = g_frameStamp[4]; out5[1] = g_frameStamp[9]; out5[2] = g_frameStamp[10]; out5[3] = g_frameStamp[11]; out5[4] = g_frameStamp[5];
    return 0;
}
