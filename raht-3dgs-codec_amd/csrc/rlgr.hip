// rlgr.hip -- host-side adaptive Run-Length / Golomb-Rice entropy coder (Malvar 2006), byte-exact
// with the reference's vendored PyRLGR (reference python/PyRLGR/src/libs/rlgr/membuf.cpp:258-423,
// parameters L=4, U0=3, D0=1, U1=2, D1=1 of membuf.h:18-22).  SURVEY.md 8f-1: the stage right after
// the RAHT hot path (reference python/encode_3dgs.py:219-245).
//
// What is different from the reference: it works directly on strided int32 coefficient columns (the
// reference goes tensor -> numpy -> Python list -> std::vector<int64_t> per channel,
// encode_3dgs.py:215-234), writes into caller buffers, and codes the D channels of a matrix on a
// pool of host threads (channels are independent streams). The bit stream is identical.
// Host code only (no kernels): RLGR is a sequential adaptive coder; the north star keeps the entropy
// stage on the CPU.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "raht_common.h"

namespace raht {
namespace rlgr {

constexpr uint64_t L = 4, U0 = 3, D0 = 1, U1 = 2, D1 = 1;

// MSB-first bit writer with a 64-bit accumulator (same byte stream as membuf::write/flush). Fewer than
// 32 bits stay pending after every put; whole 32-bit words leave as one big-endian store.
struct BitWriter {
    uint8_t *out;
    int64_t cap, size = 0;
    uint64_t acc = 0;        // pending bits, right-aligned (bits above nbits are stale)
    int nbits = 0;           // < 32 after every put
    bool overflow = false;

    inline void put(uint64_t v, int bits)                 // bits <= 32, v < 2^bits
    {
        acc = (acc << bits) | v;
        nbits += bits;
        if (nbits >= 32) {
            nbits -= 32;
            const uint32_t word = (uint32_t)(acc >> nbits);
            if (size + 4 <= cap) {
                const uint32_t be = __builtin_bswap32(word);
                memcpy(out + size, &be, 4);
            } else {
                for (int b = 0; b < 4; ++b) {
                    if (size + b < cap) out[size + b] = (uint8_t)(word >> (24 - 8 * b));
                    else overflow = true;
                }
            }
            size += 4;
        }
    }
    inline void put_wide(uint64_t v, int bits)            // membuf.cpp:172-184; any width up to 64
    {
        if (bits > 32) { put((bits == 64) ? (v >> 32) : ((v >> 32) & ((1ull << (bits - 32)) - 1)), bits - 32); put(v & 0xffffffffull, 32); }
        else put(bits ? (v & ((1ull << bits) - 1)) : 0, bits);
    }
    inline void golomb_rice(uint64_t u, int k)            // membuf.cpp:242-256 (k <= 32)
    {
        const uint64_t p = u >> k;
        if (p < 32) {
            if (p + 1 + (uint64_t)k <= 32) {              // prefix (p ones, one zero) and remainder in one go
                put((((1ull << (p + 1)) - 2) << k) | (k ? (u & ((1ull << k) - 1)) : 0), (int)p + 1 + k);
            } else {
                put((1ull << (p + 1)) - 2, (int)p + 1);
                put(k ? (u & ((1ull << k) - 1)) : 0, k);
            }
        } else {
            put(0xffffffffull, 32);                       // escape: 32 ones, then 32 raw bits
            put(u & 0xffffffffull, 32);
        }
    }
    inline void close()                                   // membuf.cpp:47-58: pad the last byte with zeros
    {
        if (nbits & 7) put(0, 8 - (nbits & 7));
        while (nbits > 0) {
            nbits -= 8;
            if (size < cap) out[size] = (uint8_t)(acc >> nbits);
            else overflow = true;
            ++size;
        }
    }
};

// MSB-first bit reader: the valid bits sit right-aligned in acc; bits past the end of the stream read
// as zeros (the reference underflows there).
struct BitReader {
    const uint8_t *in;
    int64_t size, pos = 0;
    uint64_t acc = 0;
    int nbits = 0;

    inline void fill()                                    // afterwards nbits > 56 unless the stream ended
    {
        if (nbits <= 32 && pos + 4 <= size) {
            uint32_t be;
            memcpy(&be, in + pos, 4);
            acc = (acc << 32) | __builtin_bswap32(be);
            pos += 4; nbits += 32;
        }
        while (nbits <= 56 && pos < size) { acc = (acc << 8) | in[pos++]; nbits += 8; }
    }
    inline uint32_t bit()
    {
        if (!nbits) { fill(); if (!nbits) return 0; }
        --nbits;
        return (uint32_t)((acc >> nbits) & 1u);
    }
    inline uint64_t get(int bits)                         // bits <= 56
    {
        if (!bits) return 0;
        if (nbits < bits) fill();
        if (nbits < bits) { const int miss = bits - nbits; acc <<= miss; nbits += miss; }   // zero padding
        nbits -= bits;
        return (acc >> nbits) & ((1ull << bits) - 1);
    }
    inline uint64_t get_wide(int bits)
    {
        if (bits > 56) { const uint64_t hi = get(bits - 32) << 32; return hi + get(32); }
        return get(bits);
    }
    inline uint64_t golomb_rice(int k)                    // membuf.cpp:228-240
    {
        // unary prefix by counting leading ones of the pending bits (at most 32 count)
        if (nbits < 33) fill();
        uint64_t p;
        if (nbits >= 33) {
            const uint64_t top = acc << (64 - nbits);     // pending bits, left-aligned
            p = (uint64_t)__builtin_clzll(~top | (1ull << 30));              // leading ones, at most 33 counted
            if (p >= 32) { nbits -= 32; return get(32); }
            nbits -= (int)p + 1;                          // the ones and the terminating zero
        } else {                                          // tail of the stream: bit by bit, zeros past the end
            p = 0;
            while (bit()) { if (++p >= 32) return get(32); }
        }
        return (p << k) + get(k);
    }
};

static inline uint64_t s2u(int64_t v) { return v < 0 ? (((uint64_t)(-v)) << 1) - 1 : ((uint64_t)v) << 1; }
static inline int64_t u2s(uint64_t v) { const int64_t d = (int64_t)(v >> 1); return (v & 1) ? -d - 1 : d; }

#define RLGR_ADAPT_KRP(p)                                            \
    do {                                                             \
        if (p) { k_RP += (p) - 1; if (k_RP > 32 * L) k_RP = 32 * L; } \
        else { k_RP = (k_RP < 2) ? 0 : k_RP - 2; }                   \
    } while (0)

static int64_t encode(const int32_t *seq, int64_t n, int64_t stride, int flag_signed, uint8_t *out, int64_t cap)
{
    BitWriter w;
    w.out = out; w.cap = cap;
    uint64_t u = 0, k_P = 0, k_RP = 2 * L, m = 0, k = 0;
    for (int64_t i = 0; i < n; ++i) {                     // membuf.cpp:351-408
        const int64_t v = seq[i * stride];
        u = flag_signed ? s2u(v) : (uint64_t)(uint32_t)v;
        k = k_P / L;
        const uint64_t k_R = k_RP / L;
        if (k) {                                          // run mode
            if (u) {
                --u;
                w.put(0, 1);
                w.put_wide(m, (int)k);
                w.golomb_rice(u, (int)k_R);
                const uint64_t p = u >> k_R;
                RLGR_ADAPT_KRP(p);
                k_P = (k_P < D1) ? 0 : k_P - D1;
                m = 0;
            } else if (++m == (1ull << k)) {
                w.put(1, 1);
                k_P += U1;
                m = 0;
            }
        } else {                                          // no-run mode
            w.golomb_rice(u, (int)k_R);
            const uint64_t p = u >> k_R;
            RLGR_ADAPT_KRP(p);
            if (u) k_P = (k_P < D0) ? 0 : k_P - D0;
            else k_P += U0;
            m = 0;
        }
    }
    if (n > 0 && k && !u) {                               // membuf.cpp:410-413: flush the open run
        w.put(0, 1);
        w.put_wide(m, (int)(k_P / L));
    }
    w.close();
    return w.overflow ? -1 : w.size;
}

static void decode(const uint8_t *buf, int64_t nbytes, int64_t n, int flag_signed, int32_t *seq, int64_t stride)
{
    BitReader r;
    r.in = buf; r.size = nbytes;
    uint64_t k_P = 0, k_RP = 2 * L;
    int64_t i = 0;
    while (i < n) {                                       // membuf.cpp:270-331
        uint64_t k = k_P / L;
        const uint64_t k_R = k_RP / L;
        if (k) {
            uint64_t m = 0;
            while (r.bit()) {
                m += 1ull << k;
                k_P += U1;
                k = k_P / L;
                if (m > (uint64_t)n) break;                // corrupt stream guard
            }
            m += r.get_wide((int)k);
            while (m-- && i < n) seq[(i++) * stride] = 0;
            if (i >= n) break;
            const uint64_t u = r.golomb_rice((int)k_R);
            seq[(i++) * stride] = (int32_t)(flag_signed ? u2s(u + 1) : (int64_t)(u + 1));
            const uint64_t p = u >> k_R;
            RLGR_ADAPT_KRP(p);
            k_P = (k_P < D1) ? 0 : k_P - D1;
        } else {
            const uint64_t u = r.golomb_rice((int)k_R);
            seq[(i++) * stride] = (int32_t)(flag_signed ? u2s(u) : (int64_t)u);
            const uint64_t p = u >> k_R;
            RLGR_ADAPT_KRP(p);
            if (u) k_P = (k_P < D0) ? 0 : k_P - D0;
            else k_P += U0;
        }
    }
}

// nthreads - 1 helpers + the calling thread run `work` (which pulls items until none are left). A helper that
// cannot be started (std::system_error) is simply absent: the others finish its share. Never leaves a joinable
// std::thread behind on the way out (that would be std::terminate).
template <typename W>
static void run_pool(int nthreads, W &work)
{
    std::vector<std::thread> pool;
    try {
        pool.reserve((size_t)std::max(nthreads - 1, 0));
        for (int t = 1; t < nthreads; ++t) pool.emplace_back([&]() { work(); });
    } catch (...) {
    }
    work();
    for (auto &t : pool) t.join();
}

// Channels are handed out most expensive first (cost[c]: any monotone proxy of the channel's coding time, or NULL): 56
// channels on 16 threads are 3.5 rounds, and a frame's channels differ by 10 x in bit rate -- with the long ones last, the
// pass ended on a few threads coding the xyz / DC-heavy channels alone.
template <typename F>
static void parallel_channels(int D, int nthreads, F fn, const double *cost = nullptr)
{
    nthreads = std::min(nthreads, D);
    if (nthreads <= 1) { for (int c = 0; c < D; ++c) fn(c); return; }
    std::vector<int> order((size_t)D);
    for (int c = 0; c < D; ++c) order[(size_t)c] = c;
    if (cost) std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    std::atomic<int> next(0);
    auto work = [&]() { for (int i = next++; i < D; i = next++) fn(order[(size_t)i]); };
    run_pool(nthreads, work);
}

// proxy for a channel's bit rate from ~2048 evenly spaced symbols: sum of 1 + log2(1 + |q|)
static double sampled_cost(const int32_t *q, int64_t N, int64_t stride)
{
    const int64_t step = std::max<int64_t>(1, N / 2048);
    double c = 0;
    for (int64_t i = 0; i < N; i += step) {
        const int64_t v = q[i * stride];
        const uint64_t a = (uint64_t)(v < 0 ? -v : v) + 1;
        c += 1 + (63 - __builtin_clzll(a));
    }
    return c;
}

}  // namespace rlgr
}  // namespace raht

using namespace raht;

extern "C" {

int64_t raht_rlgr_bound(int64_t n)
{
    // worst case per symbol: run prefix (1 + k <= 33 bits) + escape (64 bits) -> 13 bytes; + flush
    return 16 + 13 * (n > 0 ? n : 0);
}

int raht_rlgr_encode(const int32_t *seq, int64_t n, int64_t stride, int flag_signed, uint8_t *out, int64_t cap,
                     int64_t *nbytes)
{
    if ((!seq && n > 0) || !out || !nbytes || n < 0 || stride < 1) { set_error("raht_rlgr_encode: bad argument"); return RAHT_ERR_INVALID; }
    const int64_t r = rlgr::encode(seq, n, stride, flag_signed, out, cap);
    if (r < 0) { set_error("raht_rlgr_encode: output buffer too small (use raht_rlgr_bound)"); return RAHT_ERR_NOMEM; }
    *nbytes = r;
    return RAHT_OK;
}

int raht_rlgr_decode(const uint8_t *buf, int64_t nbytes, int64_t n, int flag_signed, int32_t *seq, int64_t stride)
{
    if ((!buf && nbytes > 0) || (!seq && n > 0) || n < 0 || nbytes < 0 || stride < 1) { set_error("raht_rlgr_decode: bad argument"); return RAHT_ERR_INVALID; }
    rlgr::decode(buf, nbytes, n, flag_signed, seq, stride);
    return RAHT_OK;
}

}  // extern "C" (helpers below have C++ linkage)
namespace raht { namespace rlgr {
// Blocked, threaded transpose of an N x D row-major int32 matrix into D contiguous channels (and
// back). Coding a strided column makes every thread stream the whole matrix through its caches.
static void transpose_to_channels(const int32_t *Q, int64_t N, int D, int64_t ldq, int32_t *T, int nthreads)
{
    const int64_t B = 2048;
    const int64_t nblk = (N + B - 1) / B;
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (int64_t b = next++; b < nblk; b = next++) {
            const int64_t r0 = b * B, r1 = std::min(N, r0 + B);
            for (int c0 = 0; c0 < D; c0 += 16)
                for (int64_t r = r0; r < r1; ++r) {
                    const int32_t *row = Q + r * ldq;
                    for (int c = c0; c < std::min(D, c0 + 16); ++c) T[(int64_t)c * N + r] = row[c];
                }
        }
    };
    raht::rlgr::run_pool(nthreads, work);
}

static void transpose_from_channels(const int32_t *T, int64_t N, int D, int32_t *Q, int64_t ldq, int nthreads)
{
    const int64_t B = 2048;
    const int64_t nblk = (N + B - 1) / B;
    std::atomic<int64_t> next(0);
    auto work = [&]() {
        for (int64_t b = next++; b < nblk; b = next++) {
            const int64_t r0 = b * B, r1 = std::min(N, r0 + B);
            for (int c0 = 0; c0 < D; c0 += 16)
                for (int64_t r = r0; r < r1; ++r) {
                    int32_t *row = Q + r * ldq;
                    for (int c = c0; c < std::min(D, c0 + 16); ++c) row[c] = T[(int64_t)c * N + r];
                }
        }
    };
    raht::rlgr::run_pool(nthreads, work);
}

// CPUs this process may actually use: the cgroup's CPU quota (containers: 16 CPUs on the one-GPU MI355X box although 256 are
// visible -- 56 coder threads there run in bursts and get throttled: 24 ... 105 ms per pass instead of a steady 65) and the
// affinity mask, not just the number of hardware threads.
static int usable_cpus()
{
    static int n = 0;
    if (n) return n;
    int v = (int)std::max(1u, std::thread::hardware_concurrency());
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {          // cgroup v2: "<quota> <period>" or "max <period>"
        char q[32] = "";
        long long period = 0;
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) {
            const long long quota = atoll(q);
            if (quota > 0) v = std::min<long long>(v, std::max<long long>(1, (quota + period - 1) / period));
        }
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {   // cgroup v1
        long long quota = -1, period = 0;
        if (fscanf(g, "%lld", &quota) == 1 && quota > 0) {
            if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
                if (fscanf(h, "%lld", &period) == 1 && period > 0) v = std::min<long long>(v, std::max<long long>(1, (quota + period - 1) / period));
                fclose(h);
            }
        }
        fclose(g);
    }
    n = std::max(1, v);
    return n;
}

static int resolve_threads(int nthreads, int D)
{
    if (nthreads <= 0) nthreads = usable_cpus();
    return std::max(1, std::min(nthreads, std::max(D, 1)));
}
}}  // namespace raht::rlgr
using raht::rlgr::resolve_threads;
extern "C" {

int raht_rlgr_encode_channels(const int32_t *Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride,
                              int flag_signed, uint8_t *out, int64_t cap_per_channel, int64_t *nbytes, int nthreads)
{
    if (!Q || !out || !nbytes || N < 0 || D < 1 || sym_stride < 1 || chan_stride < 1 || cap_per_channel < 1) {
        set_error("raht_rlgr_encode_channels: bad argument");
        return RAHT_ERR_INVALID;
    }
    // std::vector / std::thread may throw (708 MB of staging on cfg3): no exception crosses the C ABI
    return raht::guarded("raht_rlgr_encode_channels", [&]() -> int {
        const int nt = resolve_threads(nthreads, D);
        std::vector<int32_t> tmp;
        const int32_t *src = Q;
        int64_t ss = sym_stride, cs = chan_stride;
        if (sym_stride != 1 && chan_stride == 1 && D > 1 && N > 4096) {          // row-major: go channel-major first
            tmp.resize((size_t)N * (size_t)D);
            rlgr::transpose_to_channels(Q, N, D, sym_stride, tmp.data(), nt);
            src = tmp.data(); ss = 1; cs = N;
        }
        std::atomic<int> bad(0);
        std::vector<double> cost((size_t)D);
        for (int c = 0; c < D; ++c) cost[(size_t)c] = rlgr::sampled_cost(src + (int64_t)c * cs, N, ss);
        rlgr::parallel_channels(D, nt, [&](int c) {
            const int64_t r = rlgr::encode(src + (int64_t)c * cs, N, ss, flag_signed, out + (int64_t)c * cap_per_channel, cap_per_channel);
            if (r < 0) { bad = 1; nbytes[c] = -1; } else nbytes[c] = r;
        }, cost.data());
        if (bad) { set_error("raht_rlgr_encode_channels: cap_per_channel too small (use raht_rlgr_bound)"); return RAHT_ERR_NOMEM; }
        return RAHT_OK;
    });
}

int raht_rlgr_decode_channels(const uint8_t *bufs, int64_t cap_per_channel, const int64_t *nbytes, int64_t N, int D,
                              int flag_signed, int32_t *Q, int64_t sym_stride, int64_t chan_stride, int nthreads)
{
    if (!bufs || !nbytes || !Q || N < 0 || D < 1 || sym_stride < 1 || chan_stride < 1 || cap_per_channel < 0) { set_error("raht_rlgr_decode_channels: bad argument"); return RAHT_ERR_INVALID; }
    // the length table comes off the wire: a stream never extends past its channel's slot
    for (int c = 0; c < D; ++c)
        if (nbytes[c] < 0 || nbytes[c] > cap_per_channel) {
            set_error("raht_rlgr_decode_channels: nbytes[%d] = %lld outside [0, cap_per_channel = %lld]", c, (long long)nbytes[c], (long long)cap_per_channel);
            return RAHT_ERR_INVALID;
        }
    return raht::guarded("raht_rlgr_decode_channels", [&]() -> int {
        const int nt = resolve_threads(nthreads, D);
        std::vector<int32_t> tmp;
        int32_t *dst = Q;
        int64_t ss = sym_stride, cs = chan_stride;
        const bool via_tmp = (sym_stride != 1 && chan_stride == 1 && D > 1 && N > 4096);
        if (via_tmp) { tmp.resize((size_t)N * (size_t)D); dst = tmp.data(); ss = 1; cs = N; }
        std::vector<double> cost((size_t)D);
        for (int c = 0; c < D; ++c) cost[(size_t)c] = (double)nbytes[c];          // a channel's decoding time follows its bytes
        rlgr::parallel_channels(D, nt, [&](int c) {
            rlgr::decode(bufs + (int64_t)c * cap_per_channel, nbytes[c], N, flag_signed, dst + (int64_t)c * cs, ss);
        }, cost.data());
        if (via_tmp) rlgr::transpose_from_channels(tmp.data(), N, D, Q, sym_stride, nt);
        return RAHT_OK;
    });
}


/* Are two contiguous int32 arrays equal? (the drivers' round-trip assertion, python/encode_3dgs.py:242-245, on 10^8 symbols:
 * numpy's single-threaded array_equal costs more than coding them.) *first_diff = index of the first difference, or -1. */
int raht_i32_equal(const int32_t *a, const int32_t *b, int64_t n, int nthreads, int64_t *first_diff)
{
    if ((!a || !b) && n > 0) { set_error("raht_i32_equal: NULL argument"); return RAHT_ERR_INVALID; }
    if (n < 0 || !first_diff) { set_error("raht_i32_equal: bad argument"); return RAHT_ERR_INVALID; }
    return raht::guarded("raht_i32_equal", [&]() -> int {
        const int64_t B = (int64_t)1 << 20;
        const int64_t nblk = (n + B - 1) / B;
        const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(resolve_threads(nthreads, 1 << 20), nblk));
        std::atomic<int64_t> next(0), first(INT64_MAX);
        auto work = [&]() {
            for (int64_t blk = next++; blk < nblk; blk = next++) {
                const int64_t i0 = blk * B, i1 = std::min(n, i0 + B);
                if (i0 >= first.load(std::memory_order_relaxed)) continue;
                if (memcmp(a + i0, b + i0, (size_t)(i1 - i0) * 4) == 0) continue;
                for (int64_t i = i0; i < i1; ++i)
                    if (a[i] != b[i]) { int64_t cur = first.load(); while (i < cur && !first.compare_exchange_weak(cur, i)) {} break; }
            }
        };
        raht::rlgr::run_pool(nt, work);
        *first_diff = first.load() == INT64_MAX ? -1 : first.load();
        return RAHT_OK;
    });
}

}  // extern "C"
