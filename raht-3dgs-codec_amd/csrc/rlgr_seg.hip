// rlgr_seg.hip -- the RLGR entropy stage ON THE GPU, segmented (second half: "GPU-segmented").
//
// The reference's coder (python/PyRLGR/src/libs/rlgr/membuf.cpp:258-423, parameters membuf.h:18-22) is one sequential
// adaptive stream per channel: ~3 M symbols that each depend on the state all earlier ones left. csrc/rlgr.hip runs the 56
// channels of a frame on the host threads, byte-exact, and that is what bounds a frame end to end (round 3: 65 ms per pass
// on 16 CPUs next to 0.6 ms of transforms, plus 12 ms of PCIe each way for the raw integers). Here every channel is cut
// into SEGMENTS of `seg_len` symbols and every segment is its own RLGR stream -- the coder's state starts afresh (k_P = 0,
// k_RP = 2 L), the stream is padded to a byte boundary and, in the container, to a 4-byte boundary. One lane codes one
// segment; 3 M x 56 symbols at 4096 per segment are 41 k independent streams = 656 waves.
//
// What "parity" means here: segment (c, s) is BYTE-IDENTICAL to what the reference's membuf::rlgrWrite produces for the
// slice Q[c, s * seg_len : (s + 1) * seg_len] -- tests compare every segment with the host coder of rlgr.hip (itself pinned
// byte for byte by reference-built streams, tests/test_rlgr.py), so any RLGR decoder reads a segment. The CONTAINER (sizes
// table + concatenated segments) is this repo's: the reference has no segmented format. Cost of the restarts: the coder
// re-adapts within a few dozen symbols; tests / DESIGN.md quote the measured size difference.
#include "raht_common.h"

#include <cstdlib>

namespace raht {
namespace rlgr_seg {

constexpr uint32_t L = 4, U0 = 3, D0 = 1, U1 = 2, D1 = 1;

// MSB-first bit writer, same byte stream as membuf::write / flush (and as BitWriter of rlgr.hip): < 32 bits pending after
// every put, whole 32-bit words leave big-endian. WRITE = false only counts. `out` is 4-byte aligned.
template <bool WRITE>
struct DevBitWriter {
    uint32_t *out32;
    uint32_t size = 0;       // bytes (keeps counting past cap: the exact length is known either way)
    uint32_t cap = 0xffffffffu;   // bytes that may be written at out32 (a multiple of 4); past it nothing is stored
    uint64_t acc = 0;
    int nbits = 0;

    __device__ __forceinline__ void put(uint64_t v, int bits)      // bits <= 32, v < 2^bits
    {
        acc = (acc << bits) | v;
        nbits += bits;
        if (nbits >= 32) {
            nbits -= 32;
            if (WRITE && size + 4 <= cap) out32[size >> 2] = __builtin_bswap32((uint32_t)(acc >> nbits));
            size += 4;
        }
    }
    __device__ __forceinline__ void put_wide(uint32_t v, int bits)   // the run length m in k bits (k <= 32 inside a segment)
    {
        if (bits > 32) { put(0, bits - 32); put(v, 32); }
        else put(bits == 32 ? v : (bits ? (v & ((1u << bits) - 1u)) : 0u), bits);
    }
    __device__ __forceinline__ void golomb_rice(uint32_t u, int k)  // membuf.cpp:242-256; k <= 32
    {
        const uint32_t p = (k < 32) ? (u >> k) : 0u;
        const uint32_t rem = (k >= 32) ? u : (u & ((1u << k) - 1u));
        if (p < 32) {
            const uint32_t pre = (uint32_t)((1ull << (p + 1)) - 2);               // p ones and a zero
            if (p + 1 + (uint32_t)k <= 32) put(((uint64_t)pre << k) | rem, (int)p + 1 + k);
            else { put(pre, (int)p + 1); put(rem, k); }
        } else {
            put(0xffffffffull, 32);                                               // escape: 32 ones, then 32 raw bits
            put(u, 32);
        }
    }
    __device__ __forceinline__ void close()                         // membuf.cpp:47-58
    {
        if (nbits & 7) put(0, 8 - (nbits & 7));
        if (nbits > 0) {                                            // 1..3 whole bytes left: the last, partial word (zero filled)
            const uint32_t word = (uint32_t)(acc << (32 - nbits));
            if (WRITE && size + 4 <= cap) out32[size >> 2] = __builtin_bswap32(word);
            size += (uint32_t)(nbits >> 3);
            nbits = 0;
        }
    }
};

__device__ __forceinline__ uint64_t s2u(int64_t v) { return v < 0 ? (((uint64_t)(-v)) << 1) - 1 : ((uint64_t)v) << 1; }
__device__ __forceinline__ int64_t u2s(uint64_t v) { const int64_t d = (int64_t)(v >> 1); return (v & 1) ? -d - 1 : d; }

#define RLGS_ADAPT_KRP(p)                                            \
    do {                                                             \
        if (p) { k_RP += (p) - 1; if (k_RP > 32 * L) k_RP = 32 * L; } \
        else { k_RP = (k_RP < 2) ? 0 : k_RP - 2; }                   \
    } while (0)

// (p - 1 may not fit 32 bits together with k_RP: saturate first)
#define RLGS_ADAPT_KRP32(p)                                                              \
    do {                                                                                 \
        if (p) { k_RP = ((p) > 32 * L) ? 32 * L : min(k_RP + (p) - 1, 32 * L); }         \
        else { k_RP = (k_RP < 2) ? 0 : k_RP - 2; }                                       \
    } while (0)

// membuf.cpp:340-423 on one segment
// VEC: the segment starts on a 16-byte boundary -- symbols are fetched four at a time (a lane walks its own segment, so a
// wave's loads touch 64 different lines either way: 16-byte loads make it a quarter as many instructions and lookups)
// What bounds this coder is instruction issue, one wave per SIMD at most: ~170 instructions per symbol (both modes' paths
// are walked by a wave whose lanes disagree), 2.6 waves per SIMD at 1024 symbols per segment, 0.64 at 4096 -- the time per pass is
// that of ONE wave walking its segments: proportional to seg_len above ~1500. Staging the symbols / the output through LDS in
// blocks of 64 (to take the loads and stores off each other's wait counter) changed nothing for the encoder and made the
// decoder slower (round 3): memory is not what a lane waits for.
template <bool WRITE, bool VEC>
__device__ __forceinline__ uint32_t encode_segment(const int32_t *__restrict__ seq, int n, int flag_signed, uint32_t *out32, uint32_t cap = 0xffffffffu,
                                                   int64_t sstr = 1)     // sstr: distance between consecutive symbols (VEC: 1)
{
    DevBitWriter<WRITE> w;
    w.out32 = out32;
    w.cap = cap;
    // 32-bit state: u = s2u(int32) < 2^32; inside a segment of n < 2^31 symbols the run counter m and the run exponent
    // k (<= log2 n + 1) stay far below 32 bits; k_RP is capped at 32 L. (The host coder carries them in 64 bits because one
    // stream may hold 2^32 symbols and more; 64-bit integer arithmetic is several instructions per operation here.)
    uint32_t u = 0, k_P = 0, k_RP = 2 * L, m = 0, k = 0;
    int32_t nxt = (!VEC && n > 0) ? seq[0] : 0;
    int4 cur4 = make_int4(0, 0, 0, 0), nxt4 = make_int4(0, 0, 0, 0);
    if (VEC && n > 0) nxt4 = *(const int4 *)seq;                     // (whole groups of four are readable: see the kernel)
    for (int i = 0; i < n; ++i) {
        int32_t v;
        if (VEC) {
            if ((i & 3) == 0) { cur4 = nxt4; if (i + 4 < n) nxt4 = *(const int4 *)(seq + i + 4); }    // one group ahead
            const int q = i & 3;
            v = q == 0 ? cur4.x : q == 1 ? cur4.y : q == 2 ? cur4.z : cur4.w;
        } else {
            v = nxt;
            if (i + 1 < n) nxt = seq[(int64_t)(i + 1) * sstr];       // one symbol ahead: the load is off the dependent chain
        }
        u = flag_signed ? (v < 0 ? ((uint32_t)(-(int64_t)v) << 1) - 1u : (uint32_t)v << 1) : (uint32_t)v;     // _s2u, membuf.cpp:4-13
        k = k_P / L;
        const uint32_t k_R = k_RP / L;
        if (k) {                                                    // run mode
            if (u) {
                --u;
                w.put(0, 1);
                w.put_wide(m, (int)k);
                w.golomb_rice(u, (int)k_R);
                const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;      // (k_R reaches 32 after an escape: a 32-bit shift by 32 is not 0)
                RLGS_ADAPT_KRP32(p);
                k_P = (k_P < D1) ? 0 : k_P - D1;
                m = 0;
            } else if (++m == ((k < 32) ? (1u << k) : 0u)) {         // (k >= 32 needs 2^32 zeros in one segment: never)
                w.put(1, 1);
                k_P += U1;
                m = 0;
            }
        } else {                                                    // no-run mode
            w.golomb_rice(u, (int)k_R);
            const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;      // (k_R reaches 32 after an escape: a 32-bit shift by 32 is not 0)
            RLGS_ADAPT_KRP32(p);
            if (u) k_P = (k_P < D0) ? 0 : k_P - D0;
            else k_P += U0;
            m = 0;
        }
    }
    if (n > 0 && k && !u) {                                         // membuf.cpp:410-413: flush the open run
        w.put(0, 1);
        w.put_wide(m, (int)(k_P / L));
    }
    w.close();
    return w.size;
}

// MSB-first bit reader over a 4-byte aligned segment of `size` bytes; bits past the end read as zeros
struct DevBitReader {
    const uint32_t *in32;
    uint32_t size, pos = 0;      // bytes; pos is a multiple of 4 (whole words are consumed, the last one zero-extended)
    uint64_t acc = 0;
    int nbits = 0;

    __device__ __forceinline__ uint32_t word(uint32_t w) { return in32[w]; }
    __device__ __forceinline__ void fill()                           // afterwards nbits > 32 unless the stream ended
    {
        if (nbits <= 32 && pos < size) {
            uint32_t w = __builtin_bswap32(word(pos >> 2));
            const uint32_t left = size - pos;
            int got = 32;
            if (left < 4) { got = (int)left * 8; w >>= (32 - got); }   // the last, partial word: only its bytes count
            acc = (acc << got) | w;
            nbits += got;
            pos += 4;
        }
    }
    __device__ __forceinline__ uint32_t bit()
    {
        if (!nbits) { fill(); if (!nbits) return 0; }
        --nbits;
        return (uint32_t)((acc >> nbits) & 1u);
    }
    __device__ __forceinline__ uint64_t get(int bits)                 // bits <= 32
    {
        if (!bits) return 0;
        if (nbits < bits) fill();
        if (nbits < bits) { const int miss = bits - nbits; acc <<= miss; nbits += miss; }   // zero padding
        nbits -= bits;
        return (acc >> nbits) & ((1ull << bits) - 1);
    }
    __device__ __forceinline__ uint64_t get_wide(int bits)
    {
        if (bits > 32) { const uint64_t hi = get(bits - 32) << 32; return hi + get(32); }
        return get(bits);
    }
    __device__ __forceinline__ uint64_t golomb_rice(int k)            // membuf.cpp:228-240
    {
        if (nbits < 33) fill();
        uint64_t p;
        if (nbits >= 33) {
            const uint64_t top = acc << (64 - nbits);
            p = (uint64_t)__builtin_clzll(~top | (1ull << 30));      // leading ones, at most 33 counted
            if (p >= 32) { nbits -= 32; return get(32); }
            nbits -= (int)p + 1;
        } else {
            p = 0;
            while (bit()) { if (++p >= 32) return get(32); }
        }
        return (p << k) + get(k);
    }
};

// VEC: the segment starts on a 16-byte boundary: symbols leave four at a time
template <bool VEC>
__device__ __forceinline__ void decode_segment(const uint32_t *in32, uint32_t nbytes, int n, int flag_signed, int32_t *__restrict__ seq, int64_t sstr = 1)
{
    DevBitReader r;
    r.in32 = in32; r.size = nbytes;
    // 32-bit state, as in the encoder: inside a segment of n < 2^31 symbols of int32 data the symbol values (< 2^32), the run
    // length m and the exponents stay within 32 bits; a corrupt stream may overflow them -- it then decodes to different garbage
    // than a 64-bit decoder would, inside the same bounds (every store is at i < n, every read below `size`).
    uint32_t k_P = 0, k_RP = 2 * L;
    int i = 0;
    int4 buf = make_int4(0, 0, 0, 0);
    auto emit = [&](int32_t v) {
        if (VEC) {
            const int q = i & 3;
            if (q == 0) buf.x = v; else if (q == 1) buf.y = v; else if (q == 2) buf.z = v; else buf.w = v;
            ++i;
            if ((i & 3) == 0) *(int4 *)(seq + i - 4) = buf;
        } else {
            seq[(int64_t)i * sstr] = v;
            ++i;
        }
    };
    while (i < n) {                                                  // membuf.cpp:270-331
        uint32_t k = k_P / L;
        const uint32_t k_R = k_RP / L;
        if (k) {
            uint32_t m = 0;
            while (r.bit()) {
                if (k >= 31) { m = (uint32_t)n; break; }             // corrupt stream guard (a run of 2^31 zeros in one segment)
                m += 1u << k;
                k_P += U1;
                k = k_P / L;
                if (m > (uint32_t)n) break;                          // corrupt stream guard
            }
            m += (uint32_t)r.get_wide((int)min(k, 32u));
            while (m-- && i < n) emit(0);
            if (i >= n) break;
            const uint32_t u = (uint32_t)r.golomb_rice((int)k_R);
            const uint32_t u1 = u + 1u;
            emit(flag_signed ? ((u1 & 1u) ? -(int32_t)(u1 >> 1) - 1 : (int32_t)(u1 >> 1)) : (int32_t)u1);
            const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;
            RLGS_ADAPT_KRP32(p);
            k_P = (k_P < D1) ? 0 : k_P - D1;
        } else {
            const uint32_t u = (uint32_t)r.golomb_rice((int)k_R);
            emit(flag_signed ? ((u & 1u) ? -(int32_t)(u >> 1) - 1 : (int32_t)(u >> 1)) : (int32_t)u);
            const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;
            RLGS_ADAPT_KRP32(p);
            if (u) k_P = (k_P < D0) ? 0 : k_P - D0;
            else k_P += U0;
        }
    }
    if (VEC && (i & 3)) {                                            // the last, partial group (n not a multiple of four)
        const int b = i & ~3;
        seq[b] = buf.x;
        if ((i & 3) > 1) seq[b + 1] = buf.y;
        if ((i & 3) > 2) seq[b + 2] = buf.z;
    }
}

// segment g = c * nseg + s  <->  symbols [s * S, min(N, (s + 1) * S)) of channel c
template <bool WRITE>
__global__ __launch_bounds__(64) void seg_encode_kernel(const int32_t *__restrict__ Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int S, int nseg,
                                                        int flag_signed, uint32_t *__restrict__ seg_bytes, const uint32_t *__restrict__ seg_off,
                                                        uint8_t *__restrict__ out, uint64_t cap, uint32_t *__restrict__ overflow)
{
    // thread t -> segment g = c * nseg + s. Channel-major input (sym_stride == 1): t = g, a lane walks its own contiguous run.
    // Row-major input (the quantized coefficients as the transform kernels leave them: symbol n of channel c at Q[n * ld + c]):
    // t = s * D + c -- the lanes of a wave are NEIGHBOURING CHANNELS at the same position of their segments, so every step of
    // the wave reads (writes) one contiguous piece of a row: no transpose in front of (behind) the coder.
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (int64_t)D * nseg) return;
    int c, s;
    if (sym_stride == 1) { c = (int)(t / nseg); s = (int)(t - (int64_t)c * nseg); }
    else { s = (int)(t / D); c = (int)(t - (int64_t)s * D); }
    const int64_t g = (int64_t)c * nseg + s;
    const int64_t i0 = (int64_t)s * S;
    const int n = (int)min((int64_t)S, N - i0);
    const int32_t *seq = Q + (int64_t)c * chan_stride + i0 * sym_stride;
    // 16-byte loads: segment starts aligned and a whole group of four readable behind the last symbol (wave-uniform choice)
    const bool vec = sym_stride == 1 && ((((uintptr_t)Q) & 15) == 0) && ((chan_stride & 3) == 0) && ((S & 3) == 0) && ((N & 3) == 0 || chan_stride >= ((N + 3) & ~(int64_t)3));
    if (!WRITE) {
        seg_bytes[g] = vec ? encode_segment<false, true>(seq, n, flag_signed, nullptr) : encode_segment<false, false>(seq, n, flag_signed, nullptr, 0xffffffffu, sym_stride);
    } else {
        const uint64_t off = seg_off[g];                              // 4-byte aligned
        const uint32_t need = (seg_bytes[g] + 3u) & ~3u;
        if (off + need > cap) { atomicOr(overflow, 1u); return; }
        if (vec) (void)encode_segment<true, true>(seq, n, flag_signed, (uint32_t *)(out + off));
        else (void)encode_segment<true, false>(seq, n, flag_signed, (uint32_t *)(out + off), 0xffffffffu, sym_stride);
    }
}

// ONE encoding pass: every segment into a fixed slot of `slot` bytes (what the raw integers would take, + 16) of a scratch
// buffer, its exact length recorded; seg_compact_kernel then moves the streams to their places in the container. A segment that
// does not fit its slot (incompressible data) raises *overflow and the caller falls back to the two exact passes.
__global__ __launch_bounds__(64) void seg_encode_slots_kernel(const int32_t *__restrict__ Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int S, int nseg,
                                                              int flag_signed, uint32_t *__restrict__ seg_bytes, uint8_t *__restrict__ slots, uint32_t slot,
                                                              uint32_t *__restrict__ overflow)
{
    // thread t -> segment g = c * nseg + s. Channel-major input (sym_stride == 1): t = g, a lane walks its own contiguous run.
    // Row-major input (the quantized coefficients as the transform kernels leave them: symbol n of channel c at Q[n * ld + c]):
    // t = s * D + c -- the lanes of a wave are NEIGHBOURING CHANNELS at the same position of their segments, so every step of
    // the wave reads (writes) one contiguous piece of a row: no transpose in front of (behind) the coder.
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (int64_t)D * nseg) return;
    int c, s;
    if (sym_stride == 1) { c = (int)(t / nseg); s = (int)(t - (int64_t)c * nseg); }
    else { s = (int)(t / D); c = (int)(t - (int64_t)s * D); }
    const int64_t g = (int64_t)c * nseg + s;
    const int64_t i0 = (int64_t)s * S;
    const int n = (int)min((int64_t)S, N - i0);
    const int32_t *seq = Q + (int64_t)c * chan_stride + i0 * sym_stride;
    const bool vec = sym_stride == 1 && ((((uintptr_t)Q) & 15) == 0) && ((chan_stride & 3) == 0) && ((S & 3) == 0);
    uint32_t *o = (uint32_t *)(slots + (size_t)g * slot);
    const uint32_t nb = vec ? encode_segment<true, true>(seq, n, flag_signed, o, slot) : encode_segment<true, false>(seq, n, flag_signed, o, slot, sym_stride);
    seg_bytes[g] = nb;
    if (((nb + 3u) & ~3u) > slot) atomicOr(overflow, 1u);
}

// one wave per segment: its words from the slot to its offset in the container
__global__ __launch_bounds__(256) void seg_compact_kernel(const uint8_t *__restrict__ slots, uint32_t slot, const uint32_t *__restrict__ seg_bytes,
                                                          const uint32_t *__restrict__ seg_off, int64_t G, uint8_t *__restrict__ out, uint64_t cap,
                                                          uint32_t *__restrict__ overflow)
{
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (g >= G) return;
    const uint32_t nw = (seg_bytes[g] + 3u) >> 2;
    const uint64_t off = seg_off[g];
    if (off + 4ull * nw > cap) { if (lane == 0) atomicOr(overflow, 2u); return; }
    const uint32_t *src = (const uint32_t *)(slots + (size_t)g * slot);
    uint32_t *dst = (uint32_t *)(out + off);
    for (uint32_t i = lane; i < nw; i += 64) dst[i] = src[i];
}

// padded size of every segment (its slot in the container): the input of the offset scan
__global__ void seg_pad_kernel(const uint32_t *__restrict__ seg_bytes, int64_t n, uint32_t *__restrict__ padded)
{
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) padded[g] = (seg_bytes[g] + 3u) & ~3u;
}

__global__ __launch_bounds__(64) void seg_decode_kernel(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint32_t *__restrict__ seg_off,
                                                        const uint32_t *__restrict__ seg_bytes, int64_t N, int D, int S, int nseg, int flag_signed,
                                                        int32_t *__restrict__ Q, int64_t sym_stride, int64_t chan_stride, uint32_t *__restrict__ bad)
{
    // thread t -> segment g = c * nseg + s. Channel-major input (sym_stride == 1): t = g, a lane walks its own contiguous run.
    // Row-major input (the quantized coefficients as the transform kernels leave them: symbol n of channel c at Q[n * ld + c]):
    // t = s * D + c -- the lanes of a wave are NEIGHBOURING CHANNELS at the same position of their segments, so every step of
    // the wave reads (writes) one contiguous piece of a row: no transpose in front of (behind) the coder.
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (int64_t)D * nseg) return;
    int c, s;
    if (sym_stride == 1) { c = (int)(t / nseg); s = (int)(t - (int64_t)c * nseg); }
    else { s = (int)(t / D); c = (int)(t - (int64_t)s * D); }
    const int64_t g = (int64_t)c * nseg + s;
    const int64_t i0 = (int64_t)s * S;
    const int n = (int)min((int64_t)S, N - i0);
    // the tables come off the wire: a segment never reaches outside the buffer (its last word is read whole: 4-byte slots)
    const uint64_t off = seg_off[g];
    uint32_t nb = seg_bytes[g];
    if ((off & 3) || off > in_bytes || (uint64_t)((nb + 3u) & ~3u) > in_bytes - off) { nb = 0; if (bad) atomicOr(bad, 1u); }
    // (16-byte stores of four buffered symbols measured SLOWER than one 4-byte store per symbol -- 4.2 against 3.0 ms for 3 M x 56
    // at 2048 per segment: the component selects cost more instructions than the stores save; kept behind this switch)
    const bool vec = false && sym_stride == 1 && ((((uintptr_t)Q) & 15) == 0) && ((chan_stride & 3) == 0) && ((S & 3) == 0);
    if (vec) decode_segment<true>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0);
    else decode_segment<false>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0 * sym_stride, sym_stride);
}

}  // namespace rlgr_seg
}  // namespace raht

using namespace raht;

extern "C" {

/* Segmented RLGR on the device. Q: DEVICE int32, channel-major (symbol n of channel c at Q[c * chan_stride + n], what
 * raht_transpose_i32 produces); seg_len symbols per segment (>= 64), nseg = ceil(N / seg_len) segments per channel.
 * Outputs (DEVICE): seg_bytes[D * nseg] = exact byte length of every segment's stream, seg_off[D * nseg + 1] = its offset in
 * `out` (segments are laid out in order, each padded to 4 bytes; seg_off[last] = bytes used), out[cap] = the streams.
 * *total_bytes (HOST) = bytes used. RAHT_ERR_NOMEM when cap is too small (nothing useful in out; 4 N D + 64 D nseg bytes
 * always suffice for data a raw int32 dump would not beat, raht_rlgr_bound(seg_len) * D * nseg for anything). Synchronises. */
int raht_rlgr_seg_encode(const int32_t *Q, int64_t N, int D, int64_t chan_stride, int seg_len, int flag_signed, uint32_t *seg_bytes,
                         uint32_t *seg_off, uint8_t *out, int64_t cap, int64_t *total_bytes, raht_stream_t stream)
{
    if (chan_stride < N) { set_error("raht_rlgr_seg_encode: bad argument"); return RAHT_ERR_INVALID; }
    return raht_rlgr_seg_encode_strided(Q, N, D, 1, chan_stride, seg_len, flag_signed, seg_bytes, seg_off, out, cap, total_bytes, stream);
}

int raht_rlgr_seg_encode_strided(const int32_t *Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int seg_len, int flag_signed,
                                 uint32_t *seg_bytes, uint32_t *seg_off, uint8_t *out, int64_t cap, int64_t *total_bytes, raht_stream_t stream)
{
    if (!Q || !seg_bytes || !seg_off || !out || !total_bytes || N < 1 || D < 1 || sym_stride < 1 || chan_stride < 1 || seg_len < 64 || cap < 16 || ((uintptr_t)out & 3) ||
        !((sym_stride == 1 && chan_stride >= N) || (chan_stride == 1 && sym_stride >= D))) {
        set_error("raht_rlgr_seg_encode: bad argument (channel-major: sym_stride 1, chan_stride >= N; row-major: chan_stride 1, sym_stride >= D)");
        return RAHT_ERR_INVALID;
    }
    const int64_t nseg = ceil_div(N, seg_len), G = nseg * D;
    if (nseg >= ((int64_t)1 << 31) || G >= ((int64_t)1 << 31)) { set_error("raht_rlgr_seg_encode: too many segments"); return RAHT_ERR_INVALID; }
    // segment offsets and the container's total are 32-bit: refuse what could wrap them (worst case: every symbol escapes --
    // 8 bytes and a bit -- plus the 4-byte padding of every segment)
    if (raht_rlgr_bound(seg_len) * G + 4 * G >= ((int64_t)1 << 32)) {
        set_error("raht_rlgr_seg_encode: %lld x %d symbols may need a container of 4 GiB or more (32-bit segment offsets): split the frame", (long long)N, D);
        return RAHT_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    return guarded("raht_rlgr_seg_encode", [&]() -> int {
        Scratch tmp(sizeof(uint32_t) * ((size_t)G + 2), s);
        if (!tmp.ok()) return RAHT_ERR_NOMEM;
        uint32_t *padded = tmp.as<uint32_t>(), *flags = padded + G;          // flags[0] = overflow, flags[1] = total
        RAHT_HIP_CHECK(hipMemsetAsync(flags, 0, 8, s));
        const unsigned gb = (unsigned)ceil_div(G, 64);
        // ONE encoding pass into fixed slots + a compaction, when a scratch buffer of the raw size is to be had; the two exact
        // passes (sizes, then streams) otherwise, and whenever a segment outgrows its slot
        static const bool two_pass_only = getenv("RAHT_RLGR_TWO_PASS") != nullptr;      // A/B knob
        const uint32_t slot = 4u * (uint32_t)seg_len + 16u;
        if (!two_pass_only && (uint64_t)G * slot < ((uint64_t)1 << 33)) {
            Scratch slots((size_t)G * slot, s);
            if (slots.ok()) {
                hipLaunchKernelGGL(rlgr_seg::seg_encode_slots_kernel, dim3(gb), dim3(64), 0, s, Q, N, D, sym_stride, chan_stride, seg_len, (int)nseg, flag_signed,
                                   seg_bytes, slots.as<uint8_t>(), slot, flags);
                hipLaunchKernelGGL(rlgr_seg::seg_pad_kernel, dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, s, seg_bytes, G, padded);
                RAHT_RET(exclusive_scan_u32(padded, seg_off, G, flags + 1, s));
                RAHT_HIP_CHECK(hipMemcpyAsync(seg_off + G, flags + 1, 4, hipMemcpyDeviceToDevice, s));
                hipLaunchKernelGGL(rlgr_seg::seg_compact_kernel, dim3((unsigned)ceil_div(G * 64, 256)), dim3(256), 0, s, slots.as<uint8_t>(), slot, seg_bytes,
                                   seg_off, G, out, (uint64_t)cap, flags);
                RAHT_HIP_CHECK(hipGetLastError());
                uint32_t back[2] = {0, 0};
                RAHT_RET(read_back_u32(back, flags, 2, nullptr, nullptr, 0, s));    // (synchronises: the scratch may go back to the pool)
                if (!(back[0] & 1u)) {
                    *total_bytes = (int64_t)back[1];
                    if ((back[0] & 2u) || (int64_t)back[1] > cap) { set_error("raht_rlgr_seg_encode: %u bytes needed, cap = %lld", back[1], (long long)cap); return RAHT_ERR_NOMEM; }
                    return RAHT_OK;
                }
                RAHT_HIP_CHECK(hipMemsetAsync(flags, 0, 8, s));                       // a segment outgrew its slot: the exact passes
            } else {
                (void)hipGetLastError();
            }
        }
        hipLaunchKernelGGL(rlgr_seg::seg_encode_kernel<false>, dim3(gb), dim3(64), 0, s, Q, N, D, sym_stride, chan_stride, seg_len, (int)nseg, flag_signed,
                           seg_bytes, (const uint32_t *)nullptr, (uint8_t *)nullptr, (uint64_t)0, flags);
        hipLaunchKernelGGL(rlgr_seg::seg_pad_kernel, dim3((unsigned)ceil_div(G, 256)), dim3(256), 0, s, seg_bytes, G, padded);
        RAHT_RET(exclusive_scan_u32(padded, seg_off, G, flags + 1, s));       // (32-bit offsets: containers below 4 GiB)
        RAHT_HIP_CHECK(hipMemcpyAsync(seg_off + G, flags + 1, 4, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(rlgr_seg::seg_encode_kernel<true>, dim3(gb), dim3(64), 0, s, Q, N, D, sym_stride, chan_stride, seg_len, (int)nseg, flag_signed,
                           seg_bytes, seg_off, out, (uint64_t)cap, flags);
        RAHT_HIP_CHECK(hipGetLastError());
        uint32_t back[2] = {0, 0};
        RAHT_RET(read_back_u32(back, flags, 2, nullptr, nullptr, 0, s));
        *total_bytes = (int64_t)back[1];
        if (back[0] || (int64_t)back[1] > cap) { set_error("raht_rlgr_seg_encode: %u bytes needed, cap = %lld", back[1], (long long)cap); return RAHT_ERR_NOMEM; }
        return RAHT_OK;
    });
}

/* The inverse: streams `in` (DEVICE, 4-byte aligned, in_bytes long -- a multiple of 4) with their offsets / lengths (DEVICE, as
 * raht_rlgr_seg_encode wrote them) -> Q (DEVICE, channel-major). Does not synchronise. The tables come off the wire: a
 * segment whose offset / length reaches outside `in` decodes as an empty stream (zeros) and sets *bad_dev (DEVICE uint32,
 * may be NULL) instead of reading there. */
int raht_rlgr_seg_decode(const uint8_t *in, int64_t in_bytes, const uint32_t *seg_off, const uint32_t *seg_bytes, int64_t N, int D, int seg_len,
                         int flag_signed, int32_t *Q, int64_t chan_stride, uint32_t *bad_dev, raht_stream_t stream)
{
    if (chan_stride < N) { set_error("raht_rlgr_seg_decode: bad argument"); return RAHT_ERR_INVALID; }
    return raht_rlgr_seg_decode_strided(in, in_bytes, seg_off, seg_bytes, N, D, seg_len, flag_signed, Q, 1, chan_stride, bad_dev, stream);
}

int raht_rlgr_seg_decode_strided(const uint8_t *in, int64_t in_bytes, const uint32_t *seg_off, const uint32_t *seg_bytes, int64_t N, int D, int seg_len,
                                 int flag_signed, int32_t *Q, int64_t sym_stride, int64_t chan_stride, uint32_t *bad_dev, raht_stream_t stream)
{
    if (!in || in_bytes < 0 || (in_bytes & 3) || !seg_off || !seg_bytes || !Q || N < 1 || D < 1 || seg_len < 64 || ((uintptr_t)in & 3) ||
        !((sym_stride == 1 && chan_stride >= N) || (chan_stride == 1 && sym_stride >= D))) {
        set_error("raht_rlgr_seg_decode: bad argument");
        return RAHT_ERR_INVALID;
    }
    const int64_t nseg = ceil_div(N, seg_len), G = nseg * D;
    if (G >= ((int64_t)1 << 31)) { set_error("raht_rlgr_seg_decode: too many segments"); return RAHT_ERR_INVALID; }
    hipLaunchKernelGGL(rlgr_seg::seg_decode_kernel, dim3((unsigned)ceil_div(G, 64)), dim3(64), 0, (hipStream_t)stream, in, (uint64_t)in_bytes, seg_off, seg_bytes, N, D,
                       seg_len, (int)nseg, flag_signed, Q, sym_stride, chan_stride, bad_dev);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

}  // extern "C"
