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
// LDSOUT (out32 16-byte aligned): the words of a lane collect in a 16-word LDS column ([16][64] words per wave, as in the
// decoder below) and leave as one 64-byte piece -- with the steps of a frame coded together the one-word stores of 738 k lanes
// cost 8 x the streams' bytes in HBM writes (3.7 GB for 0.47 GB, rocprofv3 WRITE_SIZE).
template <bool WRITE, bool LDSOUT = false>
struct DevBitWriter {
    uint32_t *out32;
    uint32_t size = 0;       // bytes (keeps counting past cap: the exact length is known either way)
    uint32_t cap = 0xffffffffu;   // bytes that may be written at out32 (a multiple of 4); past it nothing is stored
    uint64_t acc = 0;
    int nbits = 0;
    uint32_t *col = nullptr; // LDSOUT: this lane's column

    __device__ __forceinline__ void store_word(uint32_t w)          // the word at byte offset `size`
    {
        if (!WRITE || size + 4 > cap) return;
        if (LDSOUT) {
            const uint32_t wi = size >> 2, q = wi & 15u;
            col[q * 64] = w;
            if (q == 15u) {
                uint4 x[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) x[t] = make_uint4(col[(4 * t) * 64], col[(4 * t + 1) * 64], col[(4 * t + 2) * 64], col[(4 * t + 3) * 64]);
#pragma unroll
                for (int t = 0; t < 4; ++t) *(uint4 *)(out32 + wi - 15 + 4 * t) = x[t];
            }
        } else {
            out32[size >> 2] = w;
        }
    }
    __device__ __forceinline__ void put(uint64_t v, int bits)      // bits <= 32, v < 2^bits
    {
        acc = (acc << bits) | v;
        nbits += bits;
        if (nbits >= 32) {
            nbits -= 32;
            store_word(__builtin_bswap32((uint32_t)(acc >> nbits)));
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
        uint32_t words = size >> 2;                                 // whole words so far
        if (nbits > 0) {                                            // 1..3 whole bytes left: the last, partial word (zero filled)
            const uint32_t word = (uint32_t)(acc << (32 - nbits));
            store_word(__builtin_bswap32(word));
            if (size + 4 <= cap) ++words;
            size += (uint32_t)(nbits >> 3);
            nbits = 0;
        }
        if (WRITE && LDSOUT) {                                      // the last, partial column (a full one left when its 16th word came)
            const uint32_t stored = min(words, cap >> 2);
            for (uint32_t wi = stored & ~15u; wi < stored; ++wi) out32[wi] = col[(wi & 15u) * 64];
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
template <bool WRITE, bool VEC, bool LDSOUT = false>
__device__ __forceinline__ uint32_t encode_segment(const int32_t *__restrict__ seq, int n, int flag_signed, uint32_t *out32, uint32_t cap = 0xffffffffu,
                                                   int64_t sstr = 1,     // sstr: distance between consecutive symbols (VEC: 1)
                                                   uint32_t *lds = nullptr)
{
    DevBitWriter<WRITE, LDSOUT> w;
    w.out32 = out32;
    w.cap = cap;
    w.col = lds + (threadIdx.x & 63);
    // 32-bit state: u = s2u(int32) < 2^32; inside a segment of n < 2^31 symbols the run counter m and the run exponent
    // k (<= log2 n + 1) stay far below 32 bits; k_RP is capped at 32 L. (The host coder carries them in 64 bits because one
    // stream may hold 2^32 symbols and more; 64-bit integer arithmetic is several instructions per operation here.)
    uint32_t u = 0, k_P = 0, k_RP = 2 * L, m = 0, k = 0;
    int32_t nxt = (!VEC && n > 0) ? seq[0] : 0;
    const int32_t *rp = seq;                                         // (walked by pointer: a 64-bit multiply per symbol otherwise)
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
            if (i + 1 < n) { rp += sstr; nxt = *rp; }                // one symbol ahead: the load is off the dependent chain
        }
        u = flag_signed ? (v < 0 ? ((uint32_t)(-(int64_t)v) << 1) - 1u : (uint32_t)v << 1) : (uint32_t)v;     // _s2u, membuf.cpp:4-13
        k = k_P / L;
        const uint32_t k_R = k_RP / L;
        // ONE Golomb-Rice site for both modes (the symbol that ends a run and a no-run symbol differ by what precedes the code and by
        // the k_P step): a wave whose lanes disagree about the mode walks the code path once, not twice
        bool code = true;
        const bool nz = u != 0;
        if (k) {                                                    // run mode
            if (nz) {
                --u;                                                // (in place, as membuf.cpp does: the open-run test behind the loop sees it)
                if (k < 32) w.put((uint64_t)(m & ((1u << k) - 1u)), (int)k + 1);      // a 0 bit, then the run length in k bits
                else { w.put(0, 1); w.put_wide(m, (int)k); }
            } else {
                code = false;
                if (++m == ((k < 32) ? (1u << k) : 0u)) {           // (k >= 32 needs 2^32 zeros in one segment: never)
                    w.put(1, 1);
                    k_P += U1;
                    m = 0;
                }
            }
        }
        if (code) {
            w.golomb_rice(u, (int)k_R);
            const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;        // (k_R reaches 32 after an escape: a 32-bit shift by 32 is not 0)
            RLGS_ADAPT_KRP32(p);
            if (k || nz) k_P = k_P ? k_P - 1u : 0u;                  // D0 = D1 = 1 (run mode: always; no-run mode: after a nonzero symbol)
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
// LDSIN: the stream's words come through an 8-word LDS column per lane ([8][64] words per wave), fetched as one aligned 32-byte
// piece (2 x 16 bytes) when the lane crosses into it -- a lane's 4-byte reads, far apart in time, found their line evicted
// between two of them once the steps of a frame decode together (FETCH_SIZE 12 x the streams' bytes). Pieces are aligned
// in memory, not to the segment: the first one may begin before the segment (inside the container: `lo`), the last one may
// reach past it (`hi`: words at or beyond it read as zero and are never consumed: fill() stops at `size`).
// MODE 2 (the symbol-synchronous decoder): a RING of 16 words per lane ([16][64] words per wave) that ALL lanes of the wave top up
// at the same iterations (top_up(), every 32 symbols): the wave then waits for stream words once per 32 iterations instead of
// whenever some lane crosses into a new piece -- with 64 lanes that was most iterations, and for a frame on its own (1.25 waves
// per SIMD, nothing else to run meanwhile) each such wait is a memory round trip on the wave's only chain. A lane that drains its
// ring before the next top-up (more than 16 bits per symbol over 32 symbols) reads its words one by one until then.
template <int MODE>
struct DevBitReaderT {
    static constexpr bool LDSIN = MODE == 1;
    static constexpr int RING = 16;
    uint32_t have = 0;           // MODE 2: words [0, have) have been fetched (the ring holds the last RING of them at most)
    const uint32_t *in32;
    uint32_t size, pos = 0;      // bytes; pos is a multiple of 4 (whole words are consumed, the last one zero-extended)
    uint64_t acc = 0;
    int nbits = 0;
    uint32_t *col = nullptr;     // LDSIN: this lane's column
    uint32_t w0 = 0;             // LDSIN: word index of the segment's start inside its first 32-byte piece
    int64_t hi_words = 0;        // LDSIN: words from the segment's start to the end of the container

    __device__ __forceinline__ void top_up()                         // MODE 2, called by every lane of the wave at the same time
    {
        const uint32_t next = pos >> 2, nwords = (size + 3u) >> 2;  // next word to consume; words of this stream
        if (have < next) have = next;                               // (the lane read ahead of its ring word by word)
#pragma unroll
        for (int t = 0; t < RING; ++t) {
            const uint32_t w = have + (uint32_t)t;
            if (w < next + RING && w < nwords) col[(w & (RING - 1)) * 64] = in32[w];
        }
        have = min(min(have + (uint32_t)RING, next + (uint32_t)RING), nwords);
    }
    __device__ __forceinline__ uint32_t word(uint32_t w)
    {
        if (MODE == 2) return (w < have) ? col[(w & (RING - 1)) * 64] : in32[w];
        if (!LDSIN) return in32[w];
        const uint32_t gw = w + w0, q = gw & 7u;
        if (q == 0 || w == 0) {                                      // into a new piece (or the very first word): fetch it
            const int64_t first = (int64_t)w - (int64_t)q;           // word index (from the segment's start) of the piece's first word; >= -7
            const uint4 *src = (const uint4 *)(in32 + first);
            uint4 a = make_uint4(0, 0, 0, 0), b = a;
            if (first + 8 <= hi_words) { a = src[0]; b = src[1]; }
            else {                                                   // the container ends inside this piece: word by word
                const uint32_t *sw = in32 + first;
                if (first + 0 < hi_words) a.x = sw[0];
                if (first + 1 < hi_words) a.y = sw[1];
                if (first + 2 < hi_words) a.z = sw[2];
                if (first + 3 < hi_words) a.w = sw[3];
                if (first + 4 < hi_words) b.x = sw[4];
                if (first + 5 < hi_words) b.y = sw[5];
                if (first + 6 < hi_words) b.z = sw[6];
                if (first + 7 < hi_words) b.w = sw[7];
            }
            col[0 * 64] = a.x; col[1 * 64] = a.y; col[2 * 64] = a.z; col[3 * 64] = a.w;
            col[4 * 64] = b.x; col[5 * 64] = b.y; col[6 * 64] = b.z; col[7 * 64] = b.w;
        }
        return col[q * 64];
    }
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

// How the decoded symbols leave (OUT_*). A lane decodes its own segment, so a wave's store of one symbol per lane touches 64
// different lines, four bytes each. While few lanes are in flight (one frame: 82 k) the L2 gathers a lane's consecutive symbols
// into whole lines before they go to HBM; with the steps of a frame decoded together (738 k lanes, 8 waves per SIMD) the
// half-written lines no longer fit and leave early: 35.4 GB of HBM writes for 6.05 GB of symbols (rocprofv3 WRITE_SIZE,
// profiles/r04b_rlgr_batch_sq_counters.txt), and that traffic -- not the instruction stream -- was the decoder's time.
// OUT_LDS: a lane parks its symbols in a 16-word column of LDS ([16][64] words: the bank is the lane, no conflicts) and
// writes them as one aligned 64-byte piece (4 x 16 bytes, back to back) whenever the column is full. Needs unit symbol stride
// and 16-byte aligned segment starts.
// OUT_VEC (kept behind a switch, round 3): four symbols buffered in registers, 16-byte stores -- slower than OUT_WORD on one
// frame (the component selects cost more than the stores save).
enum { OUT_WORD = 0, OUT_VEC = 1, OUT_LDS = 2 };
constexpr int DEC_LDS_WORDS = 16 * 64;

template <int OUT, bool LDSIN = false>
__device__ __forceinline__ void decode_segment(const uint32_t *in32, uint32_t nbytes, int n, int flag_signed, int32_t *__restrict__ seq, int64_t sstr = 1,
                                               int32_t *lds = nullptr, int32_t *lds_in = nullptr, int64_t hi_words = 0)
{
    constexpr bool VEC = OUT == OUT_VEC;
    DevBitReaderT<LDSIN> r;
    r.in32 = in32; r.size = nbytes;
    if (LDSIN) { r.col = (uint32_t *)lds_in + (threadIdx.x & 63); r.w0 = (uint32_t)(((uintptr_t)in32 & 31) >> 2); r.hi_words = hi_words; }
    // 32-bit state, as in the encoder: inside a segment of n < 2^31 symbols of int32 data the symbol values (< 2^32), the run
    // length m and the exponents stay within 32 bits; a corrupt stream may overflow them -- it then decodes to different garbage
    // than a 64-bit decoder would, inside the same bounds (every store is at i < n, every read below `size`).
    uint32_t k_P = 0, k_RP = 2 * L;
    int i = 0;
    int4 buf = make_int4(0, 0, 0, 0);
    int32_t *col = lds + (threadIdx.x & 63);                          // OUT_LDS: this lane's column
    auto emit = [&](int32_t v) {
        if (OUT == OUT_LDS) {
            col[(i & 15) * 64] = v;
            ++i;
            if ((i & 15) == 0) {                                     // the column is full: one aligned 64-byte piece
                int4 x[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) x[q] = make_int4(col[(4 * q) * 64], col[(4 * q + 1) * 64], col[(4 * q + 2) * 64], col[(4 * q + 3) * 64]);
#pragma unroll
                for (int q = 0; q < 4; ++q) *(int4 *)(seq + i - 16 + 4 * q) = x[q];
            }
        } else if (VEC) {
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
        }
        // ONE Golomb-Rice site for both modes (a run's closing symbol is coded minus one and always steps k_P down)
        const bool closing = k != 0;
        const uint32_t u = (uint32_t)r.golomb_rice((int)k_R);
        const uint32_t uu = closing ? u + 1u : u;
        emit(flag_signed ? ((uu & 1u) ? -(int32_t)(uu >> 1) - 1 : (int32_t)(uu >> 1)) : (int32_t)uu);
        const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;
        RLGS_ADAPT_KRP32(p);
        if (closing || u) k_P = k_P ? k_P - 1u : 0u;                 // D0 = D1 = 1
        else k_P += U0;
    }
    if (OUT == OUT_LDS && (i & 15)) {                                // the last, partial column
        const int b = i & ~15;
        for (int q = 0; q < (i & 15); ++q) seq[b + q] = col[q * 64];
    }
    if (VEC && (i & 3)) {                                            // the last, partial group (n not a multiple of four)
        const int b = i & ~3;
        seq[b] = buf.x;
        if ((i & 3) > 1) seq[b + 1] = buf.y;
        if ((i & 3) > 2) seq[b + 2] = buf.z;
    }
}

// SYMBOL-SYNCHRONOUS decoding (row-major output). decode_segment above lets every lane run ahead on its own: a lane in a run of
// zeros emits them in an inner loop while the wave's other lanes wait, and the lanes of a wave are at different symbols at any
// time -- in a row-major layout their stores then hit 64 different rows. Here every iteration of the wave produces symbol i of
// EVERY lane: a lane inside a run just counts it down (its zeros cost nothing beside the other lanes' codes), a lane that starts a
// run reads the run's header and emits its first zero, everybody else decodes one Golomb-Rice code (the run's closing symbol
// and an ordinary symbol share that code path; they differ by selects). The lanes of a wave are neighbouring channels at the
// SAME row, so the store of an iteration is one contiguous piece of a row: no LDS staging, no transpose behind the decoder.
// Same streams, same symbols (membuf.cpp:270-331 with the run's state carried across iterations: z zeros still to come, tail = its
// closing symbol still to decode -- not decoded when the segment ends first, as in the original's break).
// expect (may be NULL): what the symbols should be, same layout as seq -- the drivers' round-trip assertion
// (python/encode_3dgs.py:242-245) inside the decoder: in this layout the comparison is one more contiguous read per iteration.
// -> true when a symbol differs.
template <int RMODE>
__device__ __forceinline__ bool decode_segment_sync(const uint32_t *in32, uint32_t nbytes, int n, int flag_signed, int32_t *__restrict__ seq, int64_t sstr,
                                                    int32_t *lds_in = nullptr, int64_t hi_words = 0, const int32_t *__restrict__ expect = nullptr)
{
    bool differs = false;
    DevBitReaderT<RMODE> r;
    r.in32 = in32; r.size = nbytes;
    if (RMODE != 0) { r.col = (uint32_t *)lds_in + (threadIdx.x & 63); r.w0 = (uint32_t)(((uintptr_t)in32 & 31) >> 2); r.hi_words = hi_words; }
    uint32_t k_P = 0, k_RP = 2 * L, z = 0;
    bool tail = false;
    int32_t *wp = seq;                                               // (walked by pointer: a 64-bit multiply per symbol otherwise)
    const int32_t *ep = expect;
    for (int i = 0; i < n; ++i, wp += sstr) {
        if (RMODE == 2 && (i & 31) == 0) r.top_up();
        int32_t v = 0;
        if (z) {
            --z;
        } else {
            bool code = true;
            if (!tail) {
                uint32_t k = k_P / L;
                if (k) {                                             // a run starts: its header
                    uint32_t m = 0;
                    while (r.bit()) {
                        if (k >= 31) { m = (uint32_t)n; break; }     // corrupt stream guard (a run of 2^31 zeros in one segment)
                        m += 1u << k;
                        k_P += U1;
                        k = k_P / L;
                        if (m > (uint32_t)n) break;                  // corrupt stream guard
                    }
                    m += (uint32_t)r.get_wide((int)min(k, 32u));
                    tail = true;
                    if (m) { z = m - 1; code = false; }              // this iteration emits the first of its zeros
                }
            }
            if (code) {
                const uint32_t k_R = k_RP / L;
                const uint32_t u = (uint32_t)r.golomb_rice((int)k_R);
                const uint32_t uu = tail ? u + 1u : u;               // (membuf.cpp: the symbol that ends a run is coded minus one)
                v = flag_signed ? ((uu & 1u) ? -(int32_t)(uu >> 1) - 1 : (int32_t)(uu >> 1)) : (int32_t)uu;
                const uint32_t p = (k_R < 32) ? (u >> k_R) : 0u;
                RLGS_ADAPT_KRP32(p);
                if (tail || u) k_P = k_P ? k_P - 1u : 0u;            // D0 = D1 = 1
                else k_P += U0;
                tail = false;
            }
        }
        *wp = v;
        if (expect) { differs |= *ep != v; ep += sstr; }
    }
    return differs;
}
static_assert(D0 == 1 && D1 == 1, "decode_segment_sync folds the two decrements");

// which way the decoded symbols leave (see OUT_*): RAHT_RLGR_DECODE_OUT=word|vec|lds overrides (A/B knob)
// Default: by the number of lanes in flight. One 3 M x 56 frame (82 k lanes, 1.25 waves per SIMD) is bound by ONE wave's
// instruction stream and the L2 still gathers its lines: OUT_WORD 2.86 ms, OUT_LDS 3.36 ms, OUT_VEC 3.97 ms. Nine such frames by
// one launch (738 k lanes): OUT_WORD 1.76 ms per frame (HBM writes 5.9 x the symbols), OUT_LDS 0.99 ms, OUT_VEC 1.81 ms.
static int g_decode_out = -1;                                       // raht_debug_rlgr_decode_out
static int g_encode_out = -1;                                       // raht_debug_rlgr_encode_out (the batched encoder: words or LDS columns)
static int decode_out_mode(int64_t lanes)
{
    static const char *e = getenv("RAHT_RLGR_DECODE_OUT");
    if (g_decode_out >= 0) return g_decode_out;
    if (e) return e[0] == 'l' ? OUT_LDS : e[0] == 'v' ? OUT_VEC : OUT_WORD;
    return lanes >= 200000 ? OUT_LDS : OUT_WORD;
}

// row-major output: the symbol-synchronous decoder (RAHT_RLGR_DECODE_SYNC=0: the per-lane one with strided stores; A/B knob;
// raht_debug_rlgr_decode_out(3) / (4) force it on / off for the tests)
static int g_decode_sync = -1;
static int decode_sync_rows()
{
    static const char *e = getenv("RAHT_RLGR_DECODE_SYNC");
    if (g_decode_sync >= 0) return g_decode_sync;
    return e ? atoi(e) != 0 : 1;
}

// (with OUT_LDS) the streams' words through LDS as well: RAHT_RLGR_DECODE_IN=word switches it off (A/B knob)
static int decode_lds_in()
{
    static const char *e = getenv("RAHT_RLGR_DECODE_IN");
    return e ? (e[0] == 'l') : 1;
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
                                                        int32_t *__restrict__ Q, int64_t sym_stride, int64_t chan_stride, uint32_t *__restrict__ bad, int out_mode,
                                                        int sync_rows)
{
    __shared__ int32_t s_col[DEC_LDS_WORDS];
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
    const bool aligned = sym_stride == 1 && ((((uintptr_t)Q) & 15) == 0) && ((chan_stride & 3) == 0) && ((S & 3) == 0);
    if (sym_stride != 1 && sync_rows == 2) decode_segment_sync<2>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0 * sym_stride, sym_stride, s_col);
    else if (sym_stride != 1 && sync_rows) decode_segment_sync<0>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0 * sym_stride, sym_stride);
    else if (aligned && out_mode == OUT_LDS) decode_segment<OUT_LDS>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0, 1, s_col);
    else if (aligned && out_mode == OUT_VEC) decode_segment<OUT_VEC>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0);
    else decode_segment<OUT_WORD>((const uint32_t *)(in + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0 * sym_stride, sym_stride);
}

// ---- several frames of ONE shape in one set of launches ------------------------------------------------------------
// What bounds the coders is the number of independent lanes: one frame (3 M x 56 at 2048 symbols per segment) is 1.25 waves
// per SIMD, and a wave that is alone on its SIMD waits out the latency of every one of its ~480 k dependent instructions.
// The quantization steps of a frame (python/encode_3dgs.py:199-275: nine of them) are nine such frames of the same shape with
// nothing between them: coded by ONE launch (blockIdx.y = frame) they fill every wave slot of the chip. Each frame keeps
// its own tables and container: the bytes are those of the one-frame entry points.
constexpr int SEG_BATCH_MAX = RAHT_RLGR_BATCH_MAX;
struct SegEncJobs {
    const int32_t *Q[SEG_BATCH_MAX]; uint32_t *seg_bytes[SEG_BATCH_MAX]; uint32_t *seg_off[SEG_BATCH_MAX]; uint8_t *out[SEG_BATCH_MAX];
    uint64_t cap[SEG_BATCH_MAX];
};
struct SegDecJobs {
    const uint8_t *in[SEG_BATCH_MAX]; uint64_t in_bytes[SEG_BATCH_MAX]; const uint32_t *seg_off[SEG_BATCH_MAX]; const uint32_t *seg_bytes[SEG_BATCH_MAX];
    int32_t *Q[SEG_BATCH_MAX];
    const int32_t *expect[SEG_BATCH_MAX];        // (may be NULL) what Q[j] should become: compared inside the row-major decoder
};

// flags: two words per frame (overflow bits, container bytes)
__global__ __launch_bounds__(64) void seg_encode_slots_batch_kernel(const SegEncJobs J, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int S, int nseg,
                                                                    int flag_signed, uint8_t *__restrict__ slots, uint32_t slot, uint32_t *__restrict__ flags,
                                                                    int lds_out)
{
    __shared__ uint32_t s_col[16 * 64];
    const int j = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t G = (int64_t)D * nseg;
    if (t >= G) return;
    int c, s;
    if (sym_stride == 1) { c = (int)(t / nseg); s = (int)(t - (int64_t)c * nseg); }
    else { s = (int)(t / D); c = (int)(t - (int64_t)s * D); }
    const int64_t g = (int64_t)c * nseg + s;
    const int64_t i0 = (int64_t)s * S;
    const int n = (int)min((int64_t)S, N - i0);
    const int32_t *Q = J.Q[j];
    const int32_t *seq = Q + (int64_t)c * chan_stride + i0 * sym_stride;
    const bool vec = sym_stride == 1 && ((((uintptr_t)Q) & 15) == 0) && ((chan_stride & 3) == 0) && ((S & 3) == 0);
    uint32_t *o = (uint32_t *)(slots + ((size_t)j * (size_t)G + (size_t)g) * slot);
    uint32_t nb;
    if (lds_out) nb = vec ? encode_segment<true, true, true>(seq, n, flag_signed, o, slot, 1, s_col) : encode_segment<true, false, true>(seq, n, flag_signed, o, slot, sym_stride, s_col);
    else nb = vec ? encode_segment<true, true>(seq, n, flag_signed, o, slot) : encode_segment<true, false>(seq, n, flag_signed, o, slot, sym_stride);
    J.seg_bytes[j][g] = nb;
    if (((nb + 3u) & ~3u) > slot) atomicOr(flags + 2 * j, 1u);
}

__global__ void seg_pad_batch_kernel(const SegEncJobs J, int64_t G, uint32_t *__restrict__ padded)
{
    const int j = blockIdx.y;
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < G) padded[(size_t)j * (size_t)G + g] = (J.seg_bytes[j][g] + 3u) & ~3u;
}

// (also closes every frame's offset table: seg_off[G] = its container's bytes, left in flags[2 j + 1] by the scan)
__global__ __launch_bounds__(256) void seg_compact_batch_kernel(const SegEncJobs J, const uint8_t *__restrict__ slots, uint32_t slot, int64_t G,
                                                                uint32_t *__restrict__ flags)
{
    const int j = blockIdx.y;
    const int64_t g = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) J.seg_off[j][G] = flags[2 * j + 1];
    if (g >= G) return;
    const uint32_t nw = (J.seg_bytes[j][g] + 3u) >> 2;
    const uint64_t off = J.seg_off[j][g];
    if (off + 4ull * nw > J.cap[j]) { if (lane == 0) atomicOr(flags + 2 * j, 2u); return; }
    const uint32_t *src = (const uint32_t *)(slots + ((size_t)j * (size_t)G + (size_t)g) * slot);
    uint32_t *dst = (uint32_t *)(J.out[j] + off);
    for (uint32_t i = lane; i < nw; i += 64) dst[i] = src[i];
}

__global__ __launch_bounds__(64) void seg_decode_batch_kernel(const SegDecJobs J, int64_t N, int D, int S, int nseg, int flag_signed, int64_t sym_stride,
                                                              int64_t chan_stride, uint32_t *__restrict__ bad, int out_mode, int lds_in, int sync_rows)
{
    __shared__ int32_t s_col[DEC_LDS_WORDS];
    __shared__ int32_t s_in[8 * 64];
    const int j = blockIdx.y;
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (t >= (int64_t)D * nseg) return;
    int c, s;
    if (sym_stride == 1) { c = (int)(t / nseg); s = (int)(t - (int64_t)c * nseg); }
    else { s = (int)(t / D); c = (int)(t - (int64_t)s * D); }
    const int64_t g = (int64_t)c * nseg + s;
    const int64_t i0 = (int64_t)s * S;
    const int n = (int)min((int64_t)S, N - i0);
    const uint64_t off = J.seg_off[j][g], in_bytes = J.in_bytes[j];
    uint32_t nb = J.seg_bytes[j][g];
    if ((off & 3) || off > in_bytes || (uint64_t)((nb + 3u) & ~3u) > in_bytes - off) { nb = 0; if (bad) atomicOr(bad, 1u << j); }
    int32_t *Q = J.Q[j];
    const bool aligned = sym_stride == 1 && ((((uintptr_t)Q) & 15) == 0) && ((chan_stride & 3) == 0) && ((S & 3) == 0);
    const uint64_t o = nb ? off : 0;
    if (sym_stride != 1 && (sync_rows || J.expect[j])) {
        const int64_t at = (int64_t)c * chan_stride + i0 * sym_stride;
        const int32_t *ex = J.expect[j] ? J.expect[j] + at : nullptr;
        bool differs;
        if (lds_in) differs = decode_segment_sync<2>((const uint32_t *)(J.in[j] + o), nb, n, flag_signed, Q + at, sym_stride, s_col, 0, ex);
        else differs = decode_segment_sync<0>((const uint32_t *)(J.in[j] + o), nb, n, flag_signed, Q + at, sym_stride, nullptr, 0, ex);
        if (__ballot(differs) && (threadIdx.x & 63) == 0 && bad) atomicOr(bad, 1u << (16 + j));
    } else if (aligned && out_mode == OUT_LDS && (((uintptr_t)J.in[j]) & 31) == 0 && lds_in)
        decode_segment<OUT_LDS, true>((const uint32_t *)(J.in[j] + o), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0, 1, s_col, s_in, (int64_t)((in_bytes - o) >> 2));
    else if (aligned && out_mode == OUT_LDS) decode_segment<OUT_LDS>((const uint32_t *)(J.in[j] + o), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0, 1, s_col);
    else decode_segment<OUT_WORD>((const uint32_t *)(J.in[j] + (nb ? off : 0)), nb, n, flag_signed, Q + (int64_t)c * chan_stride + i0 * sym_stride, sym_stride);
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
                       seg_len, (int)nseg, flag_signed, Q, sym_stride, chan_stride, bad_dev, rlgr_seg::decode_out_mode(G), rlgr_seg::decode_sync_rows() ? (rlgr_seg::decode_lds_in() ? 2 : 1) : 0);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_debug_rlgr_decode_out(int mode)
{
    const int prev = rlgr_seg::g_decode_out;
    if (mode == 3 || mode == 4) { rlgr_seg::g_decode_sync = (mode == 3); return prev; }     // row-major output: symbol-synchronous decoder on / off
    rlgr_seg::g_decode_out = (mode >= 0 && mode <= 2) ? mode : -1;
    if (mode < 0) rlgr_seg::g_decode_sync = -1;
    return prev;
}

int raht_debug_rlgr_encode_out(int mode)
{
    const int prev = rlgr_seg::g_encode_out;
    rlgr_seg::g_encode_out = (mode == 0 || mode == 2) ? mode : -1;
    return prev;
}

/* k frames of one shape (N, D, strides, seg_len) in one set of launches: the quantization steps of a frame coded together.
 * Every frame j has its own input Q[j], tables seg_bytes[j] / seg_off[j], container out[j] of cap[j] bytes and total_bytes[j]:
 * exactly what raht_rlgr_seg_encode_strided leaves for that frame alone (same bytes). Synchronises. RAHT_ERR_NOMEM when a
 * container is too small (total_bytes[] holds what every frame needs). */
int raht_rlgr_seg_encode_batch(int k, const int32_t *const *Q, int64_t N, int D, int64_t sym_stride, int64_t chan_stride, int seg_len, int flag_signed,
                               uint32_t *const *seg_bytes, uint32_t *const *seg_off, uint8_t *const *out, const int64_t *cap, int64_t *total_bytes,
                               raht_stream_t stream)
{
    if (k < 1 || k > RAHT_RLGR_BATCH_MAX || !Q || !seg_bytes || !seg_off || !out || !cap || !total_bytes) {
        set_error("raht_rlgr_seg_encode_batch: bad argument (1 <= k <= %d)", RAHT_RLGR_BATCH_MAX);
        return RAHT_ERR_INVALID;
    }
    if (N < 1 || D < 1 || sym_stride < 1 || chan_stride < 1 || seg_len < 64 ||
        !((sym_stride == 1 && chan_stride >= N) || (chan_stride == 1 && sym_stride >= D))) {
        set_error("raht_rlgr_seg_encode_batch: bad argument (channel-major: sym_stride 1, chan_stride >= N; row-major: chan_stride 1, sym_stride >= D)");
        return RAHT_ERR_INVALID;
    }
    for (int j = 0; j < k; ++j)
        if (!Q[j] || !seg_bytes[j] || !seg_off[j] || !out[j] || cap[j] < 16 || ((uintptr_t)out[j] & 3)) { set_error("raht_rlgr_seg_encode_batch: bad argument (frame %d)", j); return RAHT_ERR_INVALID; }
    const int64_t nseg = ceil_div(N, seg_len), G = nseg * D;
    if (nseg >= ((int64_t)1 << 31) || G >= ((int64_t)1 << 31)) { set_error("raht_rlgr_seg_encode_batch: too many segments"); return RAHT_ERR_INVALID; }
    if (raht_rlgr_bound(seg_len) * G + 4 * G >= ((int64_t)1 << 32)) {
        set_error("raht_rlgr_seg_encode_batch: %lld x %d symbols may need a container of 4 GiB or more (32-bit segment offsets): split the frame", (long long)N, D);
        return RAHT_ERR_INVALID;
    }
    hipStream_t s = (hipStream_t)stream;
    auto one_by_one = [&]() -> int {
        int rc_all = RAHT_OK;
        for (int j = 0; j < k; ++j) {
            const int rc = raht_rlgr_seg_encode_strided(Q[j], N, D, sym_stride, chan_stride, seg_len, flag_signed, seg_bytes[j], seg_off[j], out[j], cap[j], &total_bytes[j], stream);
            if (rc != RAHT_OK && rc_all == RAHT_OK) rc_all = rc;
            if (rc != RAHT_OK && rc != RAHT_ERR_NOMEM) return rc;
        }
        return rc_all;
    };
    static const bool no_batch = getenv("RAHT_RLGR_NO_BATCH") != nullptr;          // A/B knob
    if (k == 1 || no_batch) return one_by_one();
    // slots of whole 64-byte pieces (16-byte aligned starts are what the LDS-column output needs; 64: whole pieces)
    const uint32_t slot = (4u * (uint32_t)seg_len + 16u + 63u) & ~63u;
    static const char *enc_out = getenv("RAHT_RLGR_ENCODE_OUT");                  // A/B knob: word | lds
    const int lds_out = rlgr_seg::g_encode_out >= 0 ? (rlgr_seg::g_encode_out == rlgr_seg::OUT_LDS)
                                                    : enc_out ? (enc_out[0] == 'l') : ((int64_t)k * G >= 200000);
    int rc = guarded("raht_rlgr_seg_encode_batch", [&]() -> int {
        Scratch tmp(sizeof(uint32_t) * ((size_t)k * (size_t)G + 2 * (size_t)k), s);
        Scratch slots((size_t)k * (size_t)G * slot, s);
        if (!tmp.ok() || !slots.ok()) { (void)hipGetLastError(); return RAHT_ERR_UNSUPPORTED; }      // no room for k sets of slots: frame by frame
        uint32_t *padded = tmp.as<uint32_t>(), *flags = padded + (size_t)k * (size_t)G;
        RAHT_HIP_CHECK(hipMemsetAsync(flags, 0, 8 * (size_t)k, s));
        rlgr_seg::SegEncJobs J;
        for (int j = 0; j < RAHT_RLGR_BATCH_MAX; ++j) {
            const int q = j < k ? j : 0;
            J.Q[j] = Q[q]; J.seg_bytes[j] = seg_bytes[q]; J.seg_off[j] = seg_off[q]; J.out[j] = out[q]; J.cap[j] = (uint64_t)cap[q];
        }
        hipLaunchKernelGGL(rlgr_seg::seg_encode_slots_batch_kernel, dim3((unsigned)ceil_div(G, 64), (unsigned)k), dim3(64), 0, s, J, N, D, sym_stride, chan_stride, seg_len,
                           (int)nseg, flag_signed, slots.as<uint8_t>(), slot, flags, lds_out);
        hipLaunchKernelGGL(rlgr_seg::seg_pad_batch_kernel, dim3((unsigned)ceil_div(G, 256), (unsigned)k), dim3(256), 0, s, J, G, padded);
        for (int j = 0; j < k; ++j) RAHT_RET(exclusive_scan_u32(padded + (size_t)j * (size_t)G, seg_off[j], G, flags + 2 * j + 1, s));
        hipLaunchKernelGGL(rlgr_seg::seg_compact_batch_kernel, dim3((unsigned)ceil_div(G * 64, 256), (unsigned)k), dim3(256), 0, s, J, slots.as<uint8_t>(), slot, G, flags);
        RAHT_HIP_CHECK(hipGetLastError());
        uint32_t back[2 * RAHT_RLGR_BATCH_MAX] = {0};
        RAHT_RET(read_back_u32(back, flags, 2 * k, nullptr, nullptr, 0, s));        // (synchronises: the scratch may go back to the pool)
        int rc2 = RAHT_OK;
        for (int j = 0; j < k; ++j) {
            if (back[2 * j] & 1u) return RAHT_ERR_UNSUPPORTED;                        // a segment outgrew its slot (incompressible data): the exact passes
            total_bytes[j] = (int64_t)back[2 * j + 1];
            if ((back[2 * j] & 2u) || (int64_t)back[2 * j + 1] > cap[j]) {
                set_error("raht_rlgr_seg_encode_batch: frame %d needs %u bytes, cap = %lld", j, back[2 * j + 1], (long long)cap[j]);
                rc2 = RAHT_ERR_NOMEM;
            }
        }
        return rc2;
    });
    if (rc == RAHT_ERR_UNSUPPORTED) return one_by_one();
    return rc;
}

/* The inverse for k frames of one shape: in[j] / in_bytes[j] / seg_off[j] / seg_bytes[j] -> Q[j], one launch. Does not
 * synchronise. *bad_dev (DEVICE uint32, may be NULL): bit j set when a table entry of frame j reached outside in[j]. */
static int seg_decode_batch_impl(int k, const uint8_t *const *in, const int64_t *in_bytes, const uint32_t *const *seg_off, const uint32_t *const *seg_bytes,
                                 int64_t N, int D, int seg_len, int flag_signed, int32_t *const *Q, const int32_t *const *expect, int64_t sym_stride,
                                 int64_t chan_stride, uint32_t *bad_dev, raht_stream_t stream);

int raht_rlgr_seg_decode_batch(int k, const uint8_t *const *in, const int64_t *in_bytes, const uint32_t *const *seg_off, const uint32_t *const *seg_bytes,
                               int64_t N, int D, int seg_len, int flag_signed, int32_t *const *Q, int64_t sym_stride, int64_t chan_stride,
                               uint32_t *bad_dev, raht_stream_t stream)
{
    return seg_decode_batch_impl(k, in, in_bytes, seg_off, seg_bytes, N, D, seg_len, flag_signed, Q, nullptr, sym_stride, chan_stride, bad_dev, stream);
}

/* ... and compares every frame with what it should decode to (expect[j]: DEVICE, the layout and strides of Q[j]; ROW-MAJOR only:
 * chan_stride = 1) on the way: bit 16 + j of *bad_dev (required) is set when a symbol of frame j differs -- the drivers'
 * round-trip assertion (python/encode_3dgs.py:242-245) without a pass of its own over two N x D arrays. */
int raht_rlgr_seg_decode_batch_check(int k, const uint8_t *const *in, const int64_t *in_bytes, const uint32_t *const *seg_off, const uint32_t *const *seg_bytes,
                                     int64_t N, int D, int seg_len, int flag_signed, int32_t *const *Q, const int32_t *const *expect, int64_t sym_stride,
                                     int64_t chan_stride, uint32_t *bad_dev, raht_stream_t stream)
{
    if (!expect || !bad_dev || chan_stride != 1) { set_error("raht_rlgr_seg_decode_batch_check: needs expect[], bad_dev and row-major frames (chan_stride 1)"); return RAHT_ERR_INVALID; }
    for (int j = 0; j < k && j < RAHT_RLGR_BATCH_MAX; ++j)
        if (!expect[j]) { set_error("raht_rlgr_seg_decode_batch_check: expect[%d] is NULL", j); return RAHT_ERR_INVALID; }
    return seg_decode_batch_impl(k, in, in_bytes, seg_off, seg_bytes, N, D, seg_len, flag_signed, Q, expect, sym_stride, chan_stride, bad_dev, stream);
}

static int seg_decode_batch_impl(int k, const uint8_t *const *in, const int64_t *in_bytes, const uint32_t *const *seg_off, const uint32_t *const *seg_bytes,
                                 int64_t N, int D, int seg_len, int flag_signed, int32_t *const *Q, const int32_t *const *expect, int64_t sym_stride,
                                 int64_t chan_stride, uint32_t *bad_dev, raht_stream_t stream)
{
    if (k < 1 || k > RAHT_RLGR_BATCH_MAX || !in || !in_bytes || !seg_off || !seg_bytes || !Q || N < 1 || D < 1 || seg_len < 64 ||
        !((sym_stride == 1 && chan_stride >= N) || (chan_stride == 1 && sym_stride >= D))) {
        set_error("raht_rlgr_seg_decode_batch: bad argument (1 <= k <= %d)", RAHT_RLGR_BATCH_MAX);
        return RAHT_ERR_INVALID;
    }
    for (int j = 0; j < k; ++j)
        if (!in[j] || in_bytes[j] < 0 || (in_bytes[j] & 3) || ((uintptr_t)in[j] & 3) || !seg_off[j] || !seg_bytes[j] || !Q[j]) {
            set_error("raht_rlgr_seg_decode_batch: bad argument (frame %d)", j);
            return RAHT_ERR_INVALID;
        }
    const int64_t nseg = ceil_div(N, seg_len), G = nseg * D;
    if (G >= ((int64_t)1 << 31)) { set_error("raht_rlgr_seg_decode_batch: too many segments"); return RAHT_ERR_INVALID; }
    rlgr_seg::SegDecJobs J;
    for (int j = 0; j < RAHT_RLGR_BATCH_MAX; ++j) {
        const int q = j < k ? j : 0;
        J.in[j] = in[q]; J.in_bytes[j] = (uint64_t)in_bytes[q]; J.seg_off[j] = seg_off[q]; J.seg_bytes[j] = seg_bytes[q]; J.Q[j] = Q[q];
        J.expect[j] = expect ? expect[q] : nullptr;
    }
    hipLaunchKernelGGL(rlgr_seg::seg_decode_batch_kernel, dim3((unsigned)ceil_div(G, 64), (unsigned)k), dim3(64), 0, (hipStream_t)stream, J, N, D, seg_len, (int)nseg,
                       flag_signed, sym_stride, chan_stride, bad_dev, rlgr_seg::decode_out_mode((int64_t)k * G), rlgr_seg::decode_lds_in(), rlgr_seg::decode_sync_rows());
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

}  // extern "C"
