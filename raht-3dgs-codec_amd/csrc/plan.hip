// plan.hip -- the RAHT "plan": list-free replacement of the reference's RAHT_param_reorder_fast
// (reference python/RAHT_param.py:190-279), built entirely on device from the sorted Morton keys.
//
// Structural identity used (verified against the reference lists by
// tests/test_plan_*): with d[i] = msb(key[i] ^ key[i-1]), row i >= 1 is a right sibling at exactly
// one binary level l = d[i]; its left partner is the first row of the level-l node that contains
// row i-1, and its own subtree ends at the first later row whose key differs above bit l. Both ends
// are found by a galloping + binary search over the sorted keys (O(log subtree) per row).
#include "raht_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>

namespace raht {

__global__ void compact_scatter_kernel(const uint32_t *in, const uint32_t *flag, const uint32_t *pos, uint32_t *out, int64_t n);

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_plan_device(const raht_plan *plan, const char *what)
{
    if (!plan) { set_error("%s: NULL plan", what); return RAHT_ERR_INVALID; }
    const int cur = current_device();
    if (cur != plan->device) {
        set_error("%s: the plan lives on HIP device %d but device %d is current (make the plan's device current "
                  "before calling; see raht.h, devices and threads)", what, plan->device, cur);
        return RAHT_ERR_INVALID;
    }
    return RAHT_OK;
}

// ---- device error word -------------------------------------------------------------------------
struct PlanErr {
    int code;                 // first error code seen (0 = none)
    unsigned int row;         // smallest offending row
};

__device__ __forceinline__ void report(PlanErr *e, int code, int64_t row)
{
    atomicCAS(&e->code, 0, code);
    atomicMin(&e->row, (unsigned int)row);
}

// ---- Morton keys -------------------------------------------------------------------------------
// get_morton_code (voxelize_pc.py:25-59) / RAHT_param.py:208-212: digit_k = z_k + 2 y_k + 4 x_k at
// bits [3k, 3k+2]. Implemented with the 21-bit "spread by 3" magic-number sequence.
__device__ __forceinline__ uint64_t spread3(uint64_t v)
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

__device__ __forceinline__ uint64_t morton3(uint64_t x, uint64_t y, uint64_t z)
{
    return spread3(z) | (spread3(y) << 1) | (spread3(x) << 2);
}

template <typename VT>
__global__ void keys_from_coords_kernel(const VT *__restrict__ V, int64_t N, double m0, double m1,
                                        double m2, double Q, int depth, uint64_t *__restrict__ keys,
                                        PlanErr *err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    // RAHT_param.py:205-206  Vint = floor((V - minV) / Q)
    const int64_t x = (int64_t)floor(((double)V[3 * i + 0] - m0) / Q);
    const int64_t y = (int64_t)floor(((double)V[3 * i + 1] - m1) / Q);
    const int64_t z = (int64_t)floor(((double)V[3 * i + 2] - m2) / Q);
    const int64_t hi = (int64_t)1 << depth;
    if (x < 0 || y < 0 || z < 0 || x >= hi || y >= hi || z >= hi) report(err, RAHT_ERR_BOUNDS, i);
    keys[i] = morton3((uint64_t)x, (uint64_t)y, (uint64_t)z);
}

__global__ void morton_i64_kernel(const int64_t *__restrict__ V, int64_t N, uint64_t *__restrict__ keys)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    keys[i] = morton3((uint64_t)V[3 * i + 0], (uint64_t)V[3 * i + 1], (uint64_t)V[3 * i + 2]);
}

// ---- lvl[], wl[], wr[] ---------------------------------------------------------------------------
// lvl[i] = highest bit in which key[i] differs from key[i-1] (row i starts a node at every level <= lvl[i]).
// wr[i]  = rows of the level-lvl[i] node that starts at row i  = (first m > i with lvl[m] >= lvl[i], or N) - i
// wl[i]  = rows of the level-lvl[i] node that ends at row i-1  = i - (last m < i with lvl[m] >= lvl[i]); lvl[0] = 255
//
// (how the neighbours are found: see level_extent_kernel below)
static constexpr int EXT_THREADS = 1024;
static constexpr int EXT_WAVES = EXT_THREADS / 64;

// Profiling build only (-DRAHT_PHASE_CLOCKS, tools/phase_clocks_plan.py): thread 0 of a plan-build workgroup stamps the shader
// clock at its phase boundaries. [0] level_extent blocks, [1] tile_heights tiles, [2] sched_tail (one workgroup).
#ifdef RAHT_PHASE_CLOCKS
constexpr int PL_CLK_BLOCKS = 4096, PL_CLK_SLOTS = 12;
__device__ unsigned long long g_phase_clk_plan[3][PL_CLK_BLOCKS][PL_CLK_SLOTS];
#define PL_STAMP(which, blk, k) do { if (threadIdx.x == 0 && (blk) < PL_CLK_BLOCKS) g_phase_clk_plan[which][blk][k] = __builtin_readcyclecounter(); } while (0)
#define PL_NOTE_LEVELS(blk, n) do { if (threadIdx.x == 0 && (blk) < PL_CLK_BLOCKS) g_phase_clk_plan[1][blk][PL_CLK_SLOTS - 1] = (unsigned long long)(n); } while (0)
#define PL_SLOT_DECL int pl_slot = 0
#define PL_STAMP_NEXT() do { if (pl_slot < 8) { PL_STAMP(2, 0, pl_slot); ++pl_slot; } } while (0)
#else
#define PL_STAMP(which, blk, k) do { } while (0)
#define PL_NOTE_LEVELS(blk, n) do { } while (0)
#define PL_SLOT_DECL do { } while (0)
#define PL_STAMP_NEXT() do { } while (0)
#endif

// The same pass also counts, per block of EXT_THREADS rows, the rows of every order_RAGFT bucket (ORDER_BUCKETS
// bins) and of every binary level (64 bins) -> bucket_hist[bin * gridDim.x + block]: the input of the stable
// counting sort that produces order_RAGFT (no separate histogram pass over the rows) and, scanned, the start of
// every level among the rows (max_level / len(Flags), the level engine's level offsets).
static constexpr int ORDER_BUCKETS = 32;             // bucket(0) = 0, bucket(i) = 1 + (20 - lvl / 3) <= 21

// One queued search, over a monotone predicate on the keys ((key >> l) == prefix holds exactly on the node):
// gallop away from the block by x8, then split the bracket in 8 with 7 independent probes per step -- the chain
// of dependent loads is what a search costs. Every probe address is a valid row.
// q = row within the block | direction << 31 (0 = right: where does the node STARTING at the row end;
// 1 = left: where does the node ENDING at the row before it start).
static constexpr int EXT_QCAP = 96;                  // queue slots per block handed to extent_finish_kernel
__device__ __forceinline__ void extent_search(const uint64_t *__restrict__ keys, int64_t N, int64_t b0, uint32_t q, int ql,
                                              int32_t *__restrict__ wl, int32_t *__restrict__ wr)
{
    const bool right = (q >> 31) == 0;
    const int64_t r = b0 + (q & 0x7fffffffu);
    const int64_t dir = right ? 1 : -1;
    // right: the node starting at row r reaches at least to the end of its BLOCK; left: the node
    // ending at row r - 1 reaches back at least to the row before the block (never block 0: it holds row 0)
    const uint64_t pref = keys[right ? r : r - 1] >> ql;
    int64_t in = right ? min(b0 + 1024 - 1, N - 1) : max(b0 - 1, (int64_t)0);
    int64_t out = right ? N : -1;
    for (int64_t step = 1;; step <<= 3) {
        const int64_t p = in + dir * step;
        const bool in_range = right ? p < N : p >= 0;
        const uint64_t k = keys[min(max(p, (int64_t)0), N - 1)];
        if (in_range && (k >> ql) == pref) in = p;
        else { out = right ? min(p, N) : max(p, (int64_t)-1); break; }
    }
    while ((right ? out - in : in - out) > 1) {
        const int64_t w = right ? out - in : in - out;
        uint64_t k[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) k[j] = keys[in + dir * ((w * (j + 1)) >> 3)];     // between in and out
        int64_t nin = in, nout = out;
        bool hit = false;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int64_t p = in + dir * ((w * (j + 1)) >> 3);
            if (!hit) { if ((k[j] >> ql) == pref) nin = p; else { nout = p; hit = true; } }
        }
        in = nin; out = nout;
    }
    if (right) wr[r] = (int32_t)(out - r);
    else wl[r] = (int32_t)(r - in);
}

// How the neighbours are found. A row's right (left) neighbour is the next (previous) row whose level is >= its
// own. For every binary level t < nbits every wave publishes ONE 64-bit word: which of its rows have a level
// >= t (a v_cmp into a scalar register pair, stored by one lane: 8 KiB of LDS per 1024-row block). A row of
// level l then reads the words of column l: its own wave's word gives the neighbour inside the wave, the
// following (preceding) waves' words the neighbour inside the block -- typically one to three 8-byte LDS reads,
// no dependent chain, no memory traffic. Only rows whose node leaves the block (~1 %) are queued and search
// the keys (gallop x8, then 8-way splits). Rows past the end of the scene and row 0 count as level 255: they
// end every node. (Earlier versions: one key search per row, 76 us on cfg3 -- every wave paid the instruction
// stream of its longest search; a loop over the distinct levels of each wave with the key search for the one
// row in six that left its wave, 62 us -- ~22 iterations per wave serialised on a scalar read-lane each.)
__global__ void __launch_bounds__(EXT_THREADS) level_extent_kernel(const uint64_t *__restrict__ keys, int64_t N, int nbits,
                                                                    uint8_t *__restrict__ lvl, uint8_t *__restrict__ order_bucket,
                                                                    int32_t *__restrict__ wl, int32_t *__restrict__ wr, PlanErr *err,
                                                                    uint32_t *__restrict__ bucket_hist, uint32_t *__restrict__ gq,
                                                                    uint32_t *__restrict__ gq_count)
{
    __shared__ uint64_t s_ge[EXT_WAVES][64];         // [wave][level]: rows of the wave with a level >= `level`
    __shared__ uint32_t queue[2 * EXT_THREADS];      // row within the block | direction << 31
    __shared__ uint8_t s_lvl[EXT_THREADS];
    __shared__ uint32_t n_queued;
    __shared__ uint32_t s_lh[64];
    __shared__ uint32_t s_present[2];
    PL_STAMP(0, blockIdx.x, 0);
    if (threadIdx.x == 0) { n_queued = 0; s_present[0] = 0; s_present[1] = 0; }
    if (threadIdx.x < 64) s_lh[threadIdx.x] = 0;
    __syncthreads();
    const int64_t b0 = (int64_t)blockIdx.x * EXT_THREADS;
    const int64_t i = b0 + threadIdx.x;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool valid = i < N;
    int l = 255;                                     // rows past the end (and row 0) end every node
    if (valid) {
        const uint64_t k = keys[i];
        if (nbits < 64 && (k >> nbits) != 0) report(err, RAHT_ERR_BOUNDS, i);
        if (i != 0) {
            const uint64_t p = keys[i - 1];
            if (k <= p) { report(err, RAHT_ERR_UNSORTED, i); l = 0; }
            else l = 63 - __clzll((long long)(k ^ p));
        }
        lvl[i] = (uint8_t)l;
        s_lvl[threadIdx.x] = (uint8_t)l;
        // order_RAGFT (RAHT_param.py:251-274): [root] ++ groups of rows that stop being node starts within
        // octree level g = lvl / 3, coarse to fine, ascending row index inside a group  ==  a stable bucket
        // sort by bucket(0) = 0, bucket(i) = 1 + (20 - lvl[i] / 3)
        order_bucket[i] = (i == 0) ? 0 : (uint8_t)(1 + (20 - l / 3));
    }
    const bool searching = valid && i != 0;
    // Which levels occur in this BLOCK at all: a row only ever reads the words of its own level, so only those columns
    // are needed -- typically a dozen of the up to 63 (the kernel was bound by the two wave-wide compares per level and
    // wave: 36 levels on cfg3). Wave-wide OR of 1 << l by butterfly shuffles, one LDS atomic per wave.
    {
        uint32_t plo = (l < 32) ? (1u << l) : 0u, phi = (l >= 32 && l < 64) ? (1u << (l - 32)) : 0u;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) { plo |= (uint32_t)__shfl_xor((int)plo, d, 64); phi |= (uint32_t)__shfl_xor((int)phi, d, 64); }
        if (lane == 0) { if (plo) atomicOr(&s_present[0], plo); if (phi) atomicOr(&s_present[1], phi); }
    }
    PL_STAMP(0, blockIdx.x, 1);                       // keys loaded, levels known, lvl / bucket stores issued
    __syncthreads();
    PL_STAMP(0, blockIdx.x, 2);
    // one word per (level present, wave), and the level histogram (one LDS atomic per wave and level present)
    uint64_t present = (uint64_t)s_present[0] | ((uint64_t)s_present[1] << 32);
    if (nbits < 64) present &= ((uint64_t)1 << max(nbits, 1)) - 1;    // levels are < nbits (out-of-range keys are reported, not indexed)
    present &= ~((uint64_t)1 << 63);
    // (the words of a wave collect in LANE t of three registers -- v_writelane, no branch and no LDS traffic inside the loop -- and
    // leave with one store and one atomic per wave: the loop was 28 % of this kernel, a third of it the lane-0 branches)
    {
        // (`present` came from LDS, i.e. in vector registers: made scalar, the loop's control runs on the scalar unit)
        present = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)present) |
                  ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(present >> 32)) << 32);
        const uint64_t present_all = present;
        uint32_t w_lo = 0, w_hi = 0, w_eq = 0;
        while (present) {
            const int t = __ffsll((unsigned long long)present) - 1;
            present &= present - 1;
            const uint64_t ge = __ballot(l >= t);
            const uint32_t ge_lo = (uint32_t)ge, ge_hi = (uint32_t)(ge >> 32), eq = (uint32_t)__popcll(__ballot(l == t));
            // (no writelane builtin in this hipcc; the lane select goes through M0: two SGPR operands exceed the constant bus)
            // (M0 is saved and restored: the compiler does not track it as a clobber)
            uint32_t m0_keep;
            asm volatile("s_mov_b32 %3, m0\n\ts_mov_b32 m0, %7\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0\n\ts_mov_b32 m0, %3"
                         : "+v"(w_lo), "+v"(w_hi), "+v"(w_eq), "=&s"(m0_keep) : "s"(ge_lo), "s"(ge_hi), "s"(eq), "s"(t));
        }
        if ((present_all >> lane) & 1) {
            s_ge[wv][lane] = (uint64_t)w_lo | ((uint64_t)w_hi << 32);
            if (w_eq) atomicAdd(&s_lh[lane], w_eq);
        }
    }
    if (i == 0) s_lh[63] = 1;                        // row 0 (lvl 255): alone in level bin 63
    PL_STAMP(0, blockIdx.x, 3);                       // words of every present level published
    __syncthreads();
    PL_STAMP(0, blockIdx.x, 4);
    const uint64_t below = ((uint64_t)1 << lane) - 1, above = ~(below | ((uint64_t)1 << lane));
    bool q_r = false, q_l = false;
    if (searching) {
        const int lc = min(l, 63);
        // right neighbour: own wave, then the following waves of the block
        int found = -1;
        {
            const uint64_t m = s_ge[wv][lc] & above;
            if (m) found = wv * 64 + __ffsll((unsigned long long)m) - 1;
            for (int w2 = wv + 1; found < 0 && w2 < EXT_WAVES; ++w2) {
                const uint64_t m2 = s_ge[w2][lc];
                if (m2) found = w2 * 64 + __ffsll((unsigned long long)m2) - 1;
            }
        }
        if (found >= 0) wr[i] = (int32_t)(min((int64_t)found + b0, N) - i);   // (a row past the end stands for row N)
        else if (b0 + EXT_THREADS >= N) wr[i] = (int32_t)(N - i);
        else q_r = true;
        // left neighbour: own wave, then the preceding waves
        found = -1;
        {
            const uint64_t m = s_ge[wv][lc] & below;
            if (m) found = wv * 64 + 63 - __clzll((long long)m);
            for (int w2 = wv - 1; found < 0 && w2 >= 0; --w2) {
                const uint64_t m2 = s_ge[w2][lc];
                if (m2) found = w2 * 64 + 63 - __clzll((long long)m2);
            }
        }
        if (found >= 0) wl[i] = (int32_t)((int)threadIdx.x - found);
        else q_l = true;                              // (never in block 0: it holds row 0)
    } else if (valid) {
        wl[0] = 0; wr[0] = 0;
    }
    PL_STAMP(0, blockIdx.x, 5);                       // neighbours found, wl / wr stores issued
    const uint64_t m_r = __ballot(q_r), m_l = __ballot(q_l);
    if (m_r | m_l) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&n_queued, (uint32_t)(__popcll(m_r) + __popcll(m_l)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (q_r) queue[base + __popcll(m_r & below)] = threadIdx.x;
        if (q_l) queue[base + __popcll(m_r) + __popcll(m_l & below)] = threadIdx.x | 0x80000000u;
    }
    // per-block histograms, bin-major: [ORDER_BUCKETS order buckets | 64 binary levels] x gridDim.x. ONE exclusive scan
    // over the whole array turns the first part into the order scatter's positions and, read at every level's first
    // block, the second part into the start of every level among the rows (no global atomics)
    if (threadIdx.x < 64) bucket_hist[(size_t)(ORDER_BUCKETS + threadIdx.x) * gridDim.x + blockIdx.x] = s_lh[threadIdx.x];
    if (threadIdx.x < ORDER_BUCKETS) {
        uint32_t c = 0;                               // order bucket b >= 1 = levels 3 (21 - b) .. 3 (21 - b) + 2; bucket 0 = row 0
        const int b = threadIdx.x;
        if (b == 0) c = s_lh[63];
        else if (b <= 21) { const int l0 = 3 * (21 - b); c = s_lh[l0] + s_lh[l0 + 1] + s_lh[l0 + 2]; }
        bucket_hist[(size_t)b * gridDim.x + blockIdx.x] = c;
    }
    PL_STAMP(0, blockIdx.x, 6);                       // queue + histograms written
    __syncthreads();
    PL_STAMP(0, blockIdx.x, 7);
    // The queued searches go to extent_finish_kernel: a block that ran them itself kept its 16 wave slots
    // until its slowest search (a chain of ~5 dependent loads) had finished, and the next block of the CU could
    // not start -- that tail, not the work, was two thirds of this kernel's 75 us. Each block owns EXT_QCAP slots
    // of a global queue; the (pathological) rest it still searches itself.
    const uint32_t nq = n_queued;
    if (threadIdx.x == 0) gq_count[blockIdx.x] = min(nq, (uint32_t)EXT_QCAP);
    if (threadIdx.x < min(nq, (uint32_t)EXT_QCAP)) gq[(size_t)blockIdx.x * EXT_QCAP + threadIdx.x] = queue[threadIdx.x];
    for (uint32_t t = EXT_QCAP + threadIdx.x; t < nq; t += EXT_THREADS) {
        const uint32_t q = queue[t];
        extent_search(keys, N, b0, q, s_lvl[q & 0x7fffffffu], wl, wr);
    }
    PL_STAMP(0, blockIdx.x, 8);
}

__device__ __forceinline__ void extent_search_block(uint32_t block, const uint64_t *__restrict__ keys, int64_t N, const uint8_t *__restrict__ lvl,
                                                    const uint32_t *__restrict__ gq, const uint32_t *__restrict__ gq_count,
                                                    uint32_t nblk, int32_t *__restrict__ wl, int32_t *__restrict__ wr)
{
    const uint32_t g = block * blockDim.x + threadIdx.x;
    const uint32_t b = g / EXT_QCAP, slot = g - b * EXT_QCAP;
    if (b >= nblk || slot >= gq_count[b]) return;
    const uint32_t q = gq[(size_t)b * EXT_QCAP + slot];
    const int64_t b0 = (int64_t)b * EXT_THREADS;
    extent_search(keys, N, b0, q, (int)lvl[b0 + (q & 0x7fffffffu)], wl, wr);
}

// ---- order_RAGFT -------------------------------------------------------------------------------
// Stable counting sort of the rows by order bucket, second half: bucket_pos holds, for every (bucket, block),
// where that block's first row of that bucket goes (exclusive scan of bucket_hist, bucket-major). Inside the
// block the rank of a row among the rows of its bucket is (rows of that bucket in earlier waves) + (earlier
// lanes of its wave with the same bucket). Writes the permutation AND its inverse: inv_order[i] = pos.
// First half of that sort: ONE launch, one workgroup per bin, turns the bin-major per-block counts into positions INSIDE the
// bin (exclusive, in place) and the bin's total; the scatter adds up the totals of the bins before its own by itself (96 values).
// (The generic two-launch scan over the whole array cost 17 us of a 200 us build, and a third of a launch gap.)
constexpr int BSCAN_THREADS = 256;
__device__ __forceinline__ void bucket_scan_block(uint32_t bin, uint32_t *__restrict__ hist, uint32_t nblk, uint32_t *__restrict__ bin_total)
{
    __shared__ uint32_t ws[BSCAN_THREADS / 64];
    uint32_t *row = hist + (size_t)bin * nblk;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t b0 = 0; b0 < nblk; b0 += BSCAN_THREADS * 8) {
        const uint32_t i0 = b0 + threadIdx.x * 8;
        uint32_t v[8], t = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { v[k] = (i0 + k < nblk) ? row[i0 + k] : 0u; t += v[k]; }
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        if (lane == 63) ws[wid] = inc;
        __syncthreads();
        uint32_t before = carry, all = 0;
#pragma unroll
        for (int w = 0; w < BSCAN_THREADS / 64; ++w) { const uint32_t x = ws[w]; if (w < wid) before += x; all += x; }
        __syncthreads();
        uint32_t ex = before + inc - t;
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (i0 + k < nblk) row[i0 + k] = ex; ex += v[k]; }
        carry += all;
    }
    if (threadIdx.x == 0) bin_total[bin] = carry;
}

// Second launch of a plan build, two jobs that both wait for level_extent_kernel and for nothing else: the first ORDER_BUCKETS + 64
// workgroups scan one histogram bin each, the others run the queued extent searches (one launch instead of two: a back-to-back
// launch costs ~4.6 us even when the chip could run both at once)
__global__ void __launch_bounds__(BSCAN_THREADS) extent_finish_kernel(const uint64_t *__restrict__ keys, int64_t N, const uint8_t *__restrict__ lvl,
                                                                      const uint32_t *__restrict__ gq, const uint32_t *__restrict__ gq_count,
                                                                      uint32_t nblk, int32_t *__restrict__ wl, int32_t *__restrict__ wr,
                                                                      uint32_t *__restrict__ hist, uint32_t *__restrict__ bin_total)
{
    if (blockIdx.x < ORDER_BUCKETS + 64) bucket_scan_block(blockIdx.x, hist, nblk, bin_total);
    else extent_search_block(blockIdx.x - (ORDER_BUCKETS + 64), keys, N, lvl, gq, gq_count, nblk, wl, wr);
}

__global__ void __launch_bounds__(EXT_THREADS) order_scatter_kernel(const uint8_t *__restrict__ order_bucket, int64_t N,
                                                                     const uint32_t *__restrict__ bucket_pos,
                                                                     const uint32_t *__restrict__ bin_total,
                                                                     uint32_t *__restrict__ order, uint32_t *__restrict__ inv_order,
                                                                     uint32_t *__restrict__ level_start)
{
    __shared__ uint32_t wcnt[EXT_THREADS / 64][ORDER_BUCKETS];
    __shared__ uint32_t bin_base[ORDER_BUCKETS];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    // level_start[l] = rows with a binary level < l: the level bins' totals, scanned -- gathered here for the host's one read-back
    if (blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 128) {
        const uint32_t t = bin_total[ORDER_BUCKETS + lane];
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        level_start[lane] = inc - t;
    }
    if (threadIdx.x < 64) {                             // rows of the order buckets before this one (ORDER_BUCKETS <= 64)
        const uint32_t t = lane < ORDER_BUCKETS ? bin_total[lane] : 0u;
        uint32_t inc = t;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t u = __shfl_up(inc, d, 64); if (lane >= d) inc += u; }
        if (lane < ORDER_BUCKETS) bin_base[lane] = inc - t;
    }
    const int64_t i = (int64_t)blockIdx.x * EXT_THREADS + threadIdx.x;
    const bool valid = i < N;
    const uint32_t b = valid ? order_bucket[i] : 0u;
    // lanes of this wave with the same bucket (5 ballots)
    uint64_t same = __ballot(valid);
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const uint64_t bal = __ballot((b >> k) & 1u);
        same &= ((b >> k) & 1u) ? bal : ~bal;
    }
    const uint64_t below = ((uint64_t)1 << lane) - 1;
    const uint32_t rank = (uint32_t)__popcll(same & below);
    if (threadIdx.x < ORDER_BUCKETS) {
#pragma unroll
        for (int w = 0; w < EXT_THREADS / 64; ++w) wcnt[w][threadIdx.x] = 0;
    }
    __syncthreads();
    if (valid && rank == 0) wcnt[wid][b] = (uint32_t)__popcll(same);
    __syncthreads();
    if (valid) {
        uint32_t before = 0;
        for (int w = 0; w < wid; ++w) before += wcnt[w][b];
        const uint32_t pos = bin_base[b] + bucket_pos[(size_t)b * gridDim.x + blockIdx.x] + before + rank;
        order[pos] = (uint32_t)i;
        inv_order[i] = pos;
    }
}

__global__ void order_to_identity_kernel(uint32_t *order, int64_t N)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N) order[k] = (uint32_t)k;
}

__global__ void invert_perm_kernel(const uint32_t *__restrict__ order, int64_t N, uint32_t *__restrict__ inv)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < N) inv[order[k]] = (uint32_t)k;
}

// ---- tile schedule -----------------------------------------------------------------------------
// Entry j of a stage (row r = rows ? rows[j] : j) is merged inside its tile iff the whole subtree
// [r - wl, r + wr) lies inside the tile's row range; otherwise it survives to the next stage.
__global__ void stage_survivor_kernel(const uint32_t *__restrict__ rows, int64_t n, int R, int64_t N,
                                      const int32_t *__restrict__ wl, const int32_t *__restrict__ wr,
                                      const uint8_t *__restrict__ lvl, int top_level,
                                      uint32_t *__restrict__ survivor)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int64_t r = rows ? rows[j] : j;
    const int64_t t = j / R;
    const int64_t j0 = t * R, j1 = j0 + R;
    const int64_t start = rows ? rows[j0] : j0;
    const int64_t end = (j1 < n) ? (rows ? (int64_t)rows[j1] : j1) : N;
    const bool merged = (r > 0) && ((int)lvl[r] < top_level) && (r - wl[r] >= start) && (r + wr[r] <= end);
    survivor[j] = merged ? 0u : 1u;
}

// Channels per chunk (<= max_dc) such that every chunk, the last one included, holds at least one
// whole 16-byte lane chunk (the tile kernel fetches a row's tail as the 16 bytes that end with it).
static int fit_chunk_channels(int elem_size, int D, int max_dc)
{
    const int vn = 16 / elem_size;
    const int hi = std::max(std::min(max_dc, 64), vn);
    if (D <= hi) return D;
    for (int Dc = hi; Dc >= vn; --Dc) {
        const int r = D % Dc;
        if (r == 0 || r >= vn) return Dc;
    }
    for (int Dc = hi + 1; Dc <= 64; ++Dc) {               // nothing that narrow fits: widen
        const int r = D % Dc;
        if (r == 0 || r >= vn) return Dc;
    }
    return std::min(D, 64);                               // not reached (61..64 cover every remainder for D > 64)
}

int pick_chunk_channels(int elem_size, int D)
{
    static const int forced = getenv("RAHT_STAGE0_CH") ? atoi(getenv("RAHT_STAGE0_CH")) : 0;     // tuning knob: channel chunks at stage 0
    if (forced >= 16 / elem_size && forced < D) return fit_chunk_channels(elem_size, D, forced);
    return fit_chunk_channels(elem_size, D, 64);
}

size_t tile_lds_bytes(int R, int elem_size, int Dc, bool ident, bool qm)
{
    // data tile + per-slot butterfly record (16 B float / 24 B double), row id (later stages only),
    // Q position (fused quantization only), flag + histograms (1 KiB) + survivor slot list
    // (R x uint16) + the inverse's survivor prefetch area (12 rows)
    // (must match the carve-up in transform.hip: tile_kernel)
    const int vn = 16 / elem_size;
    const size_t Dp = (size_t)((Dc + vn - 1) / vn) * vn;                 // rows padded to whole 16-byte chunks
    size_t data = (size_t)R * Dp * elem_size;
    size_t meta = (size_t)R * ((elem_size == 4 ? 16 : 24) + (ident ? 0 : 4) + (qm ? 4 : 0) + 1);
    size_t surv = ((size_t)R * 2 + 15) & ~(size_t)15;
    return data + ((meta + 15) & ~(size_t)15) + 1024 + surv + (size_t)12 * Dp * elem_size;
}

int pick_tile_rows(const raht_plan *plan, int elem_size, int Dc)
{
    if (plan->tile_rows_override > 0) return plan->tile_rows_override;
    // Three 512-thread workgroups per CU. gfx950 hands out its 160 KiB of LDS in 128 granules of
    // 1280 bytes, so each workgroup may use 42 granules. Measured best on MI355X for the
    // 59-channel float32 case (R = 192); see DESIGN.md for the sweep.
    const size_t budget = (size_t)42 * 1280;
    for (int R = 512; R >= 64; R -= 8)
        if (tile_lds_bytes(R, elem_size, Dc, true, true) <= budget) return R;
    return 0;
}

void pick_tail_geometry(const raht_plan *plan, int elem_size, int D, int stage0_rows, int *tail_rows, int *tail_chunk,
                        int *final_rows)
{
    // Later stages hold a few % of the rows. Default: the same geometry as stage 0, trimmed so that
    // three workgroups still fit per CU with the slightly larger later-stage LDS layout (row ids).
    // Much larger chunked tiles (e.g. 1024 x 32) cut the number of stages but measured slower on cfg3
    // (one workgroup per CU, no overlap): 1.02 vs 0.955 ms per fused step.
    int Dc = std::min(D, 64), R = stage0_rows;
    if (D > 64) Dc = pick_chunk_channels(elem_size, D);
    if (plan->tail_chunk_override > 0) Dc = fit_chunk_channels(elem_size, D, std::min(plan->tail_chunk_override, std::min(D, 64)));
    if (plan->tail_rows_override > 0) {
        R = plan->tail_rows_override;
        while (R > 64 && tile_lds_bytes(R, elem_size, Dc, false, true) > (size_t)128 * 1280) R -= 64;
    } else {
        while (R > 64 && tile_lds_bytes(R, elem_size, Dc, false, true) > (size_t)42 * 1280) R -= 8;
    }
    *tail_rows = R;
    *tail_chunk = Dc;
    // The top of the tree is latency-bound: once at most this many entries are left, ONE launch
    // (top_kernel: a workgroup per 16-byte channel chunk, all entries in LDS, 16 bytes per entry)
    // finishes the tree. 8192 entries = 128 KiB of the CU's 160 KiB.
    // Default 1536 (rounds 1-2: 4096): one workgroup per chunk touching EVERY entry's row (one 128-byte line per
    // 16 useful bytes, on ceil(D / 4) CUs only) costs 6 us + 7.8 ns per entry (439 entries 9.4 us, 2310 entries 24 us,
    // 7013 entries 41 us), a tile stage 11 us and leaves 1/18 of its entries: above ~1500 entries one more tile stage
    // is cheaper. The reference's own shape (J = 10, ~1 M voxels x 56) leaves 2310 entries after two tile stages:
    // 0.239 -> 0.225 ms per fused step with the third tile stage (r03 sweep, tools/sweep_tail2.sh).
    int Rf = 1536;
    if (plan->final_rows_override > 0) Rf = std::min(plan->final_rows_override, RAHT_TOP_MAX_ROWS);
    *final_rows = Rf;
}

static void free_schedule(Schedule &sc)
{
    // blocks go back to the cache and may be handed out again at once: nothing enqueued may still use
    // them (hipFree used to imply this wait; schedules are only dropped on rare, synchronous paths)
    if (!sc.stages.empty()) (void)hipDeviceSynchronize();
    for (auto &st : sc.stages) {
        if (st.rows) dev_free(st.rows);
        if (st.surv_off) dev_free(st.surv_off);
        if (st.e_wl) dev_free(st.e_wl);
        if (st.e_wr) dev_free(st.e_wr);
        if (st.e_lvl) dev_free(st.e_lvl);
        if (st.e_ht) dev_free(st.e_ht);
        if (st.arrive) dev_free(st.arrive);
        if (st.e_pos) dev_free(st.e_pos);
        if (st.t_pj) dev_free(st.t_pj);
        if (st.t_ab32) dev_free(st.t_ab32);
        if (st.t_ab64) dev_free(st.t_ab64);
        if (st.t_root) dev_free(st.t_root);
        if (st.t_lev) dev_free(st.t_lev);
        if (st.ws) dev_free(st.ws);
    }
    sc.stages.clear();
}

int ensure_workspace(Schedule *sc, size_t row_bytes, bool split)
{
    if (row_bytes <= sc->ws_row_bytes && split == sc->ws_split) return RAHT_OK;
    row_bytes = std::max(row_bytes, sc->ws_row_bytes);
    for (size_t k = 1; k < sc->stages.size(); ++k) {
        Stage &st = sc->stages[k];
        if (st.ws) { (void)hipDeviceSynchronize(); dev_free(st.ws); st.ws = nullptr; }
        const size_t one = (row_bytes * (size_t)st.n_entries + 255) & ~(size_t)255;
        if (dev_malloc(&st.ws, split ? 2 * one : one) != hipSuccess) {
            set_error("workspace allocation failed (%zu bytes)", split ? 2 * one : one);
            sc->ws_row_bytes = 0;
            return RAHT_ERR_NOMEM;
        }
        st.ws_inv_off = split ? one : 0;
    }
    sc->ws_row_bytes = row_bytes;
    sc->ws_split = split;
    return RAHT_OK;
}

__global__ void tile_start_kernel(const uint32_t *__restrict__ pos, int64_t n, int R, int64_t n_tiles,
                                  uint32_t total, uint32_t *__restrict__ surv_off)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t > n_tiles) return;
    surv_off[t] = (t < n_tiles) ? pos[t * R] : total;
    (void)n;
}

__global__ void gather_meta_kernel(const uint32_t *__restrict__ rows, int64_t n, const int32_t *__restrict__ wl,
                                   const int32_t *__restrict__ wr, const uint8_t *__restrict__ lvl,
                                   const uint32_t *__restrict__ inv_order, int32_t *__restrict__ e_wl,
                                   int32_t *__restrict__ e_wr, uint8_t *__restrict__ e_lvl, uint32_t *__restrict__ e_pos)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t r = rows[j];
    e_wl[j] = wl[r]; e_wr[j] = wr[r]; e_lvl[j] = lvl[r]; e_pos[j] = inv_order[r];
}

// ---- butterfly heights of the tile stages -----------------------------------------------------------
// One wave per tile, every tile stage of a schedule in ONE launch. The recurrence is the forward transform's own order:
// walk the binary levels present in the tile upwards; a butterfly (partner slot p, own slot j) gets h = 1 + max(cur[p],
// cur[j]) and leaves it in cur[p], the slot that carries the merged node on. Butterflies of one level touch disjoint slots,
// so a level is: every lane reads the two values of its (<= SPL) butterflies, then writes them -- two LDS round trips. A
// lane keeps its slots' level / partner / height in registers; LDS holds one byte per slot (and the row ids of a later
// stage, for the partner search). What "merged in this tile" means is the tile kernel's own predicate (transform.hip, P1).
constexpr int HT_MAX_ROWS = 1024, HT_MAX_STAGES = 8;
struct HeightStage {
    const uint32_t *rows; const int32_t *wl, *wr; const uint8_t *lvl; uint8_t *ht;
    int64_t n; int R; uint32_t first_tile;
    // launched BEFORE the host knows the stage sizes (build_schedule_fast): the entry count and whether the stage is a tile
    // stage at all come from the schedule builder's device state; first_tile then counts the tiles of the stages' CAPACITIES
    const uint32_t *n_dev, *kind_dev; uint32_t kind_tile;
};
struct HeightArgs { HeightStage st[HT_MAX_STAGES]; int n_stages; uint32_t n_tiles; int64_t N; int top_level; };

template <int SPL>
__global__ __launch_bounds__(64) void tile_heights_kernel(const HeightArgs H)
{
    extern __shared__ __align__(16) unsigned char ht_smem[];
    int k = 0;
#pragma unroll
    for (int q = 1; q < HT_MAX_STAGES; ++q) k += (q < H.n_stages && blockIdx.x >= H.st[q].first_tile) ? 1 : 0;
    PL_STAMP(1, blockIdx.x, 0);
    const HeightStage &S = H.st[k];
    const int R = S.R;
    const int lane = threadIdx.x;
    const int64_t e0 = (int64_t)(blockIdx.x - S.first_tile) * R;
    if (S.kind_dev && *S.kind_dev != S.kind_tile) return;
    const int64_t Sn = S.n_dev ? (int64_t)*S.n_dev : S.n;
    if (e0 >= Sn) return;
    const uint32_t *__restrict__ rows = S.rows;
    const int nt = (int)min((int64_t)R, Sn - e0);
    uint8_t *s_cur = ht_smem;                                   // [R]
    uint32_t *s_row = (uint32_t *)(ht_smem + ((R + 15) & ~15)); // [R], later stages only
    const int64_t start_row = rows ? (int64_t)rows[e0] : e0;
    const int64_t end_row = (e0 + R < Sn) ? (rows ? (int64_t)rows[e0 + R] : e0 + R) : H.N;
    int lv[SPL], part[SPL];
    int32_t wlv[SPL], wrv[SPL];
    int64_t r[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) {                          // all loads first: one round trip
        const int j = lane + s * 64;
        lv[s] = 255; wlv[s] = 0; wrv[s] = 0; r[s] = 0;
        if (j < nt) {
            r[s] = rows ? (int64_t)rows[e0 + j] : e0 + j;
            lv[s] = (int)S.lvl[e0 + j]; wlv[s] = S.wl[e0 + j]; wrv[s] = S.wr[e0 + j];
        }
    }
    if (rows) {
#pragma unroll
        for (int s = 0; s < SPL; ++s) { const int j = lane + s * 64; if (j < nt) s_row[j] = (uint32_t)r[s]; }
    }
#pragma unroll
    for (int s = 0; s < SPL; ++s) { const int j = lane + s * 64; if (j < nt) s_cur[j] = 0; }
    __syncthreads();
    PL_STAMP(1, blockIdx.x, 1);                       // metadata loaded
    uint64_t mask = 0;
#pragma unroll
    for (int s = 0; s < SPL; ++s) {
        const int j = lane + s * 64;
        const bool merged = (j < nt) && (r[s] > 0) && (lv[s] < H.top_level) && (r[s] - wlv[s] >= start_row) && (r[s] + wrv[s] <= end_row);
        part[s] = 0;
        if (merged) {
            if (!rows) part[s] = j - wlv[s];
            else {                                          // the partner row r - wl is an entry of this tile
                const uint32_t want = (uint32_t)(r[s] - wlv[s]);
                int lo = 0, hi = j - 1;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_row[mid] < want) lo = mid + 1; else hi = mid; }
                part[s] = lo;
            }
            mask |= (uint64_t)1 << lv[s];
        } else {
            lv[s] = 255;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)mask, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(mask >> 32), d, 64);
        mask |= (uint64_t)lo | ((uint64_t)hi << 32);
    }
    int ht[SPL];
#pragma unroll
    for (int s = 0; s < SPL; ++s) ht[s] = 0;
    PL_STAMP(1, blockIdx.x, 2);                       // partners resolved, level mask reduced
    PL_NOTE_LEVELS(blockIdx.x, __popcll(mask));
    // (the reduced mask is the same in every lane but lives in vector registers: made scalar, the loop's counter, find-first-set
    // and branch run on the scalar unit -- see level_extent_kernel)
    mask = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mask) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mask >> 32)) << 32);
    while (mask) {
        const int l = __ffsll((unsigned long long)mask) - 1;
        mask &= mask - 1;
        int h[SPL];
#pragma unroll
        for (int s = 0; s < SPL; ++s) {
            h[s] = 0;
            if (lv[s] == l) h[s] = 1 + max((int)s_cur[part[s]], (int)s_cur[lane + s * 64]);
        }
#pragma unroll
        for (int s = 0; s < SPL; ++s)
            if (lv[s] == l) { ht[s] = h[s]; s_cur[part[s]] = (uint8_t)h[s]; }
        __syncthreads();                                    // one wave: orders the LDS traffic of consecutive levels
    }
    PL_STAMP(1, blockIdx.x, 3);                       // levels walked
#pragma unroll
    for (int s = 0; s < SPL; ++s) { const int j = lane + s * 64; if (j < nt) S.ht[e0 + j] = (uint8_t)ht[s]; }
    PL_STAMP(1, blockIdx.x, 4);
}

static void launch_heights_kernel(const HeightArgs &H, int maxR, bool any_rows, hipStream_t s)
{
    const size_t lds = (size_t)((maxR + 15) & ~15) + (any_rows ? (size_t)maxR * 4 : 0);
    const int spl = (maxR + 63) / 64;
    if (spl <= 3) hipLaunchKernelGGL(tile_heights_kernel<3>, dim3(H.n_tiles), dim3(64), lds, s, H);
    else if (spl <= 4) hipLaunchKernelGGL(tile_heights_kernel<4>, dim3(H.n_tiles), dim3(64), lds, s, H);
    else if (spl <= 8) hipLaunchKernelGGL(tile_heights_kernel<8>, dim3(H.n_tiles), dim3(64), lds, s, H);
    else hipLaunchKernelGGL(tile_heights_kernel<16>, dim3(H.n_tiles), dim3(64), lds, s, H);
}

// heights of every tile stage of a finished schedule: one launch (sizes are known on the host by now; enqueued, not
// waited for: the transforms that read them run behind this on the same stream)
static int launch_stage_heights(raht_plan *plan, Schedule &sc, hipStream_t s)
{
    // up to HT_MAX_STAGES tile stages per launch; deeper schedules (deep or unbalanced key sets, small tail_rows / final_rows
    // overrides: get_schedule_exact builds up to plan->max_stages = 24 of them) take several launches
    HeightArgs H;
    int maxR = 0;
    bool any_rows = false;
    auto reset = [&]() { H.n_stages = 0; H.n_tiles = 0; H.N = plan->N; H.top_level = plan->top_level; maxR = 0; any_rows = false; };
    auto flush = [&]() -> int {
        if (H.n_stages == 0) return RAHT_OK;
        for (int q = H.n_stages; q < HT_MAX_STAGES; ++q) H.st[q] = H.st[0];
        launch_heights_kernel(H, maxR, any_rows, s);
        RAHT_HIP_CHECK(hipGetLastError());
        reset();
        return RAHT_OK;
    };
    reset();
    // (testing aid, read per schedule: RAHT_HEIGHT_STAGES_PER_LAUNCH=2 walks the several-launches path on ordinary scenes)
    const char *ge = getenv("RAHT_HEIGHT_STAGES_PER_LAUNCH");
    const int group = ge ? std::min(std::max(atoi(ge), 1), HT_MAX_STAGES) : HT_MAX_STAGES;
    for (size_t k = 0; k < sc.stages.size(); ++k) {
        Stage &st = sc.stages[k];
        if (st.is_top || st.n_entries < 1) continue;
        if (st.tile_rows > HT_MAX_ROWS) { set_error("tile heights: %d rows per tile not supported", st.tile_rows); return RAHT_ERR_UNSUPPORTED; }
        if (!st.e_ht) RAHT_HIP_CHECK(dev_malloc(&st.e_ht, (size_t)st.n_entries));
        HeightStage &h = H.st[H.n_stages++];
        h.rows = st.rows; h.wl = st.rows ? st.e_wl : plan->wl; h.wr = st.rows ? st.e_wr : plan->wr; h.lvl = st.rows ? st.e_lvl : plan->lvl;
        h.ht = st.e_ht; h.n = st.n_entries; h.R = st.tile_rows; h.first_tile = H.n_tiles;
        h.n_dev = nullptr; h.kind_dev = nullptr; h.kind_tile = 0;
        H.n_tiles += (uint32_t)st.n_tiles;
        maxR = std::max(maxR, st.tile_rows);
        any_rows = any_rows || st.rows != nullptr;
        if (H.n_stages == group) RAHT_RET(flush());
    }
    return flush();
}

// ---- TOP stage: every butterfly still to do, resolved against the stage's entry list ----------------
__global__ void top_resolve_kernel(const uint32_t *__restrict__ rows, int64_t n, const int32_t *__restrict__ wl,
                                   const int32_t *__restrict__ wr, const uint8_t *__restrict__ lvl,
                                   const int64_t *__restrict__ wsum, int top_level, uint32_t *__restrict__ pj,
                                   double *__restrict__ ab, uint8_t *__restrict__ bucket, uint32_t *__restrict__ is_root)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int64_t r = rows ? (int64_t)rows[e] : e;
    const int l = (int)lvl[r];
    const bool merged = (r > 0) && (l < top_level);
    is_root[e] = merged ? 0u : 1u;
    bucket[e] = merged ? (uint8_t)l : (uint8_t)63;       // roots sort behind every butterfly (levels are <= 62)
    uint32_t rec = 0;
    double a = 0.0, b = 0.0;
    if (merged) {
        const int64_t want = r - wl[r];                  // the partner row is an entry of this stage as well
        int64_t p = want;
        if (rows) {
            int64_t lo = 0, hi = e - 1;
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if ((int64_t)rows[mid] < want) lo = mid + 1; else hi = mid;
            }
            p = lo;
        }
        double w0, w1;
        pair_weights(r, wl[r], wr[r], wsum, w0, w1);
        const double den = w0 + w1;
        a = sqrt(w0 / den);                              // RAHT.py:321-322
        b = sqrt(w1 / den);
        rec = (uint32_t)p | ((uint32_t)e << 16);
    }
    pj[e] = rec; ab[2 * e] = a; ab[2 * e + 1] = b;
}

__global__ void top_gather_kernel(const uint32_t *__restrict__ perm, uint32_t n_merges, const uint32_t *__restrict__ pj,
                                  const double *__restrict__ ab, uint32_t *__restrict__ t_pj,
                                  float *__restrict__ t_ab32, double *__restrict__ t_ab64)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_merges) return;
    const uint32_t e = perm[k];
    t_pj[k] = pj[e];
    const double a = ab[2 * e], b = ab[2 * e + 1];
    t_ab64[2 * k] = a; t_ab64[2 * k + 1] = b;
    t_ab32[2 * k] = (float)a; t_ab32[2 * k + 1] = (float)b;
}

__global__ void top_root_rank_kernel(const uint32_t *__restrict__ is_root, const uint32_t *__restrict__ pos, int64_t n,
                                     uint32_t *__restrict__ t_root)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) t_root[e] = is_root[e] ? pos[e] : 0xffffffffu;
}

static int build_top_stage(raht_plan *plan, uint32_t *rows, int64_t n, hipStream_t s, Stage &st)
{
    st.is_top = true;
    st.n_entries = n;
    st.n_tiles = 1;
    st.rows = rows;
    st.tile_rows = (int)n;
    const unsigned gb = (unsigned)ceil_div(n, 256);
    if (rows) {
        RAHT_HIP_CHECK(dev_malloc(&st.e_wl, sizeof(int32_t) * (size_t)n));
        RAHT_HIP_CHECK(dev_malloc(&st.e_wr, sizeof(int32_t) * (size_t)n));
        RAHT_HIP_CHECK(dev_malloc(&st.e_lvl, (size_t)n));
        RAHT_HIP_CHECK(dev_malloc(&st.e_pos, sizeof(uint32_t) * (size_t)n));
        hipLaunchKernelGGL(gather_meta_kernel, dim3(gb), dim3(256), 0, s, rows, n, plan->wl, plan->wr, plan->lvl,
                           plan->inv_order, st.e_wl, st.e_wr, st.e_lvl, st.e_pos);
    }
    // scratch: pj | is_root | pos | perm | boff[65] | total | ab (double, 8-byte aligned first) | bucket
    Scratch buf(sizeof(double) * 2 * (size_t)n + sizeof(uint32_t) * (4 * (size_t)n + 66) + (size_t)n, s);
    if (!buf.ok()) return RAHT_ERR_NOMEM;
    double *ab = buf.as<double>();
    uint32_t *pj = (uint32_t *)(ab + 2 * n), *is_root = pj + n, *pos = is_root + n, *perm = pos + n, *boff = perm + n,
             *total = boff + 65;
    uint8_t *bucket = (uint8_t *)(total + 1);
    hipLaunchKernelGGL(top_resolve_kernel, dim3(gb), dim3(256), 0, s, rows, n, plan->wl, plan->wr, plan->lvl, plan->wsum,
                       plan->top_level, pj, ab, bucket, is_root);
    RAHT_RET(exclusive_scan_u32(is_root, pos, n, total, s));
    RAHT_HIP_CHECK(dev_malloc(&st.t_root, sizeof(uint32_t) * (size_t)n));
    hipLaunchKernelGGL(top_root_rank_kernel, dim3(gb), dim3(256), 0, s, is_root, pos, n, st.t_root);
    RAHT_RET(bucket_sort_u8(bucket, perm, n, 6, boff, s));
    RAHT_RET(read_back_u32(st.t_loff, boff, 65, nullptr, nullptr, 0, s));
    st.n_merges = st.t_loff[63];
    {
        // the level program: non-empty levels, ascending; the trailing run of levels with at most 64
        // butterflies each (the top of the tree) is chained by one wave
        uint32_t *lev = st.t_lev_host;                    // lives in the stage: the upload below stays asynchronous
        int nlev = 0;
        for (int l = 0; l < 63; ++l)
            if (st.t_loff[l + 1] > st.t_loff[l]) { lev[2 * nlev] = st.t_loff[l]; lev[2 * nlev + 1] = st.t_loff[l + 1]; ++nlev; }
        int nbig = nlev;
        while (nbig > 0 && lev[2 * (nbig - 1) + 1] - lev[2 * (nbig - 1)] <= 64) --nbig;
        // the chained records live in LDS next to the entries (16 B each + 12 / 20 B per record)
        const size_t lds_budget = 160 * 1024 - 1024;
        while (nbig < nlev && (size_t)n * 16 + (size_t)(st.n_merges - lev[2 * nbig]) * 20 > lds_budget) ++nbig;
        st.t_nlev = nlev; st.t_nbig = nbig;
        st.t_small_start = (nbig < nlev) ? lev[2 * nbig] : st.n_merges;
        RAHT_HIP_CHECK(dev_malloc(&st.t_lev, sizeof(uint32_t) * 2 * 64));
        // staged through a scratch-independent pageable copy: hipMemcpyAsync from pageable memory copies the
        // source before it returns, so `st` may be moved afterwards
        RAHT_HIP_CHECK(hipMemcpyAsync(st.t_lev, lev, sizeof(uint32_t) * 2 * (size_t)std::max(nlev, 1), hipMemcpyHostToDevice, s));
    }
    const size_t nm = std::max<size_t>(st.n_merges, 1);
    RAHT_HIP_CHECK(dev_malloc(&st.t_pj, sizeof(uint32_t) * nm));
    RAHT_HIP_CHECK(dev_malloc(&st.t_ab32, sizeof(float) * 2 * nm));
    RAHT_HIP_CHECK(dev_malloc(&st.t_ab64, sizeof(double) * 2 * nm));
    if (st.n_merges)
        hipLaunchKernelGGL(top_gather_kernel, dim3((unsigned)ceil_div(st.n_merges, 256)), dim3(256), 0, s, perm, st.n_merges,
                           pj, ab, st.t_pj, st.t_ab32, st.t_ab64);
    return RAHT_OK;                                  // (the scratch goes back to the pool: stream-ordered reuse)
}

// ---- schedule build, device-driven ----------------------------------------------------------------
// The exact builder below (get_schedule_exact) reads every stage's size back to the host before it can size
// and launch the next stage: 4-5 round trips of ~20 us each, during which the GPU idles -- more than half of
// a cfg3 plan build. Here the chain of stages runs on the device: every stage is two launches
// (sched_count_kernel: survivor flags + per-block counts; sched_emit_kernel: the next stage's entry list,
// its entry-ordered plan metadata and this stage's per-tile survivor offsets in one pass), the TOP stage is
// ONE single-workgroup launch (sched_top_kernel), each kernel decides from the device-resident SchedState
// whether it has anything to do, buffers are sized from generous bounds (a stage keeps < 1/3 of its entries:
// measured 1/20 at 184 rows per tile, 1/6 at 64), and ONE read-back at the end tells the host how it went.
// Anything unusual (a bound exceeded, more stages than were enqueued, no progress) -> the exact builder.
constexpr int SB_THREADS = 256, SB_ITEMS = 8, SB_BLOCK = SB_THREADS * SB_ITEMS;
constexpr int SCHED_SPEC_MAX = 8;                  // stages enqueued speculatively, at most
enum { SK_NONE = 0, SK_TILE = 1, SK_TOP = 2 };

struct SchedState {
    uint32_t n[SCHED_SPEC_MAX + 2];                // entries of stage k
    uint32_t kind[SCHED_SPEC_MAX + 2];             // what stage k is (written by the stage before it)
    uint32_t finished;                             // the tree is done: last_stage / last_is_top are valid
    uint32_t last_stage, last_is_top;
    uint32_t trouble;                              // 1 = a buffer bound was exceeded, 2 = a stage made no progress
    uint32_t top[4];                               // TOP stage: n_merges, nlev, nbig, small_start
};
constexpr int SCHED_STATE_WORDS = sizeof(SchedState) / 4;

__device__ __forceinline__ uint32_t block_sum_256(uint32_t v, uint32_t *red /* [4] */)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    const uint32_t t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// wl / wr / lvl are ENTRY-ordered for this stage (stage 0: the plan arrays, entry = row)
// n_first >= 0: this is the FIRST launch of the chain (stage 0 as a tile stage of n_first entries): nothing has
// written the state yet -- block 0 does (the launches behind this one read it), every block takes n from the argument
__device__ __forceinline__ void sched_state_init(SchedState *S, uint32_t n0, uint32_t kind0)
{
    uint32_t *w = (uint32_t *)S;
    for (int i = threadIdx.x; i < SCHED_STATE_WORDS; i += blockDim.x) w[i] = 0;
    __syncthreads();
    if (threadIdx.x == 0) { S->n[0] = n0; S->kind[0] = kind0; }
}

__global__ __launch_bounds__(SB_THREADS) void sched_count_kernel(SchedState *__restrict__ S, int k, const uint32_t *__restrict__ rows,
                                                                 const int32_t *__restrict__ wl, const int32_t *__restrict__ wr,
                                                                 const uint8_t *__restrict__ lvl, int R, int64_t N, int top_level,
                                                                 uint8_t *__restrict__ flags, uint32_t *__restrict__ blk_cnt, int64_t n_first)
{
    __shared__ uint32_t red[4];
    if (n_first >= 0) { if (blockIdx.x == 0) sched_state_init(S, (uint32_t)n_first, SK_TILE); }
    else if (S->kind[k] != SK_TILE) return;
    const int64_t n = n_first >= 0 ? n_first : (int64_t)S->n[k];
    const int64_t base = (int64_t)blockIdx.x * SB_BLOCK + (int64_t)threadIdx.x * SB_ITEMS;
    if ((int64_t)blockIdx.x * SB_BLOCK >= n) return;
    uint32_t cnt = 0;
    // tile bounds by 32-bit arithmetic, carried along the thread's 8 consecutive entries (a 64-bit division per
    // entry was most of this kernel's time)
    uint32_t j0 = (uint32_t)base / (uint32_t)R * (uint32_t)R;
    int64_t start = 0, end = 0;
    bool fresh = true;
    // the thread's 8 consecutive entries in a few wide loads (32 / 32 / 8 / 32 bytes) instead of 8 x 4 narrow ones: the launch is
    // bound by its load instructions, not by the 27 MB it reads on cfg3 (20 -> ~12 us)
    int32_t v_wl[SB_ITEMS], v_wr[SB_ITEMS];
    uint32_t v_row[SB_ITEMS];
    uint8_t v_lv[SB_ITEMS];
    if (base + SB_ITEMS <= n) {
        const int4 a0 = *(const int4 *)(wl + base), a1 = *(const int4 *)(wl + base + 4);
        const int4 b0 = *(const int4 *)(wr + base), b1 = *(const int4 *)(wr + base + 4);
        const uint2 l8 = *(const uint2 *)(lvl + base);
        v_wl[0] = a0.x; v_wl[1] = a0.y; v_wl[2] = a0.z; v_wl[3] = a0.w; v_wl[4] = a1.x; v_wl[5] = a1.y; v_wl[6] = a1.z; v_wl[7] = a1.w;
        v_wr[0] = b0.x; v_wr[1] = b0.y; v_wr[2] = b0.z; v_wr[3] = b0.w; v_wr[4] = b1.x; v_wr[5] = b1.y; v_wr[6] = b1.z; v_wr[7] = b1.w;
#pragma unroll
        for (int q = 0; q < 4; ++q) { v_lv[q] = (uint8_t)(l8.x >> (8 * q)); v_lv[4 + q] = (uint8_t)(l8.y >> (8 * q)); }
        if (rows) {
            const uint4 r0 = *(const uint4 *)(rows + base), r1 = *(const uint4 *)(rows + base + 4);
            v_row[0] = r0.x; v_row[1] = r0.y; v_row[2] = r0.z; v_row[3] = r0.w; v_row[4] = r1.x; v_row[5] = r1.y; v_row[6] = r1.z; v_row[7] = r1.w;
        }
    } else {
#pragma unroll
        for (int q = 0; q < SB_ITEMS; ++q) {
            const int64_t j = min(base + q, n - 1);
            v_wl[q] = wl[j]; v_wr[q] = wr[j]; v_lv[q] = lvl[j]; v_row[q] = rows ? rows[j] : 0u;
        }
    }
    uint8_t fl[SB_ITEMS];
#pragma unroll
    for (int q = 0; q < SB_ITEMS; ++q) {
        const int64_t j = base + q;
        fl[q] = 0;
        if (j < n) {
            if ((uint32_t)j >= j0 + (uint32_t)R) { j0 += (uint32_t)R; fresh = true; }
            if (fresh) {
                const int64_t j1 = (int64_t)j0 + R;
                start = rows ? (int64_t)rows[j0] : (int64_t)j0;
                end = (j1 < n) ? (rows ? (int64_t)rows[j1] : j1) : N;
                fresh = false;
            }
            const int64_t r = rows ? (int64_t)v_row[q] : j;
            const bool merged = (r > 0) && ((int)v_lv[q] < top_level) && (r - v_wl[q] >= start) && (r + v_wr[q] <= end);
            fl[q] = merged ? 0 : 1;
            cnt += merged ? 0u : 1u;
        }
    }
    if (base + SB_ITEMS <= n) {
        uint2 f8;
        f8.x = (uint32_t)fl[0] | ((uint32_t)fl[1] << 8) | ((uint32_t)fl[2] << 16) | ((uint32_t)fl[3] << 24);
        f8.y = (uint32_t)fl[4] | ((uint32_t)fl[5] << 8) | ((uint32_t)fl[6] << 16) | ((uint32_t)fl[7] << 24);
        *(uint2 *)(flags + base) = f8;
    } else {
#pragma unroll
        for (int q = 0; q < SB_ITEMS; ++q) if (base + q < n) flags[base + q] = fl[q];
    }
    const uint32_t tot = block_sum_256(cnt, red);
    if (threadIdx.x == 0) blk_cnt[blockIdx.x] = tot;
}

// p_* are the PLAN arrays (row-indexed): the next stage's entry-ordered copies are gathered from them here
__global__ __launch_bounds__(SB_THREADS) void sched_emit_kernel(SchedState *__restrict__ S, int k, const uint32_t *__restrict__ rows,
                                                                const uint8_t *__restrict__ flags, const uint32_t *__restrict__ blk_cnt,
                                                                int R, uint32_t Rf, uint32_t n_roots,
                                                                const int32_t *__restrict__ p_wl, const int32_t *__restrict__ p_wr,
                                                                const uint8_t *__restrict__ p_lvl, const uint32_t *__restrict__ p_inv,
                                                                uint32_t *__restrict__ n_rows, int32_t *__restrict__ n_wl, int32_t *__restrict__ n_wr,
                                                                uint8_t *__restrict__ n_lvl, uint32_t *__restrict__ n_pos, uint32_t cap_next,
                                                                uint32_t *__restrict__ surv_off)
{
    __shared__ uint32_t red[4];
    __shared__ uint32_t wsum[4];
    if (S->kind[k] != SK_TILE) return;
    const int64_t n = S->n[k];
    const int64_t nblk = (n + SB_BLOCK - 1) / SB_BLOCK;
    if ((int64_t)blockIdx.x >= nblk) return;
    // survivors in the blocks before this one (<= a few thousand words from L2)
    uint32_t part = 0;
    for (int64_t b = threadIdx.x; b < (int64_t)blockIdx.x; b += SB_THREADS) part += blk_cnt[b];
    const uint32_t block_base = block_sum_256(part, red);
    // exclusive scan of this block's flags (8 consecutive entries per thread)
    const int64_t base = (int64_t)blockIdx.x * SB_BLOCK + (int64_t)threadIdx.x * SB_ITEMS;
    uint8_t f[SB_ITEMS];
    uint32_t mine = 0;
    if (base + SB_ITEMS <= n) {                             // (8 flags in one load: the buffer is 16-byte aligned)
        const uint2 f8 = *(const uint2 *)(flags + base);
#pragma unroll
        for (int q = 0; q < 4; ++q) { f[q] = (uint8_t)(f8.x >> (8 * q)); f[4 + q] = (uint8_t)(f8.y >> (8 * q)); }
#pragma unroll
        for (int q = 0; q < SB_ITEMS; ++q) mine += f[q];
    } else {
#pragma unroll
        for (int q = 0; q < SB_ITEMS; ++q) { f[q] = (base + q < n) ? flags[base + q] : 0; mine += f[q]; }
    }
    uint32_t inc = mine;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t before = 0, block_tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wid) before += wsum[w]; block_tot += wsum[w]; }
    uint32_t pos = block_base + before + inc - mine;
    uint32_t tile = (uint32_t)base / (uint32_t)R;
    uint32_t next_start = tile * (uint32_t)R;               // first tile boundary at or after `base`
    if (next_start < (uint32_t)base) { ++tile; next_start += (uint32_t)R; }
    // three loops -- the survivors' rows, their plan entries, the stores -- so that every load of a phase is in flight before the
    // first one is used (one loop made each survivor two dependent round trips of its own: see sched_tail_kernel)
    uint32_t g_r[SB_ITEMS], g_pos[SB_ITEMS], g_inv[SB_ITEMS];
    int32_t g_wl[SB_ITEMS], g_wr[SB_ITEMS];
    uint8_t g_lv[SB_ITEMS];
    bool keep[SB_ITEMS];
#pragma unroll
    for (int q = 0; q < SB_ITEMS; ++q) {
        const int64_t j = base + q;
        keep[q] = false; g_pos[q] = 0; g_r[q] = (uint32_t)j;
        if (j < n) {
            if ((uint32_t)j == next_start) { surv_off[tile] = pos; ++tile; next_start += (uint32_t)R; }   // first survivor of the tile
            if (f[q]) {
                keep[q] = pos < cap_next;
                g_pos[q] = pos;
                if (keep[q] && rows) g_r[q] = rows[j];
                ++pos;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < SB_ITEMS; ++q) {
        g_wl[q] = 0; g_wr[q] = 0; g_lv[q] = 0; g_inv[q] = 0;
        if (keep[q]) { const uint32_t r = g_r[q]; g_wl[q] = p_wl[r]; g_wr[q] = p_wr[r]; g_lv[q] = p_lvl[r]; g_inv[q] = p_inv[r]; }
    }
#pragma unroll
    for (int q = 0; q < SB_ITEMS; ++q)
        if (keep[q]) { const uint32_t o = g_pos[q]; n_rows[o] = g_r[q]; n_wl[o] = g_wl[q]; n_wr[o] = g_wr[q]; n_lvl[o] = g_lv[q]; n_pos[o] = g_inv[q]; }
    if ((int64_t)blockIdx.x == nblk - 1 && threadIdx.x == 0) {
        const uint32_t total = block_base + block_tot;
        surv_off[(n + R - 1) / R] = total;
        S->n[k + 1] = total;
        if (total > cap_next) S->trouble = 1;                                                     // (and the chain stops: kind[k + 1] stays SK_NONE)
        else if (total == n_roots) { S->finished = 1; S->last_stage = (uint32_t)k; S->last_is_top = 0; }   // only the roots are left
        else if (total >= (uint32_t)n) S->trouble = 2;                                              // no progress
        else S->kind[k + 1] = (total <= Rf) ? SK_TOP : SK_TILE;
    }
}

// The TOP stage in one workgroup: every butterfly still to do, resolved against the stage's entry list, bucketed by
// level; root ranks; the level program (what build_top_stage does with a dozen launches and a read-back).
constexpr int ST_THREADS = 1024;
struct SchedTopOut { uint32_t *t_pj; float *t_ab32; double *t_ab64; uint32_t *t_root; uint32_t *t_lev; };

__device__ __forceinline__ void sched_top_body(SchedState *__restrict__ S, int k, int n, const uint32_t *__restrict__ rows,
                                               const int32_t *__restrict__ p_wl, const int32_t *__restrict__ p_wr,
                                               const uint8_t *__restrict__ p_lvl, const int64_t *__restrict__ wsum,
                                               int top_level, const SchedTopOut &O)
{
    __shared__ uint32_t s_rows[RAHT_TOP_MAX_ROWS];
    __shared__ uint32_t hist[64], cursor[64], wtot[ST_THREADS / 64];
    __shared__ uint32_t root_base;
    uint32_t *__restrict__ t_pj = O.t_pj, *__restrict__ t_root = O.t_root, *__restrict__ t_lev = O.t_lev;
    float *__restrict__ t_ab32 = O.t_ab32;
    double *__restrict__ t_ab64 = O.t_ab64;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid < 64) hist[tid] = 0;
    if (tid == 0) root_base = 0;
    for (int e = tid; e < n; e += ST_THREADS) s_rows[e] = rows ? rows[e] : (uint32_t)e;
    __syncthreads();
    PL_STAMP(2, 0, 8);                                              // top: entry rows in LDS
    // pass 1: level histogram of the butterflies; root ranks in entry order
    for (int e0 = 0; e0 < n; e0 += ST_THREADS) {
        const int e = e0 + tid;
        bool root = false;
        if (e < n) {
            const uint32_t r = s_rows[e];
            const int l = (int)p_lvl[r];
            const bool merged = (r > 0) && (l < top_level);
            root = !merged;
            if (merged) atomicAdd(&hist[l], 1u);
        }
        const uint64_t bal = __ballot(root);
        if (lane == 0) wtot[wid] = (uint32_t)__popcll(bal);
        __syncthreads();
        uint32_t before = root_base;
        for (int w = 0; w < wid; ++w) before += wtot[w];
        if (e < n) t_root[e] = root ? before + (uint32_t)__popcll(bal & (((uint64_t)1 << lane) - 1)) : 0xffffffffu;
        __syncthreads();
        if (tid == 0) { uint32_t t = 0; for (int w = 0; w < ST_THREADS / 64; ++w) t += wtot[w]; root_base += t; }
        __syncthreads();
    }
    PL_STAMP(2, 0, 9);                                              // top: pass 1 done
    // level offsets + the level program: wave 0, lane l = binary level l
    if (wid == 0) {
        const uint32_t h = (lane < 63) ? hist[lane] : 0u;
        uint32_t inc = h;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
        const uint32_t first = inc - h;
        cursor[lane] = first;
        const uint32_t n_merges = (uint32_t)__shfl((int)inc, 63, 64);
        const uint64_t nonempty = __ballot(h > 0), below = ((uint64_t)1 << lane) - 1;
        const int nlev = __popcll(nonempty);
        const int my = __popcll(nonempty & below);              // index of this level among the non-empty ones
        if (h > 0) { t_lev[2 * my] = first; t_lev[2 * my + 1] = first + h; }
        // the trailing run of levels with <= 64 butterflies each is chained by one wave; its records live in LDS next
        // to the entries (16 B per entry + 12 / 20 B per record; 160 KiB - 1 KiB): walk it down from the top while it fits
        const uint64_t big = __ballot(h > 64);
        int nbig = big ? __popcll(nonempty & (((uint64_t)2 << (63 - __clzll((long long)big))) - 1)) : 0;
        const size_t lds_budget = 160 * 1024 - 1024;
        uint32_t small_start = n_merges;
        // first butterfly of the non-empty level with index q: broadcast from the lane that owns it
        for (;;) {
            uint32_t cand = n_merges;
            if (nbig < nlev) {
                const uint64_t owner = __ballot(h > 0 && my == nbig);
                cand = (uint32_t)__shfl((int)first, __ffsll((unsigned long long)owner) - 1, 64);
            }
            if (nbig < nlev && (size_t)n * 16 + (size_t)(n_merges - cand) * 20 > lds_budget) { ++nbig; continue; }
            small_start = cand;
            break;
        }
        if (lane == 0) {
            S->top[0] = n_merges; S->top[1] = (uint32_t)nlev; S->top[2] = (uint32_t)nbig; S->top[3] = small_start;
            S->finished = 1; S->last_stage = (uint32_t)k; S->last_is_top = 1;
        }
    }
    __syncthreads();
    PL_STAMP(2, 0, 10);                                             // top: level program written
    // pass 2: resolve and place every butterfly (any order inside a level: they are independent)
    for (int e = tid; e < n; e += ST_THREADS) {
        const uint32_t r = s_rows[e];
        const int l = (int)p_lvl[r];
        if (!((r > 0) && (l < top_level))) continue;
        const int32_t wlr = p_wl[r], wrr = p_wr[r];
        const uint32_t want = r - (uint32_t)wlr;            // the partner row is an entry of this stage as well
        int lo = 0, hi = e - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (s_rows[mid] < want) lo = mid + 1; else hi = mid;
        }
        double w0, w1;
        pair_weights((int64_t)r, wlr, wrr, wsum, w0, w1);
        const double den = w0 + w1;
        const double a = sqrt(w0 / den), b = sqrt(w1 / den);        // RAHT.py:321-322
        const uint32_t pos = atomicAdd(&cursor[l], 1u);
        t_pj[pos] = (uint32_t)lo | ((uint32_t)e << 16);
        t_ab64[2 * pos] = a; t_ab64[2 * pos + 1] = b;
        t_ab32[2 * pos] = (float)a; t_ab32[2 * pos + 1] = (float)b;
    }
    PL_STAMP(2, 0, 11);                                             // top: pass 2 issued
}

// The END of the chain in ONE launch of one workgroup: from stage k0 on, every tile stage of at most `tail_max`
// entries (flags, survivor scan, the next stage's entry list and this stage's per-tile survivor offsets: what
// sched_count_kernel + sched_emit_kernel do for the large stages) and the TOP stage. On cfg3 that is stage 2
// (7 013 entries) and the top stage (439): two launches of work instead of the eight (two of them empty) that the
// launch-per-step chain enqueued -- back-to-back launches cost 4.6 us each even when they have nothing to do.
struct SchedStageBufs {                         // buffers of stage k (entry-ordered copies; surv: its per-tile survivor offsets)
    uint32_t *rows[SCHED_SPEC_MAX + 2]; int32_t *wl[SCHED_SPEC_MAX + 2], *wr[SCHED_SPEC_MAX + 2];
    uint8_t *lvl[SCHED_SPEC_MAX + 2]; uint32_t *pos[SCHED_SPEC_MAX + 2], *surv[SCHED_SPEC_MAX + 2];
    uint32_t cap[SCHED_SPEC_MAX + 2];
};

__global__ __launch_bounds__(ST_THREADS) void sched_tail_kernel(SchedState *S, int k_multi, int k_last, SchedStageBufs B, int R, uint32_t Rf,
                                                                uint32_t n_roots, int64_t N, int top_level, uint32_t tail_max,
                                                                const int32_t *__restrict__ p_wl, const int32_t *__restrict__ p_wr,
                                                                const uint8_t *__restrict__ p_lvl, const uint32_t *__restrict__ p_inv,
                                                                const int64_t *__restrict__ wsum, SchedTopOut O, int64_t n_first)
{
    __shared__ uint32_t wcnt[8 * (ST_THREADS / 64)], woff[8 * (ST_THREADS / 64) + 1];
    __shared__ uint32_t s_kind[SCHED_SPEC_MAX + 2], s_n[SCHED_SPEC_MAX + 2];   // the chain's state: ONE read at the start, then kept here
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    PL_SLOT_DECL;
    PL_STAMP_NEXT();                                                        // [0] start
    if (n_first >= 0) { sched_state_init(S, (uint32_t)n_first, SK_TOP); __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __syncthreads(); }   // (a tree that fits the top stage)
    // stages 1 .. k_multi - 1 had the multi-workgroup launches if they were tile stages; one of them may have turned out
    // to be the top stage already (a tree that shrank faster than expected)
    // (what this workgroup itself decides about stage k + 1 goes to the state in memory AND into these copies: a global read
    // per stage was a ~2 us round trip on a chip that runs nothing else)
    if (tid < SCHED_SPEC_MAX + 2) {
        s_kind[tid] = (n_first >= 0) ? (tid == 0 ? (uint32_t)SK_TOP : (uint32_t)SK_NONE) : __atomic_load_n(&S->kind[tid], __ATOMIC_RELAXED);
        s_n[tid] = (n_first >= 0) ? (tid == 0 ? (uint32_t)n_first : 0u) : __atomic_load_n(&S->n[tid], __ATOMIC_RELAXED);
    }
    __syncthreads();
    for (int k = (k_multi == 0) ? 0 : 1; k <= k_last; ++k) {
        const uint32_t kind = s_kind[k], n = s_n[k];
        PL_STAMP_NEXT();                                                    // state of stage k read
        if (kind == SK_TOP) {
            sched_top_body(S, k, (int)n, B.rows[k], p_wl, p_wr, p_lvl, wsum, top_level, O);
            return;
        }
        if (kind == SK_TILE && k < k_multi) continue;                       // done by its own launches
        if (kind != SK_TILE || n > tail_max || n > B.cap[k] || k == k_last) return;   // nothing left / larger than expected: exact builder
        const uint32_t *__restrict__ rows = B.rows[k];
        const int32_t *__restrict__ wl = B.wl[k], *__restrict__ wr = B.wr[k];
        const uint8_t *__restrict__ lvl = B.lvl[k];
        uint32_t *__restrict__ surv_off = B.surv[k];
        uint32_t *__restrict__ n_rows = B.rows[k + 1], *__restrict__ n_pos = B.pos[k + 1];
        int32_t *__restrict__ n_wl = B.wl[k + 1], *__restrict__ n_wr = B.wr[k + 1];
        uint8_t *__restrict__ n_lvl = B.lvl[k + 1];
        const uint32_t cap_next = B.cap[k + 1];
        uint32_t running = 0;                                               // survivors in front of this pass (uniform)
        // TI x 1024 entries per pass (entry j = base + q * 1024 + tid): the loads of a pass are independent of each
        // other, so a 7 013-entry stage is ONE round of dependent L2 round trips instead of seven
        constexpr int TI = 8, NWV = ST_THREADS / 64;
        for (uint32_t base = 0; base < n; base += ST_THREADS * TI) {
            // every load of a pass is issued before the first one is used (two loops): with the survivor test in the loading
            // loop the compiler waited per iteration -- eight dependent round trips of ~2 us for one 7 013-entry stage
            uint32_t r[TI], e_start[TI], e_end[TI];
            int32_t e_wl[TI], e_wr[TI];
            uint8_t e_lv[TI];
            uint64_t bal[TI];
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                const uint32_t j = min(base + (uint32_t)(q * ST_THREADS + tid), n - 1);
                const uint32_t j0 = j / (uint32_t)R * (uint32_t)R, j1 = j0 + (uint32_t)R;
                r[q] = rows[j]; e_lv[q] = lvl[j]; e_wl[q] = wl[j]; e_wr[q] = wr[j];
                e_start[q] = rows[j0];
                e_end[q] = rows[min(j1, n - 1)];
            }
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                const uint32_t j = base + (uint32_t)(q * ST_THREADS + tid);
                bool surv = false;
                if (j < n) {
                    const uint32_t j1 = j / (uint32_t)R * (uint32_t)R + (uint32_t)R;
                    const int64_t start = (int64_t)e_start[q], end = (j1 < n) ? (int64_t)e_end[q] : N;
                    const bool merged = (r[q] > 0) && ((int)e_lv[q] < top_level) && ((int64_t)r[q] - e_wl[q] >= start) && ((int64_t)r[q] + e_wr[q] <= end);
                    surv = !merged;
                }
                bal[q] = __ballot(surv);
                if (lane == 0) wcnt[q * NWV + wid] = (uint32_t)__popcll(bal[q]);
            }
            __syncthreads();
            if (wid == 0) {                                                 // exclusive offsets of the TI * 16 (q, wave) counts
                const uint32_t c0 = wcnt[2 * lane], c1 = wcnt[2 * lane + 1];
                uint32_t inc = c0 + c1;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) { const uint32_t t = __shfl_up(inc, d, 64); if (lane >= d) inc += t; }
                woff[2 * lane] = inc - c0 - c1;
                woff[2 * lane + 1] = inc - c1;
                if (lane == 63) woff[TI * NWV] = inc;
            }
            // the survivors' plan rows: gathered (all in flight) while wave 0 scans, stored once the offsets are known
            int32_t g_wl[TI], g_wr[TI];
            uint32_t g_inv[TI];
            uint8_t g_lv[TI];
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                g_wl[q] = 0; g_wr[q] = 0; g_inv[q] = 0; g_lv[q] = 0;
                if ((bal[q] >> lane) & 1) { const uint32_t rr = r[q]; g_wl[q] = p_wl[rr]; g_wr[q] = p_wr[rr]; g_lv[q] = p_lvl[rr]; g_inv[q] = p_inv[rr]; }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < TI; ++q) {
                const uint32_t j = base + (uint32_t)(q * ST_THREADS + tid);
                if (j < n) {
                    const uint32_t pos = running + woff[q * NWV + wid] + (uint32_t)__popcll(bal[q] & (((uint64_t)1 << lane) - 1));
                    if (j % (uint32_t)R == 0) surv_off[j / (uint32_t)R] = pos;     // first survivor of the tile
                    if (((bal[q] >> lane) & 1) && pos < cap_next) {
                        n_rows[pos] = r[q]; n_wl[pos] = g_wl[q]; n_wr[pos] = g_wr[q]; n_lvl[pos] = g_lv[q]; n_pos[pos] = g_inv[q];
                    }
                }
            }
            running += woff[TI * NWV];
            __syncthreads();
        }
        if (tid == 0) {
            const uint32_t total = running;
            surv_off[(n + (uint32_t)R - 1) / (uint32_t)R] = total;
            S->n[k + 1] = total;
            s_n[k + 1] = total;
            if (total > cap_next) S->trouble = 1;                                                      // (and the chain stops)
            else if (total == n_roots) { S->finished = 1; S->last_stage = (uint32_t)k; S->last_is_top = 0; }   // only the roots are left
            else if (total >= n) S->trouble = 2;                                                       // no progress
            else { const uint32_t nk = (total <= Rf) ? SK_TOP : SK_TILE; S->kind[k + 1] = nk; s_kind[k + 1] = nk; }
        }
        // The next stage of THIS workgroup reads what this one wrote: workgroup scope (its waves share the CU's vector cache;
        // the stores are complete before the barrier). An agent-scope __threadfence() here wrote the XCD's whole L2 back -- tens
        // of MB of the earlier kernels' dirty lines -- once per wave: 18 of this kernel's 24 us (profiles/r04b_plan_phase_clocks.txt).
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __syncthreads();
    }
}

static int get_schedule_exact(raht_plan *plan, int R0, int R1, int Rf, hipStream_t s, Schedule **out);

// -> RAHT_OK and *built = true when the schedule was built; *built = false: use the exact builder
static int build_schedule_fast(raht_plan *plan, int R0, int R1, int Rf, hipStream_t s, Schedule &sc, bool *built, bool *heights_done)
{
    *built = false;
    *heights_done = false;
    const int64_t N = plan->N;
    // Sizes are unknown on the host. Stages expected to be large (a stage keeps ~1/20 of its entries at 184 rows per
    // tile, ~1/6 at 64: assume 1/16 resp. 1/4) get the two multi-workgroup launches; from the first stage expected to
    // be small on, ONE single-workgroup launch finishes the chain (sched_tail_kernel: up to TAIL_MAX entries per stage).
    // Buffers hold 1/3 of the stage before (+ slack). A stage larger than expected at the tail, or a bound exceeded ->
    // the state says so and the exact builder takes over.
    constexpr uint32_t TAIL_MAX = 65536;
    constexpr int KB = SCHED_SPEC_MAX;                    // stages with buffers: 0 .. KB
    int64_t cap[SCHED_SPEC_MAX + 2];
    cap[0] = N;
    for (int k = 0; k <= KB; ++k) cap[k + 1] = std::min<int64_t>(cap[k], cap[k] / 3 + 2048);
    int KS = 0;                                           // stages given the multi-workgroup launches: k = 0 .. KS - 1
    if (N > Rf) {
        double expect = (double)N;
        while (KS < KB - 1 && KS < plan->max_stages && (KS == 0 || expect > (double)TAIL_MAX * 0.5)) { ++KS; expect /= (R0 >= 128 ? 16.0 : 4.0); }
    }
    struct Bufs { uint32_t *rows = nullptr; int32_t *wl = nullptr, *wr = nullptr; uint8_t *lvl = nullptr; uint32_t *pos = nullptr; uint32_t *surv = nullptr; uint8_t *ht = nullptr; };
    std::vector<Bufs> B((size_t)KB + 2);
    uint32_t *t_pj = nullptr, *t_root = nullptr, *t_lev = nullptr;
    float *t_ab32 = nullptr;
    double *t_ab64 = nullptr;
    bool ok = true;
    auto take = [&](auto **ptr, size_t bytes) { if (ok && dev_malloc(ptr, std::max<size_t>(bytes, 16)) != hipSuccess) ok = false; };
    for (int k = 0; k < KB && N > Rf; ++k) {
        const int R = (k == 0) ? R0 : R1;
        take(&B[(size_t)k].surv, sizeof(uint32_t) * (size_t)(ceil_div(cap[k], R) + 1));
        Bufs &nx = B[(size_t)k + 1];
        const size_t c = (size_t)cap[k + 1];
        take(&nx.rows, 4 * c); take(&nx.wl, 4 * c); take(&nx.wr, 4 * c); take(&nx.lvl, c); take(&nx.pos, 4 * c);
    }
    // The butterfly heights of every tile stage are enqueued right behind the chain, BEFORE the read-back below: their launch
    // takes sizes and stage kinds from the device state and a grid that covers the stages' capacities, so the host's wait for the
    // read-back (~15-20 us of wake-up and launch latency, during which the GPU used to idle) overlaps the kernel.
    // (RAHT_HEIGHT_STAGES_PER_LAUNCH, the testing aid of launch_stage_heights, keeps the launch behind the read-back.)
    const bool early_heights = N > Rf && KB <= HT_MAX_STAGES && std::max(R0, R1) <= HT_MAX_ROWS && !getenv("RAHT_HEIGHT_STAGES_PER_LAUNCH");
    if (early_heights)
        for (int k = 0; k < KB; ++k) take(&B[(size_t)k].ht, (size_t)cap[k]);
    const size_t tm = (size_t)std::max(Rf, 1);
    take(&t_pj, 4 * tm); take(&t_ab32, 8 * tm); take(&t_ab64, 16 * tm); take(&t_root, 4 * tm); take(&t_lev, 4 * 128);
    // scratch: state | per-block counts | flags
    const size_t nblk0 = (size_t)ceil_div(N, SB_BLOCK);
    Scratch scr(sizeof(SchedState) + sizeof(uint32_t) * nblk0 + (size_t)N + 16, s);
    auto release = [&]() {
        for (auto &b : B) { dev_free(b.rows); dev_free(b.wl); dev_free(b.wr); dev_free(b.lvl); dev_free(b.pos); dev_free(b.surv); dev_free(b.ht); }
        dev_free(t_pj); dev_free(t_ab32); dev_free(t_ab64); dev_free(t_root); dev_free(t_lev);
    };
    if (!ok || !scr.ok()) { (void)hipDeviceSynchronize(); release(); return RAHT_ERR_NOMEM; }
    SchedState *dS = scr.as<SchedState>();
    uint32_t *blk_cnt = (uint32_t *)(dS + 1);
    uint8_t *flags = (uint8_t *)(((uintptr_t)(blk_cnt + nblk0) + 15) & ~(uintptr_t)15);      // (sched_count_kernel stores 8 flags at a time)
    SchedStageBufs SB;
    for (int k = 0; k <= KB + 1 && k < SCHED_SPEC_MAX + 2; ++k) {
        const Bufs &b = B[(size_t)k];
        SB.rows[k] = b.rows; SB.wl[k] = b.wl; SB.wr[k] = b.wr; SB.lvl[k] = b.lvl; SB.pos[k] = b.pos; SB.surv[k] = b.surv;
        SB.cap[k] = (uint32_t)cap[k];
    }
    const SchedTopOut TO = {t_pj, t_ab32, t_ab64, t_root, t_lev};
    // the first launch of the chain also initialises the state (no upload, no memset: ~5 us each)
    for (int k = 0; k < KS; ++k) {
        const Bufs &cur = B[(size_t)k];
        const int R = (k == 0) ? R0 : R1;
        const unsigned gb = (unsigned)ceil_div(cap[k], SB_BLOCK);
        const Bufs &nx = B[(size_t)k + 1];
        hipLaunchKernelGGL(sched_count_kernel, dim3(gb), dim3(SB_THREADS), 0, s, dS, k, cur.rows, k ? cur.wl : plan->wl,
                           k ? cur.wr : plan->wr, k ? cur.lvl : plan->lvl, R, N, plan->top_level, flags, blk_cnt, k == 0 ? N : (int64_t)-1);
        hipLaunchKernelGGL(sched_emit_kernel, dim3(gb), dim3(SB_THREADS), 0, s, dS, k, cur.rows, flags, blk_cnt, R, (uint32_t)Rf,
                           (uint32_t)plan->n_roots, plan->wl, plan->wr, plan->lvl, plan->inv_order, nx.rows, nx.wl, nx.wr, nx.lvl,
                           nx.pos, (uint32_t)cap[k + 1], cur.surv);
    }
    hipLaunchKernelGGL(sched_tail_kernel, dim3(1), dim3(ST_THREADS), 0, s, dS, KS, KB, SB, R1, (uint32_t)Rf, (uint32_t)plan->n_roots, N,
                       plan->top_level, TAIL_MAX, plan->wl, plan->wr, plan->lvl, plan->inv_order, plan->wsum, TO, KS == 0 ? N : (int64_t)-1);
    auto heights_behind = [&]() {
        HeightArgs H;
        H.n_stages = 0; H.n_tiles = 0; H.N = N; H.top_level = plan->top_level;
        for (int k = 0; k < KB; ++k) {
            const Bufs &b = B[(size_t)k];
            HeightStage &h = H.st[H.n_stages++];
            h.rows = b.rows; h.wl = k ? b.wl : plan->wl; h.wr = k ? b.wr : plan->wr; h.lvl = k ? b.lvl : plan->lvl;
            h.ht = b.ht; h.n = 0; h.R = (k == 0) ? R0 : R1; h.first_tile = H.n_tiles;
            h.n_dev = &dS->n[k]; h.kind_dev = &dS->kind[k]; h.kind_tile = SK_TILE;
            H.n_tiles += (uint32_t)ceil_div(cap[k], h.R);
        }
        for (int q = H.n_stages; q < HT_MAX_STAGES; ++q) H.st[q] = H.st[0];
        launch_heights_kernel(H, std::max(R0, R1), true, s);
    };
    hipError_t e = hipGetLastError();
    SchedState hs;
    int rc = RAHT_ERR_HIP;
    if (e == hipSuccess) {
        rc = read_back_u32((uint32_t *)&hs, (const uint32_t *)dS, SCHED_STATE_WORDS, plan->pend_host, plan->pend_dev, plan->pend_n, s,
                           early_heights ? std::function<void()>(heights_behind) : std::function<void()>());
        if (rc == RAHT_OK && hipGetLastError() != hipSuccess) rc = RAHT_ERR_HIP;
        if (rc == RAHT_OK) plan->pend_n = 0;              // delivered
    }
    if (rc != RAHT_OK || !hs.finished || hs.trouble || (int)hs.last_stage >= plan->max_stages) {
        (void)hipStreamSynchronize(s);
        release();
        (void)hipGetLastError();
        return rc == RAHT_OK ? RAHT_OK : rc;             // *built stays false: the exact builder decides
    }
    const int K = (int)hs.last_stage + 1;
    for (int k = 0; k < K; ++k) {
        Stage st;
        Bufs &b = B[(size_t)k];
        st.n_entries = hs.n[k];
        st.rows = b.rows; st.e_wl = b.wl; st.e_wr = b.wr; st.e_lvl = b.lvl; st.e_pos = b.pos;
        b.rows = nullptr; b.wl = nullptr; b.wr = nullptr; b.lvl = nullptr; b.pos = nullptr;
        if (k == K - 1 && hs.last_is_top) {
            st.is_top = true;
            st.n_tiles = 1;
            st.tile_rows = (int)st.n_entries;
            st.n_merges = hs.top[0]; st.t_nlev = (int)hs.top[1]; st.t_nbig = (int)hs.top[2]; st.t_small_start = hs.top[3];
            st.t_pj = t_pj; st.t_ab32 = t_ab32; st.t_ab64 = t_ab64; st.t_root = t_root; st.t_lev = t_lev;
            t_pj = nullptr; t_ab32 = nullptr; t_ab64 = nullptr; t_root = nullptr; t_lev = nullptr;
        } else {
            st.tile_rows = (k == 0) ? R0 : R1;
            st.n_tiles = ceil_div(st.n_entries, st.tile_rows);
            st.surv_off = b.surv;
            b.surv = nullptr;
            if (early_heights) { st.e_ht = b.ht; b.ht = nullptr; }
        }
        sc.stages.push_back(st);
    }
    release();                                            // buffers of stages that were not needed
    *built = true;
    *heights_done = early_heights;
    return RAHT_OK;
}

int get_schedule(raht_plan *plan, int R0, int R1, int Rf, hipStream_t s, Schedule **out)
{
    for (auto &sc : plan->schedules)
        if (sc.tile_rows == R0 && sc.tail_rows == R1 && sc.final_rows == Rf) { *out = &sc; return RAHT_OK; }
    static const bool exact_only = getenv("RAHT_SCHEDULE_EXACT") != nullptr;     // A/B and debugging knob
    if (!exact_only && R0 >= 64 && R1 >= 64 && Rf >= 1 && Rf <= RAHT_TOP_MAX_ROWS) {
        Schedule sc;
        sc.tile_rows = R0; sc.tail_rows = R1; sc.final_rows = Rf; sc.valid = true;
        bool built = false, heights_done = false;
        RAHT_RET(build_schedule_fast(plan, R0, R1, Rf, s, sc, &built, &heights_done));
        if (built) {
            const int rch = heights_done ? RAHT_OK : launch_stage_heights(plan, sc, s);
            if (rch != RAHT_OK) { free_schedule(sc); return rch; }
            plan->schedules.push_back(sc);
            *out = &plan->schedules.back();
            return RAHT_OK;
        }
    }
    return get_schedule_exact(plan, R0, R1, Rf, s, out);
}

static int get_schedule_exact(raht_plan *plan, int R0, int R1, int Rf, hipStream_t s, Schedule **out)
{
    Schedule sc;
    sc.tile_rows = R0;
    sc.tail_rows = R1;
    sc.final_rows = Rf;
    sc.valid = true;
    const int64_t N = plan->N;
    Scratch buf(sizeof(uint32_t) * (2 * (size_t)N + 1), s);
    if (!buf.ok()) return RAHT_ERR_NOMEM;
    uint32_t *flag = buf.as<uint32_t>(), *pos = flag + N, *dtotal = pos + N;
    uint32_t *rows = nullptr;      // rows of the current stage (nullptr = identity)
    int64_t n = N;
    int rc = RAHT_OK;
    const int max_stages = std::max(1, plan->max_stages);
    for (int k = 0; k < max_stages; ++k) {
        const int R = (k == 0) ? R0 : R1;
        if (n <= Rf) {                                       // few entries left: the TOP stage finishes the tree
            Stage st;
            rc = build_top_stage(plan, rows, n, s, st);
            sc.stages.push_back(st);
            break;
        }
        Stage st;
        st.n_entries = n;
        st.n_tiles = ceil_div(n, R);
        st.rows = rows;
        st.tile_rows = R;
        if (rows) {
            if (dev_malloc(&st.e_wl, sizeof(int32_t) * (size_t)n) != hipSuccess || dev_malloc(&st.e_wr, sizeof(int32_t) * (size_t)n) != hipSuccess ||
                dev_malloc(&st.e_lvl, (size_t)n) != hipSuccess || dev_malloc(&st.e_pos, sizeof(uint32_t) * (size_t)n) != hipSuccess) { rc = RAHT_ERR_NOMEM; sc.stages.push_back(st); break; }
            hipLaunchKernelGGL(gather_meta_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, rows, n, plan->wl,
                               plan->wr, plan->lvl, plan->inv_order, st.e_wl, st.e_wr, st.e_lvl, st.e_pos);
        }
        hipLaunchKernelGGL(stage_survivor_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s,
                           rows, n, R, N, plan->wl, plan->wr, plan->lvl, plan->top_level, flag);
        rc = exclusive_scan_u32(flag, pos, n, dtotal, s);
        if (rc != RAHT_OK) break;
        uint32_t cnt32 = 0;
        rc = read_back_u32(&cnt32, dtotal, 1, nullptr, nullptr, 0, s);
        if (rc != RAHT_OK) break;
        const int64_t cnt = cnt32;
        const bool last = (cnt == plan->n_roots);            // only the roots are left: tree finished
        // no progress, or more stages than the plan allows. (With >= 64 rows per tile and <= 63 key bits a stage
        // always merges something -- the minimum-level entry of the first tile cannot reach past it, DESIGN.md
        // 4.2 -- so in practice only the stage limit, raht_plan_set_max_stages, ends up here.)
        if (!last && (cnt >= n || k == max_stages - 1)) {
            sc.stages.push_back(st);
            sc.valid = false;
            break;
        }
        if (dev_malloc(&st.surv_off, sizeof(uint32_t) * (size_t)(st.n_tiles + 1)) != hipSuccess) { rc = RAHT_ERR_NOMEM; break; }
        hipLaunchKernelGGL(tile_start_kernel, dim3((unsigned)ceil_div(st.n_tiles + 1, 256)), dim3(256), 0, s,
                           pos, n, R, st.n_tiles, cnt32, st.surv_off);
        if (last) { sc.stages.push_back(st); break; }
        uint32_t *next = nullptr;
        if (dev_malloc(&next, sizeof(uint32_t) * (size_t)cnt) != hipSuccess) { rc = RAHT_ERR_NOMEM; break; }
        hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, rows,
                           flag, pos, next, n);
        sc.stages.push_back(st);
        rows = next;
        n = cnt;
    }
    hipError_t e = hipStreamSynchronize(s);
    if (rc == RAHT_OK && e != hipSuccess) rc = RAHT_ERR_HIP;
    if (rc != RAHT_OK) {
        if (!sc.stages.empty() && sc.stages.back().rows != rows && rows) dev_free(rows);
        free_schedule(sc);
        set_error("schedule build failed");
        return rc;
    }
    if (sc.valid) {
        rc = launch_stage_heights(plan, sc, s);
        if (rc != RAHT_OK) { free_schedule(sc); return rc; }
    }
    plan->schedules.push_back(sc);                   // std::deque: earlier schedules keep their addresses
    *out = &plan->schedules.back();
    return RAHT_OK;
}

// ---- roots: rows that still carry a low-pass value when the (possibly truncated) tree is done ----
__global__ void root_flag_kernel(const uint8_t *__restrict__ lvl, int64_t N, int top_level, uint32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) flag[i] = (i == 0 || (int)lvl[i] >= top_level) ? 1u : 0u;
}

static int compute_roots(raht_plan *p, hipStream_t s)
{
    if (p->root_rows) { dev_free(p->root_rows); p->root_rows = nullptr; }
    if (p->top_level > p->max_level) {
        // untruncated tree (the usual case): row 0 is the only row left carrying a low-pass value
        p->n_roots = 1;
        RAHT_HIP_CHECK(dev_malloc(&p->root_rows, sizeof(uint32_t)));
        RAHT_HIP_CHECK(hipMemsetAsync(p->root_rows, 0, sizeof(uint32_t), s));
        return RAHT_OK;
    }
    Scratch buf(sizeof(uint32_t) * 2 * (size_t)p->N, s);
    if (!buf.ok()) return RAHT_ERR_NOMEM;
    uint32_t *flag = buf.as<uint32_t>(), *tmp = flag + p->N;
    hipLaunchKernelGGL(root_flag_kernel, dim3((unsigned)ceil_div(p->N, 256)), dim3(256), 0, s, p->lvl, p->N,
                       p->top_level, flag);
    int64_t cnt = 0;
    RAHT_RET(compact_u32(nullptr, flag, tmp, p->N, &cnt, s));
    p->n_roots = cnt;
    RAHT_HIP_CHECK(dev_malloc(&p->root_rows, sizeof(uint32_t) * (size_t)cnt));
    RAHT_HIP_CHECK(hipMemcpyAsync(p->root_rows, tmp, sizeof(uint32_t) * (size_t)cnt, hipMemcpyDeviceToDevice, s));
    RAHT_HIP_CHECK(hipStreamSynchronize(s));
    return RAHT_OK;
}

// Rows bucketed by binary level (the LEVEL engine's pair lists): built on first use only -- the tile engine
// never reads them. level_off (host) comes from the level histogram of the plan build.
int ensure_level_rows(raht_plan *p, hipStream_t s)
{
    if (p->level_rows) return RAHT_OK;
    RAHT_HIP_CHECK(dev_malloc(&p->level_rows, sizeof(uint32_t) * (size_t)p->N));
    // the low 6 bits of lvl are the bucket (row 0, lvl 255, is alone in bucket 63); stable: ascending rows per level
    return bucket_sort_u8(p->lvl, p->level_rows, p->N, 6, nullptr, s);
}

// ---- plan construction ---------------------------------------------------------------------------
// First launch of a plan build: the caller's sorted keys into the plan (keys_in == nullptr: they are there already), the
// error word and the root list of an untruncated tree (row 0) initialised -- one launch instead of a device copy, an
// 8-byte upload and a 4-byte memset (~5 us each on the stream)
__global__ void plan_begin_kernel(const uint64_t *__restrict__ keys_in, uint64_t *__restrict__ keys, int64_t N, PlanErr *derr, uint32_t *root_rows)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) { derr->code = 0; derr->row = 0xffffffffu; root_rows[0] = 0; }
    if (!keys_in) return;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x)
        keys[i] = __builtin_nontemporal_load(keys_in + i);
}

static int finish_plan(raht_plan *p, const int64_t *leaf_weights, hipStream_t s, const uint64_t *keys_in)
{
    const int64_t N = p->N;
    const unsigned gb = (unsigned)ceil_div(N, 256);
    const unsigned nblk = (unsigned)ceil_div(N, EXT_THREADS);
    // scratch: error word | level starts [64] | (order bucket + level) histograms / positions [(ORDER_BUCKETS + 64) x nblk]
    //          | search queue [EXT_QCAP x nblk] + counts [nblk] | bucket ids [N]
    Scratch tmp(sizeof(PlanErr) + sizeof(uint32_t) * (64 + (ORDER_BUCKETS + 64) + (size_t)(ORDER_BUCKETS + 64 + EXT_QCAP + 1) * nblk) + (size_t)N, s);
    if (!tmp.ok()) return RAHT_ERR_NOMEM;
    PlanErr *derr = tmp.as<PlanErr>();
    uint32_t *lhist = (uint32_t *)((char *)tmp.ptr() + sizeof(PlanErr));
    uint32_t *bin_total = lhist + 64;
    uint32_t *bhist = bin_total + (ORDER_BUCKETS + 64);
    uint32_t *gq = bhist + (size_t)(ORDER_BUCKETS + 64) * nblk, *gq_count = gq + (size_t)EXT_QCAP * nblk;
    uint8_t *bucket = (uint8_t *)(gq_count + nblk);
    static_assert(sizeof(PlanErr) == 2 * sizeof(uint32_t), "PlanErr is read back as two words");
    // top_level = 64 at creation: row 0 is the only row left carrying a low-pass value (compute_roots' first case)
    p->n_roots = 1;
    RAHT_HIP_CHECK(dev_malloc(&p->root_rows, sizeof(uint32_t)));
    hipLaunchKernelGGL(plan_begin_kernel, dim3(keys_in ? (unsigned)std::min<int64_t>(ceil_div(N, 256), 4096) : 1u), dim3(256), 0, s,
                       keys_in, p->keys, N, derr, p->root_rows);
    RAHT_HIP_CHECK(dev_malloc(&p->lvl, (size_t)N));
    RAHT_HIP_CHECK(dev_malloc(&p->wl, sizeof(int32_t) * (size_t)N));
    RAHT_HIP_CHECK(dev_malloc(&p->wr, sizeof(int32_t) * (size_t)N));
    RAHT_HIP_CHECK(dev_malloc(&p->order, sizeof(uint32_t) * (size_t)N));
    RAHT_HIP_CHECK(dev_malloc(&p->inv_order, sizeof(uint32_t) * (size_t)N));
    // everything below is enqueued speculatively; the error word is checked at the single sync
    hipLaunchKernelGGL(level_extent_kernel, dim3(nblk), dim3(EXT_THREADS), 0, s, p->keys, N,
                       p->nbits, p->lvl, bucket, p->wl, p->wr, derr, bhist, gq, gq_count);
    hipLaunchKernelGGL(extent_finish_kernel, dim3((unsigned)(ORDER_BUCKETS + 64 + ceil_div((int64_t)nblk * EXT_QCAP, BSCAN_THREADS))), dim3(BSCAN_THREADS), 0, s,
                       p->keys, N, p->lvl, gq, gq_count, nblk, p->wl, p->wr, bhist, bin_total);
    // order_RAGFT and its inverse: stable counting sort by bucket (histogram from the pass above). (Measured and dropped,
    // round 3: this pair on a second stream next to the extent searches and the first schedule pass, and the stage-0 butterfly
    // heights on a third next to the whole schedule build -- 0.245 / 0.235 ms against 0.226 ms on one stream: a cross-queue
    // dependency costs ~13 us, and kernels this small slow each other down when they share the chip.)
    hipLaunchKernelGGL(order_scatter_kernel, dim3(nblk), dim3(EXT_THREADS), 0, s, bucket, N, bhist, bin_total, p->order, p->inv_order, lhist);
    if (getenv("RAHT_DEBUG_IDENTITY_ORDER")) {    // timing experiments only: order_RAGFT := identity
        hipLaunchKernelGGL(order_to_identity_kernel, dim3(gb), dim3(256), 0, s, p->order, N);
        hipLaunchKernelGGL(invert_perm_kernel, dim3(gb), dim3(256), 0, s, p->order, N, p->inv_order);
    }
    // The error word and the level histogram travel with the schedule builder's read-back (ONE host round trip per
    // plan). Everything up to it is enqueued speculatively: kernels running on unsorted keys read and write inside
    // their arrays all the same, and their results are thrown away with the plan.
    uint32_t back[2 + 64];
    p->pend_dev = (const uint32_t *)derr; p->pend_host = back; p->pend_n = 2 + 64;
    if (leaf_weights) {
        // prefix sums of the leaf weights on the host: weighted plans are tiny (<= 512 rows when
        // they stitch the top octree levels of a sharded scene)
        std::vector<int64_t> w((size_t)N), ps((size_t)N + 1);
        RAHT_HIP_CHECK(hipMemcpy(w.data(), leaf_weights, sizeof(int64_t) * (size_t)N, hipMemcpyDeviceToHost));
        ps[0] = 0;
        for (int64_t i = 0; i < N; ++i) {
            if (w[(size_t)i] < 1) { set_error("leaf weight < 1 at row %lld", (long long)i); return RAHT_ERR_INVALID; }
            ps[(size_t)i + 1] = ps[(size_t)i] + w[(size_t)i];
        }
        RAHT_HIP_CHECK(dev_malloc(&p->wsum, sizeof(int64_t) * ((size_t)N + 1)));
        RAHT_HIP_CHECK(hipMemcpy(p->wsum, ps.data(), sizeof(int64_t) * ((size_t)N + 1), hipMemcpyHostToDevice));
    }
    RAHT_HIP_CHECK(hipGetLastError());
    // Build the default schedule now so that float32 transforms with D <= 64 never allocate.
    Schedule *sc = nullptr;
    const int R0 = pick_tile_rows(p, 4, 59);
    int R1 = 0, Dc1 = 0, Rf = 0;
    pick_tail_geometry(p, 4, 59, R0, &R1, &Dc1, &Rf);
    int rc_sched = get_schedule(p, R0, R1, Rf, s, &sc);
    if (p->pend_n) {                                // the builder did not take them along (exact path, or it failed early)
        p->pend_n = 0;
        RAHT_RET(read_back_u32(back, (const uint32_t *)derr, 2 + 64, nullptr, nullptr, 0, s));
    }
    p->pend_dev = nullptr; p->pend_host = nullptr;
    PlanErr he;
    he.code = (int)back[0]; he.row = back[1];
    const uint32_t *lh = back + 2;
    if (he.code != 0) {
        if (he.code == RAHT_ERR_UNSORTED)
            set_error("Morton keys are not strictly increasing at row %u (input must be Morton-sorted "
                      "and duplicate-free)", he.row);
        else
            set_error("coordinate / key out of bounds at row %u for depth %d", he.row, p->nbits / 3);
        return he.code;
    }
    RAHT_RET(rc_sched);
    for (int l = 0; l < 64; ++l) p->level_off[l] = lh[l];      // rows bucketed by level (bucket 63 = row 0), see ensure_level_rows
    p->level_off[64] = (uint32_t)N;
    p->max_level = -1;                              // highest level that has a pair (bucket 63 = row 0)
    for (int l = 0; l < 63; ++l)
        if (p->level_off[l + 1] > p->level_off[l]) p->max_level = l;
    return RAHT_OK;
}

}  // namespace raht

using namespace raht;

extern "C" {

#ifdef RAHT_PHASE_CLOCKS
/* profiling build only: dst[3][n_blocks][12] <- the plan-build kernels' phase stamps (tools/phase_clocks_plan.py) */
int raht_debug_read_phase_clocks_plan(unsigned long long *dst, int which, int n_blocks)
{
    if (which < 0 || which > 2) return RAHT_ERR_INVALID;
    RAHT_HIP_CHECK(hipDeviceSynchronize());
    RAHT_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase_clk_plan), sizeof(unsigned long long) * PL_CLK_SLOTS * (size_t)std::min(n_blocks, PL_CLK_BLOCKS),
                                       sizeof(unsigned long long) * PL_CLK_SLOTS * PL_CLK_BLOCKS * (size_t)which));
    return PL_CLK_SLOTS;
}
#endif

const char *raht_last_error(void) { return g_err; }
int raht_version(void) { return RAHT_VERSION; }

// Owns a half-built plan: whatever way a constructor leaves (error code or exception), the plan is destroyed.
struct PlanHolder {
    raht_plan *p = nullptr;
    ~PlanHolder() { if (p) raht_plan_destroy(p); }
    raht_plan *release() { raht_plan *q = p; p = nullptr; return q; }
};

int raht_plan_create(const void *V, int v_dtype, int64_t N, const double minV[3], double width,
                     int depth, raht_stream_t stream, raht_plan **out)
{
    if (!V || !out || !minV) { set_error("raht_plan_create: NULL argument"); return RAHT_ERR_INVALID; }
    if (N < 1 || N >= ((int64_t)1 << 31)) { set_error("raht_plan_create: N=%lld out of range", (long long)N); return RAHT_ERR_INVALID; }
    if (depth < 1 || depth > 21) { set_error("raht_plan_create: depth=%d (1..21)", depth); return RAHT_ERR_INVALID; }
    if (!(width > 0)) { set_error("raht_plan_create: width must be > 0"); return RAHT_ERR_INVALID; }
    if (v_dtype < RAHT_F32 || v_dtype > RAHT_I64) { set_error("raht_plan_create: bad v_dtype %d", v_dtype); return RAHT_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    return guarded("raht_plan_create", [&]() -> int {
        PlanHolder h;
        h.p = new raht_plan();
        raht_plan *p = h.p;
        p->device = current_device();
        p->N = N;
        p->nbits = 3 * depth;
        if (dev_malloc(&p->keys, sizeof(uint64_t) * (size_t)N) != hipSuccess) { set_error("hipMalloc keys"); return RAHT_ERR_NOMEM; }
        Scratch errw(sizeof(PlanErr), s);
        if (!errw.ok()) return RAHT_ERR_NOMEM;
        PlanErr *derr = errw.as<PlanErr>();
        PlanErr h0 = {0, 0xffffffffu};
        RAHT_HIP_CHECK(hipMemcpyAsync(derr, &h0, sizeof(h0), hipMemcpyHostToDevice, s));
        const double Q = width / (double)((uint64_t)1 << depth);
        const unsigned gb = (unsigned)ceil_div(N, 256);
        switch (v_dtype) {
        case RAHT_F64: hipLaunchKernelGGL(keys_from_coords_kernel<double>, dim3(gb), dim3(256), 0, s, (const double *)V, N, minV[0], minV[1], minV[2], Q, depth, p->keys, derr); break;
        case RAHT_F32: hipLaunchKernelGGL(keys_from_coords_kernel<float>, dim3(gb), dim3(256), 0, s, (const float *)V, N, minV[0], minV[1], minV[2], Q, depth, p->keys, derr); break;
        case RAHT_I32: hipLaunchKernelGGL(keys_from_coords_kernel<int32_t>, dim3(gb), dim3(256), 0, s, (const int32_t *)V, N, minV[0], minV[1], minV[2], Q, depth, p->keys, derr); break;
        default:       hipLaunchKernelGGL(keys_from_coords_kernel<int64_t>, dim3(gb), dim3(256), 0, s, (const int64_t *)V, N, minV[0], minV[1], minV[2], Q, depth, p->keys, derr); break;
        }
        RAHT_HIP_CHECK(hipGetLastError());
        PlanErr he;
        RAHT_RET(read_back_u32((uint32_t *)&he, (const uint32_t *)derr, 2, nullptr, nullptr, 0, s));
        if (he.code != 0) {
            set_error("coordinate out of [0, 2^%d) at row %u (reference RAHT_param.py:26-27 raises ValueError)", depth, he.row);
            return he.code;
        }
        RAHT_RET(finish_plan(p, nullptr, s, nullptr));
        *out = h.release();
        return RAHT_OK;
    });
}

static int plan_from_keys_impl(const uint64_t *keys_sorted, int64_t N, int nbits, const int64_t *leaf_weights, bool borrow,
                               raht_stream_t stream, raht_plan **out, const char *what)
{
    if (!keys_sorted || !out) { set_error("%s: NULL argument", what); return RAHT_ERR_INVALID; }
    if (N < 1 || N >= ((int64_t)1 << 31)) { set_error("%s: N=%lld out of range", what, (long long)N); return RAHT_ERR_INVALID; }
    if (nbits < 1 || nbits > 63) { set_error("%s: nbits=%d (1..63)", what, nbits); return RAHT_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    return guarded(what, [&]() -> int {
        PlanHolder h;
        h.p = new raht_plan();
        raht_plan *p = h.p;
        p->device = current_device();
        p->N = N;
        p->nbits = nbits;
        if (borrow) {
            // the caller's array IS the plan's key array (no 8 N-byte copy): it must stay alive and unchanged as long as the plan
            p->keys = const_cast<uint64_t *>(keys_sorted);
            p->keys_borrowed = true;
        } else if (dev_malloc(&p->keys, sizeof(uint64_t) * (size_t)N) != hipSuccess) { set_error("hipMalloc keys"); return RAHT_ERR_NOMEM; }
        RAHT_RET(finish_plan(p, leaf_weights, s, borrow ? nullptr : keys_sorted));
        *out = h.release();
        return RAHT_OK;
    });
}

int raht_plan_create_from_keys(const uint64_t *keys_sorted, int64_t N, int nbits,
                               const int64_t *leaf_weights, raht_stream_t stream, raht_plan **out)
{
    return plan_from_keys_impl(keys_sorted, N, nbits, leaf_weights, false, stream, out, "raht_plan_create_from_keys");
}

int raht_plan_create_from_keys_borrowed(const uint64_t *keys_sorted, int64_t N, int nbits,
                                        const int64_t *leaf_weights, raht_stream_t stream, raht_plan **out)
{
    return plan_from_keys_impl(keys_sorted, N, nbits, leaf_weights, true, stream, out, "raht_plan_create_from_keys_borrowed");
}

int raht_plan_destroy(raht_plan *p)
{
    if (!p) return RAHT_OK;
    // destruction may be called with any device current (a Python finaliser runs wherever the
    // interpreter happens to be): work on the plan's own device
    DeviceGuard on_plan_device(p->device);
    // the plan's blocks go back to the cache and may be handed to another plan at once: nothing
    // enqueued on any stream may still be using them (hipFree used to imply the same wait)
    (void)hipDeviceSynchronize();
    for (auto &sc : p->schedules) free_schedule(sc);
    if (p->keys && !p->keys_borrowed) dev_free(p->keys);
    if (p->lvl) dev_free(p->lvl);
    if (p->wl) dev_free(p->wl);
    if (p->wr) dev_free(p->wr);
    if (p->wsum) dev_free(p->wsum);
    if (p->order) dev_free(p->order);
    if (p->inv_order) dev_free(p->inv_order);
    if (p->level_rows) dev_free(p->level_rows);
    if (p->root_rows) dev_free(p->root_rows);
    if (p->row_map) dev_free(p->row_map);
    delete p;
    return RAHT_OK;
}

int64_t raht_plan_size(const raht_plan *p) { return p ? p->N : -1; }
int raht_plan_nbits(const raht_plan *p) { return p ? p->nbits : -1; }

int raht_plan_set_tail_tile(raht_plan *p, int tail_rows, int tail_channels, int final_rows)
{
    if (!p || tail_rows < 0 || tail_rows > 1024 || (tail_rows & 3) || tail_channels < 0 || tail_channels > 64 ||
        final_rows < 0 || final_rows > RAHT_TOP_MAX_ROWS) {
        set_error("raht_plan_set_tail_tile: rows multiple of 4 in [0, 1024], channels in [0, 64], top-stage rows in [0, %d]", RAHT_TOP_MAX_ROWS);
        return RAHT_ERR_INVALID;
    }
    p->tail_rows_override = tail_rows;
    p->tail_chunk_override = tail_channels;
    p->final_rows_override = final_rows;
    return RAHT_OK;
}

int raht_plan_set_max_stages(raht_plan *p, int max_stages)
{
    if (!p || max_stages < 1 || max_stages > 64) { set_error("raht_plan_set_max_stages: 1..64"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_set_max_stages"));
    if (max_stages == p->max_stages) return RAHT_OK;
    for (auto &sc : p->schedules) free_schedule(sc);       // schedules were built under the old limit
    p->schedules.clear();
    p->max_stages = max_stages;
    return RAHT_OK;
}

int raht_plan_set_concurrent_directions(raht_plan *p, int on)
{
    if (!p) { set_error("raht_plan_set_concurrent_directions: NULL plan"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_set_concurrent_directions"));
    p->split_ws = on != 0;                                  // (the workspaces are re-made by the next transform / raht_plan_prepare)
    return RAHT_OK;
}

int raht_plan_set_stage0_events(raht_plan *p, void *ev_before, void *ev_after)
{
    if (!p || ((ev_before == nullptr) != (ev_after == nullptr))) { set_error("raht_plan_set_stage0_events: need both events or none"); return RAHT_ERR_INVALID; }
    p->ev_before = (hipEvent_t)ev_before;
    p->ev_after = (hipEvent_t)ev_after;
    return RAHT_OK;
}

int raht_plan_set_engine(raht_plan *p, int engine, int tile_rows)
{
    if (!p || (engine != RAHT_ENGINE_TILE && engine != RAHT_ENGINE_LEVEL)) { set_error("raht_plan_set_engine: bad argument"); return RAHT_ERR_INVALID; }
    if (tile_rows != 0 && (tile_rows < 64 || tile_rows > 1024 || (tile_rows & 3))) { set_error("tile_rows must be a multiple of 4 in [64, 1024]"); return RAHT_ERR_INVALID; }
    p->engine = engine;
    p->tile_rows_override = tile_rows;
    return RAHT_OK;
}

// len(Flags) of the reference: the loop of RAHT_param.py:226 stops at the level where a single
// node is left, i.e. after the highest level that has a pair.
int raht_plan_levels(const raht_plan *p)
{
    if (!p) return -1;
    return p->N == 1 ? 1 : p->max_level + 1;
}

int raht_plan_export_level(const raht_plan *cp, int level, int64_t *list, uint8_t *flags,
                           int64_t *weights, int64_t *n_out)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    if (!p || !n_out) { set_error("raht_plan_export_level: NULL argument"); return RAHT_ERR_INVALID; }
    if (level < 0 || level >= raht_plan_levels(p)) { set_error("raht_plan_export_level: level %d out of range", level); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_export_level"));
    try {
    if (p->lvl_host.empty()) {
        p->lvl_host.resize((size_t)p->N);
        RAHT_HIP_CHECK(hipMemcpy(p->lvl_host.data(), p->lvl, (size_t)p->N, hipMemcpyDeviceToHost));
    }
    const uint8_t *lv = p->lvl_host.data();
    std::vector<int64_t> ws;
    if (p->wsum) {
        ws.resize((size_t)p->N + 1);
        RAHT_HIP_CHECK(hipMemcpy(ws.data(), p->wsum, sizeof(int64_t) * ((size_t)p->N + 1), hipMemcpyDeviceToHost));
    }
    // nodes alive at `level` start at row 0 and at every row whose own level is >= `level`
    int64_t n = 0, prev = -1;
    for (int64_t i = 0; i < p->N; ++i) {
        if (i != 0 && lv[i] < level) continue;
        if (prev >= 0) {
            if (weights) weights[n - 1] = p->wsum ? ws[(size_t)i] - ws[(size_t)prev] : i - prev;
            if (flags) flags[n - 1] = (lv[i] == level) ? 1 : 0;
        }
        if (list) list[n] = i;
        prev = i;
        ++n;
    }
    if (weights) weights[n - 1] = p->wsum ? ws[(size_t)p->N] - ws[(size_t)prev] : p->N - prev;
    if (flags) flags[n - 1] = 0;
    *n_out = n;
    } catch (const std::exception &) {               // host allocation of the N-sized staging copies
        set_error("raht_plan_export_level: out of host memory");
        return RAHT_ERR_NOMEM;
    }
    return RAHT_OK;
}

__global__ void order_to_i64_kernel(const uint32_t *__restrict__ o, int64_t N, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < N) out[i] = (int64_t)o[i];
}

int raht_plan_order(const raht_plan *p, int64_t *order_dev, raht_stream_t stream)
{
    if (!p || !order_dev) { set_error("raht_plan_order: NULL argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_order"));
    hipLaunchKernelGGL(order_to_i64_kernel, dim3((unsigned)ceil_div(p->N, 256)), dim3(256), 0,
                       (hipStream_t)stream, p->order, p->N, order_dev);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_plan_arrays(const raht_plan *p, const uint64_t **keys, const uint8_t **lvl,
                     const int32_t **wl, const int32_t **wr)
{
    if (!p) return RAHT_ERR_INVALID;
    if (keys) *keys = p->keys;
    if (lvl) *lvl = p->lvl;
    if (wl) *wl = p->wl;
    if (wr) *wr = p->wr;
    return RAHT_OK;
}

int raht_plan_set_top_level(raht_plan *p, int top_level, raht_stream_t stream)
{
    if (!p || top_level < 1 || top_level > 64) { set_error("raht_plan_set_top_level: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_set_top_level"));
    if (top_level == p->top_level) return RAHT_OK;
    for (auto &sc : p->schedules) free_schedule(sc);
    p->schedules.clear();
    p->top_level = top_level;
    return compute_roots(p, (hipStream_t)stream);
}

int raht_plan_roots(const raht_plan *p, int64_t *n_roots, int64_t *rows_dev, raht_stream_t stream)
{
    if (!p || !n_roots) { set_error("raht_plan_roots: NULL argument"); return RAHT_ERR_INVALID; }
    *n_roots = p->n_roots;
    if (rows_dev) {
        RAHT_RET(check_plan_device(p, "raht_plan_roots"));
        hipLaunchKernelGGL(order_to_i64_kernel, dim3((unsigned)ceil_div(p->n_roots, 256)), dim3(256), 0,
                           (hipStream_t)stream, p->root_rows, p->n_roots, rows_dev);
        RAHT_HIP_CHECK(hipGetLastError());
    }
    return RAHT_OK;
}

__global__ void row_map_kernel(const int64_t *__restrict__ map, int64_t n, int64_t n_rows, uint32_t *__restrict__ out, PlanErr *err)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t r = map[i];
    if (r < 0 || r >= n_rows) report(err, RAHT_ERR_BOUNDS, i);
    out[i] = (uint32_t)r;
}

int raht_plan_set_row_map(raht_plan *p, const int64_t *map_dev, int64_t n_matrix_rows, raht_stream_t stream)
{
    if (!p) { set_error("raht_plan_set_row_map: NULL plan"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_set_row_map"));
    hipStream_t s = (hipStream_t)stream;
    if (!map_dev) {
        if (p->row_map) { (void)hipDeviceSynchronize(); dev_free(p->row_map); p->row_map = nullptr; p->map_rows = 0; }
        return RAHT_OK;
    }
    if (p->N > RAHT_TOP_MAX_ROWS) { set_error("raht_plan_set_row_map: plans of at most %d rows", RAHT_TOP_MAX_ROWS); return RAHT_ERR_UNSUPPORTED; }
    if (n_matrix_rows < p->N || n_matrix_rows >= ((int64_t)1 << 31)) { set_error("raht_plan_set_row_map: n_matrix_rows=%lld", (long long)n_matrix_rows); return RAHT_ERR_INVALID; }
    if (!p->row_map) RAHT_HIP_CHECK(dev_malloc(&p->row_map, sizeof(uint32_t) * (size_t)p->N));
    Scratch errw(sizeof(PlanErr), s);
    if (!errw.ok()) return RAHT_ERR_NOMEM;
    PlanErr h0 = {0, 0xffffffffu}, he;
    RAHT_HIP_CHECK(hipMemcpyAsync(errw.ptr(), &h0, sizeof(h0), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(row_map_kernel, dim3((unsigned)ceil_div(p->N, 256)), dim3(256), 0, s, map_dev, p->N, n_matrix_rows, p->row_map, errw.as<PlanErr>());
    RAHT_HIP_CHECK(hipGetLastError());
    RAHT_RET(read_back_u32((uint32_t *)&he, (const uint32_t *)errw.ptr(), 2, nullptr, nullptr, 0, s));
    if (he.code != 0) {
        dev_free(p->row_map); p->row_map = nullptr; p->map_rows = 0;
        set_error("raht_plan_set_row_map: map[%u] outside [0, %lld)", he.row, (long long)n_matrix_rows);
        return RAHT_ERR_BOUNDS;
    }
    p->map_rows = n_matrix_rows;
    return RAHT_OK;
}

int raht_plan_set_root_buffer(raht_plan *p, void *buf_dev)
{
    if (!p) return RAHT_ERR_INVALID;
    p->root_buf = buf_dev;
    return RAHT_OK;
}

int raht_plan_copy_array(const raht_plan *p, int which, void *dst, raht_stream_t stream)
{
    if (!p || !dst || which < 0 || which > 3) { set_error("raht_plan_copy_array: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_copy_array"));
    const void *src[4] = {p->keys, p->lvl, p->wl, p->wr};
    const size_t es[4] = {8, 1, 4, 4};
    RAHT_HIP_CHECK(hipMemcpyAsync(dst, src[which], es[which] * (size_t)p->N, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return RAHT_OK;
}

int raht_plan_stage_stats(raht_plan *p, int elem_size, int D, int *n_stages, int64_t *rows_per_stage,
                          int max_stages, int *tile_rows)
{
    if (!p || !n_stages) return RAHT_ERR_INVALID;
    RAHT_RET(check_plan_device(p, "raht_plan_stage_stats"));
    const int Dc = pick_chunk_channels(elem_size, D);
    const int R = pick_tile_rows(p, elem_size, Dc);
    if (R == 0) { set_error("no tile size fits"); return RAHT_ERR_UNSUPPORTED; }
    Schedule *sc = nullptr;
    int R1 = 0, Dc1 = 0, Rf = 0;
    pick_tail_geometry(p, elem_size, D, R, &R1, &Dc1, &Rf);
    RAHT_RET(guarded("raht_plan_stage_stats", [&]() { return get_schedule(p, R, R1, Rf, nullptr, &sc); }));
    *n_stages = sc->valid ? (int)sc->stages.size() : -(int)sc->stages.size();
    if (tile_rows) *tile_rows = R;
    for (int k = 0; k < (int)sc->stages.size() && k < max_stages; ++k)
        if (rows_per_stage) rows_per_stage[k] = sc->stages[(size_t)k].n_entries;
    return RAHT_OK;
}

int raht_morton(const int64_t *V, int64_t N, int J, uint64_t *keys, raht_stream_t stream)
{
    if (N < 0 || J < 1 || J > 21) { set_error("raht_morton: bad argument"); return RAHT_ERR_INVALID; }
    if (N == 0) return RAHT_OK;
    if (!V || !keys) { set_error("raht_morton: NULL argument"); return RAHT_ERR_INVALID; }
    hipLaunchKernelGGL(morton_i64_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0,
                       (hipStream_t)stream, V, N, keys);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

}  // extern "C"
