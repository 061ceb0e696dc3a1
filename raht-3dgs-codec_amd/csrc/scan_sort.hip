// scan_sort.hip -- device primitives for the plan and the voxelizer: exclusive scan, stable LSD
// radix pass (wave64 ballot ranking), bucket sort, stream compaction. Hand-written for gfx950:
// 64-lane ballots, LDS histograms, no library calls.
//
// These are integer/HBM-bound kernels (no MFMA). They run at plan / voxelize time, not inside the
// transform entry points.
#include "raht_common.h"
#include "raht_device.h"
#include <map>
#include <mutex>
#include <unordered_map>

namespace raht {

// ------------------------------------------------------------------------------------------------
// Scratch pool
// ------------------------------------------------------------------------------------------------
struct PoolBlock { void *p; size_t bytes; bool used; int device; hipStream_t last_stream; bool touched; };
static thread_local std::vector<PoolBlock> g_pool;

Scratch::Scratch(size_t bytes, hipStream_t stream)
{
    if (bytes == 0) bytes = 16;
    const int dev = current_device();
    int best = -1, n_dev = 0;
    // best fit among the free blocks whose last user ran on THIS stream (or that were never used): a block last used on another
    // stream costs a device synchronisation before it may be handed out, which would serialise a caller that works on two
    // streams at once (raht_voxelize_plan: the plan build next to the voxelizer's mean kernel) -- such a block is only taken
    // when the pool is already large
    int n_free_other = 0, best_other = -1;
    for (int i = 0; i < (int)g_pool.size(); ++i) {
        const PoolBlock &b = g_pool[(size_t)i];
        if (b.device != dev || !b.p) continue;
        ++n_dev;
        if (b.used || b.bytes < bytes) continue;
        if (!b.touched || b.last_stream == stream) { if (best < 0 || b.bytes < g_pool[(size_t)best].bytes) best = i; }
        else { ++n_free_other; if (best_other < 0 || b.bytes < g_pool[(size_t)best_other].bytes) best_other = i; }
    }
    if (best < 0 && best_other >= 0 && n_dev >= 40) best = best_other;
    (void)n_free_other;
    if (best < 0) {
        // recycle the largest free block (of this device) that is too small, else grow the pool
        int victim = -1, empty = -1;
        for (int i = 0; i < (int)g_pool.size(); ++i) {
            const PoolBlock &b = g_pool[(size_t)i];
            if (!b.p && !b.used) { if (empty < 0) empty = i; continue; }
            if (b.device == dev && !b.used && (victim < 0 || b.bytes > g_pool[(size_t)victim].bytes)) victim = i;
        }
        void *q = nullptr;
        const size_t want = bytes + bytes / 4;                 // head-room against slow growth
        if (victim >= 0 && n_dev >= 48) {
            (void)hipFree(g_pool[(size_t)victim].p);
            // a failed allocation leaves an EMPTY slot behind: erasing it would shift the slot indices that
            // live Scratch objects hold, and a later destructor would release somebody else's block
            g_pool[(size_t)victim] = {nullptr, 0, false, dev, nullptr, false};
            if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); set_error("scratch: out of device memory (%zu bytes)", want); return; }
            g_pool[(size_t)victim] = {q, want, false, dev, nullptr, false};
            best = victim;
        } else {
            if (hipMalloc(&q, want) != hipSuccess) { (void)hipGetLastError(); set_error("scratch: out of device memory (%zu bytes)", want); return; }
            if (empty >= 0) { g_pool[(size_t)empty] = {q, want, false, dev, nullptr, false}; best = empty; }
            else { g_pool.push_back({q, want, false, dev, nullptr, false}); best = (int)g_pool.size() - 1; }
        }
    }
    PoolBlock &blk = g_pool[(size_t)best];
    // the block's previous user may still be running on another stream (see raht_common.h)
    if (blk.touched && blk.last_stream != stream) (void)hipDeviceSynchronize();
    blk.used = true;
    blk.touched = true;
    blk.last_stream = stream;
    p_ = blk.p;
    slot_ = best;
}

Scratch::~Scratch()
{
    if (slot_ >= 0 && slot_ < (int)g_pool.size()) g_pool[(size_t)slot_].used = false;
}

// ------------------------------------------------------------------------------------------------
// Cache of long-lived device blocks (see raht_common.h).
// ------------------------------------------------------------------------------------------------
namespace {
struct LiveBlock { size_t cls; int device; };
struct DevCache {
    std::mutex mu;
    std::unordered_map<void *, LiveBlock> live;                    // block -> (class size, device)
    std::multimap<std::pair<int, size_t>, void *> free_blocks;     // (device, class size) -> block
    size_t cached = 0, limit = 0;
};
DevCache &dev_cache()
{
    static DevCache c;
    if (c.limit == 0) {
        const char *e = getenv("RAHT_POOL_MAX_BYTES");
        c.limit = e ? (size_t)strtoull(e, nullptr, 10) : ((size_t)8 << 30);
        if (c.limit == 0) c.limit = 1;                    // "0" = cache nothing
    }
    return c;
}
size_t size_class(size_t bytes)
{
    if (bytes < 4096) return 4096;
    int hb = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (hb - 3);           // eight classes per power of two
    return (bytes + step - 1) / step * step;
}
}  // namespace

hipError_t dev_malloc(void **p, size_t bytes)
{
    DevCache &c = dev_cache();
    const size_t cls = size_class(bytes);
    const int dev = current_device();
    {
        std::lock_guard<std::mutex> g(c.mu);
        auto it = c.free_blocks.find(std::make_pair(dev, cls));
        if (it != c.free_blocks.end()) {
            *p = it->second;
            c.free_blocks.erase(it);
            c.cached -= cls;
            c.live[*p] = LiveBlock{cls, dev};
            return hipSuccess;
        }
    }
    hipError_t e = hipMalloc(p, cls);
    if (e != hipSuccess) {                               // under memory pressure: drop the cache and retry
        (void)hipGetLastError();
        raht_release_cached_memory();
        e = hipMalloc(p, cls);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(c.mu);
        c.live[*p] = LiveBlock{cls, dev};
    }
    return e;
}

void dev_free(void *p)
{
    if (!p) return;
    DevCache &c = dev_cache();
    size_t cls = 0;
    {
        std::lock_guard<std::mutex> g(c.mu);
        int dev = 0;
        auto it = c.live.find(p);
        if (it != c.live.end()) { cls = it->second.cls; dev = it->second.device; c.live.erase(it); }
        if (cls && c.cached + cls <= c.limit) {
            c.free_blocks.emplace(std::make_pair(dev, cls), p);      // back to the bucket of the block's OWN device
            c.cached += cls;
            return;
        }
    }
    (void)hipFree(p);
}

// ------------------------------------------------------------------------------------------------
// Exclusive scan (3 kernels, recursive on the block sums).
// ------------------------------------------------------------------------------------------------
constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;   // 2048 items per block

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// Block-wide exclusive scan of one value per thread (256 threads). Returns the exclusive prefix;
// *block_total receives the sum (valid in every thread).
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *block_total)
{
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = wave_incl_scan(v);
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t s = wsum[w];
        if (w < wid) base += s;
        tot += s;
    }
    __syncthreads();
    *block_total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_block_kernel(const uint32_t *in,   // may alias out
                                                                  uint32_t *out,
                                                                  uint32_t *sums,
                                                                  int64_t n,
                                                                  uint32_t *total)      // single-block scans only
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t v[SCAN_ITEMS];
    uint32_t tsum = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        v[k] = (base + k < n) ? in[base + k] : 0u;
        tsum += v[k];
    }
    uint32_t tot;
    uint32_t ex = block_excl_scan_256(tsum, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
    if (threadIdx.x == 0 && sums) sums[blockIdx.x] = tot;
    if (threadIdx.x == 0 && total) *total = tot;
}

// Second and last launch of a scan of up to SCAN_FUSE_BLOCKS blocks: every block sums the totals of the
// blocks before it by itself (<= 8 KB from L2) instead of waiting for a scan of the totals, and the last
// block also writes the grand total. (Scans of this size are launch-bound: 2 launches instead of 3-5.)
constexpr int SCAN_FUSE_BLOCKS = 2048;

__global__ __launch_bounds__(SCAN_THREADS) void scan_finish_kernel(uint32_t *__restrict__ out,
                                                                   const uint32_t *__restrict__ sums,
                                                                   int64_t n, uint32_t *__restrict__ total)
{
    uint32_t part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += sums[b];
    uint32_t add;
    (void)block_excl_scan_256(part, &add);
    if (total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = add + sums[blockIdx.x];
    if (blockIdx.x == 0) return;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] += add;
}

__global__ __launch_bounds__(SCAN_THREADS) void scan_add_kernel(uint32_t *__restrict__ out,
                                                                const uint32_t *__restrict__ sums,
                                                                int64_t n)
{
    const uint32_t add = sums[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_ITEMS;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k)
        if (base + k < n) out[base + k] += add;
}

__global__ void scan_total_kernel(const uint32_t *in_last, const uint32_t *out_last, uint32_t *total)
{
    *total = *in_last + *out_last;
}

static int scan_rec(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *ws, hipStream_t s)
{
    const int64_t nb = ceil_div(n, SCAN_BLOCK);
    if (nb == 1) {
        hipLaunchKernelGGL(scan_block_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, out,
                           (uint32_t *)nullptr, n, (uint32_t *)nullptr);
        return RAHT_OK;
    }
    uint32_t *sums = ws;
    hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, sums, n,
                       (uint32_t *)nullptr);
    RAHT_RET(scan_rec(sums, sums, nb, ws + nb, s));
    hipLaunchKernelGGL(scan_add_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, out, sums, n);
    return RAHT_OK;
}

int exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total, hipStream_t s)
{
    if (n <= 0) {
        if (total) RAHT_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(uint32_t), s));
        return RAHT_OK;
    }
    // workspace: nb + nb/2048 + ... entries, plus one word to keep in[n-1] (out may alias in)
    int64_t wsn = 1;
    for (int64_t m = ceil_div(n, SCAN_BLOCK); m > 1; m = ceil_div(m, SCAN_BLOCK)) wsn += m;
    Scratch ws(sizeof(uint32_t) * (size_t)wsn, s);
    if (!ws.ok()) return RAHT_ERR_NOMEM;
    const int64_t nb = ceil_div(n, SCAN_BLOCK);
    if (nb <= SCAN_FUSE_BLOCKS) {
        uint32_t *sums = ws.as<uint32_t>();
        if (nb == 1) {
            hipLaunchKernelGGL(scan_block_kernel, dim3(1), dim3(SCAN_THREADS), 0, s, in, out, (uint32_t *)nullptr, n, total);
        } else {
            hipLaunchKernelGGL(scan_block_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, in, out, sums, n,
                               (uint32_t *)nullptr);
            hipLaunchKernelGGL(scan_finish_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, out, sums, n, total);
        }
        RAHT_HIP_CHECK(hipGetLastError());
        return RAHT_OK;
    }
    uint32_t *last_in = ws.as<uint32_t>();
    if (total)
        RAHT_HIP_CHECK(hipMemcpyAsync(last_in, in + (n - 1), sizeof(uint32_t), hipMemcpyDeviceToDevice, s));
    int rc = scan_rec(in, out, n, ws.as<uint32_t>() + 1, s);
    if (rc == RAHT_OK && total)
        hipLaunchKernelGGL(scan_total_kernel, dim3(1), dim3(1), 0, s, last_in, out + (n - 1), total);
    RAHT_HIP_CHECK(hipGetLastError());
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Stable LSD radix pass. A block owns RP_BLOCK consecutive items; its 4 waves own consecutive
// quarter-chunks and walk them in rounds of 64 (one item per lane), so "earlier" in memory order is
// (block, wave, round, lane) -- ranks are assigned in exactly that order, which makes the pass stable.
// ------------------------------------------------------------------------------------------------
constexpr int RP_THREADS = 256;
constexpr int RP_WAVES = 4;
constexpr int RP_ROUNDS = 8;                           // 2048 items per block: 29 KiB of LDS, five blocks per CU. Measured on 3 M keys (36 / 60 bit):
                                                       // 16 rounds 0.289 / 0.446 ms, 10: 0.259 / 0.400, 8: 0.252 / 0.390, 6: 0.256 / 0.400, 4: 0.283 / 0.450 --
                                                       // occupancy against the length of the runs a digit leaves the block in
constexpr int RP_WAVE_ITEMS = 64 * RP_ROUNDS;          // 512
constexpr int RP_BLOCK = RP_WAVES * RP_WAVE_ITEMS;     // 2048 items per block

template <typename KeyT>
__device__ __forceinline__ uint32_t digit_of(KeyT k, int shift, uint32_t mask)
{
    return (uint32_t)(k >> shift) & mask;
}

template <typename KeyT>
__global__ __launch_bounds__(RP_THREADS) void radix_hist_kernel(const KeyT *__restrict__ keys,
                                                                int64_t n, int shift, uint32_t mask,
                                                                uint32_t *__restrict__ ghist,
                                                                uint32_t nblocks)
{
    __shared__ uint32_t h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RP_BLOCK;
#pragma unroll 4
    for (int k = 0; k < RP_BLOCK / RP_THREADS; ++k) {
        const int64_t i = base + (int64_t)k * RP_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[digit_of(keys[i], shift, mask)], 1u);
    }
    __syncthreads();
    if (threadIdx.x <= mask) ghist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];
}

// Lanes of the wave holding the same digit as this lane (among `valid` lanes).
__device__ __forceinline__ uint64_t match_digit(uint32_t digit, int bits, uint64_t valid)
{
    uint64_t m = valid;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        if (b < bits) {
            const uint64_t bal = __ballot((digit >> b) & 1u);
            m &= ((digit >> b) & 1u) ? bal : ~bal;
        }
    }
    return m;
}

// The block's items are first sorted by digit INSIDE LDS (stable: ranks in (wave, round, lane) order) and
// then written out slot by slot, so that every digit leaves the block as one contiguous run
// (2048 items over 256 digits: 8 items = 64-96 bytes per run) instead of one scattered 8 + 4 byte
// store per item.
template <typename KeyT, bool HAS_VALS_IN, bool WRITE_KEYS>
__global__ __launch_bounds__(RP_THREADS) void radix_scatter_kernel(
    const KeyT *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
    KeyT *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n, int shift, int bits,
    const uint32_t *__restrict__ goffs, uint32_t nblocks)
{
    __shared__ uint32_t wcnt[RP_WAVES][256];
    __shared__ uint32_t gdelta[256];                   // global position - block-local slot, per digit
    __shared__ KeyT skey[RP_BLOCK];
    __shared__ uint32_t sval[RP_BLOCK];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t mask = (1u << bits) - 1u;
    for (int k = threadIdx.x; k < RP_WAVES * 256; k += RP_THREADS) (&wcnt[0][0])[k] = 0;
    __syncthreads();

    const int64_t bbase = (int64_t)blockIdx.x * RP_BLOCK;
    const int64_t wbase = bbase + (int64_t)wid * RP_WAVE_ITEMS;
    const int nblk = (int)min((int64_t)RP_BLOCK, n - bbase);
    KeyT key[RP_ROUNDS];
    uint32_t val[RP_ROUNDS];
    // phase 1: per-wave digit histogram (keys and payloads stay in registers)
#pragma unroll
    for (int r = 0; r < RP_ROUNDS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        key[r] = (i < n) ? keys_in[i] : (KeyT)0;
        val[r] = HAS_VALS_IN ? ((i < n) ? vals_in[i] : 0u) : (uint32_t)i;
        if (i < n) atomicAdd(&wcnt[wid][digit_of(key[r], shift, mask)], 1u);
    }
    __syncthreads();
    // phase 2: counts -> block-local start slots (digit-major, then wave); remember where the digit's
    // run of this block starts globally
    {
        const uint32_t d = threadIdx.x;                 // RP_THREADS == 256 digits
        uint32_t c[RP_WAVES], tot = 0;
#pragma unroll
        for (int w = 0; w < RP_WAVES; ++w) { c[w] = wcnt[w][d]; tot += c[w]; }
        uint32_t btot;
        uint32_t run = block_excl_scan_256(tot, &btot); // (contains the barriers that order the reads above)
#pragma unroll
        for (int w = 0; w < RP_WAVES; ++w) { wcnt[w][d] = run; run += c[w]; }
        const uint32_t lbase = run - tot;
        gdelta[d] = (d <= mask ? goffs[(size_t)d * nblocks + blockIdx.x] : 0u) - lbase;
    }
    __syncthreads();
    // phase 3: rank inside the round by ballots, bump the wave's running slot, place into LDS
    // (wavefront-scope atomics, not a volatile pointer: hipcc turns volatile LDS accesses through a pointer into FLAT loads /
    // stores with system-scope cache bits and a s_waitcnt vmcnt(0) each. Within a wave LDS operations execute in order.)
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < RP_ROUNDS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint64_t vmask = __ballot(valid);
        if (vmask == 0) break;                       // wave-uniform
        const uint32_t d = digit_of(key[r], shift, mask);
        const uint64_t same = match_digit(d, bits, vmask);
        const uint32_t rank = (uint32_t)__popcll(same & lt);
        uint32_t slot = 0;
        if (valid) slot = __hip_atomic_load(&wcnt[wid][d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + rank;
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0) __hip_atomic_store(&wcnt[wid][d], slot + (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_wave_barrier();
        if (valid) { skey[slot] = key[r]; sval[slot] = val[r]; }
    }
    __syncthreads();
    // phase 4: slots in order -> global memory; consecutive slots of one digit are consecutive there
    for (int q = threadIdx.x; q < nblk; q += RP_THREADS) {
        const KeyT k = skey[q];
        const uint32_t pos = (uint32_t)q + gdelta[digit_of(k, shift, mask)];
        if (WRITE_KEYS) keys_out[pos] = k;
        vals_out[pos] = sval[q];
    }
}

__global__ void store_u32_kernel(uint32_t *p, uint32_t v) { *p = v; }

template <typename KeyT>
static int radix_pass_impl(const KeyT *keys_in, const uint32_t *vals_in, KeyT *keys_out,
                           uint32_t *vals_out, int64_t n, int shift, int bits, uint32_t *bucket_off,
                           hipStream_t s)
{
    if (n <= 0) return RAHT_OK;
    if (bits < 1 || bits > 8) { set_error("radix pass: bits=%d", bits); return RAHT_ERR_INVALID; }
    const uint32_t nb = (uint32_t)ceil_div(n, RP_BLOCK);
    const uint32_t nd = 1u << bits;
    Scratch gh(sizeof(uint32_t) * (size_t)nd * nb, s);
    if (!gh.ok()) return RAHT_ERR_NOMEM;
    uint32_t *ghist = gh.as<uint32_t>();
    hipLaunchKernelGGL(radix_hist_kernel<KeyT>, dim3(nb), dim3(RP_THREADS), 0, s, keys_in, n, shift,
                       nd - 1u, ghist, nb);
    int rc = exclusive_scan_u32(ghist, ghist, (int64_t)nd * nb, nullptr, s);
    if (rc != RAHT_OK) return rc;
    if (bucket_off) {
        // start of bucket d = goffs[d * nb + 0]; end sentinel = n
        RAHT_HIP_CHECK(hipMemcpy2DAsync(bucket_off, sizeof(uint32_t), ghist, sizeof(uint32_t) * nb,
                                        sizeof(uint32_t), nd, hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(store_u32_kernel, dim3(1), dim3(1), 0, s, bucket_off + nd, (uint32_t)n);
    }
    if (vals_in) {
        if (keys_out)
            hipLaunchKernelGGL((radix_scatter_kernel<KeyT, true, true>), dim3(nb), dim3(RP_THREADS), 0,
                               s, keys_in, vals_in, keys_out, vals_out, n, shift, bits, ghist, nb);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<KeyT, true, false>), dim3(nb), dim3(RP_THREADS), 0,
                               s, keys_in, vals_in, keys_out, vals_out, n, shift, bits, ghist, nb);
    } else {
        if (keys_out)
            hipLaunchKernelGGL((radix_scatter_kernel<KeyT, false, true>), dim3(nb), dim3(RP_THREADS), 0,
                               s, keys_in, vals_in, keys_out, vals_out, n, shift, bits, ghist, nb);
        else
            hipLaunchKernelGGL((radix_scatter_kernel<KeyT, false, false>), dim3(nb), dim3(RP_THREADS), 0,
                               s, keys_in, vals_in, keys_out, vals_out, n, shift, bits, ghist, nb);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// (A one-sweep variant -- one histogram pass for all digits up front, then ONE launch per digit whose blocks chain
// their per-digit offsets by decoupled look-back with bounded waits -- was built and measured in round 2: 57 us per
// pass against 10 + 10 + 35 us for histogram + scan + scatter here; 36-bit sort of 3 M keys 0.36 ms against 0.29 ms.
// 256 independent look-back chains per block through agent-scope loads, on 8 XCDs with separate L2s, cost more than the
// two small launches they replace. Removed again. Digits of 9 bits (36-bit keys in 4 passes instead of 5): 0.29 -> 0.27 ms, but
// every pass costs 68 us instead of 58 (512 runs of ~8 items per block write worse than 256 runs of 16; 58 KiB of LDS; a
// 512 x blocks scan), so 30-bit keys, which need 4 passes either way, got slower: not kept either. A one-launch scan of the
// digit-major histogram (workgroup per digit, digit totals by global atomics from the histogram kernel): 0.254 -> 0.360 ms --
// ~1500 atomics on each of 256 hot words serialise in L2; the two-launch generic scan stays.)
int radix_pass_u64(const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                   uint32_t *vals_out, int64_t n, int shift, int bits, hipStream_t s)
{
    return radix_pass_impl<uint64_t>(keys_in, vals_in, keys_out, vals_out, n, shift, bits, nullptr, s);
}

int bucket_sort_u8(const uint8_t *bucket, uint32_t *perm_out, int64_t n, int bits,
                   uint32_t *bucket_off, hipStream_t s)
{
    return radix_pass_impl<uint8_t>(bucket, nullptr, nullptr, perm_out, n, 0, bits, bucket_off, s);
}

// ------------------------------------------------------------------------------------------------
// Whole key sort in npass + 2 launches (round 3): ONE histogram launch for the digits of every pass, one launch
// that reduces its per-block counts, then ONE launch per digit pass whose tiles find their per-digit offsets among
// themselves instead of through a per-tile histogram, two scan launches and a scatter launch per pass. On 3 M keys a
// pass of the four-launch form takes ~50 us for 72 MB of traffic -- launch gaps and three extra dependent kernels,
// not bytes.
//
// A tile = OS_TILE consecutive items, one workgroup of four waves; with 16 items per lane a tile takes 42 of the CU's
// 128 LDS granules, so 768 tiles are resident at once and the 733 tiles of a 3 M-key pass all start together. Tile t
// is workgroup t (RAHT_SORT_TICKET=1: the t-th workgroup to take a ticket -- see "order" below). Per digit d (thread d)
// the tile publishes ONE word, state[t][d] = OS_PART | its count, as soon as its histogram is known. Tiles form groups
// of 32: thread d adds the published counts of the tiles before its own in its group (31 loads in flight at once);
// the LAST tile of a group then knows the group's total and publishes gstate[g][d] = OS_PART | total, and everybody
// walks back over the groups before its own, OS_WIN words per round trip, until a word marked OS_INCL (a group's
// last tile upgrades its word to the inclusive prefix once it knows it; group 0's is inclusive from the start).
// A plain chain of tiles (the textbook decoupled look-back) costs O(sqrt(tiles)) dependent round trips when all tiles
// start in the same microsecond; this form costs two or three whatever the tile count, and the loads of the first
// step are issued BEFORE the ranking phase and the group total leaves in the MIDDLE of it, so that most of the
// waiting hides behind the ranks (a round trip of these cache-bypassing loads is ~2 us).
// Flag and value share a word, so relaxed agent-scope atomics are all the ordering needed (they bypass the per-XCD
// L2s; everything else is ordinary kernel-boundary visibility).
// Order: a tile only ever waits for tiles with a smaller number. Workgroups are dispatched in index order (round-robin
// over the XCDs, in order inside each), so the lowest-numbered unfinished tile is always resident and the waits cannot
// cycle. The library does not stake a GPU on that: every wait is bounded by the 100 MHz wall clock (OS_WAIT_TICKS);
// a tile that gives up publishes OS_ERR, which later tiles pass on -- they write nothing, the grid drains, the host
// finds the error word set and repeats the sort pass by pass. RAHT_SORT_TICKET=1 numbers the tiles by an atomic
// ticket instead (provably free of cycles whatever the dispatch order; 733 atomics on one word: +4 us per pass).
// Ranks inside a tile are assigned in (wave, round, lane) = memory order, tiles are numbered in memory order: stable.
// ------------------------------------------------------------------------------------------------
constexpr int OS_THREADS = 256;
constexpr int OS_WAVES = 4;
constexpr int OS_MAX_PASSES = 8;
constexpr uint32_t OS_PART = 1u << 30, OS_INCL = 2u << 30, OS_ERR = 3u << 30, OS_VALUE = (1u << 30) - 1u;
constexpr int OS_WIN = 16;                             // group words fetched per round trip of the walk over the groups
constexpr uint32_t OS_GROUP_LG = 5, OS_GROUP = 1u << OS_GROUP_LG;   // tiles per group
constexpr uint64_t OS_WAIT_TICKS = 20000000ull;        // 0.2 s of the 100 MHz wall clock
constexpr int OS_SLICES = 16;                          // the histogram blocks are reduced in this many slices per pass
constexpr int OS_HIST_BLOCKS = 512;

struct OsPasses {
    int npass;
    int shift[OS_MAX_PASSES];
    int bits[OS_MAX_PASSES];
};

// The digit of a 64-bit key at a (uniform) shift without a 64-bit shift: a 32-bit window of the key.
__device__ __forceinline__ uint32_t os_digit(uint64_t k, int shift, uint32_t mask)
{
    const uint32_t lo = (uint32_t)k, hi = (uint32_t)(k >> 32);
    const uint32_t w = shift >= 32 ? (hi >> (shift - 32)) : __builtin_amdgcn_alignbit(hi, lo, (uint32_t)shift);
    return w & mask;
}

// partial[b][p * 256 + d] = number of keys of block b whose digit of pass p is d. The same launch zeroes the
// look-back words of the passes that follow, and the sort's error word. FROM_CLOUD: the keys are not read but computed
// from the cloud's coordinates (and stored): the voxelizer's key kernel and this one are ONE pass over the points.
template <bool FROM_CLOUD>
__global__ __launch_bounds__(OS_THREADS) void os_hist_kernel(uint64_t *__restrict__ keys, int64_t n, OsPasses P,
                                                             uint32_t *__restrict__ partial, uint32_t *__restrict__ zero_words,
                                                             int64_t n_zero, uint32_t *__restrict__ err, const VoxGrid G)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) *err = 0u;
    __shared__ uint32_t h[OS_MAX_PASSES * 256];
    for (int k = threadIdx.x; k < P.npass * 256; k += OS_THREADS) h[k] = 0;
    __syncthreads();
    const int64_t stride = (int64_t)gridDim.x * OS_THREADS;
    // eight independent loads in flight per thread (one at a time, the loop is a chain of HBM round trips)
    constexpr int U = FROM_CLOUD ? 4 : 8;
    for (int64_t i0 = (int64_t)blockIdx.x * OS_THREADS + threadIdx.x; i0 < n; i0 += U * stride) {
        uint64_t k[U];
        if constexpr (FROM_CLOUD) {
            Xyz pt[U];
#pragma unroll
            for (int u = 0; u < U; ++u) pt[u] = *(const Xyz *)(G.PC + min(i0 + u * stride, n - 1) * G.ld);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                k[u] = vox_key(pt[u], G);
                if (i0 + u * stride < n) keys[i0 + u * stride] = k[u];
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) k[u] = keys[min(i0 + u * stride, n - 1)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (i0 + u * stride >= n) continue;
#pragma unroll
            for (int p = 0; p < OS_MAX_PASSES; ++p)
                if (p < P.npass) atomicAdd(&h[p * 256 + os_digit(k[u], P.shift[p], (1u << P.bits[p]) - 1u)], 1u);
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * OS_THREADS + threadIdx.x; i < n_zero; i += stride) zero_words[i] = 0u;
    __syncthreads();
    for (int k = threadIdx.x; k < P.npass * 256; k += OS_THREADS) partial[(size_t)blockIdx.x * (P.npass * 256) + k] = h[k];
}

// slices[p][q][d] = keys whose digit of pass p is d, counted over the q-th slice of the histogram blocks (a chain of
// dependent strided loads is what this launch costs: many short chains; the pass tiles add the OS_SLICES words up).
__global__ __launch_bounds__(OS_THREADS) void os_base_kernel(const uint32_t *__restrict__ partial, int nblocks, int npass,
                                                             uint32_t *__restrict__ slices)
{
    const int p = blockIdx.x / OS_SLICES, q = blockIdx.x % OS_SLICES, d = threadIdx.x;
    const uint32_t *src = partial + p * 256 + d;
    const size_t row = (size_t)npass * 256;
    const int per = (nblocks + OS_SLICES - 1) / OS_SLICES, b0 = q * per, b1 = min(nblocks, b0 + per);
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int b = b0;
    for (; b + 8 <= b1; b += 8)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] += src[(size_t)(b + u) * row];
    for (; b < b1; ++b) acc[0] += src[(size_t)b * row];
    slices[((size_t)p * OS_SLICES + q) * 256 + d] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
}

__device__ __forceinline__ uint32_t os_load(const uint32_t *p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void os_store(uint32_t *p, uint32_t v)
{
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// x = a look-back word as first loaded; polls until it carries a flag (bounded).
__device__ __forceinline__ uint32_t os_wait(const uint32_t *p, uint32_t x, bool &bad)
{
    if ((x >> 30) == 0) {
        const uint64_t t0 = wall_clock64();
        do {
            __builtin_amdgcn_s_sleep(1);
            x = os_load(p);
        } while ((x >> 30) == 0 && wall_clock64() - t0 < OS_WAIT_TICKS);
        if ((x >> 30) == 0) bad = true;
    }
    if ((x >> 30) == 3u) bad = true;
    return x;
}

// Block-wide exclusive scans of TWO values per thread at once (256 threads).
__device__ __forceinline__ void block_excl_scan2_256(uint32_t a, uint32_t b, uint32_t *ea, uint32_t *eb)
{
    __shared__ uint32_t wsa[4], wsb[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t ia = wave_incl_scan(a), ib = wave_incl_scan(b);
    if (lane == 63) { wsa[wid] = ia; wsb[wid] = ib; }
    __syncthreads();
    uint32_t ba = 0, bb = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        if (w < wid) { ba += wsa[w]; bb += wsb[w]; }
    }
    __syncthreads();
    *ea = ba + ia - a;
    *eb = bb + ib - b;
}

// Lanes of `valid` that hold the same digit as this lane. Per bit: bal = lanes with the bit set; the lanes that DIFFER from
// this one in that bit are bal ^ t with t = all ones when this lane's bit is clear... (sign-extended bit: 0 or -1, inverted
// sense folded into the final complement): the differences of all bits are OR-ed (two bits per v_or3) and complemented once.
__device__ __forceinline__ uint64_t os_match(uint32_t digit, int bits, uint64_t valid)
{
    uint32_t lo = 0, hi = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        if (b < bits) {                                 // (uniform)
            const int32_t t = ((int32_t)(digit << (31 - b))) >> 31;          // -1: bit set
            const uint64_t bal = __ballot(t != 0);
            lo |= (uint32_t)bal ^ (uint32_t)t;                               // set bit: differs from the lanes NOT in bal = ~bal = bal ^ -1
            hi |= (uint32_t)(bal >> 32) ^ (uint32_t)t;
        }
    }
    return ~(((uint64_t)hi << 32) | lo) & valid;
}

#ifdef RAHT_OS_CLOCKS
// profiling build only (make EXTRA=-DRAHT_OS_CLOCKS): 100 MHz wall-clock stamps at the phase boundaries of every tile of the
// pass whose shift is os_dbg_shift (tools/sort_phase_clocks.py)
__device__ unsigned long long os_dbg[8192 * 8];
__device__ int os_dbg_shift = 8;
#define OS_STAMP(k) do { if (threadIdx.x == 0 && shift == os_dbg_shift && tile < 8192) os_dbg[tile * 8 + (k)] = wall_clock64(); } while (0)
#else
#define OS_STAMP(k) do { } while (0)
#endif

template <int ROUNDS, bool HAS_VALS_IN, bool TICKET>
__global__ __launch_bounds__(OS_THREADS) void os_pass_kernel(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                             uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, int64_t n,
                                                             int shift, int bits, const uint32_t *__restrict__ slices,
                                                             uint32_t *__restrict__ state, uint32_t *__restrict__ gstate,
                                                             uint32_t *__restrict__ ticket, uint32_t *__restrict__ err,
                                                             int64_t *__restrict__ vals64_out /* last pass: the payload widened, or NULL */,
                                                             uint32_t fail_tile /* tests: this tile gives up (RAHT_SORT_DEBUG_FAIL_TILE) */)
{
    constexpr int WAVE_ITEMS = 64 * ROUNDS, TILE = OS_WAVES * WAVE_ITEMS;
    // 53 272 bytes with 16 rounds = 42 of the CU's 128 LDS granules: three tiles per CU, 768 on the chip (the 733 tiles of a
    // 3 M-key pass are all resident at once; one granule more and a third of them would wait for a second round)
    __shared__ uint32_t wcnt[OS_WAVES][256];
    uint32_t *gdelta = wcnt[0];                        // global position - tile-local slot, per digit (once the ranks are out)
    __shared__ uint64_t skey[TILE];
    __shared__ uint32_t sval[TILE];
    __shared__ uint32_t s_tile, s_bad;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t mask = (1u << bits) - 1u;
    for (int k = threadIdx.x; k < OS_WAVES * 256; k += OS_THREADS) (&wcnt[0][0])[k] = 0;
    if (threadIdx.x == 0) { s_tile = TICKET ? atomicAdd(ticket, 1u) : blockIdx.x; s_bad = 0; }
    __syncthreads();
    const uint32_t tile = TICKET ? s_tile : blockIdx.x;
    OS_STAMP(0);
    const int64_t bbase = (int64_t)tile * TILE;
    if (bbase >= n) return;                            // (cannot happen: the grid is exactly the tile count)
    const int64_t wbase = bbase + (int64_t)wid * WAVE_ITEMS;
    const int nblk = (int)min((int64_t)TILE, n - bbase);
    uint64_t key[ROUNDS];
    uint32_t val[ROUNDS], dig[ROUNDS];
    // phase 1: per-wave digit histogram (keys, payloads and digits stay in registers)
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        key[r] = (i < n) ? keys_in[i] : 0ull;
        val[r] = HAS_VALS_IN ? ((i < n) ? vals_in[i] : 0u) : (uint32_t)i;
    }
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int64_t i = wbase + r * 64 + lane;
        dig[r] = os_digit(key[r], shift, mask);
        if (i < n) atomicAdd(&wcnt[wid][dig[r]], 1u);
    }
    __syncthreads();
    OS_STAMP(1);
    // phase 2: publish this tile's count of digit d; counts -> tile-local start slots (digit-major, then wave); digit bases
    const uint32_t d = threadIdx.x;
    uint32_t *mine = state + (size_t)tile * 256 + d;
    const uint32_t g = tile >> OS_GROUP_LG, j = tile & (OS_GROUP - 1u);
    const uint32_t *row0 = state + (size_t)(g << OS_GROUP_LG) * 256 + d;
    uint32_t tot = 0, lbase, dbase;
    uint32_t pv[OS_GROUP - 1];                          // the words of the tiles before this one in its group
    {
        uint32_t c[OS_WAVES];
#pragma unroll
        for (int w = 0; w < OS_WAVES; ++w) { c[w] = wcnt[w][d]; tot += c[w]; }
        os_store(mine, OS_PART | tot);
        uint32_t dtot = 0;
#pragma unroll
        for (int q = 0; q < OS_SLICES; ++q) dtot += slices[q * 256 + d];
        uint32_t run;
        block_excl_scan2_256(tot, dtot, &run, &dbase);  // (contains the barriers that order the reads above)
        lbase = run;
#pragma unroll
        for (int w = 0; w < OS_WAVES; ++w) { wcnt[w][d] = run; run += c[w]; }
        // in flight across the first half of the ranks (the neighbours published a moment ago, or will in a moment)
        if (j > 0) {
#pragma unroll
            for (uint32_t u = 0; u < OS_GROUP - 1; ++u) pv[u] = os_load(row0 + (size_t)min(u, j - 1u) * 256);
        }
    }
    __syncthreads();
    OS_STAMP(2);
    // phase 3: rank inside the round by ballots, bump the wave's running slot, place into LDS. Within a wave LDS operations
    // execute in order: the slot counter a round's first lane of a digit writes is what the next round reads.
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    auto rank_round = [&](int r) {
        const int64_t i = wbase + r * 64 + lane;
        const bool valid = i < n;
        const uint64_t vmask = __ballot(valid);
        if (vmask != 0) {                                // wave-uniform
            const uint32_t dg = dig[r];
            const uint64_t same = os_match(dg, bits, vmask);
            const uint32_t rank = (uint32_t)__popcll(same & lt);
            uint32_t slot = 0;
            if (valid) slot = __hip_atomic_load(&wcnt[wid][dg], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) + rank;
            __builtin_amdgcn_wave_barrier();
            if (valid && rank == 0) __hip_atomic_store(&wcnt[wid][dg], slot + (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            if (valid) { skey[slot] = key[r]; sval[slot] = val[r]; }
        }
    };
#pragma unroll
    for (int r = 0; r < ROUNDS / 2; ++r) rank_round(r);
    OS_STAMP(3);
    // phase 4a: keys of digit d in the tiles before this one in its group; the group's last tile publishes the group's total
    bool bad = (tile == fail_tile);
    uint32_t sum_in = 0;
    const bool leader = (j == OS_GROUP - 1u);
    uint32_t *gmine = gstate + (size_t)g * 256 + d;
    if (j > 0) {
#pragma unroll
        for (uint32_t u = 0; u < OS_GROUP - 1; ++u) {
            if (u >= j) continue;                                           // (j is uniform over the workgroup)
            sum_in += os_wait(row0 + (size_t)u * 256, pv[u], bad) & OS_VALUE;
        }
    }
    if (leader) os_store(gmine, bad ? OS_ERR : ((g == 0 ? OS_INCL : OS_PART) | ((sum_in + tot) & OS_VALUE)));
    OS_STAMP(4);
#pragma unroll
    for (int r = ROUNDS / 2; r < ROUNDS; ++r) rank_round(r);
    OS_STAMP(5);
    // phase 4b: ... and in the groups before it
    uint32_t gs = 0;
    if (g > 0) {
        int64_t b = (int64_t)g - 1;
        for (bool done = false; !done && !bad && b >= 0;) {
            uint32_t v[OS_WIN];
#pragma unroll
            for (int u = 0; u < OS_WIN; ++u) v[u] = os_load(gstate + (size_t)max(b - u, (int64_t)0) * 256 + d);
#pragma unroll
            for (int u = 0; u < OS_WIN; ++u) {
                if (done || bad || b - u < 0) continue;
                const uint32_t x = os_wait(gstate + (size_t)(b - u) * 256 + d, v[u], bad);
                if (bad) break;
                gs += x & OS_VALUE;
                if ((x >> 30) == 2u) done = true;
            }
            b -= OS_WIN;                                                    // (group 0 publishes OS_INCL: the walk ends there)
        }
        if (leader) os_store(gmine, bad ? OS_ERR : (OS_INCL | ((gs + sum_in + tot) & OS_VALUE)));
    }
    if (bad) { s_bad = 1; atomicOr(err, 1u); }
    __syncthreads();                                   // every wave is done with its slot counters: wcnt[0] becomes gdelta
    gdelta[d] = dbase + gs + sum_in - lbase;
    __syncthreads();
    OS_STAMP(6);
    if (s_bad) return;
    // phase 5: slots in order -> global memory; consecutive slots of one digit are consecutive there
    for (int q = threadIdx.x; q < nblk; q += OS_THREADS) {
        const uint64_t k = skey[q];
        const uint32_t pos = (uint32_t)q + gdelta[os_digit(k, shift, mask)];
        if ((int64_t)pos < n) {
            const uint32_t v = sval[q];
            keys_out[pos] = k;
            vals_out[pos] = v;
            if (vals64_out) vals64_out[pos] = (int64_t)v;
        }
    }
    OS_STAMP(7);
}

// Items per lane of a tile: the smallest tile whose tile count still fits the chip at once (2048 items: five tiles per CU,
// 3072 / 4096: three) -- more, smaller tiles overlap their phases better, but a tile that has to wait for a second round of
// workgroups costs a whole tile lifetime. RAHT_SORT_ROUNDS = 8 / 12 / 16 pins it.
static int os_rounds(int64_t n)
{
    static int forced = -1;
    if (forced < 0) {
        const char *e = getenv("RAHT_SORT_ROUNDS");
        forced = e ? atoi(e) : 0;
        if (forced != 8 && forced != 12 && forced != 16) forced = 0;
    }
    if (forced) return forced;
    static int cus[RAHT_MAX_DEVICES] = {};
    const int dev = current_device();
    if (!cus[dev]) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    if (ceil_div(n, 2048) <= (int64_t)5 * cus[dev]) return 8;
    if (ceil_div(n, 3072) <= (int64_t)3 * cus[dev]) return 12;
    return 16;
}

static bool sort_onesweep_enabled()
{
    static int on = -1;
    if (on < 0) { const char *e = getenv("RAHT_SORT_ONESWEEP"); on = (e && e[0] == '0') ? 0 : 1; }
    return on != 0;
}

template <int ROUNDS>
static void os_launch_pass(const uint64_t *kin, const uint32_t *vin, uint64_t *kout, uint32_t *vout, int64_t n, int shift, int bits,
                           const uint32_t *slices, uint32_t *state, uint32_t *gstate, uint32_t *ticket, uint32_t *err, int64_t *v64, hipStream_t s)
{
    const unsigned nt = (unsigned)ceil_div(n, (int64_t)OS_WAVES * 64 * ROUNDS);
    static int tk = -1;
    static uint32_t fail_tile = 0xffffffffu;
    if (tk < 0) {
        const char *e = getenv("RAHT_SORT_TICKET"); tk = (e && e[0] == '1') ? 1 : 0;
        const char *f = getenv("RAHT_SORT_DEBUG_FAIL_TILE"); if (f) fail_tile = (uint32_t)strtoul(f, nullptr, 10);
    }
    if (tk) {
        if (vin)
            hipLaunchKernelGGL((os_pass_kernel<ROUNDS, true, true>), dim3(nt), dim3(OS_THREADS), 0, s, kin, vin, kout, vout, n, shift, bits, slices, state, gstate, ticket, err, v64, fail_tile);
        else
            hipLaunchKernelGGL((os_pass_kernel<ROUNDS, false, true>), dim3(nt), dim3(OS_THREADS), 0, s, kin, vin, kout, vout, n, shift, bits, slices, state, gstate, ticket, err, v64, fail_tile);
    } else {
        if (vin)
            hipLaunchKernelGGL((os_pass_kernel<ROUNDS, true, false>), dim3(nt), dim3(OS_THREADS), 0, s, kin, vin, kout, vout, n, shift, bits, slices, state, gstate, ticket, err, v64, fail_tile);
        else
            hipLaunchKernelGGL((os_pass_kernel<ROUNDS, false, false>), dim3(nt), dim3(OS_THREADS), 0, s, kin, vin, kout, vout, n, shift, bits, slices, state, gstate, ticket, err, v64, fail_tile);
    }
}

// Stable sort of (key, original index) by the low `nbits` key bits. tmp_keys / tmp_idx: N-sized ping-pong buffers (unused
// when one pass is enough). err_dev: device word, zeroed here and set when a tile gave up waiting (the caller reads it back
// with whatever it reads back anyway, once the stream has drained). Returns 1 when the input does not fit this form (the
// caller then uses the pass-by-pass sort).
int sort_pairs_onesweep(const uint64_t *keys_in, int64_t n, int nbits, uint64_t *keys_out, uint32_t *idx_out, uint64_t *tmp_keys,
                        uint32_t *tmp_idx, uint32_t *err_dev, hipStream_t s, int64_t *idx64_out, const VoxGrid *grid)
{
    if (n <= 0) return RAHT_OK;
    const int npass = std::max(1, (nbits + 7) / 8);
    if (npass > OS_MAX_PASSES || n >= ((int64_t)1 << 30) || !err_dev || !sort_onesweep_enabled()) return 1;
    OsPasses P;
    P.npass = npass;
    for (int p = 0, sh = 0; p < OS_MAX_PASSES; ++p) {
        const int b = p < npass ? std::max(1, (nbits - sh + (npass - p) - 1) / (npass - p)) : 1;    // digits as even as possible: 36 bits = 8,7,7,7,7
        P.shift[p] = p < npass ? sh : 0;
        P.bits[p] = b;
        if (p < npass) sh += b;
    }
    const int R = os_rounds(n);
    const int64_t ntiles = ceil_div(n, (int64_t)OS_WAVES * 64 * R);
    // (from the cloud: every point costs a 128-byte line, the launch is bound by loads in flight -- more, shorter blocks)
    static int hb_cloud = 0;
    if (!hb_cloud) { const char *e = getenv("RAHT_SORT_CLOUD_BLOCKS"); hb_cloud = e ? std::max(1, atoi(e)) : 4 * OS_HIST_BLOCKS; }
    const int hb = (int)std::min<int64_t>(ceil_div(n, OS_THREADS * 16), grid ? hb_cloud : OS_HIST_BLOCKS);
    // [ tickets (npass) | tile and group words (npass x (ntiles + ngroups) x 256) ] zeroed by the histogram launch, then partial, slices
    const int64_t ngroups = ceil_div(ntiles, (int64_t)OS_GROUP);
    const int64_t n_zero = OS_MAX_PASSES + (int64_t)npass * (ntiles + ngroups) * 256;
    Scratch ws(sizeof(uint32_t) * ((size_t)n_zero + (size_t)hb * npass * 256 + (size_t)npass * OS_SLICES * 256), s);
    if (!ws.ok()) return RAHT_ERR_NOMEM;
    uint32_t *ticket = ws.as<uint32_t>(), *state = ticket + OS_MAX_PASSES, *err = err_dev;
    uint32_t *partial = ticket + n_zero, *slices = partial + (size_t)hb * npass * 256;
    if (grid) hipLaunchKernelGGL(os_hist_kernel<true>, dim3(hb), dim3(OS_THREADS), 0, s, (uint64_t *)keys_in, n, P, partial, ticket, n_zero, err, *grid);
    else hipLaunchKernelGGL(os_hist_kernel<false>, dim3(hb), dim3(OS_THREADS), 0, s, (uint64_t *)keys_in, n, P, partial, ticket, n_zero, err, VoxGrid{});
    hipLaunchKernelGGL(os_base_kernel, dim3(npass * OS_SLICES), dim3(OS_THREADS), 0, s, partial, hb, npass, slices);
    const uint64_t *kin = keys_in;
    const uint32_t *vin = nullptr;
    for (int ps = 0; ps < npass; ++ps) {
        const bool to_out = ((npass - 1 - ps) % 2 == 0);
        uint64_t *ko = to_out ? keys_out : tmp_keys;
        uint32_t *vo = to_out ? idx_out : tmp_idx;
        uint32_t *st = state + (size_t)ps * (ntiles + ngroups) * 256, *gst = st + (size_t)ntiles * 256;
        const uint32_t *sl = slices + (size_t)ps * OS_SLICES * 256;
        if (R == 8) os_launch_pass<8>(kin, vin, ko, vo, n, P.shift[ps], P.bits[ps], sl, st, gst, ticket + ps, err, ps == npass - 1 ? idx64_out : nullptr, s);
        else if (R == 12) os_launch_pass<12>(kin, vin, ko, vo, n, P.shift[ps], P.bits[ps], sl, st, gst, ticket + ps, err, ps == npass - 1 ? idx64_out : nullptr, s);
        else os_launch_pass<16>(kin, vin, ko, vo, n, P.shift[ps], P.bits[ps], sl, st, gst, ticket + ps, err, ps == npass - 1 ? idx64_out : nullptr, s);
        kin = ko;
        vin = vo;
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// ------------------------------------------------------------------------------------------------
// Stream compaction.
// ------------------------------------------------------------------------------------------------
__global__ void compact_scatter_kernel(const uint32_t *in, const uint32_t *flag, const uint32_t *pos,
                                       uint32_t *out, int64_t n)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && flag[j]) out[pos[j]] = in ? in[j] : (uint32_t)j;
}

int compact_u32(const uint32_t *in, const uint32_t *flag, uint32_t *out, int64_t n,
                int64_t *count_host, hipStream_t s, const uint32_t *extra_dev, uint32_t *extra_host)
{
    *count_host = 0;
    if (n <= 0) return RAHT_OK;
    Scratch pos(sizeof(uint32_t) * ((size_t)n + 1), s);
    if (!pos.ok()) return RAHT_ERR_NOMEM;
    uint32_t *total = pos.as<uint32_t>() + n;
    int rc = exclusive_scan_u32(flag, pos.as<uint32_t>(), n, total, s);
    if (rc == RAHT_OK) {
        hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, in,
                           flag, pos.as<uint32_t>(), out, n);
        uint32_t t = 0;
        rc = read_back_u32(&t, total, 1, extra_host, extra_dev, extra_dev ? 1 : 0, s);     // (the caller's word rides along)
        *count_host = t;
    }
    return rc;
}

// ------------------------------------------------------------------------------------------------
// Starts of the runs of equal keys in a sorted key array (the voxelizer's voxel boundaries, voxelize_pc.py:114-118), in TWO
// launches: per-block counts of run starts straight from the keys, then every block adds up the counts of the blocks
// before it by itself, ranks its own starts and writes them -- as uint32, and (optionally) widened to int64 and together
// with the key of every run. (Flag array + generic scan + scatter + widening: five launches and 48 N more bytes.)
// ------------------------------------------------------------------------------------------------
// Items of a block are taken in rounds of 256 consecutive keys (lane = key: loads and stores coalesce); a key's predecessor
// comes from the lane below (lane 0 of every wave loads it).
__device__ __forceinline__ bool run_start_flag(const uint64_t *__restrict__ keys, int64_t i, int64_t n, uint64_t x)
{
    const int lane = threadIdx.x & 63;
    uint64_t prev = __shfl_up(x, 1, 64);
    if (lane == 0 && i > 0 && i < n) prev = keys[i - 1];
    return i < n && (i == 0 || x != prev);
}

__global__ __launch_bounds__(SCAN_THREADS) void run_count_kernel(const uint64_t *__restrict__ keys, int64_t n, uint32_t *__restrict__ blk)
{
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint64_t x[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) x[k] = (base + k * SCAN_THREADS < n) ? keys[base + k * SCAN_THREADS] : 0ull;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) c += run_start_flag(keys, base + k * SCAN_THREADS, n, x[k]) ? 1u : 0u;
    uint32_t tot;
    (void)block_excl_scan_256(c, &tot);
    if (threadIdx.x == 0) blk[blockIdx.x] = tot;
}

__global__ __launch_bounds__(SCAN_THREADS) void run_starts_kernel(const uint64_t *__restrict__ keys, int64_t n, const uint32_t *__restrict__ blk,
                                                                  uint32_t *__restrict__ starts, int64_t *__restrict__ starts64,
                                                                  uint64_t *__restrict__ run_keys, uint32_t *__restrict__ total)
{
    __shared__ uint32_t wc[SCAN_ITEMS * 4];               // run starts per (round, wave), memory order
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + threadIdx.x;
    uint64_t x[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) x[k] = (base + k * SCAN_THREADS < n) ? keys[base + k * SCAN_THREADS] : 0ull;
    uint32_t part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_THREADS) part += blk[b];
    uint32_t f = 0, rank[SCAN_ITEMS];
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        const bool fl = run_start_flag(keys, base + k * SCAN_THREADS, n, x[k]);
        const uint64_t bal = __ballot(fl);
        rank[k] = (uint32_t)__popcll(bal & lt);
        if (fl) f |= 1u << k;
        if (lane == 0) wc[k * 4 + wid] = (uint32_t)__popcll(bal);
    }
    uint32_t before;
    (void)block_excl_scan_256(part, &before);            // (its barriers also publish wc)
    uint32_t run = before;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = wc[k * 4 + w];
            if (w == wid && (f & (1u << k))) {
                const uint32_t pos = run + rank[k];
                const int64_t i = base + k * SCAN_THREADS;
                starts[pos] = (uint32_t)i;
                if (starts64) starts64[pos] = i;
                if (run_keys) run_keys[pos] = x[k];
            }
            run += c;
        }
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total = run;
}

int run_starts_u64(const uint64_t *keys_sorted, int64_t n, uint32_t *starts, int64_t *starts64, uint64_t *run_keys,
                   uint32_t *count_dev, hipStream_t s)
{
    if (n <= 0) return RAHT_OK;
    const int64_t nb = ceil_div(n, SCAN_BLOCK);
    Scratch ws(sizeof(uint32_t) * (size_t)nb, s);
    if (!ws.ok()) return RAHT_ERR_NOMEM;
    uint32_t *blk = ws.as<uint32_t>();
    hipLaunchKernelGGL(run_count_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, keys_sorted, n, blk);
    hipLaunchKernelGGL(run_starts_kernel, dim3((unsigned)nb), dim3(SCAN_THREADS), 0, s, keys_sorted, n, blk, starts, starts64, run_keys, count_dev);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// ------------------------------------------------------------------------------------------------
// Small device -> host read-backs through a polled mailbox in mapped pinned memory.
// ------------------------------------------------------------------------------------------------
static constexpr int MAILBOX_WORDS = 192;

__global__ void __launch_bounds__(64) mailbox_publish_kernel(const uint32_t *__restrict__ a, int na,
                                                             const uint32_t *__restrict__ b, int nb,
                                                             volatile uint32_t *box, uint32_t seq)
{
    // one wave: the data stores are complete (system scope) before lane 0 raises the sequence word
    for (int t = threadIdx.x; t < na; t += 64) box[1 + t] = a[t];
    for (int t = threadIdx.x; t < nb; t += 64) box[1 + na + t] = b[t];
    __threadfence_system();
    if (threadIdx.x == 0) box[0] = seq;
}

struct Mailbox {
    std::mutex mu;
    uint32_t *box = nullptr;      // [0] = sequence word, [1 ..] = payload
    uint32_t seq = 0;
};
static Mailbox &mailbox() { static Mailbox m; return m; }

int read_back_u32(uint32_t *dst_a, const uint32_t *dev_a, int na, uint32_t *dst_b, const uint32_t *dev_b, int nb,
                  hipStream_t s, const std::function<void()> &behind)
{
    bool behind_called = false;
    if (na < 0 || nb < 0 || na + nb > MAILBOX_WORDS) { set_error("read_back_u32: too many words"); return RAHT_ERR_INVALID; }
    Mailbox &m = mailbox();
    std::lock_guard<std::mutex> g(m.mu);
    if (!m.box) {
        // portable + mapped + coherent: the same pointer is valid on the host and on every device
        if (hipHostMalloc((void **)&m.box, sizeof(uint32_t) * (MAILBOX_WORDS + 1),
                          hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) {
            m.box = nullptr;
            (void)hipGetLastError();
        } else {
            m.box[0] = 0;
        }
    }
    bool done = false;
    if (m.box) {
        const uint32_t seq = ++m.seq ? m.seq : ++m.seq;            // never 0
        hipLaunchKernelGGL(mailbox_publish_kernel, dim3(1), dim3(64), 0, s, dev_a, na, dev_b, nb, m.box, seq);
        if (hipGetLastError() == hipSuccess) {
            if (behind) { behind(); behind_called = true; }
            volatile uint32_t *flag = m.box;
            for (uint32_t it = 1;; ++it) {
                if (*flag == seq) { done = true; break; }
                if ((it & 0xfffu) == 0) {
                    // a failed or finished stream ends the wait even if the word never arrives
                    const hipError_t q = hipStreamQuery(s);
                    if (q != hipErrorNotReady) { done = (q == hipSuccess && *flag == seq); break; }
                }
                __builtin_ia32_pause();
            }
            if (done) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                for (int t = 0; t < na; ++t) dst_a[t] = flag[1 + t];
                for (int t = 0; t < nb; ++t) dst_b[t] = flag[1 + na + t];
            }
        }
    }
    if (!done) {                                                     // no mailbox / stream error: the plain way
        if (behind && !behind_called) behind();
        hipError_t e = hipSuccess;
        if (na) e = hipMemcpyAsync(dst_a, dev_a, sizeof(uint32_t) * (size_t)na, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess && nb) e = hipMemcpyAsync(dst_b, dev_b, sizeof(uint32_t) * (size_t)nb, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { set_error("read-back: %s", hipGetErrorString(e)); return RAHT_ERR_HIP; }
    }
    return RAHT_OK;
}

}  // namespace raht

#ifdef RAHT_OS_CLOCKS
extern "C" int raht_debug_sort_clocks(unsigned long long *host_out, int n_tiles, int shift)
{
    if (shift >= 0) return hipMemcpyToSymbol(HIP_SYMBOL(raht::os_dbg_shift), &shift, sizeof(int)) == hipSuccess ? 0 : -4;
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(raht::os_dbg), sizeof(unsigned long long) * 8 * (size_t)n_tiles) == hipSuccess ? 0 : -4;
}
#endif

extern "C" int raht_release_cached_memory(void)
{
    auto &c = raht::dev_cache();
    std::vector<void *> blocks;
    {
        std::lock_guard<std::mutex> g(c.mu);
        for (auto &kv : c.free_blocks) blocks.push_back(kv.second);
        c.free_blocks.clear();
        c.cached = 0;
    }
    for (void *q : blocks) (void)hipFree(q);
    return RAHT_OK;
}
