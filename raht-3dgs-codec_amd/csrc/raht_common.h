// raht_common.h -- internal declarations shared by the HIP translation units of libraht_hip.so.
// gfx950 (MI355X) only: wave = 64 lanes, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>
#include <deque>
#include <exception>
#include <functional>
#include <new>
#include <vector>

#include "raht.h"

#define RAHT_WAVE 64
#define RAHT_MAX_LEVELS 64

namespace raht {

void set_error(const char *fmt, ...);

#define RAHT_HIP_CHECK(expr)                                                              \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            raht::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                    \
            return RAHT_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)

#define RAHT_RET(expr)                 \
    do {                               \
        int _r = (expr);               \
        if (_r != RAHT_OK) return _r;  \
    } while (0)

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// No C++ exception crosses the C ABI (raht.h): every extern "C" entry point that can allocate host
// memory (std::vector / std::deque growth, std::thread) runs its body through this.
template <typename F>
static inline int guarded(const char *what, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_error("%s: out of host memory", what);
        return RAHT_ERR_NOMEM;
    } catch (const std::exception &e) {
        set_error("%s: %s", what, e.what());
        return RAHT_ERR_INVALID;
    } catch (...) {
        set_error("%s: unknown C++ exception", what);
        return RAHT_ERR_INVALID;
    }
}

// ---- devices -------------------------------------------------------------------------------------
// One process per GPU is the intended use, but nothing below assumes it: every piece of cached device
// state (block cache, scratch pool, kernel attributes, CU counts) is keyed by the HIP device ordinal, a
// plan remembers the device it was built on, and every entry point that takes a plan refuses to run
// while another device is current (RAHT_ERR_INVALID) instead of mixing memory of two devices.
constexpr int RAHT_MAX_DEVICES = 64;
static inline int current_device()
{
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= RAHT_MAX_DEVICES) d = 0;
    return d;
}
// Makes `device` current for the lifetime of the object (plan destruction from a thread whose current
// device is another one: Python finalisers run wherever the interpreter happens to be).
class DeviceGuard {
public:
    explicit DeviceGuard(int device) : prev_(current_device()), dev_(device)
    {
        if (prev_ != dev_) (void)hipSetDevice(dev_);
    }
    ~DeviceGuard() { if (prev_ != dev_) (void)hipSetDevice(prev_); }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
private:
    int prev_, dev_;
};
// Once-per-device flag for hipFuncSetAttribute calls (the attribute belongs to the function ON the current
// device): true exactly the first time it is asked about a device. Idempotent work only -- two threads racing
// on the first call may both get true.
struct PerDeviceOnce {
    bool done[RAHT_MAX_DEVICES] = {};
    bool first(int device) { if (done[device]) return false; done[device] = true; return true; }
};

// Device scratch memory from a thread-local grow-only pool (one per device): plan / sort / voxelizer
// temporaries are reused across calls instead of paying hipMalloc + hipFree (each a device
// synchronisation) per use. A block is released when its Scratch object dies, i.e. possibly while the kernels
// that use it are still queued on `stream`: reuse is STREAM-ORDERED. Every block remembers the stream of its
// last user; handing it to work on ANOTHER stream (a caller that moves between streams on one host thread, e.g.
// side streams around collectives) first waits for the device to drain -- rare, and the only way to be right
// without an event per release. Blocks belong to the device that was current when they were allocated and are
// only handed out while that device is current.
class Scratch {
public:
    Scratch(size_t bytes, hipStream_t stream);
    ~Scratch();
    Scratch(const Scratch &) = delete;
    Scratch &operator=(const Scratch &) = delete;
    void *ptr() const { return p_; }
    template <typename T> T *as() const { return (T *)p_; }
    bool ok() const { return p_ != nullptr; }
private:
    void *p_ = nullptr;
    int slot_ = -1;
};

// Long-lived device blocks (plan arrays, schedules, workspaces) come from a process-wide cache of freed
// blocks in ~12 % size classes: a codec that builds one plan per frame pays hipMalloc / hipFree (each a
// device synchronisation, ~0.1-0.3 ms for the six N-sized plan arrays) only until the cache is warm.
// dev_free keeps at most RAHT_POOL_MAX_BYTES (default 8 GiB) cached; raht_release_cached_memory() empties it.
// The cache is keyed by (device, size class): a block freed by a plan on device 0 is never handed to a
// plan on device 1.
hipError_t dev_malloc(void **p, size_t bytes);
template <typename T> inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc((void **)p, bytes); }
void dev_free(void *p);

// ---- device primitives (scan_sort.hip) ---------------------------------------------------------
// Exclusive prefix sum of n uint32 values, in place allowed (out may equal in). `total` (device
// uint32*, may be NULL) receives the sum. Allocates its own small workspace (plan/voxelize time
// only -- never called from the transform entry points).
int exclusive_scan_u32(const uint32_t *in, uint32_t *out, int64_t n, uint32_t *total, hipStream_t s);

// One stable LSD radix pass on `bits`-wide digits (bits <= 8) taken at `shift`.
// keys are uint64; payload is uint32. keys_out / vals_in / vals_out may be NULL
// (vals_in == NULL means payload = original index).
int radix_pass_u64(const uint64_t *keys_in, const uint32_t *vals_in, uint64_t *keys_out,
                   uint32_t *vals_out, int64_t n, int shift, int bits, hipStream_t s);
// The whole stable sort of (key, original index) by the low nbits key bits in npass + 2 launches: one histogram launch for
// every digit, one for the digit bases, one launch per digit pass whose tiles chain their offsets by decoupled look-back
// (scan_sort.hip). tmp_keys / tmp_idx: N-sized ping-pong buffers. err_dev: device word, zeroed by the first launch and set
// when a tile gave up waiting for its predecessors (bounded waits: the grid always drains) -- the caller reads it back
// once the stream has drained and falls back to the pass-by-pass sort. Returns 1 (nothing enqueued) when the input does
// not fit (more than 8 digit passes, N >= 2^30) or RAHT_SORT_ONESWEEP=0. idx64_out (may be NULL): the indices once more as
// int64, written by the last pass. grid (may be NULL): keys_in is not an input but the voxel keys of the cloud's points
// (raht_device.h: vox_key), which the histogram launch computes AND stores there on its way.
struct VoxGrid;
int sort_pairs_onesweep(const uint64_t *keys_in, int64_t n, int nbits, uint64_t *keys_out, uint32_t *idx_out, uint64_t *tmp_keys,
                        uint32_t *tmp_idx, uint32_t *err_dev, hipStream_t s, int64_t *idx64_out = nullptr,
                        const VoxGrid *grid = nullptr);
// Same for uint8 bucket ids (< 2^bits); produces the stable permutation and, optionally, the
// start offset of every bucket (bucket_off: device uint32[(1<<bits)+1]).
int bucket_sort_u8(const uint8_t *bucket, uint32_t *perm_out, int64_t n, int bits,
                   uint32_t *bucket_off, hipStream_t s);

// Host read-back of a few 32-bit words produced on `s` (na + nb <= 192; b may be NULL): a one-wave kernel
// stores them into a mapped pinned mailbox and the host polls it -- 9 us instead of the 22 us of a
// pageable hipMemcpyAsync + hipStreamSynchronize (tools/native/probe_readback.hip). Returns once the
// words (and therefore all earlier work on `s`) are complete. `behind` (may be empty) is called once the read-back has been
// enqueued and before the host starts waiting: work it enqueues on `s` runs while the host waits and is NOT waited for.
int read_back_u32(uint32_t *dst_a, const uint32_t *dev_a, int na, uint32_t *dst_b, const uint32_t *dev_b, int nb,
                  hipStream_t s, const std::function<void()> &behind = std::function<void()>());

// out[k] = in[j] (or j when in == NULL) for the k-th j with flag[j] != 0. *count_host gets the
// number of kept items (synchronises the stream).
int compact_u32(const uint32_t *in, const uint32_t *flag, uint32_t *out, int64_t n,
                int64_t *count_host, hipStream_t s, const uint32_t *extra_dev = nullptr, uint32_t *extra_host = nullptr);

// starts[k] = index of the first element of the k-th run of equal keys in a SORTED key array; optionally the same as int64
// (starts64) and the key of every run (run_keys). *count_dev (DEVICE word) = number of runs. Two launches, no host round trip:
// the voxelizer's next kernel reads the count on the device.
int run_starts_u64(const uint64_t *keys_sorted, int64_t n, uint32_t *starts, int64_t *starts64, uint64_t *run_keys,
                   uint32_t *count_dev, hipStream_t s);

// ---- plan (plan.hip) ---------------------------------------------------------------------------
constexpr int RAHT_TOP_MAX_ROWS = 8192;   // entries the TOP stage can hold (16 bytes each in LDS)

struct Stage {
    int64_t n_entries = 0;   // active rows entering this stage
    int64_t n_tiles = 0;
    uint32_t *rows = nullptr;      // device; nullptr for stage 0 (identity)
    uint32_t *surv_off = nullptr;  // device uint32[n_tiles + 1]: index (in the NEXT stage's entry
                                   // list) of the first survivor of every tile; nullptr on the last stage
    void *ws = nullptr;            // device workspace holding this stage's entries (stages >= 1)
    size_t ws_inv_off = 0;         // byte offset of the INVERSE direction's own copy of the workspace inside `ws` (0: both directions
                                   // share one -- the default; raht_plan_set_concurrent_directions gives each its own)
    int tile_rows = 0;             // rows per tile of THIS stage
    // entry-ordered copies of the plan metadata (stages >= 1): one contiguous, single-latency load
    // per tile instead of rows[] -> wl/wr/lvl/inv_order[row] chains. nullptr on stage 0 (entry = row).
    int32_t *e_wl = nullptr, *e_wr = nullptr;
    uint8_t *e_lvl = nullptr;
    uint32_t *e_pos = nullptr;
    // tile stages (stage 0 included): HEIGHT of every entry's butterfly inside its tile's merge tree (1 = both children are
    // single entries; 0 = the entry survives the tile). The tile kernels run their butterfly rounds by height, not by
    // binary level: a butterfly's height is 1 + the larger height of its two children, so rounds by ascending height
    // respect every dependency, butterflies of one height are independent, and a tile needs as many rounds as its tree
    // is high (9.3 on average at 184 rows) instead of one per binary level present (13.9). Heights are < 64: levels
    // strictly increase along a dependency chain.
    uint8_t *e_ht = nullptr;
    // forward chaining of the later tile stages (transform.hip, tile_kernel_chain): entries of this stage that have arrived
    // from the stage below, per tile; all zero between launches (the workgroup that completes a tile resets its counter)
    uint32_t *arrive = nullptr;
    // TOP stage (the last one, when at most `top_rows` entries are left): ONE launch of top_kernel
    // finishes the tree. A workgroup per 16-byte channel chunk keeps all entries in LDS and walks
    // the butterflies level by level from this precomputed list, sorted by level:
    bool is_top = false;
    uint32_t n_merges = 0;
    uint32_t *t_pj = nullptr;      // device [n_merges]: partner entry | own entry << 16
    float *t_ab32 = nullptr;       // device [2 * n_merges]: a, b as the float32 transform uses them
    double *t_ab64 = nullptr;      // device [2 * n_merges]: a, b in float64
    uint32_t *t_root = nullptr;    // device [n_entries]: rank among the roots (root buffer row), ~0u = not a root
    uint32_t t_loff[65] = {0};     // host: first merge of every binary level (t_loff[63] = n_merges)
    uint32_t t_lev_host[2 * 64] = {0};
    uint32_t *t_lev = nullptr;     // device [2 * t_nlev]: (first, end) butterfly of every NON-EMPTY level, ascending
    int t_nlev = 0;
    int t_nbig = 0;                // the first t_nbig of them run on the whole workgroup (a barrier each); the rest
                                   // hold <= 64 butterflies each and are chained by ONE wave without barriers
    uint32_t t_small_start = 0;    // first butterfly of the chained part (its records are staged in LDS)
};

struct Schedule {
    int tile_rows = 0;         // rows per tile of stage 0 (cache key, with tail_rows)
    int tail_rows = 0;         // rows per tile of the stages >= 1
    int final_rows = 0;        // a stage with at most this many entries becomes the TOP stage (one launch finishes the tree)
    bool valid = false;        // false: tile stages cannot finish the tree -> use the level engine
    std::vector<Stage> stages;
    size_t ws_row_bytes = 0;   // bytes per workspace row currently allocated (D * elem_size)
    bool ws_split = false;     // the workspaces currently allocated hold one copy per direction
};

}  // namespace raht

struct raht_plan {
    int device = 0;              // HIP device ordinal the plan (and every block it owns) lives on
    int64_t N = 0;
    int nbits = 0;
    int max_level = -1;          // highest binary level with a pair (-1 when N == 1)
    uint64_t *keys = nullptr;    // device, sorted Morton keys
    bool keys_borrowed = false;  // keys is the CALLER's array (raht_plan_create_from_keys_borrowed): never freed here
    uint8_t *lvl = nullptr;      // device, 255 for row 0
    int32_t *wl = nullptr;       // device
    int32_t *wr = nullptr;       // device
    int64_t *wsum = nullptr;     // device int64[N+1] prefix of leaf weights, or nullptr (all ones)
    uint32_t *order = nullptr;   // device, order_RAGFT
    uint32_t *inv_order = nullptr;  // device, inverse permutation: inv_order[order[k]] = k
    uint32_t *level_rows = nullptr;          // device, rows 1..N-1 stably sorted by lvl (LEVEL engine only: ensure_level_rows)
    uint32_t level_off[RAHT_MAX_LEVELS + 1]; // host, start of every level inside level_rows
    int top_level = 64;          // butterflies at binary levels >= top_level are NOT performed
    int64_t n_roots = 1;         // row 0 plus every row whose level is >= top_level
    uint32_t *root_rows = nullptr;   // device, ascending
    void *root_buf = nullptr;    // caller-owned device buffer (n_roots x D), see raht_plan_set_root_buffer
    uint32_t *row_map = nullptr; // device [N] or nullptr: plan row i lives in matrix row row_map[i] (raht_plan_set_row_map)
    int64_t map_rows = 0;        // rows of the mapped matrices
    int engine = RAHT_ENGINE_TILE;
    int tile_rows_override = 0;
    int tail_rows_override = 0;  // rows per tile of the later stages (0 = automatic)
    int tail_chunk_override = 0; // channels per chunk of the later stages (0 = automatic)
    int final_rows_override = 0; // single-tile finishing stage up to this many entries (0 = automatic)
    // plan construction only: words the next host read-back on the build stream should fetch along (the schedule
    // builder's single read-back also carries the plan's error word and level histogram: one round trip per plan)
    const uint32_t *pend_dev = nullptr; uint32_t *pend_host = nullptr; int pend_n = 0;
    hipEvent_t ev_before = nullptr, ev_after = nullptr;   // profiling: recorded around the stage-0 launch
    int max_stages = 24;         // a tile schedule that needs more stages than this is abandoned (level engine)
    bool split_ws = false;       // one workspace set per direction: a forward and an inverse call may run at the same time
    std::deque<raht::Schedule> schedules;    // cache keyed by tile geometry; a deque: references handed out by
                                             // get_schedule stay valid when another geometry is added
    std::vector<uint8_t> lvl_host;           // lazily downloaded for export_level
};

namespace raht {
// RAHT_OK when the plan's device is the calling thread's current device (see "devices" above).
int check_plan_device(const raht_plan *plan, const char *what);
// Weights of the two children of the butterfly at row i (left extent l, right extent r): row counts,
// or sums of leaf weights (weighted plans: prefix populations of a sharded scene).
__device__ __forceinline__ void pair_weights(int64_t i, int l, int r, const int64_t *wsum, double &w0, double &w1)
{
    if (wsum) {
        w0 = (double)(wsum[i] - wsum[i - l]);
        w1 = (double)(wsum[i + r] - wsum[i]);
    } else {
        w0 = (double)l;
        w1 = (double)r;
    }
}

// The level engine's per-level row lists (built on first use).
int ensure_level_rows(raht_plan *plan, hipStream_t s);
// Tile schedule for `tile_rows` rows per tile (built on first use, cached in the plan).
int get_schedule(raht_plan *plan, int tile_rows, int tail_rows, int final_rows, hipStream_t s, Schedule **out);
// Tile geometry of the later (small, latency-bound) stages: as many rows as one workgroup per CU can
// stage, in channel chunks.
void pick_tail_geometry(const raht_plan *plan, int elem_size, int D, int stage0_rows, int *tail_rows, int *tail_chunk,
                        int *final_rows);
// Make sure the per-stage workspaces of `sc` hold rows of at least row_bytes bytes (allocates on
// first use / growth only).
int ensure_workspace(Schedule *sc, size_t row_bytes, bool split = false);
// stage workspace as a direction sees it
static inline void *stage_ws(const Stage &st, bool inverse) { return st.ws ? (void *)((char *)st.ws + (inverse ? st.ws_inv_off : 0)) : nullptr; }
// Rows per LDS tile for an element size / channel count (0 = does not fit).
int pick_tile_rows(const raht_plan *plan, int elem_size, int chunk_channels);
int pick_chunk_channels(int elem_size, int D);

size_t tile_lds_bytes(int R, int elem_size, int Dc, bool ident, bool qm);
}  // namespace raht
