// transform.hip -- forward / inverse RAHT butterflies on gfx950.
//
// Replaces RAHT2_optimized (reference python/RAHT.py:252-336) and inverse_RAHT_optimized
// (reference python/iRAHT.py:40-114). Two kernel families:
//
//  * LEVEL engine: one launch per binary level (= per octree level per axis). A lane group
//    (>= D lanes, power of two) owns one sibling pair; the group's first lane evaluates
//    a = sqrt(w0/(w0+w1)), b = sqrt(w1/(w0+w1)) in float64 from the integer occupancy weights and
//    broadcasts them with wavefront shuffles; lanes map to attribute channels, so every row access
//    is one coalesced segment. 2 row reads + 2 row writes per pair: HBM traffic ~2x the ideal.
//
//  * TILE engine (default): a workgroup stages a run of R Morton-contiguous rows (all channels) in
//    LDS, performs EVERY butterfly whose subtree lies inside the run (levels ascending; a barrier
//    per level that needs the whole workgroup, none between levels one wave instruction can take),
//    and writes the run back once. Rows whose subtree crosses the run boundary (4.3 % at R = 184)
//    are the entries of the next, ~20x smaller stage. HBM traffic ~1.05x the ideal (read C once,
//    write T once). Stage membership is a pure function of the plan, so it is precomputed
//    (plan.hip). Every data-touching phase works on 16-byte row chunks (raht_device.h).
//
//  * TOP stage: once <= 4096 entries are left, one launch of top_kernel finishes the tree from
//    butterflies resolved at schedule time.
//
// Bandwidth-bound integer/fp32 work: no MFMA by design (3 flops per 4 bytes).
#include "raht_common.h"
#include "raht_device.h"
#include "tile_engine.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace raht {

size_t tile_lds_bytes(int R, int elem_size, int Dc, bool ident, bool qm);

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float4 type; static constexpr int n = 4; };
template <> struct Vec16<double> { typedef double2 type; static constexpr int n = 2; };

// ------------------------------------------------------------------------------------------------
// strided row copy (only used when the level engine needs dst = src first)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void copy_rows_kernel(const T *__restrict__ src, int64_t lds, T *__restrict__ dst, int64_t ldd,
                                 int64_t N, int D)
{
    const int64_t total = N * D;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = e / D;
        const int c = (int)(e - i * D);
        dst[i * ldd + c] = src[i * lds + c];
    }
}

// ------------------------------------------------------------------------------------------------
// LEVEL engine
// ------------------------------------------------------------------------------------------------
template <typename T, bool INV>
__global__ __launch_bounds__(256) void level_pass_kernel(T *__restrict__ data, int64_t ld, int D,
                                                         const uint32_t *__restrict__ level_rows,
                                                         uint32_t count, const int32_t *__restrict__ wl,
                                                         const int32_t *__restrict__ wr,
                                                         const int64_t *__restrict__ wsum, int lp_shift,
                                                         const uint32_t *__restrict__ row_map)
{
    const int lane = threadIdx.x & 63;
    const int Lp = 1 << lp_shift;
    const int gpw = 64 >> lp_shift;                       // pairs per wave step
    const int g = lane >> lp_shift, c0 = lane & (Lp - 1);
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nwaves = (gridDim.x * blockDim.x) >> 6;
    for (uint32_t mb = wave * gpw; mb < count; mb += nwaves * gpw) {
        const uint32_t m = mb + g;
        const bool act = m < count;
        int64_t i = 0;
        int l = 1;
        T a = 0, b = 0;
        if (act) {
            i = level_rows[m];
            l = wl[i];
            if (c0 == 0) {
                double w0, w1;
                pair_weights(i, l, wr[i], wsum, w0, w1);
                const double den = w0 + w1;
                a = (T)sqrt(w0 / den);                    // RAHT.py:321-322
                b = (T)sqrt(w1 / den);
            }
        }
        // occupancy-weight butterfly coefficients: one evaluation per pair, wave-shuffle broadcast
        a = __shfl(a, g << lp_shift, 64);
        b = __shfl(b, g << lp_shift, 64);
        if (act) {
            // row_map (raht_plan_set_row_map): plan row -> matrix row, e.g. the padded all-gather buffer of a sharded scene
            T *r0 = data + (row_map ? (int64_t)row_map[i - l] : i - l) * ld;
            T *r1 = data + (row_map ? (int64_t)row_map[i] : i) * ld;
            for (int c = c0; c < D; c += Lp) {
                const T x0 = r0[c], x1 = r1[c];
                if (!INV) {                               // RAHT.py:331-332
                    r0[c] = a * x0 + b * x1;
                    r1[c] = a * x1 - b * x0;
                } else {                                  // iRAHT.py:108-109
                    r0[c] = a * x0 - b * x1;
                    r1[c] = b * x0 + a * x1;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// TILE engine
//
// Data flow (K stages; stage k has n_k "entries" = rows still carrying a live low-pass value,
// n_0 = N; a tile = R consecutive entries):
//
//   forward  stage k: read its entries CONTIGUOUSLY (k = 0: the caller's C, k >= 1: workspace
//            ws_k, entry order) -> butterflies in LDS -> rows finalised here go to T[row] (or,
//            fused, quantized to Q[inv_order[row]]); the few survivors go, compacted, to ws_{k+1}.
//   inverse  stage k (K-1 ... 0): rows finalised at this stage come from T[row] (or, fused,
//            dequantized from Q[inv_order[row]]); its survivors were produced by stage k+1 and sit
//            contiguously in ws_{k+1} (prefetched at kernel start) -> butterflies, levels
//            descending -> the whole tile is written CONTIGUOUSLY to ws_k (k >= 1) or to C (k = 0).
//
// So every large transfer is a coalesced contiguous span; only finalised rows of stages >= 1 (a
// few % of N) and the fused Q rows are row-granular (236-byte segments at D = 59).
// ------------------------------------------------------------------------------------------------

// Ablation branches exist only in a profiling build (make ABLATE=1, tools/ablate.sh)
#ifdef RAHT_ABLATE
#define TILE_DBG(A) ((A).dbg)
#else
#define TILE_DBG(A) 0
#endif

// Profiling build only (-DRAHT_PHASE_CLOCKS, tools/phase_clocks.py): thread 0 of the first workgroups stamps
// the shader clock at the phase boundaries of its first tile.
#ifdef RAHT_PHASE_CLOCKS
constexpr int PHASE_CLK_TILES = 4096, PHASE_CLK_SLOTS = 10;
__device__ unsigned long long g_phase_clk[PHASE_CLK_TILES][PHASE_CLK_SLOTS];
#define PHASE_STAMP(k) do { if (threadIdx.x == 0 && tile_id < PHASE_CLK_TILES) g_phase_clk[tile_id][k] = __builtin_readcyclecounter(); } while (0)
#else
#define PHASE_STAMP(k) do { } while (0)
#endif

// 16 bytes written THROUGH the XCD's L2 (sc1: agent scope): visible to the other XCDs once the store has been acknowledged
// (s_waitcnt vmcnt(0)), without the L2 write-back an agent-scope release fence costs
template <typename T>
__device__ __forceinline__ void st_chunk_wt(T *p, const RegChunk<T> &x)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v;
    __builtin_memcpy(&v, &x, 16);
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

// IDENT = true is stage 0 (entries are the rows themselves: the HBM-heavy launch); IDENT = false
// are the later, much smaller stages. QM = true fuses quantize+reorder (forward) / un-reorder+
// dequantize (inverse). Separate instantiations keep them apart in rocprof kernel statistics.
// launch bounds: three 512-thread workgroups per CU = 6 waves per SIMD for float32 (<= 80 VGPRs)
//
// Lane mapping. The kernel is bound by instruction issue, not by HBM (rocprofv3 SQ_INSTS_VALU: the
// one-channel-per-lane version spent ~9.4k vector instructions per 192 x 59 tile and kept the VALUs
// ~70 % busy at 3.8 TB/s), so every data-touching phase works on 16-byte chunks: a lane owns VN
// consecutive channels of a row, G = 2^lg >= ceil(Dc / VN) lanes cover a row, and one wave
// instruction moves 64 / G rows (59 float channels: 15 chunks, G = 16, 4 rows). LDS rows are padded
// to Dp = Dc rounded up to VN so that every chunk is a 16-byte-aligned ds_read/write_b128; global
// rows need element alignment only. A butterfly is one ds_read_b128 per operand and lane.
//
// tile_body is the kernel; tile_kernel runs it for ONE scene (workgroup b takes tiles b, b + gridDim.x, ...), tile_kernel_batch
// for several scenes in one launch (workgroup b takes ONE tile of the scene whose tile range holds b).
template <typename T, bool INV, bool IDENT, bool QM, int SLOTS, bool WT = false, bool SQ = false, bool MULTI = false>
__device__ __forceinline__ void tile_body(const TileArgs<T> &A, const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type &ST,
                                          const int64_t first_tile, const int64_t tile_stride, const int chunk_y, const MultiQ *MQ = nullptr)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef RegChunk<T> V16;
    typedef typename std::conditional<QM, int32_t, T>::type RawT;       // element type of the inverse's input rows
    typedef RegChunk<RawT> RawChunk;
    constexpr int VN = 16 / sizeof(T);
    // float64 rows with fused quantization (the reference's own precision): a 16-byte chunk is 2 coefficients = 2
    // integers, so Q rows do not have the tile's chunk layout -- the forward stores 8 bytes per lane, the inverse gathers
    // 8 bytes per lane through registers and converts them on the way into LDS (no LDS-direct load, no lazy conversion)
    constexpr bool QM64 = QM && sizeof(T) == 8;
    typedef typename StepsFor<T>::elem StepT;
    const int R = A.R;
    const int tid0 = threadIdx.x;
    const int nthreads = blockDim.x, nw = nthreads >> 6;
    const int c_base = chunk_y * A.Dc;
    const int Dc = min(A.Dc, A.D - c_base);
    const int Dp = A.Dp;                                  // LDS row stride in elements
    const int lg = A.lg, lr = 6 - A.lg;                   // log2(lanes per row), log2(rows per wave instruction)
    const int NC = (Dc + VN - 1) / VN;                    // chunks per row
    // Row loads go STRAIGHT to LDS (glds16), lane-linear: chunk c = 64 * instruction + lane of the tile's nt * NCp chunk
    // places sits at tile + 16 * c (an LDS row is NCp whole chunks; NCp > NC only in the last channel block of a chunked
    // stage, whose rows end early: those lanes load nothing). No register holds a row between HBM and LDS, so the loads
    // stay in flight across the metadata phases for free (kept in registers they cost 24 VGPRs: the forward kernels sat at
    // the 80-register limit of three workgroups per CU with it, the gathering inverses spilled and had to wait for their
    // rows on the spot -- the one exposed HBM round trip of the round-1 step).
    //   forward, plain inverse of stage 0: the input is one contiguous span (C / ws_k / T), issued at tile start (P0b);
    //   other inverses: a GATHER (T rows of the later stages, Q rows when fused) whose addresses need the plan metadata,
    //       issued after sync #1. The fused inverse's integers are dequantized where they are consumed: in the butterfly
    //       that reads the row as its high-pass operand (every row finalised in a tile is read that way exactly once,
    //       before anything is written to it);
    //   the forward waits for its rows in front of the butterflies (sync #4), the inverse at sync #3 (P3b writes the
    //   survivors' values from the stage above over what the gather put in their slots).
    constexpr bool GATHER = INV && !(IDENT && !QM);
    const int NCp = Dp / VN;
    const uint32_t NCm = ((1u << 20) + (uint32_t)NCp - 1) / (uint32_t)NCp;   // c / NCp == (c * NCm) >> 20 for c < 2^15
    // workgroup barrier that leaves the LDS-direct loads in flight (__syncthreads() drains the vector memory counter)
    auto sync_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto sync_landed = [&]() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    // ---- LDS carve-up (must match tile_lds_bytes) ----
    size_t off = (size_t)R * Dp * sizeof(T);
    T *tile = (T *)smem;
    MRec<T> *mrec = (MRec<T> *)(smem + off); off += (size_t)R * sizeof(MRec<T>);
    int32_t *srow = (int32_t *)(smem + off); if (!IDENT) off += (size_t)R * 4;   // stage 0: row = e0 + slot
    int32_t *sdst = (int32_t *)(smem + off); if (QM) off += (size_t)R * 4;       // position of the row in Q
    uint8_t *sflag = (uint8_t *)(smem + off); off += (size_t)R;      // 0 survivor, 1 merged here, 2 final root
    off = (off + 15) & ~(size_t)15;
    uint32_t *hist = (uint32_t *)(smem + off);            // [64]
    uint32_t *loff = hist + 64;                           // [64]
    uint32_t *cursor = hist + 128;                        // [64]
    uint32_t *scnt = hist + 196;                          // [32] survivors per (slot chunk, wave)
    off += 1024;
    uint16_t *ssurv = (uint16_t *)(smem + off);           // [R] slots of this tile's survivors, by rank
    off += ((size_t)R * 2 + 15) & ~(size_t)15;
    T *spre = (T *)(smem + off);                          // inverse: [TILE_PRE_ROWS * Dp] survivor prefetch

    // ---- persistent loop over this workgroup's tiles; metadata of the next tile is prefetched ----
    const int64_t n_tiles = (A.n_entries + R - 1) / R;
    TileMeta<SLOTS> M;
    // coff: the chunk's place in the LDS row; goff: its first channel in global rows (last chunk: see ld_chunk)
    auto lane_geom = [&](int tid, int &lane, int &wid, int &g, int &coff, int &goff, bool &active) {
        lane = tid & 63;
        wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        g = lane >> lg;
        const int c4 = lane & ((1 << lg) - 1);
        active = c4 < NC;
        const int c4c = min(c4, NC - 1);                  // idle lanes shadow the last chunk (loads stay valid)
        coff = c4c * VN;
        goff = c_base + min(coff, Dc - VN);
    };
    // quantization steps of this lane's channels (and their refined reciprocals: once per channel instead of once per
    // coefficient): fetched where they are used -- in front of the write-back (forward) or of the row gather
    // (inverse) -- not at kernel start: eight registers less across the metadata phases
    StepT my_step[VN];
    float my_rcp[VN];
    // (`ln` = the lane id from the tile loop's OPAQUE copy of the thread id: computed from tid0 the eight step-table
    // addresses are loop invariants that hipcc hoists to kernel start and spills -- and any scratch use at all cost the
    // fused inverse 0.27 -> 0.35 ms)
    auto load_steps = [&](int ln) {
        if constexpr (QM) {
            const int c4c = min(ln & ((1 << lg) - 1), NC - 1);
            const int g0 = c_base + min(c4c * VN, Dc - VN);
#pragma unroll
            for (int i = 0; i < VN; ++i) {
                my_step[i] = ST.v[ST.n == 1 ? 0 : g0 + i];
                if constexpr (!QM64) my_rcp[i] = refined_rcp(my_step[i]);
            }
        }
    };

    if (first_tile < n_tiles) load_tile_meta<T, IDENT, QM, SLOTS>(A, first_tile, tid0, nthreads, M);
    for (int64_t tile_id = first_tile; tile_id < n_tiles; tile_id += tile_stride) {
    // Re-derive the lane-dependent indices every iteration from an opaque copy of the thread id:
    // otherwise the compiler hoists dozens of lane-dependent addresses out of this long loop body
    // and spills them (register budget: 80 VGPRs for three workgroups per CU).
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    int lane, wid, g, coff, goff; bool active;
    lane_geom(tid, lane, wid, g, coff, goff, active);
    const int64_t e0 = tile_id * R;
    const int nt = (int)min((int64_t)R, A.n_entries - e0);
    const int64_t start_row = M.start_row, end_row = M.end_row;
    const uint32_t surv_base = (uint32_t)__builtin_amdgcn_readlane((int)M.surv_raw, 0);
    const uint32_t surv_cnt = (uint32_t)__builtin_amdgcn_readlane((int)M.surv_raw, 1) - surv_base;
    int32_t m_row[SLOTS], m_wl[SLOTS], m_wr[SLOTS], m_pos[SLOTS];
    int m_lv[SLOTS], m_ht[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        m_row[s] = M.row[s]; m_wl[s] = M.wl[s]; m_wr[s] = M.wr[s]; m_pos[s] = M.pos[s]; m_lv[s] = M.lv[s]; m_ht[s] = M.ht[s] & 63;
    }
    if (tid < 64) hist[tid] = 0;

    // Predication. Every LDS <-> global row loop below runs inside ONE `if (active)` region (lanes beyond the row's
    // last chunk sit the whole loop out) and addresses rows past the end of the tile as its last row: those lanes
    // move the same bytes to the same place as the lane that owns the row, so no per-chunk exec-mask
    // juggling is needed (it was 4 scalar instructions and a branch per chunk; the kernel issues nearly
    // as many scalar as vector instructions).

    PHASE_STAMP(0);
    // all chunks of `rows` rows, row jr's chunk ch from src(jr) + its channels, to the LDS rows at `dst`
    auto load_rows = [&](auto stream, const T *dst, int rows, auto src) {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)dst;
        const int total = rows * NCp;
        for (int it = wid; (it << 6) < total; it += nw) {
            const int c = (it << 6) + lane;
            const int jr = (int)(((uint32_t)c * NCm) >> 20), ch = c - jr * NCp;
            if (c < total && ch < NC)
                glds16<decltype(stream)::value>(src(jr, (uint32_t)(c_base + min(ch * VN, Dc - VN))), lds0 + ((uint32_t)it << 10));
        }
    };
    // ---- P0b. row transfers whose addresses do not depend on the plan metadata ----
    if constexpr (!GATHER) {
        // forward: this stage's entries, entry order (C or ws_k); plain inverse of stage 0: T rows [e0, e0+nt)
        const uint32_t lds = (uint32_t)(INV ? A.ld_fin : A.ld_in);
        const T *src = (INV ? (const T *)A.fin : A.in) + e0 * (int64_t)lds;      // wave-uniform
        load_rows(std::integral_constant<int, (WT && !IDENT) ? 2 : (IDENT ? 1 : 0)>(), tile, nt,      // stage 0: C (or T) itself, touched once
                  [&](int jr, uint32_t go) { return row_at(src, (uint32_t)jr, lds, go); });
    }
    if (INV) {
        // survivors of this tile were produced by the stage above: one contiguous chunk of ws_{k+1}
        // (the top stage has no stage above it: its survivors are the roots, handled in P3b)
        const int npre = A.last_stage ? 0 : (int)min(surv_cnt, (uint32_t)TILE_PRE_ROWS);
        load_rows(std::false_type(), spre, npre, [&](int q, uint32_t go) {
            return row_at((const T *)A.wsn + (int64_t)surv_base * A.ld_ws, (uint32_t)q, (uint32_t)A.ld_ws, go); });
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        if (j < nt) { if (!IDENT) srow[j] = m_row[s]; if (QM && INV) sdst[j] = m_pos[s]; }
    }
    PHASE_STAMP(1);
    sync_lds();                                                            // sync #1
    PHASE_STAMP(2);

    // inverse of the later stages / fused inverse: gather every slot's coefficient row now (survivor slots get
    // overwritten in P3b) -- the addresses need srow / sdst
    if constexpr (GATHER && QM64) {
        if (Dc >= 4) {
            // float64 rows, int32 coefficients: the integer rows go STRAIGHT to LDS as well (round 3; rounds 1-2 gathered
            // them through registers and waited for them on the spot) -- packed, NCi 16-byte chunks per row, at the start
            // of the tile (they take half the room of the float64 rows they will become), lane-linear like load_rows.
            // They are converted after they have landed (P2b below). Chunk ch of a row holds its integers
            // [min(4 ch, Dc - 4), + 4): the last chunk is the 16 bytes that END the row.
            const int NCi = (Dc + 3) >> 2;
            const uint32_t NCim = ((1u << 20) + (uint32_t)NCi - 1) / (uint32_t)NCi;
            const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)tile;
            const int total = nt * NCi;
            for (int it = wid; (it << 6) < total; it += nw) {
                const int c = (it << 6) + lane;
                const int jr = (int)(((uint32_t)c * NCim) >> 20), ch = c - jr * NCi;
                if (c < total)
                    glds16<true>(row_far((const int32_t *)A.Q, (uint32_t)sdst[jr], (uint32_t)A.ldq, (uint32_t)(c_base + min(ch * 4, Dc - 4))), lds0 + ((uint32_t)it << 10));
            }
        } else {
            // rows of fewer than four integers: through registers
            load_steps(lane);
            constexpr int GU = 4;                                 // rows in flight per lane
            if (active) for (int it0 = wid; (it0 << lr) < nt; it0 += nw * GU) {
                int32_t q[GU][VN];
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int j = min(((it0 + u * nw) << lr) + g, nt - 1);
                    ld_ints<VN>(row_far((const int32_t *)A.Q, (uint32_t)sdst[j], (uint32_t)A.ldq, (uint32_t)goff), q[u]);
                }
#pragma unroll
                for (int u = 0; u < GU; ++u) {
                    const int j = min(((it0 + u * nw) << lr) + g, nt - 1);
                    V16 x;
#pragma unroll
                    for (int i = 0; i < VN; ++i) x.v[i] = (T)q[u][i] * (T)my_step[i];                 // encode_3dgs.py:261
                    *(V16 *)&tile[__mul24(j, Dp) + coff] = x;
                }
            }
        }
    } else if constexpr (GATHER) {
        load_rows(std::true_type(), tile, nt, [&](int jr, uint32_t go) {
            if constexpr (QM) return (const void *)row_far((const RawT *)A.Q, (uint32_t)sdst[jr], (uint32_t)A.ldq, go);
            else return (const void *)row_far((const RawT *)A.fin, (uint32_t)srow[jr], (uint32_t)A.ld_fin, go);
        });
    }

    // ---- P1. which slots merge inside this tile; level histogram; survivor ranks ----
    bool m_merged[SLOTS];
    int m_rank[SLOTS];
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        m_merged[s] = false;
        bool surv = false;
        if (j < nt && !(TILE_DBG(A) & 2)) {
            const int64_t r = m_row[s];
            m_merged[s] = (r > 0) && (m_lv[s] < A.top_level) && (r - m_wl[s] >= start_row) && (r + m_wr[s] <= end_row);
            surv = !m_merged[s];
            sflag[j] = m_merged[s] ? 1 : (A.last_stage ? 2 : 0);
            // fused forward: bit 31 of the row's place in Q says "final here" (roots of a truncated tree are
            // quantized by the caller's top stage): the write-back needs ONE LDS word per row
            if (QM && !INV) sdst[j] = m_pos[s] | ((m_merged[s] || (A.last_stage && !A.root_buf)) ? (int32_t)0x80000000 : 0);
            if (m_merged[s]) atomicAdd(&hist[m_ht[s]], 1u);            // rounds are keyed by HEIGHT (raht_common.h, Stage::e_ht)
        }
        if (j < nt && (TILE_DBG(A) & 2)) { sflag[j] = 1; if (QM && !INV) sdst[j] = m_pos[s] | (int32_t)0x80000000; }
        const uint64_t bal = __ballot(surv);
        m_rank[s] = __popcll(bal & lt);
        if (lane == 0 && s * nw + wid < 32) scnt[s * nw + wid] = (uint32_t)__popcll(bal);
    }
    sync_lds();                                                            // sync #2
    PHASE_STAMP(3);

    // ---- P2. level offsets (wave 0); survivor destinations ----
    if (wid == 0) {
        const uint32_t c = hist[lane];
        uint32_t inc = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(inc, d, 64);
            if (lane >= d) inc += t;
        }
        loff[lane] = inc - c;
        cursor[lane] = inc - c;
    }
    {
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int j = tid + s * nthreads;
            if (j < nt && !m_merged[s] && !(TILE_DBG(A) & 2)) {
                uint32_t before = 0;
                for (int q = 0; q < s * nw + wid; ++q) before += scnt[q];
                const uint32_t rk = before + (uint32_t)m_rank[s];            // rank among this tile's survivors
                ssurv[rk] = (uint16_t)j;
            }
        }
    }
    if constexpr (INV) sync_landed(); else sync_lds();                     // sync #3 (inverse: its rows have landed)
    PHASE_STAMP(4);
    if constexpr (INV && QM64) {
        // ---- P2b. the packed integer rows -> float64 coefficients in their rows' slots (encode_3dgs.py:261) ----
        // The integers of row j sit inside the slots of rows j / 2 (they take half the room), so within a round every
        // lane READS its integers before anyone writes (a barrier in between), and rounds walk the rows downwards: writing
        // row r only overwrites the integers of rows 2 r and 2 r + 1, which are behind us. Survivor slots are left alone
        // (P3b fills them). Usually one round: 88 rows x 30 chunks on 512 lanes = 6 tasks per lane.
        if (Dc >= 4) {
            load_steps(lane);
            constexpr int CU = 8;
            const int NCi = (Dc + 3) >> 2;
            const int rows_per_k = nw << lr;
            const int K = (nt + rows_per_k - 1) / rows_per_k;
            const int32_t *stg = (const int32_t *)tile;
            const int e0 = min(coff, Dc - VN);                    // this lane's two channels inside the channel block
            const int last0 = 4 * (NCi - 1), shift = 4 * NCi - Dc;
            const int p0 = e0 >= last0 ? e0 + shift : e0, p1 = e0 + 1 >= last0 ? e0 + 1 + shift : e0 + 1;
            for (int k1 = K; k1 > 0; k1 -= CU) {
                const int k0 = max(k1 - CU, 0);
                int32_t q0[CU], q1[CU];
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    const int k = k0 + u;
                    const int j = min(((wid + k * nw) << lr) + g, nt - 1);
                    q0[u] = 0; q1[u] = 0;
                    if (k < k1) { q0[u] = stg[__mul24(j, NCi * 4) + p0]; q1[u] = stg[__mul24(j, NCi * 4) + p1]; }
                }
                sync_lds();
#pragma unroll
                for (int u = 0; u < CU; ++u) {
                    const int k = k0 + u;
                    const int jj = ((wid + k * nw) << lr) + g;
                    if (k < k1 && active && jj < nt && sflag[jj] != 0) {
                        V16 x;
                        x.v[0] = (T)q0[u] * (T)my_step[0];
                        x.v[1] = (T)q1[u] * (T)my_step[1];
                        *(V16 *)&tile[__mul24(jj, Dp) + coff] = x;
                    }
                }
                if (k0 > 0) sync_lds();
            }
        }
    }
    if constexpr (INV && QM && !QM64) {
        load_steps(lane);
        // roots finalised here come straight from Q as well: dequantize them in place (no butterfly will)
        if (A.last_stage && !A.root_buf && active) for (int it = wid; (it << lr) < nt; it += nw) {
            const int j = (it << lr) + g;
            if (j < nt && sflag[j] == 2) {
                V16 *pr = (V16 *)&tile[__mul24(j, Dp) + coff];
                const RawChunk raw = *(const RawChunk *)pr;
                V16 x;
#pragma unroll
                for (int i = 0; i < VN; ++i) x.v[i] = (T)raw.v[i] * (T)my_step[i];          // encode_3dgs.py:261
                *pr = x;
            }
        }
    }

    // ---- P3a. resolve every butterfly of this tile into a record, bucketed by level ----
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        if (j < nt && m_merged[s]) {
            const int64_t r = m_row[s];
            const int l = m_wl[s];
            int p;
            if (IDENT) {
                p = j - l;
            } else {                                     // partner row r - l is an active row of this tile
                const int32_t want = (int32_t)(r - l);
                int lo = 0, hi = j - 1;
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (srow[mid] < want) lo = mid + 1; else hi = mid;
                }
                p = lo;
            }
            double w0, w1;
            pair_weights(r, l, m_wr[s], A.wsum, w0, w1);
            const double den = w0 + w1;
            MRec<T> rec;
            rec.po = (uint32_t)__mul24((int)p, Dp);
            rec.jo = (uint32_t)__mul24((int)j, Dp);
            rec.a = (T)sqrt(w0 / den);                    // RAHT.py:321-322
            rec.b = (T)sqrt(w1 / den);
            const uint32_t pos = atomicAdd(&cursor[m_ht[s]], 1u);
            mrec[pos] = rec;
        }
    }
    // ---- P3b. inverse: drop the survivors' low-pass rows (from the stage above) into their slots
    if (INV && !A.last_stage && !(TILE_DBG(A) & 2)) {
        // (a) the first TILE_PRE_ROWS survivors were prefetched into spre: LDS -> LDS, no wait on HBM
        //     (kept apart from (b): a value that may come from either source makes hipcc wait for
        //     every outstanding global load, including the next tile's prefetch)
        const uint32_t n_pre = min(surv_cnt, (uint32_t)TILE_PRE_ROWS);
        if (active) for (uint32_t it = wid; (it << lr) < n_pre; it += nw) {
            const uint32_t qc = min((it << lr) + g, n_pre - 1);
            const V16 x = *(const V16 *)&spre[__mul24((int)qc, Dp) + coff];
            *(V16 *)&tile[__mul24((int)ssurv[qc], Dp) + coff] = x;
        }
        // (b) the rest (tiles with many survivors) straight from the workspace
        if (active) for (uint32_t it = wid; TILE_PRE_ROWS + (it << lr) < surv_cnt; it += nw) {
            const uint32_t qc = min(TILE_PRE_ROWS + (it << lr) + g, surv_cnt - 1);
            const V16 x = ld_chunk<T>(row_at((const T *)A.wsn + (int64_t)surv_base * A.ld_ws, qc, (uint32_t)A.ld_ws, (uint32_t)goff));
            *(V16 *)&tile[__mul24((int)ssurv[qc], Dp) + coff] = x;
        }
    }
    // top stage of the inverse: the roots' low-pass values may come from a compact caller buffer
    if (INV && A.last_stage && A.root_buf && !(TILE_DBG(A) & 2)) {
        if (active) for (uint32_t it = wid; (it << lr) < surv_cnt; it += nw) {
            const uint32_t qc = min((it << lr) + g, surv_cnt - 1);
            const V16 x = ld_chunk<T>(A.root_buf + (int64_t)(surv_base + qc) * A.D + goff);
            *(V16 *)&tile[__mul24((int)ssurv[qc], Dp) + coff] = x;
        }
    }
    if constexpr (!INV) sync_landed(); else __syncthreads();               // sync #4 (forward: its rows have landed)
    PHASE_STAMP(5);

    // prefetch the next tile's plan metadata: the loads stay in flight during the butterflies
    if (tile_id + tile_stride < n_tiles) load_tile_meta<T, IDENT, QM, SLOTS>(A, tile_id + tile_stride, tid, nthreads, M);

    // ---- P4. butterflies, one round per level present; a lane group handles one butterfly ----
    {
        const uint32_t stride = (uint32_t)(nw << lr);
        bool chained = false;                 // wave 0 has run levels the other waves have not synchronised with yet
        // lane l keeps level l's offset and count: a v_readlane per level instead of a dependent LDS
        // round trip in front of every round (the rounds are a latency chain; +1.5 % on the step).
        // Fetching the next chained level's record one link ahead as well measured no further gain.
        const int loff_v = (int)loff[lane], hist_v = (int)hist[lane];
        uint64_t mask = __ballot(hist_v > 0);            // levels present in this tile
        if (TILE_DBG(A) & 1) mask = 0;
        MRec<T> pre[TILE_ROUND_U];                       // records of the next wide round's first pass, fetched ahead
        bool have_pre = false;
        while (mask) {
            const int l = INV ? (63 - __clzll((long long)mask)) : (__ffsll((long long)mask) - 1);
            mask &= ~(1ull << l);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane(loff_v, l), cnt = (uint32_t)__builtin_amdgcn_readlane(hist_v, l);
            // branch-free body, U independent LDS chains in flight per lane. Lanes past the level's last
            // butterfly (and the lanes past a row's last chunk) redo the last butterfly (chunk) in lockstep
            // with the lane that owns it: same reads, same writes, one instruction -- no exec-mask
            // juggling in the chain. Only a WHOLE wave instruction past the end must be skipped (a second
            // application by another instruction would not be harmless): a scalar branch. Most levels of a
            // tile hold fewer butterflies than one pass of the workgroup covers: those take U = 1.
            // fetch: the butterfly records of one pass (U per lane group); apply: the butterflies themselves
            auto fetch = [&](auto UC, uint32_t base_, uint32_t cnt_, uint32_t mb, MRec<T> *r) {
                constexpr int U = decltype(UC)::value;
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = mrec[base_ + min(mb + u * stride + g, cnt_ - 1)];
            };
            auto apply = [&](auto UC, uint32_t mb, const MRec<T> *r) {
                constexpr int U = decltype(UC)::value;
                uint32_t ip[U], ij[U];
                V16 x0[U], x1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { ip[u] = r[u].po + coff; ij[u] = r[u].jo + coff; }
#pragma unroll
                for (int u = 0; u < U; ++u) { x0[u] = *(const V16 *)&tile[ip[u]]; x1[u] = *(const V16 *)&tile[ij[u]]; }
                if constexpr (INV && QM && !QM64) {   // the high-pass operand is still the quantized integer (encode_3dgs.py:261)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
#pragma unroll
                        for (int i = 0; i < VN; ++i) x1[u].v[i] = (T)__float_as_int((float)x1[u].v[i]) * (T)my_step[i];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const T ca = r[u].a, cb = r[u].b;
                    V16 lo, hi;
#pragma unroll
                    for (int i = 0; i < VN; ++i) {
                        if (!INV) {                       // RAHT.py:331-332
                            lo.v[i] = ca * x0[u].v[i] + cb * x1[u].v[i];
                            hi.v[i] = ca * x1[u].v[i] - cb * x0[u].v[i];
                        } else {                          // iRAHT.py:108-109
                            lo.v[i] = ca * x0[u].v[i] - cb * x1[u].v[i];
                            hi.v[i] = cb * x0[u].v[i] + ca * x1[u].v[i];
                        }
                    }
                    if (u == 0 || mb + u * stride < cnt) { *(V16 *)&tile[ip[u]] = lo; *(V16 *)&tile[ij[u]] = hi; }
                }
            };
            // pre_loaded: the records of this round's first pass were fetched before the barrier in front of it
            auto pass = [&](auto UC, bool pre_loaded) {
                constexpr int U = decltype(UC)::value;
                uint32_t mb = (uint32_t)(wid << lr);
                MRec<T> r[U];
                if (pre_loaded && mb < cnt) {
#pragma unroll
                    for (int u = 0; u < U; ++u) r[u] = pre[u];
                    apply(UC, mb, r);
                    mb += stride * U;
                }
                for (; mb < cnt; mb += stride * U) { fetch(UC, base, cnt, mb, r); apply(UC, mb, r); }
            };
            if (cnt <= (1u << lr)) {
                // a level that fits ONE wave instruction (most levels of a tile): wave 0 takes it
                // alone. A wave's LDS operations execute in order, so a run of such levels needs no
                // workgroup barrier in between -- the rounds are a latency chain, and a 512-thread
                // barrier per level was most of it.
                if (wid == 0) pass(std::integral_constant<int, 1>(), false);
                chained = true;
                have_pre = false;
            } else {
                if (chained) { __syncthreads(); chained = false; }
                if (cnt <= stride) pass(std::integral_constant<int, 1>(), have_pre);
                else pass(std::integral_constant<int, TILE_ROUND_U>(), have_pre);
                have_pre = false;
#ifdef RAHT_REC_PREFETCH
                // (measured, off: -DRAHT_REC_PREFETCH) the NEXT wide round's first records fetched while the waves gather at the
                // barrier -- they are read-only during P4, and behind the barrier they are one more LDS round trip in every
                // round's chain. 0.594 -> 0.608 ms on the fused step (tools/ab_swap.sh, round 3): eight more live registers and
                // one more LDS access per wave in front of every barrier cost more than the round trip saves
                if (mask) {
                    const int l2 = INV ? (63 - __clzll((long long)mask)) : (__ffsll((long long)mask) - 1);
                    const uint32_t base2 = (uint32_t)__builtin_amdgcn_readlane(loff_v, l2), cnt2 = (uint32_t)__builtin_amdgcn_readlane(hist_v, l2);
                    if (cnt2 > (1u << lr) && (uint32_t)(wid << lr) < cnt2) {
                        if (cnt2 <= stride) fetch(std::integral_constant<int, 1>(), base2, cnt2, (uint32_t)(wid << lr), pre);
                        else fetch(std::integral_constant<int, TILE_ROUND_U>(), base2, cnt2, (uint32_t)(wid << lr), pre);
                        have_pre = true;
                    }
                }
#endif
                __syncthreads();
            }
        }
        if (chained) __syncthreads();
    }

    PHASE_STAMP(6);
    // ---- P5. write back ----
    // (lane geometry re-derived from a fresh opaque copy of the thread id: otherwise the write-back's per-lane addresses
    // are computed long before they are needed and sit in registers across the phases above)
    {
        int tid5 = tid0;
        asm volatile("" : "+v"(tid5));
        lane_geom(tid5, lane, wid, g, coff, goff, active);
    }
    if constexpr (INV && SQ) {
        // raht_dequant_inv_sqdiff: the reconstruction is compared with A.ref on its way out -- per column sum (x - ref)^2, differences
        // in T, squares and sums in float64 (what raht_sqdiff_columns computes from two matrices) -- and written only when the caller
        // wants it (the drivers' PSNR columns, python/encode_3dgs.py:298-310, need the sums, not C_rec). SB row instructions' worth
        // of reference chunks are in flight at a time; per tile: lanes -> the wave's row groups (shuffles) -> the workgroup's
        // waves (LDS, fixed order) -> one float64 per chunk element in sq_part: a deterministic sum.
        constexpr int SB = 4;
        double acc[VN];
#pragma unroll
        for (int i = 0; i < VN; ++i) acc[i] = 0.0;
        const T *rbase = A.ref + e0 * A.ld_ref;
        if (active) for (int it0 = wid; (it0 << lr) < nt; it0 += nw * SB) {
            V16 c[SB];
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int j = min(((it0 + u * nw) << lr) + g, nt - 1);
                c[u] = ld_chunk<T, true>(row_at(rbase, (uint32_t)j, (uint32_t)A.ld_ref, (uint32_t)goff));
            }
#pragma unroll
            for (int u = 0; u < SB; ++u) {
                const int jr = ((it0 + u * nw) << lr) + g;
                if (((it0 + u * nw) << lr) < nt) {                           // (wave-uniform)
                    const int j = min(jr, nt - 1);
                    const V16 x = *(const V16 *)&tile[__mul24(j, Dp) + coff];
                    if (A.out) st_chunk<T, true>(row_at(A.out + e0 * A.ld_out, (uint32_t)j, (uint32_t)A.ld_out, (uint32_t)goff), x);
                    if (jr < nt) {
#pragma unroll
                        for (int i = 0; i < VN; ++i) { const T d = x.v[i] - c[u].v[i]; acc[i] += (double)d * (double)d; }
                    }
                }
            }
        }
        // lanes of one chunk place across the wave's row groups
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            for (int sh = 1 << lg; sh < 64; sh <<= 1) acc[i] += __shfl_xor(acc[i], sh, 64);
        }
        __syncthreads();                                                      // every wave has read its rows: the tile's LDS is free
        double *sacc = (double *)smem;                                        // [nw][NC * VN]
        const int c4s = lane & ((1 << lg) - 1);
        if (g == 0 && c4s < NC) {
#pragma unroll
            for (int i = 0; i < VN; ++i) sacc[(wid * NC + c4s) * VN + i] = acc[i];
        }
        __syncthreads();
        if (tid0 < NC * VN) {
            double t = 0.0;
            for (int w = 0; w < nw; ++w) t += sacc[w * NC * VN + tid0];
            A.sq_part[(int64_t)tile_id * (NC * VN) + tid0] = t;
        }
    } else if (INV) {
        // the whole tile, entry order: stage 0 -> C rows [e0, e0+nt); stage k -> ws_k
        if (active) for (int it = wid; (it << lr) < nt; it += nw) {
            const int j = min((it << lr) + g, nt - 1);
            const V16 x = *(const V16 *)&tile[__mul24(j, Dp) + coff];
            st_chunk<T, IDENT>(row_at(A.out + e0 * A.ld_out, (uint32_t)j, (uint32_t)A.ld_out, (uint32_t)goff), x);   // stage 0: C itself
        }
    } else {
        // survivors, compacted, to the next stage's workspace (top stage: the caller's root buffer)
        if (!(TILE_DBG(A) & 2) && (!A.last_stage || A.root_buf)) {
            T *dstb = A.last_stage ? A.root_buf : A.wsn;
            const int64_t ldb = A.last_stage ? (int64_t)A.D : A.ld_ws;
            if (active) for (uint32_t it = wid; (it << lr) < surv_cnt; it += nw) {
                const uint32_t q = min((it << lr) + g, surv_cnt - 1);
                const V16 x = *(const V16 *)&tile[__mul24((int)ssurv[q], Dp) + coff];
                if constexpr (WT) { if (!A.last_stage) st_chunk_wt<T>(row_at(dstb + (int64_t)surv_base * ldb, q, (uint32_t)ldb, (uint32_t)goff), x);
                                    else st_chunk<T>(row_at(dstb + (int64_t)surv_base * ldb, q, (uint32_t)ldb, (uint32_t)goff), x); }
                else st_chunk<T>(row_at(dstb + (int64_t)surv_base * ldb, q, (uint32_t)ldb, (uint32_t)goff), x);
            }
        }
        // rows finalised here: T[row], or, fused, quantized to Q[inv_order[row]] (encode_3dgs.py:204,210,215)
        auto store_final = [&](auto fast_div) {
            if (active) for (int it = wid; (it << lr) < nt; it += nw) {
                const int jc = min((it << lr) + g, nt - 1);
                V16 x = *(const V16 *)&tile[__mul24(jc, Dp) + coff];
                if constexpr (QM) {
                    const uint32_t dv = (uint32_t)sdst[jc];
                    asm volatile("" : "+v"(x.v[0]));          // keep the row read next to the flag read, not behind its branch
                    if (dv >> 31) {
                        if constexpr (QM64) {
                            int32_t qv[VN];
#pragma unroll
                            for (int i = 0; i < VN; ++i) qv[i] = quantize_one_f64((double)x.v[i], (double)my_step[i]);
                            st_ints<VN>(row_far(A.Q, dv & 0x7fffffffu, (uint32_t)A.ldq, (uint32_t)goff), qv);
                        } else if constexpr (MULTI) {
                            // raht_fwd_quant_multi: the row is quantized once per step table, each into its own matrix
                            for (int kk = 0; kk < MQ->k; ++kk) {
                                const float sp = MQ->step[kk], rc = refined_rcp(sp);
                                RegChunk<int32_t> qv;
#pragma unroll
                                for (int i = 0; i < VN; ++i) qv.v[i] = quantize_one((float)x.v[i], sp, rc, decltype(fast_div)::value);
                                st_chunk<int32_t, true>(row_far(MQ->Q[kk], dv & 0x7fffffffu, (uint32_t)A.ldq, (uint32_t)goff), qv);
                            }
                        } else {
                            RegChunk<int32_t> qv;
#pragma unroll
                            for (int i = 0; i < VN; ++i) qv.v[i] = quantize_one((float)x.v[i], (float)my_step[i], my_rcp[i], decltype(fast_div)::value);
                            st_chunk<int32_t, true>(row_far(A.Q, dv & 0x7fffffffu, (uint32_t)A.ldq, (uint32_t)goff), qv);
                        }
                    }
                } else {
                    if (sflag[jc] != 0) {
                        if constexpr (IDENT) st_chunk<T, true>(row_at(A.fin + e0 * A.ld_fin, (uint32_t)jc, (uint32_t)A.ld_fin, (uint32_t)goff), x);
                        else st_chunk<T, true>(row_far(A.fin, (uint32_t)srow[jc], (uint32_t)A.ld_fin, (uint32_t)goff), x);
                    }
                }
            }
        };
        if constexpr (QM64) {
            load_steps(lane);
            store_final(std::false_type());
        } else if constexpr (MULTI) {
            if (MQ->fast_div) store_final(std::true_type()); else store_final(std::false_type());
        } else if constexpr (QM) {
            load_steps(lane);
            if (ST.fast_div) store_final(std::true_type()); else store_final(std::false_type());
        } else {
            store_final(std::false_type());
        }
    }
    PHASE_STAMP(7);
    __syncthreads();              // LDS is reused by the next tile
    PHASE_STAMP(8);
    }                             // persistent tile loop
}

template <typename T, bool INV, bool IDENT, bool QM, int SLOTS>
__global__ __launch_bounds__(512, (sizeof(T) == 4 ? 6 : 4)) void tile_kernel(const TileArgs<T> A,
                                                   const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type ST)
{
    tile_body<T, INV, IDENT, QM, SLOTS>(A, ST, (int64_t)blockIdx.x, (int64_t)gridDim.x, (int)blockIdx.y);
}

// raht_fwd_quant_multi: the fused forward kernels writing one quantization per step table (tile_body / top_body, MULTI)
template <bool IDENT, int SLOTS>
__global__ __launch_bounds__(512, 6) void tile_kernel_multi(const TileArgs<float> A, const StepTable ST, const MultiQ M)
{
    tile_body<float, false, IDENT, true, SLOTS, false, false, true>(A, ST, (int64_t)blockIdx.x, (int64_t)gridDim.x, (int)blockIdx.y, &M);
}

// stage 0 of raht_dequant_inv_sqdiff (tile_body, SQ): the fused inverse that compares its output with a reference matrix on the way out
template <int SLOTS>
__global__ __launch_bounds__(512, 6) void tile_kernel_sq(const TileArgs<float> A, const StepTable ST)
{
    tile_body<float, true, true, true, SLOTS, false, true>(A, ST, (int64_t)blockIdx.x, (int64_t)gridDim.x, (int)blockIdx.y);
}

// out[ch] = sum over tiles of the partial that holds channel ch (a row's last chunk is the 16 bytes that END the row: when D is not
// a multiple of 4 it repeats channels of its neighbour, which are counted from their own chunk only)
__global__ __launch_bounds__(256) void sq_final_kernel(const double *__restrict__ part, int64_t n_tiles, int D, int ncv, double *__restrict__ out)
{
    __shared__ double red[256];
    const int ch = blockIdx.x;
    const int full = (D / 4) * 4;                          // channels [0, full): chunk ch / 4, element ch % 4
    const int e = ch < full ? ch : (ncv - 4) + (ch - (D - 4));
    double t = 0.0;
    for (int64_t k = threadIdx.x; k < n_tiles; k += 256) t += part[k * ncv + e];
    red[threadIdx.x] = t;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[ch] = red[0];
}

// FORWARD CHAINING of the later tile stages (round 3). Stage k + 1's tile P can run as soon as the tiles of stage k that
// feed it are done, and those are a handful of CONSECUTIVE tiles. So stages 1 .. last tile stage go out as ONE launch with one
// workgroup per stage-1 tile: a workgroup that finishes a tile adds the number of survivors it delivered to the arrival
// counter of each parent tile it fed (at most two: the next stage's tiles are at least as long as this one's); whoever
// completes a parent's count runs that parent next, in the same workgroup, and so on upwards. Nobody ever waits, so nothing
// can deadlock; every tile of every chained stage is run exactly once (by the last of its children to arrive). Ordering
// (the parent may run on another XCD, whose L2 is a different one): survivor rows are written THROUGH the L2 (st_chunk_wt,
// sc1), every thread waits for its stores' acknowledgement, barrier, one relaxed agent-scope atomic; the taker loads the rows
// past its own L2 (glds16<2>). Counters return to zero (the taker resets them). MEASURED, OFF by default (RAHT_CHAIN=1):
// fused cfg3 forward 0.3038 -> 0.3112 ms -- the acknowledgement wait, the atomic and the parent's coherent loads are three
// ~2 us round trips on every tile's way up, more than the 4.6 us launch gap and the ~6 us of stage-2 work the chaining hides
// (the first version, with agent-scope fences = L2 write-backs instead of write-through stores: 0.339 ms).
// Only with all D channels in one chunk (D <= 64). The pending parents (depth-first, at most one per stage above) live at
// the end of the dynamic LDS block, behind what tile_body uses.
constexpr int CHAIN_MAX = 6;
template <typename T>
struct TileChain {
    TileArgs<T> a[CHAIN_MAX];
    uint32_t *arrive[CHAIN_MAX];       // arrive[i]: counters of a[i]'s tiles (i >= 1)
    int n;
    uint32_t stack_off;                // byte offset of the pending list in the dynamic LDS block
};

template <typename T, bool QM, int SLOTS>
__global__ __launch_bounds__(512, (sizeof(T) == 4 ? 6 : 4)) void tile_kernel_chain(const TileChain<T> C,
                                                   const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type ST)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t *pend = (uint32_t *)(smem + C.stack_off);      // [0] = count, [1 + i] = stage << 24 | tile
    int stage = 0;
    int64_t tile = blockIdx.x;
    if (threadIdx.x == 0) pend[0] = 0;
    for (;;) {
        tile_body<T, false, false, QM, SLOTS, true>(C.a[stage], ST, tile, (int64_t)1 << 40, 0);      // (ends with a barrier)
        if (stage + 1 < C.n) {
            // this tile's survivor rows were written THROUGH the L2 (st_chunk_wt): once every thread's stores are acknowledged
            // they are visible device-wide, and the parent tile loads them past its own XCD's L2 (glds16<2>). No fence: an
            // agent-scope release is a write-back of the whole L2 (round 3, first version: the chained launch 35 us SLOWER than
            // the separate launches; with a fence in every thread 220 us instead of 31).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (threadIdx.x == 0) {
                const TileArgs<T> &A = C.a[stage];
                const uint32_t sb = A.surv_off[tile], se = A.surv_off[tile + 1];
                const uint32_t Rn = (uint32_t)C.a[stage + 1].R, n_next = (uint32_t)C.a[stage + 1].n_entries;
                uint32_t *arr = C.arrive[stage + 1];
                for (uint32_t lo = sb; lo < se;) {
                    const uint32_t pt = lo / Rn, hi = min(se, (pt + 1) * Rn), add = hi - lo;
                    const uint32_t target = min(Rn, n_next - pt * Rn);
                    const uint32_t old = atomicAdd(&arr[pt], add);
                    if (old + add == target) {
                        arr[pt] = 0;                           // ours now; ready for the next launch
                        const uint32_t c = pend[0];
                        pend[1 + c] = ((uint32_t)(stage + 1) << 24) | pt;
                        pend[0] = c + 1;
                    }
                    lo = hi;
                }
            }
            __syncthreads();
        }
        const uint32_t c = pend[0];
        if (c == 0) break;
        const uint32_t top = pend[c];
        __syncthreads();                                      // everybody has read the entry before it is popped / overwritten
        if (threadIdx.x == 0) pend[0] = c - 1;
        stage = (int)(top >> 24);
        tile = (int64_t)(top & 0xffffffu);
        __syncthreads();
    }
}

// Several scenes, one launch (raht_*_batch): the same stage of up to TILE_BATCH_MAX scenes. first_tile[s] = number of tiles of the
// scenes before s; a workgroup finds its scene with a handful of scalar compares and runs ONE tile of it. A frame of ~1 M
// Gaussians fills the chip's 768 workgroup slots two and a half times and then waits ~20 us for its tail stages (a third of
// its step); batched, the partial rounds of different scenes fill each other and ALL tails are one launch per stage.
constexpr int TILE_BATCH_MAX = 8;
template <typename T>
struct TileBatch {
    TileArgs<T> a[TILE_BATCH_MAX];
    uint32_t first_tile[TILE_BATCH_MAX + 1];
    int n;
};

template <typename T, bool INV, bool IDENT, bool QM, int SLOTS>
__global__ __launch_bounds__(512, (sizeof(T) == 4 ? 6 : 4)) void tile_kernel_batch(const TileBatch<T> B,
                                                   const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type ST)
{
    int s = 0;
#pragma unroll
    for (int q = 1; q < TILE_BATCH_MAX; ++q) s += (q < B.n && blockIdx.x >= B.first_tile[q]) ? 1 : 0;
    const int64_t t = (int64_t)(blockIdx.x - B.first_tile[s]);
    tile_body<T, INV, IDENT, QM, SLOTS>(B.a[s], ST, t, (int64_t)1 << 40, (int)blockIdx.y);
}

// ------------------------------------------------------------------------------------------------
// TOP stage: the last <= RAHT_TOP_MAX_ROWS entries of the tree in ONE launch.
//
// The top of the tree is a handful of butterflies per level over a few thousand rows: as tile
// stages it was two or three launches of ~15-20 us each, all of it latency (launch, metadata round
// trip, merge resolution, ~20 barrier-separated rounds per tile). Here the butterflies are resolved
// once per schedule (plan.hip: build_top_stage; partner entry, a, b, sorted by level), a workgroup
// owns ONE 16-byte channel chunk of ALL entries (LDS: 16 bytes per entry), every thread keeps its
// <= 8 butterfly records in registers, and the levels are walked with one barrier each.
// ------------------------------------------------------------------------------------------------
constexpr int TOP_THREADS = 1024;
constexpr int TOP_SLOTS = RAHT_TOP_MAX_ROWS / TOP_THREADS;       // entries / butterflies per thread

template <typename T>
struct TopArgs {
    const T *in;  int64_t ld_in;       // fwd: the stage's entries, entry order (C when it is the only stage, else ws)
    T *fin;       int64_t ld_fin;      // T rows (fwd out / inv in)
    T *out;       int64_t ld_out;      // inv: the stage's entries, entry order (C or ws)
    int32_t *Q;   int64_t ldq;         // fused quantization
    const uint32_t *rows;              // entry -> row (nullptr: identity)
    int io_mapped;                     // the stage's entry-ordered input / output is addressed through rows[] as well
                                       // (a plan with a row map, raht_plan_set_row_map: entry e lives in matrix row rows[e])
    const uint32_t *e_pos;             // entry -> position in Q
    const uint32_t *pj;                // butterflies sorted by level: partner entry | own entry << 16
    const T *ab;                       // a, b per butterfly
    const uint32_t *root_rank;         // entry -> row of the caller's root buffer (~0u: not a root)
    T *root_buf;
    int n, n_merges, D;
    const uint32_t *lev;               // device [2 * nlev]: (first, end) butterfly of every non-empty level, ascending
    int nlev, nbig;                    // levels [0, nbig): whole workgroup, a barrier each; [nbig, nlev): chained by wave 0
    uint32_t small_start;              // first butterfly of the chained part
};

template <typename T, bool INV, bool QM, bool MULTI = false>
__device__ __forceinline__ void top_body(const TopArgs<T> &A, const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type &ST,
                                         const int chunk, const MultiQ *MQ = nullptr)
{
    constexpr bool QM64 = QM && sizeof(T) == 8;              // float64 rows, 2 quantized integers per 16-byte chunk
    extern __shared__ __align__(16) unsigned char smem[];
    typedef RegChunk<T> V16;
    constexpr int VN = 16 / sizeof(T);
    V16 *tile = (V16 *)smem;
    __shared__ uint32_t s_lev[2 * 64];
    const int tid = threadIdx.x;
    const int goff = min(chunk * VN, A.D - VN);                    // last chunk: the 16 bytes that end the row
    const int n = A.n, nm = A.n_merges;
    if (tid < 2 * A.nlev) s_lev[tid] = A.lev[tid];
    // records of the chained levels, staged in LDS behind the entries
    const int n_small = nm - (int)A.small_start;
    uint32_t *s_pj = (uint32_t *)(smem + (size_t)n * 16);
    T *s_ab = (T *)(s_pj + ((n_small + 3) & ~3));
    for (int i = tid; i < n_small; i += TOP_THREADS) {
        s_pj[i] = A.pj[A.small_start + i];
        s_ab[2 * i] = A.ab[2 * (A.small_start + i)];
        s_ab[2 * i + 1] = A.ab[2 * (A.small_start + i) + 1];
    }

    typename StepsFor<T>::elem my_step[VN];
    float my_rcp[VN];
#pragma unroll
    for (int i = 0; i < VN; ++i) { my_step[i] = 1; my_rcp[i] = 1.0f; }
    if constexpr (QM) {
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            my_step[i] = ST.v[ST.n == 1 ? 0 : goff + i];
            if constexpr (!QM64) my_rcp[i] = refined_rcp(my_step[i]);
        }
    }

    // this thread's butterflies (clamped, unconditional loads) ...
    uint32_t pj[TOP_SLOTS];
    T ra[TOP_SLOTS], rb[TOP_SLOTS];
#pragma unroll
    for (int k = 0; k < TOP_SLOTS; ++k) {
        const int idx = min(k * TOP_THREADS + tid, max(nm - 1, 0));
        pj[k] = A.pj[idx]; ra[k] = A.ab[2 * idx]; rb[k] = A.ab[2 * idx + 1];
    }
    // ... where its entries' coefficients live (T row or Q position; root buffer row), fetched up front
    // so that neither the loads nor the stores below wait on a dependent metadata load per entry ...
    uint32_t m_dst[TOP_SLOTS], m_rr[TOP_SLOTS];
#pragma unroll
    for (int k = 0; k < TOP_SLOTS; ++k) {
        const int e = min(k * TOP_THREADS + tid, n - 1);
        m_rr[k] = A.root_rank[e];
        m_dst[k] = QM ? A.e_pos[e] : (A.rows ? A.rows[e] : (uint32_t)e);
    }
    // ... and the entries themselves
    {
        typedef typename std::conditional<QM && INV && !QM64, int32_t, T>::type RawT;
        RegChunk<RawT> x[TOP_SLOTS];
#pragma unroll
        for (int k = 0; k < TOP_SLOTS; ++k) {
            const int e = min(k * TOP_THREADS + tid, n - 1);
            if constexpr (QM64 && INV) {
                int32_t q[VN];
                ld_ints<VN>(A.Q + (int64_t)m_dst[k] * A.ldq + goff, q);
#pragma unroll
                for (int i = 0; i < VN; ++i) x[k].v[i] = (T)q[i];
            } else if constexpr (!INV) {
                x[k] = ld_chunk<RawT>((const RawT *)A.in + (int64_t)(A.io_mapped ? m_dst[k] : (uint32_t)e) * A.ld_in + goff);
            } else if constexpr (QM) {
                x[k] = ld_chunk<RawT>((const RawT *)A.Q + (int64_t)m_dst[k] * A.ldq + goff);
            } else {
                const T *src = (A.root_buf && m_rr[k] != 0xffffffffu) ? A.root_buf + (int64_t)m_rr[k] * A.D
                                                                      : A.fin + (int64_t)m_dst[k] * A.ld_fin;
                x[k] = ld_chunk<RawT>((const RawT *)src + goff);
            }
        }
#pragma unroll
        for (int k = 0; k < TOP_SLOTS; ++k) {
            const int e = k * TOP_THREADS + tid;
            if (e < n) {
                V16 v;
#pragma unroll
                for (int i = 0; i < VN; ++i) {
                    v.v[i] = (T)x[k].v[i];
                    if constexpr (QM && INV) v.v[i] = v.v[i] * (T)my_step[i];            // encode_3dgs.py:261
                }
                tile[e] = v;
            }
        }
        if constexpr (QM && INV) {
            // the roots' low-pass values come from the caller's compact buffer (already dequantized)
            if (A.root_buf) {
#pragma unroll
                for (int k = 0; k < TOP_SLOTS; ++k) {
                    const int e = k * TOP_THREADS + tid;
                    if (e < n && m_rr[k] != 0xffffffffu) tile[e] = ld_chunk<T>(A.root_buf + (int64_t)m_rr[k] * A.D + goff);
                }
            }
        }
    }
    __syncthreads();

    auto butterfly = [&](uint32_t rec, T a, T b) {
        const uint32_t ip = rec & 0xffffu, ij = rec >> 16;
        const V16 x0 = tile[ip], x1 = tile[ij];
        V16 vlo, vhi;
#pragma unroll
        for (int i = 0; i < VN; ++i) {
            if (!INV) {                                   // RAHT.py:331-332
                vlo.v[i] = a * x0.v[i] + b * x1.v[i];
                vhi.v[i] = a * x1.v[i] - b * x0.v[i];
            } else {                                      // iRAHT.py:108-109
                vlo.v[i] = a * x0.v[i] - b * x1.v[i];
                vhi.v[i] = b * x0.v[i] + a * x1.v[i];
            }
        }
        tile[ip] = vlo; tile[ij] = vhi;
    };
    // levels with more than 64 butterflies: the whole workgroup, one barrier per level (loops not
    // unrolled: 63 copies of the body thrash the instruction cache)
    auto big_levels = [&]() {
#pragma unroll 1
        for (int q = 0; q < A.nbig; ++q) {
            const int li = INV ? A.nbig - 1 - q : q;
            const uint32_t lo = s_lev[2 * li], hi = s_lev[2 * li + 1];
#pragma unroll
            for (int k = 0; k < TOP_SLOTS; ++k) {
                const uint32_t idx = (uint32_t)(k * TOP_THREADS + tid);
                if ((uint32_t)(k * TOP_THREADS) < hi && (uint32_t)((k + 1) * TOP_THREADS) > lo && idx >= lo && idx < hi)
                    butterfly(pj[k], ra[k], rb[k]);
            }
            __syncthreads();
        }
    };
    // the top of the tree, <= 64 butterflies per level: ONE wave walks it without barriers (a wave's
    // LDS operations execute in order, so a level sees the previous level's stores)
    auto small_levels = [&]() {
        if (tid < 64) {
#pragma unroll 1
            for (int q = A.nbig; q < A.nlev; ++q) {
                const int li = INV ? A.nlev - 1 - (q - A.nbig) : q;
                const uint32_t lo = s_lev[2 * li], hi = s_lev[2 * li + 1];
                const uint32_t i = lo - A.small_start + (uint32_t)tid;
                if (lo + (uint32_t)tid < hi) butterfly(s_pj[i], s_ab[2 * i], s_ab[2 * i + 1]);
            }
        }
        __syncthreads();
    };
    if (!INV) { big_levels(); small_levels(); }
    else { small_levels(); big_levels(); }

    // write back
#pragma unroll
    for (int k = 0; k < TOP_SLOTS; ++k) {
        const int e = k * TOP_THREADS + tid;
        if (e >= n) continue;
        const V16 v = tile[e];
        if constexpr (INV) {
            st_chunk<T>(A.out + (int64_t)(A.io_mapped ? m_dst[k] : (uint32_t)e) * A.ld_out + goff, v);
        } else {
            const uint32_t rr = m_rr[k];
            const bool to_buf = A.root_buf && rr != 0xffffffffu;   // still a low-pass value: the caller's top stage takes it
            if (to_buf) st_chunk<T>(A.root_buf + (int64_t)rr * A.D + goff, v);
            if (QM && to_buf) {
                // roots are quantized by the caller's top stage
            } else if constexpr (QM) {
                if constexpr (QM64) {
                    int32_t qv[VN];
#pragma unroll
                    for (int i = 0; i < VN; ++i) qv[i] = quantize_one_f64((double)v.v[i], (double)my_step[i]);
                    st_ints<VN>(A.Q + (int64_t)m_dst[k] * A.ldq + goff, qv);
                } else if constexpr (MULTI) {
                    for (int kk = 0; kk < MQ->k; ++kk) {
                        const float sp = MQ->step[kk], rc = refined_rcp(sp);
                        RegChunk<int32_t> qv;
#pragma unroll
                        for (int i = 0; i < VN; ++i) qv.v[i] = quantize_one((float)v.v[i], sp, rc, MQ->fast_div);
                        st_chunk<int32_t>(MQ->Q[kk] + (int64_t)m_dst[k] * A.ldq + goff, qv);
                    }
                } else {
                    RegChunk<int32_t> qv;
#pragma unroll
                    for (int i = 0; i < VN; ++i) qv.v[i] = quantize_one((float)v.v[i], (float)my_step[i], my_rcp[i], ST.fast_div);
                    st_chunk<int32_t>(A.Q + (int64_t)m_dst[k] * A.ldq + goff, qv);
                }
            } else {
                st_chunk<T>(A.fin + (int64_t)m_dst[k] * A.ld_fin + goff, v);
            }
        }
    }
}

template <typename T, bool INV, bool QM>
__global__ __launch_bounds__(TOP_THREADS) void top_kernel(const TopArgs<T> A,
                                                          const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type ST)
{
    top_body<T, INV, QM>(A, ST, (int)blockIdx.x);
}

__global__ __launch_bounds__(TOP_THREADS) void top_kernel_multi(const TopArgs<float> A, const StepTable ST, const MultiQ M)
{
    top_body<float, false, true, true>(A, ST, (int)blockIdx.x, &M);
}

// the top stages of several scenes in one launch (raht_*_batch): blockIdx.y = scene
template <typename T>
struct TopBatch { TopArgs<T> a[TILE_BATCH_MAX]; };

template <typename T, bool INV, bool QM>
__global__ __launch_bounds__(TOP_THREADS) void top_kernel_batch(const TopBatch<T> B,
                                                                const typename std::conditional<QM, typename StepsFor<T>::type, NoSteps>::type ST)
{
    top_body<T, INV, QM>(B.a[blockIdx.y], ST, (int)blockIdx.x);
}

// node weights of RAHT.py:325-328: after its own butterfly a right sibling carries w0 + w1 and is
// never touched again; row 0 ends with the total weight.
template <typename T>
__global__ void node_weight_kernel(const int32_t *__restrict__ wl, const int32_t *__restrict__ wr,
                                   const int64_t *__restrict__ wsum, int64_t N, T *__restrict__ w)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    if (i == 0) { w[0] = (T)(wsum ? wsum[N] : N); return; }
    double w0, w1;
    pair_weights(i, wl[i], wr[i], wsum, w0, w1);
    w[i] = (T)(w0 + w1);
}

static int tile_threads()
{
    static int t = 0;
    if (t == 0) {
        const char *e = getenv("RAHT_TILE_THREADS");      // tuning knob: 512 (default) or 256
        const int v = e ? atoi(e) : 512;
        t = (v == 256 || v == 512) ? v : 512;
    }
    return t;
}

// threads per workgroup of the stages >= 1 (tuning knob RAHT_TAIL_THREADS: 256 or 512; default = tile_threads())
static int tail_threads()
{
    static int t = 0;
    if (t == 0) {
        const char *e = getenv("RAHT_TAIL_THREADS");
        const int v = e ? atoi(e) : tile_threads();
        t = (v == 256 || v == 512) ? v : tile_threads();
    }
    return t;
}

static int device_cus()
{
    static int n[RAHT_MAX_DEVICES] = {};
    const int dev = current_device();
    if (n[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n[dev] = v;
        else n[dev] = 256;                                  // MI355X
    }
    return n[dev];
}

// RAHT_TILE_PERSIST (tuning knob): 0 / unset = one tile per workgroup (default, measured fastest);
// 1 = as many workgroups as the chip keeps resident, each walking tiles b, b + grid, ...;
// k >= 2 = k tiles per workgroup.
static int persist_mode()
{
    static int v = -1;
    if (v < 0) { const char *e = getenv("RAHT_TILE_PERSIST"); v = e ? std::max(0, atoi(e)) : 0; }
    return v;
}

static int lp_shift_for(int Dc)
{
    int s = 0;
    while ((1 << s) < Dc && s < 6) ++s;
    return s;
}

// What one direction of the transform reads / writes.
template <typename T>
struct XformIO {
    const T *src = nullptr; int64_t ld_src = 0;     // fwd: C             inv: T (unless quantized)
    T *dst = nullptr; int64_t ld_dst = 0;           // fwd: T (unless q)  inv: C
    int32_t *Q = nullptr; int64_t ldq = 0;          // fused quantization (fwd out / inv in)
    const typename StepsFor<T>::elem *steps = nullptr; int n_steps = 0;
};

// ---- host side of the tile / top launches: "prepare" fills and validates the kernel arguments of one (scene, stage),
// "launch" enqueues one scene's stage, "launch_*_batch" the same stage of several scenes in one launch ----
struct TileGeom {
    int64_t n_tiles = 0;
    unsigned grid_x = 0, nchunks = 0;
    int threads = 0;
    size_t lds = 0;
    bool one = true, ident = true;
    bool same_shape(const TileGeom &o) const { return nchunks == o.nchunks && threads == o.threads && lds == o.lds && one == o.one && ident == o.ident; }
};

template <typename T, bool INV, bool IDENT, bool QM, int SLOTS>
static int tile_kernel_attr()
{
    // > 64 KiB of dynamic LDS must be allowed per function AND per device
    static PerDeviceOnce attr, attr_b;
    if (attr.first(current_device()))
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel<T, INV, IDENT, QM, SLOTS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    if (attr_b.first(current_device()))
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_batch<T, INV, IDENT, QM, SLOTS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    return RAHT_OK;
}

template <typename T, bool INV, bool IDENT, bool QM, int SLOTS>
static int launch_tile_one(const TileArgs<T> &A, const XformIO<T> &io, dim3 grid, int threads, size_t lds, hipStream_t s)
{
    RAHT_RET((tile_kernel_attr<T, INV, IDENT, QM, SLOTS>()));
    if constexpr (QM) {
        typename StepsFor<T>::type st;
        fill_step_table(st, io.steps, io.n_steps);
        hipLaunchKernelGGL((tile_kernel<T, INV, IDENT, true, SLOTS>), grid, dim3(threads), lds, s, A, st);
    } else {
        NoSteps ns{0, 0};
        hipLaunchKernelGGL((tile_kernel<T, INV, IDENT, false, SLOTS>), grid, dim3(threads), lds, s, A, ns);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

template <typename T, bool INV, bool IDENT, bool QM, int SLOTS>
static int launch_tile_batch_one(const TileBatch<T> &B, const XformIO<T> &io, dim3 grid, int threads, size_t lds, hipStream_t s)
{
    RAHT_RET((tile_kernel_attr<T, INV, IDENT, QM, SLOTS>()));
    if constexpr (QM) {
        typename StepsFor<T>::type st;
        fill_step_table(st, io.steps, io.n_steps);
        hipLaunchKernelGGL((tile_kernel_batch<T, INV, IDENT, true, SLOTS>), grid, dim3(threads), lds, s, B, st);
    } else {
        NoSteps ns{0, 0};
        hipLaunchKernelGGL((tile_kernel_batch<T, INV, IDENT, false, SLOTS>), grid, dim3(threads), lds, s, B, ns);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

template <typename T, bool INV, bool QM>
static int prepare_top_stage(const raht_plan *p, const Schedule &sc, int k, const XformIO<T> &io, int D, TopArgs<T> &A, size_t &lds)
{
    const Stage &st = sc.stages[(size_t)k];
    T *ws_k = (k >= 1) ? (T *)stage_ws(st, INV) : nullptr;
    A.in = nullptr; A.ld_in = 0; A.out = nullptr; A.ld_out = 0;
    if (!INV) { A.in = (k == 0) ? io.src : ws_k; A.ld_in = (k == 0) ? io.ld_src : D; A.fin = io.dst; A.ld_fin = io.ld_dst; }
    else { A.fin = const_cast<T *>(io.src); A.ld_fin = io.ld_src; A.out = (k == 0) ? io.dst : ws_k; A.ld_out = (k == 0) ? io.ld_dst : D; }
    A.Q = io.Q; A.ldq = io.ldq;
    A.rows = st.rows;
    A.io_mapped = 0;
    if (p->row_map) {                                      // only single-stage plans carry a row map (run_transform checks)
        if (k != 0 || st.rows || QM) { set_error("row-mapped plans run as ONE top stage without fused quantization"); return RAHT_ERR_UNSUPPORTED; }
        A.rows = p->row_map;
        A.io_mapped = 1;
    }
    A.e_pos = st.rows ? st.e_pos : p->inv_order;
    A.pj = st.t_pj;
    if constexpr (sizeof(T) == 4) A.ab = (const T *)st.t_ab32; else A.ab = (const T *)st.t_ab64;
    A.root_rank = st.t_root;
    A.root_buf = (T *)p->root_buf;
    A.n = (int)st.n_entries; A.n_merges = (int)st.n_merges; A.D = D;
    A.lev = st.t_lev; A.nlev = st.t_nlev; A.nbig = st.t_nbig; A.small_start = st.t_small_start;
    const size_t n_small = st.n_merges - st.t_small_start;
    lds = (size_t)st.n_entries * 16 + ((n_small + 3) & ~(size_t)3) * 4 + n_small * 2 * sizeof(T);   // + 512 B static
    static PerDeviceOnce attr;
    if (attr.first(current_device())) {
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)top_kernel<T, INV, QM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)top_kernel_batch<T, INV, QM>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
    }
    return RAHT_OK;
}

template <typename T, bool INV, bool QM>
static int launch_top_stage(const raht_plan *p, const Schedule &sc, int k, const XformIO<T> &io, int D, hipStream_t s)
{
    constexpr int VN = 16 / (int)sizeof(T);
    TopArgs<T> A;
    size_t lds = 0;
    RAHT_RET((prepare_top_stage<T, INV, QM>(p, sc, k, io, D, A, lds)));
    const dim3 grid((unsigned)((D + VN - 1) / VN));
    if constexpr (QM) {
        typename StepsFor<T>::type stp;
        fill_step_table(stp, io.steps, io.n_steps);
        hipLaunchKernelGGL((top_kernel<T, INV, true>), grid, dim3(TOP_THREADS), lds, s, A, stp);
    } else {
        NoSteps ns{0, 0};
        hipLaunchKernelGGL((top_kernel<T, INV, false>), grid, dim3(TOP_THREADS), lds, s, A, ns);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// the top stages of scenes idx[0..m): one launch, blockIdx.y = scene (io.steps: the batch shares one step table)
template <typename T, bool INV, bool QM>
static int launch_top_batch(int m, const TopArgs<T> *As, const size_t *ldss, const XformIO<T> &io, int D, hipStream_t s)
{
    constexpr int VN = 16 / (int)sizeof(T);
    TopBatch<T> B;
    size_t lds = 0;
    for (int i = 0; i < m; ++i) { B.a[i] = As[i]; lds = std::max(lds, ldss[i]); }
    for (int i = m; i < TILE_BATCH_MAX; ++i) B.a[i] = As[0];
    const dim3 grid((unsigned)((D + VN - 1) / VN), (unsigned)m);
    if constexpr (QM) {
        typename StepsFor<T>::type stp;
        fill_step_table(stp, io.steps, io.n_steps);
        hipLaunchKernelGGL((top_kernel_batch<T, INV, true>), grid, dim3(TOP_THREADS), lds, s, B, stp);
    } else {
        NoSteps ns{0, 0};
        hipLaunchKernelGGL((top_kernel_batch<T, INV, false>), grid, dim3(TOP_THREADS), lds, s, B, ns);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

template <typename T, bool INV, bool QM>
static int prepare_tile_stage(const raht_plan *p, const Schedule &sc, int k, const XformIO<T> &io, int D, int Dc0, int dbg,
                              TileArgs<T> &A, TileGeom &G)
{
    const Stage &st = sc.stages[(size_t)k];
    int Dc = Dc0;
    if (k >= 1) {                                    // later stages: large tiles, channel chunks
        int r1 = 0, rf = 0;
        pick_tail_geometry(p, (int)sizeof(T), D, sc.tile_rows, &r1, &Dc, &rf);
    }
    const int K = (int)sc.stages.size();
    A.rows = st.rows; A.surv_off = st.surv_off; A.n_entries = st.n_entries; A.N = p->N; A.R = st.tile_rows;
    A.D = D; A.Dc = Dc;
    constexpr int VN = 16 / (int)sizeof(T);
    A.Dp = (Dc + VN - 1) / VN * VN;
    A.lg = 0;
    while ((1 << A.lg) < A.Dp / VN) ++A.lg;               // lanes per row: power of two >= chunks per row (Dc <= 64)
    A.last_stage = (k == K - 1) ? 1 : 0;
    A.wsum = p->wsum;
    if (st.rows) { A.lvl = st.e_lvl; A.wl = st.e_wl; A.wr = st.e_wr; A.inv_order = st.e_pos; }
    else { A.lvl = p->lvl; A.wl = p->wl; A.wr = p->wr; A.inv_order = p->inv_order; }
    static const bool rounds_by_level = getenv("RAHT_ROUNDS_BY_LEVEL") != nullptr;    // A/B knob: one round per binary level present (rounds 1-2)
    A.ht = rounds_by_level ? A.lvl : st.e_ht;
    A.Q = io.Q; A.ldq = io.ldq;
    A.top_level = p->top_level; A.root_buf = (T *)p->root_buf;
    A.dbg = dbg; A.nwide = 0; A.ref = nullptr; A.ld_ref = 0; A.sq_part = nullptr;
    A.ld_ws = D;
    A.wsn = (k + 1 < K) ? (T *)stage_ws(sc.stages[(size_t)k + 1], INV) : nullptr;
    T *ws_k = (k >= 1) ? (T *)stage_ws(st, INV) : nullptr;
    if (!INV) {
        A.in = (k == 0) ? io.src : ws_k; A.ld_in = (k == 0) ? io.ld_src : D;
        A.fin = io.dst; A.ld_fin = io.ld_dst;
        A.out = nullptr; A.ld_out = 0;
    } else {
        A.in = nullptr; A.ld_in = 0;
        A.fin = const_cast<T *>(io.src); A.ld_fin = io.ld_src;
        A.out = (k == 0) ? io.dst : ws_k; A.ld_out = (k == 0) ? io.ld_dst : D;
    }
    // Every pointer the kernel will dereference for THIS (direction, stage) must be there before it is
    // launched: a tile kernel handed a null workspace reads address 0 and the process dies in ROCr's
    // fault handler at the next synchronisation (the round-1 abort). In particular the LAST
    // tile stage has no stage above it (wsn == nullptr): its survivors are the roots.
    {
        const char *bad = nullptr;
        if (!A.lvl || !A.wl || !A.wr || !A.ht) bad = "plan arrays";
        else if (!A.last_stage && (!A.wsn || !A.surv_off)) bad = "survivor workspace of a non-final stage";
        else if (k >= 1 && !ws_k) bad = "stage workspace";
        else if (QM && (!A.Q || !A.inv_order)) bad = "Q / inv_order";
        else if (!INV && (!A.in || (!QM && !A.fin))) bad = "forward input / output";
        else if (INV && (!A.out || (!QM && !A.fin))) bad = "inverse input / output";
        else if (st.tile_rows < 1 || (int64_t)st.tile_rows * std::max<int64_t>(std::max(A.ld_in, A.ld_out), A.ld_fin) * (int64_t)sizeof(T) >= ((int64_t)1 << 32)) bad = "tile geometry (32-bit row offsets)";
        if (bad) { set_error("tile stage %d (%s): missing %s", k, INV ? "inverse" : "forward", bad); return RAHT_ERR_INVALID; }
    }
    G.nchunks = (unsigned)((D + Dc - 1) / Dc);
    G.lds = tile_lds_bytes(st.tile_rows, (int)sizeof(T), Dc, st.rows == nullptr, QM);
    G.threads = (k == 0) ? tile_threads() : tail_threads();
    if (st.tile_rows > TILE_MAX_SLOTS * G.threads) {
        set_error("tile_rows %d too large for %d threads", st.tile_rows, G.threads);
        return RAHT_ERR_UNSUPPORTED;
    }
    // persistent workgroups: as many as the chip keeps resident (LDS granules of 1280 B, 32 waves
    // per CU), each walking tiles blockIdx.x, blockIdx.x + gridDim.x, ...
    const int per_cu = std::max(1, std::min((int)(128 / ((G.lds + 1279) / 1280)), 32 / (G.threads / 64)));
    const int64_t resident = (int64_t)per_cu * device_cus();
    const int pm = persist_mode();
    const int64_t gx = pm == 1 ? std::min<int64_t>(st.n_tiles, std::max<int64_t>(1, resident / G.nchunks))
                     : pm >= 2 ? ceil_div(st.n_tiles, (int64_t)pm) : st.n_tiles;
    G.n_tiles = st.n_tiles;
    G.grid_x = (unsigned)gx;
    G.one = st.tile_rows <= G.threads;
    G.ident = st.rows == nullptr;
    return RAHT_OK;
}

template <typename T, bool INV, bool QM>
static int launch_stage_impl(const raht_plan *p, const Schedule &sc, int k, const XformIO<T> &io, int D, int Dc0,
                             hipStream_t s, int dbg);

template <typename T, bool INV, bool QM>
static int launch_tile_stage(const raht_plan *p, const Schedule &sc, int k, const XformIO<T> &io, int D, int Dc0,
                             hipStream_t s, int dbg = 0)
{
    if (k == 0 && p->ev_before && dbg == 0) {          // profiling: bracket the stage-0 launch of a real transform
        RAHT_HIP_CHECK(hipEventRecord(p->ev_before, s));
        const int rc = launch_stage_impl<T, INV, QM>(p, sc, k, io, D, Dc0, s, dbg);
        RAHT_HIP_CHECK(hipEventRecord(p->ev_after, s));
        return rc;
    }
    return launch_stage_impl<T, INV, QM>(p, sc, k, io, D, Dc0, s, dbg);
}

template <typename T, bool INV, bool QM>
static int launch_stage_impl(const raht_plan *p, const Schedule &sc, int k, const XformIO<T> &io, int D, int Dc0,
                             hipStream_t s, int dbg)
{
    const Stage &st = sc.stages[(size_t)k];
    if (st.is_top) return launch_top_stage<T, INV, QM>(p, sc, k, io, D, s);
    TileArgs<T> A;
    TileGeom G;
    RAHT_RET((prepare_tile_stage<T, INV, QM>(p, sc, k, io, D, Dc0, dbg, A, G)));
    const dim3 grid(G.grid_x, G.nchunks);
    if (G.ident)
        return G.one ? launch_tile_one<T, INV, true, QM, 1>(A, io, grid, G.threads, G.lds, s)
                     : launch_tile_one<T, INV, true, QM, 2>(A, io, grid, G.threads, G.lds, s);
    return G.one ? launch_tile_one<T, INV, false, QM, 1>(A, io, grid, G.threads, G.lds, s)
                 : launch_tile_one<T, INV, false, QM, 2>(A, io, grid, G.threads, G.lds, s);
}

// stages k0 .. k1 (tile stages of one launch shape, forward direction, one channel chunk) as ONE chained launch
template <typename T, bool QM>
static int launch_tile_chain(raht_plan *p, Schedule &sc, int k0, int k1, const XformIO<T> &io, int D, int Dc0, hipStream_t s)
{
    TileChain<T> C;
    TileGeom G0;
    C.n = k1 - k0 + 1;
    for (int i = 0; i < CHAIN_MAX; ++i) C.arrive[i] = nullptr;
    for (int k = k0; k <= k1; ++k) {
        Stage &st = sc.stages[(size_t)k];
        TileGeom G;
        RAHT_RET((prepare_tile_stage<T, false, QM>(p, sc, k, io, D, Dc0, 0, C.a[k - k0], G)));
        if (G.nchunks != 1) { set_error("tile chain: channel-chunked stage"); return RAHT_ERR_INVALID; }
        if (k == k0) G0 = G;
        else if (!G.same_shape(G0) || st.tile_rows < sc.stages[(size_t)k - 1].tile_rows) { set_error("tile chain: stages of different launch shapes"); return RAHT_ERR_INVALID; }
        if (k > k0 && !st.arrive) {
            RAHT_HIP_CHECK(dev_malloc(&st.arrive, sizeof(uint32_t) * (size_t)st.n_tiles));
            RAHT_HIP_CHECK(hipMemsetAsync(st.arrive, 0, sizeof(uint32_t) * (size_t)st.n_tiles, s));
        }
        C.arrive[k - k0] = st.arrive;
    }
    for (int i = C.n; i < CHAIN_MAX; ++i) C.a[i] = C.a[0];
    C.stack_off = (uint32_t)((G0.lds + 15) & ~(size_t)15);
    const size_t lds = C.stack_off + 64;
    const dim3 grid((unsigned)G0.n_tiles, 1);
    static PerDeviceOnce attr1, attr2;
    if (G0.one) {
        if (attr1.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_chain<T, QM, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    } else {
        if (attr2.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_chain<T, QM, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if constexpr (QM) {
        typename StepsFor<T>::type st;
        fill_step_table(st, io.steps, io.n_steps);
        if (G0.one) hipLaunchKernelGGL((tile_kernel_chain<T, true, 1>), grid, dim3(G0.threads), lds, s, C, st);
        else hipLaunchKernelGGL((tile_kernel_chain<T, true, 2>), grid, dim3(G0.threads), lds, s, C, st);
    } else {
        NoSteps ns{0, 0};
        if (G0.one) hipLaunchKernelGGL((tile_kernel_chain<T, false, 1>), grid, dim3(G0.threads), lds, s, C, ns);
        else hipLaunchKernelGGL((tile_kernel_chain<T, false, 2>), grid, dim3(G0.threads), lds, s, C, ns);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// the forward direction's stage sequence: stage 0, the later tile stages (optionally chained into one launch: see below), the top stage
template <typename T, bool QM>
static int launch_forward_stages(raht_plan *p, Schedule &sc, const XformIO<T> &io, int D, int Dc, hipStream_t s)
{
    // MEASURED, OFF by default (RAHT_CHAIN=1 switches it on): parity-green, but the fused cfg3 forward takes 0.339 ms chained
    // against 0.304 ms with one launch per stage -- every tile's agent-scope release is a write-back of its XCD's L2 (the survivor
    // rows must be visible to a parent that may run on another XCD), 705 of them cost more than the one launch (4.6 us) and the
    // ~6 us of stage-2 work they hide. (With the fence in every thread: 0.50 ms.)
    static const bool chain_on = getenv("RAHT_CHAIN") && atoi(getenv("RAHT_CHAIN")) != 0;
    const int K = (int)sc.stages.size();
    int k1 = K - 1;
    while (k1 >= 1 && sc.stages[(size_t)k1].is_top) --k1;          // last tile stage
    int r1 = 0, dc1 = 0, rf = 0;
    pick_tail_geometry(p, (int)sizeof(T), D, sc.tile_rows, &r1, &dc1, &rf);
    const bool chain = chain_on && k1 >= 2 && k1 <= CHAIN_MAX && dc1 >= D && !p->row_map;
    for (int k = 0; k < K; ++k) {
        if (chain && k == 1) {
            RAHT_RET((launch_tile_chain<T, QM>(p, sc, 1, k1, io, D, Dc, s)));
            k = k1;
            continue;
        }
        RAHT_RET((launch_tile_stage<T, false, QM>(p, sc, k, io, D, Dc, s)));
    }
    return RAHT_OK;
}

// the same tile stage of m <= TILE_BATCH_MAX scenes (equal launch shape) in one launch, one tile per workgroup
template <typename T, bool INV, bool QM>
static int launch_tile_batch(int m, const TileArgs<T> *As, const TileGeom *Gs, const XformIO<T> &io, hipStream_t s)
{
    TileBatch<T> B;
    B.n = m;
    uint32_t tot = 0;
    for (int i = 0; i < TILE_BATCH_MAX; ++i) {
        B.a[i] = As[i < m ? i : 0];
        B.first_tile[i] = tot;
        if (i < m) tot += (uint32_t)Gs[i].n_tiles;
    }
    B.first_tile[TILE_BATCH_MAX] = tot;
    const TileGeom &G = Gs[0];
    const dim3 grid(tot, G.nchunks);
    if (G.ident)
        return G.one ? launch_tile_batch_one<T, INV, true, QM, 1>(B, io, grid, G.threads, G.lds, s)
                     : launch_tile_batch_one<T, INV, true, QM, 2>(B, io, grid, G.threads, G.lds, s);
    return G.one ? launch_tile_batch_one<T, INV, false, QM, 1>(B, io, grid, G.threads, G.lds, s)
                 : launch_tile_batch_one<T, INV, false, QM, 2>(B, io, grid, G.threads, G.lds, s);
}

template <typename T>
__global__ void root_rows_kernel(T *__restrict__ mat, int64_t ld, int D, const uint32_t *__restrict__ rows,
                                 int64_t n_roots, T *__restrict__ buf, int to_buf, const uint32_t *__restrict__ row_map)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_roots * D) return;
    const int64_t q = e / D;
    const int c = (int)(e - q * D);
    const int64_t r = row_map ? (int64_t)row_map[rows[q]] : (int64_t)rows[q];
    if (to_buf) buf[e] = mat[r * ld + c];
    else mat[r * ld + c] = buf[e];
}

template <typename T, bool INV>
static int run_level_engine(const raht_plan *p, const T *src, int64_t ld_src, T *dst, int64_t ld_dst,
                            int D, hipStream_t s)
{
    RAHT_RET(ensure_level_rows(const_cast<raht_plan *>(p), s));
    const int64_t mat_rows = p->row_map ? p->map_rows : p->N;
    if ((const void *)src != (const void *)dst) {
        const int64_t total = mat_rows * D;
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(total, 256), 8192);
        hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(gb), dim3(256), 0, s, src, ld_src, dst, ld_dst, mat_rows, D);
    }
    const unsigned gr = (unsigned)ceil_div(p->n_roots * D, 256);
    if (INV && p->root_buf)
        hipLaunchKernelGGL(root_rows_kernel<T>, dim3(gr), dim3(256), 0, s, dst, ld_dst, D, p->root_rows, p->n_roots,
                           (T *)p->root_buf, 0, p->row_map);
    const int lps = lp_shift_for(std::min(D, 64));
    const int gpw = 64 >> lps;
    const int top = std::min(p->max_level, p->top_level - 1);
    for (int q = 0; q <= top; ++q) {
        const int l = INV ? top - q : q;
        const uint32_t cnt = p->level_off[l + 1] - p->level_off[l];
        if (cnt == 0) continue;                                  // RAHT.py:304-305
        const int64_t steps = ceil_div(cnt, gpw * 4);
        const unsigned gb = (unsigned)std::min<int64_t>(steps, 2048);
        hipLaunchKernelGGL((level_pass_kernel<T, INV>), dim3(gb), dim3(256), 0, s, dst, ld_dst, D,
                           p->level_rows + p->level_off[l], cnt, p->wl, p->wr, p->wsum, lps, p->row_map);
    }
    if (!INV && p->root_buf)
        hipLaunchKernelGGL(root_rows_kernel<T>, dim3(gr), dim3(256), 0, s, dst, ld_dst, D, p->root_rows, p->n_roots,
                           (T *)p->root_buf, 1, p->row_map);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// Tile schedule (+ workspaces) for this element type / channel count; nullptr -> use the level engine.
template <typename T>
static int tile_setup(raht_plan *p, int D, int64_t max_ld, hipStream_t s, Schedule **sc_out, int *Dc_out)
{
    *sc_out = nullptr;
    if (p->row_map && p->N > RAHT_TOP_MAX_ROWS) {
        set_error("row-mapped plans hold at most %d rows (they run as one top stage)", RAHT_TOP_MAX_ROWS);
        return RAHT_ERR_UNSUPPORTED;
    }
    if (p->engine == RAHT_ENGINE_LEVEL) return RAHT_OK;
    if (max_ld > ((int64_t)1 << 18)) return RAHT_OK;  // row_at(): 32-bit byte offsets inside a tile; wider strides: level engine
    if (D < 16 / (int)sizeof(T)) return RAHT_OK;      // rows shorter than one 16-byte chunk: level engine
    const int Dc = pick_chunk_channels((int)sizeof(T), D);
    const int R = pick_tile_rows(p, (int)sizeof(T), Dc);
    if (R == 0) return RAHT_OK;
    Schedule *sc = nullptr;
    int R1 = 0, Dc1 = 0, Rf = 0;
    pick_tail_geometry(p, (int)sizeof(T), D, R, &R1, &Dc1, &Rf);
    if (p->row_map) Rf = RAHT_TOP_MAX_ROWS;           // ONE top stage (the only kernel that addresses rows through the map)
    RAHT_RET(get_schedule(p, R, R1, Rf, s, &sc));
    if (!sc->valid) return RAHT_OK;                   // pathological key pattern, see plan.hip
    RAHT_RET(ensure_workspace(sc, (size_t)D * sizeof(T), p->split_ws));
    *sc_out = sc;
    *Dc_out = Dc;
    return RAHT_OK;
}

template <typename T, bool INV>
static int run_transform(const raht_plan *cp, const T *src, int64_t ld_src, int D, T *dst, int64_t ld_dst,
                         T *w, hipStream_t s)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    if (!p || !src || !dst) { set_error("raht transform: NULL argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, INV ? "raht_inv" : "raht_fwd"));
    if (D < 1 || ld_src < D || ld_dst < D) { set_error("raht transform: bad D/ld (D=%d ld_src=%lld ld_dst=%lld)", D, (long long)ld_src, (long long)ld_dst); return RAHT_ERR_INVALID; }
    Schedule *sc = nullptr;
    int Dc = 0;
    RAHT_RET(tile_setup<T>(p, D, std::max(ld_src, ld_dst), s, &sc, &Dc));
    int rc = RAHT_OK;
    if (!sc) {
        rc = run_level_engine<T, INV>(p, src, ld_src, dst, ld_dst, D, s);
    } else {
        XformIO<T> io;
        io.src = src; io.ld_src = ld_src; io.dst = dst; io.ld_dst = ld_dst;
        const int K = (int)sc->stages.size();
        if constexpr (!INV) {
            rc = launch_forward_stages<T, false>(p, *sc, io, D, Dc, s);
        } else {
            for (int q = 0; q < K && rc == RAHT_OK; ++q) rc = launch_tile_stage<T, INV, false>(p, *sc, K - 1 - q, io, D, Dc, s);
        }
    }
    if (rc == RAHT_OK && w && p->row_map) { set_error("node weights are not available from a row-mapped plan"); return RAHT_ERR_UNSUPPORTED; }
    if (rc == RAHT_OK && w) {
        hipLaunchKernelGGL(node_weight_kernel<T>, dim3((unsigned)ceil_div(p->N, 256)), dim3(256), 0, s,
                           p->wl, p->wr, p->wsum, p->N, w);
        RAHT_HIP_CHECK(hipGetLastError());
    }
    return rc;
}

template <typename S>
static int check_steps(const S *steps, int n_steps, int D)
{
    if (!steps || !(n_steps == 1 || n_steps == D)) { set_error("quant: n_steps must be 1 or D"); return RAHT_ERR_INVALID; }
    if (n_steps > MAX_STEP_CH) { set_error("quant: per-channel steps support D <= %d", MAX_STEP_CH); return RAHT_ERR_UNSUPPORTED; }
    for (int c = 0; c < n_steps; ++c)
        if (!(steps[c] > (S)0)) { set_error("quant: step[%d] must be > 0", c); return RAHT_ERR_INVALID; }
    return RAHT_OK;
}

// the two-pass entry points (quant.hip), by element type: what the fused entry points fall back to
static int quant_reorder_any(const raht_plan *p, const float *T, int64_t ldt, int D, const float *st, int n, int32_t *Q, int64_t ldq, raht_stream_t s)
{ return raht_quant_reorder(p, T, ldt, D, st, n, Q, ldq, s); }
static int quant_reorder_any(const raht_plan *p, const double *T, int64_t ldt, int D, const double *st, int n, int32_t *Q, int64_t ldq, raht_stream_t s)
{ return raht_quant_reorder_f64(p, T, ldt, D, st, n, Q, ldq, s); }
static int dequant_unreorder_any(const raht_plan *p, const int32_t *Q, int64_t ldq, int D, const float *st, int n, float *T, int64_t ldt, raht_stream_t s)
{ return raht_dequant_unreorder(p, Q, ldq, D, st, n, T, ldt, s); }
static int dequant_unreorder_any(const raht_plan *p, const int32_t *Q, int64_t ldq, int D, const double *st, int n, double *T, int64_t ldt, raht_stream_t s)
{ return raht_dequant_unreorder_f64(p, Q, ldq, D, st, n, T, ldt, s); }

/* Fused forward RAHT + quantize + reorder: Q[k, c] = floor(T[order[k], c] / step_c + 0.5) without
 * ever materialising T (encode_3dgs.py:159,204,210,215 in one pass). */
template <typename T>
static int fwd_quant_impl(const raht_plan *cp, const T *C, int64_t ldc, int D, const T *steps, int n_steps,
                          int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    if (!p || !C || !Q || D < 1 || ldc < D || ldq < D) { set_error("raht_fwd_quant: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_fwd_quant"));
    if (p->row_map) { set_error("raht_fwd_quant: not available for a row-mapped plan"); return RAHT_ERR_UNSUPPORTED; }
    RAHT_RET(check_steps(steps, n_steps, D));
    Schedule *sc = nullptr;
    int Dc = 0;
    RAHT_RET(tile_setup<T>(p, D, std::max(ldc, ldq), s, &sc, &Dc));
    if (!sc) {
        // level engine (selected explicitly, or fallback for pathological key patterns): two passes
        // through a pooled temporary (stream-ordered reuse; see Scratch in raht_common.h)
        Scratch tmp(sizeof(T) * (size_t)p->N * (size_t)D, s);
        if (!tmp.ok()) return RAHT_ERR_NOMEM;
        RAHT_RET((run_level_engine<T, false>(p, C, ldc, tmp.as<T>(), D, D, s)));
        return quant_reorder_any(p, tmp.as<T>(), D, D, steps, n_steps, Q, ldq, stream);
    }
    XformIO<T> io;
    io.src = C; io.ld_src = ldc; io.Q = Q; io.ldq = ldq; io.steps = steps; io.n_steps = n_steps;
    return launch_forward_stages<T, true>(p, *sc, io, D, Dc, s);
}

/* Fused un-reorder + dequantize + inverse RAHT (encode_3dgs.py:261,267-268,274 in one pass). */
template <typename T>
static int dequant_inv_impl(const raht_plan *cp, const int32_t *Q, int64_t ldq, int D, const T *steps, int n_steps,
                            T *C, int64_t ldc, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    if (!p || !C || !Q || D < 1 || ldc < D || ldq < D) { set_error("raht_dequant_inv: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_dequant_inv"));
    if (p->row_map) { set_error("raht_dequant_inv: not available for a row-mapped plan"); return RAHT_ERR_UNSUPPORTED; }
    RAHT_RET(check_steps(steps, n_steps, D));
    Schedule *sc = nullptr;
    int Dc = 0;
    RAHT_RET(tile_setup<T>(p, D, std::max(ldc, ldq), s, &sc, &Dc));
    if (!sc) {
        Scratch tmp(sizeof(T) * (size_t)p->N * (size_t)D, s);
        if (!tmp.ok()) return RAHT_ERR_NOMEM;
        RAHT_RET(dequant_unreorder_any(p, Q, ldq, D, steps, n_steps, tmp.as<T>(), D, stream));
        return run_level_engine<T, true>(p, tmp.as<T>(), D, C, ldc, D, s);
    }
    XformIO<T> io;
    io.dst = C; io.ld_dst = ldc; io.Q = const_cast<int32_t *>(Q); io.ldq = ldq; io.steps = steps; io.n_steps = n_steps;
    const int K = (int)sc->stages.size();
    for (int k = K - 1; k >= 0; --k) RAHT_RET((launch_tile_stage<T, true, true>(p, *sc, k, io, D, Dc, s)));
    return RAHT_OK;
}

/* One forward pass, k quantizations (python/encode_3dgs.py:28,199-217: the drivers quantize ONE coefficient matrix at nine steps):
 * Q[i] = floor(T / steps[i] + 0.5), reordered, for i < k, each bit-identical to raht_fwd_quant(..., &steps[i], 1, Q[i], ...). */
static int fwd_quant_multi_impl(const raht_plan *cp, const float *C, int64_t ldc, int D, const float *steps, int k, int32_t *const *Q,
                                int64_t ldq, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    if (!p || !C || !Q || !steps || k < 1 || D < 1 || ldc < D || ldq < D) { set_error("raht_fwd_quant_multi: bad argument"); return RAHT_ERR_INVALID; }
    for (int i = 0; i < k; ++i) {
        if (!Q[i] || !(steps[i] > 0.0f)) { set_error("raht_fwd_quant_multi: Q[%d] / steps[%d]", i, i); return RAHT_ERR_INVALID; }
        for (int j = 0; j < i; ++j) if (Q[j] == Q[i]) { set_error("raht_fwd_quant_multi: Q[%d] and Q[%d] are the same matrix", j, i); return RAHT_ERR_INVALID; }
    }
    RAHT_RET(check_plan_device(p, "raht_fwd_quant_multi"));
    Schedule *sc = nullptr;
    int Dc = 0;
    if (!p->row_map) RAHT_RET(tile_setup<float>(p, D, std::max(ldc, ldq), s, &sc, &Dc));
    if (!sc || p->row_map || p->root_buf || p->top_level < 64) {
        // level engine, row-mapped / truncated plans: one call per step
        for (int i = 0; i < k; ++i) RAHT_RET(fwd_quant_impl<float>(p, C, ldc, D, &steps[i], 1, Q[i], ldq, stream));
        return RAHT_OK;
    }
    for (int k0 = 0; k0 < k; k0 += MULTI_Q_MAX) {
        MultiQ M;
        M.k = std::min(MULTI_Q_MAX, k - k0);
        M.fast_div = 1;
        for (int i = 0; i < MULTI_Q_MAX; ++i) {
            M.step[i] = steps[k0 + std::min(i, M.k - 1)];
            M.Q[i] = Q[k0 + std::min(i, M.k - 1)];
            if (!(M.step[i] >= 0x1p-100f && M.step[i] <= 0x1p100f)) M.fast_div = 0;
        }
        XformIO<float> io;
        io.src = C; io.ld_src = ldc; io.Q = M.Q[0]; io.ldq = ldq; io.steps = &steps[k0]; io.n_steps = 1;
        StepTable st;
        fill_step_table(st, io.steps, 1);
        const int K = (int)sc->stages.size();
        for (int kk = 0; kk < K; ++kk) {
            const Stage &stg = sc->stages[(size_t)kk];
            if (stg.is_top) {
                TopArgs<float> A;
                size_t lds = 0;
                RAHT_RET((prepare_top_stage<float, false, true>(p, *sc, kk, io, D, A, lds)));
                static PerDeviceOnce attr;
                if (attr.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)top_kernel_multi, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
                hipLaunchKernelGGL(top_kernel_multi, dim3((unsigned)((D + 3) / 4)), dim3(TOP_THREADS), lds, s, A, st, M);
            } else {
                TileArgs<float> A;
                TileGeom G;
                RAHT_RET((prepare_tile_stage<float, false, true>(p, *sc, kk, io, D, Dc, 0, A, G)));
                const dim3 grid(G.grid_x, G.nchunks);
                static PerDeviceOnce a11, a12, a01, a02;
                if (kk == 0 && p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_before, s));
                if (G.ident && G.one) { if (a11.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_multi<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                                        hipLaunchKernelGGL((tile_kernel_multi<true, 1>), grid, dim3(G.threads), G.lds, s, A, st, M); }
                else if (G.ident) { if (a12.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_multi<true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                                    hipLaunchKernelGGL((tile_kernel_multi<true, 2>), grid, dim3(G.threads), G.lds, s, A, st, M); }
                else if (G.one) { if (a01.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_multi<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                                  hipLaunchKernelGGL((tile_kernel_multi<false, 1>), grid, dim3(G.threads), G.lds, s, A, st, M); }
                else { if (a02.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_multi<false, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                       hipLaunchKernelGGL((tile_kernel_multi<false, 2>), grid, dim3(G.threads), G.lds, s, A, st, M); }
                if (kk == 0 && p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_after, s));
            }
            RAHT_HIP_CHECK(hipGetLastError());
        }
    }
    return RAHT_OK;
}

/* raht_dequant_inv fused with the drivers' distortion measurement (python/encode_3dgs.py:274,298-310: C_rec = iRAHT(...), then
 * torch.mean((C - C_rec) ** 2) over all / quats / scales / opacity / colour columns): the stage-0 kernel of the fused inverse
 * compares every row it reconstructs with the original on its way out and leaves per-column sums of squared differences; C_rec
 * itself is written only if the caller passes a buffer. One pass over Q and C instead of Q -> C_rec, then C and C_rec again. */
static int dequant_inv_sqdiff_impl(const raht_plan *cp, const int32_t *Q, int64_t ldq, int D, const float *steps, int n_steps,
                                   const float *Cref, int64_t ldref, float *Crec, int64_t ldc, double *sq, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    if (!p || !Q || !Cref || !sq || D < 1 || ldq < D || ldref < D || (Crec && ldc < D)) { set_error("raht_dequant_inv_sqdiff: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_dequant_inv_sqdiff"));
    if (p->row_map) { set_error("raht_dequant_inv_sqdiff: not available for a row-mapped plan"); return RAHT_ERR_UNSUPPORTED; }
    RAHT_RET(check_steps(steps, n_steps, D));
    Schedule *sc = nullptr;
    int Dc = 0;
    RAHT_RET(tile_setup<float>(p, D, std::max(std::max(ldref, ldq), Crec ? ldc : (int64_t)D), s, &sc, &Dc));
    const bool fusable = sc && Dc >= D && sc->stages.size() >= 2 && !sc->stages[0].is_top && !p->root_buf && p->top_level >= 64;
    if (!fusable) {
        // level engine, channel-chunked rows (D > 64), one-launch trees, truncated plans: the two passes
        Scratch tmp(Crec ? 16 : sizeof(float) * (size_t)p->N * (size_t)D, s);
        if (!tmp.ok()) return RAHT_ERR_NOMEM;
        float *out = Crec ? Crec : tmp.as<float>();
        const int64_t ldo = Crec ? ldc : D;
        RAHT_RET(dequant_inv_impl<float>(p, Q, ldq, D, steps, n_steps, out, ldo, stream));
        return raht_sqdiff_columns(Cref, ldref, out, ldo, p->N, D, RAHT_F32, sq, stream);
    }
    XformIO<float> io;
    io.dst = Crec; io.ld_dst = Crec ? ldc : D; io.Q = const_cast<int32_t *>(Q); io.ldq = ldq; io.steps = steps; io.n_steps = n_steps;
    const int K = (int)sc->stages.size();
    for (int k = K - 1; k >= 1; --k) RAHT_RET((launch_tile_stage<float, true, true>(p, *sc, k, io, D, Dc, s)));
    // stage 0: the comparing kernel
    TileArgs<float> A;
    TileGeom G;
    XformIO<float> io0 = io;
    if (!Crec) io0.dst = const_cast<float *>(Cref);            // (only so that the argument check sees an output; overridden below)
    RAHT_RET((prepare_tile_stage<float, true, true>(p, *sc, 0, io0, D, Dc, 0, A, G)));
    const int ncv = ((D + 3) / 4) * 4;
    Scratch part(sizeof(double) * (size_t)G.n_tiles * (size_t)ncv, s);
    if (!part.ok()) return RAHT_ERR_NOMEM;
    A.out = Crec; A.ld_out = Crec ? ldc : 0;
    A.ref = Cref; A.ld_ref = ldref; A.sq_part = part.as<double>();
    if (G.nchunks != 1 || !G.ident || G.grid_x != (unsigned)G.n_tiles || (size_t)G.threads / 64 * (size_t)ncv * 8 > G.lds) {
        set_error("raht_dequant_inv_sqdiff: unexpected stage-0 geometry");
        return RAHT_ERR_INVALID;
    }
    StepTable st;
    fill_step_table(st, steps, n_steps);
    static PerDeviceOnce attr1, attr2;
    if (p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_before, s));
    if (G.one) {
        if (attr1.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_sq<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((tile_kernel_sq<1>), dim3(G.grid_x), dim3(G.threads), G.lds, s, A, st);
    } else {
        if (attr2.first(current_device())) RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_sq<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL((tile_kernel_sq<2>), dim3(G.grid_x), dim3(G.threads), G.lds, s, A, st);
    }
    if (p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_after, s));
    hipLaunchKernelGGL(sq_final_kernel, dim3((unsigned)D), dim3(256), 0, s, part.as<double>(), G.n_tiles, D, ncv, sq);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// ---- several scenes, one set of launches (raht_*_batch) ----------------------------------------------
// Round r of the forward direction runs stage r of every scene that has one (inverse: the stages from the top of the
// deepest schedule downwards, a scene joining when its own top stage comes up); within a round the tile stages of equal
// launch shape go out TILE_BATCH_MAX scenes per launch, the top stages likewise with blockIdx.y = scene. Scenes without a
// tile schedule run through the single-scene path.
template <typename T, bool INV, bool QM>
static int run_batch(int n, raht_plan *const *plans, const XformIO<T> *ios, int D, hipStream_t s, const char *what)
{
    if (n < 1 || !plans || !ios) { set_error("%s: bad argument", what); return RAHT_ERR_INVALID; }
    std::vector<Schedule *> scs((size_t)n, nullptr);
    std::vector<int> Dcs((size_t)n, 0);
    int maxK = 0;
    for (int i = 0; i < n; ++i) {
        raht_plan *p = plans[i];
        const XformIO<T> &io = ios[i];
        if (!p) { set_error("%s: NULL plan (scene %d)", what, i); return RAHT_ERR_INVALID; }
        for (int j = 0; j < i; ++j) if (plans[j] == p) { set_error("%s: scenes %d and %d share a plan (a plan owns its workspaces)", what, j, i); return RAHT_ERR_INVALID; }
        RAHT_RET(check_plan_device(p, what));
        const int64_t ld_a = INV ? (QM ? io.ldq : io.ld_src) : io.ld_src, ld_b = INV ? io.ld_dst : (QM ? io.ldq : io.ld_dst);
        const void *pa = INV ? (QM ? (const void *)io.Q : (const void *)io.src) : (const void *)io.src;
        const void *pb = INV ? (const void *)io.dst : (QM ? (const void *)io.Q : (const void *)io.dst);
        if (!pa || !pb || D < 1 || ld_a < D || ld_b < D) { set_error("%s: bad matrix argument (scene %d)", what, i); return RAHT_ERR_INVALID; }
        if (p->row_map) continue;                              // single-scene path (one mapped top stage)
        RAHT_RET(tile_setup<T>(p, D, std::max(ld_a, ld_b), s, &scs[(size_t)i], &Dcs[(size_t)i]));
        if (scs[(size_t)i]) maxK = std::max(maxK, (int)scs[(size_t)i]->stages.size());
    }
    // scenes outside the tile engine: their own entry point, in place in the stream
    for (int i = 0; i < n; ++i) {
        if (scs[(size_t)i]) continue;
        const XformIO<T> &io = ios[i];
        int rc;
        if constexpr (QM && !INV) rc = fwd_quant_impl<T>(plans[i], io.src, io.ld_src, D, io.steps, io.n_steps, io.Q, io.ldq, (raht_stream_t)s);
        else if constexpr (QM && INV) rc = dequant_inv_impl<T>(plans[i], io.Q, io.ldq, D, io.steps, io.n_steps, io.dst, io.ld_dst, (raht_stream_t)s);
        else rc = run_transform<T, INV>(plans[i], io.src, io.ld_src, D, io.dst, io.ld_dst, nullptr, s);
        RAHT_RET(rc);
    }
    for (int r = 0; r < maxK; ++r) {
        int idx_tile[TILE_BATCH_MAX], idx_top[TILE_BATCH_MAX], n_tile = 0, n_top = 0;
        TileArgs<T> At[TILE_BATCH_MAX];
        TileGeom Gt[TILE_BATCH_MAX];
        TopArgs<T> Ap[TILE_BATCH_MAX];
        size_t Lp[TILE_BATCH_MAX];
        auto flush_tile = [&]() -> int {
            if (n_tile == 0) return RAHT_OK;
            int rc;
            if (n_tile == 1) {
                const int i = idx_tile[0];
                rc = launch_stage_impl<T, INV, QM>(plans[i], *scs[(size_t)i], INV ? maxK - 1 - r : r, ios[i], D, Dcs[(size_t)i], s, 0);
            } else {
                rc = launch_tile_batch<T, INV, QM>(n_tile, At, Gt, ios[idx_tile[0]], s);
            }
            n_tile = 0;
            return rc;
        };
        auto flush_top = [&]() -> int {
            if (n_top == 0) return RAHT_OK;
            const int rc = launch_top_batch<T, INV, QM>(n_top, Ap, Lp, ios[idx_top[0]], D, s);
            n_top = 0;
            return rc;
        };
        for (int i = 0; i < n; ++i) {
            Schedule *sc = scs[(size_t)i];
            if (!sc) continue;
            const int K = (int)sc->stages.size();
            const int k = INV ? maxK - 1 - r : r;
            if (k < 0 || k >= K) continue;
            if (sc->stages[(size_t)k].is_top) {
                RAHT_RET((prepare_top_stage<T, INV, QM>(plans[i], *sc, k, ios[i], D, Ap[n_top], Lp[n_top])));
                idx_top[n_top++] = i;
                if (n_top == TILE_BATCH_MAX) RAHT_RET(flush_top());
            } else {
                TileArgs<T> A;
                TileGeom G;
                RAHT_RET((prepare_tile_stage<T, INV, QM>(plans[i], *sc, k, ios[i], D, Dcs[(size_t)i], 0, A, G)));
                if (n_tile > 0 && !G.same_shape(Gt[0])) RAHT_RET(flush_tile());     // another launch shape: its own launch
                At[n_tile] = A; Gt[n_tile] = G; idx_tile[n_tile++] = i;
                if (n_tile == TILE_BATCH_MAX) RAHT_RET(flush_tile());
            }
        }
        RAHT_RET(flush_tile());
        RAHT_RET(flush_top());
    }
    return RAHT_OK;
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_fwd(const raht_plan *plan, const float *C, int64_t ldc, int D, float *T, int64_t ldt, float *w,
             raht_stream_t stream)
{
    return guarded("raht_fwd", [&]() { return run_transform<float, false>(plan, C, ldc, D, T, ldt, w, (hipStream_t)stream); });
}

int raht_fwd_f64(const raht_plan *plan, const double *C, int64_t ldc, int D, double *T, int64_t ldt,
                 double *w, raht_stream_t stream)
{
    return guarded("raht_fwd_f64", [&]() { return run_transform<double, false>(plan, C, ldc, D, T, ldt, w, (hipStream_t)stream); });
}

int raht_inv(const raht_plan *plan, const float *T, int64_t ldt, int D, float *C, int64_t ldc,
             raht_stream_t stream)
{
    return guarded("raht_inv", [&]() { return run_transform<float, true>(plan, T, ldt, D, C, ldc, nullptr, (hipStream_t)stream); });
}

int raht_inv_f64(const raht_plan *plan, const double *T, int64_t ldt, int D, double *C, int64_t ldc,
                 raht_stream_t stream)
{
    return guarded("raht_inv_f64", [&]() { return run_transform<double, true>(plan, T, ldt, D, C, ldc, nullptr, (hipStream_t)stream); });
}

int raht_fwd_quant(const raht_plan *plan, const float *C, int64_t ldc, int D, const float *steps, int n_steps,
                   int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    return guarded("raht_fwd_quant", [&]() { return fwd_quant_impl<float>(plan, C, ldc, D, steps, n_steps, Q, ldq, stream); });
}

/* The same at the reference's own precision (float64 coefficients are what encode_3dgs.py:204 quantizes): the float64
 * tile kernels with the float64 quantizer in their write-back. */
int raht_fwd_quant_f64(const raht_plan *plan, const double *C, int64_t ldc, int D, const double *steps, int n_steps,
                       int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    return guarded("raht_fwd_quant_f64", [&]() { return fwd_quant_impl<double>(plan, C, ldc, D, steps, n_steps, Q, ldq, stream); });
}

int raht_dequant_inv(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps, int n_steps,
                     float *C, int64_t ldc, raht_stream_t stream)
{
    return guarded("raht_dequant_inv", [&]() { return dequant_inv_impl<float>(plan, Q, ldq, D, steps, n_steps, C, ldc, stream); });
}

int raht_fwd_quant_multi(const raht_plan *plan, const float *C, int64_t ldc, int D, const float *steps, int k, int32_t *const *Q,
                         int64_t ldq, raht_stream_t stream)
{
    return guarded("raht_fwd_quant_multi", [&]() { return fwd_quant_multi_impl(plan, C, ldc, D, steps, k, Q, ldq, stream); });
}

int raht_dequant_inv_sqdiff(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const float *steps, int n_steps,
                            const float *C_ref, int64_t ld_ref, float *C_rec, int64_t ldc, double *sqdiff, raht_stream_t stream)
{
    return guarded("raht_dequant_inv_sqdiff", [&]() { return dequant_inv_sqdiff_impl(plan, Q, ldq, D, steps, n_steps, C_ref, ld_ref, C_rec, ldc, sqdiff, stream); });
}

int raht_dequant_inv_f64(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps,
                         double *C, int64_t ldc, raht_stream_t stream)
{
    return guarded("raht_dequant_inv_f64", [&]() { return dequant_inv_impl<double>(plan, Q, ldq, D, steps, n_steps, C, ldc, stream); });
}

/* Pre-build the tile schedule and workspaces for (elem_size, D) so that later transform calls
 * neither allocate nor synchronise (e.g. before hipGraph capture). */
int raht_plan_prepare(raht_plan *p, int elem_size, int D, raht_stream_t stream)
{
    if (!p || (elem_size != 4 && elem_size != 8) || D < 1) { set_error("raht_plan_prepare: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_plan_prepare"));
    return guarded("raht_plan_prepare", [&]() {
        Schedule *sc = nullptr;
        int Dc = 0;
        if (elem_size == 4) return tile_setup<float>(p, D, D, (hipStream_t)stream, &sc, &Dc);
        return tile_setup<double>(p, D, D, (hipStream_t)stream, &sc, &Dc);
    });
}

#ifdef RAHT_PHASE_CLOCKS
int raht_debug_read_phase_clocks(unsigned long long *dst, int n_tiles)
{
    RAHT_HIP_CHECK(hipDeviceSynchronize());
    RAHT_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase_clk), sizeof(unsigned long long) * PHASE_CLK_SLOTS * (size_t)std::min(n_tiles, PHASE_CLK_TILES)));
    return PHASE_CLK_SLOTS;
}
#endif

/* Profiling aid: enqueue ONE stage of the float32 tile schedule (stage 0 = the HBM-heavy launch).
 * Results are only meaningful as part of a full transform; bench.py uses this to time the dominant
 * kernel in isolation with HIP events. Q != NULL selects the fused-quantization kernels
 * (forward: mat = C in, Q out; inverse: Q in, mat = C out), Q == NULL the plain ones
 * (forward: mat = C in, mat2 = T out; inverse: mat = T in, mat2 = C out). */
static int debug_run_stage_impl(const raht_plan *cp, int inverse, int stage, const float *mat, int64_t ld_mat, int D,
                         float *mat2, int64_t ld_mat2, int32_t *Q, int64_t ldq, float step, int ablate,
                         raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    if (!p || D < 1 || (!Q && (!mat || !mat2)) || (Q && !(inverse ? (const void *)mat2 : (const void *)mat))) {
        set_error("raht_debug_run_stage: bad argument");
        return RAHT_ERR_INVALID;
    }
    RAHT_RET(check_plan_device(p, "raht_debug_run_stage"));
#ifndef RAHT_ABLATE
    if (ablate != 0) { set_error("raht_debug_run_stage: ablations need a -DRAHT_ABLATE build of the library (make ABLATE=1)"); return RAHT_ERR_UNSUPPORTED; }
#endif
    Schedule *sc = nullptr;
    int Dc = 0;
    RAHT_RET(tile_setup<float>(p, D, std::max(std::max(ld_mat, ld_mat2), ldq), s, &sc, &Dc));
    if (!sc) { set_error("raht_debug_run_stage: tile engine unavailable"); return RAHT_ERR_UNSUPPORTED; }
    if (stage < 0 || stage >= (int)sc->stages.size()) { set_error("raht_debug_run_stage: stage out of range"); return RAHT_ERR_INVALID; }
    XformIO<float> io;
    if (!Q) {
        io.src = mat; io.ld_src = ld_mat; io.dst = mat2; io.ld_dst = ld_mat2;
        if (inverse) return launch_tile_stage<float, true, false>(p, *sc, stage, io, D, Dc, s, ablate);
        return launch_tile_stage<float, false, false>(p, *sc, stage, io, D, Dc, s, ablate);
    }
    io.Q = Q; io.ldq = ldq; io.steps = &step; io.n_steps = 1;
    if (inverse) {
        io.dst = mat2; io.ld_dst = ld_mat2;
        return launch_tile_stage<float, true, true>(p, *sc, stage, io, D, Dc, s, ablate);
    }
    io.src = mat; io.ld_src = ld_mat;
    return launch_tile_stage<float, false, true>(p, *sc, stage, io, D, Dc, s, ablate);
}


int raht_fwd_batch(int n, raht_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                   float *const *T, const int64_t *ldt, raht_stream_t stream)
{
    return guarded("raht_fwd_batch", [&]() -> int {
        if (n < 1 || !plans || !C || !ldc || !T || !ldt) { set_error("raht_fwd_batch: bad argument"); return RAHT_ERR_INVALID; }
        std::vector<XformIO<float>> ios((size_t)n);
        for (int i = 0; i < n; ++i) { ios[(size_t)i].src = C[i]; ios[(size_t)i].ld_src = ldc[i]; ios[(size_t)i].dst = T[i]; ios[(size_t)i].ld_dst = ldt[i]; }
        return run_batch<float, false, false>(n, plans, ios.data(), D, (hipStream_t)stream, "raht_fwd_batch");
    });
}

int raht_inv_batch(int n, raht_plan *const *plans, const float *const *T, const int64_t *ldt, int D,
                   float *const *C, const int64_t *ldc, raht_stream_t stream)
{
    return guarded("raht_inv_batch", [&]() -> int {
        if (n < 1 || !plans || !C || !ldc || !T || !ldt) { set_error("raht_inv_batch: bad argument"); return RAHT_ERR_INVALID; }
        std::vector<XformIO<float>> ios((size_t)n);
        for (int i = 0; i < n; ++i) { ios[(size_t)i].src = T[i]; ios[(size_t)i].ld_src = ldt[i]; ios[(size_t)i].dst = C[i]; ios[(size_t)i].ld_dst = ldc[i]; }
        return run_batch<float, true, false>(n, plans, ios.data(), D, (hipStream_t)stream, "raht_inv_batch");
    });
}

int raht_fwd_quant_batch(int n, raht_plan *const *plans, const float *const *C, const int64_t *ldc, int D,
                         const float *steps, int n_steps, int32_t *const *Q, const int64_t *ldq, raht_stream_t stream)
{
    return guarded("raht_fwd_quant_batch", [&]() -> int {
        if (n < 1 || !plans || !C || !ldc || !Q || !ldq) { set_error("raht_fwd_quant_batch: bad argument"); return RAHT_ERR_INVALID; }
        RAHT_RET(check_steps(steps, n_steps, D));
        std::vector<XformIO<float>> ios((size_t)n);
        for (int i = 0; i < n; ++i) {
            XformIO<float> &io = ios[(size_t)i];
            io.src = C[i]; io.ld_src = ldc[i]; io.Q = Q[i]; io.ldq = ldq[i]; io.steps = steps; io.n_steps = n_steps;
        }
        return run_batch<float, false, true>(n, plans, ios.data(), D, (hipStream_t)stream, "raht_fwd_quant_batch");
    });
}

int raht_dequant_inv_batch(int n, raht_plan *const *plans, const int32_t *const *Q, const int64_t *ldq, int D,
                           const float *steps, int n_steps, float *const *C, const int64_t *ldc, raht_stream_t stream)
{
    return guarded("raht_dequant_inv_batch", [&]() -> int {
        if (n < 1 || !plans || !C || !ldc || !Q || !ldq) { set_error("raht_dequant_inv_batch: bad argument"); return RAHT_ERR_INVALID; }
        RAHT_RET(check_steps(steps, n_steps, D));
        std::vector<XformIO<float>> ios((size_t)n);
        for (int i = 0; i < n; ++i) {
            XformIO<float> &io = ios[(size_t)i];
            io.dst = C[i]; io.ld_dst = ldc[i]; io.Q = const_cast<int32_t *>(Q[i]); io.ldq = ldq[i]; io.steps = steps; io.n_steps = n_steps;
        }
        return run_batch<float, true, true>(n, plans, ios.data(), D, (hipStream_t)stream, "raht_dequant_inv_batch");
    });
}

int raht_debug_run_stage(const raht_plan *plan, int inverse, int stage, const float *mat, int64_t ld_mat, int D,
                         float *mat2, int64_t ld_mat2, int32_t *Q, int64_t ldq, float step, int ablate,
                         raht_stream_t stream)
{
    return guarded("raht_debug_run_stage", [&]() { return debug_run_stage_impl(plan, inverse, stage, mat, ld_mat, D, mat2, ld_mat2, Q, ldq, step, ablate, stream); });
}

}  // extern "C"
