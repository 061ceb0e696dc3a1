// transform_mx.hip -- MIXED-PRECISION fused RAHT kernels: float32 rows whose first `n_wide` channels are carried in float64.
//
// Why: the reference quantizes float64 coefficients (python/encode_3dgs.py:82-83 DTYPE = float64, :204 floor(Coeff / step + 0.5)).
// On a 59-column frame (python/voxelize_pc.py:155: columns 0-2 of PCvox are the voxel coordinates, 0 .. 2^J - 1) the xyz
// coefficients reach 1e6, so at step 0.01 the quotient exceeds 2^24 and no float32 pipeline can return the reference's integers
// there (round 3: up to 56 units off), while the 56 attribute columns are fine in float32. The all-float64 kernels return the
// reference's integers everywhere at half the throughput. Here ONE set of launches carries the wide-range columns in float64 --
// float32 input converted exactly, float64 butterflies with float64 a / b, IEEE double division by the float64 step -- and the
// other columns in float32, bit-identical to raht_fwd_quant / raht_dequant_inv.
//
// Tile layout. An LDS row is NCp 16-byte chunk places: NW2 = ceil(n_wide / 2) places of two doubles, then NF = ceil((D - n_wide) / 4)
// places of four floats (the last one = the 16 bytes that END the row, as in transform.hip). 59 channels, 3 wide: 2 + 14 = 16
// places = 256 bytes = one lane group of 16 with no idle lane. A lane owns one chunk place of a row for the whole kernel; the
// wide lanes take the float64 branch of every butterfly (v_fma_f64 issues at the rate of v_fma_f32 on this chip), the others the
// float32 branch. Rows travel HBM -> LDS with global_load_lds_dwordx4 as in the float32 kernels: the float chunks to their places,
// the row's first 16 bytes (the n_wide <= 4 wide channels, raw float32 / int32) into the wide area, where the lane that loaded
// them widens them in place once they have landed. The workspaces between stages hold LDS row images (16 NCp bytes per row), so
// the later stages and the top stage read and write whole chunks. Butterfly records carry a and b in float64 (24 bytes); the
// float32 lanes round them once, exactly as the float32 kernels round sqrt(w0 / (w0 + w1)).
//
// Replaces, on the wide columns, what raht_fwd_quant_f64 / raht_dequant_inv_f64 compute (same arithmetic, same order), and on the
// other columns what raht_fwd_quant / raht_dequant_inv compute. Reference: python/RAHT.py:252-336, python/iRAHT.py:40-114,
// python/encode_3dgs.py:204,210,215,261,267-268,274.
#include "raht_common.h"
#include "raht_device.h"
#include "tile_engine.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

namespace raht {

// Profiling build only (-DRAHT_PHASE_CLOCKS, tools/phase_clocks_mx.py): thread 0 of the first stage-0 workgroups stamps the
// shader clock at the phase boundaries of its tile.
#ifdef RAHT_PHASE_CLOCKS
constexpr int MX_CLK_TILES = 4096, MX_CLK_SLOTS = 12;
__device__ unsigned long long g_phase_clk_mx[MX_CLK_TILES][MX_CLK_SLOTS];
#define MX_STAMP(k) do { if (IDENT && threadIdx.x == 0 && tile_id < MX_CLK_TILES) g_phase_clk_mx[tile_id][k] = __builtin_readcyclecounter(); } while (0)
#else
#define MX_STAMP(k) do { } while (0)
#endif

constexpr int MX_MAX_WIDE = 4;          // the wide channels are the row's first 16 bytes
constexpr int MX_PRE_ROWS = 12;         // survivor rows prefetched by the inverse before flags are known
constexpr int MX_THREADS = 512;
constexpr int MX_TOP_THREADS = 1024;
constexpr int MX_TOP_SLOTS = RAHT_TOP_MAX_ROWS / MX_TOP_THREADS;

struct StepTableMX {
    StepTable f;                        // float32 steps of every channel ((float)step, what raht_fwd_quant would be given)
    double w[MX_MAX_WIDE];              // float64 steps of the wide channels
};

// (must match the carve-up in tile_body_mx)
static size_t tile_lds_bytes_mx(int R, int NF, int nwide, bool ident)
{
    const size_t data = (size_t)R * NF * 16 + (((size_t)R * nwide * 8 + 15) & ~(size_t)15);   // float tile + wide tile (8 bytes per wide channel)
    const size_t meta = (size_t)R * (16 + 4 + (ident ? 0 : 4) + 4 + 1);        // a, b (float64) + operand slots; row id; Q position; flag
    const size_t surv = ((size_t)R * 2 + 15) & ~(size_t)15;
    return data + ((meta + 15) & ~(size_t)15) + 1024 + surv;                  // (the inverse's survivor prefetch shares the records' bytes)
}

// the wide parts of the workspaces a stage touches (a workspace row is stored as two dense arrays: the float places of every
// entry, then the wide places of every entry; TileArgs::in / out / wsn point at the float parts)
struct MxPtrs {
    const double *in_w;      // fwd, stages >= 1: wide part of ws_k (n_wide doubles per entry)
    double *out_w;           // inv, stages >= 1: wide part of ws_k
    double *wsn_w;           // wide part of ws_{k+1}
};

template <bool INV, bool IDENT, int SLOTS>
__device__ __forceinline__ void tile_body_mx(const TileArgs<float> &A, const MxPtrs &P, const StepTableMX &ST, const int64_t tile_id)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef RegChunk<float> V16;
    typedef RegChunk<double> W16;
    typedef RegChunk<int32_t> I16;
    const int R = A.R;
    const int tid0 = threadIdx.x;
    const int nthreads = blockDim.x, nwv = nthreads >> 6;
    const int nwide = A.nwide;                             // wide channels (1 .. 4): one double each per row of the wide tile
    const int lgw = nwide > 2 ? 2 : nwide - 1;             // log2(lanes per butterfly in the wide pass)
    const int Df = A.D;                                    // the float tile holds ALL D channels, laid out as in the float32 kernels
    const int Fp = A.Dp, NF = Fp >> 2;                     // float tile: row stride in floats, chunk places per row
    const int lg = A.lg, lr = 6 - A.lg;                    // 2^lg >= NF lanes per row
    const uint32_t NFm = ((1u << 20) + (uint32_t)NF - 1) / (uint32_t)NF;     // c / NF == (c * NFm) >> 20 for c < 2^15
    auto sync_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto wait_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- LDS carve-up (must match tile_lds_bytes_mx) ----
    size_t off = 0;
    float *ftile = (float *)smem; off += (size_t)R * Fp * 4;              // float32 channels, NF places per row
    double *wd = (double *)(smem + off); off += ((size_t)R * nwide * 8 + 15) & ~(size_t)15;   // wide channels: nwide doubles per row
    unsigned char *rec_base = smem + off;
    W16 *rec_ab = (W16 *)(smem + off); off += (size_t)R * 16;             // butterfly records: a, b in float64 ...
    uint32_t *rec_pj = (uint32_t *)(smem + off); off += (size_t)R * 4;    // ... and the two operand slots (partner | own << 16)
    // inverse: the survivor rows prefetched at kernel start wait in the records' bytes (records are written once they have moved
    // into their slots: P3b, a barrier, P3a) -- as many as fit, at most MX_PRE_ROWS
    const int pre_rows = min(MX_PRE_ROWS, (R * 20 - 8) / (Fp * 4 + nwide * 8));      // (- 8: the wide part arrives in 16-byte chunks)
    float *spre_f = (float *)rec_base;
    double *spre_w = (double *)(rec_base + (size_t)pre_rows * Fp * 4);
    int32_t *srow = (int32_t *)(smem + off); if (!IDENT) off += (size_t)R * 4;
    int32_t *sdst = (int32_t *)(smem + off); off += (size_t)R * 4;
    uint8_t *sflag = (uint8_t *)(smem + off); off += (size_t)R;
    off = (off + 15) & ~(size_t)15;
    uint32_t *hist = (uint32_t *)(smem + off);
    uint32_t *loff = hist + 64;
    uint32_t *cursor = hist + 128;
    uint32_t *scnt = hist + 196;
    off += 1024;
    uint16_t *ssurv = (uint16_t *)(smem + off);
    off += ((size_t)R * 2 + 15) & ~(size_t)15;

    TileMeta<SLOTS> M;
    load_tile_meta<float, IDENT, true, SLOTS>(A, tile_id, tid0, nthreads, M);

    // lane geometry of the row loops, as in the float32 kernels: lane c4 of a group of 2^lg works on float place fl = min(c4, NF - 1)
    // of one row (channels goff .. goff + 3; lanes past the last place shadow it: same reads, same writes). The lane of place 0
    // is the row's HEAD lane: the wide channels are the first n_wide <= 4 channels of its chunk, and in the write-backs it
    // substitutes their float64 results for what the float32 arithmetic made of them.
    auto lane_geom = [&](int tid, int &lane, int &wid, int &g, int &c4, int &fl, int &sp, int &goff, bool &head) {
        lane = tid & 63;
        wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        g = lane >> lg;
        c4 = lane & ((1 << lg) - 1);
        fl = min(c4, NF - 1);
        head = c4 == 0;
        sp = fl;
        goff = min(fl * 4, Df - 4);
    };
    float my_step[4], my_rcp[4];
    auto load_steps = [&](int place) {
        const int g0 = min(place * 4, Df - 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            my_step[i] = ST.f.v[ST.f.n == 1 ? 0 : g0 + i];
            my_rcp[i] = refined_rcp(my_step[i]);
        }
    };

    int tid = tid0;
    asm volatile("" : "+v"(tid));
    int lane, wid, g, c4, fl, sp, goff; bool head;
    lane_geom(tid, lane, wid, g, c4, fl, sp, goff, head);
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
    MX_STAMP(0);

    // LDS-direct transfers, lane-linear: instruction `it` of a transfer fills LDS bytes [1024 it, 1024 it + 1024) of its region.
    //   contiguous: row images (both parts of a workspace row are stored as separate, dense arrays)
    //   caller rows (C / Q: 4-byte elements either way): the float places from channels n_wide + 4 ch .., and the row's first 16
    //   bytes -- the raw wide channels, widened in place by widen_row once landed -- into the row's first wide place
    auto load_linear = [&](const float *dst, int chunks, const float *src) {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)dst;
        for (int it = wid; (it << 6) < chunks; it += nwv) {
            const int c = (it << 6) + lane;
            if (c < chunks) glds16<0>(src + (uint32_t)c * 4u, lds0 + ((uint32_t)it << 10));
        }
    };
    auto load_caller_rows = [&](int rows, auto src) {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)ftile;
        const int total = rows * NF;
        for (int it = wid; (it << 6) < total; it += nwv) {
            const int c = (it << 6) + lane;
            const int jr = (int)(((uint32_t)c * NFm) >> 20), ch = c - jr * NF;
            if (c < total) glds16<1>(src(jr, (uint32_t)min(ch * 4, Df - 4)), lds0 + ((uint32_t)it << 10));
        }
    };
    // Widening of the wide channels a caller row arrived with (the first n_wide elements of its first float chunk): float32 ->
    // float64 (forward), int32 * float64 step (inverse, encode_3dgs.py:261), into the wide tile. One ROW per thread, after the
    // barrier behind which every wave's rows have landed. (The float tile keeps those elements and carries them through its
    // float32 butterflies like any other channel: the write-backs drop what comes out of that.)
    auto widen_row = [&](int j, auto is_int) {
        double *row = wd + __mul24(j, nwide);
        const V16 raw = *(const V16 *)(ftile + __mul24(j, Fp));          // the row's first chunk, as it arrived
        double d[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (decltype(is_int)::value) d[i] = i < nwide ? (double)__float_as_int(raw.v[i]) * ST.w[i] : 0.0;
            else d[i] = i < nwide ? (double)raw.v[i] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) if (i < nwide) row[i] = d[i];
    };

    // ---- P0b. transfers whose addresses do not depend on the plan metadata ----
    if constexpr (!INV) {
        if constexpr (IDENT) {
            const uint32_t ldc = (uint32_t)A.ld_in;
            const float *src = A.in + e0 * (int64_t)ldc;
            load_caller_rows(nt, [&](int jr, uint32_t go) { return row_at(src, (uint32_t)jr, ldc, go); });
        } else {
            load_linear(ftile, nt * NF, A.in + e0 * (int64_t)Fp);
            load_linear((const float *)wd, (nt * nwide + 1) >> 1, (const float *)(P.in_w + e0 * (int64_t)nwide));      // (16-byte chunks: may read one double past the tile's rows -- the next tile's, or the array's slack)
        }
    } else {
        const int npre = A.last_stage ? 0 : (int)min(surv_cnt, (uint32_t)pre_rows);
        load_linear(spre_f, npre * NF, (const float *)A.wsn + (int64_t)surv_base * Fp);
        load_linear((const float *)spre_w, (npre * nwide + 1) >> 1, (const float *)(P.wsn_w + (int64_t)surv_base * nwide));
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        if (j < nt) { if (!IDENT) srow[j] = m_row[s]; if (INV) sdst[j] = m_pos[s]; }
    }
    MX_STAMP(1);
    sync_lds();                                                            // sync #1
    MX_STAMP(2);

    if constexpr (INV) {
        // every slot's quantized row (survivor slots are overwritten in P3b)
        load_caller_rows(nt, [&](int jr, uint32_t go) {
            return (const void *)row_far((const int32_t *)A.Q, (uint32_t)sdst[jr], (uint32_t)A.ldq, go); });
    }

    // ---- P1. which slots merge inside this tile; height histogram; survivor ranks ----
    bool m_merged[SLOTS];
    int m_rank[SLOTS];
    const uint64_t lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        m_merged[s] = false;
        bool surv = false;
        if (j < nt) {
            const int64_t r = m_row[s];
            m_merged[s] = (r > 0) && (m_lv[s] < A.top_level) && (r - m_wl[s] >= start_row) && (r + m_wr[s] <= end_row);
            surv = !m_merged[s];
            sflag[j] = m_merged[s] ? 1 : (A.last_stage ? 2 : 0);
            if (!INV) sdst[j] = m_pos[s] | ((m_merged[s] || A.last_stage) ? (int32_t)0x80000000 : 0);
            if (m_merged[s]) atomicAdd(&hist[m_ht[s]], 1u);
        }
        const uint64_t bal = __ballot(surv);
        m_rank[s] = __popcll(bal & lt);
        if (lane == 0 && s * nwv + wid < 32) scnt[s * nwv + wid] = (uint32_t)__popcll(bal);
    }
    sync_lds();                                                            // sync #2
    MX_STAMP(3);

    // ---- P2. round offsets (wave 0); survivor destinations ----
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
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        if (j < nt && !m_merged[s]) {
            uint32_t before = 0;
            for (int q = 0; q < s * nwv + wid; ++q) before += scnt[q];
            ssurv[before + (uint32_t)m_rank[s]] = (uint16_t)j;
        }
    }
    if constexpr (INV) wait_landed();                                      // this wave's Q rows (and survivor prefetch) are in LDS
    sync_lds();                                                            // sync #3 (inverse: every row has landed)
    MX_STAMP(4);
    if constexpr (INV) {
        // ---- P3b (first part): the prefetched survivor rows (images, from the stage above) move into their slots; the butterfly
        // records are then written over the bytes they waited in, hence the barrier
        if (!A.last_stage) {
            const uint32_t n_pre = min(surv_cnt, (uint32_t)pre_rows);
            if (c4 < NF) for (uint32_t it = wid; (it << lr) < n_pre; it += nwv) {
                const uint32_t qc = min((it << lr) + g, n_pre - 1);
                const V16 x = *(const V16 *)&spre_f[__mul24((int)qc, Fp) + fl * 4];
                *(V16 *)&ftile[__mul24((int)ssurv[qc], Fp) + fl * 4] = x;
            }
            for (uint32_t c = (uint32_t)tid; c < n_pre * (uint32_t)nwide; c += (uint32_t)nthreads) {
                const uint32_t qc = c / (uint32_t)nwide, i = c - qc * (uint32_t)nwide;
                wd[__mul24((int)ssurv[qc], nwide) + i] = spre_w[c];
            }
        }
        sync_lds();
        load_steps(fl);
        // roots finalised by a last TILE stage come straight from Q as well: dequantize them in place (no butterfly will)
        if (A.last_stage && c4 < NF) for (int it = wid; (it << lr) < nt; it += nwv) {
            const int j = (it << lr) + g;
            if (j < nt && sflag[j] == 2) {
                V16 *pr = (V16 *)&ftile[__mul24(j, Fp) + fl * 4];
                const I16 raw = *(const I16 *)pr;
                V16 x;
#pragma unroll
                for (int i = 0; i < 4; ++i)       // encode_3dgs.py:261 (the wide elements stay raw: widen_row reads them in this phase)
                    x.v[i] = (head && i < nwide) ? __int_as_float(raw.v[i]) : (float)raw.v[i] * my_step[i];
                *pr = x;
            }
        }
        // the wide channels of the rows finalised here (survivor slots are filled by P3b, with images)
        // (by the thread half a workgroup away from the row's own: tiles of <= 256 rows keep waves 0 .. 3 busy with P3a
        // below, the widening then runs next to it on the idle ones; sflag: 0 = survivor)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int j = ((tid + (nthreads >> 1)) & (nthreads - 1)) + s * nthreads;
            if (j < nt && (sflag[j] != 0)) widen_row(j, std::true_type());
        }
    }

    // ---- P3a. resolve every butterfly of this tile into a record, bucketed by height ----
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        if (j < nt && m_merged[s]) {
            const int64_t r = m_row[s];
            const int l = m_wl[s];
            int p;
            if (IDENT) {
                p = j - l;
            } else {
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
            W16 ab;
            ab.v[0] = sqrt(w0 / den);                     // RAHT.py:321-322 (float64; the float32 lanes round it once)
            ab.v[1] = sqrt(w1 / den);
            const uint32_t pos = atomicAdd(&cursor[m_ht[s]], 1u);
            rec_ab[pos] = ab;
            rec_pj[pos] = (uint32_t)p | ((uint32_t)j << 16);
        }
    }
    // ---- P3b (second part) ----
    if (INV && !A.last_stage) {
        // (second part: the survivors past the prefetched ones, straight from the workspace)
        if (c4 < NF) for (uint32_t it = wid; pre_rows + (it << lr) < surv_cnt; it += nwv) {
            const uint32_t qc = min(pre_rows + (it << lr) + g, surv_cnt - 1);
            const V16 x = ld_chunk<float>(row_at((const float *)A.wsn + (int64_t)surv_base * Fp, qc, (uint32_t)Fp, (uint32_t)(fl * 4)));
            *(V16 *)&ftile[__mul24((int)ssurv[qc], Fp) + fl * 4] = x;
        }
        for (uint32_t c = (uint32_t)(pre_rows * nwide) + (uint32_t)tid; c < surv_cnt * (uint32_t)nwide; c += (uint32_t)nthreads) {
            const uint32_t qc = c / (uint32_t)nwide, i = c - qc * (uint32_t)nwide;
            wd[__mul24((int)ssurv[qc], nwide) + i] = P.wsn_w[(int64_t)surv_base * nwide + c];
        }
    }
    if constexpr (!INV) {
        wait_landed();                                                     // this wave's rows are in LDS
        if constexpr (IDENT) {
            sync_lds();                                                    // every wave's
#pragma unroll
            for (int s = 0; s < SLOTS; ++s) {
                const int j = tid + s * nthreads;
                if (j < nt) widen_row(j, std::false_type());
            }
        }
        sync_lds();                                                        // sync #4
    } else {
        __syncthreads();
    }
    MX_STAMP(5);

    // ---- P4. butterflies, one round per height present ----
    // The float32 channels run exactly as in the float32 kernels: a lane group per butterfly (head and idle lanes shadow the
    // group's last float lane: same reads, same writes, no exec-mask juggling). The wide channels of a level are a SEPARATE, dense
    // pass over their own LDS array -- a lane per (butterfly, channel), 16 to 64 butterflies per wave instruction -- on other waves.
    // Levels that fit one wave instruction (most: the chain of small levels needs no workgroup barrier, a wave's LDS operations
    // execute in order) are walked by wave 0 (float) and wave 1 (wide) side by side. (First version: one row array with the wide
    // places in front, the wide lanes taking a float64 branch inside every float butterfly instruction: both branches issued for
    // every instruction of every wave, and every same-place access of many 256-byte rows a bank conflict: LDS busy twice as long
    // as in the float32 kernel, rocprofv3 SQ_LDS_BANK_CONFLICT 39 M against 11 M cycles per launch.)
    {
        const uint32_t stride = (uint32_t)(nwv << lr);
        const uint32_t gw = (uint32_t)lane >> lgw, cw = (uint32_t)min(lane & ((1 << lgw) - 1), nwide - 1);   // (three channels: the group's 4th lane shadows the 3rd)
        const uint32_t bw = 64u >> lgw;                        // wide butterflies per wave instruction
        const uint32_t cf = (uint32_t)fl * 4u;
        bool chained = false;
        const int loff_v = (int)loff[lane], hist_v = (int)hist[lane];
        uint64_t mask = __ballot(hist_v > 0);
        while (mask) {
            const int l = INV ? (63 - __clzll((long long)mask)) : (__ffsll((long long)mask) - 1);
            mask &= ~(1ull << l);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane(loff_v, l), cnt = (uint32_t)__builtin_amdgcn_readlane(hist_v, l);
            auto apply_f = [&](auto UC, uint32_t mb) {
                constexpr int U = decltype(UC)::value;
                W16 ab[U];
                uint32_t pj[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const uint32_t m = base + min(mb + u * stride + g, cnt - 1); pj[u] = rec_pj[m]; ab[u] = rec_ab[m]; }
                uint32_t ip[U], ij[U];
                V16 x0[U], x1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { ip[u] = __umul24(pj[u] & 0xffffu, (uint32_t)Fp) + cf; ij[u] = __umul24(pj[u] >> 16, (uint32_t)Fp) + cf; }
#pragma unroll
                for (int u = 0; u < U; ++u) { x0[u] = *(const V16 *)&ftile[ip[u]]; x1[u] = *(const V16 *)&ftile[ij[u]]; }
                if constexpr (INV) {                          // the high-pass operand is still the quantized integer (encode_3dgs.py:261)
#pragma unroll
                    for (int u = 0; u < U; ++u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) x1[u].v[i] = (float)__float_as_int(x1[u].v[i]) * my_step[i];
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const float ca = (float)ab[u].v[0], cb = (float)ab[u].v[1];
                    V16 lo, hi;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!INV) {                           // RAHT.py:331-332
                            lo.v[i] = ca * x0[u].v[i] + cb * x1[u].v[i];
                            hi.v[i] = ca * x1[u].v[i] - cb * x0[u].v[i];
                        } else {                              // iRAHT.py:108-109
                            lo.v[i] = ca * x0[u].v[i] - cb * x1[u].v[i];
                            hi.v[i] = cb * x0[u].v[i] + ca * x1[u].v[i];
                        }
                    }
                    if (u == 0 || mb + u * stride < cnt) { *(V16 *)&ftile[ip[u]] = lo; *(V16 *)&ftile[ij[u]] = hi; }
                }
            };
            auto pass_f = [&](auto UC) {
                constexpr int U = decltype(UC)::value;
                for (uint32_t mb = (uint32_t)(wid << lr); mb < cnt; mb += stride * U) apply_f(UC, mb);
            };
            // butterflies m0 + gw of this level, wide channels (lanes past the last one redo it in lockstep with its owner)
            auto apply_w = [&](uint32_t m0) {
                const uint32_t m = base + min(m0 + gw, cnt - 1);
                const uint32_t pj = rec_pj[m];
                const W16 ab = rec_ab[m];
                const uint32_t ip = __umul24(pj & 0xffffu, (uint32_t)nwide) + cw, ij = __umul24(pj >> 16, (uint32_t)nwide) + cw;
                const double d0 = wd[ip], d1 = wd[ij];
                const double ca = ab.v[0], cb = ab.v[1];
                double lo, hi;
                if (!INV) {                                   // RAHT.py:331-332
                    lo = ca * d0 + cb * d1;
                    hi = ca * d1 - cb * d0;
                } else {                                      // iRAHT.py:108-109
                    lo = ca * d0 - cb * d1;
                    hi = cb * d0 + ca * d1;
                }
                wd[ip] = lo; wd[ij] = hi;
            };
            if (cnt <= (1u << lr)) {
                if (wid == 0) pass_f(std::integral_constant<int, 1>());
                else if (wid == 1) { for (uint32_t m0 = 0; m0 < cnt; m0 += bw) apply_w(m0); }     // (narrow rows: a float instruction may hold more butterflies than a wide one)
                chained = true;
            } else {
                if (chained) { __syncthreads(); chained = false; }
                if (cnt <= stride) pass_f(std::integral_constant<int, 1>());
                else pass_f(std::integral_constant<int, TILE_ROUND_U>());
                for (uint32_t m0 = (uint32_t)(nwv - 1 - wid) * bw; m0 < cnt; m0 += (uint32_t)nwv * bw) apply_w(m0);   // from the last wave down
                __syncthreads();
            }
        }
        if (chained) __syncthreads();
    }

    MX_STAMP(6);
    // ---- P5. write back ----
    {
        int tid5 = tid0;
        asm volatile("" : "+v"(tid5));
        lane_geom(tid5, lane, wid, g, c4, fl, sp, goff, head);
    }
    const bool rowlane = c4 < NF;
    if constexpr (INV) {
        if constexpr (IDENT) {
            // stage 0 -> the caller's C rows [e0, e0 + nt). ONE store instruction writes a whole row: the float lanes their chunks,
            // the head lane the row's first 16 bytes -- the wide channels rounded to float32 and, behind them, the first
            // 4 - n_wide float channels once more (the same values their own lane stores). A row start written by a second,
            // narrower store instruction becomes a partial-line write of its own: the fused forward took 0.92 ms instead of
            // 0.31 ms that way (rows are 236 bytes: every line is shared by two rows)
            float *base = A.out + e0 * A.ld_out;
            // Two row instructions per trip. The head lanes alone read their rows' wide results (clamped indices: no branch per
            // channel, all reads of a trip in flight together); what follows is branch-free -- selects, then ONE store instruction
            // per row group, pinned behind an empty asm so that the compiler cannot sink it into a head / non-head diamond again
            // (it did: the rows' first 16 bytes left as a second, narrow store instruction, i.e. a partial-line write per row, and
            // every wide channel was its own LDS round trip)
            if (rowlane) for (int it = wid; (it << lr) < nt; it += 2 * nwv) {
                int j[2]; V16 x[2]; double w[2][4] = {};
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    j[u] = min(((it + u * nwv) << lr) + g, nt - 1);
                    x[u] = *(const V16 *)&ftile[__mul24(j[u], Fp) + sp * 4];
                }
                if (head) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) w[u][i] = wd[__mul24(j[u], nwide) + min(i, nwide - 1)];
                    }
                }
                asm volatile("" : "+v"(w[0][0]), "+v"(w[0][1]), "+v"(w[0][2]), "+v"(w[0][3]), "+v"(w[1][0]), "+v"(w[1][1]), "+v"(w[1][2]), "+v"(w[1][3]));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) { const float f = (float)w[u][i]; x[u].v[i] = (head && i < nwide) ? f : x[u].v[i]; }
                    asm volatile("" : "+v"(x[u].v[0]), "+v"(x[u].v[1]), "+v"(x[u].v[2]), "+v"(x[u].v[3]));
                    if (u == 0 || ((it + nwv) << lr) < nt)                  // (wave-uniform)
                        st_chunk<float, true>(row_at(base, (uint32_t)j[u], (uint32_t)A.ld_out, (uint32_t)goff), x[u]);
                }
            }
        } else {
            // stage k -> ws_k: both parts are contiguous runs of chunks
            float *bf = A.out + e0 * (int64_t)Fp;
            double *bw_ = P.out_w + e0 * (int64_t)nwide;
            for (int c = tid; c < nt * NF; c += nthreads) st_chunk<float>(bf + c * 4, *(const V16 *)&ftile[c * 4]);
            for (int c = tid; c < nt * nwide; c += nthreads) bw_[c] = wd[c];
        }
    } else {
        // survivors, compacted, to the next stage's workspace (row images, two dense arrays)
        if (!A.last_stage) {
            float *bf = A.wsn + (int64_t)surv_base * Fp;
            double *bw_ = P.wsn_w + (int64_t)surv_base * nwide;
            if (c4 < NF) for (uint32_t it = wid; (it << lr) < surv_cnt; it += nwv) {
                const uint32_t q = min((it << lr) + g, surv_cnt - 1);
                const V16 x = *(const V16 *)&ftile[__mul24((int)ssurv[q], Fp) + fl * 4];
                st_chunk<float>(row_at(bf, q, (uint32_t)Fp, (uint32_t)(fl * 4)), x);
            }
            for (uint32_t c = (uint32_t)tid; c < surv_cnt * (uint32_t)nwide; c += (uint32_t)nthreads) {
                const uint32_t q = c / (uint32_t)nwide, i = c - q * (uint32_t)nwide;
                bw_[c] = wd[__mul24((int)ssurv[q], nwide) + i];
            }
        }
        // rows finalised here, quantized to Q[inv_order[row]] (encode_3dgs.py:204,210,215).
        // (a) the wide channels: one ROW per thread -- the IEEE double division is ~40 instructions a channel, so it runs on
        //     whole waves of rows -- and the integers go back into the row's first wide place (a row finalised here is nobody's
        //     survivor)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int j = tid0 + s * nthreads;
            if (j < nt && ((uint32_t)sdst[j] >> 31)) {
                double *row = &wd[__mul24(j, nwide)];
                double d[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < nwide) d[i] = row[i];
                asm volatile("" ::: "memory");                // (the integers go over the doubles they were made of: every read is above)
                int32_t *qrow = (int32_t *)row;
#pragma unroll
                for (int i = 0; i < 4; ++i) if (i < nwide) qrow[i] = quantize_one_f64(d[i], ST.w[i]);
            }
        }
        sync_lds();
        MX_STAMP(7);
        // (b) ONE store instruction per row group writes whole rows of Q: the float lanes their quantized chunks, the head lane the
        //     row's first 16 bytes = the wide integers and, behind them, the first 4 - n_wide float channels quantized once more
        //     (same values as their own lane's). See the inverse's write-back for why.
        load_steps(sp);
        auto store_final = [&](auto fast_div) {
            if (rowlane) for (int it = wid; (it << lr) < nt; it += 2 * nwv) {
                int jc[2]; V16 x[2]; I16 qi[2] = {}; uint32_t dv[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    jc[u] = min(((it + u * nwv) << lr) + g, nt - 1);
                    x[u] = *(const V16 *)&ftile[__mul24(jc[u], Fp) + sp * 4];
                    dv[u] = (uint32_t)sdst[jc[u]];
                }
                if (head) {                                   // (only the head lanes; clamped indices: no branch per channel)
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int32_t *qrow = (const int32_t *)&wd[__mul24(jc[u], nwide)];
#pragma unroll
                        for (int i = 0; i < 4; ++i) qi[u].v[i] = qrow[min(i, nwide - 1)];
                    }
                }
                asm volatile("" : "+v"(x[0].v[0]), "+v"(x[1].v[0]), "+v"(qi[0].v[0]), "+v"(qi[0].v[1]), "+v"(qi[0].v[2]), "+v"(qi[0].v[3]),
                             "+v"(qi[1].v[0]), "+v"(qi[1].v[1]), "+v"(qi[1].v[2]), "+v"(qi[1].v[3]));
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    // branch-free up to ONE store instruction per row group (see the inverse's write-back)
                    I16 qv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int32_t q = quantize_one(x[u].v[i], my_step[i], my_rcp[i], decltype(fast_div)::value);
                        qv.v[i] = (head && i < nwide) ? qi[u].v[i] : q;
                    }
                    asm volatile("" : "+v"(qv.v[0]), "+v"(qv.v[1]), "+v"(qv.v[2]), "+v"(qv.v[3]));
                    if ((dv[u] >> 31) && (u == 0 || ((it + nwv) << lr) < nt))
                        st_chunk<int32_t, true>(row_far(A.Q, dv[u] & 0x7fffffffu, (uint32_t)A.ldq, (uint32_t)goff), qv);
                }
            }
        };
        if (ST.f.fast_div) store_final(std::true_type()); else store_final(std::false_type());
    }
    MX_STAMP(8);
}

template <bool INV, bool IDENT, int SLOTS>
__global__ __launch_bounds__(MX_THREADS, 6) void tile_kernel_mx(const TileArgs<float> A, const MxPtrs P, const StepTableMX ST)
{
    tile_body_mx<INV, IDENT, SLOTS>(A, P, ST, (int64_t)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// TOP stage, mixed: one workgroup per chunk place keeps that place of ALL entries in LDS (transform.hip: top_kernel);
// workgroups [0, NW2) run the float64 instantiation on the wide places, the others the float32 one.
// ------------------------------------------------------------------------------------------------
struct TopArgsMX {
    const float *in_rows;  int64_t ld_in;    // fwd, single-stage schedule: the caller's C rows (entry = row)
    const float *in_img; const double *in_img_w;   // fwd, later stage: workspace row images, entry order (float places; n_wide doubles)
    float *out_rows;       int64_t ld_out;   // inv, single-stage schedule: the caller's C rows
    float *out_img; double *out_img_w;       // inv, later stage: workspace row images
    int32_t *Q;            int64_t ldq;
    const uint32_t *e_pos;                   // entry -> position in Q
    const uint32_t *pj;                      // butterflies sorted by level: partner entry | own entry << 16
    const float *ab32;  const double *ab64;  // a, b per butterfly
    int n, n_merges, D, nwide, Fp;           // Fp: floats per row of the float part of an image
    const uint32_t *lev;
    int nlev, nbig;
    uint32_t small_start;
};

template <bool WIDE, bool INV>
__device__ __forceinline__ void top_body_mx(const TopArgsMX &A, const StepTableMX &ST, const int chunk)
{
    typedef typename std::conditional<WIDE, double, float>::type T;
    constexpr int VN = WIDE ? 2 : 4;
    typedef RegChunk<T> V16;
    extern __shared__ __align__(16) unsigned char smem[];
    V16 *tile = (V16 *)smem;
    __shared__ uint32_t s_lev[2 * 64];
    const int tid = threadIdx.x;
    const int nwide = A.nwide;
    const int goff = WIDE ? 2 * chunk : min(chunk * 4, A.D - 4);            // first channel of this chunk in the caller's rows
    // (float chunk 0 holds the wide channels too, as float32 ballast: it never writes them to Q / C -- the wide workgroups do)
    const int ioff = chunk * 4;                                              // a float chunk's place in a row image (floats)
    const int n = A.n, nm = A.n_merges;
    if (tid < 2 * A.nlev) s_lev[tid] = A.lev[tid];
    const T *ab = WIDE ? (const T *)A.ab64 : (const T *)A.ab32;
    const int n_small = nm - (int)A.small_start;
    uint32_t *s_pj = (uint32_t *)(smem + (size_t)n * 16);
    T *s_ab = (T *)(s_pj + ((n_small + 3) & ~3));
    for (int i = tid; i < n_small; i += MX_TOP_THREADS) {
        s_pj[i] = A.pj[A.small_start + i];
        s_ab[2 * i] = ab[2 * (A.small_start + i)];
        s_ab[2 * i + 1] = ab[2 * (A.small_start + i) + 1];
    }
    T my_step[VN];
    float my_rcp[VN];
    bool live[VN];                                         // wide: channel goff + i exists
#pragma unroll
    for (int i = 0; i < VN; ++i) {
        if constexpr (WIDE) { live[i] = goff + i < nwide; my_step[i] = live[i] ? ST.w[goff + i] : 1.0; my_rcp[i] = 1.0f; }
        else { live[i] = !(chunk == 0 && i < nwide); my_step[i] = ST.f.v[ST.f.n == 1 ? 0 : goff + i]; my_rcp[i] = refined_rcp(my_step[i]); }
    }
    uint32_t pj[MX_TOP_SLOTS];
    T ra[MX_TOP_SLOTS], rb[MX_TOP_SLOTS];
#pragma unroll
    for (int k = 0; k < MX_TOP_SLOTS; ++k) {
        const int idx = min(k * MX_TOP_THREADS + tid, max(nm - 1, 0));
        pj[k] = A.pj[idx]; ra[k] = ab[2 * idx]; rb[k] = ab[2 * idx + 1];
    }
    uint32_t m_dst[MX_TOP_SLOTS];
#pragma unroll
    for (int k = 0; k < MX_TOP_SLOTS; ++k) m_dst[k] = A.e_pos[min(k * MX_TOP_THREADS + tid, n - 1)];
    // the entries
#pragma unroll
    for (int k = 0; k < MX_TOP_SLOTS; ++k) {
        const int e = k * MX_TOP_THREADS + tid;
        if (e >= n) continue;
        V16 v;
        if constexpr (!INV) {
            if (A.in_img) {
                if constexpr (WIDE) {
#pragma unroll
                    for (int i = 0; i < VN; ++i) v.v[i] = live[i] ? A.in_img_w[(int64_t)e * nwide + goff + i] : 0.0;
                } else {
                    v = *(const V16 *)(A.in_img + (int64_t)e * A.Fp + ioff);
                }
            } else if constexpr (WIDE) {
#pragma unroll
                for (int i = 0; i < VN; ++i) v.v[i] = live[i] ? (double)A.in_rows[(int64_t)e * A.ld_in + goff + i] : 0.0;
            } else {
                v = ld_chunk<float>(A.in_rows + (int64_t)e * A.ld_in + goff);
            }
        } else {
            const int32_t *q = A.Q + (int64_t)m_dst[k] * A.ldq + goff;
            if constexpr (WIDE) {
#pragma unroll
                for (int i = 0; i < VN; ++i) v.v[i] = live[i] ? (double)q[i] * my_step[i] : 0.0;        // encode_3dgs.py:261
            } else {
                const RegChunk<int32_t> raw = ld_chunk<int32_t>(q);
#pragma unroll
                for (int i = 0; i < VN; ++i) v.v[i] = (float)raw.v[i] * my_step[i];
            }
        }
        tile[e] = v;
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
    auto big_levels = [&]() {
#pragma unroll 1
        for (int q = 0; q < A.nbig; ++q) {
            const int li = INV ? A.nbig - 1 - q : q;
            const uint32_t lo = s_lev[2 * li], hi = s_lev[2 * li + 1];
#pragma unroll
            for (int k = 0; k < MX_TOP_SLOTS; ++k) {
                const uint32_t idx = (uint32_t)(k * MX_TOP_THREADS + tid);
                if ((uint32_t)(k * MX_TOP_THREADS) < hi && (uint32_t)((k + 1) * MX_TOP_THREADS) > lo && idx >= lo && idx < hi)
                    butterfly(pj[k], ra[k], rb[k]);
            }
            __syncthreads();
        }
    };
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

#pragma unroll
    for (int k = 0; k < MX_TOP_SLOTS; ++k) {
        const int e = k * MX_TOP_THREADS + tid;
        if (e >= n) continue;
        const V16 v = tile[e];
        if constexpr (INV) {
            if (A.out_img) {
                if constexpr (WIDE) {
#pragma unroll
                    for (int i = 0; i < VN; ++i) if (live[i]) A.out_img_w[(int64_t)e * nwide + goff + i] = v.v[i];
                } else {
                    *(V16 *)(A.out_img + (int64_t)e * A.Fp + ioff) = v;
                }
            } else if constexpr (WIDE) {
#pragma unroll
                for (int i = 0; i < VN; ++i) if (live[i]) A.out_rows[(int64_t)e * A.ld_out + goff + i] = (float)v.v[i];
            } else if (chunk == 0) {
#pragma unroll
                for (int i = 0; i < VN; ++i) if (live[i]) A.out_rows[(int64_t)e * A.ld_out + goff + i] = v.v[i];
            } else {
                st_chunk<float>(A.out_rows + (int64_t)e * A.ld_out + goff, v);
            }
        } else {
            int32_t *q = A.Q + (int64_t)m_dst[k] * A.ldq + goff;
            if constexpr (WIDE) {
#pragma unroll
                for (int i = 0; i < VN; ++i) if (live[i]) q[i] = quantize_one_f64(v.v[i], my_step[i]);
            } else {
                RegChunk<int32_t> qv;
#pragma unroll
                for (int i = 0; i < VN; ++i) qv.v[i] = quantize_one(v.v[i], my_step[i], my_rcp[i], ST.f.fast_div);
                if (chunk == 0) {
#pragma unroll
                    for (int i = 0; i < VN; ++i) if (live[i]) q[i] = qv.v[i];
                } else {
                    st_chunk<int32_t>(q, qv);
                }
            }
        }
    }
}

template <bool INV>
__global__ __launch_bounds__(MX_TOP_THREADS) void top_kernel_mx(const TopArgsMX A, const StepTableMX ST)
{
    const int NW2 = (A.nwide + 1) >> 1;
    if ((int)blockIdx.x < NW2) top_body_mx<true, INV>(A, ST, (int)blockIdx.x);
    else top_body_mx<false, INV>(A, ST, (int)blockIdx.x - NW2);
}

// the wide columns of caller rows <-> a compact float64 matrix (fallback path only)
__global__ void mx_cols_to_f64_kernel(const float *__restrict__ C, int64_t ldc, int64_t N, int nwide, double *__restrict__ W)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * nwide) return;
    const int64_t i = e / nwide;
    W[e] = (double)C[i * ldc + (e - i * nwide)];
}
__global__ void mx_cols_from_f64_kernel(const double *__restrict__ W, int64_t N, int nwide, float *__restrict__ C, int64_t ldc)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * nwide) return;
    const int64_t i = e / nwide;
    C[i * ldc + (e - i * nwide)] = (float)W[e];
}

// ---- host side -------------------------------------------------------------------------------------------------------
struct MxGeom {
    int nwide = 0, NCp = 0, Dp = 0, lg = 0;     // NCp: chunk places per row (wide + float); Dp: floats per row of the float tile
    int R0 = 0, R1 = 0, Rf = 0;
};

// Chunk places and tile rows for (D, n_wide). false: the mixed tile kernels do not cover this shape (fallback).
static bool mx_geometry(const raht_plan *p, int D, int nwide, MxGeom &g)
{
    // D - 4 >= n_wide: the row's LAST 16-byte chunk (channels [D - 4, D), which overlaps its neighbour when D % 4 != 0) must not
    // reach into the wide channels, whose float32 copies are ballast that only the first chunk's lane knows to replace
    if (D - 4 < nwide || D > 64 + MX_MAX_WIDE) return false;
    g.nwide = nwide;
    const int NF = (D + 3) / 4;                           // float places: ALL D channels, the float32 kernels' row layout
    g.NCp = (nwide + 1) / 2 + NF;
    g.Dp = NF * 4;
    g.lg = 0;
    while ((1 << g.lg) < NF) ++g.lg;
    int r1 = 0, dc1 = 0;
    pick_tail_geometry(p, 4, D, 512, &r1, &dc1, &g.Rf);
    const size_t budget = (size_t)42 * 1280;              // three workgroups per CU
    auto fit = [&](int hi, bool ident, size_t cap) {
        for (int R = hi; R >= 64; --R) if (tile_lds_bytes_mx(R, NF, nwide, ident) <= cap) return R;
        return 0;
    };
    g.R0 = p->tile_rows_override > 0 ? fit(std::min(p->tile_rows_override, TILE_MAX_SLOTS * MX_THREADS), true, (size_t)128 * 1280)
                                     : fit(512, true, budget);
    if (p->tile_rows_override > 0 && p->tile_rows_override < 64) g.R0 = 0;
    if (g.R0 == 0) return false;
    g.R1 = p->tail_rows_override > 0 ? fit(std::min(p->tail_rows_override, TILE_MAX_SLOTS * MX_THREADS), false, (size_t)128 * 1280)
                                     : fit(g.R0, false, budget);
    if (p->tail_rows_override > 0 && p->tail_rows_override < 64) g.R1 = 0;
    return g.R1 != 0;
}

static void fill_steps_mx(StepTableMX &t, const double *steps, int n_steps, int nwide)
{
    float f[MAX_STEP_CH];
    for (int c = 0; c < n_steps; ++c) f[c] = (float)steps[c];
    fill_step_table(t.f, f, n_steps);
    for (int i = 0; i < MX_MAX_WIDE; ++i) t.w[i] = i < nwide ? steps[n_steps == 1 ? 0 : i] : 1.0;
}

template <bool INV, bool IDENT, int SLOTS>
static int launch_tile_mx_one(const TileArgs<float> &A, const MxPtrs &P, const StepTableMX &st, unsigned n_tiles, size_t lds, hipStream_t s)
{
    static PerDeviceOnce attr;
    if (attr.first(current_device()))
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_mx<INV, IDENT, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((tile_kernel_mx<INV, IDENT, SLOTS>), dim3(n_tiles), dim3(MX_THREADS), lds, s, A, P, st);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

struct MxIO {
    const float *C_in = nullptr; float *C_out = nullptr; int64_t ldc = 0;
    int32_t *Q = nullptr; int64_t ldq = 0;
};

template <bool INV>
static int launch_stage_mx(const raht_plan *p, const Schedule &sc, int k, const MxIO &io, int D, const MxGeom &g,
                           const StepTableMX &stp, hipStream_t s)
{
    const Stage &st = sc.stages[(size_t)k];
    const int K = (int)sc.stages.size();
    // a stage's workspace: the float places of every entry, then the wide places of every entry
    float *ws_k = (k >= 1) ? (float *)stage_ws(st, INV) : nullptr;
    float *ws_n = (k + 1 < K) ? (float *)stage_ws(sc.stages[(size_t)k + 1], INV) : nullptr;
    double *ws_k_w = ws_k ? (double *)(ws_k + (size_t)st.n_entries * g.Dp) : nullptr;
    double *ws_n_w = ws_n ? (double *)(ws_n + (size_t)sc.stages[(size_t)k + 1].n_entries * g.Dp) : nullptr;
    if (k >= 1 && !ws_k) { set_error("mixed stage %d: missing stage workspace", k); return RAHT_ERR_INVALID; }
    if (st.is_top) {
        TopArgsMX A;
        A.in_rows = nullptr; A.ld_in = 0; A.in_img = nullptr; A.in_img_w = nullptr;
        A.out_rows = nullptr; A.ld_out = 0; A.out_img = nullptr; A.out_img_w = nullptr;
        if (!INV) { if (k == 0) { A.in_rows = io.C_in; A.ld_in = io.ldc; } else { A.in_img = ws_k; A.in_img_w = ws_k_w; } }
        else { if (k == 0) { A.out_rows = io.C_out; A.ld_out = io.ldc; } else { A.out_img = ws_k; A.out_img_w = ws_k_w; } }
        A.Q = io.Q; A.ldq = io.ldq;
        A.e_pos = st.rows ? st.e_pos : p->inv_order;
        A.pj = st.t_pj; A.ab32 = st.t_ab32; A.ab64 = st.t_ab64;
        A.n = (int)st.n_entries; A.n_merges = (int)st.n_merges; A.D = D; A.nwide = g.nwide; A.Fp = g.Dp;
        A.lev = st.t_lev; A.nlev = st.t_nlev; A.nbig = st.t_nbig; A.small_start = st.t_small_start;
        if (!A.e_pos || !A.pj || !A.ab32 || !A.ab64 || !A.lev || !A.Q) { set_error("mixed top stage: missing plan arrays"); return RAHT_ERR_INVALID; }
        const size_t n_small = st.n_merges - st.t_small_start;
        const size_t lds = (size_t)st.n_entries * 16 + ((n_small + 3) & ~(size_t)3) * 4 + n_small * 2 * sizeof(double);
        static PerDeviceOnce attr;
        if (attr.first(current_device()))
            RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)top_kernel_mx<INV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024));
        hipLaunchKernelGGL((top_kernel_mx<INV>), dim3((unsigned)g.NCp), dim3(MX_TOP_THREADS), lds, s, A, stp);
        RAHT_HIP_CHECK(hipGetLastError());
        return RAHT_OK;
    }
    TileArgs<float> A;
    A.rows = st.rows; A.surv_off = st.surv_off; A.n_entries = st.n_entries; A.N = p->N; A.R = st.tile_rows;
    A.D = D; A.Dc = D; A.Dp = g.Dp; A.lg = g.lg; A.nwide = g.nwide;
    A.last_stage = (k == K - 1) ? 1 : 0;
    A.wsum = p->wsum;
    if (st.rows) { A.lvl = st.e_lvl; A.wl = st.e_wl; A.wr = st.e_wr; A.inv_order = st.e_pos; }
    else { A.lvl = p->lvl; A.wl = p->wl; A.wr = p->wr; A.inv_order = p->inv_order; }
    A.ht = st.e_ht;
    A.Q = io.Q; A.ldq = io.ldq;
    A.top_level = p->top_level; A.root_buf = nullptr; A.dbg = 0; A.ref = nullptr; A.ld_ref = 0; A.sq_part = nullptr;
    A.ld_ws = g.Dp; A.wsn = ws_n;
    A.fin = nullptr; A.ld_fin = 0;
    if (!INV) { A.in = (k == 0) ? io.C_in : ws_k; A.ld_in = (k == 0) ? io.ldc : g.Dp; A.out = nullptr; A.ld_out = 0; }
    else { A.in = nullptr; A.ld_in = 0; A.out = (k == 0) ? io.C_out : ws_k; A.ld_out = (k == 0) ? io.ldc : g.Dp; }
    {
        // every pointer the kernel dereferences for this (direction, stage) must be there before the launch
        const char *bad = nullptr;
        if (!A.lvl || !A.wl || !A.wr || !A.ht) bad = "plan arrays";
        else if (!A.last_stage && (!A.wsn || !A.surv_off)) bad = "survivor workspace of a non-final stage";
        else if (!A.Q || !A.inv_order) bad = "Q / inv_order";
        else if (!INV && !A.in) bad = "forward input";
        else if (INV && !A.out) bad = "inverse output";
        else if ((st.rows == nullptr) != (k == 0)) bad = "stage order";
        else if (st.tile_rows < 1 || st.tile_rows > TILE_MAX_SLOTS * MX_THREADS ||
                 (int64_t)st.tile_rows * std::max<int64_t>(io.ldc, g.Dp) * 4 >= ((int64_t)1 << 31)) bad = "tile geometry (32-bit row offsets)";
        if (bad) { set_error("mixed tile stage %d (%s): missing %s", k, INV ? "inverse" : "forward", bad); return RAHT_ERR_INVALID; }
    }
    MxPtrs P;
    P.in_w = (!INV && k >= 1) ? ws_k_w : nullptr;
    P.out_w = (INV && k >= 1) ? ws_k_w : nullptr;
    P.wsn_w = ws_n_w;
    const size_t lds = tile_lds_bytes_mx(st.tile_rows, g.Dp / 4, g.nwide, st.rows == nullptr);
    const bool one = st.tile_rows <= MX_THREADS;
    const unsigned nt = (unsigned)st.n_tiles;
    if (k == 0 && p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_before, s));
    int rc;
    if (k == 0) rc = one ? launch_tile_mx_one<INV, true, 1>(A, P, stp, nt, lds, s) : launch_tile_mx_one<INV, true, 2>(A, P, stp, nt, lds, s);
    else rc = one ? launch_tile_mx_one<INV, false, 1>(A, P, stp, nt, lds, s) : launch_tile_mx_one<INV, false, 2>(A, P, stp, nt, lds, s);
    if (k == 0 && p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_after, s));
    return rc;
}

static int mx_check_args(const raht_plan *p, const void *a, const void *b, int D, int64_t lda, int64_t ldb, const double *steps,
                         int n_steps, int n_wide, const char *what)
{
    if (!p || !a || !b || D < 1 || lda < D || ldb < D) { set_error("%s: bad argument", what); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, what));
    if (n_wide < 1 || n_wide > MX_MAX_WIDE || n_wide > D) { set_error("%s: n_wide must be 1..%d (and <= D)", what, MX_MAX_WIDE); return RAHT_ERR_INVALID; }
    if (!steps || !(n_steps == 1 || n_steps == D)) { set_error("%s: n_steps must be 1 or D", what); return RAHT_ERR_INVALID; }
    if (n_steps > MAX_STEP_CH) { set_error("%s: per-channel steps support D <= %d", what, MAX_STEP_CH); return RAHT_ERR_UNSUPPORTED; }
    for (int c = 0; c < n_steps; ++c)
        if (!(steps[c] > 0.0) || !((float)steps[c] > 0.0f)) { set_error("%s: step[%d] must be > 0 (also as float32)", what, c); return RAHT_ERR_INVALID; }
    if (p->row_map || p->root_buf || p->top_level < 64) { set_error("%s: not available for row-mapped or truncated plans", what); return RAHT_ERR_UNSUPPORTED; }
    return RAHT_OK;
}

// schedule for the mixed tile kernels, or *sc_out = nullptr when this (plan, D, n_wide) takes the fallback
static int mx_setup(raht_plan *p, int D, int n_wide, int64_t max_ld, hipStream_t s, Schedule **sc_out, MxGeom &g)
{
    *sc_out = nullptr;
    if (p->engine == RAHT_ENGINE_LEVEL || max_ld > ((int64_t)1 << 18)) return RAHT_OK;
    if (!mx_geometry(p, D, n_wide, g)) return RAHT_OK;
    Schedule *sc = nullptr;
    RAHT_RET(get_schedule(p, g.R0, g.R1, g.Rf, s, &sc));
    if (!sc->valid) return RAHT_OK;
    RAHT_RET(ensure_workspace(sc, (size_t)g.Dp * 4 + (size_t)g.nwide * 8 + 8, p->split_ws));      // float chunks + n_wide doubles (+ slack: the wide part is fetched in 16-byte chunks)
    *sc_out = sc;
    return RAHT_OK;
}

static int fwd_quant_mixed_impl(const raht_plan *cp, const float *C, int64_t ldc, int D, const double *steps, int n_steps, int n_wide,
                                int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    RAHT_RET(mx_check_args(p, C, Q, D, ldc, ldq, steps, n_steps, n_wide, "raht_fwd_quant_mixed"));
    Schedule *sc = nullptr;
    MxGeom g;
    RAHT_RET(mx_setup(p, D, n_wide, std::max(ldc, ldq), s, &sc, g));
    if (!sc) {
        // shapes / plans the mixed tile kernels do not cover (level engine, D - n_wide < 4, very wide rows): the float32 path for
        // every channel, then the wide columns once more in float64 through a compact N x n_wide matrix
        float f[MAX_STEP_CH];
        for (int c = 0; c < n_steps; ++c) f[c] = (float)steps[c];
        RAHT_RET(raht_fwd_quant(p, C, ldc, D, f, n_steps, Q, ldq, stream));
        Scratch tmp(sizeof(double) * 2 * (size_t)p->N * (size_t)n_wide, s);
        if (!tmp.ok()) return RAHT_ERR_NOMEM;
        double *W = tmp.as<double>(), *TW = W + (size_t)p->N * n_wide;
        hipLaunchKernelGGL(mx_cols_to_f64_kernel, dim3((unsigned)ceil_div(p->N * n_wide, 256)), dim3(256), 0, s, C, ldc, p->N, n_wide, W);
        RAHT_HIP_CHECK(hipGetLastError());
        RAHT_RET(raht_fwd_f64(p, W, n_wide, n_wide, TW, n_wide, nullptr, stream));
        double ws[MX_MAX_WIDE];
        for (int i = 0; i < n_wide; ++i) ws[i] = steps[n_steps == 1 ? 0 : i];
        return raht_quant_reorder_f64(p, TW, n_wide, n_wide, ws, n_wide, Q, ldq, stream);
    }
    StepTableMX stp;
    fill_steps_mx(stp, steps, n_steps, n_wide);
    MxIO io;
    io.C_in = C; io.ldc = ldc; io.Q = Q; io.ldq = ldq;
    const int K = (int)sc->stages.size();
    for (int k = 0; k < K; ++k) RAHT_RET((launch_stage_mx<false>(p, *sc, k, io, D, g, stp, s)));
    return RAHT_OK;
}

static int dequant_inv_mixed_impl(const raht_plan *cp, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps, int n_wide,
                                  float *C, int64_t ldc, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    hipStream_t s = (hipStream_t)stream;
    RAHT_RET(mx_check_args(p, Q, C, D, ldq, ldc, steps, n_steps, n_wide, "raht_dequant_inv_mixed"));
    Schedule *sc = nullptr;
    MxGeom g;
    RAHT_RET(mx_setup(p, D, n_wide, std::max(ldc, ldq), s, &sc, g));
    if (!sc) {
        float f[MAX_STEP_CH];
        for (int c = 0; c < n_steps; ++c) f[c] = (float)steps[c];
        RAHT_RET(raht_dequant_inv(p, Q, ldq, D, f, n_steps, C, ldc, stream));
        Scratch tmp(sizeof(double) * 2 * (size_t)p->N * (size_t)n_wide, s);
        if (!tmp.ok()) return RAHT_ERR_NOMEM;
        double *W = tmp.as<double>(), *TW = W + (size_t)p->N * n_wide;
        double ws[MX_MAX_WIDE];
        for (int i = 0; i < n_wide; ++i) ws[i] = steps[n_steps == 1 ? 0 : i];
        RAHT_RET(raht_dequant_unreorder_f64(p, Q, ldq, n_wide, ws, n_wide, TW, n_wide, stream));
        RAHT_RET(raht_inv_f64(p, TW, n_wide, n_wide, W, n_wide, stream));
        hipLaunchKernelGGL(mx_cols_from_f64_kernel, dim3((unsigned)ceil_div(p->N * n_wide, 256)), dim3(256), 0, s, W, p->N, n_wide, C, ldc);
        RAHT_HIP_CHECK(hipGetLastError());
        return RAHT_OK;
    }
    StepTableMX stp;
    fill_steps_mx(stp, steps, n_steps, n_wide);
    MxIO io;
    io.C_out = C; io.ldc = ldc; io.Q = const_cast<int32_t *>(Q); io.ldq = ldq;
    const int K = (int)sc->stages.size();
    for (int k = K - 1; k >= 0; --k) RAHT_RET((launch_stage_mx<true>(p, *sc, k, io, D, g, stp, s)));
    return RAHT_OK;
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_fwd_quant_mixed(const raht_plan *plan, const float *C, int64_t ldc, int D, const double *steps, int n_steps, int n_wide,
                         int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    return guarded("raht_fwd_quant_mixed", [&]() { return fwd_quant_mixed_impl(plan, C, ldc, D, steps, n_steps, n_wide, Q, ldq, stream); });
}

int raht_dequant_inv_mixed(const raht_plan *plan, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps, int n_wide,
                           float *C, int64_t ldc, raht_stream_t stream)
{
    return guarded("raht_dequant_inv_mixed", [&]() { return dequant_inv_mixed_impl(plan, Q, ldq, D, steps, n_steps, n_wide, C, ldc, stream); });
}

#ifdef RAHT_PHASE_CLOCKS
int raht_debug_read_phase_clocks_mx(unsigned long long *dst, int n_tiles)
{
    RAHT_HIP_CHECK(hipDeviceSynchronize());
    RAHT_HIP_CHECK(hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_phase_clk_mx), sizeof(unsigned long long) * MX_CLK_SLOTS * (size_t)std::min(n_tiles, MX_CLK_TILES)));
    return MX_CLK_SLOTS;
}
#endif

/* Tile rows and stage sizes the mixed kernels use for (D, n_wide): tile_rows = 0 when this shape takes the two-pass fallback. */
int raht_plan_mixed_stats(raht_plan *plan, int D, int n_wide, int *tile_rows, int *n_stages, int64_t *rows_per_stage, int max_stages)
{
    return guarded("raht_plan_mixed_stats", [&]() -> int {
        if (!plan || !tile_rows || !n_stages || D < 1 || n_wide < 1 || n_wide > MX_MAX_WIDE) { set_error("raht_plan_mixed_stats: bad argument"); return RAHT_ERR_INVALID; }
        RAHT_RET(check_plan_device(plan, "raht_plan_mixed_stats"));
        *tile_rows = 0; *n_stages = 0;
        if (plan->row_map || plan->root_buf || plan->top_level < 64) return RAHT_OK;
        Schedule *sc = nullptr;
        MxGeom g;
        RAHT_RET(mx_setup(plan, D, n_wide, D, nullptr, &sc, g));
        if (!sc) return RAHT_OK;
        *tile_rows = g.R0;
        *n_stages = (int)sc->stages.size();
        for (int k = 0; k < *n_stages && k < max_stages && rows_per_stage; ++k) rows_per_stage[k] = sc->stages[(size_t)k].n_entries;
        return RAHT_OK;
    });
}

}  // extern "C"
