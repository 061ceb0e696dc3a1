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
static size_t tile_lds_bytes_mx(int R, int NCp, bool ident)
{
    const size_t data = (size_t)R * NCp * 16;
    const size_t meta = (size_t)R * (sizeof(MRec<double>) + (ident ? 0 : 4) + 4 + 1);
    const size_t surv = ((size_t)R * 2 + 15) & ~(size_t)15;
    return data + ((meta + 15) & ~(size_t)15) + 1024 + surv + (size_t)MX_PRE_ROWS * NCp * 16;
}

template <bool INV, bool IDENT, int SLOTS>
__device__ __forceinline__ void tile_body_mx(const TileArgs<float> &A, const StepTableMX &ST, const int64_t tile_id)
{
    extern __shared__ __align__(16) unsigned char smem[];
    typedef RegChunk<float> V16;
    typedef RegChunk<double> W16;
    typedef RegChunk<int32_t> I16;
    const int R = A.R;
    const int tid0 = threadIdx.x;
    const int nthreads = blockDim.x, nwv = nthreads >> 6;
    const int nwide = A.nwide, NW2 = (nwide + 1) >> 1;     // wide channels, wide chunk places
    const int Df = A.D - nwide;                            // float32 channels (>= 4)
    const int Dp = A.Dp, NCp = Dp >> 2;                    // LDS row stride in floats, chunk places per row
    const int lg = A.lg, lr = 6 - A.lg;
    const uint32_t NCm = ((1u << 20) + (uint32_t)NCp - 1) / (uint32_t)NCp;   // c / NCp == (c * NCm) >> 20 for c < 2^15
    auto sync_lds = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    auto wait_landed = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };

    // ---- LDS carve-up (must match tile_lds_bytes_mx) ----
    size_t off = (size_t)R * Dp * 4;
    float *tile = (float *)smem;
    MRec<double> *mrec = (MRec<double> *)(smem + off); off += (size_t)R * sizeof(MRec<double>);
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
    float *spre = (float *)(smem + off);

    TileMeta<SLOTS> M;
    load_tile_meta<float, IDENT, true, SLOTS>(A, tile_id, tid0, nthreads, M);

    // lane geometry: lane c4 of a group of 2^lg owns chunk place c4 of the group's row. Places [0, NW2) are wide (two doubles:
    // channels 2 c4, 2 c4 + 1), places [NW2, NCp) float (channels goff .. goff + 3 of the global row)
    auto lane_geom = [&](int tid, int &lane, int &wid, int &g, int &c4c, int &coff, int &goff, bool &active, bool &wide) {
        lane = tid & 63;
        wid = __builtin_amdgcn_readfirstlane(tid >> 6);
        g = lane >> lg;
        const int c4 = lane & ((1 << lg) - 1);
        active = c4 < NCp;
        c4c = min(c4, NCp - 1);
        coff = c4c * 4;
        wide = c4c < NW2;
        goff = nwide + min(max(c4c - NW2, 0) * 4, Df - 4);
    };
    float my_step[4], my_rcp[4];
    auto load_steps = [&](int ln) {
        const int c4c = min(ln & ((1 << lg) - 1), NCp - 1);
        const int g0 = nwide + min(max(c4c - NW2, 0) * 4, Df - 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            my_step[i] = ST.f.v[ST.f.n == 1 ? 0 : g0 + i];
            my_rcp[i] = refined_rcp(my_step[i]);
        }
    };

    int tid = tid0;
    asm volatile("" : "+v"(tid));
    int lane, wid, g, c4c, coff, goff; bool active, wide;
    lane_geom(tid, lane, wid, g, c4c, coff, goff, active, wide);
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

    // Row transfers, lane-linear over the nt * NCp chunk places of the tile (place c = 64 * instruction + lane -> LDS byte 16 c).
    //   image rows (workspaces): every place from the row's own chunk;
    //   caller rows (C / Q, element type = 4 bytes either way): float places from channels goff.., place 0 the row's first 16
    //   bytes (raw wide channels, widened in place by widen_rows once landed), other wide places nothing
    auto load_image_rows = [&](const float *dst, int rows, const float *src) {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)dst;
        const int total = rows * NCp;
        for (int it = wid; (it << 6) < total; it += nwv) {
            const int c = (it << 6) + lane;
            if (c < total) glds16<0>(src + (uint32_t)c * 4u, lds0 + ((uint32_t)it << 10));
        }
    };
    auto load_caller_rows = [&](const float *dst, int rows, auto src) {
        const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)dst;
        const int total = rows * NCp;
        for (int it = wid; (it << 6) < total; it += nwv) {
            const int c = (it << 6) + lane;
            const int jr = (int)(((uint32_t)c * NCm) >> 20), ch = c - jr * NCp;
            if (c < total && (ch >= NW2 || ch == 0))
                glds16<1>(src(jr, (uint32_t)(ch >= NW2 ? nwide + min((ch - NW2) * 4, Df - 4) : 0)), lds0 + ((uint32_t)it << 10));
        }
    };
    // the lane that loaded a row's raw first 16 bytes widens them: float32 -> float64 (forward), int32 * float64 step (inverse,
    // encode_3dgs.py:261); its own load has landed (s_waitcnt vmcnt(0) in front), nobody else touches the wide places before
    // the barrier that follows
    auto widen_rows = [&](float *dst, int rows, auto is_int) {
        const int total = rows * NCp;
        for (int it = wid; (it << 6) < total; it += nwv) {
            const int c = (it << 6) + lane;
            const int jr = (int)(((uint32_t)c * NCm) >> 20), ch = c - jr * NCp;
            if (c < total && ch == 0) {
                float *row = dst + __mul24(jr, Dp);
                double d[4];
                if constexpr (decltype(is_int)::value) {
                    const I16 raw = *(const I16 *)row;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = i < nwide ? (double)raw.v[i] * ST.w[i] : 0.0;
                } else {
                    const V16 raw = *(const V16 *)row;
#pragma unroll
                    for (int i = 0; i < 4; ++i) d[i] = i < nwide ? (double)raw.v[i] : 0.0;
                }
                W16 w0; w0.v[0] = d[0]; w0.v[1] = d[1];
                *(W16 *)row = w0;
                if (NW2 > 1) { W16 w1; w1.v[0] = d[2]; w1.v[1] = d[3]; *(W16 *)(row + 4) = w1; }
            }
        }
    };

    // ---- P0b. transfers whose addresses do not depend on the plan metadata ----
    if constexpr (!INV) {
        if constexpr (IDENT) {
            const uint32_t ldc = (uint32_t)A.ld_in;
            const float *src = A.in + e0 * (int64_t)ldc;
            load_caller_rows(tile, nt, [&](int jr, uint32_t go) { return row_at(src, (uint32_t)jr, ldc, go); });
        } else {
            load_image_rows(tile, nt, A.in + e0 * (int64_t)Dp);
        }
    } else {
        const int npre = A.last_stage ? 0 : (int)min(surv_cnt, (uint32_t)MX_PRE_ROWS);
        load_image_rows(spre, npre, (const float *)A.wsn + (int64_t)surv_base * Dp);
    }
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        if (j < nt) { if (!IDENT) srow[j] = m_row[s]; if (INV) sdst[j] = m_pos[s]; }
    }
    sync_lds();                                                            // sync #1

    if constexpr (INV) {
        // every slot's quantized row (survivor slots are overwritten in P3b)
        load_caller_rows(tile, nt, [&](int jr, uint32_t go) {
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
    if constexpr (INV) {
        wait_landed();                                                     // this wave's Q rows (and survivor prefetch) are in LDS
        widen_rows(tile, nt, std::true_type());
    }
    sync_lds();                                                            // sync #3
    if constexpr (INV) {
        load_steps(lane);
        // roots finalised by a last TILE stage come straight from Q as well: dequantize them in place (no butterfly will)
        if (A.last_stage && active && !wide) for (int it = wid; (it << lr) < nt; it += nwv) {
            const int j = (it << lr) + g;
            if (j < nt && sflag[j] == 2) {
                V16 *pr = (V16 *)&tile[__mul24(j, Dp) + coff];
                const I16 raw = *(const I16 *)pr;
                V16 x;
#pragma unroll
                for (int i = 0; i < 4; ++i) x.v[i] = (float)raw.v[i] * my_step[i];            // encode_3dgs.py:261
                *pr = x;
            }
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
            MRec<double> rec;
            rec.po = (uint32_t)__mul24((int)p, Dp);
            rec.jo = (uint32_t)__mul24((int)j, Dp);
            rec.a = sqrt(w0 / den);                       // RAHT.py:321-322 (float64; the float32 lanes round it once)
            rec.b = sqrt(w1 / den);
            const uint32_t pos = atomicAdd(&cursor[m_ht[s]], 1u);
            mrec[pos] = rec;
        }
    }
    // ---- P3b. inverse: the survivors' low-pass rows (images, from the stage above) into their slots ----
    if (INV && !A.last_stage) {
        const uint32_t n_pre = min(surv_cnt, (uint32_t)MX_PRE_ROWS);
        if (active) for (uint32_t it = wid; (it << lr) < n_pre; it += nwv) {
            const uint32_t qc = min((it << lr) + g, n_pre - 1);
            const V16 x = *(const V16 *)&spre[__mul24((int)qc, Dp) + coff];
            *(V16 *)&tile[__mul24((int)ssurv[qc], Dp) + coff] = x;
        }
        if (active) for (uint32_t it = wid; MX_PRE_ROWS + (it << lr) < surv_cnt; it += nwv) {
            const uint32_t qc = min(MX_PRE_ROWS + (it << lr) + g, surv_cnt - 1);
            const V16 x = ld_chunk<float>(row_at((const float *)A.wsn + (int64_t)surv_base * Dp, qc, (uint32_t)Dp, (uint32_t)coff));
            *(V16 *)&tile[__mul24((int)ssurv[qc], Dp) + coff] = x;
        }
    }
    if constexpr (!INV) {
        wait_landed();                                                     // this wave's rows are in LDS
        if constexpr (IDENT) widen_rows(tile, nt, std::false_type());
        sync_lds();                                                        // sync #4
    } else {
        __syncthreads();
    }

    // ---- P4. butterflies, one round per height present; a lane group handles one butterfly ----
    {
        const uint32_t stride = (uint32_t)(nwv << lr);
        bool chained = false;
        const int loff_v = (int)loff[lane], hist_v = (int)hist[lane];
        uint64_t mask = __ballot(hist_v > 0);
        while (mask) {
            const int l = INV ? (63 - __clzll((long long)mask)) : (__ffsll((long long)mask) - 1);
            mask &= ~(1ull << l);
            const uint32_t base = (uint32_t)__builtin_amdgcn_readlane(loff_v, l), cnt = (uint32_t)__builtin_amdgcn_readlane(hist_v, l);
            auto apply = [&](auto UC, uint32_t mb) {
                constexpr int U = decltype(UC)::value;
                MRec<double> r[U];
#pragma unroll
                for (int u = 0; u < U; ++u) r[u] = mrec[base + min(mb + u * stride + g, cnt - 1)];
                uint32_t ip[U], ij[U];
                V16 x0[U], x1[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { ip[u] = r[u].po + coff; ij[u] = r[u].jo + coff; }
#pragma unroll
                for (int u = 0; u < U; ++u) { x0[u] = *(const V16 *)&tile[ip[u]]; x1[u] = *(const V16 *)&tile[ij[u]]; }
                if (wide) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        W16 d0, d1, lo, hi;
                        __builtin_memcpy(&d0, &x0[u], 16);
                        __builtin_memcpy(&d1, &x1[u], 16);
                        const double ca = r[u].a, cb = r[u].b;
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            if (!INV) {                       // RAHT.py:331-332
                                lo.v[i] = ca * d0.v[i] + cb * d1.v[i];
                                hi.v[i] = ca * d1.v[i] - cb * d0.v[i];
                            } else {                          // iRAHT.py:108-109
                                lo.v[i] = ca * d0.v[i] - cb * d1.v[i];
                                hi.v[i] = cb * d0.v[i] + ca * d1.v[i];
                            }
                        }
                        if (u == 0 || mb + u * stride < cnt) { *(W16 *)&tile[ip[u]] = lo; *(W16 *)&tile[ij[u]] = hi; }
                    }
                } else {
                    if constexpr (INV) {                      // the high-pass operand is still the quantized integer (encode_3dgs.py:261)
#pragma unroll
                        for (int u = 0; u < U; ++u) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) x1[u].v[i] = (float)__float_as_int(x1[u].v[i]) * my_step[i];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float ca = (float)r[u].a, cb = (float)r[u].b;
                        V16 lo, hi;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            if (!INV) {
                                lo.v[i] = ca * x0[u].v[i] + cb * x1[u].v[i];
                                hi.v[i] = ca * x1[u].v[i] - cb * x0[u].v[i];
                            } else {
                                lo.v[i] = ca * x0[u].v[i] - cb * x1[u].v[i];
                                hi.v[i] = cb * x0[u].v[i] + ca * x1[u].v[i];
                            }
                        }
                        if (u == 0 || mb + u * stride < cnt) { *(V16 *)&tile[ip[u]] = lo; *(V16 *)&tile[ij[u]] = hi; }
                    }
                }
            };
            auto pass = [&](auto UC) {
                constexpr int U = decltype(UC)::value;
                for (uint32_t mb = (uint32_t)(wid << lr); mb < cnt; mb += stride * U) apply(UC, mb);
            };
            if (cnt <= (1u << lr)) {
                if (wid == 0) pass(std::integral_constant<int, 1>());     // fits one wave instruction: wave 0 alone, no barrier
                chained = true;
            } else {
                if (chained) { __syncthreads(); chained = false; }
                if (cnt <= stride) pass(std::integral_constant<int, 1>());
                else pass(std::integral_constant<int, TILE_ROUND_U>());
                __syncthreads();
            }
        }
        if (chained) __syncthreads();
    }

    // ---- P5. write back ----
    {
        int tid5 = tid0;
        asm volatile("" : "+v"(tid5));
        lane_geom(tid5, lane, wid, g, c4c, coff, goff, active, wide);
    }
    if constexpr (INV) {
        if constexpr (IDENT) {
            // stage 0 -> the caller's C rows [e0, e0 + nt): float chunks as they are, the wide lanes round their two channels
            float *base = A.out + e0 * A.ld_out;
            if (active) for (int it = wid; (it << lr) < nt; it += nwv) {
                const int j = min((it << lr) + g, nt - 1);
                const V16 x = *(const V16 *)&tile[__mul24(j, Dp) + coff];
                if (!wide) {
                    st_chunk<float, true>(row_at(base, (uint32_t)j, (uint32_t)A.ld_out, (uint32_t)goff), x);
                } else {
                    W16 d;
                    __builtin_memcpy(&d, &x, 16);
                    float *pr = row_at(base, (uint32_t)j, (uint32_t)A.ld_out, (uint32_t)(2 * c4c));
#pragma unroll
                    for (int i = 0; i < 2; ++i) if (2 * c4c + i < nwide) __builtin_nontemporal_store((float)d.v[i], pr + i);
                }
            }
        } else {
            // stage k -> ws_k, row images
            float *base = A.out + e0 * (int64_t)Dp;
            if (active) for (int it = wid; (it << lr) < nt; it += nwv) {
                const int j = min((it << lr) + g, nt - 1);
                const V16 x = *(const V16 *)&tile[__mul24(j, Dp) + coff];
                st_chunk<float>(row_at(base, (uint32_t)j, (uint32_t)Dp, (uint32_t)coff), x);
            }
        }
    } else {
        // survivors, compacted, to the next stage's workspace (row images)
        if (!A.last_stage) {
            float *base = A.wsn + (int64_t)surv_base * Dp;
            if (active) for (uint32_t it = wid; (it << lr) < surv_cnt; it += nwv) {
                const uint32_t q = min((it << lr) + g, surv_cnt - 1);
                const V16 x = *(const V16 *)&tile[__mul24((int)ssurv[q], Dp) + coff];
                st_chunk<float>(row_at(base, q, (uint32_t)Dp, (uint32_t)coff), x);
            }
        }
        // rows finalised here, quantized to Q[inv_order[row]] (encode_3dgs.py:204,210,215): the float32 channels ...
        load_steps(lane);
        auto store_final = [&](auto fast_div) {
            if (active && !wide) for (int it = wid; (it << lr) < nt; it += nwv) {
                const int jc = min((it << lr) + g, nt - 1);
                V16 x = *(const V16 *)&tile[__mul24(jc, Dp) + coff];
                const uint32_t dv = (uint32_t)sdst[jc];
                asm volatile("" : "+v"(x.v[0]));
                if (dv >> 31) {
                    I16 qv;
#pragma unroll
                    for (int i = 0; i < 4; ++i) qv.v[i] = quantize_one(x.v[i], my_step[i], my_rcp[i], decltype(fast_div)::value);
                    st_chunk<int32_t, true>(row_far(A.Q, dv & 0x7fffffffu, (uint32_t)A.ldq, (uint32_t)goff), qv);
                }
            }
        };
        if (ST.f.fast_div) store_final(std::true_type()); else store_final(std::false_type());
        // ... and the wide channels: one ROW per thread (the IEEE double division is ~40 instructions a channel: run on whole
        // waves of rows instead of on the few wide lanes of every row instruction)
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) {
            const int j = tid0 + s * nthreads;
            if (j < nt) {
                const uint32_t dv = (uint32_t)sdst[j];
                if (dv >> 31) {
                    const float *row = &tile[__mul24(j, Dp)];
                    double d[4];
                    { const W16 w0 = *(const W16 *)row; d[0] = w0.v[0]; d[1] = w0.v[1]; }
                    d[2] = 0.0; d[3] = 0.0;
                    if (NW2 > 1) { const W16 w1 = *(const W16 *)(row + 4); d[2] = w1.v[0]; d[3] = w1.v[1]; }
                    int32_t *qrow = row_far(A.Q, dv & 0x7fffffffu, (uint32_t)A.ldq, 0u);
#pragma unroll
                    for (int i = 0; i < 4; ++i) if (i < nwide) __builtin_nontemporal_store(quantize_one_f64(d[i], ST.w[i]), qrow + i);
                }
            }
        }
    }
}

template <bool INV, bool IDENT, int SLOTS>
__global__ __launch_bounds__(MX_THREADS, 6) void tile_kernel_mx(const TileArgs<float> A, const StepTableMX ST)
{
    tile_body_mx<INV, IDENT, SLOTS>(A, ST, (int64_t)blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// TOP stage, mixed: one workgroup per chunk place keeps that place of ALL entries in LDS (transform.hip: top_kernel);
// workgroups [0, NW2) run the float64 instantiation on the wide places, the others the float32 one.
// ------------------------------------------------------------------------------------------------
struct TopArgsMX {
    const float *in_rows;  int64_t ld_in;    // fwd, single-stage schedule: the caller's C rows (entry = row)
    const float *in_img;                     // fwd, later stage: workspace row images, entry order
    float *out_rows;       int64_t ld_out;   // inv, single-stage schedule: the caller's C rows
    float *out_img;                          // inv, later stage: workspace row images
    int32_t *Q;            int64_t ldq;
    const uint32_t *e_pos;                   // entry -> position in Q
    const uint32_t *pj;                      // butterflies sorted by level: partner entry | own entry << 16
    const float *ab32;  const double *ab64;  // a, b per butterfly
    int n, n_merges, D, nwide, Dp;
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
    const int nwide = A.nwide, NW2 = (nwide + 1) >> 1, Df = A.D - nwide;
    const int goff = WIDE ? 2 * chunk : nwide + min(chunk * 4, Df - 4);     // first channel of this chunk in the caller's rows
    const int ioff = WIDE ? chunk * 4 : (NW2 + chunk) * 4;                    // its place in a row image (floats)
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
        else { live[i] = true; my_step[i] = ST.f.v[ST.f.n == 1 ? 0 : goff + i]; my_rcp[i] = refined_rcp(my_step[i]); }
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
                v = *(const V16 *)(A.in_img + (int64_t)e * A.Dp + ioff);
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
                *(V16 *)(A.out_img + (int64_t)e * A.Dp + ioff) = v;
            } else if constexpr (WIDE) {
#pragma unroll
                for (int i = 0; i < VN; ++i) if (live[i]) A.out_rows[(int64_t)e * A.ld_out + goff + i] = (float)v.v[i];
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
                st_chunk<int32_t>(q, qv);
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
    int nwide = 0, NCp = 0, Dp = 0, lg = 0;
    int R0 = 0, R1 = 0, Rf = 0;
};

// Chunk places and tile rows for (D, n_wide). false: the mixed tile kernels do not cover this shape (fallback).
static bool mx_geometry(const raht_plan *p, int D, int nwide, MxGeom &g)
{
    const int Df = D - nwide;
    if (Df < 4 || D > 64 + MX_MAX_WIDE) return false;
    g.nwide = nwide;
    g.NCp = (nwide + 1) / 2 + (Df + 3) / 4;
    if (g.NCp > 32) return false;
    g.Dp = g.NCp * 4;
    g.lg = 0;
    while ((1 << g.lg) < g.NCp) ++g.lg;
    int r1 = 0, dc1 = 0;
    pick_tail_geometry(p, 4, D, 512, &r1, &dc1, &g.Rf);
    const size_t budget = (size_t)42 * 1280;              // three workgroups per CU
    auto fit = [&](int hi, bool ident, size_t cap) {
        for (int R = hi; R >= 64; R -= 4) if (tile_lds_bytes_mx(R, g.NCp, ident) <= cap) return R;
        return 0;
    };
    g.R0 = p->tile_rows_override > 0 ? fit(std::min(p->tile_rows_override, TILE_MAX_SLOTS * MX_THREADS) / 4 * 4 + 0, true, (size_t)128 * 1280)
                                     : fit(512, true, budget);
    if (p->tile_rows_override > 0 && p->tile_rows_override < 64) g.R0 = 0;
    if (g.R0 == 0) return false;
    g.R1 = p->tail_rows_override > 0 ? fit(std::min(p->tail_rows_override, TILE_MAX_SLOTS * MX_THREADS) / 4 * 4, false, (size_t)128 * 1280)
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
static int launch_tile_mx_one(const TileArgs<float> &A, const StepTableMX &st, unsigned n_tiles, size_t lds, hipStream_t s)
{
    static PerDeviceOnce attr;
    if (attr.first(current_device()))
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel_mx<INV, IDENT, SLOTS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((tile_kernel_mx<INV, IDENT, SLOTS>), dim3(n_tiles), dim3(MX_THREADS), lds, s, A, st);
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
    float *ws_k = (k >= 1) ? (float *)st.ws : nullptr;
    float *ws_n = (k + 1 < K) ? (float *)sc.stages[(size_t)k + 1].ws : nullptr;
    if (k >= 1 && !ws_k) { set_error("mixed stage %d: missing stage workspace", k); return RAHT_ERR_INVALID; }
    if (st.is_top) {
        TopArgsMX A;
        A.in_rows = nullptr; A.ld_in = 0; A.in_img = nullptr; A.out_rows = nullptr; A.ld_out = 0; A.out_img = nullptr;
        if (!INV) { if (k == 0) { A.in_rows = io.C_in; A.ld_in = io.ldc; } else A.in_img = ws_k; }
        else { if (k == 0) { A.out_rows = io.C_out; A.ld_out = io.ldc; } else A.out_img = ws_k; }
        A.Q = io.Q; A.ldq = io.ldq;
        A.e_pos = st.rows ? st.e_pos : p->inv_order;
        A.pj = st.t_pj; A.ab32 = st.t_ab32; A.ab64 = st.t_ab64;
        A.n = (int)st.n_entries; A.n_merges = (int)st.n_merges; A.D = D; A.nwide = g.nwide; A.Dp = g.Dp;
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
    A.top_level = p->top_level; A.root_buf = nullptr; A.dbg = 0;
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
    const size_t lds = tile_lds_bytes_mx(st.tile_rows, g.NCp, st.rows == nullptr);
    const bool one = st.tile_rows <= MX_THREADS;
    const unsigned nt = (unsigned)st.n_tiles;
    if (k == 0 && p->ev_before) RAHT_HIP_CHECK(hipEventRecord(p->ev_before, s));
    int rc;
    if (k == 0) rc = one ? launch_tile_mx_one<INV, true, 1>(A, stp, nt, lds, s) : launch_tile_mx_one<INV, true, 2>(A, stp, nt, lds, s);
    else rc = one ? launch_tile_mx_one<INV, false, 1>(A, stp, nt, lds, s) : launch_tile_mx_one<INV, false, 2>(A, stp, nt, lds, s);
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
    RAHT_RET(ensure_workspace(sc, (size_t)g.NCp * 16));
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
