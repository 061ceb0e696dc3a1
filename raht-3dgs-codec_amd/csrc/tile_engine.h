// tile_engine.h -- device-side pieces shared by the tile kernels of transform.hip (float32 / float64) and
// transform_mx.hip (mixed precision): kernel arguments, per-tile plan metadata, row addressing, LDS-direct loads.
#pragma once
#include "raht_common.h"
#include "raht_device.h"

#include <type_traits>

namespace raht {

template <typename T>
struct TileArgs {
    const T *in;         // fwd: this stage's entries, entry order (stage 0: C; k >= 1: ws_k)
    int64_t ld_in;
    T *fin;              // row-indexed coefficient matrix T (fwd: output, inv: input)
    int64_t ld_fin;
    T *wsn;              // ws_{k+1}: survivors of this stage, entry order of stage k+1 (fwd out / inv in)
    T *out;              // inv: this stage's entries, entry order (stage 0: C; k >= 1: ws_k)
    int64_t ld_out;
    int64_t ld_ws;       // row stride of the workspaces (= D)
    int32_t *Q;          // fused quantization: row-permuted integer coefficients (fwd out / inv in)
    int64_t ldq;
    const uint32_t *inv_order;
    const uint32_t *rows;      // active rows of this stage (nullptr: identity, stage 0)
    const uint32_t *surv_off;  // [n_tiles + 1] first survivor index per tile (nullptr on the last stage)
    int64_t n_entries;
    int64_t N;
    int R;
    int D;
    int Dc;              // channels per chunk (blockIdx.y selects the chunk)
    int Dp;              // LDS row stride: Dc rounded up to a whole 16-byte chunk
    int lg;              // log2 of the lanes per row (2^lg >= chunks per row)
    int last_stage;      // this is the top stage of the schedule
    int dbg;             // profiling ablations, -DRAHT_ABLATE builds only (raht_debug_run_stage): 1 = skip
                         // butterflies, 2 = skip merge resolution too (pure staged copy). The product build
                         // compiles TILE_DBG to the constant 0: its kernels carry no ablation branches.
    const uint8_t *lvl;
    const uint8_t *ht;   // entry-ordered: height of the entry's butterfly inside its tile (Stage::e_ht); keys the rounds
    const int32_t *wl;
    const int32_t *wr;
    const int64_t *wsum;
    int top_level;       // butterflies at levels >= top_level are left to the caller (sharded scenes)
    T *root_buf;         // optional compact (n_roots x D) buffer: forward writes the rows still carrying
                         // a low-pass value there, inverse reads them from there (nullptr: T / Q rows)
    int nwide;           // mixed precision (transform_mx.hip): the first nwide channels are carried as float64; 0 elsewhere
    // raht_dequant_inv_sqdiff (stage 0 of the fused inverse only): the matrix the reconstruction is compared with, and where this
    // tile's per-chunk-element sums of squared differences go ([n_tiles x 4 ceil(D / 4)] float64); out may then be nullptr
    const T *ref; int64_t ld_ref; double *sq_part;
};

// One butterfly, resolved: LDS element offsets of the partner (low-pass) row and of the own
// (high-pass) row, plus the two coefficients.
template <typename T> struct MRec;
template <> struct __align__(16) MRec<float> { uint32_t po; uint32_t jo; float a; float b; };
template <> struct __align__(8) MRec<double> { uint32_t po; uint32_t jo; double a; double b; };

constexpr int TILE_MAX_SLOTS = 2;      // slots per thread (template SLOTS = 1 or 2): R <= SLOTS * blockDim
#ifndef RAHT_ROUND_U
#define RAHT_ROUND_U 2
#endif
constexpr int TILE_ROUND_U = RAHT_ROUND_U;        // butterflies in flight per lane group in a round
constexpr int TILE_PRE_ROWS = 12;      // survivor rows prefetched by the inverse before flags are known

// Plan metadata of one tile, held in registers (slot j = tid + s * blockDim). The persistent tile
// loop loads the NEXT tile's metadata while the current tile's butterflies run, so no tile waits on
// an HBM round trip before it can issue its own data loads.
template <int SLOTS>
struct TileMeta {
    int32_t row[SLOTS], wl[SLOTS], wr[SLOTS], pos[SLOTS];
    int lv[SLOTS], ht[SLOTS];
    int64_t start_row, end_row;
    uint32_t surv_raw;                 // lane l holds surv_off[t + (l & 1)]: a lane-dependent load, so that hipcc
                                       // does not scalarise it on the spot (s_waitcnt + v_readfirstlane right
                                       // after the load would also drain every older load, i.e. the prefetch)
};

template <typename T, bool IDENT, bool QM, int SLOTS>
__device__ __forceinline__ void load_tile_meta(const TileArgs<T> &A, int64_t t, int tid, int nthreads, TileMeta<SLOTS> &M)
{
    const int R = A.R;
    const int64_t e0 = t * R;
    const int nt = (int)min((int64_t)R, A.n_entries - e0);
    M.start_row = IDENT ? e0 : (int64_t)A.rows[e0];
    M.end_row = (e0 + R < A.n_entries) ? (IDENT ? e0 + R : (int64_t)A.rows[e0 + R]) : A.N;
    M.surv_raw = 0;
    if (A.surv_off) M.surv_raw = A.surv_off[t + (tid & 1)];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int j = tid + s * nthreads;
        M.row[s] = 0; M.wl[s] = 0; M.wr[s] = 0; M.lv[s] = 0; M.ht[s] = 0; M.pos[s] = 0;
        if (j < nt) {
            // A.wl / wr / lvl / inv_order are ENTRY-ordered for this stage (stage 0: entry = row,
            // the plan arrays themselves; later stages: per-stage gathered copies), so all five
            // loads are contiguous and independent of each other
            const int64_t r = IDENT ? e0 + j : (int64_t)A.rows[e0 + j];
            M.row[s] = (int32_t)r;
            M.wl[s] = A.wl[e0 + j];
            M.wr[s] = A.wr[e0 + j];
            M.lv[s] = A.lvl[e0 + j];
            M.ht[s] = A.ht[e0 + j];
            M.pos[s] = QM ? (int32_t)A.inv_order[e0 + j] : (int32_t)r;    // where the final coefficient lives
        }
    }
}

// raht_fwd_quant_multi: one forward pass, several quantizations (the drivers quantize ONE coefficient matrix at nine steps,
// python/encode_3dgs.py:28,199-217): k scalar steps, k output matrices
constexpr int MULTI_Q_MAX = 12;
struct MultiQ {
    int k;
    int fast_div;                       // every step within [2^-100, 2^100] (raht_device.h: quantize_one)
    float step[MULTI_Q_MAX];
    int32_t *Q[MULTI_Q_MAX];
};

// Row addressing. The kernel is bound by vector-instruction issue, and a 64 x 64-bit row * stride product
// per 16-byte chunk (3 quarter-rate multiplies + 5 more instructions) was a tenth of it.
//  row_at:  rows of the tile being processed: a wave-uniform base (scalar registers) plus a 32-bit byte
//           offset per lane, one v_mad_u32_u24 and a shift (tile_setup guarantees R * ld * 8 < 2^31);
//  row_far: rows anywhere in a matrix (scatter / gather through the plan's permutations): the 32 x 32 -> 64
//           bit product is ONE v_mad_u64_u32.
template <typename P>
__device__ __forceinline__ P *row_at(P *uniform_base, uint32_t j, uint32_t ld, uint32_t goff)
{
    typedef typename std::conditional<std::is_const<P>::value, const char, char>::type Byte;
    return (P *)((Byte *)uniform_base + (uint32_t)((__umul24(j, ld) + goff) * (uint32_t)sizeof(P)));
}
template <typename P>
__device__ __forceinline__ P *row_far(P *base, uint32_t row, uint32_t ld, uint32_t goff)
{
    return base + ((uint64_t)row * ld + goff);
}

// LDS-direct load: 16 bytes per lane from the lane's global address to LDS byte (lds_base + 16 * lane), no register in
// between (global_load_lds_dwordx4; M0 = the wave's LDS base, inactive lanes write nothing, the global address needs
// element alignment only: tools/probes/glds_probe.hip). Issued through inline assembly ON PURPOSE: hipcc does not know
// that LDS is being written, so it neither drains the load at the next LDS access it cannot tell apart from the
// destination (every one, with one dynamic LDS block) nor counts it -- its own s_waitcnt vmcnt(n) are then merely
// stricter than needed (the counter retires in order). The kernel waits for the data itself: tile_kernel, sync #3.
template <int MODE = 0>      // 0: default policy; 1: nontemporal (matrices touched once: C, T, Q); 2: agent-coherent (sc1: rows another
                             // workgroup of the SAME launch has just written through -- forward chaining, tile_kernel_chain)
__device__ __forceinline__ void glds16(const void *g, uint32_t lds_base)
{
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
    if constexpr (MODE == 1) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(g), "s"(b) : "memory");
    else if constexpr (MODE == 2) asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" :: "v"(g), "s"(b) : "memory");
    else asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(b) : "memory");
}

}  // namespace raht
