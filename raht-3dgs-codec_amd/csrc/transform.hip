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
//    LDS with 16-byte coalesced loads, performs EVERY butterfly whose subtree lies inside the run
//    (levels ascending, one barrier per level present), and writes the run back once. Rows whose
//    subtree crosses the run boundary (~3-4 % at R = 256) are the active rows of the next, much
//    smaller stage, which gathers them by index. HBM traffic ~1.04x the ideal (read C once, write T
//    once). Stage membership is a pure function of the plan, so it is precomputed (plan.hip).
//
// Bandwidth-bound integer/fp32 work: no MFMA by design (3 flops per 4 bytes).
#include "raht_common.h"

namespace raht {

size_t tile_lds_bytes(int R, int elem_size, int Dc);

template <typename T> struct Vec16;
template <> struct Vec16<float> { typedef float4 type; static constexpr int n = 4; };
template <> struct Vec16<double> { typedef double2 type; static constexpr int n = 2; };

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
                                                         const int64_t *__restrict__ wsum, int lp_shift)
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
            T *r0 = data + (i - l) * ld;
            T *r1 = data + i * ld;
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
// ------------------------------------------------------------------------------------------------
template <typename T>
struct TileArgs {
    const T *src;        // pristine input (fwd: C, inv: T)
    int64_t ld_src;
    T *dst;              // output, also the carrier of intermediate low-pass rows between stages
    int64_t ld_dst;
    const uint32_t *rows;  // active rows of this stage (nullptr: identity, stage 0)
    int64_t n_entries;
    int64_t N;
    int R;
    int D;
    int Dc;              // channels per chunk (blockIdx.y selects the chunk)
    int lp_shift;        // log2 of the lane-group size (>= Dc, power of two, <= 64)
    int last_stage;      // this is the top stage of the schedule
    int vec_ok;          // stage 0 fast path allowed (contiguous rows, 16-byte aligned bases)
    const uint8_t *lvl;
    const int32_t *wl;
    const int32_t *wr;
    const int64_t *wsum;
};

// IDENT = true is stage 0 (the tile is rows [e0, e0 + R) of the matrix itself: the HBM-heavy
// launch); IDENT = false are the gathered later stages. Separate instantiations keep the two
// apart in rocprof kernel statistics.
template <typename T, bool INV, bool IDENT>
__global__ __launch_bounds__(256) void tile_kernel(const TileArgs<T> A)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int R = A.R;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int nthreads = blockDim.x, nw = nthreads >> 6;
    const int c_base = blockIdx.y * A.Dc;
    const int Dc = min(A.Dc, A.D - c_base);

    // ---- LDS carve-up (must match tile_lds_bytes) ----
    size_t off = ((size_t)R * A.Dc * sizeof(T) + 15) & ~(size_t)15;
    T *tile = (T *)smem;
    T *sa = (T *)(smem + off); off += (size_t)R * sizeof(T);
    T *sb = (T *)(smem + off); off += (size_t)R * sizeof(T);
    int32_t *srow = (int32_t *)(smem + off); off += (size_t)R * 4;
    int32_t *swl = (int32_t *)(smem + off); off += (size_t)R * 4;
    int32_t *swr = (int32_t *)(smem + off); off += (size_t)R * 4;
    uint16_t *spart = (uint16_t *)(smem + off); off += (size_t)R * 2;
    uint16_t *smlist = (uint16_t *)(smem + off); off += (size_t)R * 2;
    uint8_t *slv = (uint8_t *)(smem + off); off += (size_t)R;
    off = (off + 15) & ~(size_t)15;
    uint32_t *hist = (uint32_t *)(smem + off);            // [64]
    uint32_t *loff = hist + 64;                           // [64]
    uint32_t *cursor = hist + 128;                        // [64]
    uint32_t *lmask = hist + 192;                         // [2]

    const int64_t e0 = (int64_t)blockIdx.x * R;
    const int nt = (int)min((int64_t)R, A.n_entries - e0);
    constexpr bool ident = IDENT;
    const int64_t start_row = ident ? e0 : (int64_t)A.rows[e0];
    const int64_t end_row = (e0 + R < A.n_entries) ? (ident ? e0 + R : (int64_t)A.rows[e0 + R]) : A.N;
    const bool fast = ident && A.vec_ok;                  // contiguous 16-byte path

    // ---- 1. stage-0 bulk load: the tile is one contiguous span of src ----
    if (fast) {
        typedef typename Vec16<T>::type V;
        constexpr int VN = Vec16<T>::n;
        const T *gsrc = A.src + e0 * A.ld_src;
        const int nelem = nt * Dc;
        const int nvec = nelem / VN;
        const V *g4 = (const V *)gsrc;
        V *l4 = (V *)tile;
        int v = tid;
        for (; v + 3 * nthreads < nvec; v += 4 * nthreads) {      // 4 loads in flight per lane
            const V x0 = g4[v], x1 = g4[v + nthreads], x2 = g4[v + 2 * nthreads], x3 = g4[v + 3 * nthreads];
            l4[v] = x0; l4[v + nthreads] = x1; l4[v + 2 * nthreads] = x2; l4[v + 3 * nthreads] = x3;
        }
        for (; v < nvec; v += nthreads) l4[v] = g4[v];
        for (int e = nvec * VN + tid; e < nelem; e += nthreads) tile[e] = gsrc[e];
    }

    // ---- 2. per-slot metadata ----
    if (tid < 64) hist[tid] = 0;
    for (int j = tid; j < nt; j += nthreads) {
        const int64_t r = ident ? e0 + j : (int64_t)A.rows[e0 + j];
        srow[j] = (int32_t)r;
        swl[j] = A.wl[r];
        swr[j] = A.wr[r];
        slv[j] = A.lvl[r];
    }
    __syncthreads();

    // ---- 3. which slots merge here, their partner slot and butterfly coefficients ----
    for (int j = tid; j < nt; j += nthreads) {
        const int64_t r = srow[j];
        const int l = swl[j], rr = swr[j];
        const bool merged = (r > 0) && (r - l >= start_row) && (r + rr <= end_row);
        uint16_t part = 0xffffu;
        if (merged) {
            int p;
            if (ident) {
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
            part = (uint16_t)p;
            double w0, w1;
            pair_weights(r, l, rr, A.wsum, w0, w1);
            const double den = w0 + w1;
            sa[j] = (T)sqrt(w0 / den);
            sb[j] = (T)sqrt(w1 / den);
            atomicAdd(&hist[slv[j]], 1u);
        }
        spart[j] = part;
    }
    __syncthreads();

    // ---- 4. bucket the merging slots by level (counting sort in LDS) ----
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
        const uint64_t m = __ballot(c > 0);
        if (lane == 0) { lmask[0] = (uint32_t)m; lmask[1] = (uint32_t)(m >> 32); }
    }
    __syncthreads();
    for (int j = tid; j < nt; j += nthreads) {
        if (spart[j] != 0xffffu) {
            const uint32_t pos = atomicAdd(&cursor[slv[j]], 1u);
            smlist[pos] = (uint16_t)j;
        }
    }

    // ---- 5. row-wise loads (gathered stages, strided / unaligned stage 0, inverse patches) ----
    // forward : stage 0 reads src, later stages read the low-pass rows carried in dst
    // inverse : rows finalised by THIS stage still hold pristine coefficients in src; rows that
    //           survive this stage were already rewritten by the stages above it (unless this is
    //           the top stage) and are read from dst
    {
        const bool need_all = !fast;
        if (need_all || (INV && !A.last_stage)) {
            for (int j = wid; j < nt; j += nw) {
                const bool survivor = (spart[j] == 0xffffu);
                bool from_dst;
                if (!INV) from_dst = !ident;
                else from_dst = survivor && !A.last_stage;
                if (!need_all && !from_dst) continue;     // already in LDS from the bulk load
                const int64_t r = srow[j];
                const T *gp = from_dst ? (A.dst + r * A.ld_dst + c_base) : (A.src + r * A.ld_src + c_base);
                if (lane < Dc) tile[j * Dc + lane] = gp[lane];
            }
        }
    }
    __syncthreads();

    // ---- 6. butterflies, one round per level present ----
    {
        const int Lp = 1 << A.lp_shift;
        const int gpw = 64 >> A.lp_shift;
        const int g = lane >> A.lp_shift, c = lane & (Lp - 1);
        uint64_t mask = ((uint64_t)lmask[1] << 32) | lmask[0];
        while (mask) {
            const int l = INV ? (63 - __clzll((long long)mask)) : (__ffsll((long long)mask) - 1);
            mask &= ~(1ull << l);
            const uint32_t base = loff[l], cnt = hist[l];
            for (uint32_t mb = wid * gpw; mb < cnt; mb += nw * gpw) {
                const uint32_t m = mb + g;
                if (m < cnt && c < Dc) {
                    const int j = smlist[base + m];
                    const int p = spart[j];
                    const T a = sa[j], b = sb[j];
                    const T x0 = tile[p * Dc + c], x1 = tile[j * Dc + c];
                    if (!INV) {                           // RAHT.py:331-332
                        tile[p * Dc + c] = a * x0 + b * x1;
                        tile[j * Dc + c] = a * x1 - b * x0;
                    } else {                              // iRAHT.py:108-109
                        tile[p * Dc + c] = a * x0 - b * x1;
                        tile[j * Dc + c] = b * x0 + a * x1;
                    }
                }
            }
            __syncthreads();
        }
    }

    // ---- 7. write back ----
    if (fast) {
        typedef typename Vec16<T>::type V;
        constexpr int VN = Vec16<T>::n;
        T *gdst = A.dst + e0 * A.ld_dst;
        const int nelem = nt * Dc;
        const int nvec = nelem / VN;
        V *g4 = (V *)gdst;
        const V *l4 = (const V *)tile;
        for (int v = tid; v < nvec; v += nthreads) g4[v] = l4[v];
        for (int e = nvec * VN + tid; e < nelem; e += nthreads) gdst[e] = tile[e];
    } else {
        for (int j = wid; j < nt; j += nw) {
            const int64_t r = srow[j];
            if (lane < Dc) A.dst[r * A.ld_dst + c_base + lane] = tile[j * Dc + lane];
        }
    }
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

static int lp_shift_for(int Dc)
{
    int s = 0;
    while ((1 << s) < Dc && s < 6) ++s;
    return s;
}

template <typename T, bool INV>
static int launch_tile_stage(const raht_plan *p, const Schedule &sc, int k, const T *src, int64_t ld_src,
                             T *dst, int64_t ld_dst, int D, int Dc, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel<T, INV, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        RAHT_HIP_CHECK(hipFuncSetAttribute((const void *)tile_kernel<T, INV, false>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const Stage &st = sc.stages[(size_t)k];
    TileArgs<T> A;
    A.src = src; A.ld_src = ld_src; A.dst = dst; A.ld_dst = ld_dst;
    A.rows = st.rows; A.n_entries = st.n_entries; A.N = p->N; A.R = sc.tile_rows;
    A.D = D; A.Dc = Dc; A.lp_shift = lp_shift_for(Dc);
    A.last_stage = (k == (int)sc.stages.size() - 1) ? 1 : 0;
    A.vec_ok = (Dc == D && ld_src == D && ld_dst == D && ((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0)) ? 1 : 0;
    A.lvl = p->lvl; A.wl = p->wl; A.wr = p->wr; A.wsum = p->wsum;
    const int nchunks = (D + Dc - 1) / Dc;
    const size_t lds = tile_lds_bytes(sc.tile_rows, (int)sizeof(T), Dc);
    if (st.rows == nullptr)
        hipLaunchKernelGGL((tile_kernel<T, INV, true>), dim3((unsigned)st.n_tiles, (unsigned)nchunks), dim3(256), lds, s, A);
    else
        hipLaunchKernelGGL((tile_kernel<T, INV, false>), dim3((unsigned)st.n_tiles, (unsigned)nchunks), dim3(256), lds, s, A);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

template <typename T, bool INV>
static int run_level_engine(const raht_plan *p, const T *src, int64_t ld_src, T *dst, int64_t ld_dst,
                            int D, hipStream_t s)
{
    if ((const void *)src != (const void *)dst) {
        const int64_t total = p->N * D;
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(total, 256), 8192);
        hipLaunchKernelGGL(copy_rows_kernel<T>, dim3(gb), dim3(256), 0, s, src, ld_src, dst, ld_dst, p->N, D);
    }
    const int lps = lp_shift_for(std::min(D, 64));
    const int gpw = 64 >> lps;
    for (int q = 0; q <= p->max_level; ++q) {
        const int l = INV ? p->max_level - q : q;
        const uint32_t cnt = p->level_off[l + 1] - p->level_off[l];
        if (cnt == 0) continue;                                  // RAHT.py:304-305
        const int64_t steps = ceil_div(cnt, gpw * 4);
        const unsigned gb = (unsigned)std::min<int64_t>(steps, 2048);
        hipLaunchKernelGGL((level_pass_kernel<T, INV>), dim3(gb), dim3(256), 0, s, dst, ld_dst, D,
                           p->level_rows + p->level_off[l], cnt, p->wl, p->wr, p->wsum, lps);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

template <typename T, bool INV>
static int run_transform(const raht_plan *cp, const T *src, int64_t ld_src, int D, T *dst, int64_t ld_dst,
                         T *w, hipStream_t s)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    if (!p || !src || !dst) { set_error("raht transform: NULL argument"); return RAHT_ERR_INVALID; }
    if (D < 1 || ld_src < D || ld_dst < D) { set_error("raht transform: bad D/ld (D=%d ld_src=%lld ld_dst=%lld)", D, (long long)ld_src, (long long)ld_dst); return RAHT_ERR_INVALID; }
    int rc = RAHT_OK;
    bool use_level = (p->engine == RAHT_ENGINE_LEVEL);
    const Schedule *sc = nullptr;
    int Dc = 0;
    if (!use_level) {
        Dc = pick_chunk_channels((int)sizeof(T), D);
        const int R = pick_tile_rows(p, (int)sizeof(T), Dc);
        if (R == 0) use_level = true;
        else {
            RAHT_RET(get_schedule(p, R, s, &sc));
            if (!sc->valid) use_level = true;              // pathological key pattern, see plan.hip
        }
    }
    if (use_level) {
        rc = run_level_engine<T, INV>(p, src, ld_src, dst, ld_dst, D, s);
    } else {
        const int K = (int)sc->stages.size();
        for (int q = 0; q < K && rc == RAHT_OK; ++q) {
            const int k = INV ? K - 1 - q : q;
            rc = launch_tile_stage<T, INV>(p, *sc, k, src, ld_src, dst, ld_dst, D, Dc, s);
        }
    }
    if (rc == RAHT_OK && w) {
        hipLaunchKernelGGL(node_weight_kernel<T>, dim3((unsigned)ceil_div(p->N, 256)), dim3(256), 0, s,
                           p->wl, p->wr, p->wsum, p->N, w);
        RAHT_HIP_CHECK(hipGetLastError());
    }
    return rc;
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_fwd(const raht_plan *plan, const float *C, int64_t ldc, int D, float *T, int64_t ldt, float *w,
             raht_stream_t stream)
{
    return run_transform<float, false>(plan, C, ldc, D, T, ldt, w, (hipStream_t)stream);
}

int raht_fwd_f64(const raht_plan *plan, const double *C, int64_t ldc, int D, double *T, int64_t ldt,
                 double *w, raht_stream_t stream)
{
    return run_transform<double, false>(plan, C, ldc, D, T, ldt, w, (hipStream_t)stream);
}

int raht_inv(const raht_plan *plan, const float *T, int64_t ldt, int D, float *C, int64_t ldc,
             raht_stream_t stream)
{
    return run_transform<float, true>(plan, T, ldt, D, C, ldc, nullptr, (hipStream_t)stream);
}

int raht_inv_f64(const raht_plan *plan, const double *T, int64_t ldt, int D, double *C, int64_t ldc,
                 raht_stream_t stream)
{
    return run_transform<double, true>(plan, T, ldt, D, C, ldc, nullptr, (hipStream_t)stream);
}

/* Profiling aid: enqueue ONE stage of the float32 tile schedule (stage 0 = the HBM-heavy launch).
 * Results are only meaningful as part of a full transform; bench.py uses this to time the dominant
 * kernel in isolation with HIP events. */
int raht_debug_run_stage(const raht_plan *cp, int inverse, int stage, const float *src, int64_t ld_src, int D,
                         float *dst, int64_t ld_dst, raht_stream_t stream)
{
    raht_plan *p = const_cast<raht_plan *>(cp);
    if (!p || !src || !dst || D < 1) { set_error("raht_debug_run_stage: bad argument"); return RAHT_ERR_INVALID; }
    const int Dc = pick_chunk_channels(4, D);
    const int R = pick_tile_rows(p, 4, Dc);
    if (R == 0) { set_error("raht_debug_run_stage: no tile size"); return RAHT_ERR_UNSUPPORTED; }
    const Schedule *sc = nullptr;
    RAHT_RET(get_schedule(p, R, (hipStream_t)stream, &sc));
    if (stage < 0 || stage >= (int)sc->stages.size()) { set_error("raht_debug_run_stage: stage out of range"); return RAHT_ERR_INVALID; }
    if (inverse) return launch_tile_stage<float, true>(p, *sc, stage, src, ld_src, dst, ld_dst, D, Dc, (hipStream_t)stream);
    return launch_tile_stage<float, false>(p, *sc, stage, src, ld_src, dst, ld_dst, D, Dc, (hipStream_t)stream);
}

}  // extern "C"
