// quant.hip -- quantize + reorder / dequantize + un-reorder, the driver-inline arithmetic around
// the transform (reference python/encode_3dgs.py:204 floor(x/step+0.5), :210 index_select(0,
// order_RAGFT), :215 int32; :261 x*step, :267-268 gather by argsort(order_RAGFT)).
//
// 16-byte row chunks (raht_device.h): a lane moves 4 consecutive channels of a row, G lanes cover a
// row, a wave instruction moves 64 / G rows; the last chunk of a row whose length is not a multiple
// of 4 is the 16 bytes that END it (overlapping its neighbour: same value written twice). The
// division is the hoisted-reciprocal refinement of the fused kernels (bit-identical to x / step).
// HBM-bound, 8 bytes/element. Rows narrower than one chunk (D < 4) take the scalar kernels.
#include "raht_common.h"
#include "raht_device.h"

#include <algorithm>

namespace raht {

template <bool QUANT>
__global__ __launch_bounds__(256) void reorder_chunk_kernel(const void *__restrict__ src_, int64_t lds, int D, int lg,
                                                            const uint32_t *__restrict__ perm, int64_t N,
                                                            const StepTable steps, void *__restrict__ dst_, int64_t ldd)
{
    // Both directions GATHER rows through a permutation and write consecutive rows (scattered row
    // writes measured 0.47 ms against 0.35 ms for scattered row reads on cfg3):
    // QUANT: dst = Q row k      <- src = T row perm[k],  perm = order_RAGFT                 (:210)
    // else : dst = T row k      <- src = Q row perm[k],  perm = its inverse (argsort, :267-268)
    const int lane = threadIdx.x & 63;
    const int G = 1 << lg, rpi = 64 >> lg;
    const int g = lane >> lg, c4 = lane & (G - 1);
    const int NC = (D + 3) >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int cc = c4; cc < NC; cc += G) {                 // one pass unless D > 256
        const int goff = min(cc * 4, D - 4);
        float sp[4], rc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sp[i] = steps.v[steps.n == 1 ? 0 : goff + i];
            rc[i] = refined_rcp(sp[i]);
        }
        for (int64_t k0 = wave * rpi * 4; k0 < N; k0 += nwaves * rpi * 4) {
            // 4 row groups in flight per lane: all loads are issued before the first is consumed
            int64_t k[4], r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                k[u] = min(k0 + u * rpi + g, N - 1);
                r[u] = (int64_t)perm[k[u]];
            }
            if constexpr (QUANT) {
                RegChunk<float> x[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) x[u] = ld_chunk<float, true>((const float *)src_ + r[u] * lds + goff);     // :210
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    RegChunk<int32_t> q;
#pragma unroll
                    for (int i = 0; i < 4; ++i) q.v[i] = quantize_one(x[u].v[i], sp[i], rc[i], steps.fast_div);      // :204, :215
                    if (k0 + u * rpi + g < N) st_chunk<int32_t, true>((int32_t *)dst_ + k[u] * ldd + goff, q);
                }
            } else {
                RegChunk<int32_t> q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = ld_chunk<int32_t, true>((const int32_t *)src_ + r[u] * lds + goff);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    RegChunk<float> x;
#pragma unroll
                    for (int i = 0; i < 4; ++i) x.v[i] = (float)q[u].v[i] * sp[i];                                     // :261
                    if (k0 + u * rpi + g < N) st_chunk<float, true>((float *)dst_ + k[u] * ldd + goff, x);
                }
            }
        }
    }
}

// scalar fall-backs for matrices narrower than one chunk (D < 4): one wave per row, lanes = channels
__global__ __launch_bounds__(256) void quant_reorder_kernel(const float *__restrict__ T, int64_t ldt, int D,
                                                            const uint32_t *__restrict__ order, int64_t N,
                                                            const StepTable steps, int32_t *__restrict__ Q,
                                                            int64_t ldq)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t k = wave; k < N; k += nwaves) {
        const float *src = T + (int64_t)order[k] * ldt;          // :210
        int32_t *dst = Q + k * ldq;
        for (int c = lane; c < D; c += 64) {
            const float st = steps.v[steps.n == 1 ? 0 : c];
            dst[c] = (int32_t)floorf(src[c] / st + 0.5f);         // :204, :215
        }
    }
}

__global__ __launch_bounds__(256) void dequant_unreorder_kernel(const int32_t *__restrict__ Q, int64_t ldq, int D,
                                                                const uint32_t *__restrict__ order, int64_t N,
                                                                const StepTable steps, float *__restrict__ T,
                                                                int64_t ldt)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t k = wave; k < N; k += nwaves) {
        const int32_t *src = Q + k * ldq;
        float *dst = T + (int64_t)order[k] * ldt;                 // inverse of :210 == :267-268
        for (int c = lane; c < D; c += 64) {
            const float st = steps.v[steps.n == 1 ? 0 : c];
            dst[c] = (float)src[c] * st;                          // :261
        }
    }
}

// A few rows at explicit positions (the <= 512 top coefficients of a Morton-prefix sharded scene, which
// the shard-local fused kernels leave to the caller): Q[pos[i], :] = quantize(X[i, :]) and back.
__global__ __launch_bounds__(256) void quant_rows_kernel(const float *__restrict__ X, int64_t ldx, int64_t n, int D,
                                                         const int64_t *__restrict__ pos, const StepTable steps,
                                                         int32_t *__restrict__ Q, int64_t ldq)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const float *src = X + i * ldx;
        int32_t *dst = Q + (pos ? pos[i] : i) * ldq;
        for (int c = lane; c < D; c += 64) dst[c] = (int32_t)floorf(src[c] / steps.v[steps.n == 1 ? 0 : c] + 0.5f);
    }
}

__global__ __launch_bounds__(256) void dequant_rows_kernel(const int32_t *__restrict__ Q, int64_t ldq,
                                                           const int64_t *__restrict__ pos, int64_t n, int D,
                                                           const StepTable steps, float *__restrict__ X, int64_t ldx)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int32_t *src = Q + (pos ? pos[i] : i) * ldq;
        float *dst = X + i * ldx;
        for (int c = lane; c < D; c += 64) dst[c] = (float)src[c] * steps.v[steps.n == 1 ? 0 : c];
    }
}

// float64 (the reference's own precision, python/encode_3dgs.py:82-83,204): one lane per element, rows gathered
// through the permutation as above. The division is the IEEE double division of the reference's CPU path.

template <bool QUANT>
__global__ __launch_bounds__(256) void reorder_f64_kernel(const void *__restrict__ src_, int64_t lds, int D,
                                                          const uint32_t *__restrict__ perm, int64_t N,
                                                          const StepTable64 steps, void *__restrict__ dst_, int64_t ldd)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t k = wave; k < N; k += nwaves) {
        const int64_t r = (int64_t)perm[k];
        for (int c = lane; c < D; c += 64) {
            const double st = steps.v[steps.n == 1 ? 0 : c];
            if constexpr (QUANT) ((int32_t *)dst_)[k * ldd + c] = quantize_one_f64(((const double *)src_)[r * lds + c], st);        // :204, :210, :215
            else ((double *)dst_)[k * ldd + c] = (double)((const int32_t *)src_)[r * lds + c] * st;                                // :261, :267-268
        }
    }
}

static int fill_steps64(StepTable64 &t, const double *steps, int n_steps, int D)
{
    if (!steps || !(n_steps == 1 || n_steps == D)) { set_error("quant: n_steps must be 1 or D"); return RAHT_ERR_INVALID; }
    if (n_steps > MAX_STEP_CH) { set_error("quant: per-channel steps support D <= %d", MAX_STEP_CH); return RAHT_ERR_UNSUPPORTED; }
    t.n = n_steps;
    for (int c = 0; c < n_steps; ++c) {
        if (!(steps[c] > 0.0)) { set_error("quant: step[%d] must be > 0", c); return RAHT_ERR_INVALID; }
        t.v[c] = steps[c];
    }
    return RAHT_OK;
}

// Rows of 32-bit words at explicit positions, without arithmetic: GATHER dst[i, :] = src[pos[i], :] or
// SCATTER dst[pos[i], :] = src[i, :]. The <= 512 root rows of a Morton-prefix sharded scene travel between the
// coefficient matrix and the all-gather buffers with one launch of this instead of a chain of torch index ops.
template <bool SCATTER>
__global__ __launch_bounds__(256) void rows_move_kernel(const uint32_t *__restrict__ src, int64_t lds, const int64_t *__restrict__ pos,
                                                        int64_t n, int words, uint32_t *__restrict__ dst, int64_t ldd)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < n; i += nwaves) {
        const int64_t r = pos[i];
        const uint32_t *a = src + (SCATTER ? i : r) * lds;
        uint32_t *b = dst + (SCATTER ? r : i) * ldd;
        for (int c = lane; c < words; c += 64) b[c] = a[c];
    }
}

// int32 matrix transpose through an LDS tile (64 rows x 64 columns, padded): row-major N x D  <->
// channel-major D x N, so that the entropy stage reads / writes contiguous channels. Both global
// sides are coalesced (lanes along the contiguous dimension).
__global__ __launch_bounds__(256) void transpose_i32_kernel(const int32_t *__restrict__ in, int64_t ld_in, int64_t rows,
                                                            int64_t cols, int32_t *__restrict__ out, int64_t ld_out)
{
    __shared__ int32_t t[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;            // 64 x 4
    for (int k = ly; k < 64; k += 4) {
        const int64_t r = r0 + k, c = c0 + lx;
        if (r < rows && c < cols) t[k][lx] = in[r * ld_in + c];
    }
    __syncthreads();
    for (int k = ly; k < 64; k += 4) {
        const int64_t c = c0 + k, r = r0 + lx;                         // out is cols x rows
        if (r < rows && c < cols) out[c * ld_out + r] = t[lx][k];
    }
}

static int fill_steps(StepTable &t, const float *steps, int n_steps, int D)
{
    if (!steps || !(n_steps == 1 || n_steps == D)) { set_error("quant: n_steps must be 1 or D"); return RAHT_ERR_INVALID; }
    if (n_steps > MAX_STEP_CH) { set_error("quant: per-channel steps support D <= %d", MAX_STEP_CH); return RAHT_ERR_UNSUPPORTED; }
    for (int c = 0; c < n_steps; ++c)
        if (!(steps[c] > 0.0f)) { set_error("quant: step[%d] must be > 0", c); return RAHT_ERR_INVALID; }
    fill_step_table(t, steps, n_steps);
    return RAHT_OK;
}

// lanes per row of the chunked kernels: power of two >= chunks per row, at most a whole wave
static int lanes_log2(int D)
{
    const int nc = (D + 3) / 4;
    int lg = 0;
    while ((1 << lg) < nc && lg < 6) ++lg;
    return lg;
}

// ---- per-column sums of squared differences (the drivers' five PSNR columns, python/encode_3dgs.py:298-310) -------------
// One pass over both matrices instead of five torch.mean((a - b) ** 2) reductions with a host round trip each (2.0 ms per
// quantization step of a 3 M x 56 frame). A lane owns one column of a group of rows (lanes = columns: coalesced rows), differences
// in the matrices' own type (as torch forms them), squares and sums in float64; partial[block][column] by a fixed tree, then one
// workgroup adds the blocks' rows in block order: the result does not depend on the launch's timing.
constexpr int SQD_THREADS = 256;
template <typename T>
__global__ __launch_bounds__(SQD_THREADS) void sqdiff_partial_kernel(const T *__restrict__ A, int64_t lda, const T *__restrict__ B, int64_t ldb,
                                                                     int64_t N, int D, int cpr /* lanes per row: power of two >= min(D, 256) */,
                                                                     double *__restrict__ partial)
{
    __shared__ double red[SQD_THREADS];
    const int rows_per_iter = SQD_THREADS / cpr;
    const int c0 = threadIdx.x & (cpr - 1), r0 = threadIdx.x / cpr;
    for (int cb = 0; cb < D; cb += cpr) {                  // (one pass unless D > 256)
        const int c = cb + c0;
        double acc = 0.0;
        if (c < D) {
            // eight rows in flight per lane (one at a time the loop is a chain of HBM round trips: 0.79 ms for 1.3 GB)
            const int64_t S = (int64_t)gridDim.x * rows_per_iter;
            for (int64_t i = (int64_t)blockIdx.x * rows_per_iter + r0; i < N; i += 8 * S) {
                T a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t k = min(i + u * S, N - 1);
                    a[u] = __builtin_nontemporal_load(A + k * lda + c);
                    b[u] = __builtin_nontemporal_load(B + k * ldb + c);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const T dlt = a[u] - b[u];
                    if (i + u * S < N) acc += (double)dlt * (double)dlt;
                }
            }
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        for (int h = rows_per_iter >> 1; h >= 1; h >>= 1) {          // fixed tree over the row groups
            if (r0 < h) red[threadIdx.x] += red[threadIdx.x + h * cpr];
            __syncthreads();
        }
        if (r0 == 0 && c < D) partial[(size_t)blockIdx.x * D + c] = red[c0];
        __syncthreads();
    }
}

// one wave per column: lane l adds the blocks l, l + 64, ... in order, then a fixed butterfly over the lanes (one thread per
// column walking 2048 partial rows was a chain of dependent loads: 0.5 ms of the call's 0.8)
__global__ __launch_bounds__(256) void sqdiff_final_kernel(const double *__restrict__ partial, int nblocks, int D, double *__restrict__ out)
{
    const int c = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= D) return;
    double s = 0.0;
    for (int b = lane; b < nblocks; b += 64) s += partial[(size_t)b * D + c];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    if (lane == 0) out[c] = s;
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_quant_reorder(const raht_plan *p, const float *T, int64_t ldt, int D, const float *steps, int n_steps,
                       int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (!p || !T || !Q || D < 1 || ldt < D || ldq < D) { set_error("raht_quant_reorder: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_quant_reorder"));
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    if (D >= 4) {
        const int lg = lanes_log2(D);
        const int64_t rows_per_block = (int64_t)4 * 4 * (64 >> lg);           // 4 waves x 4 row groups in flight
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, rows_per_block), 16384);
        hipLaunchKernelGGL(reorder_chunk_kernel<true>, dim3(gb), dim3(256), 0, (hipStream_t)stream, (const void *)T, ldt, D,
                           lg, p->order, p->N, st, (void *)Q, ldq);
    } else {
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, 4), 4096);
        hipLaunchKernelGGL(quant_reorder_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, T, ldt, D, p->order,
                           p->N, st, Q, ldq);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_dequant_unreorder(const raht_plan *p, const int32_t *Q, int64_t ldq, int D, const float *steps,
                           int n_steps, float *T, int64_t ldt, raht_stream_t stream)
{
    if (!p || !T || !Q || D < 1 || ldt < D || ldq < D) { set_error("raht_dequant_unreorder: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_dequant_unreorder"));
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    if (D >= 4) {
        const int lg = lanes_log2(D);
        const int64_t rows_per_block = (int64_t)4 * 4 * (64 >> lg);
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, rows_per_block), 16384);
        hipLaunchKernelGGL(reorder_chunk_kernel<false>, dim3(gb), dim3(256), 0, (hipStream_t)stream, (const void *)Q, ldq, D,
                           lg, p->inv_order, p->N, st, (void *)T, ldt);
    } else {
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, 4), 4096);
        hipLaunchKernelGGL(dequant_unreorder_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, Q, ldq, D,
                           p->order, p->N, st, T, ldt);
    }
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_quant_reorder_f64(const raht_plan *p, const double *T, int64_t ldt, int D, const double *steps, int n_steps,
                           int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (!p || !T || !Q || D < 1 || ldt < D || ldq < D) { set_error("raht_quant_reorder_f64: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_quant_reorder_f64"));
    StepTable64 st;
    RAHT_RET(fill_steps64(st, steps, n_steps, D));
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, 4), 16384);
    hipLaunchKernelGGL(reorder_f64_kernel<true>, dim3(gb), dim3(256), 0, (hipStream_t)stream, (const void *)T, ldt, D, p->order,
                       p->N, st, (void *)Q, ldq);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_dequant_unreorder_f64(const raht_plan *p, const int32_t *Q, int64_t ldq, int D, const double *steps, int n_steps,
                               double *T, int64_t ldt, raht_stream_t stream)
{
    if (!p || !T || !Q || D < 1 || ldt < D || ldq < D) { set_error("raht_dequant_unreorder_f64: bad argument"); return RAHT_ERR_INVALID; }
    RAHT_RET(check_plan_device(p, "raht_dequant_unreorder_f64"));
    StepTable64 st;
    RAHT_RET(fill_steps64(st, steps, n_steps, D));
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, 4), 16384);
    hipLaunchKernelGGL(reorder_f64_kernel<false>, dim3(gb), dim3(256), 0, (hipStream_t)stream, (const void *)Q, ldq, D,
                       p->inv_order, p->N, st, (void *)T, ldt);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_quant_rows(const float *X, int64_t ldx, int64_t n, int D, const float *steps, int n_steps,
                    const int64_t *pos, int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (n < 0 || D < 1) { set_error("raht_quant_rows: bad argument"); return RAHT_ERR_INVALID; }
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    if (n == 0) return RAHT_OK;                              // no rows: no buffers needed (empty device tensors are NULL, their strides 0)
    if (!X || !Q || ldx < D || ldq < D) { set_error("raht_quant_rows: bad argument"); return RAHT_ERR_INVALID; }
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(n, 4), 4096);
    hipLaunchKernelGGL(quant_rows_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, X, ldx, n, D, pos, st, Q, ldq);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_dequant_rows(const int32_t *Q, int64_t ldq, const int64_t *pos, int64_t n, int D, const float *steps,
                      int n_steps, float *X, int64_t ldx, raht_stream_t stream)
{
    if (n < 0 || D < 1) { set_error("raht_dequant_rows: bad argument"); return RAHT_ERR_INVALID; }
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    if (n == 0) return RAHT_OK;
    if (!X || !Q || ldx < D || ldq < D) { set_error("raht_dequant_rows: bad argument"); return RAHT_ERR_INVALID; }
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(n, 4), 4096);
    hipLaunchKernelGGL(dequant_rows_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, Q, ldq, pos, n, D, st, X, ldx);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

static int rows_move(bool scatter, const void *src, int64_t ld_src, const int64_t *pos, int64_t n, int D, int elem_size,
                     void *dst, int64_t ld_dst, raht_stream_t stream, const char *what)
{
    if (n < 0 || D < 1 || (elem_size != 4 && elem_size != 8)) { set_error("%s: bad argument", what); return RAHT_ERR_INVALID; }
    if (n == 0) return RAHT_OK;                              // no rows: no buffers needed (empty device tensors are NULL, their strides 0)
    if (!src || !dst || !pos || ld_src < D || ld_dst < D) { set_error("%s: bad argument", what); return RAHT_ERR_INVALID; }
    const int wpe = elem_size / 4;
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(n, 4), 4096);
    if (scatter)
        hipLaunchKernelGGL(rows_move_kernel<true>, dim3(gb), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)src, ld_src * wpe, pos, n, D * wpe, (uint32_t *)dst, ld_dst * wpe);
    else
        hipLaunchKernelGGL(rows_move_kernel<false>, dim3(gb), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)src, ld_src * wpe, pos, n, D * wpe, (uint32_t *)dst, ld_dst * wpe);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_rows_gather(const void *src, int64_t ld_src, const int64_t *pos, int64_t n, int D, int elem_size, void *dst,
                     int64_t ld_dst, raht_stream_t stream)
{
    return rows_move(false, src, ld_src, pos, n, D, elem_size, dst, ld_dst, stream, "raht_rows_gather");
}

int raht_rows_scatter(const void *src, int64_t ld_src, const int64_t *pos, int64_t n, int D, int elem_size, void *dst,
                      int64_t ld_dst, raht_stream_t stream)
{
    return rows_move(true, src, ld_src, pos, n, D, elem_size, dst, ld_dst, stream, "raht_rows_scatter");
}

int raht_transpose_i32(const int32_t *in, int64_t ld_in, int64_t rows, int64_t cols, int32_t *out, int64_t ld_out,
                       raht_stream_t stream)
{
    if (!in || !out || rows < 0 || cols < 0 || ld_in < cols || ld_out < rows) { set_error("raht_transpose_i32: bad argument"); return RAHT_ERR_INVALID; }
    if (rows == 0 || cols == 0) return RAHT_OK;
    const dim3 grid((unsigned)ceil_div(rows, 64), (unsigned)ceil_div(cols, 64));
    hipLaunchKernelGGL(transpose_i32_kernel, grid, dim3(256), 0, (hipStream_t)stream, in, ld_in, rows, cols, out, ld_out);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_sqdiff_columns(const void *A, int64_t lda, const void *B, int64_t ldb, int64_t N, int D, int dtype, double *out,
                        raht_stream_t stream)
{
    if (N < 0 || D < 1 || (dtype != RAHT_F32 && dtype != RAHT_F64) || !out) { set_error("raht_sqdiff_columns: bad argument"); return RAHT_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    if (N == 0) { RAHT_HIP_CHECK(hipMemsetAsync(out, 0, sizeof(double) * (size_t)D, s)); return RAHT_OK; }
    if (!A || !B || lda < D || ldb < D) { set_error("raht_sqdiff_columns: bad argument"); return RAHT_ERR_INVALID; }
    int cpr = 1;
    while (cpr < D && cpr < SQD_THREADS) cpr <<= 1;
    const int rows_per_iter = SQD_THREADS / cpr;
    const int nb = (int)std::min<int64_t>(ceil_div(N, (int64_t)rows_per_iter * 8), 1024);
    Scratch ws(sizeof(double) * (size_t)nb * (size_t)D, s);
    if (!ws.ok()) return RAHT_ERR_NOMEM;
    if (dtype == RAHT_F32)
        hipLaunchKernelGGL(sqdiff_partial_kernel<float>, dim3(nb), dim3(SQD_THREADS), 0, s, (const float *)A, lda, (const float *)B, ldb, N, D, cpr, ws.as<double>());
    else
        hipLaunchKernelGGL(sqdiff_partial_kernel<double>, dim3(nb), dim3(SQD_THREADS), 0, s, (const double *)A, lda, (const double *)B, ldb, N, D, cpr, ws.as<double>());
    hipLaunchKernelGGL(sqdiff_final_kernel, dim3((unsigned)ceil_div(D, 4)), dim3(256), 0, s, ws.as<double>(), nb, D, out);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

}  // extern "C"
