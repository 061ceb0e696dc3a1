// quant.hip -- quantize + reorder / dequantize + un-reorder, the driver-inline arithmetic around
// the transform (reference python/encode_3dgs.py:204 floor(x/step+0.5), :210 index_select(0,
// order_RAGFT), :215 int32; :261 x*step, :267-268 gather by argsort(order_RAGFT)).
//
// One wave per coefficient row: lanes map to channels, so the permuted row read (gather through
// order_RAGFT) and the row write are each a single coalesced segment. HBM-bound, 8 bytes/element.
#include "raht_common.h"

#include <algorithm>

namespace raht {

constexpr int MAX_STEP_CH = 256;

struct StepTable {
    int n;                         // 1 or D
    float v[MAX_STEP_CH];
};

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
    t.n = n_steps;
    for (int c = 0; c < n_steps; ++c) {
        if (!(steps[c] > 0.0f)) { set_error("quant: step[%d] must be > 0", c); return RAHT_ERR_INVALID; }
        t.v[c] = steps[c];
    }
    return RAHT_OK;
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_quant_reorder(const raht_plan *p, const float *T, int64_t ldt, int D, const float *steps, int n_steps,
                       int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (!p || !T || !Q || D < 1 || ldt < D || ldq < D) { set_error("raht_quant_reorder: bad argument"); return RAHT_ERR_INVALID; }
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, 4), 4096);
    hipLaunchKernelGGL(quant_reorder_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, T, ldt, D, p->order,
                       p->N, st, Q, ldq);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_dequant_unreorder(const raht_plan *p, const int32_t *Q, int64_t ldq, int D, const float *steps,
                           int n_steps, float *T, int64_t ldt, raht_stream_t stream)
{
    if (!p || !T || !Q || D < 1 || ldt < D || ldq < D) { set_error("raht_dequant_unreorder: bad argument"); return RAHT_ERR_INVALID; }
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(p->N, 4), 4096);
    hipLaunchKernelGGL(dequant_unreorder_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, Q, ldq, D,
                       p->order, p->N, st, T, ldt);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_quant_rows(const float *X, int64_t ldx, int64_t n, int D, const float *steps, int n_steps,
                    const int64_t *pos, int32_t *Q, int64_t ldq, raht_stream_t stream)
{
    if (!X || !Q || n < 0 || D < 1 || ldx < D || ldq < D) { set_error("raht_quant_rows: bad argument"); return RAHT_ERR_INVALID; }
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    if (n == 0) return RAHT_OK;
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(n, 4), 4096);
    hipLaunchKernelGGL(quant_rows_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, X, ldx, n, D, pos, st, Q, ldq);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_dequant_rows(const int32_t *Q, int64_t ldq, const int64_t *pos, int64_t n, int D, const float *steps,
                      int n_steps, float *X, int64_t ldx, raht_stream_t stream)
{
    if (!X || !Q || n < 0 || D < 1 || ldx < D || ldq < D) { set_error("raht_dequant_rows: bad argument"); return RAHT_ERR_INVALID; }
    StepTable st;
    RAHT_RET(fill_steps(st, steps, n_steps, D));
    if (n == 0) return RAHT_OK;
    const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(n, 4), 4096);
    hipLaunchKernelGGL(dequant_rows_kernel, dim3(gb), dim3(256), 0, (hipStream_t)stream, Q, ldq, pos, n, D, st, X, ldx);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
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

}  // extern "C"
