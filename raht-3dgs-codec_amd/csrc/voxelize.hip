// voxelize.hip -- on-device voxelizer: shift / quantize / clamp, 3J-bit Morton key, stable LSD radix
// sort (8-bit digits, wave64 ballot ranking -- scan_sort.hip), voxel boundaries, per-voxel mean.
// Replaces voxelize_pc_batched (reference python/voxelize_pc.py:62-172). Float32 arithmetic, as
// torch performs it on a float32 point cloud; integer outputs are bit-exact.
#include "raht_common.h"
#include <atomic>
#include "raht_device.h"

#include <algorithm>
#include <cmath>

namespace raht {

__device__ __forceinline__ uint32_t vx_compact3(uint64_t v)
{
    v &= 0x1249249249249249ull;
    v = (v | (v >> 2)) & 0x10c30c30c30c30c3ull;
    v = (v | (v >> 4)) & 0x100f00f00f00f00full;
    v = (v | (v >> 8)) & 0x001f0000ff0000ffull;
    v = (v | (v >> 16)) & 0x001f00000000ffffull;
    v = (v | (v >> 32)) & 0x1fffffull;
    return (uint32_t)v;
}

// ---- per-axis min / global max reductions (two-stage: block partials, then host) ---------------
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ PC, int64_t ld, int64_t N,
                                                     float s0, float s1, float s2,
                                                     float *__restrict__ part /* [grid][4]: min x,y,z, max */)
{
    float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mx = -INFINITY;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const float x = PC[i * ld + 0], y = PC[i * ld + 1], z = PC[i * ld + 2];
        mn0 = fminf(mn0, x); mn1 = fminf(mn1, y); mn2 = fminf(mn2, z);
        mx = fmaxf(mx, fmaxf(x - s0, fmaxf(y - s1, z - s2)));      // voxelize_pc.py:92,95
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        mn0 = fminf(mn0, __shfl_xor(mn0, d, 64)); mn1 = fminf(mn1, __shfl_xor(mn1, d, 64));
        mn2 = fminf(mn2, __shfl_xor(mn2, d, 64)); mx = fmaxf(mx, __shfl_xor(mx, d, 64));
    }
    __shared__ float sm[4][4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (lane == 0) { sm[wid][0] = mn0; sm[wid][1] = mn1; sm[wid][2] = mn2; sm[wid][3] = mx; }
    __syncthreads();
    if (threadIdx.x < 4) {
        float v = sm[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) v = (threadIdx.x < 3) ? fminf(v, sm[w][threadIdx.x]) : fmaxf(v, sm[w][threadIdx.x]);
        part[blockIdx.x * 4 + threadIdx.x] = v;
    }
}

// ---- keys --------------------------------------------------------------------------------------
// Every load instruction of a wave touches 64 different 128-byte lines here (row stride 236 bytes at d = 56): the kernel
// moves one line per point (384 MB on 3 M points) whatever the instruction mix -- 126 us with three dword loads per row and
// with one dwordx3 alike (3 TB/s of lines for 36 MB of coordinates). Only a layout with the positions apart from the
// attributes would change that, and the reference's PC matrix is the boundary. (raht_voxelize computes its keys inside the
// sort's histogram launch -- scan_sort.hip, os_hist_kernel<true> -- this kernel serves raht_voxel_keys.)
__global__ __launch_bounds__(256) void vox_keys_kernel(const VoxGrid G, int64_t N, uint64_t *__restrict__ keys)
{
    constexpr int U = 4;                                             // rows per thread: four independent loads in flight
    const int64_t i0 = ((int64_t)blockIdx.x * blockDim.x) * U + threadIdx.x;
    Xyz p[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = min(i0 + (int64_t)u * blockDim.x, N - 1);
        p[u] = *(const Xyz *)(G.PC + i * G.ld);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = i0 + (int64_t)u * blockDim.x;
        if (i < N) keys[i] = vox_key(p[u], G);
    }
}

__global__ void boundary_kernel(const uint64_t *__restrict__ keys, int64_t N, uint32_t *__restrict__ flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1u : 0u;       // voxelize_pc.py:114-118
}

// The voxelizer's secondary outputs (voxelize_pc.py:103-111, 147-156): one lane per output element, rows gathered
// through the sort permutation. vid[k] = voxel of sorted point k (boundary flags scanned).
__global__ __launch_bounds__(256) void residual_kernel(const float *__restrict__ PC, int64_t ldpc, int64_t N, int ld,
                                                       const int64_t *__restrict__ sort_idx, const uint32_t *__restrict__ pos,
                                                       const uint32_t *__restrict__ flag, const float *__restrict__ PCvox,
                                                       float m0, float m1, float m2, float vs, float *__restrict__ PCsorted,
                                                       float *__restrict__ Delta)
{
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= N * ld) return;
    const int64_t k = e / ld;
    const int c = (int)(e - k * ld);
    const float x = PC[sort_idx[k] * ldpc + c];
    if (PCsorted) PCsorted[e] = x;                                      // :103-108
    if (c < 3) {
        const float v0 = x - (c == 0 ? m0 : (c == 1 ? m1 : m2));        // :92, :103
        // :110-111: IEEE divide as torch on the CPU, and product and difference rounded SEPARATELY (two torch ops): no fma
        Delta[e] = __fsub_rn(v0, __fmul_rn(vs, floorf(__fdiv_rn(v0, vs))));
    } else {
        const int64_t v = (int64_t)pos[k] + (int64_t)flag[k] - 1;       // :129-132
        Delta[e] = x - PCvox[v * ld + c];                               // :147-148
    }
}

// The same in 16-byte row chunks (ld >= 4): a lane owns 4 consecutive columns of a row, 2^lg lanes cover a row, four row
// groups in flight per lane; the last chunk of a row is the 16 bytes that END it (overlapping columns are computed twice
// with identical results). One lane per element spent a 64-bit division per element and moved 4 bytes per instruction:
// 0.86 ms for 3 M x 59 (2.8 GB of gathers and streams), against ~0.55 ms for this form.
__global__ __launch_bounds__(256) void residual_chunk_kernel(const float *__restrict__ PC, int64_t ldpc, int64_t N, int ld, int lg,
                                                             const int64_t *__restrict__ sort_idx, const uint32_t *__restrict__ pos,
                                                             const uint32_t *__restrict__ flag, const float *__restrict__ PCvox,
                                                             float m0, float m1, float m2, float vs, float *__restrict__ PCsorted,
                                                             float *__restrict__ Delta)
{
    const int lane = threadIdx.x & 63;
    const int G = 1 << lg, rpi = 64 >> lg;
    const int g = lane >> lg, c4 = lane & (G - 1);
    const int NC = (ld + 3) >> 2;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int cc = c4; cc < NC; cc += G) {                 // one pass unless ld > 256
        const int goff = min(cc * 4, ld - 4);
        for (int64_t k0 = wave * rpi * 4; k0 < N; k0 += nwaves * rpi * 4) {
            int64_t k[4];
            RegChunk<float> x[4], pv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                k[u] = min(k0 + u * rpi + g, N - 1);
                const int64_t r = sort_idx[k[u]];
                x[u] = ld_chunk<float, true>(PC + r * ldpc + goff);
                if (goff + 3 >= 3 && PCvox) {
                    const int64_t v = (int64_t)pos[k[u]] + (int64_t)flag[k[u]] - 1;       // :129-132
                    pv[u] = ld_chunk<float>(PCvox + v * ld + goff);
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (k0 + u * rpi + g >= N) continue;
                if (PCsorted) st_chunk<float, true>(PCsorted + k[u] * ld + goff, x[u]);   // :103-108
                RegChunk<float> dl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = goff + i;
                    if (c < 3) {
                        const float v0 = x[u].v[i] - (c == 0 ? m0 : (c == 1 ? m1 : m2));   // :92, :103
                        dl.v[i] = __fsub_rn(v0, __fmul_rn(vs, floorf(__fdiv_rn(v0, vs))));  // :110-111 (no fma: two torch ops)
                    } else {
                        dl.v[i] = x[u].v[i] - pv[u].v[i];                                 // :147-148
                    }
                }
                st_chunk<float, true>(Delta + k[u] * ld + goff, dl);
            }
        }
    }
}

__global__ void u32_to_i64_kernel(const uint32_t *__restrict__ in, int64_t n, int64_t *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (int64_t)in[i];
}

// Per-voxel attribute mean. A wave takes 16 voxels per iteration: lanes fetch the 16 (+1) voxel
// starts, first-member indices and keys with vector loads, then the first-member rows are loaded 8 at
// a time (lanes = attribute columns) so that several HBM round trips overlap; further members (rare:
// most voxels hold one point) are added sequentially in sorted order -- the same order as a CPU
// scatter_add_ (voxelize_pc.py:140-144), so the float32 sums are bit-reproducible.
__global__ __launch_bounds__(256) void voxel_mean_kernel(const float *__restrict__ PC, int64_t ld, int64_t N, int d,
                                                         const uint64_t *__restrict__ keys_sorted,
                                                         const uint32_t *__restrict__ sort_idx,
                                                         const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ nvox_dev,
                                                         float *__restrict__ PCvox, int64_t *__restrict__ Vvox)
{
    // the voxel count comes from the launch before this one and the sort's error word from the launches before that: no
    // host round trip in between. A sort that gave up (never seen; the host then repeats it pass by pass) left stale
    // indices behind: nothing may be gathered through them.
    if (nvox_dev[-1] != 0u) return;
    const int64_t nvox = *nvox_dev;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int ldo = 3 + d;
    for (int64_t v0 = wave * 16; v0 < nvox; v0 += nwaves * 16) {
        const int64_t vi = v0 + lane;
        const uint32_t vs = (lane <= 16 && vi < nvox) ? vstart[vi] : (uint32_t)N;
        uint32_t first = 0, klo = 0, khi = 0;
        if (lane < 16 && vi < nvox) {
            first = sort_idx[vs];
            const uint64_t k = keys_sorted[vs];
            klo = (uint32_t)k; khi = (uint32_t)(k >> 32);
        }
        // integer voxel coordinates from the key (:152,:155)
        if (lane < 16 && vi < nvox) {
            const uint64_t key = ((uint64_t)khi << 32) | klo;
            const uint32_t x = vx_compact3(key >> 2), y = vx_compact3(key >> 1), z = vx_compact3(key);
            if (PCvox) { PCvox[vi * ldo + 0] = (float)x; PCvox[vi * ldo + 1] = (float)y; PCvox[vi * ldo + 2] = (float)z; }
            if (Vvox) { Vvox[vi * 3 + 0] = x; Vvox[vi * 3 + 1] = y; Vvox[vi * 3 + 2] = z; }
        }
        if (!PCvox || d == 0) continue;
        for (int c0 = 0; c0 < d; c0 += 64) {
            const int c = c0 + lane;
            const int cc = min(c, d - 1);
            for (int u0 = 0; u0 < 16; u0 += 8) {
                float acc[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const uint32_t i0 = (uint32_t)__builtin_amdgcn_readlane((int)first, u0 + u);
                    acc[u] = PC[(int64_t)i0 * ld + 3 + cc];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t v = v0 + u0 + u;
                    if (v >= nvox) break;                                      // wave-uniform
                    const int64_t s = (uint32_t)__builtin_amdgcn_readlane((int)vs, u0 + u);
                    const int64_t e = (uint32_t)__builtin_amdgcn_readlane((int)vs, u0 + u + 1);
                    float a = acc[u];
                    for (int64_t i = s + 1; i < e; ++i) a += PC[(int64_t)sort_idx[i] * ld + 3 + cc];
                    if (c < d) PCvox[v * ldo + 3 + c] = __fdiv_rn(a, (float)(e - s));   // :137,:144
                }
            }
        }
    }
}

// Mean AND the voxelizer's secondary outputs in one pass over the gathered rows (ld = 3 + d >= 8 columns, 16-byte chunks of
// the WHOLE row: chunk 0 = x, y, z and the first attribute). Two kernels (voxel_mean_chunk_kernel, then residual_chunk_kernel)
// gather every point's row twice and read PCvox back once per point: 5 GB on 3 M x 59 where this form moves 3.2 GB. Per voxel
// and chunk: first member's chunk (four voxel groups in flight), further members added in sorted order (:140-144), the mean
// (:137,:144; columns < 3: the voxel's integer coordinates from its key, :152,:155), then -- members of a multi-point voxel are
// gathered once more, the usual single point is still in its register -- PCsorted[k] = the row (:103-108) and DeltaPC[k] =
// row - mean, positions: V0 - voxel_size * floor(V0 / voxel_size) (:110-111, :147-156). Same operations on the same operands
// in the same order as the two kernels: bit-identical outputs.
__global__ __launch_bounds__(256) void voxel_full_chunk_kernel(const float *__restrict__ PC, int64_t ldin, int64_t N, int ld, int lg,
                                                               const uint64_t *__restrict__ keys_sorted,
                                                               const uint32_t *__restrict__ sort_idx,
                                                               const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ nvox_dev,
                                                               float *__restrict__ PCvox, int64_t *__restrict__ Vvox,
                                                               float *__restrict__ PCsorted, float *__restrict__ Delta,
                                                               float m0, float m1, float m2, float vsz)
{
    // the voxel count comes from the launch before this one and the sort's error word from the launches before that: no
    // host round trip in between. A sort that gave up (never seen; the host then repeats it pass by pass) left stale
    // indices behind: nothing may be gathered through them.
    if (nvox_dev[-1] != 0u) return;
    const int64_t nvox = *nvox_dev;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int G = 1 << lg, rpi = 64 >> lg;                    // lg >= 1: rpi <= 32
    const int U = min(4, 32 / rpi);                           // voxel groups in flight; U * rpi <= 32 voxels per iteration
    const int vpi = U * rpi;
    const int g = lane >> lg, c4 = lane & (G - 1);
    const int NC = (ld + 3) >> 2;
    for (int64_t v0 = wave * vpi; v0 < nvox; v0 += nwaves * vpi) {
        const int64_t vi = v0 + lane;
        const uint32_t vs = (lane <= vpi && vi < nvox) ? vstart[vi] : (uint32_t)N;
        uint32_t first = 0, klo = 0, khi = 0;
        if (lane < vpi && vi < nvox) {
            first = sort_idx[vs];
            const uint64_t k = keys_sorted[vs];
            klo = (uint32_t)k; khi = (uint32_t)(k >> 32);
            if (Vvox) {
                Vvox[vi * 3 + 0] = vx_compact3(k >> 2); Vvox[vi * 3 + 1] = vx_compact3(k >> 1); Vvox[vi * 3 + 2] = vx_compact3(k);
            }
        }
        // this lane's voxels: extents, first member, key. Shuffled here, with every lane active -- inside the chunk loop the
        // idle lanes of a row group are masked off and a shuffle would read garbage from them (an extent of garbage is a
        // multi-million-iteration member loop)
        uint32_t s0[4], e0[4], i0[4], kl[4], kh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sel = min(u, U - 1) * rpi + g;
            s0[u] = (uint32_t)__shfl((int)vs, sel, 64);
            e0[u] = (uint32_t)__shfl((int)vs, sel + 1, 64);
            i0[u] = (uint32_t)__shfl((int)first, sel, 64);
            kl[u] = (uint32_t)__shfl((int)klo, sel, 64);
            kh[u] = (uint32_t)__shfl((int)khi, sel, 64);
        }
        for (int cc = c4; cc < NC; cc += G) {
            const int goff = min(cc * 4, ld - 4);
            RegChunk<float> acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[u] = ld_chunk<float, true>(PC + (int64_t)i0[u] * ldin + goff);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t v = v0 + u * rpi + g;
                if (u >= U || v >= nvox) continue;
                RegChunk<float> a = acc[u];
                for (uint32_t i = s0[u] + 1; i < e0[u]; ++i) {       // further members (rare), sorted order (:140-144)
                    const RegChunk<float> b = ld_chunk<float, true>(PC + (int64_t)sort_idx[i] * ldin + goff);
#pragma unroll
                    for (int q = 0; q < 4; ++q) a.v[q] += b.v[q];
                }
                const float cnt = (float)(e0[u] - s0[u]);
#pragma unroll
                for (int q = 0; q < 4; ++q) a.v[q] = __fdiv_rn(a.v[q], cnt);                  // :137,:144
                if (goff < 3) {                                                                 // integer voxel coordinates from the key (:152,:155)
                    const uint64_t key = ((uint64_t)kh[u] << 32) | kl[u];
                    const float cx = (float)vx_compact3(key >> 2), cy = (float)vx_compact3(key >> 1), cz = (float)vx_compact3(key);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = goff + q;
                        if (c < 3) a.v[q] = (c == 0) ? cx : (c == 1 ? cy : cz);
                    }
                }
                if (PCvox) st_chunk<float, true>(PCvox + v * ld + goff, a);
                if (!PCsorted && !Delta) continue;
                for (uint32_t i = s0[u]; i < e0[u]; ++i) {
                    const RegChunk<float> x = (i == s0[u]) ? acc[u] : ld_chunk<float, true>(PC + (int64_t)sort_idx[i] * ldin + goff);
                    if (PCsorted) st_chunk<float, true>(PCsorted + (int64_t)i * ld + goff, x);  // :103-108
                    if (Delta) {
                        RegChunk<float> dl;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c = goff + q;
                            if (c < 3) {
                                const float w0 = x.v[q] - (c == 0 ? m0 : (c == 1 ? m1 : m2));    // :92, :103
                                dl.v[q] = __fsub_rn(w0, __fmul_rn(vsz, floorf(__fdiv_rn(w0, vsz))));  // :110-111 (no fma: two torch ops)
                            } else {
                                dl.v[q] = x.v[q] - a.v[q];                                      // :147-148
                            }
                        }
                        st_chunk<float, true>(Delta + (int64_t)i * ld + goff, dl);
                    }
                }
            }
        }
    }
}

// How often a one-sweep sort gave up (a tile's bounded wait for its predecessors ran out) and the call repeated the sort pass
// by pass: raht_sort_fallbacks(). Expected to stay 0; the result of the call is correct either way.
static std::atomic<int64_t> g_sort_fallbacks{0};

// Stable sort of (key, original index). `sort_err` (device word, may be NULL) selects the one-sweep form
// (scan_sort.hip: npass + 2 launches); the caller checks the word once the stream has drained and repeats the call with
// onesweep = false -- the pass-by-pass form, four launches per digit; the word is cleared -- in the never-seen case that it is set.
// `grid` (may be NULL): keys_in is an OUTPUT as well -- the keys of the cloud's points, computed on the way (inside the
// one-sweep form's histogram launch, or by vox_keys_kernel in front of the pass-by-pass form).
static int sort_keys_u32idx(const uint64_t *keys_in, int64_t N, int nbits, uint64_t *keys_out,
                                  uint32_t *idx_out, hipStream_t s, uint32_t *sort_err = nullptr, int64_t *idx64_out = nullptr,
                                  const VoxGrid *grid = nullptr, bool onesweep = true)
{
    // LSD passes of 8 bits; ping-pong between two (key, index) buffers, last pass lands in *_out
    const int npass = std::max(1, (nbits + 7) / 8);
    Scratch tmp(npass > 1 ? (sizeof(uint64_t) + sizeof(uint32_t)) * (size_t)N : 16, s);
    if (!tmp.ok()) return RAHT_ERR_NOMEM;
    uint64_t *ktmp = tmp.as<uint64_t>();
    uint32_t *itmp = (uint32_t *)(ktmp + N);
    if (sort_err) {
        const int rc1 = onesweep ? sort_pairs_onesweep(keys_in, N, nbits, keys_out, idx_out, ktmp, itmp, sort_err, s, idx64_out, grid) : 1;
        if (rc1 <= 0) return rc1;                        // enqueued (or failed); 1 = not applicable
        RAHT_HIP_CHECK(hipMemsetAsync(sort_err, 0, sizeof(uint32_t), s));
    }
    if (grid) hipLaunchKernelGGL(vox_keys_kernel, dim3((unsigned)ceil_div(N, 256 * 4)), dim3(256), 0, s, *grid, N, (uint64_t *)keys_in);
    const uint64_t *kin = keys_in;
    const uint32_t *iin = nullptr;
    int rc = RAHT_OK;
    for (int ps = 0; ps < npass && rc == RAHT_OK; ++ps) {
        const bool to_out = ((npass - 1 - ps) % 2 == 0);
        uint64_t *ko = to_out ? keys_out : ktmp;
        uint32_t *io = to_out ? idx_out : itmp;
        const int bits = std::min(8, nbits - 8 * ps);
        rc = radix_pass_u64(kin, iin, ko, io, N, 8 * ps, bits < 1 ? 1 : bits, s);
        kin = ko;
        iin = io;
    }
    if (rc == RAHT_OK && idx64_out)
        hipLaunchKernelGGL(u32_to_i64_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, idx_out, N, idx64_out);
    return rc;
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_sort_keys(const uint64_t *keys_in, int64_t N, int nbits, uint64_t *keys_out, int64_t *idx_out,
                   raht_stream_t stream)
{
    // an empty input is valid and needs no buffers (an empty device tensor's data pointer is NULL): ranks of a sharded scene
    // may hold no points, and must reach the next collective like every other rank
    if (N < 0 || nbits < 1 || nbits > 64) { set_error("raht_sort_keys: bad argument"); return RAHT_ERR_INVALID; }
    if (N == 0) return RAHT_OK;
    if (!keys_in || !keys_out) { set_error("raht_sort_keys: NULL argument"); return RAHT_ERR_INVALID; }
    if (N >= ((int64_t)1 << 31)) { set_error("raht_sort_keys: N too large"); return RAHT_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    Scratch ib(sizeof(uint32_t) * ((size_t)N + 1), s);
    if (!ib.ok()) return RAHT_ERR_NOMEM;
    uint32_t *idx32 = ib.as<uint32_t>(), *sort_err = idx32 + N;
    for (int attempt = 0; attempt < 2; ++attempt) {
        RAHT_RET(sort_keys_u32idx(keys_in, N, nbits, keys_out, idx32, s, sort_err, idx_out, nullptr, attempt == 0));
        RAHT_HIP_CHECK(hipGetLastError());
        // idx32 returns to the pool when this frame ends: the read-back of the sort's error word is also the wait for the stream
        uint32_t bad = 0;
        RAHT_RET(read_back_u32(&bad, sort_err, 1, nullptr, nullptr, 0, s));
        if (!bad) break;
        g_sort_fallbacks.fetch_add(1, std::memory_order_relaxed);
    }
    return RAHT_OK;
}

int64_t raht_sort_fallbacks(void) { return g_sort_fallbacks.load(std::memory_order_relaxed); }

int raht_voxel_keys(const float *PC, int64_t ldpc, int64_t N, const float vmin[3], double width, int J,
                    uint64_t *keys, raht_stream_t stream)
{
    if (!vmin || N < 0 || J < 1 || J > 21 || !(width > 0)) { set_error("raht_voxel_keys: bad argument"); return RAHT_ERR_INVALID; }
    if (N == 0) return RAHT_OK;                              // empty rank of a sharded cloud: nothing to do; an empty tensor has
                                                             // neither a data pointer nor meaningful strides
    if (!PC || !keys || ldpc < 3) { set_error("raht_voxel_keys: bad argument"); return RAHT_ERR_INVALID; }
    const float vs = (float)(width / (double)((uint64_t)1 << J));        // voxelize_pc.py:97, as raht_voxelize
    const VoxGrid G = {PC, ldpc, vmin[0], vmin[1], vmin[2], vs, J};
    hipLaunchKernelGGL(vox_keys_kernel, dim3((unsigned)ceil_div(N, 256 * 4)), dim3(256), 0, (hipStream_t)stream, G, N, keys);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

int raht_voxelize_residuals(const float *PC, int64_t ldpc, int64_t N, int d, const uint64_t *keys_sorted,
                            const int64_t *sort_idx, const float *PCvox, const float vmin[3], double voxel_size,
                            float *PCsorted, float *DeltaPC, raht_stream_t stream)
{
    if (!PC || !keys_sorted || !sort_idx || !vmin || !DeltaPC || N < 1 || d < 0 || ldpc < 3 + d || !(voxel_size > 0) || (d > 0 && !PCvox)) {
        set_error("raht_voxelize_residuals: bad argument");
        return RAHT_ERR_INVALID;
    }
    if (N >= ((int64_t)1 << 31)) { set_error("raht_voxelize_residuals: N too large"); return RAHT_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    Scratch buf(sizeof(uint32_t) * 2 * (size_t)N, s);
    if (!buf.ok()) return RAHT_ERR_NOMEM;
    uint32_t *flag = buf.as<uint32_t>(), *pos = flag + N;
    hipLaunchKernelGGL(boundary_kernel, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, s, keys_sorted, N, flag);
    RAHT_RET(exclusive_scan_u32(flag, pos, N, nullptr, s));
    const int ld = 3 + d;
    if (ld >= 4) {
        int lg = 0;
        while ((1 << lg) < (ld + 3) / 4 && lg < 6) ++lg;
        const int64_t rows_per_block = (int64_t)4 * 4 * (64 >> lg);            // 4 waves x 4 row groups in flight
        const unsigned gb = (unsigned)std::min<int64_t>(ceil_div(N, rows_per_block), 16384);
        hipLaunchKernelGGL(residual_chunk_kernel, dim3(gb), dim3(256), 0, s, PC, ldpc, N, ld, lg, sort_idx, pos, flag,
                           PCvox, vmin[0], vmin[1], vmin[2], (float)voxel_size, PCsorted, DeltaPC);
    } else
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)ceil_div(N * ld, 256)), dim3(256), 0, s, PC, ldpc, N, ld, sort_idx, pos, flag,
                       PCvox, vmin[0], vmin[1], vmin[2], (float)voxel_size, PCsorted, DeltaPC);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

// VOXELIZE + MERGE in one pass over the gathered rows ("fused into the voxelize dedup"). The reference runs the
// voxelizer and then its CUDA merge kernel back to back (python/test_voxelize_3dgs.py:203-257; cuda/merge_cluster.cu:2-111): the
// sort permutation and the voxel starts ARE the cluster indices / offsets (:225-233). Here the kernel that walks every voxel's
// member rows merges them on the way: rows are whole Gaussians [xyz(3) | quat(4) | scale(3) | opacity(1) | colour(cd)], weight =
// opacity, and per column exactly the arithmetic of merge.hip / merge_cluster.cu -- members in sorted (= index) order, acc =
// fma(x, w, acc), means / scales acc / (tw == 0 ? 1 : tw), quaternion acc / |acc| (identity if 0), opacity min(sum, 1), colours
// tw > 0 ? acc / tw : 0 -- so the result is bit-identical to raht_voxelize followed by raht_merge_clusters. Output rows: the
// voxel's integer coordinates (from its key, as PCvox carries them, voxelize_pc.py:152,155) and the merged attributes; the
// merged means optionally on their own. One read of every member row instead of two gathers (mean pass + merge pass over five
// separate arrays).
__global__ __launch_bounds__(256) void voxel_merge_chunk_kernel(const float *__restrict__ PC, int64_t ldin, int64_t N, int ld, int lg,
                                                                const uint64_t *__restrict__ keys_sorted, const uint32_t *__restrict__ sort_idx,
                                                                const uint32_t *__restrict__ vstart, const uint32_t *__restrict__ nvox_dev,
                                                                float *__restrict__ Gvox, float *__restrict__ merged_means, int weight_by_opacity)
{
    if (nvox_dev[-1] != 0u) return;                           // (a sort that gave up left stale indices: see voxel_full_chunk_kernel)
    const int64_t nvox = *nvox_dev;
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int G = 1 << lg, rpi = 64 >> lg;
    const int U = min(4, 32 / rpi);
    const int vpi = U * rpi;
    const int g = lane >> lg, c4 = lane & (G - 1);
    const int NC = (ld + 3) >> 2;
    for (int64_t v0 = wave * vpi; v0 < nvox; v0 += nwaves * vpi) {
        const int64_t vi = v0 + lane;
        const uint32_t vs = (lane <= vpi && vi < nvox) ? vstart[vi] : (uint32_t)N;
        uint32_t first = 0, klo = 0, khi = 0;
        if (lane < vpi && vi < nvox) {
            first = sort_idx[vs];
            const uint64_t k = keys_sorted[vs];
            klo = (uint32_t)k; khi = (uint32_t)(k >> 32);
        }
        uint32_t s0[4], e0[4], i0[4], kl[4], kh[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sel = min(u, U - 1) * rpi + g;
            s0[u] = (uint32_t)__shfl((int)vs, sel, 64);
            e0[u] = (uint32_t)__shfl((int)vs, sel + 1, 64);
            i0[u] = (uint32_t)__shfl((int)first, sel, 64);
            kl[u] = (uint32_t)__shfl((int)klo, sel, 64);
            kh[u] = (uint32_t)__shfl((int)khi, sel, 64);
        }
        for (int cb = 0; cb < NC; cb += G) {                   // (every lane walks every block of chunks: the shuffles below need them all)
            const int cc = cb + c4;
            const bool act = cc < NC;
            const int goff = min(min(cc, NC - 1) * 4, ld - 4);
            RegChunk<float> x0[4];
            float w0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x0[u] = ld_chunk<float, true>(PC + (int64_t)i0[u] * ldin + goff);
                w0[u] = weight_by_opacity ? PC[(int64_t)i0[u] * ldin + 10] : 1.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int64_t v = v0 + u * rpi + g;
                const bool live = u < U && v < nvox;
                float tw = w0[u];
                RegChunk<float> acc;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc.v[q] = (goff + q == 10) ? x0[u].v[q] : __fmaf_rn(x0[u].v[q], w0[u], 0.0f);     // merge_cluster.cu:38-63
                if (live) for (uint32_t i = s0[u] + 1; i < e0[u]; ++i) {      // further members, sorted (= index) order
                    const int64_t r = (int64_t)sort_idx[i] * ldin;
                    const RegChunk<float> b = ld_chunk<float, true>(PC + r + goff);
                    const float w = weight_by_opacity ? PC[r + 10] : 1.0f;
                    tw += w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc.v[q] = (goff + q == 10) ? acc.v[q] + b.v[q] : __fmaf_rn(b.v[q], w, acc.v[q]);
                }
                // quaternion norm (merge_cluster.cu:76-78): columns 3 .. 6 sit in chunks 0 and 1 of this row group's first block
                float n2 = 0.0f;
                if (cb == 0) {
                    const int l0 = (g << lg), l1 = (g << lg) + 1;
                    const float qx = __shfl(acc.v[3], l0, 64), qy = __shfl(acc.v[0], l1, 64), qz = __shfl(acc.v[1], l1, 64), qw = __shfl(acc.v[2], l1, 64);
                    n2 = qx * qx;
                    n2 = __fmaf_rn(qy, qy, n2);
                    n2 = __fmaf_rn(qz, qz, n2);
                    n2 = __fmaf_rn(qw, qw, n2);
                }
                if (!live || !act) continue;
                RegChunk<float> r;
                float mm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = goff + q;
                    float y;
                    if (c < 3 || (c >= 7 && c < 10)) y = __fdiv_rn(acc.v[q], tw == 0.0f ? 1.0f : tw);              // :66-73, :91-93
                    else if (c < 7) { const float nrm = __fsqrt_rn(n2); y = (nrm > 0.0f) ? __fdiv_rn(acc.v[q], nrm) : (c == 6 ? 1.0f : 0.0f); }   // :76-89
                    else if (c == 10) y = fminf(acc.v[q], 1.0f);                                                   // :96
                    else y = (tw > 0.0f) ? __fdiv_rn(acc.v[q], tw) : 0.0f;                                         // :98-110
                    mm[q] = y;
                    r.v[q] = y;
                }
                if (goff < 3) {
                    const uint64_t key = ((uint64_t)kh[u] << 32) | kl[u];
                    const float cx = (float)vx_compact3(key >> 2), cy = (float)vx_compact3(key >> 1), cz = (float)vx_compact3(key);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = goff + q;
                        if (c < 3) { if (merged_means) merged_means[v * 3 + c] = mm[q]; r.v[q] = (c == 0) ? cx : (c == 1 ? cy : cz); }
                    }
                }
                st_chunk<float, true>(Gvox + v * ld + goff, r);
            }
        }
    }
}

static int voxelize_impl(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                         int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                         int64_t *Vvox, int64_t *n_vox, float vmin_out[3], double *width_out,
                         double *voxel_size_out, raht_stream_t stream, uint64_t *voxel_keys,
                         float *PCsorted = nullptr, float *DeltaPC = nullptr, hipEvent_t keys_ready = nullptr,
                         float *merge_means = nullptr, int merge = 0 /* 1: weight = opacity, 2: weight = 1 */)
{
    if (!PC || N < 1 || d < 0 || ldpc < 3 + d || J < 1 || J > 21 || !n_vox) { set_error("raht_voxelize: bad argument"); return RAHT_ERR_INVALID; }
    if (N >= ((int64_t)1 << 31)) { set_error("raht_voxelize: N too large"); return RAHT_ERR_INVALID; }
    hipStream_t s = (hipStream_t)stream;
    float vmin[3];
    double width = width_in;
    // ---- vmin / width (voxelize_pc.py:87-95) ----
    const int nb = (int)std::min<int64_t>(ceil_div(N, 256), 1024);
    Scratch partb(sizeof(float) * 4 * (size_t)nb, s);
    if (!partb.ok()) return RAHT_ERR_NOMEM;
    float *part = partb.as<float>();
    float hp_buf[1024 * 4];                          // nb <= 1024 partial results (no host allocation: nothing can throw)
    struct { float *p; size_t n; float *data() { return p; } size_t size() const { return n; } float &operator[](size_t i) { return p[i]; } } hp = {hp_buf, (size_t)nb * 4};
    if (vmin_in) { vmin[0] = vmin_in[0]; vmin[1] = vmin_in[1]; vmin[2] = vmin_in[2]; }
    else {
        hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, s, PC, ldpc, N, 0.f, 0.f, 0.f, part);
        RAHT_HIP_CHECK(hipMemcpyAsync(hp.data(), part, sizeof(float) * hp.size(), hipMemcpyDeviceToHost, s));
        RAHT_HIP_CHECK(hipStreamSynchronize(s));
        for (int a = 0; a < 3; ++a) {
            float m = hp[(size_t)a];
            for (int b = 1; b < nb; ++b) m = std::fmin(m, hp[(size_t)b * 4 + a]);
            vmin[a] = m;
        }
    }
    if (width_in < 0) {
        hipLaunchKernelGGL(minmax_kernel, dim3(nb), dim3(256), 0, s, PC, ldpc, N, vmin[0], vmin[1], vmin[2], part);
        RAHT_HIP_CHECK(hipMemcpyAsync(hp.data(), part, sizeof(float) * hp.size(), hipMemcpyDeviceToHost, s));
        RAHT_HIP_CHECK(hipStreamSynchronize(s));
        float m = hp[3];
        for (int b = 1; b < nb; ++b) m = std::fmax(m, hp[(size_t)b * 4 + 3]);
        width = (double)m;
    }
    if (!(width > 0)) { set_error("raht_voxelize: width must be > 0 (got %g)", width); return RAHT_ERR_INVALID; }
    const double voxel_size = width / (double)((uint64_t)1 << J);   // :97
    const float vs = (float)voxel_size;

    // ---- keys, sort ----
    Scratch kb(sizeof(uint64_t) * 2 * (size_t)N, s), ib(sizeof(uint32_t) * (2 * (size_t)N + 2), s);
    if (!kb.ok() || !ib.ok()) return RAHT_ERR_NOMEM;
    uint64_t *keys = kb.as<uint64_t>();
    uint64_t *ks = keys_sorted ? keys_sorted : keys + N;
    uint32_t *idx = ib.as<uint32_t>(), *vstart = idx + N, *sort_err = vstart + N, *nv_dev = sort_err + 1;      // (the voxel kernels read nv_dev[-1])
    int64_t nv = 0;
    const bool want_res = PCsorted || DeltaPC;
    const bool fused = PCvox && d >= 5;                                 // whole rows in 16-byte chunks: 3 + d >= 8 columns
    if (want_res && !fused && !sort_idx) { set_error("raht_voxelize: residuals of a cloud with fewer than 5 attribute columns need sort_idx"); return RAHT_ERR_INVALID; }
    {
        // ONE host round trip per call, at the end: keys + histogram, digit bases, the sort's passes, voxel starts (count on the
        // device), means / residuals are enqueued back to back; the read-back of the count and of the sort's error word is the wait
        const VoxGrid G = {PC, ldpc, vmin[0], vmin[1], vmin[2], vs, J};
        for (int attempt = 0; attempt < 2; ++attempt) {
            RAHT_RET(sort_keys_u32idx(keys, N, 3 * J, ks, idx, s, sort_err, sort_idx, &G, attempt == 0));
            RAHT_RET(run_starts_u64(ks, N, vstart, voxel_indices, voxel_keys, nv_dev, s));
            if (keys_ready) {
                // raht_voxelize_plan: the voxel count comes back NOW, and the event marks the voxel keys as complete -- the plan is
                // then built from them on a second stream while this one forms the means (the read-back at the end is skipped)
                RAHT_HIP_CHECK(hipGetLastError());
                uint32_t back[2] = {0, 0};
                RAHT_RET(read_back_u32(back, sort_err, 2, nullptr, nullptr, 0, s));
                nv = back[1];
                if (back[0]) { g_sort_fallbacks.fetch_add(1, std::memory_order_relaxed); continue; }
                RAHT_HIP_CHECK(hipEventRecord(keys_ready, s));
            }
            if (PCvox || Vvox || want_res) {
                const unsigned gv = (unsigned)std::min<int64_t>(ceil_div(N, 64), 8192);        // (N >= the voxel count)
                if (merge) {
                    int lg = 1;
                    while ((1 << lg) < (3 + d + 3) / 4 && lg < 4) ++lg;           // at most 16 lanes per row: rows wider than 64 columns take several blocks of chunks
                    hipLaunchKernelGGL(voxel_merge_chunk_kernel, dim3(gv), dim3(256), 0, s, PC, ldpc, N, 3 + d, lg, ks, idx, vstart, nv_dev, PCvox, merge_means, merge == 1 ? 1 : 0);
                } else if (fused) {
                    int lg = 1;
                    while ((1 << lg) < (3 + d + 3) / 4 && lg < 6) ++lg;
                    hipLaunchKernelGGL(voxel_full_chunk_kernel, dim3(gv), dim3(256), 0, s, PC, ldpc, N, 3 + d, lg, ks, idx, vstart, nv_dev, PCvox, Vvox,
                                       PCsorted, DeltaPC, vmin[0], vmin[1], vmin[2], vs);
                } else {
                    hipLaunchKernelGGL(voxel_mean_kernel, dim3(gv), dim3(256), 0, s, PC, ldpc, N, d, ks, idx, vstart, nv_dev, PCvox, Vvox);
                }
            }
            RAHT_HIP_CHECK(hipGetLastError());
            if (keys_ready) break;
            uint32_t back[2] = {0, 0};                       // { sort error, voxel count }: adjacent device words
            RAHT_RET(read_back_u32(back, sort_err, 2, nullptr, nullptr, 0, s));
            nv = back[1];
            if (!back[0]) break;                             // (else: once more with the pass-by-pass sort)
            g_sort_fallbacks.fetch_add(1, std::memory_order_relaxed);
        }
        if (want_res && !fused) {
            // narrow clouds: the two-call sequence (sort_idx as int64 is what raht_voxelize_residuals takes)
            if (DeltaPC) RAHT_RET(raht_voxelize_residuals(PC, ldpc, N, d, ks, sort_idx, PCvox, vmin, voxel_size, PCsorted, DeltaPC, stream));
            else RAHT_RET(raht_rows_gather(PC, ldpc, sort_idx, N, 3 + d, 4, PCsorted, 3 + d, stream));
            hipError_t e = hipStreamSynchronize(s);
            if (e != hipSuccess) { set_error("raht_voxelize: %s", hipGetErrorString(e)); return RAHT_ERR_HIP; }
        }
    }
    *n_vox = nv;
    if (vmin_out) { vmin_out[0] = vmin[0]; vmin_out[1] = vmin[1]; vmin_out[2] = vmin[2]; }
    if (width_out) *width_out = width;
    if (voxel_size_out) *voxel_size_out = voxel_size;
    return RAHT_OK;
}

int raht_voxelize(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                  int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                  int64_t *Vvox, int64_t *n_vox, float vmin_out[3], double *width_out,
                  double *voxel_size_out, raht_stream_t stream)
{
    return voxelize_impl(PC, ldpc, N, d, vmin_in, width_in, J, keys_sorted, sort_idx, voxel_indices, PCvox, Vvox, n_vox, vmin_out,
                         width_out, voxel_size_out, stream, nullptr);
}

int raht_voxelize_merge(const float *G, int64_t ldg, int64_t N, int color_dim, int weight_by_opacity, const float *vmin_in, double width_in, int J,
                        uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *Gvox, float *merged_means, int64_t *n_vox,
                        float vmin_out[3], double *width_out, double *voxel_size_out, raht_stream_t stream)
{
    if (color_dim < 0 || !Gvox) { set_error("raht_voxelize_merge: bad argument"); return RAHT_ERR_INVALID; }
    return voxelize_impl(G, ldg, N, 8 + color_dim, vmin_in, width_in, J, keys_sorted, sort_idx, voxel_indices, Gvox, nullptr, n_vox, vmin_out, width_out,
                         voxel_size_out, stream, nullptr, nullptr, nullptr, nullptr, merged_means, weight_by_opacity ? 1 : 2);
}

int raht_voxelize_all(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                      int J, uint64_t *keys_sorted, int64_t *sort_idx, int64_t *voxel_indices, float *PCvox,
                      int64_t *Vvox, float *PCsorted, float *DeltaPC, int64_t *n_vox, float vmin_out[3], double *width_out,
                      double *voxel_size_out, raht_stream_t stream)
{
    if (DeltaPC && d > 0 && !PCvox) { set_error("raht_voxelize_all: DeltaPC needs PCvox"); return RAHT_ERR_INVALID; }
    return voxelize_impl(PC, ldpc, N, d, vmin_in, width_in, J, keys_sorted, sort_idx, voxel_indices, PCvox, Vvox, n_vox, vmin_out,
                         width_out, voxel_size_out, stream, nullptr, PCsorted, DeltaPC);
}

int raht_voxelize_plan(const float *PC, int64_t ldpc, int64_t N, int d, const float *vmin_in, double width_in,
                       int J, uint64_t *voxel_keys, int64_t *voxel_indices, float *PCvox, int64_t *n_vox,
                       float vmin_out[3], double *width_out, double *voxel_size_out, raht_stream_t stream, raht_plan **plan)
{
    if (!voxel_keys || !plan) { set_error("raht_voxelize_plan: NULL argument"); return RAHT_ERR_INVALID; }
    *plan = nullptr;
    int64_t nv = 0;
    // The plan needs the voxels' sorted keys only, the per-voxel means (the HBM-bound half of the voxelizer: a gather of every
    // point's row) need the plan not at all: the voxelizer reads the voxel count back as soon as the keys are there, enqueues the
    // mean kernel on the caller's stream and returns; the plan is built on a side stream meanwhile (a chain of small,
    // latency-bound launches with a host round trip of its own); the caller's stream then waits for it.
    // MEASURED, OFF by default (RAHT_VOXPLAN_OVERLAP=1 switches it on): 0.847 ms per cfg3 frame against 0.825 ms on one stream --
    // next to a kernel that keeps every CU busy the plan's dozen small launches get their workgroup slots late and stretch from
    // 0.2 to ~0.35 ms, which is the mean kernel's own length: nothing is hidden.
    static const bool serial = !(getenv("RAHT_VOXPLAN_OVERLAP") && atoi(getenv("RAHT_VOXPLAN_OVERLAP")) != 0);
    static hipStream_t side[RAHT_MAX_DEVICES] = {};
    static hipEvent_t ev_keys[RAHT_MAX_DEVICES] = {}, ev_plan[RAHT_MAX_DEVICES] = {};
    const int dev = current_device();
    hipStream_t s = (hipStream_t)stream;
    if (!serial && !side[dev]) {
        RAHT_HIP_CHECK(hipStreamCreateWithFlags(&side[dev], hipStreamNonBlocking));
        RAHT_HIP_CHECK(hipEventCreateWithFlags(&ev_keys[dev], hipEventDisableTiming));
        RAHT_HIP_CHECK(hipEventCreateWithFlags(&ev_plan[dev], hipEventDisableTiming));
    }
    RAHT_RET(voxelize_impl(PC, ldpc, N, d, vmin_in, width_in, J, nullptr, nullptr, voxel_indices, PCvox, nullptr, &nv, vmin_out,
                           width_out, voxel_size_out, stream, voxel_keys, nullptr, nullptr, serial ? nullptr : ev_keys[dev]));
    if (n_vox) *n_vox = nv;
    // the voxels' keys are sorted and unique by construction; the plan build checks them anyway (it reads them to find the
    // levels) and BORROWS the caller's array
    if (serial) return raht_plan_create_from_keys_borrowed(voxel_keys, nv, 3 * J, nullptr, stream, plan);
    RAHT_HIP_CHECK(hipStreamWaitEvent(side[dev], ev_keys[dev], 0));
    const int rc = raht_plan_create_from_keys_borrowed(voxel_keys, nv, 3 * J, nullptr, (raht_stream_t)side[dev], plan);
    RAHT_HIP_CHECK(hipEventRecord(ev_plan[dev], side[dev]));
    RAHT_HIP_CHECK(hipStreamWaitEvent(s, ev_plan[dev], 0));     // (also after a failed build: whatever it enqueued is ordered before the caller's next work)
    return rc;
}

}  // extern "C"
