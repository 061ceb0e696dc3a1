// raht_device.h -- device-side helpers shared by the HIP sources: 16-byte row chunks and the
// quantizer's arithmetic.
#pragma once
#include "raht_common.h"

namespace raht {

constexpr int MAX_STEP_CH = 256;
struct StepTable {
    int n;                         // 0 = no quantization, 1 = one step, D = per channel
    int fast_div;                  // every step within [2^-100, 2^100]: the forward may divide without range scaling
    float v[MAX_STEP_CH];
};
struct NoSteps { int n; int fast_div; };
// float64 steps (the reference's own precision, python/encode_3dgs.py:82-83,204): always the IEEE double division
struct StepTable64 { int n; double v[MAX_STEP_CH]; };
template <typename T> struct StepsFor;
template <> struct StepsFor<float> { typedef StepTable type; typedef float elem; };
template <> struct StepsFor<double> { typedef StepTable64 type; typedef double elem; };


// ---- row chunks: one lane moves 16 bytes (VN = 16 / sizeof(T) consecutive channels) of one row --------
// Registers / LDS (16-byte aligned: ds_read_b128 / ds_write_b128) ...
template <typename E> struct alignas(16) RegChunk { E v[16 / sizeof(E)]; };
// ... and global memory, where a row starts on an element boundary only (59 channels: 236-byte rows).
// gfx950 global loads / stores of 8..16 bytes need element alignment only.
template <typename E> struct __attribute__((packed, aligned(sizeof(E)))) MemChunk { E v[16 / sizeof(E)]; };

// A chunk is always a whole 16 bytes, in global memory too: when a row's length is not a multiple of
// VN, its LAST chunk is the 16 bytes that END the row (channels [Dc - VN, Dc)), i.e. it overlaps its
// neighbour by VN - Dc % VN channels. The overlapped channels live twice in LDS, go through the same
// butterflies with the same operands in both copies, and are written back twice with identical
// values. That keeps every load and store a plain, unpredicated 16-byte access (a load inside a
// divergent branch costs an exec-mask region with its own s_waitcnt, i.e. one serialised HBM round
// trip per chunk; masks and shifts cost VALU issue slots, which is what bounds this kernel). The host
// only runs the tile kernel on channel chunks of at least VN channels (plan.hip: fit_chunk_channels).
// STREAM = true marks the once-touched matrices (C, T, Q): nontemporal loads / stores (`nt`), measured
// +1.5 % on the fused cfg3 step when applied to both directions of the big streams (loads alone: -3 %).
// Workspace rows, which the next stage re-reads from L2, keep the default policy.
template <typename E, bool STREAM = false>
__device__ __forceinline__ RegChunk<E> ld_chunk(const E *__restrict__ p)
{
    constexpr int VN = 16 / sizeof(E);
    RegChunk<E> x;
    if constexpr (STREAM) {
#pragma unroll
        for (int i = 0; i < VN; ++i) x.v[i] = __builtin_nontemporal_load(p + i);      // one global_load_dwordx4 ... nt
    } else {
        const MemChunk<E> t = *(const MemChunk<E> *)p;
#pragma unroll
        for (int i = 0; i < VN; ++i) x.v[i] = t.v[i];
    }
    return x;
}

template <typename E, bool STREAM = false>
__device__ __forceinline__ void st_chunk(E *__restrict__ p, const RegChunk<E> &x)
{
    constexpr int VN = 16 / sizeof(E);
    if constexpr (STREAM) {
#pragma unroll
        for (int i = 0; i < VN; ++i) __builtin_nontemporal_store(x.v[i], p + i);
    } else {
        MemChunk<E> t;
#pragma unroll
        for (int i = 0; i < VN; ++i) t.v[i] = x.v[i];
        *(MemChunk<E> *)p = t;
    }
}

// VN quantized integers of one row chunk (float64 rows: 2 per 16-byte chunk), element-aligned in global memory
template <int VN> struct __attribute__((packed, aligned(4))) IntPack { int32_t v[VN]; };
template <int VN>
__device__ __forceinline__ void ld_ints(const int32_t *__restrict__ p, int32_t (&v)[VN])
{
    const IntPack<VN> t = *(const IntPack<VN> *)p;
#pragma unroll
    for (int i = 0; i < VN; ++i) v[i] = t.v[i];
}
template <int VN>
__device__ __forceinline__ void st_ints(int32_t *__restrict__ p, const int32_t (&v)[VN])
{
    IntPack<VN> t;
#pragma unroll
    for (int i = 0; i < VN; ++i) t.v[i] = v[i];
    *(IntPack<VN> *)p = t;
}
__device__ __forceinline__ int32_t quantize_one_f64(double x, double step) { return (int32_t)floor(x / step + 0.5); }   // encode_3dgs.py:204

__device__ __forceinline__ int32_t quantize_one(float x, float sp, float r, int fast_div)
{
    float q;
    if (fast_div) {
        // x / step, correctly rounded: the quotient refinement of hipcc's float division (mul, 4 fma)
        // without its range scaling and special-case fixup, which the host has ruled out (steps
        // within [2^-100, 2^100]); r is the refined reciprocal of the step (refined_rcp)
        const float q0 = x * r;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-sp, q0, x), r, q0);
        q = __builtin_fmaf(__builtin_fmaf(-sp, q1, x), r, q1);
    } else {
        q = x / sp;
    }
    return (int32_t)floorf(q + 0.5f);                     // encode_3dgs.py:204
}

// v_rcp_f32 + one Newton step: exactly how hipcc's own float division refines 1 / step per quotient
__device__ __forceinline__ float refined_rcp(float sp)
{
    const float r0 = __builtin_amdgcn_rcpf(sp);
    return __builtin_fmaf(__builtin_fmaf(-sp, r0, 1.0f), r0, r0);
}


// ---- voxel keys (voxelize.hip; the key sort's histogram launch can compute them on the way: scan_sort.hip) ----------
__device__ __forceinline__ uint64_t vx_spread3(uint64_t v)
{
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x001f00000000ffffull;
    v = (v | (v << 16)) & 0x001f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}
// xyz of one point: 12 bytes at the start of a (3 + d)-float row, as ONE global_load_dwordx3 (element alignment)
struct __attribute__((packed, aligned(4))) Xyz { float x, y, z; };
// the cloud and its grid (voxelize_pc.py:87-98)
struct VoxGrid {
    const float *PC;
    int64_t ld;
    float m0, m1, m2, vs;
    int J;
};
// Morton key of the voxel of one point (voxelize_pc.py:92, 98, 100): shift, IEEE divide, floor, clamp, interleave
__device__ __forceinline__ uint64_t vox_key(const Xyz &p, const VoxGrid &g)
{
    const int64_t hi = ((int64_t)1 << g.J) - 1;
    const float v[3] = {p.x - g.m0, p.y - g.m1, p.z - g.m2};          // voxelize_pc.py:92
    int64_t q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        int64_t t = (int64_t)floorf(__fdiv_rn(v[a], g.vs));           // :98 (IEEE divide, not rcp*mul)
        q[a] = t < 0 ? 0 : (t > hi ? hi : t);
    }
    return vx_spread3((uint64_t)q[2]) | (vx_spread3((uint64_t)q[1]) << 1) | (vx_spread3((uint64_t)q[0]) << 2);
}

// StepTable from the caller's steps (host)
inline void fill_step_table(StepTable64 &t, const double *steps, int n_steps)
{
    t.n = n_steps;
    for (int c = 0; c < n_steps; ++c) t.v[c] = steps[c];
}
inline void fill_step_table(StepTable &t, const float *steps, int n_steps)
{
    t.n = n_steps;
    t.fast_div = 1;
    for (int c = 0; c < n_steps; ++c) {
        t.v[c] = steps[c];
        if (!(steps[c] >= 0x1p-100f && steps[c] <= 0x1p100f)) t.fast_div = 0;
    }
}

}  // namespace raht
