// Probe: global_load_lds_dwordx4 on gfx950 -- dword-aligned (not 16-byte aligned) global addresses, EXEC-masked lanes,
// M0 as the wave's LDS base. Build: hipcc --offload-arch=gfx950 -O2 tools/probes/glds_probe.hip -o gpurun_out/glds_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ __align__(16) unsigned char smem[];

__device__ __forceinline__ void glds16(const void *g, uint32_t lds_base)
{
    const uint32_t b = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(g), "s"(b) : "memory");
}

// every wave gathers 64 16-byte chunks: chunk c = 15 per row of 59 floats (row stride 59 floats = 236 bytes: only
// dword-aligned), LDS rows of 60 floats, lane-linear; lanes with (c % 7 == 3) are masked off and must keep the fill value
__global__ void probe(const float *src, float *dst, int nrows, int lds_off)
{
    float *tile = (float *)(smem + lds_off);          // (also beyond 64 KiB: M0 must carry the full LDS address)
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int i = threadIdx.x; i < nrows * 60; i += blockDim.x) tile[i] = -1.0f;
    __syncthreads();
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)tile;
    const int total = nrows * 15;
    for (int it = wid; it * 64 < total; it += nw) {
        const int c = it * 64 + lane;
        if (c < total && (c % 7) != 3) {
            const int row = c / 15, ch = c % 15;
            const int goff = min(ch * 4, 59 - 4);
            glds16(src + (size_t)row * 59 + goff, lds0 + (uint32_t)(it * 1024));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < nrows * 60; i += blockDim.x) dst[i] = tile[i];
}

int main()
{
    const int nrows = 184;
    std::vector<float> h((size_t)nrows * 59 + 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)i;
    float *src, *dst;
    if (hipMalloc(&src, h.size() * 4) != hipSuccess || hipMalloc(&dst, nrows * 60 * 4) != hipSuccess) return 3;
    if (hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice) != hipSuccess) return 3;
    int bad = 0;
    if (hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return 3;
    const int offs[3] = {0, 60 * 1024, 112 * 1024};
    for (int t = 0; t < 3; ++t) {
    hipMemset(dst, 0, nrows * 60 * 4);
    hipLaunchKernelGGL(probe, dim3(1), dim3(512), offs[t] + 48 * 1024, 0, src, dst, nrows, offs[t]);
    std::vector<float> o((size_t)nrows * 60);
    if (hipMemcpy(o.data(), dst, o.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 2; }
    const int bad0 = bad;
    for (int c = 0; c < nrows * 15; ++c) {
        const int row = c / 15, ch = c % 15, goff = ch * 4 < 55 ? ch * 4 : 55;
        for (int i = 0; i < 4; ++i) {
            const float want = (c % 7) != 3 ? h[(size_t)row * 59 + goff + i] : -1.0f;
            const float got = o[(size_t)row * 60 + ch * 4 + i];
            if (want != got) { if (bad < 10) printf("chunk %d elem %d: want %g got %g\n", c, i, want, got); ++bad; }
        }
    }
    printf("glds probe, tile at LDS byte %d: %d mismatches of %d\n", offs[t], bad - bad0, nrows * 60);
    }
    return bad ? 1 : 0;
}
