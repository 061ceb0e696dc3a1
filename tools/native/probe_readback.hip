// How long does the host wait for one 4-byte count produced by a kernel?
//   a) hipMemcpyAsync to pageable memory + hipStreamSynchronize   (what plan.hip did)
//   b) hipMemcpyAsync to pinned memory + hipStreamSynchronize
//   c) the kernel stores into mapped pinned memory (system-scope release), the host polls
// build: hipcc --offload-arch=gfx950 -O2 -o probe_readback probe_readback.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <immintrin.h>

__global__ void produce(uint32_t *d, uint32_t v, int spin)
{
    uint32_t x = v;
    for (int i = 0; i < spin; ++i) x = x * 1664525u + 1013904223u;
    if (threadIdx.x == 0) d[0] = v + (x & 0u);
}
__global__ void publish(const uint32_t *d, volatile uint32_t *host_val, volatile uint32_t *host_flag, uint32_t seq)
{
    host_val[0] = d[0];
    __threadfence_system();
    host_flag[0] = seq;
}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
    hipStream_t s; hipStreamCreate(&s);
    uint32_t *d; hipMalloc(&d, 64);
    uint32_t *pin; hipHostMalloc(&pin, 64, hipHostMallocDefault);
    uint32_t *mapped; hipHostMalloc(&mapped, 64, hipHostMallocMapped | hipHostMallocCoherent);
    uint32_t *mapped_dev; hipHostGetDevicePointer((void **)&mapped_dev, mapped, 0);
    mapped[0] = mapped[1] = 0;
    const int reps = 200;
    for (int spin : {0, 20000}) {
        double ta = 0, tb = 0, tc = 0, tk = 0;
        for (int i = 0; i < reps; ++i) {                      // kernel alone + sync, as the floor
            double t0 = now();
            hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, s, d, (uint32_t)i, spin);
            hipStreamSynchronize(s);
            tk += now() - t0;
        }
        for (int i = 0; i < reps; ++i) {
            uint32_t h = 0;
            double t0 = now();
            hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, s, d, (uint32_t)i, spin);
            hipMemcpyAsync(&h, d, 4, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            ta += now() - t0;
            if (h != (uint32_t)i) { printf("a: wrong value\n"); return 1; }
        }
        for (int i = 0; i < reps; ++i) {
            double t0 = now();
            hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, s, d, (uint32_t)i, spin);
            hipMemcpyAsync(pin, d, 4, hipMemcpyDeviceToHost, s);
            hipStreamSynchronize(s);
            tb += now() - t0;
            if (pin[0] != (uint32_t)i) { printf("b: wrong value\n"); return 1; }
        }
        for (int i = 0; i < reps; ++i) {
            const uint32_t seq = (uint32_t)(i + 1 + (spin ? reps : 0));
            double t0 = now();
            hipLaunchKernelGGL(produce, dim3(1), dim3(64), 0, s, d, (uint32_t)i, spin);
            hipLaunchKernelGGL(publish, dim3(1), dim3(1), 0, s, d, mapped_dev, mapped_dev + 1, seq);
            volatile uint32_t *flag = mapped + 1;
            while (*flag != seq) _mm_pause();
            uint32_t v = ((volatile uint32_t *)mapped)[0];
            tc += now() - t0;
            if (v != (uint32_t)i) { printf("c: wrong value\n"); return 1; }
        }
        hipStreamSynchronize(s);
        printf("spin %5d: kernel+sync %.1f us | a pageable copy+sync %.1f us | b pinned copy+sync %.1f us | c mapped store + poll %.1f us\n",
               spin, tk / reps, ta / reps, tb / reps, tc / reps);
    }
    return 0;
}
