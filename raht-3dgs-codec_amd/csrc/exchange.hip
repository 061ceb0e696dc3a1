// exchange.hip -- DIRECT all-gather of the root rows of a Morton-prefix sharded scene.
//
// The exchange of a sharded step is <= 512 rows in all (<= 15 KB per rank at D = 59): latency-bound on xGMI, and xGMI is
// point to point -- every GPU has its own link to each of the 7 others. So instead of a ring (RCCL's all_gather_into_tensor,
// the default path) each rank WRITES its slot straight into every peer's gather buffer, one workgroup per peer, and then
// raises that peer's flag; it leaves when all of its own flags have been raised. One launch per direction, no host
// involvement, no staging copy.
//
//   exchange block of a rank (fine-grained device memory, opened by every peer through hipIpc):
//     [ 2 x world x slot_bytes ]  data, double-buffered by the parity of the sequence number: a rank that is one gather ahead
//                                 writes buffer (g + 1) & 1 while its peer still reads buffer g & 1 -- and nobody can be two
//                                 ahead, because gather g + 1 only completes once every peer has STARTED it
//     [ world x 64 bytes ]        flags: flag[src] = sequence number of the last gather whose slot from `src` has landed
//     [ 64 bytes ]                status: 1 = a wait timed out (a peer never arrived)
//
// Every wait is bounded (XCHG_TIMEOUT_TICKS of the 100 MHz wall clock): a missing peer ends in an error status, not a hung GPU.
#include "raht_common.h"

#include <cstring>

namespace raht {

constexpr int XCHG_MAX_WORLD = 8;
constexpr uint64_t XCHG_TIMEOUT_TICKS = 100000000ull * 20;     // 20 s of the 100 MHz constant clock

struct XchgPeers { unsigned char *base[XCHG_MAX_WORLD]; };

__host__ __device__ inline size_t xchg_data_bytes(int world, int64_t slot_bytes) { return (size_t)2 * (size_t)world * (size_t)slot_bytes; }
__host__ __device__ inline size_t xchg_flags_off(int world, int64_t slot_bytes) { return (xchg_data_bytes(world, slot_bytes) + 255) & ~(size_t)255; }
__host__ __device__ inline size_t xchg_status_off(int world, int64_t slot_bytes) { return xchg_flags_off(world, slot_bytes) + (size_t)world * 64; }
__host__ __device__ inline size_t xchg_total_bytes(int world, int64_t slot_bytes) { return xchg_status_off(world, slot_bytes) + 64; }

__global__ __launch_bounds__(256) void xchg_gather_kernel(const uint4 *__restrict__ send, int64_t slot_bytes, XchgPeers P, int rank, int world, uint32_t seq)
{
    const int p = blockIdx.x;                                   // this workgroup serves peer p
    const size_t buf_off = (size_t)(seq & 1u) * (size_t)world * (size_t)slot_bytes;
    // 1. my slot -> peer p's buffer (p == rank: my own copy)
    uint4 *dst = (uint4 *)(P.base[p] + buf_off + (size_t)rank * (size_t)slot_bytes);
    const int64_t nvec = slot_bytes >> 4;
    for (int64_t i = threadIdx.x; i < nvec; i += blockDim.x) dst[i] = send[i];
    __threadfence_system();                                     // the slot is visible to the peer before its flag says so
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t *peer_flag = (uint32_t *)(P.base[p] + xchg_flags_off(world, slot_bytes) + (size_t)rank * 64);
        __hip_atomic_store(peer_flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        // 2. wait for peer p's slot in MY buffer
        uint32_t *my_flag = (uint32_t *)(P.base[rank] + xchg_flags_off(world, slot_bytes) + (size_t)p * 64);
        const uint64_t t0 = wall_clock64();
        bool ok = true;
        // (sequence numbers wrap after 2^32 gathers: compare as a signed distance)
        while ((int32_t)(__hip_atomic_load(my_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
            __builtin_amdgcn_s_sleep(32);
            if (wall_clock64() - t0 > XCHG_TIMEOUT_TICKS) { ok = false; break; }
        }
        if (!ok) {
            uint32_t *status = (uint32_t *)(P.base[rank] + xchg_status_off(world, slot_bytes));
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

}  // namespace raht

using namespace raht;

extern "C" {

int raht_xchg_bytes(int world, int64_t slot_bytes, int64_t *total)
{
    if (world < 1 || world > XCHG_MAX_WORLD || slot_bytes < 16 || (slot_bytes & 15) || !total) { set_error("raht_xchg_bytes: world in [1, %d], slot_bytes a multiple of 16", XCHG_MAX_WORLD); return RAHT_ERR_INVALID; }
    *total = (int64_t)xchg_total_bytes(world, slot_bytes);
    return RAHT_OK;
}

int raht_xchg_alloc(int world, int64_t slot_bytes, void **base, void *handle64)
{
    int64_t total = 0;
    RAHT_RET(raht_xchg_bytes(world, slot_bytes, &total));
    if (!base || !handle64) { set_error("raht_xchg_alloc: NULL argument"); return RAHT_ERR_INVALID; }
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handles travel as 64 bytes");
    void *p = nullptr;
    // fine-grained: peers' writes and the flag traffic must be coherent INSIDE running kernels, not only at kernel boundaries
    RAHT_HIP_CHECK(hipExtMallocWithFlags(&p, (size_t)total, hipDeviceMallocFinegrained));
    hipError_t e = hipMemset(p, 0, (size_t)total);
    if (e == hipSuccess) e = hipIpcGetMemHandle((hipIpcMemHandle_t *)handle64, p);
    if (e != hipSuccess) { (void)hipFree(p); set_error("raht_xchg_alloc: %s", hipGetErrorString(e)); return RAHT_ERR_HIP; }
    *base = p;
    return RAHT_OK;
}

int raht_xchg_open(const void *handle64, void **base)
{
    if (!handle64 || !base) { set_error("raht_xchg_open: NULL argument"); return RAHT_ERR_INVALID; }
    hipIpcMemHandle_t h;
    memcpy(&h, handle64, sizeof(h));
    RAHT_HIP_CHECK(hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess));
    return RAHT_OK;
}

int raht_xchg_close(void *base)
{
    if (!base) return RAHT_OK;
    RAHT_HIP_CHECK(hipIpcCloseMemHandle(base));
    return RAHT_OK;
}

int raht_xchg_free(void *base)
{
    if (!base) return RAHT_OK;
    RAHT_HIP_CHECK(hipFree(base));
    return RAHT_OK;
}

/* One direct all-gather: `send` (slot_bytes of device memory) lands in slot `rank` of buffer (seq & 1) of every peer's block;
 * returns (in stream order) once every peer's slot has landed in this rank's block. peers: HOST array of `world` block base
 * pointers as mapped in THIS process (peers[rank] = the own block). seq: 1, 2, 3, ... the same on every rank. */
int raht_xchg_gather(const void *send, int64_t slot_bytes, void *const *peers, int rank, int world, uint32_t seq, raht_stream_t stream)
{
    int64_t total = 0;
    RAHT_RET(raht_xchg_bytes(world, slot_bytes, &total));
    if (!send || !peers || rank < 0 || rank >= world) { set_error("raht_xchg_gather: bad argument"); return RAHT_ERR_INVALID; }
    XchgPeers P;
    for (int i = 0; i < XCHG_MAX_WORLD; ++i) {
        P.base[i] = (unsigned char *)peers[i < world ? i : 0];
        if (i < world && !peers[i]) { set_error("raht_xchg_gather: peer %d is not mapped", i); return RAHT_ERR_INVALID; }
    }
    hipLaunchKernelGGL(xchg_gather_kernel, dim3((unsigned)world), dim3(256), 0, (hipStream_t)stream, (const uint4 *)send, slot_bytes, P, rank, world, seq);
    RAHT_HIP_CHECK(hipGetLastError());
    return RAHT_OK;
}

/* Device address of buffer (seq & 1) of a block, and (synchronising) its status word: 0 = fine, 1 = a wait timed out. */
int raht_xchg_buffer(void *base, int world, int64_t slot_bytes, uint32_t seq, void **buf)
{
    int64_t total = 0;
    RAHT_RET(raht_xchg_bytes(world, slot_bytes, &total));
    if (!base || !buf) { set_error("raht_xchg_buffer: NULL argument"); return RAHT_ERR_INVALID; }
    *buf = (unsigned char *)base + (size_t)(seq & 1u) * (size_t)world * (size_t)slot_bytes;
    return RAHT_OK;
}

int raht_xchg_status(void *base, int world, int64_t slot_bytes, raht_stream_t stream, int *status)
{
    int64_t total = 0;
    RAHT_RET(raht_xchg_bytes(world, slot_bytes, &total));
    if (!base || !status) { set_error("raht_xchg_status: NULL argument"); return RAHT_ERR_INVALID; }
    uint32_t v = 0;
    RAHT_HIP_CHECK(hipMemcpyAsync(&v, (unsigned char *)base + xchg_status_off(world, slot_bytes), 4, hipMemcpyDeviceToHost, (hipStream_t)stream));
    RAHT_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *status = (int)v;
    return RAHT_OK;
}

}  // extern "C"
