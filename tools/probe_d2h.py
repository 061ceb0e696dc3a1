# GPU probe: device -> host copy of the quantized coefficient matrix (cfg3: 708 MB), pageable vs pinned
import time, torch
N, D = 2999072, 59
q = torch.randint(-100, 100, (D, N), dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    t = time.perf_counter(); h = q.cpu(); dt = time.perf_counter() - t
print("pageable .cpu()        : %.1f ms  (%.1f GB/s)" % (dt * 1e3, q.numel() * 4 / dt / 1e9))
t = time.perf_counter(); pin = torch.empty((D, N), dtype=torch.int32, pin_memory=True); print("pinned alloc %.1f ms" % ((time.perf_counter() - t) * 1e3))
for _ in range(2):
    t = time.perf_counter(); pin.copy_(q, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("pinned copy_           : %.1f ms  (%.1f GB/s)" % (dt * 1e3, q.numel() * 4 / dt / 1e9))
for _ in range(2):
    t = time.perf_counter(); q2 = pin.to("cuda", non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("pinned H2D             : %.1f ms  (%.1f GB/s)" % (dt * 1e3, q.numel() * 4 / dt / 1e9))
h2 = h.clone()
for _ in range(2):
    t = time.perf_counter(); q3 = h2.to("cuda"); torch.cuda.synchronize(); dt = time.perf_counter() - t
print("pageable H2D           : %.1f ms  (%.1f GB/s)" % (dt * 1e3, q.numel() * 4 / dt / 1e9))
