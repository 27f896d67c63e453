"""Raw host->device copy bandwidth from pinned memory (the ceiling of the host-pointer API), via torch."""
import time
import torch
for mb in (38, 154, 616):
    n = mb * (1 << 20) // 4
    h = torch.empty(n, dtype=torch.float32, pin_memory=True)
    h.fill_(1.0)
    d = torch.empty(n, dtype=torch.float32, device="cuda")
    for _ in range(2):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    t = time.perf_counter()
    reps = 5
    for _ in range(reps):
        d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("pinned H2D %4d MB: %.2f ms  %.1f GB/s" % (mb, dt * 1e3, mb * 1.048576e-3 / dt))
    hp = torch.empty(n, dtype=torch.float32)
    hp.fill_(1.0)
    d.copy_(hp); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        d.copy_(hp)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print("pageable H2D %4d MB: %.2f ms  %.1f GB/s" % (mb, dt * 1e3, mb * 1.048576e-3 / dt))
    o = torch.empty(n, dtype=torch.float32, pin_memory=True)
    d2 = d[: n // 300]
    t = time.perf_counter()
    o[: n // 300].copy_(d2, non_blocking=True); torch.cuda.synchronize()
    print("  small D2H %.3f ms" % ((time.perf_counter() - t) * 1e3))
