"""Experiment: device-to-host copy rate into pinned and pageable memory on the GPU box (33.5 MB vectors)."""
import time
import torch
n = 1 << 22
x = torch.zeros(n, dtype=torch.float64, device="cuda")
pin = torch.empty(n, dtype=torch.float64, pin_memory=True)
pag = torch.empty(n, dtype=torch.float64)
pag.zero_()
for name, dst in (("pinned", pin), ("pageable", pag)):
    for _ in range(2):
        dst.copy_(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        dst.copy_(x, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    print("%s: %.2f ms per 33.5 MB = %.1f GB/s" % (name, dt * 1e3, 8 * n / dt / 1e9), flush=True)
