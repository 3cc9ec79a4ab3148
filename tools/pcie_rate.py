import torch, time
d = torch.empty(58_000_000, dtype=torch.uint8, device='cuda')
h = torch.empty(58_000_000, dtype=torch.uint8).pin_memory()
u = torch.empty(58_000_000, dtype=torch.uint8)
for name, dst in (("pinned", h), ("pageable", u)):
    for _ in range(3):
        torch.cuda.synchronize(); t = time.perf_counter(); dst.copy_(d, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(name, "%.2f ms  %.1f GB/s" % (dt * 1e3, 58e6 / dt / 1e9))
t = time.perf_counter(); u.copy_(h); dt = time.perf_counter() - t
print("host memcpy %.2f ms" % (dt * 1e3))
