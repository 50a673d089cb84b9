import torch, time
x = torch.empty(50_000_000, dtype=torch.uint8).pin_memory()
y = torch.empty_like(x, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); y.copy_(x, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    print("H2D pinned 50 MB: %.2f ms = %.1f GB/s" % ((t1 - t0) * 1e3, 0.05 / (t1 - t0)))
import numpy as np
a = np.random.randint(0, 255, 50_000_000, dtype=np.uint8); b = np.empty_like(a)
t0 = time.perf_counter(); b[:] = a; t1 = time.perf_counter(); print("host memcpy 50 MB: %.2f ms" % ((t1 - t0) * 1e3))
