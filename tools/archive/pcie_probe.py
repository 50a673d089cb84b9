"""Development: H2D rate from pinned memory by copy size (what piece size phx_upload should use)."""
import time, torch
dev = torch.device("cuda:0")
total = 25 << 20
h = torch.empty(total, dtype=torch.uint8).pin_memory()
d = torch.empty(total, dtype=torch.uint8, device=dev)
for piece in (256 << 10, 512 << 10, 1 << 20, 2 << 20, 4 << 20, 8 << 20, total):
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for o in range(0, total, piece):
            d[o:o + piece].copy_(h[o:o + piece], non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
    print("piece %8d B: %.3f ms  %.1f GB/s" % (piece, t * 1e3, total / t / 1e9))
