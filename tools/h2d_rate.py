"""Host-to-device copy rate of the box (pinned and pageable memory, several sizes): what an upload of a batch's sequences can
reach at best.  GPU box: python tools/h2d_rate.py"""
import time
import torch

dev = torch.device("cuda", 0)
for mb in (1, 4, 24, 96):
    n = mb << 20
    pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
    pageable = torch.empty(n, dtype=torch.uint8)
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    for name, src in (("pinned", pinned), ("pageable", pageable)):
        dst.copy_(src, non_blocking=True); torch.cuda.synchronize()
        best = 1e9
        for _ in range(10):
            t0 = time.perf_counter()
            dst.copy_(src, non_blocking=True)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%3d MB %-8s %.3f ms  %.1f GB/s" % (mb, name, best * 1e3, n / best / 1e9))
