#!/usr/bin/env python3
"""Pinned host -> device copy bandwidth by chunk size (context for pdog_detect_batch_host's copy-bound rate)."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (1, 8, 32, 128, 512):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    reps = max(4, 2048 // mb)
    t0 = time.perf_counter()
    for _ in range(reps): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mb:4d} MB chunks: {n * reps / dt / 1e9:6.1f} GB/s")
