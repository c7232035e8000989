#!/usr/bin/env python3
"""Kernel time vs batch size: roll kernel (one wave per strip, v100), two-pass (many workgroups per window, v200),
fused (one workgroup per window, one launch, v300 — windows whose padded tile fits in LDS)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pawsometracker_jl_amd as pt
from bench import make_frames
dev = torch.device("cuda", 0)
for (ws, label) in (((256, 256), "257x257"), ((270, 480), "271x481"), ((45, 45), "45x45")):
    for n in (1, 8, 64, 256, 1024):
        frames, gh, _ = make_frames(torch, n, 1080, 1920, 25, (ws[0] // 2, ws[1] // 2), 0, 3, dev)
        g = torch.from_numpy(gh).cuda()
        row = [f"{label} n={n:5d}"]
        for variant in (100, 200, 300):
            bt = pt.BatchTracker(1080, 1920, 25, ws, True, 128)
            try:
                bt.set_variant(variant)
            except pt.PdogError:      # 300 = fused one-workgroup kernel: needs the window's padded tile in LDS
                bt.close(); continue
            bt.use_torch_stream(); bt.reserve(n)
            out = torch.empty((n, 2), dtype=torch.int32, device=dev)
            for _ in range(3): bt.detect(frames, g, out=out)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); e0.record()
            for _ in range(20): bt.detect(frames, g, out=out)
            e1.record(); torch.cuda.synchronize()
            row.append(f"v{variant}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us")
            bt.close()
        print("  ".join(row))
