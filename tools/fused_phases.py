#!/usr/bin/env python3
"""Diagnostic build only (make -C pawsometracker.jl_amd/csrc diag; PAWSOME_DOG_LIB=…/libpawsome_dog_diag.so
PDOG_FUSED_DIAG=1): where one frame of the fused one-workgroup kernel spends its time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pawsometracker_jl_amd as pt
from oracle import synth
for (h, w, tw, ws) in ((1080, 1920, 25, (45, 45)), (1080, 1920, 10, (21, 21)), (1080, 1920, 25, (31, 91))):
    f = synth.disc_frame(h, w, (500, 900), tw, True)
    frames = torch.from_numpy(f[None]).cuda()
    g = torch.tensor([[498, 903]], dtype=torch.int32).cuda()
    bt = pt.BatchTracker(h, w, tw, ws, True, 128)
    rows = []
    for _ in range(6):
        out, resp = bt.detect(frames, g, want_resp=True)
        torch.cuda.synchronize()
        rows.append(resp.flatten()[:16].cpu().numpy())
    r = np.median(np.array(rows[1:]), 0)
    cyc, tick = r[:8], r[8:]
    order = [(0, "samples+first loads (wave 0)"), (4, "barrier"), (1, "tile staged"), (2, "row pass"), (5, "col pass + wave peak (wave 0)"),
             (6, "barrier"), (3, "finalize")]
    prev_c = prev_t = 0.0
    parts = []
    for i, name in order:
        parts.append(f"{name} {cyc[i] - prev_c:.0f} cyc ({(tick[i] - prev_t) / 100:.2f} us)")
        prev_c, prev_t = cyc[i], tick[i]
    print(f"window {ws} tw {tw}: total {tick[3] / 100:.2f} us, clock {cyc[3] / tick[3] * 100:.0f} MHz; " + "; ".join(parts), "pos", out.tolist())
    bt.close()
