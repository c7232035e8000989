#!/usr/bin/env python3
"""Diagnostic (ablation build only): per-wave shader cycles and real time of the roll kernel's main loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pawsometracker_jl_amd as pt
from bench import make_frames
n = 4096
frames, gh, _ = make_frames(torch, n, 1080, 1920, 25, (128, 128), 0, 3, torch.device("cuda", 0))
fill = pt.mode(frames[0].cpu().numpy())
bt = pt.BatchTracker(1080, 1920, 25, (256, 256), True, fill)
bt.set_variant(106); bt.use_torch_stream()
g = torch.from_numpy(gh).cuda()
for _ in range(2):
    bt.detect(frames, g)
out, resp = bt.detect(frames, g, want_resp=True)
torch.cuda.synchronize()
info = bt.info()
nb = n * info.n_strips
st = resp.flatten()[: 2 * nb].cpu().numpy().reshape(nb, 2)
cyc, ticks = st[:, 0], st[:, 1]
clk = cyc / (ticks / 100e6) / 1e9
print("waves", nb, "strips", info.n_strips)
print("cycles per wave: median %.0f min %.0f max %.0f" % (np.median(cyc), cyc.min(), cyc.max()))
print("wave lifetime us: median %.1f" % np.median(ticks / 100))
print("clock GHz: median %.3f min %.3f max %.3f" % (np.median(clk), clk.min(), clk.max()))
