#!/usr/bin/env python3
"""Diagnostic (ablation build): cycles per sub-chunk spent in the stage / row / column phases of the roll kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pawsometracker_jl_amd as pt
from bench import make_frames
n = 4096
frames, gh, _ = make_frames(torch, n, 1080, 1920, 25, (128, 128), 0, 3, torch.device("cuda", 0))
fill = pt.mode(frames[0].cpu().numpy())
bt = pt.BatchTracker(1080, 1920, 25, (256, 256), True, fill)
bt.set_variant(107); bt.use_torch_stream()
g = torch.from_numpy(gh).cuda()
for _ in range(2):
    bt.detect(frames, g)
out, resp = bt.detect(frames, g, want_resp=True)
torch.cuda.synchronize()
nb = n * bt.info().n_strips
st = resp.flatten()[: 4 * nb].cpu().numpy().reshape(nb, 4)
per = st[:, :3] / st[:, 3:4]
print("cycles per sub-chunk (median over waves): stage %.0f  row %.0f  col %.0f  total %.0f" % (*np.median(per, 0), np.median(per.sum(1))))
