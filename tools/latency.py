#!/usr/bin/env python3
"""Single-clip latencies: host functor call (upload + kernels + readback) and the device chain."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pawsometracker_jl_amd as pt
from oracle import synth
h, w, tw = 1080, 1920, 25
for ws in (45, 256):
    f = synth.disc_frame(h, w, (500, 900), tw, True)
    t = pt.Tracker(f, tw, (ws, ws), True)
    g = (500, 900)
    for _ in range(20): g = t(g)
    t0 = time.perf_counter(); n = 200
    for _ in range(n): g = t(g)
    dt = (time.perf_counter() - t0) / n
    print(f"host functor 1080p window {ws}: {dt*1e6:.1f} us/call -> {1/dt:.0f} frames/s (upload of the window tile + kernels + readback)")
    t.close()
    nf = 512
    frames = torch.from_numpy(np.broadcast_to(f, (nf, h, w)).copy()).cuda()
    bt = pt.BatchTracker(h, w, tw, (ws, ws), True, 128)
    out = bt.detect_chain(frames, (500, 900)); bt.sync()
    t0 = time.perf_counter()
    out = bt.detect_chain(frames, (500, 900)); bt.sync()
    dt = (time.perf_counter() - t0) / nf
    print(f"device chain 1080p window {ws}: {dt*1e6:.1f} us/frame -> {1/dt:.0f} frames/s; last {out[-1].tolist()}")
    bt.close()
