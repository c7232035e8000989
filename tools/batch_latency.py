import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pawsometracker_jl_amd as pt
from oracle import synth
h, w, tw, ws = 1080, 1920, 25, int(sys.argv[1])
f = synth.disc_frame(h, w, (500, 900), tw, True)
frames = torch.from_numpy(np.broadcast_to(f, (32, h, w)).copy()).cuda()
bt = pt.BatchTracker(h, w, tw, (ws, ws), True, 128)
for n in (1, 2, 4, 8, 16, 32):
    g = torch.tensor([[480, 880]] * n, dtype=torch.int32).cuda()
    out = torch.empty((n, 2), dtype=torch.int32, device="cuda")
    for _ in range(10): bt.detect(frames[:n], g, out=out)
    bt.sync()
    t0 = time.perf_counter(); R = 200
    for _ in range(R): bt.detect(frames[:n], g, out=out)
    bt.sync()
    dt = (time.perf_counter() - t0) / R
    print(f"window {ws} batch {n}: {dt*1e6:.1f} us per batch (back-to-back launches), kernel {bt.kernel_for_batch(n)}; {out[0].tolist()}")
