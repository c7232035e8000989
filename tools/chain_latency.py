import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pawsometracker_jl_amd as pt
from oracle import synth
h, w, tw, ws = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
off = (40, 60) if ws > 120 else (ws // 5, ws // 4)
f = synth.disc_frame(h, w, (h // 2 - off[0], w // 2 + off[1]), tw, True)
nf = 256
frames = torch.from_numpy(np.broadcast_to(f, (nf, h, w)).copy()).cuda()
bt = pt.BatchTracker(h, w, tw, (ws, ws), True, 128)
out = bt.detect_chain(frames, (h // 2, w // 2)); bt.sync()
t0 = time.perf_counter()
out = bt.detect_chain(frames, (h // 2, w // 2)); bt.sync()
dt = (time.perf_counter() - t0) / nf
print(f"chain {h}x{w} tw {tw} window {ws}: {dt*1e6:.1f} us/frame; last {out[-1].tolist()} kernel {bt.kernel_for_batch(1) if hasattr(bt,'kernel_for_batch') else ''}")
