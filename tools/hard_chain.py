import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pawsometracker_jl_amd as pt
h, w, tw, ws = 1080, 1920, 25, int(sys.argv[1])
rng = np.random.Generator(np.random.PCG64(1))
nf = 128
kind = sys.argv[2]
amp = 2 if kind == "noise" else 1
frames = (128 + rng.integers(-amp, amp + 1, (nf, h, w))).astype(np.uint8)
if kind == "faint":
    yy, xx = np.ogrid[0:h, 0:w]
    m = (yy - 540) ** 2 + (xx - 960) ** 2 <= 144
    for k in range(nf): frames[k][m] -= 1
d = torch.from_numpy(frames).cuda()
for exact in (1, 0):
    bt = pt.BatchTracker(h, w, tw, (ws, ws), True, 128)
    bt.set_exact(exact)
    out = bt.detect_chain(d, (545, 955)); bt.sync()
    t0 = time.perf_counter()
    out = bt.detect_chain(d, (545, 955)); bt.sync()
    dt = (time.perf_counter() - t0) / nf
    print(f"{kind} window {ws} exact {exact}: {dt*1e6:.1f} us/frame; refined {bt.exact_stats()[2]}; last {out[-1].tolist()}")
    bt.close()
