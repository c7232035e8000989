import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pawsometracker_jl_amd as pt
h, w, tw, ws, n = 1080, 1920, 25, 256, 4096
kind = sys.argv[1] if len(sys.argv) > 1 else "noise"
rng = np.random.Generator(np.random.PCG64(1))
nf = 64
if kind == "noise":
    frames = (128 + rng.integers(-2, 3, (nf, h, w))).astype(np.uint8)
else:  # low-contrast disc: one grey level darker under +-1 noise
    frames = (128 + rng.integers(-1, 2, (nf, h, w))).astype(np.uint8)
    yy, xx = np.ogrid[0:h, 0:w]
    for k in range(nf):
        ci, cj = rng.integers(300, 780), rng.integers(400, 1500)
        m = (yy - ci) ** 2 + (xx - cj) ** 2 <= 144
        frames[k][m] -= 1
d_f = torch.from_numpy(frames).cuda()
fi = torch.from_numpy(rng.integers(0, nf, n).astype(np.int32)).cuda()
g = torch.from_numpy(np.stack([rng.integers(200, 880, n), rng.integers(300, 1600, n)], 1).astype(np.int32)).cuda()
bt = pt.BatchTracker(h, w, tw, (ws, ws), True, 128)
out = bt.detect(d_f, g, fi); bt.sync()
s0 = bt.exact_detail()
t0 = time.perf_counter(); R = 5
for _ in range(R): out = bt.detect(d_f, g, fi)
bt.sync()
dt = (time.perf_counter() - t0) / R
s1 = bt.exact_detail()
print(f"{kind}: {dt*1e3:.2f} ms per 4096-batch; per batch: refined {(s1[0]-s0[0])/R:.0f}, blocks {(s1[1]-s0[1])/R:.0f}, candidates {(s1[2]-s0[2])/R:.0f}, chains {(s1[3]-s0[3])/R:.0f}")
bt.set_exact(0)
out = bt.detect(d_f, g, fi); bt.sync()
t0 = time.perf_counter()
for _ in range(R): out = bt.detect(d_f, g, fi)
bt.sync()
print(f"{kind}: exact off {(time.perf_counter() - t0) / R*1e3:.2f} ms")
