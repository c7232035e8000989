import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import pawsometracker_jl_amd as pt
from oracle import synth
h, w, tw, ws = 1080, 1920, 25, 256
f = synth.disc_frame(h, w, (500, 900), tw, True)
t = pt.Tracker(f, tw, (ws, ws), True)
g = (480, 880)
t0 = time.perf_counter()
for i in range(20000):
    g = t(g)
    assert g == (500, 900), (i, g)
print("functor 20000 calls ok, %.1f us/call" % ((time.perf_counter() - t0) / 20000 * 1e6))
t.close()
nf = 64
rng = np.random.default_rng(0)
frames = np.stack([synth.disc_frame(h, w, (500 + int(rng.integers(-5, 6)), 900 + int(rng.integers(-5, 6))), tw, True) for _ in range(nf)])
d = torch.from_numpy(frames).cuda()
bt = pt.BatchTracker(h, w, tw, (ws, ws), True, 128)
ref = bt.detect_chain(d, (480, 880)).cpu().numpy()
t0 = time.perf_counter()
for i in range(2000):
    out = bt.detect_chain(d, (480, 880))
    if i % 100 == 0:
        bt.sync()
        assert np.array_equal(out.cpu().numpy(), ref), i
bt.sync()
print("chain 2000 launches x %d frames ok, %.1f us/frame" % (nf, (time.perf_counter() - t0) / 2000 / nf * 1e6))
# batches of 1 and 2 windows back to back with chains in between
g2 = torch.tensor([[480, 880], [510, 910]], dtype=torch.int32).cuda()
for i in range(3000):
    o = bt.detect(d[:2], g2)
    if i % 500 == 0:
        bt.sync(); assert o.cpu().numpy().tolist() == [list(map(int, ref[0])), o.cpu().numpy().tolist()[1]]
bt.sync()
print("small batches ok")
bt.close()
