import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import pawsometracker_jl_amd as pt
from test_gpu_exact import _hard_windows
tw, ws = 25, (45, 45)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
frames, guesses, fam = _hard_windows(n, 128, 128, tw, seed=77)
def run(variant, want_resp=False):
    bt = pt.BatchTracker(128, 128, tw, ws, True, 128)
    bt.set_variant(variant); bt.set_exact(True)
    r = bt.detect(torch.from_numpy(frames).cuda(), torch.from_numpy(guesses).cuda(), want_resp=want_resp)
    bt.sync()
    out = (r[0] if want_resp else r).cpu().numpy()
    print(variant, want_resp, bt.exact_stats(), bt.exact_detail()); bt.close(); return out
ref = run(100)
a = run(200)
b = run(200, True)
for name, x in (("map", a), ("userresp", b)):
    bad = np.flatnonzero((x != ref).any(1))
    print(name, bad.size, bad[:10], bad[-10:])
