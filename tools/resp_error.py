#!/usr/bin/env python3
"""Prints max|gpu - oracle| / max|oracle| of the DoG response for every golden case and kernel family."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import pawsometracker_jl_amd as pt
z = np.load(os.path.join(ROOT, "tests", "golden", "dog_cases.npz"))
worst = {}
for name in z["names"]:
    name = str(name)
    tw, wh, ww, darker, g1, g2, fill, l = (int(v) for v in z[name + "/params"])
    ref = z[name + "/resp"]
    row = [f"{name:30s} l={l:3d}"]
    for variant in (-1, 2):
        t = pt.Tracker(z[name + "/frame"], tw, (wh, ww), bool(darker))
        try:
            if variant >= 0:
                t.set_variant(variant)
        except pt.PdogError:
            t.close(); continue
        ij, resp = t((g1, g2), want_resp=True)
        v = t.info().variant
        t.close()
        scale = np.abs(ref).max()
        err = np.abs(resp.astype(np.float64) - ref).max()
        rel = err / scale if scale > 1e-6 else float("nan")
        row.append(f"v{v}: abs {err:.2e} rel {rel:.2e} pos_ok={ij == tuple(int(x) for x in z[name + '/ij'])}")
        if scale > 1e-6:
            worst[v] = max(worst.get(v, 0), rel)
    print("  ".join(row))
print("worst relative error per variant:", {k: f"{v:.2e}" for k, v in worst.items()})
