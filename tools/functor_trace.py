#!/usr/bin/env python3
"""Diagnostic: PDOG_HOST_TRACE=1 python tools/functor_trace.py — median pack / launch / sync time of the host functor."""
import os, re, subprocess, sys
import numpy as np
if os.environ.get("PDOG_TRACE_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import pawsometracker_jl_amd as pt
    from oracle import synth
    f = synth.disc_frame(1080, 1920, (500, 900), 25, True)
    t = pt.Tracker(f, 25, (45, 45), True)
    g = (500, 900)
    for _ in range(300):
        g = t(g)
    sys.exit(0)
env = dict(os.environ, PDOG_TRACE_CHILD="1", PDOG_HOST_TRACE="1")
err = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True).stderr
rows = np.array([[float(x) for x in m] for m in re.findall(r"pack ([\d.]+) us, launch ([\d.]+) us, sync ([\d.]+) us", err)])
print("calls", len(rows), "median us: pack %.1f launch %.1f sync %.1f" % tuple(np.median(rows[50:], 0)))
