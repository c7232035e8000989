#!/usr/bin/env python3
"""Soak: N functor calls (in-place tile + ticket polling) on a moving target, every answer checked against the known
disc centre; prints calls/s.  python tools/soak_functor.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pawsometracker_jl_amd as pt
from oracle import synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
h, w, tw = 480, 640, 25
frames, cs = [], []
for k in range(64):
    c = (100 + 3 * k, 120 + 5 * k)
    frames.append(synth.disc_frame(h, w, c, tw, True)); cs.append(c)
t = pt.Tracker(frames[0], tw, (45, 45), True)
g = cs[0]
bad = 0
t0 = time.perf_counter()
for i in range(n):
    k = i % 128
    k = k if k < 64 else 127 - k          # walk forth and back
    t.img.data = frames[k]
    g = t(g)
    bad += g != cs[k]
dt = time.perf_counter() - t0
print(f"{n} calls, {bad} wrong, {n / dt:.0f} calls/s ({dt / n * 1e6:.1f} us per call)")
sys.exit(1 if bad else 0)
