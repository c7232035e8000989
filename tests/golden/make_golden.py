"""Generates tests/golden/dog_cases.npz — golden input/output vectors for the
DoG + argmax functor (reference: src/PawsomeTracker.jl:39-62).

PARITY UNPINNED: the reference (Julia) cannot run here and its tests hold no
numeric vectors for this path, so these vectors come from OUR C oracle
(oracle/dog_oracle.c) and are accepted only where the independent NumPy/SciPy
statement (oracle/dog_oracle_np.py) reproduces them (positions exactly,
response to 1e-12).  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import dog_oracle_np as onp  # noqa: E402
from oracle import synth  # noqa: E402
from oracle.dog_oracle import Oracle  # noqa: E402


def cases():
    """(name, frame, target_width, window_size(h,w), darker, guess)"""
    out = []
    rng = np.random.Generator(np.random.PCG64(1234))

    def noisy(f, amp):
        return np.clip(f.astype(np.int16) + rng.integers(-amp, amp + 1, f.shape), 0, 255).astype(np.uint8)

    # reference test defaults: 100x100, tw=10, start (50,50), dark (test/test-basic-test.jl:1-10)
    f = synth.disc_frame(100, 100, (50, 50), 10, True)
    out.append(("defaults_tw10_centre", f, 10, (21, 21), True, (50, 50)))
    out.append(("defaults_tw10_offset", f, 10, (21, 21), True, (46, 55)))
    # tw=25 (the reference default), 240x320 (BASELINE cfg 1 shape), default window 45
    f = synth.disc_frame(240, 320, (120, 160), 25, True)
    out.append(("tw25_centre", f, 25, (45, 45), True, (120, 160)))
    out.append(("tw25_offset", f, 25, (45, 45), True, (110, 171)))
    out.append(("tw25_noise", noisy(f, 3), 25, (45, 45), True, (125, 150)))
    out.append(("tw25_flat_window_ties", f, 25, (45, 45), True, (40, 40)))      # all-equal response
    out.append(("tw25_target_outside_window", f, 25, (45, 45), True, (120, 110)))  # peak on the window edge
    # borders: disc clipped by the frame, window hanging over the corner (fill pad + clamp)
    f = synth.disc_frame(240, 320, (3, 4), 25, True)
    out.append(("tw25_corner_clipped", f, 25, (45, 45), True, (1, 1)))
    f = synth.disc_frame(240, 320, (238, 318), 25, True)
    out.append(("tw25_far_corner", noisy(f, 2), 25, (45, 45), True, (240, 320)))
    # guess outside the frame but inside the reference's pad (a6: not pre-clamped)
    f = synth.disc_frame(240, 320, (6, 160), 25, True)
    out.append(("tw25_guess_outside_frame", f, 25, (45, 45), True, (-20, 160)))
    # bright target, darker_target=false
    f = synth.disc_frame(240, 320, (100, 200), 25, False)
    out.append(("tw25_bright", noisy(f, 3), 25, (45, 45), False, (95, 190)))
    # fill != background: mode is 128 but the window sits on a 90-valued plateau
    f = synth.disc_frame(240, 320, (60, 60), 25, True)
    f[150:, 200:] = 90
    out.append(("tw25_plateau_not_fill", f, 25, (45, 45), True, (200, 260)))
    out.append(("tw25_plateau_edge", f, 25, (45, 45), True, (150, 200)))
    # rectangular window, even window_size (radii = size .÷ 2)
    f = synth.disc_frame(200, 300, (90, 140), 25, True)
    out.append(("tw25_rect_window", noisy(f, 3), 25, (40, 90), True, (100, 120)))
    # window wider than one strip and taller than one chunk (multi-strip / multi-chunk path)
    f = synth.disc_frame(300, 400, (150, 222), 25, True)
    out.append(("tw25_window_129", noisy(f, 3), 25, (129, 129), True, (140, 200)))
    # pure noise (argmax decided by noise only)
    f = rng.integers(0, 256, (120, 160), dtype=np.uint8)
    out.append(("tw25_pure_noise", f, 25, (45, 45), True, (60, 80)))
    out.append(("tw10_pure_noise", f, 10, (31, 31), True, (60, 80)))
    # large sigma (BASELINE cfg 5 kernel, l = 293) on a small window
    f = synth.disc_frame(400, 500, (200, 250), 120, True)
    out.append(("tw120_small_window", noisy(f, 3), 120, (21, 21), True, (195, 258)))
    # other target widths (runtime-L kernels): l = 4*ceil(sqrt2*sigma)+1
    f = synth.disc_frame(160, 200, (80, 100), 16, True)
    out.append(("tw16", noisy(f, 3), 16, (33, 33), True, (84, 95)))
    f = synth.disc_frame(160, 200, (80, 100), 40, True)
    out.append(("tw40", noisy(f, 3), 40, (61, 61), True, (70, 110)))
    return out


def main():
    o = Oracle()
    store = {}
    names = []
    for name, frame, tw, ws, darker, guess in cases():
        sig = o.sigma(tw)
        K = o.dog_kernel(sig, darker)
        fill = o.mode_u8(frame)
        radii = (ws[0] // 2, ws[1] // 2)
        ij, resp = o.detect(frame, fill, K, radii, guess, want_resp=True)
        ij2, resp2 = onp.detect(frame, onp.mode_u8(frame), onp.dog_kernel(onp.sigma(tw), darker), radii, guess)
        assert tuple(int(v) for v in ij2) == ij, (name, ij, ij2)
        assert np.abs(resp - resp2).max() <= 1e-12, name
        ij3, resp3 = o.detect_separable(frame, fill, sig, darker, K.shape[0], radii, guess, want_resp=True)
        assert np.abs(resp - resp3).max() <= 1e-12, name
        names.append(name)
        store[name + "/frame"] = frame
        store[name + "/params"] = np.array([tw, ws[0], ws[1], int(darker), guess[0], guess[1], fill, K.shape[0]], np.int32)
        store[name + "/ij"] = np.array(ij, np.int32)
        store[name + "/resp"] = np.ascontiguousarray(resp)  # [win_h, win_w] float64 (C order copy)
        print(f"{name:32s} l={K.shape[0]:3d} fill={fill:3d} guess={guess} -> ij={ij} max|resp|={np.abs(resp).max():.4f}")
    store["names"] = np.array(names)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dog_cases.npz")
    np.savez_compressed(path, **store)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
