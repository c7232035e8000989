"""Deterministic synthetic frames for tests and bench.

Mirrors the reference's test recipe (test/test-basic-test.jl:65-68): a flat
Gray{N0f8}(0.5) background (raw 128) with one filled disc of radius
target_width ÷ 2, value 0 (dark target) or 255 (bright target).  The reference
pushes frames through JPEG + H.264; here they stay raw u8 (no ffmpeg).
TEST/BENCH INPUT GENERATOR — holds no reference algorithm.
"""
import numpy as np


def disc_frame(h, w, centre, target_width, darker=True, bkgd=128, out=None):
    """One h x w u8 frame; centre is 1-based (row, col) like CartesianIndex."""
    f = np.full((h, w), bkgd, np.uint8) if out is None else out
    if out is not None:
        f[...] = bkgd
    rad = int(target_width) // 2
    ci, cj = centre[0] - 1, centre[1] - 1
    i0, i1 = max(0, ci - rad), min(h - 1, ci + rad)
    j0, j1 = max(0, cj - rad), min(w - 1, cj + rad)
    if i0 <= i1 and j0 <= j1:
        ii, jj = np.ogrid[i0:i1 + 1, j0:j1 + 1]
        mask = (ii - ci) ** 2 + (jj - cj) ** 2 <= rad * rad
        f[i0:i1 + 1, j0:j1 + 1][mask] = 0 if darker else 255
    return f


def make_batch(n, h, w, target_width, radii, darker=True, seed=0, noise=0, margin_frac=0.5):
    """n frames + guesses: centres uniform over the frame (borders included),
    guess = centre + uniform integer offset within ±radii*margin_frac, clamped to the frame."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ci = rng.integers(1, h + 1, n)
    cj = rng.integers(1, w + 1, n)
    di = rng.integers(-int(radii[0] * margin_frac), int(radii[0] * margin_frac) + 1, n)
    dj = rng.integers(-int(radii[1] * margin_frac), int(radii[1] * margin_frac) + 1, n)
    frames = np.empty((n, h, w), np.uint8)
    for b in range(n):
        disc_frame(h, w, (int(ci[b]), int(cj[b])), target_width, darker, out=frames[b])
    if noise:
        nrng = np.random.Generator(np.random.PCG64(seed + 1))
        nz = nrng.integers(-noise, noise + 1, frames.shape, dtype=np.int16)
        frames = np.clip(frames.astype(np.int16) + nz, 0, 255).astype(np.uint8)
    guesses = np.stack([np.clip(ci + di, 1, h), np.clip(cj + dj, 1, w)], 1).astype(np.int32)
    centres = np.stack([ci, cj], 1).astype(np.int32)
    return frames, guesses, centres
