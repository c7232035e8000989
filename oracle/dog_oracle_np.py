"""Independent NumPy/SciPy statement of the DoG + argmax functor.

TEST INFRASTRUCTURE ONLY (see dog_oracle.c).  PARITY UNPINNED.
Written against the same reference lines as the C oracle but with library
primitives (scipy.signal.correlate2d, numpy argmax) so that a transcription
error in one of the two shows up as a disagreement.
"""
import math

import numpy as np
from scipy.signal import correlate2d


def sigma(target_width):                      # src/PawsomeTracker.jl:30
    return target_width / (2 * math.sqrt(2 * math.log(2)))


def default_window(target_width):             # src/PawsomeTracker.jl:64-68
    return 4 * math.ceil(sigma(target_width)) + 1


def kernel_len(sig):                          # ImageFiltering Kernel.DoG default length
    return 4 * math.ceil(sig * math.sqrt(2)) + 1


def gaussian_1d(sig, l):                      # ImageFiltering KernelFactors.gaussian
    w = l // 2
    x = np.arange(-w, w + 1, dtype=np.float64)
    g = np.exp(-x * x / (2 * sig * sig))
    return g / g.sum()


def dog_kernel(sig, darker, l=None):          # src/PawsomeTracker.jl:41-43
    l = kernel_len(sig) if l is None else l
    gp, gm = gaussian_1d(sig, l), gaussian_1d(sig * math.sqrt(2), l)
    return (-1.0 if darker else 1.0) * (np.outer(gp, gp) - np.outer(gm, gm))


def mode_u8(img):                             # src/PawsomeTracker.jl:47 (StatsBase.mode)
    flat = np.asarray(img, np.uint8).T.ravel()  # column-major scan of the h x w view
    cnt = np.zeros(256, np.int64)
    mc, mv = 0, int(flat[0])
    for v in flat:
        cnt[v] += 1
        if cnt[v] > mc:
            mc, mv = cnt[v], int(v)
    return mv


def padded_tile(frame, fill, top, left, th, tw):
    """tile[a, b] = frame[top+a, left+b] (1-based top/left) or fill outside (PaddedView, :48)."""
    h, w = frame.shape
    tile = np.full((th, tw), fill, np.uint8)
    a0, a1 = max(1, top), min(h, top + th - 1)
    b0, b1 = max(1, left), min(w, left + tw - 1)
    if a0 <= a1 and b0 <= b1:
        tile[a0 - top:a1 - top + 1, b0 - left:b1 - left + 1] = frame[a0 - 1:a1, b0 - 1:b1]
    return tile


def detect(frame, fill, K, radii, guess):     # src/PawsomeTracker.jl:55-62
    h, w = frame.shape
    l = K.shape[0]
    hw = l // 2
    r1, r2 = radii
    i0, j0 = guess[0] - r1, guess[1] - r2
    tile = padded_tile(frame, fill, i0 - hw, j0 - hw, 2 * r1 + 1 + 2 * hw, 2 * r2 + 1 + 2 * hw)
    resp = correlate2d(tile.astype(np.float64) / 255.0, K, mode="valid")
    flat = np.argmax(resp.T.ravel())          # first max, column-major
    bj, bi = divmod(int(flat), 2 * r1 + 1)
    ij = (min(max(i0 + bi, 1), h), min(max(j0 + bj, 1), w))
    return ij, resp
