"""ctypes binding of the C oracle (oracle/dog_oracle.c).

TEST INFRASTRUCTURE ONLY — see the header of dog_oracle.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
PARITY UNPINNED (no Julia here, no golden vectors in the reference's tests).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")


def build(force=False):
    """Compile both oracle libraries with gcc (no-op when they are up to date)."""
    src = os.path.join(_HERE, "dog_oracle.c")
    outs = [os.path.join(_BUILD, n) for n in ("libdog_oracle.so", "libdog_oracle_fast.so")]
    fresh = all(os.path.exists(o) and os.path.getmtime(o) >= os.path.getmtime(src) for o in outs)
    if force or not fresh:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return outs


class Oracle:
    """Thin wrapper; `fast=True` loads the -Ofast build (timing only)."""

    def __init__(self, fast=False):
        name = "libdog_oracle_fast.so" if fast else "libdog_oracle.so"
        path = os.path.join(_BUILD, name)
        if not os.path.exists(path):
            build()
        L = self.lib = C.CDLL(path)
        d, i, p = C.c_double, C.c_int, C.c_void_p
        L.pdo_sigma.restype = d; L.pdo_sigma.argtypes = [d]
        L.pdo_default_window.restype = i; L.pdo_default_window.argtypes = [d]
        L.pdo_kernel_len.restype = i; L.pdo_kernel_len.argtypes = [d]
        L.pdo_gaussian_1d.restype = None; L.pdo_gaussian_1d.argtypes = [d, i, p]
        L.pdo_dog_kernel.restype = None; L.pdo_dog_kernel.argtypes = [d, i, i, p]
        L.pdo_mode_u8.restype = i; L.pdo_mode_u8.argtypes = [p, i, i, C.c_int64]
        L.pdo_detect_dense.restype = None
        L.pdo_detect_dense.argtypes = [p, i, i, C.c_int64, i, p, i, i, i, i, i, p, p, p, i]
        L.pdo_detect_separable.restype = None
        L.pdo_detect_separable.argtypes = [p, i, i, C.c_int64, i, d, i, i, i, i, i, i, p, p, p, i]
        L.pdo_detect_batch_dense.restype = None
        L.pdo_detect_batch_dense.argtypes = [p, C.c_int64, i, i, i, C.c_int64, i, p, i, i, i, p, p, i]
        L.pdo_detect_batch_par.restype = None
        L.pdo_detect_batch_par.argtypes = [p, C.c_int64, i, i, i, C.c_int64, i, p, d, i, i, i, i, p, p, i, i]
        L.pdo_max_threads.restype = i

    # --- scalars (src/PawsomeTracker.jl:30, :64-68; ImageFiltering Kernel.DoG) ---
    def sigma(self, tw):
        return self.lib.pdo_sigma(float(tw))

    def default_window(self, tw):
        return self.lib.pdo_default_window(float(tw))

    def kernel_len(self, sigma):
        return self.lib.pdo_kernel_len(float(sigma))

    def gaussian_1d(self, sigma, l):
        g = np.empty(l, np.float64)
        self.lib.pdo_gaussian_1d(float(sigma), int(l), g.ctypes.data)
        return g

    def dog_kernel(self, sigma, darker, l=None):
        l = self.kernel_len(sigma) if l is None else l
        K = np.empty((l, l), np.float64, order="F")
        self.lib.pdo_dog_kernel(float(sigma), int(bool(darker)), int(l), K.ctypes.data)
        return K

    def mode_u8(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        return self.lib.pdo_mode_u8(img.ctypes.data, h, w, w)

    def max_threads(self):
        return self.lib.pdo_max_threads()

    # --- the functor (src/PawsomeTracker.jl:55-62) ---
    def detect(self, frame, fill, K, radii, guess, want_resp=False, nthreads=0):
        frame = np.ascontiguousarray(frame, np.uint8)
        h, w = frame.shape
        l = K.shape[0]
        K = np.asfortranarray(K, np.float64)
        r1, r2 = radii
        oi, oj = C.c_int(), C.c_int()
        resp = np.empty((2 * r1 + 1, 2 * r2 + 1), np.float64, order="F") if want_resp else None
        self.lib.pdo_detect_dense(frame.ctypes.data, h, w, w, int(fill), K.ctypes.data, l, r1, r2,
                                  int(guess[0]), int(guess[1]), C.byref(oi), C.byref(oj),
                                  resp.ctypes.data if want_resp else None, nthreads)
        return ((oi.value, oj.value), resp) if want_resp else (oi.value, oj.value)

    def detect_separable(self, frame, fill, sigma, darker, l, radii, guess, want_resp=False, nthreads=0):
        frame = np.ascontiguousarray(frame, np.uint8)
        h, w = frame.shape
        r1, r2 = radii
        oi, oj = C.c_int(), C.c_int()
        resp = np.empty((2 * r1 + 1, 2 * r2 + 1), np.float64, order="F") if want_resp else None
        self.lib.pdo_detect_separable(frame.ctypes.data, h, w, w, int(fill), float(sigma),
                                      int(bool(darker)), int(l), r1, r2, int(guess[0]), int(guess[1]),
                                      C.byref(oi), C.byref(oj),
                                      resp.ctypes.data if want_resp else None, nthreads)
        return ((oi.value, oj.value), resp) if want_resp else (oi.value, oj.value)

    def detect_batch(self, frames, fill, K, radii, guesses, nthreads=0):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        K = np.asfortranarray(K, np.float64)
        g = np.ascontiguousarray(guesses, np.int32)
        out = np.empty((n, 2), np.int32)
        self.lib.pdo_detect_batch_dense(frames.ctypes.data, h * w, n, h, w, w, int(fill),
                                        K.ctypes.data, K.shape[0], radii[0], radii[1],
                                        g.ctypes.data, out.ctypes.data, nthreads)
        return out


    def detect_batch_par(self, frames, fill, K, sigma, darker, radii, guesses, separable=False, nthreads=0):
        """n windows, threads split across windows (each window single-threaded)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        K = np.asfortranarray(K, np.float64)
        g = np.ascontiguousarray(guesses, np.int32)
        out = np.empty((n, 2), np.int32)
        self.lib.pdo_detect_batch_par(frames.ctypes.data, h * w, n, h, w, w, int(fill), K.ctypes.data,
                                      float(sigma), int(bool(darker)), K.shape[0], radii[0], radii[1],
                                      g.ctypes.data, out.ctypes.data, int(bool(separable)), nthreads)
        return out


class OracleTracker:
    """The reference `Tracker` (src/PawsomeTracker.jl:32-62) on top of the C oracle."""

    def __init__(self, img, target_width, window_size, darker_target, oracle=None):
        self.o = oracle or Oracle()
        img = np.ascontiguousarray(img, np.uint8)
        self.sz = img.shape                                   # :40
        self.sigma = self.o.sigma(target_width)               # :41
        self.darker = bool(darker_target)                     # :42
        self.l = self.o.kernel_len(self.sigma)
        self.kernel = self.o.dog_kernel(self.sigma, darker_target, self.l)  # :43
        self.radii = (window_size[0] // 2, window_size[1] // 2)             # :44
        self.fill = self.o.mode_u8(img)                       # :47
        self.data = img.copy()                                # trckr.img.data, :166

    def __call__(self, guess, want_resp=False):
        return self.o.detect(self.data, self.fill, self.kernel, self.radii, guess, want_resp)
