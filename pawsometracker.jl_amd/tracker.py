"""Host-side mirror of the reference `Tracker` on top of the C ABI.

Same spellings as /root/reference/src/PawsomeTracker.jl so that call sites
(:94-95, :103-105, :166-167) and the parity tests read like the reference:

    trckr = Tracker(img, target_width, window_size, darker_target)   # :39-52
    trckr.img.data[...] = next_frame                                  # :166
    ij = trckr(guess)                                                 # :55-62

The arithmetic runs in the HIP library (libpawsome_dog.so); this file holds no
filter code and never imports oracle/.
"""
import ctypes as C

import numpy as np

from . import _lib


def get_sigma(target_width):
    """src/PawsomeTracker.jl:30"""
    return _lib.lib().pdog_sigma(float(target_width))


def guess_window_size(target_width):
    """src/PawsomeTracker.jl:64-68"""
    return _lib.lib().pdog_default_window(float(target_width))


def fix_window_size(window_size):
    """src/PawsomeTracker.jl:70-72: (w, h) -> (h, w); Int -> (l, l)"""
    if isinstance(window_size, (tuple, list)):
        w, h = window_size
        return (int(h), int(w))
    return (int(window_size), int(window_size))


def mode(img):
    """mode(_img), src/PawsomeTracker.jl:47 (StatsBase tie rule, column-major scan)."""
    img = np.asarray(img)
    if img.dtype != np.uint8 or img.ndim != 2:
        raise TypeError("frame must be a 2-D uint8 (GRAY8) array")
    if not img.flags.c_contiguous:
        img = np.ascontiguousarray(img)
    out = C.c_int()
    _lib.check(_lib.lib().pdog_mode_u8(img.ctypes.data, img.shape[0], img.shape[1], img.strides[0], C.byref(out)))
    return out.value


class _PaddedFrame:
    """Stands in for the PaddedView at :48: `.data` is the live frame buffer the
    caller overwrites in place each frame (:166); everything outside reads as `fillvalue`."""

    def __init__(self, data, fillvalue):
        self.data = data
        self.fillvalue = fillvalue


class Tracker:
    """src/PawsomeTracker.jl:32-62 — constructor (:39-52) and functor (:55-62)."""

    def __init__(self, img, target_width, window_size, darker_target, device=0):
        img = np.asarray(img)
        if img.dtype != np.uint8 or img.ndim != 2:
            raise TypeError("frame must be a 2-D uint8 (GRAY8) array")
        self.sz = (int(img.shape[0]), int(img.shape[1]))            # :40
        self.radii = (int(window_size[0]) // 2, int(window_size[1]) // 2)  # :44
        self.target_width = float(target_width)
        self.darker_target = bool(darker_target)
        fillvalue = mode(img)                                        # :47
        self.img = _PaddedFrame(np.array(img, dtype=np.uint8, order="C", copy=True), fillvalue)  # :48
        h = C.c_void_p()
        _lib.check(_lib.lib().pdog_create(int(device), self.sz[0], self.sz[1], self.target_width,
                                          int(window_size[0]), int(window_size[1]),
                                          int(self.darker_target), fillvalue, C.byref(h)))
        self._h = h
        self._resp = None

    # -- the functor, :55-62 --
    def __call__(self, guess, want_resp=False):
        g = (C.c_int32 * 2)(int(guess[0]), int(guess[1]))
        out = (C.c_int32 * 2)()
        data = self.img.data
        resp_ptr = None
        if want_resp:
            info = self.info()
            resp = np.empty((info.win_h, info.win_w), np.float32, order="F")
            resp_ptr = resp.ctypes.data
        _lib.check(_lib.lib().pdog_detect_host(self._h, data.ctypes.data, data.strides[0], g, out, resp_ptr))
        ij = (int(out[0]), int(out[1]))
        return (ij, resp) if want_resp else ij

    def info(self):
        o = _lib.PdogInfo()
        _lib.check(_lib.lib().pdog_get_info(self._h, C.byref(o)))
        return o

    def set_exact(self, on):
        """Exact mode (default on): near-ties of the FP32 ranking are re-decided in the reference's Float64 arithmetic."""
        _lib.check(_lib.lib().pdog_set_exact(self._h, int(on)))   # 0 off, 1 on, 2 re-evaluate everything (self-check)

    def set_tuning(self, key, value=1):
        """Pin one of the library's alternative code paths (pdog_set_tuning): tests and A/B only."""
        _lib.check(_lib.lib().pdog_set_tuning(self._h, key.encode(), int(value)))

    def exact_stats(self):
        """(on, threshold 2δ, windows re-evaluated so far) — pdog_get_exact."""
        on, thr, n = C.c_int(), C.c_double(), C.c_uint64()
        _lib.check(_lib.lib().pdog_get_exact(self._h, C.byref(on), C.byref(thr), C.byref(n)))
        return bool(on.value), thr.value, int(n.value)

    def exact_detail(self):
        """(windows refined, column blocks rescanned, candidates, sequential chains) — pdog_get_exact_detail."""
        out = (C.c_uint64 * 4)()
        _lib.check(_lib.lib().pdog_get_exact_detail(self._h, out))
        return tuple(int(v) for v in out)

    def set_variant(self, variant):
        _lib.check(_lib.lib().pdog_set_variant(self._h, int(variant)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().pdog_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def get_guess(start_location, img, sar=1.0):
    """src/PawsomeTracker.jl:74-90: CartesianIndex -> Tuple; (x, y) -> round.((y, x/sar));
    missing (None) -> size(img) .÷ 2.  A CartesianIndex is spelled `("ij", (i, j))` here,
    a bare 2-tuple is (x, y) like the reference's NTuple{2}."""
    if start_location is None:
        return (img.shape[0] // 2, img.shape[1] // 2)          # :86-90
    if isinstance(start_location, tuple) and len(start_location) == 2 and start_location[0] == "ij":
        return (int(start_location[1][0]), int(start_location[1][1]))   # :74-77
    x, y = start_location                                       # :79-84
    return (int(_round_half_even(y)), int(_round_half_even(x / sar)))


def _round_half_even(v):
    return int(np.rint(v))  # Julia's round(Int, x) rounds half to even, like rint


def get_start_ij_and_tracker(start_location, img, target_width, window_size, darker_target, sar=1.0, device=0):
    """src/PawsomeTracker.jl:92-107, both methods."""
    guess = get_guess(start_location, img, sar)
    if start_location is None:                                   # :99-107 auto-detect
        sz = img.shape
        window_size2 = (sz[0] // 4, sz[1] // 4)                  # :102
        trckr = Tracker(img, target_width, window_size2, darker_target, device)   # :103
        ij = trckr(guess)                                        # :104
        trckr.close()
        trckr = Tracker(img, target_width, window_size, darker_target, device)    # :105
        return trckr, ij
    trckr = Tracker(img, target_width, window_size, darker_target, device)        # :94
    return trckr, trckr(guess)                                   # :95


def track_frames(frames, target_width=25, start_location=None, window_size=None, darker_target=True,
                 sar=1.0, device=0):
    """The frame loop of track_one (src/PawsomeTracker.jl:159-169) on already-decoded GRAY8
    frames (decode stays with the host application): indices[1] from the bootstrap (:161),
    then indices[k] = trckr(indices[k-1]) per frame (:166-167, the intended loop).
    `frames` is an iterable of h x w uint8 arrays.  Returns a list of 1-based (row, col)."""
    if window_size is None:
        window_size = guess_window_size(target_width)            # :136
    window_size = fix_window_size(window_size)                   # :142
    it = iter(frames)
    img = np.asarray(next(it))                                   # :159
    trckr, ij = get_start_ij_and_tracker(start_location, img, target_width, window_size, darker_target, sar, device)
    indices = [ij]
    try:
        for frame in it:
            trckr.img.data[...] = frame                          # :166
            indices.append(trckr(indices[-1]))                   # :167
    finally:
        trckr.close()
    return indices
