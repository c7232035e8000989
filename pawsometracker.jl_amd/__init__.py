"""MI355X-native DoG + argmax hot path behind PawsomeTracker's `Tracker` API.

Directory name follows the build contract (`pawsometracker.jl_amd/`); import it
as `pawsometracker_jl_amd` (the loader stub at the repo root registers it).
"""
from ._lib import LIB_PATH, PdogError, lib  # noqa: F401
from .tracker import (Tracker, fix_window_size, get_guess, get_sigma,  # noqa: F401
                      get_start_ij_and_tracker, guess_window_size, mode, track_frames)
from .batch import BatchTracker, GroupTracker, gather_positions, mode_device, shard_range  # noqa: F401
