"""ctypes binding of include/pawsome_dog.h (the C-ABI drop-in boundary).

The HIP extension is the product path: if libpawsome_dog.so is missing this
module raises — there is no CPU fallback and nothing here touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PAWSOME_DOG_LIB selects another build of the same library (tuning builds with ablation variants)
LIB_PATH = os.environ.get("PAWSOME_DOG_LIB") or os.path.join(_HERE, "libpawsome_dog.so")

PDOG_OK, PDOG_E_ARG, PDOG_E_HIP, PDOG_E_NODEV, PDOG_E_RANGE, PDOG_E_ALLOC = range(6)

# every symbol include/pawsome_dog.h declares (tests check the library exports them all)
SYMBOLS = (
    "pdog_abi_version", "pdog_last_error", "pdog_sigma", "pdog_default_window", "pdog_kernel_len",
    "pdog_gaussian_taps", "pdog_mode_u8", "pdog_mode_u8_device", "pdog_create", "pdog_destroy", "pdog_get_info",
    "pdog_set_fill", "pdog_set_stream", "pdog_reserve", "pdog_set_variant", "pdog_kernel_for_batch", "pdog_sync",
    "pdog_detect_batch", "pdog_detect_host", "pdog_window_tile", "pdog_detect_batch_host", "pdog_detect_chain", "pdog_detect_chains",
    "pdog_alloc_host", "pdog_free_host", "pdog_detect_chain_progress", "pdog_get_stream",
    "pdog_group_create", "pdog_group_destroy", "pdog_group_size", "pdog_group_tracker", "pdog_group_shard",
    "pdog_group_detect_batch", "pdog_group_sync", "pdog_shard_range", "pdog_shard_owner", "pdog_group_test_compact",
    "pdog_set_exact", "pdog_get_exact", "pdog_get_exact_detail", "pdog_dense_kernel", "pdog_set_tuning",
)


class PdogInfo(C.Structure):
    _fields_ = [
        ("frame_h", C.c_int32), ("frame_w", C.c_int32),
        ("radius_h", C.c_int32), ("radius_w", C.c_int32),
        ("win_h", C.c_int32), ("win_w", C.c_int32),
        ("kernel_len", C.c_int32), ("fill", C.c_int32), ("darker_target", C.c_int32),
        ("n_strips", C.c_int32), ("strip_w", C.c_int32), ("variant", C.c_int32),
        ("sigma", C.c_double), ("target_width", C.c_double),
        ("algorithmic_bytes_per_window", C.c_int64), ("algorithmic_fma_per_window", C.c_int64),
    ]


class PdogError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pawsome_dog status {code}: {msg}")
        self.code = code


_lib = None


def _preload_torch_hip_runtime():
    """PyTorch-ROCm ships its own libamdhip64 / libhsa-runtime64 / librccl (same SONAMEs as /opt/rocm's) and opens
    them by path.  If this library were loaded first it would bind to /opt/rocm's copies, `import torch` would then
    bring in a second HIP runtime and a second RCCL, and only the runtime initialised first gets the GPU (seen as
    "no HIP device" from the other).  So when torch is installed it is imported FIRST: its copies are then the ones
    in the process and this library's NEEDED entries resolve to them by SONAME.  (Loading torch's libraries one by
    one instead — as round 1 did for the HIP runtime — breaks their destructor order once librccl is among them:
    `double free` at interpreter exit.)  Without torch nothing happens and /opt/rocm's libraries are used, as for
    any C or Julia host."""
    import sys
    if "torch" in sys.modules:
        return
    try:
        import importlib.util
        if importlib.util.find_spec("torch") is None:
            return
        import torch  # noqa: F401
    except (ImportError, ValueError, OSError):
        return


def lib():
    """Load the shared library once; fail loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    i, d, p, i64 = C.c_int, C.c_double, C.c_void_p, C.c_int64
    L.pdog_abi_version.restype = i
    L.pdog_last_error.restype = C.c_char_p
    L.pdog_sigma.restype = d; L.pdog_sigma.argtypes = [d]
    L.pdog_default_window.restype = i; L.pdog_default_window.argtypes = [d]
    L.pdog_kernel_len.restype = i; L.pdog_kernel_len.argtypes = [d]
    L.pdog_gaussian_taps.restype = i; L.pdog_gaussian_taps.argtypes = [d, i, p, i]
    L.pdog_mode_u8.restype = i; L.pdog_mode_u8.argtypes = [p, i, i, i64, C.POINTER(i)]
    if hasattr(L, "pdog_mode_u8_device"):
        L.pdog_mode_u8_device.restype = i; L.pdog_mode_u8_device.argtypes = [i, p, i, i, i64, p, C.POINTER(i)]
    L.pdog_create.restype = i; L.pdog_create.argtypes = [i, i, i, d, i, i, i, i, C.POINTER(p)]
    L.pdog_destroy.restype = i; L.pdog_destroy.argtypes = [p]
    L.pdog_get_info.restype = i; L.pdog_get_info.argtypes = [p, C.POINTER(PdogInfo)]
    L.pdog_set_fill.restype = i; L.pdog_set_fill.argtypes = [p, i]
    L.pdog_set_stream.restype = i; L.pdog_set_stream.argtypes = [p, p]
    L.pdog_reserve.restype = i; L.pdog_reserve.argtypes = [p, i]
    L.pdog_set_variant.restype = i; L.pdog_set_variant.argtypes = [p, i]
    L.pdog_sync.restype = i; L.pdog_sync.argtypes = [p]
    L.pdog_detect_batch.restype = i
    L.pdog_detect_batch.argtypes = [p, p, i64, i64, i, p, p, i, p, p]
    L.pdog_detect_host.restype = i; L.pdog_detect_host.argtypes = [p, p, i64, p, p, p]
    L.pdog_detect_chain.restype = i; L.pdog_detect_chain.argtypes = [p, p, i64, i64, i, p, p]
    if hasattr(L, "pdog_kernel_for_batch"):
        L.pdog_kernel_for_batch.restype = i; L.pdog_kernel_for_batch.argtypes = [p, i, C.POINTER(i)]
    if hasattr(L, "pdog_window_tile"):
        L.pdog_window_tile.restype = i; L.pdog_window_tile.argtypes = [p, i, i, i64, i, d, i, i, p, p, i64]
    if hasattr(L, "pdog_detect_batch_host"):
        L.pdog_detect_batch_host.restype = i; L.pdog_detect_batch_host.argtypes = [p, p, i64, i64, i, p, p, i, p]
    if hasattr(L, "pdog_detect_chain_progress"):
        L.pdog_alloc_host.restype = i; L.pdog_alloc_host.argtypes = [C.c_size_t, C.POINTER(p)]
        L.pdog_free_host.restype = i; L.pdog_free_host.argtypes = [p]
        L.pdog_detect_chain_progress.restype = i; L.pdog_detect_chain_progress.argtypes = [p, p, i64, i64, i, p, p, p]
    if hasattr(L, "pdog_detect_chains"):  # absent only in older A/B builds selected through PAWSOME_DOG_LIB
        L.pdog_detect_chains.restype = i; L.pdog_detect_chains.argtypes = [p, p, i64, i64, i, i, p, p]
    if hasattr(L, "pdog_group_create"):
        L.pdog_get_stream.restype = i; L.pdog_get_stream.argtypes = [p, C.POINTER(p)]
        L.pdog_group_create.restype = i; L.pdog_group_create.argtypes = [i, p, i, i, d, i, i, i, i, C.POINTER(p)]
        L.pdog_group_destroy.restype = i; L.pdog_group_destroy.argtypes = [p]
        L.pdog_group_size.restype = i; L.pdog_group_size.argtypes = [p]
        L.pdog_group_tracker.restype = i; L.pdog_group_tracker.argtypes = [p, i, C.POINTER(p)]
        L.pdog_group_shard.restype = i; L.pdog_group_shard.argtypes = [p, i, i, C.POINTER(i), C.POINTER(i)]
        L.pdog_group_detect_batch.restype = i; L.pdog_group_detect_batch.argtypes = [p, p, i64, i64, p, p, p, i, p]
        L.pdog_group_sync.restype = i; L.pdog_group_sync.argtypes = [p]
        L.pdog_shard_range.restype = i; L.pdog_shard_range.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
        L.pdog_shard_owner.restype = i; L.pdog_shard_owner.argtypes = [i, i, i, C.POINTER(i), C.POINTER(i)]
        if hasattr(L, "pdog_group_test_compact"):   # (absent from older builds loaded through PAWSOME_DOG_LIB for an A/B)
            L.pdog_group_test_compact.restype = i; L.pdog_group_test_compact.argtypes = [p, i, i, p]
    if hasattr(L, "pdog_dense_kernel"):
        L.pdog_dense_kernel.restype = i; L.pdog_dense_kernel.argtypes = [d, i, p, i]
    if hasattr(L, "pdog_set_tuning"):
        L.pdog_set_tuning.restype = i; L.pdog_set_tuning.argtypes = [p, C.c_char_p, i]
    if hasattr(L, "pdog_set_exact"):
        L.pdog_set_exact.restype = i; L.pdog_set_exact.argtypes = [p, i]
        L.pdog_get_exact.restype = i; L.pdog_get_exact.argtypes = [p, C.POINTER(i), C.POINTER(d), C.POINTER(C.c_uint64)]
        L.pdog_get_exact_detail.restype = i; L.pdog_get_exact_detail.argtypes = [p, C.POINTER(C.c_uint64)]
    _lib = L
    return L


def check(code):
    if code != PDOG_OK:
        raise PdogError(code, lib().pdog_last_error().decode("utf-8", "replace"))
