"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/pawsome_dog.h declares, and its host-side helpers (Float64 scalar code, no kernels)
agree with the oracle.  No compute calls without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pawsometracker_jl_amd as pt
from pawsometracker_jl_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pawsome_dog.h")).read()
    declared = set(re.findall(r"\b(pdog_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert L.pdog_abi_version() == 1


def test_scalar_helpers_match_oracle(oracle):
    L = pt.lib()
    for tw in (5, 10, 16, 25, 40, 77.5, 120):
        assert L.pdog_sigma(tw) == oracle.sigma(tw)
        assert L.pdog_default_window(tw) == oracle.default_window(tw)
        l = L.pdog_kernel_len(tw)
        assert l == oracle.kernel_len(oracle.sigma(tw))
        for which, s in ((0, oracle.sigma(tw)), (1, oracle.sigma(tw) * 2 ** 0.5)):
            out = np.empty(l)
            assert L.pdog_gaussian_taps(tw, which, out.ctypes.data, l) == 0
            assert np.array_equal(out, oracle.gaussian_1d(s, l))
    assert L.pdog_gaussian_taps(25.0, 0, np.empty(3).ctypes.data, 3) == _lib.PDOG_E_ARG


def test_mode_matches_oracle(oracle, golden):
    for c in golden:
        assert pt.mode(c["frame"]) == c["fill"]
    assert pt.mode(np.array([[1, 2], [2, 1]], np.uint8)) == 2
    view = np.zeros((6, 10), np.uint8)[:, 2:7]     # strided rows
    view[:] = 9
    assert pt.mode(view) == 9


def test_host_bookkeeping():
    assert pt.fix_window_size((30, 50)) == (50, 30)       # (w,h) -> (h,w), :70
    assert pt.fix_window_size(45) == (45, 45)             # :72
    assert pt.guess_window_size(25) == 45 and pt.guess_window_size(120) == 205
    img = np.zeros((1080, 1920), np.uint8)
    assert pt.get_guess(None, img) == (540, 960)          # :86-90
    assert pt.get_guess(("ij", (5, 7)), img) == (5, 7)    # :74-77
    assert pt.get_guess((100, 40), img, sar=2.0) == (40, 50)   # (x,y) -> round.((y, x/sar)), :79-84


def test_shard_range_partitions():
    for n, w in ((4096, 8), (8192, 8), (10, 3), (3, 8), (0, 2)):
        spans = [pt.shard_range(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        sizes = [hi - lo for lo, hi in spans]
        assert max(sizes) - min(sizes) <= 1


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pt.PdogError) as e:
        pt.Tracker(np.full((32, 32), 128, np.uint8), 25, (45, 45), True)
    assert e.value.code == _lib.PDOG_E_NODEV and "no CPU path" in str(e.value)


def test_window_tile_is_the_padded_view(oracle):
    """pdog_window_tile (the host-side packer behind pdog_detect_host / pdog_detect_batch_host, no GPU involved)
    reproduces PaddedView(fill, img, …) (src/PawsomeTracker.jl:45-48) on the rectangle the functor reads:
    window ± l÷2, any guess up to l÷2 outside the frame, strided frames, padded output pitch."""
    L = pt.lib()
    rng = np.random.default_rng(3)
    for (fh, fw, tw, ws) in ((60, 80, 10, (21, 21)), (50, 40, 25, (45, 45)), (30, 30, 16, (9, 71)), (20, 90, 6, (63, 5)), (7, 5, 25, (45, 45))):
        wide = rng.integers(0, 256, (fh, fw + 13), dtype=np.uint8)
        frame = wide[:, :fw]                               # row stride fw + 13
        l = oracle.kernel_len(oracle.sigma(tw))
        hw, r1, r2 = l // 2, ws[0] // 2, ws[1] // 2
        th, tww = 2 * r1 + l, 2 * r2 + l                   # (2r+1) + (l-1)
        fill = 77
        padded = np.full((fh + 2 * (r1 + l), fw + 2 * (r2 + l)), fill, np.uint8)   # pad = radii + l per side, :45-46
        padded[r1 + l:r1 + l + fh, r2 + l:r2 + l + fw] = frame
        for g in [(-hw, -hw), (fh + hw + 1, fw + hw + 1), (1, 1), (fh, fw), (fh // 2, fw // 2), (-hw, fw), (fh + hw + 1, 1)] + \
                 [tuple(int(v) for v in rng.integers(-hw, [fh + hw + 2, fw + hw + 2])) for _ in range(20)]:
            pitch = tww + 11
            out = np.full((th, pitch), 200, np.uint8)
            gg = (C.c_int32 * 2)(*g)
            assert L.pdog_window_tile(frame.ctypes.data, fh, fw, frame.strides[0], fill, float(tw), ws[0], ws[1], gg,
                                      out.ctypes.data, pitch) == 0
            i0, j0 = g[0] - r1 - hw - 1 + (r1 + l), g[1] - r2 - hw - 1 + (r2 + l)    # 0-based origin in the padded array
            assert np.array_equal(out[:, :tww], padded[i0:i0 + th, j0:j0 + tww]), (fh, fw, tw, ws, g)
            assert (out[:, tww:] == fill).all()
    bad = (C.c_int32 * 2)(1, 1)
    assert L.pdog_window_tile(frame.ctypes.data, fh, fw, frame.strides[0], fill, 25.0, 45, 45, bad, out.ctypes.data, 10) == _lib.PDOG_E_ARG
    assert L.pdog_window_tile(None, fh, fw, fw, fill, 25.0, 45, 45, bad, out.ctypes.data, 200) == _lib.PDOG_E_ARG


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: the header must compile as C99 (and as C++) on its own, and a C translation unit
    that only includes it must link against the library by name."""
    import subprocess
    hdr = os.path.join(ROOT, "include", "pawsome_dog.h")
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", hdr])
    subprocess.check_call(["g++", "-x", "c++", "-fsyntax-only", "-Wall", "-Werror", hdr])
    src = tmp_path / "use.c"
    src.write_text('#include "pawsome_dog.h"\n#include <stdio.h>\n'
                   'int main(void) { printf("%d %d %.6f\\n", pdog_abi_version(), pdog_kernel_len(25.0), pdog_sigma(25.0)); return 0; }\n')
    exe = tmp_path / "use"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.check_output([str(exe)], text=True).split()
    assert out[0] == "1" and out[1] == "65" and abs(float(out[2]) - 10.616525) < 1e-5


def test_c_abi_shard_partition_matches_the_python_one():
    """pdog_shard_range / pdog_shard_owner (the partition pdog_group_* and its gather compaction use) against
    shard_range (what the torch.distributed path uses): same contiguous shards, and owner() inverts range()."""
    L = pt.lib()
    lo, hi, r, k = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    for n, w in ((4096, 8), (8192, 8), (10, 3), (3, 8), (7, 2), (1, 1), (0, 2), (1000, 7)):
        for rank in range(w):
            assert L.pdog_shard_range(n, w, rank, C.byref(lo), C.byref(hi)) == 0
            assert (lo.value, hi.value) == pt.shard_range(n, rank, w)
            for win in range(lo.value, hi.value):
                assert L.pdog_shard_owner(n, w, win, C.byref(r), C.byref(k)) == 0
                assert (r.value, k.value) == (rank, win - lo.value)
    assert L.pdog_shard_range(10, 0, 0, C.byref(lo), C.byref(hi)) == _lib.PDOG_E_ARG
    assert L.pdog_shard_owner(10, 3, 10, C.byref(r), C.byref(k)) == _lib.PDOG_E_ARG


def test_library_does_not_link_rccl():
    """RCCL is opened by pdog_group_create (dlopen), not linked: a host that only uses the single-device ABI can load the
    library where there is no librccl on the loader path."""
    import subprocess
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-d", pt.LIB_PATH], capture_output=True, text=True)
    if out.returncode != 0:
        out = subprocess.run(["readelf", "-d", pt.LIB_PATH], capture_output=True, text=True)
    needed = [l for l in out.stdout.splitlines() if "NEEDED" in l]
    assert needed and not any("rccl" in l for l in needed), needed


def test_group_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pt.PdogError) as e:
        pt.GroupTracker([0], 64, 64, 25, (45, 45), True, 128)
    assert e.value.code == _lib.PDOG_E_NODEV and "no CPU path" in str(e.value)


def test_bench_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` starts its own ranks; with fewer than N GPUs visible it must say so and exit
    non-zero before touching anything (round 1 died on an assert here)."""
    import subprocess
    import sys
    import torch
    n = torch.cuda.device_count() + 1
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 2 and f"--gpus {n} requested but only {n - 1} GPU(s) are visible" in p.stderr, (p.returncode, p.stderr[-500:])
    assert p.stdout.strip() == ""


def test_dense_kernel_is_bit_equal_to_the_oracle(oracle):
    """Exact mode re-decides near-ties with the reference's dense Float64 kernel (src/PawsomeTracker.jl:41-43); last
    bits decide those ties, so the table must equal the oracle's bit for bit (hipcc would contract a*b - c*d into an
    FMA by default)."""
    L = pt.lib()
    for tw in (5, 10, 25, 40, 120):
        for darker in (0, 1):
            l = L.pdog_kernel_len(float(tw))
            K = np.empty((l, l), np.float64, order="F")
            assert L.pdog_dense_kernel(float(tw), darker, K.ctypes.data, l * l) == 0
            ref = oracle.dog_kernel(oracle.sigma(tw), bool(darker), l)
            assert np.array_equal(K.view(np.uint64), ref.view(np.uint64)), (tw, darker)
    assert L.pdog_dense_kernel(25.0, 1, np.empty(4).ctypes.data, 4) == _lib.PDOG_E_ARG
