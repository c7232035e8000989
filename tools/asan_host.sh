#!/bin/bash
# AddressSanitizer + UBSan over the HOST-only entry points of the C ABI (pdog_window_tile, pdog_mode_u8,
# pdog_gaussian_taps) on a CPU box: 3000 random geometries with frame buffers of exactly the bytes a strided
# frame owns, so any read past the frame or write past the tile is caught.  (GPU-side sanitizers are not
# available on the pool.)  usage: tools/asan_host.sh   — needs ~2 min for the instrumented build.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${TMPDIR:-/tmp}/pdog_asan; mkdir -p "$OUT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fno-slp-vectorize -fsanitize=address,undefined \
    -fno-omit-frame-pointer -o "$OUT/libpawsome_dog_asan.so" "$ROOT/pawsometracker.jl_amd/csrc/pawsome_dog.hip"
/opt/rocm/lib/llvm/bin/clang -fsanitize=address,undefined -g -I "$ROOT/include" "$ROOT/tools/asan_harness.c" -o "$OUT/asan_harness" \
    -L"$OUT" -l:libpawsome_dog_asan.so -Wl,-rpath,"$OUT" -Wl,-rpath,/opt/rocm/lib
ASAN_OPTIONS=detect_leaks=0 "$OUT/asan_harness"
