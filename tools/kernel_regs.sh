#!/bin/bash
# usage: tools/kernel_regs.sh <object-or-so> [name-filter]  — VGPR/SGPR/spill/scratch/LDS figures of every gfx950 kernel in a code object
F=$1; PAT=${2:-.}
TMP=$(mktemp -d)
B=/opt/rocm/lib/llvm/bin
$B/llvm-objcopy --dump-section .hip_fatbin=$TMP/fat.bin $F 2>/dev/null
T=$($B/clang-offload-bundler --list --type=o --input=$TMP/fat.bin 2>/dev/null | grep gfx950 | head -1)
[ -n "$T" ] && $B/clang-offload-bundler --unbundle --type=o --input=$TMP/fat.bin --targets=$T --output=$TMP/dev.co 2>/dev/null
[ -s $TMP/dev.co ] || { echo "no gfx950 bundle in $F"; exit 1; }
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.co | python3 -c "
import sys,re
txt=sys.stdin.read()
for blk in txt.split('  - .agpr_count:')[1:]:
    g=lambda k:(re.search(r'\.'+k+r':\s+(\S+)',blk) or [None,'?'])[1]
    name=g('name')
    if re.search(sys.argv[1],name): print(f\"{name[:90]:90s} vgpr={g('vgpr_count'):>4} sgpr={g('sgpr_count'):>4} vspill={g('vgpr_spill_count'):>3} sspill={g('sgpr_spill_count'):>3} scratch={g('private_segment_fixed_size'):>5} lds={g('group_segment_fixed_size'):>6}\")
" "$PAT"
rm -rf $TMP
