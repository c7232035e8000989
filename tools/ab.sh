#!/bin/bash
# usage: tools/ab.sh <libA.so> <libB.so> [bench args]  — same-session A/B of two builds (boards differ by ≈8 %): A B A B, kernel_ms each
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in $A $B; do
    PAWSOME_DOG_LIB=$PWD/$L python bench.py --no-cpu --steps 30 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$L', 'ms_per_step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done
