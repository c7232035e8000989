#!/bin/bash
# usage: tools/ab3.sh "<lib1> <lib2> ..." [bench args]  — same-session comparison of several builds, two rounds, kernel_ms each
LIBS=$1; shift
for rep in 1 2 3; do
  for L in $LIBS; do
    PAWSOME_DOG_LIB=$PWD/$L python bench.py --no-cpu --steps 30 "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$L', 'ms_per_step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done
