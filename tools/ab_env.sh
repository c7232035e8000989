#!/bin/bash
# usage: tools/ab_env.sh <bench flag(s) of variant B, quoted> [bench args]  — same-session A/B of one library path:
#   tools/ab_env.sh "--tuning no_fold=1" --workload cfg4        A = default, B = with the flag; A B A B, ms_per_step and kernel_ms each
B=$1; shift
for rep in 1 2; do
  for on in 0 1; do
    if [ $on = 1 ]; then X="$B"; else X=""; fi
    python bench.py --no-cpu --steps 30 "$@" $X 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('[$X]', 'ms_per_step %.4f kernel_ms %.4f' % (r['ms_per_step'], r['roofline']['kernel_ms']))"
  done
done
