#!/bin/bash
# Developer script (GPU box): bench.py ms/step of the in-tree library against other builds of it, REPS interleaved rounds (boxes differ by ~4 %:
# only same-box, interleaved runs rank builds).  usage: tools/dev/ab_lib.sh build/lib_r4.so [more.so ...]     env: BENCH_ARGS, REPS (default 3)
R=${GRAFT_REPO_ROOT:-.}
REPS=${REPS:-3}
VARS=("-" "$@")
declare -A RES
for r in $(seq 1 $REPS); do
  for v in "${VARS[@]}"; do
    lib=""; if [ "$v" != "-" ]; then lib="--lib $R/$v"; fi
    ms=$(python3 $R/bench.py --steps 150 --warmup 30 --no-kernel-times --no-cpu-baseline --no-llh-eval $lib $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): print(json.loads(l)['ms_per_step'])")
    RES[$v]="${RES[$v]} $ms"
  done
done
for v in "${VARS[@]}"; do
  python3 -c "
import sys
x = sorted(float(t) for t in sys.argv[2:])
print('%-40s min %.4f  median %.4f  max %.4f ms' % (sys.argv[1], x[0], x[len(x)//2], x[-1]))" "$v" ${RES[$v]}
done
