#!/bin/bash
# Developer script (GPU box): bench.py ms/step under a list of option variants (iwae_set_option switches), REPS interleaved rounds each (run-to-run
# spread on a box is ~3 %: single runs cannot rank variants closer than that).  usage: tools/dev/ab.sh "name=1" "a=2 b=3" ...
# (the first line is the default build); env BENCH_ARGS adds bench.py arguments, REPS (default 3) the rounds
R=${GRAFT_REPO_ROOT:-.}
REPS=${REPS:-3}
VARS=("-" "$@")
declare -A RES
for r in $(seq 1 $REPS); do
  for v in "${VARS[@]}"; do
    opts=""; if [ "$v" != "-" ]; then for o in $v; do opts="$opts --opt $o"; done; fi
    ms=$(python3 $R/bench.py --steps 150 --warmup 30 --no-kernel-times --no-cpu-baseline --no-llh-eval $opts $BENCH_ARGS 2>/dev/null | python3 -c "
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
print('%-50s min %.4f  median %.4f  max %.4f ms' % (sys.argv[1], x[0], x[len(x)//2], x[-1]))" "$v" ${RES[$v]}
done
