#!/bin/bash
# Developer script (GPU box): bench.py ms/step under a list of environment variants, REPS interleaved rounds each (run-to-run
# spread on a box is ~3 %: single runs cannot rank variants closer than that).  usage: tools/dev/ab.sh "VAR=1" "A=2 B=3" ...
# (the first line is the default build); env BENCH_ARGS adds bench.py arguments, REPS (default 3) the rounds
R=${GRAFT_REPO_ROOT:-.}
REPS=${REPS:-3}
VARS=("IWAE_AB_DEFAULT=1" "$@")
declare -A RES
for r in $(seq 1 $REPS); do
  for v in "${VARS[@]}"; do
    ms=$(env $v python3 $R/bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-llh-eval $BENCH_ARGS 2>/dev/null | python3 -c "
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
