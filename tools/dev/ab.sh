#!/bin/bash
# Developer script (GPU box): bench.py ms/step under a list of environment variants.  usage: tools/dev/ab.sh "VAR=1" "A=2 B=3" ...
# (the first line is the default build); env BENCH_ARGS adds bench.py arguments
R=${GRAFT_REPO_ROOT:-.}
run() {
  env $1 python3 $R/bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-llh-eval $BENCH_ARGS 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d.get('roofline',{}).get('all_kernels',{})
print('%-44s %.4f ms  %s' % ('$1' or 'default', d['ms_per_step'], ' '.join('%s=%.0f' % (k[:12], v['us']) for k, v in r.items())))"
}
run "IWAE_AB_DEFAULT=1"
for v in "$@"; do run "$v"; done
