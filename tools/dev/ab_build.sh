#!/bin/bash
# Developer script (GPU box): A/B of two BUILDS of the library, interleaved: tools/dev/_old/libiwae_amd.so (built from the previous
# commit by hand) against the in-tree one.  usage: tools/dev/ab_build.sh [bench args]
R=${GRAFT_REPO_ROOT:-.}
cp $R/iwae_amd/libiwae_amd.so /tmp/new.so
REPS=${REPS:-3}
for r in $(seq 1 $REPS); do
  for v in old new; do
    if [ $v = old ]; then cp $R/tools/dev/_old/libiwae_amd.so $R/iwae_amd/libiwae_amd.so; else cp /tmp/new.so $R/iwae_amd/libiwae_amd.so; fi
    ms=$(python3 $R/bench.py --steps 150 --warmup 30 --no-cpu-baseline --no-llh-eval "$@" 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); k = d['roofline'].get('all_kernels', {})
        print(d['ms_per_step'], ' '.join('%s=%.1f' % (n, v['us']) for n, v in k.items()))")
    echo "$v $ms"
  done
done
cp /tmp/new.so $R/iwae_amd/libiwae_amd.so
