#!/bin/bash
# Developer script (GPU box): bench.py once per environment setting given as arguments ("-" = default build), one summary line each.
# usage: tools/dev/ab_bench.sh - IWAE_NO_BERN_PIPE=1 ...
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  if [ "$e" = "-" ]; then envs=""; else envs="$e"; fi
  env $envs python bench.py --no-cpu-baseline 2> gpurun_out/ab.err | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); r = d['roofline']
    print('%-28s ms/step %.4f  dominant %.2f us (%s)  other %s  elbo %s' % ('$e', d['ms_per_step'], r['avg_launch_us'], r['kernel'][:24], r['other'], d['config']['iwae_elbo_after']))
"
done
