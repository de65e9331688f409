#!/bin/bash
# Developer script (GPU box): bench.py once per option setting given as arguments ("-" = defaults), one summary line each.
# An argument is a space-separated list of iwae_set_option switches: tools/dev/ab_bench.sh - no_bern_pipe=1 "wg16=160 no_side2=1" ...
# (BENCH_ARGS: extra bench.py arguments, e.g. "--config c2")
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  opts=""
  if [ "$e" != "-" ]; then for o in $e; do opts="$opts --opt $o"; done; fi
  python bench.py --no-cpu-baseline --no-llh-eval $opts $BENCH_ARGS 2> gpurun_out/ab.err | python3 -c "
import sys, json
for l in sys.stdin:
    if not l.startswith('{'): continue
    d = json.loads(l); r = d['roofline']
    print('%-28s ms/step %.4f  dominant %.2f us (%s)  %s  elbo %s' % ('$e', d['ms_per_step'], r['avg_launch_us'], r['timed_as'], ' '.join('%s=%.1f' % (k, v['us']) for k, v in r.get('all_kernels', {}).items()), d['config']['iwae_elbo_after']))
"
done
