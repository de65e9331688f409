#!/bin/bash
# Developer script (GPU box): the data-parallel step's code path on ONE GPU (world size 1, RCCL collectives are trivial):
# what the exchange costs before any real communication.  usage: tools/dev/dp_rehearsal.sh [option settings ...] ("-" = defaults)
cd $GRAFT_REPO_ROOT
port=29520
for e in "$@"; do
  opts=""
  if [ "$e" != "-" ]; then for o in $e; do opts="$opts --opt $o"; done; fi
  port=$((port+1))
  env MASTER_ADDR=127.0.0.1 MASTER_PORT=$port RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 \
    timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --no-llh-eval $opts 2> gpurun_out/dp.err | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('%-24s forced-dist ms/step %.4f' % ('$e', d['ms_per_step']))
"
done
