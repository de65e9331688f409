#!/bin/bash
# Developer script (GPU box): how the --steps 20 (driver-shaped) number depends on the untimed settling run in front of it.
cd $GRAFT_REPO_ROOT
for s in 100 100 400 400 1000 1000 100; do python bench.py --steps 20 --warmup 5 --settle $s --no-kernel-times --no-cpu-baseline --no-llh-eval 2>/dev/null | python3 -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('settle $s steps', d['steps'], 'ms/step', d['ms_per_step'])"; done
