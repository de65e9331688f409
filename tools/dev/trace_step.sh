#!/bin/bash
# Developer script (GPU box): kernel trace of the bench loop, then the timeline of one step.
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 60 --warmup 10 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval $BENCH_ARGS > $OUT/log.txt 2>&1 || echo failed
python3 $R/tools/dev/timeline.py $OUT
