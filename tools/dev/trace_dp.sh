#!/bin/bash
# Developer script (GPU box): kernel timeline of one DATA-PARALLEL step rehearsed on one GPU (world size 1).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_dp_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --force-dist --steps 60 --warmup 10 --settle 0 --no-kernel-times --no-cpu-baseline --no-llh-eval > $OUT/log.txt 2>&1 || echo failed
python3 $R/tools/dev/timeline.py $OUT
