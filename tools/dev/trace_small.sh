#!/bin/bash
# Developer script (GPU box): kernel timeline of one small-batch train step (dataset path, B=20, k=5).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_small_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/runs.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np
from iwae_amd.native import NativeModel
from iwae_amd import utils
rng = np.random.default_rng(0)
m = NativeModel(1, 200, 100, seed=1)
m.dataset_upload((rng.random((60000, 784)) * 255).astype(np.uint8)); m.dataset_begin_epoch(0, rng.permutation(60000))
for i in range(300): m.train_step_dataset(i * 20, 20, 5, 1.0, 1e-3, "iwae_elbo", scalars=False)
m.sync()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/runs.py > $OUT/log.txt 2>&1 || echo failed
python3 $R/tools/dev/timeline.py $OUT 250
