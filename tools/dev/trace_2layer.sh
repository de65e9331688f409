#!/bin/bash
# Developer script (GPU box): kernel timeline of one 2-layer train step (B=1024, k=50).
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace2_${1:-x}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/run2.py <<PY
import sys; sys.path.insert(0, "$R")
import numpy as np
from iwae_amd.native import NativeModel
from iwae_amd import utils
p = utils.synthetic_pixel_means(); rng = np.random.default_rng(0)
x = (rng.random((1024, 784)) < p[None]).astype(np.float32)
import torch
xd = torch.tensor(x, device="cuda")
m = NativeModel(2, [200, 100], [100, 50], seed=1); m.set_output_bias(utils.bias_from_mean(p))
for i in range(40):
    m.train_step_devptr(xd.data_ptr(), 1024, 50, 1.0, 1e-3, 1)
m.sync()
PY
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/run2.py > $OUT/log.txt 2>&1 || echo failed
python3 $R/tools/dev/timeline.py $OUT
