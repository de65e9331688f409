"""Developer script (GPU box): host enqueue time vs end-to-end time per train step (is a configuration host-bound?).
usage: python tools/dev/host_time.py B k [layers] [option=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from iwae_amd.native import NativeModel
B, k = int(sys.argv[1]), int(sys.argv[2])
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 1
opts = dict((kv.split('=')[0], int(kv.split('=')[1])) for kv in sys.argv[4:])
m = NativeModel(layers, 200 if layers == 1 else [200, 100], 100 if layers == 1 else [100, 50], seed=1, options=opts)
x = torch.tensor((np.random.default_rng(0).random((B, 784)) < 0.2).astype(np.float32), device="cuda")
for _ in range(50):
    m.train_step_devptr(x.data_ptr(), B, k, 1.0, 1e-3, 1)
m.sync()
N = 400
t0 = time.perf_counter()
for _ in range(N):
    m.train_step_devptr(x.data_ptr(), B, k, 1.0, 1e-3, 1)
t1 = time.perf_counter()
m.sync()
t2 = time.perf_counter()
print("B=%d k=%d %s: host enqueue %.1f us/step, end-to-end %.1f us/step" % (B, k, opts, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6))
