"""Developer script (GPU box): end-to-end time per train step for a (B, k) shape.  usage: python tools/dev/step_time.py B k [layers] [objective id]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from iwae_amd.native import NativeModel
B, k = int(sys.argv[1]), int(sys.argv[2])
layers = int(sys.argv[3]) if len(sys.argv) > 3 else 1
obj = int(sys.argv[4]) if len(sys.argv) > 4 else 1
m = NativeModel(layers, 200 if layers == 1 else [200, 100], 100 if layers == 1 else [100, 50], seed=1)
x = torch.tensor((np.random.default_rng(0).random((B, 784)) < 0.2).astype(np.float32), device="cuda")
best = 1e9
for rep in range(3):
    for _ in range(40):
        m.train_step_devptr(x.data_ptr(), B, k, 1.0, 1e-3, obj)
    m.sync()
    N = 300
    t0 = time.perf_counter()
    for _ in range(N):
        m.train_step_devptr(x.data_ptr(), B, k, 1.0, 1e-3, obj)
    m.sync()
    best = min(best, (time.perf_counter() - t0) / N * 1e6)
print("%-40s B=%d k=%d layers=%d: %.1f us/step" % (os.environ.get("TAG", ""), B, k, layers, best))
