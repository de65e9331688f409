"""Developer script (GPU box): step / evaluator times of the model variants outside bench.py's configs (2-layer evaluator, conditional models).
usage: python tools/dev/variant_times.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel

def timeit(f, n):
    f(); f()
    t = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t) / n

B, k = 1024, 50
x = O.synthetic_binarized(B, 1)
y = np.eye(10, dtype=np.float32)[np.arange(B) % 10]
for name, kw in (("1-layer", {}), ("cond (task05)", {"cond_dim": 10}), ("cond prior (task04)", {"cond_dim": 10, "cond_prior": True})):
    m = NativeModel(1, 200, 100, seed=5, **kw)
    if kw: m.set_condition(y)
    def step(): m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
    def sync_step(): step(); m.sync()
    for _ in range(20): step()
    m.sync()
    t = time.perf_counter()
    for _ in range(100): step()
    m.sync()
    print("%-22s train step %.4f ms" % (name, (time.perf_counter() - t) * 10))
    m.close()
for layers, nh, nl in ((1, 200, 100), (2, [200, 100], [100, 50])):
    m = NativeModel(layers, nh, nl, seed=5)
    xe = O.synthetic_binarized(1000, 2)
    for prec in ("bf16", "fp32"):
        m.set_eval_precision(prec)
        m.eval_llh(xe[:32], 5000); m.sync()
        t = time.perf_counter(); m.eval_llh(xe, 5000); dt = time.perf_counter() - t
        print("%d-layer evaluator %s: %.0f images/s" % (layers, prec, 1000 / dt))
    m.close()
