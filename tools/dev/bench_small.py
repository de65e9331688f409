import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from iwae_amd.native import NativeModel
from iwae_amd import utils
p = utils.synthetic_pixel_means(); rng = np.random.default_rng(0)
def batch(n): return (rng.random((n, 784)) < p[None]).astype(np.float32)
m1 = NativeModel(1, 200, 100, seed=1); m1.set_output_bias(utils.bias_from_mean(p))
for (B, k, obj) in [(20, 5, "iwae_elbo"), (20, 1, "vae_elbo"), (20, 1, "iwae_elbo"), (20, 5, "iwae_elbo"), (20, 1, "vae_elbo")]:
    x = batch(B)
    for _ in range(50): m1.train_step(x, k, 1.0, 1e-3, obj, scalars=False)
    m1.sync(); t0 = time.perf_counter()
    for _ in range(500): m1.train_step(x, k, 1.0, 1e-3, obj, scalars=False)
    m1.sync(); dt = (time.perf_counter() - t0) / 500
    print(B, k, obj, "%.4f ms" % (dt * 1e3))
# resident dataset path (what main.py uses): no host traffic per step
g = (rng.random((60000, 784)) * 255).astype(np.uint8)
m1.dataset_upload(g)
m1.dataset_begin_epoch(0, rng.permutation(60000))
for (B, k) in [(20, 1), (20, 5), (20, 50), (100, 5)]:
    for i in range(50): m1.train_step_dataset(i * B, B, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
    m1.sync(); t0 = time.perf_counter()
    for i in range(500): m1.train_step_dataset((i * B) % 59000, B, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
    m1.sync(); dt = (time.perf_counter() - t0) / 500
    print("dataset path", B, k, "%.4f ms  %.0f images/s" % (dt * 1e3, B / dt))
