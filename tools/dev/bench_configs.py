"""Developer script (GPU box): step times of the other BASELINE.json configs + the k=5000 LLH evaluator."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from iwae_amd.native import NativeModel
from iwae_amd import utils

p = utils.synthetic_pixel_means()
rng = np.random.default_rng(0)


def batch(n):
    return (rng.random((n, 784)) < p[None]).astype(np.float32)


def time_steps(m, x, k, obj, steps, warm=10):
    for _ in range(warm):
        m.train_step(x, k, 1.0, 1e-3, obj, scalars=False)
    m.sync(); t0 = time.perf_counter()
    for _ in range(steps):
        m.train_step(x, k, 1.0, 1e-3, obj, scalars=False)
    m.sync()
    return (time.perf_counter() - t0) / steps


rows = []
m1 = NativeModel(1, 200, 100, seed=1); m1.set_output_bias(utils.bias_from_mean(p))
for (B, k, obj, steps) in [(20, 1, "vae_elbo", 500), (20, 5, "iwae_elbo", 500), (20, 50, "iwae_elbo", 300), (1024, 50, "iwae_elbo", 100), (1024, 50, "dreg", 100), (1024, 50, "vae_elbo_kl", 100), (1024, 5, "iwae_elbo", 200)]:
    dt = time_steps(m1, batch(B), k, obj, steps)
    rows.append(("1-layer", B, k, obj, dt))
m2 = NativeModel(2, [200, 100], [100, 50], seed=1); m2.set_output_bias(utils.bias_from_mean(p))
for (B, k, obj, steps) in [(20, 5, "iwae_elbo", 300), (1024, 50, "iwae_elbo", 60)]:
    dt = time_steps(m2, batch(B), k, obj, steps)
    rows.append(("2-layer", B, k, obj, dt))
print("%-8s %6s %5s %-12s %10s %14s" % ("model", "B", "k", "objective", "ms/step", "images/s"))
for r in rows:
    print("%-8s %6d %5d %-12s %10.4f %14.0f" % (r[0], r[1], r[2], r[3], r[4] * 1e3, r[1] / r[4]))
# LLH evaluator (main.py:170-184): N test images, k = 5000
for N, chunk in [(2000, 0), (2000, 32)]:
    x = batch(N)
    m1.eval_llh(x[:64], 5000)
    m1.sync(); t0 = time.perf_counter()
    llh = m1.eval_llh(x, 5000, chunk)
    dt = time.perf_counter() - t0
    print("eval_llh 1-layer N=%d k=5000 chunk=%d: %.3f s (%.0f images/s, %.1f TFLOP/s fwd GEMM) llh=%.3f  -> 10 000 images: %.2f s" % (N, chunk, dt, N / dt, N * 2.168e9 / dt / 1e12, llh, dt * 10000 / N))
x = batch(10000)
m1.sync(); t0 = time.perf_counter(); r = m1.forward(x, 50); dt = time.perf_counter() - t0
print("val_step on 10 000 images, k=50 (main.py:152): %.3f s" % dt)
