"""Developer script (GPU box): quick step timing at the C2 shape. Not a pytest."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
B, k = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 50
nl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
x = O.synthetic_binarized(B, 1)
m = NativeModel(nl, [200, 100][:nl] if nl == 2 else 200, [100, 50][:nl] if nl == 2 else 100, seed=5)
m.set_output_bias(O.output_bias_from_mean(O.synthetic_pixel_means()))
for i in range(5):
    r = m.train_step(x, k, 1.0, 1e-3, "iwae_elbo")
print("warm", r["iwae_elbo"])
m.sync(); t0 = time.time()
for i in range(steps):
    m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
m.sync(); dt = (time.time() - t0) / steps
r = m.forward(x, k)
print("B=%d k=%d layers=%d: %.3f ms/step  %.0f images/s  iwae_elbo after %d steps: %.3f" % (B, k, nl, dt * 1e3, B / dt, steps + 5, r["iwae_elbo"]))
