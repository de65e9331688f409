"""Developer script (GPU box, library built with STAMPS=1): phase cycle shares inside bern_pipe_kernel (the whole decoder forward).
usage: python tools/dev/dbg_bern_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
B, k = 1024, 50
x = O.synthetic_binarized(B, 1)
m = NativeModel(1, 200, 100, seed=5, options={"dense_stamps_epi": 4, "dense_stamps_kt": 7})      # diagnostic option names: STAMPS=1 build only
for i in range(10):
    m.forward_backward(x, k, 1.0, "iwae_elbo")
s = m.debug_tensor("dense_stamps").astype(np.float64)
s = s[s.sum(1) > 0]
names = ["prologue+z", "layer 1", "layer 2", "fill", "main compute", "tail", "end", "main barrier"]
tot = s.sum(1)
print("waves %d, mean total cycles/wave %.0f (%.1f us @2.1GHz) min %.0f max %.0f" % (s.shape[0], tot.mean(), tot.mean() / 2100, tot.min(), tot.max()))
for i, n in enumerate(names):
    print("%-14s mean %9.0f cyc  %5.1f%%   full waves %9.0f  quarter waves %9.0f" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean(),
          s.reshape(-1, 16, 8)[:, :12, i].mean() if s.shape[0] % 16 == 0 else 0, s.reshape(-1, 16, 8)[:, 12:, i].mean() if s.shape[0] % 16 == 0 else 0))
