"""Developer script (GPU box, library built with STAMPS=1): phase cycle shares inside one dense_kernel launch.
usage: IWAE_DENSE_STAMPS=<epi>:<KT> python tools/dev/dbg_dense_stamps.py      (epi 4 = BERN, 0 = TANH, 2 = DX, 3 = F32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
B, k = 1024, 50
x = O.synthetic_binarized(B, 1)
m = NativeModel(1, 200, 100, seed=5)
for i in range(10):
    m.forward_backward(x, k, 1.0, "iwae_elbo")
s = m.debug_tensor("dense_stamps")
names = ["prologue", "vmcnt wait", "barrier", "issue st/ld", "mfma", "epilogue", "tail", "-"]
tot = s.sum(1)
print("launch %s: waves %d, mean total cycles/wave %.0f (%.1f us @2.4GHz) min %.0f max %.0f" % (os.environ.get("IWAE_DENSE_STAMPS"), s.shape[0], tot.mean(), tot.mean() / 2400, tot.min(), tot.max()))
for i, n in enumerate(names[:7]):
    print("%-12s mean %9.0f cyc  %5.1f%%" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean()))
