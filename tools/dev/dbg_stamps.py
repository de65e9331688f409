"""Developer script (GPU box): phase cycle shares inside out_bwd (diagnostic STAMPS build)."""
import os, sys
os.environ["IWAE_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
B, k = 1024, 50
x = O.synthetic_binarized(B, 1)
m = NativeModel(1, 200, 100, seed=5)
for i in range(20):
    m.forward_backward(x, k, 1.0, "iwae_elbo")
s = m.debug_tensor("stamps")
names = ["prologue", "vmcnt wait", "barrier", "issue dma/st/ld", "mfma1", "epilogue", "mfma2", "final"]
tot = s.sum(1)
print("waves", s.shape[0], "mean total cycles/wave %.0f (%.1f us @2.4GHz) min %.0f max %.0f" % (tot.mean(), tot.mean() / 2400, tot.min(), tot.max()))
for i, n in enumerate(names):
    print("%-16s mean %9.0f cyc  %5.1f%%   (per group %.0f)" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean(), s[:, i].mean() / 13))
