"""Developer script (GPU box, library built with STAMPS=1): phase cycle shares inside wgradws_kernel<true,4,4> (the output layer's
weight gradient): loader waves (issue | wait | weight + LDS store | barrier) and compute waves (G reads | A reads + MFMA | barrier).
usage: python tools/dev/dbg_wgrad_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd import _capi
if os.environ.get("IWAE_LIB"):
    _capi.LIB_PATH = os.environ["IWAE_LIB"]      # the STAMPS=1 build, kept beside the shipped library
from iwae_amd.native import NativeModel
B, k = 1024, 50
x = O.synthetic_binarized(B, 1)
m = NativeModel(1, 200, 100, seed=5, options={"dense_stamps_epi": 10, "dense_stamps_kt": 0})      # diagnostic option names: STAMPS=1 build only
for i in range(10):
    m.forward_backward(x, k, 1.0, "iwae_elbo")
m.sync()
s = m.debug_tensor("dense_stamps").astype(np.float64).reshape(-1, 12, 8)
comp, load = s[:, :8, :], s[:, 8:, :]
for nm, arr, names in (("compute waves", comp, ["prologue", "G reads", "A reads + MFMA", "barrier", "slab stores", "-", "-", "-"]),
                       ("loader waves", load, ["prologue", "requests", "wait", "weight + store", "barrier", "-", "-", "-"])):
    tot = arr.sum(2)
    print("%s: %d waves, mean total %.0f cycles (%.1f us @2.1 GHz), min %.0f max %.0f" % (nm, tot.size, tot.mean(), tot.mean() / 2100, tot.min(), tot.max()))
    for i, n in enumerate(names):
        if n != "-":
            print("    %-16s mean %9.0f cyc  %5.1f%%   per stage (50) %7.0f" % (n, arr[:, :, i].mean(), 100 * arr[:, :, i].mean() / tot.mean(), arr[:, :, i].mean() / 50))
# the last column block (64 real columns of 256) separately: blocks are numbered bz * gx + bx with gx = 4
last = s.reshape(-1, 4, 12, 8)[:, 3]
print("last column block: compute total %.0f, loader total %.0f" % (last[:, :8].sum(2).mean(), last[:, 8:].sum(2).mean()))
