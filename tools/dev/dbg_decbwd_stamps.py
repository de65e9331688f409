"""Developer script (GPU box, library built with STAMPS=1): phase cycle shares inside dec_bwd_kernel.
usage: python tools/dev/dbg_decbwd_stamps.py [option=value ...]      (e.g. dec_bwd_nw=4: the 4 waves x 32 rows shape)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
B, k = 1024, 50
x = O.synthetic_binarized(B, 1)
opts = {"dense_stamps_epi": 9, "dense_stamps_kt": 0}
opts.update(dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[1:]))
m = NativeModel(1, 200, 100, seed=5, options=opts)      # diagnostic option names: STAMPS=1 build only
for i in range(10):
    m.forward_backward(x, k, 1.0, "iwae_elbo")
s = m.debug_tensor("dense_stamps").astype(np.float64)
s = s[s.sum(1) > 0]
names = ["p1 wait", "p1 multiply", "dpre2", "p2/p3 wait", "p2/p3 mult+epi", "-", "store drain", "-"]
tot = s.sum(1)
print("waves %d, mean total cycles/wave %.0f (%.1f us @2.1GHz) min %.0f max %.0f" % (s.shape[0], tot.mean(), tot.mean() / 2100, tot.min(), tot.max()))
for i, n in enumerate(names):
    print("%-16s mean %9.0f cyc  %5.1f%%" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean()))
