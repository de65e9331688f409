"""Developer script (GPU box, library built with STAMPS=1): phase cycle shares inside gemm_f32_v2_kernel in the float32 training step --
the output layer's dX product (epi 12) or its weight gradient (epi 13).   usage: python tools/dev/dbg_gemmf32_stamps.py 12|13"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
epi = int(sys.argv[1]) if len(sys.argv) > 1 else 12
x = O.synthetic_binarized(1024, 1)
m = NativeModel(1, 200, 50, seed=5, precision="fp32", options={"dense_stamps_epi": epi, "no_f32_side": 1})
for i in range(4):
    m.train_step(x, 50, 1.0, 1e-3, "iwae_elbo")
s = m.debug_tensor("dense_stamps").astype(np.float64)
s = s[s.sum(1) > 0]
names = ["set-up + first k-step", "requests of the next k-step", "fragment reads + MFMAs", "wait for the quads", "LDS stores", "barrier", "epilogue", "-"]
tot = s.sum(1)
print("epi %d: waves %d, mean total cycles/wave %.0f min %.0f max %.0f" % (epi, s.shape[0], tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    print("%-30s mean %9.0f cyc  %5.1f%%" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean()))
