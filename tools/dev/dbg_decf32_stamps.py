"""Developer script (GPU box, library built with STAMPS=1): phase cycle shares inside dec_fwd_f32_kernel (the float32 decoder forward) in the k = 5000 evaluator.
usage: python tools/dev/dbg_decf32_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
x = O.synthetic_binarized(100, 1)
m = NativeModel(1, 200, 100, seed=5, options={"dense_stamps_epi": 11})      # diagnostic option name: STAMPS=1 build only
m.set_eval_precision("fp32")
for i in range(3):
    m.eval_llh(x, 5000)
s = m.debug_tensor("dense_stamps").astype(np.float64)
s = s[s.sum(1) > 0]
names = ["z staging", "pass set-up", "barriers", "epilogue operand requests", "wait for own DMA", "DMA issue", "fragment reads + MFMAs", "epilogues"]
tot = s.sum(1)
print("waves %d, mean total cycles/wave %.0f min %.0f max %.0f   (matrix pipe: 3 744 MFMAs x 32 = 119 808 cycles per wave, two waves per SIMD)" % (s.shape[0], tot.mean(), tot.min(), tot.max()))
for i, n in enumerate(names):
    print("%-28s mean %9.0f cyc  %5.1f%%" % (n, s[:, i].mean(), 100 * s[:, i].mean() / tot.mean()))
