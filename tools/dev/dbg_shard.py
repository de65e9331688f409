"""Developer script (GPU box): how far two half-batch gradients (global noise keys) are from the full-batch gradient, and how far the decoder kernel's
two workgroup shapes are from each other on the full batch -- for any library file.  usage: python tools/dev/dbg_shard.py [path/to/lib.so]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from iwae_amd import _capi
if len(sys.argv) > 1:
    _capi.LIB_PATH = os.path.abspath(sys.argv[1])
    import ctypes
    lib = ctypes.CDLL(_capi.LIB_PATH)
    for n in list(_capi.SYMBOLS):
        if not hasattr(lib, n):
            del _capi.SYMBOLS[n]
from iwae_amd.native import NativeModel
from oracle import iwae_np as O

B, k = 1024, 50
x = O.synthetic_binarized(B, 77)
P = O.init_params(1, 200, 100, 31, x_mean=O.synthetic_pixel_means())


def grads(opts, parts):
    m = NativeModel(1, 200, 100, seed=123, options=opts)
    m.set_params(O.flatten_params(P))
    out = []
    n = B // parts
    for h in range(parts):
        m.set_step(11, n * h)
        r = m.forward_backward(x[n * h:n * (h + 1)], k, 1.0, "iwae_elbo", want=("lpz", "lqzx", "lpxz"))
        out.append((m.get_grads().astype(np.float64), r))
    m.close()
    return out


rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
(gf, rf), = grads(None, 1)
(g8, r8), = grads({"no_bern_qw": 1}, 1)
hv = grads(None, 2)
print("full batch, 16-wave vs 8-wave shape: grad rel %.3e  max|d lpz| %.2e  max|d lqzx| %.2e  max|d lpxz| %.2e" %
      (rel(g8, gf), np.abs(r8["lpz"] - rf["lpz"]).max(), np.abs(r8["lqzx"] - rf["lqzx"]).max(), np.abs(r8["lpxz"] - rf["lpxz"]).max()))
print("two halves vs full: grad rel %.3e" % rel(0.5 * (hv[0][0] + hv[1][0]), gf))
h8 = grads({"no_bern_qw": 1}, 2)
print("two halves vs full, both on the 8-wave shape: grad rel %.3e" % rel(0.5 * (h8[0][0] + h8[1][0]), g8))
