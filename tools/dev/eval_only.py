"""Developer script (GPU box): the k = 5000 evaluator alone (for rocprofv3 --kernel-trace --stats).  usage: python tools/dev/eval_only.py [fp32|bf16] [images] [option=value ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
x = O.synthetic_binarized(n, 1)
opts = dict((kv.split("=")[0], int(kv.split("=")[1])) for kv in sys.argv[3:] if "=" in kv)
for kv in sys.argv[3:]:
    if kv.endswith(".so"):      # another build of the library (A/B)
        import ctypes
        from iwae_amd import _capi
        _capi.LIB_PATH = os.path.abspath(kv)
        _probe = ctypes.CDLL(_capi.LIB_PATH)
        for _n in [n_ for n_ in _capi.SYMBOLS if not hasattr(_probe, n_)]:
            del _capi.SYMBOLS[_n]
m = NativeModel(1, 200, 100, seed=5, options=opts)
m.set_eval_precision(prec)
m.eval_llh(x[:min(n, 500)], 5000)      # (warm-up on full-size launches: the buffers grow here, not in the timed call)
m.sync()
t = time.perf_counter()
llh = m.eval_llh(x, 5000)
dt = time.perf_counter() - t
print("%s %s: %d images in %.3f s = %.0f images/s, llh %.4f" % (prec, opts, n, dt, n / dt, llh))
