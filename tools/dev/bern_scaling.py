"""Developer script (GPU box): Bernoulli-forward / out_bwd launch time against the row count (k = 50), to tell a
throughput bound (time ~ rows) from a per-workgroup latency bound (time ~ rounds).  Not a pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
import iwae_amd._capi as _c
if os.environ.get('IWAE_LIB'): _c.LIB_PATH = os.environ['IWAE_LIB']      # diagnostic builds
from iwae_amd.native import NativeModel
k = 50
m = NativeModel(1, 200, 100, seed=5)
m.set_output_bias(O.output_bias_from_mean(O.synthetic_pixel_means()))
for B in [int(a) for a in sys.argv[1:]] or [328, 656, 1024, 1312, 1968, 2624]:
    x = O.synthetic_binarized(B, 1)
    for i in range(3):
        m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
    m.sync()
    m.enable_timing(1)
    for i in range(20):
        m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
    m.sync()
    tb, nb = m.kernel_time("bernoulli_fwd"); to, no = m.kernel_time("out_bwd")
    m.enable_timing(0)
    rows = B * k
    print("B=%5d rows=%6d WGs=%4d (%.2f per CU)  bernoulli_fwd %.1f us (%d)  out_bwd %.1f us   ns/row %.2f" % (B, rows, (rows + 127) // 128, (rows + 127) // 128 / 256, tb, nb, to, tb * 1e3 / rows))
