"""Developer script (GPU box): random shapes through iwae_forward_backward against the rounding-aware oracle.
usage: python tools/dev/fuzz_parity.py [n_cases] [seed]      env FUZZ_LARGE=1: >= 8 192 rows; FUZZ_FUSED=1: the 200-row decoder kernel with its in-kernel log-mean-exp;
FUZZ_F32=1: float32 mode against the exact oracle at the float32 tolerances (scalars 1e-5, gradients 1e-4 relative)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden"))
import numpy as np
from oracle import iwae_np as O
import make_golden as MG
from iwae_amd.native import NativeModel

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
objs1 = ["vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl", "dreg"]
worst = 0.0
for c in range(n_cases):
    layers = 1 if rng.random() < 0.7 else 2
    B, k = int(rng.integers(1, 48)), int(rng.integers(1, 72))
    if os.environ.get("FUZZ_LARGE"):      # row counts that take the large-row kernel choices (>= 8192 / >= 16384 rows)
        B, k = int(rng.integers(130, 420)), int(rng.integers(45, 70))
    xd = int(rng.choice([48, 100, 784, 1000]))
    fused = bool(os.environ.get("FUZZ_FUSED"))
    if fused:      # the decoder kernel's 16-wave / 200-row shape with its in-kernel log-mean-exp: k a divisor of 200, hidden width 200, >= 8 192 rows
        k = int(rng.choice([20, 25, 40, 50, 100, 200]))
        B = int(rng.integers(8192 // k + 1, 12000 // k + 2))
        xd = int(rng.choice([784, 500, 300]))
    if layers == 1 and fused:
        nh, nl = 200, int(rng.choice([20, 50, 64, 100, 128]))
        obj = str(rng.choice(objs1)); beta = float(rng.choice([1.0, 0.5]))
    elif fused:
        nh = [200, 100]; nl = [100, 50]
        obj = str(rng.choice(["vae_elbo", "iwae_elbo", "iwae_eq14"])); beta = 1.0
    elif layers == 1:
        nh, nl = int(rng.choice([16, 40, 64, 100, 128, 200, 256])), int(rng.choice([2, 4, 10, 32, 50, 100, 128]))
        obj = str(rng.choice(objs1)); beta = float(rng.choice([1.0, 0.5]))
    else:
        nh = [int(rng.choice([32, 64, 200])), int(rng.choice([16, 100]))]; nl = [int(rng.choice([4, 20, 100])), int(rng.choice([2, 50]))]
        obj = str(rng.choice(["vae_elbo", "iwae_elbo", "iwae_eq14"])); beta = 1.0
    x, P, eps = MG.inputs(layers, nh, nl, xd, B, k, 1000 + c)
    f32 = bool(os.environ.get("FUZZ_F32"))
    m = NativeModel(layers, nh, nl, x_dim=xd, seed=1, precision="fp32" if f32 else "bf16", options={"bern_qw_force": 1} if fused else None)
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, beta, obj, eps=eps)
    rnd = None if f32 else O.bf16_round
    if layers == 1:
        res, g = O.loss_grads_1layer(P, x, eps, beta, obj, **({} if f32 else {"rnd": rnd}))
    else:
        res, g = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj, **({} if f32 else {"rnd": rnd}))
    key = "iwae_elbo" if obj == "dreg" else obj
    flat = m.get_grads(); off = 0; errs = []
    for dW, db in g:
        for t in (dW, db):
            got = flat[off:off + t.size].reshape(t.shape).astype(np.float64); off += t.size
            errs.append(np.linalg.norm(got - t) / (np.linalg.norm(t) + 1e-30))
    ds = abs(r[key] - res[key])
    worst = max(worst, max(errs))
    flag = "" if ((max(errs) < 1e-4 and ds < 1e-5 * abs(res[key]) + 2e-4) if f32 else (max(errs) < 2e-2 and ds < 0.05)) else "   <-- CHECK"
    print("case %2d L%d B=%3d k=%3d x=%4d h=%s z=%s %-11s |dscalar| %.4f  max grad rel %.2e%s" % (c, layers, B, k, xd, nh, nl, obj, ds, max(errs), flag), flush=True)
    m.close()
print("worst gradient relative error %.3e" % worst)
