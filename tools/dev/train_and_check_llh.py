"""Developer script (GPU box): train a few epochs on synthetic MNIST-like data with the device pipeline,
then compare the k=5000 test LLH of the trained (bf16-GEMM) model with the exact float64 oracle on a few images."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from iwae_amd import iwae1, utils
from iwae_amd.optimizers import Adam
from oracle import iwae_np as O, philox_np

np.random.seed(123)
Xtrain, Xtest = utils.synthetic_mnist(20000, 512)
# make the data less trivial: shift the blob per image
rng = np.random.default_rng(0)
def jitter(X):
    out = np.empty_like(X)
    for i in range(X.shape[0]):
        out[i] = np.roll(np.roll(X[i].reshape(28, 28), rng.integers(-4, 5), 0), rng.integers(-4, 5), 1).reshape(-1)
    return out
Xtrain, Xtest = jitter(Xtrain), jitter(Xtest)
model = iwae1.IWAE(200, 100, output_bias=utils.get_bias(Xtrain))
opt = Adam(1e-3, epsilon=1e-4)
model.set_dataset(Xtrain)
B, k = 100, 50
Xt = utils.bernoullisample(Xtest)
t0 = time.time()
for epoch in range(6):
    model.begin_epoch(epoch, np.random.permutation(Xtrain.shape[0]))
    for lo in range(0, Xtrain.shape[0], B):
        res = model.train_step_dataset(lo, B, k, 1.0, opt, objective="iwae_elbo")
    v = model.val_step(Xt, k, 1.0)
    print("epoch %d train iwae_elbo %.3f  val iwae_elbo %.3f  (%.1f s)" % (epoch, float(res["iwae_elbo"]), float(v["iwae_elbo"]), time.time() - t0), flush=True)
# k=5000 LLH: device vs exact oracle on 12 test images with the SAME noise (device Philox restated in NumPy)
net = model._net
n = 12
net.set_step(999, 0)
llh_dev, per = net.eval_llh(Xt[:n], 5000, chunk=n, per_image=True)
P = O.unflatten_params(net.get_params().astype(np.float64), 1, 200, 100)
per_o = []
for i in range(n):
    eps = philox_np.device_eps(123, 999, 1, 5000, 100, batch_offset=i)
    r = O.forward_1layer(P, Xt[i:i + 1], eps)
    per_o.append(float(r["iwae_elbo"]))
per_o = np.array(per_o)
print("k=5000 LLH device %.4f  exact-oracle %.4f  |mean diff| %.4f  max per-image |diff| %.4f" % (per.mean(), per_o.mean(), abs(per.mean() - per_o.mean()), np.max(np.abs(per - per_o))))
