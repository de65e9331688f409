"""Training-level evidence for the bf16 headline (missing #3): the SAME model trained from the same initial weights on the same
batches and the same device noise, once with bf16 GEMM operands and once in float32 mode (the reference's arithmetic, src/iwae1.py:31-34), then the
k = 5000 test log-likelihood (main.py:170-184) of both trained models on held-out images with the float32 evaluator.  Also float32 runs with other
noise seeds: the run-to-run spread the bf16 / float32 difference has to be read against.
Data: sklearn.datasets.load_digits (1 797 real 8 x 8 digits, bundled offline) upsampled to 28 x 28, or the synthetic blobs.
usage: python tools/dev/train_bf16_vs_fp32.py [digits|synthetic] [steps] [B] [k]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from iwae_amd import iwae1, utils  # noqa: E402
from iwae_amd.optimizers import Adam  # noqa: E402


def digits_28():
    from sklearn.datasets import load_digits
    d = load_digits().images.astype(np.float64) / 16.0          # [1797, 8, 8] in [0, 1]
    up = np.kron(d, np.ones((1, 3, 3)))                          # 24 x 24
    out = np.zeros((d.shape[0], 28, 28))
    out[:, 2:26, 2:26] = up
    return out.reshape(d.shape[0], 784)


def run(X, Xtest_bin, precision, seed, steps, B, k):
    model = iwae1.IWAE(200, 100, output_bias=utils.get_bias(X), precision=precision, seed=seed)
    P0 = np.load("/tmp/_p0.npy") if os.path.exists("/tmp/_p0.npy") else None
    if P0 is None:
        P0 = model._net.get_params().copy()
        np.save("/tmp/_p0.npy", P0)
    model._net.set_params(P0)                                    # every run starts from the same weights
    opt = Adam(1e-3, epsilon=1e-4)
    model.set_dataset(X)
    rs = np.random.RandomState(5)
    t0 = time.time()
    step, epoch = 0, 0
    n = (X.shape[0] // B) * B
    while step < steps:
        model.begin_epoch(epoch, rs.permutation(X.shape[0]))
        for lo in range(0, n, B):
            res = model.train_step_dataset(lo, B, k, 1.0, opt, objective="iwae_elbo")
            step += 1
            if step >= steps:
                break
        epoch += 1
    last = float(res["iwae_elbo"])
    net = model._net
    net.set_eval_precision("fp32")
    net.set_step(999, 0)
    llh, per = net.eval_llh(Xtest_bin, 5000, chunk=64, per_image=True)
    dt = time.time() - t0
    net.close()
    return llh, per, last, dt


def main():
    data = sys.argv[1] if len(sys.argv) > 1 else "digits"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    if os.path.exists("/tmp/_p0.npy"):
        os.remove("/tmp/_p0.npy")
    if data == "digits":
        X = digits_28()
        rs = np.random.RandomState(0)
        perm = rs.permutation(X.shape[0])
        Xtr, Xte = X[perm[:1500]], X[perm[1500:]]
    else:
        Xtr, Xte = utils.synthetic_mnist(20000, 256)
    np.random.seed(3)
    Xte_bin = utils.bernoullisample(Xte)
    rows = []
    for prec, seed in (("bf16", 123), ("fp32", 123), ("fp32", 124), ("fp32", 125), ("bf16", 124)):
        llh, per, last, dt = run(Xtr, Xte_bin, prec, seed, steps, B, k)
        rows.append((prec, seed, llh, per))
        print("%s seed %d: k=5000 LLH on %d held-out images %.4f (last train iwae_elbo %.2f, %.1f s)" % (prec, seed, Xte_bin.shape[0], llh, last, dt), flush=True)
    base = rows[1]
    for prec, seed, llh, per in rows:
        print("  %s/%d - fp32/123: mean %.4f, per-image max |d| %.3f, std %.3f" % (prec, seed, llh - base[2], np.max(np.abs(per - base[3])), np.std(per - base[3])))


if __name__ == "__main__":
    main()
