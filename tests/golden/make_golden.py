"""Generates tests/golden/*.npz from the CPU oracle (oracle/iwae_np.py, float64).

The reference ships no fixtures and cannot be executed here (TensorFlow absent), so these vectors
pin the ORACLE (against regressions) and give the GPU tests fixed targets; they are not outputs of
the reference.  Every file holds inputs AND expected outputs as plain arrays.  For the full-size
model (455k / 521k parameters) the parameters and noise are regenerated from the stored seeds with
numpy's PCG64 (a stable stream) and only compact expectations are stored.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import iwae_np as O  # noqa: E402


def inputs(n_layers, nh, nl, x_dim, B, k, seed, cond=0, cond_prior=False):
    """Deterministic inputs shared by the generator and the tests.  cond > 0: the conditional models of tasks/task05.py /
    task04.py (cond_prior); the one-hot condition y is returned as a 4th value."""
    rng = np.random.default_rng(seed)
    if x_dim == 784:
        x = O.synthetic_binarized(B, seed + 1)
        mean = O.synthetic_pixel_means()
    else:
        x = (rng.random((B, x_dim)) < 0.3).astype(np.float32)
        mean = np.full(x_dim, 0.3)
    P = O.init_params(n_layers, nh, nl, seed + 2, x_mean=mean, x_dim=x_dim, cond_dim=cond, cond_prior=cond_prior)
    P = [(W, b + 0.05 * rng.standard_normal(b.shape)) for W, b in P]
    if n_layers == 1:
        eps = rng.standard_normal((k, B, nl)).astype(np.float32)
    else:
        eps = (rng.standard_normal((k, B, nl[0])).astype(np.float32), rng.standard_normal((k, B, nl[1])).astype(np.float32))
    if cond:
        y = np.eye(cond, dtype=np.float32)[rng.integers(0, cond, B)]
        return x, P, eps, y
    return x, P, eps


def grad_summary(grads):
    """[n_tensors, 3]: sum, L2 norm, max|.| of every dW and db (Keras order)."""
    rows = []
    for dW, db in grads:
        for g in (dW, db):
            rows.append([g.sum(), np.sqrt((g * g).sum()), np.abs(g).max()])
    return np.array(rows)


def probe_indices(grads, per_tensor=256, seed=20240229):
    """Fixed (seeded) positions in the FLAT gradient, up to `per_tensor` per dW / db: the full-size fixtures keep the gradient's
    values there, so the GPU test compares element by element (a sign flip or a permuted tensor cannot hide behind a norm)
    without storing 455k floats per objective."""
    rng = np.random.default_rng(seed)
    idx, off = [], 0
    for dW, db in grads:
        for g in (dW, db):
            n = g.size
            take = np.arange(n) if n <= per_tensor else np.sort(rng.choice(n, per_tensor, replace=False))
            idx.append(off + take)
            off += n
    return np.concatenate(idx).astype(np.int64)


def run(n_layers, nh, nl, x_dim, B, k, seed, objective, beta, rnd, cond=0, cond_prior=False):
    y = None
    if cond:
        x, P, eps, y = inputs(n_layers, nh, nl, x_dim, B, k, seed, cond, cond_prior)
    else:
        x, P, eps = inputs(n_layers, nh, nl, x_dim, B, k, seed)
    if n_layers == 1:
        res, g = O.loss_grads_1layer(P, x, eps, beta, objective, rnd=rnd, y=y)
    else:
        res, g = O.loss_grads_2layer(P, x, eps[0], eps[1], beta, objective, rnd=rnd)
    flat = O.flatten_params(P)
    gflat = O.flatten_grads(g)
    p1, m1, v1 = O.adam_update(flat, gflat, 0.0, 0.0, 1, 1e-3)
    return x, P, eps, res, g, gflat, p1


def save(name, n_layers, nh, nl, x_dim, B, k, seed, objectives, beta=1.0, full_arrays=False, cond=0, cond_prior=False):
    out = {"n_layers": n_layers, "n_hidden": np.array(nh), "n_latent": np.array(nl), "x_dim": x_dim, "B": B, "k": k,
           "seed": seed, "beta": beta, "objectives": np.array(objectives)}
    if cond:
        out["cond_dim"], out["cond_prior"] = cond, int(cond_prior)
    for tag, rnd in (("exact", None), ("bf16", O.bf16_round)):
        for obj in objectives:
            x, P, eps, res, g, gflat, p1 = run(n_layers, nh, nl, x_dim, B, k, seed, obj, beta, rnd, cond, cond_prior)
            pre = "%s/%s/" % (tag, obj)
            for key, v in res.items():
                if np.ndim(v) == 0 or key in ("lpxz", "lpz", "lqzx", "lpxz1", "lpz1z2", "lpz2", "lqz1x", "lqz2z1", "al") or full_arrays:
                    out[pre + key] = np.asarray(v)
            out[pre + "grad_summary"] = grad_summary(g)
            out["grad_probe_idx"] = probe_indices(g)
            out[pre + "grad_probe"] = gflat[out["grad_probe_idx"]]
            out[pre + "adam_param_sum"] = np.array([p1.sum(), np.abs(p1).sum()])
            if full_arrays:
                out[pre + "grad_flat"] = gflat
                out[pre + "adam_params"] = p1
    if full_arrays:
        out["x"] = x
        out["params_flat"] = O.flatten_params(P)
        if n_layers == 1:
            out["eps"] = eps
        else:
            out["eps1"], out["eps2"] = eps
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "%.1f KB" % (os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))


if __name__ == "__main__":
    save("tiny_1layer", 1, 16, 4, 48, 5, 3, 11, ["vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl", "dreg"], beta=0.7, full_arrays=True)
    save("tiny_2layer", 2, [16, 8], [4, 2], 48, 4, 3, 12, ["vae_elbo", "iwae_elbo", "iwae_eq14"], full_arrays=True)
    save("full_1layer_B8_k50", 1, 200, 100, 784, 8, 50, 13, ["iwae_elbo", "vae_elbo_kl", "dreg"])
    save("full_1layer_B20_k1", 1, 200, 100, 784, 20, 1, 14, ["vae_elbo", "iwae_elbo"])
    save("full_2layer_B4_k5", 2, [200, 100], [100, 50], 784, 4, 5, 15, ["iwae_elbo", "vae_elbo"])
    save("full_cond_B6_k5", 1, 200, 100, 784, 6, 5, 16, ["iwae_elbo", "vae_elbo_kl"], cond=10)                         # tasks/task05.py
    save("full_condprior_B6_k5", 1, 200, 100, 784, 6, 5, 17, ["iwae_elbo", "vae_elbo"], beta=0.8, cond=10, cond_prior=True)  # tasks/task04.py
