"""Developer script (run on the GPU box): per-tensor parity table HIP vs oracle. Not a pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel


def rows(a):  # [k,B,F] -> [B*k, F]
    a = np.asarray(a)
    if a.ndim == 2:
        return a.T.reshape(-1, 1)
    return a.transpose(1, 0, 2).reshape(-1, a.shape[-1])


def err(name, got, ref, scale=None):
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    if got.shape != ref.shape:
        print("%-14s SHAPE got %s ref %s" % (name, got.shape, ref.shape)); return
    d = np.abs(got - ref)
    s = np.max(np.abs(ref)) + 1e-30
    i = np.unravel_index(np.argmax(d), d.shape)
    print("%-14s max|d| %.3e  rel-to-max %.3e  (ref max %.3e) at %s got %.5g ref %.5g" % (name, d.max(), d.max() / s, s, i, got[i], ref[i]))


def run(B, k, obj, n_hidden=200, n_latent=100, beta=1.0, seed=0):
    print("=" * 100); print("1-layer B=%d k=%d obj=%s H=%d D=%d" % (B, k, obj, n_hidden, n_latent))
    rng = np.random.default_rng(seed)
    x = O.synthetic_binarized(B, seed + 1)
    P = O.init_params(1, n_hidden, n_latent, seed + 2, x_mean=O.synthetic_pixel_means())
    P = [(W, b + 0.05 * rng.standard_normal(b.shape)) for W, b in P]
    eps = rng.standard_normal((k, B, n_latent)).astype(np.float32)
    m = NativeModel(1, n_hidden, n_latent, seed=5)
    m.set_params(O.flatten_params(P))
    tape = {}
    res_o, g_o = O.loss_grads_1layer(P, x, eps, beta, obj, rnd=O.bf16_round, tape=tape)
    res_x, g_x = O.loss_grads_1layer(P, x, eps, beta, obj)
    want = ("z", "snis_z", "al", "logits", "lpxz", "lpz", "lqzx", "log_w")
    r = m.forward_backward(x, k, beta, obj, eps=eps, want=want)
    enc, dec = tape["enc"], tape["dec"]
    err("enc.h1", m.debug_tensor("enc.h1"), enc.l1.y)
    err("enc.h2", m.debug_tensor("enc.h2"), enc.l2.y)
    head = m.debug_tensor("enc.head")
    Dp = head.shape[1] // 2
    err("mu", head[:, :n_latent], tape["mu"])
    err("sigma", head[:, Dp:Dp + n_latent], tape["sigma"])
    err("z(bf16)", m.debug_tensor("z"), rows(O.bf16_round(tape["z"])))
    err("z export", r["z"], tape["z"])
    err("dec.g1", m.debug_tensor("dec.g1"), rows(dec.d1.y))
    err("dec.g2", m.debug_tensor("dec.g2"), rows(dec.d2.y))
    err("logits", r["logits"], tape["logits"])
    for nm in ("lpxz", "lpz", "lqzx"):
        err(nm, r[nm], res_o[nm]); err(nm + " vs exact", r[nm], res_x[nm])
    err("log_w", r["log_w"], tape["log_w"])
    err("al", r["al"], res_o["al"])
    err("snis_z", r["snis_z"], res_o["snis_z"])
    for nm in ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14"):
        print("%-14s got %.6f emu %.6f exact %.6f" % (nm, r[nm], res_o[nm], res_x[nm]))
    if obj == "dreg":
        print("inference_loss got %.6f emu %.6f exact %.6f" % (r["inference_loss"], res_o["inference_loss"], res_x["inference_loss"]))
    err("gx", m.debug_tensor("gx"), rows(tape["G"]))
    err("dec.dl", m.debug_tensor("dec.dl"), rows(dec.out.dpre))
    err("dec.d2", m.debug_tensor("dec.d2"), rows(dec.d2.dpre))
    err("dec.d1", m.debug_tensor("dec.d1"), rows(dec.d1.dpre))
    dz = m.debug_tensor("dec.dz")
    err("dec.dz", dz[:, :n_latent], rows(tape["dz_dec"]))
    dh = m.debug_tensor("enc.dhead")
    err("enc.dmu", dh[:, :n_latent], enc.lmu.dpre)
    err("enc.da", dh[:, Dp:Dp + n_latent], enc.lstd.dpre)
    err("enc.d2", m.debug_tensor("enc.d2"), enc.l2.dpre)
    err("enc.d1", m.debug_tensor("enc.d1"), enc.l1.dpre)
    g = m.get_grads()
    off = 0
    names = [n for n, _ in O.layer_shapes(1, n_hidden, n_latent)]
    for nm, (dW, db), (dWx, dbx) in zip(names, g_o, g_x):
        gw = g[off:off + dW.size].reshape(dW.shape); off += dW.size
        gb = g[off:off + db.size]; off += db.size
        err("dW " + nm, gw, dW); err("  vs exact", gw, dWx); err("db " + nm, gb, db)
    # one Adam step
    flat0 = O.flatten_params(P)
    m.adam_step(1e-3)
    p1 = m.get_params()
    f_ref, _, _ = O.adam_update(flat0, g.astype(np.float64), 0.0, 0.0, 1, 1e-3)
    err("adam(params)", p1, f_ref)
    m.close()


if __name__ == "__main__":
    run(4, 3, "iwae_elbo")
    run(8, 50, "iwae_elbo", seed=3)
    run(5, 7, "vae_elbo_kl", beta=0.7, seed=4)
    run(6, 5, "dreg", seed=5)
    run(3, 2, "vae_elbo", n_hidden=64, n_latent=2, seed=6)
