"""Developer script (GPU box): 2-layer parity table HIP vs oracle. Not a pytest."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import iwae_np as O
from iwae_amd.native import NativeModel
from dbg_parity import rows, err


def run(B, k, obj, nh=(200, 100), nl=(100, 50), seed=0):
    print("=" * 100); print("2-layer B=%d k=%d obj=%s" % (B, k, obj))
    rng = np.random.default_rng(seed)
    x = O.synthetic_binarized(B, seed + 1)
    P = O.init_params(2, list(nh), list(nl), seed + 2, x_mean=O.synthetic_pixel_means())
    P = [(W, b + 0.05 * rng.standard_normal(b.shape)) for W, b in P]
    e1 = rng.standard_normal((k, B, nl[0])).astype(np.float32)
    e2 = rng.standard_normal((k, B, nl[1])).astype(np.float32)
    m = NativeModel(2, list(nh), list(nl), seed=5)
    m.set_params(O.flatten_params(P))
    t = {}
    res_o, g_o = O.loss_grads_2layer(P, x, e1, e2, 1.0, obj, rnd=O.bf16_round, tape=t)
    res_x, g_x = O.loss_grads_2layer(P, x, e1, e2, 1.0, obj)
    want = ("z", "z2", "snis_z", "snis_z2", "al", "lpxz", "lpz", "lqzx", "lpz2", "lqzx2", "log_w")
    r = m.forward_backward(x, k, 1.0, obj, eps=(e1, e2), want=want)
    err("z1 export", r["z"], t["z1"]); err("z2 export", r["z2"], t["z2"])
    h2 = m.debug_tensor("enc2.head"); D2p = h2.shape[1] // 2
    err("mu2", h2[:, :nl[1]], rows(t["mu2"])); err("sig2", h2[:, D2p:D2p + nl[1]], rows(t["sig2"]))
    hp = m.debug_tensor("dec2.head"); D1p = hp.shape[1] // 2
    err("mup", hp[:, :nl[0]], rows(t["mup"])); err("sigp", hp[:, D1p:D1p + nl[0]], rows(t["sigp"]))
    for a, b in (("lpxz", "lpxz1"), ("lpz", "lpz1z2"), ("lpz2", "lpz2"), ("lqzx", "lqz1x"), ("lqzx2", "lqz2z1")):
        err(b, r[a], res_o[b]); err(b + " vs exact", r[a], res_x[b])
    err("al", r["al"], res_o["al"]); err("snis_z1", r["snis_z"], res_o["snis_z1"]); err("snis_z2", r["snis_z2"], res_o["snis_z2"])
    for nm in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
        print("%-14s got %.6f emu %.6f exact %.6f" % (nm, r[nm], res_o[nm], res_x[nm]))
    dh = m.debug_tensor("dec2.dhead")
    err("dec2.dmup", dh[:, :nl[0]], rows(t["dec2"].lmu.dpre)); err("dec2.dap", dh[:, D1p:D1p + nl[0]], rows(t["dec2"].lstd.dpre))
    dh = m.debug_tensor("enc2.dhead")
    err("enc2.dmu2", dh[:, :nl[1]], rows(t["enc2"].lmu.dpre)); err("enc2.da2", dh[:, D2p:D2p + nl[1]], rows(t["enc2"].lstd.dpre))
    dh = m.debug_tensor("enc.dhead"); 
    err("enc1.dmu", dh[:, :nl[0]], t["enc1"].lmu.dpre); err("enc1.da", dh[:, D1p:D1p + nl[0]], t["enc1"].lstd.dpre)
    g = m.get_grads(); off = 0
    names = [n for n, _ in O.layer_shapes(2, list(nh), list(nl))]
    for nmL, (dW, db), (dWx, dbx) in zip(names, g_o, g_x):
        gw = g[off:off + dW.size].reshape(dW.shape); off += dW.size
        gb = g[off:off + db.size]; off += db.size
        err("dW " + nmL, gw, dW); err("  vs exact", gw, dWx); err("db " + nmL, gb, db)
    m.close()


if __name__ == "__main__":
    run(4, 3, "iwae_elbo")
    run(6, 50, "vae_elbo", seed=2)
    run(3, 5, "iwae_eq14", nh=(64, 32), nl=(4, 2), seed=3)
