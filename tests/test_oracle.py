"""CPU tests of the oracle: closed-form backward vs autograd, golden-fixture regression, the noise
generator's published known-answer vectors, Keras-form Adam, bf16 rounding."""
import os

import numpy as np
import pytest
import torch

from oracle import iwae_np as O, iwae_torch as T, philox_np
import make_golden as MG

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _max_rel(ga, gb):
    return max(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-300) for (aw, ab), (bw, bb) in zip(ga, gb) for a, b in ((aw, bw), (ab, bb)))


@pytest.mark.parametrize("objective", ["vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl", "dreg"])
def test_closed_form_backward_matches_autograd_1layer(objective):
    x, P, eps = MG.inputs(1, 16, 4, 48, 5, 7, 3)
    r1, g1 = O.loss_grads_1layer(P, x, eps, 0.7, objective)
    r2, g2 = T.loss_grads(P, x, eps, 0.7, objective, 1)
    assert _max_rel(g1, g2) < 1e-11
    for k in r1:
        if k in r2:
            np.testing.assert_allclose(r1[k], r2[k], rtol=1e-11, atol=1e-11)


@pytest.mark.parametrize("objective", ["vae_elbo", "iwae_elbo", "iwae_eq14"])
def test_closed_form_backward_matches_autograd_2layer(objective):
    x, P, eps = MG.inputs(2, [16, 8], [4, 2], 48, 4, 6, 4)
    r1, g1 = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, objective)
    r2, g2 = T.loss_grads(P, x, eps, 1.0, objective, 2)
    assert _max_rel(g1, g2) < 1e-11
    for k in r1:
        if k in r2:
            np.testing.assert_allclose(r1[k], r2[k], rtol=1e-11, atol=1e-11)


def test_2layer_has_no_vae_elbo_kl():          # src/iwae2.py:154-173: KeyError in the reference
    x, P, eps = MG.inputs(2, [16, 8], [4, 2], 48, 2, 2, 5)
    with pytest.raises(KeyError):
        O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, "vae_elbo_kl")


def test_parameter_counts_match_survey():       # 455,384 and 521,084
    assert sum(a * b + b for _, (a, b) in O.layer_shapes(1, 200, 100)) == 455384
    assert sum(a * b + b for _, (a, b) in O.layer_shapes(2, [200, 100], [100, 50])) == 521084


GOLDEN = ["tiny_1layer", "tiny_2layer", "full_1layer_B8_k50", "full_1layer_B20_k1", "full_2layer_B4_k5",
          "full_cond_B6_k5", "full_condprior_B6_k5"]


@pytest.mark.parametrize("name", GOLDEN)
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nl = int(g["n_layers"])
    nh = g["n_hidden"].tolist() if nl == 2 else int(g["n_hidden"])
    nlat = g["n_latent"].tolist() if nl == 2 else int(g["n_latent"])
    cond = int(g["cond_dim"]) if "cond_dim" in g else 0
    cprior = bool(int(g["cond_prior"])) if "cond_prior" in g else False
    for tag, rnd in (("exact", None), ("bf16", O.bf16_round)):
        for obj in [str(o) for o in g["objectives"]]:
            x, P, eps, res, gr, gflat, p1 = MG.run(nl, nh, nlat, int(g["x_dim"]), int(g["B"]), int(g["k"]), int(g["seed"]), obj, float(g["beta"]), rnd,
                                                   cond, cprior)
            pre = "%s/%s/" % (tag, obj)
            for key in res:
                if pre + key in g:
                    np.testing.assert_allclose(res[key], g[pre + key], rtol=1e-10, atol=1e-10)
            np.testing.assert_allclose(MG.grad_summary(gr), g[pre + "grad_summary"], rtol=1e-9, atol=1e-12)
            np.testing.assert_array_equal(MG.probe_indices(gr), g["grad_probe_idx"])
            np.testing.assert_allclose(gflat[g["grad_probe_idx"]], g[pre + "grad_probe"], rtol=1e-9, atol=1e-13)
            if "x" in g:      # fixtures that carry their inputs: the regenerated inputs must be those
                np.testing.assert_array_equal(x, g["x"])
                np.testing.assert_allclose(O.flatten_params(P), g["params_flat"], rtol=0, atol=0)


def test_logmeanexp_is_stable_and_exact():       # src/utils.py:6-8
    lw = np.array([[-1000.0, -1.0], [-1001.0, -2.0], [-999.0, -3.0]])
    out = O.logmeanexp(lw, axis=0)
    ref0 = -999.0 + np.log((np.exp(-1.0) + np.exp(-2.0) + 1.0) / 3.0)
    ref1 = -1.0 + np.log((1.0 + np.exp(-1.0) + np.exp(-2.0)) / 3.0)
    np.testing.assert_allclose(out, [ref0, ref1], rtol=1e-13)
    assert np.isclose(O.logmeanexp(np.full((7, 2), -3.25), 0), -3.25).all()      # k identical weights


def test_keras_adam_first_steps():
    # Keras form (epsilon outside the bias correction): after step 1, m_hat/sqrt(v_hat) = sign(g), so
    # theta moves by lr * |g| / (|g| + eps*sqrt(1-b2)/(... )) -- checked against a hand evaluation
    g = np.array([0.5, -2.0, 1e-5])
    th, m, v = O.adam_update(np.zeros(3), g, 0.0, 0.0, 1, 1e-3)
    alpha = 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9)
    ref = -alpha * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-4)
    np.testing.assert_allclose(th, ref, rtol=1e-14)
    th2, m2, v2 = O.adam_update(th, g, m, v, 2, 1e-3)
    assert np.all(np.abs(th2) > np.abs(th))


def test_learning_rate_schedule():                # main.py:44-51
    d, epochs = O.learning_rate_schedule()
    assert epochs == 3280
    assert sorted(d) == [0, 1, 4, 13, 40, 121, 364, 1093]
    np.testing.assert_allclose(d[0], 1e-3)
    np.testing.assert_allclose(d[1093], 1e-4, rtol=1e-12)


def test_bf16_round_matches_torch():
    a = np.random.default_rng(0).standard_normal(10000).astype(np.float32) * 37.0
    ref = torch.tensor(a).to(torch.bfloat16).to(torch.float32).numpy()
    np.testing.assert_array_equal(O.bf16_round(a).astype(np.float32), ref)


def test_philox_known_answer_vectors():
    """Random123 kat_vectors for philox4x32-10 (the published algorithm the device generator uses)."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for c, k, exp in kat:
        r = philox_np.philox4x32_10(*[np.uint32(v) for v in c], *k)
        assert tuple(int(v) for v in r) == exp


def test_device_eps_is_standard_normal_and_split_invariant():
    e = philox_np.device_eps(123, 5, 64, 50, 100)
    assert e.shape == (50, 64, 100)
    assert abs(e.mean()) < 0.01 and abs(e.std() - 1.0) < 0.01
    # rows are keyed by the global image index: a shard with batch_offset reproduces the slice
    e2 = philox_np.device_eps(123, 5, 16, 50, 100, batch_offset=32)
    np.testing.assert_array_equal(e[:, 32:48], e2)


def test_device_binarize_restatement_is_bernoulli_of_grey_level():
    """src/utils.py:26-27 semantics: P(x=1) = g/255; one draw per (epoch, image); integer thresholds."""
    g = np.tile(np.arange(256, dtype=np.uint8), (400, 1))           # 400 images x 256 "pixels" = all grey levels
    x = philox_np.device_binarize(123, 3, g, np.arange(400))
    assert set(np.unique(x)) <= {0.0, 1.0}
    assert x[:, 0].sum() == 0 and x[:, 255].sum() == 400              # g=0 never fires, g=255 always
    np.testing.assert_allclose(x.mean(0), np.arange(256) / 255.0, atol=0.09)
    assert abs(x.mean() - 0.5) < 0.01
    np.testing.assert_array_equal(x, philox_np.device_binarize(123, 3, g, np.arange(400)))      # deterministic
    assert (x != philox_np.device_binarize(123, 4, g, np.arange(400))).mean() > 0.2                 # new epoch, new draw
    np.testing.assert_array_equal(x[[7, 2]], philox_np.device_binarize(123, 3, g, [7, 2]))       # keyed by image id


def test_conditional_model_closed_form_matches_autograd():
    """tasks/task05.py:108-122: encoder on concat(x, y), decoder on concat(z, y).  The closed-form backward of the NumPy
    oracle against torch autograd of the op-for-op forward."""
    from oracle import iwae_torch as T
    rng = np.random.default_rng(4)
    B, k, X, H, D, C = 5, 4, 48, 16, 6, 10
    P = O.init_params(1, H, D, 3, x_dim=X, cond_dim=C)
    assert P[0][0].shape == (X + C, H) and P[4][0].shape == (D + C, H)
    x = (rng.random((B, X)) < 0.3).astype(np.float64)
    y = np.eye(C)[rng.integers(0, C, B)]
    eps = rng.standard_normal((k, B, D))
    for obj in ("iwae_elbo", "vae_elbo", "vae_elbo_kl", "iwae_eq14"):
        res, g = O.loss_grads_1layer(P, x, eps, 0.8, obj, y=y)
        rt, gt = T.loss_grads(P, x, eps, 0.8, obj, y=y)
        assert abs(res[obj] - rt[obj]) < 1e-12
        for (a, b), (c, d) in zip(g, gt):
            assert np.max(np.abs(a - c)) < 1e-12 and np.max(np.abs(b - d)) < 1e-12
    # without y the shapes are the unconditional ones and the condition is ignored everywhere
    P0 = O.init_params(1, H, D, 3, x_dim=X)
    assert P0[0][0].shape == (X, H) and P0[4][0].shape == (D, H)


def test_conditional_prior_closed_form_matches_autograd():
    """tasks/task04.py:124-136: lpz against the learned conditional prior p(z|y); its network's gradient, and the zero
    gradient under vae_elbo_kl (the analytic KL there stays against N(0,1))."""
    from oracle import iwae_torch as T
    rng = np.random.default_rng(6)
    B, k, X, H, D, C = 4, 5, 48, 16, 6, 10
    P = O.init_params(1, H, D, 3, x_dim=X, cond_dim=C, cond_prior=True)
    assert len(P) == 11 and P[7][0].shape == (C, H) and P[10][0].shape == (H, D)
    x = (rng.random((B, X)) < 0.3).astype(np.float64)
    y = np.eye(C)[rng.integers(0, C, B)]
    eps = rng.standard_normal((k, B, D))
    for obj in ("iwae_elbo", "vae_elbo", "vae_elbo_kl", "iwae_eq14"):
        res, g = O.loss_grads_1layer(P, x, eps, 0.7, obj, y=y)
        rt, gt = T.loss_grads(P, x, eps, 0.7, obj, y=y)
        assert abs(res[obj] - rt[obj]) < 1e-12
        for (a, b), (c, d) in zip(g, gt):
            assert np.max(np.abs(a - c)) < 1e-12 and np.max(np.abs(b - d)) < 1e-12
        assert (np.abs(g[7][0]).max() == 0.0) == (obj == "vae_elbo_kl")


# ---------------------------------------------------------------- third-party pin of the TFP semantics (missing #4)
def test_log_densities_match_torch_distributions():
    """The oracle's log-density formulas (oracle/iwae_np.py, re-typed in oracle/iwae_torch.py) restate the published semantics of
    tfd.Normal.log_prob, tfd.Bernoulli(logits).log_prob and tfd.kl_divergence (reference call sites: src/iwae1.py:105-116).  TensorFlow
    Probability is not installed; torch.distributions is an INDEPENDENT implementation of the same distributions: element by element."""
    import torch.distributions as D
    rng = np.random.default_rng(7)
    x = rng.standard_normal((5, 9, 13)) * 3.0
    loc = rng.standard_normal((9, 13))
    scale = np.exp(rng.standard_normal((9, 13))) + 1e-6
    got = O.normal_log_prob(x, loc[None], scale[None])
    ref = D.Normal(torch.from_numpy(loc), torch.from_numpy(scale)).log_prob(torch.from_numpy(x)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    # Bernoulli(logits): binary targets (the reference's binarised pixels) over a wide range of logits, incl. +-40 (softplus tails)
    logits = np.concatenate([rng.standard_normal(200) * 8.0, [-40.0, 40.0, 0.0, -1e-9, 700.0, -700.0]])
    xb = (rng.random(logits.shape) < 0.5).astype(np.float64)
    got = O.bernoulli_log_prob(xb, logits)
    ref = D.Bernoulli(logits=torch.from_numpy(logits)).log_prob(torch.from_numpy(xb)).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.sigmoid(logits), torch.sigmoid(torch.from_numpy(logits)).numpy(), rtol=1e-12, atol=1e-15)      # (tanh form: absolute, not relative, in the far tail)
    # KL(N(mu, sigma) || N(0, 1)), the closed form TFP registers for two Normals (iwae1.py:116)
    got = O.kl_normal_std(loc, scale)
    ref = D.kl_divergence(D.Normal(torch.from_numpy(loc), torch.from_numpy(scale)), D.Normal(torch.zeros(9, 13, dtype=torch.float64), torch.ones(9, 13, dtype=torch.float64))).numpy()
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-12)
    # utils.py:6-8 against torch.logsumexp
    lw = rng.standard_normal((50, 11)) * 30.0 - 300.0
    np.testing.assert_allclose(O.logmeanexp(lw, 0), (torch.logsumexp(torch.from_numpy(lw), 0) - np.log(50.0)).numpy(), rtol=1e-12)


@pytest.mark.parametrize("beta", [1.0, 0.7])
def test_forward_dict_matches_a_torch_distributions_restatement(beta):
    """src/iwae1.py:98-151 written a THIRD time, with torch.distributions objects where the reference uses tfd objects (Normal(...).log_prob,
    Bernoulli(logits=...).log_prob, kl_divergence, softmax, logsumexp) instead of hand-typed formulas: every entry of the result dict."""
    import torch.distributions as D
    x, P, eps = MG.inputs(1, 16, 4, 48, 6, 9, 11)
    res = O.forward_1layer(P, x, eps, beta)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64))
    (W1, b1), (W2, b2), (Wm, bm), (Ws, bs), (V1, c1), (V2, c2), (V3, c3) = [(t(W), t(b)) for W, b in P]
    xt, et = t(x), t(eps)
    h = torch.tanh(torch.tanh(xt @ W1 + b1) @ W2 + b2)
    qzx = D.Normal(h @ Wm + bm, torch.exp(h @ Ws + bs) + 1e-6)                    # iwae1.py:31-44
    z = qzx.loc + qzx.scale * et                                                   # :59 (the draw made explicit)
    pxz = D.Bernoulli(logits=torch.tanh(torch.tanh(z @ V1 + c1) @ V2 + c2) @ V3 + c3)   # :79-85
    pz = D.Normal(torch.zeros_like(qzx.loc), torch.ones_like(qzx.loc))
    lpz = pz.log_prob(z).sum(-1)                                                   # :107
    lqzx = qzx.log_prob(z).sum(-1)                                                 # :109
    lpxz = pxz.log_prob(xt.expand(eps.shape[0], *xt.shape)).sum(-1)                # :111
    log_w = lpxz + beta * (lpz - lqzx)                                             # :113
    kl = D.kl_divergence(qzx, pz).sum(-1)                                          # :116
    k = eps.shape[0]
    want = {
        "vae_elbo": log_w.mean(0).mean(),                                          # :120
        "vae_elbo_kl": (lpxz.mean(0) - beta * kl).mean(),                          # :121
        "iwae_elbo": (torch.logsumexp(log_w, 0) - np.log(k)).mean(),               # :125, utils.py:6-8
        "iwae_eq14": (torch.softmax(log_w, 0) * log_w).sum(0).mean(),              # :128-134
        "snis_z": (torch.softmax(log_w, 0).unsqueeze(-1) * z).sum(0),              # :137-139
        "lpxz": lpxz, "lpz": lpz, "lqzx": lqzx, "z": z, "logits": pxz.logits,
    }
    for key, v in want.items():
        np.testing.assert_allclose(res[key], v.numpy(), rtol=1e-10, atol=1e-10, err_msg=key)


def test_log_densities_and_logmeanexp_match_scipy():
    """A second independent implementation of the same semantics (after torch.distributions above): scipy.stats / scipy.special.  tfd.Normal.log_prob =
    norm.logpdf, tfd.Bernoulli(logits).log_prob = bernoulli.logpmf at p = expit(logits) (in the range where p is representable), utils.logmeanexp =
    logsumexp - log k (/root/reference/src/utils.py:6-8, src/iwae1.py:105-125)."""
    import scipy.special as sp
    import scipy.stats as st
    rng = np.random.default_rng(17)
    x = rng.standard_normal((4, 7, 11)) * 2.5
    loc = rng.standard_normal((7, 11))
    scale = np.exp(rng.standard_normal((7, 11)) * 0.7) + 1e-6
    np.testing.assert_allclose(O.normal_log_prob(x, loc[None], scale[None]), st.norm.logpdf(x, loc[None], scale[None]), rtol=1e-12, atol=1e-12)
    logits = rng.standard_normal(300) * 6.0
    xb = (rng.random(300) < 0.5).astype(np.float64)
    np.testing.assert_allclose(O.bernoulli_log_prob(xb, logits), st.bernoulli.logpmf(xb, sp.expit(logits)), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(O.softplus(logits), np.logaddexp(0.0, logits), rtol=1e-13, atol=1e-15)
    lw = rng.standard_normal((37, 5)) * 40.0 - 250.0
    np.testing.assert_allclose(O.logmeanexp(lw, 0), sp.logsumexp(lw, axis=0) - np.log(37.0), rtol=1e-13)
    # KL(N(mu, sigma) || N(0, 1)) as the expectation it is: Gauss-Hermite quadrature of q (log q - log p), no closed form involved (iwae1.py:116)
    xs, ws = np.polynomial.hermite_e.hermegauss(80)
    mu, sg = loc[:3, :4], scale[:3, :4]
    zq = mu[..., None] + sg[..., None] * xs
    kl_quad = ((st.norm.logpdf(zq, mu[..., None], sg[..., None]) - st.norm.logpdf(zq)) * ws).sum(-1) / np.sqrt(2.0 * np.pi)
    np.testing.assert_allclose(O.kl_normal_std(mu, sg), kl_quad, rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("objective", ["iwae_elbo", "vae_elbo", "iwae_eq14", "vae_elbo_kl"])
def test_closed_form_gradients_match_autograd_through_torch_distributions(objective):
    """The oracle's closed-form backward (oracle/iwae_np.py: loss_grads_1layer, the derivation SURVEY 3.3 writes out) against autograd through the
    torch.distributions restatement of the forward above -- tfd objects' semantics from an independent library AND the gradient from an independent
    differentiator (the reference: tape.gradient of -res[objective], src/iwae1.py:153-162).  iwae_eq14's normalised weights are constants of the differentiation
    (tf.stop_gradient, iwae1.py:132) -- .detach() here."""
    import torch.distributions as D
    beta = 0.7 if objective == "vae_elbo_kl" else 1.0
    x, P, eps = MG.inputs(1, 16, 4, 48, 6, 9, 23)
    res, g = O.loss_grads_1layer(P, x, eps, beta, objective)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).requires_grad_(True)
    Pt = [(t(W), t(b)) for W, b in P]
    (W1, b1), (W2, b2), (Wm, bm), (Ws, bs), (V1, c1), (V2, c2), (V3, c3) = Pt
    xt, et = torch.from_numpy(x.astype(np.float64)), torch.from_numpy(eps.astype(np.float64))
    h = torch.tanh(torch.tanh(xt @ W1 + b1) @ W2 + b2)
    qzx = D.Normal(h @ Wm + bm, torch.exp(h @ Ws + bs) + 1e-6)
    z = qzx.loc + qzx.scale * et
    pxz = D.Bernoulli(logits=torch.tanh(torch.tanh(z @ V1 + c1) @ V2 + c2) @ V3 + c3)
    pz = D.Normal(torch.zeros_like(qzx.loc), torch.ones_like(qzx.loc))
    lpxz = pxz.log_prob(xt.expand(eps.shape[0], *xt.shape)).sum(-1)
    log_w = lpxz + beta * (pz.log_prob(z).sum(-1) - qzx.log_prob(z).sum(-1))
    k = eps.shape[0]
    val = {"vae_elbo": log_w.mean(0).mean(),
           "vae_elbo_kl": (lpxz.mean(0) - beta * D.kl_divergence(qzx, pz).sum(-1)).mean(),
           "iwae_elbo": (torch.logsumexp(log_w, 0) - np.log(k)).mean(),
           "iwae_eq14": (torch.softmax(log_w, 0).detach() * log_w).sum(0).mean()}[objective]      # iwae1.py:128-134: the weights are stop_gradient'ed (:132)
    (-val).backward()
    np.testing.assert_allclose(res[objective], val.item(), rtol=1e-10)
    for (dW, db), (Wt, bt) in zip(g, Pt):
        np.testing.assert_allclose(dW, Wt.grad.numpy(), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("beta", [1.0, 0.5])
def test_dreg_gradients_match_autograd_through_torch_distributions(beta):
    """The DReG estimator (/root/reference/tasks/task02.py:34-101): encoder gradients from inference_loss = -mean_b sum_s stop(w~)^2 (lpz + lpxz - lqzx_stopped), where lqzx_stopped
    evaluates z under Normal(stop(mu), stop(sigma) + 1e-6) -- z itself keeps its dependence on the encoder through the reparameterisation -- and decoder gradients from
    -iwae_elbo.  The oracle differentiates this in closed form (its own derivation of dz); here the same construction is written with torch.distributions objects and .detach()
    and differentiated by autograd, the two parameter groups by the two losses as the reference's train_step does (:86-93)."""
    import torch.distributions as D
    x, P, eps = MG.inputs(1, 16, 4, 48, 6, 9, 29)
    res, g = O.loss_grads_1layer(P, x, eps, beta, "dreg")
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).requires_grad_(True)
    Pt = [(t(W), t(b)) for W, b in P]
    (W1, b1), (W2, b2), (Wm, bm), (Ws, bs), (V1, c1), (V2, c2), (V3, c3) = Pt
    xt, et = torch.from_numpy(x.astype(np.float64)), torch.from_numpy(eps.astype(np.float64))
    h = torch.tanh(torch.tanh(xt @ W1 + b1) @ W2 + b2)
    qzx = D.Normal(h @ Wm + bm, torch.exp(h @ Ws + bs) + 1e-6)
    z = qzx.loc + qzx.scale * et
    pxz = D.Bernoulli(logits=torch.tanh(torch.tanh(z @ V1 + c1) @ V2 + c2) @ V3 + c3)
    lpz = D.Normal(torch.zeros_like(qzx.loc), torch.ones_like(qzx.loc)).log_prob(z).sum(-1)
    lqzx = qzx.log_prob(z).sum(-1)
    lpxz = pxz.log_prob(xt.expand(eps.shape[0], *xt.shape)).sum(-1)
    log_w = lpxz + beta * (lpz - lqzx)
    iwae_elbo = (torch.logsumexp(log_w, 0) - np.log(eps.shape[0])).mean()
    al = torch.softmax(log_w, 0)
    lqzx_stopped = D.Normal(qzx.loc.detach(), qzx.scale.detach() + 1e-6).log_prob(z).sum(-1)
    inference_loss = -((al.detach() ** 2) * (lpz + lpxz - lqzx_stopped)).sum(0).mean()
    enc = [p for pair in Pt[:4] for p in pair]
    dec = [p for pair in Pt[4:] for p in pair]
    g_enc = torch.autograd.grad(inference_loss, enc, retain_graph=True)
    g_dec = torch.autograd.grad(-iwae_elbo, dec)
    np.testing.assert_allclose(res["iwae_elbo"], iwae_elbo.item(), rtol=1e-10)
    np.testing.assert_allclose(res["inference_loss"], inference_loss.item(), rtol=1e-10)
    flat_oracle = [a for pair in g for a in pair]
    for got, ref in zip(flat_oracle, list(g_enc) + list(g_dec)):
        np.testing.assert_allclose(got, ref.numpy(), rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("objective", ["iwae_elbo", "vae_elbo", "iwae_eq14"])
def test_two_layer_gradients_match_autograd_through_torch_distributions(objective):
    """The hierarchical model (/root/reference/src/iwae2.py:58-167): q(z1|x) q(z2|z1), p(z2) p(z1|z2) p(x|z1), log_w = lpxz1 + lpz1z2 + lpz2 - lqz1x - lqz2z1 (:128, no beta
    inside), written with torch.distributions objects and differentiated by autograd, against the oracle's closed-form backward of the same step."""
    import torch.distributions as D
    nh, nl = [16, 8], [6, 3]
    x, P, eps = MG.inputs(2, nh, nl, 48, 5, 7, 31)
    res, g = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, objective)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).requires_grad_(True)
    Pt = [(t(W), t(b)) for W, b in P]
    xt = torch.from_numpy(x.astype(np.float64))
    e1, e2 = torch.from_numpy(eps[0].astype(np.float64)), torch.from_numpy(eps[1].astype(np.float64))

    def block(inp, p4):      # BasicBlock: two tanh layers, mu head, sigma = exp(.) + 1e-6 head (iwae2.py:24-44)
        (A1, a1), (A2, a2), (Am, am), (As, a_s) = p4
        hh = torch.tanh(torch.tanh(inp @ A1 + a1) @ A2 + a2)
        return D.Normal(hh @ Am + am, torch.exp(hh @ As + a_s) + 1e-6)

    # parameter order of the oracle / the flat vector: enc1 (4 layers), enc2 (4), dec2 (4), dec1 (3)
    qz1x = block(xt, Pt[0:4])
    z1 = qz1x.loc + qz1x.scale * e1                                   # [k, B, D1]
    qz2z1 = block(z1, Pt[4:8])
    z2 = qz2z1.loc + qz2z1.scale * e2
    pz1z2 = block(z2, Pt[8:12])
    (V1, c1), (V2, c2), (V3, c3) = Pt[12:15]
    pxz1 = D.Bernoulli(logits=torch.tanh(torch.tanh(z1 @ V1 + c1) @ V2 + c2) @ V3 + c3)
    lpz2 = D.Normal(torch.zeros_like(z2), torch.ones_like(z2)).log_prob(z2).sum(-1)
    lpz1z2 = pz1z2.log_prob(z1).sum(-1)
    lqz1x = qz1x.log_prob(z1).sum(-1)
    lqz2z1 = qz2z1.log_prob(z2).sum(-1)
    lpxz1 = pxz1.log_prob(xt.expand(e1.shape[0], *xt.shape)).sum(-1)
    log_w = lpxz1 + lpz1z2 + lpz2 - lqz1x - lqz2z1
    k = e1.shape[0]
    val = {"vae_elbo": log_w.mean(0).mean(),
           "iwae_elbo": (torch.logsumexp(log_w, 0) - np.log(k)).mean(),
           "iwae_eq14": (torch.softmax(log_w, 0).detach() * log_w).sum(0).mean()}[objective]
    (-val).backward()
    np.testing.assert_allclose(res[objective], val.item(), rtol=1e-10)
    for (dW, db), (Wt, bt) in zip(g, Pt):
        np.testing.assert_allclose(dW, Wt.grad.numpy(), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-8, atol=1e-11)


@pytest.mark.parametrize("cond_prior", [False, True])
def test_conditional_model_gradients_match_autograd_through_torch_distributions(cond_prior):
    """The conditional models: tasks/task05.py:108-135 (encoder on concat(x, y), decoder on concat(z, y), N(0, 1) prior) and tasks/task04.py:100-135 (the same plus the learned
    prior p(z|y) = BasicBlock(y), whose four layers sit AFTER the decoder's in the weight order), beta = 0.5 -- torch.distributions objects + autograd against the oracle's
    closed-form backward (the prior block's gradient included)."""
    import torch.distributions as D
    beta = 0.5
    x, P, eps, y = MG.inputs(1, 16, 4, 48, 6, 9, 37, cond=5, cond_prior=cond_prior)
    res, g = O.loss_grads_1layer(P, x, eps, beta, "iwae_elbo", y=y)
    t = lambda a: torch.from_numpy(np.asarray(a, dtype=np.float64)).requires_grad_(True)
    Pt = [(t(W), t(b)) for W, b in P]
    xt, et, yt = (torch.from_numpy(a.astype(np.float64)) for a in (x, eps, y))

    def block(inp, p4):
        (A1, a1), (A2, a2), (Am, am), (As, a_s) = p4
        hh = torch.tanh(torch.tanh(inp @ A1 + a1) @ A2 + a2)
        return D.Normal(hh @ Am + am, torch.exp(hh @ As + a_s) + 1e-6)

    qzxy = block(torch.cat([xt, yt], -1), Pt[0:4])
    z = qzxy.loc + qzxy.scale * et
    zy = torch.cat([z, yt.expand(eps.shape[0], *yt.shape)], -1)
    (V1, c1), (V2, c2), (V3, c3) = Pt[4:7]
    pxzy = D.Bernoulli(logits=torch.tanh(torch.tanh(zy @ V1 + c1) @ V2 + c2) @ V3 + c3)
    pz = block(yt, Pt[7:11]) if cond_prior else D.Normal(torch.zeros_like(qzxy.loc), torch.ones_like(qzxy.loc))
    log_w = pxzy.log_prob(xt.expand(eps.shape[0], *xt.shape)).sum(-1) + beta * (pz.log_prob(z).sum(-1) - qzxy.log_prob(z).sum(-1))
    val = (torch.logsumexp(log_w, 0) - np.log(eps.shape[0])).mean()
    (-val).backward()
    np.testing.assert_allclose(res["iwae_elbo"], val.item(), rtol=1e-10)
    assert len(g) == len(Pt) == (11 if cond_prior else 7)
    for (dW, db), (Wt, bt) in zip(g, Pt):
        np.testing.assert_allclose(dW, Wt.grad.numpy(), rtol=1e-8, atol=1e-11)
        np.testing.assert_allclose(db, bt.grad.numpy(), rtol=1e-8, atol=1e-11)


def test_keras_adam_matches_torch_adam_with_the_epsilon_moved():
    """Keras Adam (main.py:56, epsilon = 1e-4): theta -= lr sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps) -- epsilon OUTSIDE the bias correction of v.  torch.optim.Adam
    divides by sqrt(v / (1 - b2^t)) + eps', i.e. the same update with eps' = eps / sqrt(1 - b2^t): an independent implementation of the moment recurrences and both bias
    corrections, driven for 25 steps with that epsilon set per step."""
    rng = np.random.default_rng(3)
    n, lr = 257, 1e-3
    th = rng.standard_normal(n)
    p = torch.nn.Parameter(torch.from_numpy(th.copy()))
    opt = torch.optim.Adam([p], lr=lr, betas=(0.9, 0.999), eps=1e-4)
    m = v = 0.0
    for t in range(1, 26):
        gr = rng.standard_normal(n) * 10.0 ** rng.integers(-4, 1)
        th, m, v = O.adam_update(th, gr, m, v, t, lr)
        opt.param_groups[0]["eps"] = 1e-4 / np.sqrt(1.0 - 0.999 ** t)
        p.grad = torch.from_numpy(gr.copy())
        opt.step()
        np.testing.assert_allclose(th, p.detach().numpy(), rtol=1e-12, atol=1e-14)
