"""CPU oracle for the IWAE hot path -- TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (nbip/IWAE) ships no tests, golden vectors or
fixtures, and TensorFlow / TensorFlow-Probability (un-pinned third-party
dependencies that hold all of its arithmetic) are not installed in this image, so
this restatement could not be checked against reference outputs.  It follows the
reference source line by line (citations below) and the published semantics of
the TF/TFP ops it calls; it is cross-checked against an
independent torch-autograd twin (oracle/iwae_torch.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product path (iwae_amd/) never does.

Everything is NumPy float64.  Reference tensor layout is kept: sample axis
first, [k, B, ...] (iwae1.py:59, :107-125).

An optional `rnd` callable models the points where the MI355X kernels round a
GEMM operand to bf16 (see DESIGN.md "rounding points"); rnd=None is the exact
restatement.
"""
import numpy as np

LOG2PI = float(np.log(2.0 * np.pi))
SIGMA_EPS = 1e-6  # iwae1.py:42, iwae2.py:43

OBJECTIVES_1L = ("vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl")  # main.py:23, iwae1.py:141-144
OBJECTIVES_2L = ("vae_elbo", "iwae_elbo", "iwae_eq14")                # iwae2.py:154-156


# --------------------------------------------------------------------------
# rounding helpers
# --------------------------------------------------------------------------
def bf16_round(a):
    """Round-to-nearest-even to bfloat16, returned as float64 (via float32)."""
    a32 = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    u = a32.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).astype(np.float64).reshape(a32.shape)


def _id(a):
    return a


# --------------------------------------------------------------------------
# parameters (Keras trainable_weights creation order)
# --------------------------------------------------------------------------
def layer_shapes(n_layers, n_hidden, n_latent, x_dim=784, cond_dim=0, cond_prior=False):
    """[(name, (in, out))] for every Dense, in Keras creation order.
    cond_dim > 0 (1-layer only): the conditional model of tasks/task05.py, encoder input x_dim + cond_dim,
    decoder input n_latent + cond_dim.  cond_prior: tasks/task04.py:108 adds a BasicBlock(cond_dim -> n_latent) AFTER the
    decoder (Keras creation order), the learned prior p(z|y).

    1-layer: iwae1.py:31-34 (BasicBlock), :72-75 (decoder).
    2-layer: iwae2.py:55-56 (two encoder blocks), :77-87 (decoder block + MLP).
    """
    if n_layers == 1:
        H, D = int(n_hidden), int(n_latent)
        out = [("enc.l1", (x_dim + cond_dim, H)), ("enc.l2", (H, H)), ("enc.lmu", (H, D)), ("enc.lstd", (H, D)),
               ("dec.d1", (D + cond_dim, H)), ("dec.d2", (H, H)), ("dec.out", (H, x_dim))]
        if cond_prior:
            out += [("prior.l1", (cond_dim, H)), ("prior.l2", (H, H)), ("prior.lmu", (H, D)), ("prior.lstd", (H, D))]
        return out
    H1, H2 = int(n_hidden[0]), int(n_hidden[1])
    D1, D2 = int(n_latent[0]), int(n_latent[1])
    return [("enc1.l1", (x_dim, H1)), ("enc1.l2", (H1, H1)), ("enc1.lmu", (H1, D1)), ("enc1.lstd", (H1, D1)),
            ("enc2.l1", (D1, H2)), ("enc2.l2", (H2, H2)), ("enc2.lmu", (H2, D2)), ("enc2.lstd", (H2, D2)),
            ("dec2.l1", (D2, H2)), ("dec2.l2", (H2, H2)), ("dec2.lmu", (H2, D1)), ("dec2.lstd", (H2, D1)),
            ("dec1.d1", (D1, H1)), ("dec1.d2", (H1, H1)), ("dec1.out", (H1, x_dim))]


def output_bias_from_mean(train_mean):
    """utils.py:19-21: logit of the clipped per-pixel training mean."""
    m = np.clip(np.asarray(train_mean, dtype=np.float64), 0.001, 0.999)
    return -np.log(1.0 / m - 1.0)


def init_params(n_layers, n_hidden, n_latent, seed, x_mean=None, x_dim=784, cond_dim=0, cond_prior=False):
    """Keras Dense defaults: glorot-uniform kernel, zero bias; final decoder bias
    from the data mean (iwae1.py:74-75, utils.py:11-23).  Returns list of
    (W [in,out], b [out]) float64."""
    rng = np.random.default_rng(seed)
    params = []
    shapes = layer_shapes(n_layers, n_hidden, n_latent, x_dim, cond_dim, cond_prior)
    for idx, (name, (fi, fo)) in enumerate(shapes):
        lim = np.sqrt(6.0 / (fi + fo))
        W = rng.uniform(-lim, lim, size=(fi, fo))
        b = np.zeros(fo)
        if name.endswith(".out") and x_mean is not None:
            b = output_bias_from_mean(x_mean)
        params.append((W, b))
    return params


def flatten_params(params):
    return np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for W, b in params])


def unflatten_params(flat, n_layers, n_hidden, n_latent, x_dim=784, cond_dim=0, cond_prior=False):
    out, o = [], 0
    for _, (fi, fo) in layer_shapes(n_layers, n_hidden, n_latent, x_dim, cond_dim, cond_prior):
        W = np.asarray(flat[o:o + fi * fo], dtype=np.float64).reshape(fi, fo); o += fi * fo
        b = np.asarray(flat[o:o + fo], dtype=np.float64).copy(); o += fo
        out.append((W, b))
    return out


# --------------------------------------------------------------------------
# building blocks
# --------------------------------------------------------------------------
def logmeanexp(log_w, axis):
    """utils.py:6-8."""
    m = np.max(log_w, axis=axis)
    return np.log(np.mean(np.exp(log_w - np.expand_dims(m, axis)), axis=axis)) + m


def softplus(l):
    return np.maximum(l, 0.0) + np.log1p(np.exp(-np.abs(l)))


def sigmoid(l):
    return 0.5 * (1.0 + np.tanh(0.5 * l))


def normal_log_prob(x, loc, scale):
    """tfd.Normal.log_prob (iwae1.py:107,109): -0.5((x-loc)/scale)^2 - 0.5 log 2pi - log scale."""
    u = (x - loc) / scale
    return -0.5 * u * u - 0.5 * LOG2PI - np.log(scale)


def bernoulli_log_prob(x, logits):
    """tfd.Bernoulli(logits).log_prob (iwae1.py:83,111): x*l - softplus(l)."""
    return x * logits - softplus(logits)


def kl_normal_std(mu, sigma):
    """tfd.kl_divergence(Normal(mu,sigma), Normal(0,1)) (iwae1.py:116)."""
    ls = np.log(sigma)
    return 0.5 * mu * mu + 0.5 * np.expm1(2.0 * ls) - ls


class _Dense:
    """y = act(rnd(x) @ rnd(W) + b) with a tape for the closed-form backward."""

    def __init__(self, W, b, act, rnd):
        self.W, self.b, self.act, self.rnd = W, b, act, rnd

    def fwd(self, x_r):
        # x_r is already in the precision the kernel stores it in
        self.x = x_r
        pre = x_r @ self.rnd(self.W) + self.b
        if self.act == "tanh":
            self.y = self.rnd(np.tanh(pre))      # stored activation (bf16 on device)
        else:
            self.y = pre                         # heads / logits stay fp32-accumulated
        return self.y

    def bwd(self, dy, need_dx=True):
        """dy = dLoss/dy (post-activation, except for act None where it is dpre).
        Returns dx; leaves self.dW, self.db."""
        if self.act == "tanh":
            dpre = dy * (1.0 - self.y * self.y)
        else:
            dpre = dy
        dpre_r = self.rnd(dpre)                  # gradient operand as stored (bf16)
        self.dpre = dpre_r
        x2 = self.x.reshape(-1, self.x.shape[-1])
        d2 = dpre_r.reshape(-1, dpre_r.shape[-1])
        self.dW = x2.T @ d2
        self.db = d2.sum(axis=0)
        if not need_dx:
            return None
        return dpre_r @ self.rnd(self.W).T


class _Block:
    """BasicBlock (iwae1.py:24-44): two tanh layers + mu head + exp-sigma head (+1e-6)."""

    def __init__(self, p4, rnd):
        (W1, b1), (W2, b2), (Wm, bm), (Ws, bs) = p4
        self.l1 = _Dense(W1, b1, "tanh", rnd)
        self.l2 = _Dense(W2, b2, "tanh", rnd)
        self.lmu = _Dense(Wm, bm, None, rnd)
        self.lstd = _Dense(Ws, bs, None, rnd)
        self.rnd = rnd

    def fwd(self, x_r):
        h2 = self.l2.fwd(self.l1.fwd(x_r))
        self.mu = self.lmu.fwd(h2)
        self.a = self.lstd.fwd(h2)
        self.sigma = np.exp(self.a) + SIGMA_EPS      # iwae1.py:34,42
        return self.mu, self.sigma

    def bwd(self, dmu, dsigma, need_dx=True):
        da = dsigma * np.exp(self.a)
        dh2 = self.lmu.bwd(dmu) + self.lstd.bwd(da)
        dh1 = self.l2.bwd(dh2)
        return self.l1.bwd(dh1, need_dx=need_dx)

    def grads(self):
        return [(l.dW, l.db) for l in (self.l1, self.l2, self.lmu, self.lstd)]


class _MLP3:
    """decode_z_to_x (iwae1.py:70-77): tanh, tanh, linear."""

    def __init__(self, p3, rnd):
        (W1, b1), (W2, b2), (W3, b3) = p3
        self.d1 = _Dense(W1, b1, "tanh", rnd)
        self.d2 = _Dense(W2, b2, "tanh", rnd)
        self.out = _Dense(W3, b3, None, rnd)

    def fwd(self, z_r):
        return self.out.fwd(self.d2.fwd(self.d1.fwd(z_r)))

    def bwd(self, dlogits):
        return self.d1.bwd(self.d2.bwd(self.out.bwd(dlogits)))

    def grads(self):
        return [(l.dW, l.db) for l in (self.d1, self.d2, self.out)]


def _weights_over_k(log_w):
    m = np.max(log_w, axis=0, keepdims=True)
    w = np.exp(log_w - m)
    return w / np.sum(w, axis=0, keepdims=True)


# --------------------------------------------------------------------------
# 1-layer model: iwae1.py:98-151 (+ DReG: tasks/task02.py:34-85)
# --------------------------------------------------------------------------
def forward_1layer(params, x, eps, beta=1.0, rnd=None, dreg=False, _tape=None, y=None):
    """x [B,X] in {0,1}; eps [k,B,D] ~ N(0,1) (the draw of qzx.sample, iwae1.py:59).
    Returns the reference's result dict (iwae1.py:141-151) as float64 arrays.
    y [B,C] (optional): the conditional model of tasks/task05.py:108-122 -- the encoder sees concat(x, y), the
    decoder concat(z, y) (y one-hot there; any float condition here), prior N(0,1), likelihood over x only.
    With 11 parameter pairs (tasks/task04.py:101-173) the last four are the conditional prior network: p(z|y) =
    N(mu_p(y), sigma_p(y)) replaces N(0,1) in lpz; the analytic KL of vae_elbo_kl stays against N(0,1) (:131)."""
    rnd = rnd or _id
    x = np.asarray(x, dtype=np.float64)
    eps = np.asarray(eps, dtype=np.float64)
    k, B, D = eps.shape
    enc = _Block(params[0:4], rnd)
    dec = _MLP3(params[4:7], rnd)
    if y is None:
        mu, sigma = enc.fwd(rnd(x))                   # iwae1.py:57
    else:
        y = np.asarray(y, dtype=np.float64)
        mu, sigma = enc.fwd(rnd(np.concatenate([x, y], axis=-1)))                     # task05.py:113-114
    z = mu[None] + sigma[None] * eps                  # iwae1.py:59 (reparameterised sample)
    if y is None:
        logits = dec.fwd(rnd(z))                      # iwae1.py:81
    else:
        logits = dec.fwd(rnd(np.concatenate([z, np.broadcast_to(y[None], (k, B, y.shape[-1]))], axis=-1)))   # task05.py:117-118
    prior = None
    if y is not None and len(params) == 11:
        prior = _Block(params[7:11], rnd)
        mu_p, sig_p = prior.fwd(rnd(y))                                # task04.py:124
        lpz = np.sum(normal_log_prob(z, mu_p[None], sig_p[None]), axis=-1)   # :130
    else:
        lpz = np.sum(normal_log_prob(z, 0.0, 1.0), axis=-1)        # :107
    lqzx = np.sum(normal_log_prob(z, mu[None], sigma[None]), axis=-1)  # :109
    lpxz = np.sum(bernoulli_log_prob(x[None], logits), axis=-1)    # :111
    log_w = lpxz + beta * (lpz - lqzx)                # :113
    kl = np.sum(kl_normal_std(mu, sigma), axis=-1)    # :116
    res = {}
    res["vae_elbo"] = np.mean(np.mean(log_w, axis=0), axis=-1)      # :120
    res["vae_elbo_kl"] = np.mean(lpxz) - beta * np.mean(kl)         # :121
    res["iwae_elbo"] = np.mean(logmeanexp(log_w, axis=0), axis=-1)  # :125
    wn = _weights_over_k(log_w)                                     # :128-132
    res["iwae_eq14"] = np.mean(np.sum(wn * log_w, axis=0))          # :134
    al = wn                                                         # :137 softmax over axis 0
    res["snis_z"] = np.sum(al[:, :, None] * z, axis=0)              # :139
    res.update(z=z, al=al, logits=logits, lpxz=lpxz, lpz=lpz, lqzx=lqzx)
    if dreg:
        # task02.py:61-76.  qzx.scale already holds +1e-6 and :63 adds another.
        sig2 = sigma + SIGMA_EPS
        lq_st = np.sum(normal_log_prob(z, mu[None], sig2[None]), axis=-1)
        stopped_log_w = lpz + lpxz - lq_st                          # :70 (no beta)
        res["inference_loss"] = -np.mean(np.sum(al * al * stopped_log_w, axis=0), axis=-1)  # :73-76
    if _tape is not None:
        _tape.update(enc=enc, dec=dec, mu=mu, sigma=sigma, eps=eps, z=z, x=x, log_w=log_w,
                     wn=wn, logits=logits, beta=beta, k=k, B=B, prior=prior)
        if prior is not None:
            _tape.update(mu_p=mu_p, sig_p=sig_p)
    return res


def loss_grads_1layer(params, x, eps, beta=1.0, objective="iwae_elbo", rnd=None, tape=None, y=None):
    """Closed-form gradient of loss = -res[objective] (iwae1.py:155-159) w.r.t. the
    14 tensors; objective "dreg" = tasks/task02.py:87-101 (encoder <- inference_loss,
    decoder <- -iwae_elbo).  Returns (res, [(dW, db)...])."""
    rnd = rnd or _id
    tape = {} if tape is None else tape
    dreg = objective == "dreg"
    res = forward_1layer(params, x, eps, beta, rnd, dreg=dreg, _tape=tape, y=y)
    enc, dec = tape["enc"], tape["dec"]
    mu, sigma, z, wn, xx = tape["mu"], tape["sigma"], tape["z"], tape["wn"], tape["x"]
    k, B = tape["k"], tape["B"]
    p = sigmoid(tape["logits"])
    dmu_extra = 0.0
    dsig_extra = 0.0
    if objective in ("iwae_elbo", "iwae_eq14", "dreg"):
        G = -wn / B                       # dLoss/dlog_w  (eq14 has the same gradient)
    elif objective == "vae_elbo":
        G = -np.ones_like(wn) / (k * B)
    elif objective == "vae_elbo_kl":
        G = -np.ones_like(wn) / (k * B)
    else:
        raise KeyError(objective)         # iwae1.py:157 raises KeyError for unknown keys
    dlogits = G[:, :, None] * (xx[None] - p)          # d lpxz / d logits = x - sigmoid(l)
    dz_dec = dec.bwd(dlogits)[..., :z.shape[-1]]     # conditional model: the decoder input is concat(z, y)
    tape.update(G=G, dz_dec=dz_dec)
    if objective == "dreg":
        # encoder gets d inference_loss; decoder backward is linear in the row weight
        sig2 = sigma + SIGMA_EPS
        coef = (wn * wn) / B                          # stop-gradient weights squared / B
        dz = wn[:, :, None] * dz_dec + coef[:, :, None] * (z - (z - mu[None]) / (sig2[None] ** 2))
        dmu = dz.sum(axis=0)
        dsig = (dz * tape["eps"]).sum(axis=0)
    elif objective == "vae_elbo_kl":
        # loss = -(mean lpxz - beta mean_b kl): only lpxz reaches z; KL is analytic in mu, sigma
        dz = dz_dec
        dmu = dz.sum(axis=0) + beta * mu / B
        dsig = (dz * tape["eps"]).sum(axis=0) + beta * (sigma - 1.0 / sigma) / B
    else:
        Gb = G * beta
        if tape.get("prior") is None:
            dz = dz_dec - Gb[:, :, None] * z          # lpz: d/dz = -z
        else:                                          # lpz = log N(z; mu_p, sig_p): d/dz = -(z - mu_p)/sig_p^2
            mu_p, sig_p = tape["mu_p"], tape["sig_p"]
            u = (z - mu_p[None]) / sig_p[None]
            dz = dz_dec - Gb[:, :, None] * u / sig_p[None]
            dmu_p = (Gb[:, :, None] * u / sig_p[None]).sum(axis=0)
            dsig_p = (Gb[:, :, None] * (u * u - 1.0) / sig_p[None]).sum(axis=0)
        dmu = dz.sum(axis=0)
        dsig = (dz * tape["eps"]).sum(axis=0) + (Gb[:, :, None] / sigma[None]).sum(axis=0)
    enc.bwd(dmu, dsig, need_dx=False)
    grads = enc.grads() + dec.grads()
    prior = tape.get("prior")
    if prior is not None:
        if objective == "vae_elbo_kl":                 # lpz is not part of that objective (task04.py:136)
            prior.bwd(np.zeros_like(tape["mu_p"]), np.zeros_like(tape["sig_p"]), need_dx=False)
        else:
            prior.bwd(dmu_p, dsig_p, need_dx=False)
        grads = grads + prior.grads()
    return res, grads


# --------------------------------------------------------------------------
# 2-layer model: iwae2.py:58-67, :89-96, :109-167 (beta is ignored there)
# --------------------------------------------------------------------------
def forward_2layer(params, x, eps1, eps2, beta=1.0, rnd=None, _tape=None):
    rnd = rnd or _id
    x = np.asarray(x, dtype=np.float64)
    eps1 = np.asarray(eps1, dtype=np.float64)
    eps2 = np.asarray(eps2, dtype=np.float64)
    k, B, _ = eps1.shape
    enc1 = _Block(params[0:4], rnd)
    enc2 = _Block(params[4:8], rnd)
    dec2 = _Block(params[8:12], rnd)
    dec1 = _MLP3(params[12:15], rnd)
    mu1, sig1 = enc1.fwd(rnd(x))                      # iwae2.py:59
    z1 = mu1[None] + sig1[None] * eps1                # :61
    mu2, sig2 = enc2.fwd(rnd(z1))                     # :63
    z2 = mu2 + sig2 * eps2                            # :65
    mup, sigp = dec2.fwd(rnd(z2))                     # :90
    logits = dec1.fwd(rnd(z1))                        # :92
    lpz2 = np.sum(normal_log_prob(z2, 0.0, 1.0), axis=-1)          # :118
    lqz2z1 = np.sum(normal_log_prob(z2, mu2, sig2), axis=-1)       # :120
    lpz1z2 = np.sum(normal_log_prob(z1, mup, sigp), axis=-1)       # :122
    lqz1x = np.sum(normal_log_prob(z1, mu1[None], sig1[None]), axis=-1)  # :124
    lpxz1 = np.sum(bernoulli_log_prob(x[None], logits), axis=-1)   # :126
    log_w = lpxz1 + lpz1z2 + lpz2 - lqz1x - lqz2z1                 # :128
    res = {}
    res["vae_elbo"] = np.mean(np.mean(log_w, axis=0), axis=-1)      # :132
    res["iwae_elbo"] = np.mean(logmeanexp(log_w, axis=0), axis=-1)  # :136
    wn = _weights_over_k(log_w)
    res["iwae_eq14"] = np.mean(np.sum(wn * log_w, axis=0))          # :145
    res["snis_z1"] = np.sum(wn[:, :, None] * z1, axis=0)            # :150
    res["snis_z2"] = np.sum(wn[:, :, None] * z2, axis=0)            # :152
    res.update(z1=z1, z2=z2, al=wn, logits=logits, lpxz1=lpxz1, lpz1z2=lpz1z2, lpz2=lpz2,
               lqz1x=lqz1x, lqz2z1=lqz2z1)
    if _tape is not None:
        _tape.update(enc1=enc1, enc2=enc2, dec2=dec2, dec1=dec1, mu1=mu1, sig1=sig1, mu2=mu2, sig2=sig2,
                     mup=mup, sigp=sigp, z1=z1, z2=z2, eps1=eps1, eps2=eps2, x=x, wn=wn, logits=logits,
                     k=k, B=B)
    return res


def loss_grads_2layer(params, x, eps1, eps2, beta=1.0, objective="iwae_elbo", rnd=None, tape=None):
    """Closed form of tape.gradient(-res[objective]) for iwae2.py:169-178."""
    rnd = rnd or _id
    if objective not in OBJECTIVES_2L:
        raise KeyError(objective)         # iwae2.py:173: 'vae_elbo_kl' is a KeyError in the reference
    t = {} if tape is None else tape
    res = forward_2layer(params, x, eps1, eps2, beta, rnd, _tape=t)
    k, B, wn = t["k"], t["B"], t["wn"]
    G = (-wn / B) if objective in ("iwae_elbo", "iwae_eq14") else (-np.ones_like(wn) / (k * B))
    G3 = G[:, :, None]
    p = sigmoid(t["logits"])
    dz1 = t["dec1"].bwd(G3 * (t["x"][None] - p))                    # via lpxz1
    u = (t["z1"] - t["mup"]) / t["sigp"]                            # via lpz1z2
    dz1 = dz1 + G3 * (-u / t["sigp"])
    dmup = G3 * (u / t["sigp"])
    dsigp = G3 * ((u * u - 1.0) / t["sigp"])
    dz2 = t["dec2"].bwd(dmup, dsigp)
    dz2 = dz2 + G3 * (-t["z2"])                                     # via lpz2
    dmu2 = dz2                                                      # -lqz2z1: total d/dmu2 = 0
    dsig2 = dz2 * t["eps2"] + G3 / t["sig2"]                        # -lqz2z1: total d/dsig2 = +G/sig2
    dz1 = dz1 + t["enc2"].bwd(dmu2, dsig2)
    dmu1 = dz1.sum(axis=0)
    dsig1 = (dz1 * t["eps1"]).sum(axis=0) + (G3 / t["sig1"][None]).sum(axis=0)
    t["enc1"].bwd(dmu1, dsig1, need_dx=False)
    return res, t["enc1"].grads() + t["enc2"].grads() + t["dec2"].grads() + t["dec1"].grads()


# --------------------------------------------------------------------------
# Keras Adam (main.py:93): epsilon OUTSIDE the bias correction
# --------------------------------------------------------------------------
def adam_update(flat, grad, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-4):
    """One step, t is the 1-based step count AFTER increment. Returns (flat, m, v)."""
    m = beta1 * m + (1.0 - beta1) * grad
    v = beta2 * v + (1.0 - beta2) * grad * grad
    alpha = lr * np.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)
    flat = flat - alpha * m / (np.sqrt(v) + eps)
    return flat, m, v


def flatten_grads(grads):
    return np.concatenate([np.concatenate([dW.ravel(), db.ravel()]) for dW, db in grads])


def learning_rate_schedule():
    """main.py:44-51: {first epoch: lr}, total epochs (3280)."""
    epochs, d = 0, {}
    for i in range(8):
        d[epochs] = 0.001 * 10 ** (-i / 7)
        epochs += 3 ** i
    return d, epochs


# --------------------------------------------------------------------------
# synthetic MNIST-like data: smooth centred blob, global mean ~0.13
# --------------------------------------------------------------------------
def synthetic_pixel_means(x_dim=784):
    side = int(round(np.sqrt(x_dim)))
    yy, xx = np.mgrid[0:side, 0:side]
    c = (side - 1) / 2.0
    r2 = ((yy - c) ** 2 + (xx - c) ** 2) / (0.30 * side) ** 2
    p = 0.62 * np.exp(-r2)
    return p.reshape(-1)[:x_dim]


def synthetic_binarized(n, seed, x_dim=784):
    rng = np.random.default_rng(seed)
    p = synthetic_pixel_means(x_dim)
    return (rng.random((n, x_dim)) < p[None]).astype(np.float32)   # utils.py:26-27 semantics
