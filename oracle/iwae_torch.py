"""Torch-autograd twin of oracle/iwae_np.py -- TEST INFRASTRUCTURE ONLY (parity unpinned,
see iwae_np.py header).

Purpose: (1) an independent check of the closed-form backward in iwae_np.py: the
forward below is written op-for-op after the reference (iwae1.py:98-151,
iwae2.py:109-167, tasks/task02.py:34-101) and the gradient comes from autograd,
exactly how the reference obtains it from tf.GradientTape (iwae1.py:155-159);
(2) the CPU baseline that bench.py times ("port" kind): the same graph in fp32 on
all host cores, Adam(eps=1e-4) included.
"""
import math
import torch

LOG2PI = math.log(2.0 * math.pi)
SIGMA_EPS = 1e-6


def _normal_lp(x, loc, scale):
    u = (x - loc) / scale
    return -0.5 * u * u - 0.5 * LOG2PI - torch.log(scale)


def _bern_lp(x, logits):
    return x * logits - torch.nn.functional.softplus(logits)


def _logmeanexp(log_w, dim):
    m = torch.max(log_w, dim=dim).values          # utils.py:7 (gradient flows through max in TF too)
    return torch.log(torch.mean(torch.exp(log_w - m.unsqueeze(dim)), dim=dim)) + m


def _block(p4, h):
    (W1, b1), (W2, b2), (Wm, bm), (Ws, bs) = p4
    h = torch.tanh(h @ W1 + b1)
    h = torch.tanh(h @ W2 + b2)
    return h @ Wm + bm, torch.exp(h @ Ws + bs) + SIGMA_EPS


def _mlp3(p3, z):
    (W1, b1), (W2, b2), (W3, b3) = p3
    return torch.tanh(torch.tanh(z @ W1 + b1) @ W2 + b2) @ W3 + b3


def forward_1layer(params, x, eps, beta=1.0, dreg=False, y=None):
    """y [B,C]: conditional model of tasks/task05.py:108-122 (encoder on concat(x,y), decoder on concat(z,y))."""
    mu, sigma = _block(params[0:4], x if y is None else torch.cat([x, y], dim=-1))
    z = mu.unsqueeze(0) + sigma.unsqueeze(0) * eps
    logits = _mlp3(params[4:7], z if y is None else torch.cat([z, y.unsqueeze(0).expand(z.shape[0], -1, -1)], dim=-1))
    if y is not None and len(params) == 11:           # tasks/task04.py:124-130: learned conditional prior p(z|y)
        mu_p, sig_p = _block(params[7:11], y)
        lpz = _normal_lp(z, mu_p.unsqueeze(0), sig_p.unsqueeze(0)).sum(-1)
    else:
        lpz = _normal_lp(z, torch.zeros((), dtype=z.dtype), torch.ones((), dtype=z.dtype)).sum(-1)
    lqzx = _normal_lp(z, mu.unsqueeze(0), sigma.unsqueeze(0)).sum(-1)
    lpxz = _bern_lp(x.unsqueeze(0), logits).sum(-1)
    log_w = lpxz + beta * (lpz - lqzx)
    ls = torch.log(sigma)
    kl = (0.5 * mu * mu + 0.5 * torch.expm1(2.0 * ls) - ls).sum(-1)
    res = {}
    res["vae_elbo"] = log_w.mean(0).mean(-1)
    res["vae_elbo_kl"] = lpxz.mean() - beta * kl.mean()
    res["iwae_elbo"] = _logmeanexp(log_w, 0).mean(-1)
    m = log_w.max(dim=0, keepdim=True).values
    w = torch.exp(log_w - m)
    wn = (w / w.sum(0, keepdim=True)).detach()
    res["iwae_eq14"] = (wn * log_w).sum(0).mean()
    al = torch.softmax(log_w, dim=0)
    res["snis_z"] = (al.unsqueeze(-1) * z).sum(0)
    res.update(z=z, al=al, logits=logits, lpxz=lpxz, lpz=lpz, lqzx=lqzx)
    if dreg:
        mu_s, sig_s = mu.detach(), sigma.detach()
        lq_st = _normal_lp(z, mu_s.unsqueeze(0), (sig_s + SIGMA_EPS).unsqueeze(0)).sum(-1)
        sq = al.detach() ** 2
        res["inference_loss"] = -((sq * (lpz + lpxz - lq_st)).sum(0)).mean(-1)
    return res


def forward_2layer(params, x, eps1, eps2, beta=1.0):
    mu1, sig1 = _block(params[0:4], x)
    z1 = mu1.unsqueeze(0) + sig1.unsqueeze(0) * eps1
    mu2, sig2 = _block(params[4:8], z1)
    z2 = mu2 + sig2 * eps2
    mup, sigp = _block(params[8:12], z2)
    logits = _mlp3(params[12:15], z1)
    zero, one = torch.zeros((), dtype=x.dtype), torch.ones((), dtype=x.dtype)
    lpz2 = _normal_lp(z2, zero, one).sum(-1)
    lqz2z1 = _normal_lp(z2, mu2, sig2).sum(-1)
    lpz1z2 = _normal_lp(z1, mup, sigp).sum(-1)
    lqz1x = _normal_lp(z1, mu1.unsqueeze(0), sig1.unsqueeze(0)).sum(-1)
    lpxz1 = _bern_lp(x.unsqueeze(0), logits).sum(-1)
    log_w = lpxz1 + lpz1z2 + lpz2 - lqz1x - lqz2z1
    res = {}
    res["vae_elbo"] = log_w.mean(0).mean(-1)
    res["iwae_elbo"] = _logmeanexp(log_w, 0).mean(-1)
    m = log_w.max(dim=0, keepdim=True).values
    w = torch.exp(log_w - m)
    wn = (w / w.sum(0, keepdim=True)).detach()
    res["iwae_eq14"] = (wn * log_w).sum(0).mean()
    al = torch.softmax(log_w, dim=0)
    res["snis_z1"] = (al.unsqueeze(-1) * z1).sum(0)
    res["snis_z2"] = (al.unsqueeze(-1) * z2).sum(0)
    res.update(z1=z1, z2=z2, al=al, logits=logits, lpxz1=lpxz1, lpz1z2=lpz1z2, lpz2=lpz2,
               lqz1x=lqz1x, lqz2z1=lqz2z1)
    return res


def to_torch_params(params_np, dtype=torch.float64, requires_grad=True):
    out = []
    for W, b in params_np:
        out.append((torch.tensor(W, dtype=dtype, requires_grad=requires_grad),
                    torch.tensor(b, dtype=dtype, requires_grad=requires_grad)))
    return out


def loss_grads(params_np, x, eps, beta=1.0, objective="iwae_elbo", n_layers=1, dtype=torch.float64, y=None):
    """autograd gradient of the reference's loss; eps is eps (1-layer) or (eps1, eps2)."""
    P = to_torch_params(params_np, dtype)
    xt = torch.tensor(x, dtype=dtype)
    flatP = [t for Wb in P for t in Wb]
    if n_layers == 1:
        e = torch.tensor(eps, dtype=dtype)
        if objective == "dreg":
            res = forward_1layer(P, xt, e, beta, dreg=True)
            enc = [t for Wb in P[0:4] for t in Wb]
            dec = [t for Wb in P[4:7] for t in Wb]
            ge = torch.autograd.grad(res["inference_loss"], enc, retain_graph=True)   # task02.py:95
            gd = torch.autograd.grad(-res["iwae_elbo"], dec)                           # task02.py:96
            g = list(ge) + list(gd)
        else:
            res = forward_1layer(P, xt, e, beta, y=None if y is None else torch.tensor(y, dtype=dtype))
            g = torch.autograd.grad(-res[objective], flatP, allow_unused=True)
            g = [gi if gi is not None else torch.zeros_like(p) for gi, p in zip(g, flatP)]
    else:
        e1 = torch.tensor(eps[0], dtype=dtype)
        e2 = torch.tensor(eps[1], dtype=dtype)
        res = forward_2layer(P, xt, e1, e2, beta)
        g = torch.autograd.grad(-res[objective], flatP)
    grads = [(g[2 * i].numpy(), g[2 * i + 1].numpy()) for i in range(len(P))]
    res_np = {k_: v.detach().numpy() for k_, v in res.items()}
    return res_np, grads


class CpuTrainer:
    """fp32 CPU train step (forward, autograd backward, Keras-form Adam eps=1e-4) used as
    bench.py's cpu_baseline ("port"): the same graph the reference runs under tf.function
    (iwae1.py:153-162), eps drawn by torch.randn."""

    def __init__(self, params_np, n_layers=1, lr=1e-3, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        self.n_layers = n_layers
        self.P = to_torch_params(params_np, torch.float32)
        self.flat = [t for Wb in self.P for t in Wb]
        self.m = [torch.zeros_like(t) for t in self.flat]
        self.v = [torch.zeros_like(t) for t in self.flat]
        self.t = 0
        self.lr = lr

    def step(self, x, k, objective="iwae_elbo", beta=1.0):
        B = x.shape[0]
        if self.n_layers == 1:
            D = self.P[2][0].shape[1]
            res = forward_1layer(self.P, x, torch.randn(k, B, D), beta)
        else:
            D1 = self.P[2][0].shape[1]
            D2 = self.P[6][0].shape[1]
            res = forward_2layer(self.P, x, torch.randn(k, B, D1), torch.randn(k, B, D2), beta)
        g = torch.autograd.grad(-res[objective], self.flat)
        self.t += 1
        alpha = self.lr * math.sqrt(1 - 0.999 ** self.t) / (1 - 0.9 ** self.t)
        with torch.no_grad():
            for p, gi, m, v in zip(self.flat, g, self.m, self.v):
                m.mul_(0.9).add_(gi, alpha=0.1)
                v.mul_(0.999).addcmul_(gi, gi, value=0.001)
                p.sub_(alpha * m / (v.sqrt() + 1e-4))
        return float(res[objective].detach())
