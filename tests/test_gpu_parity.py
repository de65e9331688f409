"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs, against the committed golden fixtures, and -- at BASELINE.json's full size --
through size-independent properties.

Tolerances (the GEMM operands are bf16 with fp32 accumulation, BASELINE.json configs[1]):
  * against the oracle run with the SAME bf16 rounding points ("emu"): per-sample log densities
    |d| <= 0.03 nat, scalars |d| <= 0.02 nat, gradients relative L2 error <= 1e-2 per tensor
    (differences are single bf16-ulp flips from fp32 summation order);
  * against the exact float64 oracle: scalars |d| <= 0.15 nat at random init where |log_w| ~ 300-450
    (north_star budget: +-0.1 nat on the trained test LLH ~ -85), gradients relative L2 <= 3e-2.
"""
import os

import numpy as np
import pytest

from oracle import iwae_np as O, philox_np
import make_golden as MG

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

EMU_ROW_ATOL, EMU_SCALAR_ATOL, EMU_GRAD_REL = 0.03, 0.02, 1e-2
O_ID = {"vae_elbo": 0, "iwae_elbo": 1, "iwae_eq14": 2, "vae_elbo_kl": 3, "dreg": 4}
EXACT_SCALAR_ATOL, EXACT_GRAD_REL = 0.15, 3e-2


def _model(n_layers, nh, nl, x_dim=784, options=None):
    """options: kernel-selection switches (iwae_set_option) -- the library does not read the environment."""
    from iwae_amd.native import NativeModel
    return NativeModel(n_layers, nh, nl, x_dim=x_dim, seed=123, options=options)


def _grad_rel_errors(flat, grads):
    out, off = [], 0
    for dW, db in grads:
        for g in (dW, db):
            got = flat[off:off + g.size].reshape(g.shape).astype(np.float64)
            off += g.size
            out.append(np.linalg.norm(got - g) / (np.linalg.norm(g) + 1e-30))
    return out


def _densities_at_device_head(m, P, x, eps, nl):
    """Per-row log p(x|z), log p(z), log q(z|x) of the 1-layer model evaluated by the ORACLE at the DEVICE's own encoder head
    (mu, sigma as float32, iwae_debug_tensor "enc.head") and the given draws: removes the one sensitivity the per-row comparison with
    the pure oracle has -- a bf16 ulp flip of one encoder activation moves an image's mu, hence log p(z) of all its samples -- so the
    usual per-row bound holds for EVERY row (src/iwae1.py:59,105-111)."""
    head = m.debug_tensor("enc.head").astype(np.float64)
    Dp = head.shape[1] // 2
    mu, sig = head[:, :nl], head[:, Dp:Dp + nl]
    z = mu[None] + sig[None] * np.asarray(eps, dtype=np.float64)
    dec = O._MLP3(P[4:7], O.bf16_round)
    lpxz = np.sum(O.bernoulli_log_prob(np.asarray(x, dtype=np.float64)[None], dec.fwd(O.bf16_round(z))), axis=-1)
    lpz = np.sum(O.normal_log_prob(z, 0.0, 1.0), axis=-1)
    lqzx = np.sum(O.normal_log_prob(z, mu[None], sig[None]), axis=-1)
    return {"lpxz": lpxz, "lpz": lpz, "lqzx": lqzx, "mu": mu, "sigma": sig}


def _densities_at_device_heads_2layer(m, eps1, eps2, B, k, nl, P=None, x=None):
    """The 2-layer model's four latent log-densities (src/iwae2.py:118-124) evaluated by the oracle at the DEVICE's own three Gaussian
    heads (float32: "enc.head" on the images, "enc2.head" / "dec2.head" per sample; device rows are image-major, r = b*k + s) and the given
    draws.  A bf16 ulp flip in one hidden activation moves a head, and log p(z1|z2) divides by sigma_p^2: evaluated at the device's heads
    the comparison is free of that and holds per row at float32-level tolerances."""
    def split(name, D):
        h = m.debug_tensor(name).astype(np.float64)
        Dp = h.shape[1] // 2
        return h[:, :D], h[:, Dp:Dp + D]
    km = lambda a: a.reshape(B, k, -1).transpose(1, 0, 2)      # [M, D] image-major -> [k, B, D]
    mu1, sig1 = split("enc.head", nl[0])
    mu2, sig2 = [km(a) for a in split("enc2.head", nl[1])]
    mup, sigp = [km(a) for a in split("dec2.head", nl[0])]
    z1 = mu1[None] + sig1[None] * np.asarray(eps1, dtype=np.float64)
    z2 = mu2 + sig2 * np.asarray(eps2, dtype=np.float64)
    out = {"lpz2": np.sum(O.normal_log_prob(z2, 0.0, 1.0), axis=-1), "lqz2z1": np.sum(O.normal_log_prob(z2, mu2, sig2), axis=-1),
           "lpz1z2": np.sum(O.normal_log_prob(z1, mup, sigp), axis=-1), "lqz1x": np.sum(O.normal_log_prob(z1, mu1[None], sig1[None]), axis=-1)}
    if P is not None:      # log p(x|z1) through the oracle's decoder (the last three layers) at the device's own z1 (src/iwae2.py:96,121)
        dec = O._MLP3(P[-3:], O.bf16_round)
        out["lpxz1"] = np.sum(O.bernoulli_log_prob(np.asarray(x, dtype=np.float64)[None], dec.fwd(O.bf16_round(z1))), axis=-1)
    return out


CASES_1L = [  # (B, k, objective, beta, n_hidden, n_latent, x_dim)
    (4, 3, "iwae_elbo", 1.0, 200, 100, 784),
    (8, 50, "iwae_elbo", 1.0, 200, 100, 784),
    (20, 1, "vae_elbo", 1.0, 200, 100, 784),        # BASELINE configs[0]: the reference's default regime
    (5, 7, "vae_elbo_kl", 0.7, 200, 100, 784),
    (6, 5, "iwae_eq14", 1.0, 200, 100, 784),
    (6, 5, "dreg", 1.0, 200, 100, 784),
    (1, 1, "iwae_elbo", 1.0, 200, 100, 784),        # smallest possible call
    (3, 130, "iwae_elbo", 1.0, 200, 100, 784),      # k > 64: strided wave reduction over k
    (5, 3, "iwae_elbo", 0.7, 16, 4, 48),            # tiny dims (padding paths), task01-like small latent
    (3, 2, "vae_elbo", 1.0, 64, 2, 784),            # 2-D latent of tasks/task01.py
    (5, 4, "iwae_elbo", 1.0, 128, 32, 784),         # hidden 128: the KT=4 instantiations of every MFMA kernel
    (2, 3, "iwae_elbo", 1.0, 256, 128, 784),        # hidden 256 / latent 128: generic (run-time KT) fallbacks, largest dims
    (300, 2, "iwae_elbo", 1.0, 200, 100, 784),      # ragged row count (600 rows: partial 128-row tile)
]


@pytest.mark.parametrize("B,k,obj,beta,nh,nl,xd", CASES_1L)
def test_train_step_1layer_matches_oracle(gpu, B, k, obj, beta, nh, nl, xd):
    x, P, eps = MG.inputs(1, nh, nl, xd, B, k, 100 + B + k)
    res_e, g_e = O.loss_grads_1layer(P, x, eps, beta, obj, rnd=O.bf16_round)
    res_x, g_x = O.loss_grads_1layer(P, x, eps, beta, obj)
    m = _model(1, nh, nl, xd)
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, beta, obj, eps=eps, want=("z", "snis_z", "al", "logits", "lpxz", "lpz", "lqzx"))
    for key in ("lpxz", "lpz", "lqzx"):
        assert np.max(np.abs(r[key] - res_e[key])) < EMU_ROW_ATOL, key
    np.testing.assert_allclose(r["z"], res_e["z"], rtol=0, atol=1e-2)     # a bf16-ulp flip in h1/h2 moves mu by ~1e-3
    np.testing.assert_allclose(r["al"], res_e["al"], atol=2e-2)
    np.testing.assert_allclose(r["al"].sum(0), 1.0, atol=1e-5)
    np.testing.assert_allclose(r["snis_z"], res_e["snis_z"], atol=5e-2)
    assert np.max(np.abs(r["logits"] - res_e["logits"])) < 2e-2
    for key in ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14"):
        assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
        assert abs(r[key] - res_x[key]) < EXACT_SCALAR_ATOL, (key, r[key], res_x[key])
    if obj == "dreg":
        assert abs(r["inference_loss"] - res_e["inference_loss"]) < 5e-3 * abs(res_e["inference_loss"]) + 0.05
    g = m.get_grads()
    assert max(_grad_rel_errors(g, g_e)) < EMU_GRAD_REL
    assert max(_grad_rel_errors(g, g_x)) < EXACT_GRAD_REL
    # Keras Adam, eps = 1e-4 (main.py:93): one step from the device gradient
    m.adam_step(1e-3)
    ref, _, _ = O.adam_update(O.flatten_params(P), g.astype(np.float64), 0.0, 0.0, 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6
    m.close()


CASES_2L = [(4, 3, "iwae_elbo", [200, 100], [100, 50], 784), (6, 50, "vae_elbo", [200, 100], [100, 50], 784),
            (3, 5, "iwae_eq14", [64, 32], [4, 2], 784), (4, 3, "iwae_elbo", [16, 8], [4, 2], 48)]


@pytest.mark.parametrize("B,k,obj,nh,nl,xd", CASES_2L)
def test_train_step_2layer_matches_oracle(gpu, B, k, obj, nh, nl, xd):
    x, P, eps = MG.inputs(2, nh, nl, xd, B, k, 200 + B + k)
    res_e, g_e = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj, rnd=O.bf16_round)
    res_x, g_x = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj)
    m = _model(2, nh, nl, xd)
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, 1.0, obj, eps=eps, want=("z", "z2", "al", "lpxz", "lpz", "lqzx", "lpz2", "lqzx2", "snis_z", "snis_z2"))
    # per-row densities: against the oracle evaluated at the DEVICE's own Gaussian heads (float32) -- every row within 0.02 nat
    # (round 5: was a 0.4-nat window on log p(z1|z2), which divides by sigma_p^2 of a head fed by bf16 activations; at the device's
    # heads that sensitivity is gone, as in the headline-size test) -- and against the pure rounding-aware oracle within the bf16-flip bound
    at = _densities_at_device_heads_2layer(m, eps[0], eps[1], B, k, nl, P, x)
    for a, b in (("lpxz", "lpxz1"), ("lpz", "lpz1z2"), ("lpz2", "lpz2"), ("lqzx", "lqz1x"), ("lqzx2", "lqz2z1")):
        d_at = float(np.max(np.abs(r[a] - at[b])))
        assert d_at < (EMU_ROW_ATOL if b == "lpxz1" else 2e-2), (b, d_at)
        d_e = float(np.max(np.abs(r[a] - res_e[b])))
        assert d_e < (0.4 if b == "lpz1z2" else 0.05), (b, d_e)      # (the pure oracle: loose, explained by the bound above)
    np.testing.assert_allclose(r["z"], res_e["z1"], rtol=0, atol=1e-2)
    for key in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
        assert abs(r[key] - res_e[key]) < 0.05, (key, r[key], res_e[key])
        assert abs(r[key] - res_x[key]) < 0.3, (key, r[key], res_x[key])
    # the objective values recomputed by the ORACLE's own reductions from the device's per-row terms: the device's log-mean-exp / means
    # are held to 1e-3 nat (the 0.05 / 0.3 above are the bf16 operands' effect on the terms, not slack in the reduction)
    lw = (r["lpxz"] + r["lpz"] + r["lpz2"] - r["lqzx"] - r["lqzx2"]).astype(np.float64)
    assert abs(r["iwae_elbo"] - float(np.mean(O.logmeanexp(lw, axis=0)))) < 1e-3
    assert abs(r["vae_elbo"] - float(np.mean(lw))) < 1e-3
    g = m.get_grads()
    assert max(_grad_rel_errors(g, g_e)) < 2e-2
    assert max(_grad_rel_errors(g, g_x)) < 5e-2
    m.close()


@pytest.mark.parametrize("obj", ["iwae_elbo", "dreg"])
def test_large_row_count_kernels_match_oracle(gpu, obj):
    """8 500 data rows: the row-count-dependent kernel choices of the full-size step (8-wave x 16-row dense shape,
    s = x - sigmoid(l) kept by the forward pass + out_bwd_s_kernel, 16-wave row-weighted weight gradient, one grouped
    launch for the encoder's weight gradients) against the oracle, plus the fused Adam of iwae_train_step.
    obj = "dreg" (tasks/task02.py:61-101, BASELINE configs[3]): at this size DReG takes its own kernel chain -- the separate
    sampling kernel with the stop-gradient density lq_dreg, the decoder kernel reading the stored z rows, the row-weighted
    output-layer weight gradient -- which the small-row DReG cases do not reach."""
    B, k, nh, nl, xd = 170, 50, 200, 100, 784
    x, P, eps = MG.inputs(1, nh, nl, xd, B, k, 4242)
    res_e, g_e = O.loss_grads_1layer(P, x, eps, 1.0, obj, rnd=O.bf16_round)
    res_x, g_x = O.loss_grads_1layer(P, x, eps, 1.0, obj)
    m = _model(1, nh, nl, xd)
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, 1.0, obj, eps=eps, want=("lpxz", "lqzx", "lpz"))
    for key in ("lpxz", "lqzx", "lpz"):
        assert np.max(np.abs(r[key] - res_e[key])) < EMU_ROW_ATOL, key
    for key in ("iwae_elbo",) if obj == "dreg" else ("vae_elbo", "iwae_elbo", "iwae_eq14"):
        assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    if obj == "dreg":
        assert abs(r["inference_loss"] - res_e["inference_loss"]) < 5e-3 * abs(res_e["inference_loss"]) + 0.05
    g = m.get_grads()
    errs_e, errs_x = _grad_rel_errors(g, g_e), _grad_rel_errors(g, g_x)
    assert max(errs_e) < EMU_GRAD_REL, errs_e
    assert max(errs_x) < EXACT_GRAD_REL, errs_x
    # elementwise, per tensor, against the rounding-aware oracle: |d| <= 3 % of the tensor's largest element
    off = 0
    for dW, db in g_e:
        for t in (dW, db):
            got = g[off:off + t.size].reshape(t.shape).astype(np.float64)
            off += t.size
            assert np.max(np.abs(got - t)) <= 3e-2 * np.max(np.abs(t)) + 1e-9
    # the same step through iwae_train_step (Adam fused into the slab reduction) lands on the same parameters
    r2 = m.train_step(x, k, 1.0, 1e-3, obj, eps=eps)
    assert abs(r2["iwae_elbo"] - r["iwae_elbo"]) < 1e-5
    np.testing.assert_array_equal(m.get_grads(), g)
    ref, _, _ = O.adam_update(O.flatten_params(P), g.astype(np.float64), 0.0, 0.0, 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6
    m.close()


@pytest.mark.parametrize("force_qw", [False, True])
@pytest.mark.parametrize("B,k,xd", [(170, 50, 100), (260, 33, 1000), (2, 5000, 784), (175, 48, 48)])
def test_pipelined_bernoulli_forward_shapes(gpu, B, k, xd, force_qw):
    """bern_pipe_kernel (>= 8 192 rows, hidden 200, k >= 32) beyond the reference's 784 pixels and k = 50: pixel counts whose
    last 32-pixel half is partial (100 = 3 x 32 + 4) or that fill an even / minimal number of halves (1000 -> 32, 48 -> 2),
    blocks of 128 rows that straddle images at other k, a ragged last block, and k = 5000 (the test-LLH evaluator's regime:
    all rows of a block belong to one image).  Both instantiations: forward only (no s kept) and training step (s kept,
    checked through the gradient it feeds), per-row log p(x|z) against the oracle with the same bf16 rounding points."""
    nh, nl = 200, 20
    x, P, eps = MG.inputs(1, nh, nl, xd, B, k, 77 + xd)
    res_e, g_e = O.loss_grads_1layer(P, x, eps, 1.0, "iwae_elbo", rnd=O.bf16_round)
    # force_qw: the 16-wave / 200-row workgroup shape (12 full-tile waves + four waves sharing the 13th tile), which the
    # library otherwise takes only where its workgroups fill the machine's CUs evenly (the full-size benchmark shape)
    m = _model(1, nh, nl, xd, options={"bern_qw_force": 1} if force_qw else None)
    m.set_params(O.flatten_params(P))
    r0 = m.forward(x, k, 1.0, eps=eps, want=("lpxz",))
    assert np.max(np.abs(r0["lpxz"] - res_e["lpxz"])) < EMU_ROW_ATOL
    r = m.forward_backward(x, k, 1.0, "iwae_elbo", eps=eps, want=("lpxz",))
    np.testing.assert_allclose(r["lpxz"], r0["lpxz"], rtol=2e-6, atol=1e-4)   # the s-keeping instantiation: same sums, prod*(1+e) as mul vs fma
    assert abs(r["iwae_elbo"] - res_e["iwae_elbo"]) < EMU_SCALAR_ATOL
    assert max(_grad_rel_errors(m.get_grads(), g_e)) < EMU_GRAD_REL
    m.close()


@pytest.mark.parametrize("opts", [{}, {"out_recompute": 1}])
def test_output_layer_remainder_strip_matches_oracle(gpu, opts):
    """x_dim = 300 at 8 500 rows: the output layer is 320 padded columns = one 256-wide column block + a remainder of 64, which
    wgradws_kernel gives the narrow 64-wide strip shape (BJ = 1).  Default: the row-weighted instantiation (G = the stored s); with
    out_recompute the forward keeps no s and the gradient runs UNWEIGHTED on dl -- the unweighted strip instantiation, which no test at the
    reference's 784 pixels reaches (round-4 advisor finding).  Every gradient tensor against the oracle with the same rounding points."""
    B, k, nh, nl, xd = 170, 50, 200, 20, 300
    x, P, eps = MG.inputs(1, nh, nl, xd, B, k, 991)
    res_e, g_e = O.loss_grads_1layer(P, x, eps, 1.0, "iwae_elbo", rnd=O.bf16_round)
    m = _model(1, nh, nl, xd, options=opts or None)
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, 1.0, "iwae_elbo", eps=eps, want=("lpxz",))
    assert np.max(np.abs(r["lpxz"] - res_e["lpxz"])) < EMU_ROW_ATOL
    g = m.get_grads()
    errs = _grad_rel_errors(g, g_e)
    assert max(errs) < EMU_GRAD_REL, errs
    off = 0
    for dW, db in g_e:      # elementwise: a mis-addressed strip would show as a block of wrong columns, not in a norm
        for t in (dW, db):
            got = g[off:off + t.size].reshape(t.shape).astype(np.float64)
            off += t.size
            assert np.max(np.abs(got - t)) <= 3e-2 * np.max(np.abs(t)) + 1e-9
    m.close()


def test_row_weight_pads_stay_finite_after_a_float32_evaluation(gpu):
    """Round-4 advisor finding: the row-weighted job of wgrad_rows_kernel (the decoder's output-layer gradient on <= 2 048 data rows) reads
    the row weights gx in whole 32-row stages, i.e. up to 31 rows past M; the G rows there are zero, but 0 x a non-finite pad is NaN.  A
    float32 evaluation sizes gx FIRST here (3 images x 40 samples), then a bf16 step with M = 100 (not a multiple of 32) fits inside that
    capacity: its gradient must be finite and bitwise the gradient of a fresh handle that never ran the float32 path."""
    B, k = 20, 5
    x = O.synthetic_binarized(B, 5)
    P = O.init_params(1, 200, 100, 9, x_mean=O.synthetic_pixel_means())
    outs = []
    for warm in (True, False):
        m = _model(1, 200, 100)
        m.set_params(O.flatten_params(P))
        if warm:
            m.eval_llh(x[:3], k=40)           # float32 by default (iwae_set_eval_precision)
        m.set_step(4, 0)
        m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
        outs.append((m.get_grads().copy(), m.get_params().copy()))
        m.close()
    assert np.all(np.isfinite(outs[0][0])) and np.all(np.isfinite(outs[0][1]))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,k,obj", [(170, 50, "iwae_elbo"), (172, 48, "vae_elbo")])
def test_two_layer_large_row_count_matches_oracle(gpu, B, k, obj):
    """The 2-layer model (iwae2.py) at >= 8 192 rows: the per-sample blocks on M rows take the large-row dense shapes and
    the z1 -> x decoder the one-launch decoder kernel (z1 read from the sampling kernel's rows instead of being made in
    the kernel); gradients of all 26 tensors against the oracle with the same bf16 rounding points."""
    nh, nl, xd = [200, 100], [100, 50], 784
    x, P, eps = MG.inputs(2, nh, nl, xd, B, k, 4321)
    res_e, g_e = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj, rnd=O.bf16_round)
    m = _model(2, nh, nl, xd)
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, 1.0, obj, eps=eps)
    assert abs(r[obj] - res_e[obj]) < EMU_SCALAR_ATOL, (r[obj], res_e[obj])
    assert max(_grad_rel_errors(m.get_grads(), g_e)) < EMU_GRAD_REL
    m.close()


@pytest.mark.parametrize("B,k", [(170, 50), (1650, 5)])
def test_two_layer_kernel_variants_agree(gpu, B, k):
    """2-layer model at >= 8 192 rows: the per-sample blocks fused into one launch each way (chain2_fwd_kernel / chain2_bwd_kernel: every
    layer of q(z2|z1) and p(z1|z2), the z2 sampling, the three log-densities and their gradients without a round trip through HBM)
    against the same mathematics as separate launches (dense_kernel x 6, sample_kernel, gauss_lp_kernel, gauss_bwd_kernel x 2,
    dense_kernel<EPI_DX> x 6): values, per-row densities and all 26 gradient tensors, on host noise and on the device's own."""
    nh, nl = [200, 100], [100, 50]
    x, P, eps = MG.inputs(2, nh, nl, 784, B, k, 5150 + B)

    def run(opts, e):
        m = _model(2, nh, nl, options=opts)
        m.set_params(O.flatten_params(P))
        m.set_step(9, 2)
        r = m.forward_backward(x, k, 1.0, "iwae_elbo", eps=e, want=("lpz", "lpz2", "lqzx", "lqzx2", "z2"))
        # round 4: every variant's per-row densities against the oracle evaluated at THAT run's own Gaussian heads -- no window for outliers
        e1, e2 = e if e is not None else (philox_np.device_eps(123, 9, B, k, nl[0], stream=0, batch_offset=2),
                                          philox_np.device_eps(123, 9, B, k, nl[1], stream=1, batch_offset=2))
        at = _densities_at_device_heads_2layer(m, e1, e2, B, k, nl)
        for a, b in (("lpz", "lpz1z2"), ("lpz2", "lpz2"), ("lqzx", "lqz1x"), ("lqzx2", "lqz2z1")):
            err = np.abs(r[a] - at[b])
            assert err.max() < (5e-3 if e is not None else 2e-2), (opts, b, err.max())      # (device noise: the host restates the float32 draws in float64, d eps ~ 1e-7, divided by sigma_p)
        g = m.get_grads().astype(np.float64)
        m.close()
        return r, g

    for e, ref_opts in ((eps, {"no_chain2": 1}), (None, {"no_chain2": 1}), (eps, {"no_chain2_bwd": 1})):
        r0, g0 = run(ref_opts, e)
        r1, g1 = run({}, e)
        # the fused kernels start their accumulators from the bias, the separate launches add it at the end: float32 sums in another
        # order, so a bf16 activation flips by an ulp here and there and moves the densities of THAT row (log p(z1|z2) of a row with a small
        # sigma_p by several 0.1 nat at random init, |lpz| ~ 200-400) -- the typical row agrees to float32 rounding; the rows that do not are
        # accounted for inside run(): each variant matches the oracle at its own heads on every row
        for key in ("lpz", "lpz2", "lqzx2"):
            err = np.abs(r1[key] - r0[key])
            assert np.quantile(err, 0.9) < 2e-3 and err.max() < 1.0, (key, np.quantile(err, 0.9), err.max())
        errz = np.abs(r1["z2"] - r0["z2"])
        assert np.quantile(errz, 0.98) < 2e-3 and errz.max() < 0.05
        assert abs(r1["iwae_elbo"] - r0["iwae_elbo"]) < 2e-2
        assert np.linalg.norm(g1 - g0) / np.linalg.norm(g0) < 5e-3


@pytest.mark.parametrize("layers,B,k,obj", [(1, 120, 50, "iwae_elbo"), (1, 120, 50, "dreg"), (1, 90, 50, "iwae_eq14"), (1, 2000, 5, "iwae_elbo"),
                                            (1, 2000, 5, "dreg"), (1, 1650, 5, "vae_elbo"), (2, 120, 50, "iwae_elbo"), (2, 2000, 5, "iwae_elbo")])
def test_kernel_family_boundaries_match_oracle(gpu, layers, B, k, obj):
    """Row counts and sample counts at which the host picks ANOTHER kernel family than the cases above: 4 097 - 8 191 rows
    (6 000 = 120 x 50 and 4 500 = 90 x 50: beyond block_fwd_kernel's 4 096 rows, below the 8 192 of the one-launch decoder kernel:
    sample_kernel + dense_kernel launches, per-pixel-group Bernoulli sums, dec_bwd_kernel in its small-row use), and >= 8 192
    rows with FEW samples per image (B = 2 000 / 1 650, k = 5 -- the reference's default --n_samples, main.py:18 -- where a 128-row
    block spans ~26 images: the pipelined Bernoulli kernel needs k >= ~32, so dense_kernel<EPI_BERN> runs at a large row count,
    with the large-row weight gradients behind it).  1-layer iwae_elbo / iwae_eq14 / vae_elbo / DReG and the 2-layer model, every
    gradient tensor and the per-row log-densities against the oracle with the same bf16 rounding points."""
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x, P, eps = MG.inputs(layers, nh, nl, 784, B, k, 9000 + B + k)
    m = _model(layers, nh, nl)
    m.set_params(O.flatten_params(P))
    if layers == 1:
        res_e, g_e = O.loss_grads_1layer(P, x, eps, 1.0, obj, rnd=O.bf16_round)
        r = m.forward_backward(x, k, 1.0, obj, eps=eps, want=("lpxz", "lqzx", "lpz"))
        # Per-row densities with up to 2 000 images: a bf16 ulp flip of ONE hidden activation of the encoder (fp32 summation order
        # of the device vs the oracle's float64) moves an image's mu by ~1e-3 and with it log p(z) = -1/2 sum z^2 of all its samples
        # by a few 1e-2 (round 3 widened the bound to a 10 x window for that).  Round 4: the encoder head itself is held to the oracle's
        # (|d mu|, |d sigma| small: a flipped activation times one weight), and the densities are compared with the oracle evaluated AT the
        # device's own head, where the usual bound holds for EVERY row -- an error in the sampling / density / decoder kernels cannot hide.
        at = _densities_at_device_head(m, P, x, eps, nl)
        enc = O._Block(P[0:4], O.bf16_round)
        mu_o, sig_o = enc.fwd(O.bf16_round(np.asarray(x, dtype=np.float64)))
        assert np.max(np.abs(at["mu"] - mu_o)) < 1e-2 and np.max(np.abs(at["sigma"] / sig_o - 1.0)) < 1e-2
        for key in ("lpxz", "lqzx", "lpz"):
            err = np.abs(r[key] - at[key])
            assert err.max() < EMU_ROW_ATOL, (key, err.max())
            err = np.abs(r[key] - res_e[key])       # and against the pure oracle: the typical row
            assert np.quantile(err, 0.98) < EMU_ROW_ATOL, (key, np.quantile(err, 0.98), err.max())
        if obj == "dreg":
            assert abs(r["inference_loss"] - res_e["inference_loss"]) < 5e-3 * abs(res_e["inference_loss"]) + 0.05
            assert abs(r["iwae_elbo"] - res_e["iwae_elbo"]) < EMU_SCALAR_ATOL
        else:
            for key in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
                assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    else:
        res_e, g_e = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj, rnd=O.bf16_round)
        r = m.forward_backward(x, k, 1.0, obj, eps=eps, want=("lpxz",))
        err = np.abs(r["lpxz"] - res_e["lpxz1"])
        assert np.quantile(err, 0.98) < EMU_ROW_ATOL and err.max() < 10 * EMU_ROW_ATOL
        for key in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
            assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    g = m.get_grads()
    errs = _grad_rel_errors(g, g_e)
    assert max(errs) < EMU_GRAD_REL, errs
    # the same step on the device's own noise agrees with the oracle on the same Philox draws (the decoder kernel / block kernels
    # make z themselves on that path)
    m.set_step(17, 3)
    r2 = m.forward_backward(x, k, 1.0, obj)
    if layers == 1:
        e_dev = philox_np.device_eps(123, 17, B, k, nl, batch_offset=3)
        res_d, g_d = O.loss_grads_1layer(P, x, e_dev, 1.0, obj, rnd=O.bf16_round)
    else:
        e1 = philox_np.device_eps(123, 17, B, k, nl[0], stream=0, batch_offset=3)
        e2 = philox_np.device_eps(123, 17, B, k, nl[1], stream=1, batch_offset=3)
        res_d, g_d = O.loss_grads_2layer(P, x, e1, e2, 1.0, obj, rnd=O.bf16_round)
    assert abs(r2["iwae_elbo"] - res_d["iwae_elbo"]) < EMU_SCALAR_ATOL
    assert max(_grad_rel_errors(m.get_grads(), g_d)) < EMU_GRAD_REL
    m.close()


@pytest.mark.parametrize("B,k,nh,nl,obj", [(170, 50, 200, 100, "iwae_elbo"), (20, 3, 200, 100, "iwae_elbo"), (9, 4, 64, 10, "iwae_elbo"), (6, 5, 64, 2, "iwae_elbo"),
                                           (170, 50, 200, 100, "dreg"), (20, 3, 200, 100, "dreg"),
                                           # round 5 (the decoder kernel's rewritten sampling prologue at 8 500 rows): a latent of 2 k-steps whose width is not
                                           # a multiple of 4 (50: the masked chunk straddles D), one that fills its k-steps exactly (64), a ragged 4-step one (98),
                                           # and rows that leave the last workgroup's shared tile half empty (163 x 52 = 8 476)
                                           (170, 50, 200, 50, "iwae_elbo"), (170, 50, 200, 64, "dreg"), (170, 50, 200, 98, "iwae_elbo"), (163, 52, 200, 100, "iwae_elbo")])
def test_device_noise_step_matches_oracle_on_the_same_draws(gpu, B, k, nh, nl, obj):
    """The training step on the DEVICE's own noise (the path bench.py times: noise drawn ahead by eps_gen_kernel, the first
    decoder layer making z = mu + sigma*eps itself; on few rows block_fwd_kernel's sampling mode) against the oracle fed the same
    draws, restated on the host from the published Philox4x32-10 + Box-Muller (oracle/philox_np.py).  The host draws are float64,
    the device's float32: the tolerances are the bf16-emulation ones.  Latent widths 10 and 2: cached draws in rows of 12 / 4 floats."""
    step = 9
    x = O.synthetic_binarized(B, 23)
    P = O.init_params(1, nh, nl, 29, x_mean=O.synthetic_pixel_means())
    eps = philox_np.device_eps(123, step, B, k, nl)                      # [k, B, D], seed 123 = _model's
    res_e, g_e = O.loss_grads_1layer(P, x, eps, 1.0, obj, rnd=O.bf16_round)
    m = _model(1, nh, nl)
    m.set_params(O.flatten_params(P))
    m.set_step(step, 0)
    r = m.forward_backward(x, k, 1.0, obj, want=("lpxz", "lpz", "lqzx", "z"))
    np.testing.assert_allclose(r["z"], res_e["z"], rtol=0, atol=1e-2)
    # every row against the oracle evaluated at the device's own encoder head (a bf16 ulp flip of one encoder activation moves an image's mu, hence
    # log p(z) of all its samples: nl = 50 has such a row at 0.039 nat against the pure oracle), and the typical row against the pure oracle
    at = _densities_at_device_head(m, P, x, eps, nl)
    for key in ("lpxz", "lpz", "lqzx"):
        assert np.max(np.abs(r[key] - at[key])) < EMU_ROW_ATOL, key
        assert np.quantile(np.abs(r[key] - res_e[key]), 0.98) < EMU_ROW_ATOL, key
        assert np.max(np.abs(r[key] - res_e[key])) < 10 * EMU_ROW_ATOL, key
    for key in (("iwae_elbo",) if obj == "dreg" else ("vae_elbo", "iwae_elbo", "iwae_eq14")):
        assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    if obj == "dreg":      # (tasks/task02.py:61-76; at >= 8 192 rows the second log q is summed in the decoder kernel's prologue)
        assert abs(r["inference_loss"] - res_e["inference_loss"]) < 5e-3 * abs(res_e["inference_loss"]) + 0.05
    assert max(_grad_rel_errors(m.get_grads(), g_e)) < EMU_GRAD_REL
    m.close()


@pytest.mark.parametrize("B,k", [(170, 50), (24, 5)])
def test_kernel_variants_agree(gpu, B, k):
    """The tuning switches select different kernels for the same mathematics: recomputing the logits in out_bwd
    instead of reading the stored s, the 4-wave x 32-row dense shape instead of 8 x 16, the separate sampling kernel
    instead of the first decoder layer making z itself, the Bernoulli forward on dense_kernel<EPI_BERN> instead of the
    software-pipelined bern_pipe_kernel, the decoder's tanh layers / the encoder block as separate launches instead of fused ones;
    round 2: the decoder's dX chain as three launches instead of dec_bwd_kernel, the general weight-gradient kernel instead of the
    specialised-wave one and that one's 8 + 8-wave shape, one side stream instead of two, the grouped launch of the hidden layers'
    gradients, one lse_kernel instead of the side stream's own copy; the per-pixel-group dX kernels at small row counts; the decoder's
    dX chain by dec_bwd_kernel at few rows and by dec_bwd_rows_kernel at many; the few-row decoder's output layer as its own launch."""
    x = O.synthetic_binarized(B, 3)
    P = O.init_params(1, 200, 100, 7, x_mean=O.synthetic_pixel_means())

    def run(env):
        m = _model(1, 200, 100, options=env)
        m.set_params(O.flatten_params(P))
        m.set_step(5, 0)
        r = m.forward_backward(x, k, 1.0, "iwae_elbo")
        g = m.get_grads().astype(np.float64)
        m.close()
        return r["iwae_elbo"], g

    e0, g0 = run({})
    for env in ({"out_recompute": 1}, {"dense_g1": 0}, {"no_zin": 1}, {"no_bern_pipe": 1},
                {"no_dec_fused": 1}, {"no_block_fused": 1}, {"no_early_wout": 1}, {"bern_qw_force": 1},
                {"no_dec_bwd": 1}, {"no_wg7": 1}, {"wg9": 3}, {"no_side2": 1}, {"wg_group": 1}, {"no_lse_dup": 1}, {"no_lse_fused": 1}, {"no_wg3": 1}, {"dz_f32": 1}, {"no_small_dec_bwd": 1},
                {"dec_rows": 0}, {"dec_rows": 16384}, {"no_out_in_block": 1}, {"no_wgrad_rows": 1}, {"no_dec_rows": 1}, {"no_lse_in_bwd": 1}, {"no_lat_in_block": 1}, {"lat_rows4": 1}, {"g2w": 1}, {"dec_bwd_nw": 4}, {"defer_split": 1}, {"wout_split": 35}):
        e1, g1 = run(env)
        assert abs(e1 - e0) < 2e-3, env
        assert np.linalg.norm(g1 - g0) / np.linalg.norm(g0) < 5e-3, env


@pytest.mark.parametrize("layers,B,k,obj", [(1, 170, 50, "iwae_elbo"), (1, 170, 50, "dreg"), (1, 170, 50, "vae_elbo_kl"), (1, 340, 25, "iwae_eq14"),
                                            (1, 425, 20, "vae_elbo"), (1, 90, 100, "iwae_elbo"), (1, 45, 200, "dreg"), (2, 170, 50, "iwae_elbo")])
def test_decoder_kernel_does_the_log_mean_exp_bitwise(gpu, layers, B, k, obj):
    """Round 3: where the one-launch decoder kernel's workgroups own whole images (its 16-wave / 200-row shape, k a divisor of 200) it
    also does lse_kernel's work for them (iwae1.py:113-139: log_w, logmeanexp, softmax weights, the per-image objective values) -- the
    same device function on the same numbers, so everything downstream must be BITWISE what the separate launch gives: scalars, log_w,
    every gradient.  k <= 64 (one sample per lane) and k > 64, a last workgroup holding fewer images than the others (8 500 rows =
    42.5 x 200), host noise and the device's own (then the kernel's prior / posterior terms come from its LDS copies), DReG's second
    log q, the 2-layer model (terms of the chain kernel, read from memory); and against the oracle for the iwae_elbo cases."""
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x, P, eps = MG.inputs(layers, nh, nl, 784, B, k, 4400 + B + k)

    def run(opts, e):
        m = _model(layers, nh, nl, options=dict(opts, bern_qw_force=1))
        m.set_params(O.flatten_params(P))
        m.set_step(3, 1)
        r = m.forward_backward(x, k, 0.7 if obj == "vae_elbo_kl" else 1.0, obj, eps=e, want=("log_w",))
        g = m.get_grads()
        m.close()
        return r, g

    def run_fwd(opts):      # val_step: forward only (z from the sampling kernel, the kernel's terms read from memory; DReG's value reported too)
        m = _model(layers, nh, nl, options=dict(opts, bern_qw_force=1))
        m.set_params(O.flatten_params(P))
        m.set_step(3, 1)
        r = m.forward(x, k, 1.0, eps=eps)
        m.close()
        return r

    f0, f1 = run_fwd({"no_lse_fused": 1}), run_fwd({})
    for key in f0:
        if np.isscalar(f0[key]) or getattr(f0[key], "ndim", 1) == 0:
            assert f1[key] == f0[key] or (np.isnan(f1[key]) and np.isnan(f0[key])), ("forward", key, f1[key], f0[key])
    for e in (eps, None):
        r0, g0 = run({"no_lse_fused": 1}, e)
        r1, g1 = run({}, e)
        np.testing.assert_array_equal(r1["log_w"], r0["log_w"])
        for key in r0:
            if np.isscalar(r0[key]) or getattr(r0[key], "ndim", 1) == 0:
                assert r1[key] == r0[key] or (np.isnan(r1[key]) and np.isnan(r0[key])), (key, r1[key], r0[key])
        np.testing.assert_array_equal(g1, g0)
    if obj == "iwae_elbo":
        res_e, g_e = (O.loss_grads_1layer(P, x, eps, 1.0, obj, rnd=O.bf16_round) if layers == 1 else
                      O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj, rnd=O.bf16_round))
        r1, g1 = run({}, eps)
        assert abs(r1["iwae_elbo"] - res_e["iwae_elbo"]) < EMU_SCALAR_ATOL
        assert max(_grad_rel_errors(g1, g_e)) < EMU_GRAD_REL


@pytest.mark.parametrize("B,k,layers", [(170, 50, 1), (20, 5, 1), (24, 6, 2), (1024, 50, 1), (1024, 50, 2)])
def test_in_library_data_parallel_step_world1_is_bitwise_the_single_gpu_step(gpu, B, k, layers):
    """iwae_comm_init + iwae_train_step (the data-parallel step inside the library: ncclAllReduce on the library's streams between
    gradient and Adam, the decoder segment's exchange + update deferred on the side stream) rehearsed with ONE rank -- all this
    box has -- against the single-GPU step: 12 steps on device noise must land on bit-identical parameters and Adam state
    (grad_scale 1/1, a one-rank all-reduce is the identity), with parameter reads in between seeing completed updates.
    (1024, 50): the per-rank workload of BASELINE configs[4] (1-layer, 8 x 1 024 images) and of its 2-layer counterpart -- the kernels, streams and
    deferred updates the N = 8 run will take on every rank; what one GPU can exercise of it (round-4 verdict)."""
    from iwae_amd.native import NativeModel
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x = O.synthetic_binarized(B, 13)
    P = O.init_params(layers, nh, nl, 21, x_mean=O.synthetic_pixel_means())
    outs = []
    for dp in (False, True):
        m = _model(layers, nh, nl)
        m.set_params(O.flatten_params(P))
        if dp:
            blob = NativeModel.comm_unique_id()
            m.comm_preflight(blob, 1, 0)                                 # the non-collective checks pass ...
            with pytest.raises(ValueError):
                m.comm_preflight(blob, 2, 1)                             # ... and refuse what comm_init would refuse, without entering the rendezvous
            m.comm_init(blob, 1, 0)
            with pytest.raises(RuntimeError):
                m.comm_preflight(NativeModel.comm_unique_id(), 1, 0)     # communicators already there
            with pytest.raises(RuntimeError):
                m.comm_init(NativeModel.comm_unique_id(), 1, 0)          # already initialised: call order error, not a second communicator
        for t in range(12):
            m.set_step(t, 0)
            r = m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=(t % 4 == 0))
            if t == 5:
                m.get_params()
        outs.append((m.get_params().copy(), m.get_adam_state(), m.get_grads().copy()))
        if dp:
            m.comm_destroy()
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo")                    # back on the single-GPU step
        m.close()
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1][0], outs[1][1][0])
    np.testing.assert_array_equal(outs[0][1][1], outs[1][1][1])
    assert outs[0][1][2] == outs[1][1][2] == 12
    with pytest.raises(ValueError):
        m2 = _model(1, 200, 100)
        try:
            m2.comm_init(NativeModel.comm_unique_id(), 2, 1)              # does not match the handle's iwae_config (world_size 1)
        finally:
            m2.close()


@pytest.mark.parametrize("B,k", [(170, 50), (20, 5)])
def test_split_backward_equals_joined_backward(gpu, B, k):
    """iwae_forward_backward_split (the data-parallel step's entry: decoder gradient completed on the side stream, unjoined,
    so its all-reduce overlaps the encoder's backward pass) leaves exactly the gradient of iwae_forward_backward, reports the
    decoder's first tensor as the start of the side segment, and the following Adam step lands on the same parameters."""
    x = O.synthetic_binarized(B, 31)
    P = O.init_params(1, 200, 100, 37, x_mean=O.synthetic_pixel_means())
    outs = []
    g_n = O.flatten_params(P).size
    for split in (False, True):
        m = _model(1, 200, 100)
        m.set_params(O.flatten_params(P))
        m.set_step(3, 0)
        if split:
            side, off = m.forward_backward_split_devptr(x.ctypes.data, B, k, 1.0, 1)
            names = [t[0] for t in m.tensor_table()]
            first_dec = [t for t in m.tensor_table() if t[0].startswith("dec")][0]
            # (<= 2 048 data rows, round 4: the decoder's weight gradients ride in the encoder's launch on the main stream -- nothing is left
            # on the side stream, which the ABI reports as an offset of n, include/iwae_amd.h)
            assert off == (first_dec[2] if B * k > 2048 else g_n), (off, names)
            assert side != 0
        else:
            m.forward_backward(x, k, 1.0, "iwae_elbo")
        g = m.get_grads().copy()
        m.adam_step(1e-3, 0.5)
        outs.append((g, m.get_params().copy()))
        m.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("B,k", [(96, 20), (170, 50)])
def test_deferred_decoder_update_is_bitwise_equivalent(gpu, B, k):
    """iwae_train_step leaves the decoder's slab reduction + Adam on the side stream and joins it lazily (before the next
    sampling kernel / any parameter access): scheduling only -- 25 steps with device noise must land on exactly the
    parameters of the run that joins at the end of every step, and reads in between must see completed updates.
    (170, 50) = 8 500 rows takes the full-size step's kernels and streams (the decoder's weight gradients on two side streams);
    round 5: option defer_split = one deferred update per side stream, each behind the gradients it carried."""
    x = O.synthetic_binarized(B, 11)
    P = O.init_params(1, 200, 100, 3, x_mean=O.synthetic_pixel_means())

    def run(env, poke):
        m = _model(1, 200, 100, options=env)
        m.set_params(O.flatten_params(P))
        for t in range(25):
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
            if poke and t % 7 == 3:
                m.get_grads()                       # entry points in between must join the pending update first
        out = m.get_params().copy(), m.get_adam_state()
        m.close()
        return out

    p0, (m0, v0, t0) = run({"no_defer": 1}, False)
    p1, (m1, v1, t1) = run({}, False)
    p2, _ = run({}, True)
    p3, (m3, v3, _) = run({"defer_split": 1}, True)
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(p0, p2)
    np.testing.assert_array_equal(p0, p3)
    np.testing.assert_array_equal(m0, m1)
    np.testing.assert_array_equal(v0, v1)
    np.testing.assert_array_equal(m0, m3)
    np.testing.assert_array_equal(v0, v3)
    assert t0 == t1 == 25


@pytest.mark.parametrize("nh,nl,opts", [(256, 128, {}), (16, 4, {}), (200, 100, {"no_early_wout": 1}), (200, 100, {"out_recompute": 1}),
                                        (96, 20, {}), (200, 100, {"no_wg3": 1}), (200, 100, {}), (200, 100, {"no_dec_rows": 1}),
                                        (256, 128, {"no_dec_rows": 1}), (200, 100, {"no_dec_rows": 1, "out_recompute": 1}),
                                        (200, 100, {"no_lse_in_bwd": 1}), (200, 100, {"no_lat_in_block": 1})])
def test_speculative_noise_draw_is_ordered_on_every_kernel_path(gpu, nh, nl, opts):
    """The next step's noise is drawn a step ahead on a side stream (forward_impl).  Which side stream must follow what the backward
    pass of THAT step uses: hidden widths without a stored-s instantiation (256, 16, 96) and the options out_recompute / no_early_wout
    never touch the second side stream (round-3 advisor finding: the draw went there unordered).  Scheduling only: 12 steps at B = 20,
    k = 5 enqueued back to back land bitwise on the parameters of (a) the same steps with the host waiting for the device after each
    one and (b) the same steps fed the device's own draws as host noise (iwae_debug_eps), which never speculates."""
    B, k = 20, 5
    x = O.synthetic_binarized(B, 31)
    P = O.init_params(1, nh, nl, 5, x_mean=O.synthetic_pixel_means())

    def run(mode):
        m = _model(1, nh, nl, options=opts)
        m.set_params(O.flatten_params(P))
        for t in range(12):
            e = None
            if mode == "host_eps":
                m.set_step(t, 0)
                e = m.debug_eps(B, k)
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", eps=e, scalars=False)
            if mode == "sync":
                m.sync()
        out = m.get_params().copy()
        m.close()
        return out

    p_async, p_sync, p_host = run("async"), run("sync"), run("host_eps")
    np.testing.assert_array_equal(p_async, p_sync)
    np.testing.assert_array_equal(p_async, p_host)


def test_two_layer_deferred_update_is_bitwise_equivalent(gpu):
    """2-layer training step at >= 8 192 rows (round 3): the layers behind the image encoder (q(z2|z1), p(z1|z2), the decoder) are summed
    from their slabs and updated on the side streams that carry their weight gradients (each stream its own layers), joined by the next forward in front of z1 --
    scheduling only: 12 steps on the device's noise land on exactly the parameters and Adam moments of the run whose main stream joins
    both side streams and updates everything itself (option no_defer2), with and without reads between the steps."""
    B, k, nh, nl = 170, 50, [200, 100], [100, 50]
    x = O.synthetic_binarized(B, 12)
    P = O.init_params(2, nh, nl, 4, x_mean=O.synthetic_pixel_means())

    def run(env, poke):
        m = _model(2, nh, nl, options=env)
        m.set_params(O.flatten_params(P))
        for t in range(12):
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=(t % 5 == 4))
            if poke and t % 4 == 1:
                m.get_params()
        out = m.get_params().copy(), m.get_adam_state()
        m.close()
        return out

    p0, (m0, v0, t0) = run({"no_defer2": 1}, False)
    p1, (m1, v1, t1) = run({}, False)
    p2, _ = run({}, True)
    p3, _ = run({"no_defer2_split": 1}, True)      # (one deferred update on the last side stream instead of one per side stream)
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(p0, p2)
    np.testing.assert_array_equal(p0, p3)
    np.testing.assert_array_equal(m0, m1)
    np.testing.assert_array_equal(v0, v1)
    assert t0 == t1 == 12


@pytest.mark.parametrize("B,k,obj,nh,nl,xd", [(6, 5, "iwae_elbo", 200, 100, 784), (9, 3, "vae_elbo_kl", 64, 20, 48), (170, 50, "iwae_elbo", 200, 100, 784)])
def test_conditional_model_matches_oracle(gpu, B, k, obj, nh, nl, xd):
    """tasks/task05.py:101-168 (CIWAE): encoder on concat(x, onehot(y)), decoder on concat(z, onehot(y)); the condition
    travels through iwae_set_condition.  Small and large row counts (both kernel families)."""
    from iwae_amd.native import NativeModel
    C = 10
    rng = np.random.default_rng(B * 7 + k)
    x = O.synthetic_binarized(B, 31, x_dim=xd)
    y = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]
    eps = rng.standard_normal((k, B, nl)).astype(np.float32)
    P = O.init_params(1, nh, nl, 17, x_mean=O.synthetic_pixel_means(xd), x_dim=xd, cond_dim=C)
    res_e, g_e = O.loss_grads_1layer(P, x, eps, 1.0, obj, rnd=O.bf16_round, y=y)
    m = NativeModel(1, nh, nl, x_dim=xd, seed=123, cond_dim=C)
    assert m.n_params == sum(W.size + b.size for W, b in P)
    m.set_params(O.flatten_params(P))
    with pytest.raises(RuntimeError):
        m.forward_backward(x, k, 1.0, obj, eps=eps)            # no condition set yet: fails loudly
    m.set_condition(y)
    r = m.forward_backward(x, k, 1.0, obj, eps=eps, want=("lpxz", "lqzx", "lpz"))
    for key in ("lpxz", "lqzx", "lpz"):
        assert np.max(np.abs(r[key] - res_e[key])) < EMU_ROW_ATOL, key
    for key in ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14"):
        assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    assert max(_grad_rel_errors(m.get_grads(), g_e)) < EMU_GRAD_REL
    # sample(z, y) (tasks/task05.py:185-198): decoder on concat(z, y)
    n = 7
    z = rng.standard_normal((n, nl)).astype(np.float32)
    yz = np.eye(C, dtype=np.float32)[np.full(n, 3)]
    m.set_condition(yz)
    dec = O._MLP3(P[4:7], O.bf16_round)
    ref = O.sigmoid(dec.fwd(O.bf16_round(np.concatenate([z, yz], axis=-1).astype(np.float64))))
    assert np.max(np.abs(m.decode(z) - ref)) < 2e-2
    # the k = 64 likelihood estimate walks the condition rows chunk by chunk
    m.set_condition(y)
    m.set_step(5, 0)
    a = m.eval_llh(x[:B], k=64, chunk=0)
    m.set_step(5, 0)                                      # same noise keys: only the chunking differs
    b2 = m.eval_llh(x[:B], k=64, chunk=max(1, B // 3))
    assert abs(a - b2) < 1e-3
    m.close()


@pytest.mark.parametrize("B,k,obj,beta", [(7, 6, "iwae_elbo", 1.0), (5, 4, "vae_elbo", 0.7), (6, 3, "vae_elbo_kl", 1.0), (170, 50, "iwae_eq14", 1.0)])
def test_conditional_prior_model_matches_oracle(gpu, B, k, obj, beta):
    """tasks/task04.py:101-173: the conditional model with the learned prior p(z|y) = N(mu_p(y), sigma_p(y)) (a BasicBlock on
    onehot(y), created after the decoder): lpz against that prior, its gradient through latent_bwd_kernel's second output."""
    from iwae_amd.native import NativeModel
    C, nh, nl, xd = 10, 200, 100, 784
    rng = np.random.default_rng(B + 100 * k)
    x = O.synthetic_binarized(B, 41)
    y = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]
    eps = rng.standard_normal((k, B, nl)).astype(np.float32)
    P = O.init_params(1, nh, nl, 19, x_mean=O.synthetic_pixel_means(xd), x_dim=xd, cond_dim=C, cond_prior=True)
    assert len(P) == 11
    res_e, g_e = O.loss_grads_1layer(P, x, eps, beta, obj, rnd=O.bf16_round, y=y)
    m = NativeModel(1, nh, nl, x_dim=xd, seed=123, cond_dim=C, cond_prior=True)
    assert m.n_params == sum(W.size + b.size for W, b in P)
    m.set_params(O.flatten_params(P))
    m.set_condition(y)
    r = m.forward_backward(x, k, beta, obj, eps=eps, want=("lpxz", "lqzx", "lpz"))
    for key in ("lpxz", "lqzx", "lpz"):
        # lpz divides by sigma_p^2 of a head that comes out of bf16-fed GEMMs: a bf16-ulp flip upstream moves it more than
        # the N(0,1) prior's lpz (same reason as lpz1z2 of the 2-layer model)
        assert np.max(np.abs(r[key] - res_e[key])) < (0.2 if key == "lpz" else EMU_ROW_ATOL), key
    for key in ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14"):
        assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    errs = _grad_rel_errors(m.get_grads(), g_e)
    if obj == "vae_elbo_kl":      # lpz is not in that objective: the prior network's gradient is exactly zero
        assert np.all(m.get_grads()[-sum(W.size + b.size for W, b in P[7:]):] == 0.0)
        errs = errs[:14]
    assert max(errs) < EMU_GRAD_REL
    with pytest.raises(ValueError):
        m.forward_backward(x, k, beta, "dreg", eps=eps)
    # sample(z, y) (tasks/task04.py:190-204): z -> mu_p(y) + sigma_p(y) * z, then the decoder on concat(z_new, y)
    n = 5
    z = rng.standard_normal((n, nl)).astype(np.float32)
    yz = np.eye(C, dtype=np.float32)[np.full(n, 8)]
    m.set_condition(yz)
    prior = O._Block(P[7:11], O.bf16_round)
    mu_p, sig_p = prior.fwd(O.bf16_round(yz.astype(np.float64)))
    dec = O._MLP3(P[4:7], O.bf16_round)
    ref = O.sigmoid(dec.fwd(O.bf16_round(np.concatenate([mu_p + sig_p * z, yz], axis=-1))))
    assert np.max(np.abs(m.decode(z) - ref)) < 2e-2
    m.close()


def test_changing_batch_shapes_do_not_leak_state(gpu):
    """One handle driven through changing (B, k) -- growing and shrinking buffers, the speculative noise prefetch missing its
    guess, the deferred decoder update pending across calls: every step must equal the same step on a fresh handle."""
    P = O.init_params(1, 200, 100, 5, x_mean=O.synthetic_pixel_means())
    m1 = _model(1, 200, 100)
    m1.set_params(O.flatten_params(P))
    shapes = [(64, 10), (16, 50), (200, 3), (64, 10), (64, 10), (180, 50), (5, 1)]
    for t, (B, k) in enumerate(shapes):
        x = O.synthetic_binarized(B, 100 + t)
        before = m1.get_params().copy()
        m2 = _model(1, 200, 100)
        m2.set_params(before)
        mo, ve, ts = m1.get_adam_state()
        m2.set_adam_state(mo, ve, ts)
        for m in (m1, m2):
            m.set_step(40 + t, 0)
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
        np.testing.assert_array_equal(m1.get_grads(), m2.get_grads())
        np.testing.assert_array_equal(m1.get_params(), m2.get_params())
        m2.close()
    m1.close()


def test_2layer_rejects_vae_elbo_kl_and_dreg(gpu):
    m = _model(2, [200, 100], [100, 50])
    x = O.synthetic_binarized(2, 1)
    for obj in ("vae_elbo_kl", "dreg"):
        with pytest.raises(ValueError):
            m.forward_backward(x, 2, 1.0, obj)
    m.close()


@pytest.mark.parametrize("name", ["tiny_1layer", "tiny_2layer", "full_1layer_B8_k50", "full_1layer_B20_k1", "full_2layer_B4_k5",
                                  "full_cond_B6_k5", "full_condprior_B6_k5"])
def test_against_golden_fixtures(gpu, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    nl = int(g["n_layers"])
    nh = g["n_hidden"].tolist() if nl == 2 else int(g["n_hidden"])
    nlat = g["n_latent"].tolist() if nl == 2 else int(g["n_latent"])
    B, k, beta = int(g["B"]), int(g["k"]), float(g["beta"])
    cond = int(g["cond_dim"]) if "cond_dim" in g else 0
    cprior = bool(int(g["cond_prior"])) if "cond_prior" in g else False
    if cond:
        from iwae_amd.native import NativeModel
        x, P, eps, y = MG.inputs(nl, nh, nlat, int(g["x_dim"]), B, k, int(g["seed"]), cond, cprior)
        m = NativeModel(nl, nh, nlat, x_dim=int(g["x_dim"]), seed=123, cond_dim=cond, cond_prior=cprior)
        m.set_condition(y)
    else:
        x, P, eps = MG.inputs(nl, nh, nlat, int(g["x_dim"]), B, k, int(g["seed"]))
        m = _model(nl, nh, nlat, int(g["x_dim"]))
    for obj in [str(o) for o in g["objectives"]]:
        m.set_params(O.flatten_params(P))
        r = m.forward_backward(x, k, beta, obj, eps=eps, want=("lpxz", "al", "lpz"))
        pre = "bf16/%s/" % obj
        px = "lpxz" if nl == 1 else "lpxz1"
        assert np.max(np.abs(r["lpxz"] - g[pre + px])) < 0.05
        if not cond:      # every row at the device's own Gaussian heads (round 5): 0.03 nat on log p(x|z), 0.02 on log p(z1|z2)
            if nl == 2:
                at = _densities_at_device_heads_2layer(m, eps[0], eps[1], B, k, nlat, P, x)
                assert np.max(np.abs(r["lpxz"] - at["lpxz1"])) < EMU_ROW_ATOL
                assert np.max(np.abs(r["lpz"] - at["lpz1z2"])) < 2e-2
            else:
                at = _densities_at_device_head(m, P, x, eps, nlat)
                assert np.max(np.abs(r["lpxz"] - at["lpxz"])) < EMU_ROW_ATOL
                assert np.max(np.abs(r["lpz"] - at["lpz"])) < 2e-2
        for key in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
            if pre + key in g:
                assert abs(r[key] - float(g[pre + key])) < 0.05, (obj, key)
                assert abs(r[key] - float(g["exact/%s/%s" % (obj, key)])) < 0.3
        gs = g[pre + "grad_summary"]
        flat = m.get_grads().astype(np.float64)
        # element by element: the whole flat gradient where the fixture carries it (tiny models), else the fixture's seeded
        # probe positions (<= 256 per tensor); per tensor the error is measured against that tensor's largest |element|
        off = 0
        shapes = O.layer_shapes(nl, nh, nlat, int(g["x_dim"]), cond, cprior)
        pidx = g["grad_probe_idx"]
        for li, (_, (fi, fo)) in enumerate(shapes):
            for ti, n in enumerate((fi * fo, fo)):
                t = flat[off:off + n]
                summ = gs[2 * li + ti]            # (sum, L2 norm, max|.|) of the oracle's tensor
                l2, amax = summ[1], summ[2]
                assert abs(np.sqrt((t * t).sum()) - l2) < 2e-2 * l2 + 1e-6, (obj, li, ti)
                assert abs(t.sum() - summ[0]) < 0.1 * l2 + 2e-2 * abs(summ[0]) + 1e-6, (obj, li, ti, "sum")     # signed: a flipped block shows here too
                sel = (pidx >= off) & (pidx < off + n)
                want = g[pre + "grad_probe"][sel]
                got = flat[pidx[sel]]
                assert np.max(np.abs(got - want)) <= 3e-2 * amax + 1e-7, (obj, li, ti, "probe", float(np.max(np.abs(got - want))), float(amax))
                assert np.linalg.norm(got - want) <= 2e-2 * np.linalg.norm(want) + 1e-7, (obj, li, ti, "probe l2")
                if pre + "grad_flat" in g:
                    full = g[pre + "grad_flat"][off:off + n]
                    assert np.linalg.norm(t - full) <= 2e-2 * np.linalg.norm(full) + 1e-7, (obj, li, ti, "flat")
                off += n
    m.close()


def test_device_noise_matches_published_philox(gpu):
    """The device Philox4x32-10 + Box-Muller stream equals its NumPy restatement (which is pinned to the
    Random123 known-answer vectors in tests/test_oracle.py); fast v_log/v_sin/v_cos: abs err <= 2e-5."""
    m = _model(1, 200, 100)
    m.set_step(7, 0)
    e = m.debug_eps(5, 3, 0)
    ref = philox_np.device_eps(123, 7, 5, 3, 100)
    assert np.max(np.abs(e - ref)) < 2e-5
    m.set_step(7, 2)                       # batch_offset 2: same draws as images 2.. of the unsplit batch
    e2 = m.debug_eps(3, 3, 0)
    np.testing.assert_array_equal(e2, e[:, 2:5])
    # and the forward pass consumes exactly these draws
    x = O.synthetic_binarized(5, 3)
    m.set_step(9, 0)
    ed = m.debug_eps(5, 3, 0)
    r1 = m.forward(x, 3, want=("lpxz", "lpz"))
    r2 = m.forward(x, 3, eps=ed, want=("lpxz", "lpz"))
    np.testing.assert_allclose(r1["lpz"], r2["lpz"], atol=1e-4)
    np.testing.assert_allclose(r1["lpxz"], r2["lpxz"], atol=1e-3)
    m.close()


def test_decode_matches_oracle(gpu):
    x, P, eps = MG.inputs(1, 200, 100, 784, 4, 2, 55)
    m = _model(1, 200, 100)
    m.set_params(O.flatten_params(P))
    z = np.random.default_rng(0).standard_normal((37, 100)).astype(np.float32)
    probs = m.decode(z)
    dec = O._MLP3(P[4:7], O.bf16_round)
    ref = O.sigmoid(dec.fwd(O.bf16_round(z)))
    assert probs.shape == (37, 784) and np.max(np.abs(probs - ref)) < 5e-3
    m.close()


def test_decode_2layer_matches_oracle_mean_path(gpu):
    """src/iwae2.py:184-196 with the device's own z1 draw: probs = sigmoid(dec1(mu_p + sigma_p * eps)), eps = Philox stream 2."""
    x, P, eps = MG.inputs(2, [200, 100], [100, 50], 784, 4, 2, 66)
    m = _model(2, [200, 100], [100, 50])
    m.set_params(O.flatten_params(P))
    z2 = np.random.default_rng(1).standard_normal((21, 50)).astype(np.float32)
    m.set_step(4, 0)
    probs = m.decode(z2)
    dec2 = O._Block(P[8:12], O.bf16_round)
    mup, sigp = dec2.fwd(O.bf16_round(z2))
    e = philox_np.device_eps(123, 4, 21, 1, 100, stream=2)[0]
    dec1 = O._MLP3(P[12:15], O.bf16_round)
    ref = O.sigmoid(dec1.fwd(O.bf16_round(mup + sigp * e)))
    assert probs.shape == (21, 784) and np.max(np.abs(probs - ref)) < 2e-2
    m.close()


# ---------------------------------------------------------------- full-size properties (BASELINE configs[1])
@pytest.fixture(scope="module")
def big(gpu):
    m = _model(1, 200, 100)
    P = O.init_params(1, 200, 100, 123, x_mean=O.synthetic_pixel_means())
    m.set_params(O.flatten_params(P))
    x = O.synthetic_binarized(1024, 9)
    yield m, x
    m.close()


def test_full_size_invariants(big):
    m, x = big
    m.set_step(3, 0)
    r = m.forward_backward(x, 50, 1.0, "iwae_elbo", want=("al", "lpxz", "lpz", "lqzx", "log_w"))
    np.testing.assert_allclose(r["al"].sum(0), 1.0, atol=1e-5)                     # softmax over k
    np.testing.assert_allclose(r["log_w"], r["lpxz"] + r["lpz"] - r["lqzx"], rtol=0, atol=1e-3)   # iwae1.py:113
    assert r["iwae_elbo"] >= r["vae_elbo"] - 1e-3                                   # Jensen: L_k >= L_1
    lme = np.log(np.mean(np.exp(r["log_w"].astype(np.float64) - r["log_w"].max(0)), 0)) + r["log_w"].max(0)
    assert abs(lme.mean() - r["iwae_elbo"]) < 2e-3                                  # utils.py:6-8 on the device log_w
    g1 = m.get_grads()
    m.set_step(3, 0)
    m.forward_backward(x, 50, 1.0, "iwae_elbo")
    np.testing.assert_array_equal(g1, m.get_grads())                                # deterministic (no float atomics)
    assert np.all(np.isfinite(g1)) and np.linalg.norm(g1) > 0


def test_full_size_batch_permutation_and_shard_equivalence(big):
    """Images are independent units: permuting the batch permutes per-image outputs, and the mean of two
    half-batch gradients (noise keyed by the GLOBAL image index) equals the full-batch gradient --
    the identity the 8-GPU data-parallel step rests on."""
    m, x = big
    k = 50
    m.set_step(11, 0)
    rf = m.forward_backward(x, k, 1.0, "iwae_elbo", want=("lpxz",))
    gf = m.get_grads().astype(np.float64)
    halves = []
    for h in range(2):
        m.set_step(11, 512 * h)
        rh = m.forward_backward(x[512 * h:512 * (h + 1)], k, 1.0, "iwae_elbo", want=("lpxz",))
        np.testing.assert_allclose(rh["lpxz"], rf["lpxz"][:, 512 * h:512 * (h + 1)], atol=1e-3)
        halves.append(m.get_grads().astype(np.float64))
    gavg = 0.5 * (halves[0] + halves[1])
    assert np.linalg.norm(gavg - gf) / np.linalg.norm(gf) < 1e-4
    # permutation with explicit noise
    rng = np.random.default_rng(0)
    eps = rng.standard_normal((k, 64, 100)).astype(np.float32)
    perm = rng.permutation(64)
    a = m.forward(x[:64], k, eps=eps, want=("lpxz", "lqzx"))
    b = m.forward(x[:64][perm], k, eps=eps[:, perm], want=("lpxz", "lqzx"))
    np.testing.assert_allclose(a["lpxz"][:, perm], b["lpxz"], atol=1e-3)
    np.testing.assert_allclose(a["lqzx"][:, perm], b["lqzx"], atol=1e-4)


def test_eval_llh_chunking_and_definition(big):
    """main.py:170-184: mean over images of iwae_elbo(k, B=1).  Chunking must not change the estimate
    (noise is keyed by the global image index) and it must equal per-image forward calls."""
    m, x = big
    for prec in ("fp32", "bf16"):          # the evaluator's default arithmetic (float32) and the fast path
        m.set_eval_precision(prec)
        m.set_step(21, 0)
        a, pa = m.eval_llh(x[:48], k=500, chunk=48, per_image=True)
        m.set_step(21, 0)
        b, pb = m.eval_llh(x[:48], k=500, chunk=7, per_image=True)
        np.testing.assert_allclose(pa, pb, atol=2e-3)
        assert abs(a - b) < 1e-3
        m.set_step(21, 5)
        single = m.forward(x[5:6], 500)["iwae_elbo"]          # this handle's forward is the bf16 path
        assert abs(single - pa[5]) < (2e-3 if prec == "bf16" else EXACT_SCALAR_ATOL)
    # k = 5000 (the reference's L) on a few images: runs, finite, and tighter than k = 50 on average
    m.set_step(22, 0)
    l5000 = m.eval_llh(x[:8], k=5000)
    m.set_step(22, 0)
    l50 = m.eval_llh(x[:8], k=50)
    assert np.isfinite(l5000) and l5000 > l50 - 0.5
    m.set_eval_precision("fp32")


@pytest.mark.parametrize("layers", [1, 2])
def test_eval_llh_k_chunking_is_invisible(gpu, layers):
    """iwae_eval_llh walks an image's k samples in chunks when k exceeds the rows-per-launch cap (IWAE_EVAL_ROWS) and merges the
    chunks' log-mean-exps with a running log-sum-exp (src/utils.py:6-8 in associative form); the Philox rows are those of the
    unchunked call, so the per-image estimates must not move -- k = 1500 in chunks of 256 (5 full + 1 ragged) against one launch,
    both arithmetics, 1- and 2-layer model."""
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    P = O.init_params(layers, nh, nl, 77, x_mean=O.synthetic_pixel_means())
    x = O.synthetic_binarized(6, 5)
    outs = {}
    for rows in (0, 256):
        m = _model(layers, nh, nl, options={"eval_rows": rows} if rows else None)
        m.set_params(O.flatten_params(P))
        for prec in ("fp32", "bf16"):
            m.set_eval_precision(prec)
            m.set_step(31, 4)
            outs[(rows, prec)] = m.eval_llh(x, k=1500, per_image=True)
            if rows:
                # the chunked evaluator's last launch drew (step 31, image 4 + 5, the 220 samples 1280.. of 1500): a training step whose
                # (step, offset, rows) happen to equal that launch's must NOT take those draws for its own (the ring slot's tag is
                # invalidated by both arithmetics' chunked forwards)
                m.set_step(31, 9)
                outs[("after", prec)] = (m.forward_backward(x[5:6], 220, 1.0, "iwae_elbo")["iwae_elbo"], m.get_grads().copy())
        m.close()
    fresh = _model(layers, nh, nl)
    fresh.set_params(O.flatten_params(P))
    fresh.set_step(31, 9)
    e_ref, g_ref = fresh.forward_backward(x[5:6], 220, 1.0, "iwae_elbo")["iwae_elbo"], fresh.get_grads().copy()
    fresh.close()
    for prec in ("fp32", "bf16"):
        assert outs[("after", prec)][0] == e_ref, prec
        np.testing.assert_array_equal(outs[("after", prec)][1], g_ref)
    for prec in ("fp32", "bf16"):
        (a, pa), (b, pb) = outs[(0, prec)], outs[(256, prec)]
        np.testing.assert_allclose(pa, pb, atol=2e-3)
        assert abs(a - b) < 1e-3


def test_eval_llh_images_per_launch_are_invisible(gpu):
    """The k = 5000 evaluator (main.py:170-184) packs as many images into a launch as the row cap allows -- 419 at the reference's dims since round 4
    (2^21 rows), 104 before (2^19) -- and keeps every launch's per-image estimates on the device until ONE copy at the end.  The draws are keyed by the
    global image index, so an image's estimate must not depend on which launch it rode in or on its neighbours: 450 images (two launches, the second
    ragged) against launches of 104 and of 7 images, per image, both arithmetics.  Bitwise, except bf16 at 7 images per launch: 35 000 rows take the
    kernel family that samples z inside the decoder kernel (another summation order of the 100 density terms per row): within 2 float32 ulps there."""
    P = O.init_params(1, 200, 100, 78, x_mean=O.synthetic_pixel_means())
    x = O.synthetic_binarized(450, 9)
    outs = {}
    for rows in (0, 1 << 19, 7 * 5000):
        m = _model(1, 200, 100, options={"eval_rows": rows} if rows else None)
        m.set_params(O.flatten_params(P))
        for prec in ("fp32", "bf16"):
            m.set_eval_precision(prec)
            m.set_step(3, 0)
            outs[(rows, prec)] = m.eval_llh(x, k=5000, per_image=True)
        m.close()
    for prec in ("fp32", "bf16"):
        a, pa = outs[(0, prec)]
        assert np.all(np.isfinite(pa)) and pa.shape == (450,)
        for rows in (1 << 19, 7 * 5000):
            b, pb = outs[(rows, prec)]
            if prec == "bf16" and rows < 1 << 19:
                np.testing.assert_allclose(pa, pb, rtol=0, atol=7e-5, err_msg="%s rows %d" % (prec, rows))
                assert abs(a - b) < 1e-5
            else:
                np.testing.assert_array_equal(pa, pb, err_msg="%s rows %d" % (prec, rows))
                assert a == b


# ---------------------------------------------------------------- float32 mode (iwae_config.precision = IWAE_PREC_FP32)
# "fp32 kernels rel 1e-5 on scalars / 1e-4 on grads vs fp64 oracle".  The reference computes in float32
# (Keras Dense defaults, src/iwae1.py:31-34,72-75); every GEMM of this mode is an exact-f32 MFMA (v_mfma_f32_16x16x4_f32).
F32_SCALAR_REL, F32_GRAD_REL, F32_ROW_ATOL = 1e-5, 1e-4, 2e-3

CASES_F32 = [  # (layers, B, k, objective, beta, n_hidden, n_latent, x_dim)
    (1, 20, 1, "vae_elbo", 1.0, 200, 100, 784),          # BASELINE configs[0]
    (1, 8, 50, "iwae_elbo", 1.0, 200, 100, 784),
    (1, 5, 7, "vae_elbo_kl", 0.7, 200, 100, 784),
    (1, 6, 5, "iwae_eq14", 1.0, 200, 100, 784),
    (1, 6, 5, "dreg", 1.0, 200, 100, 784),               # tasks/task02.py
    (1, 5, 3, "iwae_elbo", 0.7, 16, 4, 48),              # the tiny fixture's dims
    (1, 3, 130, "iwae_elbo", 1.0, 64, 2, 784),           # 2-D latent (tasks/task01.py), k > 64
    (1, 170, 50, "iwae_elbo", 1.0, 200, 100, 784),       # 8 500 rows: row-split weight gradients
    (2, 4, 3, "iwae_elbo", 1.0, [200, 100], [100, 50], 784),
    (2, 6, 50, "vae_elbo", 1.0, [200, 100], [100, 50], 784),
    (2, 3, 5, "iwae_eq14", 1.0, [16, 8], [4, 2], 48),
]


@pytest.mark.parametrize("layers,B,k,obj,beta,nh,nl,xd", CASES_F32)
def test_float32_mode_matches_exact_oracle(gpu, layers, B, k, obj, beta, nh, nl, xd):
    from iwae_amd.native import NativeModel
    x, P, eps = MG.inputs(layers, nh, nl, xd, B, k, 300 + B + k)
    if layers == 1:
        res, g = O.loss_grads_1layer(P, x, eps, beta, obj)
        rows = (("lpxz", "lpxz"), ("lpz", "lpz"), ("lqzx", "lqzx"))
        keys = ("iwae_elbo",) if obj == "dreg" else ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14")
    else:
        res, g = O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj)
        rows = (("lpxz", "lpxz1"), ("lpz", "lpz1z2"), ("lpz2", "lpz2"), ("lqzx", "lqz1x"), ("lqzx2", "lqz2z1"))
        keys = ("vae_elbo", "iwae_elbo", "iwae_eq14")
    m = NativeModel(layers, nh, nl, x_dim=xd, seed=123, precision="fp32")
    m.set_params(O.flatten_params(P))
    r = m.forward_backward(x, k, beta, obj, eps=eps, want=tuple(a for a, _ in rows) + ("al", "logits", "z"))
    for a, b in rows:
        assert np.max(np.abs(r[a] - res[b])) < F32_ROW_ATOL, (a, float(np.max(np.abs(r[a] - res[b]))))
    assert np.max(np.abs(r["logits"] - res["logits"])) < 2e-4
    np.testing.assert_allclose(r["z"], res["z"] if layers == 1 else res["z1"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(r["al"], res["al"], atol=2e-4)
    for key in keys:
        assert abs(r[key] - res[key]) <= F32_SCALAR_REL * abs(res[key]) + 2e-4, (key, r[key], res[key])
    if obj == "dreg":
        assert abs(r["inference_loss"] - res["inference_loss"]) <= 1e-4 * abs(res["inference_loss"]) + 1e-4
    flat = m.get_grads()
    errs = _grad_rel_errors(flat, g)
    assert max(errs) < F32_GRAD_REL, errs
    # the forward-only call and the fused train step agree with the two-call path; Keras Adam from the device gradient
    r0 = m.forward(x, k, beta, eps=eps)
    for key in keys:
        assert abs(r0[key] - r[key]) <= 1e-6 * abs(r[key]) + 1e-5
    r2 = m.train_step(x, k, beta, 1e-3, obj, eps=eps)
    # (bitwise at small row counts; from ~8 000 rows on the call without `logits` takes the output layer's fused epilogue: the same sums in another order)
    g2 = m.get_grads().astype(np.float64)
    assert np.linalg.norm(g2 - flat) / np.linalg.norm(flat) < 1e-5
    ref, _, _ = O.adam_update(O.flatten_params(P), flat.astype(np.float64), 0.0, 0.0, 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6
    m.close()


@pytest.mark.parametrize("layers,B,k,obj", [(1, 200, 50, "iwae_elbo"), (1, 200, 50, "dreg"), (2, 192, 50, "iwae_elbo")])
def test_float32_mode_fused_output_layer_matches_exact_oracle(gpu, layers, B, k, obj):
    """float32 mode at >= 9 400 rows (round 3): the output layer's 128-tile GEMM takes log p(x|z) in its epilogue (per half tile partial sums,
    added by lse_kernel in a fixed order) and, in a training step, leaves s = x - sigmoid(l) where the logits would have gone; the backward pass
    takes the row weight g_r inside the weight-gradient GEMM's operand fetch and the dX GEMM's epilogue instead of a pass that makes
    dl = g_r s.  Scalars and every gradient tensor against the exact float64 oracle at the float32 tolerances, and against the same mode with
    the separate bern_f32 / dl_f32 passes (option no_f32_bern_fused)."""
    from iwae_amd.native import NativeModel
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x, P, eps = MG.inputs(layers, nh, nl, 784, B, k, 640 + B + k)
    res, g = (O.loss_grads_1layer(P, x, eps, 1.0, obj) if layers == 1 else O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, obj))
    keys = ("iwae_elbo",) if obj == "dreg" else ("vae_elbo", "iwae_elbo", "iwae_eq14")
    out = []
    for opts in ({}, {"no_f32_bern_fused": 1}):
        m = NativeModel(layers, nh, nl, x_dim=784, seed=123, precision="fp32", options=opts)
        m.set_params(O.flatten_params(P))
        r = m.forward_backward(x, k, 1.0, obj, eps=eps)
        for key in keys:
            assert abs(r[key] - res[key]) <= F32_SCALAR_REL * abs(res[key]) + 2e-4, (opts, key, r[key], res[key])
        flat = m.get_grads()
        assert max(_grad_rel_errors(flat, g)) < F32_GRAD_REL, opts
        r0 = m.forward(x, k, 1.0, eps=eps)          # (forward only: nothing is written but the partial sums)
        for key in keys:
            assert abs(r0[key] - r[key]) <= 1e-6 * abs(r[key]) + 1e-5
        out.append((r, flat.astype(np.float64)))
        m.close()
    (r1, g1), (r2, g2) = out
    for key in keys:
        assert abs(r1[key] - r2[key]) <= 2e-6 * abs(r2[key])
    assert np.linalg.norm(g1 - g2) / np.linalg.norm(g2) < 1e-5


@pytest.mark.parametrize("B,k,nh,nl,xd,obj", [(100, 50, 200, 100, 784, "iwae_elbo"), (1433, 3, 200, 100, 784, "iwae_elbo"), (90, 47, 64, 20, 100, "vae_elbo"),
                                              (4200, 1, 200, 100, 784, "vae_elbo"), (101, 41, 208, 128, 52, "dreg")])
def test_float32_decoder_forward_in_one_launch_matches_exact_oracle(gpu, B, k, nh, nl, xd, obj):
    """float32 mode at >= 4 096 rows (round 4): the decoder forward is ONE launch (dec_fwd_f32_kernel: a wave owns 16 rows through the two tanh
    layers and the output layer, activations in LDS, log p(x|z) whole per row; a training step also keeps g1, g2 and s = x - sigmoid(l)).
    Shapes: the reference's; k = 3 and k = 1 (a wave's 16 rows span more than three images: x straight from memory instead of through LDS); a
    ragged row count (4 230 = 66 workgroups + 6 rows) with narrow layers and 100 pixels (one pass of 7 tiles); the widest layers the kernel
    takes (208 / 128 / 52).  Per-row log p(x|z), the objective values and every gradient tensor against the exact float64 oracle at the float32
    tolerances, and against the same mode with three GEMM launches (option no_f32_dec_fused)."""
    from iwae_amd.native import NativeModel
    x, P, eps = MG.inputs(1, nh, nl, xd, B, k, 770 + B + k)
    res, g = O.loss_grads_1layer(P, x, eps, 1.0, obj)
    keys = ("iwae_elbo",) if obj == "dreg" else ("vae_elbo", "iwae_elbo", "iwae_eq14")
    out = []
    # (round 5: a training step takes the three launches by default -- the fused kernel is the forward-only calls' and option f32_dec_fused_train's)
    for opts in ({"f32_dec_fused_train": 1}, {"no_f32_dec_fused": 1}, {}):
        m = NativeModel(1, nh, nl, x_dim=xd, seed=123, precision="fp32", options=opts)
        m.set_params(O.flatten_params(P))
        r = m.forward_backward(x, k, 1.0, obj, eps=eps, want=("lpxz",))
        assert np.max(np.abs(r["lpxz"] - res["lpxz"])) < F32_ROW_ATOL, opts
        for key in keys:
            assert abs(r[key] - res[key]) <= F32_SCALAR_REL * abs(res[key]) + 2e-4, (opts, key, r[key], res[key])
        flat = m.get_grads()
        assert max(_grad_rel_errors(flat, g)) < F32_GRAD_REL, opts
        r0 = m.forward(x, k, 1.0, eps=eps, want=("lpxz",))          # (forward only: g1, g2, s are not written)
        np.testing.assert_allclose(r0["lpxz"], r["lpxz"], rtol=1e-6, atol=1e-4)
        out.append((r, flat.astype(np.float64)))
        m.close()
    (r1, g1), (r2, g2), (r3, g3) = out
    np.testing.assert_allclose(r1["lpxz"], r2["lpxz"], rtol=2e-6, atol=2e-4)
    assert np.linalg.norm(g1 - g2) / np.linalg.norm(g2) < 1e-5
    np.testing.assert_array_equal(r3["lpxz"], r2["lpxz"])
    np.testing.assert_array_equal(g3, g2)


@pytest.mark.parametrize("layers,B,k,obj", [(1, 100, 50, "iwae_elbo"), (1, 1024, 5, "vae_elbo"), (2, 96, 50, "iwae_elbo")])
def test_float32_step_on_two_streams_is_bitwise_the_one_stream_step(gpu, layers, B, k, obj):
    """Round 5, float32 mode at >= 4 096 rows: the decoder's three weight gradients, their slab reduction and their Adam update run on the side
    stream, the update deferred past the next step's encoder forward and sampling (backward_f32; joined in front of the decoder forward, by every
    call that reads parameters or gradients, and by a forward-only call in between).  Same kernels in the same order per stream: parameters, Adam
    state and objective values after four steps (device noise, a forward-only call and a gradient read in between) are BITWISE those of the
    one-stream step (option no_f32_side)."""
    from iwae_amd.native import NativeModel
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x = O.synthetic_binarized(B, 11)
    out = []
    for opts in ({}, {"no_f32_side": 1}, {"f32_wout_last": 1}):
        m = NativeModel(layers, nh, nl, seed=77, precision="fp32", options=opts)
        vals = []
        for step in range(4):
            r = m.train_step(x, k, 1.0, 1e-3, obj)
            vals.append([r["vae_elbo"], r["iwae_elbo"]])
            if step == 1:
                vals.append([m.forward(x, k, 1.0)["iwae_elbo"], 0.0])          # forward only: joins the deferred update first
            if step == 2:
                m.forward_backward(x, k, 1.0, obj)
                vals.append([float(np.abs(m.get_grads()).sum()), 0.0])         # the gradient of a two-call step, read while the side stream may still hold it
        out.append((np.array(vals), m.get_params().copy(), [a.copy() for a in m.get_adam_state()[:2]]))
        m.close()
    for other in out[1:]:
        np.testing.assert_array_equal(out[0][0], other[0])
        np.testing.assert_array_equal(out[0][1], other[1])
        for a, b in zip(out[0][2], other[2]):
            np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("layers,B,k", [(1, 100, 50), (1, 1024, 5), (2, 96, 50)])
def test_float32_gemm_kernel_variants_agree(gpu, layers, B, k):
    """Round 5: the float32 GEMMs' k loop was rewritten (gemm_f32_v2_kernel: 16-byte conflict-free LDS stores and reads with lane quad q contracting
    k = 4q + j, transposed 16-byte epilogues, 8-wave 128 x 224 / 224 x 128 tiles, the v2 loop on 64 x 64 tiles, K-split few-row products with the
    epilogue in the reduction).  Every variant against the exact float64 oracle at the float32 tolerances, and against each other to 1e-5 -- they
    differ in the ORDER of float32 sums only.  (The switches are process-wide: each is put back.)"""
    from iwae_amd.native import NativeModel
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x, P, eps = MG.inputs(layers, nh, nl, 784, B, k, 880 + B + k)
    res, g = (O.loss_grads_1layer(P, x, eps, 1.0, "iwae_elbo") if layers == 1 else O.loss_grads_2layer(P, x, eps[0], eps[1], 1.0, "iwae_elbo"))
    grads = {}
    for name in (None, "f32_gemm_v1", "f32_gemm_w4", "f32_gemm_small_v1", "f32_no_ksplit"):
        m = NativeModel(layers, nh, nl, seed=123, precision="fp32")
        if name:
            m.set_option(name, 1)
        try:
            m.set_params(O.flatten_params(P))
            r = m.forward_backward(x, k, 1.0, "iwae_elbo", eps=eps)
            for key in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
                assert abs(r[key] - res[key]) <= F32_SCALAR_REL * abs(res[key]) + 2e-4, (name, key, r[key], res[key])
            flat = m.get_grads()
            assert max(_grad_rel_errors(flat, g)) < F32_GRAD_REL, name
            grads[name] = flat.astype(np.float64)
        finally:
            if name:
                m.set_option(name, 0)
            m.close()
    for name, gv in grads.items():
        assert np.linalg.norm(gv - grads[None]) / np.linalg.norm(grads[None]) < 1e-5, name


@pytest.mark.parametrize("layers,B,k", [(1, 100, 50), (2, 96, 50), (1, 20, 5)])
def test_float32_data_parallel_step_world1_is_bitwise_the_single_gpu_step(gpu, layers, B, k):
    """float32 mode through the in-library data-parallel step with ONE rank (iwae_comm_init + iwae_train_step): the gradient comes off both streams
    (round 5: the decoder's weight gradients on the side stream), is joined, all-reduced whole and applied by one Adam launch -- against the
    single-GPU step, which applies the encoder's segment on the main stream and the decoder's, deferred, on the side stream.  Eight steps on device
    noise: bit-identical parameters, Adam state and last gradient."""
    from iwae_amd.native import NativeModel
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    x = O.synthetic_binarized(B, 17)
    P = O.init_params(layers, nh, nl, 23, x_mean=O.synthetic_pixel_means())
    outs = []
    for dp in (False, True):
        m = NativeModel(layers, nh, nl, seed=123, precision="fp32")
        m.set_params(O.flatten_params(P))
        if dp:
            m.comm_init(NativeModel.comm_unique_id(), 1, 0)
        for t in range(8):
            m.set_step(t, 0)
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=(t % 3 == 0))
            if t == 4:
                m.get_params()
        outs.append((m.get_params().copy(), m.get_adam_state(), m.get_grads().copy()))
        if dp:
            m.comm_destroy()
            m.train_step(x, k, 1.0, 1e-3, "iwae_elbo")
        m.close()
    np.testing.assert_array_equal(outs[0][2], outs[1][2])
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1][0], outs[1][1][0])
    np.testing.assert_array_equal(outs[0][1][1], outs[1][1][1])
    assert outs[0][1][2] == outs[1][1][2] == 8


def test_float32_mode_against_golden_fixtures(gpu):
    """The tiny fixtures' exact (float64) expectations, element by element, at SURVEY 8(c)'s float32 tolerances."""
    from iwae_amd.native import NativeModel
    for name in ("tiny_1layer", "tiny_2layer", "full_1layer_B20_k1"):
        g = np.load(os.path.join(GOLD, name + ".npz"))
        nl = int(g["n_layers"])
        nh = g["n_hidden"].tolist() if nl == 2 else int(g["n_hidden"])
        nlat = g["n_latent"].tolist() if nl == 2 else int(g["n_latent"])
        B, k, beta, xd = int(g["B"]), int(g["k"]), float(g["beta"]), int(g["x_dim"])
        x, P, eps = MG.inputs(nl, nh, nlat, xd, B, k, int(g["seed"]))
        m = NativeModel(nl, nh, nlat, x_dim=xd, seed=123, precision="fp32")
        for obj in [str(o) for o in g["objectives"]]:
            m.set_params(O.flatten_params(P))
            r = m.forward_backward(x, k, beta, obj, eps=eps)
            pre = "exact/%s/" % obj
            for key in ("vae_elbo", "iwae_elbo", "iwae_eq14"):
                if pre + key in g:
                    assert abs(r[key] - float(g[pre + key])) <= F32_SCALAR_REL * abs(float(g[pre + key])) + 2e-4, (name, obj, key)
            flat = m.get_grads().astype(np.float64)
            want = g[pre + "grad_probe"]
            got = flat[g["grad_probe_idx"]]
            assert np.linalg.norm(got - want) <= F32_GRAD_REL * np.linalg.norm(want) + 1e-9, (name, obj)
            if pre + "grad_flat" in g:
                full = g[pre + "grad_flat"]
                assert np.linalg.norm(flat - full) <= F32_GRAD_REL * np.linalg.norm(full), (name, obj)
        m.close()


def test_float32_mode_with_device_noise_and_dataset_pipeline(gpu):
    """float32 mode on the device's own Philox noise (oracle fed the NumPy restatement of the same draws) and through the
    resident-dataset input path; 15 fused train steps track the float64 oracle trajectory."""
    from iwae_amd.native import NativeModel
    B, k, step = 40, 10, 3
    x = O.synthetic_binarized(B, 23)
    P = O.init_params(1, 200, 100, 29, x_mean=O.synthetic_pixel_means())
    eps = philox_np.device_eps(123, step, B, k, 100)
    res, g = O.loss_grads_1layer(P, x, eps, 1.0, "iwae_elbo")
    m = NativeModel(1, 200, 100, seed=123, precision="fp32")
    m.set_params(O.flatten_params(P))
    m.set_step(step, 0)
    r = m.forward_backward(x, k, 1.0, "iwae_elbo", want=("lpxz", "lqzx"))
    assert abs(r["iwae_elbo"] - res["iwae_elbo"]) < 2e-2        # float32 Box-Muller with fast sin / cos / log: eps differs by <= 2e-5
    assert max(_grad_rel_errors(m.get_grads(), g)) < 2e-3
    # trajectory with explicit noise
    flat = O.flatten_params(P)
    mo = vo = 0.0
    rng = np.random.default_rng(5)
    for t in range(1, 16):
        e = rng.standard_normal((k, B, 100)).astype(np.float32)
        rr = m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", eps=e)
        Pt = O.unflatten_params(flat, 1, 200, 100)
        rt, gt = O.loss_grads_1layer(Pt, x, e, 1.0, "iwae_elbo")
        flat, mo, vo = O.adam_update(flat, O.flatten_grads(gt), mo, vo, t, 1e-3)
        assert abs(rr["iwae_elbo"] - rt["iwae_elbo"]) < 5e-3
    assert np.max(np.abs(m.get_params() - flat)) < 2e-5
    # resident dataset path == host batch path
    rng = np.random.default_rng(3)
    gray = (rng.random((200, 784)) * 256).astype(np.uint8)
    m.dataset_upload(gray)
    m.dataset_begin_epoch(2, rng.permutation(200).astype(np.int32))
    xb = m.dataset_get_batch(11, 64)
    p0 = m.get_params().copy(); mo_, ve_, ts_ = m.get_adam_state()
    m.set_step(9, 0)
    a = m.train_step_dataset(11, 64, 5, 1.0, 1e-3, "iwae_elbo")
    pa = m.get_params().copy()
    m.set_params(p0); m.set_adam_state(mo_, ve_, ts_)
    m.set_step(9, 0)
    b = m.train_step(xb, 5, 1.0, 1e-3, "iwae_elbo")
    assert a["iwae_elbo"] == b["iwae_elbo"]
    np.testing.assert_array_equal(pa, m.get_params())
    m.close()


@pytest.mark.parametrize("prior,B,k,obj,beta", [(False, 6, 5, "iwae_elbo", 1.0), (False, 9, 3, "vae_elbo_kl", 0.7), (False, 170, 50, "iwae_eq14", 1.0),
                                                 (True, 7, 6, "iwae_elbo", 1.0), (True, 5, 4, "vae_elbo", 0.7), (True, 170, 50, "iwae_elbo", 1.0)])
def test_float32_mode_conditional_models_match_exact_oracle(gpu, prior, B, k, obj, beta):
    """float32 mode (exact-f32 MFMA GEMMs) of the conditional models: tasks/task05.py:101-168 (encoder on concat(x, y), decoder on
    concat(z, y)) and tasks/task04.py:101-173 (the same with the learned prior p(z|y)), against the exact float64 oracle at the
    float32 tolerances of SURVEY 8(c) -- every gradient tensor incl. the prior network's, the per-row densities, and the float32
    k-chunked evaluator (the evaluator's default arithmetic) walking the condition rows."""
    from iwae_amd.native import NativeModel
    C, nh, nl, xd = 10, 200, 100, 784
    rng = np.random.default_rng(1000 * prior + B + 10 * k)
    x = O.synthetic_binarized(B, 51)
    y = np.eye(C, dtype=np.float32)[rng.integers(0, C, B)]
    eps = rng.standard_normal((k, B, nl)).astype(np.float32)
    P = O.init_params(1, nh, nl, 23, x_mean=O.synthetic_pixel_means(xd), x_dim=xd, cond_dim=C, cond_prior=prior)
    res, g = O.loss_grads_1layer(P, x, eps, beta, obj, y=y)
    m = NativeModel(1, nh, nl, x_dim=xd, seed=123, cond_dim=C, cond_prior=prior, precision="fp32")
    m.set_params(O.flatten_params(P))
    with pytest.raises(RuntimeError):
        m.forward_backward(x, k, beta, obj, eps=eps)            # no condition set yet: fails loudly
    m.set_condition(y)
    r = m.forward_backward(x, k, beta, obj, eps=eps, want=("lpxz", "lqzx", "lpz", "logits"))
    for key in ("lpxz", "lqzx", "lpz"):
        assert np.max(np.abs(r[key] - res[key])) < F32_ROW_ATOL, (key, float(np.max(np.abs(r[key] - res[key]))))
    assert np.max(np.abs(r["logits"] - res["logits"])) < 2e-4
    for key in ("vae_elbo", "vae_elbo_kl", "iwae_elbo", "iwae_eq14"):
        assert abs(r[key] - res[key]) <= F32_SCALAR_REL * abs(res[key]) + 2e-4, (key, r[key], res[key])
    errs = _grad_rel_errors(m.get_grads(), g)
    assert max(errs) < F32_GRAD_REL, errs
    # Keras Adam from that gradient through the fused train step
    flat = m.get_grads()
    m.train_step(x, k, beta, 1e-3, obj, eps=eps)
    ref, _, _ = O.adam_update(O.flatten_params(P), flat.astype(np.float64), 0.0, 0.0, 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6
    # the evaluator (float32 by default) on the conditional model: chunking over images does not move the estimate
    m.set_params(O.flatten_params(P))
    m.set_condition(y)
    m.set_step(5, 0)
    a = m.eval_llh(x, k=64, chunk=0)
    m.set_step(5, 0)
    b2 = m.eval_llh(x, k=64, chunk=max(1, B // 3))
    assert abs(a - b2) < 1e-4
    m.close()


def test_full_size_dreg_invariants(big):
    """BASELINE configs[3] at its full size (B = 1024, k = 50, DReG step of tasks/task02.py:87-101) through size-independent
    properties: the decoder is trained on -iwae_elbo (:95-96), so its gradient must equal the iwae_elbo step's decoder gradient
    on the same noise; the encoder gradient differs (inference_loss); the step is deterministic; the two half-batch gradients
    average to the full-batch one; the forward dict obeys the log_w identity and softmax normalisation."""
    m, x = big
    k = 50
    nenc = sum(n for name, shape, off in m.tensor_table() if name.startswith("enc") for n in [int(np.prod(shape))])
    m.set_step(31, 0)
    r = m.forward_backward(x, k, 1.0, "dreg", want=("al", "lpxz", "lpz", "lqzx", "log_w"))
    gd = m.get_grads().astype(np.float64)
    np.testing.assert_allclose(r["al"].sum(0), 1.0, atol=1e-5)
    np.testing.assert_allclose(r["log_w"], r["lpxz"] + r["lpz"] - r["lqzx"], rtol=0, atol=1e-3)
    lme = np.log(np.mean(np.exp(r["log_w"].astype(np.float64) - r["log_w"].max(0)), 0)) + r["log_w"].max(0)
    assert abs(lme.mean() - r["iwae_elbo"]) < 2e-3
    # tasks/task02.py:70-76 on the device's own tensors: inference_loss = -mean_b sum_s al^2 * (lpz + lpxz - lq_stopgrad)
    al = r["al"].astype(np.float64)
    lq_sg = r["lqzx"]       # sigma + 1e-6 vs sigma: below the 1e-3 the comparison allows
    il = -np.mean(np.sum(al * al * (r["lpz"] + r["lpxz"] - lq_sg), 0))
    assert abs(il - r["inference_loss"]) < 2e-3 * abs(il) + 2e-2, (il, r["inference_loss"])
    m.set_step(31, 0)
    m.forward_backward(x, k, 1.0, "dreg")
    np.testing.assert_array_equal(gd, m.get_grads().astype(np.float64))             # deterministic
    m.set_step(31, 0)
    m.forward_backward(x, k, 1.0, "iwae_elbo")
    gi = m.get_grads().astype(np.float64)
    dec_d, dec_i = gd[nenc:], gi[nenc:]
    assert np.linalg.norm(dec_d - dec_i) / np.linalg.norm(dec_i) < 2e-3             # decoder <- -iwae_elbo in both
    assert np.linalg.norm(gd[:nenc] - gi[:nenc]) / np.linalg.norm(gi[:nenc]) > 1e-2  # encoder <- inference_loss: a different estimator
    assert np.all(np.isfinite(gd)) and np.linalg.norm(gd[:nenc]) > 0
    halves = []
    for h in range(2):
        m.set_step(31, 512 * h)
        m.forward_backward(x[512 * h:512 * (h + 1)], k, 1.0, "dreg")
        halves.append(m.get_grads().astype(np.float64))
    gavg = 0.5 * (halves[0] + halves[1])
    assert np.linalg.norm(gavg - gd) / np.linalg.norm(gd) < 1e-4
    # the fused train step applies exactly this gradient
    p0 = m.get_params().copy()
    mo, ve, ts = m.get_adam_state()
    m.set_step(31, 0)
    m.train_step(x, k, 1.0, 1e-3, "dreg", scalars=False)
    np.testing.assert_array_equal(m.get_grads().astype(np.float64), gd)
    ref, _, _ = O.adam_update(p0.astype(np.float64), gd, mo.astype(np.float64), ve.astype(np.float64), ts + 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6
    m.set_params(p0); m.set_adam_state(mo, ve, ts)


@pytest.fixture(scope="module")
def big2(gpu):
    m = _model(2, [200, 100], [100, 50])
    P = O.init_params(2, [200, 100], [100, 50], 321, x_mean=O.synthetic_pixel_means())
    m.set_params(O.flatten_params(P))
    x = O.synthetic_binarized(1024, 19)
    yield m, x
    m.close()


def test_full_size_two_layer_invariants(big2):
    """BASELINE configs[2] at its full size (2-layer model, B = 1024, k = 50; src/iwae2.py:109-178) through size-independent
    properties: log_w is the five-term sum of :128, al the softmax over k, iwae_elbo the logmeanexp of the device's log_w,
    Jensen L_k >= L_1, determinism, the half-batch / full-batch gradient identity of the data-parallel step, and the fused
    train step applying exactly the gradient iwae_forward_backward leaves."""
    m, x = big2
    k = 50
    m.set_step(41, 0)
    r = m.forward_backward(x, k, 1.0, "iwae_elbo", want=("al", "lpxz", "lpz", "lqzx", "lpz2", "lqzx2", "log_w"))
    g = m.get_grads().astype(np.float64)
    np.testing.assert_allclose(r["al"].sum(0), 1.0, atol=1e-5)
    np.testing.assert_allclose(r["log_w"], r["lpxz"] + r["lpz"] + r["lpz2"] - r["lqzx"] - r["lqzx2"], rtol=0, atol=2e-3)   # iwae2.py:128
    lme = np.log(np.mean(np.exp(r["log_w"].astype(np.float64) - r["log_w"].max(0)), 0)) + r["log_w"].max(0)
    assert abs(lme.mean() - r["iwae_elbo"]) < 2e-3
    assert abs(r["log_w"].astype(np.float64).mean() - r["vae_elbo"]) < 2e-3          # iwae2.py:131
    assert r["iwae_elbo"] >= r["vae_elbo"] - 1e-3
    m.set_step(41, 0)
    m.forward_backward(x, k, 1.0, "iwae_elbo")
    np.testing.assert_array_equal(g, m.get_grads().astype(np.float64))
    assert np.all(np.isfinite(g)) and np.linalg.norm(g) > 0
    for _, shape, off in m.tensor_table():       # every one of the 26 tensors receives a gradient
        assert np.linalg.norm(g[off:off + int(np.prod(shape))]) > 0
    halves = []
    for h in range(2):
        m.set_step(41, 512 * h)
        rh = m.forward_backward(x[512 * h:512 * (h + 1)], k, 1.0, "iwae_elbo", want=("lpxz",))
        np.testing.assert_allclose(rh["lpxz"], r["lpxz"][:, 512 * h:512 * (h + 1)], atol=1e-3)
        halves.append(m.get_grads().astype(np.float64))
    gavg = 0.5 * (halves[0] + halves[1])
    assert np.linalg.norm(gavg - g) / np.linalg.norm(g) < 1e-4
    p0 = m.get_params().copy()
    m.set_step(41, 0)
    m.train_step(x, k, 1.0, 1e-3, "iwae_elbo", scalars=False)
    np.testing.assert_array_equal(m.get_grads().astype(np.float64), g)
    ref, _, _ = O.adam_update(p0.astype(np.float64), g, 0.0, 0.0, 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6


@pytest.mark.parametrize("cfg", ["c1", "c2", "c3"])
def test_headline_size_step_matches_oracle(gpu, cfg):
    """BASELINE configs[1] / [2] / [3] AT their full size -- B = 1 024, k = 50: 51 200 data rows; 1-layer iwae_elbo, the 2-layer model,
    the 1-layer DReG step -- on the path bench.py times (the device's own noise, drawn ahead; z made inside the decoder kernel; the
    one-launch decoder, dec_bwd_kernel, the specialised-wave weight gradients, the deferred decoder update), against the ORACLE fed the
    same Philox draws with the same bf16 rounding points: the objective values, per-row densities (at the device's own encoder head,
    every row; vs the pure oracle, the typical row), every gradient tensor, and one fused Adam step.  (Round 3 checked invariants only
    at this size: VERDICT missing #2.)  src/iwae1.py:98-162, src/iwae2.py:109-178, tasks/task02.py:34-101."""
    B, k, step = 1024, 50, 6
    two = cfg == "c2"
    obj = "dreg" if cfg == "c3" else "iwae_elbo"
    nh, nl = ([200, 100], [100, 50]) if two else (200, 100)
    x = O.synthetic_binarized(B, 77)
    P = O.init_params(2 if two else 1, nh, nl, 31, x_mean=O.synthetic_pixel_means())
    m = _model(2 if two else 1, nh, nl)
    m.set_params(O.flatten_params(P))
    m.set_step(step, 0)
    if two:
        e1 = philox_np.device_eps(123, step, B, k, nl[0], stream=0)
        e2 = philox_np.device_eps(123, step, B, k, nl[1], stream=1)
        res_e, g_e = O.loss_grads_2layer(P, x, e1, e2, 1.0, obj, rnd=O.bf16_round)
        r = m.forward_backward(x, k, 1.0, obj, want=("lpxz", "lpz2", "lqzx", "lqzx2", "lpz"))
        at = _densities_at_device_heads_2layer(m, e1, e2, B, k, nl)
        for a, b in (("lpz", "lpz1z2"), ("lpz2", "lpz2"), ("lqzx", "lqz1x"), ("lqzx2", "lqz2z1")):
            assert np.max(np.abs(r[a] - at[b])) < 2e-2, b
            assert np.quantile(np.abs(r[a] - res_e[b]), 0.9) < 0.05, b
        assert np.quantile(np.abs(r["lpxz"] - res_e["lpxz1"]), 0.98) < EMU_ROW_ATOL
        keys = ("vae_elbo", "iwae_elbo", "iwae_eq14")
    else:
        e = philox_np.device_eps(123, step, B, k, nl)
        res_e, g_e = O.loss_grads_1layer(P, x, e, 1.0, obj, rnd=O.bf16_round)
        r = m.forward_backward(x, k, 1.0, obj, want=("lpxz", "lpz", "lqzx"))
        at = _densities_at_device_head(m, P, x, e, nl)
        for key in ("lpxz", "lpz", "lqzx"):
            assert np.max(np.abs(r[key] - at[key])) < EMU_ROW_ATOL, key
            assert np.quantile(np.abs(r[key] - res_e[key]), 0.98) < EMU_ROW_ATOL, key
        keys = ("iwae_elbo",) if obj == "dreg" else ("vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl")
        if obj == "dreg":
            assert abs(r["inference_loss"] - res_e["inference_loss"]) < 5e-3 * abs(res_e["inference_loss"]) + 0.05
    for key in keys:
        assert abs(r[key] - res_e[key]) < EMU_SCALAR_ATOL, (key, r[key], res_e[key])
    g = m.get_grads()
    errs = _grad_rel_errors(g, g_e)
    assert max(errs) < EMU_GRAD_REL, errs
    off = 0
    for dW, db in g_e:      # elementwise: |d| <= 3 % of the tensor's largest element
        for t in (dW, db):
            got = g[off:off + t.size].reshape(t.shape).astype(np.float64)
            off += t.size
            assert np.max(np.abs(got - t)) <= 3e-2 * np.max(np.abs(t)) + 1e-9
    # the fused step (Adam inside the slab reduction, the decoder's share deferred to the side stream) applies exactly that gradient
    p0 = m.get_params().astype(np.float64)
    m.set_step(step, 0)
    m.train_step(x, k, 1.0, 1e-3, obj, scalars=False)
    np.testing.assert_array_equal(m.get_grads(), g)
    ref, _, _ = O.adam_update(p0, g.astype(np.float64), 0.0, 0.0, 1, 1e-3)
    assert np.max(np.abs(m.get_params() - ref)) < 2e-6
    m.close()


def _jitter(X, rng):
    out = np.empty_like(X)
    for i in range(X.shape[0]):
        out[i] = np.roll(np.roll(X[i].reshape(28, 28), rng.integers(-4, 5), 0), rng.integers(-4, 5), 1).reshape(-1)
    return out


@pytest.mark.parametrize("kind", ["iwae1", "dreg", "iwae2"])
def test_trained_model_k5000_llh_within_north_star_tolerance(gpu, kind):
    """north_star's stated tolerance: the test-set LLH at k = 5000 (main.py:170-184) within +-0.1 nat of the reference
    arithmetic.  A model is trained for 1 200 steps (B = 100, k = 50) on synthetic MNIST-like data with the device pipeline
    (resident dataset, per-epoch binarisation, fused step), then iwae_eval_llh(k = 5000) on 16 test images is compared with the
    exact float64 oracle evaluating the SAME weights on the SAME noise (the device's Philox stream restated in NumPy,
    oracle/philox_np.py; the 2-layer model draws z1 from stream 0 and z2 from stream 1): |difference of the means| <= 0.1 nat, and per
    image <= 0.1 nat.  kind: the 1-layer model trained on iwae_elbo (README.md:14-16), the same model trained with the DReG estimator
    (tasks/task02.py, README.md:44-46), the 2-layer model of src/iwae2.py (README.md:21-23) -- both evaluators each."""
    from iwae_amd import iwae1, iwae2, task02, utils
    from iwae_amd.optimizers import Adam
    np.random.seed(123)
    rng = np.random.default_rng(0)
    Xtrain, Xtest = utils.synthetic_mnist(20000, 256)
    Xtrain, Xtest = _jitter(Xtrain, rng), _jitter(Xtest, rng)
    layers = 2 if kind == "iwae2" else 1
    nh, nl = ([200, 100], [100, 50]) if layers == 2 else (200, 100)
    if kind == "iwae2":
        model = iwae2.IWAE(nh, nl, output_bias=utils.get_bias(Xtrain))
    elif kind == "dreg":
        model = task02.IWAEDReG(nh, nl, output_bias=utils.get_bias(Xtrain))
    else:
        model = iwae1.IWAE(nh, nl, output_bias=utils.get_bias(Xtrain))
    opt = Adam(1e-3, epsilon=1e-4)
    model.set_dataset(Xtrain)
    B, k = 100, 50
    first = last = None
    for epoch in range(6):
        model.begin_epoch(epoch, np.random.permutation(Xtrain.shape[0]))
        for lo in range(0, Xtrain.shape[0], B):
            res = model.train_step_dataset(lo, B, k, 1.0, opt, objective="dreg" if kind == "dreg" else "iwae_elbo")
            first = float(res["iwae_elbo"]) if first is None else first
        last = float(res["iwae_elbo"])
    assert last > first + 20.0, (first, last)            # it did train
    Xt = utils.bernoullisample(Xtest)
    net = model._net
    n = 16
    net.set_step(999, 0)
    llh_dev, per = net.eval_llh(Xt[:n], 5000, chunk=n, per_image=True)        # default arithmetic of the evaluator: float32
    net.set_eval_precision("bf16")
    net.set_step(999, 0)
    _, per_bf = net.eval_llh(Xt[:n], 5000, chunk=n, per_image=True)            # the fast path (bf16 GEMM operands)
    net.set_eval_precision("fp32")
    P = O.unflatten_params(net.get_params().astype(np.float64), layers, nh, nl)
    if layers == 1:
        per_o = np.array([float(O.forward_1layer(P, Xt[i:i + 1], philox_np.device_eps(123, 999, 1, 5000, 100, batch_offset=i))["iwae_elbo"])
                          for i in range(n)])
    else:
        per_o = np.array([float(O.forward_2layer(P, Xt[i:i + 1], philox_np.device_eps(123, 999, 1, 5000, 100, stream=0, batch_offset=i),
                                                 philox_np.device_eps(123, 999, 1, 5000, 50, stream=1, batch_offset=i))["iwae_elbo"]) for i in range(n)])
    d_mean, d_max = abs(per.mean() - per_o.mean()), np.max(np.abs(per - per_o))
    print("%s k=5000 LLH: device %.4f, exact fp64 oracle %.4f, |mean diff| %.4f, max per-image |diff| %.4f (trained %.2f -> %.2f)"
          % (kind, per.mean(), per_o.mean(), d_mean, d_max, first, last))
    d_bf_mean, d_bf_max = abs(per_bf.mean() - per_o.mean()), np.max(np.abs(per_bf - per_o))
    print("          bf16 evaluator: |mean diff| %.4f, max per-image |diff| %.4f" % (d_bf_mean, d_bf_max))
    assert abs(llh_dev - per.mean()) < 1e-3
    assert d_mean <= 0.1 and d_max <= 0.1, (d_mean, d_max)                 # north_star: +-0.1 nat
    assert d_mean <= 2e-3 and d_max <= 5e-3, (d_mean, d_max)               # ... and float32 arithmetic is two orders inside it
    assert d_bf_mean <= 0.1 and d_bf_max <= 0.1, (d_bf_mean, d_bf_max)     # the bf16 evaluator also stays inside the budget


def _digits_28():
    """sklearn.datasets.load_digits (1 797 real 8 x 8 digits, bundled offline: SURVEY 8c's real-data sanity set) as 28 x 28 grey images."""
    from sklearn.datasets import load_digits
    d = load_digits().images.astype(np.float64) / 16.0
    out = np.zeros((d.shape[0], 28, 28))
    out[:, 2:26, 2:26] = np.kron(d, np.ones((1, 3, 3)))
    return out.reshape(d.shape[0], 784)


@pytest.mark.parametrize("data", ["synthetic", "digits"])
def test_bf16_training_reaches_the_float32_trained_llh(gpu, data):
    """Training-level evidence for the bf16 headline (missing #3; the reference trains in float32, src/iwae1.py:31-34,
    main.py:114-184): the same model from the same initial weights on the same batches and the same device noise, trained with bf16 GEMM
    operands and in float32 mode, then the k = 5000 test log-likelihood of each trained model on held-out images, float32 evaluator.
    synthetic (20 000 blob images, 2 000 steps): the two trajectories stay together -- same-seed difference <= 0.1 nat (measured 0.0004;
    other noise seeds move a float32 run by 0.04-0.05).
    digits (1 500 real digits, 3 000 steps, held-out 297): training is chaotic at this size -- float32 runs that differ only in the noise
    seed end 0.1-0.5 nat apart (measured: -152.75 / -152.23 / -152.64), so a single pair says nothing.  Statement tested: over S = 6 noise seeds
    each, the bf16 runs' mean LLH equals the float32 runs' mean within max(0.1 nat, 2.5 standard errors of the difference), and no bf16 run
    lies outside the float32 runs' range by more than their spread."""
    from iwae_amd import iwae1, utils
    from iwae_amd.optimizers import Adam
    if data == "digits":
        X = _digits_28()
        perm = np.random.RandomState(0).permutation(X.shape[0])
        Xtr, Xte = X[perm[:1500]], X[perm[1500:]]
        steps, seeds = 3000, (123, 124, 125, 126, 127, 128)
    else:
        Xtr, Xte = utils.synthetic_mnist(20000, 256)
        steps, seeds = 2000, (123,)
    np.random.seed(3)
    Xte_bin = utils.bernoullisample(Xte)
    B, k = 100, 50
    P0 = [None]

    def run(precision, seed):
        model = iwae1.IWAE(200, 100, output_bias=utils.get_bias(Xtr), precision=precision, seed=seed)
        if P0[0] is None:
            P0[0] = model._net.get_params().copy()
        model._net.set_params(P0[0])
        opt = Adam(1e-3, epsilon=1e-4)
        model.set_dataset(Xtr)
        rs = np.random.RandomState(5)
        step, epoch, n = 0, 0, (Xtr.shape[0] // B) * B
        while step < steps:
            model.begin_epoch(epoch, rs.permutation(Xtr.shape[0]))
            for lo in range(0, n, B):
                model.train_step_dataset(lo, B, k, 1.0, opt, objective="iwae_elbo")
                step += 1
                if step >= steps:
                    break
            epoch += 1
        net = model._net
        net.set_eval_precision("fp32")
        net.set_step(999, 0)
        llh = net.eval_llh(Xte_bin, 5000, chunk=64)
        net.close()
        return float(llh)

    bf = np.array([run("bf16", s) for s in seeds])
    f32 = np.array([run("fp32", s) for s in seeds])
    print("%s: k=5000 LLH after %d steps -- bf16-trained %s, float32-trained %s" % (data, steps, np.round(bf, 3), np.round(f32, 3)))
    if len(seeds) == 1:
        assert abs(bf[0] - f32[0]) <= 0.1, (bf, f32)
    else:
        S = len(seeds)
        se = np.sqrt(bf.var(ddof=1) / S + f32.var(ddof=1) / S)
        d = abs(bf.mean() - f32.mean())
        spread = f32.max() - f32.min()
        print("   mean difference %.3f nat, standard error %.3f, float32 seed-to-seed spread %.3f" % (d, se, spread))
        assert d <= max(0.1, 2.5 * se), (d, se)
        assert bf.min() >= f32.min() - spread and bf.max() <= f32.max() + spread, (bf, f32)


@pytest.mark.parametrize("nh,nl,xd,B,k,steps,lr", [(16, 4, 48, 16, 5, 20, 1e-2), (200, 100, 784, 20, 5, 8, 1e-3), (200, 100, 784, 20, 1, 8, 1e-3),
                                                  (200, 100, 784, 170, 50, 5, 1e-3)])
def test_training_reduces_loss_and_matches_oracle_trajectory(gpu, nh, nl, xd, B, k, steps, lr):
    """Adam steps with explicit noise: the device tracks the oracle's trajectory -- every step's objective is computed from the weights the PREVIOUS
    step's epilogue wrote into the bf16 images (a misplaced image chunk would show here), the parameters at the end against the oracle's.
    Round 4: the reference's dims at B = 20 with k = 5 and k = 1 (wgrad_rows_kernel updates all seven layers and writes forward, MG-major and K-major
    backward images; log-mean-exp and latent sums inside the backward kernels) and at 8 500 rows (that kernel for the encoder, the deferred slab
    reduction for the decoder)."""
    x, P, _ = MG.inputs(1, nh, nl, xd, B, k, 77)
    m = _model(1, nh, nl, xd)
    m.set_params(O.flatten_params(P))
    flat = O.flatten_params(P)
    mo = vo = 0.0
    rng = np.random.default_rng(5)
    first = last = None
    exact = nh == 16
    for t in range(1, steps + 1):
        eps = rng.standard_normal((k, B, nl)).astype(np.float32)
        r = m.train_step(x, k, 1.0, lr, "iwae_elbo", eps=eps)
        Pt = O.unflatten_params(flat, 1, nh, nl, xd)
        res, g = O.loss_grads_1layer(Pt, x, eps, 1.0, "iwae_elbo", rnd=None if exact else O.bf16_round)
        flat, mo, vo = O.adam_update(flat, O.flatten_grads(g), mo, vo, t, lr)
        first = r["iwae_elbo"] if first is None else first
        last = r["iwae_elbo"]
        assert abs(r["iwae_elbo"] - res["iwae_elbo"]) < 0.05, (t, r["iwae_elbo"], res["iwae_elbo"])
    assert last > first + (0.5 if exact else 0.2)
    # Adam's step is ~lr whatever the gradient's size: an element whose gradient is near zero may take another sign on the device -- at most one lr per step
    d = np.abs(m.get_params() - flat)
    assert d.max() < (5e-3 if exact else 2.0 * lr * steps) and np.mean(d) < (1e-3 if exact else 0.05 * lr * steps), (d.max(), np.mean(d))
    m.close()


@pytest.mark.parametrize("layers,obj,steps", [(1, "iwae_elbo", 400), (1, "dreg", 150), (2, "iwae_elbo", 150)])
def test_full_size_training_is_reproducible_run_to_run(gpu, layers, obj, steps):
    """The full-size step (B = 1 024, k = 50: BASELINE configs[1..3]) runs its kernels on three streams and sums every weight gradient from
    fp32 slabs in a fixed order -- no atomics anywhere -- so a training run on the device's own counter-based noise must be BITWISE
    reproducible: two models stepped the same number of times from the same seed end on identical parameters and Adam moments.  A missing
    order between two streams (a weight image rewritten while its last reader runs, a row weight read before it is written) shows up here
    as a difference; NaNs or a rising loss would too."""
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    B, k = 1024, 50
    x = O.synthetic_binarized(B, 11)

    def run():
        m = _model(layers, nh, nl)
        first = m.train_step(x, k, 1.0, 1e-3, obj)["iwae_elbo"]
        for _ in range(steps - 2):
            m.train_step(x, k, 1.0, 1e-3, obj, scalars=False)
        last = m.train_step(x, k, 1.0, 1e-3, obj)["iwae_elbo"]
        p = m.get_params()
        mo, ve, t = m.get_adam_state()
        m.close()
        return first, last, p, mo, ve, t

    f0, l0, p0, mo0, ve0, t0 = run()
    f1, l1, p1, mo1, ve1, t1 = run()
    assert np.all(np.isfinite(p0)) and np.isfinite(l0)
    assert l0 > f0 + 20.0, (f0, l0)          # (random init: about -545 nat; a few hundred steps on one batch gain > 100)
    assert t0 == t1 == steps
    assert f0 == f1 and l0 == l1
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(mo0, mo1)
    np.testing.assert_array_equal(ve0, ve1)


def test_bad_arguments_fail_loudly(gpu):
    from iwae_amd.native import NativeModel
    with pytest.raises(ValueError):
        NativeModel(3, 200, 100)
    with pytest.raises(ValueError):
        NativeModel(1, 300, 100)
    m = _model(1, 200, 100)
    with pytest.raises(ValueError):
        m.set_params(np.zeros(10, dtype=np.float32))
    m.close()


def test_dataset_gather_binarize_is_bit_exact_and_equivalent(gpu):
    """Device data pipeline (main.py:117-120, src/utils.py:26-27): gather by the epoch's order + dynamic
    binarisation, byte/integer work -> bit-exact against the NumPy restatement; and a train step fed from the
    resident dataset equals a train step fed the same batch through the host path."""
    rng = np.random.default_rng(3)
    N = 300
    gray = (rng.random((N, 784)) * 256).astype(np.uint8)
    gray[:, :40] = 0
    gray[:, 40:80] = 255
    m = _model(1, 200, 100)
    P = O.init_params(1, 200, 100, 5, x_mean=O.synthetic_pixel_means())
    m.set_params(O.flatten_params(P))
    m.dataset_upload(gray)
    order = rng.permutation(N).astype(np.int32)
    for epoch in (0, 7):
        m.dataset_begin_epoch(epoch, order)
        xb = m.dataset_get_batch(37, 150)                       # ragged: not a multiple of 64/128
        ref = philox_np.device_binarize(123, epoch, gray, order[37:187])
        np.testing.assert_array_equal(xb, ref)
    assert xb[:, :40].sum() == 0 and xb[:, 40:80].min() == 1
    # equivalence of the two input paths (same noise counters)
    m.dataset_begin_epoch(7, order)
    m.set_step(5, 0)
    a = m.train_step_dataset(37, 150, 5, 1.0, 1e-3, "iwae_elbo")
    pa = m.get_params()
    m.set_params(O.flatten_params(P)); m.set_adam_state(np.zeros(m.n_params), np.zeros(m.n_params), 0)
    m.set_step(5, 0)
    b = m.train_step(ref, 5, 1.0, 1e-3, "iwae_elbo")
    assert a["iwae_elbo"] == b["iwae_elbo"]
    np.testing.assert_array_equal(pa, m.get_params())
    with pytest.raises(ValueError):
        m.train_step_dataset(200, 150, 5)                       # range outside the dataset
    m.close()


@pytest.mark.parametrize("prior,precision", [(False, "bf16"), (True, "bf16"), (False, "fp32")])
def test_labelled_dataset_feeds_the_conditional_models(gpu, prior, precision):
    """tasks/task05.py:296-322 / tasks/task04.py train on (x, y) batches of a labelled set.  Round 5: the resident dataset carries one class id per
    image (iwae_dataset_set_labels) and the input kernel emits onehot(y) of the batch's images where a host-fed step takes iwae_set_condition's
    rows -- integer / byte work: the gathered one-hot rows and pixels are bit-exact against the NumPy restatement, and a train step fed from the
    resident set lands bitwise on the parameters of the same batch fed through the host path (x, onehot(y)); the float32 mode and the learned
    prior p(z|y) included; errors (no labels, a label >= cond_dim, an unconditional model) fail loudly."""
    from iwae_amd.native import NativeModel
    from iwae_amd import task05
    rng = np.random.default_rng(11)
    N, B, k = 260, 150, 5
    gray = (rng.random((N, 784)) * 256).astype(np.uint8)
    labels = rng.integers(0, 10, N).astype(np.uint8)
    P = O.init_params(1, 200, 100, 5, x_mean=O.synthetic_pixel_means(), cond_dim=10, cond_prior=prior)
    m = NativeModel(1, 200, 100, seed=123, cond_dim=10, cond_prior=prior, precision=precision)
    m.set_params(O.flatten_params(P))
    m.dataset_upload(gray)
    order = rng.permutation(N).astype(np.int32)
    m.dataset_begin_epoch(3, order)
    with pytest.raises(RuntimeError):
        m.train_step_dataset(37, B, k)                             # a conditional model without labels
    bad = labels.copy(); bad[17] = 10
    with pytest.raises(ValueError):
        m.dataset_set_labels(bad)                                  # not below cond_dim
    with pytest.raises(ValueError):
        m.dataset_set_labels(labels[:-1])                          # one label per image
    m.dataset_set_labels(labels)
    yb = m.dataset_get_labels(37, B)
    np.testing.assert_array_equal(yb, task05.one_hot(labels[order[37:37 + B]]))
    xb = m.dataset_get_batch(37, B)
    np.testing.assert_array_equal(xb, philox_np.device_binarize(123, 3, gray, order[37:37 + B]))
    m.set_step(5, 0)
    a = m.train_step_dataset(37, B, k, 1.0, 1e-3, "iwae_elbo")
    pa, ga = m.get_params().copy(), m.get_grads().copy()
    m.set_params(O.flatten_params(P)); m.set_adam_state(np.zeros(m.n_params), np.zeros(m.n_params), 0)
    m.set_step(5, 0)
    m.set_condition(yb)
    b = m.train_step(xb, k, 1.0, 1e-3, "iwae_elbo")
    assert a["iwae_elbo"] == b["iwae_elbo"]
    np.testing.assert_array_equal(ga, m.get_grads())
    np.testing.assert_array_equal(pa, m.get_params())
    m.close()
    m0 = _model(1, 200, 100)
    m0.dataset_upload(gray)
    with pytest.raises(RuntimeError):
        m0.dataset_set_labels(labels)                              # cond_dim = 0
    m0.close()
