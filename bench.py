"""bench.py -- images/sec of the IWAE train step (forward + backward + Adam [+ RCCL all-reduce]).

  python bench.py --gpus 1 --steps 100 --warmup 10 [--config c1]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Default workload (BASELINE.json configs[1], --config c1): 1-layer IWAE, k=50, batch 1024 PER GPU (weak scaling; N=8
is configs[4], global batch 8192), bf16 GEMM operands with fp32 accumulation, objective iwae_elbo, synthetic binarised
28x28 images already resident in HBM, Keras-style random-init weights.  The other BASELINE configs are selectable
(--config c0 | c2 | c3: the reference's default regime B=20,k=1; the 2-layer model; the DReG step) and print the same
line for their own workload.  Rank 0 prints ONE JSON line.  The oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

X_DIM = 784
TIMING_EVERY = 2             # the per-kernel pass AFTER the timed region: HIP events bracket the timed kernels on every 2nd step
TIMING_STEPS = 48            # steps of that pass (>= 16 launches per kernel; an event record costs a few us of stream bubble,
                             # so none of it happens inside the timed region)
SETTLE_STEPS = 100           # untimed steps run in any case before the timed region (--warmup if that is larger): clocks ramped, the noise
                             # ring primed, every buffer at its final size.  A fixed COUNT: every rank must issue the same collectives
PEAK_HBM_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md); a plain 1 GiB device copy reaches 4.8 TB/s on these boxes (profiles/r03_copy_yardstick.txt)
PEAK_BF16_TFLOPS = 2500.0    # dense bf16 MFMA peak (same guide)
PEAK_FP32_TFLOPS = 157.3     # v_mfma_f32_16x16x4_f32 (precision fp32)
RIDGE = PEAK_BF16_TFLOPS * 1e12 / (PEAK_HBM_GBS * 1e9)      # 312 FLOP/B

CONFIGS = {
    # BASELINE.json configs[0..3]; B is per GPU
    "c0": dict(layers=1, B=20, k=1, obj="vae_elbo", nh=200, nl=100,
               label="1-layer IWAE k=1 (VAE ELBO) batch 20 (BASELINE configs[0], the reference's default regime)"),
    "c1": dict(layers=1, B=1024, k=50, obj="iwae_elbo", nh=200, nl=100,
               label="1-layer IWAE k=50 batch 1024 per GPU (BASELINE configs[1]; N=8 is configs[4])"),
    "c2": dict(layers=2, B=1024, k=50, obj="iwae_elbo", nh=[200, 100], nl=[100, 50],
               label="2-layer IWAE k=50 batch 1024 (BASELINE configs[2], src/iwae2.py)"),
    "c3": dict(layers=1, B=1024, k=50, obj="dreg", nh=200, nl=100,
               label="1-layer IWAE k=50 batch 1024 with the DReG estimator (BASELINE configs[3], tasks/task02.py)"),
}
OBJ_ID = {"vae_elbo": 0, "iwae_elbo": 1, "iwae_eq14": 2, "vae_elbo_kl": 3, "dreg": 4}


def step_flops(cfg):
    """Algorithmic GEMM FLOPs of one train step: fwd = encoder on B images + per-sample work on B*k rows,
    bwd = 2x fwd minus the dX of the first encoder layer.  1-layer: 433 600 FLOP per sample, 2-layer: 563 600."""
    per_sample = 433600 if cfg["layers"] == 1 else 563600
    return 3 * (473600 * cfg["B"] + per_sample * cfg["B"] * cfg["k"]) - 313600 * cfg["B"]


def step_algorithmic_bytes(cfg):
    """compulsory bytes of a step (x in, parameters + Adam state read and written) + save-for-backward
    traffic if the per-sample activations (1-layer: z, g1, g2 as bf16) are written once and read once."""
    B, M = cfg["B"], cfg["B"] * cfg["k"]
    nparam = 455384 if cfg["layers"] == 1 else 521084
    compulsory = 4 * X_DIM * B + 7 * 4 * nparam
    if cfg["layers"] == 1:
        acts = 2 * M * (2 * 100 + 2 * 200 + 2 * 200)
    else:   # z1, z2, the two per-sample blocks' h1, h2 (100 wide) and the decoder's g1, g2
        acts = 2 * M * (2 * 100 + 2 * 50 + 4 * 2 * 100 + 2 * 200 + 2 * 200)
    return compulsory, acts


def kernel_models(cfg):
    """Timed kernels of the step (names: include/iwae_amd.h, iwae_kernel_time): what each is, its ALGORITHMIC bytes per launch
    (unpadded operands read once + results written once) and GEMM FLOPs, and the substring of its
    device-side name (to pair with rocprofv3 rows).  D, H, X: latent, hidden, pixels of the z -> x decoder; M data rows."""
    M, k, D, H, X = cfg["B"] * cfg["k"], cfg["k"], 100, 200, X_DIM
    fused_z = cfg["layers"] == 1 and cfg["obj"] != "dreg" and M >= 8192      # the decoder kernel makes z itself from the draws
    zin = 4 * D + (8.0 * D + 2.0 * X) / k + 2 * D + 8 if fused_z else 2 * D + 2.0 * X / k      # draws + head + x in, z + 2 densities out | z in
    big = M >= 8192
    out = {
        "decoder_fwd": dict(
            name=("bern_pipe_kernel<7,true,true,*> (whole decoder forward in one launch: " if big else
                  "block_fwd_kernel<6,true> (few rows: whole decoder forward in one launch: " if M <= 4096 else "dense_kernel<EPI_BERN> (output layer: ") +
                 ("z = mu + sigma*eps, " if (fused_z or M <= 4096) else "") + ("two tanh layers, " if (big or M <= 4096) else "") +
                 "Bernoulli log-likelihood, keeps s = x - sigmoid(l))",
            bytes=M * (zin + (4 * H if (big or M <= 4096) else 2 * H) + 2 * X + 4),
            flop=2 * M * ((D * H + H * H if (big or M <= 4096) else 0) + H * X),
            match="bern_pipe_kernel" if big else "block_fwd_kernel<6, true" if M <= 4096 else "dense_kernel<4"),
        "out_bwd": dict(name="out_bwd_s_kernel<7> (output-layer backward from the stored s: dg2 = s W^T, dpre2)",
                        bytes=M * (2 * X + 2 * H + 4 + 2 * H), flop=2 * M * H * X, match="out_bwd_s_kernel"),
        "decoder_bwd": dict(name=("dec_bwd_rows_kernel (<= 1 024 rows: 16-row workgroups, weights straight from L2)" if M <= 1024 else "dec_bwd_kernel<7,8,1> (8 waves x 16 rows)") +
                                 " (decoder dX chain in one launch: dg2 = s W3^T -> dpre2 -> dpre1 -> dz; s, g2, g1 in, dpre2, dpre1, dz out)",
                            bytes=M * (2 * X + 2 * H + 4 + 2 * H + 2 * H + 2 * H + (2 if cfg["layers"] == 1 else 4) * D), flop=2 * M * (H * X + H * H + H * D),
                            match="dec_bwd_rows_kernel" if M <= 1024 else "dec_bwd_kernel"),
        "wgrad_out": dict(name="wgradws_kernel<true,4,4> (output-layer weight gradient dV3 = g2^T (g_r s): 8 compute + 4 loader waves, s weighted on its way through the loaders' registers, side stream)" if big else
                               "wgradp_kernel<8,4,4,4,true> (output-layer weight gradient, side stream)",
                          bytes=M * (2 * H + 2 * X + 4) + 4 * (H * X + X), flop=2 * M * H * X, match="wgradws_kernel<true" if big else "wgradp_kernel<8, 4, 4, 4, true",
                          note=("runs on the side stream beside dec_bwd_kernel and the hidden layers' gradients, on 96 one-per-CU workgroups by choice (DESIGN.md section 3, items 49 "
                                "and 57: on 128 workgroups THIS kernel is faster -- 52.5 us alone instead of 66, 94 us in the step instead of 108 -- and the step slower, 0.2142 vs "
                                "0.2042 ms; the backward phase of the step moves ~580 MB at 4.2 TB/s, 88 % of the copy rate of this pool: what one kernel gains the kernels beside it lose)"
                                if big and cfg["layers"] == 1 else None)),
        "dx_hidden": dict(name="dense_kernel<EPI_DX,7> (dpre1 = (dpre2 V2^T) * (1 - g1^2))", bytes=M * 6 * H, flop=2 * M * H * H, match="dense_kernel<2, 7"),
        "dx_latent": dict(name="dense_kernel<EPI_F32,7> (dz = dpre1 V1^T, fp32)", bytes=M * (2 * H + 4 * D), flop=2 * M * H * D, match="dense_kernel<3, 7"),
        "wgrad_hidden": dict(name=("wgradws_kernel<false,4,4>" if big else "wgradp_kernel<8,4,4,4,false>") + " (dV2 = g1^T dpre2, second side stream)",
                             bytes=M * 4 * H + 4 * (H * H + H), flop=2 * M * H * H, match="wgradws_kernel<false" if big else "wgradp_kernel<8, 4, 4, 4, false"),
        "wgrad_latent": dict(name=("wgradws_kernel<false,4,4>" if big else "wgradp_kernel<8,4,4,4,false>") + " (dV1 = z^T dpre1, second side stream)",
                             bytes=M * (2 * D + 2 * H) + 4 * (D * H + H), flop=2 * M * D * H, match="wgradws_kernel<false" if big else "wgradp_kernel<8, 4, 4, 4, false"),
        "latent_bwd": dict(name="latent_bwd_kernel (d mu, d sigma per image: sum over the k samples)",
                           bytes=M * ((2 if (big and cfg["layers"] == 1) else 4) * D + 4 * D + 16), flop=0, match="latent_bwd_kernel"),
        "encoder_fwd": dict(name="block_fwd_kernel (encoder BasicBlock on the B images, one launch)",
                            bytes=cfg["B"] * (4 * X + 4 * H + 8 * D) + 2 * (X * H + H * H + 2 * H * D), flop=2 * cfg["B"] * (X * H + H * H + 2 * H * D), match="block_fwd_kernel"),
        # round 4: the main stream's last kernel is wgrad_rows_kernel -- the image encoder's three weight gradients over ALL rows, Keras Adam and the weight-image
        # refresh in its epilogue (on <= 2 048 data rows the decoder's three layers too); bytes: operands read once, theta / m / v read and written, gradient + images written
        "reduce_adam": dict(name="wgrad_rows_kernel (encoder weight gradients, whole row reduction per workgroup, + fused Adam + weight-image refresh"
                                 + (", + the decoder's three layers" if M <= 2048 and cfg["layers"] == 1 else "") + ")",
                            bytes=cfg["B"] * 2 * (X + 2 * H + 2 * H + 2 * D + H) + 30 * (X * H + H * H + 2 * H * D)
                                  + (M * 2 * (2 * H + X + H + D + 2 * H) + 30 * (D * H + H * H + H * X) if M <= 2048 and cfg["layers"] == 1 else 0),
                            flop=2 * cfg["B"] * (X * H + H * H + 2 * H * D) + (2 * M * (D * H + H * H + H * X) if M <= 2048 and cfg["layers"] == 1 else 0),
                            match="wgrad_rows_kernel"),
    }
    return out


def synthetic_batch(n, seed):
    from iwae_amd import utils
    rng = np.random.default_rng(seed)
    p = utils.synthetic_pixel_means(X_DIM)
    return (rng.random((n, X_DIM)) < p[None]).astype(np.float32), p


def host_cores():
    """Cores this job may actually use: cgroup quota / affinity, capped at 16 (a one-GPU box's CPU share;
    oversubscribing the 256 logical CPUs of the host makes the CPU leg ~50x slower, not faster)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(cfg, budget_s=12.0):
    """The same train step on the host cores (oracle/iwae_torch.py, fp32, autograd + Adam(eps=1e-4)):
    a bounded sample of the SAME workload, reported, never the target."""
    from oracle import iwae_np as O, iwae_torch as T
    threads = host_cores()
    torch.set_num_threads(threads)
    x_np, p = synthetic_batch(cfg["B"], 7)
    P = O.init_params(cfg["layers"], cfg["nh"], cfg["nl"], 123, x_mean=p)
    tr = T.CpuTrainer(P, cfg["layers"], lr=1e-3, threads=threads)
    x = torch.tensor(x_np)
    obj = "iwae_elbo" if cfg["obj"] == "dreg" else cfg["obj"]     # the DReG port's cost is the iwae_elbo step's (same graph, two gradient targets)
    tr.step(x, cfg["k"], obj)      # warm-up
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < budget_s and n < 2000):
        tr.step(x, cfg["k"], obj)
        n += 1
    dt = time.time() - t0
    return {"value": round(cfg["B"] * n / dt, 1), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps of the same workload (B=%d, k=%d, %d-layer, fp32 torch-CPU autograd + Adam) in %.1f s; "
                      "TF2 (the reference runtime) is not installed, this is the oracle port" % (n, cfg["B"], cfg["k"], cfg["layers"], dt)}


def load_profile_traffic(kernel_match, config, build_id):
    """HBM bytes per launch of the kernel whose device-side name contains `kernel_match`, from the PMC passes of
    tools/profile_bench.sh kept under profiles/ (FETCH_SIZE x2 + WRITE_SIZE: MI355X_MICROARCH.md, HBM).  Returned WITH its
    source, or None when no kept profile holds that instantiation.  The number is a profile artefact, not a live measurement, so it is
    only quoted when the profile was taken ON THIS BUILD: the table carries the iwae_build_id() of the library it measured (round 5), and
    a table of another build -- a kernel edited since, or a round-4 table, which carries no id -- gives None with the reason as source."""
    stale = None
    for tag in ("r05", "r05_c2", "r04", "r04_c2"):      # (newest kept profile of this configuration first)
        path = os.path.join(ROOT, "profiles", "%s_kernel_traffic.json" % tag)
        if not os.path.exists(path):
            continue
        try:
            tab = json.load(open(path))
        except Exception:
            continue
        if tab.get("config", "c1") != config:
            continue
        if tab.get("build_id") != build_id:
            stale = stale or "profiles/%s_kernel_traffic.json was taken on build %s, this library is %s: not quoted" % (tag, tab.get("build_id"), build_id)
            continue
        for ent in tab.get("kernels", []):
            if kernel_match in ent.get("kernel", ""):
                return ent.get("hbm_bytes_per_launch"), "profiles/%s_kernel_traffic.json: %s" % (tag, ent["kernel"]), tab.get("step_hbm_bytes")
    return None, stale, None


class _StdoutToStderr:
    """RCCL prints a version banner on STDOUT when its first communicator is made; the contract is ONE JSON line there.
    Everything before the result line runs with file descriptor 1 pointing at stderr."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False


def main():
    with _StdoutToStderr():
        out, dist = run()
    if out is not None:
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


def run():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c1", help="BASELINE.json configs[0..3] (default c1 = configs[1], the headline)")
    ap.add_argument("--precision", choices=("bf16", "fp32"), default="bf16", help="GEMM arithmetic (iwae_config.precision); the headline is bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-llh-eval", action="store_true", help="skip the untimed k = 5000 evaluator run (profiling: keeps its kernels out of the statistics)")
    ap.add_argument("--settle", type=int, default=SETTLE_STEPS, help="untimed steps in front of the timed region in any case (profiling scripts pass 0: then exactly --warmup)")
    ap.add_argument("--no-kernel-times", action="store_true", help="skip the per-kernel event pass behind the timed region (profiling)")
    ap.add_argument("--timing-every", type=int, default=TIMING_EVERY, help="the per-kernel event pass brackets the kernels of every n-th step")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="iwae_set_option switch for A/B measurements (repeatable; tools/README.md)")
    ap.add_argument("--lib", default=None, metavar="PATH", help="developer A/B only: load this build of libiwae_amd.so instead of the in-tree one (recorded in config.lib)")
    ap.add_argument("--force-dist", action="store_true", help="rehearse the data-parallel code path with ONE rank (trivial collectives)")
    ap.add_argument("--dp-torch", action="store_true", help="data-parallel exchange through torch.distributed.all_reduce instead of the library's own RCCL calls")
    args = ap.parse_args()
    force_dist = args.force_dist
    options = {}
    for kv in args.opt:
        name, _, val = kv.partition("=")
        options[name] = int(val or "1")
    cfg = CONFIGS[args.config]
    B, K = cfg["B"], cfg["k"]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torchrun with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or force_dist:      # --force-dist rehearses the RCCL path on one GPU
        import torch.distributed as dist
        if world == 1:               # (a rehearsal outside torchrun: a one-rank rendezvous of its own)
            for key, val in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29517"), ("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")):
                os.environ.setdefault(key, val)
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    if args.lib:      # (an older build may lack newer symbols: bind what it has)
        import ctypes
        from iwae_amd import _capi
        _capi.LIB_PATH = os.path.abspath(args.lib)
        probe = ctypes.CDLL(_capi.LIB_PATH)
        for name in [n for n in _capi.SYMBOLS if not hasattr(probe, n)]:
            del _capi.SYMBOLS[name]
    from iwae_amd.native import NativeModel
    from iwae_amd.parallel import DataParallelStep
    from iwae_amd import utils

    x_np, p = synthetic_batch(B * world, 123)
    lo = rank * B
    x_dev = torch.tensor(x_np[lo:lo + B], device="cuda")       # inputs resident in HBM before timing
    net = NativeModel(cfg["layers"], cfg["nh"], cfg["nl"], x_dim=X_DIM, device=local_rank, seed=123, world_size=world, rank=rank,
                      precision=args.precision, options=options)
    net.set_output_bias(utils.bias_from_mean(p))                       # identical init on every rank (same seed)
    # world > 1: the in-library RCCL exchange.  If it cannot be brought up on EVERY rank the run RAISES on every rank
    # (round 5: a line measured on the torch.distributed fallback is not the path the design describes; --dp-torch selects that path on purpose)
    dp = DataParallelStep(net, rank, world, in_library=not args.dp_torch, force_dist=force_dist, allow_fallback=False)
    rccl_ranks = net.comm_info()[0] if dp.in_library else (dist.get_world_size() if dist and dp.path == "torch_fallback" else 0)
    lr = 1e-3
    obj_id = OBJ_ID[cfg["obj"]]

    def step():
        dp.step(x_dev.data_ptr(), B, K, 1.0, lr, obj_id)

    # untimed: --warmup steps, and in any case enough steps / time for the clocks, the noise ring and the allocator to settle
    # (the driver's --warmup 5 alone left the timed region 7-11 % slower than a 200-step run of the same build)
    settle = max(args.warmup, args.settle)
    for i in range(settle):
        step()
        if i % 16 == 15:
            net.sync()
    net.sync()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    net.sync()                    # the library's own streams (a deferred decoder update may still run on its side stream)
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # per-kernel launch durations: a SEPARATE pass behind the timed region (HIP events on the stream each kernel runs on)
    models = kernel_models(cfg)
    ktimes = {k: (0.0, 0) for k in models}
    allreduce_us = None
    if not args.no_kernel_times:
        net.enable_timing(args.timing_every)
        for _ in range(TIMING_STEPS):
            step()
        net.sync()
        ktimes = {k: net.kernel_time(k) for k in models}
        if dp.in_library:      # the data-parallel step's two all-reduces (events around each ncclAllReduce on the stream that carries it)
            allreduce_us = {seg: round(net.kernel_time("allreduce_" + seg)[0], 2) for seg in ("enc", "dec")}
        net.enable_timing(0)
    elbo = net.forward(x_np[lo:lo + B], K)["iwae_elbo"]
    llh_eval = None
    if world == 1 and not args.no_llh_eval and cfg["layers"] == 1:      # the other half of BASELINE's metric: the test-LLH protocol of main.py:170-184 (k = 5000 per image), untimed extra
        n_eval = 4190      # ten full launches of the evaluator (419 images x 5000 samples = 2^21 rows each at the reference's dims)
        xe = np.tile(x_np, (n_eval // x_np.shape[0] + 1, 1))[:n_eval]
        llh_eval = {"k": 5000, "images": n_eval}
        for prec in ("fp32", "bf16"):      # the evaluator's default arithmetic (float32, as the reference's) and the fast path
            net.set_eval_precision(prec)
            net.set_step(1 << 20, 0)
            net.eval_llh(xe[:838], 5000)      # (two full-size launches: every buffer of the evaluator reaches its size here, not in the timed call)
            net.sync()
            net.set_step(1 << 20, 0)
            t1 = time.perf_counter()
            llh = net.eval_llh(xe, 5000)
            dt_eval = time.perf_counter() - t1
            llh_eval[prec] = {"images_per_s": round(n_eval / dt_eval, 1), "llh": round(float(llh), 4),
                              "gemm_tflops": round(n_eval * 2168473600.0 / dt_eval / 1e12, 1), "s_per_10000_images": round(dt_eval * 10000 / n_eval, 3)}
        net.set_eval_precision("fp32")
        llh_eval["note"] = "forward-only evaluator on synthetic images with the just-trained weights, same noise in both arithmetics"

    if rank == 0:
        ms = dt * 1e3 / args.steps
        value = B * world * args.steps / dt
        from iwae_amd import _capi
        try:
            build_id = _capi.library_build_id()
        except Exception:      # (--lib with a build older than iwae_build_id)
            build_id = "unknown"
        launched = {k: v for k, v in ktimes.items() if v[1] > 0}
        dom = max(launched, key=lambda k: launched[k][0]) if launched else None    # the kernel with the longest average launch, over ALL timed kernels
        flop_step = step_flops(cfg)
        comp_b, act_b = step_algorithmic_bytes(cfg)
        peak_tf = PEAK_FP32_TFLOPS if args.precision == "fp32" else PEAK_BF16_TFLOPS
        out = {
            "metric": "images/sec (train step) IWAE k=50 batch 1024 @1/2/4/8 GPU; test LLH@k=5000",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            # the timed region by the wall clock (barrier + synchronise on both sides) and the build the line was measured on
            "timed_region_s": round(dt, 6), "build_id": build_id,
            "config": {"workload": cfg["label"] + ", objective %s, train step = forward + backward + Adam(eps=1e-4)" % cfg["obj"]
                                   + (" + RCCL all-reduce of the flat fp32 gradient" if world > 1 else ""),
                       "config_id": args.config, "global_batch": B * world, "n_samples": K, "parallelism": "dp%d" % world,
                       # which exchange path ran, and over how many ranks RCCL itself says (ncclCommCount on the library's communicator)
                       "dp_path": dp.path, "rccl_ranks": rccl_ranks, "allreduce_us": allreduce_us, "settle_steps": settle, "options": options, "lib": args.lib,
                       "step_gemm_tflops": round(flop_step * world / (dt / args.steps) / 1e12, 1),
                       "iwae_elbo_after": round(float(elbo), 3), "llh_eval_k5000": llh_eval},
        }
        if dom:
            dom_us, dom_n = launched[dom]
            km = models[dom]
            hbm_bound = km["flop"] / max(km["bytes"], 1) < RIDGE
            traffic, tsrc, step_hbm = load_profile_traffic(km["match"], args.config, build_id)
            ach = km["bytes"] / (dom_us * 1e-6) / 1e9 if hbm_bound else km["flop"] / (dom_us * 1e-6) / 1e12
            peak = PEAK_HBM_GBS if hbm_bound else peak_tf
            out["roofline"] = {
                "bound": "hbm" if hbm_bound else "mfma", "kernel": km["name"], "timed_as": dom,
                "stream": "side" if dom.startswith("wgrad") else "main",
                "achieved": round(ach, 1), "peak": peak, "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": tsrc,
                "avg_launch_us": round(dom_us, 2), "launches": dom_n,
                "algorithmic_bytes_per_launch": int(km["bytes"]), "flop_per_launch": km["flop"],
                "mfma_tflops": round(km["flop"] / (dom_us * 1e-6) / 1e12, 1) if dom_us > 0 else 0.0,
                "note": km.get("note"),
                # every timed kernel: average launch (us), algorithmic GB/s and its fraction of the HBM peak
                "all_kernels": {k: {"us": round(v[0], 2), "GBps": round(models[k]["bytes"] / (v[0] * 1e-6) / 1e9, 1) if v[0] > 0 else 0.0,
                                    "hbm_frac": round(models[k]["bytes"] / (v[0] * 1e-6) / 1e9 / PEAK_HBM_GBS, 3) if v[0] > 0 else 0.0,
                                    "tflops": round(models[k]["flop"] / (v[0] * 1e-6) / 1e12, 1) if v[0] > 0 else 0.0}
                                for k, v in launched.items()},
                # the step as a whole (per GPU): GEMM FLOP / step time against the MFMA peak, SURVEY 8(d)'s algorithmic bytes
                # (compulsory + activations written once and read once) / step time against the HBM peak, and -- from the
                # kept profile, when it is of this configuration -- the bytes the step really moved
                "step": {"gemm_flop": flop_step, "mfma_frac": round(flop_step / (ms * 1e-3) / (peak_tf * 1e12), 4),
                         "algorithmic_bytes": int(comp_b + act_b), "algorithmic_hbm_frac": round((comp_b + act_b) / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4),
                         "measured_hbm_bytes": step_hbm, "measured_hbm_frac": round(step_hbm / (ms * 1e-3) / (PEAK_HBM_GBS * 1e9), 4) if step_hbm else None},
            }
        if not dom:      # float32 mode (one generic GEMM kernel, no per-kernel events): the step as a whole against the f32 MFMA peak
            out["roofline"] = {"bound": "mfma", "kernel": "gemm_f32_v2_kernel (every product of the step: v_mfma_f32_16x16x4_f32), whole step", "timed_as": "step",
                               "stream": "main + side (the decoder's weight gradients and its deferred update)", "achieved": round(flop_step / (ms * 1e-3) / 1e12, 2), "peak": peak_tf, "unit": "TFLOP/s",
                               "frac": round(flop_step / (ms * 1e-3) / (peak_tf * 1e12), 4), "traffic": None, "traffic_source": None,
                               "avg_launch_us": round(ms * 1e3, 2), "launches": args.steps, "flop_per_launch": flop_step}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        return out, dist
    return None, dist


if __name__ == "__main__":
    main()
