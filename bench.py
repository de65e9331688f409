"""bench.py -- images/sec of the IWAE train step (forward + backward + Adam [+ RCCL all-reduce]).

  python bench.py --gpus 1 --steps 100 --warmup 10
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): 1-layer IWAE, k=50, batch 1024 PER GPU (weak scaling; N=8 is
configs[4], global batch 8192), bf16 GEMM operands with fp32 accumulation, objective iwae_elbo,
synthetic binarised 28x28 images already resident in HBM, Keras-style random-init weights.
Rank 0 prints ONE JSON line.  The oracle is used only for the cpu_baseline leg.
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

B_PER_GPU, K_SAMPLES, N_HIDDEN, N_LATENT, X_DIM = 1024, 50, 200, 100, 784
OBJ_IWAE_ELBO = 1
# algorithmic GEMM FLOPs of one step: 3*(473600*B + 433600*B*k) - 313600*B
FLOP_PER_STEP = 3 * (473600 * B_PER_GPU + 433600 * B_PER_GPU * K_SAMPLES) - 313600 * B_PER_GPU
TIMING_EVERY = 8             # HIP events bracket the candidate kernels on every 8th step of the timed region
PEAK_HBM_GBS = 8000.0        # MI355X HBM3E peak (MI355X_MICROARCH.md); a plain device copy reaches ~4.8 TB/s
M_ROWS = B_PER_GPU * K_SAMPLES
# The two heaviest kernels of the step, both below the roofline ridge (2.5 PFLOP/s / 8 TB/s = 312 FLOP/B), i.e. HBM-bound
# by the model.  Algorithmic bytes per data row (unpadded):
#   bernoulli_fwd  (bern_pipe_kernel<7,true,true>: the WHOLE decoder forward in one launch): the step's draws in (4D) + the image's
#                  encoder head and x, shared by its k rows ((8D + 2X)/k) + z, g1, g2 out for the backward pass (2D + 2H + 2H) +
#                  s = x - sigmoid(l) out (2X) + log p(x|z), log p(z), log q(z|x) out (12);   FLOP 2(DH + HH + HX)
#   out_bwd        (out_bwd_s_kernel): s in (2X) + g2 in (2H) + row weight (4) + dpre2 out (2H);   FLOP 2HX
KERNELS = {
    "bernoulli_fwd": {"name": "bern_pipe_kernel<7,true,true,true> (whole decoder forward, 16-wave / 200-row workgroups: z = mu + sigma*eps, two tanh layers, output layer + "
                              "Bernoulli log-likelihood, keeps s = x - sigmoid(l))",
                      "bytes": M_ROWS * (4 * N_LATENT + (8.0 * N_LATENT + 2.0 * X_DIM) / K_SAMPLES + 2 * N_LATENT + 4 * N_HIDDEN + 2 * X_DIM + 12),
                      "flop": 2 * M_ROWS * (N_LATENT * N_HIDDEN + N_HIDDEN * N_HIDDEN + N_HIDDEN * X_DIM),
                      "match": "bern_pipe_kernel"},
    "out_bwd": {"name": "out_bwd_s_kernel<7> (decoder output-layer backward from the stored s)",
                "bytes": M_ROWS * (2 * X_DIM + 2 * N_HIDDEN + 4 + 2 * N_HIDDEN), "flop": 2 * M_ROWS * N_HIDDEN * X_DIM,
                "match": "out_bwd"},
}


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


def cpu_baseline(budget_s=12.0):
    """The same train step on the host cores (oracle/iwae_torch.py, fp32, autograd + Adam(eps=1e-4)):
    a bounded sample of the SAME workload (B=1024, k=50), reported, never the target."""
    from oracle import iwae_np as O, iwae_torch as T
    threads = host_cores()
    torch.set_num_threads(threads)
    x_np, p = synthetic_batch(B_PER_GPU, 7)
    P = O.init_params(1, N_HIDDEN, N_LATENT, 123, x_mean=p)
    tr = T.CpuTrainer(P, 1, lr=1e-3, threads=threads)
    x = torch.tensor(x_np)
    tr.step(x, K_SAMPLES)      # warm-up
    t0 = time.time()
    n = 0
    while n < 3 or (time.time() - t0 < budget_s and n < 200):
        tr.step(x, K_SAMPLES)
        n += 1
    dt = time.time() - t0
    return {"value": round(B_PER_GPU * n / dt, 1), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d train steps of the same workload (B=1024, k=50, 1-layer, fp32 torch-CPU autograd + Adam) in %.1f s; "
                      "TF2 (the reference runtime) is not installed, this is the oracle port" % (n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-llh-eval", action="store_true", help="skip the untimed k = 5000 evaluator run (profiling: keeps its kernels out of the statistics)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d needs torchrun with %d ranks (WORLD_SIZE=%d)" % (args.gpus, args.gpus, world))
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("IWAE_BENCH_FORCE_DIST"):      # the env switch rehearses the RCCL path on one GPU
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    from iwae_amd.native import NativeModel
    from iwae_amd.parallel import DataParallelStep
    from iwae_amd import utils

    x_np, p = synthetic_batch(B_PER_GPU * world, 123)
    lo = rank * B_PER_GPU
    x_dev = torch.tensor(x_np[lo:lo + B_PER_GPU], device="cuda")       # inputs resident in HBM before timing
    net = NativeModel(1, N_HIDDEN, N_LATENT, x_dim=X_DIM, device=local_rank, seed=123, world_size=world, rank=rank)
    net.set_output_bias(utils.bias_from_mean(p))                       # identical init on every rank (same seed)
    dp = DataParallelStep(net, rank, world)
    lr = 1e-3

    def step():
        dp.step(x_dev.data_ptr(), B_PER_GPU, K_SAMPLES, 1.0, lr, OBJ_IWAE_ELBO)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # HIP events around out_bwd on every TIMING_EVERY-th step of the timed region (each record is a small stream bubble)
    net.enable_timing(int(os.environ.get("IWAE_BENCH_TIMING", TIMING_EVERY)))
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ktimes = {k: net.kernel_time(k) for k in KERNELS}
    net.enable_timing(0)
    elbo = net.forward(x_np[lo:lo + B_PER_GPU], K_SAMPLES)["iwae_elbo"]
    llh_eval = None
    if world == 1 and not args.no_llh_eval:      # the other half of BASELINE's metric: the test-LLH protocol of main.py:170-184 (k = 5000 per image), untimed extra
        n_eval = 1000
        net.eval_llh(x_np[:32], 5000)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        llh = net.eval_llh(x_np[:n_eval], 5000)
        dt_eval = time.perf_counter() - t1
        llh_eval = {"k": 5000, "images": n_eval, "images_per_s": round(n_eval / dt_eval, 1), "llh": round(float(llh), 3),
                    "note": "forward-only evaluator on synthetic images with the just-trained weights; 10 000 images take %.2f s" % (dt_eval * 10000 / n_eval)}

    if rank == 0:
        ms = dt * 1e3 / args.steps
        value = B_PER_GPU * world * args.steps / dt
        dom = max(KERNELS, key=lambda k: ktimes[k][0])          # the kernel with the longest average launch
        dom_us, dom_n = ktimes[dom]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_kernel_traffic.json")     # PMC passes of tools/profile_bench.sh
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        ach = KERNELS[dom]["bytes"] / (dom_us * 1e-6) / 1e9 if dom_us > 0 else 0.0
        out = {
            "metric": "images/sec (train step) IWAE k=50 batch 1024 @1/2/4/8 GPU; test LLH@k=5000",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "1-layer IWAE k=50 batch 1024 per GPU (BASELINE configs[1]; N=8 is configs[4]), "
                                   "objective iwae_elbo, train step = forward + backward + Adam(eps=1e-4)"
                                   + (" + RCCL all-reduce of 455384 fp32 grads" if world > 1 else ""),
                       "global_batch": B_PER_GPU * world, "n_samples": K_SAMPLES, "parallelism": "dp%d" % world,
                       "step_gemm_tflops": round(FLOP_PER_STEP * world / (dt / args.steps) / 1e12, 1),
                       "iwae_elbo_after": round(float(elbo), 3), "llh_eval_k5000": llh_eval},
            "roofline": {"bound": "hbm", "kernel": KERNELS[dom]["name"],
                         "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": traffic,
                         "avg_launch_us": round(dom_us, 2), "launches": dom_n,
                         "algorithmic_bytes_per_launch": int(KERNELS[dom]["bytes"]),
                         "flop_per_launch": KERNELS[dom]["flop"],
                         "mfma_tflops": round(KERNELS[dom]["flop"] / (dom_us * 1e-6) / 1e12, 1) if dom_us > 0 else 0.0,
                         "other": {k: round(v[0], 2) for k, v in ktimes.items() if k != dom}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
