"""Child process of tests/test_gpu_multi.py: one rank of a 2-GPU data-parallel run through the library's own RCCL exchange.
Usage: python _dp_gpu_worker.py <rank> <world> <port> <out.npz>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)       # ships the id blob + the agreement flag only
    torch.cuda.set_device(rank)
    from iwae_amd.native import NativeModel
    from iwae_amd.parallel import DataParallelStep
    from oracle import iwae_np as O

    layers = int(os.environ.get("DP_TEST_LAYERS", "1"))
    nh, nl = (200, 100) if layers == 1 else ([200, 100], [100, 50])
    Bl, k, steps = 96, 50, 6                                            # 4 800 rows per rank; the global batch is world * Bl images
    x = O.synthetic_binarized(Bl * world, 5)
    P = O.init_params(layers, nh, nl, 11, x_mean=O.synthetic_pixel_means())
    net = NativeModel(layers, nh, nl, device=rank, seed=123, world_size=world, rank=rank)
    net.set_params(O.flatten_params(P))
    dp = DataParallelStep(net, rank, world)                             # raises on every rank if RCCL cannot be brought up
    assert dp.in_library and net.comm_info() == (world, rank), (dp.path, net.comm_info())
    xd = torch.tensor(x[rank * Bl:(rank + 1) * Bl], device="cuda")
    for _ in range(steps):
        dp.step(xd.data_ptr(), Bl, k, 1.0, 1e-3, 1)                     # objective 1 = iwae_elbo
    net.sync()
    params = net.get_params()
    mo, ve, t = net.get_adam_state()
    res = {"params": params, "mom": mo, "vel": ve, "t": np.array([t])}
    if rank == 0:
        # the same global batch on ONE handle without communicators: noise is keyed by the global image index, so the N-rank
        # run must reproduce it up to the fp32 summation order of the gradient (1e-6 relative after one step)
        ref = NativeModel(layers, nh, nl, device=0, seed=123)
        ref.set_params(O.flatten_params(P))
        xa = torch.tensor(x, device="cuda")
        for s in range(steps):
            ref.set_step(s, 0)
            ref.train_step_devptr(xa.data_ptr(), Bl * world, k, 1.0, 1e-3, 1)
        ref.sync()
        res["ref_params"] = ref.get_params()
        ref.close()
    np.savez(out, **res)
    net.comm_destroy()
    net.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
