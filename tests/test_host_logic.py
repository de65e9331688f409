"""CPU tests of the host-side logic: data layouts the kernels rely on (Python mirror of
iwae_amd/csrc/layout.h), reference-API helpers, and the data-parallel exchange over gloo (2 ranks)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- mirror of layout.h -----------------------------------------------------------------
def hperm(x):
    return (0x78 >> (2 * x)) & 3


def p_pos(f):
    return (f & ~31) + 8 * ((f & 15) >> 2) + 4 * ((f & 31) >> 4) + (f & 3)


def img_mg_byte(m, kf, KT):
    mg, tile, rr = m >> 6, (m >> 4) & 3, m & 15
    ks, h, q, i = kf >> 5, (kf >> 4) & 1, (kf >> 2) & 3, kf & 3
    return mg * (4 * KT + 1) * 1024 + (ks * 4 + tile) * 1024 + rr * 64 + ((q ^ hperm(rr >> 2)) * 16) + (4 * h + i) * 2


def test_layout_header_matches_mirror():
    src = open(os.path.join(ROOT, "iwae_amd", "csrc", "layout.h")).read()
    assert "(0x78 >> (2 * x)) & 3" in src
    assert "(f & ~31) + 8 * ((f & 15) >> 2) + 4 * ((f & 31) >> 4) + (f & 3)" in src
    assert "(4 * KT + 1) * 1024" in src


def test_p_layout_is_a_permutation_with_lane_chunks():
    pos = [p_pos(f) for f in range(256)]
    assert sorted(pos) == list(range(256))
    # lane quad q owns features {32t+4q+i} U {32t+16+4q+i}: 8 contiguous positions = one 16-byte load
    for t in range(4):
        for q in range(4):
            own = [32 * t + 16 * h + 4 * q + i for h in range(2) for i in range(4)]
            assert sorted(p_pos(f) for f in own) == list(range(32 * t + 8 * q, 32 * t + 8 * q + 8))


def test_image_blocks_are_bijective_and_bank_conflict_free():
    KT = 7
    offs = {img_mg_byte(m, kf, KT) for m in range(64) for kf in range(32 * KT)}
    assert len(offs) == 64 * 32 * KT and max(offs) < (4 * KT) * 1024 and all(o % 2 == 0 for o in offs)
    # ds_read_b128 services a wave in four 16-lane groups (MI355X_MICROARCH.md, LDS table); bank = (addr/4) % 64
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
              list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
              list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
    for g in groups:
        banks = {}
        for lane in g:
            rho, q = lane & 15, lane >> 4
            a = rho * 64 + ((q ^ hperm(rho >> 2)) * 16)
            for b in range(4):
                banks.setdefault((a // 4 + b) % 64, set()).add(a)
        assert max(len(s) for s in banks.values()) == 1


def test_utils_and_optimizer_shim():
    from iwae_amd import utils
    from iwae_amd.optimizers import Adam
    lw = np.random.default_rng(0).standard_normal((5, 3)) * 30
    ref = np.log(np.mean(np.exp(lw - lw.max(0)), 0)) + lw.max(0)
    np.testing.assert_allclose(utils.logmeanexp(lw, 0), ref)
    np.random.seed(1)
    xb = utils.bernoullisample(np.full((100, 50), 0.25))
    assert xb.dtype == np.float32 and set(np.unique(xb)) <= {0.0, 1.0} and abs(xb.mean() - 0.25) < 0.03
    m = utils.MyMetric()
    m.update_state(np.array([[1.0]])); m.update_state(np.array([[3.0]]))
    assert float(m.result()) == 2.0
    b = utils.bias_from_mean(np.array([0.0, 0.5, 1.0]))
    np.testing.assert_allclose(b, [-np.log(1 / 0.001 - 1), 0.0, -np.log(1 / 0.999 - 1)], rtol=1e-6)
    opt = Adam(1e-3, epsilon=1e-4)
    assert np.isclose(opt.learning_rate.numpy(), 1e-3)
    opt.learning_rate.assign(5e-4)
    assert np.isclose(float(opt.learning_rate), 5e-4)
    assert opt.hyper() == (0.9, 0.999, 1e-4)
    assert Adam(1e-3).hyper() == (0.9, 0.999, 1e-7)      # Keras defaults are accepted and handed to the device optimizer (iwae_set_adam)
    with pytest.raises(ValueError):
        Adam(1e-3, beta_1=1.0)


def test_task_driver_flags_identical_to_reference():
    """tasks/task02.py:15-22 and tasks/task04.py / task05.py:34-43 of the reference: the drivers under tasks/ parse the same flags with the same defaults."""
    sys.path.insert(0, os.path.join(ROOT, "tasks"))
    try:
        import importlib
        common = importlib.import_module("_common")
        a = common.parser_task02().parse_args([])
        assert (a.n_samples, a.batch_size, a.epochs, a.gpu) == (5, 20, -1, "0")
        with pytest.raises(SystemExit):
            common.parser_task02().parse_args(["--objective", "iwae_elbo"])      # task02's script has no such flag
        c = common.parser_conditional().parse_args(["--objective", "vae_elbo_kl", "--stochastic_layers", "2"])
        assert (c.stochastic_layers, c.n_samples, c.batch_size, c.epochs, c.objective, c.gpu) == (2, 5, 20, -1, "vae_elbo_kl", "0")
    finally:
        sys.path.remove(os.path.join(ROOT, "tasks"))
        sys.modules.pop("_common", None)


def test_main_cli_flags_identical_to_reference():
    sys.argv = ["main.py"]
    import importlib
    main = importlib.import_module("main")
    a = main.parser.parse_args([])
    assert (a.stochastic_layers, a.n_samples, a.batch_size, a.epochs, a.objective, a.gpu) == (1, 5, 20, -1, "iwae_elbo", "0")
    a = main.parser.parse_args(["--stochastic_layers", "2", "--n_samples", "50", "--objective", "iwae_eq14", "--batch_size", "1024", "--epochs", "3", "--gpu", "1"])
    assert a.stochastic_layers == 2 and a.objective == "iwae_eq14"
    with pytest.raises(SystemExit):
        main.parser.parse_args(["--objective", "nope"])
    with pytest.raises(SystemExit):
        main.parser.parse_args(["--stochastic_layers", "3"])


# ---- data-parallel exchange, 2 ranks over gloo ---------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from oracle import iwae_np as O
    from iwae_amd import parallel
    import make_golden as MG
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    B, k = 8, 5
    x, P, eps = MG.inputs(1, 16, 4, 48, B, k, 21)
    lo, hi = parallel.shard_bounds(B, rank, world)
    # each rank: mean-over-its-shard gradient (what iwae_forward_backward leaves on the device)
    _, g = O.loss_grads_1layer(P, x[lo:hi], eps[:, lo:hi], 1.0, "iwae_elbo")
    flat = torch.tensor(O.flatten_grads(g))
    split = flat.clone()
    parallel.allreduce_sum_(flat)
    # the overlapped exchange (decoder segment first, then the encoder's): the same sums, message by message
    parallel.exchange_split_(split, split.numel() // 3, None)
    assert torch.equal(split, flat)
    whole = flat.clone()
    parallel.exchange_split_(whole, whole.numel(), None)      # no side segment: one message
    assert torch.equal(whole, flat * world)
    flat = flat / world                       # grad_scale = 1/world in iwae_adam_step
    p1, _, _ = O.adam_update(O.flatten_params(P), flat.numpy(), 0.0, 0.0, 1, 1e-3)
    if rank == 0:
        _, gfull = O.loss_grads_1layer(P, x, eps, 1.0, "iwae_elbo")
        pf, _, _ = O.adam_update(O.flatten_params(P), O.flatten_grads(gfull), 0.0, 0.0, 1, 1e-3)
        q.put((float(np.max(np.abs(flat.numpy() - O.flatten_grads(gfull)))), float(np.max(np.abs(p1 - pf)))))
    # replicas stay identical
    t = torch.tensor(p1)
    lst = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(lst, t)
    assert all(torch.equal(lst[0], o) for o in lst)
    # the RCCL id blob of the in-library exchange (iwae_comm_init) travels the same way: rank 0 makes it, every rank gets it
    blob = parallel.share_comm_id(lambda: bytes(range(256)), rank)
    assert blob == bytes(range(256)), (rank, blob[:8])
    assert parallel.all_ranks_ok(True) and not parallel.all_ranks_ok(rank == 0)
    dist.destroy_process_group()


def test_data_parallel_gradient_exchange_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    gerr, perr = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert gerr < 1e-12 and perr < 1e-12


class _FakeNet:
    """Stands in for NativeModel in the host-logic tests of the exchange initialisation (no GPU, no RCCL)."""
    fail_id = False

    def __init__(self, fail_init, fail_preflight=False):
        self.fail_init, self.fail_preflight, self.inited, self.destroyed, self.entered = fail_init, fail_preflight, False, False, False

    @classmethod
    def comm_unique_id(cls):
        if cls.fail_id:
            raise OSError("librccl.so: cannot open shared object file")
        return bytes(range(256))

    def comm_preflight(self, blob, world, rank):      # iwae_comm_preflight: what comm_init can refuse without another rank
        assert blob == bytes(range(256))
        if self.fail_preflight:
            raise RuntimeError("comm_init: communicators already exist (iwae_comm_destroy first)")

    def comm_init(self, blob, world, rank):
        assert blob == bytes(range(256))
        self.entered = True
        # ncclCommInitRank is a blocking rendezvous: model it -- a rank that never arrives would leave the others here (the test would
        # time out), which is exactly what the agreed preflight in front of it has to rule out
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
        if self.fail_init:
            raise RuntimeError("ncclCommInitRank: unhandled system error")
        self.inited = True

    def comm_destroy(self):
        self.destroyed = True


class _FakeNetNoId(_FakeNet):
    fail_id = True


def _dp_init_worker(rank, world, port, scenario, q):
    sys.path.insert(0, ROOT)
    from iwae_amd import parallel
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if scenario == "id_fails_on_rank0":          # ranks 1.. sit in the broadcast: rank 0 must ship its failure, everyone raises
        net = _FakeNetNoId(False)
    elif scenario == "init_fails_on_rank1":      # rank-asymmetric failure: rank 0's communicators are given back, everyone raises
        net = _FakeNet(rank == 1)
    elif scenario == "preflight_fails_on_rank1": # rank 1 would refuse BEFORE the rendezvous: nobody may enter it (round-3 advisor finding)
        net = _FakeNet(False, fail_preflight=(rank == 1))
    else:
        net = _FakeNet(False)
    out = {"rank": rank}
    try:
        dp = parallel.DataParallelStep(net, rank, world)
        out.update(raised=False, path=dp.path, in_library=dp.in_library)
    except RuntimeError as e:
        out.update(raised=True, msg=str(e))
    out.update(inited=net.inited, destroyed=net.destroyed, entered=net.entered)
    # whatever happened, the ranks are still in step: one more collective must complete
    t = torch.ones(1)
    dist.all_reduce(t)
    out["still_in_step"] = float(t.item()) == world
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.parametrize("scenario", ["ok", "id_fails_on_rank0", "init_fails_on_rank1", "preflight_fails_on_rank1"])
def test_exchange_initialisation_is_collective_and_never_falls_back_silently(scenario):
    """world 2 over gloo.  A failing RCCL initialisation -- rank 0 cannot make the id, or comm_init fails on ONE rank -- must
    raise on EVERY rank (no rank left behind in a broadcast, none on a different exchange path), with the communicators of
    the ranks that did succeed destroyed again; the healthy case takes the in-library path on both."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_init_worker, args=(r, 2, port, scenario, q)) for r in range(2)]
    for p in procs:
        p.start()
    outs = sorted((q.get(timeout=120) for _ in procs), key=lambda o: o["rank"])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert all(o["still_in_step"] for o in outs)
    if scenario == "ok":
        assert all(not o["raised"] and o["in_library"] and o["path"] == "rccl_in_library" and o["inited"] and not o["destroyed"] for o in outs)
    else:
        assert all(o["raised"] for o in outs), outs
        assert all("in-library RCCL exchange could not be initialised" in o["msg"] for o in outs)
        if scenario == "id_fails_on_rank0":
            assert all("librccl.so" in o["msg"] and not o["inited"] for o in outs)
        elif scenario == "preflight_fails_on_rank1":
            assert all(not o["entered"] and not o["inited"] for o in outs), outs      # no rank entered the rendezvous
        else:
            assert outs[0]["inited"] and outs[0]["destroyed"] and not outs[1]["inited"]      # rank 0 gave its communicators back


def test_single_rank_needs_no_exchange():
    from iwae_amd import parallel
    dp = parallel.DataParallelStep(_FakeNet(True), 0, 1)
    assert dp.path == "single" and not dp.in_library


def test_shard_bounds():
    from iwae_amd import parallel
    assert [parallel.shard_bounds(8192, r, 8) for r in (0, 7)] == [(0, 1024), (7168, 8192)]
    with pytest.raises(ValueError):
        parallel.shard_bounds(10, 0, 4)
