"""Data-parallel training step: one process per GPU, gradients exchanged with RCCL over xGMI.  The reference has no
counterpart (single device, main.py:32); the path shards naturally because the loss is a mean over images
(src/iwae1.py:120-134):

  rank r takes images [r*B/N, (r+1)*B/N) -> forward + backward on its shard (mean over ITS images)
  all-reduce(SUM) of the flat fp32 gradient buffer (455,384 floats, in place on the device) in two messages: the
  decoder's layers (done early, on the library's side stream) beside the encoder's backward pass, then the encoder's
  Adam with grad_scale = 1/N on every rank (replicas stay bit-identical)

Default: the collective is issued BY THE LIBRARY (iwae_comm_init: ncclAllReduce on its own streams, the decoder segment's
exchange + update deferred beside the next encoder forward like the single-GPU step's); torch.distributed only ships the
256-byte RCCL ids to the ranks and carries the collective agreement that every rank's initialisation succeeded.  A failing
RCCL initialisation raises on every rank (it never falls back silently); in_library=False or allow_fallback=True select the
round-1 path instead: iwae_forward_backward -> torch.distributed.all_reduce of iwae_grad_devptr() -> iwae_adam_step.

Noise is keyed by the GLOBAL image index (batch_offset), so N ranks draw the same eps as 1 rank.
The helpers take plain torch tensors so the host logic is testable on CPU with gloo."""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world_size):
    """Contiguous, equal shards (the all-reduce average assumes equal shard sizes)."""
    if n % world_size:
        raise ValueError("global batch %d not divisible by world size %d" % (n, world_size))
    per = n // world_size
    return rank * per, (rank + 1) * per


def allreduce_sum_(flat, group=None, force=False):
    """In-place sum of the flat gradient over ranks; one message, not one per tensor.  force: also with one rank (rehearsal)."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class _DevArray:
    """Expose a raw device pointer of the C ABI to torch (no copy) via __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def exchange_split_(flat, side_offset, side_stream, group=None, force=False):
    """Sum the flat gradient over ranks in two messages: [side_offset, n) ordered behind `side_stream` (the stream the
    decoder's gradient is completed on, well before the step ends) and [0, side_offset) behind the current stream; the
    current stream then waits for both.  side_stream = None (CPU tensors / gloo in the host-logic tests): same two
    messages, in the same order on every rank, without streams."""
    n = flat.numel()
    if side_offset >= n:
        return allreduce_sum_(flat, group, force)
    active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
    if not active:
        if side_stream is not None:
            torch.cuda.current_stream().wait_stream(side_stream)
        return flat
    if side_stream is not None:
        with torch.cuda.stream(side_stream):
            w = dist.all_reduce(flat[side_offset:], op=dist.ReduceOp.SUM, group=group, async_op=True)
    else:
        w = dist.all_reduce(flat[side_offset:], op=dist.ReduceOp.SUM, group=group, async_op=True)
    dist.all_reduce(flat[:side_offset], op=dist.ReduceOp.SUM, group=group)
    w.wait()
    return flat


def share_comm_id(make_id, rank, group=None):
    """Rank 0 makes the RCCL id blob (NativeModel.comm_unique_id), every rank receives it: the only thing torch.distributed
    carries for the in-library exchange.  Works on any backend (gloo on CPU in the host-logic tests).
    A failure on rank 0 (RCCL not loadable ...) is SHIPPED, not raised there: the other ranks sit in the broadcast, so rank 0
    must take part in it either way -- every rank then raises the same error together."""
    box = [None]
    if rank == 0:
        try:
            box = [("ok", make_id())]
        except Exception as e:          # noqa: BLE001 -- whatever it was, the other ranks must hear of it
            box = [("error", "%s: %s" % (type(e).__name__, e))]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    kind, payload = box[0]
    if kind != "ok":
        raise RuntimeError("rank 0 could not make the RCCL id: %s" % payload)
    return payload


def all_ranks_ok(ok, group=None, device=None):
    """Collective agreement on a per-rank success flag (all-reduce MIN): True only if every rank passed True."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1):
        return bool(ok)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def init_in_library_exchange(net, rank, world_size, group=None, device=None):
    """Brings up the library's own RCCL communicators on every rank, or on none: the id blob (or rank 0's failure) reaches every
    rank, every rank runs the non-collective preflight and the ranks agree on it, then every rank enters comm_init (the rendezvous) and the
    outcome is agreed on collectively once more.  Returns None on success, else the reason as
    a string -- the SAME decision on every rank, with the communicators of the ranks that did succeed destroyed again, so no rank
    is left issuing ncclAllReduce while another issues something else (a rank-asymmetric failure used to be a hang)."""
    err = None
    try:
        blob = share_comm_id(type(net).comm_unique_id, rank, group)
    except Exception as e:              # noqa: BLE001 -- identical on every rank (see share_comm_id)
        return str(e)
    # ncclCommInitRank is a blocking rendezvous: a rank that would refuse before reaching it (RCCL not loadable, world / rank mismatch,
    # communicators already there) must say so BEFORE anyone enters it -- a non-collective preflight, agreed on collectively
    try:
        net.comm_preflight(blob, world_size, rank)
    except Exception as e:              # noqa: BLE001
        err = "rank %d: comm_init refused: %s: %s" % (rank, type(e).__name__, e)
    if not all_ranks_ok(err is None, group, device):
        return err or ("rank %d: ready, but another rank refused comm_init before the rendezvous" % rank)
    try:
        net.comm_init(blob, world_size, rank)
    except Exception as e:              # noqa: BLE001
        err = "rank %d: comm_init failed: %s: %s" % (rank, type(e).__name__, e)
    if all_ranks_ok(err is None, group, device):
        return None
    if err is None:                     # this rank succeeded, another did not: give the communicators back
        try:
            net.comm_destroy()
        except Exception:               # noqa: BLE001
            pass
        err = "rank %d: comm_init succeeded here but failed on another rank" % rank
    return err


class DataParallelStep:
    """in_library (default True): the collective is issued by the library itself (iwae_comm_init).  With world_size > 1 a failing
    RCCL initialisation RAISES on every rank -- it never falls back silently: a benchmark or a training run would otherwise
    measure a different exchange path than the one it reports.  allow_fallback=True opts into round 1's torch.distributed path
    when (and only when) every rank agrees the in-library path is unavailable.  force_dist rehearses the multi-rank code path
    on one rank (trivial collectives)."""

    def __init__(self, net, rank, world_size, group=None, overlap=False, in_library=True, allow_fallback=False, force_dist=False):
        self.net, self.rank, self.world, self.group = net, int(rank), int(world_size), group
        self.step_idx = 0
        self._side = None
        self.overlap = False
        self.force_dist = bool(force_dist)
        self.in_library = False
        self.path = "single"
        if not (self.world > 1 or self.force_dist):
            return
        if in_library:
            dev = None
            if dist.is_available() and dist.is_initialized() and dist.get_backend(group) == "nccl":
                dev = torch.device("cuda", torch.cuda.current_device())
            why = init_in_library_exchange(net, self.rank, self.world, group, dev)
            if why is None:
                self.in_library = True
                self.path = "rccl_in_library"
                return
            if not allow_fallback:
                raise RuntimeError("iwae_amd.parallel: the in-library RCCL exchange could not be initialised on every rank (%s); "
                                   "pass allow_fallback=True to use torch.distributed.all_reduce instead" % why)
            print("iwae_amd.parallel: in-library RCCL exchange unavailable (%s); every rank uses torch.distributed.all_reduce" % why, flush=True)
        self.path = "torch_fallback"
        ptr, n = net.grad_devptr()
        self.grad = torch.as_tensor(_DevArray(ptr, n), device="cuda")
        # run the library on torch's current stream so the collective is ordered after the backward
        net.set_stream(torch.cuda.current_stream().cuda_stream)
        # overlap (opt-in): the decoder's 54 % of the gradient bytes are exchanged beside the encoder's backward pass.  Off by
        # default: with one rank (trivial collectives) the two messages and their three cross-stream waits cost +22 us per step.
        self.overlap = bool(overlap)

    def step(self, x_devptr, b_local, k, beta, lr, objective_id, global_batch_offset=0):
        """x_devptr: this rank's shard, already resident in HBM ([b_local, x_dim] float32)."""
        self.net.set_step(self.step_idx, global_batch_offset + self.rank * b_local)
        if self.in_library or self.path == "single":
            # the library's own train step: single GPU (Adam fused into the gradient reduction) or, after comm_init, with its
            # ncclAllReduce between gradient and update
            self.net.train_step_devptr(x_devptr, b_local, k, beta, lr, objective_id)
        elif self.overlap:
            side, off = self.net.forward_backward_split_devptr(x_devptr, b_local, k, beta, objective_id)
            if self._side is None or self._side.cuda_stream != side:
                self._side = torch.cuda.ExternalStream(side)
            exchange_split_(self.grad, off, self._side, self.group, force=self.force_dist)
            self.net.adam_step(lr, 1.0 / self.world)
        else:
            self.net.forward_backward_devptr(x_devptr, b_local, k, beta, objective_id)
            allreduce_sum_(self.grad, self.group, force=self.force_dist)
            self.net.adam_step(lr, 1.0 / self.world)
        self.step_idx += 1
