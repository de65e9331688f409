"""Data-parallel training step: one process per GPU, gradients exchanged with RCCL over xGMI.  The reference has no
counterpart (single device, main.py:32); the path shards naturally because the loss is a mean over images
(src/iwae1.py:120-134):

  rank r takes images [r*B/N, (r+1)*B/N) -> forward + backward on its shard (mean over ITS images)
  all-reduce(SUM) of the flat fp32 gradient buffer (455,384 floats, in place on the device) in two messages: the
  decoder's layers (done early, on the library's side stream) beside the encoder's backward pass, then the encoder's
  Adam with grad_scale = 1/N on every rank (replicas stay bit-identical)

Default: the collective is issued BY THE LIBRARY (iwae_comm_init: ncclAllReduce on its own streams, the decoder segment's
exchange + update deferred beside the next encoder forward like the single-GPU step's); torch.distributed only ships the
128-byte RCCL ids to the ranks.  IWAE_DP_TORCH=1 (or a failing RCCL initialisation) selects the round-1 path instead:
iwae_forward_backward -> torch.distributed.all_reduce of iwae_grad_devptr() -> iwae_adam_step.

Noise is keyed by the GLOBAL image index (batch_offset), so N ranks draw the same eps as 1 rank.
The helpers take plain torch tensors so the host logic is testable on CPU with gloo."""
import os

import torch
import torch.distributed as dist


def shard_bounds(n, rank, world_size):
    """Contiguous, equal shards (the all-reduce average assumes equal shard sizes)."""
    if n % world_size:
        raise ValueError("global batch %d not divisible by world size %d" % (n, world_size))
    per = n // world_size
    return rank * per, (rank + 1) * per


def allreduce_sum_(flat, group=None):
    """In-place sum of the flat gradient over ranks; one message, not one per tensor."""
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("IWAE_BENCH_FORCE_DIST")):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


class _DevArray:
    """Expose a raw device pointer of the C ABI to torch (no copy) via __cuda_array_interface__."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False), "version": 2}


def exchange_split_(flat, side_offset, side_stream, group=None):
    """Sum the flat gradient over ranks in two messages: [side_offset, n) ordered behind `side_stream` (the stream the
    decoder's gradient is completed on, well before the step ends) and [0, side_offset) behind the current stream; the
    current stream then waits for both.  side_stream = None (CPU tensors / gloo in the host-logic tests): same two
    messages, in the same order on every rank, without streams."""
    n = flat.numel()
    if side_offset >= n:
        return allreduce_sum_(flat, group)
    active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("IWAE_BENCH_FORCE_DIST"))
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
    carries for the in-library exchange.  Works on any backend (gloo on CPU in the host-logic tests)."""
    box = [make_id() if rank == 0 else None]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


class DataParallelStep:
    def __init__(self, net, rank, world_size, group=None, overlap=None, in_library=None):
        self.net, self.rank, self.world, self.group = net, int(rank), int(world_size), group
        self.step_idx = 0
        self._side = None
        self.overlap = False
        forced = bool(os.environ.get("IWAE_BENCH_FORCE_DIST"))
        want_lib = (os.environ.get("IWAE_DP_TORCH") is None) if in_library is None else bool(in_library)
        self.in_library = False
        if (self.world > 1 or forced) and want_lib:
            try:
                blob = share_comm_id(type(net).comm_unique_id, self.rank, group)
                net.comm_init(blob, self.world, self.rank)
                self.in_library = True
                return
            except Exception as e:      # an RCCL that cannot be loaded / initialised: the torch.distributed path still works
                print("iwae_amd.parallel: in-library RCCL exchange unavailable (%s); using torch.distributed.all_reduce" % e, flush=True)
        ptr, n = net.grad_devptr()
        self.grad = torch.as_tensor(_DevArray(ptr, n), device="cuda")
        # run the library on torch's current stream so the collective is ordered after the backward
        net.set_stream(torch.cuda.current_stream().cuda_stream)
        # overlap (opt-in, IWAE_DP_OVERLAP=1): the decoder's 54 % of the gradient bytes are exchanged beside the encoder's
        # backward pass.  Off by default: with one rank (trivial collectives) the two messages and their three cross-stream
        # waits cost +22 us per step (0.338 -> 0.360 ms), the one-message path's exposed all-reduce could not be measured
        # on the one-GPU development box, so the default stays the path with the fewest stream hand-offs.
        self.overlap = (os.environ.get("IWAE_DP_OVERLAP") is not None) if overlap is None else bool(overlap)

    def step(self, x_devptr, b_local, k, beta, lr, objective_id, global_batch_offset=0):
        """x_devptr: this rank's shard, already resident in HBM ([b_local, x_dim] float32)."""
        self.net.set_step(self.step_idx, global_batch_offset + self.rank * b_local)
        if self.in_library or (self.world == 1 and not os.environ.get("IWAE_BENCH_FORCE_DIST")):
            # the library's own train step: single GPU (Adam fused into the gradient reduction) or, after comm_init, with its
            # ncclAllReduce between gradient and update
            self.net.train_step_devptr(x_devptr, b_local, k, beta, lr, objective_id)
        elif self.overlap:
            side, off = self.net.forward_backward_split_devptr(x_devptr, b_local, k, beta, objective_id)
            if self._side is None or self._side.cuda_stream != side:
                self._side = torch.cuda.ExternalStream(side)
            exchange_split_(self.grad, off, self._side, self.group)
            self.net.adam_step(lr, 1.0 / self.world)
        else:
            self.net.forward_backward_devptr(x_devptr, b_local, k, beta, objective_id)
            allreduce_sum_(self.grad, self.group)
            self.net.adam_step(lr, 1.0 / self.world)
        self.step_idx += 1
