"""Data-parallel training over RCCL/xGMI: one process per GPU, ONE flat gradient all-reduce.

The reference wraps the model in DistributedDataParallel (trainer.py:78-82), whose bucketed
hooks all-reduce 1-2 small buckets per step at these model sizes (0.1-13 MB). Here the
gradients already live in FlatAdam's single contiguous buffer, so the exchange step is exactly
one `all_reduce(SUM)` of that buffer — small enough to be latency bound, which on the
point-to-point xGMI mesh favours one direct message over a multi-hop ring of buckets — issued
on a side HIP stream right after backward, with the 1/world scaling folded into the optimiser's
grad pre-scale (no extra pass). Global grad norm is computed after the reduction, as DDP does.
"""

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, optimizer, process_group=None, device=None):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat_grad = optimizer.flat_grad
        self.flat_param = optimizer.flat_param
        self._side = None
        if self.flat_grad.is_cuda:
            self._side = torch.cuda.Stream(device=self.flat_grad.device)
        optimizer.set_grad_prescale(1.0 / self.world)

    def broadcast_parameters(self, src=0):
        """DDP's constructor broadcast: every rank starts from rank `src`'s parameters."""
        if self.world > 1:
            dist.broadcast(self.flat_param, src=src, group=self.group)

    def all_reduce(self):
        """Sum the flat gradient over ranks (call after backward, before optimizer.step())."""
        if self.world == 1:
            return
        if self._side is None:  # CPU tensors (gloo tests)
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            return
        cur = torch.cuda.current_stream(self.flat_grad.device)
        self._side.wait_stream(cur)
        with torch.cuda.stream(self._side):
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        cur.wait_stream(self._side)
