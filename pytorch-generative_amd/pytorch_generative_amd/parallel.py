"""Data-parallel training over RCCL/xGMI: one process per GPU, ONE flat gradient all-reduce.

The reference wraps the model in DistributedDataParallel (trainer.py:78-82), whose bucketed
hooks all-reduce 1-2 small buckets per step at these model sizes (0.1-13 MB). Here the
gradients already live in FlatAdam's single contiguous buffer, so the exchange step is exactly
one `all_reduce(SUM)` of that buffer, stream-ordered right after backward, with the 1/world scaling
folded into the optimiser's grad pre-scale (no extra pass) and the global grad norm computed after
the reduction, as DDP does. Messages this small are latency bound on the point-to-point xGMI mesh,
which favours one direct message over a multi-hop ring of buckets.

The collective is NOT overlapped with compute: reference semantics need every gradient before the
norm / update, and the message (0.11 MB ImageGPT ... 3.6 MB PixelSNAIL ... 13.5 MB GatedPixelCNN)
costs tens of microseconds next to a 12-70 ms step; what keeps the step cheap around it is that
both halves (forward+backward | norm+Adam) replay from hipGraphs (graph.py).

`backend="gloo"` (CPU tensors, or GPU tensors staged through pinned host memory) exists so that the
multi-process path can be exercised on a single-GPU development box; production runs use
`backend="nccl"`, which is RCCL on ROCm.
"""

import torch
import torch.distributed as dist


class FlatGradAllReduce:
    def __init__(self, optimizer, process_group=None):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.flat_grad = optimizer.flat_grad
        self.flat_param = optimizer.flat_param
        self._via_host = (self.world > 1 and self.flat_grad.is_cuda
                          and dist.get_backend(process_group) == "gloo")
        self._host = None
        if self._via_host:
            self._host = torch.empty(self.flat_grad.shape, dtype=self.flat_grad.dtype, pin_memory=True)
        optimizer.set_grad_prescale(1.0 / self.world)

    def _collective(self, fn, tensor):
        if not self._via_host:
            fn(tensor)
            return
        self._host.copy_(tensor)  # synchronous D2H on the current stream
        fn(self._host)
        tensor.copy_(self._host, non_blocking=False)

    def broadcast_parameters(self, src=0):
        """DDP's constructor broadcast: every rank starts from rank `src`'s parameters."""
        if self.world > 1:
            self._collective(lambda t: dist.broadcast(t, src=src, group=self.group), self.flat_param)

    def all_reduce(self):
        """Sum the flat gradient over ranks (call after backward, before optimizer.step())."""
        if self.world > 1:
            self._collective(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group),
                             self.flat_grad)
