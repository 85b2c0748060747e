"""Data-parallel training over RCCL/xGMI: one process per GPU, ONE flat gradient all-reduce.

The reference wraps the model in DistributedDataParallel (trainer.py:78-82), whose bucketed
hooks all-reduce 1-2 small buckets per step at these model sizes (0.1-13 MB). Here the
gradients already live in FlatAdam's single contiguous buffer, so the exchange step is exactly
one all-reduce(SUM) of that buffer, stream-ordered right after backward, with the 1/world scaling
folded into the optimiser's grad pre-scale (no extra pass) and the global grad norm computed after
the reduction, as DDP does. Messages this small are latency bound on the point-to-point xGMI mesh,
which favours one direct message over a multi-hop ring of buckets.

Transports:
  * "rccl" — the production path: `pg_allreduce_sum` of libpg_hip.so (csrc/comm.hip), a direct
    librccl call on the step's own stream. It neither allocates nor synchronises, so it is captured
    INSIDE the step's hipGraph (graph.py): backward -> all-reduce -> norm + Adam is one graph launch.
    `torch.distributed` is used only as the rendezvous that ships rank 0's 128-byte unique id.
    Chosen when the process group's backend is nccl (= RCCL on ROCm), or forced with
    `transport="rccl"` (a world of one still creates a communicator and runs the collective: the
    single-GPU test of this path).
  * "gloo" — development only: GPU tensors staged through pinned host memory, so that the
    multi-process logic can be exercised with two processes on ONE GPU (RCCL refuses two ranks on a
    device). Not capturable: the step is split into two graphs around it.

The collective is NOT overlapped with compute: reference semantics need every gradient before the
norm / update, and the message (0.11 MB ImageGPT ... 3.6 MB PixelSNAIL ... 13.5 MB GatedPixelCNN)
costs tens of microseconds next to a 1.4-140 ms step.
"""

import ctypes

import torch
import torch.distributed as dist

from pytorch_generative_amd import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatGradAllReduce:
    def __init__(self, optimizer, process_group=None, transport=None):
        self.opt = optimizer
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(process_group) if dist.is_initialized() else 0
        self.flat_grad = optimizer.flat_grad
        self.flat_param = optimizer.flat_param
        if transport is None:
            if self.world > 1 and self.flat_grad.is_cuda and dist.get_backend(process_group) == "nccl":
                transport = "rccl"
            elif self.world > 1:
                transport = "gloo"
            else:
                transport = "none"
        if transport not in ("rccl", "gloo", "none"):
            raise ValueError(f"unknown transport {transport!r}")
        self.transport = transport
        self._host = None
        self._owns_comm = False
        if transport == "rccl":
            self._init_rccl()
        elif transport == "gloo" and self.flat_grad.is_cuda:
            self._host = torch.empty(self.flat_grad.shape, dtype=self.flat_grad.dtype, pin_memory=True)
        optimizer.set_grad_prescale(1.0 / self.world)

    # ---- direct RCCL ---------------------------------------------------------------------------
    def _init_rccl(self):
        if not self.flat_grad.is_cuda:
            raise ValueError("transport='rccl' needs the flat buffers on the GPU")
        lib = _lib.load()
        if lib.pg_comm_world() not in (0, self.world):
            raise RuntimeError("a communicator of another world size is alive in this process")
        if lib.pg_comm_world() == self.world:
            return  # one communicator per process, shared by every reducer of that process
        buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        if self.rank == 0:
            _lib.check(lib.pg_comm_unique_id(buf), "pg_comm_unique_id")
        ident = [buf.raw]
        if self.world > 1:  # the rendezvous: ship rank 0's id, nothing else goes through torch.distributed
            dist.broadcast_object_list(ident, src=0, group=self.group)
        torch.cuda.set_device(self.flat_grad.device)
        _lib.check(lib.pg_comm_init(self.rank, self.world, ident[0]), "pg_comm_init")
        self._owns_comm = True

    @property
    def capturable(self):
        """True if all_reduce() may be captured inside the step's hipGraph."""
        return self.transport in ("rccl", "none")

    @property
    def active(self):
        """True if all_reduce() does anything (world > 1, or a forced RCCL world of one)."""
        return self.transport == "rccl" or self.world > 1

    def close(self):
        if self._owns_comm:
            _lib.check(_lib.load().pg_comm_destroy(), "pg_comm_destroy")
            self._owns_comm = False

    # ---- collectives ---------------------------------------------------------------------------
    def _via_gloo(self, fn, tensor):
        if self._host is None:
            fn(tensor)
            return
        self._host.copy_(tensor)  # synchronous D2H on the current stream
        fn(self._host)
        tensor.copy_(self._host, non_blocking=False)

    def broadcast_parameters(self, src=0):
        """DDP's constructor broadcast: every rank starts from rank `src`'s parameters."""
        if self.transport == "rccl":
            t = self.flat_param
            _lib.check(_lib.load().pg_broadcast(t.data_ptr(), t.numel(), _lib.DTYPE_F32, src, _stream()),
                       "pg_broadcast")
        elif self.world > 1:
            self._via_gloo(lambda t: dist.broadcast(t, src=src, group=self.group), self.flat_param)

    def all_reduce(self):
        """Sum the flat gradient over ranks (call after backward, before optimizer.step())."""
        if self.transport == "rccl":
            t = self.flat_grad
            _lib.check(_lib.load().pg_allreduce_sum(t.data_ptr(), t.numel(), _lib.DTYPE_F32, _stream()),
                       "pg_allreduce_sum")
        elif self.world > 1:
            self._via_gloo(lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group),
                           self.flat_grad)
