"""ops._common — shared helpers: stream / pointer plumbing, activation ids, zero fills, gradient sinks, the row-decode context.

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import os

import torch

from pytorch_generative_amd import _lib


ACT_NONE, ACT_RELU, ACT_ELU, ACT_GELU = 0, 1, 2, 3
ACT_ELU_OUT = 4  # dgrad epilogue only: derivative of ELU from its output (include/pg_hip.h)
GATE_TANH, GATE_IDENTITY = 0, 1
_ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "elu": ACT_ELU, "gelu": ACT_GELU}


def _stream():
    return torch.cuda.current_stream().cuda_stream


class RowDecode:
    """Context of row-cached incremental sampling (models/base.py).

    While active, the convolutional models' ordinary forward() is called on ONE image row
    (N, C, 1, W): every layer of these models is row-causal (its output row r depends on input rows
    <= r only), so nn.Conv2d keeps the last k input rows it needs (k = upward reach of its taps) in a
    private band buffer and evaluates only the current row; nn.CausalAttention keeps its q / k / v
    maps; image_positional_encoding returns the current row of the full-size encoding. `commit`
    marks the pass that runs once a row is final and pushes it into the caches."""

    current = None

    def __init__(self, height):
        self.height, self.row, self.commit = int(height), 0, False

    def __enter__(self):
        RowDecode.current = self
        return self

    def __exit__(self, *exc):
        RowDecode.current = None
        return False


def _chk(t, name):
    if not t.is_cuda:
        raise RuntimeError(
            f"{name}: expected a tensor on the MI355X (cuda) device, got {t.device}; "
            "the HIP operator path has no CPU fallback"
        )
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if t.device.index != torch.cuda.current_device():
        # kernels are enqueued on the CURRENT device's stream with raw pointers: a tensor of another
        # GPU would be dereferenced from the wrong device (Trainer / recipes call set_device)
        raise RuntimeError(f"{name}: tensor lives on {t.device} but the current device is "
                           f"cuda:{torch.cuda.current_device()}; call torch.cuda.set_device first")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return 0 if t is None else t.data_ptr()


def zeros(shape, device):
    """torch.zeros(shape, dtype=float32) with the fill done by the library's own kernel (pg_fill): gradient sinks, loss scalars,
    KL accumulators — no ATen fill kernel inside a captured step."""
    t = torch.empty(shape, device=device, dtype=torch.float32)
    if t.numel():
        _lib.check(_lib.load().pg_fill(t.data_ptr(), 0.0, t.numel(), _stream()), "pg_fill")
    return t


def zeros_like(t):
    return zeros(tuple(t.shape), t.device)


def _sink(param):
    """Returns the direct gradient sink of a parameter (or None)."""
    return getattr(param, "_pg_grad", None) if param is not None else None


CONV_FMT_F32, CONV_FMT_B3, CONV_FMT_B3_GATE = 1, 2, 3  # include/pg_hip.h PG_CONV_FMT_*
FUSE_SKIP = os.environ.get("PG_FUSE_SKIP", "1") != "0"  # A/B: 0 = plain fan-out, autograd sums the gradients


def _dense_per_image(t):
    """True if every image of the (N, C, H, W) tensor is a dense (C, H, W) block and the data is fp32 on
    the GPU: the batch stride may be larger than C*H*W (a channel slice of a wider tensor)."""
    if not (t.is_cuda and t.dtype == torch.float32 and t.dim() == 4):
        return False
    _, c, h, w = t.shape
    st = t.stride()
    return st[3] == 1 and st[2] == w and st[1] == h * w and st[0] >= c * h * w and t.data_ptr() % 16 == 0
