"""ops.elementwise — activations, gate, adds, positional encoding, channel concatenation, gradient fan-out sums.

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import torch

from pytorch_generative_amd import _lib
from pytorch_generative_amd.ops._common import (
    ACT_ELU,
    ACT_GELU,
    ACT_RELU,
    FUSE_SKIP,
    _chk,
    _dense_per_image,
    _sink,
    _stream,
    zeros,
)


# --------------------------------------------------------------------------------------------
# elementwise
# --------------------------------------------------------------------------------------------
class _Act(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        lib = _lib.load()
        x = _chk(x, "act.x")
        y = torch.empty_like(x)
        _lib.check(lib.pg_act_fwd(x.data_ptr(), y.data_ptr(), x.numel(), act, _stream()), "pg_act_fwd")
        ctx.save_for_backward(x)
        ctx.act = act
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "act.dy")
        dx = torch.empty_like(x)
        _lib.check(lib.pg_act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), ctx.act,
                                  _stream()), "pg_act_bwd")
        return dx, None


def relu(x):
    return _Act.apply(x, ACT_RELU)


def elu(x):
    return _Act.apply(x, ACT_ELU)


def gelu(x):
    return _Act.apply(x, ACT_GELU)


class _Gated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gate, res=None):
        lib = _lib.load()
        x = _chk(x, "gated.x")
        n, c2, h, w = x.shape
        assert c2 % 2 == 0, "x must have an even number of channels."
        y = torch.empty((n, c2 // 2, h, w), device=x.device, dtype=torch.float32)
        if res is None:
            _lib.check(lib.pg_gated_fwd(x.data_ptr(), y.data_ptr(), n, c2 // 2, h * w, gate, _stream()),
                       "pg_gated_fwd")
        else:
            res = _chk(res, "gated.res")
            if res.shape != y.shape:
                raise ValueError("gated_activation: residual shape mismatch")
            _lib.check(lib.pg_gated_fwd_res(x.data_ptr(), res.data_ptr(), y.data_ptr(), n, c2 // 2,
                                            h * w, gate, _stream()), "pg_gated_fwd_res")
        ctx.save_for_backward(x)
        ctx.gate, ctx.has_res = gate, res is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        dy = _chk(dy, "gated.dy")
        n, c2, h, w = x.shape
        dx = torch.empty_like(x)
        _lib.check(lib.pg_gated_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), n, c2 // 2, h * w,
                                    ctx.gate, _stream()), "pg_gated_bwd")
        return dx, None, (dy if ctx.has_res else None)


def gated_activation(x, gate, res=None):
    """act(x[:, :C]) * sigmoid(x[:, C:]) (+ res): GatedActivation, optionally fused with the
    residual add that follows it in PixelSNAIL's ResidualBlock (pixel_snail.py:55-56)."""
    if res is not None and ((x.shape[2] * x.shape[3]) % 4 != 0):
        return add(res, _Gated.apply(x, gate))
    return _Gated.apply(x, gate, res)


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a = _chk(a, "add.a")
        b = _chk(b, "add.b")
        if a.shape != b.shape:
            raise ValueError(f"add: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty_like(a)
        _lib.check(lib.pg_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _stream()), "pg_add")
        return out

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _Add.apply(a, b)


class _AddBcast(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, gp):
        lib = _lib.load()
        x = _chk(x, "add_bcast.x")
        p = _chk(p, "add_bcast.p")
        per = p.numel()
        if x.numel() % per or tuple(x.shape[1:]) != tuple(p.shape[-(x.dim() - 1):]):
            raise ValueError(f"add_bcast: {tuple(p.shape)} does not broadcast over {tuple(x.shape)}")
        y = torch.empty_like(x)
        _lib.check(lib.pg_add_bcast_fwd(x.data_ptr(), p.data_ptr(), y.data_ptr(), x.shape[0], per,
                                        _stream()), "pg_add_bcast_fwd")
        ctx.gp = gp
        ctx.pshape = p.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        dy = _chk(dy, "add_bcast.dy")
        dp = None
        gp = ctx.gp
        if ctx.needs_input_grad[1]:
            if gp is None:
                dp = zeros(ctx.pshape, dy.device)
                gp = dp
            per = gp.numel()
            _lib.check(lib.pg_add_bcast_bwd(dy.data_ptr(), gp.data_ptr(), dy.shape[0], per, _stream()),
                       "pg_add_bcast_bwd")
        return (dy if ctx.needs_input_grad[0] else None), dp, None


def add_broadcast_batch(x, p):
    """x + p where p has batch dim 1 (the learned positional map of ImageGPT)."""
    return _AddBcast.apply(x, p, _sink(p))


def image_positional_encoding(shape, device):
    lib = _lib.load()
    n, _, h, w = shape
    out = torch.empty((n, 2, h, w), device=device, dtype=torch.float32)
    if not out.is_cuda:
        raise RuntimeError("image_positional_encoding: the HIP path needs a cuda device")
    _lib.check(lib.pg_image_positional_encoding(out.data_ptr(), n, h, w, _stream()),
               "pg_image_positional_encoding")
    return out


def mul_inplace_(w, mask):
    lib = _lib.load()
    if not (w.is_cuda and mask.is_cuda and w.is_contiguous() and mask.is_contiguous()):
        raise RuntimeError("mul_inplace_: expects contiguous cuda tensors")
    _lib.check(lib.pg_mul_inplace(w.data_ptr(), mask.data_ptr(), w.numel(), _stream()), "pg_mul_inplace")
    return w


class _ConcatChannels(torch.autograd.Function):
    """torch.cat(tensors, dim=1) of (N, C_i, H, W) tensors on pg_copy_rows; the gradients handed back are
    channel-slice VIEWS of the incoming gradient (no copies): the convolution they flow into adds them in
    its data-gradient epilogue (n_skip protocol) or reads them with their batch stride."""

    @staticmethod
    def forward(ctx, *tensors):
        lib = _lib.load()
        parts = [_chk(t, "concat.part") for t in tensors]
        n, _, h, w = parts[0].shape
        L = h * w
        ctot = sum(int(t.shape[1]) for t in parts)
        out = torch.empty((n, ctot, h, w), device=parts[0].device, dtype=torch.float32)
        off = 0
        for t in parts:
            if tuple(t.shape[0:1] + t.shape[2:]) != (n, h, w):
                raise ValueError("concat_channels: batch / spatial shape mismatch")
            c = int(t.shape[1])
            _lib.check(lib.pg_copy_rows(t.data_ptr(), out.data_ptr() + 4 * off * L, n, c * L, c * L, ctot * L, 0,
                                        _stream()), "pg_copy_rows")
            off += c
        ctx.sizes = [int(t.shape[1]) for t in parts]
        return out

    @staticmethod
    def backward(ctx, dy):
        grads, off = [], 0
        for i, c in enumerate(ctx.sizes):
            grads.append(dy.narrow(1, off, c) if ctx.needs_input_grad[i] else None)
            off += c
        return tuple(grads)


def concat_channels(tensors):
    return _ConcatChannels.apply(*tensors)


def _sum_into(out, ts):
    """out = sum of the tensors ts (same shape as out; each dense, or dense per image with a larger batch stride) in one
    pg_sum_rows launch per 31 tensors."""
    import ctypes

    lib = _lib.load()
    n_batch, per = (int(out.shape[0]), out.numel() // max(int(out.shape[0]), 1)) if out.dim() == 4 else (1, out.numel())
    acc = None
    for i in range(0, len(ts), 31):  # 32 rows per launch, the running sum among them
        grp = ([acc] if acc is not None else []) + list(ts[i:i + 31])
        rows = (ctypes.c_void_p * len(grp))(*[t.data_ptr() for t in grp])
        bs = (ctypes.c_long * len(grp))(*[(int(t.stride(0)) if t.dim() == 4 else per) for t in grp])
        _lib.check(lib.pg_sum_rows(rows, bs, len(grp), out.data_ptr(), n_batch, per, _stream()), "pg_sum_rows")
        acc = out
    return out


class _SumVectors(torch.autograd.Function):
    """out = sum of k equally shaped tensors in ONE launch (pg_sum_rows); every input's gradient is the output's."""

    @staticmethod
    def forward(ctx, *ts):
        ts = [_chk(t, "sum_vectors.t") for t in ts]
        if any(t.shape != ts[0].shape for t in ts):
            raise ValueError("sum_vectors: shape mismatch")
        ctx.k = len(ts)
        return _sum_into(torch.empty_like(ts[0]), ts)

    @staticmethod
    def backward(ctx, g):
        return (g,) * ctx.k


class _Fanout(torch.autograd.Function):
    """k pass-through aliases of x for k readers: autograd then never sums gradients for x with its own chain of k - 1 `add`
    kernels — the k gradients come back HERE and are summed by one launch (pg_sum_rows), each read where it lies (a gradient
    that is a channel slice of a wider tensor, e.g. out of concat_channels' backward, with its batch stride: no copy)."""

    @staticmethod
    def forward(ctx, x, k):
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *gs):
        gs = [g if (g.dim() == 4 and _dense_per_image(g)) else _chk(g, "fanout.g") for g in gs if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        out = torch.empty(gs[0].shape, device=gs[0].device, dtype=torch.float32)
        return _sum_into(out, gs), None


def fanout(x, k):
    """k aliases of x, one per reader (see _Fanout); x itself when it needs no gradient or k < 2."""
    if k < 2 or not FUSE_SKIP or not (torch.is_grad_enabled() and x.requires_grad):
        return (x,) * max(int(k), 1)
    return _Fanout.apply(x, int(k))


def sum_vectors(ts):
    """torch.stack(ts).sum(dim=0) (vd_vae.py:400, the sum of the per-block KL terms) without the stacked tensor."""
    ts = list(ts)
    if not ts:
        raise ValueError("sum_vectors: empty list")
    return ts[0] if len(ts) == 1 else _SumVectors.apply(*ts)
