"""VectorQuantizer on the HIP operator path (reference nn/utils.py:16-96) — SURVEY.md §8(f) rank 4.

Same constructor signature, the same state_dict keys / shapes (buffers `_embedding`, `_cluster_size`,
`_embedding_avg`), EMA codebook update inside forward when training. Kernels: csrc/vq.hip (assignment
with the reference's distance form and first-minimum rule, EMA update, straight-through backward, MSE
loss). Validated on MI355X against outputs of the reference (tests/golden/vq_*.pt, tests/test_gpu_f4.py).
Only `use_ema=True` (the reference default and what the VQ-VAE models use) is implemented; the
gradient-descent codebook raises. `ReZeroWrapper` (nn/utils.py:7-13) is not on the named path.
"""

import torch
from torch import nn
from torch.nn import init

from pytorch_generative_amd import _lib, ops


class _VectorQuantize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, embedding, cluster_size, embedding_avg, decay, training):
        lib = _lib.load()
        x = ops._chk(x, "vq.x")
        n, d, h, w = x.shape
        k = embedding.shape[0]
        L = h * w
        idx = torch.empty(n * L, device=x.device, dtype=torch.int32)
        q = torch.empty_like(x)
        st = torch.empty_like(x)
        loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        _lib.check(lib.pg_vq_assign(x.data_ptr(), embedding.data_ptr(), idx.data_ptr(), q.data_ptr(),
                                    st.data_ptr(), loss.data_ptr(), n, d, L, k, ops._stream()), "pg_vq_assign")
        if training:  # EMA codebook update, in place on the buffers like the reference's .data updates
            count = torch.empty(k, device=x.device, dtype=torch.float32)
            total = torch.empty(k * d, device=x.device, dtype=torch.float32)
            _lib.check(lib.pg_vq_ema_update(x.data_ptr(), idx.data_ptr(), cluster_size.data_ptr(),
                                            embedding_avg.data_ptr(), embedding.data_ptr(), count.data_ptr(),
                                            total.data_ptr(), n, d, L, k, float(decay), ops._stream()),
                       "pg_vq_ema_update")
        ctx.save_for_backward(x, q)
        ctx.mark_non_differentiable(idx)
        return st, loss.view(()), idx

    @staticmethod
    def backward(ctx, d_st, d_loss, _d_idx):
        lib = _lib.load()
        x, q = ctx.saved_tensors
        d_st = ops._chk(d_st, "vq.d_quantized")
        g = ops._chk(d_loss.reshape(1), "vq.d_loss")
        dx = torch.empty_like(x)
        _lib.check(lib.pg_vq_bwd(x.data_ptr(), q.data_ptr(), d_st.data_ptr(), g.data_ptr(), dx.data_ptr(),
                                 x.numel(), ops._stream()), "pg_vq_bwd")
        return dx, None, None, None, None, None


class _MSE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        lib = _lib.load()
        a = ops._chk(a, "mse.a")
        b = ops._chk(b, "mse.b")
        if a.shape != b.shape:
            raise ValueError("mse_loss: shape mismatch")
        loss = torch.zeros(1, device=a.device, dtype=torch.float32)
        _lib.check(lib.pg_mse_fwd(a.data_ptr(), b.data_ptr(), loss.data_ptr(), a.numel(), ops._stream()),
                   "pg_mse_fwd")
        ctx.save_for_backward(a, b)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        a, b = ctx.saved_tensors
        g = ops._chk(g.reshape(1), "mse.grad")
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b) if ctx.needs_input_grad[1] else None
        _lib.check(lib.pg_mse_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(),
                                  da.data_ptr() if da is not None else 0,
                                  db.data_ptr() if db is not None else 0, a.numel(), ops._stream()),
                   "pg_mse_bwd")
        return da, db


def mse_loss(a, b):
    """F.mse_loss(a, b) (mean over every element), gradients to both arguments."""
    return _MSE.apply(a, b)


class VectorQuantizer(nn.Module):
    """nn/utils.py:16-96. forward(x) -> (quantized with the straight-through gradient, commitment loss)."""

    def __init__(self, n_embeddings, embedding_dim, use_ema=True, ema_decay=0.99):
        super().__init__()
        if not use_ema:
            raise NotImplementedError("VectorQuantizer(use_ema=False): only the EMA codebook is on the HIP path")
        if embedding_dim > 64:
            raise ValueError("VectorQuantizer: embedding_dim > 64 is not covered by pg_vq_assign")
        self.n_embeddings = n_embeddings
        self.embedding_dim = embedding_dim
        self._use_ema = use_ema
        self._decay = ema_decay
        embedding = torch.zeros(n_embeddings, embedding_dim)
        init.kaiming_uniform_(embedding, nonlinearity="linear")
        self.register_buffer("_embedding", embedding)
        self.register_buffer("_cluster_size", torch.zeros(n_embeddings))
        self.register_buffer("_embedding_avg", embedding.clone())
        self.last_indices = None  # (N*H*W,) int32 of the latest forward (extension, for inspection)

    def forward(self, x):
        assert x.shape[1] == self.embedding_dim, "Input channels must equal embedding_dim."
        st, loss, idx = _VectorQuantize.apply(x, self._embedding, self._cluster_size, self._embedding_avg,
                                              self._decay, self.training)
        self.last_indices = idx
        return st, loss
