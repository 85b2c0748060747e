"""Attention-side operators of the hot path, on hand-written gfx950 kernels.

Mirrors CausalAttention (reference nn/attention.py:66-161) and image_positional_encoding
(:37-57). The L x L score / mask tensors of the reference are never materialised.
"""

import functools

import torch
from torch import nn

from pytorch_generative_amd import ops
from pytorch_generative_amd.nn.convolution import Conv2d


@functools.lru_cache(maxsize=32)
def _posenc_cached(shape, device):
    return ops.image_positional_encoding(shape, device)


def image_positional_encoding(shape, device=None):
    """(N, 2, H, W) tensor of (row, col) pixel coordinates scaled to [-.5, .5).

    Generated on the GPU and cached per (shape, device) — the reference builds it on the CPU
    and copies it to the device once per block per step (pixel_snail.py:113).
    """
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    ctx = ops.RowDecode.current
    if ctx is not None:  # row-cached sampling: `shape` is one image row; return that row of the full map
        n, c, _, w = (int(s) for s in shape)
        full = _posenc_cached((n, c, ctx.height, w), torch.device(device))
        return full[:, :, ctx.row:ctx.row + 1, :].contiguous()
    return _posenc_cached(tuple(int(s) for s in shape), torch.device(device))


class CausalAttention(nn.Module):
    """Autoregressively masked multi-head self attention over the pixels of an image."""

    def __init__(
        self,
        in_channels,
        n_heads=1,
        embed_channels=None,
        out_channels=None,
        mask_center=False,
        extra_input_channels=0,
    ):
        super().__init__()
        self._n_heads = n_heads
        self._embed_channels = embed_channels or in_channels
        self._out_channels = out_channels or in_channels
        self._mask_center = mask_center
        self._q = Conv2d(in_channels=in_channels, out_channels=self._embed_channels, kernel_size=1)
        self._kv = Conv2d(
            in_channels=in_channels + extra_input_channels,
            out_channels=self._embed_channels + self._out_channels,
            kernel_size=1,
        )
        self._proj = Conv2d(
            in_channels=self._out_channels, out_channels=self._out_channels, kernel_size=1
        )
        if extra_input_channels == 0:
            # layout hint for FlatAdam: q and kv parameters back to back, so that the two projections
            # of the same input run as ONE convolution over the merged views (ops.conv_pair_views)
            self._kv.weight._pg_follows = self._q.weight
            self._kv.bias._pg_follows = self._q.bias

    def _row_reset(self):
        self._row_q = self._row_kv = None

    def _row_forward(self, ctx, x, extra_x, res):
        """Row-cached sampling: x / extra_x are row `ctx.row`. The row's q / k / v are written into
        full-size caches and the attention core runs over rows <= ctx.row (everything an earlier
        position contributed is final); only the current row of the output is kept."""
        q = self._q(x)
        kv = self._kv(x if extra_x is None else torch.cat((x, extra_x), dim=1))
        n, _, _, w = q.shape
        if getattr(self, "_row_q", None) is None or self._row_q.shape[0] != n or self._row_q.shape[3] != w:
            self._row_q = torch.zeros((n, q.shape[1], ctx.height, w), device=q.device)
            self._row_kv = torch.zeros((n, kv.shape[1], ctx.height, w), device=q.device)
        r = ctx.row
        self._row_q[:, :, r:r + 1].copy_(q)
        self._row_kv[:, :, r:r + 1].copy_(kv)
        out = ops.causal_attention(self._row_q[:, :, :r + 1].contiguous(), self._row_kv[:, :, :r + 1].contiguous(),
                                   self._n_heads, self._embed_channels, self._out_channels, self._mask_center)
        return self._proj(out[:, :, r:r + 1].contiguous(), res=res)

    def forward(self, x, extra_x=None, *, res=None):
        """x feeds q, k and v; extra_x (optional) is concatenated for k and v only.

        `res` (extension) is added to the projected output inside the projection kernel.
        x may be a tuple of tensors (extension): they are concatenated on dim 1 — together with
        extra_x in ONE copy when the merged projection below runs.
        """
        parts = list(x) if isinstance(x, (tuple, list)) else [x]
        ctx = ops.RowDecode.current
        if ctx is not None:
            return self._row_forward(ctx, parts[0] if len(parts) == 1 else torch.cat(parts, dim=1), extra_x, res)
        if extra_x is not None and ops.FUSE_QKV_EXTRA and extra_x.is_cuda:
            # q reads a channel PREFIX of what k / v read (attention.py:139-143): one 1x1 convolution
            # over cat(x, extra_x) with q's weight rows zero-padded over the extra channels gives
            # [q | k | v] in one tensor — one concatenation, one projection, one data gradient and
            # one weight gradient instead of two each plus the sum of the two input gradients.
            # The parameters keep the reference's shapes; the merged weight is rebuilt per step
            # (40 x 69 values for PixelSNAIL) and its gradient is split back row block by row block.
            x_all = ops.concat_channels(parts + [extra_x])  # HIP copy; its backward hands out slice views
            w, b = ops.merge_qkv_weight(self._q, self._kv)  # gradients go straight into the parameters' sinks
            qkv = ops.conv2d_taps(x_all, w, b, self._kv._conv_spec())
            out = ops.causal_attention_qkv(
                qkv, self._n_heads, self._embed_channels, self._out_channels, self._mask_center
            )
            return self._proj(out, res=res)
        x = parts[0] if len(parts) == 1 else torch.cat(parts, dim=1)
        if extra_x is None and ops.FUSE_PAIR:
            views = ops.conv_pair_views(self._q, self._kv)
            if views is not None:
                qkv = ops.conv2d_pair(x, self._q, self._kv, views, self._q._conv_spec())
                out = ops.causal_attention_qkv(
                    qkv, self._n_heads, self._embed_channels, self._out_channels, self._mask_center
                )
                return self._proj(out, res=res)
        q = self._q(x)
        if extra_x is not None:
            x = torch.cat((x, extra_x), dim=1)
        kv = self._kv(x)
        out = ops.causal_attention(
            q, kv, self._n_heads, self._embed_channels, self._out_channels, self._mask_center
        )
        return self._proj(out, res=res)
