"""ImageGPT on the MI355X operator path.

Same constructor, attribute names (= state_dict keys) and forward graph as the reference's
models/autoregressive/image_gpt.py:21-109 — including its doubled residual (the block returns
x + ..., and the model loop adds x again, :50-52 vs :107-108) — with every op on HIP kernels:
NCHW LayerNorm, 1x1 convs (residual adds fused into the projection / MLP-out kernels), exact
GELU, and the fused causal attention core.
"""

import torch
from torch import nn

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models import base


class TransformerBlock(nn.Module):
    def __init__(self, n_channels, n_attention_heads):
        super().__init__()
        self._ln1 = pg_nn.NCHWLayerNorm(n_channels)
        self._ln2 = pg_nn.NCHWLayerNorm(n_channels)
        self._attn = pg_nn.CausalAttention(
            in_channels=n_channels,
            n_heads=n_attention_heads,
            embed_channels=n_channels,
            out_channels=n_channels,
        )
        self._out = nn.Sequential(
            pg_nn.Conv2d(in_channels=n_channels, out_channels=4 * n_channels, kernel_size=1),
            nn.GELU(),  # placeholder keeping the Sequential indices; fused into _out[2]'s load
            pg_nn.Conv2d(in_channels=4 * n_channels, out_channels=n_channels, kernel_size=1),
        )

    def _fused_ok(self, x):
        return ops.gpt_block_supported(x, self._ln1, self._attn._q, self._attn._kv, self._attn._proj,
                                       self._ln2, self._out[0], self._out[2])

    def forward_plus_input(self, x):
        """x + self(x) — what the model loop computes (reference image_gpt.py:107) — on the fused
        head / attention / tail kernels (gpt_block.hip) when the block has the BASELINE shape."""
        if not self._fused_ok(x):
            return ops.add(x, self.forward(x))
        attn = self._attn
        qkv, xs = ops.gpt_block_head(x, self._ln1, attn._q, attn._kv)
        o = ops.causal_attention_qkv(qkv, attn._n_heads, attn._embed_channels, attn._out_channels,
                                     attn._mask_center)
        return ops.gpt_block_tail(o, xs, attn._proj, self._ln2, self._out[0], self._out[2])

    def forward(self, x):
        # skip=True: the residual branch's gradient is added inside the LayerNorm backward kernel
        y, x = self._ln1(x, skip=True)
        x = self._attn(y, res=x)                         # x + attn(ln1(x))
        y, x = self._ln2(x, skip=True)
        if ops.mlp_gelu_supported(y, self._out[0], self._out[2]):
            # fc1 -> GELU -> fc2 -> + x in one kernel each way: the hidden tensor stays in registers
            return ops.mlp_gelu(y, self._out[0], self._out[2], res=x)
        hidden = self._out[0](y)
        # exact GELU fused into the MLP-out kernel's input load, residual into its epilogue
        return self._out[2](hidden, in_act="gelu", res=x)  # x + mlp(ln2(x))


class ImageGPT(base.AutoregressiveModel):
    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        in_size=28,
        n_transformer_blocks=8,
        n_attention_heads=4,
        n_embedding_channels=16,
        sample_fn=None,
    ):
        super().__init__(sample_fn)
        self._pos = nn.Parameter(torch.zeros(1, in_channels, in_size, in_size))
        self._input = pg_nn.CausalConv2d(
            mask_center=True,
            in_channels=in_channels,
            out_channels=n_embedding_channels,
            kernel_size=3,
            padding=1,
        )
        self._transformer = nn.ModuleList(
            TransformerBlock(n_channels=n_embedding_channels, n_attention_heads=n_attention_heads)
            for _ in range(n_transformer_blocks)
        )
        self._ln = pg_nn.NCHWLayerNorm(n_embedding_channels)
        self._out = pg_nn.Conv2d(
            in_channels=n_embedding_channels, out_channels=out_channels, kernel_size=1
        )

    def forward(self, x):
        x = self._input(ops.add_broadcast_batch(x, self._pos))
        for block in self._transformer:
            x = block.forward_plus_input(x)
        return self._out(self._ln(x))
