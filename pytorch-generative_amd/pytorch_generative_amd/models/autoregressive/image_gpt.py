"""ImageGPT on the MI355X operator path.

Same constructor, attribute names (= state_dict keys) and forward graph as the reference's
models/autoregressive/image_gpt.py:21-109 — including its doubled residual (the block returns
x + ..., and the model loop adds x again, :50-52 vs :107-108) — with every op on HIP kernels:
NCHW LayerNorm, 1x1 convs (residual adds fused into the projection / MLP-out kernels), exact
GELU, and the fused causal attention core.
"""

import os

import torch
from torch import nn

from pytorch_generative_amd import _lib
from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models import base

# A/B (measurement only): 0 = one weight-gradient reduction launch per block instead of one per model (round 6)
_BLOCK_CHAIN = os.environ.get("PG_BLOCK_CHAIN", "1") != "0"


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

    def forward_plus_input(self, x, chain=None, flush=False):
        """x + self(x) — what the model loop computes (reference image_gpt.py:107) — on the fused
        head / attention / tail kernels (gpt_block.hip) when the block has the BASELINE shape.
        chain / flush (round 6): a queue shared by all blocks of the model, flushed by the block whose backward runs last —
        the weight-gradient rows of all blocks are then reduced by ONE launch (ops.gpt_block_head)."""
        if not self._fused_ok(x):
            return ops.add(x, self.forward(x))
        attn = self._attn
        pair = {}  # lets the block's two backward kernels share one weight-gradient reduction launch
        if chain is not None:
            pair["chain"], pair["flush"] = chain, flush
        qkv, xs = ops.gpt_block_head(x, self._ln1, attn._q, attn._kv, pair)
        o = ops.causal_attention_qkv(qkv, attn._n_heads, attn._embed_channels, attn._out_channels,
                                     attn._mask_center)
        return ops.gpt_block_tail(o, xs, attn._proj, self._ln2, self._out[0], self._out[2], pair)

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

    # ---- incremental sampling -------------------------------------------------
    def _incremental_ok(self, n):
        ld = (n + 15) // 16 * 16
        probe = torch.empty(1, self._input.weight.shape[0], 1, ld, device="meta")
        return all(blk._fused_ok(probe) for blk in self._transformer) and self._input.bias is not None

    @torch.no_grad()
    def sample(self, n_samples=None, conditioned_on=None, *, incremental=True, return_logits=False):
        """Same contract as the reference's AutoregressiveModel.sample (models/base.py:97-120: raster
        order, only entries < 0 are drawn, `_sample_fn` called once per pixel on (n, c) logits) — but
        each step evaluates ONLY the current position: the model is causal, so the activations of
        pixel p are final once pixels < p are. One position of all samples is a (channels, batch)
        matrix, i.e. the (C, L) plane layout of the block kernels with the batch as the pixel axis:
        the fused head / tail kernels run unchanged, attention becomes a one-query-per-(n, head)
        decode against per-layer K / V caches. The reference (and `incremental=False`) instead runs
        one full forward per pixel: H*W times the work.

        return_logits (extension): also returns the (H*W, n, c) logits the draws were made from."""
        canvas = self._start_canvas(n_samples, conditioned_on)
        n, c, h, w = canvas.shape
        if not incremental or not self._incremental_ok(n):
            if return_logits:
                raise ValueError("return_logits needs the incremental sampler")
            return super().sample(conditioned_on=canvas)
        lib = _lib.load()
        dev, L = canvas.device, h * w
        canvas = canvas.contiguous()
        ld = (n + 15) // 16 * 16
        conv = self._input
        emb = conv.weight.shape[0]
        kh, kw = conv.weight.shape[2], conv.weight.shape[3]
        ops.mul_inplace_(conv.weight.data, conv.mask)  # nn/convolution.py:42, as on every forward
        blocks = list(self._transformer)
        caches = []
        for blk in blocks:
            a = blk._attn
            caches.append((torch.zeros(n, a._embed_channels, L, device=dev),
                           torch.zeros(n, a._out_channels, L, device=dev)))
        x = torch.zeros(1, emb, 1, ld, device=dev)
        pos_dev = torch.zeros(1, dtype=torch.int32, device=dev)  # raster position, advanced on the device

        def position_logits():
            """logits (n, c) of the position in pos_dev; appends its keys / values to the caches"""
            st = torch.cuda.current_stream().cuda_stream
            _lib.check(lib.pg_sample_embed(canvas.data_ptr(), self._pos.data_ptr(),
                                           conv.weight.data_ptr(), conv.bias.data_ptr(), x.data_ptr(),
                                           n, c, h, w, emb, kh, kw, 0, 0, ld, pos_dev.data_ptr(), st),
                       "pg_sample_embed")
            cur = x
            for blk, (kc, vc) in zip(blocks, caches):
                a = blk._attn
                qkv, xs = ops.gpt_block_head(cur, blk._ln1, a._q, a._kv)
                o = torch.empty(1, a._out_channels, 1, ld, device=dev)
                _lib.check(lib.pg_attn_decode(qkv.data_ptr(), kc.data_ptr(), vc.data_ptr(),
                                              o.data_ptr(), n, a._n_heads, L, 0,
                                              a._embed_channels // a._n_heads,
                                              a._out_channels // a._n_heads, ld, int(a._mask_center),
                                              pos_dev.data_ptr(), st), "pg_attn_decode")
                cur = ops.gpt_block_tail(o, xs, a._proj, blk._ln2, blk._out[0], blk._out[2])
            return self._out(self._ln(cur))[0, :, 0, :n].t().contiguous()

        # One hipGraph = one position of every layer (~30 launches) + the position increment; only the
        # draw (`_sample_fn` is user code) and the canvas update stay eager. Falls back to eager
        # launches if the capture is refused.
        graph, static_logits = None, None
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                position_logits()  # warm-up at position 0 (rewrites cache column 0 with the same values)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                static_logits = position_logits()
                pos_dev.add_(1)
        except Exception:  # noqa: BLE001 — capture is an optimisation only
            graph = None
            torch.cuda.synchronize()
        pos_dev.zero_()
        all_logits = []
        for row in range(h):
            for col in range(w):
                if graph is not None:
                    graph.replay()
                    logits = static_logits.clone() if return_logits else static_logits
                else:
                    logits = position_logits()
                    pos_dev.add_(1)
                if return_logits:
                    all_logits.append(logits)
                drawn = self._sample_fn(logits).view(n, c)
                current = canvas[:, :, row, col]
                canvas[:, :, row, col] = torch.where(current < 0, drawn, current)
        if return_logits:
            return canvas, torch.stack(all_logits)
        return canvas

    def forward(self, x):
        x = self._input(ops.add_broadcast_batch(x, self._pos))
        # one queue for the blocks' weight-gradient partial rows, flushed by the FIRST block (its backward runs last) — only when
        # every block takes the fused kernels and a backward pass with a gradient for the first block's input will run
        blocks = list(self._transformer)
        chain = None
        if (ops.DEFER_BLOCK_REDUCE and _BLOCK_CHAIN and torch.is_grad_enabled() and x.requires_grad and len(blocks) > 1
                and all(b._fused_ok(x) for b in blocks)):
            chain = ops.new_block_chain()
        for i, block in enumerate(blocks):
            x = block.forward_plus_input(x, chain, flush=(i == 0))
        return self._out(self._ln(x))


def reproduce(n_epochs=457, batch_size=64, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (image_gpt.py:112-175: same model
    hyper-parameters, Adam lr 5e-3, per-batch lr decay 0.999977) on the MI355X path. Arguments as the reference;
    `debug_loader` replaces both loaders (any iterable of (x, y) batches). Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: ImageGPT(in_channels=1, out_channels=1, in_size=28, n_transformer_blocks=8,
                         n_attention_heads=2, n_embedding_channels=64),
        loaders=recipes.binarized_mnist, loss_fn=recipes.bce_loss, lr=5e-3, lr_decay=0.999977,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
