"""PixelSNAIL on the MI355X operator path
(reference models/autoregressive/pixel_snail.py:27-187).

2x2 convolutions with padding 1 cropped to [:h, :w] become four tap offsets
(-1,-1),(-1,0),(0,-1),(0,0); ELU is fused into the 2x2 kernels' LDS staging; the positional
encoding is generated on the device once and cached.
"""

import torch
from torch import nn

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models import base


def _elu_conv_elu(conv, x, res=None, n_skip=0):
    """elu(conv(elu(x))) (+ res): both activations and the add are fused into the convolution."""
    return conv(x, in_act="elu", out_act="elu", res=res, n_skip=n_skip)


class ResidualBlock(nn.Module):
    """Residual block with a gated (identity * sigmoid) activation."""

    def __init__(self, n_channels):
        super().__init__()
        self._input_conv = pg_nn.Conv2d(
            in_channels=n_channels, out_channels=n_channels, kernel_size=2, padding=1
        )
        self._output_conv = pg_nn.Conv2d(
            in_channels=n_channels, out_channels=2 * n_channels, kernel_size=2, padding=1
        )
        self._activation = pg_nn.GatedActivation(activation_fn=nn.Identity())

    def forward(self, x, extra_skip=False):
        """extra_skip=True (extension) also returns one more pass-through alias of x (for the block's
        final residual): x then has a single autograd consumer — the first convolution — whose data
        gradient adds the skip gradients in its epilogue."""
        _, _, h, w = x.shape
        k = 2 if extra_skip else 1
        # reference: elu(conv(elu(x)))[:, :, :h, :w] -> conv[:, :, :h, :w] -> gate -> + x
        if self._input_conv.mfma_ok(x, (h, w)) and self._output_conv.mfma_ok(x, (h, w)):
            # the inner ELU lives in the first convolution's epilogue; its derivative (from the
            # stored output) in the second convolution's data-gradient epilogue
            out, *xs = self._input_conv(x, crop=(h, w), in_act="elu", out_act="elu", out_pre_scaled=True,
                                        n_skip=k)
            if self._output_conv.gate_ok(out, (h, w)):
                # round 6: the gate and the block's residual add in the second convolution's launch (the wide kernel's two
                # 64-channel chunks are the gate's halves): the standalone gate kernel read the 128-channel output back
                y = self._output_conv(out, crop=(h, w), in_post="elu", gate=self._activation._gate, gate_res=xs[0])
                return (y, xs[1]) if extra_skip else y
            out = self._output_conv(out, crop=(h, w), in_post="elu")
        else:
            out, *xs = self._input_conv(x, crop=(h, w), in_act="elu", n_skip=k)
            out = self._output_conv(out, crop=(h, w), in_act="elu")
        y = self._activation(out, res=xs[0])
        return (y, xs[1]) if extra_skip else y


class PixelSNAILBlock(nn.Module):
    def __init__(
        self,
        n_channels,
        input_img_channels=1,
        n_residual_blocks=2,
        attention_key_channels=4,
        attention_value_channels=32,
    ):
        super().__init__()

        def conv(in_channels):
            return pg_nn.Conv2d(in_channels, out_channels=n_channels, kernel_size=1)

        self._residual = nn.Sequential(
            *[ResidualBlock(n_channels) for _ in range(n_residual_blocks)]
        )
        self._attention = pg_nn.CausalAttention(
            in_channels=n_channels + 2,
            embed_channels=attention_key_channels,
            out_channels=attention_value_channels,
            mask_center=True,
            extra_input_channels=input_img_channels,
        )
        self._residual_out = conv(n_channels)
        self._attention_out = conv(attention_value_channels)
        self._out = conv(n_channels)

    def forward(self, x, input_img, *, add_input=False):
        """add_input=True (extension) returns x + block(x): the model loop's residual
        (pixel_snail.py:186) fused into the block's last convolution."""
        # every tensor with several readers goes through the FIRST reader's convolution with pass-through
        # aliases for the others: no gradient-sum kernels in backward (ops.conv2d_taps, n_skip)
        blocks = list(self._residual)
        x_skip = None
        res = x
        for i, rb in enumerate(blocks):
            if i == 0 and add_input:
                res, x_skip = rb(res, extra_skip=True)
            else:
                res = rb(res)
        if add_input and x_skip is None:
            x_skip = x
        # round 6: both = elu(a) + elu(r) has one reader, the last convolution — its data gradient's epilogue multiplies by the two
        # ELU derivatives (from both and r_out) and writes BOTH producers' gradients (ops.GradSlot carries the second); the two
        # producers skip their pg_act_bwd_from_out launches (out_pre_scaled). Forward launches are the same either way.
        dual = (res.requires_grad and self._out.dual_ok(res) and self._residual_out.mfma_ok(res)
                and self._attention_out.mfma_ok(res))  # (mfma_ok reads the image size only: the attention output has it too)
        r_out, res_skip = self._residual_out(res, in_act="elu", out_act="elu", out_pre_scaled=dual, n_skip=1)
        pos = pg_nn.image_positional_encoding(input_img.shape, res.device)
        attn = self._attention((pos, res_skip), input_img)  # cat(pos, res[, img]) happens inside, once
        if dual:
            slot = ops.GradSlot()
            both = self._attention_out(attn, in_act="elu", out_act="elu", res=r_out, out_pre_scaled=True, res_slot=slot)
            return self._out(both, in_act="elu", out_act="elu", res=x_skip if add_input else None, in_sum=(r_out, slot))
        both = _elu_conv_elu(self._attention_out, attn, res=r_out)  # elu(conv(elu(attn))) + res
        return _elu_conv_elu(self._out, both, res=x_skip if add_input else None)


class PixelSNAIL(base.AutoregressiveModel):
    _row_decode = True  # every layer is row-causal: sample() runs row by row (models/base.py)
    _row_decode_min_batch = 64  # its row step (attention over a growing prefix) runs eagerly: pays off at larger batches

    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        n_channels=64,
        n_pixel_snail_blocks=8,
        n_residual_blocks=2,
        attention_key_channels=4,
        attention_value_channels=32,
        sample_fn=None,
    ):
        super().__init__(sample_fn)
        self._input = pg_nn.CausalConv2d(
            mask_center=True,
            in_channels=in_channels,
            out_channels=n_channels,
            kernel_size=3,
            padding=1,
        )
        self._pixel_snail_blocks = nn.ModuleList(
            [
                PixelSNAILBlock(
                    n_channels=n_channels,
                    input_img_channels=in_channels,
                    n_residual_blocks=n_residual_blocks,
                    attention_key_channels=attention_key_channels,
                    attention_value_channels=attention_value_channels,
                )
                for _ in range(n_pixel_snail_blocks)
            ]
        )
        self._output = nn.Sequential(
            pg_nn.Conv2d(in_channels=n_channels, out_channels=n_channels // 2, kernel_size=1),
            pg_nn.Conv2d(in_channels=n_channels // 2, out_channels=out_channels, kernel_size=1),
        )

    def forward(self, x):
        input_img = x
        x = self._input(x)
        for block in self._pixel_snail_blocks:
            x = block(x, input_img, add_input=True)
        return self._output[1](self._output[0](x))


def reproduce(n_epochs=457, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (pixel_snail.py:190-254: same model
    hyper-parameters, Adam lr 1e-3, per-batch lr decay 0.999977) on the MI355X path. Arguments as the reference;
    `debug_loader` replaces both loaders (any iterable of (x, y) batches). Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: PixelSNAIL(in_channels=1, out_channels=1, n_channels=64, n_pixel_snail_blocks=8, n_residual_blocks=2,
                           attention_value_channels=32, attention_key_channels=4),
        loaders=recipes.binarized_mnist, loss_fn=recipes.bce_loss, lr=1e-3, lr_decay=0.999977,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
