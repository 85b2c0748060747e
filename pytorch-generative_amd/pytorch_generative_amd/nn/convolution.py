"""Convolution-side operators of the hot path, on hand-written gfx950 kernels.

Mirrors the public surface of the reference's pytorch_generative/nn/convolution.py
(CausalConv2d :12-43, GatedActivation :46-66, NCHWLayerNorm :69-75) — same constructor
arguments, parameter/buffer names and shapes (so reference checkpoints load with
strict=True) — but forward/backward run in libpg_hip.so instead of ATen.
"""

import torch
from torch import nn

from pytorch_generative_amd import ops

_ACTS = {None: ops.ACT_NONE, "relu": ops.ACT_RELU, "elu": ops.ACT_ELU, "gelu": ops.ACT_GELU}


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Conv2d(nn.Conv2d):
    """Stride-1 nn.Conv2d whose arithmetic is the HIP tap-list convolution.

    Extra (keyword-only) forward arguments expose the fusions the kernels offer:
      crop:   (h, w) — compute only the first h rows / w columns of the output (the
              reference computes the padded conv and slices it, e.g. gated_pixel_cnn.py:115,
              pixel_snail.py:54-55).
      in_act: "relu" | "elu" | "gelu" applied to the input on load.
      res:    tensor added to the output (fused residual).
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if _pair(self.stride) != (1, 1) or _pair(self.dilation) != (1, 1) or self.groups != 1:
            raise ValueError("pytorch_generative_amd.nn.Conv2d supports stride=1, dilation=1, groups=1")
        if self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise ValueError("pytorch_generative_amd.nn.Conv2d supports integer zero padding only")
        self._spec = None

    def _active_taps(self):
        return None  # all taps

    def _conv_spec(self):
        if self._spec is None:
            kh, kw = _pair(self.kernel_size)
            ph, pw = _pair(self.padding)
            self._spec = ops.ConvSpec(kh, kw, ph, pw, active=self._active_taps(), wgrad_all=True)
        return self._spec

    def forward(self, x, *, crop=None, in_act=None, res=None):
        return ops.conv2d_taps(
            x, self.weight, self.bias, self._conv_spec(), out_hw=crop, in_act=_ACTS[in_act],
            res=res, weight_param=self.weight, bias_param=self.bias,
        )


class CausalConv2d(Conv2d):
    """A Conv2d masked so that each output pixel sees only pixels above / to the left.

    mask_center=True additionally hides the pixel itself ('type A' of the PixelCNN paper,
    False = 'type B'). Like the reference, there is no channel-wise autoregressive masking,
    the 0/1 `mask` buffer is part of the state_dict, and the masked entries of `weight` are
    zeroed in place on every forward. The kernel never touches masked taps at all.
    """

    def __init__(self, mask_center, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._mask_center = bool(mask_center)
        kh, kw = self.weight.shape[2], self.weight.shape[3]
        mask = torch.zeros_like(self.weight)
        mask[:, :, : kh // 2, :] = 1.0
        mask[:, :, kh // 2, : kw // 2 + (0 if self._mask_center else 1)] = 1.0
        self.register_buffer("mask", mask)

    def _active_taps(self):
        m = self.mask[0, 0].detach().to("cpu")
        return [(u, v) for u in range(m.shape[0]) for v in range(m.shape[1]) if m[u, v] != 0]

    def forward(self, x, *, crop=None, in_act=None, res=None):
        ops.mul_inplace_(self.weight.data, self.mask)
        return super().forward(x, crop=crop, in_act=in_act, res=res)


class GatedActivation(nn.Module):
    """activation_fn(f) * sigmoid(g), f / g = first / second half of the channels.

    The HIP kernel implements the two activations the reference's models use: torch.tanh
    (GatedPixelCNN) and identity (PixelSNAIL).
    """

    def __init__(self, activation_fn=torch.tanh):
        super().__init__()
        self._activation_fn = activation_fn
        if activation_fn is torch.tanh or isinstance(activation_fn, nn.Tanh):
            self._gate = ops.GATE_TANH
        elif isinstance(activation_fn, nn.Identity):
            self._gate = ops.GATE_IDENTITY
        else:
            raise ValueError(
                "GatedActivation on the HIP path supports torch.tanh and nn.Identity() only"
            )

    def forward(self, x):
        _, c, _, _ = x.shape
        assert c % 2 == 0, "x must have an even number of channels."
        return ops.gated_activation(x, self._gate)


class NCHWLayerNorm(nn.LayerNorm):
    """LayerNorm over the channel dimension of NCHW tensors (no permutes: one lane per pixel)."""

    def forward(self, x):
        if len(self.normalized_shape) != 1 or not self.elementwise_affine:
            raise ValueError("NCHWLayerNorm expects a single affine channel dimension")
        return ops.nchw_layernorm(x, self.weight, self.bias, self.eps)
