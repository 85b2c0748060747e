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
      out_act: "relu" | "elu" applied to conv + bias BEFORE `res` is added (epilogue fusion).
      out_pre_scaled / in_post: the producer / consumer halves of a fused activation between two
              convolutions (see ops.conv2d_taps); only valid when both report mfma_ok().
    """

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if _pair(self.dilation) != (1, 1) or self.groups != 1:
            raise ValueError("pytorch_generative_amd.nn.Conv2d supports dilation=1, groups=1")
        if self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise ValueError("pytorch_generative_amd.nn.Conv2d supports integer zero padding only")
        self._down2 = False
        if _pair(self.stride) != (1, 1):
            # the one strided shape on the path: the VAE encoder's 4x4 / stride 2 / padding 1
            # down-sampling convolution (reference models/vae/vaes.py:153-160)
            if (_pair(self.stride), _pair(self.kernel_size), _pair(self.padding)) != ((2, 2), (4, 4), (1, 1)):
                raise ValueError(
                    "pytorch_generative_amd.nn.Conv2d supports stride 1, or stride 2 with kernel 4 and padding 1"
                )
            self._down2 = True
        self._spec = None

    def _active_taps(self):
        return None  # all taps

    def _conv_spec(self):
        if self._spec is None:
            kh, kw = _pair(self.kernel_size)
            ph, pw = _pair(self.padding)
            self._spec = ops.ConvSpec(kh, kw, ph, pw, active=self._active_taps(), wgrad_all=True)
        return self._spec

    def mfma_ok(self, x, crop=None):
        """True when this convolution and its data gradient run on the matrix-core kernels (then
        out_act / out_pre_scaled / in_post are available)."""
        return (not self._down2) and ops.conv_mfma_ok(x, self.weight, self._conv_spec(), crop)

    # ---- row-cached incremental sampling (ops.RowDecode) ------------------------------------------
    def _row_reset(self):
        self._row_band = None
        self._row_spec_cached = None

    def _row_forward(self, ctx, x, crop, in_act, res, out_act):
        """x is row `ctx.row` of this convolution's input: evaluates the same row of the output from
        a band of the k + 1 most recent input rows (k = how far the active taps reach upward)."""
        if self._down2:
            raise ValueError("row decode: strided convolutions are not row-causal")
        spec = self._conv_spec()
        reach = [spec.pad_h - u for u, _ in spec.fwd_taps]  # rows above the output row a tap reads
        if min(reach) < 0:
            raise ValueError("row decode: a tap below the output row (the layer is not row-causal)")
        k = max(reach)
        n, c, one, w = x.shape
        if one != 1:
            raise ValueError("row decode expects one image row")
        if getattr(self, "_row_spec_cached", None) is None:
            self._row_spec_cached = ops.ConvSpec(spec.kh, spec.kw, spec.pad_h - k, spec.pad_w,
                                                 active=spec.fwd_taps)
        if k == 0:
            band = x
        else:
            if getattr(self, "_row_band", None) is None or self._row_band.shape != (n, c, k + 1, w):
                self._row_band = torch.zeros((n, c, k + 1, w), device=x.device, dtype=x.dtype)
            band = self._row_band
            band[:, :, k:, :].copy_(x)
        ow = crop[1] if crop is not None else spec.full_out(1, w)[1]
        y = ops.conv2d_taps(band, self.weight, self.bias, self._row_spec_cached, out_hw=(1, ow),
                            in_act=_ACTS[in_act], res=None if out_act is not None else res)
        if out_act is not None:  # (one row: the unfused epilogue is three tiny launches)
            y = ops._Act.apply(y, _ACTS[out_act])
            if res is not None:
                y = ops.add(y, res)
        if ctx.commit and k > 0:  # the row is final: it becomes the newest cached row
            band[:, :, :k, :].copy_(band[:, :, 1:, :].clone())
        return y

    def two_residuals_ok(self, x, crop=None):
        return (not self._down2) and ops.RowDecode.current is None and ops.conv_two_residuals_ok(
            x, self.weight, self._conv_spec(), crop)

    def forward_cat2(self, xa, xb, *, in_act=None, skip_b=False):
        """self(torch.cat((xa, xb), dim=1), in_act=in_act) for a 1x1 convolution WITHOUT the concatenated tensor:
        W [xa; xb] = W[:, :Ca] xa + W[:, Ca:] xb, the second product chained through the first one's residual input
        (extension; VD-VAE's posterior reads cat(x, mixin), vd_vae.py:177).
        skip_b=True returns (y, xb_alias): a pass-through alias of xb for its NEXT reader, whose gradient this convolution's
        data-gradient kernel adds in its epilogue (ops.conv2d_taps, n_skip) — a tensor read by k blocks then needs no
        gradient-sum kernels of autograd at all."""
        if (self._down2 or ops.RowDecode.current is not None or tuple(self.weight.shape[2:]) != (1, 1)
                or xa.shape[1] + xb.shape[1] != self.weight.shape[1]):
            y = self.forward(torch.cat((xa, xb), dim=1), in_act=in_act)
            return (y, xb) if skip_b else y
        wa, wb = ops.split_in_channels(self.weight, xa.shape[1])
        spec = self._conv_spec()
        h = ops.conv2d_taps(xa, wa, None, spec, in_act=_ACTS[in_act])
        return ops.conv2d_taps(xb, wb, self.bias, spec, in_act=_ACTS[in_act], res=h, bias_param=self.bias,
                               n_skip=1 if skip_b else 0)

    def gate_ok(self, x, crop=None):
        """True if forward(..., gate=...) can run this convolution's GatedActivation in the same launch (ops.conv_gate_ok)."""
        return (not self._down2) and ops.RowDecode.current is None and ops.conv_gate_ok(x, self.weight, self._conv_spec(), crop)

    def dual_ok(self, x, crop=None):
        """True if forward(x, in_act="elu", in_sum=(r, slot)) can compute both gradients of x = elu(a) + r in its data gradient's
        epilogue (ops.conv_dual_ok: a 1x1 convolution on the bf16x3 1x1 kernel; training graph only)."""
        return ((not self._down2) and ops.RowDecode.current is None and crop is None and self.mfma_ok(x, crop)
                and ops.conv_dual_ok(x, self.weight, self._conv_spec()))

    def forward(self, x, *, crop=None, in_act=None, res=None, out_act=None, out_pre_scaled=False,
                in_post=None, n_skip=0, res2=None, gate=None, gate_res=None, res_slot=None, in_sum=None):
        """n_skip > 0 (extension) returns (y, x_1, .., x_n): pass-through aliases of x for the skip
        connections that also read x, see ops.conv2d_taps.
        gate = ops.GATE_TANH / GATE_IDENTITY (extension, round 6; only where gate_ok()): returns the GatedActivation of this
        convolution's output (conv + res, + gate_res behind the gate) — half the channels — from the same launch.
        in_sum = (r, slot) / res_slot = slot (extension, round 6; only where the consumer's dual_ok()): the two ends of the dual
        data gradient, see ops.GradSlot. Forward values do not change."""
        if res_slot is not None or in_sum is not None:
            if (gate is not None or res2 is not None or n_skip or self._down2 or ops.RowDecode.current is not None
                    or not self.mfma_ok(x, crop)):
                raise ValueError("Conv2d: res_slot / in_sum need the plain matrix-core training path (check dual_ok first)")
            return ops.conv2d_taps(
                x, self.weight, self.bias, self._conv_spec(), out_hw=crop, in_act=_ACTS[in_act],
                res=res, weight_param=self.weight, bias_param=self.bias, out_act=_ACTS[out_act],
                out_pre_scaled=out_pre_scaled, in_post=_ACTS[in_post], res_slot=res_slot, in_sum=in_sum,
            )
        if gate is not None:
            if res2 is not None or out_act is not None or out_pre_scaled or not self.gate_ok(x, crop):
                raise ValueError("Conv2d: gate= needs a convolution (at most one residual) on a shape gate_ok() accepts")
            if n_skip and not x.requires_grad:
                y = ops.conv2d_taps(x, self.weight, self.bias, self._conv_spec(), out_hw=crop, in_act=_ACTS[in_act], res=res,
                                    weight_param=self.weight, bias_param=self.bias, in_post=_ACTS[in_post],
                                    gate=gate, gate_res=gate_res)
                return (y,) + (x,) * n_skip
            return ops.conv2d_taps(x, self.weight, self.bias, self._conv_spec(), out_hw=crop, in_act=_ACTS[in_act], res=res,
                                   weight_param=self.weight, bias_param=self.bias, in_post=_ACTS[in_post], n_skip=n_skip,
                                   gate=gate, gate_res=gate_res)
        if res2 is not None and not self.two_residuals_ok(x, crop):
            y = self.forward(x, crop=crop, in_act=in_act, res=res, out_act=out_act,
                             out_pre_scaled=out_pre_scaled, in_post=in_post, n_skip=n_skip)
            return (ops.add(y[0], res2),) + tuple(y[1:]) if n_skip else ops.add(y, res2)
        if n_skip:
            ctx = ops.RowDecode.current
            if ctx is not None or self._down2 or not x.requires_grad:
                y = self.forward(x, crop=crop, in_act=in_act, res=res, out_act=out_act,
                                 out_pre_scaled=out_pre_scaled, in_post=in_post, res2=res2)
                return (y,) + (x,) * n_skip
        ctx = ops.RowDecode.current
        if ctx is not None:
            return self._row_forward(ctx, x, crop, in_act, res, out_act)
        if self._down2:
            if out_act is not None or in_post is not None:
                raise ValueError("Conv2d: fused output activations are not available for stride 2")
            return self._forward_down2(x, in_act, res)
        if (out_pre_scaled or in_post is not None) and not self.mfma_ok(x, crop):
            # the two halves of a fused activation between two convolutions only exist on the matrix-core
            # path; the caller must have checked mfma_ok() on BOTH convolutions (fail here, not in backward)
            raise ValueError("Conv2d: out_pre_scaled / in_post need the matrix-core path for this shape "
                             "(check mfma_ok first)")
        if out_act is not None and not out_pre_scaled and not self.mfma_ok(x, crop):
            # shapes the matrix-core path does not take: same result from separate kernels
            y = ops.conv2d_taps(x, self.weight, self.bias, self._conv_spec(), out_hw=crop,
                                in_act=_ACTS[in_act], weight_param=self.weight, bias_param=self.bias)
            y = ops._Act.apply(y, _ACTS[out_act])
            y = y if res is None else ops.add(y, res)
            return (y,) + (x,) * n_skip if n_skip else y
        return ops.conv2d_taps(
            x, self.weight, self.bias, self._conv_spec(), out_hw=crop, in_act=_ACTS[in_act],
            res=res, weight_param=self.weight, bias_param=self.bias, out_act=_ACTS[out_act],
            out_pre_scaled=out_pre_scaled, in_post=_ACTS[in_post], n_skip=n_skip, res2=res2,
        )

    def _forward_down2(self, x, in_act, res):
        """out[a, b] = sum_{u,v} w[u, v] x[2a + u - 1, 2b + v - 1] evaluated on the four 2x2 phases
        of x: phase 0 (even rows/cols) pairs with taps u in (1, 3) at offsets (0, +1), phase 1 with
        u in (0, 2) at offsets (-1, 0) — four stride-1 2x2 tap convolutions chained through the
        fused residual input."""
        xs = ops.phase_split4(x)  # four separate tensors: their gradients come back as four tensors, interleaved by one launch
        _, _, h, w = xs[0].shape
        out = res
        wph = ops.phase_weights(self.weight)  # [2 pr + pc] = weight[:, :, (1 - pr)::2, (1 - pc)::2], one copy
        for pr in (0, 1):
            for pc in (0, 1):
                wsel = wph[2 * pr + pc]
                spec = _PHASE_SPECS[(pr, pc)]
                last = (pr, pc) == (1, 1)
                out = ops.conv2d_taps(xs[2 * pr + pc], wsel, self.bias if last else None, spec,
                                      out_hw=(h, w), in_act=_ACTS[in_act], res=out)
        return out


# phase p of the input contributes through a 2-tap filter at offsets (0, +1) (p = 0, padding 0)
# or (-1, 0) (p = 1, padding 1); see Conv2d._forward_down2 / ConvTranspose2d.forward.
_PHASE_SPECS = {
    (pr, pc): ops.ConvSpec(2, 2, pr, pc) for pr in (0, 1) for pc in (0, 1)
}


class ConvTranspose2d(nn.ConvTranspose2d):
    """The VAE decoder's 4x4 / stride 2 / padding 1 transposed convolution
    (reference models/vae/vaes.py:228-235) on the HIP tap kernels:
    out[2a + p] = sum_{u = p+1 (mod 2)} w[u] x[a + (p + 1 - u) / 2] — every output phase is a
    stride-1 2x2 tap convolution of x; the four phases are interleaved by one merge kernel."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        ok = (_pair(self.stride), _pair(self.kernel_size), _pair(self.padding)) == ((2, 2), (4, 4), (1, 1))
        if not ok or _pair(self.output_padding) != (0, 0) or self.groups != 1 or _pair(self.dilation) != (1, 1):
            raise ValueError("pytorch_generative_amd.nn.ConvTranspose2d supports kernel 4, stride 2, padding 1")

    def forward(self, x, *, in_act=None):
        # output phase 0 uses kernel rows (3, 1) at offsets (-1, 0), phase 1 rows (2, 0) at (0, +1):
        # [2 pr + pc] = weight.transpose(0, 1)[:, :, (1 - pr)::2, (1 - pc)::2].flip(2, 3), one copy for all four
        wph = ops.phase_weights(self.weight, transposed=True)
        phases = []
        xs = [x] * 4
        for pr in (0, 1):
            for pc in (0, 1):
                k = 2 * pr + pc
                spec = _PHASE_SPECS[(1 - pr, 1 - pc)]
                if k == 0 and x.requires_grad:
                    # x has four readers: the other three read pass-through aliases, whose gradients the first
                    # convolution's data-gradient kernel adds in its epilogue (ops.conv2d_taps, n_skip)
                    # (bias_param: with a flat-gradient sink the four bias gradients are added by the kernels — without it autograd
                    # summed them with three ATen adds per layer)
                    y, *al = ops.conv2d_taps(x, wph[k], self.bias, spec, out_hw=x.shape[2:], in_act=_ACTS[in_act],
                                             n_skip=3, bias_param=self.bias)
                    xs = [x] + list(al)
                else:
                    y = ops.conv2d_taps(xs[k], wph[k], self.bias, spec, out_hw=x.shape[2:], in_act=_ACTS[in_act],
                                        bias_param=self.bias)
                phases.append(y)
        return ops.phase_merge4(phases)  # interleaves the four phase outputs without a stacked copy


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

    def forward(self, x, *, crop=None, in_act=None, res=None, **kw):
        ops.mul_inplace_(self.weight.data, self.mask)
        return super().forward(x, crop=crop, in_act=in_act, res=res, **kw)


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

    def forward(self, x, *, res=None):
        """res (extension): tensor added to the gated output in the same kernel."""
        _, c, _, _ = x.shape
        assert c % 2 == 0, "x must have an even number of channels."
        return ops.gated_activation(x, self._gate, res)


class GatedConv(nn.Module):
    """Convolution to 2 x out_channels followed by GatedActivation, as ONE operator: activation_fn(f) * sigmoid(g)
    with f / g the two channel halves of conv(x).

    BASELINE.json's north_star lists `GatedConv` on the nn surface; the reference spells it out as a convolution
    feeding `GatedActivation` (nn/convolution.py:46-66) — GatedPixelCNNLayer's `_vstack_1xN` / `_hstack_Nx1` -> gate
    (models/autoregressive/gated_pixel_cnn.py:63-96, :99-130) and PixelSNAIL's ResidualBlock
    (pixel_snail.py:41-56: `Conv2d(n, 2n, 2, padding=1)` -> `GatedActivation(Identity)`). Parameter names are
    `conv.weight` / `conv.bias` (+ `conv.mask` when causal).

    mask_center: None = plain convolution, True / False = CausalConv2d of type A / B.
    forward(x, crop=, in_act=, res=): `crop` / `in_act` as Conv2d.forward; `res` is added to the gated output in
    the gate kernel. Where the convolution has 128 output channels on the wide bf16x3 kernel (Conv2d.gate_ok: PixelSNAIL's
    ResidualBlock) the gate and `res` run in the convolution's own epilogue — one launch, the 128-channel pre-gate tensor is
    written once for backward and never read back in forward.
    """

    def __init__(self, in_channels, out_channels, kernel_size=1, padding=0, mask_center=None,
                 activation_fn=torch.tanh, bias=True):
        super().__init__()
        if mask_center is None:
            self.conv = Conv2d(in_channels, 2 * out_channels, kernel_size, padding=padding, bias=bias)
        else:
            self.conv = CausalConv2d(mask_center, in_channels, 2 * out_channels, kernel_size, padding=padding,
                                     bias=bias)
        self.gate = GatedActivation(activation_fn)

    def forward(self, x, *, crop=None, in_act=None, res=None):
        if self.conv.gate_ok(x, crop):
            return self.conv(x, crop=crop, in_act=in_act, gate=self.gate._gate, gate_res=res)
        return self.gate(self.conv(x, crop=crop, in_act=in_act), res=res)


class NCHWLayerNorm(nn.LayerNorm):
    """LayerNorm over the channel dimension of NCHW tensors (no permutes: one lane per pixel)."""

    def forward(self, x, *, skip=False):
        """skip=True (extension) returns (LN(x), x): feed the second value to the residual add of
        `x + f(LN(x))` and its gradient is folded into LN's backward kernel."""
        if len(self.normalized_shape) != 1 or not self.elementwise_affine:
            raise ValueError("NCHWLayerNorm expects a single affine channel dimension")
        if skip and not ops.FUSE_LNSKIP:
            return ops.nchw_layernorm(x, self.weight, self.bias, self.eps), x
        if skip:
            return ops.nchw_layernorm_skip(x, self.weight, self.bias, self.eps)
        return ops.nchw_layernorm(x, self.weight, self.bias, self.eps)
