"""PixelCNN++ on the MI355X operator path — SURVEY.md §8(f) rank 4 (BASELINE.json configs[2] names it).

The reference repository has NO PixelCNN++; this module follows the published architecture — Salimans,
Karpathy, Chen, Kingma, "PixelCNN++", ICLR 2017 (section 2: discretized logistic mixture likelihood,
conditioning on whole pixels, two streams of down-shifted / down-right-shifted convolutions, gated
residual blocks on concatenated ELUs, down- and up-sampling by strided convolutions with short-cut
connections across the three resolutions). Its CPU restatement is oracle/pixelcnnpp.py, written on
torch's strided / transposed convolutions; since there is no reference implementation, parity means HIP
path == that oracle plus the autoregressive property (tests/test_gpu_f4.py, tests/test_dmol_cpu.py).

How the pieces map onto this package's kernels:
  * down-shifted (2 x 3) and down-right-shifted (2 x 2) convolutions, and the extra row / column shift
    of the input layers, are tap lists of the masked-convolution kernels (padding + crop, no shifting
    copies);
  * stride-2 down-sampling = the stride-1 shifted convolution followed by ops.subsample2; stride-2
    up-sampling = ops.zero_insert2 followed by the same shifted convolution (the transposed convolution
    of the paper written as a gather);
  * the gate a * sigmoid(b) + residual is GatedActivation's fused kernel; concat_elu is ops.concat_elu;
  * the loss is ops.dmol_loss_sum_mean on images scaled to [-1, 1].
"""

import torch
from torch import nn

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models import base


class ShiftedConv2d(pg_nn.Conv2d):
    """kind "ds": down-shifted — the (kh x kw) window ends at the output's row and is centred on its
    column; kind "drs": down-right-shifted — the window ends at the output's row AND column.
    shift_down / shift_right move the whole window one more row up / column left (the input layers)."""

    def __init__(self, in_channels, out_channels, kernel_size, kind, shift_down=False, shift_right=False):
        kh, kw = kernel_size
        assert kind in ("ds", "drs") and (kind == "drs" or kw % 2 == 1)
        ph = kh - 1 + int(shift_down)
        pw = (kw - 1 if kind == "drs" else (kw - 1) // 2) + int(shift_right)
        super().__init__(in_channels, out_channels, kernel_size=(kh, kw), padding=(ph, pw))

    def forward(self, x, **kw):
        return super().forward(x, crop=(x.shape[2], x.shape[3]), **kw)


class GatedResnet(nn.Module):
    """x + a' * sigmoid(b'), [a' | b'] = conv2(concat_elu(conv1(concat_elu(x)) + nin(concat_elu(aux))))."""

    def __init__(self, n_filters, kind, aux_channels=0):
        super().__init__()
        ks = (2, 3) if kind == "ds" else (2, 2)
        self._conv_in = ShiftedConv2d(2 * n_filters, n_filters, ks, kind)
        self._nin = pg_nn.Conv2d(2 * aux_channels, n_filters, kernel_size=1) if aux_channels else None
        self._conv_out = ShiftedConv2d(2 * n_filters, 2 * n_filters, ks, kind)
        self._gate = pg_nn.GatedActivation(activation_fn=nn.Identity())

    def forward(self, x, aux=None):
        # x has two readers (the activation and the gate's residual): two pass-through aliases, whose gradients one launch sums
        # (ops.fanout) instead of autograd's add kernel
        x, x_res = ops.fanout(x, 2)
        c1 = self._conv_in(ops.concat_elu(x))
        if aux is not None:
            c1 = self._nin(ops.concat_elu(aux), res=c1)
        c2 = self._conv_out(ops.concat_elu(c1))
        return self._gate(c2, res=x_res)


class PixelCNNpp(base.AutoregressiveModel):
    """forward(x) -> (N, 10 * n_mix, H, W) mixture parameters for images x in [-1, 1] with 3 channels
    (H, W multiples of 4). Paper configuration: n_filters=160, n_resnet=5, n_mix=10."""

    def __init__(self, in_channels=3, n_filters=160, n_resnet=5, n_mix=10, sample_fn=None):
        super().__init__(sample_fn)
        if in_channels != 3:
            raise ValueError("PixelCNNpp: the discretized logistic mixture conditions R, G, B sub-pixels (3 channels)")
        f, cin = n_filters, in_channels + 1  # + a channel of ones (so that the shifted convolutions see the border)
        self._n_mix, self._n_resnet = n_mix, n_resnet
        self._u_in = ShiftedConv2d(cin, f, (2, 3), "ds", shift_down=True)
        self._ul_in_a = ShiftedConv2d(cin, f, (1, 3), "ds", shift_down=True)
        self._ul_in_b = ShiftedConv2d(cin, f, (2, 1), "drs", shift_right=True)
        self._up_u = nn.ModuleList([nn.ModuleList([GatedResnet(f, "ds") for _ in range(n_resnet)]) for _ in range(3)])
        self._up_ul = nn.ModuleList([nn.ModuleList([GatedResnet(f, "drs", aux_channels=f) for _ in range(n_resnet)])
                                     for _ in range(3)])
        self._down_u_conv = nn.ModuleList([ShiftedConv2d(f, f, (2, 3), "ds") for _ in range(2)])
        self._down_ul_conv = nn.ModuleList([ShiftedConv2d(f, f, (2, 2), "drs") for _ in range(2)])
        counts = [n_resnet, n_resnet + 1, n_resnet + 1]
        self._dn_u = nn.ModuleList([nn.ModuleList([GatedResnet(f, "ds", aux_channels=f) for _ in range(c)])
                                    for c in counts])
        self._dn_ul = nn.ModuleList([nn.ModuleList([GatedResnet(f, "drs", aux_channels=2 * f) for _ in range(c)])
                                     for c in counts])
        self._up_u_conv = nn.ModuleList([ShiftedConv2d(f, f, (2, 3), "ds") for _ in range(2)])
        self._up_ul_conv = nn.ModuleList([ShiftedConv2d(f, f, (2, 2), "drs") for _ in range(2)])
        self._out = pg_nn.Conv2d(f, 10 * n_mix, kernel_size=1)

    def forward(self, x):
        return self._net(x)

    def _net(self, x):
        """The network body on images in [-1, 1]. forward() and sample() both call THIS, so that a subclass whose
        forward() rescales its input (reproduce()'s wrapper for [0, 1] loaders) does not rescale the sampler's canvas."""
        n, _, h, w = x.shape
        if h % 4 or w % 4:
            raise ValueError("PixelCNNpp: H and W must be multiples of 4 (two stride-2 levels)")
        xp = ops.concat_channels([x, torch.ones((n, 1, h, w), device=x.device, dtype=x.dtype)])
        # Every stream tensor of the up pass has two readers — the next layer and, later, the down pass's short-cut (a tensor
        # of the u stream a third one: the ul stream's layer of the same step): each reader takes its own pass-through alias
        # (ops.fanout), so that the gradients are summed by one launch per tensor instead of autograd's chains of add kernels.
        # `u_cur` / `ul_cur` = the alias for the next layer, u / ul = the stacks of short-cut aliases.
        u_cur, a = ops.fanout(self._u_in(xp), 2)
        ul_cur, b = ops.fanout(self._ul_in_b(xp, res=self._ul_in_a(xp)), 2)
        u, ul = [a], [b]
        for s in range(3):  # up pass: towards the coarse resolution
            for ru, rul in zip(self._up_u[s], self._up_ul[s]):
                u_cur, u_aux, a = ops.fanout(ru(u_cur), 3)
                ul_cur, b = ops.fanout(rul(ul_cur, aux=u_aux), 2)
                u.append(a)
                ul.append(b)
            if s < 2:
                u_cur, a = ops.fanout(ops.subsample2(self._down_u_conv[s](u_cur)), 2)
                ul_cur, b = ops.fanout(ops.subsample2(self._down_ul_conv[s](ul_cur)), 2)
                u.append(a)
                ul.append(b)
        hu, hul = u.pop(), ul.pop()  # (the last tensors' "next layer" aliases stay unused: they carry no gradient)
        for s in range(3):  # down pass: back to the fine resolution, short-cuts from the up pass
            for ru, rul in zip(self._dn_u[s], self._dn_ul[s]):
                hu, hu_cat = ops.fanout(ru(hu, aux=u.pop()), 2)  # read by the ul stream's concatenation and by the next layer
                hul = rul(hul, aux=ops.concat_channels([hu_cat, ul.pop()]))
            if s < 2:
                hu = self._up_u_conv[s](ops.zero_insert2(hu))
                hul = self._up_ul_conv[s](ops.zero_insert2(hul))
        assert not u and not ul
        return self._out(hul, in_act="elu")

    @staticmethod
    def sample_from_mixture(params, n_mix):
        """One draw per image from the discretized logistic mixture at ONE pixel: params (N, 10 K) in the channel
        layout of the loss (mixture logits, then per sub-pixel means / log-scales / coefficients) -> (N, 3) in
        [-1, 1]. The published sampler (Salimans et al. 2017, section 2.1-2.2 and the authors' implementation):
        component by the Gumbel-max trick, a logistic variate by inverse CDF, then G and B shifted by their
        linear dependence on the already drawn sub-pixels (eq. 3), each clamped to the image range."""
        n, k = params.shape[0], n_mix
        logits = params[:, :k]
        rest = params[:, k:].reshape(n, 3, 3 * k)
        u = torch.rand_like(logits).clamp_(1e-5, 1.0 - 1e-5)
        sel = torch.argmax(logits - torch.log(-torch.log(u)), dim=1)                       # (N,)
        pick = lambda t: t.gather(2, sel.view(n, 1, 1).expand(n, 3, 1)).squeeze(2)          # noqa: E731  (N, 3)
        means = pick(rest[:, :, :k])
        log_scales = pick(rest[:, :, k:2 * k]).clamp(min=-7.0)
        coeffs = torch.tanh(pick(rest[:, :, 2 * k:]))
        v = torch.rand_like(means).clamp_(1e-5, 1.0 - 1e-5)
        x = means + torch.exp(log_scales) * (torch.log(v) - torch.log1p(-v))
        x0 = x[:, 0].clamp(-1.0, 1.0)
        x1 = (x[:, 1] + coeffs[:, 0] * x0).clamp(-1.0, 1.0)
        x2 = (x[:, 2] + coeffs[:, 1] * x0 + coeffs[:, 2] * x1).clamp(-1.0, 1.0)
        return torch.stack((x0, x1, x2), dim=1)

    @torch.no_grad()
    def sample(self, n_samples=None, conditioned_on=None, *, image_size=None):
        """Raster-order sampling as base.AutoregressiveModel.sample (reference models/base.py:97-120: one full
        forward per pixel, only the unknown entries replaced), with the draw made from the logistic mixture.
        Images live in [-1, 1]; entries of `conditioned_on` below -1 are the unknown ones. `image_size` (H, W)
        is needed when the model has not seen a batch yet. (The strided levels make the network non-row-causal
        for ops.RowDecode: full forwards only.)"""
        if conditioned_on is not None:
            canvas = conditioned_on.clone()
        else:
            assert n_samples is not None, 'Must provided one, and only one, of "n_samples" or "conditioned_on"'
            h, w = image_size if image_size is not None else (int(self._h), int(self._w))
            canvas = torch.full((n_samples, 3, h, w), -2.0, device=self.device)
        n, _, h, w = canvas.shape
        unknown = canvas < -1.0
        canvas = torch.where(unknown, torch.zeros_like(canvas), canvas)  # any finite value: later pixels are never read
        for row in range(h):
            for col in range(w):
                if not bool(unknown[:, :, row, col].any()):
                    continue
                params = self._net(canvas)[:, :, row, col]
                drawn = self.sample_from_mixture(params, self._n_mix)
                canvas[:, :, row, col] = torch.where(unknown[:, :, row, col], drawn, canvas[:, :, row, col])
        return canvas


class PixelCNNppUnitRange(PixelCNNpp):
    """PixelCNNpp for loaders that deliver images in [0, 1] (the recipe's model): forward() maps its input to the
    network's [-1, 1]; sample() takes / returns images in [0, 1] (negative entries of `conditioned_on` are the
    unknown ones, as in the reference's base.AutoregressiveModel.sample, models/base.py:97-120) and runs the
    sampler's conditioning forwards on the UNSCALED network body."""

    def forward(self, x):
        return self._net(x * 2.0 - 1.0)

    @torch.no_grad()
    def sample(self, n_samples=None, conditioned_on=None, *, image_size=None):
        if conditioned_on is not None:
            conditioned_on = torch.where(conditioned_on < 0, torch.full_like(conditioned_on, -2.0),
                                         conditioned_on * 2.0 - 1.0)
        out = super().sample(n_samples, conditioned_on, image_size=image_size)
        return (out + 1.0) * 0.5


def dmol_loss(x, _, preds, n_mix=10):
    """loss_fn(x, y, preds) of the recipe: images in [0, 1] (the loaders' range) are mapped to [-1, 1]."""
    return ops.dmol_loss_sum_mean(preds, x * 2.0 - 1.0, n_mix)


def reproduce(n_epochs=457, batch_size=16, log_dir="/tmp/run", n_gpus=1, device_id=0, debug_loader=None,
              n_filters=160, n_resnet=5, n_mix=10):
    """Training recipe in the shape of the reference's reproduce() functions (there is no PixelCNN++ in the
    reference): CIFAR-10-shaped batches, the paper's model size by default, Adam lr 1e-3 with the per-batch
    decay 0.999995 of the paper, the discretized logistic mixture loss. Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: PixelCNNppUnitRange(in_channels=3, n_filters=n_filters, n_resnet=n_resnet, n_mix=n_mix),
        loaders=lambda b: recipes.datasets.get_cifar10_loaders(b), loss_fn=lambda x, y, p: dmol_loss(x, y, p, n_mix),
        lr=1e-3, lr_decay=0.999995, n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
