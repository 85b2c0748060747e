"""Very Deep VAE on the MI355X operator path (reference models/vae/vd_vae.py:48-412).

Same constructor, parameter names (= state_dict keys) and forward graph, including the
reference's quirks: the decoder's `_out` bottlenecks get the kernel size left over from the
ENCODER loop (the decoder loop assigns a misspelt variable, vd_vae.py:354 vs :361), `_prior` /
`_posterior` always use 3x3 bottlenecks, and last-conv weights are pre-scaled by
1/sqrt(total blocks) (:337,:365-366). GELUs are fused into the following convolution's input
load, residual adds into its epilogue; the Gaussian heads (split, KL, reparameterised sample) are
one fused kernel reading the conv output in place.
"""

from dataclasses import dataclass

import numpy as np
import torch
from torch import nn

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models.vae import vaes


@dataclass
class StackConfig:
    """Encoder / decoder block counts of one resolution level."""

    n_encoder_blocks: int
    n_decoder_blocks: int


DEFAULT_MODEL = [StackConfig(n_encoder_blocks=1, n_decoder_blocks=1) for _ in range(6)]


class BottleneckBlock(nn.Module):
    def __init__(self, in_channels, out_channels, bottleneck_channels, bottleneck_kernel_size=3,
                 is_residual=True):
        super().__init__()
        self._is_residual = is_residual
        padding = 1 if bottleneck_kernel_size == 3 else 0  # like the reference: k in {1, 3}
        self._net = nn.Sequential(
            nn.GELU(),
            pg_nn.Conv2d(in_channels, bottleneck_channels, kernel_size=1),
            nn.GELU(),
            pg_nn.Conv2d(bottleneck_channels, bottleneck_channels,
                         kernel_size=bottleneck_kernel_size, padding=padding),
            nn.GELU(),
            pg_nn.Conv2d(bottleneck_channels, bottleneck_channels,
                         kernel_size=bottleneck_kernel_size, padding=padding),
            nn.GELU(),
            pg_nn.Conv2d(bottleneck_channels, out_channels, kernel_size=1),
        )

    def forward(self, x, n_alias=0, x2=None, x2_alias=False):
        """n_alias > 0 (extension) returns (out, x_1, .., x_n): pass-through aliases of x for the caller's other
        readers of x (ops.conv2d_taps, n_skip): their gradients are added in the first convolution's data-gradient
        epilogue instead of by autograd's gradient-sum kernels. x2 (extension): the block reads cat((x, x2), dim=1)
        without that tensor being written (Conv2d.forward_cat2); x2_alias=True then returns (out, x2_next): the pass-through
        alias of x2 for its next reader."""
        res = None
        k = n_alias + (1 if self._is_residual else 0)
        if x2 is not None:
            if k:
                raise ValueError("BottleneckBlock: x2 is for the non-residual block without aliases")
            if x2_alias:
                h, x2_next = self._net[1].forward_cat2(x, x2, in_act="gelu", skip_b=True)
                al = []
            else:
                h, al = self._net[1].forward_cat2(x, x2, in_act="gelu"), []
        elif k:
            # the block's own residual add reads the first alias
            h, *al = self._net[1](x, in_act="gelu", n_skip=k)
            if self._is_residual:
                res, al = al[0], al[1:]
        else:
            h, al = self._net[1](x, in_act="gelu"), []
        h = self._net[3](h, in_act="gelu")
        h = self._net[5](h, in_act="gelu")
        out = self._net[7](h, in_act="gelu", res=res)
        if x2 is not None and x2_alias:
            return out, x2_next
        return (out, *al) if n_alias else out


class TopDownBlock(nn.Module):
    def __init__(self, n_channels, latent_channels, bottleneck_channels, bottleneck_kernel_size):
        super().__init__()
        self._n_channels = n_channels
        self._latent_channels = latent_channels
        self._prior = BottleneckBlock(n_channels, 2 * latent_channels + n_channels,
                                      bottleneck_channels, is_residual=False)
        self._posterior = BottleneckBlock(2 * n_channels, 2 * latent_channels,
                                          bottleneck_channels, is_residual=False)
        self._latents = pg_nn.Conv2d(latent_channels, n_channels, kernel_size=1)
        self._out = BottleneckBlock(n_channels, n_channels, bottleneck_channels,
                                    bottleneck_kernel_size=bottleneck_kernel_size, is_residual=True)

    def forward(self, x, mixin=None, more_readers=False):
        """more_readers=True (extension, training): also returns the pass-through alias of `mixin` for the NEXT top-down block of
        the stack — the encoder's feature map is read by every block of its resolution (vd_vae.py:177), and chained aliases
        let each block's data-gradient kernel add the later blocks' gradients instead of autograd's gradient-sum kernels."""
        c = self._latent_channels
        n, _, h, w = x.shape
        # x has three readers (the prior's first convolution, the concatenation, the residual add): the last two
        # read pass-through aliases handed out by the first
        prior, x_cat, x = self._prior(x, n_alias=2)  # prior = [p_mean | p_log_std | p_h]
        eps = vaes.draw_noise((n, c, h, w), x.device)
        if mixin is None:  # generation: sample from the prior
            z, kl_div = ops.gaussian_head_prior(prior, eps, c), None
            p_h = prior[:, 2 * c:]
        else:              # training: sample from the approximate posterior
            if more_readers:
                post, mixin = self._posterior(x_cat, x2=mixin, x2_alias=True)
            else:
                post = self._posterior(x_cat, x2=mixin)  # the block of cat((x, mixin)), concatenation not materialised
            # p_h = prior[:, 2c:] comes out of the head function: its gradient goes back into the prior's
            # gradient with one copy (no slice backward + full-size add)
            z, kl_div, p_h = ops.gaussian_head_pair(post, prior, eps, c, split_rest=True)
        if self._latents.two_residuals_ok(z):
            # x + p_h + latents(z): both adds in the 1x1 convolution's epilogue, p_h read in place (batch-strided)
            y = self._out(self._latents(z, res=x, res2=p_h))
        else:
            x_ph = ops.add(x, p_h.contiguous())
            y = self._out(self._latents(z, res=x_ph))
        return (y, kl_div, mixin) if more_readers else (y, kl_div)


class EncoderStack(nn.Module):
    def __init__(self, n_residual_blocks, pool, n_channels, bottleneck_channels,
                 bottleneck_kernel_size):
        super().__init__()
        self._residuals = nn.Sequential(*[
            BottleneckBlock(n_channels, n_channels, bottleneck_channels,
                            bottleneck_kernel_size=bottleneck_kernel_size, is_residual=True)
            for _ in range(n_residual_blocks)
        ])
        self._pool = nn.AvgPool2d(kernel_size=2, stride=2) if pool else None  # parameterless marker

    def forward(self, x):
        features = self._residuals(x)
        if self._pool is None:
            return features, features
        # the feature map has two kinds of readers — the pooling here and the decoder's top-down blocks (through `mixin`): the
        # latter read a pass-through alias whose gradient the pooling's backward kernel adds
        x, features = ops.avg_pool2(features, n_skip=1)
        return x, features


class DecoderStack(nn.Module):
    def __init__(self, n_topdown_blocks, unpool, n_channels, latent_channels, bottleneck_channels,
                 bottleneck_kernel_size):
        super().__init__()
        self._unpool = nn.Upsample(scale_factor=2, mode="nearest") if unpool else None  # marker
        self._topdowns = nn.ModuleList([
            TopDownBlock(n_channels, latent_channels, bottleneck_channels, bottleneck_kernel_size)
            for _ in range(n_topdown_blocks)
        ])

    def forward(self, x, mixin=None):
        if self._unpool is not None:
            x = ops.upsample2_nearest(x)
        kl_divs = []
        last = len(self._topdowns) - 1
        for i, topdown in enumerate(self._topdowns):
            if mixin is not None and i < last:
                x, kl_div, mixin = topdown(x, mixin, more_readers=True)
            else:
                x, kl_div = topdown(x, mixin)
            kl_divs.append(kl_div)
        return x, kl_divs


class VeryDeepVAE(vaes.VariationalAutoEncoder):
    def __init__(self, in_channels=1, out_channels=1, input_resolution=32,
                 stack_configs=DEFAULT_MODEL, latent_channels=4, hidden_channels=16,
                 bottleneck_channels=8, sample_fn=None):
        super().__init__(sample_fn)
        stack_configs = [c if isinstance(c, StackConfig) else StackConfig(*c) for c in stack_configs]
        self._hidden_channels = hidden_channels

        self._input = pg_nn.Conv2d(in_channels, hidden_channels, kernel_size=3, padding=1)
        self._encoder = nn.ModuleList()
        resolutions = [input_resolution // 2**i for i in range(len(stack_configs))]
        encoder_blocks = [conf.n_encoder_blocks for conf in stack_configs]
        total_encoder_blocks = sum(encoder_blocks)
        bottleneck_kernel_size = 3
        for i, (res, n_blocks) in enumerate(zip(resolutions, encoder_blocks)):
            bottleneck_kernel_size = 3 if res >= 3 else 1
            stack = EncoderStack(n_blocks, i < len(stack_configs) - 1, hidden_channels,
                                 bottleneck_channels, bottleneck_kernel_size)
            for block in stack._residuals:
                block._net[-1].weight.data /= np.sqrt(total_encoder_blocks)
            self._encoder.append(stack)

        self._biases = nn.ParameterList([
            nn.Parameter(torch.zeros(1, hidden_channels, size, size))
            for size in resolutions[1:] + [resolutions[-1]]
        ])

        self._decoder = nn.ModuleList()
        decoder_blocks = [conf.n_decoder_blocks for conf in stack_configs]
        total_decoder_blocks = sum(decoder_blocks)
        for i, (res, n_blocks) in enumerate(zip(reversed(resolutions), reversed(decoder_blocks))):
            # NB (reference quirk, vd_vae.py:354-361): the decoder stacks receive the kernel size
            # the ENCODER loop ended with, not one derived from their own resolution.
            stack = DecoderStack(n_blocks, i > 0, hidden_channels, latent_channels,
                                 bottleneck_channels, bottleneck_kernel_size)
            for block in stack._topdowns:
                block._out._net[-1].weight.data /= np.sqrt(total_decoder_blocks)
                block._latents.weight.data /= np.sqrt(total_decoder_blocks)
            self._decoder.append(stack)
        self._output = pg_nn.Conv2d(hidden_channels, out_channels, kernel_size=1)

    def _top(self, n):
        b = self._biases[-1]
        return ops.zeros((n,) + tuple(b.shape[1:]), b.device)

    def forward(self, x):
        """Returns (logits, kl_div) with kl_div the per-sample sum over all latents (not normalised)."""
        n = x.shape[0]
        x = self._input(x)
        mixins = []
        for stack in self._encoder:
            x, mixin = stack(x)
            mixins.append(mixin)

        x = self._top(n)
        kl_divs = []
        for stack, mixin, bias in zip(self._decoder, reversed(mixins), reversed(list(self._biases))):
            x = ops.add_broadcast_batch(x, bias)
            x, divs = stack(x, mixin)
            kl_divs.extend(divs)
        kl_div = ops.sum_vectors(kl_divs)  # torch.stack(kl_divs).sum(dim=0) in one launch, no stacked tensor
        return self._output(x), kl_div

    def _sample(self, n_samples):
        x = self._top(n_samples)
        for stack, bias in zip(self._decoder, reversed(list(self._biases))):
            x = ops.add_broadcast_batch(x, bias)
            x, _ = stack(x)
        return self._output(x)


def reproduce(n_epochs=500, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (vd_vae.py:415-494: six stacks
    (3,5)(3,5)(2,4)(2,3)(2,2)(1,1) on 32x32, Adam lr 5e-4) on the MI355X path. Arguments as the
    reference; `debug_loader` replaces both loaders. Returns the Trainer."""
    from pytorch_generative_amd import recipes

    stacks = [StackConfig(3, 5), StackConfig(3, 5), StackConfig(2, 4), StackConfig(2, 3),
              StackConfig(2, 2), StackConfig(1, 1)]
    return recipes.run(
        lambda: VeryDeepVAE(in_channels=1, out_channels=1, input_resolution=32, stack_configs=stacks,
                            latent_channels=16, hidden_channels=64, bottleneck_channels=32),
        loaders=recipes.binarized_mnist_32, loss_fn=recipes.elbo_loss, lr=5e-4,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
