"""Shared VAE pieces on the MI355X operator path (reference models/vae/vaes.py).

Noise: the reference draws eps with torch.randn_like inside sample_from_gaussian
(vaes.py:31-33). Here every Gaussian head asks `draw_noise(shape, device)`; parity tests replace
it (set_noise_fn) with a function that replays pre-drawn eps so both sides see identical noise.
"""

import torch

from pytorch_generative_amd.models import base


def _default_noise(shape, device):
    return torch.randn(shape, device=device, dtype=torch.float32)


_noise_fn = _default_noise


def set_noise_fn(fn=None):
    """Installs `fn(shape, device) -> eps` as the noise source (None restores torch.randn)."""
    global _noise_fn
    _noise_fn = fn or _default_noise


def draw_noise(shape, device):
    return _noise_fn(tuple(shape), device)


class VariationalAutoEncoder(base.GenerativeModel):
    """Base class of the VAE family (reference models/base.py:123-134)."""

    def __init__(self, sample_fn=None):
        super().__init__()
        self._sample_fn = sample_fn or base._bernoulli_from_logits

    def _sample(self, n_samples):
        raise NotImplementedError

    @torch.no_grad()
    def sample(self, n_samples):
        return self._sample_fn(self._sample(n_samples))


# ---- encoder / decoder stacks (reference models/vae/vaes.py:69-241) ----------------------------
from torch import nn  # noqa: E402

from pytorch_generative_amd import nn as pg_nn  # noqa: E402
from pytorch_generative_amd import ops  # noqa: E402
from pytorch_generative_amd.nn import utils as nn_utils  # noqa: E402


class ResidualBlock(nn.Module):
    """x + conv1x1(relu(conv3x3(relu(x)))); ReLUs fused into the conv input loads."""

    def __init__(self, n_channels, hidden_channels):
        super().__init__()
        self._net = nn.Sequential(
            nn.ReLU(),
            pg_nn.Conv2d(n_channels, hidden_channels, kernel_size=3, padding=1),
            nn.ReLU(),
            pg_nn.Conv2d(hidden_channels, n_channels, kernel_size=1),
        )

    def forward(self, x):
        # x has two readers (the 3x3 convolution and the residual add): the add reads a pass-through alias whose
        # gradient the 3x3 convolution's data-gradient kernel adds in its epilogue (ops.conv2d_taps, n_skip)
        h, res = self._net[1](x, in_act="relu", n_skip=1)
        return self._net[3](h, in_act="relu", res=res)


class ResidualStack(nn.Module):
    def __init__(self, n_channels, hidden_channels, n_residual_blocks=1):
        super().__init__()
        self._net = nn.Sequential(
            *[ResidualBlock(n_channels, hidden_channels) for _ in range(n_residual_blocks)]
            + [nn.ReLU()]
        )

    def forward(self, x, *, fuse_final_relu=False):
        """With fuse_final_relu the trailing ReLU is left to the consumer's `in_act`."""
        for block in list(self._net)[:-1]:
            x = block(x)
        return x if fuse_final_relu else ops.relu(x)


class Encoder(nn.Module):
    """Down-samples by `stride` with stride//2 4x4/s2 convolutions (+ReLU), a residual stack and a
    3x3 output convolution."""

    def __init__(self, in_channels, out_channels, hidden_channels, n_residual_blocks,
                 residual_channels, stride):
        super().__init__()
        assert stride % 2 == 0, '"stride" must be even.'
        net = []
        for i in range(stride // 2):
            first, last = 0, stride // 2 - 1
            in_c = in_channels if i == first else hidden_channels // 2
            out_c = hidden_channels // 2 if i < last else hidden_channels
            net.append(pg_nn.Conv2d(in_c, out_c, kernel_size=4, stride=2, padding=1))
            net.append(nn.ReLU())
        net.append(ResidualStack(hidden_channels, residual_channels, n_residual_blocks))
        net.append(pg_nn.Conv2d(hidden_channels, out_channels, kernel_size=3, padding=1))
        self._net = nn.Sequential(*net)
        self._n_down = stride // 2

    def forward(self, x):
        mods = list(self._net)
        pending_relu = False
        for i in range(self._n_down):
            x = mods[2 * i](x, in_act="relu" if pending_relu else None)
            pending_relu = True  # the ReLU after each down conv is fused into the next consumer
        # the residual stack adds x itself, so the ReLU before it must be materialised
        x = ops.relu(x)
        x = mods[2 * self._n_down](x, fuse_final_relu=True)
        return mods[2 * self._n_down + 1](x, in_act="relu")


class Decoder(nn.Module):
    """3x3 conv, residual stack, then stride//2 4x4/s2 transposed convolutions (ReLU between)."""

    def __init__(self, in_channels, out_channels, hidden_channels, n_residual_blocks,
                 residual_channels, stride):
        super().__init__()
        assert stride % 2 == 0, '"stride" must be even.'
        net = [
            pg_nn.Conv2d(in_channels, hidden_channels, kernel_size=3, padding=1),
            ResidualStack(hidden_channels, residual_channels, n_residual_blocks),
        ]
        for i in range(stride // 2):
            first, last = 0, stride // 2 - 1
            in_c = hidden_channels if i == first else hidden_channels // 2
            out_c = hidden_channels // 2 if i < last else out_channels
            net.append(pg_nn.ConvTranspose2d(in_c, out_c, kernel_size=4, stride=2, padding=1))
            if i < last:
                net.append(nn.ReLU())
        self._net = nn.Sequential(*net)

    def forward(self, x):
        mods = list(self._net)
        x = mods[1](mods[0](x), fuse_final_relu=True)
        pending_relu = True
        for m in mods[2:]:
            if isinstance(m, pg_nn.ConvTranspose2d):
                x = m(x, in_act="relu" if pending_relu else None)
                pending_relu = False
            else:  # nn.ReLU marker
                pending_relu = True
        return x


class Quantizer(nn.Module):
    """vaes.py:244-264: 1x1 convolution to the embedding width, then the VectorQuantizer."""

    def __init__(self, in_channels, n_embeddings, embedding_dim):
        super().__init__()
        self._net = nn.Sequential(
            pg_nn.Conv2d(in_channels=in_channels, out_channels=embedding_dim, kernel_size=1),
            nn_utils.VectorQuantizer(n_embeddings, embedding_dim),
        )

    def forward(self, x):
        return self._net[1](self._net[0](x))
