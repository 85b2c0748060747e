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
