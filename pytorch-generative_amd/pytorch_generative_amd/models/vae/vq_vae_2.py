"""VQ-VAE-2 on the MI355X operator path (reference models/vae/vq_vae_2.py:21-110)."""

import torch

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd.models.vae import vaes
from pytorch_generative_amd.nn import utils as nn_utils


class VectorQuantizedVAE2(vaes.VariationalAutoEncoder):
    """vq_vae_2.py:21-110."""

    def __init__(self, in_channels=1, out_channels=1, hidden_channels=128, n_residual_blocks=2,
                 residual_channels=32, n_embeddings=128, embedding_dim=16, sample_fn=None):
        super().__init__(sample_fn)
        enc = dict(hidden_channels=hidden_channels, n_residual_blocks=n_residual_blocks,
                   residual_channels=residual_channels, stride=2)
        self._encoder_b = vaes.Encoder(in_channels=in_channels, out_channels=hidden_channels, **enc)
        self._encoder_t = vaes.Encoder(in_channels=hidden_channels, out_channels=hidden_channels, **enc)
        self._quantizer_t = vaes.Quantizer(hidden_channels, n_embeddings, embedding_dim)
        self._quantizer_b = vaes.Quantizer(hidden_channels, n_embeddings, embedding_dim)
        self._decoder_t = vaes.Decoder(in_channels=embedding_dim, out_channels=hidden_channels, **enc)
        self._conv = pg_nn.Conv2d(in_channels=hidden_channels, out_channels=embedding_dim, kernel_size=1)
        self._decoder_b = vaes.Decoder(in_channels=2 * embedding_dim, out_channels=out_channels, **enc)

    def forward(self, x):
        encoded_b = self._encoder_b(x)
        encoded_t = self._encoder_t(encoded_b)
        quantized_t, vq_loss_t = self._quantizer_t(encoded_t)
        quantized_b, vq_loss_b = self._quantizer_b(encoded_b)
        decoded_t = self._decoder_t(quantized_t)
        xhat = self._decoder_b(torch.cat((self._conv(decoded_t), quantized_b), dim=1))
        # 0.5 * (vq_b + vq_t) + mse(decoded_t, encoded_b), vq_vae_2.py:110 (gradients to both arguments)
        return xhat, (vq_loss_b + vq_loss_t) * 0.5 + nn_utils.mse_loss(decoded_t, encoded_b)

    def _sample(self, n_samples):
        raise NotImplementedError("VQ-VAE-2 does not support sampling.")


def reproduce(n_epochs=457, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (vq_vae_2.py:117-186: as VQ-VAE with 64 residual
    channels and loss = MSE + 0.25 * quantization loss) on the MI355X path. Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: VectorQuantizedVAE2(in_channels=3, out_channels=3, hidden_channels=128,
                                    n_residual_blocks=2, residual_channels=64, n_embeddings=512,
                                    embedding_dim=64),
        loaders=recipes.cifar10, loss_fn=recipes.vq_loss(0.25), lr=2e-4, lr_decay=0.999977,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
