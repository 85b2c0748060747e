"""VAE on the MI355X operator path (reference models/vae/vae.py:15-101): same constructor and
state_dict; the [mean | log_std] split, the
unit-Gaussian KL and the reparameterised sample are one fused kernel."""

import torch
from torch import nn

from pytorch_generative_amd import ops
from pytorch_generative_amd.models.vae import vaes


class VAE(vaes.VariationalAutoEncoder):
    def __init__(self, in_channels=1, out_channels=1, latent_channels=16, strides=[4],
                 hidden_channels=64, residual_channels=32, sample_fn=None):
        super().__init__(sample_fn)
        self._latent_channels = latent_channels
        self._total_stride = sum(strides)
        encoder = []
        for i, stride in enumerate(strides):
            in_c = in_channels if i == 0 else hidden_channels
            out_c = hidden_channels if i < len(strides) - 1 else 2 * latent_channels
            encoder.append(vaes.Encoder(in_c, out_c, hidden_channels, n_residual_blocks=2,
                                        residual_channels=residual_channels, stride=stride))
        self._encoder = nn.Sequential(*encoder)
        decoder = []
        for i, stride in enumerate(reversed(strides)):
            in_c = latent_channels if i == 0 else hidden_channels
            out_c = hidden_channels if i < len(strides) - 1 else out_channels
            decoder.append(vaes.Decoder(in_c, out_c, hidden_channels, n_residual_blocks=2,
                                        residual_channels=residual_channels, stride=stride))
        self._decoder = nn.Sequential(*decoder)

    def forward(self, x):
        """Returns (logits, kl_div per sample — summed over the latent dims, not normalised)."""
        h = self._encoder(x)  # [mean | log_std]
        n, _, lh, lw = h.shape
        eps = vaes.draw_noise((n, self._latent_channels, lh, lw), h.device)
        latents, kl_div = ops.gaussian_head_unit(h, eps, self._latent_channels)
        return self._decoder(latents), kl_div

    def _sample(self, n_samples):
        latent_size = int(self._h) // 2 ** (self._total_stride // 2)
        shape = (n_samples, self._latent_channels, latent_size, latent_size)
        return self._decoder(torch.randn(shape, device=self.device))


def reproduce(n_epochs=457, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (vae.py:104-167: same model
    hyper-parameters, Adam lr 5e-4) on the MI355X path. Arguments as the reference;
    `debug_loader` replaces both loaders (any iterable of (x, y) batches). Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: VAE(in_channels=1, out_channels=1, latent_channels=16, strides=[2, 2, 2, 2],
                    hidden_channels=64, residual_channels=32),
        loaders=recipes.binarized_mnist_32, loss_fn=recipes.elbo_loss, lr=5e-4, lr_decay=1.0,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
