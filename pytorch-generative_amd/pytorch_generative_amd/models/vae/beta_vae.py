"""Beta-VAE on the MI355X operator path (reference models/vae/beta_vae.py:16-60)."""

from pytorch_generative_amd.models.vae import vae


class BetaVAE(vae.VAE):
    """VAE whose KL term is scaled by `beta` (beta=1 is the plain VAE)."""

    def __init__(self, in_channels=1, out_channels=1, beta=4.0, latent_channels=16, strides=[4],
                 hidden_channels=64, residual_channels=32, sample_fn=None):
        super().__init__(in_channels, out_channels, latent_channels, strides, hidden_channels,
                         residual_channels, sample_fn)
        self._beta = beta

    def forward(self, x):
        out, kl_div = super().forward(x)
        return out, self._beta * kl_div


def reproduce(n_epochs=500, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (beta_vae.py:63-127: same model
    hyper-parameters, Adam lr 1e-3) on the MI355X path. Arguments as the reference;
    `debug_loader` replaces both loaders (any iterable of (x, y) batches). Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: BetaVAE(in_channels=1, out_channels=1, beta=4.0, latent_channels=16,
                        strides=[2, 2, 2, 2], hidden_channels=64, residual_channels=32),
        loaders=recipes.binarized_mnist_32, loss_fn=recipes.elbo_loss, lr=1e-3, lr_decay=1.0,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
