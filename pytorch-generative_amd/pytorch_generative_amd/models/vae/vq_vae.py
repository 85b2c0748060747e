"""VQ-VAE on the MI355X operator path (reference models/vae/vq_vae.py:19-81)."""

from pytorch_generative_amd.models.vae import vaes


class VectorQuantizedVAE(vaes.VariationalAutoEncoder):
    """vq_vae.py:19-81."""

    def __init__(self, in_channels=1, out_channels=1, hidden_channels=128, n_residual_blocks=2,
                 residual_channels=32, n_embeddings=128, embedding_dim=16, sample_fn=None):
        super().__init__(sample_fn)
        self._encoder = vaes.Encoder(in_channels, hidden_channels, hidden_channels, n_residual_blocks,
                                     residual_channels, stride=4)
        self._quantizer = vaes.Quantizer(hidden_channels, n_embeddings, embedding_dim)
        self._decoder = vaes.Decoder(embedding_dim, out_channels, hidden_channels, n_residual_blocks,
                                     residual_channels, stride=4)

    def forward(self, x):
        quantized, quantization_loss = self._quantizer(self._encoder(x))
        return self._decoder(quantized), quantization_loss

    def _sample(self, n_samples):
        raise NotImplementedError("VQ-VAE does not support sampling.")


def reproduce(n_epochs=457, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (vq_vae.py:84-153: CIFAR-10 shaped data, 128 hidden
    channels, 512 codes of width 64, Adam lr 2e-4 with the per-batch 0.999977 decay, loss = MSE +
    quantization loss) on the MI355X path. Arguments as the reference; returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: VectorQuantizedVAE(in_channels=3, out_channels=3, hidden_channels=128,
                                   residual_channels=32, n_residual_blocks=2, n_embeddings=512,
                                   embedding_dim=64),
        loaders=recipes.cifar10, loss_fn=recipes.vq_loss(1.0), lr=2e-4, lr_decay=0.999977,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
