"""VAE family of the hot path (layout of pytorch_generative.models.vae)."""

from pytorch_generative_amd.models.vae import beta_vae, vae, vaes, vd_vae, vq_vae, vq_vae_2  # noqa: F401
