"""Model constructors of the hot path (same names / signatures / module layout as
pytorch_generative.models: models.autoregressive.<module>, models.vae.<module>)."""

from pytorch_generative_amd.models import autoregressive, base, vae  # noqa: F401
from pytorch_generative_amd.models.autoregressive.gated_pixel_cnn import GatedPixelCNN
from pytorch_generative_amd.models.autoregressive.image_gpt import ImageGPT
from pytorch_generative_amd.models.autoregressive.pixel_cnn import PixelCNN
from pytorch_generative_amd.models.autoregressive.pixel_cnn_pp import PixelCNNpp
from pytorch_generative_amd.models.autoregressive.pixel_snail import PixelSNAIL
from pytorch_generative_amd.models.vae.beta_vae import BetaVAE
from pytorch_generative_amd.models.vae.vae import VAE
from pytorch_generative_amd.models.vae.vd_vae import VeryDeepVAE
from pytorch_generative_amd.models.vae.vq_vae import VectorQuantizedVAE
from pytorch_generative_amd.models.vae.vq_vae_2 import VectorQuantizedVAE2

__all__ = ["GatedPixelCNN", "ImageGPT", "PixelCNN", "PixelCNNpp", "PixelSNAIL", "VAE", "BetaVAE", "VeryDeepVAE",
           "VectorQuantizedVAE", "VectorQuantizedVAE2"]
