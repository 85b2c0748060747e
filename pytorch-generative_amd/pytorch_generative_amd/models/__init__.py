"""Model constructors of the hot path (same names/signatures as pytorch_generative.models)."""

from pytorch_generative_amd.models.gated_pixel_cnn import GatedPixelCNN
from pytorch_generative_amd.models.image_gpt import ImageGPT
from pytorch_generative_amd.models.pixel_cnn import PixelCNN
from pytorch_generative_amd.models.pixel_snail import PixelSNAIL
from pytorch_generative_amd.models.vae import VAE, BetaVAE
from pytorch_generative_amd.models.vd_vae import VeryDeepVAE

__all__ = ["GatedPixelCNN", "ImageGPT", "PixelCNN", "PixelSNAIL", "VAE", "BetaVAE", "VeryDeepVAE"]
