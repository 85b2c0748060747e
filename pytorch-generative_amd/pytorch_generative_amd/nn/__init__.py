"""MI355X-native building blocks with the pytorch_generative.nn operator surface."""

from pytorch_generative_amd.nn.attention import CausalAttention, image_positional_encoding
from pytorch_generative_amd.nn.convolution import (
    CausalConv2d,
    Conv2d,
    ConvTranspose2d,
    GatedActivation,
    GatedConv,
    NCHWLayerNorm,
)
from pytorch_generative_amd.nn.utils import VectorQuantizer

__all__ = [
    "CausalAttention",
    "image_positional_encoding",
    "CausalConv2d",
    "Conv2d",
    "ConvTranspose2d",
    "GatedActivation",
    "GatedConv",
    "NCHWLayerNorm",
    "VectorQuantizer",
]
