"""Autoregressive image models of the hot path (layout of pytorch_generative.models.autoregressive)."""

from pytorch_generative_amd.models.autoregressive import (  # noqa: F401
    gated_pixel_cnn,
    image_gpt,
    pixel_cnn,
    pixel_cnn_pp,
    pixel_snail,
)
