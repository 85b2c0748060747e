"""pytorch_generative_amd — MI355X-native drop-in for pytorch_generative's masked-convolution +
causal-attention training path (nn.* operators, models.* constructors, Trainer)."""

from pytorch_generative_amd import _lib, models, nn, ops  # noqa: F401

__all__ = ["models", "nn", "ops"]
__version__ = "0.1.0"
