"""pytorch_generative_amd — MI355X-native drop-in for pytorch_generative's masked-convolution +
causal-attention training path: the nn.* operators, the models.* constructors (with each module's
reproduce() recipe) and the Trainer, on hand-written gfx950 kernels behind a C-ABI."""

from pytorch_generative_amd import _lib, datasets, models, nn, ops, trainer  # noqa: F401

__all__ = ["datasets", "models", "nn", "ops", "trainer"]
__version__ = "0.2.0"
