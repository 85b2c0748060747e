"""Drop-in aliasing: makes `import pytorch_generative` resolve to this package.

    import pytorch_generative_amd.compat as compat
    compat.install_alias()               # registers pytorch_generative[.models[.autoregressive|.vae]|.nn|...]
    import pytorch_generative as pg      # -> the MI355X path

Code written against the reference's public surface (pytorch_generative/__init__.py:1-3,
nn/__init__.py:3-13, models/__init__.py:3-24, models/<family>/<module>.reproduce, trainer.Trainer)
then runs unmodified on the HIP operator path for the components this package covers; names of
out-of-scope components (NADE, MADE, NICE, KDE, mixtures, `models.flow`) resolve to
placeholders that raise on use.
"""

import sys
import types

import pytorch_generative_amd as _pkg

_OUT_OF_SCOPE_MODULES = {
    "pytorch_generative.models.flow": ("nice",),
    "pytorch_generative.models.autoregressive": ("nade", "made", "fvbn"),
}
_OUT_OF_SCOPE_MODELS = ("NADE", "MADE", "FullyVisibleBeliefNetwork", "NICE", "GaussianKernel", "ParzenWindowKernel",
                        "KernelDensityEstimator", "BernoulliMixtureModel", "GaussianMixtureModel")


class _NotOnThisPath:
    def __init__(self, name):
        self._name = name

    def _fail(self, *a, **k):
        raise NotImplementedError(
            f"{self._name} is outside the masked-convolution / causal-attention path that "
            "pytorch_generative_amd implements (SURVEY.md §8)")

    __call__ = reproduce = _fail

    def __getattr__(self, item):
        if item.startswith("__"):
            raise AttributeError(item)
        return _NotOnThisPath(f"{self._name}.{item}")


def install_alias(name="pytorch_generative"):
    """Registers `name` (and its submodules) in sys.modules as aliases of this package."""
    pairs = {
        name: _pkg,
        f"{name}.nn": _pkg.nn,
        f"{name}.nn.utils": _pkg.nn.utils,
        f"{name}.nn.attention": _pkg.nn.attention,
        f"{name}.nn.convolution": _pkg.nn.convolution,
        f"{name}.models": _pkg.models,
        f"{name}.models.base": _pkg.models.base,
        f"{name}.models.autoregressive": _pkg.models.autoregressive,
        f"{name}.models.vae": _pkg.models.vae,
        f"{name}.trainer": _pkg.trainer,
        f"{name}.datasets": _pkg.datasets,
    }
    for fam in (_pkg.models.autoregressive, _pkg.models.vae):
        for k, v in vars(fam).items():
            if isinstance(v, types.ModuleType) and v.__name__.startswith(fam.__name__ + "."):
                pairs[f"{name}.models.{fam.__name__.rsplit('.', 1)[1]}.{k}"] = v
    for full, children in _OUT_OF_SCOPE_MODULES.items():
        full = full.replace("pytorch_generative", name, 1)
        mod = pairs.get(full)
        if mod is None:
            mod = types.ModuleType(full)
            pairs[full] = mod
            setattr(_pkg.models, full.rsplit(".", 1)[1], mod)
        for child in children:
            if not hasattr(mod, child):
                setattr(mod, child, _NotOnThisPath(f"{full}.{child}"))
    for cls in _OUT_OF_SCOPE_MODELS:
        if not hasattr(_pkg.models, cls):
            setattr(_pkg.models, cls, _NotOnThisPath(f"{name}.models.{cls}"))
    sys.modules.update(pairs)
    return _pkg
