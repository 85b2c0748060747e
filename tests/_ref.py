"""Loads the REAL reference (read-only checkout at /root/reference) for oracle pinning and golden
generation. Only usable in the build container; on the GPU box the path does not exist."""

import os
import sys
import types

REF_ROOT = "/root/reference"


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "pytorch_generative"))


def load():
    """Returns the reference's `pytorch_generative` package with its top-level __init__ bypassed
    (it imports torchvision / tensorboard, neither installed here — SURVEY.md §8c)."""
    if "pytorch_generative" in sys.modules and hasattr(sys.modules["pytorch_generative"], "_pg_ref"):
        return sys.modules["pytorch_generative"]
    pkg = types.ModuleType("pytorch_generative")
    pkg.__path__ = [os.path.join(REF_ROOT, "pytorch_generative")]
    pkg._pg_ref = True
    sys.modules["pytorch_generative"] = pkg
    import importlib

    pkg.nn = importlib.import_module("pytorch_generative.nn")
    # models/__init__ imports every model file; only torch/numpy/sklearn are needed
    pkg.models = importlib.import_module("pytorch_generative.models")
    return pkg


def clone_state(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}
