"""GPU: the reference's own smoke tier (pytorch_generative/models/tests.py) against this package
aliased as `pytorch_generative`.

* `test_reference_file_unmodified` loads the reference's tests.py itself and runs its cases for
  the seven in-scope models — only where the reference checkout is readable (it is not shipped
  to the GPU boxes and is never copied into this repository).
* The other tests restate exactly those cases (same constructor arguments, same calls, same
  assertions: tests.py:30-77 IntegrationTests via module.reproduce(n_epochs=1, n_gpus=0,
  debug_loader=DummyLoader), :80-161 MultipleChannelsTests, :269-296 MiscTests) so that the tier
  runs on every GPU box.
torch's default device is set to the GPU for the duration (the reference's cases build CPU tensors;
this path has no CPU arithmetic).
"""

import importlib.util
import os
import sys
import unittest

import pytest
import torch

pytestmark = pytest.mark.gpu

REF_TESTS = "/root/reference/pytorch_generative/models/tests.py"
IN_SCOPE = {
    "IntegrationTests": ["test_PixelCNN", "test_GatedPixelCNN", "test_PixelSnail", "test_ImageGPT",
                         "test_VAE", "test_BetaVAE", "test_VeryDeepVAE", "test_VectorQuantizedVAE", "test_VectorQuantizedVAE2"],
    "MultipleChannelsTests": ["test_PixelCNN", "test_GatedPixelCNN", "test_PixelSNAIL", "test_ImageGPT",
                              "test_VAE", "test_VeryDeepVAE"],
    "MiscTests": ["test_sampling_after_load"],
}


@pytest.fixture()
def pg_alias():
    import pytorch_generative_amd.compat as compat

    saved = {k: v for k, v in sys.modules.items() if k == "pytorch_generative" or k.startswith("pytorch_generative.")}
    pkg = compat.install_alias()
    torch.set_default_device("cuda")
    try:
        yield pkg
    finally:
        torch.set_default_device("cpu")
        for k in [k for k in sys.modules if k == "pytorch_generative" or k.startswith("pytorch_generative.")]:
            del sys.modules[k]
        sys.modules.update(saved)


@pytest.mark.skipif(not os.path.exists(REF_TESTS), reason="reference checkout not present on this box")
def test_reference_file_unmodified(pg_alias):
    spec = importlib.util.spec_from_file_location("_ref_models_tests", REF_TESTS)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    suite = unittest.TestSuite()
    for cls, names in IN_SCOPE.items():
        for n in names:
            suite.addTest(getattr(mod, cls)(n))
    result = unittest.TextTestRunner(verbosity=0).run(suite)
    assert result.wasSuccessful(), (result.failures, result.errors)
    assert result.testsRun == sum(len(v) for v in IN_SCOPE.values())


class _DummyLoader:
    """One (x, y) batch per epoch: x ~ N(0, 1) of shape (1, C, S, S) (tests.py:12-27)."""

    def __init__(self, channels, size):
        self._batch = (torch.randn((1, channels, size, size)), torch.tensor([0]))

    def __iter__(self):
        return iter([self._batch])


@pytest.mark.parametrize("family,module,size,channels", [
    ("autoregressive", "pixel_cnn", 28, 1), ("autoregressive", "gated_pixel_cnn", 28, 1),
    ("autoregressive", "pixel_snail", 28, 1), ("autoregressive", "image_gpt", 28, 1),
    ("vae", "vae", 32, 1), ("vae", "beta_vae", 32, 1), ("vae", "vd_vae", 32, 1),
    ("vae", "vq_vae", 28, 3), ("vae", "vq_vae_2", 28, 3),  # tests.py:70-74
])
def test_integration_reproduce(pg_alias, tmp_path, family, module, size, channels):
    import pytorch_generative as pg  # the alias

    mod = getattr(getattr(pg.models, family), module)
    with pytest.warns(UserWarning, match="n_gpus=0"):
        t = mod.reproduce(n_epochs=1, log_dir=str(tmp_path), n_gpus=0, debug_loader=_DummyLoader(channels, size))
    assert t._epoch == 1 and t._step == 1
    assert os.path.exists(os.path.join(tmp_path, "trainer_state_1.ckpt"))
    assert all(torch.isfinite(p).all() for p in t.model.parameters())


def _multiple_channels(model, conditional_sample):
    batch = torch.rand(2, 3, 8, 8)
    model(batch)
    assert model.sample(n_samples=2).dim() == 4
    if conditional_sample:
        batch[:, :, 1:, :] = -1
        sample = model.sample(conditioned_on=batch)
        assert (sample[:, :, 0, :] == batch[:, :, 0, :]).all()


def test_multiple_channels(pg_alias):
    from pytorch_generative import models
    from pytorch_generative.models.vae.vd_vae import StackConfig

    _multiple_channels(models.PixelCNN(in_channels=3, out_channels=3, n_residual=1, residual_channels=1,
                                       head_channels=1), True)
    _multiple_channels(models.GatedPixelCNN(in_channels=3, out_channels=3, n_gated=1, gated_channels=1,
                                            head_channels=1), True)
    _multiple_channels(models.PixelSNAIL(in_channels=3, out_channels=3, n_channels=2, n_pixel_snail_blocks=1,
                                         n_residual_blocks=1, attention_key_channels=1,
                                         attention_value_channels=1), True)
    _multiple_channels(models.ImageGPT(in_channels=3, out_channels=3, in_size=8, n_transformer_blocks=1,
                                       n_attention_heads=2, n_embedding_channels=4), True)
    _multiple_channels(models.VAE(in_channels=3, out_channels=3, latent_channels=1, strides=[2, 2],
                                  hidden_channels=1, residual_channels=1), False)
    _multiple_channels(models.VeryDeepVAE(in_channels=3, out_channels=3, input_resolution=8,
                                          stack_configs=[StackConfig(1, 1), StackConfig(1, 1)],
                                          latent_channels=1, bottleneck_channels=1), False)


def test_sampling_after_load(pg_alias):
    from pytorch_generative import models

    kw = dict(in_channels=3, out_channels=3, n_residual=1, residual_channels=1, head_channels=1)
    model = models.PixelCNN(**kw)
    model(torch.rand(2, 3, 8, 8))
    model.sample(2)
    fresh = models.PixelCNN(**kw)
    fresh.load_state_dict(model.state_dict())
    assert fresh.sample(2).shape == (2, 3, 8, 8)


def test_out_of_scope_names_fail_loudly(pg_alias):
    from pytorch_generative import models

    with pytest.raises(NotImplementedError, match="outside the masked-convolution"):
        models.NADE(input_dim=4, hidden_dim=2)
    with pytest.raises(NotImplementedError):
        models.flow.nice.reproduce(n_epochs=1)
