"""SURVEY.md §8(f) rank 4 on the GPU: `nn.VectorQuantizer`, `models.VectorQuantizedVAE` / `VectorQuantizedVAE2`
against the golden vectors of the reference (tests/golden/vq_*.pt, the same files that pin the oracle in
tests/test_oracle_golden.py). First run on MI355X in round 3 (green); part of the default `-m gpu` tier."""

import pytest
import torch

import _util

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need the MI355X"
    from pytorch_generative_amd import _lib

    _lib.load()
    return torch.device("cuda:0")


@pytest.mark.parametrize("case", ["ema_train", "ema_eval"])
def test_vector_quantizer_matches_reference_golden(dev, case):
    import pytorch_generative_amd as pg

    g = _util.load_golden("vq_quantizer")["cases"][case]
    m = pg.nn.VectorQuantizer(n_embeddings=12, embedding_dim=8).to(dev)
    m.load_state_dict(g["before"])
    m.train(g["training"])
    x = g["x"].to(dev).requires_grad_(True)
    q, loss = m(x)
    (q.sum() * 0.5 + loss).backward()
    # the distance arithmetic is not bit-identical to torch's matmul: a near-tie may pick another code;
    # none is expected on this fixture (smallest gap between the two best distances ~1e-2)
    assert torch.equal(q.detach().cpu(), g["quantized"]), "a different codebook row was chosen"
    _util.assert_close(loss, g["loss"], 1e-5, "commitment loss")
    _util.assert_close(x.grad, g["dx"], 1e-5, "dx")
    for key in ("_embedding", "_cluster_size", "_embedding_avg"):
        _util.assert_close(m.state_dict()[key], g["after"][key], 1e-5, f"buffer {key}")


@pytest.mark.parametrize("name,ctor", [("vq_vae_small", "VectorQuantizedVAE"),
                                       ("vq_vae_2_small", "VectorQuantizedVAE2")])
def test_vq_vae_models_match_reference_golden(dev, name, ctor):
    import pytorch_generative_amd as pg
    from pytorch_generative_amd.nn import utils as vq

    g = _util.load_golden(name)
    model = getattr(pg.models, ctor)(**g["kwargs"]).to(dev)
    model.load_state_dict(g["state0"])
    model.train()
    x = g["x"].to(dev)
    recon, vq_loss = model(x)
    loss = vq.mse_loss(recon, x) + vq_loss
    loss.backward()
    _util.assert_close(recon, g["recon"], TOL, "reconstruction")
    _util.assert_close(vq_loss, g["vq_loss"], TOL, "quantization loss")
    _util.assert_close(loss, g["loss"], TOL, "loss")
    for k, p in model.named_parameters():
        want = g["grads"][k]
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
        else:
            _util.assert_close(p.grad, want, 1e-3, f"grad {k}")
    after = g["state_after_forward"]
    for k, v in model.state_dict().items():
        if k.endswith(("_embedding", "_cluster_size", "_embedding_avg")):
            _util.assert_close(v, after[k], 1e-5, f"buffer {k} after forward")
