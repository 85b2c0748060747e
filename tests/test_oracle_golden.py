"""CPU: the oracle reproduces the reference's golden vectors (tests/golden/*.pt, generated from
the real reference by tests/golden/make_golden.py). Tolerances: fp32, 1e-5 relative — the oracle
calls the same torch CPU primitives, so any larger gap is an algorithmic difference."""

import pytest
import torch

import _util
from oracle import models as omodels
from oracle import train as otrain


@pytest.mark.parametrize("name", _util.golden_names())
def test_oracle_matches_golden_step(name):
    g = _util.load_golden(name)
    state = {k: v.clone() for k, v in g["state0"].items()}
    fwd = omodels.FORWARDS[_util.ORACLE_FWD[g["ctor"]]]
    logits, loss, grads = otrain.loss_and_grads(fwd, state, g["x"], **_util.oracle_kwargs(g))
    _util.assert_close(logits, g["logits"], 1e-5, "logits")
    _util.assert_close(loss, g["loss"], 1e-5, "loss")
    for k, want in g["grads"].items():
        if want is None:
            assert grads[k] is None, k
        else:
            _util.assert_close(grads[k], want, 2e-4, f"grad {k}")
    norm = otrain.adam_step_(state, grads, otrain.new_opt_state(), lr=g["lr"])
    assert abs(norm - float(g["grad_norm"])) <= 1e-5 * float(g["grad_norm"])
    for k, want in g["state1"].items():
        if otrain.is_param(k):
            _util.assert_close(state[k], want, 1e-5, f"param {k} after Adam")


def test_mask_patterns_known_answers():
    """3x3 A/B and 7x7 A tap counts and exact patterns."""
    from oracle import ops as oops

    assert oops.causal_mask(3, 3, True).tolist() == [[1, 1, 1], [1, 0, 0], [0, 0, 0]]
    assert oops.causal_mask(3, 3, False).tolist() == [[1, 1, 1], [1, 1, 0], [0, 0, 0]]
    assert int(oops.causal_mask(7, 7, True).sum()) == 24
    m = oops.attention_mask(5, True)
    assert m[0].sum() == 0 and torch.equal(m, torch.tril(torch.ones(5, 5), -1))


def test_strict_attention_first_pixel_is_bias():
    """A strictly-causal attention row with no allowed key yields zeros, so the first output
    pixel equals the projection bias."""
    from oracle import ops as oops

    torch.manual_seed(0)
    p = {"_q.weight": torch.randn(4, 6, 1, 1), "_q.bias": torch.randn(4),
         "_kv.weight": torch.randn(12, 6, 1, 1), "_kv.bias": torch.randn(12),
         "_proj.weight": torch.randn(8, 8, 1, 1), "_proj.bias": torch.randn(8)}
    out = oops.causal_attention(torch.randn(2, 6, 4, 4), None, p, "", 1, 4, True)
    assert torch.allclose(out[:, :, 0, 0], p["_proj.bias"].expand(2, 8), atol=1e-6)
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("name", _util.vae_golden_names())
def test_oracle_matches_vae_golden(name):
    g = _util.load_golden(name)
    leaves = {k: v.clone().requires_grad_(True) for k, v in g["state0"].items() if otrain.is_param(k)}
    if g["ctor"] == "VeryDeepVAE":
        logits, kl = omodels.vd_vae(leaves, g["x"], g["eps"])
    else:
        logits, kl = omodels.vae(leaves, g["x"], g["eps"][0], beta=g["kwargs"].get("beta", 1.0))
    recon, klm, elbo = omodels.elbo_terms(logits, g["x"], kl)
    _util.assert_close(logits, g["logits"], 1e-5, "logits")
    _util.assert_close(kl, g["kl"], 1e-5, "kl")
    _util.assert_close(elbo, g["loss"], 1e-5, "elbo")
    grads = torch.autograd.grad(elbo, list(leaves.values()), allow_unused=True)
    for k, go in zip(leaves, grads):
        want = g["grads"][k]
        if want is None:
            assert go is None or float(go.abs().max()) == 0.0, k
        else:
            _util.assert_close(go, want, 2e-4, f"grad {k}")
