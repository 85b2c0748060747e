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


# ---- VectorQuantizer / VQ-VAE — the oracle restatement against outputs of the
# ---- reference (tests/golden/make_vq_golden.py). The HIP path for this row is not built yet.
@pytest.mark.parametrize("case", ["ema_train", "ema_eval", "sgd_train"])
def test_vector_quantizer_oracle_matches_reference_golden(case):
    from oracle import ops as oops

    g = _util.load_golden("vq_quantizer")["cases"][case]
    b = g["before"]
    x = g["x"].clone().requires_grad_(True)
    emb = b["_embedding"].clone().requires_grad_(not g["use_ema"])
    out = oops.vector_quantize(x, emb, b.get("_cluster_size"), b.get("_embedding_avg"),
                               use_ema=g["use_ema"], training=g["training"])
    assert torch.equal(out["quantized"].detach(), g["quantized"])  # same codebook rows, bit for bit
    _util.assert_close(out["loss"], g["loss"], 1e-6, "vq loss")
    (out["quantized"].sum() * 0.5 + out["loss"]).backward()
    _util.assert_close(x.grad, g["dx"], 1e-6, "dx (straight-through + commitment)")
    if not g["use_ema"]:
        _util.assert_close(emb.grad, g["d_embedding"], 1e-6, "d embedding")
    for key in ("_embedding", "_cluster_size", "_embedding_avg"):
        if key in g["after"]:
            _util.assert_close(out[key[1:]], g["after"][key], 1e-6, f"buffer {key} after forward")
    if g["use_ema"] and not g["training"]:
        assert all(torch.equal(g["after"][k], b[k]) for k in b)  # eval leaves the buffers alone


def test_vq_vae_oracle_matches_reference_golden_step():
    g = _util.load_golden("vq_vae_small")
    state = {k: v.clone() for k, v in g["state0"].items()}
    leaves = {k: v.requires_grad_(True) for k, v in state.items() if otrain.is_param(k)}
    p = dict(state)
    p.update(leaves)
    recon, vq_loss, vq = omodels.vq_vae(p, g["x"], training=True)
    loss = omodels.vq_vae_loss(recon, g["x"], vq_loss)
    _util.assert_close(recon, g["recon"], 1e-5, "reconstruction")
    _util.assert_close(vq_loss, g["vq_loss"], 1e-5, "vq loss")
    _util.assert_close(loss, g["loss"], 1e-5, "loss")
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)))
    for k, want in g["grads"].items():
        if want is None:
            assert grads[k] is None, k
        else:
            _util.assert_close(grads[k], want, 2e-4, f"grad {k}")
    pre = "_quantizer._net.1."
    for key in ("_embedding", "_cluster_size", "_embedding_avg"):  # EMA buffers move in forward
        _util.assert_close(vq[key[1:]], g["state_after_forward"][pre + key], 1e-6, key)


def test_vq_vae_2_oracle_matches_reference_golden():
    g = _util.load_golden("vq_vae_2_small")
    state = {k: v.clone() for k, v in g["state0"].items()}
    leaves = {k: v.requires_grad_(True) for k, v in state.items() if otrain.is_param(k)}
    p = dict(state)
    p.update(leaves)
    xhat, vq_loss, (vq_t, vq_b) = omodels.vq_vae_2(p, g["x"], training=True)
    loss = omodels.vq_vae_loss(xhat, g["x"], vq_loss)
    _util.assert_close(xhat, g["recon"], 1e-5, "reconstruction")
    _util.assert_close(vq_loss, g["vq_loss"], 1e-5, "quantization loss")
    _util.assert_close(loss, g["loss"], 1e-5, "loss")
    grads = dict(zip(leaves, torch.autograd.grad(loss, list(leaves.values()), allow_unused=True)))
    for k, want in g["grads"].items():
        if want is None:
            assert grads[k] is None, k
        else:
            _util.assert_close(grads[k], want, 2e-4, f"grad {k}")
    for pre, vq in (("_quantizer_t._net.1.", vq_t), ("_quantizer_b._net.1.", vq_b)):
        for key in ("_embedding", "_cluster_size", "_embedding_avg"):
            _util.assert_close(vq[key[1:]], g["state_after_forward"][pre + key], 1e-6, pre + key)
