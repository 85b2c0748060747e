"""Pins the CPU oracle against the REAL reference (runs only where /root/reference exists).

The reference's own tests hold no numeric vectors for this path, so the oracle is
checked op by op, model by model and for a full train step against the live reference modules."""

import copy

import pytest
import torch

import _ref
from oracle import models as omodels
from oracle import ops as oops
from oracle import train as otrain

pytestmark = pytest.mark.skipif(not _ref.available(), reason="/root/reference not present")


@pytest.fixture(scope="module")
def ref():
    return _ref.load()


@pytest.mark.parametrize("k,mc", [(3, True), (3, False), (7, True), (5, False)])
def test_causal_mask_bit_exact(ref, k, mc):
    conv = ref.nn.CausalConv2d(mc, 2, 3, kernel_size=k, padding=k // 2)
    assert torch.equal(conv.mask[0, 0], oops.causal_mask(k, k, mc))
    assert torch.equal(conv.mask, oops.causal_mask(k, k, mc).expand_as(conv.mask))


@pytest.mark.parametrize("size,mc", [(16, True), (16, False), (784, False), (1024, True)])
def test_attention_mask_bit_exact(ref, size, mc):
    from pytorch_generative.nn import attention

    assert torch.equal(attention._get_causal_mask(size, mc), oops.attention_mask(size, mc))


@pytest.mark.parametrize("shape", [(2, 3, 32, 32), (1, 1, 28, 28), (2, 3, 8, 8), (1, 3, 64, 64)])
def test_posenc_bit_exact(ref, shape):
    assert torch.equal(ref.nn.image_positional_encoding(shape), oops.image_positional_encoding(shape))


def test_causal_conv_and_ln_and_gate(ref):
    torch.manual_seed(0)
    x = torch.randn(2, 4, 9, 9)
    conv = ref.nn.CausalConv2d(True, 4, 6, kernel_size=3, padding=1)
    w0 = conv.weight.detach().clone()
    y = conv(x)
    assert torch.allclose(y, oops.causal_conv2d(x, w0, conv.bias.detach(), True, 1), atol=1e-6)
    ln = ref.nn.NCHWLayerNorm(4)
    with torch.no_grad():
        ln.weight.uniform_(0.5, 1.5)
        ln.bias.uniform_(-1, 1)
    assert torch.allclose(ln(x), oops.nchw_layernorm(x, ln.weight, ln.bias), atol=1e-6)
    assert torch.allclose(ref.nn.GatedActivation()(x), oops.gated_activation(x, "tanh"), atol=1e-7)
    ga = ref.nn.GatedActivation(activation_fn=torch.nn.Identity())
    assert torch.allclose(ga(x), oops.gated_activation(x, "identity"), atol=1e-7)


@pytest.mark.parametrize("mc,heads,extra", [(False, 2, 0), (True, 1, 3)])
def test_attention_module(ref, mc, heads, extra):
    torch.manual_seed(1)
    attn = ref.nn.CausalAttention(6, n_heads=heads, embed_channels=4, out_channels=8,
                                  mask_center=mc, extra_input_channels=extra)
    x = torch.randn(2, 6, 5, 5)
    ex = torch.randn(2, extra, 5, 5) if extra else None
    p = {k: v.detach() for k, v in attn.state_dict().items()}
    got = oops.causal_attention(x, ex, p, "", heads, 4, mc)
    assert torch.allclose(attn(x, ex), got, atol=1e-6)


def _build(ref, name):
    torch.manual_seed(0)
    m = ref.models
    if name == "image_gpt":
        return m.ImageGPT(1, 1, in_size=8, n_transformer_blocks=2, n_attention_heads=2,
                          n_embedding_channels=8), (2, 1, 8, 8), {"n_heads": 2}
    if name == "pixel_cnn":
        return m.PixelCNN(1, 1, n_residual=2, residual_channels=4, head_channels=4), (2, 1, 9, 9), {}
    if name == "gated_pixel_cnn":
        return m.GatedPixelCNN(3, 3, n_gated=2, gated_channels=4, head_channels=4), (2, 3, 9, 9), {}
    return m.PixelSNAIL(3, 3, n_channels=8, n_pixel_snail_blocks=2, n_residual_blocks=2,
                        attention_key_channels=2, attention_value_channels=4), (2, 3, 8, 8), {}


@pytest.mark.parametrize("name", ["image_gpt", "pixel_cnn", "gated_pixel_cnn", "pixel_snail"])
def test_model_forward_loss_grads_and_adam_step(ref, name):
    model, shape, kw = _build(ref, name)
    if name == "image_gpt":  # _pos is zero-initialised; make it matter
        with torch.no_grad():
            model._pos.normal_(0, 0.1)
    g = torch.Generator().manual_seed(7)
    x = torch.rand(shape, generator=g)
    state = _ref.clone_state(model)
    # --- reference step (trainer.py:173-193 semantics)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    opt.zero_grad()
    logits = model(x)
    loss = oops.bce_sum_mean(logits, x)  # same formula as every reproduce().loss_fn
    loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e50)
    ref_grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in model.named_parameters()}
    opt.step()
    # --- oracle step on the same initial state
    o_logits, o_loss, o_grads = otrain.loss_and_grads(omodels.FORWARDS[name], state, x, **kw)
    assert torch.allclose(o_logits, logits.detach(), rtol=1e-5, atol=1e-5)
    assert abs(float(o_loss) - float(loss.detach())) <= 1e-5 * abs(float(loss.detach()))
    for k, g_ref in ref_grads.items():
        if g_ref is None:
            assert o_grads[k] is None, k
        else:
            assert torch.allclose(o_grads[k], g_ref, rtol=1e-4, atol=1e-5), k
    o_norm = otrain.adam_step_(state, o_grads, otrain.new_opt_state(), lr=1e-3)
    assert abs(o_norm - float(norm)) <= 1e-5 * float(norm)
    for k, p in model.named_parameters():
        assert torch.allclose(state[k], p.detach(), rtol=1e-5, atol=1e-6), k


def test_vd_vae_forward_elbo_and_grads(ref):
    """VD-VAE (vd_vae.py) with the reference's noise replayed: vaes.sample_from_gaussian is
    resolved by attribute lookup at call time, so it is patched to consume pre-drawn eps."""
    from pytorch_generative.models.vae import vaes as rvaes
    from pytorch_generative.models.vae import vd_vae as rvd

    torch.manual_seed(0)
    cfg = [rvd.StackConfig(2, 2), rvd.StackConfig(1, 2), rvd.StackConfig(1, 1)]
    model = ref.models.VeryDeepVAE(3, 3, input_resolution=8, stack_configs=cfg, latent_channels=4,
                                   hidden_channels=8, bottleneck_channels=4)
    with torch.no_grad():
        for b in model._biases:
            b.normal_(0, 0.1)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 8, 8, generator=g)
    state = _ref.clone_state(model)
    shapes = omodels.vd_vae_noise_shapes(state, 2, 8)
    eg = torch.Generator().manual_seed(4321)
    eps = [torch.randn(s, generator=eg) for s in shapes]
    it = iter(eps)
    orig = rvaes.sample_from_gaussian
    rvaes.sample_from_gaussian = lambda mu, log_sig: mu + log_sig.exp() * next(it)
    try:
        logits, kl = model(x)
    finally:
        rvaes.sample_from_gaussian = orig
    recon, klm, elbo = omodels.elbo_terms(logits, x, kl)
    elbo.backward()
    # oracle
    leaves = {k: v.clone().requires_grad_(True) for k, v in state.items() if otrain.is_param(k)}
    o_logits, o_kl = omodels.vd_vae(leaves, x, eps)
    _, _, o_elbo = omodels.elbo_terms(o_logits, x, o_kl)
    grads = torch.autograd.grad(o_elbo, list(leaves.values()), allow_unused=True)
    assert torch.allclose(o_logits, logits.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(o_kl, kl.detach(), rtol=1e-5, atol=1e-4)
    for (k, p), go in zip(model.named_parameters(), grads):
        if p.grad is None:
            assert go is None or float(go.abs().max()) == 0.0, k
        else:
            assert torch.allclose(go, p.grad, rtol=1e-4, atol=1e-5), k


def test_beta_vae_forward_elbo_and_grads(ref):
    from pytorch_generative.models.vae import vaes as rvaes

    torch.manual_seed(0)
    model = ref.models.BetaVAE(3, 3, beta=4.0, latent_channels=4, strides=[2, 4], hidden_channels=8,
                               residual_channels=4)
    g = torch.Generator().manual_seed(3)
    x = torch.rand(2, 3, 16, 16, generator=g)
    state = _ref.clone_state(model)
    eps = torch.randn(2, 4, 2, 2, generator=torch.Generator().manual_seed(4321))
    orig = rvaes.sample_from_gaussian
    rvaes.sample_from_gaussian = lambda mu, log_sig: mu + log_sig.exp() * eps
    try:
        logits, kl = model(x)
    finally:
        rvaes.sample_from_gaussian = orig
    _, _, elbo = omodels.elbo_terms(logits, x, kl)
    elbo.backward()
    leaves = {k: v.clone().requires_grad_(True) for k, v in state.items() if otrain.is_param(k)}
    o_logits, o_kl = omodels.vae(leaves, x, eps, beta=4.0)
    _, _, o_elbo = omodels.elbo_terms(o_logits, x, o_kl)
    grads = torch.autograd.grad(o_elbo, list(leaves.values()), allow_unused=True)
    assert torch.allclose(o_logits, logits.detach(), rtol=1e-5, atol=1e-5)
    assert torch.allclose(o_kl, kl.detach(), rtol=1e-5, atol=1e-4)
    for (k, p), go in zip(model.named_parameters(), grads):
        assert torch.allclose(go, p.grad, rtol=1e-4, atol=1e-5), k


@pytest.mark.parametrize("use_ema,training", [(True, True), (True, False), (False, True)])
def test_vector_quantizer_live(ref, use_ema, training):
    """oracle.ops.vector_quantize against the live VectorQuantizer (nn/utils.py:53-96), fresh seeds."""
    torch.manual_seed(11)
    vq = ref.nn.VectorQuantizer(n_embeddings=32, embedding_dim=6, use_ema=use_ema)
    vq.train(training)
    before = _ref.clone_state(vq)
    x = torch.randn(4, 6, 7, 5)
    q, loss = vq(x)
    out = oops.vector_quantize(x, before["_embedding"], before.get("_cluster_size"),
                               before.get("_embedding_avg"), use_ema=use_ema, training=training)
    assert torch.equal(out["quantized"], q.detach())
    assert torch.allclose(out["loss"], loss.detach(), rtol=1e-6, atol=0)
    after = _ref.clone_state(vq)
    for key in ("_embedding", "_cluster_size", "_embedding_avg"):
        if key in after:
            assert torch.allclose(out[key[1:]], after[key], rtol=1e-6, atol=1e-7), key


@pytest.mark.parametrize("heads,embed,out", [(1, 4, 32), (1, 4, 16), (2, 8, 64), (1, 16, 16), (1, 32, 32), (2, 64, 32),
                                              (1, 64, 64), (2, 16, 40), (1, 8, 20)],
                         ids=lambda v: str(v))
@pytest.mark.parametrize("mc", [False, True], ids=["incl_centre", "strict"])
def test_attention_module_head_dims_forward_and_gradients(ref, heads, embed, out, mc):
    """The head dimensions the GPU tier exercises (d_k = embed / heads in {4, 8, 16, 32, 64}, d_v = out / heads in
    {16, 20, 32, 64}; tests/test_gpu_ops.py ATTN_CASES compare the HIP kernels with `oracle.ops.causal_attention_core`):
    the oracle against the LIVE reference module — output, input gradient and every parameter gradient."""
    torch.manual_seed(3)
    cin = 6
    attn = ref.nn.CausalAttention(cin, n_heads=heads, embed_channels=embed, out_channels=out, mask_center=mc)
    x = torch.randn(2, cin, 4, 5)
    d_o = torch.randn(2, out, 4, 5)
    xr = x.clone().requires_grad_(True)
    attn(xr).backward(d_o)
    p = {k: v.detach().clone().requires_grad_(True) for k, v in attn.state_dict().items()}
    xo = x.clone().requires_grad_(True)
    got = oops.causal_attention(xo, None, p, "", heads, embed, mc)
    got.backward(d_o)
    assert torch.allclose(attn(x), got.detach(), atol=2e-6, rtol=1e-5)
    assert torch.allclose(xr.grad, xo.grad, atol=2e-6, rtol=1e-5)
    for k, v in attn.named_parameters():
        assert torch.allclose(v.grad, p[k].grad, atol=5e-6, rtol=1e-5), k
