"""Model-level restatement (torch-CPU) as pure functions of a reference-keyed state_dict.
Test infrastructure only. Each function follows the cited reference forward()."""

import torch
import torch.nn.functional as F

from oracle import ops


def _conv(x, p, name, padding=0):
    return F.conv2d(x, p[name + ".weight"], p.get(name + ".bias"), padding=padding)


def _cconv(x, p, name, mask_center, padding):
    return ops.causal_conv2d(x, p[name + ".weight"], p.get(name + ".bias"), mask_center, padding)


def apply_masks_(p):
    """The side effect of CausalConv2d.forward (nn/convolution.py:42): every weight that has a
    sibling `.mask` buffer is multiplied by it IN PLACE (outside autograd)."""
    with torch.no_grad():
        for k in list(p.keys()):
            if k.endswith(".mask"):
                p[k[: -len("mask")] + "weight"].mul_(p[k])
    return p


# ------------------------------------------------------------------------------- ImageGPT
def image_gpt(p, x, n_heads):
    """models/autoregressive/image_gpt.py:105-109 (+ TransformerBlock.forward :50-52)."""
    n_blocks = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("_transformer."))
    c = p["_input.weight"].shape[0]
    h = _cconv(x + p["_pos"], p, "_input", True, 1)
    for i in range(n_blocks):
        pre = f"_transformer.{i}."
        b = h
        a = ops.nchw_layernorm(b, p[pre + "_ln1.weight"], p[pre + "_ln1.bias"])
        b = b + ops.causal_attention(a, None, p, pre + "_attn.", n_heads, c, False)
        m = ops.nchw_layernorm(b, p[pre + "_ln2.weight"], p[pre + "_ln2.bias"])
        m = _conv(F.gelu(_conv(m, p, pre + "_out.0")), p, pre + "_out.2")
        h = h + (b + m)  # the model loop adds x again (image_gpt.py:108)
    return _conv(ops.nchw_layernorm(h, p["_ln.weight"], p["_ln.bias"]), p, "_out")


# ------------------------------------------------------------------------------- PixelCNN
def pixel_cnn(p, x):
    """models/autoregressive/pixel_cnn.py:106-110 (+ CausalResidualBlock :52-53)."""
    n_res = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("_causal_layers."))
    h = _cconv(x, p, "_input", True, 3)
    for i in range(n_res):
        pre = f"_causal_layers.{i}._net."
        t = _conv(F.relu(h), p, pre + "1")
        t = _cconv(F.relu(t), p, pre + "3", False, 1)
        t = _conv(F.relu(t), p, pre + "5")
        h = h + (h + t)  # doubled residual (pixel_cnn.py:109 with :53)
    t = _conv(F.relu(h), p, "_head.1")
    return _conv(F.relu(t), p, "_head.3")


# -------------------------------------------------------------------------- GatedPixelCNN
def _gated_layer(p, pre, v_in, h_in, k, mask_center):
    """GatedPixelCNNLayer.forward, gated_pixel_cnn.py:99-130."""
    _, _, h, w = v_in.shape
    pad = (k - 1) // 2
    v = F.conv2d(v_in, p[pre + "_vstack_1xN.weight"], p[pre + "_vstack_1xN.bias"], padding=(0, pad))
    v = F.conv2d(v, p[pre + "_vstack_Nx1.weight"], p[pre + "_vstack_Nx1.bias"], padding=(pad + 1, 0))
    v = v[:, :, :h, :]
    link = _conv(v, p, pre + "_link")
    v = ops.gated_activation(v + _conv(v_in, p, pre + "_vstack_1x1"), "tanh")
    hs = F.conv2d(h_in, p[pre + "_hstack_1xN.weight"], p[pre + "_hstack_1xN.bias"],
                  padding=(0, pad + int(mask_center)))[:, :, :, :w]
    hs = ops.gated_activation(link + hs, "tanh")
    skip = _conv(hs, p, pre + "_hstack_skip")
    hs = _conv(hs, p, pre + "_hstack_residual")
    if not mask_center:
        hs = hs + h_in
    return v, hs, skip


def gated_pixel_cnn(p, x):
    """models/autoregressive/gated_pixel_cnn.py:185-190."""
    n_gated = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("_gated_layers."))
    v, hs, skips = _gated_layer(p, "_input.", x, x, 7, True)
    for i in range(n_gated):
        v, hs, skip = _gated_layer(p, f"_gated_layers.{i}.", v, hs, 3, False)
        skips = skips + skip
    t = _conv(F.relu(skips), p, "_head.1")
    return _conv(F.relu(t), p, "_head.3")


# ----------------------------------------------------------------------------- PixelSNAIL
def _snail_residual(p, pre, x):
    """ResidualBlock.forward, pixel_snail.py:52-56."""
    _, c, h, w = x.shape
    out = F.elu(F.conv2d(F.elu(x), p[pre + "_input_conv.weight"], p[pre + "_input_conv.bias"], padding=1))
    out = out[:, :, :h, :w]
    out = F.conv2d(out, p[pre + "_output_conv.weight"], p[pre + "_output_conv.bias"], padding=1)
    return x + ops.gated_activation(out[:, :, :h, :w], "identity")


def pixel_snail(p, x):
    """models/autoregressive/pixel_snail.py:182-187 (+ PixelSNAILBlock.forward :103-119)."""
    n_blocks = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("_pixel_snail_blocks."))
    img = x
    h = _cconv(x, p, "_input", True, 1)

    def ece(name, t):  # _elu_conv_elu, pixel_snail.py:27-28
        return F.elu(_conv(F.elu(t), p, name))

    for i in range(n_blocks):
        pre = f"_pixel_snail_blocks.{i}."
        n_res = 1 + max(int(k[len(pre + "_residual."):].split(".")[0]) for k in p
                        if k.startswith(pre + "_residual."))
        res = h
        for j in range(n_res):
            res = _snail_residual(p, f"{pre}_residual.{j}.", res)
        pos = ops.image_positional_encoding(tuple(img.shape))
        embed = p[pre + "_attention._q.weight"].shape[0]
        attn = ops.causal_attention(torch.cat((pos, res), dim=1), img, p, pre + "_attention.", 1,
                                    embed, True)
        blk = ece(pre + "_out", ece(pre + "_residual_out", res) + ece(pre + "_attention_out", attn))
        h = h + blk
    return _conv(_conv(h, p, "_output.0"), p, "_output.1")


# -------------------------------------------------------------------------------- VD-VAE
def _bottleneck(p, pre, x, residual):
    """BottleneckBlock.forward, vd_vae.py:102-104 (kernel size / padding read off the weights)."""
    h = x
    for idx in ("1", "3", "5", "7"):
        w = p[f"{pre}_net.{idx}.weight"]
        h = F.conv2d(F.gelu(h), w, p[f"{pre}_net.{idx}.bias"], padding=w.shape[-1] // 2)
    return x + h if residual else h


def _count(p, prefix):
    return 1 + max(int(k[len(prefix):].split(".")[0]) for k in p if k.startswith(prefix))


def vd_vae(p, x, eps_list):
    """VeryDeepVAE.forward, vd_vae.py:375-405 (+ TopDownBlock.forward :147-189, EncoderStack
    :226-229, DecoderStack :270-284). `eps_list` supplies the noise of every TopDownBlock in
    call order (top-down). Returns (logits, kl_div per sample)."""
    n = x.shape[0]
    n_stacks = _count(p, "_encoder.")
    latent = p["_decoder.0._topdowns.0._latents.weight"].shape[1]
    h = F.conv2d(x, p["_input.weight"], p["_input.bias"], padding=1)
    mixins = []
    for i in range(n_stacks):
        for j in range(_count(p, f"_encoder.{i}._residuals.")):
            h = _bottleneck(p, f"_encoder.{i}._residuals.{j}.", h, True)
        mixins.append(h)
        if i < n_stacks - 1:
            h = F.avg_pool2d(h, kernel_size=2, stride=2)
    h = torch.zeros_like(p[f"_biases.{n_stacks - 1}"]).repeat(n, 1, 1, 1)
    kl = torch.zeros(n)
    eps_iter = iter(eps_list)
    for i in range(n_stacks):
        mixin, bias = mixins[n_stacks - 1 - i], p[f"_biases.{n_stacks - 1 - i}"]
        h = h + bias.repeat(n, 1, 1, 1)
        if i > 0:
            h = F.interpolate(h, scale_factor=2, mode="nearest")
        for j in range(_count(p, f"_decoder.{i}._topdowns.")):
            pre = f"_decoder.{i}._topdowns.{j}."
            prior = _bottleneck(p, pre + "_prior.", h, False)
            p_mean, p_log_std, p_h = prior[:, :latent], prior[:, latent:2 * latent], prior[:, 2 * latent:]
            post = _bottleneck(p, pre + "_posterior.", torch.cat((h, mixin), dim=1), False)
            q_mean, q_log_std = post[:, :latent], post[:, latent:]
            z = ops.sample_from_gaussian(q_mean, q_log_std, next(eps_iter))
            kl = kl + ops.gaussian_kl_div(q_mean, q_log_std, p_mean, p_log_std).sum(dim=(1, 2, 3))
            lat = F.conv2d(z, p[pre + "_latents.weight"], p[pre + "_latents.bias"])
            h = _bottleneck(p, pre + "_out.", h + p_h + lat, True)
    return F.conv2d(h, p["_output.weight"], p["_output.bias"]), kl


def vd_vae_noise_shapes(p, n, input_resolution):
    """Shapes of the eps tensors vd_vae() consumes, in call order."""
    n_stacks = _count(p, "_encoder.")
    latent = p["_decoder.0._topdowns.0._latents.weight"].shape[1]
    shapes = []
    for i in range(n_stacks):
        res = input_resolution // 2 ** (n_stacks - 1 - i)
        shapes += [(n, latent, res, res)] * _count(p, f"_decoder.{i}._topdowns.")
    return shapes


# ------------------------------------------------------------------------- VAE / Beta-VAE
def _residual_stack(p, pre, x):
    """ResidualStack / ResidualBlock forward, vaes.py:92-95,118-119."""
    for b in range(_count(p, pre + "_net.") if any(k.startswith(pre + "_net.") for k in p) else 0):
        if f"{pre}_net.{b}._net.1.weight" not in p:
            continue
        h = F.conv2d(F.relu(x), p[f"{pre}_net.{b}._net.1.weight"], p[f"{pre}_net.{b}._net.1.bias"], padding=1)
        x = x + F.conv2d(F.relu(h), p[f"{pre}_net.{b}._net.3.weight"], p[f"{pre}_net.{b}._net.3.bias"])
    return F.relu(x)


def _seq_indices(p, pre):
    return sorted({int(k[len(pre):].split(".")[0]) for k in p if k.startswith(pre)})


def _vae_encoder(p, pre, x):
    """Encoder.forward, vaes.py:140-181: (4x4 s2 conv, ReLU)*, ResidualStack, 3x3 conv."""
    for idx in _seq_indices(p, pre + "_net."):
        key = f"{pre}_net.{idx}."
        if key + "weight" in p:
            w = p[key + "weight"]
            if w.shape[-1] == 4:
                x = F.relu(F.conv2d(x, w, p[key + "bias"], stride=2, padding=1))
            else:
                x = F.conv2d(x, w, p[key + "bias"], padding=1)
        else:
            x = _residual_stack(p, key, x)
    return x


def _vae_decoder(p, pre, x):
    """Decoder.forward, vaes.py:208-241: 3x3 conv, ResidualStack, (4x4 s2 ConvTranspose, ReLU)* —
    no ReLU after the last transposed convolution."""
    idxs = _seq_indices(p, pre + "_net.")
    tconvs = [i for i in idxs if f"{pre}_net.{i}.weight" in p and p[f"{pre}_net.{i}.weight"].shape[-1] == 4]
    for idx in idxs:
        key = f"{pre}_net.{idx}."
        if key + "weight" in p:
            w = p[key + "weight"]
            if w.shape[-1] == 4:
                x = F.conv_transpose2d(x, w, p[key + "bias"], stride=2, padding=1)
                if idx != tconvs[-1]:
                    x = F.relu(x)
            else:
                x = F.conv2d(x, w, p[key + "bias"], padding=1)
        else:
            x = _residual_stack(p, key, x)
    return x


def vae(p, x, eps, beta=1.0):
    """VAE.forward (vae.py:79-94) / BetaVAE.forward (beta_vae.py:58-60): returns (logits, beta*kl)."""
    h = x
    for i in range(_count(p, "_encoder.")):
        h = _vae_encoder(p, f"_encoder.{i}.", h)
    c = h.shape[1] // 2
    mean, log_std = h[:, :c], h[:, c:]
    kl = ops.unit_gaussian_kl_div(mean, log_std).sum(dim=(1, 2, 3))
    z = ops.sample_from_gaussian(mean, log_std, eps)
    for i in range(_count(p, "_decoder.")):
        z = _vae_decoder(p, f"_decoder.{i}.", z)
    return z, beta * kl


def vq_vae(p, x, training=True, decay=0.99):
    """VectorQuantizedVAE.forward (reference models/vae/vq_vae.py:69-81): Encoder (stride 4) ->
    Quantizer = 1x1 convolution + VectorQuantizer (vaes.py:244-264) -> Decoder (stride 4).
    Returns (reconstruction, quantization loss, vq) with vq = ops.vector_quantize's dict: the EMA
    buffers after the step are vq["embedding"], vq["cluster_size"], vq["embedding_avg"]."""
    h = _vae_encoder(p, "_encoder.", x)
    h = F.conv2d(h, p["_quantizer._net.0.weight"], p["_quantizer._net.0.bias"])
    pre = "_quantizer._net.1."
    use_ema = pre + "_cluster_size" in p
    vq = ops.vector_quantize(h, p[pre + "_embedding"], p.get(pre + "_cluster_size"),
                             p.get(pre + "_embedding_avg"), use_ema=use_ema, training=training,
                             decay=decay)
    return _vae_decoder(p, "_decoder.", vq["quantized"]), vq["loss"], vq


def _quantizer(p, pre, h, training, decay):
    """vaes.Quantizer (vaes.py:244-264): 1x1 convolution to the embedding width, then VectorQuantizer."""
    h = F.conv2d(h, p[pre + "_net.0.weight"], p[pre + "_net.0.bias"])
    q = pre + "_net.1."
    return ops.vector_quantize(h, p[q + "_embedding"], p.get(q + "_cluster_size"), p.get(q + "_embedding_avg"),
                               use_ema=q + "_cluster_size" in p, training=training, decay=decay)


def vq_vae_2(p, x, training=True, decay=0.99):
    """VectorQuantizedVAE2.forward (reference models/vae/vq_vae_2.py:96-110): bottom and top encoders
    (stride 2 each), both levels quantized, the top level decoded back to the bottom resolution, then
    the bottom decoder over cat(conv1x1(decoded_t), quantized_b). Returns (xhat, quantization loss,
    (vq_t, vq_b)); loss = 0.5 (vq_b + vq_t) + mse(decoded_t, encoded_b) (:110, no detach)."""
    enc_b = _vae_encoder(p, "_encoder_b.", x)
    enc_t = _vae_encoder(p, "_encoder_t.", enc_b)
    vq_t = _quantizer(p, "_quantizer_t.", enc_t, training, decay)
    vq_b = _quantizer(p, "_quantizer_b.", enc_b, training, decay)
    dec_t = _vae_decoder(p, "_decoder_t.", vq_t["quantized"])
    xhat = _vae_decoder(p, "_decoder_b.", torch.cat(
        (F.conv2d(dec_t, p["_conv.weight"], p["_conv.bias"]), vq_b["quantized"]), dim=1))
    return xhat, 0.5 * (vq_b["loss"] + vq_t["loss"]) + F.mse_loss(dec_t, enc_b), (vq_t, vq_b)


def vq_vae_loss(recon, x, vq_loss):
    """loss_fn of vq_vae.reproduce (vq_vae.py:127-136): mse(preds, x) + vq_loss."""
    return F.mse_loss(recon, x) + vq_loss


def elbo_terms(logits, x, kl):
    """loss_fn of the VAE reproduce()s, vae.py:149-159: (recon.mean(), kl.mean(), elbo.mean())."""
    recon = F.binary_cross_entropy_with_logits(logits, x, reduction="none").sum(dim=(1, 2, 3))
    return recon.mean(), kl.mean(), (recon + kl).mean()


FORWARDS = {
    "image_gpt": image_gpt,
    "pixel_cnn": pixel_cnn,
    "gated_pixel_cnn": gated_pixel_cnn,
    "pixel_snail": pixel_snail,
}
