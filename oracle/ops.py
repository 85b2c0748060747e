"""Operator-level restatement of pytorch_generative/nn (torch-CPU). Test infrastructure only."""

import functools
import math

import torch
import torch.nn.functional as F


def causal_mask(kh, kw, mask_center):
    """0/1 filter mask. Follows nn/convolution.py:35-39: all rows above the centre row, plus the
    centre row up to (mask_center: excluding / else including) the centre column."""
    m = torch.zeros(kh, kw)
    m[: kh // 2, :] = 1
    m[kh // 2, : kw // 2 + int(not mask_center)] = 1
    return m


def causal_conv2d(x, weight, bias, mask_center, padding):
    """nn/convolution.py:41-43 — `weight.data *= mask` IN PLACE and outside autograd, then an
    ordinary cross-correlation with the (now masked) weight. Consequently d loss / d weight is
    the full, unmasked correlation — exactly the reference's behaviour."""
    kh, kw = weight.shape[2:]
    weight.data *= causal_mask(kh, kw, mask_center).to(weight)
    return F.conv2d(x, weight, bias, padding=padding)


def gated_activation(x, kind="tanh"):
    """nn/convolution.py:62-66 — act(first half) * sigmoid(second half)."""
    c = x.shape[1]
    assert c % 2 == 0
    a, b = x[:, : c // 2], x[:, c // 2 :]
    a = torch.tanh(a) if kind == "tanh" else a
    return a * torch.sigmoid(b)


def nchw_layernorm(x, gamma, beta, eps=1e-5):
    """nn/convolution.py:72-75 — LayerNorm over C via permute / nn.LayerNorm / permute."""
    y = F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), gamma, beta, eps)
    return y.permute(0, 3, 1, 2)


def attention_mask(size, mask_center):
    """nn/attention.py:60-63."""
    return torch.tril(torch.ones(size, size), diagonal=-int(mask_center))


def causal_attention_core(q, k, v, n_heads, mask_center):
    """nn/attention.py:131-160 on already-projected q (N,E,H,W), k (N,E,H,W), v (N,V,H,W)."""
    n, _, h, w = q.shape

    def heads(t):  # (N, C, H, W) -> (N, heads, L, C/heads)   (attention.py:131-135)
        return t.reshape(n, n_heads, t.shape[1] // n_heads, -1).transpose(2, 3)

    q, k, v = heads(q), heads(k), heads(v)
    mask = attention_mask(h * w, mask_center).view(1, 1, h * w, h * w)
    attn = (q @ k.transpose(2, 3)) / math.sqrt(k.shape[-1])
    attn = attn.masked_fill(mask == 0, -float("inf"))
    attn = F.softmax(attn, dim=-1).masked_fill(mask == 0, 0)  # all-masked row: NaN -> 0
    return (attn @ v).transpose(2, 3).contiguous().view(n, -1, h, w)


def causal_attention(x, extra_x, p, prefix, n_heads, embed, mask_center):
    """CausalAttention.forward, nn/attention.py:120-161. `p` maps reference state_dict keys."""
    q = F.conv2d(x, p[prefix + "_q.weight"], p[prefix + "_q.bias"])
    if extra_x is not None:
        x = torch.cat((x, extra_x), dim=1)
    kv = F.conv2d(x, p[prefix + "_kv.weight"], p[prefix + "_kv.bias"])
    k, v = kv[:, :embed], kv[:, embed:]
    out = causal_attention_core(q, k, v, n_heads, mask_center)
    return F.conv2d(out, p[prefix + "_proj.weight"], p[prefix + "_proj.bias"])


@functools.lru_cache(maxsize=32)
def image_positional_encoding(shape):
    """nn/attention.py:37-57."""
    n, _, h, w = shape
    zeros = torch.zeros(n, 1, h, w)
    rows = torch.arange(-0.5, 0.5, 1 / h)[None, None, :, None] + zeros
    cols = torch.arange(-0.5, 0.5, 1 / w)[None, None, None, :] + zeros
    return torch.cat((rows, cols), dim=1)


def bce_sum_mean(logits, x):
    """The loss of every AR reproduce(), e.g. image_gpt.py:158-162."""
    n = x.shape[0]
    loss = F.binary_cross_entropy_with_logits(logits.reshape(n, -1), x.reshape(n, -1), reduction="none")
    return loss.sum(dim=1).mean()


def unit_gaussian_kl_div(mean, log_std):
    """models/vae/vaes.py:17-19."""
    return -0.5 * (1 + 2 * log_std - log_std.exp().pow(2) - mean**2)


def gaussian_kl_div(p_mean, p_log_std, q_mean, q_log_std):
    """models/vae/vaes.py:23-27 — KL(p || q); note q_var = 2 * var(q) folded in."""
    mean_delta, log_std_delta = (p_mean - q_mean) ** 2, q_log_std - p_log_std
    p_var, q_var = p_log_std.exp().pow(2), 2 * q_log_std.exp().pow(2)
    return -0.5 + log_std_delta + (p_var + mean_delta) / q_var


def sample_from_gaussian(mu, log_sig, eps):
    """models/vae/vaes.py:31-33 with the noise made explicit."""
    return mu + log_sig.exp() * eps


def vector_quantize(x, embedding, cluster_size=None, embedding_avg=None, *, use_ema=True,
                    training=True, decay=0.99):
    """VectorQuantizer.forward (reference nn/utils.py:53-96), functional: nothing is updated in place.

    x: (N, D, H, W); embedding: (K, D). Every position's D-vector goes to its nearest codebook row
    (squared Euclidean distance in the expanded form |x|^2 + |e|^2 - 2 x.e of :61-65, argmin keeps the
    FIRST minimum like torch.argmin :68). Returns a dict:
      quantized     x + (q - x).detach()           straight-through estimator (:95)
      loss          mse(x, q.detach())             commitment loss (:79), + mse(q, x.detach()) when
                                                   use_ema is False (:93)
      idxs          (N*H*W,) codebook indices, positions in NHW order (:56)
      embedding / cluster_size / embedding_avg     the buffers AFTER the EMA update of :80-90 when
                                                   use_ema and training, else the inputs unchanged
    """
    n, d, h, w = x.shape
    flat = x.permute(0, 2, 3, 1).contiguous().view(-1, d)
    dist = (flat ** 2).sum(dim=1, keepdim=True) + (embedding ** 2).sum(dim=1) - 2 * flat @ embedding.t()
    idxs = torch.argmin(dist, dim=1)
    q = embedding[idxs].view(n, h, w, d).permute(0, 3, 1, 2).contiguous()
    loss = F.mse_loss(x, q.detach())
    out = {"idxs": idxs, "embedding": embedding, "cluster_size": cluster_size, "embedding_avg": embedding_avg}
    if use_ema and training:
        k = embedding.shape[0]
        one_hot = F.one_hot(idxs, k).to(flat.dtype)
        batch_cluster = one_hot.sum(dim=0)
        batch_avg = (flat.detach().t() @ one_hot).t()
        cs = cluster_size * decay + batch_cluster * (1 - decay)
        ea = embedding_avg * decay + batch_avg * (1 - decay)
        out.update(cluster_size=cs, embedding_avg=ea, embedding=ea / (cs + 1e-5).unsqueeze(1))
    elif not use_ema:
        loss = loss + F.mse_loss(q, x.detach())
    out["quantized"] = x + (q - x).detach()
    out["loss"] = loss
    return out
