"""ops.losses — BCE-with-logits, discretized mixture of logistics, ELBO terms.

Part of the operator layer (pytorch_generative_amd.ops): HIP kernels behind torch.autograd.Function, called through the C-ABI
with tensor.data_ptr() and the current stream. No CPU / ATen fallback: a missing library, a CPU tensor or an unsupported shape raises."""

import torch

from pytorch_generative_amd import _lib
from pytorch_generative_amd.ops._common import _chk, _stream, zeros


# --------------------------------------------------------------------------------------------
# loss
# --------------------------------------------------------------------------------------------
class _BCEWithLogitsSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z, x):
        lib = _lib.load()
        z = _chk(z, "bce.logits")
        x = _chk(x, "bce.targets")
        if z.numel() != x.numel():
            raise ValueError("bce: logits/targets size mismatch")
        n = z.shape[0]
        loss = zeros((1,), z.device)
        _lib.check(lib.pg_bce_logits_fwd(z.data_ptr(), x.data_ptr(), loss.data_ptr(), n,
                                         z.numel() // n, _stream()), "pg_bce_logits_fwd")
        ctx.save_for_backward(z, x)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        z, x = ctx.saved_tensors
        g = _chk(g.reshape(1), "bce.grad")
        n = z.shape[0]
        dz = torch.empty_like(z)
        _lib.check(lib.pg_bce_logits_bwd(z.data_ptr(), x.data_ptr(), g.data_ptr(), dz.data_ptr(), n,
                                         z.numel() // n, _stream()), "pg_bce_logits_bwd")
        return dz, None


class _DmolLossSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l, x, n_mix):
        lib = _lib.load()
        l = _chk(l, "dmol.params")
        x = _chk(x, "dmol.images")
        n, c, h, w = l.shape
        if c != 10 * n_mix or tuple(x.shape) != (n, 3, h, w):
            raise ValueError("dmol: expected (N, 10 * n_mix, H, W) parameters and (N, 3, H, W) images in [-1, 1]")
        loss = zeros((1,), l.device)
        _lib.check(lib.pg_dmol_fwd(l.data_ptr(), x.data_ptr(), loss.data_ptr(), n, n_mix, h * w, _stream()),
                   "pg_dmol_fwd")
        ctx.save_for_backward(l, x)
        ctx.n_mix = n_mix
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        l, x = ctx.saved_tensors
        g = _chk(g.reshape(1), "dmol.grad")
        n, _, h, w = l.shape
        dl = torch.empty_like(l)
        _lib.check(lib.pg_dmol_bwd(l.data_ptr(), x.data_ptr(), g.data_ptr(), dl.data_ptr(), n, ctx.n_mix, h * w,
                                   _stream()), "pg_dmol_bwd")
        return dl, None, None


def dmol_loss_sum_mean(params, images, n_mix=10):
    """Discretized mixture-of-logistics negative log-likelihood (PixelCNN++, Salimans et al. 2017), nats,
    summed over pixels and averaged over the batch. params (N, 10 * n_mix, H, W), images (N, 3, H, W) in
    [-1, 1]. Not in the reference (BASELINE.json configs[2] names it): parity is against oracle/dmol.py."""
    return _DmolLossSumMean.apply(params, images, int(n_mix))


def bce_with_logits_sum_mean(logits, targets):
    """F.binary_cross_entropy_with_logits(reduction='none').sum(pixels).mean(batch)
    (reference image_gpt.py:158-162 and every other AR reproduce())."""
    return _BCEWithLogitsSumMean.apply(logits, targets)


class _ElboMean(torch.autograd.Function):
    """loss = mean_n(recon_n) + mean_n(kl_n) with recon_n the per-sample BCE-with-logits sum."""

    @staticmethod
    def forward(ctx, logits, x, kl):
        lib = _lib.load()
        logits, x, kl = _chk(logits, "elbo.logits"), _chk(x, "elbo.x"), _chk(kl, "elbo.kl")
        n = logits.shape[0]
        recon = zeros((1,), logits.device)
        klm = zeros((1,), logits.device)
        _lib.check(lib.pg_bce_logits_fwd(logits.data_ptr(), x.data_ptr(), recon.data_ptr(), n,
                                         logits.numel() // n, _stream()), "pg_bce_logits_fwd")
        _lib.check(lib.pg_vec_mean_accum(kl.data_ptr(), n, klm.data_ptr(), _stream()),
                   "pg_vec_mean_accum")
        ctx.save_for_backward(logits, x)
        ctx.n = n
        return recon.view(()), klm.view(())

    @staticmethod
    def backward(ctx, g_recon, g_kl):
        lib = _lib.load()
        logits, x = ctx.saved_tensors
        n = ctx.n
        dz = torch.empty_like(logits)
        g_recon = _chk(g_recon.reshape(1), "elbo.g")
        _lib.check(lib.pg_bce_logits_bwd(logits.data_ptr(), x.data_ptr(), g_recon.data_ptr(),
                                         dz.data_ptr(), n, logits.numel() // n, _stream()),
                   "pg_bce_logits_bwd")
        dkl = torch.empty(n, device=logits.device, dtype=torch.float32)
        g_kl = _chk(g_kl.reshape(1), "elbo.gk")
        _lib.check(lib.pg_fill_scaled(g_kl.data_ptr(), 1.0 / n, dkl.data_ptr(), n, _stream()),
                   "pg_fill_scaled")
        return dz, None, dkl


def elbo_terms(logits, x, kl):
    """Returns (recon_loss.mean(), kl_div.mean()) of the reference VAE loss_fn (vae.py:149-159)."""
    return _ElboMean.apply(logits, x, kl)
