"""Discretized mixture-of-logistics likelihood (PixelCNN++) — CPU restatement. Test infrastructure only.

The reference repository has NO implementation of this loss (BASELINE.json configs[2] names it; SURVEY.md
§8(f) rank 4), so there is no reference file:line to follow: "parity unpinned" against the reference.
What is restated here is the PUBLISHED algorithm:

  Salimans, Karpathy, Chen, Kingma: "PixelCNN++: Improving the PixelCNN with Discretized Logistic Mixture
  Likelihood and Other Modifications", ICLR 2017 — eq. (2) (the discretized logistic mixture with the edge
  bins 0 and 255 absorbing the tails) and eq. (3) (sub-pixel conditioning: the mean of green depends
  linearly on red, the mean of blue on red and green), with the numerical safeguards of the authors'
  public implementation (log-scales clamped at -7; log of the bin mass replaced by the log-density at
  the bin centre minus log 127.5 when the mass is below 1e-5; masses floored at 1e-12).

It is pinned by analytic known answers instead (tests/test_dmol_cpu.py): for every parameter setting the
masses of the 256 values of one sub-pixel sum to 1; a single component with a huge scale gives the uniform
1/256 in the interior; the likelihood is invariant to a permutation of the mixture components; gradients
agree with finite differences.

Channel layout of the network output `l` (N, 10 K, H, W) for K components, images x (N, 3, H, W) in [-1, 1]
(x = v / 127.5 - 1 for an 8-bit value v):
  [0, K)                     mixture logits
  K + c * 3K + [0, K)        mean of sub-pixel c (c = 0, 1, 2 = R, G, B)
  K + c * 3K + [K, 2K)       log-scale of sub-pixel c
  K + c * 3K + [2K, 3K)      raw coefficient c (tanh applied): c = 0 couples G to R, 1 couples B to R,
                             2 couples B to G
"""

import math

import torch
import torch.nn.functional as F

LOG_SCALE_MIN = -7.0
BIN = 1.0 / 255.0      # half a bin in [-1, 1] units
MASS_SWITCH = 1e-5     # below this the bin mass is replaced by the density at the centre
MASS_FLOOR = 1e-12


def split_params(l, n_mix):
    """-> logits (N,K,H,W), means / log_scales (N,3,K,H,W), coeffs (N,3,K,H,W) (tanh applied)."""
    n, c, h, w = l.shape
    assert c == 10 * n_mix, "expected 10 * n_mix channels"
    logits = l[:, :n_mix]
    rest = l[:, n_mix:].reshape(n, 3, 3 * n_mix, h, w)
    means = rest[:, :, :n_mix]
    log_scales = rest[:, :, n_mix:2 * n_mix].clamp(min=LOG_SCALE_MIN)
    coeffs = torch.tanh(rest[:, :, 2 * n_mix:])
    return logits, means, log_scales, coeffs


def component_log_probs(x, means, log_scales, coeffs):
    """log P(x_c | component k) per sub-pixel: (N, 3, K, H, W). Eq. (2) + (3)."""
    xr, xg = x[:, 0:1], x[:, 1:2]  # (N,1,H,W) broadcast over K
    m0 = means[:, 0]
    m1 = means[:, 1] + coeffs[:, 0] * xr
    m2 = means[:, 2] + coeffs[:, 1] * xr + coeffs[:, 2] * xg
    m = torch.stack((m0, m1, m2), dim=1)
    xx = x.unsqueeze(2)  # (N,3,1,H,W)
    centered = xx - m
    inv_s = torch.exp(-log_scales)
    plus_in = inv_s * (centered + BIN)
    min_in = inv_s * (centered - BIN)
    cdf_plus, cdf_min = torch.sigmoid(plus_in), torch.sigmoid(min_in)
    log_cdf_plus = plus_in - F.softplus(plus_in)          # log sigmoid: the value 0 takes everything below
    log_one_minus_cdf_min = -F.softplus(min_in)           # the value 255 takes everything above
    cdf_delta = cdf_plus - cdf_min
    mid_in = inv_s * centered
    log_pdf_mid = mid_in - log_scales - 2.0 * F.softplus(mid_in)
    inner = torch.where(cdf_delta > MASS_SWITCH, torch.log(cdf_delta.clamp(min=MASS_FLOOR)),
                        log_pdf_mid - math.log(127.5))
    return torch.where(xx < -0.999, log_cdf_plus, torch.where(xx > 0.999, log_one_minus_cdf_min, inner))


def dmol_log_likelihood(l, x, n_mix):
    """log p(x) per pixel, (N, H, W): logsumexp_k [log_softmax(logits)_k + sum_c log P(x_c | k)]."""
    logits, means, log_scales, coeffs = split_params(l, n_mix)
    lp = component_log_probs(x, means, log_scales, coeffs).sum(dim=1) + F.log_softmax(logits, dim=1)
    return torch.logsumexp(lp, dim=1)


def dmol_loss_sum_mean(l, x, n_mix):
    """Negative log-likelihood in nats: summed over pixels, mean over the batch (the reduction every
    reproduce() of the reference uses for its losses, e.g. image_gpt.py:158-162)."""
    return -dmol_log_likelihood(l, x, n_mix).sum(dim=(1, 2)).mean()
