"""The branch-free GELU arithmetic of csrc/common.h (pg_gelu_parts), restated in numpy fp32: its distance from the
float64 erf GELU is the size of one fp32 rounding, like ATen's own fp32 GELU (the oracle's)."""
import numpy as np
import torch
from scipy.special import erf

F = np.float32


def gelu_parts(x):
    z = (np.abs(x) * F(0.70710678118654752440)).astype(F)
    t = (F(1) / (F(0.3275911) * z + F(1))).astype(F)
    p = (t * F(1.061405429) + F(-1.453152027)).astype(F)
    p = (p * t + F(1.421413741)).astype(F)
    p = (p * t + F(-0.284496736)).astype(F)
    p = (p * t + F(0.254829592)).astype(F)
    p = (p * t).astype(F)
    e = np.exp((-z * z).astype(F)).astype(F)
    q = (F(0.5) * p * e).astype(F)
    return np.where(x >= 0, F(1) - q, q).astype(F), e


def test_gelu_and_its_derivative_against_float64():
    x = np.linspace(-8, 8, 400001).astype(F)
    xd = x.astype(np.float64)
    cdf, e = gelu_parts(x)
    ref = 0.5 * xd * (1 + erf(xd / np.sqrt(2)))
    ref_g = 0.5 * (1 + erf(xd / np.sqrt(2))) + xd * np.exp(-0.5 * xd * xd) / np.sqrt(2 * np.pi)
    ours = (x * cdf).astype(F)
    ours_g = (cdf + x * F(0.39894228040143267794) * e).astype(F)
    aten = torch.nn.functional.gelu(torch.from_numpy(x)).numpy()
    assert np.abs(ours - ref).max() <= 1e-6
    assert np.abs(ours - ref).max() <= 2 * max(np.abs(aten - ref).max(), 5e-7)
    assert np.abs(ours_g - ref_g).max() <= 1e-6
    # deep negative tail (values below 1e-3 in magnitude): relative error bounded like ATen's, whose 1 + erf cancels
    tail = (xd < -3) & (xd > -5)
    rel = (np.abs(ours - ref)[tail] / np.abs(ref[tail])).max()
    rel_aten = (np.abs(aten - ref)[tail] / np.abs(ref[tail])).max()
    assert rel <= max(rel_aten, 0.05)
