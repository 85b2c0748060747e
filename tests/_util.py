"""Shared helpers for the parity tests."""

import glob
import os

import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_names():
    """Autoregressive-model fixtures (the VAE fixtures, `vae_*.pt`, carry noise and KL terms)."""
    names = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.pt")))
    return [n for n in names if not n.startswith(("vae_", "vq_", "posenc"))]  # vq_*: SURVEY §8(f) rank 4, oracle only so far


def vae_golden_names():
    names = sorted(os.path.basename(p)[:-3] for p in glob.glob(os.path.join(GOLDEN_DIR, "vae_*.pt")))
    return names


def load_golden(name):
    return torch.load(os.path.join(GOLDEN_DIR, name + ".pt"), map_location="cpu", weights_only=False)


ORACLE_FWD = {"ImageGPT": "image_gpt", "PixelCNN": "pixel_cnn", "GatedPixelCNN": "gated_pixel_cnn",
              "PixelSNAIL": "pixel_snail"}


def oracle_kwargs(g):
    return {"n_heads": g["kwargs"]["n_attention_heads"]} if g["ctor"] == "ImageGPT" else {}


def rel_err(got, want):
    """max |got - want| / max |want| (the 'relative' tolerance of BASELINE.json's north_star)."""
    got, want = got.detach().double().cpu(), want.detach().double().cpu()
    return float((got - want).abs().max() / want.abs().max().clamp_min(1e-30))


def assert_close(got, want, tol, what=""):
    e = rel_err(got, want)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"


# Gradient tolerances (round 6: what the kernels deliver, not a bound 500x above it).
#   * per tensor, max-normalised:  max|got - want| <= GRAD_TOL * max|want|            (north_star's 1e-4)
#   * per ELEMENT:                 |got - want| <= GRAD_TOL * |want| + GRAD_FLOOR * max|want|
#     the absolute floor is the summation-order noise of an fp32 reduction over N * H * W pixels (measured ~1e-6 of the
#     tensor's maximum on both sides — torch-CPU's own order differs too); a per-tensor max-norm alone hides errors in
#     small-magnitude entries, this does not.
GRAD_TOL = 1e-4
GRAD_FLOOR = 5e-6


class GradReport:
    """Collects every gradient tensor of a case, then asserts once — so a failing run names EVERY offending tensor with
    its measured figures, and a passing run can print the worst measured error per workload (PG_PARITY_REPORT=<file>
    appends one JSON line per case)."""

    def __init__(self, what, tol=GRAD_TOL, floor=GRAD_FLOOR):
        self.what, self.tol, self.floor, self.rows = what, tol, floor, []

    def add(self, name, got, want):
        got, want = got.detach().double().cpu(), want.detach().double().cpu()
        m = float(want.abs().max())
        d = (got - want).abs()
        max_norm = float(d.max()) / max(m, 1e-30)
        # element-wise: the largest |d| / (tol * |want| + floor * max|want|); <= 1 passes
        elem = float((d / (self.tol * want.abs() + self.floor * m).clamp_min(1e-300)).max())
        self.rows.append((name, max_norm, elem, m))

    def finish(self):
        import json

        assert self.rows, f"{self.what}: no gradient compared"
        worst_mn = max(self.rows, key=lambda r: r[1])
        worst_el = max(self.rows, key=lambda r: r[2])
        rec = {"case": self.what, "tensors": len(self.rows), "worst_max_norm_err": worst_mn[1],
               "worst_max_norm_tensor": worst_mn[0], "worst_elementwise_ratio": worst_el[2],
               "worst_elementwise_tensor": worst_el[0], "tol": self.tol, "floor": self.floor}
        print("[parity] " + json.dumps(rec))
        path = os.environ.get("PG_PARITY_REPORT")
        if path:
            with open(path, "a") as f:
                f.write(json.dumps(rec) + "\n")
        bad = [r for r in self.rows if r[1] > self.tol or r[2] > 1.0]
        assert not bad, f"{self.what}: {len(bad)} of {len(self.rows)} gradient tensors out of tolerance: " + "; ".join(
            f"{n}: max-norm {a:.2e} (tol {self.tol:.0e}), element-wise ratio {b:.2f} (max|want| {m:.2e})" for n, a, b, m in bad[:8])
