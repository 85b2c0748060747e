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
