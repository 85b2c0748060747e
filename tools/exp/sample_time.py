"""Sampling time: row-cached incremental vs the reference's full forward per pixel."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch
import pytorch_generative_amd as pg

dev = torch.device("cuda:0")
CASES = [
    ("PixelCNN", dict(in_channels=1, out_channels=1, n_residual=15, residual_channels=32, head_channels=32), (1, 28, 28)),
    ("GatedPixelCNN", dict(in_channels=3, out_channels=3, n_gated=10, gated_channels=128, head_channels=32), (3, 32, 32)),
    ("PixelSNAIL", dict(in_channels=3, out_channels=3, n_channels=64, n_pixel_snail_blocks=8, n_residual_blocks=2,
                        attention_key_channels=4, attention_value_channels=32), (3, 32, 32)),
]
for n in (16, 256):
    for ctor, kw, chw in CASES:
        torch.manual_seed(0)
        model = getattr(pg.models, ctor)(**kw).to(dev)
        model(torch.rand(2, *chw, device=dev))
        out = {}
        for inc in (True, False):
            if not inc and n > 16 and ctor != "PixelCNN":
                continue  # minutes of full forwards
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.sample(n_samples=n, incremental=inc)
            torch.cuda.synchronize(); out[inc] = time.perf_counter() - t0
        print(f"{ctor:14s} n={n:4d}: row-cached {out[True]:.2f} s" + (f", full forward per pixel {out[False]:.2f} s" if False in out else ""), flush=True)
