"""Generates tests/golden/*.pt from the REAL reference (run in the build container only):

    python tests/golden/make_golden.py

Each fixture = one reference training step (trainer.py:173-193 semantics) of a small model:
initial state_dict, input, logits, loss, every parameter gradient, the global grad norm and the
state_dict after one torch.optim.Adam step. /root/reference is not available on the GPU box, so
these files are what pins the HIP path (and the oracle) to the reference there.
"""

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref  # noqa: E402

CONFIGS = {
    # name: (model ctor name, kwargs, input shape, lr, input kind)
    "image_gpt_baseline": ("ImageGPT", dict(in_channels=1, out_channels=1, in_size=28,
                                            n_transformer_blocks=8, n_attention_heads=4,
                                            n_embedding_channels=16), (2, 1, 28, 28), 5e-3, "mnist"),
    "image_gpt_small": ("ImageGPT", dict(in_channels=3, out_channels=3, in_size=7,
                                         n_transformer_blocks=2, n_attention_heads=2,
                                         n_embedding_channels=8), (3, 3, 7, 7), 5e-3, "cifar"),
    "pixel_cnn_small": ("PixelCNN", dict(in_channels=1, out_channels=1, n_residual=3,
                                         residual_channels=8, head_channels=8), (2, 1, 28, 28), 1e-3, "mnist"),
    "gated_pixel_cnn_small": ("GatedPixelCNN", dict(in_channels=3, out_channels=3, n_gated=2,
                                                    gated_channels=8, head_channels=8), (2, 3, 12, 12), 1e-3, "cifar"),
    "pixel_snail_small": ("PixelSNAIL", dict(in_channels=3, out_channels=3, n_channels=16,
                                             n_pixel_snail_blocks=2, n_residual_blocks=2,
                                             attention_key_channels=4, attention_value_channels=8),
                          (2, 3, 16, 16), 1e-3, "cifar"),
}


def make_input(shape, kind, seed=1234):
    g = torch.Generator().manual_seed(seed)
    if kind == "mnist":  # dynamically binarised MNIST-shaped (datasets.py:16-17)
        return torch.bernoulli(torch.full(shape, 0.1307), generator=g)
    return torch.randint(0, 256, shape, generator=g).float() / 255  # ToTensor()-style CIFAR


def main():
    ref = _ref.load()
    import torch.nn.functional as F

    for name, (ctor, kwargs, shape, lr, kind) in CONFIGS.items():
        torch.manual_seed(0)
        model = getattr(ref.models, ctor)(**kwargs)
        if hasattr(model, "_pos"):
            with torch.no_grad():
                model._pos.normal_(0, 0.1)
        x = make_input(shape, kind)
        state0 = _ref.clone_state(model)
        opt = torch.optim.Adam(model.parameters(), lr=lr)
        opt.zero_grad()
        logits = model(x)
        n = x.shape[0]
        loss = F.binary_cross_entropy_with_logits(
            logits.reshape(n, -1), x.reshape(n, -1), reduction="none").sum(dim=1).mean()
        loss.backward()
        norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e50)
        grads = {k: (p.grad.detach().clone() if p.grad is not None else None)
                 for k, p in model.named_parameters()}
        opt.step()
        state1 = _ref.clone_state(model)
        out = {
            "ctor": ctor, "kwargs": kwargs, "lr": lr, "x": x, "state0": state0,
            "logits": logits.detach().clone(), "loss": loss.detach().clone(), "grads": grads,
            "grad_norm": norm.detach().clone(), "state1": state1,
            "torch_version": torch.__version__,
        }
        path = os.path.join(HERE, name + ".pt")
        torch.save(out, path)
        print(f"{name}: loss={float(loss.detach()):.6f} norm={float(norm):.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_vd_vae():
    """VD-VAE fixture: the reference's noise (torch.randn_like inside vaes.sample_from_gaussian)
    is replaced by pre-drawn eps (saved in the fixture) so the step is reproducible anywhere."""
    ref = _ref.load()
    from pytorch_generative.models.vae import vaes as rvaes
    from pytorch_generative.models.vae import vd_vae as rvd

    from oracle import models as omodels

    torch.manual_seed(0)
    cfg = [(2, 3), (2, 2), (1, 2), (1, 1)]
    kwargs = dict(in_channels=3, out_channels=3, input_resolution=16, stack_configs=cfg,
                  latent_channels=4, hidden_channels=16, bottleneck_channels=8)
    rk = dict(kwargs, stack_configs=[rvd.StackConfig(*c) for c in cfg])
    model = ref.models.VeryDeepVAE(**rk)
    with torch.no_grad():
        for b in model._biases:
            b.normal_(0, 0.1)
    x = make_input((2, 3, 16, 16), "cifar")
    state0 = _ref.clone_state(model)
    eg = torch.Generator().manual_seed(4321)
    eps = [torch.randn(s, generator=eg) for s in omodels.vd_vae_noise_shapes(state0, 2, 16)]
    it = iter(eps)
    orig = rvaes.sample_from_gaussian
    rvaes.sample_from_gaussian = lambda mu, log_sig: mu + log_sig.exp() * next(it)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4)
    try:
        opt.zero_grad()
        logits, kl = model(x)
    finally:
        rvaes.sample_from_gaussian = orig
    import torch.nn.functional as F
    recon = F.binary_cross_entropy_with_logits(logits, x, reduction="none").sum(dim=(1, 2, 3))
    loss = (recon + kl).mean()
    loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e50)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None)
             for k, p in model.named_parameters()}
    opt.step()
    out = {"ctor": "VeryDeepVAE", "kwargs": kwargs, "lr": 5e-4, "x": x, "eps": eps, "state0": state0,
           "logits": logits.detach().clone(), "kl": kl.detach().clone(),
           "recon_mean": recon.mean().detach().clone(), "kl_mean": kl.mean().detach().clone(),
           "loss": loss.detach().clone(), "grads": grads, "grad_norm": norm.detach().clone(),
           "state1": _ref.clone_state(model), "torch_version": torch.__version__}
    path = os.path.join(HERE, "vae_vd_vae_small.pt")
    torch.save(out, path)
    print(f"vd_vae_small: elbo={float(loss):.6f} kl={float(kl.mean()):.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


def make_beta_vae():
    """Beta-VAE fixture (stride-2 4x4 convs + transposed convs, unit-Gaussian KL), noise replayed."""
    ref = _ref.load()
    from pytorch_generative.models.vae import vaes as rvaes
    import torch.nn.functional as F

    torch.manual_seed(0)
    kwargs = dict(in_channels=3, out_channels=3, beta=4.0, latent_channels=8, strides=[2, 4],
                  hidden_channels=16, residual_channels=8)
    model = ref.models.BetaVAE(**kwargs)
    x = make_input((2, 3, 32, 32), "cifar")
    state0 = _ref.clone_state(model)
    eps = torch.randn(2, 8, 4, 4, generator=torch.Generator().manual_seed(4321))
    orig = rvaes.sample_from_gaussian
    rvaes.sample_from_gaussian = lambda mu, log_sig: mu + log_sig.exp() * eps
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    try:
        opt.zero_grad()
        logits, kl = model(x)
    finally:
        rvaes.sample_from_gaussian = orig
    recon = F.binary_cross_entropy_with_logits(logits, x, reduction="none").sum(dim=(1, 2, 3))
    loss = (recon + kl).mean()
    loss.backward()
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e50)
    grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
    opt.step()
    out = {"ctor": "BetaVAE", "kwargs": kwargs, "lr": 1e-3, "x": x, "eps": [eps], "state0": state0,
           "logits": logits.detach().clone(), "kl": kl.detach().clone(),
           "recon_mean": recon.mean().detach().clone(), "kl_mean": kl.mean().detach().clone(),
           "loss": loss.detach().clone(), "grads": grads, "grad_norm": norm.detach().clone(),
           "state1": _ref.clone_state(model), "torch_version": torch.__version__}
    path = os.path.join(HERE, "vae_beta_vae_small.pt")
    torch.save(out, path)
    print(f"beta_vae_small: elbo={float(loss):.6f} kl={float(kl.mean()):.6f} -> {os.path.getsize(path)/1024:.0f} KiB")


if __name__ == "__main__":
    main()
    make_vd_vae()
    make_beta_vae()
