"""Generates tests/golden/vq_*.pt from the REAL reference (run in the build container only):

    python tests/golden/make_vq_golden.py

SURVEY.md §8(f) rank 4: VectorQuantizer / VQ-VAE have no numeric vectors in the reference's tests,
so the oracle restatement (oracle.ops.vector_quantize, oracle.models.vq_vae) is pinned against
outputs of the reference itself:
  vq_quantizer.pt   VectorQuantizer.forward (nn/utils.py:53-96) in the three modes — EMA + training
                    (buffers before / after), EMA + eval, gradient-descent codebook — on a seeded input
  vq_vae_2_small.pt one forward + backward of a small VectorQuantizedVAE2 (vq_vae_2.py:96-110)
  vq_vae_small.pt   one training step of a small VectorQuantizedVAE (vq_vae.py:69-81, loss of
                    vq_vae.py:127-136): state before, reconstruction, losses, every gradient,
                    state after forward (EMA buffers move in forward) and after the Adam step
"""

import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import _ref  # noqa: E402


def main():
    ref = _ref.load()
    import torch.nn.functional as F

    out = {"torch_version": torch.__version__, "cases": {}}
    g = torch.Generator().manual_seed(99)
    x = torch.randn(3, 8, 5, 6, generator=g)
    for name, use_ema, training in (("ema_train", True, True), ("ema_eval", True, False),
                                    ("sgd_train", False, True)):
        torch.manual_seed(3)
        vq = ref.nn.VectorQuantizer(n_embeddings=12, embedding_dim=8, use_ema=use_ema)
        vq.train(training)
        if use_ema:  # a used codebook: non-trivial cluster sizes / averages
            with torch.no_grad():
                vq._cluster_size.uniform_(0.5, 4.0)
                vq._embedding_avg.copy_(vq._embedding * vq._cluster_size.unsqueeze(1))
        before = _ref.clone_state(vq)
        xin = x.clone().requires_grad_(True)
        q, loss = vq(xin)
        (q.sum() * 0.5 + loss).backward()
        out["cases"][name] = {
            "use_ema": use_ema, "training": training, "x": x.clone(), "before": before,
            "quantized": q.detach().clone(), "loss": loss.detach().clone(),
            "dx": xin.grad.detach().clone(),
            "d_embedding": None if use_ema else vq._embedding.grad.detach().clone(),
            "after": _ref.clone_state(vq),
        }
    path = os.path.join(HERE, "vq_quantizer.pt")
    torch.save(out, path)
    print(f"vq_quantizer: {os.path.getsize(path) / 1024:.0f} KiB")

    kwargs = dict(in_channels=3, out_channels=3, hidden_channels=16, n_residual_blocks=1,
                  residual_channels=8, n_embeddings=10, embedding_dim=4)
    torch.manual_seed(0)
    model = ref.models.VectorQuantizedVAE(**kwargs)
    model.train()
    x = torch.randint(0, 256, (2, 3, 16, 16), generator=torch.Generator().manual_seed(1234)).float() / 255
    state0 = _ref.clone_state(model)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    opt.zero_grad()
    recon, vq_loss = model(x)
    loss = F.mse_loss(recon, x) + vq_loss
    loss.backward()
    state_fwd = _ref.clone_state(model)
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 1e50)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None)
             for k, p in model.named_parameters()}
    opt.step()
    rec = {"ctor": "VectorQuantizedVAE", "kwargs": kwargs, "lr": 2e-4, "x": x, "state0": state0,
           "recon": recon.detach().clone(), "vq_loss": vq_loss.detach().clone(),
           "loss": loss.detach().clone(), "grads": grads, "grad_norm": norm.detach().clone(),
           "state_after_forward": state_fwd, "state1": _ref.clone_state(model),
           "torch_version": torch.__version__}
    path = os.path.join(HERE, "vq_vae_small.pt")
    torch.save(rec, path)
    print(f"vq_vae_small: loss={float(loss.detach()):.6f} vq={float(vq_loss.detach()):.6f} norm={float(norm):.6f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")

    # VQ-VAE-2 (vq_vae_2.py:96-110), one forward + backward of a small model
    kwargs = dict(in_channels=3, out_channels=3, hidden_channels=16, n_residual_blocks=1,
                  residual_channels=8, n_embeddings=10, embedding_dim=4)
    torch.manual_seed(1)
    model = ref.models.VectorQuantizedVAE2(**kwargs)
    model.train()
    state0 = _ref.clone_state(model)
    xhat, vq_loss = model(x)
    loss = F.mse_loss(xhat, x) + vq_loss
    loss.backward()
    grads = {k: (p.grad.detach().clone() if p.grad is not None else None)
             for k, p in model.named_parameters()}
    rec = {"ctor": "VectorQuantizedVAE2", "kwargs": kwargs, "x": x, "state0": state0,
           "recon": xhat.detach().clone(), "vq_loss": vq_loss.detach().clone(),
           "loss": loss.detach().clone(), "grads": grads,
           "state_after_forward": _ref.clone_state(model), "torch_version": torch.__version__}
    path = os.path.join(HERE, "vq_vae_2_small.pt")
    torch.save(rec, path)
    print(f"vq_vae_2_small: loss={float(loss.detach()):.6f} vq={float(vq_loss.detach()):.6f} "
          f"-> {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
