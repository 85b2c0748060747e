"""The shared body of every model module's reproduce() (reference e.g.
models/autoregressive/image_gpt.py:112-175, models/vae/vae.py:104-167).

A reference recipe is: loaders -> model -> Adam (+ per-batch MultiplicativeLR) -> loss_fn ->
Trainer(...).interleaved_train_and_eval(n_epochs). The MI355X version keeps the hyper-parameters
and the Trainer contract and swaps the moving parts for the native ones: the model is built on
its GPU, `optim.FlatAdam` holds parameters / gradients / moments in flat buffers with the lr decay
applied on the device, the losses are the HIP loss kernels, and the Trainer replays the step from
a hipGraph.
"""

import torch

from pytorch_generative_amd import datasets, ops, optim, trainer


def bce_loss(x, _, preds):
    """sum-over-pixels, mean-over-batch BCE with logits (image_gpt.py:158-162 and the other
    autoregressive recipes)."""
    return ops.bce_with_logits_sum_mean(preds, x)


def elbo_loss(x, _, preds):
    """{"recon_loss", "kl_div", "loss"} of the VAE recipes (vae.py:149-159)."""
    logits, kl_div = preds
    recon, kl = ops.elbo_terms(logits, x, kl_div)
    return {"recon_loss": recon, "kl_div": kl, "loss": recon + kl}


def _device(n_gpus, device_id):
    if not torch.cuda.is_available():
        raise RuntimeError("reproduce(): this path needs an MI355X (no CPU fallback)")
    index = device_id if (n_gpus > 1 and device_id is not None) else torch.cuda.current_device()
    torch.cuda.set_device(index)
    return torch.device("cuda", index)


def run(build_model, *, loaders, loss_fn, lr, lr_decay=1.0, n_epochs, batch_size, log_dir, n_gpus,
        device_id, debug_loader):
    """Builds everything a recipe names and trains for `n_epochs`; returns the Trainer."""
    device = _device(n_gpus, device_id)
    if debug_loader is not None:
        train_loader = test_loader = debug_loader
    else:
        train_loader, test_loader = loaders(batch_size)
    model = build_model().to(device)
    optimizer = optim.FlatAdam(model.parameters(), lr=lr, lr_decay=lr_decay)
    t = trainer.Trainer(model=model, loss_fn=loss_fn, optimizer=optimizer, train_loader=train_loader,
                        eval_loader=test_loader, log_dir=log_dir, n_gpus=n_gpus, device_id=device_id)
    t.interleaved_train_and_eval(n_epochs)
    return t


def binarized_mnist(batch_size):
    return datasets.get_mnist_loaders(batch_size, dynamically_binarize=True)


def binarized_mnist_32(batch_size):
    return datasets.get_mnist_loaders(batch_size, dynamically_binarize=True, resize_to_32=True)


def cifar10(batch_size):
    return datasets.get_cifar10_loaders(batch_size, normalize=True)


def vq_loss(vq_weight):
    """loss_fn of the VQ-VAE recipes (vq_vae.py:127-136, vq_vae_2.py:160-169):
    {"vq_loss", "reconstruction_loss", "loss" = MSE + vq_weight * quantization loss}."""
    from pytorch_generative_amd.nn import utils as nn_utils

    def loss_fn(x, _, preds):
        recon, quantization_loss = preds
        recon_loss = nn_utils.mse_loss(recon, x)
        return {"vq_loss": quantization_loss, "reconstruction_loss": recon_loss,
                "loss": recon_loss + vq_weight * quantization_loss}

    return loss_fn
