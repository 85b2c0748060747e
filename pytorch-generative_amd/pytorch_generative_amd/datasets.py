"""Data loaders for the reproduce() recipes.

The reference's datasets.py (pytorch_generative/datasets.py:28-178) wraps torchvision's MNIST /
CIFAR-10 downloads. Neither torchvision nor a network exists on the MI355X boxes this path is
built for, and the training path only needs tensors of the right shape and value range, so the
loaders here produce SYNTHETIC batches with the statistics SURVEY.md §8(d) fixes (binarised-MNIST
shaped Bernoulli(0.1307) pixels; CIFAR shaped uniform 8-bit values / 255), generated on the device
once per loader. Data parallel: the reference has no DistributedSampler — every rank shuffles the whole
set with its own RNG (datasets.py:53-58) — so here every rank draws its OWN batches: the seed is
`base seed + rank` ("seed 1234 + r"), never the same batches on all ranks. Real data: pass any iterable of (x, y) batches as `debug_loader` to reproduce(),
or to Trainer directly.
"""

import warnings

import torch


class SyntheticLoader:
    """`n_batches` fixed random batches per epoch, resident on `device` (no host traffic)."""

    def __init__(self, shape, n_batches, kind, device=None, seed=1234):
        device = device or torch.device("cuda", torch.cuda.current_device())
        g = torch.Generator().manual_seed(seed)
        self.batches = []
        for _ in range(n_batches):
            if kind == "bernoulli":
                x = torch.bernoulli(torch.full(shape, 0.1307), generator=g)
            elif kind == "uniform8":
                x = torch.randint(0, 256, shape, generator=g).float() / 255
            elif kind == "uniform8_normalized":  # transforms.Normalize of datasets.py:170-174
                x = torch.randint(0, 256, shape, generator=g).float() / 255
                mean = torch.tensor((0.4914, 0.4822, 0.4465)).view(1, 3, 1, 1)
                std = torch.tensor((0.2023, 0.1994, 0.2010)).view(1, 3, 1, 1)
                x = (x - mean) / std
            elif kind == "dequantized":
                x = (torch.randint(0, 256, shape, generator=g).float() + torch.rand(shape, generator=g)) / 256
            else:
                raise ValueError(f"unknown synthetic data kind {kind!r}")
            self.batches.append((x.to(device), torch.zeros(shape[0], dtype=torch.long, device=device)))

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def _rank():
    import torch.distributed as dist

    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _pair(shape, kind, train_batches, test_batches):
    warnings.warn("pytorch_generative_amd.datasets: no dataset download on this path — using synthetic "
                  f"{kind} batches of shape {tuple(shape)}")
    r = _rank()  # per-rank batches (the loaders are built after Trainer / train.py initialised the group)
    return (SyntheticLoader(shape, train_batches, kind, seed=1234 + r),
            SyntheticLoader(shape, test_batches, kind, seed=4321 + r))


def get_mnist_loaders(batch_size, dynamically_binarize=False, dequantize=False, resize_to_32=False,
                      train_batches=64, test_batches=8):
    """(train_loader, test_loader) of MNIST-shaped synthetic batches; arguments as the reference's
    get_mnist_loaders (datasets.py:28-67)."""
    if dynamically_binarize and dequantize:
        raise ValueError("Cannot specify both dynamically_binarize and dequantize.")
    size = 32 if resize_to_32 else 28
    kind = "bernoulli" if dynamically_binarize else ("dequantized" if dequantize else "uniform8")
    return _pair((batch_size, 1, size, size), kind, train_batches, test_batches)


def get_cifar10_loaders(batch_size, normalize=False, train_batches=64, test_batches=8):
    """(train_loader, test_loader) of CIFAR-10-shaped synthetic batches (datasets.py:156-178)."""
    return _pair((batch_size, 3, 32, 32), "uniform8_normalized" if normalize else "uniform8",
                 train_batches, test_batches)
