"""PixelCNN on the MI355X operator path (reference models/autoregressive/pixel_cnn.py:23-110).

ReLUs are fused into the following convolution's input load and the block residual into the
last 1x1 kernel's epilogue; the doubled residual of the reference (`x + layer(x)` where
layer(x) = x + net(x), :52-53 vs :109) is reproduced.
"""

from torch import nn

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models import base


class CausalResidualBlock(nn.Module):
    def __init__(self, n_channels):
        super().__init__()
        self._net = nn.Sequential(
            nn.ReLU(),
            pg_nn.Conv2d(in_channels=n_channels, out_channels=n_channels // 2, kernel_size=1),
            nn.ReLU(),
            pg_nn.CausalConv2d(
                mask_center=False,
                in_channels=n_channels // 2,
                out_channels=n_channels // 2,
                kernel_size=3,
                padding=1,
            ),
            nn.ReLU(),
            pg_nn.Conv2d(in_channels=n_channels // 2, out_channels=n_channels, kernel_size=1),
        )

    def forward(self, x, double=False):
        """double=True (extension) returns x + (x + net(x)) — the block residual AND the model loop's
        (pixel_cnn.py:52-53, :109) — with both adds in the last convolution's epilogue; x's three readers
        go through the first convolution's pass-through aliases, so backward has no gradient-sum kernels."""
        if double:
            h, x1, x2 = self._net[1](x, in_act="relu", n_skip=2)
            h = self._net[3](h, in_act="relu")
            return self._net[5](h, in_act="relu", res=x1, res2=x2)
        h = self._net[1](x, in_act="relu")
        h = self._net[3](h, in_act="relu")
        return self._net[5](h, in_act="relu", res=x)


class PixelCNN(base.AutoregressiveModel):
    _row_decode = True  # every layer is row-causal: sample() runs row by row (models/base.py)
    _row_graph = True

    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        n_residual=15,
        residual_channels=128,
        head_channels=32,
        sample_fn=None,
    ):
        super().__init__(sample_fn)
        self._input = pg_nn.CausalConv2d(
            mask_center=True,
            in_channels=in_channels,
            out_channels=2 * residual_channels,
            kernel_size=7,
            padding=3,
        )
        self._causal_layers = nn.ModuleList(
            [CausalResidualBlock(n_channels=2 * residual_channels) for _ in range(n_residual)]
        )
        self._head = nn.Sequential(
            nn.ReLU(),
            pg_nn.Conv2d(
                in_channels=2 * residual_channels, out_channels=head_channels, kernel_size=1
            ),
            nn.ReLU(),
            pg_nn.Conv2d(in_channels=head_channels, out_channels=out_channels, kernel_size=1),
        )

    def forward(self, x):
        x = self._input(x)
        for layer in self._causal_layers:
            x = layer(x, double=True)
        return self._head[3](self._head[1](x, in_act="relu"), in_act="relu")


def reproduce(n_epochs=457, batch_size=256, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (pixel_cnn.py:113-174: same model
    hyper-parameters, Adam lr 1e-3, per-batch lr decay 0.999977) on the MI355X path. Arguments as the reference;
    `debug_loader` replaces both loaders (any iterable of (x, y) batches). Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: PixelCNN(in_channels=1, out_channels=1, n_residual=15, residual_channels=16, head_channels=32),
        loaders=recipes.binarized_mnist, loss_fn=recipes.bce_loss, lr=1e-3, lr_decay=0.999977,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
