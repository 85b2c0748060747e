"""Gated PixelCNN on the MI355X operator path
(reference models/autoregressive/gated_pixel_cnn.py:31-190).

The pad-then-crop convolutions of the reference ((k//2+1)x1 with padding k//2+1 cropped to the
first h rows, 1x(k//2+1) with padding k//2+mask_center cropped to the first w columns) are
expressed as tap offsets + an output extent, so the cropped rows/columns are never computed.
"""

from torch import nn

from pytorch_generative_amd import nn as pg_nn
from pytorch_generative_amd import ops
from pytorch_generative_amd.models import base


class GatedPixelCNNLayer(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=3, mask_center=False):
        super().__init__()
        assert kernel_size % 2 == 1, "kernel_size cannot be even"
        self._in_channels = in_channels
        self._out_channels = out_channels
        self._activation = pg_nn.GatedActivation()
        self._kernel_size = kernel_size
        self._padding = (kernel_size - 1) // 2
        self._mask_center = mask_center

        self._vstack_1xN = pg_nn.Conv2d(
            in_channels=in_channels,
            out_channels=out_channels,
            kernel_size=(1, kernel_size),
            padding=(0, self._padding),
        )
        self._vstack_Nx1 = pg_nn.Conv2d(
            in_channels=out_channels,
            out_channels=2 * out_channels,
            kernel_size=(kernel_size // 2 + 1, 1),
            padding=(self._padding + 1, 0),
        )
        self._vstack_1x1 = pg_nn.Conv2d(
            in_channels=in_channels, out_channels=2 * out_channels, kernel_size=1
        )
        self._link = pg_nn.Conv2d(
            in_channels=2 * out_channels, out_channels=2 * out_channels, kernel_size=1
        )
        self._hstack_1xN = pg_nn.Conv2d(
            in_channels=in_channels,
            out_channels=2 * out_channels,
            kernel_size=(1, kernel_size // 2 + 1),
            padding=(0, self._padding + int(mask_center)),
        )
        self._hstack_residual = pg_nn.Conv2d(
            in_channels=out_channels, out_channels=out_channels, kernel_size=1
        )
        self._hstack_skip = pg_nn.Conv2d(
            in_channels=out_channels, out_channels=out_channels, kernel_size=1
        )

    def forward(self, vstack_input, hstack_input, skip_acc=None):
        """skip_acc (extension): the running sum of the skip connections, added in `_hstack_skip`'s epilogue
        (gated_pixel_cnn.py:186-189 accumulates it in place). Every tensor with two readers goes through its
        first reader's convolution with a pass-through alias for the second (ops.conv2d_taps, n_skip): no
        gradient-sum kernels in backward."""
        _, _, h, w = vstack_input.shape
        # vertical stack
        t, vin = self._vstack_1xN(vstack_input, n_skip=1)
        vconv = self._vstack_Nx1(t, crop=(h, w))
        link, vconv_s = self._link(vconv, n_skip=1)
        # round 6: both gates run in the epilogue of the convolution that produces their input where Conv2d.gate_ok (the wide bf16x3
        # kernel, gate-interleaved fragments): the gate kernel read the 2C-channel tensor back
        gate = self._activation._gate
        if self._vstack_1x1.gate_ok(vin):
            vstack = self._vstack_1x1(vin, res=vconv_s, gate=gate)
        else:
            vstack = self._activation(self._vstack_1x1(vin, res=vconv_s))
        # horizontal stack
        if self._hstack_1xN.gate_ok(hstack_input, (h, w)):
            hstack, hin = self._hstack_1xN(hstack_input, crop=(h, w), res=link, n_skip=1, gate=gate)
        else:
            hx, hin = self._hstack_1xN(hstack_input, crop=(h, w), res=link, n_skip=1)
            hstack = self._activation(hx)
        skip, hstack_s = self._hstack_skip(hstack, res=skip_acc, n_skip=1)
        # a causal (mask_center) layer must not see its own input through the residual
        hstack = self._hstack_residual(hstack_s, res=None if self._mask_center else hin)
        return vstack, hstack, skip


class GatedPixelCNN(base.AutoregressiveModel):
    _row_decode = True  # every layer is row-causal: sample() runs row by row (models/base.py)
    _row_graph = True
    _row_decode_min_batch = 32  # measured: 16 samples 3.7 s row-cached vs 3.2 s full forwards; 256 samples 3.9 s vs ~24 s

    def __init__(
        self,
        in_channels=1,
        out_channels=1,
        n_gated=10,
        gated_channels=128,
        head_channels=32,
        sample_fn=None,
    ):
        super().__init__(sample_fn)
        self._input = GatedPixelCNNLayer(
            in_channels=in_channels, out_channels=gated_channels, kernel_size=7, mask_center=True
        )
        self._gated_layers = nn.ModuleList(
            [
                GatedPixelCNNLayer(
                    in_channels=gated_channels,
                    out_channels=gated_channels,
                    kernel_size=3,
                    mask_center=False,
                )
                for _ in range(n_gated)
            ]
        )
        self._head = nn.Sequential(
            nn.ReLU(),
            pg_nn.Conv2d(in_channels=gated_channels, out_channels=head_channels, kernel_size=1),
            nn.ReLU(),
            pg_nn.Conv2d(in_channels=head_channels, out_channels=out_channels, kernel_size=1),
        )

    def forward(self, x):
        vstack, hstack, skip_connections = self._input(x, x)
        for gated_layer in self._gated_layers:
            vstack, hstack, skip_connections = gated_layer(vstack, hstack, skip_acc=skip_connections)
        return self._head[3](self._head[1](skip_connections, in_act="relu"), in_act="relu")


def reproduce(n_epochs=457, batch_size=128, log_dir="/tmp/run", n_gpus=1, device_id=0,
              debug_loader=None):
    """The reference's training recipe for this model (gated_pixel_cnn.py:193-250: same model
    hyper-parameters, Adam lr 1e-3, per-batch lr decay 0.9999) on the MI355X path. Arguments as the reference;
    `debug_loader` replaces both loaders (any iterable of (x, y) batches). Returns the Trainer."""
    from pytorch_generative_amd import recipes

    return recipes.run(
        lambda: GatedPixelCNN(in_channels=1, out_channels=1, n_gated=10, gated_channels=128, head_channels=32),
        loaders=recipes.binarized_mnist, loss_fn=recipes.bce_loss, lr=1e-3, lr_decay=0.9999,
        n_epochs=n_epochs, batch_size=batch_size, log_dir=log_dir, n_gpus=n_gpus,
        device_id=device_id, debug_loader=debug_loader)
