"""Model base classes (behavioural mirror of the reference's models/base.py:28-120).

* GenerativeModel remembers the (C, H, W) of the first 4-D batch it is called with in lazily
  registered `_c/_h/_w` buffers (they are part of the state_dict, base.py:41-61).
* AutoregressiveModel.sample() fills pixels in raster order, one full forward per pixel,
  replacing only entries < 0 of `conditioned_on` (base.py:97-120).
"""

import abc

import torch
from torch import distributions, nn


def _bernoulli_from_logits(logits):
    return distributions.Bernoulli(logits=logits).sample()


class GenerativeModel(abc.ABC, nn.Module):
    def __call__(self, x, *args, **kwargs):
        if getattr(self, "_c", None) is None and x.dim() == 4:
            self._register_shape(*x.shape[1:])
        return super().__call__(x, *args, **kwargs)

    def _register_shape(self, c, h, w):
        for name, value in (("_c", c), ("_h", h), ("_w", w)):
            self.register_buffer(name, value if torch.is_tensor(value) else torch.tensor(value))

    def load_state_dict(self, state_dict, strict=True):
        if "_c" in state_dict and not getattr(self, "_c", None):
            self._register_shape(state_dict["_c"], state_dict["_h"], state_dict["_w"])
        return super().load_state_dict(state_dict, strict)

    @property
    def device(self):
        return next(self.parameters()).device

    @abc.abstractmethod
    def sample(self, n_samples):
        ...


class AutoregressiveModel(GenerativeModel):
    def __init__(self, sample_fn=None):
        """sample_fn: fn(logits) -> sample; defaults to Bernoulli(logits)."""
        super().__init__()
        self._sample_fn = sample_fn or _bernoulli_from_logits

    def _start_canvas(self, n_samples, conditioned_on):
        assert (
            n_samples is not None or conditioned_on is not None
        ), 'Must provided one, and only one, of "n_samples" or "conditioned_on"'
        if conditioned_on is not None:
            return conditioned_on.clone()
        shape = (n_samples, int(self._c), int(self._h), int(self._w))
        return torch.full(shape, -1.0, device=self.device)

    # Models whose every layer is row-causal set this: sample() then evaluates one image ROW per step
    # against per-layer caches of the rows above (ops.RowDecode) instead of the whole image.
    _row_decode = False
    _row_graph = False            # the row step is identical for every row (no attention): hipGraph it
    _row_decode_min_batch = 1     # below this batch the per-pixel full forward is used (launch-bound row steps)

    @torch.no_grad()
    def sample(self, n_samples=None, conditioned_on=None, *, incremental=True, return_logits=False):
        """Generates samples; entries of `conditioned_on` that are >= 0 are kept as given
        (reference models/base.py:97-120: raster order, `_sample_fn` once per pixel).

        incremental (extension): the reference runs a full forward per pixel; row-causal models here
        run the forward on the CURRENT ROW only, every convolution reading the few rows above it from
        a private cache and attention its cached keys / values (same logits, ~H x less arithmetic).
        return_logits (extension): also returns the (N, C, H, W) logits the draws were made from."""
        canvas = self._start_canvas(n_samples, conditioned_on)
        n, c, h, w = canvas.shape
        logits_map = torch.zeros_like(canvas) if return_logits else None
        if incremental and self._row_decode and (n >= self._row_decode_min_batch or return_logits):
            from pytorch_generative_amd import ops

            def reset():
                for m in self.modules():
                    if hasattr(m, "_row_reset"):
                        m._row_reset()

            reset()
            row_in = torch.empty((n, c, 1, w), device=canvas.device, dtype=canvas.dtype)
            with ops.RowDecode(h) as ctx:
                # Row-invariant models (no attention): the row step touches the same buffers for every
                # row, so it is captured ONCE into two hipGraphs (evaluate / evaluate-and-commit) and
                # replayed H*W + H times — the eager row step is ~100 tiny launches, launch bound.
                step_graph = commit_graph = step_out = None
                if self._row_graph:
                    try:
                        row_in.fill_(0.0)
                        side = torch.cuda.Stream()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):  # warm-up: allocates every band buffer
                            ctx.commit = False
                            self.forward(row_in)
                            ctx.commit = True
                            self.forward(row_in)
                        torch.cuda.current_stream().wait_stream(side)
                        step_graph, commit_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                        ctx.commit = False
                        with torch.cuda.graph(step_graph, capture_error_mode="thread_local"):
                            step_out = self.forward(row_in)
                        ctx.commit = True
                        with torch.cuda.graph(commit_graph, capture_error_mode="thread_local"):
                            self.forward(row_in)
                        for m in self.modules():  # the warm-up pushed rows of zeros: start clean
                            band = getattr(m, "_row_band", None)
                            if band is not None:
                                band.zero_()
                    except Exception as e:  # noqa: BLE001 — capture is an optimisation only
                        import os
                        if os.environ.get("PG_DEBUG"):
                            print(f"[sample] row-step capture failed: {type(e).__name__}: {e}")
                        step_graph = commit_graph = None
                        torch.cuda.synchronize()
                        reset()
                for row in range(h):
                    ctx.row, ctx.commit = row, False
                    for col in range(w):
                        row_in.copy_(canvas[:, :, row:row + 1, :])
                        if step_graph is not None:
                            step_graph.replay()
                            logits = step_out[:, :, 0, col]
                        else:
                            logits = self.forward(row_in)[:, :, 0, col]
                        if return_logits:
                            logits_map[:, :, row, col] = logits
                        drawn = self._sample_fn(logits).view(n, c)
                        current = canvas[:, :, row, col]
                        canvas[:, :, row, col] = torch.where(current < 0, drawn, current)
                    ctx.commit = True  # the row is final: push it into every layer's cache
                    row_in.copy_(canvas[:, :, row:row + 1, :])
                    if commit_graph is not None:
                        commit_graph.replay()
                    else:
                        self.forward(row_in)
            reset()
            return (canvas, logits_map) if return_logits else canvas
        for row in range(h):
            for col in range(w):
                logits = self.forward(canvas)[:, :, row, col]
                if return_logits:
                    logits_map[:, :, row, col] = logits
                drawn = self._sample_fn(logits).view(n, c)
                current = canvas[:, :, row, col]
                canvas[:, :, row, col] = torch.where(current < 0, drawn, current)
        return (canvas, logits_map) if return_logits else canvas
