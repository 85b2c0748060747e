"""hipGraph-captured training step.

At the model sizes of this path (27 k - 3.4 M parameters) an eager step is ~100-600 kernel
launches whose host cost (~3-4 us each + Python/autograd dispatch) rivals their device time, so
the whole reference step (trainer.py:173-193: zero_grad, forward, loss, backward, grad-norm,
Adam, lr decay) is captured ONCE into a hipGraph and replayed per batch with static buffers.
Every kernel of libpg_hip.so is capture-safe by construction (no allocation, no sync, launches
only on the stream it is given).

With data parallelism the RCCL all-reduce of the flat gradient sits between two graphs
(forward+backward | all-reduce | norm+Adam) so the collective itself stays an ordinary eager
RCCL call on a side stream.
"""

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, example_x, reducer=None, warmup_iters=2):
        """Args:
            model: a pytorch_generative_amd model on the GPU, in train mode.
            optimizer: optim.FlatAdam over model.parameters().
            loss_fn: fn(x, preds) -> scalar loss tensor (HIP op, e.g. ops.bce_with_logits_sum_mean).
            example_x: a batch with the static shape to capture.
            reducer: parallel.FlatGradAllReduce or None (single GPU).
        NOTE: warm-up iterations run real optimisation steps on `example_x`.
        """
        self.model, self.opt, self.loss_fn, self.reducer = model, optimizer, loss_fn, reducer
        self.split = reducer is not None and reducer.world > 1
        self.static_x = example_x.clone()
        self.static_loss = None

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                self._fwd_bwd()
                if self.split:
                    reducer.all_reduce()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()

        # thread_local: RCCL's watchdog thread may touch the HIP runtime while we capture
        mode = dict(capture_error_mode="thread_local")
        self.graph_a = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_a, **mode):
            self.static_loss = self._fwd_bwd()
            if not self.split:
                self.opt.step()
        self.graph_b = None
        if self.split:
            self.graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_b, **mode):
                self.opt.step()

    def _fwd_bwd(self):
        self.opt.zero_grad()
        loss = self.loss_fn(self.static_x, self.model(self.static_x))
        loss.backward()
        return loss.detach()

    def __call__(self, x=None):
        """Runs one step on batch `x` (None: reuse the resident batch). Returns the device loss."""
        if x is not None:
            self.static_x.copy_(x, non_blocking=True)
        self.graph_a.replay()
        if self.split:
            self.reducer.all_reduce()
            self.graph_b.replay()
        return self.static_loss
