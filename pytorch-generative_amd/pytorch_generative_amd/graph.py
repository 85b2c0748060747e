"""hipGraph-captured training step.

At the model sizes of this path (27 k - 3.4 M parameters) an eager step is ~100-600 kernel
launches whose host cost (~3-4 us each + Python/autograd dispatch) rivals their device time, so
the whole reference step (trainer.py:173-193: zero_grad, forward, loss, backward, grad-norm,
Adam, lr decay) is captured ONCE into a hipGraph and replayed per batch with static buffers.
Every kernel of libpg_hip.so is capture-safe by construction (no allocation, no sync, launches
only on the stream it is given).

With data parallelism the RCCL all-reduce of the flat gradient (`pg_allreduce_sum`, csrc/comm.hip: a
direct librccl call on the step's stream) is captured INSIDE the same graph, between backward and
the norm / Adam kernels: still one graph launch per step. Only the development transport (gloo
through pinned host memory, two processes on one GPU) cannot be captured; there the step is split
into two graphs (forward+backward | all-reduce | norm+Adam) around the eager collective.

Warm-up iterations before the capture are real steps; with `preserve_state=True` everything they
touched is rolled back: parameters, Adam moments, step / lr counters AND every module buffer (e.g. the
VectorQuantizer's EMA codebook, which its forward updates in place).

Python inside `forward_fn` / `loss_fn` runs during warm-up and capture only — NOT on every replay.
Step-dependent host logic (annealing schedules, counters, `.item()`) must live outside the captured
step; `trainer.Trainer` therefore uses the graph only when its `train_one_batch` hook is not
overridden (or the caller opts in).
"""

import torch


class GraphedTrainStep:
    def __init__(self, model, optimizer, loss_fn, example_x, reducer=None, warmup_iters=2,
                 example_y=None, forward_fn=None, preserve_state=False):
        """Args:
            model: a pytorch_generative_amd model on the GPU, in train mode.
            optimizer: optim.FlatAdam over model.parameters().
            loss_fn: fn(x, preds) -> scalar loss tensor (HIP op, e.g. ops.bce_with_logits_sum_mean);
                ignored when `forward_fn` is given.
            example_x / example_y: a batch with the static shapes to capture (y may be None).
            reducer: parallel.FlatGradAllReduce or None (single GPU).
            forward_fn: fn(x, y) -> loss tensor or {"loss": ..., other metrics}: the forward + loss
                part of the step (the Trainer passes its overridable train_one_batch hook).
            preserve_state: the warm-up iterations (needed before capture: allocator warm-up, lazy
                module state) are real steps on the example batch; True rolls parameters, Adam
                moments and the step / lr counters back afterwards so the first replay is step 1.
        """
        self.model, self.opt, self.reducer = model, optimizer, reducer
        if forward_fn is None:
            forward_fn = lambda x, y: loss_fn(x, model(x))  # noqa: E731
        self.forward_fn = forward_fn
        self.reduce = reducer is not None and reducer.active
        # force_split: keep the collective OUT of the graph even if it is capturable (fallback when a
        # capture next to a live multi-rank communicator fails: two graphs around an eager collective)
        self.split = self.reduce and (not reducer.capturable or getattr(reducer, "force_split", False))
        self.static_x = example_x.clone()
        self.static_y = None if example_y is None else example_y.clone()
        self.static_out = None

        saved = None
        if preserve_state:
            saved = [t.clone() for t in (optimizer.flat_param, optimizer.exp_avg, optimizer.exp_avg_sq,
                                         optimizer.state_block)]
            saved_buffers = [(b, b.clone()) for b in model.buffers()]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup_iters):
                self._fwd_bwd()
                if self.reduce:
                    reducer.all_reduce()
                self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if saved is not None:
            for dst, src in zip((optimizer.flat_param, optimizer.exp_avg, optimizer.exp_avg_sq,
                                 optimizer.state_block), saved):
                dst.copy_(src)
            for b, src in saved_buffers:
                b.copy_(src)

        # thread_local: RCCL's watchdog thread may touch the HIP runtime while we capture
        mode = dict(capture_error_mode="thread_local")
        # No cyclic garbage collection while a capture is open: a collection that happens to run inside the region may destroy
        # an older step's graph / tensors (closures of the step functions form cycles), and freeing device memory or a graph
        # while the stream is capturing aborts the process (round 6: the reference-suite tier died in the VD-VAE capture once the
        # model's forward created enough objects to trigger a collection there). torch.cuda.graph collects BEFORE it begins.
        import gc

        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            self.graph_a = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph_a, **mode):
                self.static_out = self._fwd_bwd()
                if not self.split:
                    if self.reduce:
                        reducer.all_reduce()  # captured: an ordinary stream op of the C-ABI
                    self.opt.step()
            self.graph_b = None
            if self.split:
                self.graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_b, **mode):
                    self.opt.step()
        finally:
            if gc_was_on:
                gc.enable()

    def _fwd_bwd(self):
        self.opt.zero_grad()
        out = self.forward_fn(self.static_x, self.static_y)
        loss = out["loss"] if isinstance(out, dict) else out
        loss.backward()
        if isinstance(out, dict):
            return {k: v.detach() for k, v in out.items()}
        return loss.detach()

    def __call__(self, x=None, y=None):
        """Runs one step on batch `x` (None: reuse the resident batch). Returns the device loss
        (or the metrics dict of `forward_fn`) — static tensors, overwritten by the next call."""
        if x is not None:
            self.static_x.copy_(x, non_blocking=True)
        if y is not None and self.static_y is not None:
            self.static_y.copy_(y, non_blocking=True)
        self.graph_a.replay()
        if self.split:
            self.reducer.all_reduce()
            self.graph_b.replay()
        return self.static_out
