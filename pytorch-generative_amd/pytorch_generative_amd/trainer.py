"""Trainer for the MI355X operator path.

Contract kept from the reference (pytorch_generative/trainer.py): the constructor signature
(:23-38), what one training step means (:173-193 — zero_grad, model(x), loss_fn(x, y, preds),
backward, global grad norm with optional clip / skip, optimizer.step, lr_scheduler.step), the
overridable `train_one_batch` / `eval_one_batch` hooks, example-weighted evaluation averages, and
the on-disk checkpoint `trainer_state_{epoch}.ckpt` with keys model / optimizer / step / epoch /
examples_processed / time_taken (/ lr_scheduler), written by rank 0 only (:98-148). A checkpoint
written by the reference's single-process Trainer restores here (tested against one written by the
reference itself); reference multi-GPU checkpoints carry DDP's "module." key prefix, which is stripped
on load. The other direction: checkpoints written here restore into the reference when the optimizer
state is converted (`FlatAdam.state_dict()` has `torch.optim.Adam`'s layout) — recipes whose lr decay
runs on the device (`FlatAdam(lr_decay=...)`, no `lr_scheduler` object) also write an equivalent
`MultiplicativeLR` state under "lr_scheduler" so that the reference's `restore_checkpoint` finds the key.

How a step runs here:
  * with `optim.FlatAdam` the whole step (zero_grad .. Adam, incl. the clip) is captured into a
    hipGraph per input shape (graph.GraphedTrainStep) and replayed: one launch instead of
    hundreds; only the final metric read-back touches the host (every `metrics_every` steps).
    A captured step runs the Python of `train_one_batch` / `loss_fn` ONLY while it is captured, not
    on every batch as the reference does: a Trainer subclass that overrides `train_one_batch`
    (step-dependent host logic: annealing, counters, `.item()`) therefore runs eager launches unless
    it passes `graph="force"`; a capture that fails falls back to eager launches with a warning;
  * `skip_grad_norm` needs the norm on the host before the update, and stock torch optimizers have
    no flat buffer: both run as eager kernel launches;
  * `n_gpus > 1`: one process per GPU (launched by train.py / torchrun), this process pinned to
    `device_id`, ONE flat RCCL all-reduce per step (`pg_allreduce_sum`), captured inside the graph.
TensorBoard is used when it is importable; otherwise metrics are only returned to the caller.
"""

import glob
import os
import re
import tempfile
import time
import warnings

import torch
import torch.distributed as dist

from pytorch_generative_amd import graph as pg_graph
from pytorch_generative_amd import optim as pg_optim
from pytorch_generative_amd import parallel

_CKPT_RE = re.compile(r"trainer_state_(\d+)\.ckpt$")


def _ckpt_name(epoch):
    return f"trainer_state_{epoch}.ckpt"


class _Scalars:
    """TensorBoard sink if the package is installed, else a no-op with the same three calls."""

    def __init__(self, log_dir, purge_step=None):
        self._w = None
        try:
            from torch.utils import tensorboard

            kw = {} if purge_step is None else {"purge_step": purge_step}
            self._w = tensorboard.SummaryWriter(log_dir, max_queue=100, **kw)
        except Exception:  # not installed in this image
            pass

    @property
    def active(self):
        return self._w is not None

    def group(self, tag, values, step):
        if self._w is not None:
            self._w.add_scalars(tag, values, step)

    def scalar(self, tag, value, step):
        if self._w is not None:
            self._w.add_scalar(tag, value, step)

    def images(self, tag, tensor, step):
        if self._w is not None:
            self._w.add_images(tag, tensor, step)

    def close(self):
        if self._w is not None:
            self._w.close()


def _split_batch(batch):
    if isinstance(batch, (tuple, list)):
        return (batch[0], batch[1]) if len(batch) > 1 else (batch[0], None)
    return batch, None


def _as_metrics(out):
    metrics = out if isinstance(out, dict) else {"loss": out}
    assert "loss" in metrics, 'Metrics dictionary does not contain "loss" key.'
    return metrics


class Trainer:
    def __init__(
        self,
        model,
        loss_fn,
        optimizer,
        train_loader,
        eval_loader,
        lr_scheduler=None,
        clip_grad_norm=None,
        skip_grad_norm=None,
        log_dir=None,
        sample_epochs=3,
        save_checkpoint_epochs=1,
        n_gpus=0,
        device_id=None,
        *,
        graph=True,
        metrics_every=1,
    ):
        """Arguments as the reference Trainer (trainer.py:40-76). Extensions (keyword only):
        graph: True (default) captures FlatAdam steps into hipGraphs unless `train_one_batch` is
        overridden by a subclass; "force" captures even then (the hook's Python then runs at capture
        time only); False = eager launches. metrics_every: read the step
        metrics back to the host every N steps (1 = the reference's per-step logging)."""
        self.loss_fn = loss_fn
        self.train_loader, self.eval_loader = train_loader, eval_loader
        self.clip_grad_norm, self.skip_grad_norm = clip_grad_norm, skip_grad_norm
        self.log_dir = log_dir or tempfile.mkdtemp()
        os.makedirs(self.log_dir, exist_ok=True)  # the reference's SummaryWriter creates it (trainer.py:87)
        self.sample_epochs, self.save_checkpoint_epochs = sample_epochs, save_checkpoint_epochs

        self.device = self._resolve_device(model, n_gpus, device_id)
        self.device_id = self.device.index if n_gpus > 1 else 0
        # every raw kernel launch of this process goes to the CURRENT device's streams: pin it
        torch.cuda.set_device(self.device)
        self._flat = isinstance(optimizer, pg_optim.FlatAdam)
        if next(model.parameters()).device != self.device:
            if self._flat:
                raise RuntimeError("move the model to its GPU before building FlatAdam")
            model = model.to(self.device)
        self.model, self.optimizer, self.lr_scheduler = model, optimizer, lr_scheduler

        self._reducer = None
        # rank 0 writes checkpoints (the reference uses device_id == 0 for that, trainer.py:99; the
        # rank is what is meant, and the two differ when ranks are pinned to one device for testing)
        self._is_main = (dist.get_rank() == 0) if (n_gpus > 1 and dist.is_initialized()) else True
        if n_gpus > 1:
            if not self._flat:
                raise RuntimeError("multi-GPU training requires pytorch_generative_amd.optim.FlatAdam")
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised (backend 'nccl' = RCCL)")
            self._reducer = parallel.FlatGradAllReduce(optimizer)
            self._reducer.broadcast_parameters(src=0)
        if self._flat:
            optimizer.set_max_norm(clip_grad_norm or skip_grad_norm)
        hook_overridden = type(self).train_one_batch is not Trainer.train_one_batch
        self._use_graph = (bool(graph) and self._flat and not skip_grad_norm
                           and (graph == "force" or not hook_overridden))
        self._graphs = {}
        self._metrics_every = max(1, int(metrics_every))

        self._step = self._epoch = self._examples_processed = 0
        self._time_taken = 0.0
        self._log = _Scalars(self.log_dir)

    @staticmethod
    def _resolve_device(model, n_gpus, device_id):
        if not torch.cuda.is_available():
            raise RuntimeError("pytorch_generative_amd.Trainer needs an MI355X: the HIP operator "
                               "path has no CPU fallback")
        if n_gpus > 1:
            assert device_id is not None, "'device_id' must be provided if n_gpus > 1."
            return torch.device("cuda", device_id)
        where = next(model.parameters()).device
        if n_gpus == 0:
            # the reference trains on the CPU here (trainer.py:77); this path has no CPU arithmetic
            warnings.warn("n_gpus=0 requested, but the HIP operator path only runs on the GPU: "
                          "training on " + (str(where) if where.type == "cuda" else "cuda:0"))
        if where.type == "cuda":
            return where
        return torch.device("cuda", device_id if device_id is not None else torch.cuda.current_device())

    # ------------------------------------------------------------------ checkpoints
    def _path(self, name):
        return os.path.join(self.log_dir, name)

    def _save_checkpoint(self):
        if not self._is_main or self._epoch % self.save_checkpoint_epochs:
            return
        payload = dict(
            model=self.model.state_dict(),
            optimizer=self.optimizer.state_dict(),
            step=self._step,
            epoch=self._epoch,
            examples_processed=self._examples_processed,
            time_taken=self._time_taken,
        )
        if self.lr_scheduler is not None:
            payload["lr_scheduler"] = self.lr_scheduler.state_dict()
        elif self._flat and self.optimizer._lr_decay != 1.0:
            # device-side per-batch decay: the state a MultiplicativeLR(optimizer, lambda _: decay)
            # (e.g. image_gpt.py:156) would hold after `step` calls, so the reference finds its key
            payload["lr_scheduler"] = {
                "base_lrs": [float(self.optimizer.defaults["lr"])], "last_epoch": self._step, "verbose": False,
                "_step_count": self._step + 1, "_get_lr_called_within_step": False,
                "_last_lr": [self.optimizer.current_lr()], "lr_lambdas": [None],
            }
        torch.save(payload, self._path(_ckpt_name(self._epoch)))

    def _latest_epoch(self):
        found = [int(m.group(1)) for m in map(_CKPT_RE.search, glob.glob(self._path("trainer_state_*.ckpt"))) if m]
        if not found:
            raise FileNotFoundError(f"No checkpoints found in {self.log_dir}.")
        print(f"Found {len(found)} saved checkpoints.")
        return max(found)

    def restore_checkpoint(self, epoch=None):
        """Restores model / optimizer / counters from `epoch` (default: the newest checkpoint)."""
        epoch = epoch or self._latest_epoch()
        print(f"Restoring trainer state from checkpoint {_ckpt_name(epoch)}.")
        state = torch.load(self._path(_ckpt_name(epoch)), map_location=self.device, weights_only=False)
        model_state = state["model"]
        if model_state and all(k.startswith("module.") for k in model_state):  # written under DDP
            model_state = {k[len("module."):]: v for k, v in model_state.items()}
        self.model.load_state_dict(model_state)
        self.optimizer.load_state_dict(state["optimizer"])
        if self.lr_scheduler is not None and "lr_scheduler" in state:
            self.lr_scheduler.load_state_dict(state["lr_scheduler"])
        self._step, self._epoch = state["step"], state["epoch"]
        self._examples_processed, self._time_taken = state["examples_processed"], state["time_taken"]
        self._log.close()
        self._log = _Scalars(self.log_dir, purge_step=self._step)

    # ------------------------------------------------------------------ hooks (overridable)
    def train_one_batch(self, x, y):
        """Forward + loss for one training batch; override for custom steps."""
        return self.loss_fn(x, y, self.model(x))

    def eval_one_batch(self, x, y):
        return self.loss_fn(x, y, self.model(x))

    # ------------------------------------------------------------------ one step
    def _to_device(self, x, y):
        x = x.to(self.device, non_blocking=True)
        if y is not None:
            y = y.to(self.device, non_blocking=True)
        return x, y

    def _graphed(self, x, y):
        key = (tuple(x.shape), None if y is None else (tuple(y.shape), y.dtype))
        g = self._graphs.get(key)
        if g is None:
            if len(self._graphs) >= 4:  # e.g. ragged batch sizes: do not hoard graphs
                return None
            try:
                g = pg_graph.GraphedTrainStep(
                    self.model, self.optimizer, None, x, reducer=self._reducer, example_y=y,
                    forward_fn=lambda xx, yy: _as_metrics(self.train_one_batch(xx, yy)),
                    preserve_state=True,
                )
            except Exception as e:  # e.g. a hook that synchronises: run this Trainer eagerly from now on
                warnings.warn(f"hipGraph capture of the training step failed ({type(e).__name__}: {e}); "
                              "falling back to eager kernel launches")
                torch.cuda.synchronize()
                self._use_graph = False
                return None
            self._graphs[key] = g
        return g

    def _train_one_batch(self, x, y, want_metrics=True):
        """One optimisation step; returns the step's metrics as Python floats (or None when
        `want_metrics` is False: nothing is read back, the host does not wait for the GPU)."""
        self.model.train()
        x, y = self._to_device(x, y)
        g = self._graphed(x, y) if self._use_graph else None
        if g is not None:
            metrics = dict(g(x, y))
            metrics["grad_norm"] = self.optimizer.grad_norm()
            stepped = True
        else:
            metrics, stepped = self._eager_step(x, y)
        if stepped and self.lr_scheduler is not None:
            self.lr_scheduler.step()
            if self._flat:
                self.optimizer.sync_lr_from_groups()
        if not want_metrics:
            return None
        return {k: float(v) for k, v in metrics.items()}

    def _eager_step(self, x, y):
        opt = self.optimizer
        opt.zero_grad()
        metrics = dict(_as_metrics(self.train_one_batch(x, y)))
        metrics["loss"].backward()
        if self._flat:
            if self._reducer is not None:
                self._reducer.all_reduce()
            if self.skip_grad_norm:  # the decision needs the norm on the host (reference :188)
                norm = opt.measure_grad_norm()
                metrics["grad_norm"] = norm
                if float(norm) > self.skip_grad_norm:
                    return metrics, False
            opt.step()
            metrics.setdefault("grad_norm", opt.grad_norm().clone())
            return metrics, True
        limit = self.clip_grad_norm or self.skip_grad_norm or 1e50
        norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), limit)
        metrics["grad_norm"] = norm
        if self.skip_grad_norm and float(norm) > self.skip_grad_norm:
            return metrics, False
        opt.step()
        return metrics, True

    @torch.no_grad()
    def _eval_one_batch(self, x, y):
        self.model.eval()
        x, y = self._to_device(x, y)
        return {k: float(v) for k, v in _as_metrics(self.eval_one_batch(x, y)).items()}

    @torch.no_grad()
    def sample_one_batch(self):
        self.model.eval()
        try:
            self._log.images("sample", self.model.sample(n_samples=16), self._step)
        except Exception as e:  # sampling is best effort, as in the reference (trainer.py:215-220)
            print(f"Failed to sample from the model: {e}")

    # ------------------------------------------------------------------ epochs
    def _current_lrs(self):
        if self._flat:
            return {"group_0": self.optimizer.current_lr()}
        return {f"group_{i}": g["lr"] for i, g in enumerate(self.optimizer.param_groups)}

    def _train_epoch(self):
        tick = time.time()
        for batch in self.train_loader:
            x, y = _split_batch(batch)
            self._examples_processed += x.shape[0]
            report = self._step % self._metrics_every == 0
            if report and self._log.active:
                self._log.group("metrics/lr", self._current_lrs(), self._step)
            metrics = self._train_one_batch(x, y, want_metrics=report)
            now = time.time()
            self._time_taken += now - tick
            tick = now
            if report:
                for key, value in metrics.items():
                    self._log.group(f"metrics/{key}", {"train": value}, self._step)
                rate = self._examples_processed / max(self._time_taken, 1e-9)
                self._log.scalar("speed/examples_per_sec", rate, self._step)
                self._log.scalar("speed/millis_per_example", 1000.0 / rate, self._step)
                self._log.scalar("speed/epoch", self._epoch, self._step)
                self._log.scalar("speed/step", self._step, self._step)
            self._step += 1

    def _eval_epoch(self):
        """Example-weighted means of every metric over the evaluation set."""
        seen, totals = 0, {}
        for batch in self.eval_loader:
            x, y = _split_batch(batch)
            n = x.shape[0]
            seen += n
            for key, value in self._eval_one_batch(x, y).items():
                totals[key] = totals.get(key, 0.0) + value * n
        means = {k: v / max(seen, 1) for k, v in totals.items()}
        for key, value in means.items():
            self._log.group(f"metrics/{key}", {"eval": value}, self._step)
        return means

    def interleaved_train_and_eval(self, max_epochs, restore=True):
        """Trains until `max_epochs` epochs are done (counting restored ones), evaluating,
        checkpointing and sampling after each epoch as configured."""
        if restore:
            try:
                self.restore_checkpoint()
            except FileNotFoundError:
                print(f"No checkpoint found in {self.log_dir}. Training from scratch.")
        while self._epoch < max_epochs:
            self._train_epoch()
            self.last_eval_metrics = self._eval_epoch()
            self._epoch += 1
            self._save_checkpoint()
            if self._epoch % self.sample_epochs == 0:
                self.sample_one_batch()
        self._log.close()
