"""Training loop for the MI355X operator path.

Keeps the reference Trainer's constructor, step semantics and checkpoint layout
(pytorch_generative/trainer.py:15-287): per batch `zero_grad -> model(x) -> loss_fn(x, y, preds)
-> backward -> global grad-norm (clip / skip) -> optimizer.step -> lr_scheduler.step`, metrics
returned as Python floats, `trainer_state_{epoch}.ckpt` holding model / optimizer / step / epoch /
examples_processed / time_taken (/ lr_scheduler), written by device 0 only.

Differences that make it MI355X-native:
  * `optimizer` may be a `pytorch_generative_amd.optim.FlatAdam`; the grad norm, clipping and the
    Adam update then run as one fused device-side chain (no per-tensor kernels, no host sync
    besides the final metric read), and gradients are accumulated by the backward kernels
    directly into its flat buffer.
  * data parallelism is one process per GPU with ONE flat RCCL all-reduce per step
    (parallel.FlatGradAllReduce) instead of DistributedDataParallel's bucket hooks.
A stock torch optimizer still works (then `torch.nn.utils.clip_grad_norm_` is used).
"""

import collections
import glob
import os
import re
import tempfile
import time

import torch
import torch.distributed as dist

from pytorch_generative_amd import optim as pg_optim
from pytorch_generative_amd import parallel


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    add_scalars = add_images = add_scalar

    def close(self):
        pass


def _make_writer(log_dir, **kwargs):
    try:
        from torch.utils import tensorboard

        return tensorboard.SummaryWriter(log_dir, max_queue=100, **kwargs)
    except Exception:  # tensorboard not installed: metrics are still returned to the caller
        return _NullWriter()


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
    ):
        self.loss_fn = loss_fn
        self.train_loader = train_loader
        self.eval_loader = eval_loader
        self.clip_grad_norm = clip_grad_norm
        self.skip_grad_norm = skip_grad_norm
        self.log_dir = log_dir or tempfile.mkdtemp()
        self.save_checkpoint_epochs = save_checkpoint_epochs
        self.sample_epochs = sample_epochs

        if n_gpus < 1:
            raise RuntimeError(
                "pytorch_generative_amd.Trainer runs on MI355X GPUs only (n_gpus >= 1); the HIP "
                "operator path has no CPU fallback"
            )
        self.device_id = 0 if device_id is None and n_gpus == 1 else device_id
        if n_gpus > 1:
            assert device_id is not None, "'device_id' must be provided if n_gpus > 1."
        self.device = torch.device("cuda", self.device_id if n_gpus > 1 else torch.cuda.current_device())
        if next(model.parameters()).device != self.device:
            if isinstance(optimizer, pg_optim.FlatAdam):
                raise RuntimeError("move the model to the GPU before building FlatAdam")
            model = model.to(self.device)

        self.model = model
        self.optimizer = optimizer
        self.lr_scheduler = lr_scheduler
        self._flat = isinstance(optimizer, pg_optim.FlatAdam)
        self._reducer = None
        if n_gpus > 1:
            if not self._flat:
                raise RuntimeError("multi-GPU training requires pytorch_generative_amd.optim.FlatAdam")
            if not dist.is_initialized():
                raise RuntimeError("torch.distributed must be initialised (backend 'nccl' = RCCL)")
            self._reducer = parallel.FlatGradAllReduce(optimizer)
            self._reducer.broadcast_parameters(src=0)
        if self._flat:
            max_norm = clip_grad_norm or skip_grad_norm
            optimizer.state_block[pg_optim._MAXNORM] = float("inf") if max_norm is None else float(max_norm)

        self._step = 0
        self._epoch = 0
        self._examples_processed = 0
        self._time_taken = 0
        self._summary_writer = _make_writer(self.log_dir)

    # ------------------------------------------------------------------ checkpoints
    def _path(self, file_name):
        return os.path.join(self.log_dir, file_name)

    def _save_checkpoint(self):
        if self.device_id != 0 or self._epoch % self.save_checkpoint_epochs != 0:
            return
        checkpoint = {
            "model": self.model.state_dict(),
            "optimizer": self.optimizer.state_dict(),
            "step": self._step,
            "epoch": self._epoch,
            "examples_processed": self._examples_processed,
            "time_taken": self._time_taken,
        }
        if self.lr_scheduler is not None:
            checkpoint["lr_scheduler"] = self.lr_scheduler.state_dict()
        torch.save(checkpoint, self._path(f"trainer_state_{self._epoch}.ckpt"))

    def _find_latest_epoch(self):
        files = glob.glob(self._path("trainer_state_[0-9]*.ckpt"))
        epochs = sorted(int(re.findall(r"trainer_state_(\d+)\.ckpt", f)[0]) for f in files)
        if not epochs:
            raise FileNotFoundError(f"No checkpoints found in {self.log_dir}.")
        print(f"Found {len(epochs)} saved checkpoints.")
        return epochs[-1]

    def restore_checkpoint(self, epoch=None):
        epoch = epoch or self._find_latest_epoch()
        name = f"trainer_state_{epoch}.ckpt"
        print(f"Restoring trainer state from checkpoint {name}.")
        checkpoint = torch.load(self._path(name), map_location=self.device, weights_only=False)
        self.model.load_state_dict(checkpoint["model"])
        self.optimizer.load_state_dict(checkpoint["optimizer"])
        self._step = checkpoint["step"]
        self._epoch = checkpoint["epoch"]
        self._examples_processed = checkpoint["examples_processed"]
        self._time_taken = checkpoint["time_taken"]
        if self.lr_scheduler is not None:
            self.lr_scheduler.load_state_dict(checkpoint["lr_scheduler"])
        self._summary_writer.close()
        self._summary_writer = _make_writer(self.log_dir, purge_step=self._step)

    # ------------------------------------------------------------------ one batch
    def _get_metrics_dict(self, loss_or_metrics):
        metrics = loss_or_metrics
        if not isinstance(metrics, dict):
            metrics = {"loss": metrics}
        assert "loss" in metrics, 'Metrics dictionary does not contain "loss" key.'
        return metrics

    def _log_metrics(self, metrics, training):
        for key, metric in metrics.items():
            self._summary_writer.add_scalars(
                f"metrics/{key}", {"train" if training else "eval": metric}, self._step
            )

    def train_one_batch(self, x, y):
        """Override for custom training steps."""
        preds = self.model(x)
        return self.loss_fn(x, y, preds)

    def _train_one_batch(self, x, y):
        self.model.train()
        x = x.to(self.device, non_blocking=True)
        if y is not None:
            y = y.to(self.device, non_blocking=True)
        self.optimizer.zero_grad()
        metrics = self._get_metrics_dict(self.train_one_batch(x, y))
        metrics["loss"].backward()

        if self._flat:
            if self._reducer is not None:
                self._reducer.all_reduce()
            if self.skip_grad_norm:
                # the decision needs the norm on the host, as in the reference (trainer.py:188)
                norm = self._flat_grad_norm()
                metrics["grad_norm"] = norm
                if norm.item() <= self.skip_grad_norm:
                    self._optimizer_step()
            else:
                self._optimizer_step()
                metrics["grad_norm"] = self.optimizer.grad_norm().clone()
        else:
            max_norm = self.clip_grad_norm or self.skip_grad_norm or 1e50
            norm = torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm)
            metrics["grad_norm"] = norm
            if not self.skip_grad_norm or norm.item() <= self.skip_grad_norm:
                self._optimizer_step()
        return {k: v.item() for k, v in metrics.items()}

    def _flat_grad_norm(self):
        scale = float(self.optimizer.state_block[pg_optim._PRESCALE].item())
        return self.optimizer.flat_grad.norm() * scale

    def _optimizer_step(self):
        self.optimizer.step()
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
            if self._flat:
                self.optimizer.sync_lr_from_groups()

    def eval_one_batch(self, x, y):
        preds = self.model(x)
        return self.loss_fn(x, y, preds)

    @torch.no_grad()
    def _eval_one_batch(self, x, y):
        self.model.eval()
        x = x.to(self.device)
        if y is not None:
            y = y.to(self.device)
        metrics = self._get_metrics_dict(self.eval_one_batch(x, y))
        return {k: v.item() for k, v in metrics.items()}

    @torch.no_grad()
    def sample_one_batch(self):
        self.model.eval()
        try:
            tensor = self.model.sample(n_samples=16)
            self._summary_writer.add_images("sample", tensor, self._step)
        except Exception as e:  # same policy as the reference (trainer.py:215-220)
            print(f"Failed to sample from the model: {e}")

    # ------------------------------------------------------------------ epochs
    def interleaved_train_and_eval(self, max_epochs, restore=True):
        if restore:
            try:
                self.restore_checkpoint()
            except FileNotFoundError:
                print(f"No checkpoint found in {self.log_dir}. Training from scratch.")

        for _ in range(max_epochs - self._epoch):
            start_time = time.time()
            for batch in self.train_loader:
                x, y = batch if isinstance(batch, (tuple, list)) else (batch, None)
                self._examples_processed += x.shape[0]
                lrs = {f"group_{i}": g["lr"] for i, g in enumerate(self.optimizer.param_groups)}
                self._summary_writer.add_scalars("metrics/lr", lrs, self._step)
                metrics = self._train_one_batch(x, y)
                self._log_metrics(metrics, training=True)

                self._time_taken += time.time() - start_time
                start_time = time.time()
                self._summary_writer.add_scalar(
                    "speed/examples_per_sec", self._examples_processed / self._time_taken, self._step
                )
                self._summary_writer.add_scalar(
                    "speed/millis_per_example",
                    self._time_taken / self._examples_processed * 1000,
                    self._step,
                )
                self._summary_writer.add_scalar("speed/epoch", self._epoch, self._step)
                self._summary_writer.add_scalar("speed/step", self._step, self._step)
                self._step += 1

            n_examples, sums = 0, collections.defaultdict(float)
            for batch in self.eval_loader:
                x, y = batch if isinstance(batch, (tuple, list)) else (batch, None)
                n_examples += x.shape[0]
                for key, metric in self._eval_one_batch(x, y).items():
                    sums[key] += metric * x.shape[0]
            self._log_metrics({k: v / n_examples for k, v in sums.items()}, training=False)

            self._epoch += 1
            self._save_checkpoint()
            if self._epoch % self.sample_epochs == 0:
                self.sample_one_batch()

        self._summary_writer.close()
