"""Writes tests/golden/ref_trainer/ — a checkpoint produced by the REFERENCE's own Trainer
(pytorch_generative/trainer.py, run here on the CPU with a no-op SummaryWriter stub, SURVEY.md
§8c), plus what that Trainer does next: the metrics and the parameters after ONE more training
batch from the restored state. Build container only:

    python tests/golden/make_ckpt_golden.py

The GPU test restores `trainer_state_2.ckpt` into pytorch_generative_amd.trainer.Trainer and must
(a) land on the same counters / parameters / Adam moments / lr and (b) take the same next step.
"""

import os
import shutil
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
OUT = os.path.join(HERE, "ref_trainer")

import _ref  # noqa: E402


def _stub_tensorboard():
    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        add_scalars = add_images = add_scalar

        def close(self):
            pass

    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    import torch.utils

    torch.utils.tensorboard = tb


def batches(seed, n, shape=(4, 1, 8, 8)):
    g = torch.Generator().manual_seed(seed)
    return [(torch.bernoulli(torch.full(shape, 0.3), generator=g), torch.zeros(shape[0], dtype=torch.long))
            for _ in range(n)]


MODEL_KW = dict(in_channels=1, out_channels=1, n_residual=2, residual_channels=4, head_channels=4)
LR, DECAY = 1e-3, 0.9


def main():
    _stub_tensorboard()
    ref = _ref.load()
    import importlib

    rtrainer = importlib.import_module("pytorch_generative.trainer")
    import torch.nn.functional as F

    def loss_fn(x, _, preds):
        n = x.shape[0]
        loss = F.binary_cross_entropy_with_logits(preds.view(n, -1), x.view(n, -1), reduction="none")
        return loss.sum(dim=1).mean()

    shutil.rmtree(OUT, ignore_errors=True)
    os.makedirs(OUT)
    torch.manual_seed(0)
    model = ref.models.PixelCNN(**MODEL_KW)
    opt = torch.optim.Adam(model.parameters(), lr=LR)
    sched = torch.optim.lr_scheduler.MultiplicativeLR(opt, lr_lambda=lambda _: DECAY)
    data = batches(0, 3)
    t = rtrainer.Trainer(model, loss_fn, opt, data, data, lr_scheduler=sched, log_dir=OUT, n_gpus=0,
                         device_id=0)  # device_id=0: the reference only checkpoints when device_id == 0
    t.interleaved_train_and_eval(2)  # writes trainer_state_1.ckpt, trainer_state_2.ckpt
    os.remove(os.path.join(OUT, "trainer_state_1.ckpt"))
    for f in os.listdir(OUT):  # tensorboard stub writes nothing; keep only the checkpoint
        if not f.endswith(".ckpt"):
            os.remove(os.path.join(OUT, f))
    # what the reference does next, from the state it just saved
    nxt = batches(99, 1)[0]
    metrics = t._train_one_batch(*nxt)
    torch.save({
        "model_kwargs": MODEL_KW, "lr": LR, "decay": DECAY, "next_x": nxt[0], "next_metrics": metrics,
        "state_after_next": {k: v.detach().clone() for k, v in model.state_dict().items()},
        "lr_after_next": opt.param_groups[0]["lr"], "train_batches": [b[0] for b in data],
    }, os.path.join(OUT, "next_step.pt"))
    print("wrote", sorted(os.listdir(OUT)), metrics)


if __name__ == "__main__":
    main()
