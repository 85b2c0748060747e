"""GPU: the callers either side of the hot path — Trainer (reference trainer.py:15-287 semantics),
checkpoint round trip in the reference's on-disk layout, and AutoregressiveModel.sample()
(reference models/tests.py:33-95 style smoke + behavioural checks)."""

import os

import pytest
import torch

import _util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


class _Loader:
    """A few fixed batches per epoch (the reference's DummyLoader idea, models/tests.py:12-27)."""

    def __init__(self, shape, n_batches=3, seed=0):
        g = torch.Generator().manual_seed(seed)
        self.batches = [torch.bernoulli(torch.full(shape, 0.3), generator=g) for _ in range(n_batches)]

    def __iter__(self):
        return iter(self.batches)


def _loss_fn(x, _, preds):
    from pytorch_generative_amd import ops

    return ops.bce_with_logits_sum_mean(preds, x)


def _make(dev, lr=1e-3):
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import optim

    torch.manual_seed(0)
    model = pg.models.PixelCNN(1, 1, n_residual=2, residual_channels=4, head_channels=4).to(dev)
    return model, optim.FlatAdam(model.parameters(), lr=lr)


def test_trainer_trains_checkpoints_and_restores(dev, tmp_path):
    from pytorch_generative_amd import trainer

    loader = _Loader((4, 1, 8, 8))
    model, opt = _make(dev)
    sched = torch.optim.lr_scheduler.MultiplicativeLR(opt, lr_lambda=lambda _: 0.9)
    t = trainer.Trainer(model, _loss_fn, opt, loader, loader, lr_scheduler=sched,
                        log_dir=str(tmp_path), n_gpus=1, sample_epochs=100)
    t.interleaved_train_and_eval(max_epochs=2)
    assert t._step == 6 and t._epoch == 2 and t._examples_processed == 24
    ckpt = torch.load(os.path.join(tmp_path, "trainer_state_2.ckpt"), map_location="cpu", weights_only=False)
    # the reference's checkpoint layout (trainer.py:98-112)
    assert set(ckpt) == {"model", "optimizer", "step", "epoch", "examples_processed", "time_taken", "lr_scheduler"}
    assert set(ckpt["optimizer"]) == {"state", "param_groups"}
    st0 = ckpt["optimizer"]["state"][0]
    assert set(st0) == {"step", "exp_avg", "exp_avg_sq"} and float(st0["step"]) == 6.0
    assert abs(ckpt["optimizer"]["param_groups"][0]["lr"] - 1e-3 * 0.9**6) < 1e-9
    assert "_c" in ckpt["model"] and "_input.mask" in ckpt["model"]
    params_after = {k: v.clone() for k, v in model.state_dict().items()}

    # a fresh trainer restores everything and continues
    model2, opt2 = _make(dev)
    sched2 = torch.optim.lr_scheduler.MultiplicativeLR(opt2, lr_lambda=lambda _: 0.9)
    t2 = trainer.Trainer(model2, _loss_fn, opt2, loader, loader, lr_scheduler=sched2,
                         log_dir=str(tmp_path), n_gpus=1, sample_epochs=100)
    t2.restore_checkpoint()
    assert t2._step == 6 and t2._epoch == 2
    for k, v in model2.state_dict().items():
        assert torch.equal(v.cpu(), params_after[k].cpu()), k
    assert abs(float(opt2.state_block[1]) - 1e-3 * 0.9**6) < 1e-9
    assert float(opt2.state_block[0]) == 6.0
    _util.assert_close(opt2.exp_avg, opt.exp_avg, 1e-7, "exp_avg")
    m = t2._train_one_batch(loader.batches[0], None)
    assert set(m) == {"loss", "grad_norm"} and all(isinstance(v, float) for v in m.values())


def test_trainer_loads_reference_style_checkpoint(dev, tmp_path):
    """A checkpoint written by the reference (torch.optim.Adam state_dict keyed by parameter
    index, model state_dict with masks and lazy _c/_h/_w buffers) restores into this Trainer."""
    from pytorch_generative_amd import trainer

    model, opt = _make(dev)
    loader = _Loader((2, 1, 8, 8), n_batches=1)
    cpu_sd = {k: torch.randn_like(v.cpu()) * 0.1 if v.dtype.is_floating_point and not k.endswith("mask")
              else v.cpu().clone() for k, v in model.state_dict().items()}
    cpu_sd.update({"_c": torch.tensor(1), "_h": torch.tensor(8), "_w": torch.tensor(8)})
    names = [k for k, _ in model.named_parameters()]
    adam_state = {i: {"step": torch.tensor(7.0), "exp_avg": torch.randn_like(cpu_sd[k]) * 1e-3,
                      "exp_avg_sq": torch.rand_like(cpu_sd[k]) * 1e-6} for i, k in enumerate(names)}
    ref_ckpt = {
        "model": cpu_sd,
        "optimizer": {"state": adam_state,
                      "param_groups": [{"lr": 5e-4, "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0,
                                        "amsgrad": False, "params": list(range(len(names)))}]},
        "step": 70, "epoch": 7, "examples_processed": 140, "time_taken": 1.5,
    }
    torch.save(ref_ckpt, os.path.join(tmp_path, "trainer_state_7.ckpt"))
    t = trainer.Trainer(model, _loss_fn, opt, loader, loader, log_dir=str(tmp_path), n_gpus=1)
    t.restore_checkpoint()
    assert (t._step, t._epoch) == (70, 7) and float(opt.state_block[0]) == 7.0
    assert abs(float(opt.state_block[1]) - 5e-4) < 1e-9  # lr lives in an fp32 state block
    for i, (k, p) in enumerate(model.named_parameters()):
        assert torch.equal(p.detach().cpu(), cpu_sd[k]), k
        o = opt._offsets[i]
        assert torch.equal(opt.exp_avg[o:o + p.numel()].cpu().view(p.shape), adam_state[i]["exp_avg"])
    assert int(model._h) == 8
    t._train_one_batch(loader.batches[0], None)  # parameters are still views of the flat buffer
    assert all(p.data_ptr() >= opt.flat_param.data_ptr() for p in model.parameters())


@pytest.mark.parametrize("ctor,kw", [
    ("ImageGPT", dict(in_channels=3, out_channels=3, in_size=5, n_transformer_blocks=1,
                      n_attention_heads=2, n_embedding_channels=4)),
    ("PixelSNAIL", dict(in_channels=3, out_channels=3, n_channels=8, n_pixel_snail_blocks=1,
                        n_residual_blocks=1, attention_value_channels=4, attention_key_channels=2)),
    ("GatedPixelCNN", dict(in_channels=3, out_channels=3, n_gated=1, gated_channels=4, head_channels=4)),
])
def test_conditional_sampling_keeps_given_pixels(dev, ctor, kw):
    """Reference models/tests.py:91-95: entries of `conditioned_on` that are >= 0 are untouched."""
    import pytorch_generative_amd as pg

    torch.manual_seed(0)
    model = getattr(pg.models, ctor)(**kw).to(dev)
    model(torch.rand(2, 3, 5, 5, device=dev))  # records (C, H, W)
    out = model.sample(n_samples=2)
    assert out.shape == (2, 3, 5, 5) and bool(((out == 0) | (out == 1)).all())
    cond = torch.full((2, 3, 5, 5), -1.0, device=dev)
    cond[:, :, :2, :] = 0.25
    out = model.sample(conditioned_on=cond)
    assert torch.equal(out[:, :, :2, :], cond[:, :, :2, :]) and bool((out[:, :, 2:, :] >= 0).all())


def test_trainer_restores_a_checkpoint_written_by_the_reference_trainer(dev, tmp_path):
    """tests/golden/ref_trainer/trainer_state_2.ckpt was written by the reference's own Trainer
    (make_ckpt_golden.py). Restoring it here must reproduce the counters, parameters, Adam moments
    and the scheduled lr — and the NEXT training batch must give the reference's metrics and
    parameters (the fixture records what the reference did next)."""
    import shutil

    import pytorch_generative_amd as pg
    from pytorch_generative_amd import optim, trainer

    src = os.path.join(_util.GOLDEN_DIR, "ref_trainer")
    shutil.copy(os.path.join(src, "trainer_state_2.ckpt"), tmp_path)
    nxt = torch.load(os.path.join(src, "next_step.pt"), map_location="cpu", weights_only=False)
    ref = torch.load(os.path.join(src, "trainer_state_2.ckpt"), map_location="cpu", weights_only=False)

    model = pg.models.PixelCNN(**nxt["model_kwargs"]).to(dev)
    opt = optim.FlatAdam(model.parameters(), lr=nxt["lr"])
    sched = torch.optim.lr_scheduler.MultiplicativeLR(opt, lr_lambda=lambda _: nxt["decay"])
    loader = [(b, None) for b in nxt["train_batches"]]
    t = trainer.Trainer(model, _loss_fn, opt, loader, loader, lr_scheduler=sched, log_dir=str(tmp_path),
                        n_gpus=1)
    t.restore_checkpoint()
    assert (t._step, t._epoch, t._examples_processed) == (ref["step"], ref["epoch"], ref["examples_processed"])
    for k, v in model.state_dict().items():
        assert torch.equal(v.cpu(), ref["model"][k]), k
    assert abs(opt.current_lr() - ref["optimizer"]["param_groups"][0]["lr"]) < 1e-10
    for i, p in enumerate(model.parameters()):
        o = opt._offsets[i]
        assert torch.equal(opt.exp_avg[o:o + p.numel()].view(p.shape).cpu(), ref["optimizer"]["state"][i]["exp_avg"])
    assert float(opt.state_block[0]) == float(ref["optimizer"]["state"][0]["step"])
    # ... and it continues exactly like the reference did
    m = t._train_one_batch(nxt["next_x"], None)
    assert abs(m["loss"] - nxt["next_metrics"]["loss"]) <= 1e-4 * abs(nxt["next_metrics"]["loss"])
    assert abs(m["grad_norm"] - nxt["next_metrics"]["grad_norm"]) <= 1e-4 * nxt["next_metrics"]["grad_norm"]
    for k, v in model.state_dict().items():
        _util.assert_close(v, nxt["state_after_next"][k], 1e-4, f"{k} after the next step")
    assert abs(opt.current_lr() - nxt["lr_after_next"]) < 1e-10


def test_flat_adam_state_dict_round_trips_through_torch_adam(dev):
    """FlatAdam.state_dict() is a complete torch.optim.Adam state_dict: it loads into torch's Adam,
    which can step, and torch's state loads back."""
    from pytorch_generative_amd import ops

    model, opt = _make(dev)
    x = _Loader((4, 1, 8, 8)).batches[0].to(dev)
    for _ in range(2):
        opt.zero_grad()
        ops.bce_with_logits_sum_mean(model(x), x).backward()
        opt.step()
    sd = opt.state_dict()
    clones = [p.detach().clone().requires_grad_(True) for p in model.parameters()]
    adam = torch.optim.Adam(clones, lr=1e-3)
    adam.load_state_dict(sd)
    for c, p in zip(clones, model.parameters()):
        c.grad = p.grad.clone()
    adam.step()  # raises KeyError if a param_group key is missing
    opt.step()
    for c, p in zip(clones, model.parameters()):
        _util.assert_close(p, c, 1e-5, "FlatAdam vs torch.optim.Adam continuing from the same state")
    model2, opt2 = _make(dev)
    opt2.load_state_dict(adam.state_dict())
    assert float(opt2.state_block[0]) == 3.0
    bad = adam.state_dict()
    bad["param_groups"][0]["weight_decay"] = 0.1
    with pytest.raises(ValueError, match="weight_decay"):
        opt2.load_state_dict(bad)


@pytest.mark.parametrize("ctor,kw,hw", [
    ("PixelCNN", dict(in_channels=1, out_channels=1, n_residual=3, residual_channels=8, head_channels=8), 12),
    ("GatedPixelCNN", dict(in_channels=3, out_channels=3, n_gated=2, gated_channels=8, head_channels=8), 8),
    ("PixelSNAIL", dict(in_channels=3, out_channels=3, n_channels=16, n_pixel_snail_blocks=2,
                        n_residual_blocks=2, attention_key_channels=4, attention_value_channels=16), 8),
])
def test_row_cached_sampling_equals_full_forward(dev, ctor, kw, hw):
    """Teacher forcing (the check ImageGPT's incremental sampler has): feed a known image pixel by
    pixel through the row-cached sampler — canvas entries not yet visited are still -1, the 'draw' is
    the teacher's pixel — and every logit the sampler produced must equal the full forward's."""
    import pytorch_generative_amd as pg

    torch.manual_seed(0)
    model = getattr(pg.models, ctor)(**kw).to(dev)
    c = kw["in_channels"]
    x = torch.bernoulli(torch.full((3, c, hw, hw), 0.4)).to(dev)
    full = model(x).detach()
    pos = iter([(r, q) for r in range(hw) for q in range(hw)])

    def teacher(_logits):
        r, q = next(pos)
        return x[:, :, r, q]

    model._sample_fn = teacher
    canvas, inc = model.sample(conditioned_on=torch.full_like(x, -1.0), return_logits=True)
    assert torch.equal(canvas, x)
    _util.assert_close(inc, full, 1e-5, f"{ctor}: row-cached logits vs full forward")
    # the reference procedure (a full forward per pixel) is still available and agrees
    pos = iter([(r, q) for r in range(hw) for q in range(hw)])
    canvas2, ref = model.sample(conditioned_on=torch.full_like(x, -1.0), incremental=False, return_logits=True)
    assert torch.equal(canvas2, x)
    _util.assert_close(ref, full, 1e-5, f"{ctor}: per-pixel full forwards vs one full forward")
