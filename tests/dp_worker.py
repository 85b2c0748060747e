"""One rank of the 2-process data-parallel check (spawned by tests/test_gpu_dp.py).

Both ranks share the box's single MI355X (device 0) and talk over gloo; everything else is the
production path: optim.FlatAdam, parallel.FlatGradAllReduce (broadcast + one flat all-reduce),
graph.GraphedTrainStep in its two-graph data-parallel form. usage:
    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <mode> <out.pt> [model]
mode "same": every rank trains on the full batches; "shard": rank r on its half of each batch.
model "igpt" (default) or "gated": a GatedPixelCNN, whose last layer's `_vstack_1x1` / `_link` ... never
receive a gradient — their slice of the flat buffer must stay zero through the
all-reduce and leave the parameters untouched.

Also: `python tests/dp_worker.py rccl1 <out.pt>` — ONE process, no torch.distributed: a real RCCL
communicator of world size 1 (RCCL refuses two ranks on one device, so this is what a 1-GPU box can
run of the production transport): init, broadcast, all-reduce captured INSIDE the step's hipGraph,
replay, destroy; the result must be bit-identical to the step without a reducer.
"""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(dev, seed, kind="igpt"):
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import optim

    torch.manual_seed(seed)
    if kind == "gated":
        model = pg.models.GatedPixelCNN(in_channels=1, out_channels=1, n_gated=2, gated_channels=16,
                                        head_channels=8).to(dev)
    elif kind == "snail":
        # BASELINE.json configs[3] (north_star's DDP workload) at its own channel counts — 64 channels, key 4 / value 32 —
        # on 16x16x3, two blocks: the bf16x3 convolutions, the 66 / 69-channel fp32-MFMA projections and the k4 attention
        model = pg.models.PixelSNAIL(in_channels=3, out_channels=3, n_channels=64, n_pixel_snail_blocks=2,
                                     n_residual_blocks=2, attention_key_channels=4,
                                     attention_value_channels=32).to(dev)
    elif kind == "vd_vae":
        # BASELINE.json configs[4]: the bench's channel counts (64 hidden / 32 bottleneck / 16 latent) on 32x32x3
        model = pg.models.VeryDeepVAE(in_channels=3, out_channels=3, input_resolution=32,
                                      stack_configs=[(1, 2), (1, 2), (1, 1), (1, 1)], latent_channels=16,
                                      hidden_channels=64, bottleneck_channels=32).to(dev)
    else:
        model = pg.models.ImageGPT(1, 1, in_size=8, n_transformer_blocks=2, n_attention_heads=4,
                                   n_embedding_channels=16).to(dev)
        with torch.no_grad():
            model._pos.normal_(0, 0.1)
    model.train()
    return model, optim.FlatAdam(model.parameters(), lr=5e-3, lr_decay=0.999)


def batches(n_steps=3, b=8, kind="igpt"):
    g = torch.Generator().manual_seed(77)
    if kind == "snail":
        return [torch.randint(0, 256, (b, 3, 16, 16), generator=g).float() / 255 for _ in range(n_steps)]
    if kind == "vd_vae":
        return [torch.randint(0, 256, (b, 3, 32, 32), generator=g).float() / 255 for _ in range(n_steps)]
    return [torch.bernoulli(torch.full((b, 1, 8, 8), 0.3), generator=g) for _ in range(n_steps)]


class FixedNoise:
    """Noise source of the VAE family for captured steps (a replay only ever sees the tensors of capture time): the i-th
    draw of EVERY step is the same pre-drawn tensor, identical in every process that uses the same seed."""

    def __init__(self, seed=3):
        self.bank, self.i, self.gen = [], 0, torch.Generator().manual_seed(seed)

    def reset(self):
        self.i = 0

    def __call__(self, shape, device):
        if self.i == len(self.bank):
            self.bank.append(torch.randn(shape, generator=self.gen).to(device))
        eps = self.bank[self.i]
        assert tuple(eps.shape) == tuple(shape)
        self.i += 1
        return eps


def step_functions(kind, model):
    """(loss_fn, forward_fn) for graph.GraphedTrainStep: the VAE's ELBO with its noise replayed, BCE otherwise."""
    from pytorch_generative_amd import ops

    if kind != "vd_vae":
        return (lambda x, preds: ops.bce_with_logits_sum_mean(preds, x)), None
    from pytorch_generative_amd.models.vae import vaes

    noise = FixedNoise()
    vaes.set_noise_fn(noise)

    def fwd(x, y=None):
        noise.reset()
        preds = model(x)
        recon, klm = ops.elbo_terms(preds[0], x, preds[1])
        return recon + klm

    return None, fwd


def rccl_world1(out):
    from pytorch_generative_amd import _lib, graph, ops, parallel

    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    loss_fn = lambda x, preds: ops.bce_with_logits_sum_mean(preds, x)  # noqa: E731
    data = [b.to(dev) for b in batches()]
    lib = _lib.load()
    assert lib.pg_comm_rccl_version() > 0 and lib.pg_comm_world() == 0
    model, opt = build(dev, seed=0)
    red = parallel.FlatGradAllReduce(opt, transport="rccl")
    assert lib.pg_comm_world() == 1 and red.capturable and red.active
    red.broadcast_parameters(src=0)
    g0 = opt.flat_grad.clone().normal_()
    opt.flat_grad.copy_(g0)
    red.all_reduce()  # eager, on the current stream
    torch.cuda.synchronize()
    assert torch.equal(opt.flat_grad, g0), "a world of one must reduce to the identity"
    step = graph.GraphedTrainStep(model, opt, loss_fn, data[0], reducer=red, preserve_state=True)
    assert step.reduce and not step.split and step.graph_b is None  # the collective is INSIDE graph_a
    losses = [float(step(b)) for b in data]
    torch.cuda.synchronize()
    model2, opt2 = build(dev, seed=0)
    step2 = graph.GraphedTrainStep(model2, opt2, loss_fn, data[0], preserve_state=True)
    losses2 = [float(step2(b)) for b in data]
    torch.cuda.synchronize()
    same = torch.equal(opt.flat_param, opt2.flat_param)
    del step
    red.close()
    assert lib.pg_comm_world() == 0
    torch.save({"same": same, "losses": losses, "losses_plain": losses2,
                "rccl_version": lib.pg_comm_rccl_version()}, out)


def comm_failure_paths(out):
    """Error behaviour of the pg_comm_* entry points (include/pg_hip.h): argument errors -> ValueError, RCCL
    errors -> RuntimeError, and a failed init leaves NO communicator behind (the next init works)."""
    import ctypes

    import pytest

    from pytorch_generative_amd import _lib

    torch.cuda.set_device(0)
    lib = _lib.load()
    buf = torch.ones(16, device="cuda")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    with pytest.raises(ValueError, match="no communicator"):
        _lib.check(lib.pg_allreduce_sum(buf.data_ptr(), 16, _lib.DTYPE_F32, st), "pg_allreduce_sum")
    ident = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    _lib.check(lib.pg_comm_unique_id(ident), "pg_comm_unique_id")
    with pytest.raises(ValueError, match="bad rank"):
        _lib.check(lib.pg_comm_init(1, 1, ident.raw), "pg_comm_init")
    assert lib.pg_comm_world() == 0
    # an id no rank 0 ever issued: RCCL's bootstrap must refuse it (zeroed address family), not hang
    bad = bytes(_lib.COMM_ID_BYTES)
    bad_id_error = None
    try:
        _lib.check(lib.pg_comm_init(0, 1, bad), "pg_comm_init")
    except RuntimeError as e:
        bad_id_error = str(e)
    if bad_id_error is None:  # this RCCL accepted the id for a world of one: a communicator exists, drop it
        _lib.check(lib.pg_comm_destroy(), "pg_comm_destroy")
    assert lib.pg_comm_world() == 0, "a failed pg_comm_init left a communicator behind"
    _lib.check(lib.pg_comm_init(0, 1, ident.raw), "pg_comm_init")  # ... and the next good init works
    assert lib.pg_comm_world() == 1
    with pytest.raises(ValueError, match="already exists"):
        _lib.check(lib.pg_comm_init(0, 1, ident.raw), "pg_comm_init")
    with pytest.raises(ValueError, match="root"):
        _lib.check(lib.pg_broadcast(buf.data_ptr(), 16, _lib.DTYPE_F32, 3, st), "pg_broadcast")
    with pytest.raises(ValueError, match="dtype"):
        _lib.check(lib.pg_allreduce_sum(buf.data_ptr(), 16, 7, st), "pg_allreduce_sum")
    _lib.check(lib.pg_allreduce_sum(buf.data_ptr(), 16, _lib.DTYPE_F32, st), "pg_allreduce_sum")
    torch.cuda.synchronize()
    assert torch.equal(buf.cpu(), torch.ones(16))
    _lib.check(lib.pg_comm_destroy(), "pg_comm_destroy")
    _lib.check(lib.pg_comm_destroy(), "pg_comm_destroy")  # idempotent
    assert lib.pg_comm_world() == 0
    torch.save({"bad_id_error": bad_id_error}, out)


def main():
    mode, out = sys.argv[1], sys.argv[2]
    kind = sys.argv[3] if len(sys.argv) > 3 else "igpt"
    if mode == "rccl1":
        return rccl_world1(out)
    if mode == "commfail":
        return comm_failure_paths(out)
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_generative_amd import graph, ops, parallel

    if mode == "same":
        # identical batches on every rank: with the bit-reproducible kernels the averaged gradient IS the 1-GPU gradient
        # (x + x and the 1/2 pre-scale are exact), so the comparison needs no allowance for Adam amplifying the last-bit
        # run-to-run differences of the fused attention backwards' atomic dQ deposits
        ops_mod = __import__("pytorch_generative_amd.ops", fromlist=["ops"])
        ops_mod.set_deterministic(True)
    model, opt = build(dev, seed=rank, kind=kind)  # different seeds: the broadcast must make them equal
    red = parallel.FlatGradAllReduce(opt)
    red.broadcast_parameters(src=0)
    loss_fn, fwd = step_functions(kind, model)
    data = batches(kind=kind)
    if mode == "shard":
        per = data[0].shape[0] // world
        data = [b[rank * per:(rank + 1) * per] for b in data]
    step = graph.GraphedTrainStep(model, opt, loss_fn, data[0].to(dev), reducer=red, preserve_state=True, forward_fn=fwd)
    assert step.split and step.graph_b is not None
    losses = [float(step(b.to(dev))) for b in data]
    torch.cuda.synchronize()
    mine = opt.flat_param.detach().cpu()
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    assert all(torch.equal(gathered[0], t) for t in gathered), "ranks diverged"
    all_losses = [None] * world
    dist.all_gather_object(all_losses, losses)
    if rank == 0:
        torch.save({"flat_param": mine, "losses": all_losses, "lr": opt.current_lr(),
                    "params": {k: v.detach().cpu() for k, v in model.named_parameters()},
                    "step": float(opt.state_block[0])}, out)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
