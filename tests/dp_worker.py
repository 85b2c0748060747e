"""One rank of the 2-process data-parallel check (spawned by tests/test_gpu_dp.py).

Both ranks share the box's single MI355X (device 0) and talk over gloo; everything else is the
production path: optim.FlatAdam, parallel.FlatGradAllReduce (broadcast + one flat all-reduce),
graph.GraphedTrainStep in its two-graph data-parallel form. usage:
    RANK=r WORLD_SIZE=2 MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/dp_worker.py <mode> <out.pt>
mode "same": every rank trains on the full batches; "shard": rank r on its half of each batch.
"""

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def build(dev, seed):
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import optim

    torch.manual_seed(seed)
    model = pg.models.ImageGPT(1, 1, in_size=8, n_transformer_blocks=2, n_attention_heads=4,
                               n_embedding_channels=16).to(dev)
    with torch.no_grad():
        model._pos.normal_(0, 0.1)
    model.train()
    return model, optim.FlatAdam(model.parameters(), lr=5e-3, lr_decay=0.999)


def batches(n_steps=3, b=8):
    g = torch.Generator().manual_seed(77)
    return [torch.bernoulli(torch.full((b, 1, 8, 8), 0.3), generator=g) for _ in range(n_steps)]


def main():
    mode, out = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from pytorch_generative_amd import graph, ops, parallel

    model, opt = build(dev, seed=rank)  # different seeds: the broadcast must make them equal
    red = parallel.FlatGradAllReduce(opt)
    red.broadcast_parameters(src=0)
    loss_fn = lambda x, preds: ops.bce_with_logits_sum_mean(preds, x)  # noqa: E731
    data = batches()
    if mode == "shard":
        per = data[0].shape[0] // world
        data = [b[rank * per:(rank + 1) * per] for b in data]
    step = graph.GraphedTrainStep(model, opt, loss_fn, data[0].to(dev), reducer=red, preserve_state=True)
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
