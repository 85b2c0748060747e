"""bench.py — training throughput of the hot path on MI355X (driver contract: see README/DESIGN).

    python bench.py --gpus 1 --steps K --warmup W          # single GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the reference's Trainer._train_one_batch (trainer.py:173-193): zero_grad, forward,
BCE-with-logits loss, backward, global grad norm, Adam, lr decay — on one resident batch of
synthetic binarised-MNIST-shaped images, replayed from a hipGraph. Workload: BASELINE.json
configs[1], ImageGPT 8 blocks / 4 heads / 16 embedding channels on 28x28x1 (fp32: the reference
is fp32 end to end and parity is gated at 1e-4; see DESIGN.md for the bf16 note).
Rank 0 prints ONE JSON line.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "pytorch-generative_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MODEL_KW = dict(in_channels=1, out_channels=1, in_size=28, n_transformer_blocks=8,
                n_attention_heads=4, n_embedding_channels=16)
LR, LR_DECAY = 5e-3, 0.999977  # reference reproduce(): image_gpt.py:155-156
# secondary workloads (same step definition; selected with --model, reported under the same
# contract but NOT the default bench line): BASELINE.json configs[0], [2], [3]
OTHER_MODELS = {
    "pixel_snail": ("PixelSNAIL", dict(in_channels=3, out_channels=3, n_channels=64,
                                       n_pixel_snail_blocks=8, n_residual_blocks=2,
                                       attention_key_channels=4, attention_value_channels=32),
                    (3, 32, 32), 1e-3, 0.999977, 7.97e9, 110.5e6),
    "gated_pixel_cnn": ("GatedPixelCNN", dict(in_channels=3, out_channels=3, n_gated=10,
                                              gated_channels=128, head_channels=32),
                        (3, 32, 32), 1e-3, 0.9999, 21.23e9, 266e6),
    "pixel_cnn": ("PixelCNN", dict(in_channels=1, out_channels=1, n_residual=15,
                                   residual_channels=32, head_channels=32),
                  (1, 28, 28), 1e-3, 0.999977, 0.964e9, 31.6e6),
    # BASELINE.json configs[4]: VAE conv stacks + KL on 64x64x3 (ELBO loss, vae.py:149-159)
    "beta_vae": ("BetaVAE", dict(in_channels=3, out_channels=3, beta=4.0, latent_channels=16,
                                 strides=[2, 2, 2, 2], hidden_channels=64, residual_channels=32),
                 (3, 64, 64), 1e-3, 1.0, 1.57e9, 17.6e6),
    "vd_vae": ("VeryDeepVAE", dict(in_channels=3, out_channels=3, input_resolution=64,
                                   stack_configs=[(3, 5), (3, 5), (2, 4), (2, 3), (2, 2), (1, 1)],
                                   latent_channels=16, hidden_channels=64, bottleneck_channels=32),
               (3, 64, 64), 5e-4, 1.0, 10.96e9, 354e6),
}
HEADS, DK, DV, L = 4, 4, 4, 784
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix == vector peak (no TF32 on gfx950)


def synthetic_batch(batch, rank, chw=(1, 28, 28)):
    g = torch.Generator().manual_seed(1234 + rank)
    if chw[0] == 1:   # dynamically binarised MNIST-shaped (datasets.py:16-17)
        return torch.bernoulli(torch.full((batch, *chw), 0.1307), generator=g)
    return torch.randint(0, 256, (batch, *chw), generator=g).float() / 255  # CIFAR-shaped


def attention_kernel_roofline(batch, device, iters=10):
    """Times the three causal-attention kernels live with HIP events on the stream they are
    launched on, through the C-ABI, at the bench's exact shapes and on random data. The dominant
    kernel of the step is attn_dkv_m44_kernel (pg_causal_attn_bwd_dkv; matrix-core path, d_k = d_v = 4).
    Algorithmic FLOPs: pairs = N*heads*L*(L+1)/2 allowed (query, key) pairs;
      fwd   2*dk + 2*dv          (QK^T, PV)
      dQ    2*dk + 2*dv + 2*dk   (QK^T recompute, dP = dO V^T, dQ = dS K)
      dK/dV 2*dk + 2*dv + 2*dv + 2*dk (QK^T recompute, dP, dV = P^T dO, dK = dS^T Q)."""
    from pytorch_generative_amd import _lib

    lib = _lib.load()
    e = HEADS * DK
    g = torch.Generator().manual_seed(7)
    mk = lambda c: torch.randn(batch, c, 28, 28, generator=g).to(device)  # noqa: E731
    q, kv, d_o = mk(e), mk(2 * e), mk(e)
    o, dq, dkv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(kv)
    lse = torch.empty(batch, HEADS, L, device=device)
    delta = torch.empty_like(lse)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    kvs = 2 * e * L

    def fwd():
        _lib.check(lib.pg_causal_attn_fwd(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * e * L,
                                          o.data_ptr(), lse.data_ptr(), batch, HEADS, L, DK, DV,
                                          e * L, kvs, kvs, e * L, 0, st), "fwd")

    def bwd(fn):
        _lib.check(fn(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * e * L, o.data_ptr(),
                      d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                      dkv.data_ptr(), dkv.data_ptr() + 4 * e * L, batch, HEADS, L, DK, DV, e * L,
                      kvs, kvs, e * L, e * L, e * L, kvs, kvs, 0, st), "bwd")

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(iters)]
        for a, b in evs:
            a.record(stream)
            fn()
            b.record(stream)
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / iters

    t_fwd = timed(fwd)
    t_dq = timed(lambda: bwd(lib.pg_causal_attn_bwd_dq))
    t_dkv = timed(lambda: bwd(lib.pg_causal_attn_bwd_dkv))
    pairs = batch * HEADS * L * (L + 1) / 2
    fl = {"fwd": pairs * (2 * DK + 2 * DV), "dq": pairs * (4 * DK + 2 * DV),
          "dkv": pairs * (4 * DK + 4 * DV)}
    ms = {"fwd": t_fwd, "dq": t_dq, "dkv": t_dkv}
    return {k: {"launch_ms": ms[k], "flop_per_launch": fl[k], "tflops": fl[k] / ms[k] / 1e9}
            for k in ms}


def measured_traffic(batch):
    """HBM bytes per launch of the dominant kernel from the committed PMC profile (collected in
    separate rocprofv3 --pmc passes, see profiles/README.md); None when no profile matches."""
    path = os.path.join(ROOT, "profiles", "r01_traffic.json")
    try:
        with open(path) as f:
            t = json.load(f)
        if t.get("per_gpu_batch") == batch:
            return t.get("attn_bwd_dkv_bytes_per_launch")
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(batch=16, steps=2):
    """The oracle's restatement of the same training step on the host cores (kind 'port': the
    reference is Python and cannot travel to the GPU box; the oracle dispatches the same torch
    CPU primitives). Bounded sample: 1 warm-up + `steps` timed steps at a reduced batch."""
    from oracle import models as omodels
    from oracle import train as otrain

    import pytorch_generative_amd as pg

    torch.manual_seed(0)
    model = pg.models.ImageGPT(**MODEL_KW)
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = synthetic_batch(batch, 0)
    opt_state = otrain.new_opt_state()
    times = []
    for i in range(steps + 1):
        t0 = time.perf_counter()
        _, loss, grads = otrain.loss_and_grads(omodels.image_gpt, state, x, n_heads=HEADS)
        otrain.adam_step_(state, grads, opt_state, lr=LR)
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / steps
    return {
        "value": batch / dt, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"oracle train step (torch-CPU fp32), batch {batch}, {steps} timed steps after 1 warm-up, "
                  f"{dt * 1e3:.0f} ms/step",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024,
                    help="per-GPU batch (weak scaling); 1024 = the saturating batch SURVEY.md §8(d) names for 28x28 models")
    ap.add_argument("--model", default="image_gpt", choices=["image_gpt", *OTHER_MODELS])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    # debugging hooks (1-GPU dev box): PG_FORCE_DEVICE pins every rank to one GPU and
    # PG_DIST_BACKEND=gloo replaces RCCL so the multi-process code path can be exercised there
    local_rank = int(os.environ.get("PG_FORCE_DEVICE", local_rank))
    backend = os.environ.get("PG_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend=backend, rank=rank, world_size=world)

    import pytorch_generative_amd as pg
    from pytorch_generative_amd import graph, ops, optim, parallel

    torch.manual_seed(0)  # identical initial weights on every rank (then broadcast anyway)
    chw, gflop_img, bytes_img = (1, 28, 28), 1.223e9, 26.2e6  # SURVEY.md §8(d), per image
    if args.model == "image_gpt":
        model = pg.models.ImageGPT(**MODEL_KW).to(device)
        lr, lr_decay = LR, LR_DECAY
    else:
        ctor, kw, chw, lr, lr_decay, gflop_img, bytes_img = OTHER_MODELS[args.model]
        model = getattr(pg.models, ctor)(**kw).to(device)
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=lr, lr_decay=lr_decay)
    reducer = None
    if world > 1:
        reducer = parallel.FlatGradAllReduce(opt)
        reducer.broadcast_parameters(src=0)
    x = synthetic_batch(args.batch, rank, chw).to(device)
    if args.model in ("beta_vae", "vd_vae"):
        def loss_fn(xx, preds):  # ELBO: recon.mean() + kl.mean()
            recon, klm = ops.elbo_terms(preds[0], xx, preds[1])
            return recon + klm
    else:
        loss_fn = lambda xx, preds: ops.bce_with_logits_sum_mean(preds, xx)  # noqa: E731

    def eager_step():
        opt.zero_grad()
        loss = loss_fn(x, model(x))
        loss.backward()
        if reducer is not None:
            reducer.all_reduce()
        opt.step()
        return loss.detach()

    launch = "eager"
    step = eager_step
    if not args.no_graph:
        try:
            gstep = graph.GraphedTrainStep(model, opt, loss_fn, x, reducer=reducer, warmup_iters=2)
            step = lambda: gstep()  # noqa: E731
            launch = "hipGraph replay"
        except Exception as e:  # keep the bench alive (e.g. capture refused next to a live RCCL comm)
            print(f"[bench] rank {rank}: hipGraph capture failed ({type(e).__name__}: {e}); "
                  "falling back to eager launches", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
    if world > 1:  # every rank must take the same path (graphs imply a different collective order)
        flag = torch.tensor([1 if launch == "eager" else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) == 1 and launch != "eager":
            step, launch = eager_step, "eager"

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(loss.item())

    if rank == 0:
        global_batch = args.batch * world
        value = global_batch * args.steps / elapsed
        out = {
            "metric": "training images/sec (ImageGPT 8-block/4-head/16-embed, 28x28x1)",
            "value": value,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE.json configs[1]: ImageGPT 8 blocks / 4 heads / 16 embed on "
                            "28x28x1 binarised-MNIST-shaped synthetic; one step = zero_grad + fwd + "
                            "BCE + bwd + global grad-norm + Adam + lr decay (reference trainer.py:173-193)",
                "per_gpu_batch": args.batch,
                "global_batch": global_batch,
                "parallelism": f"dp{world}",
                "launch": launch,
            },
            "loss_nats_per_image": loss_val,
            "bits_per_dim": loss_val / (784 * 0.6931471805599453),
        }
        if args.model != "image_gpt":
            ctor, kw = OTHER_MODELS[args.model][:2]
            out["metric"] = f"training images/sec ({ctor}, {chw[1]}x{chw[2]}x{chw[0]})"
            out["config"]["workload"] = (f"{ctor}({kw}) on {chw[1]}x{chw[2]}x{chw[0]} synthetic; same step "
                                         "definition as the default ImageGPT line (secondary workload)")
            out["bits_per_dim"] = loss_val / (chw[0] * chw[1] * chw[2] * 0.6931471805599453)
            out["roofline"] = {"step_tflops": value * gflop_img / 1e12,
                               "step_hbm_gbps_algorithmic": value * bytes_img / 1e9,
                               "note": "whole-step view against SURVEY.md §8(d) per-image work"}
        elif world == 1:
            r = attention_kernel_roofline(args.batch, device)
            out["roofline"] = {
                "bound": "mfma",  # fp32: matrix peak == vector peak == 157.3 TF on gfx950; the
                                  # kernel is fp32-MFMA + v_exp issue bound
                "kernel": "attn_dkv_m44_kernel (pg_causal_attn_bwd_dkv)",
                "achieved": r["dkv"]["tflops"],
                "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": r["dkv"]["tflops"] / FP32_PEAK_TFLOPS,
                "traffic": measured_traffic(args.batch),
                "launch_ms": r["dkv"]["launch_ms"],
                "flop_per_launch": r["dkv"]["flop_per_launch"],
                "other_kernels": {
                    "attn_fwd_m44_kernel": r["fwd"],
                    "attn_dq_m44_kernel": r["dq"],
                },
                # whole step against SURVEY.md §8(d)'s per-image algorithmic work (1.223 GF, 26.2 MB)
                "step_tflops": value * 1.223e9 / 1e12,
                "step_hbm_gbps_algorithmic": value * 26.2e6 / 1e9,
            }
            if not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
