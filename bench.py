"""bench.py — training throughput of the hot path on MI355X (driver contract: see README/DESIGN).

    python bench.py --gpus 1 --steps K --warmup W          # single GPU
    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks (one per GPU) when no launcher did
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = the reference's Trainer._train_one_batch (trainer.py:173-193): zero_grad, forward,
BCE-with-logits loss, backward, global grad norm, Adam, lr decay — on one resident batch of
synthetic images, replayed from a hipGraph (two graphs with one flat RCCL all-reduce between them
when data parallel). Rank 0 prints ONE JSON line:

  * headline (`value`, `config.workload`): BASELINE.json configs[1], ImageGPT 8 blocks / 4 heads /
    16 embedding channels on 28x28x1 at the saturating per-GPU batch 1024, fp32 (the reference is
    fp32 end to end and parity is gated at 1e-4; DESIGN.md has the bf16 note);
  * `imagegpt_b64`: the same model at the reference's default batch 64 (image_gpt.py:114);
  * `pixel_snail`: the other half of BASELINE.json's metric — configs[3] PixelSNAIL(3, 3, 64, 8, 2,
    4, 32) on 32x32x3 — at a saturating batch and at the reference default 128, with its own
    dominant-kernel roofline;
  * `roofline`: the headline's dominant kernel timed live with HIP events; `cpu_baseline`: the
    oracle's restatement of the same step on the host cores.
`python bench.py --model <name>` benches one workload only (profiling helper; not the driver line).
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

BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (MI355X_MICROARCH.md)
FP32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32 matrix == vector peak (no TF32 on gfx950)
LN2 = 0.6931471805599453


def _attn_flops_per_pair(dk, dv):
    """ALGORITHMIC FLOPs per allowed (query, key) pair (no recomputation counted twice): forward
    QK^T + PV; backward S, dP = dO V^T, dV = P^T dO, dK = dS^T Q, dQ = dS K once each = 6 dk + 4 dv
    (for d_k = d_v this is SURVEY.md §8(d)'s "3.5 x forward"). The fused backward kernel (d_k = d_v = 4)
    executes exactly this; the two-kernel backward executes S and dP twice (dq: 4 dk + 2 dv, dkv:
    4 dk + 4 dv) but is priced on the same algorithmic work split over its two launches."""
    return {"fwd": 2 * dk + 2 * dv, "bwd": 6 * dk + 4 * dv, "dq": 4 * dk + 2 * dv, "dkv": 4 * dk + 4 * dv}


def _causal_gflop_per_img(dense_total, dense_attn, heads, L, dk, dv, blocks, strict):
    """SURVEY.md §8(d) counts attention as dense L^2 at 3.5x forward; the kernels only touch the
    allowed pairs. Returns the per-image training GFLOP with the attention core counted on the causal
    triangle, algorithmic (forward + backward once each, no recomputation)."""
    pairs = heads * (L * (L - 1) / 2 if strict else L * (L + 1) / 2)
    per = _attn_flops_per_pair(dk, dv)
    attn = blocks * pairs * (per["fwd"] + per["bwd"]) / 1e9
    return dense_total - dense_attn + attn


def _pixelcnnpp_gflop_per_img(f=160, n_resnet=5, n_mix=10, hw=32):
    """ALGORITHMIC training GFLOP per image of models.PixelCNNpp (Salimans et al. 2017; the reference has no such model, so
    BASELINE.md section 3 has no row for it): 3 x the forward multiply-adds of every convolution (forward, data gradient,
    weight gradient), 2 FLOP per multiply-add, as the table's other rows are counted. A stride-2 convolution is counted
    at its OUTPUT resolution and a stride-2 transposed convolution at its INPUT resolution — the kernels evaluate both at
    the fine resolution (stride-1 shifted convolution + subsample / zero-insert, 4 x the algorithmic products), which is
    NOT counted. Elementwise work (concat_elu, gates, the mixture loss) is not counted either."""
    macs = 0.0
    px = [hw * hw, (hw // 2) ** 2, (hw // 4) ** 2]
    ds, drs = 6, 4  # taps of the down-shifted 2x3 and the down-right-shifted 2x2 windows
    macs += px[0] * 4 * f * (6 + 3 + 2)  # input layers on [x | ones]: 2x3, 1x3, 2x1
    for s in range(3):  # up pass
        u = (2 * f * f + 2 * f * 2 * f) * ds                       # conv_in 2f -> f, conv_out 2f -> 2f
        ul = (2 * f * f + 2 * f * 2 * f) * drs + 2 * f * f          # + nin (aux = u stream, 2f -> f)
        macs += px[s] * n_resnet * (u + ul)
        if s < 2:
            macs += px[s + 1] * f * f * (ds + drs)                  # stride-2 down-sampling, at the output resolution
    for s, cnt in enumerate([n_resnet, n_resnet + 1, n_resnet + 1]):  # down pass at px[2], px[1], px[0]
        r = px[2 - s]
        u = (2 * f * f + 2 * f * 2 * f) * ds + 2 * f * f            # nin 2f -> f
        ul = (2 * f * f + 2 * f * 2 * f) * drs + 4 * f * f          # nin on [u | ul short-cut]: 4f -> f
        macs += r * cnt * (u + ul)
        if s < 2:
            macs += r * f * f * (ds + drs)                          # stride-2 up-sampling, at the input resolution
    macs += px[0] * f * 10 * n_mix
    return 3 * 2 * macs / 1e9


# name -> constructor, kwargs, (C, H, W), lr, per-batch lr decay, SURVEY §8(d) GFLOP / MB per image
WORKLOADS = {
    "image_gpt": dict(ctor="ImageGPT", kw=dict(in_channels=1, out_channels=1, in_size=28, n_transformer_blocks=8,
                                               n_attention_heads=4, n_embedding_channels=16),
                      chw=(1, 28, 28), lr=5e-3, decay=0.999977, gflop=1.223, mbytes=26.2,
                      gflop_causal=_causal_gflop_per_img(1.223, 1.10, 4, 784, 4, 4, 8, False)),
    # the reference's own reproduce() hyper-parameters (image_gpt.py:147-154): 64 embed / 2 heads -> d = 32
    "image_gpt_repro": dict(ctor="ImageGPT", kw=dict(in_channels=1, out_channels=1, in_size=28, n_transformer_blocks=8,
                                                     n_attention_heads=2, n_embedding_channels=64),
                            chw=(1, 28, 28), lr=5e-3, decay=0.999977, gflop=0.0, mbytes=0.0),
    "pixel_snail": dict(ctor="PixelSNAIL", kw=dict(in_channels=3, out_channels=3, n_channels=64,
                                                   n_pixel_snail_blocks=8, n_residual_blocks=2,
                                                   attention_key_channels=4, attention_value_channels=32),
                        chw=(3, 32, 32), lr=1e-3, decay=0.999977, gflop=7.97, mbytes=110.5,
                        gflop_causal=_causal_gflop_per_img(7.97, 2.114, 1, 1024, 4, 32, 8, True)),
    "gated_pixel_cnn": dict(ctor="GatedPixelCNN", kw=dict(in_channels=3, out_channels=3, n_gated=10,
                                                          gated_channels=128, head_channels=32),
                            chw=(3, 32, 32), lr=1e-3, decay=0.9999, gflop=21.23, mbytes=266.0),
    "pixel_cnn": dict(ctor="PixelCNN", kw=dict(in_channels=1, out_channels=1, n_residual=15,
                                               residual_channels=32, head_channels=32),
                      chw=(1, 28, 28), lr=1e-3, decay=0.999977, gflop=0.964, mbytes=31.6),
    # BASELINE.json configs[2] names PixelCNN++ with the discretized-mixture-of-logistics loss; the reference has no such
    # model: the paper's configuration (Salimans et al. 2017: 160 filters, 5 gated resnets per
    # level, 10 mixture components), images mapped to [-1, 1], loss = ops.dmol_loss_sum_mean
    "pixel_cnn_pp": dict(ctor="PixelCNNpp", kw=dict(in_channels=3, n_filters=160, n_resnet=5, n_mix=10),
                         chw=(3, 32, 32), lr=1e-3, decay=0.999995, gflop=_pixelcnnpp_gflop_per_img(160, 5, 10, 32), mbytes=0.0),
    # BASELINE.json configs[4]: VAE conv stacks + KL on 64x64x3 (ELBO loss, vae.py:149-159)
    "beta_vae": dict(ctor="BetaVAE", kw=dict(in_channels=3, out_channels=3, beta=4.0, latent_channels=16,
                                             strides=[2, 2, 2, 2], hidden_channels=64, residual_channels=32),
                     chw=(3, 64, 64), lr=1e-3, decay=1.0, gflop=1.57, mbytes=17.6),
    "vd_vae": dict(ctor="VeryDeepVAE", kw=dict(in_channels=3, out_channels=3, input_resolution=64,
                                               stack_configs=[(3, 5), (3, 5), (2, 4), (2, 3), (2, 2), (1, 1)],
                                               latent_channels=16, hidden_channels=64, bottleneck_channels=32),
                   chw=(3, 64, 64), lr=5e-4, decay=1.0, gflop=10.96, mbytes=354.0),
}
WORKLOAD_TEXT = {
    "image_gpt": "BASELINE.json configs[1]: ImageGPT 8 blocks / 4 heads / 16 embed on 28x28x1 "
                 "binarised-MNIST-shaped synthetic",
    "pixel_snail": "BASELINE.json configs[3]: PixelSNAIL(3, 3, n_channels=64, 8 blocks, 2 residual "
                   "blocks, key 4 / value 32 channels) on 32x32x3 CIFAR-shaped synthetic",
}
STEP_TEXT = ("one step = zero_grad + fwd + BCE + bwd + global grad-norm + Adam + lr decay "
             "(reference trainer.py:173-193)")


def synthetic_batch(batch, rank, chw=(1, 28, 28)):
    g = torch.Generator().manual_seed(1234 + rank)
    if chw[0] == 1:   # dynamically binarised MNIST-shaped (datasets.py:16-17)
        return torch.bernoulli(torch.full((batch, *chw), 0.1307), generator=g)
    return torch.randint(0, 256, (batch, *chw), generator=g).float() / 255  # CIFAR-shaped


def workload_input(name, x):
    """The tensor a workload's network sees for the synthetic batch x in [0, 1]."""
    return x * 2.0 - 1.0 if name == "pixel_cnn_pp" else x  # PixelCNN++ and its logistic mixture see [-1, 1]


def make_loss_fn(name):
    """loss_fn(x, preds) of a workload, as its reproduce() recipe defines it (e.g. image_gpt.py:158-162, vae.py:149-159)."""
    from pytorch_generative_amd import ops

    if name in ("beta_vae", "vd_vae"):
        def loss_fn(xx, preds):  # ELBO: recon.mean() + kl.mean()
            recon, klm = ops.elbo_terms(preds[0], xx, preds[1])
            return recon + klm
        return loss_fn
    if name == "pixel_cnn_pp":
        n_mix = WORKLOADS[name]["kw"]["n_mix"]
        return lambda xx, preds: ops.dmol_loss_sum_mean(preds, xx, n_mix)
    return lambda xx, preds: ops.bce_with_logits_sum_mean(preds, xx)


class Env:
    def __init__(self, args):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:  # under torch.distributed.run the launcher's world size wins
            args.gpus = self.world
        assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
        # debugging hooks (1-GPU dev box): PG_FORCE_DEVICE pins every rank to one GPU and
        # PG_DIST_BACKEND=gloo replaces RCCL so the multi-process code path can be exercised there
        local_rank = int(os.environ.get("PG_FORCE_DEVICE", local_rank))
        backend = os.environ.get("PG_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        torch.cuda.set_device(local_rank)
        self.device = torch.device("cuda", local_rank)
        if self.world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if backend == "nccl":
                dist.init_process_group(backend="nccl", rank=self.rank, world_size=self.world,
                                        device_id=self.device)
            else:
                dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)

    def barrier(self):
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()


class _Watchdog:
    """A rank that BLOCKS (instead of raising) while it captures or warms up next to a live communicator would hang the whole job:
    every other rank then waits inside a collective that never completes. While armed, a daemon thread ends THIS process with
    exit code 3 after `seconds`; the launcher (bench.py's own _self_spawn, or torch.distributed.run) then stops the other ranks,
    so the command fails instead of hanging until the caller's timeout. PG_BENCH_CAPTURE_TIMEOUT_S (default 300; 0 disables)."""

    def __init__(self, what, rank):
        self.seconds = float(os.environ.get("PG_BENCH_CAPTURE_TIMEOUT_S", "300"))
        self.what, self.rank, self.timer = what, rank, None

    def _fire(self):
        sys.stderr.write(f"[bench] FATAL rank {self.rank}: {self.what} did not finish within {self.seconds:g} s "
                         "(PG_BENCH_CAPTURE_TIMEOUT_S); aborting this rank so that the job fails instead of hanging\n")
        sys.stderr.flush()
        os._exit(3)

    def __enter__(self):
        if self.seconds > 0:
            import threading

            self.timer = threading.Timer(self.seconds, self._fire)
            self.timer.daemon = True
            self.timer.start()
        return self

    def __exit__(self, *exc):
        if self.timer is not None:
            self.timer.cancel()
        return False


def run_workload(env, name, batch, steps, warmup, use_graph=True, require_graph=False, min_seconds=0.0,
                 same_batch=False, single=False, keep_params=False, x_override=None):
    """Builds the model, captures the step and times EXACTLY `steps` steps between barriers
    (max over ranks). Returns the record of this workload at this per-GPU batch.

    min_seconds > 0 (secondary records only, never the headline): if the K timed steps took less, a second window
    of enough steps to fill `min_seconds` is timed and reported instead (`timed_steps` says how many).
    same_batch: every rank trains on RANK 0's batch (--dp-parity); single: no gradient exchange even when the job has
    several ranks (the 1-GPU replica --dp-parity compares with); keep_params: rec["_flat_param"] = the parameters after
    the last step."""
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import graph, ops, optim, parallel

    w = WORKLOADS[name]
    torch.manual_seed(0)  # identical initial weights on every rank (then broadcast anyway)
    model = getattr(pg.models, w["ctor"])(**w["kw"]).to(env.device)
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=w["lr"], lr_decay=w["decay"])
    reducer = None
    if (env.world > 1 and not single) or os.environ.get("PG_BENCH_FORCE_RCCL") == "1":
        # world > 1 under the nccl backend: the direct-RCCL transport (pg_allreduce_sum), captured inside
        # the step's graph; PG_BENCH_FORCE_RCCL=1 runs the same collective with a communicator of ONE
        # rank (what a 1-GPU box can measure of it)
        reducer = parallel.FlatGradAllReduce(opt, transport="rccl" if env.world == 1 else None)
        reducer.broadcast_parameters(src=0)
        if os.environ.get("PG_BENCH_FORCE_SPLIT") == "1":  # diagnosis: two graphs around an eager collective even where it is capturable
            reducer.force_split = True
    if x_override is not None:  # --dp-parity's shard equality: this rank's shard of a global batch / the whole global batch
        x, batch = x_override.to(env.device), int(x_override.shape[0])
    else:
        x = workload_input(name, synthetic_batch(batch, 0 if same_batch else env.rank, w["chw"])).to(env.device)
    loss_fn = make_loss_fn(name)

    def eager_step():
        opt.zero_grad()
        loss = loss_fn(x, model(x))
        loss.backward()
        if reducer is not None:
            reducer.all_reduce()
        opt.step()
        return loss.detach()

    launch, fallback, step = "eager", None, eager_step
    if use_graph:
        try:
            try:
                with _Watchdog(f"hipGraph capture of {name} (warm-up steps + capture"
                               + (", RCCL all-reduce inside)" if reducer is not None and reducer.capturable else ")"), env.rank):
                    gstep = graph.GraphedTrainStep(model, opt, loss_fn, x, reducer=reducer, warmup_iters=2)
            except Exception as e1:  # capture with the collective inside failed: retry with it between two graphs
                if reducer is None or not reducer.capturable:
                    raise
                print(f"[bench] rank {env.rank}: capture with the all-reduce inside the graph failed "
                      f"({type(e1).__name__}: {e1}); retrying with two graphs around an eager collective",
                      file=sys.stderr, flush=True)
                torch.cuda.synchronize()
                reducer.force_split = True
                with _Watchdog(f"hipGraph capture of {name} (two graphs around an eager all-reduce)", env.rank):
                    gstep = graph.GraphedTrainStep(model, opt, loss_fn, x, reducer=reducer, warmup_iters=2)
            step = lambda: gstep()  # noqa: E731
            launch = "hipGraph replay" if not gstep.split else "two hipGraphs around an eager all-reduce"
        except Exception as e:  # e.g. capture refused next to a live RCCL communicator
            fallback = f"{type(e).__name__}: {e}"
            print(f"[bench] WARNING rank {env.rank}: hipGraph capture of {name} FAILED ({fallback}); "
                  "the numbers below are EAGER launches", file=sys.stderr, flush=True)
            torch.cuda.synchronize()
            if require_graph:
                raise SystemExit(f"--require-graph: capture failed on rank {env.rank}: {fallback}")
    launch_by_rank = [launch]
    if env.world > 1 and not single:  # every rank must take the same path (graphs imply a different collective order)
        with _Watchdog("agreeing on the launch mode with the other ranks", env.rank):
            modes = [None] * env.world
            dist.all_gather_object(modes, launch)
        launch_by_rank = modes  # what every rank ended its capture attempt in, before any fallback to a common mode
        kinds = set(modes)
        if len(kinds) > 1:  # mixed: a graph with the collective inside, split graphs and eager launches order it differently
            if require_graph:
                raise SystemExit(f"--require-graph: the ranks ended in different launch modes: {modes}")
            step, launch = eager_step, "eager"
            fallback = fallback or f"the ranks ended in different launch modes ({modes}); all of them run eager launches"

    for _ in range(warmup):
        loss = step()
    env.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    env.barrier()
    elapsed = time.perf_counter() - t0
    if min_seconds > 0:  # a thin window (secondary records): time a second one long enough; every rank decides alike
        t = torch.tensor([elapsed], device=env.device, dtype=torch.float64)
        if env.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if float(t.item()) < min_seconds:
            steps = int(min_seconds / (float(t.item()) / steps)) + 1
            env.barrier()
            t0 = time.perf_counter()
            for _ in range(steps):
                loss = step()
            env.barrier()
            elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=env.device, dtype=torch.float64)
    per_rank = [elapsed]
    if env.world > 1:
        every = [torch.zeros_like(t) for _ in range(env.world)]
        dist.all_gather(every, t)
        per_rank = [float(e.item()) for e in every]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    loss_val = float(loss.item())
    nranks = 1 if single else env.world  # single: every rank ran its own replica; the rate is ONE replica's
    value = batch * nranks * steps / elapsed
    dims = w["chw"][0] * w["chw"][1] * w["chw"][2]
    rec = {
        "images_per_s": value, "ms_per_step": elapsed / steps * 1e3, "timed_steps": steps, "timed_seconds": elapsed,
        "per_gpu_batch": batch,
        "global_batch": batch * nranks, "launch": launch, "graph_fallback": fallback, "launch_by_rank": launch_by_rank,
        # each rank's own clock between the two barriers (the job's rate uses the MAX)
        "per_rank_images_per_s": {"min": batch * steps / max(per_rank), "max": batch * steps / min(per_rank)},
        "grad_exchange": None if reducer is None else
        f"{reducer.transport} all-reduce of the flat gradient ({opt.flat_grad.numel() * 4 / 1e6:.2f} MB), "
        + ("ONE message inside the step graph, stream-ordered between backward and grad-norm/Adam; not overlapped "
           "with compute (the norm needs every gradient, DESIGN.md section 5)"
           if launch == "hipGraph replay" and reducer.capturable else "eager launch between two graphs"),
        "loss_nats_per_image": loss_val, "bits_per_dim": loss_val / (dims * LN2),
    }
    if w["gflop"] > 0:  # whole step against the per-image algorithmic work of SURVEY.md §8(d) (omitted where not accounted)
        rec["step_tflops_dense_attention_count"] = value * w["gflop"] / 1e3
    if w["mbytes"] > 0:
        rec["step_hbm_gbps_algorithmic"] = value * w["mbytes"] * 1e6 / 1e9
    if keep_params:
        torch.cuda.synchronize()
        rec["_flat_param"] = opt.flat_param.detach().clone()
    if "gflop_causal" in w:  # attention counted on the allowed pairs only (what the kernels execute)
        rec["step_tflops"] = value * w["gflop_causal"] / 1e3
        rec["step_frac_of_fp32_peak"] = rec["step_tflops"] / env.world / FP32_PEAK_TFLOPS
    del model, opt, step
    if reducer is not None:
        torch.cuda.synchronize()
        reducer.close()
    torch.cuda.empty_cache()
    return rec


def _event_time(fn, stream, iters=10):
    fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
    for a, b in evs:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in evs) / iters


def attention_kernel_roofline(batch, device, heads, dk, dv, hw, strict):
    """Times the three causal-attention kernels live with HIP events on the stream they are
    launched on, through the C-ABI, at the bench's exact shapes and on random data."""
    from pytorch_generative_amd import _lib

    lib = _lib.load()
    e, vd, L = heads * dk, heads * dv, hw * hw
    g = torch.Generator().manual_seed(7)
    mk = lambda c: torch.randn(batch, c, hw, hw, generator=g).to(device)  # noqa: E731
    q, kv, d_o = mk(e), mk(e + vd), mk(vd)
    o, dq, dkv = torch.empty_like(d_o), torch.empty_like(q), torch.empty_like(kv)
    lse = torch.empty(batch, heads, L, device=device)
    delta = torch.empty_like(lse)
    stream = torch.cuda.current_stream()
    st = stream.cuda_stream
    kvs = (e + vd) * L

    def fwd():
        _lib.check(lib.pg_causal_attn_fwd(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * e * L,
                                          o.data_ptr(), lse.data_ptr(), batch, heads, L, dk, dv,
                                          e * L, kvs, kvs, vd * L, int(strict), st), "fwd")

    def bwd(fn):
        _lib.check(fn(q.data_ptr(), kv.data_ptr(), kv.data_ptr() + 4 * e * L, o.data_ptr(),
                      d_o.data_ptr(), lse.data_ptr(), delta.data_ptr(), dq.data_ptr(),
                      dkv.data_ptr(), dkv.data_ptr() + 4 * e * L, batch, heads, L, dk, dv, e * L,
                      kvs, kvs, vd * L, vd * L, e * L, kvs, kvs, int(strict), st), "bwd")

    ms = {"fwd": _event_time(fwd, stream),
          "dq": _event_time(lambda: bwd(lib.pg_causal_attn_bwd_dq), stream),
          "dkv": _event_time(lambda: bwd(lib.pg_causal_attn_bwd_dkv), stream)}
    if dk == 4 and dv in (4, 16, 32) and lib.pg_attn_fused_bwd(-1) == 1:  # one fused launch (attn_bwd_m44 / attn_bwd_k4)
        ms["bwd"] = _event_time(lambda: bwd(lib.pg_causal_attn_bwd), stream)
    pairs = batch * heads * (L * (L - 1) / 2 if strict else L * (L + 1) / 2)
    per = _attn_flops_per_pair(dk, dv)
    return {k: {"launch_ms": ms[k], "flop_per_launch": pairs * per[k],
                "tflops": pairs * per[k] / ms[k] / 1e9} for k in ms}


def conv_kernel_roofline(batch, device, cin=64, cout=64, hw=32):
    """PixelSNAIL's dominant kernel: conv_mfma_kernel<4, 4> on its most frequent problem — the 2x2
    (pad 1, cropped) 64 -> 64 convolution of ResidualBlock with the fused ELU prologue
    (pixel_snail.py:41-55), forward launch, timed through the C-ABI with HIP events."""
    from pytorch_generative_amd import _lib, ops

    lib = _lib.load()
    spec = ops.ConvSpec(2, 2, 1, 1)
    g = torch.Generator().manual_seed(11)
    x = torch.randn(batch, cin, hw, hw, generator=g).to(device)
    wt = (torch.randn(cout, cin, 2, 2, generator=g) * 0.05).to(device)
    bias = torch.zeros(cout, device=device)
    out = torch.empty(batch, cout, hw, hw, device=device)
    fmt = ops._use_mfma(lib, cin, cout, spec, (hw, hw), hw)
    wfrag = ops._pack_frag(lib, wt, spec, False, fmt)
    stream = torch.cuda.current_stream()
    T = len(spec.fwd_taps)

    def run():
        _lib.check(lib.pg_conv2d_mfma(x.data_ptr(), wfrag.data_ptr(), bias.data_ptr(), 0, out.data_ptr(),
                                      batch, cin, hw, hw, cout, hw, hw, T, spec.f_dr, spec.f_dc,
                                      ops.ACT_ELU, 0, ops.ACT_NONE, ops.ACT_NONE, fmt, stream.cuda_stream),
                   "pg_conv2d_mfma")

    ms = _event_time(run, stream)
    flop = 2.0 * batch * hw * hw * cin * cout * T
    return {"launch_ms": ms, "flop_per_launch": flop, "tflops": flop / ms / 1e9}


def wgrad_kernel_roofline(batch, device, cin=64, cout=64, hw=32, k=(2, 2, 1, 1)):
    """The weight gradient of the same 2x2 64 -> 64 convolution (conv_wgrad_b3r_kernel<4, 1, 2>, the row-ring kernel of round 6, +
    wgrad_reduce_kernel behind pg_conv2d_wgrad), timed through the C-ABI with HIP events.
    k = (kh, kw, pad_h, pad_w) selects another window (profiling helper: tools/exp/pmc_launch.py)."""
    from pytorch_generative_amd import _lib, ops

    lib = _lib.load()
    spec = ops.ConvSpec(*k)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(batch, cin, hw, hw, generator=g).to(device)
    dy = torch.randn(batch, cout, hw, hw, generator=g).to(device)
    dw = torch.zeros(cout, cin, k[0], k[1], device=device)
    db = torch.zeros(cout, device=device)
    T = len(spec.wg_taps)
    ws_n = lib.pg_conv2d_wgrad_workspace_floats(cout, cin, T)
    ws = torch.empty(ws_n, device=device)
    stream = torch.cuda.current_stream()

    def run():
        _lib.check(lib.pg_conv2d_wgrad(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), db.data_ptr(), batch, cin,
                                       hw, hw, cout, hw, hw, k[0], k[1], T, spec.w_dr, spec.w_dc, spec.w_u, spec.w_v,
                                       ops.ACT_ELU, ws.data_ptr(), ws_n, stream.cuda_stream),
                   "pg_conv2d_wgrad")

    ms = _event_time(run, stream)
    flop = 2.0 * batch * hw * hw * cin * cout * T
    return {"launch_ms": ms, "flop_per_launch": flop, "tflops": flop / ms / 1e9}


def measured_traffic(batch, kernel, files=("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json",
                                           "r02_traffic.json", "r01_traffic.json"), with_counters=False):
    """(HBM bytes per launch, source file) of a kernel from the COMMITTED PMC profiles (collected in separate
    rocprofv3 --pmc passes, see profiles/README.md; newest round first); (None, None) when no committed profile holds
    this kernel at this batch. The figure is not measured by this run — the line names the file it comes from.
    with_counters: a third value, the kernel's counter record of that file ({"mfma_busy_frac", "waves_per_simd", ...} or None)."""
    none = (None, None, None) if with_counters else (None, None)
    for name in files:
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                t = json.load(f)
            if t.get("per_gpu_batch") != batch:
                continue
            per = t.get("kernels", {})
            for k, v in per.items():
                if kernel in k and "hbm_read_bytes" in v:
                    out = (v["hbm_read_bytes"] + v["hbm_write_bytes"], "profiles/" + name)
                    return out + (v,) if with_counters else out
            if kernel.startswith("attn_dkv") and "attn_bwd_dkv_bytes_per_launch" in t:
                out = (t["attn_bwd_dkv_bytes_per_launch"], "profiles/" + name)
                return out + (None,) if with_counters else out
        except (OSError, ValueError, AttributeError):
            pass
    return none


def _counter_fields(traffic, counters, launch_ms):
    """north_star: "rocprof counters reported as achieved HBM GB/s and MFMA utilisation": the committed counter pass of the
    kernel (bytes, matrix-pipe busy fraction, resident waves) next to THIS run's launch time."""
    out = {"hbm_gbps_achieved": None if traffic is None else traffic / (launch_ms * 1e-3) / 1e9,
           "hbm_gbps_what": "HBM bytes per launch of the committed rocprofv3 --pmc pass / this run's HIP-event launch time; "
                            "peak 8000 GB/s spec, 6290 GB/s measured copy (MI355X_MICROARCH.md)"}
    if counters:
        out["mfma_busy_frac"] = counters.get("mfma_busy_frac")
        out["waves_per_simd"] = counters.get("waves_per_simd")
        if counters.get("SQ_INSTS_MFMA"):
            out["valu_per_mfma"] = counters.get("SQ_INSTS_VALU", 0.0) / counters["SQ_INSTS_MFMA"]
    return out


def _physical_cores():
    try:
        seen = set()
        with open("/proc/cpuinfo") as f:
            phys = core = None
            for line in f:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        return len(seen) or None
    except OSError:
        return None


def _cpu_step_timer(forward, state0, x, lr, **fwd_kw):
    """Returns timed(threads, n_steps) -> (seconds per oracle train step after one warm-up step, loss)."""
    from oracle import train as otrain

    def timed(threads, n_steps):
        torch.set_num_threads(threads)
        state = {k: v.clone() for k, v in state0.items()}
        opt_state = otrain.new_opt_state()
        ts = []
        for _ in range(n_steps + 1):
            t0 = time.perf_counter()
            _, loss, grads = otrain.loss_and_grads(forward, state, x, **fwd_kw)
            otrain.adam_step_(state, grads, opt_state, lr=lr)
            ts.append(time.perf_counter() - t0)
        return sum(ts[1:]) / n_steps, float(loss)

    return timed


def _ref_ratio(model):
    try:
        with open(os.path.join(ROOT, "profiles", "r04_cpu_reference_vs_oracle.json")) as f:
            d = json.load(f)
        m = d["models"][model]
        return {"ratio": m["oracle_over_reference_step_time"], "threads": d["threads"], "batch": m["batch"],
                "timed_steps": d["timed_steps"], "where": d["where"]}
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(batch=32, point_cap_s=20.0):
    """The oracle's restatement of the same ImageGPT training step on the host cores (kind 'port':
    the reference is Python and cannot travel to the GPU box; the oracle dispatches the same torch
    CPU primitives — oneDNN / MKL — in the same order). Bounded sample: a thread sweep over
    {16, 32, 64, physical cores} with 1 warm-up + 1 timed step per point (a point whose step exceeds
    `point_cap_s` ends the sweep), then 3 timed steps at the best setting, at batch 32 (the survey's
    probe batch; the reference default 64 gives the same images/s within noise but doubles the
    sample's wall time)."""
    from oracle import models as omodels

    import pytorch_generative_amd as pg

    w = WORKLOADS["image_gpt"]
    torch.manual_seed(0)
    model = pg.models.ImageGPT(**w["kw"])
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = synthetic_batch(batch, 0)
    logical, physical = os.cpu_count(), _physical_cores()
    default_threads = torch.get_num_threads()
    timed = _cpu_step_timer(omodels.image_gpt, state0, x, w["lr"], n_heads=4)
    sweep = {}
    for threads in sorted({t for t in (16, 32, 64, physical or logical) if t and t <= logical}):
        sweep[threads] = timed(threads, 1)[0]
        if sweep[threads] > point_cap_s:
            break
    best = min(sweep, key=sweep.get)
    dt, loss = timed(best, 5)
    torch.set_num_threads(default_threads)
    return {
        "value": batch / dt, "unit": "images/s", "cores": best, "cores_logical": logical,
        "cores_physical": physical, "kind": "port", "torch": torch.__version__,
        "ms_per_step": dt * 1e3, "loss_after": loss,
        "thread_sweep_ms_per_step": {str(k): v * 1e3 for k, v in sweep.items()},
        "sample": f"oracle train step (torch-CPU fp32, ImageGPT 8/4/16), batch {batch}, 5 timed steps after "
                  f"1 warm-up at {best} threads (best of a sweep over {sorted(sweep)} threads, 1 timed step each)",
        # kind "port": how the oracle's step relates to the REAL reference trainer step, timed side by side in the build
        # container (same model, batch, threads; tools/cpu_ref_vs_oracle.py -> profiles/r04_cpu_reference_vs_oracle.json)
        "oracle_over_reference_step_time": _ref_ratio("image_gpt"),
    }


def cpu_baseline_pixel_snail(threads, batch=8):
    """The oracle's PixelSNAIL (configs[3]) training step on the host cores at `threads` threads (the
    best setting of the ImageGPT sweep): 1 warm-up + 2 timed steps at batch 8."""
    from oracle import models as omodels

    import pytorch_generative_amd as pg

    w = WORKLOADS["pixel_snail"]
    torch.manual_seed(0)
    model = pg.models.PixelSNAIL(**w["kw"])
    state0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = synthetic_batch(batch, 0, w["chw"])
    default_threads = torch.get_num_threads()
    dt, loss = _cpu_step_timer(omodels.pixel_snail, state0, x, w["lr"])(threads, 5)
    torch.set_num_threads(default_threads)
    return {"value": batch / dt, "unit": "images/s", "cores": threads, "kind": "port",
            "ms_per_step": dt * 1e3, "loss_after": loss,
            "sample": f"oracle train step (torch-CPU fp32, PixelSNAIL cfg3), batch {batch}, 5 timed steps after "
                      f"1 warm-up at {threads} threads",
            "oracle_over_reference_step_time": _ref_ratio("pixel_snail")}


# fp32-compute / HBM ceilings per GPU in images/s (BASELINE.md §3 = SURVEY.md §8(d)) for the compact records:
# frac_of_fp32_compute_ceiling = images_per_s / ceiling, ceiling = 157.3 TFLOP/s / (GFLOP per image of WORKLOADS[...])
CEILINGS = {"pixel_cnn": 163e3, "gated_pixel_cnn": 7.4e3, "beta_vae": 100e3, "vd_vae": 14.3e3,
            # no BASELINE.md row (not in the reference): the same formula on _pixelcnnpp_gflop_per_img's algorithmic count
            "pixel_cnn_pp": FP32_PEAK_TFLOPS * 1e3 / WORKLOADS["pixel_cnn_pp"]["gflop"]}
# dominant kernel of each compact record and its share of the step's kernel time, from the tracked rocprofv3 tables
# (profiles/r05_<model>_kernel_stats.csv, tools/collect_profiles_r05.sh stats); None = read the table
DOMINANT = {}


def _dominant_kernel(model):
    """(name, share of kernel time, average microseconds) of the top kernel in the newest committed rocprofv3 table of this
    workload, profiles/r06_<model>_kernel_stats.csv (else r05, r04): a committed profile, not a measurement of this run."""
    import csv

    try:
        path = next(p for p in (os.path.join(ROOT, "profiles", f"{r}_{model}_kernel_stats.csv") for r in ("r06", "r05", "r04"))
                    if os.path.exists(p))
        with open(path) as f:
            rows = list(csv.DictReader(f))
        tot = sum(float(r["TotalDurationNs"]) for r in rows)
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        name = top["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name[:name.find("(")] if "(" in name else name
        return {"kernel": name, "share_of_kernel_time": float(top["TotalDurationNs"]) / tot,
                "avg_us": float(top["AverageNs"]) / 1e3, "source": "profiles/" + os.path.basename(path)}
    except (OSError, ValueError, KeyError, StopIteration):
        return None
OTHER_MIN_SECONDS = 0.5  # every secondary record is timed over at least this long (and at least the headline's steps)
# how parity is gated (tests/) — quoted beside the errors MEASURED in this run (measured_parity)
PARITY_GATE_TEXT = ("tests gate fp32 outputs / losses / grad norm at 1e-4 and every parameter gradient at 1e-4 of its tensor's "
                    "maximum plus element-wise |d| <= 1e-4 |want| + 5e-6 max|want| (tests/_util.py) against the torch-CPU oracle "
                    "and the reference's golden vectors; causal masks, the attention kernels' admitted (query, key) set, masked "
                    "weights and positional encodings bit-exact; K graph replays == K eager steps for every timed workload")


def measured_parity(name, batch, device, n_images=4):
    """the speed is only valid next to the error of the SAME kernels on the SAME data. The first
    `n_images` images of this workload's timed batch (rank 0's synthetic batch) go through one forward + loss + backward of
    the HIP path (the default kernels the bench times, eager launches) and through the CPU oracle (reference step:
    trainer.py:173-193 up to the gradients); returns the measured errors. The oracle is used here as the checker only."""
    from oracle import models as omodels
    from oracle import train as otrain

    import pytorch_generative_amd as pg

    w = WORKLOADS[name]
    torch.manual_seed(0)
    model = getattr(pg.models, w["ctor"])(**w["kw"])
    state = {k: v.detach().clone() for k, v in model.state_dict().items()}
    x = workload_input(name, synthetic_batch(batch, 0, w["chw"]))[:n_images].contiguous()
    kw = {"n_heads": w["kw"]["n_attention_heads"]} if w["ctor"] == "ImageGPT" else {}
    fwd = omodels.FORWARDS["image_gpt" if w["ctor"] == "ImageGPT" else name]
    threads = torch.get_num_threads()
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    o_logits, o_loss, o_grads = otrain.loss_and_grads(fwd, state, x, **kw)
    torch.set_num_threads(threads)
    model = model.to(device)
    model.train()
    xg = x.to(device)
    logits = model(xg)
    loss = make_loss_fn(name)(xg, logits)
    loss.backward()
    torch.cuda.synchronize()
    rel = lambda got, want: float((got.detach().double().cpu() - want.double()).abs().max() / want.double().abs().max())  # noqa: E731
    worst, worst_name, worst_elem, worst_elem_name, n_t = 0.0, None, 0.0, None, 0
    for k, prm in model.named_parameters():
        want = o_grads.get(k)
        if want is None or prm.grad is None or float(want.abs().max()) == 0.0:
            continue
        n_t += 1
        d = (prm.grad.detach().double().cpu() - want.double()).abs()
        m = float(want.abs().max())
        e = float(d.max()) / m
        el = float((d / (1e-4 * want.double().abs() + 5e-6 * m)).max())
        if e > worst:
            worst, worst_name = e, k
        if el > worst_elem:
            worst_elem, worst_elem_name = el, k
    del model
    torch.cuda.empty_cache()
    return {"images": n_images, "of_per_gpu_batch": batch, "logits_max_norm_err": rel(logits, o_logits),
            "loss_rel_err": abs(float(loss) - float(o_loss)) / abs(float(o_loss)),
            "loss_nats_per_image": {"hip": float(loss), "oracle": float(o_loss)},
            "gradient_tensors": n_t, "worst_gradient_max_norm_err": worst, "worst_gradient_tensor": worst_name,
            "worst_gradient_elementwise_ratio": worst_elem, "worst_gradient_elementwise_tensor": worst_elem_name,
            "what": "measured in THIS run: one forward + loss + backward of the timed kernels on the first images of the timed "
                    "batch against the torch-CPU oracle (oracle/, pinned to the reference by tests/test_oracle_pin.py); max-norm "
                    "err = max|got - want| / max|want| per tensor; element-wise ratio = max |d| / (1e-4 |want| + 5e-6 max|want|), "
                    "<= 1 passes the tests' gate", "gate": PARITY_GATE_TEXT}


OTHER_CONFIGS = [  # (record key, workload, per-GPU batch, BASELINE.json config)
    ("pixel_cnn", "pixel_cnn", 1024, "configs[0]"),
    ("gated_pixel_cnn", "gated_pixel_cnn", 512, "configs[2]"),
    ("pixel_cnn_pp", "pixel_cnn_pp", 64, "configs[2]"),
    ("beta_vae", "beta_vae", 1024, "configs[4]"),
    ("vd_vae", "vd_vae", 512, "configs[4]"),
]


def _self_spawn(n):
    """`python bench.py --gpus N` without a launcher: start one process per GPU ourselves (the same command with
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), forward rank 0's JSON line, return the
    worst exit code. Equivalent to `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py ...`."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    # all ranks are watched together: a rank that dies leaves the others inside a collective that never completes, so the
    # survivors get a grace period and are then stopped — the command fails instead of hanging until the caller's timeout
    import time

    rc, failed_at = 0, None
    grace = float(os.environ.get("PG_BENCH_RANK_GRACE_S", "15"))
    try:
        while any(p.poll() is None for p in procs):
            codes = [p.poll() for p in procs]
            if failed_at is None and any(c not in (None, 0) for c in codes):
                failed_at = time.monotonic()
                bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
                sys.stderr.write(f"[bench] rank(s) exited with an error: {bad}; stopping the others in {grace:g} s\n")
            if failed_at is not None and time.monotonic() - failed_at > grace:
                break
            time.sleep(0.2)
        for p in procs:
            c = p.poll()
            rc = max(rc, 1 if c is None else abs(c))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def dp_parity(env, args, run):
    """`bench.py --gpus N --dp-parity`: DDP's defining equality (reference trainer.py:78-82, train.py:27-44; SURVEY.md
    section 8(e)) checked ON THE TRANSPORT THE JOB USES (RCCL over xGMI on a multi-GPU node; gloo in the 1-GPU tests):
    with every rank fed rank 0's batch, the all-reduced mean gradient is the 1-GPU gradient, so after the same number
    of steps every rank must hold (a) exactly rank 0's parameters and (b) the parameters of a 1-rank run of the same
    steps — bit for bit when the world size is a power of two (sum of N equal values and the 1/N pre-scale are exact)
    and the bit-reproducible kernels are selected (ops.set_deterministic). Both runs are timed like any other bench
    run, so the same command yields the N-GPU rate and the 1-rank rate of one rank (scaling_efficiency_vs_n1)."""
    import hashlib

    from pytorch_generative_amd import ops

    was = ops.set_deterministic(True)
    name = args.model or "image_gpt"
    dp = run(name, args.batch, same_batch=True, keep_params=True)
    one = run(name, args.batch, same_batch=True, single=True, keep_params=True)
    ops.set_deterministic(was)
    # SURVEY.md section 8(e)'s second equality: N DISTINCT shards == the 1-GPU step on the concatenated batch (loss = mean of the
    # per-rank means, gradient = mean of the per-rank gradients; DistributedDataParallel's averaging, trainer.py:78-82). Not bit
    # exact: the one replica sums N * B images in another order than N ranks sum B each (~1e-6 of a gradient tensor's maximum),
    # and Adam's normalised update turns a relative change e of a gradient entry into e * lr per step.
    w = WORKLOADS[name]
    shards = [workload_input(name, synthetic_batch(args.batch, r, w["chw"])) for r in range(env.world)]
    shard = run(name, args.batch, x_override=shards[env.rank], keep_params=True)
    whole = run(name, args.batch, x_override=torch.cat(shards, 0), single=True, keep_params=True)
    p_shard, p_whole = shard.pop("_flat_param"), whole.pop("_flat_param")
    n_steps = (2 if shard["launch"] != "eager" else 0) + args.warmup + args.steps
    shard_pmax = float(p_whole.abs().max())
    shard_tol = 5e-5 * shard_pmax + 3e-3 * n_steps * w["lr"]
    d_sh = (p_shard - p_whole).abs()
    # entries whose true gradient is ~0 (e.g. the key half of an attention `_kv.bias`: softmax is shift invariant) take +-lr steps
    # whose sign is round-off — in the reference too (tests/test_gpu_models.py::test_golden_step_flat_adam); they are bounded by
    # Adam's step bound and must stay a small fraction of the parameters
    sh = torch.tensor([float(d_sh.max()), float((d_sh > shard_tol).double().mean())], device=env.device, dtype=torch.float64)
    if env.world > 1:
        dist.all_reduce(sh, op=dist.ReduceOp.MAX)
    shard_diff, shard_frac_out = (float(v) for v in sh)
    shard_ok = shard_diff <= 2.0 * n_steps * w["lr"] * 1.001 and shard_frac_out <= 0.01
    mine, alone = dp.pop("_flat_param"), one.pop("_flat_param")
    root = mine.clone()
    if env.world > 1:
        dist.broadcast(root, src=0)
    stats = torch.tensor([float((mine - root).abs().max()), float((mine - alone).abs().max()),
                          float(alone.abs().max())], device=env.device, dtype=torch.float64)
    if env.world > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.MAX)
    vs_root, vs_one, pmax = (float(v) for v in stats)
    pow2 = env.world & (env.world - 1) == 0
    ok = vs_root == 0.0 and (vs_one == 0.0 if pow2 else vs_one <= 1e-6 * pmax) and shard_ok
    if env.rank == 0:
        digest = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]  # noqa: E731
        print(json.dumps({
            "metric": f"training images/sec ({w['ctor']}, {w['chw'][1]}x{w['chw'][2]}x{w['chw'][0]}) with the data-parallel "
                      "equality check", "value": dp["images_per_s"], "unit": "images/s", "n_gpus": env.world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dp["ms_per_step"], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic (every rank: rank 0's batch)",
            "config": {"workload": f"{w['ctor']}({w['kw']}); {STEP_TEXT}", "per_gpu_batch": args.batch,
                       "global_batch": args.batch * env.world, "parallelism": f"dp{env.world}", "launch": dp["launch"],
                       "graph_fallback": dp["graph_fallback"]},
            "grad_exchange": dp["grad_exchange"],
            "dp_parity": {
                "ok": ok, "max_abs_diff_vs_rank0": vs_root, "max_abs_diff_vs_one_rank_run": vs_one,
                "param_abs_max": pmax, "exact_expected": pow2, "kernels": "bit-reproducible (ops.set_deterministic)",
                "steps_compared": 2 + args.warmup + args.steps if dp["launch"] != "eager" else args.warmup + args.steps,
                "params_sha256_16": {"rank0_of_the_job": digest(mine), "one_rank_run": digest(alone)},
                "what": "max over ranks of |params - rank 0's params| and |params - params of a 1-rank run of the same "
                        "steps on the same batch| (SURVEY.md section 8(e): same batch on all ranks == 1 GPU)",
                "shards": {"ok": shard_ok, "max_abs_diff_vs_one_rank_run_on_the_concatenated_batch": shard_diff,
                           "tolerance": shard_tol, "fraction_of_parameters_beyond_tolerance": shard_frac_out,
                           "adam_step_bound": 2.0 * n_steps * w["lr"], "param_abs_max": shard_pmax, "steps_compared": n_steps,
                           "launch": shard["launch"], "launch_by_rank": shard["launch_by_rank"],
                           "what": "rank r trains on its own shard (seed 1234 + r), one replica on the concatenation of all "
                                   "shards: SURVEY.md section 8(e)'s 'N distinct shards == 1 GPU on the concatenated batch' "
                                   "(not bit exact: another summation order; tolerance 5e-5 max|p| + 3e-3 steps lr for >= 99 % of "
                                   "the parameters, Adam's step bound 2 steps lr for the rest — zero-gradient entries take "
                                   "+-lr steps of round-off sign)"}},
            "launch_by_rank": dp["launch_by_rank"],
            "one_rank_run": {"images_per_s": one["images_per_s"], "ms_per_step": one["ms_per_step"], "launch": one["launch"]},
            "scaling_efficiency_vs_n1": dp["images_per_s"] / (env.world * one["images_per_s"]),
        }), flush=True)
    if not ok:
        raise SystemExit(f"--dp-parity FAILED on rank {env.rank}: |p - p_rank0| = {vs_root:.3e}, |p - p_1rank| = {vs_one:.3e}, "
                         f"shards: |p - p_concatenated| = {shard_diff:.3e} (tolerance {shard_tol:.3e})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1024,
                    help="per-GPU batch of the headline (weak scaling); 1024 = the saturating batch "
                         "SURVEY.md §8(d) names for 28x28 models")
    ap.add_argument("--snail-batch", type=int, default=1024,
                    help="per-GPU batch of the PixelSNAIL record (saturating; the reference default 128 is reported beside it)")
    ap.add_argument("--model", default=None, choices=sorted(WORKLOADS),
                    help="bench this one workload only (profiling helper)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline only (no batch-64 / PixelSNAIL records)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--n1-value", type=float, default=None,
                    help="images/s of the same headline at --gpus 1: adds scaling_efficiency_vs_n1 to the line")
    ap.add_argument("--dp-parity", action="store_true",
                    help="SURVEY.md section 8(e) on the job's own transport: every rank trains on rank 0's batch with the "
                         "bit-reproducible kernels, then its parameters are compared with rank 0's and with a 1-rank replica "
                         "of the same steps; prints ONE JSON line with dp_parity and the job's rate")
    ap.add_argument("--require-graph", action="store_true",
                    help="exit non-zero if hipGraph capture fails on any rank instead of falling back to eager")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(_self_spawn(args.gpus))
    env = Env(args)
    graph_on = not args.no_graph

    def run(name, batch, steps=None, **kw):
        return run_workload(env, name, batch, steps or args.steps, args.warmup, graph_on, args.require_graph, **kw)

    if args.dp_parity:
        dp_parity(env, args, run)
        if env.world > 1:
            dist.destroy_process_group()
        return

    if args.model is not None:  # single-workload helper line
        w = WORKLOADS[args.model]
        rec = run(args.model, args.batch)
        if env.rank == 0:
            print(json.dumps({
                "metric": f"training images/sec ({w['ctor']}, {w['chw'][1]}x{w['chw'][2]}x{w['chw'][0]})",
                "value": rec["images_per_s"], "unit": "images/s", "n_gpus": env.world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": rec["ms_per_step"], "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"{w['ctor']}({w['kw']}) on synthetic {w['chw']}; {STEP_TEXT} "
                                       "(secondary workload, not the driver line)",
                           "per_gpu_batch": args.batch, "global_batch": args.batch * env.world,
                           "parallelism": f"dp{env.world}", "launch": rec["launch"],
                           "graph_fallback": rec["graph_fallback"]},
                "record": rec}), flush=True)
        if env.world > 1:
            dist.destroy_process_group()
        return

    head = run("image_gpt", args.batch)
    extras = {}
    if not args.no_extras:
        extras["imagegpt_b64"] = run("image_gpt", 64)
        snail = run("pixel_snail", args.snail_batch)
        snail["reference_default_batch_128"] = run("pixel_snail", 128)
        extras["pixel_snail"] = snail
        others = {}
        for key, name, batch, cfg in OTHER_CONFIGS:  # compact driver-run records of the other BASELINE configs
            r = run(name, batch, min_seconds=OTHER_MIN_SECONDS)
            others[key] = {"baseline_config": cfg, "images_per_s": r["images_per_s"], "ms_per_step": r["ms_per_step"],
                           "timed_steps": r["timed_steps"], "timed_seconds": r["timed_seconds"],
                           "per_gpu_batch": batch, "launch": r["launch"],
                           "frac_of_fp32_compute_ceiling": (r["images_per_s"] / env.world / CEILINGS[key]
                                                            if key in CEILINGS else None),
                           "loss_nats_per_image": r["loss_nats_per_image"],
                           "dominant_kernel": _dominant_kernel(name)}
        extras["other_configs"] = others

    if env.rank == 0:
        out = {
            "metric": "training images/sec (ImageGPT 8-block/4-head/16-embed, 28x28x1)",
            "value": head["images_per_s"],
            "unit": "images/s",
            "n_gpus": env.world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": WORKLOAD_TEXT["image_gpt"] + "; " + STEP_TEXT,
                "per_gpu_batch": args.batch,
                "global_batch": args.batch * env.world,
                "parallelism": f"dp{env.world}",
                "launch": head["launch"],
                "graph_fallback": head["graph_fallback"],
                "launch_by_rank": head["launch_by_rank"],  # what every rank's capture attempt ended in (before a common fallback)
            },
            "loss_nats_per_image": head["loss_nats_per_image"],
            "bits_per_dim": head["bits_per_dim"],
            "parity": PARITY_GATE_TEXT,  # replaced below by the errors measured in this run (rank 0, unless --no-cpu-baseline)
            "grad_exchange": head["grad_exchange"],
            "per_rank_images_per_s": head["per_rank_images_per_s"],
        }
        if args.n1_value:  # the driver computes efficiency itself; this is a convenience for manual runs
            out["scaling_efficiency_vs_n1"] = head["images_per_s"] / (env.world * args.n1_value)
        if "imagegpt_b64" in extras:
            b64 = extras["imagegpt_b64"]
            out["imagegpt_b64"] = {"what": "same model and step at the reference's default per-GPU batch 64 "
                                           "(image_gpt.py:114)", **b64}
        if "pixel_snail" in extras:
            out["pixel_snail"] = {"workload": WORKLOAD_TEXT["pixel_snail"] + "; " + STEP_TEXT,
                                  "unit": "images/s", "dtype": "f32", **extras["pixel_snail"]}
        if "other_configs" in extras:
            out["other_configs"] = {"what": "the other BASELINE.json configurations, same step definition, the headline's "
                                            f"number of timed steps or {OTHER_MIN_SECONDS:g} s, whichever is longer "
                                            "(timed_steps / timed_seconds per record); ceilings: BASELINE.md §3 (pixel_cnn_pp: the "
                                            "same formula on bench._pixelcnnpp_gflop_per_img)", "unit": "images/s",
                                    **extras["other_configs"]}
        if env.world == 1:
            r = attention_kernel_roofline(args.batch, env.device, 4, 4, 4, 28, False)
            dom = "bwd" if "bwd" in r else "dkv"
            traffic, traffic_src, counters = measured_traffic(args.batch, "attn_bwd_m44_kernel" if dom == "bwd" else "attn_dkv_m44_kernel",
                                                              with_counters=True)
            out["roofline"] = {
                "bound": "mfma",  # fp32: matrix peak == vector peak == 157.3 TF on gfx950; the
                                  # kernel is 4x4x1-MFMA + v_exp issue bound
                "kernel": "attn_bwd_m44_kernel (pg_causal_attn_bwd: dQ, dK, dV fused)" if dom == "bwd"
                          else "attn_dkv_m44_kernel (pg_causal_attn_bwd_dkv)",
                "achieved": r[dom]["tflops"],
                "peak": FP32_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": r[dom]["tflops"] / FP32_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,  # a committed rocprofv3 --pmc pass of this kernel at this batch, not this run
                "launch_ms": r[dom]["launch_ms"],
                **_counter_fields(traffic, counters, r[dom]["launch_ms"]),
                "flop_per_launch": r[dom]["flop_per_launch"],
                "flop_accounting": "algorithmic: 6 d_k + 4 d_v = 40 FLOP per allowed (query, key) pair, every product once",
                "other_kernels": {"attn_fwd_m44_kernel": r["fwd"],
                                  "two_kernel_backward (pg_attn_fused_bwd(0) = ops.set_deterministic)": {"attn_dq_m44_kernel": r["dq"],
                                                                                "attn_dkv_m44_kernel": r["dkv"]}},
                # whole step; attention counted on the causal triangle, algorithmic (no recomputation)
                "step_tflops": head["step_tflops"],
                "step_frac": head["step_frac_of_fp32_peak"],
                "step_tflops_dense_attention_count": head["step_tflops_dense_attention_count"],
                "step_hbm_gbps_algorithmic": head["step_hbm_gbps_algorithmic"],
            }
            if "pixel_snail" in out:
                c = conv_kernel_roofline(args.snail_batch, env.device)
                a = attention_kernel_roofline(args.snail_batch, env.device, 1, 4, 32, 32, True)
                w = wgrad_kernel_roofline(args.snail_batch, env.device)
                b3_ceiling = BF16_PEAK_TFLOPS / 6.0  # six bf16 MFMAs per fp32 product
                # which kernel pg_conv2d_mfma routes this launch to: the production library always takes the default route (its
                # A/B switches are compiled out, csrc/common.h PG_AB_ENV); only a variant library (PG_HIP_LIB) reads the environment
                e = os.environ.get if os.environ.get("PG_HIP_LIB") else (lambda k, d=None: d)
                if e("PG_CONV_B3", "1")[:1] == "0":
                    ck, cdesc = "conv_mfma_kernel", "fp32 MFMA"
                elif e("PG_CONV_B3P", "1")[:1] == "0":
                    ck, cdesc = "conv_b3_kernel", "fp32 products as 6 bf16 MFMAs, 16-channel chunks"
                else:
                    waves = 4 if e("PG_CONV_B3P_WAVES") == "4" else 8
                    ck, cdesc = f"conv_b3p_kernel<2, {waves}>", f"fp32 products as 6 bf16 MFMAs, {waves} waves per workgroup"
                snail_traffic, snail_src, snail_ctr = measured_traffic(
                    args.snail_batch, ck.split("<")[0], ("r06_snail_conv_pmc.json", "r05_snail_conv_pmc.json", "r04_snail_conv_pmc.json"),
                    with_counters=True)
                out["pixel_snail"]["roofline"] = {
                    "bound": "mfma",
                    "kernel": f"{ck} (pg_conv2d_mfma: 2x2 64->64 convolution, ELU prologue; {cdesc})",
                    # algorithmic fp32 flops against the scheme's OWN ceiling: bf16 dense peak / 6 (the fp32 matrix peak,
                    # 157.3 TF, is not this kernel's ceiling: it does not run fp32 MFMAs)
                    "achieved": c["tflops"], "peak": b3_ceiling, "unit": "TFLOP/s",
                    "frac": c["tflops"] / b3_ceiling,
                    "frac_of_fp32_matrix_peak": c["tflops"] / FP32_PEAK_TFLOPS,
                    "launch_ms": c["launch_ms"],
                    **_counter_fields(snail_traffic, snail_ctr, c["launch_ms"]),
                    "flop_per_launch": c["flop_per_launch"],
                    # HBM bytes of the SAME launch (shape and batch) from separate rocprofv3 --pmc passes
                    # (profiles/r04_snail_conv_pmc.json: calibrated on an add kernel in the same process); algorithmic =
                    # x read once + out written once
                    "traffic": snail_traffic,
                    "traffic_source": snail_src,
                    "traffic_algorithmic": 2.0 * args.snail_batch * 64 * 32 * 32 * 4,
                    "other_kernels": {"conv_wgrad_b3r_kernel<4, 1, 2> (row ring) + wgrad_reduce_kernel": w,
                                      "attn_fwd_k4_kernel": a["fwd"],
                                      "attn_delta_k4 + attn_bwd_k4_kernel (fused backward)": a.get("bwd"),
                                      "two_kernel_backward (pg_attn_fused_bwd(0) = ops.set_deterministic)": {"attn_dq_k4_kernel": a["dq"],
                                                                                     "attn_dkv_k4_kernel": a["dkv"]}},
                }
            if not args.no_cpu_baseline:
                out["parity"] = measured_parity("image_gpt", args.batch, env.device)
                if "pixel_snail" in out:
                    out["pixel_snail"]["parity"] = measured_parity("pixel_snail", args.snail_batch, env.device)
                out["cpu_baseline"] = cpu_baseline()
                if "pixel_snail" in out:
                    out["pixel_snail"]["cpu_baseline"] = cpu_baseline_pixel_snail(out["cpu_baseline"]["cores"])
        print(json.dumps(out), flush=True)
    if env.world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
