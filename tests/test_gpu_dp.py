"""GPU: data-parallel semantics of the real step objects (SURVEY.md §8e "Parity check"), with two
processes sharing the box's one MI355X over gloo:
  * every rank fed the SAME batch      == the 1-GPU step on that batch;
  * ranks fed DISTINCT halves of a batch == the 1-GPU step on the whole batch
    (loss = mean of the per-rank means, gradient = mean of the per-rank gradients).
Reference semantics: DistributedDataParallel averaging, trainer.py:78-82."""

import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def _single_gpu_reference(kind="igpt", deterministic=False):
    import dp_worker
    from pytorch_generative_amd import graph, ops

    from pytorch_generative_amd.models.vae import vaes

    dev = torch.device("cuda", 0)
    model, opt = dp_worker.build(dev, seed=0, kind=kind)
    loss_fn, fwd = dp_worker.step_functions(kind, model)
    was = ops.set_deterministic(deterministic)
    try:
        data = dp_worker.batches(kind=kind)
        step = graph.GraphedTrainStep(model, opt, loss_fn, data[0].to(dev), preserve_state=True, forward_fn=fwd)
        losses = [float(step(b.to(dev))) for b in data]
        torch.cuda.synchronize()
    finally:
        vaes.set_noise_fn(None)
        ops.set_deterministic(was)
    return {k: v.detach().cpu() for k, v in model.named_parameters()}, losses, opt.current_lr()


def _run_world2(mode, out, kind="igpt"):
    port = 29600 + os.getpid() % 300 + (0 if mode == "same" else 1) + 2 * ["igpt", "gated", "snail", "vd_vae"].index(kind)
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), mode, out, kind],
                              env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    logs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return torch.load(out, map_location="cpu", weights_only=False)


# snail = BASELINE.json configs[3], north_star's data-parallel workload (pixel_snail.py:122-187) and the one whose two-rank run
# exposed round 5's LDS race; vd_vae = configs[4] (vd_vae.py:141-189) with its noise replayed identically on every rank
@pytest.mark.parametrize("mode,kind", [("same", "igpt"), ("shard", "igpt"), ("shard", "gated"), ("same", "snail"),
                                       ("shard", "snail"), ("same", "vd_vae")])
def test_two_ranks_equal_one_gpu(tmp_path, mode, kind):
    want, want_losses, want_lr = _single_gpu_reference(kind, deterministic=mode == "same")
    got = _run_world2(mode, str(tmp_path / f"dp_{mode}.pt"), kind)
    if kind == "gated":  # parameters that never receive a gradient stay exactly where the broadcast put them
        import dp_worker

        m0, _ = dp_worker.build(torch.device("cuda", 0), seed=0, kind="gated")
        never = [k for k, p in m0.named_parameters() if "_gated_layers.1._vstack_1x1" in k
                 or "_gated_layers.1._link" in k or "_gated_layers.1._hstack_residual" in k]
        assert never
        init = dict(m0.named_parameters())
        frozen = [k for k in never if torch.equal(got["params"][k], init[k].detach().cpu())]
        assert frozen, "no never-gradient parameter found unchanged"
    assert got["step"] == 3.0 and abs(got["lr"] - want_lr) < 1e-12
    for k, w in want.items():
        g = got["params"][k]
        if k.endswith("_kv.bias"):
            # the key half of this bias has a true gradient of 0 (softmax is shift invariant): Adam turns
            # its round-off noise into +-lr steps whose sign depends on summation order — in the
            # reference too (see test_golden_step_flat_adam); only bounded by lr per step here
            assert float((g - w).abs().max()) <= 3 * 2 * 5e-3, k
            continue
        d, pmax = float((g - w).abs().max()), float(w.abs().max().clamp_min(1e-12))
        # "same" (bit-reproducible kernels on both sides): equal up to the rounding of the flat all-reduce; "shard": the two
        # half-batch gradients are summed in another order than the 1-GPU kernels sum the whole batch (~1e-6 of a tensor's
        # max), and Adam's normalised update turns a RELATIVE change e of a gradient entry into e * lr per step — small
        # entries (1e-3 of the max) therefore move by a few 1e-3 * lr per step, independent of the parameter's own scale
        tol = 2e-6 * pmax if mode == "same" else 5e-5 * pmax + 3e-3 * 3 * 5e-3
        assert d <= tol, f"{k} after 3 data-parallel steps: |diff| {d:.2e} > {tol:.2e} (max |param| {pmax:.2e})"
    for i, w in enumerate(want_losses):
        per_rank = [l[i] for l in got["losses"]]
        mean = sum(per_rank) / len(per_rank)  # global loss = mean of the per-rank means
        assert abs(mean - w) <= 1e-5 * abs(w), (mode, i, per_rank, w)
        if mode == "same":
            assert max(per_rank) - min(per_rank) <= 1e-5 * abs(w)  # fp32 atomic order of the loss sum


def test_rccl_world_of_one_captured_in_the_step_graph(tmp_path):
    """The production transport on the one GPU this box has: a real RCCL communicator (world size 1)
    created through the C-ABI, its all-reduce captured inside the step's hipGraph next to the live
    communicator, replayed, destroyed (tests/dp_worker.py rccl_world1)."""
    out = str(tmp_path / "rccl1.pt")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "dp_worker.py"), "rccl1", out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    got = torch.load(out, map_location="cpu", weights_only=False)
    assert got["rccl_version"] >= 20000
    assert got["same"], "the step with the captured world-1 all-reduce differs from the plain step"
    for a, b in zip(got["losses"], got["losses_plain"]):
        assert abs(a - b) <= 2e-6 * abs(b)


def test_comm_failure_paths(tmp_path):
    """pg_comm_* error behaviour (tests/dp_worker.py commfail; its own process: the communicator is per-process
    state): argument errors raise ValueError, a refused id RuntimeError, and a failed init leaves no communicator."""
    out = str(tmp_path / "commfail.pt")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "dp_worker.py"), "commfail", out], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    assert os.path.exists(out)


def test_bench_py_spawns_its_own_ranks():
    """`python bench.py --gpus 2` as ONE bare command (no torch.distributed.run, no WORLD_SIZE): bench.py starts
    a process per rank itself and rank 0 prints one JSON line with n_gpus == 2. On this 1-GPU box both ranks are
    pinned to device 0 and talk over gloo (RCCL refuses two ranks on one device); with 2+ GPUs the same command
    runs rank r on device r over RCCL."""
    import json

    root = os.path.dirname(HERE)
    env = dict(os.environ, PG_FORCE_DEVICE="0", PG_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--batch", "64", "--no-extras", "--no-cpu-baseline"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-4000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 5 and rec["config"]["global_batch"] == 128
    assert rec["value"] > 0 and rec["scaling"] == "weak"
    assert "all-reduce of the flat gradient" in rec["grad_exchange"]
    lo, hi = rec["per_rank_images_per_s"]["min"], rec["per_rank_images_per_s"]["max"]
    assert 0 < lo <= hi and rec["value"] <= 2 * hi * 1.001


def test_bench_py_dp_parity_mode():
    """`python bench.py --gpus 2 --dp-parity`: every rank on rank 0's batch; after the run each rank's parameters equal
    rank 0's AND those of a 1-rank run of the same steps, bit for bit (world of two: the sum of two equal gradients and
    the 1/2 pre-scale are exact) — SURVEY.md section 8(e)'s equality on whatever transport the job uses (gloo here, RCCL
    on a multi-GPU node), printed in the one JSON line next to the rate."""
    import json

    root = os.path.dirname(HERE)
    env = dict(os.environ, PG_FORCE_DEVICE="0", PG_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--batch", "16", "--dp-parity"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode(errors="replace")[-4000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    rec = json.loads(lines[0])
    d = rec["dp_parity"]
    assert rec["n_gpus"] == 2 and d["ok"] and d["exact_expected"]
    assert d["max_abs_diff_vs_rank0"] == 0.0 and d["max_abs_diff_vs_one_rank_run"] == 0.0
    assert d["params_sha256_16"]["rank0_of_the_job"] == d["params_sha256_16"]["one_rank_run"]
    assert d["param_abs_max"] > 0 and rec["one_rank_run"]["images_per_s"] > 0
    assert 0 < rec["scaling_efficiency_vs_n1"] < 1.5
    # the second equality of SURVEY.md section 8(e): distinct shards == one replica on the concatenated batch
    sh = d["shards"]
    assert sh["ok"] and sh["fraction_of_parameters_beyond_tolerance"] <= 0.01
    assert 0 <= sh["max_abs_diff_vs_one_rank_run_on_the_concatenated_batch"] <= sh["adam_step_bound"] * 1.001
    assert len(rec["launch_by_rank"]) == 2 and len(set(rec["launch_by_rank"])) == 1, rec["launch_by_rank"]


def test_train_py_two_workers_through_trainer(tmp_path):
    """`train.py --gpus 2` end to end on one GPU (PG_FORCE_DEVICE=0, gloo): the spawned workers run the
    model module's reproduce() -> recipes.run -> Trainer(n_gpus=2) -> GraphedTrainStep + FlatGradAllReduce.
    Ranks draw DIFFERENT batches (loader seeds offset by the rank) and must end with IDENTICAL
    parameters; only rank 0 writes checkpoints."""
    root = os.path.dirname(HERE)
    dump = tmp_path / "dump"
    dump.mkdir()
    logdir = tmp_path / "log"
    env = dict(os.environ, PG_FORCE_DEVICE="0", PG_DIST_BACKEND="gloo", PG_TRAIN_DUMP=str(dump),
               PYTHONPATH=os.path.join(root, "pytorch-generative_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("WORLD_SIZE", None)
    port = 29900 + os.getpid() % 90
    p = subprocess.run([sys.executable, os.path.join(root, "pytorch-generative_amd", "train.py"), "--model",
                        "gated_pixel_cnn", "--gpus", "2", "--epochs", "1", "--batch-size", "2", "--logdir",
                        str(logdir), "--port", str(port)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert p.returncode == 0, p.stdout.decode(errors="replace")[-4000:]
    r0 = torch.load(dump / "rank0.pt", map_location="cpu", weights_only=False)
    r1 = torch.load(dump / "rank1.pt", map_location="cpu", weights_only=False)
    assert r0["step"] == r1["step"] and r0["step"] > 0
    assert not torch.equal(r0["first_batch"], r1["first_batch"]), "both ranks drew the same data"
    for k, v in r0["params"].items():
        assert torch.equal(v, r1["params"][k]), f"ranks diverged on {k}"
        assert torch.isfinite(v).all()
    assert sorted(os.listdir(logdir)) == ["trainer_state_1.ckpt"] or "trainer_state_1.ckpt" in os.listdir(logdir)
