"""The two-rank gloo data-parallel path of PixelSNAIL against a lone process, tensor by tensor (see profiles/README.md round 5 item 16).
    python tools/exp/two_proc_dp_divergence.py
Workers: fixed seed, PixelSNAIL (bench constructor), batch 32 (every rank the SAME batch), bit-reproducible kernels, 4 eager steps; recorded per
step: logits, every parameter gradient BEFORE the all-reduce, the flat gradient AFTER it (times 1/world), the flat parameters after Adam."""
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402


def worker(out, world, rank, port):
    import bench
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import ops, optim, parallel

    ops.set_deterministic(True)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    if world > 1:
        import torch.distributed as dist
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    w = bench.WORKLOADS[os.environ.get("WORKLOAD", "pixel_snail")]  # BCE workloads: image_gpt, pixel_cnn, gated_pixel_cnn, pixel_snail
    torch.manual_seed(0)
    model = getattr(pg.models, w["ctor"])(**w["kw"]).to(dev)
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=w["lr"], lr_decay=w["decay"])
    red = parallel.FlatGradAllReduce(opt) if world > 1 else None
    if red is not None and os.environ.get("DP_IDLE") == "1":  # a process group exists, but the loop below never uses it
        opt.set_grad_prescale(1.0)
        world, red = 1, None
    fake_host = None
    if world == 1 and os.environ.get("FAKE_STAGING") == "1":  # one process, no process group: only the pinned-host round trip of the gradient
        fake_host = torch.empty(opt.flat_grad.shape, dtype=opt.flat_grad.dtype, pin_memory=True)
    if red is not None:
        red.broadcast_parameters(src=0)
    x = bench.synthetic_batch(int(os.environ.get("BATCH", "32")), 0, w["chw"]).to(dev)
    rec, order = {}, []

    def put(k, t):
        rec[k] = t.detach().float().cpu().clone()
        order.append(k)

    nosync = os.environ.get("NOSYNC") == "1"  # as bench.py's eager step: nothing between backward, all-reduce and Adam
    keep = []
    for step in range(6):
        opt.zero_grad()
        logits = model(x)
        if not nosync:
            put(f"s{step}.logits", logits)
        loss = ops.bce_with_logits_sum_mean(logits, x)
        loss.backward()
        if not nosync:
            torch.cuda.synchronize()
            put(f"s{step}.flat_grad_local", opt.flat_grad)
        if nosync:  # stream-ordered device clones, no host synchronisation
            keep.append((f"s{step}.logits", logits.detach().clone()))
            keep.append((f"s{step}.flat_grad_local", opt.flat_grad.detach().clone()))
        if fake_host is not None:
            fake_host.copy_(opt.flat_grad)
            opt.flat_grad.copy_(fake_host, non_blocking=False)
        if red is not None:
            red.all_reduce()
            if nosync:
                keep.append((f"s{step}.flat_grad_mean", opt.flat_grad.detach() * (1.0 / world)))
            if not nosync:
                torch.cuda.synchronize()
                put(f"s{step}.flat_grad_mean", opt.flat_grad * (1.0 / world))
        elif not nosync:
            put(f"s{step}.flat_grad_mean", opt.flat_grad)
        else:
            keep.append((f"s{step}.flat_grad_mean", opt.flat_grad.detach().clone()))
        opt.step()
        if nosync:
            keep.append((f"s{step}.flat_param", opt.flat_param.detach().clone()))
        else:
            torch.cuda.synchronize()
            put(f"s{step}.flat_param", opt.flat_param)
    torch.cuda.synchronize()
    for k, t in keep:
        put(k, t)
    torch.save({"rec": rec, "order": order}, out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
        sys.exit(0)
    tmp = "/tmp/tpdp"
    os.makedirs(tmp, exist_ok=True)
    me = os.path.abspath(__file__)
    subprocess.run([sys.executable, me, "worker", f"{tmp}/ref.pt", "1", "0", "0"], check=True)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [subprocess.Popen([sys.executable, me, "worker", f"{tmp}/r{i}.pt", "2", str(i), str(port)]) for i in range(2)]
    for p in ps:
        assert p.wait() == 0
    ref = torch.load(f"{tmp}/ref.pt")
    for name in ("r0", "r1"):
        d = torch.load(f"{tmp}/{name}.pt")
        print(name)
        for k in ref["order"]:
            a, b = d["rec"][k], ref["rec"][k]
            if not torch.equal(a, b):
                diff = (a - b).abs()
                print(f"   {k}: {int((diff > 0).sum())} of {a.numel()} elements differ, max |diff| {float(diff.max()):.3e} (|ref| max {float(b.abs().max()):.3e})")
