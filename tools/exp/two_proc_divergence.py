"""Where does a PixelSNAIL step computed by one of TWO processes sharing the GPU first differ (bitwise) from the same step computed by a
process that has the GPU to itself?  python tools/exp/two_proc_divergence.py
Each worker: fixed seed, PixelSNAIL (bench constructor), batch 32, bit-reproducible kernels, three eager steps; records the outputs of every
module of the first forward, the logits, and every parameter gradient of every step."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402


def worker(out):
    import bench
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import ops, optim

    ops.set_deterministic(True)
    dev = torch.device("cuda:0")
    w = bench.WORKLOADS["pixel_snail"]
    torch.manual_seed(0)
    model = getattr(pg.models, w["ctor"])(**w["kw"]).to(dev)
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=w["lr"], lr_decay=w["decay"])
    x = bench.synthetic_batch(32, 0, w["chw"]).to(dev)
    rec, order = {}, []

    def hook(name):
        def f(mod, inp, outp):
            t = outp[0] if isinstance(outp, (tuple, list)) else outp
            if torch.is_tensor(t) and name not in rec:
                rec[name] = t.detach().float().cpu().clone()
                order.append(name)
        return f

    hs = [m.register_forward_hook(hook(n)) for n, m in model.named_modules() if n]
    for step in range(3):
        opt.zero_grad()
        logits = model(x)
        if step == 0:
            for h in hs:
                h.remove()
        loss = ops.bce_with_logits_sum_mean(logits, x)
        loss.backward()
        for k, p in model.named_parameters():
            rec[f"step{step}.grad.{k}"] = p.grad.detach().cpu().clone()
            order.append(f"step{step}.grad.{k}")
        opt.step()
        torch.cuda.synchronize()
    torch.save({"rec": rec, "order": order}, out)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "worker":
        worker(sys.argv[2])
        sys.exit(0)
    tmp = "/tmp/tpd"
    os.makedirs(tmp, exist_ok=True)
    me = os.path.abspath(__file__)
    subprocess.run([sys.executable, me, "worker", f"{tmp}/ref.pt"], check=True)
    subprocess.run([sys.executable, me, "worker", f"{tmp}/ref2.pt"], check=True)
    ps = [subprocess.Popen([sys.executable, me, "worker", f"{tmp}/c{i}.pt"]) for i in range(2)]
    for p in ps:
        assert p.wait() == 0
    ref = torch.load(f"{tmp}/ref.pt")
    for name in ("ref2", "c0", "c1"):
        d = torch.load(f"{tmp}/{name}.pt")
        bad = [(k, float((d["rec"][k] - ref["rec"][k]).abs().max()), float(ref["rec"][k].abs().max())) for k in ref["order"]
               if not torch.equal(d["rec"][k], ref["rec"][k])]
        print(f"{name}: {len(bad)} of {len(ref['order'])} recorded tensors differ from the lone run" + (f"; first: {bad[:4]}" if bad else ""))
