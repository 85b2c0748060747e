"""Which module's output changes when two processes share the GPU? (profiles/README.md round 5 item 16)
Each worker builds PixelSNAIL (bench constructor, fixed seed, bit-reproducible kernels), runs R forwards of the SAME batch with the SAME
parameters without any host synchronisation, and keeps a stream-ordered clone of EVERY module's output. Afterwards every repeat is compared
with repeat 0 of the same process, module by module in execution order. One worker alone, then two at the same time.
    python tools/exp/conc_forward_selfcheck.py [workload] [repeats]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402


def worker(tag, name, repeats):
    import bench
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import ops

    ops.set_deterministic(True)
    dev = torch.device("cuda:0")
    w = bench.WORKLOADS[name]
    torch.manual_seed(0)
    model = getattr(pg.models, w["ctor"])(**w["kw"]).to(dev)
    model.train()
    x = bench.synthetic_batch(32, 0, w["chw"]).to(dev)
    runs, cur = [], []

    def hook(label):
        def fn(mod, inp, out):
            if torch.is_tensor(out):
                cur.append((label, type(mod).__name__, out.detach().clone()))
            elif isinstance(out, (tuple, list)):
                for i, o in enumerate(out):
                    if torch.is_tensor(o):
                        cur.append((f"{label}[{i}]", type(mod).__name__, o.detach().clone()))
        return fn

    for label, mod in model.named_modules():
        mod.register_forward_hook(hook(label or "<model>"))

    def traced(fname):  # the functional steps inside CausalAttention.forward (no module of their own)
        inner = getattr(ops, fname)

        def fn(*a, **k):
            out = inner(*a, **k)
            for i, o in enumerate(out if isinstance(out, (tuple, list)) else [out]):
                if torch.is_tensor(o):
                    cur.append((f"ops.{fname}[{i}]", "function", o.detach().clone()))
            return out
        setattr(ops, fname, fn)

    for fname in ("concat_channels", "merge_qkv_weight", "conv2d_taps", "conv2d_pair", "causal_attention_qkv", "causal_attention"):
        traced(fname)
    if os.environ.get("VA_SHIFT"):  # move this process's later allocations to other virtual addresses than its twin's
        shift = torch.empty(int(os.environ["VA_SHIFT"]), dtype=torch.uint8, device=dev)  # noqa: F841
    with torch.no_grad() if os.environ.get("NOGRAD") == "1" else torch.enable_grad():
        for _ in range(repeats):
            cur = []
            model(x)
            runs.append(cur)
    torch.cuda.synchronize()
    bad_runs = 0
    for r in range(1, repeats):
        diffs = []
        for (label, kind, a), (_, _, b) in zip(runs[r], runs[0]):
            if not torch.equal(a, b):
                d = (a - b).abs()
                diffs.append(f"{label} ({kind} {tuple(a.shape)}): {int((d > 0).sum())} of {a.numel()} differ, max {float(d.max()):.3e} of {float(b.abs().max()):.3e}")
        if diffs:
            bad_runs += 1
            print(f"[{tag}] repeat {r}: {len(diffs)} of {len(runs[0])} module outputs differ; first three in execution order:")
            for s in diffs[:3]:
                print(f"[{tag}]      {s}")
    print(f"[{tag}] {bad_runs} of {repeats - 1} repeats differ from repeat 0")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        worker(sys.argv[2], sys.argv[3], int(sys.argv[4]))
        sys.exit(0)
    name = sys.argv[1] if len(sys.argv) > 1 else "pixel_snail"
    repeats = sys.argv[2] if len(sys.argv) > 2 else "10"
    me = os.path.abspath(__file__)
    subprocess.run([sys.executable, me, "worker", "alone", name, repeats], check=True)
    mode = os.environ.get("MODE", "twins")
    if mode == "twins":  # two identical processes
        ps = [subprocess.Popen([sys.executable, me, "worker", f"shared{i}", name, repeats]) for i in range(2)]
    elif mode == "shifted":  # the same two, the second one with its allocations at other virtual addresses
        ps = [subprocess.Popen([sys.executable, me, "worker", f"shifted{i}", name, repeats],
                               env=dict(os.environ, **({"VA_SHIFT": str((1 << 30) + 12345 * 512)} if i else {}))) for i in range(2)]
    else:  # one model process next to a process that runs matrix products (tools/exp/attn_k4_concurrency.py load)
        ps = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "exp", "attn_k4_concurrency.py"), "load", "25"]),
              subprocess.Popen([sys.executable, me, "worker", "next-to-matmul", name, repeats])]
    for p in ps:
        assert p.wait() == 0
