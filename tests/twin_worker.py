"""One of two processes that share the box's MI355X (spawned by tests/test_gpu_shared_device.py).

Every kernel of the bit-reproducible set (ops.set_deterministic) is a pure function of its inputs, so two runs of the same
forward + backward on the same batch with the same parameters must agree bit for bit — ANY difference between repeats is a
race inside a kernel (round 5: `conv_mfma_kernel` zero-filled LDS and committed without a barrier in between; alone on the GPU
the fill always won, next to a twin process a late wave zeroed committed entries in 2-4 of 15 forwards; 275 single-process
tests never saw it, profiles/README.md round 5 item 16).

For every workload of bench.WORKLOADS at the bench's exact constructor this worker runs R repeats of
zero_grad -> forward -> loss -> backward WITHOUT any host synchronisation and compares, stream-ordered on the device,
every module output of the forward and the whole flat gradient buffer with repeat 0 of the SAME process. The two twins meet at a
file rendezvous before each workload so that they run the same kernels side by side.

    python tests/twin_worker.py <tag> <rendezvous dir> <n twins> <repeats> <workload>[:batch] ...
prints one line per workload:  [<tag>] <workload>: <bad> of <R-1> repeats differ (<n> tensors compared per repeat)
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

# per-process batch of each workload: a few workgroups per CU for the main kernels, small enough that two processes with every
# module output of one repeat retained fit comfortably
BATCH = {"image_gpt": 64, "image_gpt_repro": 16, "pixel_snail": 32, "gated_pixel_cnn": 16, "pixel_cnn": 64,
         "pixel_cnn_pp": 4, "beta_vae": 32, "vd_vae": 8}


def rendezvous(folder, tag, key, n, timeout=300.0):
    open(os.path.join(folder, f"{key}.{tag}"), "w").close()
    t0 = time.monotonic()
    while sum(1 for f in os.listdir(folder) if f.startswith(key + ".")) < n:
        if time.monotonic() - t0 > timeout:
            raise SystemExit(f"[{tag}] rendezvous {key} timed out")
        time.sleep(0.005)


def run_workload(tag, name, batch, repeats, dev):
    import bench
    import dp_worker
    import pytorch_generative_amd as pg
    from pytorch_generative_amd import optim
    from pytorch_generative_amd.models.vae import vaes

    w = bench.WORKLOADS[name]
    torch.manual_seed(0)
    model = getattr(pg.models, w["ctor"])(**w["kw"]).to(dev)
    model.train()
    opt = optim.FlatAdam(model.parameters(), lr=w["lr"])  # its flat gradient buffer is what the backward kernels add into
    x = bench.workload_input(name, bench.synthetic_batch(batch, 0, w["chw"])).to(dev)
    loss_fn = bench.make_loss_fn(name)
    noise = dp_worker.FixedNoise() if name in ("beta_vae", "vd_vae") else None
    vaes.set_noise_fn(noise)
    ref, labels, mism, state = [], [], [], {"i": 0, "first": True}

    def seen(label, t):
        t = t.detach()
        if state["first"]:
            ref.append(t.clone())
            labels.append(f"{label} {tuple(t.shape)}")
        else:
            mism[-1].append((t != ref[state["i"]]).sum())  # stays on the device: no host synchronisation
        state["i"] += 1

    def hook(label):
        def fn(mod, inp, out):
            if torch.is_tensor(out):
                seen(label, out)
            elif isinstance(out, (tuple, list)):
                for j, o in enumerate(out):
                    if torch.is_tensor(o):
                        seen(f"{label}[{j}]", o)
        return fn

    for label, mod in model.named_modules():
        mod.register_forward_hook(hook(label or "<model>"))
    for r in range(repeats):
        state["i"], state["first"] = 0, r == 0
        if r:
            mism.append([])
        if noise is not None:
            noise.reset()
        opt.zero_grad()
        loss = loss_fn(x, model(x))
        loss.backward()
        seen("<flat gradient>", opt.flat_grad)
    torch.cuda.synchronize()
    vaes.set_noise_fn(None)
    bad = 0
    for r, counts in enumerate(mism, start=1):
        assert len(counts) == len(ref), (len(counts), len(ref))
        wrong = [(labels[i], int(c)) for i, c in enumerate(counts) if int(c)]
        if wrong:
            bad += 1
            print(f"[{tag}] {name} repeat {r}: {len(wrong)} of {len(ref)} tensors differ from repeat 0; first three in "
                  f"execution order: " + "; ".join(f"{lb}: {c} elements" for lb, c in wrong[:3]), flush=True)
    assert float(opt.flat_grad.abs().max()) > 0 and bool(torch.isfinite(opt.flat_grad).all()), "degenerate gradient"
    print(f"[{tag}] {name}: {bad} of {repeats - 1} repeats differ ({len(ref)} tensors compared per repeat)", flush=True)


def main():
    tag, folder, n, repeats = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    from pytorch_generative_amd import ops

    ops.set_deterministic(True)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    for spec in sys.argv[5:]:
        name, _, b = spec.partition(":")
        rendezvous(folder, tag, name, n)
        run_workload(tag, name, int(b) if b else BATCH[name], repeats, dev)
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
