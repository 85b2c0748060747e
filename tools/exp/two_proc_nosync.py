"""Two processes WITHOUT any process group, each running 6 fully asynchronous eager PixelSNAIL steps (bit-reproducible kernels) on the SAME
GPU at the same time, against one process that has the GPU to itself (profiles/README.md round 5 item 16).
    python tools/exp/two_proc_nosync.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

me = os.path.join(ROOT, "tools", "exp", "two_proc_dp_divergence.py")
tmp = "/tmp/tpn"
os.makedirs(tmp, exist_ok=True)
env = dict(os.environ, NOSYNC="1")
subprocess.run([sys.executable, me, "worker", f"{tmp}/ref.pt", "1", "0", "0"], check=True, env=env)
ps = [subprocess.Popen([sys.executable, me, "worker", f"{tmp}/c{i}.pt", "1", "0", "0"], env=env) for i in range(2)]
for p in ps:
    assert p.wait() == 0
ref = torch.load(f"{tmp}/ref.pt")
for name in ("c0", "c1"):
    d = torch.load(f"{tmp}/{name}.pt")
    bad = [k for k in ref["order"] if not torch.equal(d["rec"][k], ref["rec"][k])]
    first = bad[0] if bad else None
    extra = ""
    if first:
        a, b = d["rec"][first], ref["rec"][first]
        extra = f"; first: {first}: {int((a != b).sum())} of {a.numel()} elements, max |diff| {float((a - b).abs().max()):.3e} (|ref| max {float(b.abs().max()):.3e})"
    print(f"{name}: {len(bad)} of {len(ref['order'])} recorded tensors differ from the lone run{extra}")
