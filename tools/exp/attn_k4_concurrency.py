"""Is attn_fwd_k4_kernel's output bit-stable while ANOTHER PROCESS keeps the same GPU busy? (round 5: bench.py --dp-parity with two
ranks on one GPU over gloo reproduces the 1-rank run bit for bit for every workload except PixelSNAIL, and routing only the
d_k = 4 / d_v = 32 FORWARD attention to the VALU kernels makes that one exact too.)
    python tools/exp/attn_k4_concurrency.py [iters]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "pytorch-generative_amd")]
import torch  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "load":  # the disturbing process: any kernels, for a while
    a = torch.randn(4096, 4096, device="cuda")
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(50):
            a = torch.tanh(a @ a * 1e-4)
        torch.cuda.synchronize()
    sys.exit(0)

from pytorch_generative_amd import ops  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
n, e, v, hw = 32, 4, 32, 32
q = torch.randn(n, e, hw, hw, generator=g).to(dev)
kv = torch.randn(n, e + v, hw, hw, generator=g).to(dev)


def run():
    with torch.no_grad():
        return ops.causal_attention(q, kv, 1, e, v, True).clone()


for phase in ("alone", "with another process on the GPU"):
    child = None
    if phase != "alone":
        child = subprocess.Popen([sys.executable, os.path.abspath(__file__), "load", "25"])
        time.sleep(6)
    ref = run()
    torch.cuda.synchronize()
    bad, worst = 0, 0.0
    for i in range(iters):
        out = run()
        if not torch.equal(out, ref):
            bad += 1
            worst = max(worst, float((out - ref).abs().max()))
    print(f"{phase}: {bad} of {iters} forward launches differ from the first (max |diff| {worst:.3e}; |o| max {float(ref.abs().max()):.3f})")
    if child is not None:
        child.wait()
